// flatten.hpp -- the host Scene as the flat arrays rtmi_scene_create takes (no GPU involved).
#pragma once
#include <vector>

#include "raytracer.hpp"

namespace raytracer {

struct FlatScene {
  std::vector<rtmi_object> objects;
  std::vector<rtmi_sphere> spheres;
  std::vector<rtmi_triangle> triangles;
  std::vector<rtmi_material> materials;
  rtmi_scene_desc desc(std::uint32_t hit_rule) const {
    rtmi_scene_desc d{};
    d.objects = objects.data(); d.n_objects = std::uint32_t(objects.size());
    d.spheres = spheres.data(); d.n_spheres = std::uint32_t(spheres.size());
    d.triangles = triangles.data(); d.n_triangles = std::uint32_t(triangles.size());
    d.materials = materials.data(); d.n_materials = std::uint32_t(materials.size());
    d.hit_rule = hit_rule;
    return d;
  }
};

// One rtmi_object per shape, in scene order; a mesh contributes its triangles in file order.
FlatScene flatten(const std::vector<Shape>& shapes);

}  // namespace raytracer
