// capi.cpp -- a few C entry points over the C++ host so that tests can drive the loaders and the
// integrators from Python: load a scene file and read back the flattened arrays, render a scene
// file through RandomWalk/Albedo/Normal integrators, write a BMP.
#include <cstring>
#include <string>

#include "flatten.hpp"
#include "raytracer.hpp"

using namespace raytracer;

namespace {
thread_local std::string g_err;
struct Loaded { FlatScene flat; rtmi_camera cam; Scene scene; Camera camera; Loaded(Scene s, Camera c) : scene(std::move(s)), camera(c) {} };
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// fan: 0 = last-vertex fan (default), 1 = first-vertex fan
void* rth_scene_load(const char* toml_path, int fan) {
  try {
    const toml::Table t = parse_file(toml_path);
    LoadOptions opt;
    opt.fan = fan ? FanOrigin::First : FanOrigin::Last;
    const std::string p(toml_path);
    const auto slash = p.find_last_of('/');
    if (slash != std::string::npos) opt.base_dir = p.substr(0, slash);
    Camera cam = parse_camera(t);
    Scene scene = parse_scene(t, opt);
    auto* l = new Loaded(std::move(scene), cam);
    l->flat = flatten(l->scene.shapes());
    l->cam = l->camera.raw();
    return l;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

void rth_scene_free(void* h) { delete static_cast<Loaded*>(h); }

void rth_scene_arrays(void* h, const rtmi_object** objects, uint32_t* n_objects, const rtmi_sphere** spheres, uint32_t* n_spheres,
                      const rtmi_triangle** triangles, uint32_t* n_triangles, const rtmi_material** materials,
                      uint32_t* n_materials, rtmi_camera* cam) {
  Loaded* l = static_cast<Loaded*>(h);
  *objects = l->flat.objects.data(); *n_objects = uint32_t(l->flat.objects.size());
  *spheres = l->flat.spheres.data(); *n_spheres = uint32_t(l->flat.spheres.size());
  *triangles = l->flat.triangles.data(); *n_triangles = uint32_t(l->flat.triangles.size());
  *materials = l->flat.materials.data(); *n_materials = uint32_t(l->flat.materials.size());
  *cam = l->cam;
}

// integrator: RTMI_INT_*; rgb8 must hold width*height*3 bytes of the loaded camera. Returns 0 on success.
int rth_scene_render(void* h, int integrator, uint32_t depth, uint32_t spp, uint64_t seed, uint32_t hit_rule, uint8_t* rgb8,
                     rtmi_stats* stats) {
  try {
    Loaded* l = static_cast<Loaded*>(h);
    l->scene.set_hit_rule(hit_rule);
    Image img;
    auto go = [&](SamplerIntegrator&& it) { it.seed = seed; img = it.render(); if (stats) *stats = it.last_stats; };
    if (integrator == RTMI_INT_RANDOM_WALK) go(RandomWalkIntegrator(l->camera, l->scene, depth, spp));
    else if (integrator == RTMI_INT_ALBEDO) go(AlbedoIntegrator(l->camera, l->scene, spp));
    else go(NormalIntegrator(l->camera, l->scene, spp));
    std::memcpy(rgb8, img.rgb.data(), img.rgb.size());
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

int rth_save_bmp(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
  try {
    Image img;
    img.width = width; img.height = height;
    img.rgb.assign(rgb8, rgb8 + size_t(width) * height * 3);
    img.save(path);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
