// raytracer.hpp -- C++ host side above the C ABI (include/rtmi.h).
//
// The reference's host is Rust; no Rust toolchain exists where this is built, so the host layer is
// restated in C++ with the reference's own type and method names, argument meaning and error
// behaviour. It owns scene description and I/O (TOML / OBJ / MTL loading, BMP output, CLI) and
// hands every per-pixel computation to librtmi.so. Citations are file:line in the reference.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/rtmi.h"

namespace raytracer {

// anyhow::Error of the reference: a message that bubbles up to main (src/main.rs:51-53)
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

struct Point { double x = 0, y = 0, z = 0; };
using Vector = Point;
using Normal = Point;

// Affine3<f64> (src/geometry/common.rs:8): row-major 3x4
struct Transform {
  std::array<double, 12> m{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  static Transform identity() { return Transform{}; }
  Transform operator*(const Transform& rhs) const;
  Point operator*(const Point& p) const;      // as a point
  Vector apply_vector(const Vector& v) const; // as a vector
};
Transform create_translation(const Vector& offset);              // src/geometry/common.rs:14
Transform create_rotation(const Vector& axis, double angle);     // src/geometry/common.rs:18
Transform create_scaling(const Vector& scale);                   // src/geometry/common.rs:22

// Ray (src/geometry/ray.rs:5-36)
struct Ray {
  Point origin; Vector dir; double max_t;
  Ray(const Point& origin, const Vector& dir, double max_t);     // normalises dir
  std::optional<Point> at(double t) const;
};

enum class SpectrumType { Reflectance, Illuminant };             // src/spectrum/mod.rs:8-11

// SampledSpectrum = CoefficientSpectrum<60> (src/spectrum/mod.rs:14,264-265)
struct SampledSpectrum {
  std::array<float, RTMI_NBINS> coefficients{};
  static SampledSpectrum constant(float v);
  static SampledSpectrum from_RGB(float r, float g, float b, SpectrumType type);  // :525-581
  std::array<float, 3> to_XYZ() const;                                            // :590-609
  std::array<float, 3> to_RGB() const;                                            // :624-626
};

// src/material.rs:5-9, plus two extension types that finally read Ns / Ni (no counterpart in the reference)
enum class Type { Emitter, Lambertian, Specular, Glossy, Dielectric };

struct Material {                                                // src/material.rs:12-37
  Type tipo = Type::Emitter;
  std::optional<SampledSpectrum> ambient_color, emitted_color, diffused_color, specular_color;
  std::optional<double> specular_coefficient, optical_density;
};

// One parsed `newmtl` block (the fields wavefront_obj::mtl::Material carries)
struct MtlMaterial {
  std::string name;
  double specular_coefficient = 0.0;
  std::array<double, 3> color_ambient{0, 0, 0}, color_diffuse{0, 0, 0}, color_specular{0, 0, 0};
  std::optional<std::array<double, 3>> color_emissive;
  std::optional<double> optical_density;
};
Material material_from_mtl(const MtlMaterial& m);                // From<&mtl::Material>, src/material.rs:39-53
std::vector<MtlMaterial> parse_mtl(const std::string& text);

struct Sphere {                                                  // src/shapes/sphere.rs:8-34
  Transform local_to_world; double radio; Material material;
  Sphere(const Transform& t, double radio, const Material& m) : local_to_world(t), radio(radio), material(m) {}
};

struct Triangle {                                                // src/shapes/triangle.rs:9-32
  std::array<Point, 3> vertices; Transform local_to_world; Material material;
  Triangle(const Point& p1, const Point& p2, const Point& p3, const Transform& t, const Material& m);
};

// How polygons of an OBJ file become triangles. No reference test pins the third-party crate's
// order (parity unpinned); Last is what wavefront_obj 10.0 is believed to emit.
enum class FanOrigin { Last, First };

struct ModelObj {                                                // src/shapes/model_obj.rs:8-73
  std::vector<std::array<Point, 3>> triangles;                   // file order, positions only
  Material material;
  explicit ModelObj(const std::string& file, FanOrigin fan = FanOrigin::Last, const std::string& base_dir = "");
  ModelObj() = default;
};

using Shape = std::variant<Sphere, Triangle, ModelObj>;          // src/shapes/shape.rs:34-39

struct Intersection {                                            // src/shapes/common.rs:9-17, ids instead of a shape copy
  std::size_t object; std::uint32_t primitive; Point point; Normal normal; double t;
};

class Camera {                                                   // src/camera.rs
 public:
  Camera(const Point& focus, double focal_distance, double field_of_view, std::array<double, 3> rotation,
         std::array<std::uint32_t, 2> resolution);
  std::uint32_t width() const { return cam_.width; }
  std::uint32_t height() const { return cam_.height; }
  Ray get_ray(double i, double j) const;
  const rtmi_camera& raw() const { return cam_; }
 private:
  rtmi_camera cam_{};
};

struct GpuScene;  // the flattened copy resident on the GPU

class Scene {                                                    // src/scene.rs
 public:
  Scene();
  ~Scene();
  Scene(const Scene& other);
  Scene& operator=(const Scene& other);
  void add_shape(const Shape& s);
  std::size_t len() const { return objetos_.size(); }
  const std::vector<Shape>& shapes() const { return objetos_; }
  // Scene::intersect_ray (src/scene.rs:135-150); the batch form is the efficient one on a GPU
  std::optional<Intersection> intersect_ray(const Ray& ray) const;
  std::vector<std::optional<Intersection>> intersect_rays(const std::vector<Ray>& rays) const;
  // 0 = first triangle in file order inside a mesh (the reference), 1 = nearest
  void set_hit_rule(std::uint32_t rule);
  rtmi_scene* gpu() const;  // flattens and uploads on first use
 private:
  std::vector<Shape> objetos_;
  std::uint32_t hit_rule_ = RTMI_HIT_FIRST_INDEX;
  mutable std::shared_ptr<GpuScene> gpu_;
};

struct Image {                                                   // ImageBuffer<Rgb<u8>> (src/integrators/auxiliar.rs:4)
  std::uint32_t width = 0, height = 0;
  std::vector<std::uint8_t> rgb;  // row-major, y = 0 top
  void save(const std::string& path) const;  // 24-bit BMP, like image.save("x.bmp") (src/main.rs:79)
};

class SamplerIntegrator {                                        // src/integrators/integrator.rs:17-143
 public:
  virtual ~SamplerIntegrator() = default;
  const Camera& camera() const { return camera_; }
  const Scene& scene() const { return scene_; }
  virtual std::size_t max_depth() const = 0;
  std::size_t total_samples() const { return iterations_; }
  Image render() const;
  std::uint64_t seed = 0;            // Philox key; the reference's RNG cannot be seeded
  mutable rtmi_stats last_stats{};
 protected:
  SamplerIntegrator(const Camera& c, const Scene& s, std::size_t iterations, std::uint32_t kind)
      : camera_(c), scene_(s), iterations_(iterations), kind_(kind) {}
  Camera camera_; Scene scene_; std::size_t iterations_; std::uint32_t kind_;
};

class RandomWalkIntegrator : public SamplerIntegrator {          // src/integrators/random_walk.rs
 public:
  RandomWalkIntegrator(const Camera& c, const Scene& s, std::size_t depth, std::size_t iterations)
      : SamplerIntegrator(c, s, iterations, RTMI_INT_RANDOM_WALK), depth_(depth) {}
  std::size_t max_depth() const override { return depth_; }
 private:
  std::size_t depth_;
};
class AlbedoIntegrator : public SamplerIntegrator {              // src/integrators/albedo.rs
 public:
  AlbedoIntegrator(const Camera& c, const Scene& s, std::size_t iterations) : SamplerIntegrator(c, s, iterations, RTMI_INT_ALBEDO) {}
  std::size_t max_depth() const override { return 0; }
};
class NormalIntegrator : public SamplerIntegrator {              // src/integrators/normal.rs
 public:
  NormalIntegrator(const Camera& c, const Scene& s, std::size_t iterations) : SamplerIntegrator(c, s, iterations, RTMI_INT_NORMAL) {}
  std::size_t max_depth() const override { return 0; }
};

// ---- scene_config (src/scene_config.rs) ------------------------------------------------------
namespace toml {
struct Value;
using Array = std::vector<Value>;
struct Table { std::vector<std::pair<std::string, Value>> items; const Value* get(const std::string& key) const; };
struct Value {
  std::variant<std::monostate, std::int64_t, double, std::string, bool, Array, Table> v;
  const std::int64_t* as_integer() const { return std::get_if<std::int64_t>(&v); }
  const double* as_float() const { return std::get_if<double>(&v); }  // an integer is NOT a float, as in the toml crate
  const std::string* as_str() const { return std::get_if<std::string>(&v); }
  const Array* as_array() const { return std::get_if<Array>(&v); }
  const Table* as_table() const { return std::get_if<Table>(&v); }
};
Table parse(const std::string& text);  // the subset the scene files use
}  // namespace toml

struct LoadOptions { FanOrigin fan = FanOrigin::Last; std::string base_dir; };
toml::Table parse_file(const std::string& path);                                   // src/scene_config.rs:11
Camera parse_camera(const toml::Table& table);                                     // :15
Scene parse_scene(const toml::Table& table, const LoadOptions& opt = {});          // :75
std::string read_file(const std::string& name);                                    // src/auxiliar.rs:17

// process-wide GPU context (device 0 unless RTMI_DEVICE is set)
rtmi_ctx* context();

}  // namespace raytracer
