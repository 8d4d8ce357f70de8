// raytracer.cpp -- geometry, spectra, materials, camera, scene flattening, integrators, BMP output.
#include "raytracer.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <mutex>

#include "flatten.hpp"

namespace raytracer {

namespace {
void check(int rc, const char* what) {
  if (rc != RTMI_OK) throw Error(std::string(what) + ": " + rtmi_last_error());
}
}  // namespace

// ---- transforms (src/geometry/common.rs) ----------------------------------------------------

Transform Transform::operator*(const Transform& r) const {
  Transform o;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = (m[i * 4] * r.m[j] + m[i * 4 + 1] * r.m[4 + j]) + m[i * 4 + 2] * r.m[8 + j];
      if (j == 3) s += m[i * 4 + 3];
      o.m[i * 4 + j] = s;
    }
  return o;
}
Vector Transform::apply_vector(const Vector& v) const {
  return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}
Point Transform::operator*(const Point& p) const {
  Vector r = apply_vector(p);
  return {r.x + m[3], r.y + m[7], r.z + m[11]};
}
Transform create_translation(const Vector& o) { Transform t; t.m[3] = o.x; t.m[7] = o.y; t.m[11] = o.z; return t; }
Transform create_scaling(const Vector& s) { Transform t; t.m[0] = s.x; t.m[5] = s.y; t.m[10] = s.z; return t; }
Transform create_rotation(const Vector& axis, double angle) {
  const double len = std::sqrt((axis.x * axis.x + axis.y * axis.y) + axis.z * axis.z);
  const double x = axis.x / len, y = axis.y / len, z = axis.z / len, s = std::sin(angle), c = std::cos(angle), k = 1.0 - c;
  Transform t;
  t.m = {x * x * k + c, x * y * k - z * s, x * z * k + y * s, 0, x * y * k + z * s, y * y * k + c, y * z * k - x * s, 0,
         x * z * k - y * s, y * z * k + x * s, z * z * k + c, 0};
  return t;
}

Ray::Ray(const Point& o, const Vector& d, double max_t_) : origin(o), max_t(max_t_) {
  const double len = std::sqrt((d.x * d.x + d.y * d.y) + d.z * d.z);
  dir = {d.x / len, d.y / len, d.z / len};
}
std::optional<Point> Ray::at(double t) const {
  if (t < 0.0 || t > max_t) return std::nullopt;
  return Point{origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t};
}

// ---- spectra ---------------------------------------------------------------------------------

SampledSpectrum SampledSpectrum::constant(float v) { SampledSpectrum s; s.coefficients.fill(v); return s; }
SampledSpectrum SampledSpectrum::from_RGB(float r, float g, float b, SpectrumType type) {
  SampledSpectrum s;
  check(rtmi_spectrum_from_rgb(r, g, b, type == SpectrumType::Illuminant ? RTMI_SPEC_ILLUMINANT : RTMI_SPEC_REFLECTANCE,
                               s.coefficients.data()), "from_RGB");
  return s;
}
std::array<float, 3> SampledSpectrum::to_XYZ() const {
  std::array<float, 3> xyz{};
  check(rtmi_spectrum_to_xyz(coefficients.data(), xyz.data()), "to_XYZ");
  return xyz;
}
std::array<float, 3> SampledSpectrum::to_RGB() const {
  std::array<float, 3> xyz = to_XYZ(), rgb{};
  check(rtmi_xyz_to_rgb(xyz.data(), rgb.data()), "XYZ_to_RGB");
  return rgb;
}

// ---- materials -------------------------------------------------------------------------------

Material material_from_mtl(const MtlMaterial& m) {
  auto refl = [](const std::array<double, 3>& c) {
    return SampledSpectrum::from_RGB(float(c[0]), float(c[1]), float(c[2]), SpectrumType::Reflectance);
  };
  Material out;
  out.tipo = Type::Lambertian;
  out.ambient_color = refl(m.color_ambient);
  if (m.color_emissive) out.emitted_color = refl(*m.color_emissive);
  out.diffused_color = refl(m.color_diffuse);
  out.specular_color = refl(m.color_specular);
  out.specular_coefficient = m.specular_coefficient;
  out.optical_density = m.optical_density;
  return out;
}

Triangle::Triangle(const Point& p1, const Point& p2, const Point& p3, const Transform& t, const Material& m)
    : vertices{p1, p2, p3}, local_to_world(t), material(m) {
  // every loader of the reference passes the identity (src/scene_config.rs:189, src/shapes/model_obj.rs:54);
  // a general map is folded into the vertices, which is what the GPU path stores
  if (t.m != Transform::identity().m) for (auto& v : vertices) v = t * v;
  local_to_world = Transform::identity();
}

// ---- camera ----------------------------------------------------------------------------------

Camera::Camera(const Point& focus, double focal_distance, double field_of_view, std::array<double, 3> rotation,
               std::array<std::uint32_t, 2> resolution) {
  const double f[3] = {focus.x, focus.y, focus.z};
  check(rtmi_camera_new(f, focal_distance, field_of_view, rotation.data(), resolution[0], resolution[1], &cam_), "Camera::new");
}
Ray Camera::get_ray(double i, double j) const {
  rtmi_ray r;
  check(rtmi_camera_get_ray(&cam_, i, j, &r), "Camera::get_ray");
  return Ray({r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, std::numeric_limits<double>::infinity());
}

// ---- flattening ------------------------------------------------------------------------------

namespace {
std::uint32_t add_material(FlatScene& f, const Material& m) {
  rtmi_material r;
  std::memset(&r, 0, sizeof r);
  switch (m.tipo) {
    case Type::Emitter: r.type = RTMI_MAT_EMITTER; break;
    case Type::Lambertian: r.type = RTMI_MAT_LAMBERTIAN; break;
    case Type::Specular: r.type = RTMI_MAT_SPECULAR; break;
    case Type::Glossy: r.type = RTMI_MAT_GLOSSY; break;
    case Type::Dielectric: r.type = RTMI_MAT_DIELECTRIC; break;
  }
  r.specular_coefficient = float(m.specular_coefficient.value_or(0.0));
  r.optical_density = float(m.optical_density.value_or(1.0));
  auto put = [&](const std::optional<SampledSpectrum>& s, std::uint32_t bit, float* dst) {
    if (!s) return;
    r.present |= bit;
    std::memcpy(dst, s->coefficients.data(), sizeof(float) * RTMI_NBINS);
  };
  put(m.ambient_color, RTMI_HAS_AMBIENT, r.ambient); put(m.emitted_color, RTMI_HAS_EMITTED, r.emitted);
  put(m.diffused_color, RTMI_HAS_DIFFUSED, r.diffused); put(m.specular_color, RTMI_HAS_SPECULAR, r.specular);
  f.materials.push_back(r);
  return std::uint32_t(f.materials.size() - 1);
}
rtmi_triangle make_tri(const std::array<Point, 3>& v) {
  rtmi_triangle t;
  for (int k = 0; k < 3; ++k) { t.v[k * 3] = v[k].x; t.v[k * 3 + 1] = v[k].y; t.v[k * 3 + 2] = v[k].z; }
  return t;
}
}  // namespace

FlatScene flatten(const std::vector<Shape>& shapes) {
  FlatScene f;
  for (const Shape& s : shapes) {
    rtmi_object o{};
    if (const Sphere* sp = std::get_if<Sphere>(&s)) {
      rtmi_sphere r;
      std::memcpy(r.l2w, sp->local_to_world.m.data(), sizeof r.l2w);
      r.radius = sp->radio;
      o.kind = RTMI_OBJ_SPHERE; o.first = std::uint32_t(f.spheres.size()); o.count = 1;
      o.material = add_material(f, sp->material);
      f.spheres.push_back(r);
    } else if (const Triangle* tr = std::get_if<Triangle>(&s)) {
      o.kind = RTMI_OBJ_TRIANGLE; o.first = std::uint32_t(f.triangles.size()); o.count = 1;
      o.material = add_material(f, tr->material);
      f.triangles.push_back(make_tri(tr->vertices));
    } else {
      const ModelObj& mo = std::get<ModelObj>(s);
      o.kind = RTMI_OBJ_MODEL; o.first = std::uint32_t(f.triangles.size()); o.count = std::uint32_t(mo.triangles.size());
      o.material = add_material(f, mo.material);
      for (const auto& t : mo.triangles) f.triangles.push_back(make_tri(t));
    }
    f.objects.push_back(o);
  }
  return f;
}

// ---- scene -----------------------------------------------------------------------------------

struct GpuScene {
  FlatScene flat;
  rtmi_scene* handle = nullptr;
  ~GpuScene() { if (handle) rtmi_scene_destroy(handle); }
};

rtmi_ctx* context() {
  static rtmi_ctx* ctx = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* dev = std::getenv("RTMI_DEVICE");
    check(rtmi_ctx_create(dev ? std::atoi(dev) : 0, 0, &ctx), "rtmi_ctx_create");
  });
  return ctx;
}

Scene::Scene() = default;
Scene::~Scene() = default;
Scene::Scene(const Scene&) = default;              // integrators clone the scene (src/integrators/random_walk.rs:24);
Scene& Scene::operator=(const Scene&) = default;   // the clone shares the immutable GPU copy

void Scene::add_shape(const Shape& s) { objetos_.push_back(s); gpu_.reset(); }
void Scene::set_hit_rule(std::uint32_t rule) { hit_rule_ = rule; gpu_.reset(); }

rtmi_scene* Scene::gpu() const {
  if (!gpu_) {
    auto g = std::make_shared<GpuScene>();
    g->flat = flatten(objetos_);
    rtmi_scene_desc d = g->flat.desc(hit_rule_);
    check(rtmi_scene_create(context(), &d, &g->handle), "rtmi_scene_create");
    gpu_ = g;
  }
  return gpu_->handle;
}

std::vector<std::optional<Intersection>> Scene::intersect_rays(const std::vector<Ray>& rays) const {
  std::vector<rtmi_ray> in(rays.size());
  for (std::size_t i = 0; i < rays.size(); ++i) {
    in[i].o[0] = rays[i].origin.x; in[i].o[1] = rays[i].origin.y; in[i].o[2] = rays[i].origin.z;
    in[i].d[0] = rays[i].dir.x; in[i].d[1] = rays[i].dir.y; in[i].d[2] = rays[i].dir.z;
  }
  std::vector<rtmi_hit> hits(rays.size());
  check(rtmi_intersect_batch(gpu(), in.data(), in.size(), hits.data(), nullptr), "rtmi_intersect_batch");
  std::vector<std::optional<Intersection>> out(rays.size());
  for (std::size_t i = 0; i < rays.size(); ++i)
    if (hits[i].obj != RTMI_MISS)
      out[i] = Intersection{hits[i].obj, hits[i].prim, {hits[i].p[0], hits[i].p[1], hits[i].p[2]},
                            {hits[i].n[0], hits[i].n[1], hits[i].n[2]}, hits[i].t};
  return out;
}

std::optional<Intersection> Scene::intersect_ray(const Ray& ray) const { return intersect_rays({ray})[0]; }

// ---- integrators -----------------------------------------------------------------------------

Image SamplerIntegrator::render() const {
  Image img;
  img.width = camera_.width(); img.height = camera_.height();
  img.rgb.assign(std::size_t(img.width) * img.height * 3, 0);
  rtmi_render_params p{};
  p.integrator = kind_; p.max_depth = std::uint32_t(max_depth()); p.spp = std::uint32_t(iterations_); p.seed = seed;
  check(rtmi_render(scene_.gpu(), &camera_.raw(), &p, img.rgb.data(), nullptr, &last_stats), "rtmi_render");
  return img;
}

// ---- BMP -------------------------------------------------------------------------------------

void Image::save(const std::string& path) const {
  const std::uint32_t row = (width * 3 + 3) & ~3u, data = row * height, off = 54;
  std::vector<std::uint8_t> buf(off + data, 0);
  auto u32 = [&](std::size_t at, std::uint32_t v) { for (int k = 0; k < 4; ++k) buf[at + k] = std::uint8_t(v >> (8 * k)); };
  auto u16 = [&](std::size_t at, std::uint16_t v) { buf[at] = std::uint8_t(v); buf[at + 1] = std::uint8_t(v >> 8); };
  buf[0] = 'B'; buf[1] = 'M';
  u32(2, off + data); u32(10, off);
  u32(14, 40); u32(18, width); u32(22, height); u16(26, 1); u16(28, 24); u32(34, data); u32(38, 2835); u32(42, 2835);
  for (std::uint32_t y = 0; y < height; ++y) {
    std::uint8_t* dst = buf.data() + off + std::size_t(height - 1 - y) * row;  // bottom-up rows, BGR
    const std::uint8_t* src = rgb.data() + std::size_t(y) * width * 3;
    for (std::uint32_t x = 0; x < width; ++x) { dst[x * 3] = src[x * 3 + 2]; dst[x * 3 + 1] = src[x * 3 + 1]; dst[x * 3 + 2] = src[x * 3]; }
  }
  std::ofstream f(path, std::ios::binary);
  if (!f) throw Error("No se pudo guardar la imagen en \"" + path + "\"");
  f.write(reinterpret_cast<const char*>(buf.data()), std::streamsize(buf.size()));
}

std::string read_file(const std::string& name) {
  std::ifstream f(name, std::ios::binary);
  if (!f) throw Error("Archivo \"\\\"" + name + "\\\"\" no encontrado.");  // src/auxiliar.rs:21-23 ({:?} of a &str)
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return text;
}

}  // namespace raytracer
