// scene_config.cpp -- the scene description schema of src/scene_config.rs:11-271 over a small
// TOML reader (tables, arrays of tables, dotted table headers, strings, integers, floats,
// booleans, nested and multi-line arrays, comments): what the reference's scene files use.
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "raytracer.hpp"

namespace raytracer {

// ---- TOML subset ------------------------------------------------------------------------------
namespace toml {

const Value* Table::get(const std::string& key) const {
  for (const auto& kv : items) if (kv.first == key) return &kv.second;
  return nullptr;
}

namespace {

struct Reader {
  const std::string& s;
  std::size_t p = 0;
  int line = 1;
  explicit Reader(const std::string& text) : s(text) {}
  [[noreturn]] void fail(const std::string& what) const { throw Error("TOML parse error at line " + std::to_string(line) + ": " + what); }
  bool eof() const { return p >= s.size(); }
  char peek() const { return eof() ? '\0' : s[p]; }
  void skip_space(bool newlines) {
    while (!eof()) {
      const char c = s[p];
      if (c == ' ' || c == '\t' || c == '\r') ++p;
      else if (c == '\n' && newlines) { ++p; ++line; }
      else if (c == '#') while (!eof() && s[p] != '\n') ++p;
      else break;
    }
  }
  std::string key() {
    skip_space(false);
    std::string k;
    if (peek() == '"') { ++p; while (!eof() && s[p] != '"') k += s[p++]; if (eof()) fail("unterminated key"); ++p; return k; }
    while (!eof() && (std::isalnum(static_cast<unsigned char>(s[p])) || s[p] == '_' || s[p] == '-')) k += s[p++];
    if (k.empty()) fail("expected a key");
    return k;
  }
  Value value() {
    skip_space(false);
    Value v;
    const char c = peek();
    if (c == '"' || c == '\'') {
      const char q = c; ++p;
      std::string out;
      while (!eof() && s[p] != q) {
        if (q == '"' && s[p] == '\\' && p + 1 < s.size()) {
          const char e = s[p + 1]; p += 2;
          out += e == 'n' ? '\n' : e == 't' ? '\t' : e;
        } else { if (s[p] == '\n') fail("newline in string"); out += s[p++]; }
      }
      if (eof()) fail("unterminated string");
      ++p;
      v.v = out;
    } else if (c == '[') {
      ++p;
      Array arr;
      for (;;) {
        skip_space(true);
        if (peek() == ']') { ++p; break; }
        arr.push_back(value());
        skip_space(true);
        if (peek() == ',') { ++p; continue; }
        if (peek() == ']') { ++p; break; }
        fail("expected ',' or ']' in array");
      }
      v.v = std::move(arr);
    } else if (s.compare(p, 4, "true") == 0) { p += 4; v.v = true; }
    else if (s.compare(p, 5, "false") == 0) { p += 5; v.v = false; }
    else {
      std::string tok;
      while (!eof() && (std::isalnum(static_cast<unsigned char>(s[p])) || s[p] == '+' || s[p] == '-' || s[p] == '.' || s[p] == '_')) {
        if (s[p] != '_') tok += s[p];
        ++p;
      }
      if (tok.empty()) fail("expected a value");
      const std::string bare = (tok[0] == '+' || tok[0] == '-') ? tok.substr(1) : tok;
      const bool is_float = tok.find_first_of(".eE") != std::string::npos || bare == "inf" || bare == "nan";
      char* end = nullptr;
      if (is_float) {
        const double d = std::strtod(tok.c_str(), &end);
        if (*end) fail("bad float '" + tok + "'");
        v.v = d;
      } else {
        const long long i = std::strtoll(tok.c_str(), &end, 10);
        if (*end) fail("bad integer '" + tok + "'");
        v.v = std::int64_t(i);
      }
    }
    return v;
  }
};

Table* child_table(Table& parent, const std::string& key, Reader& r) {
  for (auto& kv : parent.items)
    if (kv.first == key) {
      if (auto* t = std::get_if<Table>(&kv.second.v)) return t;
      if (auto* a = std::get_if<Array>(&kv.second.v)) {  // [Scene.material] addresses the LAST element of [[Scene]]
        if (a->empty()) r.fail("empty array of tables '" + key + "'");
        if (auto* t = std::get_if<Table>(&a->back().v)) return t;
      }
      r.fail("'" + key + "' is not a table");
    }
  Value v; v.v = Table{};
  parent.items.emplace_back(key, std::move(v));
  return std::get_if<Table>(&parent.items.back().second.v);
}

}  // namespace

Table parse(const std::string& text) {
  Table root;
  Table* cur = &root;
  Reader r(text);
  for (;;) {
    r.skip_space(true);
    if (r.eof()) break;
    if (r.peek() == '[') {
      ++r.p;
      const bool array = r.peek() == '[';
      if (array) ++r.p;
      std::vector<std::string> path{r.key()};
      r.skip_space(false);
      while (r.peek() == '.') { ++r.p; path.push_back(r.key()); r.skip_space(false); }
      if (r.peek() != ']') r.fail("expected ']'");
      ++r.p;
      if (array) { if (r.peek() != ']') r.fail("expected ']]'"); ++r.p; }
      Table* t = &root;
      for (std::size_t i = 0; i + 1 < path.size(); ++i) t = child_table(*t, path[i], r);
      if (array) {
        Value* slot = nullptr;
        for (auto& kv : t->items) if (kv.first == path.back()) slot = &kv.second;
        if (!slot) { Value v; v.v = Array{}; t->items.emplace_back(path.back(), std::move(v)); slot = &t->items.back().second; }
        auto* arr = std::get_if<Array>(&slot->v);
        if (!arr) r.fail("'" + path.back() + "' is not an array of tables");
        Value e; e.v = Table{};
        arr->push_back(std::move(e));
        cur = std::get_if<Table>(&arr->back().v);
      } else {
        cur = child_table(*t, path.back(), r);
      }
    } else {
      const std::string k = r.key();
      r.skip_space(false);
      if (r.peek() != '=') r.fail("expected '=' after key '" + k + "'");
      ++r.p;
      Value v = r.value();
      if (cur->get(k)) r.fail("duplicate key '" + k + "'");
      cur->items.emplace_back(k, std::move(v));
    }
    r.skip_space(false);
    if (!r.eof() && r.peek() != '\n') r.fail("unexpected characters after value");
  }
  return root;
}

}  // namespace toml

// ---- schema -----------------------------------------------------------------------------------

namespace {

std::vector<double> float_array(const toml::Value* v, std::size_t n, const std::string& err) {
  const toml::Array* a = v ? v->as_array() : nullptr;
  if (!a) throw Error(err);
  std::vector<double> out;
  for (const auto& e : *a) {
    const double* d = e.as_float();     // as_float(): TOML integers are rejected, as in the reference
    if (!d) throw Error(err);
    out.push_back(*d);
  }
  if (n && out.size() != n) throw Error(err);
  return out;
}

// Material::from_toml (src/scene_config.rs:197-250)
Material material_from_toml(const toml::Table& t) {
  const std::string err = "No se pudo cargar el material";
  const std::vector<double> c = float_array(t.get("albedo"), 3, err);
  const toml::Value* tv = t.get("type");
  if (!tv) throw Error(err);
  const std::string* type = tv->as_str();
  const bool emitter = type && *type == "Emitter";
  const SampledSpectrum color = SampledSpectrum::from_RGB(float(c[0]), float(c[1]), float(c[2]),
                                                          emitter ? SpectrumType::Illuminant : SpectrumType::Reflectance);
  Material m;
  if (type && *type == "Lambertian") { m.tipo = Type::Lambertian; m.ambient_color = color; }
  else if (type && *type == "Specular") { m.tipo = Type::Specular; m.specular_color = color; }
  else if (emitter) { m.tipo = Type::Emitter; m.emitted_color = color; }
  // extensions (no counterpart in the reference): type = "Glossy" with specular_coefficient, "Dielectric" with optical_density
  else if (type && (*type == "Glossy" || *type == "Dielectric")) {
    m.tipo = *type == "Glossy" ? Type::Glossy : Type::Dielectric;
    m.specular_color = color;
    const char* key = *type == "Glossy" ? "specular_coefficient" : "optical_density";
    const toml::Value* v = t.get(key);
    const double* d = v ? v->as_float() : nullptr;
    if (!d) throw Error(err);
    if (*type == "Glossy") m.specular_coefficient = *d; else m.optical_density = *d;
  }
  else throw Error("not yet implemented: material type " + (type ? "\"" + *type + "\"" : std::string("<none>")));  // todo!() :225,244,247
  return m;
}

// get_material (src/scene_config.rs:265-271)
Material get_material(const toml::Table& t, const std::string& err) {
  const toml::Value* m = t.get("material");
  if (!m) return Material{};
  if (const toml::Table* mt = m->as_table()) return material_from_toml(*mt);
  throw Error(err);
}

Point point_from_toml(const toml::Value& v) {  // create_point_from_toml :253-263
  const std::vector<double> p = float_array(&v, 3, "No se pudo cargar el punto");
  return {p[0], p[1], p[2]};
}

}  // namespace

toml::Table parse_file(const std::string& path) { return toml::parse(read_file(path)); }

Camera parse_camera(const toml::Table& table) {
  const toml::Value* cv = table.get("Camera");
  const toml::Table* t = cv ? cv->as_table() : nullptr;
  if (!t) throw Error("No se ha especificado la cámara.");
  const std::string err = "No se pudo cargar la cámara.";
  auto integer = [&](const char* k) {
    const toml::Value* v = t->get(k);
    const std::int64_t* i = v ? v->as_integer() : nullptr;
    if (!i) throw Error(err);
    return std::uint32_t(*i);
  };
  auto real = [&](const char* k) {
    const toml::Value* v = t->get(k);
    const double* d = v ? v->as_float() : nullptr;
    if (!d) throw Error(err);
    return *d;
  };
  const std::uint32_t width = integer("width"), height = integer("height");
  const double focal_distance = real("focal_distance"), field_of_view = real("field_of_view");
  const std::vector<double> pos = float_array(t->get("position"), 3, err), rot = float_array(t->get("rotation"), 3, err);
  return Camera({pos[0], pos[1], pos[2]}, focal_distance, field_of_view, {rot[0], rot[1], rot[2]}, {width, height});
}

Scene parse_scene(const toml::Table& table, const LoadOptions& opt) {
  const toml::Value* sv = table.get("Scene");
  const toml::Array* models = sv ? sv->as_array() : nullptr;
  if (!models) throw Error("No se ha especificado la escena.");
  const std::string err = "Error con la escena definida.";
  Scene scene;
  for (const toml::Value& mv : *models) {
    const toml::Table* model = mv.as_table();
    if (!model) throw Error(err);
    const toml::Value* tv = model->get("type");
    if (!tv) throw Error(err);
    const std::string* type = tv->as_str();
    if (type && *type == "Sphere") {  // Sphere::from_toml :128-166
      const std::string e = "No se pudo cargar el modelo de esfera";
      const std::vector<double> c = float_array(model->get("center"), 3, e);
      std::vector<double> s{1.0, 1.0, 1.0};
      if (model->get("scale")) s = float_array(model->get("scale"), 3, e);
      const toml::Value* rv = model->get("radius");
      const double* radius = rv ? rv->as_float() : nullptr;
      if (!radius) throw Error(e);
      const Material m = get_material(*model, e);
      scene.add_shape(Sphere(create_translation({c[0], c[1], c[2]}) * create_scaling({s[0], s[1], s[2]}), *radius, m));
    } else if (type && *type == "Triangle") {  // Triangle::from_toml :170-192
      const std::string e = "No se pudo cargar el modelo de triángulo.";
      const toml::Value* vv = model->get("vertices");
      const toml::Array* verts = vv ? vv->as_array() : nullptr;
      if (!verts || verts->size() != 3) throw Error(e);
      const Point p1 = point_from_toml((*verts)[0]), p2 = point_from_toml((*verts)[1]), p3 = point_from_toml((*verts)[2]);
      scene.add_shape(Triangle(p1, p2, p3, Transform::identity(), get_material(*model, e)));
    } else if (type && *type == "ModelObj") {  // ModelObj::from_toml :118-124
      const std::string e = "No se pudo cargar el modelo de objeto";
      const toml::Value* pv = model->get("path");
      const std::string* path = pv ? pv->as_str() : nullptr;
      if (!path) throw Error(e);
      ModelObj obj(*path, opt.fan, opt.base_dir);
      // extension the reference's author planned (:122): a [Scene.material] on a mesh replaces the MTL material
      if (model->get("material")) obj.material = get_material(*model, e);
      scene.add_shape(std::move(obj));
    } else {
      std::cerr << "[scene_config] s = " << (type ? "\"" + *type + "\"" : std::string("\"No type\"")) << std::endl;  // dbg! :104-109
    }
  }
  return scene;
}

}  // namespace raytracer
