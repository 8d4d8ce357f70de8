// model_obj.cpp -- OBJ / MTL loading with the semantics of src/shapes/model_obj.rs:15-73:
// positions only (normals and texture coordinates are ignored, :40-53), the FIRST material of the
// MTL library for the whole model (:23), polygons fan-triangulated in face order, identity
// transform, lines and points dropped. The reference delegates parsing to wavefront_obj 10.0.
#include <fstream>
#include <sstream>

#include "raytracer.hpp"

namespace raytracer {

namespace {

std::vector<std::string> split_ws(const std::string& line) {
  std::istringstream in(line);
  std::vector<std::string> tok;
  for (std::string t; in >> t;) tok.push_back(t);
  return tok;
}

std::array<double, 3> rgb_of(const std::vector<std::string>& tok, const std::string& what) {
  if (tok.size() < 4) throw Error("MTL: se esperaban tres componentes en " + what);
  return {std::stod(tok[1]), std::stod(tok[2]), std::stod(tok[3])};
}

// The reference opens paths relative to the working directory (src/shapes/model_obj.rs:16,21);
// the directory of the scene file is tried second so that scenes can be loaded from elsewhere.
std::string resolve(const std::string& file, const std::string& base_dir) {
  if (std::ifstream(file).good() || base_dir.empty()) return file;
  const std::string alt = base_dir + "/" + file;
  return std::ifstream(alt).good() ? alt : file;
}

}  // namespace

std::vector<MtlMaterial> parse_mtl(const std::string& text) {
  std::vector<MtlMaterial> out;
  std::istringstream in(text);
  for (std::string line; std::getline(in, line);) {
    const auto tok = split_ws(line.substr(0, line.find('#')));
    if (tok.empty()) continue;
    if (tok[0] == "newmtl") { out.emplace_back(); if (tok.size() > 1) out.back().name = tok[1]; continue; }
    if (out.empty()) continue;
    MtlMaterial& m = out.back();
    if (tok[0] == "Ns" && tok.size() > 1) m.specular_coefficient = std::stod(tok[1]);
    else if (tok[0] == "Ka") m.color_ambient = rgb_of(tok, "Ka");
    else if (tok[0] == "Kd") m.color_diffuse = rgb_of(tok, "Kd");
    else if (tok[0] == "Ks") m.color_specular = rgb_of(tok, "Ks");
    else if (tok[0] == "Ke") m.color_emissive = rgb_of(tok, "Ke");
    else if (tok[0] == "Ni" && tok.size() > 1) m.optical_density = std::stod(tok[1]);
    // d, illum, map_*: parsed by the crate, never read on this path
  }
  return out;
}

ModelObj::ModelObj(const std::string& file, FanOrigin fan, const std::string& base_dir) {
  const std::string text = read_file(resolve(file, base_dir));
  std::vector<Point> vertices;
  std::string mtllib;
  std::istringstream in(text);
  for (std::string line; std::getline(in, line);) {
    const auto tok = split_ws(line.substr(0, line.find('#')));
    if (tok.empty()) continue;
    if (tok[0] == "v") {
      if (tok.size() < 4) throw Error("OBJ: vértice incompleto");
      vertices.push_back({std::stod(tok[1]), std::stod(tok[2]), std::stod(tok[3])});
    } else if (tok[0] == "mtllib" && tok.size() > 1) {
      mtllib = tok[1];
    } else if (tok[0] == "f") {
      std::vector<std::size_t> idx;
      for (std::size_t k = 1; k < tok.size(); ++k) {
        const long i = std::stol(tok[k].substr(0, tok[k].find('/')));
        const long at = i > 0 ? i - 1 : long(vertices.size()) + i;
        if (at < 0 || std::size_t(at) >= vertices.size()) throw Error("OBJ: índice de vértice fuera de rango");
        idx.push_back(std::size_t(at));
      }
      const std::size_t n = idx.size();
      if (n < 3) continue;  // points and lines are not Primitive::Triangle (src/shapes/model_obj.rs:40)
      if (fan == FanOrigin::Last)
        for (std::size_t i = 0; i + 2 < n; ++i) triangles.push_back({vertices[idx[n - 1]], vertices[idx[i]], vertices[idx[i + 1]]});
      else
        for (std::size_t i = 1; i + 1 < n; ++i) triangles.push_back({vertices[idx[0]], vertices[idx[i]], vertices[idx[i + 1]]});
    }
  }
  if (!mtllib.empty()) {
    const auto mats = parse_mtl(read_file(resolve(mtllib, base_dir)));
    if (mats.empty()) throw Error("No se pudo cargar el material de \"" + mtllib + "\"");
    material = material_from_mtl(mats.front());
  }
}

}  // namespace raytracer
