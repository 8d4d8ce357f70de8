// main.cpp -- `raytracer [scene.toml] [file.bmp]`: the reference's CLI (src/main.rs:19-101) over the GPU path.
// Renders the albedo and normal AOVs (20 samples each) and the depth-10, 100-sample random walk.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "raytracer.hpp"

using namespace raytracer;

static void print_help() { std::cout << "Uso: raytracer [scene.toml] [file.bmp]" << std::endl; }

static bool ends_with(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

static int program(int argc, char** argv) {
  std::vector<std::string> args(argv + 1, argv + argc);
  if (args.size() == 1 && (args[0] == "-h" || args[0] == "--help")) { print_help(); return 0; }
  std::string output = "render.bmp", scene_file = "scene.toml";
  for (const std::string& a : args) {
    if (ends_with(a, ".toml")) scene_file = a;
    else if (ends_with(a, ".bmp")) output = a;
    else { print_help(); return 0; }
  }
  // knobs the reference hard-codes (src/main.rs:75,84,93); the defaults are the reference's values
  auto env_u = [](const char* k, unsigned long d) { const char* v = std::getenv(k); return v ? std::strtoul(v, nullptr, 10) : d; };
  const std::size_t aov_spp = env_u("RAYTRACER_AOV_SPP", 20), spp = env_u("RAYTRACER_SPP", 100), depth = env_u("RAYTRACER_DEPTH", 10);

  const toml::Table input = parse_file(scene_file);
  const Camera camera = parse_camera(input);
  LoadOptions opt;
  const auto slash = scene_file.find_last_of('/');
  if (slash != std::string::npos) opt.base_dir = scene_file.substr(0, slash);
  if (const char* fan = std::getenv("RAYTRACER_FAN")) opt.fan = std::string(fan) == "first" ? FanOrigin::First : FanOrigin::Last;
  Scene scene = parse_scene(input, opt);
  if (const char* rule = std::getenv("RAYTRACER_HIT_RULE")) scene.set_hit_rule(std::string(rule) == "nearest" ? 1u : 0u);

  auto run = [&](const SamplerIntegrator& integrator, const std::string& file) {
    const auto t0 = std::chrono::steady_clock::now();
    const Image img = integrator.render();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    img.save(file);
    std::printf("%llu segmentos en %.3f s (%.1f Mseg/s)\n", (unsigned long long)integrator.last_stats.segments, s,
                integrator.last_stats.segments / s / 1e6);
  };
  run(AlbedoIntegrator(camera, scene, aov_spp), "output-albedo.bmp");
  std::cout << "Imagen guardada en output-albedo.bmp." << std::endl;
  run(NormalIntegrator(camera, scene, aov_spp), "output-normal.bmp");
  std::cout << "Imagen guardada en output-normal.bmp." << std::endl;
  run(RandomWalkIntegrator(camera, scene, depth, spp), output);
  std::cout << "Imagen guardada en \"" << output << "\"." << std::endl;
  return 0;
}

int main(int argc, char** argv) {
  try {
    return program(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;  // src/main.rs:51-53
    return 0;
  }
}
