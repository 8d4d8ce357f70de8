// lbvh.h -- device-side builder of the per-mesh BVH (Morton order + binary radix tree), see lbvh.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "rt_types.h"

namespace rtmi {

struct LbvhOut {
  std::vector<RtWord4> words;          // quantised nodes in the requested format, node 0 = root
  std::vector<DevMeshTri> tris;        // leaf order, each leaf sorted by file-order index
  std::vector<DevTriFrame> frames;     // parallel to tris
  uint32_t n_nodes = 0, max_depth = 0;
  double center[3] = {0, 0, 0}, radius = 0;
  float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};       // padded binary32 box of the whole mesh, relative to center
  float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
  double device_ms = 0;                // upload of the vertices to the last kernel's completion
};

// Builds the tree of `n` >= 2 triangles (9 doubles each, host memory, file order) on the current device.
hipError_t build_lbvh(const double* host_vertices, uint32_t n, int format, LbvhOut* out, std::string* error);

}  // namespace rtmi
