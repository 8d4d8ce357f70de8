// bvh.h -- host-side builder of the per-mesh binary BVH the kernels traverse.
#pragma once
#include <cstdint>
#include <vector>

#include "rt_types.h"

namespace rtmi {

struct BvhBuild {
  std::vector<BvhNode> nodes;        // nodes[0] is the root; child refs are relative to it
  std::vector<DevMeshTri> tris;      // leaf order; .index = position in the input array
  uint32_t max_depth = 0;            // deepest chain of inner nodes (bounds the traversal stack)
  double center[3] = {0, 0, 0};      // node boxes are stored relative to this point ...
  double radius = 0;                 // ... and every triangle lies within this distance of it
};

// Binned-SAH build over `n` triangles given as 9 doubles each (world space, file order).
// The reference has no acceleration structure (src/shapes/model_obj.rs:77-87 scans every
// triangle); the tree only prunes work, it never changes which triangle wins under either
// hit rule. Boxes are stored in binary32, relative to the mesh centre, rounded outward and
// padded; together with the per-ray margin the kernel adds, a binary32 box test against them can
// never reject a triangle the exact binary64 test would accept.
BvhBuild build_bvh(const double* tri_vertices, uint32_t n, uint32_t leaf_size);

// The grid a mesh's node boxes are quantised on (relative to the mesh centre) and the nodes in the form the kernels
// walk (rt_types.h: RT_NODES_*). Every quantised box contains the binary32 box it was made from, with at least one
// grid step to spare on every side.
struct QuantizedNodes {
  float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
  std::vector<RtWord4> words;        // 2 (SMALL, NEAREST) or 3 (WIDE) per node, in node order
};
QuantizedNodes quantize_nodes(const BvhBuild& build, int format);
inline uint32_t node_words4(int format) { return format == RT_NODES_QUAD ? 4u : (format == RT_NODES_WIDE ? 3u : 2u); }

// The same tree with four children per node (RT_NODES_QUAD): starting at the root, a node's inner child with the largest
// box is replaced by that child's own two children until the node has four children or only leaves; the children that
// are still inner nodes become four-wide nodes the same way. Leaves and their triangle order are the binary tree's.
// n_nodes = four-wide nodes made; max_depth = deepest chain of them; max_stack = the most references a walk can have
// stacked at once (the other children of every node along one path, plus one), which is what sizes the kernels' stacks.
struct QuadNodes : QuantizedNodes { uint32_t n_nodes = 0, max_depth = 0, max_stack = 0; };
QuadNodes quantize_nodes_quad(const BvhBuild& build);
// Every reference of the four-wide nodes is in range, every node but the root has exactly one parent before it, and the
// leaves cover every triangle exactly once: what the kernels rely on when they follow these references.
bool verify_quad_nodes(const BvhBuild& build, const QuadNodes& q);

}  // namespace rtmi
