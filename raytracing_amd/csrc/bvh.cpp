// bvh.cpp -- binned-SAH BVH2 builder (host). See bvh.h.
#include "bvh.h"
#include "bvh_quant.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rtmi {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kForceMedianDepth = 40;  // past this depth splits are balanced, so depth stays bounded
// Relative cost of a box-test step and a triangle test in the surface-area heuristic, measured on the GPU (Suzanne,
// first-index rule, leaves of at most 4, 1920x1080x64 spp): with the box test at 48 vector instructions for both children
// (binary16 planes) and the binary64 triangle test at about 100, the step takes 16.99 ms at 0.5, 16.5 at 0.7, 16.35 at 0.9,
// 16.25 at 1.1, 16.20 at 1.4 and 16.30 at 1.8; the nearest rule and the million-triangle mesh move by 1 % the same way.
constexpr double kTraversalCost = 1.0;
static double kTriangleCost = 1.2;   // RTMI_SAH_TRIANGLE_COST overrides it (read when a tree is built; for measurements)

struct Box {
  double lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; }
  }
  void grow(const double* p) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  double half_area() const {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : dx * dy + dy * dz + dz * dx;
  }
};

struct Prim { Box box; double c[3]; uint32_t index; };

struct Builder {
  std::vector<Prim> prims;
  const double* verts;
  BvhBuild out;
  uint32_t leaf_size;
  double pad;
  double center[3];

  // binary32 box, relative to the mesh centre, that contains the binary64 one: pad, then round outward
  void store_box(const Box& b, float* lo, float* hi) const {
    for (int k = 0; k < 3; ++k) {
      const double dl = (b.lo[k] - center[k]) - pad, dh = (b.hi[k] - center[k]) + pad;
      float l = float(dl), h = float(dh);
      if (double(l) > dl) l = std::nextafterf(l, -FLT_MAX);
      if (double(h) < dh) h = std::nextafterf(h, FLT_MAX);
      lo[k] = l; hi[k] = h;
    }
  }

  uint32_t make_leaf(uint32_t b, uint32_t e) {
    std::sort(prims.begin() + b, prims.begin() + e, [](const Prim& x, const Prim& y) { return x.index < y.index; });
    const uint32_t first = uint32_t(out.tris.size());
    for (uint32_t i = b; i < e; ++i) {
      DevMeshTri t;
      std::memcpy(t.v, verts + size_t(prims[i].index) * 9, sizeof t.v);
      t.index = prims[i].index; t.pad = 0;
      out.tris.push_back(t);
    }
    return RT_LEAF_FLAG | (first << 4) | (e - b);
  }

  // Chooses where to cut prims[b,e); returns the split position, or b when a leaf is cheaper.
  uint32_t partition(uint32_t b, uint32_t e, const Box& bounds, uint32_t depth) {
    const uint32_t n = e - b;
    Box cb; cb.reset();
    for (uint32_t i = b; i < e; ++i) cb.grow(prims[i].c);
    int best_axis = -1, best_plane = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    if (depth < kForceMedianDepth) {
      for (int axis = 0; axis < 3; ++axis) {
        const double ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.0)) continue;
        Box bin_box[kBins]; uint32_t bin_n[kBins] = {0};
        for (auto& bb : bin_box) bb.reset();
        const double scale = kBins / ext;
        for (uint32_t i = b; i < e; ++i) {
          int k = std::min(kBins - 1, std::max(0, int((prims[i].c[axis] - cb.lo[axis]) * scale)));
          bin_box[k].grow(prims[i].box); bin_n[k]++;
        }
        double right_area[kBins]; Box acc; acc.reset(); uint32_t cnt = 0;
        for (int k = kBins - 1; k > 0; --k) { acc.grow(bin_box[k]); right_area[k] = acc.half_area(); }
        acc.reset();
        uint32_t right_n[kBins]; uint32_t rn = 0;
        for (int k = kBins - 1; k > 0; --k) { rn += bin_n[k]; right_n[k] = rn; }
        for (int k = 1; k < kBins; ++k) {
          acc.grow(bin_box[k - 1]); cnt += bin_n[k - 1];
          if (cnt == 0 || right_n[k] == 0) continue;
          const double cost = acc.half_area() * cnt + right_area[k] * right_n[k];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_plane = k; }
        }
      }
    }
    if (best_axis >= 0) {
      const double parent = bounds.half_area();
      const double split_cost = kTraversalCost + kTriangleCost * (parent > 0 ? best_cost / parent : double(n));
      if (n <= leaf_size && split_cost >= kTriangleCost * n) return b;  // a leaf is cheaper
      const double ext = cb.hi[best_axis] - cb.lo[best_axis], scale = kBins / ext;
      auto mid = std::partition(prims.begin() + b, prims.begin() + e, [&](const Prim& p) {
        int k = std::min(kBins - 1, std::max(0, int((p.c[best_axis] - cb.lo[best_axis]) * scale)));
        return k < best_plane;
      });
      const uint32_t m = uint32_t(mid - prims.begin());
      if (m > b && m < e) return m;
    }
    if (n <= leaf_size) return b;
    // degenerate centroids or forced balance: median along the widest centroid axis
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
    const uint32_t m = b + n / 2;
    std::nth_element(prims.begin() + b, prims.begin() + m, prims.begin() + e,
                     [axis](const Prim& x, const Prim& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.index < y.index); });
    return m;
  }

  struct Child { uint32_t ref, min_index; Box box; };

  Child build_range(uint32_t b, uint32_t e, uint32_t depth) {
    Child c; c.box.reset(); c.min_index = 0xFFFFFFFFu;
    for (uint32_t i = b; i < e; ++i) { c.box.grow(prims[i].box); c.min_index = std::min(c.min_index, prims[i].index); }
    const uint32_t m = partition(b, e, c.box, depth);
    if (m == b) { c.ref = make_leaf(b, e); return c; }
    c.ref = build_inner(b, m, e, depth);
    return c;
  }

  uint32_t build_inner(uint32_t b, uint32_t m, uint32_t e, uint32_t depth) {
    const uint32_t id = uint32_t(out.nodes.size());
    out.nodes.emplace_back();
    out.max_depth = std::max(out.max_depth, depth + 1);
    Child l = build_range(b, m, depth + 1);
    Child r = build_range(m, e, depth + 1);
    BvhNode& nd = out.nodes[id];
    store_box(l.box, nd.lo0, nd.hi0); store_box(r.box, nd.lo1, nd.hi1);
    nd.ref0 = l.ref; nd.ref1 = r.ref; nd.min_index0 = l.min_index; nd.min_index1 = r.min_index;
    return id;
  }
};

}  // namespace

BvhBuild build_bvh(const double* tri_vertices, uint32_t n, uint32_t leaf_size) {
  if (const char* e = getenv("RTMI_SAH_TRIANGLE_COST")) { const double c = atof(e); if (c > 0.0) kTriangleCost = c; }
  Builder bld;
  bld.verts = tri_vertices;
  bld.leaf_size = std::max(1u, std::min(leaf_size, uint32_t(RT_LEAF_MAX_TRIS)));
  bld.prims.resize(n);
  Box all; all.reset();
  for (uint32_t i = 0; i < n; ++i) {
    Prim& p = bld.prims[i];
    p.box.reset(); p.index = i;
    for (int v = 0; v < 3; ++v) p.box.grow(tri_vertices + size_t(i) * 9 + v * 3);
    for (int k = 0; k < 3; ++k) p.c[k] = 0.5 * (p.box.lo[k] + p.box.hi[k]);
    all.grow(p.box);
  }
  double extent = 0.0, r2 = 0.0;
  for (int k = 0; k < 3; ++k) {
    bld.center[k] = n ? 0.5 * (all.lo[k] + all.hi[k]) : 0.0;
    const double half = n ? 0.5 * (all.hi[k] - all.lo[k]) : 0.0;
    extent = std::max(extent, half);
    r2 += half * half;
    bld.out.center[k] = bld.center[k];
  }
  bld.out.radius = std::sqrt(r2) * (1.0 + 0x1.0p-20);
  bld.pad = extent * 0x1.0p-20 + 1e-30;
  bld.out.nodes.reserve(n ? 2 * size_t(n) / bld.leaf_size + 2 : 1);
  bld.out.tris.reserve(n);

  // The root is always an inner node, so a mesh of any size starts the same way on the GPU.
  if (n == 0) {
    BvhNode nd;
    std::memset(&nd, 0, sizeof nd);
    for (int k = 0; k < 3; ++k) { nd.lo0[k] = nd.lo1[k] = FLT_MAX; nd.hi0[k] = nd.hi1[k] = -FLT_MAX; }
    nd.ref0 = nd.ref1 = RT_REF_NONE; nd.min_index0 = nd.min_index1 = 0xFFFFFFFFu;
    bld.out.nodes.push_back(nd);
    bld.out.max_depth = 1;
    return std::move(bld.out);
  }
  Box root_box; root_box.reset();
  for (uint32_t i = 0; i < n; ++i) root_box.grow(bld.prims[i].box);
  uint32_t m = bld.partition(0, n, root_box, 0);
  if (m == 0) m = n > 1 ? n / 2 : n;  // tiny mesh: still hang it under a root node
  if (n == 1) {
    bld.out.nodes.emplace_back();
    bld.out.max_depth = 1;
    Builder::Child l = bld.build_range(0, 1, 1);
    BvhNode& nd = bld.out.nodes[0];
    bld.store_box(l.box, nd.lo0, nd.hi0);
    for (int k = 0; k < 3; ++k) { nd.lo1[k] = FLT_MAX; nd.hi1[k] = -FLT_MAX; }
    nd.ref0 = l.ref; nd.ref1 = RT_REF_NONE; nd.min_index0 = l.min_index; nd.min_index1 = 0xFFFFFFFFu;
  } else {
    bld.build_inner(0, m, n, 0);
  }
  return std::move(bld.out);
}

QuantizedNodes quantize_nodes(const BvhBuild& build, int format) {
  QuantizedNodes q;
  // extent of every stored (non-empty) child box
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (const BvhNode& n : build.nodes)
    for (int c = 0; c < 2; ++c) {
      if ((c ? n.ref1 : n.ref0) == RT_REF_NONE) continue;
      const float* l = c ? n.lo1 : n.lo0; const float* h = c ? n.hi1 : n.hi0;
      for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], double(l[k])); hi[k] = std::max(hi[k], double(h[k])); }
    }
  double widest = 0.0;
  for (int k = 0; k < 3; ++k) if (hi[k] >= lo[k]) widest = std::max(widest, hi[k] - lo[k]);
  const bool half_planes = format == RT_NODES_SMALL && RT_SMALL_F16;
  double reach = 0.0;
  for (int k = 0; k < 3; ++k) if (hi[k] >= lo[k]) reach = std::max(reach, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
  for (int k = 0; k < 3; ++k) {
    if (half_planes) half_grid_axis(reach, &q.grid_lo[k], &q.grid_step[k]);
    else quant_grid_axis(lo[k], hi[k], widest, &q.grid_lo[k], &q.grid_step[k]);
  }
  const uint32_t w4 = node_words4(format);
  q.words.assign(build.nodes.size() * w4, RtWord4{0, 0, 0, 0});
  for (size_t i = 0; i < build.nodes.size(); ++i) {
    const BvhNode& n = build.nodes[i];
    uint32_t b[12];
    for (int c = 0; c < 2; ++c) {
      const bool none = (c ? n.ref1 : n.ref0) == RT_REF_NONE;
      const float* l = c ? n.lo1 : n.lo0; const float* h = c ? n.hi1 : n.hi0;
      for (int k = 0; k < 3; ++k) {
        if (half_planes) {   // an empty slot's box is inside out
          b[c * 6 + k] = none ? 0x7BFFu : half_plane_lo(double(l[k]), q.grid_step[k]);
          b[c * 6 + 3 + k] = none ? 0xFBFFu : half_plane_hi(double(h[k]), q.grid_step[k]);
        } else {
          b[c * 6 + k] = none ? 65535u : quant_level_lo(double(l[k]), q.grid_lo[k], q.grid_step[k]);
          b[c * 6 + 3 + k] = none ? 0u : quant_level_hi(double(h[k]), q.grid_lo[k], q.grid_step[k]);
        }
      }
    }
    pack_node(&q.words[i * w4], format, b, n.ref0, n.ref1, n.min_index0, n.min_index1);
  }
  return q;
}

}  // namespace rtmi
