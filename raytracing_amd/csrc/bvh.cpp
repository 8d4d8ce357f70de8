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


namespace {

struct QuadChild { const float* lo; const float* hi; uint32_t ref; };   // ref in the binary tree's terms

float half_area32(const float* lo, const float* hi) {
  const double dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
  return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : float(dx * dy + dy * dz + dz * dx);
}

struct QuadBuilder {
  const BvhBuild& in;
  struct Node { QuadChild child[4]; uint32_t n = 0; uint32_t quad_ref[4]; };
  std::vector<Node> nodes;
  uint32_t max_depth = 0;
  uint32_t max_pending = 0;    // most references a walk can have stacked at once: the siblings waiting along one path

  // the four-wide node that takes the place of binary node `id`; returns its index. `pending` = references already
  // stacked when a walk arrives here, if every node above had all its other children waiting.
  uint32_t emit(uint32_t id, uint32_t depth, uint32_t pending = 0) {
    const uint32_t me = uint32_t(nodes.size());
    nodes.emplace_back();
    max_depth = std::max(max_depth, depth + 1);
    QuadChild c[4];
    uint32_t n = 0;
    const BvhNode& b = in.nodes[id];
    if (b.ref0 != RT_REF_NONE) c[n++] = QuadChild{b.lo0, b.hi0, b.ref0};
    if (b.ref1 != RT_REF_NONE) c[n++] = QuadChild{b.lo1, b.hi1, b.ref1};
    while (n < 4) {
      int widest = -1; float area = -1.0f;
      for (uint32_t i = 0; i < n; ++i)
        if (!(c[i].ref & RT_LEAF_FLAG)) { const float a = half_area32(c[i].lo, c[i].hi); if (a > area) { area = a; widest = int(i); } }
      if (widest < 0) break;
      const BvhNode& g = in.nodes[c[widest].ref];
      const bool has0 = g.ref0 != RT_REF_NONE, has1 = g.ref1 != RT_REF_NONE;
      if (has0 && has1) { c[widest] = QuadChild{g.lo0, g.hi0, g.ref0}; c[n++] = QuadChild{g.lo1, g.hi1, g.ref1}; }
      else if (has0) c[widest] = QuadChild{g.lo0, g.hi0, g.ref0};
      else if (has1) c[widest] = QuadChild{g.lo1, g.hi1, g.ref1};
      else { c[widest] = c[n - 1]; --n; }          // an inner node without children: nothing below it
    }
    max_pending = std::max(max_pending, pending + (n ? n - 1 : 0));
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t ref = (c[i].ref & RT_LEAF_FLAG) ? c[i].ref : emit(c[i].ref, depth + 1, pending + n - 1);
      nodes[me].child[i] = c[i];
      nodes[me].quad_ref[i] = ref;
    }
    nodes[me].n = n;
    return me;
  }
};

}  // namespace

QuadNodes quantize_nodes_quad(const BvhBuild& build) {
  QuadBuilder qb{build, {}, 0, 0};
  qb.nodes.reserve(build.nodes.size() / 2 + 2);
  qb.emit(0, 0);
  QuadNodes q;
  q.n_nodes = uint32_t(qb.nodes.size());
  q.max_depth = qb.max_depth;
  q.max_stack = qb.max_pending + 1;
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (const QuadBuilder::Node& n : qb.nodes)
    for (uint32_t c = 0; c < n.n; ++c)
      for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], double(n.child[c].lo[k])); hi[k] = std::max(hi[k], double(n.child[c].hi[k])); }
  double widest = 0.0;
  for (int k = 0; k < 3; ++k) if (hi[k] >= lo[k]) widest = std::max(widest, hi[k] - lo[k]);
  for (int k = 0; k < 3; ++k) quant_grid_axis(lo[k], hi[k], widest, &q.grid_lo[k], &q.grid_step[k]);
  q.words.assign(qb.nodes.size() * 4, RtWord4{0, 0, 0, 0});
  for (size_t i = 0; i < qb.nodes.size(); ++i) {
    const QuadBuilder::Node& n = qb.nodes[i];
    uint32_t level[24], ref[4];
    for (uint32_t c = 0; c < 4; ++c) {
      const bool none = c >= n.n;
      ref[c] = none ? RT_REF_NONE : n.quad_ref[c];
      for (int k = 0; k < 3; ++k) {   // an unused slot's box is inside out
        level[c * 6 + k] = none ? 65535u : quant_level_lo(double(n.child[c].lo[k]), q.grid_lo[k], q.grid_step[k]);
        level[c * 6 + 3 + k] = none ? 0u : quant_level_hi(double(n.child[c].hi[k]), q.grid_lo[k], q.grid_step[k]);
      }
    }
    uint32_t w[16];
    for (int j = 0; j < 12; ++j) w[j] = level[2 * j] | level[2 * j + 1] << 16;
    for (int c = 0; c < 4; ++c) w[12 + c] = ref[c];
    for (int j = 0; j < 4; ++j) q.words[i * 4 + j] = RtWord4{w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]};
  }
  return q;
}


bool verify_quad_nodes(const BvhBuild& build, const QuadNodes& q) {
  if (q.words.size() != size_t(q.n_nodes) * 4 || q.n_nodes == 0) return false;
  std::vector<uint8_t> seen_tri(build.tris.size(), 0), seen_node(q.n_nodes, 0);
  seen_node[0] = 1;
  for (uint32_t i = 0; i < q.n_nodes; ++i) {
    const RtWord4& r = q.words[size_t(i) * 4 + 3];
    const uint32_t ref[4] = {r.x, r.y, r.z, r.w};
    for (uint32_t c = 0; c < 4; ++c) {
      if (ref[c] == RT_REF_NONE) continue;
      if (ref[c] & RT_LEAF_FLAG) {
        const uint32_t first = (ref[c] & ~RT_LEAF_FLAG) >> 4, count = ref[c] & 15u;
        if (count == 0 || size_t(first) + count > build.tris.size()) return false;
        for (uint32_t t = first; t < first + count; ++t) { if (seen_tri[t]) return false; seen_tri[t] = 1; }
      } else {
        if (ref[c] >= q.n_nodes || ref[c] <= i || seen_node[ref[c]]) return false;   // children come after their parent, once
        seen_node[ref[c]] = 1;
      }
    }
  }
  for (uint8_t v : seen_tri) if (!v) return false;
  for (uint8_t v : seen_node) if (!v) return false;
  return true;
}


namespace {

// Host restatement of what the kernels do with a four-wide node (rt_kernels.hip: make_box_ray, make_quant_ray,
// test_children4), for the self-check below: binary32 arithmetic with the same operations, no upper bound on t.
struct HostQuantRay { float sx, sy, sz, lx, ly, lz, hx, hy, hz; };

HostQuantRay host_quant_ray(const double* o, const double* d, const double* center, double radius, const float* grid_lo, const float* grid_step) {
  const float ox = float(o[0] - center[0]), oy = float(o[1] - center[1]), oz = float(o[2] - center[2]);
  const float delta = 0x1.0p-20f * (std::fmax(std::fmax(std::fabs(ox), std::fabs(oy)), std::fabs(oz)) + float(radius) * 1.0000005f);
  const float ix = std::fmin(std::fmax(1.0f / float(d[0]), -0x1.0p40f), 0x1.0p40f), iy = std::fmin(std::fmax(1.0f / float(d[1]), -0x1.0p40f), 0x1.0p40f),
              iz = std::fmin(std::fmax(1.0f / float(d[2]), -0x1.0p40f), 0x1.0p40f);
  HostQuantRay q;
  q.sx = grid_step[0] * ix; q.sy = grid_step[1] * iy; q.sz = grid_step[2] * iz;
  q.lx = (grid_lo[0] - (ox + delta)) * ix; q.hx = (grid_lo[0] - (ox - delta)) * ix;
  q.ly = (grid_lo[1] - (oy + delta)) * iy; q.hy = (grid_lo[1] - (oy - delta)) * iy;
  q.lz = (grid_lo[2] - (oz + delta)) * iz; q.hz = (grid_lo[2] - (oz - delta)) * iz;
  return q;
}

bool host_moller_trumbore(const double* v, const double* o, const double* d) {
  const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
  const double tv[3] = {o[0] - v[0], o[1] - v[1], o[2] - v[2]};
  const double p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
  const double q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
  const double det = (p[0] * e1[0] + p[1] * e1[1]) + p[2] * e1[2];
  if (std::fabs(det) < 1e-10) return false;
  const double inv = 1.0 / det;
  const double t = inv * ((q[0] * e2[0] + q[1] * e2[1]) + q[2] * e2[2]);
  if (t < 0.0) return false;
  const double u = inv * ((p[0] * tv[0] + p[1] * tv[1]) + p[2] * tv[2]);
  if (!(u >= 0.0 && u <= 1.0)) return false;
  const double w = inv * ((q[0] * d[0] + q[1] * d[1]) + q[2] * d[2]);
  return !(w < 0.0 || u + w > 1.0);
}

}  // namespace

}  // namespace rtmi

// Host-only self-check of the four-wide tree (no GPU involved; bound by the CPU tests through ctypes, not part of the
// public ABI): builds the binary tree and its four-wide form for `n` triangles, checks the structure, then casts
// `n_rays` pseudo-random rays and verifies that the walk the kernels do over the quantised boxes reaches every triangle
// the exact binary64 test accepts. out[0..5] = four-wide nodes, their depth, binary nodes, binary depth, triangles that
// a walk failed to reach (must be 0), box tests per ray (rounded).
extern "C" int rtmi_internal_quad_tree_self_check(const double* tri_vertices, uint32_t n, uint32_t n_rays, uint64_t seed, uint64_t* out) {
  using namespace rtmi;
  if (!tri_vertices || !out) return -1;
  const BvhBuild bb = build_bvh(tri_vertices, n, 4);
  const QuadNodes q = quantize_nodes_quad(bb);
  if (!verify_quad_nodes(bb, q)) return -7;
  out[0] = q.n_nodes; out[1] = q.max_depth; out[2] = bb.nodes.size(); out[3] = bb.max_depth; out[4] = 0; out[5] = 0;
  uint64_t state = seed * 0x9E3779B97F4A7C15ull + 1, steps = 0, max_stack = 0;
  auto uniform = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return double(state >> 11) * 0x1.0p-53; };
  std::vector<uint8_t> reached(bb.tris.size());
  std::vector<uint32_t> stack;
  for (uint32_t r = 0; r < n_rays; ++r) {
    // from a point on a sphere around the mesh towards a point inside it; every eighth ray runs along a coordinate axis
    double o[3], t[3], d[3];
    double len = 0.0;
    do { len = 0.0; for (int k = 0; k < 3; ++k) { o[k] = 2.0 * uniform() - 1.0; len += o[k] * o[k]; } } while (len < 1e-4 || len > 1.0);
    len = std::sqrt(len);
    for (int k = 0; k < 3; ++k) {
      o[k] = bb.center[k] + 2.5 * bb.radius * o[k] / len;
      t[k] = bb.center[k] + bb.radius * 0.6 * (2.0 * uniform() - 1.0);
      d[k] = t[k] - o[k];
    }
    if (r % 8 == 7) { const int a = int(r / 8 % 3); for (int k = 0; k < 3; ++k) if (k != a) { o[k] = t[k]; d[k] = 0.0; } }
    len = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) d[k] /= len;
    const HostQuantRay qr = host_quant_ray(o, d, bb.center, bb.radius, q.grid_lo, q.grid_step);
    std::fill(reached.begin(), reached.end(), uint8_t(0));
    stack.clear();
    stack.push_back(0);
    while (!stack.empty()) {
      const uint32_t ref = stack.back(); stack.pop_back();
      if (ref & RT_LEAF_FLAG) {
        const uint32_t first = (ref & ~RT_LEAF_FLAG) >> 4, count = ref & 15u;
        for (uint32_t i = first; i < first + count; ++i) reached[i] = 1;
        continue;
      }
      ++steps;
      const RtWord4* w4 = &q.words[size_t(ref) * 4];
      const uint32_t w[12] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w, w4[2].x, w4[2].y, w4[2].z, w4[2].w};
      const uint32_t refs[4] = {w4[3].x, w4[3].y, w4[3].z, w4[3].w};
      for (int c = 0; c < 4; ++c) {
        const uint32_t a = w[3 * c], b = w[3 * c + 1], e = w[3 * c + 2];
        const float ax = std::fmaf(float(a & 0xFFFFu), qr.sx, qr.lx), bx = std::fmaf(float(b >> 16), qr.sx, qr.hx);
        const float ay = std::fmaf(float(a >> 16), qr.sy, qr.ly), by = std::fmaf(float(e & 0xFFFFu), qr.sy, qr.hy);
        const float az = std::fmaf(float(b & 0xFFFFu), qr.sz, qr.lz), bz = std::fmaf(float(e >> 16), qr.sz, qr.hz);
        const float tn = std::fmax(std::fmax(std::fmin(ax, bx), std::fmin(ay, by)), std::fmax(std::fmin(az, bz), 0.0f));
        const float tf = std::fmin(std::fmin(std::fmax(ax, bx), std::fmax(ay, by)), std::fmax(az, bz));
        if (tn <= tf * 1.000002f && refs[c] != RT_REF_NONE) stack.push_back(refs[c]);
      }
      max_stack = std::max<uint64_t>(max_stack, stack.size());
    }
    for (size_t i = 0; i < bb.tris.size(); ++i)
      if (!reached[i] && host_moller_trumbore(bb.tris[i].v, o, d)) ++out[4];
  }
  out[5] = n_rays ? steps / n_rays : 0;
  out[1] |= uint64_t(q.max_stack) << 32;
  if (max_stack > q.max_stack) return -7;                // what the kernels size their stacks by
  return 0;
}
