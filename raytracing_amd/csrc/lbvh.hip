// lbvh.hip -- the per-mesh BVH built on the GPU: Morton codes of the triangle centroids, a radix sort, the binary
// radix tree of Karras ("Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", 2012), subtrees
// of at most four triangles collapsed into leaves, boxes and minimum file-order indices fitted bottom-up, and the nodes
// written in the quantised form the traversal kernels walk (rt_types.h). The reference has no acceleration structure
// (src/shapes/model_obj.rs:77-87 scans every triangle); like the host's SAH tree this one only prunes work and never
// changes which triangle wins under either hit rule. It is faster to build and slower to traverse than the SAH tree.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <vector>

#include "bvh_quant.h"
#include "lbvh.h"

// the per-triangle normal and length must round exactly as the host's (and the reference's) unfused arithmetic
#pragma clang fp contract(off)

namespace rtmi {
namespace {

constexpr int kThreads = 256;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafSize = 4;       // a subtree of at most this many triangles becomes one leaf

#define LBVH_TRY(expr)                                                  \
  do {                                                                  \
    hipError_t e_ = (expr);                                             \
    if (e_ != hipSuccess) { if (error) *error = std::string(#expr) + ": " + hipGetErrorString(e_); return e_; } \
  } while (0)

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 256)); }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---- 1. bounds of the vertices and of the triangle-box centres: [vlo3, vhi3, clo3, chi3] ----------------------------
__device__ __forceinline__ void tri_box(const double* __restrict__ v, double lo[3], double hi[3]) {
  for (int k = 0; k < 3; ++k) {
    lo[k] = fmin(v[k], fmin(v[3 + k], v[6 + k]));
    hi[k] = fmax(v[k], fmax(v[3 + k], v[6 + k]));
  }
}

__global__ void __launch_bounds__(kThreads) bounds_kernel(const double* __restrict__ verts, uint32_t n, double* __restrict__ partial) {
  double m[12];
  for (int k = 0; k < 3; ++k) { m[k] = INFINITY; m[3 + k] = -INFINITY; m[6 + k] = INFINITY; m[9 + k] = -INFINITY; }
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
    double lo[3], hi[3];
    tri_box(verts + size_t(i) * 9, lo, hi);
    for (int k = 0; k < 3; ++k) {
      const double c = 0.5 * (lo[k] + hi[k]);
      m[k] = fmin(m[k], lo[k]); m[3 + k] = fmax(m[3 + k], hi[k]);
      m[6 + k] = fmin(m[6 + k], c); m[9 + k] = fmax(m[9 + k], c);
    }
  }
  __shared__ double s[kThreads / 64][12];
  for (int k = 0; k < 12; ++k) {
    const bool is_min = (k % 6) < 3;
    double v = m[k];
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(v, off); v = is_min ? fmin(v, o) : fmax(v, o); }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x / 64][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const bool is_min = (threadIdx.x % 6) < 3;
    double v = s[0][threadIdx.x];
    for (int w = 1; w < kThreads / 64; ++w) v = is_min ? fmin(v, s[w][threadIdx.x]) : fmax(v, s[w][threadIdx.x]);
    partial[size_t(blockIdx.x) * 12 + threadIdx.x] = v;
  }
}

__global__ void final_bounds_kernel(const double* __restrict__ partial, uint32_t blocks, double* __restrict__ out) {
  if (threadIdx.x >= 12) return;
  const bool is_min = (threadIdx.x % 6) < 3;
  double v = partial[threadIdx.x];
  for (uint32_t b = 1; b < blocks; ++b) { const double o = partial[size_t(b) * 12 + threadIdx.x]; v = is_min ? fmin(v, o) : fmax(v, o); }
  out[threadIdx.x] = v;
}

// ---- 2. Morton codes: 21 bits per axis of the box centre inside the centre bounds --------------------------------------
__device__ __forceinline__ uint64_t spread21(uint64_t x) {
  x &= 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull;
  x = (x | x << 16) & 0x1F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

struct MortonFrame { double lo[3], scale[3]; };

__global__ void __launch_bounds__(kThreads) morton_kernel(const double* __restrict__ verts, uint32_t n, MortonFrame f,
                                                          uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  double lo[3], hi[3];
  tri_box(verts + size_t(i) * 9, lo, hi);
  uint64_t q[3];
  for (int k = 0; k < 3; ++k) {
    double x = (0.5 * (lo[k] + hi[k]) - f.lo[k]) * f.scale[k];
    x = fmin(fmax(x, 0.0), 2097151.0);
    q[k] = uint64_t(x);
  }
  keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
  vals[i] = i;
}

// ---- 3. the binary radix tree over the sorted codes (Karras 2012, section 4) ------------------------------------------
// length of the common prefix of positions i and j; equal codes are told apart by their positions
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

struct Tree {
  uint32_t* left; uint32_t* right;   // per inner node: child = position | kLeafBit for a single triangle, else an inner node
  uint32_t* parent;                  // [0, n-1): parents of the inner nodes; [n-1, 2n-1): parents of the sorted positions
  uint32_t* first; uint32_t* last;   // per inner node: the sorted positions it covers
};

__global__ void __launch_bounds__(kThreads) radix_tree_kernel(const uint64_t* __restrict__ keys, int n, Tree t) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n - 1) return;
  const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = common_prefix(keys, n, i, i - d);
  int lmax = 2;
  while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int s = lmax >> 1; s >= 1; s >>= 1)
    if (common_prefix(keys, n, i, i + (l + s) * d) > dmin) l += s;
  const int j = i + l * d;
  const int dnode = common_prefix(keys, n, i, j);
  int s = 0, step = l;
  do {
    step = (step + 1) >> 1;
    if (common_prefix(keys, n, i, i + (s + step) * d) > dnode) s += step;
  } while (step > 1);
  const int gamma = i + s * d + (d < 0 ? d : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
  t.left[i] = left_leaf ? uint32_t(gamma) | kLeafBit : uint32_t(gamma);
  t.right[i] = right_leaf ? uint32_t(gamma + 1) | kLeafBit : uint32_t(gamma + 1);
  t.parent[left_leaf ? (n - 1) + gamma : gamma] = uint32_t(i);
  t.parent[right_leaf ? (n - 1) + gamma + 1 : gamma + 1] = uint32_t(i);
  t.first[i] = uint32_t(lo); t.last[i] = uint32_t(hi);
  if (i == 0) t.parent[0] = 0xFFFFFFFFu;
}

// An inner node stays an inner node when it covers more than kLeafSize triangles (the root always stays one); then so
// do all its ancestors. A smaller one becomes a leaf, and everything below it is dropped.
__device__ __forceinline__ bool stays_inner(const Tree& t, uint32_t i) { return i == 0 || t.last[i] - t.first[i] + 1 > kLeafSize; }

__global__ void __launch_bounds__(kThreads) flag_kernel(Tree t, uint32_t n_inner, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n_inner) flags[i] = stays_inner(t, i) ? 1u : 0u;
}

// ---- 4. boxes and minimum indices, bottom-up: the second child to arrive at a node does the node -------------------
struct Fit {
  double* box;          // (2n-1) x 6: inner nodes, then the sorted positions
  uint32_t* min_index;  // (2n-1)
  uint32_t* arrived;    // n-1, zeroed
  uint32_t* depth;      // one word: most inner nodes on a path from the root
  uint32_t n_inner;     // units [0, n_inner) are inner nodes, [n_inner, 2 n_inner + 1) the sorted positions
};

__device__ __forceinline__ void climb(const Tree& t, const Fit& f, uint32_t unit, double b[6], uint32_t mi) {
  for (int k = 0; k < 6; ++k) f.box[size_t(unit) * 6 + k] = b[k];
  f.min_index[unit] = mi;
  uint32_t node = t.parent[unit];
  for (;;) {
    // release what this thread wrote, and if it is the second to arrive, acquire what the first one wrote
    if (__hip_atomic_fetch_add(&f.arrived[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const uint32_t children[2] = {t.left[node], t.right[node]};
    double u[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t m = 0xFFFFFFFFu;
    for (int c = 0; c < 2; ++c) {
      const uint32_t ref = children[c];
      // a child that is a dropped inner node or a dropped position never arrives: only live units climb, and a live
      // inner node's two children are both live units (or the collapsed node itself climbed in their place)
      const uint32_t child = (ref & kLeafBit) ? f.n_inner + (ref & ~kLeafBit) : ref;
      const double* cb = f.box + size_t(child) * 6;
      for (int k = 0; k < 3; ++k) { u[k] = fmin(u[k], cb[k]); u[3 + k] = fmax(u[3 + k], cb[3 + k]); }
      const uint32_t cm = f.min_index[child];
      m = cm < m ? cm : m;
    }
    for (int k = 0; k < 6; ++k) f.box[size_t(node) * 6 + k] = u[k];
    f.min_index[node] = m;
    if (node == 0) return;
    node = t.parent[node];
  }
}

// One thread per unit that hangs directly under a remaining inner node: a collapsed inner node (2..4 triangles) or a
// single sorted position.
__global__ void __launch_bounds__(kThreads) fit_kernel(const double* __restrict__ verts, const uint32_t* __restrict__ order, uint32_t n,
                                                        Tree t, Fit f) {
  const uint32_t id = blockIdx.x * kThreads + threadIdx.x;
  if (id >= 2 * n - 1) return;
  const bool is_position = id >= n - 1;
  uint32_t first, last;
  if (is_position) { first = last = id - (n - 1); }
  else { if (stays_inner(t, id)) return; first = t.first[id]; last = t.last[id]; }
  const uint32_t parent = t.parent[id];
  if (!stays_inner(t, parent)) return;     // somewhere inside a collapsed subtree
  double b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
  uint32_t mi = 0xFFFFFFFFu;
  for (uint32_t p = first; p <= last; ++p) {
    const uint32_t tri = order[p];
    double lo[3], hi[3];
    tri_box(verts + size_t(tri) * 9, lo, hi);
    for (int k = 0; k < 3; ++k) { b[k] = fmin(b[k], lo[k]); b[3 + k] = fmax(b[3 + k], hi[k]); }
    mi = tri < mi ? tri : mi;
  }
  // inner nodes on the way to the root (the stack the traversal needs)
  uint32_t depth = 0;
  for (uint32_t a = parent; a != 0xFFFFFFFFu; a = t.parent[a]) ++depth;
  atomicMax(&f.depth[0], depth);
  climb(t, f, id, b, mi);
}

// ---- 5. the nodes in the device's form ----------------------------------------------------------------------------------
struct Frame { double center[3]; double pad; float grid_lo[3], grid_step[3]; };

__global__ void __launch_bounds__(kThreads) emit_nodes_kernel(Tree t, Fit f, const uint32_t* __restrict__ compact, uint32_t n, Frame fr,
                                                               int format, RtWord4* __restrict__ words) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n - 1 || !stays_inner(t, i)) return;
  const uint32_t w4 = format == RT_NODES_WIDE ? 3u : 2u;
  uint32_t levels[12], ref[2], mins[2];
  const uint32_t child_ref[2] = {t.left[i], t.right[i]};
  for (int c = 0; c < 2; ++c) {
    const uint32_t r = child_ref[c];
    uint32_t unit;
    if (r & kLeafBit) { unit = (n - 1) + (r & ~kLeafBit); ref[c] = RT_LEAF_FLAG | (r & ~kLeafBit) << 4 | 1u; }
    else if (stays_inner(t, r)) { unit = r; ref[c] = compact[r]; }
    else { unit = r; ref[c] = RT_LEAF_FLAG | t.first[r] << 4 | (t.last[r] - t.first[r] + 1u); }
    mins[c] = f.min_index[unit];
    for (int k = 0; k < 3; ++k) {
      // binary32 box relative to the mesh centre that contains the binary64 one: pad, then round outward
      const float lo = __double2float_rd((f.box[size_t(unit) * 6 + k] - fr.center[k]) - fr.pad);
      const float hi = __double2float_ru((f.box[size_t(unit) * 6 + 3 + k] - fr.center[k]) + fr.pad);
      if (format == RT_NODES_SMALL && RT_SMALL_F16) {
        levels[c * 6 + k] = half_plane_lo(double(lo), fr.grid_step[k]);
        levels[c * 6 + 3 + k] = half_plane_hi(double(hi), fr.grid_step[k]);
      } else {
        levels[c * 6 + k] = quant_level_lo(double(lo), fr.grid_lo[k], fr.grid_step[k]);
        levels[c * 6 + 3 + k] = quant_level_hi(double(hi), fr.grid_lo[k], fr.grid_step[k]);
      }
    }
  }
  pack_node(words + size_t(compact[i]) * w4, format, levels, ref[0], ref[1], mins[0], mins[1]);
}

// ---- 6. the triangles in leaf order, each leaf sorted by file-order index, with their normal records --------------
__device__ __forceinline__ DevTriFrame triangle_frame(const double* v) {
  // the host's tri_frame, operation for operation (src/shapes/triangle.rs:30,83,91-92)
  const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
  const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  const double cl = sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
  const double n0[3] = {c[0] / cl, c[1] / cl, c[2] / cl};
  const double n1[3] = {(1.0 * n0[0] + 0.0 * n0[1]) + 0.0 * n0[2], (0.0 * n0[0] + 1.0 * n0[1]) + 0.0 * n0[2],
                        (0.0 * n0[0] + 0.0 * n0[1]) + 1.0 * n0[2]};
  DevTriFrame f;
  f.len = sqrt((n1[0] * n1[0] + n1[1] * n1[1]) + n1[2] * n1[2]);
  for (int k = 0; k < 3; ++k) f.n[k] = n1[k] / f.len;
  return f;
}

__global__ void __launch_bounds__(kThreads) emit_triangles_kernel(const double* __restrict__ verts, const uint32_t* __restrict__ order,
                                                                   uint32_t n, Tree t, DevMeshTri* __restrict__ tris,
                                                                   DevTriFrame* __restrict__ frames) {
  const uint32_t id = blockIdx.x * kThreads + threadIdx.x;
  if (id >= 2 * n - 1) return;
  uint32_t first, last;
  if (id >= n - 1) { first = last = id - (n - 1); }
  else { if (stays_inner(t, id)) return; first = t.first[id]; last = t.last[id]; }
  if (!stays_inner(t, t.parent[id])) return;
  uint32_t idx[kLeafSize];
  const uint32_t count = last - first + 1;
  for (uint32_t k = 0; k < count; ++k) idx[k] = order[first + k];
  for (uint32_t a = 1; a < count; ++a)           // a leaf is tested in file order
    for (uint32_t b = a; b > 0 && idx[b] < idx[b - 1]; --b) { const uint32_t x = idx[b]; idx[b] = idx[b - 1]; idx[b - 1] = x; }
  for (uint32_t k = 0; k < count; ++k) {
    DevMeshTri out;
    for (int c = 0; c < 9; ++c) out.v[c] = verts[size_t(idx[k]) * 9 + c];
    out.index = idx[k]; out.pad = 0;
    tris[first + k] = out;
    frames[first + k] = triangle_frame(out.v);
  }
}

}  // namespace

hipError_t build_lbvh(const double* host_vertices, uint32_t n, int format, LbvhOut* out, std::string* error) {
  if (n < 2) { if (error) *error = "the device builder needs at least two triangles"; return hipErrorInvalidValue; }
  const auto t0 = std::chrono::steady_clock::now();
  const uint32_t n_inner = n - 1, n_units = 2 * n - 1;
  const uint32_t blocks_n = (n + kThreads - 1) / kThreads, blocks_units = (n_units + kThreads - 1) / kThreads;
  const uint32_t red_blocks = std::min<uint32_t>(blocks_n, 1024);
  Buf verts, partial, bounds, keys_a, keys_b, vals_a, vals_b, left, right, parent, first, last, flags, compact, box, min_index, arrived, depth, temp;
  LBVH_TRY(verts.alloc(size_t(n) * 72));
  LBVH_TRY(hipMemcpy(verts.p, host_vertices, size_t(n) * 72, hipMemcpyHostToDevice));
  LBVH_TRY(partial.alloc(size_t(red_blocks) * 12 * 8));
  LBVH_TRY(bounds.alloc(12 * 8));
  hipLaunchKernelGGL(bounds_kernel, dim3(red_blocks), kThreads, 0, nullptr, verts.as<double>(), n, partial.as<double>());
  hipLaunchKernelGGL(final_bounds_kernel, dim3(1), 64, 0, nullptr, partial.as<double>(), red_blocks, bounds.as<double>());
  double b[12];
  LBVH_TRY(hipMemcpy(b, bounds.p, sizeof b, hipMemcpyDeviceToHost));

  // the mesh's centre, bounding radius, padded binary32 box and quantisation grid: the host builder's formulas
  double extent = 0.0, r2 = 0.0;
  Frame fr{};
  for (int k = 0; k < 3; ++k) {
    out->center[k] = fr.center[k] = 0.5 * (b[k] + b[3 + k]);
    const double half = 0.5 * (b[3 + k] - b[k]);
    extent = std::max(extent, half);
    r2 += half * half;
  }
  out->radius = std::sqrt(r2) * (1.0 + 0x1.0p-20);
  fr.pad = extent * 0x1.0p-20 + 1e-30;
  double widest = 0.0, flo[3], fhi[3];
  for (int k = 0; k < 3; ++k) {
    const double dl = (b[k] - fr.center[k]) - fr.pad, dh = (b[3 + k] - fr.center[k]) + fr.pad;
    float l = float(dl), h = float(dh);
    if (double(l) > dl) l = std::nextafterf(l, -FLT_MAX);
    if (double(h) < dh) h = std::nextafterf(h, FLT_MAX);
    out->box_lo[k] = l; out->box_hi[k] = h;
    flo[k] = double(l); fhi[k] = double(h);
    widest = std::max(widest, fhi[k] - flo[k]);
  }
  double reach = 0.0;
  for (int k = 0; k < 3; ++k) reach = std::max(reach, std::max(std::fabs(flo[k]), std::fabs(fhi[k])));
  for (int k = 0; k < 3; ++k) {
    if (format == RT_NODES_SMALL && RT_SMALL_F16) half_grid_axis(reach, &out->grid_lo[k], &out->grid_step[k]);
    else quant_grid_axis(flo[k], fhi[k], widest, &out->grid_lo[k], &out->grid_step[k]);
    fr.grid_lo[k] = out->grid_lo[k]; fr.grid_step[k] = out->grid_step[k];
  }

  MortonFrame mf;
  for (int k = 0; k < 3; ++k) {
    const double ext = b[9 + k] - b[6 + k];
    mf.lo[k] = b[6 + k];
    mf.scale[k] = ext > 0.0 ? 2097152.0 / ext : 0.0;
  }
  LBVH_TRY(keys_a.alloc(size_t(n) * 8)); LBVH_TRY(keys_b.alloc(size_t(n) * 8));
  LBVH_TRY(vals_a.alloc(size_t(n) * 4)); LBVH_TRY(vals_b.alloc(size_t(n) * 4));
  hipLaunchKernelGGL(morton_kernel, dim3(blocks_n), kThreads, 0, nullptr, verts.as<double>(), n, mf, keys_a.as<uint64_t>(), vals_a.as<uint32_t>());
  size_t temp_bytes = 0;
  LBVH_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(),
                                     vals_b.as<uint32_t>(), n, 0, 63, nullptr));
  size_t scan_bytes = 0;
  LBVH_TRY(flags.alloc(size_t(n_inner) * 4)); LBVH_TRY(compact.alloc(size_t(n_inner) * 4));
  LBVH_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags.as<uint32_t>(), compact.as<uint32_t>(), 0u, n_inner, rocprim::plus<uint32_t>(), nullptr));
  LBVH_TRY(temp.alloc(std::max(temp_bytes, scan_bytes)));
  LBVH_TRY(rocprim::radix_sort_pairs(temp.p, temp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<uint32_t>(),
                                     vals_b.as<uint32_t>(), n, 0, 63, nullptr));
  const uint64_t* keys = keys_b.as<uint64_t>();
  const uint32_t* order = vals_b.as<uint32_t>();

  Tree t;
  LBVH_TRY(left.alloc(size_t(n_inner) * 4)); LBVH_TRY(right.alloc(size_t(n_inner) * 4)); LBVH_TRY(parent.alloc(size_t(n_units) * 4));
  LBVH_TRY(first.alloc(size_t(n_inner) * 4)); LBVH_TRY(last.alloc(size_t(n_inner) * 4));
  t.left = left.as<uint32_t>(); t.right = right.as<uint32_t>(); t.parent = parent.as<uint32_t>();
  t.first = first.as<uint32_t>(); t.last = last.as<uint32_t>();
  hipLaunchKernelGGL(radix_tree_kernel, dim3((n_inner + kThreads - 1) / kThreads), kThreads, 0, nullptr, keys, int(n), t);
  hipLaunchKernelGGL(flag_kernel, dim3((n_inner + kThreads - 1) / kThreads), kThreads, 0, nullptr, t, n_inner, flags.as<uint32_t>());
  LBVH_TRY(rocprim::exclusive_scan(temp.p, scan_bytes, flags.as<uint32_t>(), compact.as<uint32_t>(), 0u, n_inner, rocprim::plus<uint32_t>(), nullptr));

  Fit f;
  LBVH_TRY(box.alloc(size_t(n_units) * 48)); LBVH_TRY(min_index.alloc(size_t(n_units) * 4));
  LBVH_TRY(arrived.alloc(size_t(n_inner) * 4)); LBVH_TRY(depth.alloc(4));
  f.box = box.as<double>(); f.min_index = min_index.as<uint32_t>(); f.arrived = arrived.as<uint32_t>(); f.depth = depth.as<uint32_t>();
  f.n_inner = n_inner;
  LBVH_TRY(hipMemsetAsync(arrived.p, 0, size_t(n_inner) * 4, nullptr));
  LBVH_TRY(hipMemsetAsync(depth.p, 0, 4, nullptr));
  hipLaunchKernelGGL(fit_kernel, dim3(blocks_units), kThreads, 0, nullptr, verts.as<double>(), order, n, t, f);

  // how many inner nodes stay: the scan's last value plus the last flag
  uint32_t tail[2] = {0, 0};
  LBVH_TRY(hipMemcpy(&tail[0], compact.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost));
  LBVH_TRY(hipMemcpy(&tail[1], flags.as<uint32_t>() + (n_inner - 1), 4, hipMemcpyDeviceToHost));
  const uint32_t n_nodes = tail[0] + tail[1];
  const uint32_t w4 = format == RT_NODES_WIDE ? 3u : 2u;
  Buf words, tris, frames;
  LBVH_TRY(words.alloc(size_t(n_nodes) * w4 * sizeof(RtWord4)));
  LBVH_TRY(tris.alloc(size_t(n) * sizeof(DevMeshTri))); LBVH_TRY(frames.alloc(size_t(n) * sizeof(DevTriFrame)));
  hipLaunchKernelGGL(emit_nodes_kernel, dim3((n_inner + kThreads - 1) / kThreads), kThreads, 0, nullptr, t, f, compact.as<uint32_t>(), n, fr,
                     format, words.as<RtWord4>());
  hipLaunchKernelGGL(emit_triangles_kernel, dim3(blocks_units), kThreads, 0, nullptr, verts.as<double>(), order, n, t, tris.as<DevMeshTri>(),
                     frames.as<DevTriFrame>());
  LBVH_TRY(hipGetLastError());
  LBVH_TRY(hipDeviceSynchronize());
  out->device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  uint32_t depth_out = 0;
  LBVH_TRY(hipMemcpy(&depth_out, depth.p, sizeof depth_out, hipMemcpyDeviceToHost));
  out->max_depth = std::max(1u, depth_out);
  out->n_nodes = n_nodes;
  out->words.resize(size_t(n_nodes) * w4);
  out->tris.resize(n); out->frames.resize(n);
  LBVH_TRY(hipMemcpy(out->words.data(), words.p, out->words.size() * sizeof(RtWord4), hipMemcpyDeviceToHost));
  LBVH_TRY(hipMemcpy(out->tris.data(), tris.p, size_t(n) * sizeof(DevMeshTri), hipMemcpyDeviceToHost));
  LBVH_TRY(hipMemcpy(out->frames.data(), frames.p, size_t(n) * sizeof(DevTriFrame), hipMemcpyDeviceToHost));
  return hipSuccess;
}

}  // namespace rtmi
