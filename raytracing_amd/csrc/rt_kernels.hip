// rt_kernels.hip -- gfx950 kernels of the path-tracing hot path.
//
// Wavefront path tracing, no megakernel and no recursion. Scenes without meshes run one fused kernel per
// bounce (bounce_kernel: closest hit, shading, append of the paths that continue). Scenes with meshes run two
// per bounce: shade_kernel (dense: exact sphere / triangle tests, shading, append) and traverse_kernel
// (persistent waves that walk the BVHs for the mesh-bound rays only and refill their idle lanes). Paths are
// appended with one wave-wide ballot and one atomic per wave. Geometry arithmetic is binary64 like the
// reference's; spectra are binary32. A path does not carry a 60-bin throughput: it records, per hit, which
// spectrum-table row it picked up (one byte), and the 60-bin product is formed once, when the path reaches an
// emitter, in the reference's multiplication order.
//
// Citations are file:line in the reference repository.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>

#include "rt_kernels.h"

// Products and sums stay separately rounded in this file, as in the reference's (unfused)
// binary64/binary32 arithmetic: a ray through an edge or a vertex, a grazing sphere hit and
// every 60-bin product then decide and round exactly as the reference does. Fused multiply-adds
// are written out only in operator/(V3, double), whose refinement steps are those of the correctly
// rounded division itself.
#pragma clang fp contract(off)

namespace rtmi {

// 60-bin CIE curves, to_XYZ scale and Smits reflectance bases (src/spectrum/mod.rs:268-279)
__constant__ float c_cie[3][RT_NBINS];
__constant__ float c_smits_refl[7][RT_NBINS];
__constant__ float c_xyz_scale;

hipError_t upload_constants(const SpectrumTables& t) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_cie), t.cie, sizeof t.cie);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_smits_refl), t.smits[0], sizeof t.smits[0]);
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_xyz_scale), &t.xyz_scale, sizeof(float));
  return e;
}

namespace {

// ------------------------------------------------------------------------------- vectors

struct V3 { double x, y, z; };

__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
// a / s, three IEEE quotients. The compiler expands each binary64 division into a reciprocal estimate, two
// Newton steps on it and one correction of the quotient; here the reciprocal and its refinement are shared by
// the three, which yields the same correctly rounded results in 14 instructions fewer per division. Operands
// far from 1 (where the expansion rescales its inputs) take the plain divisions.
__device__ __forceinline__ V3 operator/(V3 a, double s) {
  const double big = fmax(fmax(fabs(a.x), fabs(a.y)), fmax(fabs(a.z), fabs(s)));
  const double as = fabs(s);
  if (as > 0x1.0p-200 && big < 0x1.0p200) {
    double r = __builtin_amdgcn_rcp(s);
    double e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-s, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double qx = a.x * r, qy = a.y * r, qz = a.z * r;
    // the sign is taken from the first estimate: for a zero numerator the correction step would yield +0 where the
    // IEEE quotient of -0 is -0
    return mk(__builtin_copysign(__builtin_fma(__builtin_fma(-s, qx, a.x), r, qx), qx),
              __builtin_copysign(__builtin_fma(__builtin_fma(-s, qy, a.y), r, qy), qy),
              __builtin_copysign(__builtin_fma(__builtin_fma(-s, qz, a.z), r, qz), qz));
  }
  return mk(a.x / s, a.y / s, a.z / s);
}
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 normalize(V3 a) { return a / sqrt(dot(a, a)); }  // Ray::new, src/geometry/ray.rs:16
__device__ __forceinline__ V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }

// row-major 3x4 affine map applied to a point / a vector / transposed linear part to a vector
__device__ __forceinline__ V3 xf_vec(const double* m, V3 v) {
  return mk((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
            (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
__device__ __forceinline__ V3 xf_pt(const double* m, V3 p) {
  V3 r = xf_vec(m, p);
  return mk(r.x + m[3], r.y + m[7], r.z + m[11]);
}
__device__ __forceinline__ V3 xf_vec_t(const double* m, V3 v) {
  return mk((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
            (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}

// A record of one of the scene's small tables (objects, spheres, lone triangles) at an index every lane of the wave
// agrees on, fetched through the constant address space: the compiler then issues scalar loads (one fetch per wave
// through the scalar cache, results in scalar registers) where it would otherwise issue a vector load in which all
// 64 lanes ask for the same address -- it cannot prove by itself that nothing in the kernel writes these tables --
// and the fetch no longer queues behind the wave's outstanding vector-memory operations.
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
  static_assert(sizeof(T) % 4 == 0 && __is_trivially_copyable(T), "copied word by word");
  typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
  const ConstWords src = (ConstWords)(const void*)p;
  union Words { T v; uint32_t w[sizeof(T) / 4]; __device__ Words() {} } u;
#pragma unroll
  // the words nobody reads are never fetched; the first-lane read pins a word to a scalar register, so that the fetch
  // stays a scalar load also where every use of the word is a vector instruction
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.w[k] = uint32_t(__builtin_amdgcn_readfirstlane(int(src[k])));
  return u.v;
}

// ------------------------------------------------------------------------------- Philox4x32-10

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of separate high and low halves
    const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
    const uint32_t hi0 = uint32_t(p0 >> 32), lo0 = uint32_t(p0), hi1 = uint32_t(p1 >> 32), lo1 = uint32_t(p1);
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two uniforms in [0,1) with 53 random bits each, counter = (pixel, sample, bounce, stream)
__device__ __forceinline__ void uniform_pair(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce,
                                             uint32_t stream, double& u0, double& u1) {
  uint32_t r[4];
  philox4x32_10(pixel, sample, bounce, stream, uint32_t(seed), uint32_t(seed >> 32), r);
  const uint64_t a = (uint64_t(r[1]) << 32) | r[0], b = (uint64_t(r[3]) << 32) | r[2];
  u0 = double(a >> 11) * 0x1.0p-53;
  u1 = double(b >> 11) * 0x1.0p-53;
}

// ------------------------------------------------------------------------------- primitives

// Moller-Trumbore, two-sided, inclusive edges, t >= 0 (src/geometry/mod.rs:16-80; same
// rejection order and thresholds).
__device__ __forceinline__ bool moller_trumbore(V3 v0, V3 v1, V3 v2, V3 o, V3 d, double& t_out) {
  const V3 e1 = v1 - v0, e2 = v2 - v0, tv = o - v0;
  const V3 p = cross(d, e2), q = cross(tv, e1);
  const double det = dot(p, e1);
  if (fabs(det) < 1e-10) return false;
  const double inv = 1.0 / det;
  const double t = inv * dot(q, e2);
  if (t < 0.0) return false;
  const double u = inv * dot(p, tv);
  if (!(u >= 0.0 && u <= 1.0)) return false;
  const double w = inv * dot(q, d);
  if (w < 0.0 || u + w > 1.0) return false;
  t_out = t;
  return true;
}

// Triangle::intersect under the identity map every loader uses (src/shapes/triangle.rs:73-97).
// `dn` is the ray direction after the reference's re-normalisation of the "local" ray
// (src/geometry/ray.rs:45). Returns Moller-Trumbore's t (for the hit point) and the t the
// reference reports, t / |n| with n the flat unit normal (src/shapes/triangle.rs:30,92); |n| is a
// function of the vertices alone and comes precomputed (DevTriFrame::len).
__device__ __forceinline__ bool triangle_hit(V3 v0, V3 v1, V3 v2, V3 o, V3 dn, double len, double& t_raw, double& t_reported) {
  double t;
  if (!moller_trumbore(v0, v1, v2, o, dn, t)) return false;
  t_raw = t;
  t_reported = t / len;
  return true;
}

// The ray in a sphere's local frame: origin, and direction after the reference's re-normalisation
// (src/shapes/sphere.rs:52, src/geometry/ray.rs:45). When the sphere's map is a pure translation (every
// sphere the stock scenes have but one) the linear part multiplies the direction by exactly 1 and adds exact
// zeros, so the local direction is the world direction and its normalisation is the `dn` the caller already
// holds: a square root and three quotients saved per test. The one thing a product with the identity can
// change is the sign of a zero, so directions with a zero component take the general path, which keeps the
// results identical to the bit. (The origin always takes it: it costs little, and camera positions with a
// zero coordinate are common.)
__device__ __forceinline__ void sphere_local_ray(const DevSphere& s, V3 o, V3 d, V3 dn, V3& lo, V3& ld) {
  lo = xf_pt(s.w2l, o);
  if (s.translation_only && d.x != 0.0 && d.y != 0.0 && d.z != 0.0) ld = dn;
  else ld = normalize(xf_vec(s.w2l, d));
}

__device__ __forceinline__ bool sphere_t(const DevSphere& s, V3 o, V3 d, V3 dn, double& t_out) {
  V3 lo, ld;
  sphere_local_ray(s, o, d, dn, lo, ld);
  const double h = dot(ld, lo);
  const double c = dot(lo, lo) - s.radius * s.radius;
  const double disc = h * h - c;
  if (disc < 0.0) return false;
  const double sq = sqrt(disc);
  const double t1 = -h - sq, t2 = -h + sq;
  if (t2 < 0.0) return false;
  t_out = t1 < 0.0 ? t2 : t1;
  return true;
}

// The ray as the binary32 box tests see it: origin relative to an object's centre, direction and
// its reciprocal, and delta, a bound on how far the binary32 ray can stray from the binary64 one
// anywhere inside the object's bounding sphere (rounding of the origin, 2^-24 |o - c|, plus rounding
// of the direction over the longest useful distance, 2^-24 (|o - c| + R)) plus the rounding of the few
// binary32 operations that place a plane along the ray (another 2^-22 (|o - c| + R) at most). Growing
// every box by delta keeps the box test conservative; 2^-20 (|o - c| + R) leaves a factor of two to spare.
struct RayDir { float dx, dy, dz, ix, iy, iz; };
struct BoxRay { float ox, oy, oz, dx, dy, dz, ix, iy, iz, delta; };

__device__ __forceinline__ RayDir make_ray_dir(V3 d) {
  RayDir r;
  r.dx = float(d.x); r.dy = float(d.y); r.dz = float(d.z);
  // hardware reciprocal (1 ulp): both faces of a slab are scaled by the same factor, so the error only
  // enters the comparison between axes, well inside slab32's slack
  r.ix = __builtin_amdgcn_rcpf(r.dx); r.iy = __builtin_amdgcn_rcpf(r.dy); r.iz = __builtin_amdgcn_rcpf(r.dz);
  return r;
}

__device__ __forceinline__ BoxRay make_box_ray(V3 o, const RayDir& d, const DevObject& ob) {
  BoxRay r;
  r.ox = float(o.x - ob.center[0]); r.oy = float(o.y - ob.center[1]); r.oz = float(o.z - ob.center[2]);
  r.delta = 0x1.0p-20f * (fmaxf(fmaxf(fabsf(r.ox), fabsf(r.oy)), fabsf(r.oz)) + float(ob.radius) * 1.0000005f);
  r.dx = d.dx; r.dy = d.dy; r.dz = d.dz; r.ix = d.ix; r.iy = d.iy; r.iz = d.iz;
  return r;
}

__device__ __forceinline__ BoxRay make_box_ray(V3 o, V3 d, const DevObject& ob) { return make_box_ray(o, make_ray_dir(d), ob); }

// Slab test in binary32 against a box grown by the ray's delta. A zero direction component gives
// +-inf or NaN; fminf/fmaxf drop a NaN operand, which only makes the test more permissive. The
// final comparison allows for the rounding of the six subtract-multiply pairs.
__device__ __forceinline__ bool slab32(const float* lo, const float* hi, const BoxRay& r, float tmax, float& tnear) {
  const float ax = (lo[0] - (r.ox + r.delta)) * r.ix, bx = (hi[0] - (r.ox - r.delta)) * r.ix;
  const float ay = (lo[1] - (r.oy + r.delta)) * r.iy, by = (hi[1] - (r.oy - r.delta)) * r.iy;
  const float az = (lo[2] - (r.oz + r.delta)) * r.iz, bz = (hi[2] - (r.oz - r.delta)) * r.iz;
  const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
  const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
  tnear = tn;
  return tn <= tf * 1.000002f;
}

// Can the ray touch the object at all? Every object carries a padded binary32 box around itself
// (an ellipsoid's, a triangle's or a mesh's extent); a ray that misses it skips the exact binary64
// test, which for a sphere or a triangle costs five to ten times as much as this.
__device__ __forceinline__ bool meets_box(const DevObject& ob, V3 o, const RayDir& dir) {
  const BoxRay br = make_box_ray(o, dir, ob);
  float tn;
  return slab32(ob.box_lo, ob.box_hi, br, INFINITY, tn);
}

// ModelObj::intersect's own box test (src/shapes/model_obj.rs:78 -> src/geometry/aabb.rs:136-247), exactly as the
// reference evaluates it: the ray as it was built (direction normalised once, at Ray::new), binary64 quotients, the
// interval starting at [0, inf), a zero direction component decided by containment. A model whose box the ray fails
// reports no hit, whatever Moller-Trumbore would say about a triangle that touches the box's surface.
__device__ __forceinline__ bool model_gate_exact(const double* lo, const double* hi, V3 o, V3 d) {
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  double t_lo = 0.0, t_hi = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (dd[k] == 0.0) {
      if (oo[k] < lo[k] || oo[k] > hi[k]) return false;
    } else {
      const double t0 = (lo[k] - oo[k]) / dd[k], t1 = (hi[k] - oo[k]) / dd[k];
      const double t_min = dd[k] > 0.0 ? t0 : t1, t_max = dd[k] > 0.0 ? t1 : t0;
      if (t_min > t_lo) t_lo = t_min;
      if (t_max < t_hi) t_hi = t_max;
      if (t_lo > t_hi) return false;
    }
  }
  return true;
}

// Does the ray certainly pass that test? It does when its binary32 image crosses the model's inner box shrunk further
// by four times the ray's own delta: the slack this leaves in every slab interval (margin / |d_k|) exceeds both the
// binary32 errors of this test and the rounding of the exact binary64 quotients, which scale the same way. Directions
// with a (near-)zero component are left to the exact test.
__device__ __forceinline__ bool model_gate_certain(const DevObject& ob, const BoxRay& r) {
  if (!(fabsf(r.dx) > 1e-18f && fabsf(r.dy) > 1e-18f && fabsf(r.dz) > 1e-18f)) return false;
  const float m = 4.0f * r.delta;
  const float ax = ((ob.inner_lo[0] + m) - r.ox) * r.ix, bx = ((ob.inner_hi[0] - m) - r.ox) * r.ix;
  const float ay = ((ob.inner_lo[1] + m) - r.oy) * r.iy, by = ((ob.inner_hi[1] - m) - r.oy) * r.iy;
  const float az = ((ob.inner_lo[2] + m) - r.oz) * r.iz, bz = ((ob.inner_hi[2] - m) - r.oz) * r.iz;
  // an inner box turned inside out by the margins (a flat model) has min > max on that axis and can never pass: the
  // comparisons below are ordered ones, which a NaN also fails
  if (!(ob.inner_lo[0] + m < ob.inner_hi[0] - m && ob.inner_lo[1] + m < ob.inner_hi[1] - m && ob.inner_lo[2] + m < ob.inner_hi[2] - m))
    return false;
  const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
  const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  return tn <= tf;
}

// ------------------------------------------------------------------------------- quantised BVH nodes

// How a node format (rt_types.h: RT_NODES_*) names children. `kLeaf` marks a leaf reference, `kNone` an empty slot
// and, as the current node of a lane, "traversal finished" (it carries the leaf bit, so it also ends a node loop).
template <int FMT> struct NodeFmt {
  static constexpr uint32_t kWords4 = FMT == RT_NODES_QUAD ? 4u : (FMT == RT_NODES_WIDE ? 3u : 2u);
  static constexpr uint32_t kLeaf = FMT == RT_NODES_SMALL ? RT_SMALL_LEAF_FLAG : RT_LEAF_FLAG;
  static constexpr uint32_t kNone = FMT == RT_NODES_SMALL ? RT_SMALL_REF_NONE : RT_REF_NONE;
  // what a traversal-stack entry is: the 16-bit references of the small format halve the stacks' LDS, which is what
  // lets a fifth workgroup share a CU with the staged tree
  typedef typename std::conditional<FMT == RT_NODES_SMALL, uint16_t, uint32_t>::type stack_t;
  // a leaf reference in the common form RT_LEAF_FLAG | first << 4 | count
  static __device__ __forceinline__ uint32_t leaf(uint32_t ref) {
    return FMT == RT_NODES_SMALL ? (RT_LEAF_FLAG | ((ref & 0x7FFCu) << 2) | ((ref & 3u) + 1u)) : ref;
  }
};

// The ray against a mesh's quantisation grid: a plane at level q of axis k lies at grid_lo + q * grid_step, so its
// distance along the ray is q * (step * inv_d) + (grid_lo - o) * inv_d -- one fused multiply-add per plane, with the
// origin moved by -+delta for the low / high planes (the box grown by the ray's delta, as in slab32). The rounding
// of these few binary32 operations stays within the part of delta that is kept for it and within the grid step every
// box was pushed outward by.
struct QuantRay { float sx, sy, sz, lx, ly, lz, hx, hy, hz; };

__device__ __forceinline__ QuantRay make_quant_ray(const BoxRay& r, const DevObject& ob) {
  // A direction component that is zero, or so small that its reciprocal overflows, would make these products infinite, and
  // a plane distance "finite * inf + (-+inf)" is an infinity or a NaN depending on the plane's sign: boxes the ray runs
  // inside of would be rejected. The reciprocals are therefore held to +-2^40. A ray that shallow moves less than
  // 2^-40 of the distance it travels along that axis -- far less than delta over any distance that matters -- so its slab
  // test is a containment test of the origin, which is what these finite products then compute: planes on either side of
  // the origin land at -+(their distance) * 2^40, planes on one side both far out on that side.
  const float ix = fminf(fmaxf(r.ix, -0x1.0p40f), 0x1.0p40f), iy = fminf(fmaxf(r.iy, -0x1.0p40f), 0x1.0p40f),
              iz = fminf(fmaxf(r.iz, -0x1.0p40f), 0x1.0p40f);
  QuantRay q;
  q.sx = ob.grid_step[0] * ix; q.sy = ob.grid_step[1] * iy; q.sz = ob.grid_step[2] * iz;
  q.lx = (ob.grid_lo[0] - (r.ox + r.delta)) * ix; q.hx = (ob.grid_lo[0] - (r.ox - r.delta)) * ix;
  q.ly = (ob.grid_lo[1] - (r.oy + r.delta)) * iy; q.hy = (ob.grid_lo[1] - (r.oy - r.delta)) * iy;
  q.lz = (ob.grid_lo[2] - (r.oz + r.delta)) * iz; q.hz = (ob.grid_lo[2] - (r.oz - r.delta)) * iz;
  return q;
}

struct Children { bool h0, h1; float tn0, tn1; uint32_t ref0, ref1, min0, min1; };

// The two planes a node word holds, as the binary32 factors of the fused multiply-adds below: integer levels, or (the
// small format with RT_SMALL_F16) binary16 values, whose conversion the multiply-add instruction does by itself.
typedef _Float16 HalfPair __attribute__((ext_vector_type(2)));
template <int FMT> __device__ __forceinline__ float q_lo16(uint32_t w) {
  if (FMT == RT_NODES_SMALL && RT_SMALL_F16) return float(__builtin_bit_cast(HalfPair, w).x);
  return float(w & 0xFFFFu);
}
template <int FMT> __device__ __forceinline__ float q_hi16(uint32_t w) {
  if (FMT == RT_NODES_SMALL && RT_SMALL_F16) return float(__builtin_bit_cast(HalfPair, w).y);
  return float(w >> 16);
}

// Both children of the node at `np` against the ray: slab tests on the quantised planes (a zero direction component
// gives +-inf or NaN; fminf/fmaxf drop a NaN operand, which only makes the test more permissive), the final comparison
// allowing for the rounding of the plane distances; then the references and minimum indices in the format's form.
template <int FMT, bool BOUNDED = true>
__device__ __forceinline__ Children test_children(const uint4* __restrict__ np, const QuantRay& r, float tmax) {
  const uint4 w0 = np[0], w1 = np[1];
  Children c;
  {
    const float ax = __builtin_fmaf(q_lo16<FMT>(w0.x), r.sx, r.lx), bx = __builtin_fmaf(q_hi16<FMT>(w0.y), r.sx, r.hx);
    const float ay = __builtin_fmaf(q_hi16<FMT>(w0.x), r.sy, r.ly), by = __builtin_fmaf(q_lo16<FMT>(w0.z), r.sy, r.hy);
    const float az = __builtin_fmaf(q_lo16<FMT>(w0.y), r.sz, r.lz), bz = __builtin_fmaf(q_hi16<FMT>(w0.z), r.sz, r.hz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    // BOUNDED = false: the caller has no upper bound on t (first-index rule), and the min with +inf is left out
    const float tf = BOUNDED ? fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax))
                             : fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    c.tn0 = tn; c.h0 = tn <= tf * 1.000002f;
  }
  {
    const float ax = __builtin_fmaf(q_lo16<FMT>(w0.w), r.sx, r.lx), bx = __builtin_fmaf(q_hi16<FMT>(w1.x), r.sx, r.hx);
    const float ay = __builtin_fmaf(q_hi16<FMT>(w0.w), r.sy, r.ly), by = __builtin_fmaf(q_lo16<FMT>(w1.y), r.sy, r.hy);
    const float az = __builtin_fmaf(q_lo16<FMT>(w1.x), r.sz, r.lz), bz = __builtin_fmaf(q_hi16<FMT>(w1.y), r.sz, r.hz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = BOUNDED ? fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax))
                             : fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    c.tn1 = tn; c.h1 = tn <= tf * 1.000002f;
  }
  if (FMT == RT_NODES_SMALL) {
    c.ref0 = w1.z & 0xFFFFu; c.ref1 = w1.z >> 16; c.min0 = w1.w & 0xFFFFu; c.min1 = w1.w >> 16;
  } else if (FMT == RT_NODES_NEAREST) {
    c.ref0 = w1.z; c.ref1 = w1.w; c.min0 = 0u; c.min1 = 0u;
  } else {
    const uint4 w2 = np[2];
    c.ref0 = w1.z; c.ref1 = w1.w; c.min0 = w2.x; c.min1 = w2.y;
  }
  c.h0 = c.h0 & (c.ref0 != NodeFmt<FMT>::kNone);   // no branch: empty slots are rare
  c.h1 = c.h1 & (c.ref1 != NodeFmt<FMT>::kNone);
  return c;
}

// RT_NODES_QUAD: the four children of the node at `np` against the ray, then ordered by entry distance. Returns how many
// the ray meets; ref[0 .. n) are their references, nearest first (the nearest rule walks into ref[0] and stacks the others
// so that ref[1] is popped next). An unused slot has an inside-out box and the reference RT_REF_NONE; it never counts.
struct Quad { uint32_t ref[4]; uint32_t n; };

__device__ __forceinline__ Quad test_children4(const uint4* __restrict__ np, const QuantRay& r, float tmax) {
  const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
  const uint32_t w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
  uint32_t ref[4] = {w3.x, w3.y, w3.z, w3.w};
  uint32_t key[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // child c: levels lo.x lo.y | lo.z hi.x | hi.y hi.z in words 3c .. 3c + 2
    const uint32_t a = w[3 * c], b = w[3 * c + 1], d = w[3 * c + 2];
    const float ax = __builtin_fmaf(float(a & 0xFFFFu), r.sx, r.lx), bx = __builtin_fmaf(float(b >> 16), r.sx, r.hx);
    const float ay = __builtin_fmaf(float(a >> 16), r.sy, r.ly), by = __builtin_fmaf(float(d & 0xFFFFu), r.sy, r.hy);
    const float az = __builtin_fmaf(float(b & 0xFFFFu), r.sz, r.lz), bz = __builtin_fmaf(float(d >> 16), r.sz, r.hz);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    const bool hit = (tn <= tf * 1.000002f) & (ref[c] != RT_REF_NONE);
    // tn >= 0 (it is a maximum with 0 that drops NaNs), so its bit pattern orders like its value
    key[c] = hit ? __float_as_uint(tn) : 0xFFFFFFFFu;
  }
  // sorting network for four (key, reference) pairs
#define RTMI_CSWAP(i, j) do { const bool s_ = key[j] < key[i]; const uint32_t ki_ = s_ ? key[j] : key[i], kj_ = s_ ? key[i] : key[j], \
                                ri_ = s_ ? ref[j] : ref[i], rj_ = s_ ? ref[i] : ref[j]; key[i] = ki_; key[j] = kj_; ref[i] = ri_; ref[j] = rj_; } while (0)
  RTMI_CSWAP(0, 1); RTMI_CSWAP(2, 3); RTMI_CSWAP(0, 2); RTMI_CSWAP(1, 3); RTMI_CSWAP(1, 2);
#undef RTMI_CSWAP
  Quad q;
  q.ref[0] = ref[0]; q.ref[1] = ref[1]; q.ref[2] = ref[2]; q.ref[3] = ref[3];
  q.n = uint32_t(key[0] != 0xFFFFFFFFu) + uint32_t(key[1] != 0xFFFFFFFFu) + uint32_t(key[2] != 0xFFFFFFFFu) + uint32_t(key[3] != 0xFFFFFFFFu);
  return q;
}

struct MeshHit { double t, t_raw; uint32_t index; uint32_t slot; };

// Closest hit inside one mesh through its BVH. RULE = RT_HIT_FIRST_INDEX returns the hit
// triangle with the smallest file-order index (what the reference's linear scan returns,
// src/shapes/model_obj.rs:80-84): subtrees whose smallest index cannot beat the current
// winner are skipped, distance never prunes. RULE = RT_HIT_NEAREST returns the smallest t,
// ties to the smaller index. The traversal stack lives in LDS, one column per lane.
//
// Two nested loops: lanes first walk inner nodes (cheap binary32 box tests) until each holds a
// leaf or is done, then the wave does its leaves together, so the expensive triangle code runs
// once per leaf a lane visits instead of once per node step of the slowest lane.
template <int FMT, int RULE, bool COUNT>
__device__ __forceinline__ bool mesh_closest(const uint4* __restrict__ nodes, const DevMeshTri* __restrict__ tris,
                                             const DevTriFrame* __restrict__ frames, const DevObject& ob, V3 o, V3 d_built, V3 d,
                                             uint32_t* stack, MeshHit& best, uint32_t& n_nodes, uint32_t& n_prims) {
  typedef NodeFmt<FMT> F;
  const BoxRay br = make_box_ray(o, d, ob);
  best.t = INFINITY; best.t_raw = 0.0; best.index = RT_MISS; best.slot = 0;
  if (!model_gate_certain(ob, br) && !model_gate_exact(ob.gate_lo, ob.gate_hi, o, d_built)) return false;
  const QuantRay qr = make_quant_ray(br, ob);
  constexpr uint32_t kDone = F::kNone;
  uint32_t cur = 0;
  int sp = 0;
  for (;;) {
    while (!(cur & F::kLeaf)) {
      if (COUNT) ++n_nodes;
      // nearest rule: nothing farther than the best hit so far (rounded up, with the same slack) can win
      const float tmax = RULE == RT_HIT_NEAREST ? __double2float_ru(best.t) * 1.000002f : INFINITY;
      if (FMT == RT_NODES_QUAD) {
        const Quad c = test_children4(nodes + size_t(cur) * F::kWords4, qr, tmax);
        if (c.n > 3) stack[(sp++) * kBlock] = c.ref[3];
        if (c.n > 2) stack[(sp++) * kBlock] = c.ref[2];
        if (c.n > 1) stack[(sp++) * kBlock] = c.ref[1];
        cur = c.n ? c.ref[0] : (sp ? stack[(--sp) * kBlock] : kDone);
        continue;
      }
      const Children c = test_children<FMT>(nodes + size_t(cur) * F::kWords4, qr, tmax);   // from LDS when the tree is staged
      bool h0 = c.h0, h1 = c.h1;
      if (RULE == RT_HIT_FIRST_INDEX) { h0 = h0 && c.min0 < best.index; h1 = h1 && c.min1 < best.index; }
      if (h0 && h1) {
        const bool swap = RULE == RT_HIT_FIRST_INDEX ? (c.min1 < c.min0) : (c.tn1 < c.tn0);
        stack[(sp++) * kBlock] = swap ? c.ref0 : c.ref1;
        cur = swap ? c.ref1 : c.ref0;
      } else if (h0) {
        cur = c.ref0;
      } else if (h1) {
        cur = c.ref1;
      } else {
        cur = sp ? stack[(--sp) * kBlock] : kDone;
      }
    }
    if (cur == kDone) break;
    const uint32_t leaf = F::leaf(cur);
    const uint32_t first = (leaf & ~RT_LEAF_FLAG) >> 4, cnt = leaf & 15u;
    for (uint32_t i = 0; i < cnt; ++i) {
      const DevMeshTri* tp = tris + first + i;
      const uint32_t index = tp->index;
      if (RULE == RT_HIT_FIRST_INDEX && index >= best.index) break;  // a leaf is sorted by index
      if (COUNT) ++n_prims;
      const double2* q = reinterpret_cast<const double2*>(tp->v);
      const double2 a = q[0], b = q[1], c = q[2], e = q[3];
      const double last = tp->v[8];
      double t_raw, t;
      if (triangle_hit(mk(a.x, a.y, b.x), mk(b.y, c.x, c.y), mk(e.x, e.y, last), o, d, frames[first + i].len, t_raw, t)) {
        if (RULE == RT_HIT_FIRST_INDEX) {
          best.t = t; best.t_raw = t_raw; best.index = index; best.slot = first + i;
          break;
        } else if (t < best.t || (t == best.t && index < best.index)) {
          best.t = t; best.t_raw = t_raw; best.index = index; best.slot = first + i;
        }
      }
    }
    cur = sp ? stack[(--sp) * kBlock] : kDone;
  }
  return best.index != RT_MISS;
}

struct Hit {
  uint32_t obj, prim;
  double t;
  V3 p, n;
  DevMaterial mat;
};

__device__ __forceinline__ void sphere_surface(const DevSphere& s, V3 o, V3 d, V3 dn, double t, V3& p, V3& n) {
  V3 lo, ld;
  sphere_local_ray(s, o, d, dn, lo, ld);
  const V3 pl = lo + ld * t;
  p = xf_pt(s.l2w, pl);                        // src/shapes/sphere.rs:98
  n = normalize(xf_vec_t(s.w2l, pl));          // src/shapes/sphere.rs:37-43
}

// Hit point, normal and material of the winning object only. When every lane that is here hit the same object (primary
// rays on one sphere, bounce rays on the ground) its records come through scalar loads; otherwise each lane gathers its own.
__device__ __forceinline__ void hit_surface(const DevScene& sc, uint32_t obj, V3 o, V3 d, V3 dn, double t, double t_raw,
                                            uint32_t tri_slot, V3& p, V3& n, DevMaterial& mat) {
  const uint32_t obj0 = uint32_t(__builtin_amdgcn_readfirstlane(int(obj)));
  if (__ballot(obj != obj0) == 0) {
    const DevObject ob = load_uniform(sc.objects + obj0);
    if (ob.kind == RT_OBJ_SPHERE) {
      const DevSphere s = load_uniform(sc.spheres + ob.first);
      sphere_surface(s, o, d, dn, t, p, n);
    } else {
      p = o + dn * t_raw;                                        // src/shapes/triangle.rs:78-81
      if (ob.kind == RT_OBJ_TRIANGLE) { const DevTopTri tr = load_uniform(sc.top_tris + ob.first); n = ld3(tr.frame.n); }
      else n = ld3(sc.mesh_frames[ob.first + tri_slot].n);       // src/shapes/triangle.rs:30, then :91
    }
    mat = load_uniform(sc.materials + ob.material);
    return;
  }
  const DevObject& ob = sc.objects[obj];
  if (ob.kind == RT_OBJ_SPHERE) {
    sphere_surface(sc.spheres[ob.first], o, d, dn, t, p, n);
  } else {
    const DevTriFrame& f = ob.kind == RT_OBJ_TRIANGLE ? sc.top_tris[ob.first].frame : sc.mesh_frames[ob.first + tri_slot];
    p = o + dn * t_raw;
    n = ld3(f.n);
  }
  mat = sc.materials[ob.material];
}

// Scene::intersect_ray (src/scene.rs:135-150): closest over the object list in scene order,
// strict '<' so the earlier object keeps a tie. The hit point and normal are formed once,
// for the winner only.
// MESH = false: the scene has no model at all, and the traversal code is left out of the kernel (it would cost the
// analytic-only scenes registers, i.e. spills at four waves per SIMD).
template <int RULE, bool COUNT, bool MESH = true>
__device__ __forceinline__ bool closest_hit(const DevScene& sc, const uint4* __restrict__ node_words, V3 o, V3 d,
                                            uint32_t* stack, Hit& hit, uint32_t& n_nodes, uint32_t& n_prims) {
  bool found = false;
  double best_t = 0.0, best_raw = 0.0;
  uint32_t best_obj = 0, best_prim = 0, best_slot = 0;
  const V3 dn = normalize(d);  // what `identity.inverse() * ray` leaves of the direction (src/geometry/ray.rs:45)
  const RayDir dir = make_ray_dir(dn);
  for (uint32_t i = 0; i < sc.n_objects; ++i) {
    const DevObject ob = load_uniform(sc.objects + i);
    double t, t_raw = 0.0;
    uint32_t prim = 0, slot = 0;
    bool ok;
    if (ob.kind != RT_OBJ_MODEL && !meets_box(ob, o, dir)) continue;
    if (ob.kind == RT_OBJ_SPHERE) {
      const DevSphere sp = load_uniform(sc.spheres + ob.first);
      ok = sphere_t(sp, o, d, dn, t);
    } else if (ob.kind == RT_OBJ_TRIANGLE) {
      const DevTopTri tr = load_uniform(sc.top_tris + ob.first);
      ok = triangle_hit(ld3(tr.v), ld3(tr.v + 3), ld3(tr.v + 6), o, dn, tr.frame.len, t_raw, t);
    } else if (!MESH) {
      continue;
    } else {
      MeshHit mh;
      const DevMeshTri* tris = sc.mesh_tris + ob.first;
      const DevTriFrame* frames = sc.mesh_frames + ob.first;
      // the format is the same for every lane (a property of the scene); the nearest-only form never meets the first-index rule
      if (sc.node_format == RT_NODES_SMALL)
        ok = mesh_closest<RT_NODES_SMALL, RULE, COUNT>(node_words + size_t(ob.root) * 2, tris, frames, ob, o, d, dn, stack, mh, n_nodes, n_prims);
      else if (RULE == RT_HIT_NEAREST && sc.node_format == RT_NODES_NEAREST)
        ok = mesh_closest<RT_NODES_NEAREST, RULE, COUNT>(node_words + size_t(ob.root) * 2, tris, frames, ob, o, d, dn, stack, mh, n_nodes, n_prims);
      else if (RULE == RT_HIT_NEAREST && sc.node_format == RT_NODES_QUAD)
        ok = mesh_closest<RT_NODES_QUAD, RULE, COUNT>(node_words + size_t(ob.root) * 4, tris, frames, ob, o, d, dn, stack, mh, n_nodes, n_prims);
      else
        ok = mesh_closest<RT_NODES_WIDE, RULE, COUNT>(node_words + size_t(ob.root) * 3, tris, frames, ob, o, d, dn, stack, mh, n_nodes, n_prims);
      t = mh.t; t_raw = mh.t_raw; prim = mh.index; slot = mh.slot;
    }
    if (ok && (!found || t < best_t)) {
      found = true; best_t = t; best_raw = t_raw; best_obj = i; best_prim = prim; best_slot = slot;
    }
  }
  if (!found) return false;
  hit.obj = best_obj; hit.prim = best_prim; hit.t = best_t;
  hit_surface(sc, best_obj, o, d, dn, best_t, best_raw, best_slot, hit.p, hit.n, hit.mat);
  return true;
}

// The same loop restricted to spheres and lone triangles (meshes are traverse_kernel's business).
struct Prelim { double t, t_raw; uint32_t obj; };

__device__ __forceinline__ void simple_objects(const DevScene& sc, V3 o, V3 d, V3 dn, Prelim& best) {
  best.t = 0.0; best.t_raw = 0.0; best.obj = RT_MISS;
  const RayDir dir = make_ray_dir(dn);
  for (uint32_t i = 0; i < sc.n_objects; ++i) {
    const DevObject ob = load_uniform(sc.objects + i);
    double t, t_raw = 0.0;
    bool ok;
    if (ob.kind == RT_OBJ_MODEL || !meets_box(ob, o, dir)) continue;
    if (ob.kind == RT_OBJ_SPHERE) {
      const DevSphere sp = load_uniform(sc.spheres + ob.first);
      ok = sphere_t(sp, o, d, dn, t);
    } else {
      const DevTopTri tr = load_uniform(sc.top_tris + ob.first);
      ok = triangle_hit(ld3(tr.v), ld3(tr.v + 3), ld3(tr.v + 6), o, dn, tr.frame.len, t_raw, t);
    }
    if (ok && (best.obj == RT_MISS || t < best.t)) { best.t = t; best.t_raw = t_raw; best.obj = i; }
  }
}

// Does the ray meet the (padded binary32) box of any mesh, i.e. does traverse_kernel have to look at it?
// `d` need not be exactly normalised: the box test only uses its direction, with margins far above an ulp.
__device__ __forceinline__ bool meets_any_mesh(const DevScene& sc, V3 o, V3 d) {
  const RayDir dir = make_ray_dir(d);
  bool meets = false;
  for (uint32_t k = 0; k < sc.n_meshes; ++k) {
    const DevObject ob = load_uniform(sc.objects + load_uniform(sc.mesh_objects + k));
    meets = meets || meets_box(ob, o, dir);
  }
  return meets;
}

// ------------------------------------------------------------------------------- spectra

// Per-hit spectrum rows, one byte per hit, kept in two uint4 registers.
struct Chain { uint4 a, b; };

__device__ __forceinline__ uint32_t& chain_word(Chain& c, int w) {
  switch (w) {
    case 0: return c.a.x; case 1: return c.a.y; case 2: return c.a.z; case 3: return c.a.w;
    case 4: return c.b.x; case 5: return c.b.y; case 6: return c.b.z; default: return c.b.w;
  }
}
__device__ __forceinline__ void chain_set(Chain& c, uint32_t k, uint32_t row) {
#pragma unroll
  for (int w = 0; w < 8; ++w)
    if (int(k >> 2) == w) chain_word(c, w) = (chain_word(c, w) & ~(0xFFu << ((k & 3u) * 8u))) | (row << ((k & 3u) * 8u));
}
__device__ __forceinline__ uint32_t chain_get(const Chain& c, uint32_t k) {
  uint32_t word = 0;
  Chain& m = const_cast<Chain&>(c);
#pragma unroll
  for (int w = 0; w < 8; ++w)
    if (int(k >> 2) == w) word = chain_word(m, w);
  return (word >> ((k & 3u) * 8u)) & 0xFFu;
}

// The sample's spectrum when a path ends on an emitter: ((Le * c_k) * c_{k-1}) ... * c_1, the
// innermost-first order of the reference's recursion (src/scene.rs:75,111), then to_XYZ
// (src/spectrum/mod.rs:590-609). Exact binary32 products and sums, no contraction.
__device__ __forceinline__ void emit_sample(const float* __restrict__ spectra, uint32_t emitted_row, const Chain& chain,
                                            uint32_t n_chain, float xyz[3], float* spectrum_out) {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  const float4* table = reinterpret_cast<const float4*>(spectra);
  for (int q = 0; q < RT_NBINS / 4; ++q) {
    float4 v = table[emitted_row * (RT_NBINS / 4) + q];
    for (int k = int(n_chain) - 1; k >= 0; --k) {
      const uint32_t row = chain_get(chain, uint32_t(k));
      if (row != RT_NO_SPECTRUM) {
        const float4 c = table[row * (RT_NBINS / 4) + q];
        v.x = (v.x * c.x); v.y = (v.y * c.y); v.z = (v.z * c.z); v.w = (v.w * c.w);
      }
    }
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = q * 4 + j;
      x = (x + (c_cie[0][i] * e[j]));
      y = (y + (c_cie[1][i] * e[j]));
      z = (z + (c_cie[2][i] * e[j]));
      if (spectrum_out) spectrum_out[i] = e[j];
    }
  }
  xyz[0] = (x * c_xyz_scale); xyz[1] = (y * c_xyz_scale); xyz[2] = (z * c_xyz_scale);
}

// NormalIntegrator: from_RGB(n * 0.5 + 0.5, Reflectance) then to_XYZ
// (src/integrators/normal.rs:48-59, src/spectrum/mod.rs:525-581).
__device__ __forceinline__ void normal_sample(V3 n, float xyz[3], float* spectrum_out) {
  const float r = float(fma(n.x, 0.5, 0.5)), g = float(fma(n.y, 0.5, 0.5)), b = float(fma(n.z, 0.5, 0.5));
  enum { W, C, M, Y, R, G, B };
  int s1, s2; float k0, k1, k2;
  if (r < g && r < b) {
    k0 = r;
    if (g < b) { s1 = C; k1 = g - r; s2 = B; k2 = b - g; } else { s1 = C; k1 = b - r; s2 = G; k2 = g - b; }
  } else if (g < r && g < b) {
    k0 = g;
    if (r < b) { s1 = M; k1 = r - g; s2 = B; k2 = b - r; } else { s1 = M; k1 = b - g; s2 = R; k2 = r - b; }
  } else {
    k0 = b;
    if (r < g) { s1 = Y; k1 = r - b; s2 = G; k2 = g - r; } else { s1 = Y; k1 = g - b; s2 = R; k2 = r - g; }
  }
  float x = 0.0f, y = 0.0f, z = 0.0f;
  for (int i = 0; i < RT_NBINS; ++i) {
    float v = (0.0f + (c_smits_refl[W][i] * k0));
    v = (v + (c_smits_refl[s1][i] * k1));
    v = (v + (c_smits_refl[s2][i] * k2));
    x = (x + (c_cie[0][i] * v));
    y = (y + (c_cie[1][i] * v));
    z = (z + (c_cie[2][i] * v));
    if (spectrum_out) spectrum_out[i] = v;
  }
  xyz[0] = (x * c_xyz_scale); xyz[1] = (y * c_xyz_scale); xyz[2] = (z * c_xyz_scale);
}

// ------------------------------------------------------------------------------- sampling

// Cosine-weighted direction about n (src/geometry/mod.rs:85-116): r1 first, then r2.
__device__ __forceinline__ V3 cosine_direction(V3 n, double r1, double r2) {
  const double sin_t = sqrt(r1);
  const double cos_t = sqrt(1.0 - sin_t * sin_t);
  const double phi = 2.0 * M_PI * r2;
  double sp, cp;
  sincos(phi, &sp, &cp);
  const V3 v = mk(cp * sin_t, sp * sin_t, cos_t);
  V3 b1 = fabs(n.x) > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
  b1 = b1 - n * dot(b1, n);
  b1 = b1 * (1.0 / sqrt(dot(b1, b1)));
  const V3 b2 = cross(n, b1);
  return mk((b1.x * v.x + b2.x * v.y) + n.x * v.z, (b1.y * v.x + b2.y * v.y) + n.y * v.z,
            (b1.z * v.x + b2.z * v.y) + n.z * v.z);
}

// Extension materials (no counterpart in the reference). Phong lobe about the unit vector `a`:
// cos(theta) = r1^(1/(ns+1)), phi = 2 pi r2, in the basis construction of the cosine sampler.
__device__ __forceinline__ V3 phong_lobe_direction(V3 a, double ns, double r1, double r2) {
  const double cos_t = pow(r1, 1.0 / (ns + 1.0));
  const double sin_t = sqrt(1.0 - cos_t * cos_t);
  const double phi = 2.0 * M_PI * r2;
  double sp, cp;
  sincos(phi, &sp, &cp);
  const V3 v = mk(cp * sin_t, sp * sin_t, cos_t);
  V3 b1 = fabs(a.x) > 0.9 ? mk(0.0, 1.0, 0.0) : mk(1.0, 0.0, 0.0);
  b1 = b1 - a * dot(b1, a);
  b1 = b1 * (1.0 / sqrt(dot(b1, b1)));
  const V3 b2 = cross(a, b1);
  return mk((b1.x * v.x + b2.x * v.y) + a.x * v.z, (b1.y * v.x + b2.y * v.y) + a.y * v.z,
            (b1.z * v.x + b2.z * v.y) + a.z * v.z);
}

// Smooth dielectric of index ni met by unit direction i; nf = unit normal facing against i. u < R
// (Schlick's reflectance at the vacuum-side angle) or total internal reflection reflects, otherwise
// the ray refracts. Returns whether it reflected; `out` is not normalised.
__device__ __forceinline__ bool dielectric_direction(V3 i, V3 nf, bool entering, double ni, double u, V3& out) {
  if (!(ni > 0.0)) ni = 1.0;
  const double cosi = -dot(i, nf);
  const double eta = entering ? 1.0 / ni : ni;
  const double sin2t = (eta * eta) * (1.0 - cosi * cosi);
  bool reflect;
  double cost = 0.0;
  if (sin2t > 1.0) {
    reflect = true;
  } else {
    cost = sqrt(1.0 - sin2t);
    double r0 = (1.0 - ni) / (1.0 + ni);
    r0 = r0 * r0;
    const double c = entering ? cosi : cost;
    const double x = 1.0 - c, x2 = x * x;
    const double refl = r0 + (1.0 - r0) * ((x2 * x2) * x);
    reflect = u < refl;
  }
  out = reflect ? i + nf * (2.0 * cosi) : i * eta + nf * (eta * cosi - cost);
  return reflect;
}

// ------------------------------------------------------------------------------- queues

// words of the chain a path of queue level `level` carries: one byte per hit it has behind it
__device__ __forceinline__ uint32_t chain_words(uint32_t level) { return (level + 3u) >> 2; }

__device__ __forceinline__ void queue_store(const PathQueue& q, uint32_t at, V3 o, V3 d, uint32_t slot, const Chain& chain, uint32_t n_words) {
  q.o_xy()[at] = make_double2(o.x, o.y); q.oz_dx()[at] = make_double2(o.z, d.x); q.d_yz()[at] = make_double2(d.y, d.z);
  q.slot()[at] = slot;
  const uint32_t w[8] = {chain.a.x, chain.a.y, chain.a.z, chain.a.w, chain.b.x, chain.b.y, chain.b.z, chain.b.w};
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) if (k < n_words) q.chain_word(k)[at] = w[k];   // n_words is the same for the whole launch
}

// local pixel of a batch -> image coordinates. Tiles are numbered x-major as the reference
// enumerates them (src/integrators/integrator.rs:50); inside a tile, 64 consecutive local
// pixels are one 8x8 block, so the primary rays of a wave stay together.
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f) { return uint32_t((uint64_t(n) * f.m) >> f.shift); }

__device__ __forceinline__ bool local_pixel_to_xy(const BatchDesc& b, uint32_t lp, uint32_t& x, uint32_t& y) {
  const uint32_t tile_index = fast_div(lp, b.by_tile_pixels);
  const uint32_t tile = b.tile_list[tile_index];
  const uint32_t within = lp - tile_index * b.by_tile_pixels.d;
  const uint32_t block = within / (kPixelBlock * kPixelBlock), in_block = within % (kPixelBlock * kPixelBlock);
  const uint32_t tile_x = fast_div(tile, b.by_tiles_y), tile_y = tile - tile_x * b.by_tiles_y.d;
  const uint32_t block_y = fast_div(block, b.by_blocks_per_row), block_x = block - block_y * b.by_blocks_per_row.d;
  x = tile_x * b.tile_size + block_x * kPixelBlock + (in_block % kPixelBlock);
  y = tile_y * b.tile_size + block_y * kPixelBlock + (in_block / kPixelBlock);
  return x < b.cam.width && y < b.cam.height;
}

// Dynamic LDS of the traversal kernels: [BVH nodes, when they are staged][traversal stacks].
// Small trees (sc.lds_nodes > 0) are copied into LDS once per workgroup: a node fetch then costs
// an LDS read instead of an L2 round trip, which is what a long first-index traversal waits on.
template <bool STAGED>
__device__ __forceinline__ const uint4* stage_nodes(const DevScene& sc, uint4* lds, uint32_t*& stack) {
  const uint32_t words = STAGED ? sc.lds_nodes : 0;
  stack = reinterpret_cast<uint32_t*>(lds + words) + threadIdx.x;
  const uint4* src = reinterpret_cast<const uint4*>(sc.nodes);
  if (!STAGED) return src;
  for (uint32_t i = threadIdx.x; i < words; i += kBlock) lds[i] = src[i];
  __syncthreads();
  return lds;
}

// Numbers the chunks (kBlock entries each) of a sharded queue region after region: fills
// s_prefix[0..kShards] and s_fill[0..kShards) and returns the total number of chunks.
__device__ __forceinline__ uint32_t number_chunks(const uint32_t* __restrict__ counts, uint32_t region_entries,
                                                  uint32_t* s_prefix, uint32_t* s_fill) {
  if (threadIdx.x < kShards) {
    uint32_t fill = counts[threadIdx.x * kCountStride];
    fill = fill < region_entries ? fill : region_entries;
    uint32_t scan = (fill + kBlock - 1) / kBlock;
    for (int off = 1; off < int(kShards); off <<= 1) {
      const uint32_t up = __shfl_up(scan, off);
      if (int(threadIdx.x) >= off) scan += up;
    }
    s_fill[threadIdx.x] = fill;
    s_prefix[threadIdx.x + 1] = scan;
    if (threadIdx.x == 0) s_prefix[0] = 0;
  }
  __syncthreads();
  return s_prefix[kShards];
}

// chunk -> (region, first entry of the chunk inside the region)
__device__ __forceinline__ uint32_t find_region(const uint32_t* s_prefix, uint32_t chunk) {
  uint32_t lo = 0, hi = kShards;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (s_prefix[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------- pieces of a bounce

// What one path brings into a bounce. pixel and sample are the Philox counter words of the path's random numbers; a
// path that comes out of a queue recomputes them from its slot when it needs them (path_ids).
struct PathState {
  V3 o, d;
  uint32_t pixel, sample, slot;
  Chain chain;
};

// Pixel (y * width + x) and sample of the path whose contribution slot is `slot`.
__device__ __forceinline__ bool path_ids(const BatchDesc& b, uint32_t slot, uint32_t& pixel, uint32_t& sample, uint32_t& x, uint32_t& y) {
  if (b.rays) { pixel = b.pixel_base + slot; sample = b.sample0; x = y = 0; return true; }
  const uint32_t sample_local = fast_div(slot, b.by_local_pixels);
  const uint32_t lp = slot - sample_local * b.n_local_pixels;
  sample = b.sample0 + sample_local;
  const bool inside = local_pixel_to_xy(b, lp, x, y);
  pixel = y * b.cam.width + x;
  return inside;
}

// Path `idx` of a batch at its start: the camera ray through its jittered pixel position
// (evaluate_pixel_sample, src/integrators/integrator.rs:130-135 + Camera::get_ray, src/camera.rs:80-84),
// or the caller's ray. Returns false for a pixel that the last tiles hang over the image edge.
__device__ __forceinline__ bool start_path(const BatchDesc& b, uint32_t idx, PathState& s) {
  s.slot = idx;
  s.chain.a = make_uint4(0, 0, 0, 0); s.chain.b = make_uint4(0, 0, 0, 0);
  uint32_t x, y;
  const bool inside = path_ids(b, idx, s.pixel, s.sample, x, y);
  if (b.rays) {
    s.o = ld3(b.rays + size_t(idx) * 6); s.d = ld3(b.rays + size_t(idx) * 6 + 3);
    return true;
  }
  if (!inside) return false;
  double u0, u1;
  uniform_pair(b.seed, s.pixel, s.sample, 0, 0, u0, u1);
  const double fi = double(x) + u0 - 0.5, fj = double(y) + u1 - 0.5;
  const V3 p1 = ld3(b.cam.p1), p2 = ld3(b.cam.p2), p3 = ld3(b.cam.p3), f = ld3(b.cam.focus);
  const V3 right = ((p2 - p1) * fi) / double(b.cam.width);
  const V3 down = ((p3 - p1) * fj) / double(b.cam.height);
  s.o = f;
  s.d = normalize(((p1 + right) + down) - f);
  return true;
}

__device__ __forceinline__ void load_path(const PathQueue& q, uint32_t idx, uint32_t n_words, PathState& s) {
  const double2 o_xy = q.o_xy()[idx], oz_dx = q.oz_dx()[idx], d_yz = q.d_yz()[idx];
  s.o = mk(o_xy.x, o_xy.y, oz_dx.x);
  s.d = mk(oz_dx.y, d_yz.x, d_yz.y);
  s.slot = q.slot()[idx];
  s.pixel = s.sample = 0;
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) if (k < n_words) w[k] = q.chain_word(k)[idx];
  s.chain.a = make_uint4(w[0], w[1], w[2], w[3]); s.chain.b = make_uint4(w[4], w[5], w[6], w[7]);
}

// What shading a hit decides.
//   miss / black material / depth exhausted : the path ends, contributing nothing
//   emitter                                 : the path ends, contributing Le * (chain product)
//   Lambertian / specular                   : the path continues along (no, nd)
struct Shaded {
  bool alive;       // continues to the next bounce
  bool emits;       // ended on an emitter whose spectrum emit_kernel will form
  uint32_t emit_info;
  V3 no, nd;
};

// first_hit: the path's first intersection keeps its normal as is (src/integrators/random_walk.rs:74);
// later hits turn it against the ray (src/scene.rs:30-32).
// EXT: the scene has glossy or dielectric materials (extensions). Their code (a binary64 pow above all) needs
// some 40 registers more than the reference's three material types, which costs every scene a resident wave
// per SIMD, so kernels are instantiated with and without it.
// level: ray casts the path has made before this one (0 = this is its first intersection). IDS: s.pixel and s.sample are
// valid (the path was started in this kernel); otherwise they are recomputed from the slot where random numbers are drawn.
template <bool EXT, bool IDS>
__device__ __forceinline__ Shaded shade_hit(const DevScene& sc, const BatchDesc& b, const TraceOutputs& o, bool found, Hit& hit,
                                            uint32_t level, PathState& s) {
  const bool first_hit = level == 0;
  // ray casts the path may still make, this one included (the reference's `iteraciones`)
  const uint32_t remaining = b.integrator == RT_INT_RANDOM_WALK ? b.max_depth - level : 0;
  Shaded r;
  r.alive = false; r.emits = false; r.emit_info = 0; r.no = s.o; r.nd = s.d;
  float xyz[3];
  bool contributes = false;
  float* spectrum_out = o.spectra ? o.spectra + size_t(s.slot) * RT_NBINS : nullptr;
  if (found) {
    const DevMaterial m = hit.mat;
    if (b.integrator == RT_INT_ALBEDO) {
      // first present of ambient / emitted / diffused / specular (src/integrators/albedo.rs:54-68)
      uint32_t row = RT_NO_SPECTRUM;
      if (m.present & RT_HAS_AMBIENT) row = m.ambient;
      else if (m.present & RT_HAS_EMITTED) row = m.emitted;
      else if (m.present & RT_HAS_DIFFUSED) row = m.diffused;
      else if (m.present & RT_HAS_SPECULAR) row = m.specular;
      if (row != RT_NO_SPECTRUM) { emit_sample(sc.spectra, row, s.chain, 0, xyz, spectrum_out); contributes = true; }
    } else if (b.integrator == RT_INT_NORMAL) {
      normal_sample(hit.n, xyz, spectrum_out);   // un-turned geometric normal (src/integrators/normal.rs:50)
      contributes = true;
    } else if (remaining != 0) {                 // depth == 0 returns black after intersecting (random_walk.rs:68-70)
      const bool entering = dot(hit.n, s.d) < 0.0;   // against the outward normal (dielectric only)
      if (!first_hit && dot(hit.n, s.d) > 0.0) hit.n = neg(hit.n);
      const uint32_t hit_index = level;
      if (m.type == RT_MAT_EMITTER) {            // src/scene.rs:56-62
        if (m.present & RT_HAS_EMITTED) {
          if (o.emit_counts) { r.emits = true; r.emit_info = uint32_t(m.emitted) | (hit_index << 8); }
          else { emit_sample(sc.spectra, m.emitted, s.chain, hit_index, xyz, spectrum_out); contributes = true; }
        }
      } else if (remaining > 1) {
        // the continuing materials choose a direction, a side of the surface and a factor; the new ray is formed once
        // for all of them, so a wave with several materials runs the normalisation once
        V3 dir = s.d, side = hit.n;
        uint32_t factor = (m.present & RT_HAS_SPECULAR) ? uint32_t(m.specular) : uint32_t(RT_NO_SPECTRUM);
        bool continues = true;
        uint32_t pixel = s.pixel, sample = s.sample;
        if (!IDS && (m.type != RT_MAT_SPECULAR)) { uint32_t x, y; path_ids(b, s.slot, pixel, sample, x, y); }
        if (m.type == RT_MAT_LAMBERTIAN) {       // src/scene.rs:63-79
          continues = (m.present & RT_HAS_AMBIENT) != 0;
          if (continues) {
            double r1, r2;
            uniform_pair(b.seed, pixel, sample, hit_index, 1, r1, r2);
            dir = cosine_direction(hit.n, r1, r2);
            factor = m.ambient;
          }
        } else if (EXT && m.type == RT_MAT_GLOSSY) {      // extension: Phong lobe about the mirror direction
          const V3 axis = normalize(s.d - hit.n * (2.0 * dot(s.d, hit.n)));
          double r1, r2;
          uniform_pair(b.seed, pixel, sample, hit_index, 1, r1, r2);
          dir = phong_lobe_direction(axis, double(m.ns), r1, r2);
          continues = dot(dir, hit.n) > 0.0;     // a sample under the surface is absorbed
        } else if (EXT && m.type == RT_MAT_DIELECTRIC) {  // extension: smooth dielectric with Schlick's reflectance
          const V3 nf = dot(hit.n, s.d) > 0.0 ? neg(hit.n) : hit.n;
          double u0, u1;
          uniform_pair(b.seed, pixel, sample, hit_index, 1, u0, u1);
          const bool reflected = dielectric_direction(s.d, nf, entering, double(m.ni), u0, dir);
          side = reflected ? nf : neg(nf);       // a refracted ray starts on the far side
        } else {                                 // Specular, src/scene.rs:80-112
          dir = s.d - hit.n * (2.0 * dot(s.d, hit.n));
        }
        if (continues) {
          r.nd = normalize(dir);
          r.no = hit.p + side * 1e-10;
          chain_set(s.chain, hit_index, factor);
          r.alive = true;
        }
      }
    }
  }
  if (contributes) {
    o.contrib_x[s.slot] = xyz[0]; o.contrib_y[s.slot] = xyz[1]; o.contrib_z[s.slot] = xyz[2];
  } else if (spectrum_out && !r.alive && !r.emits) {
    for (int i = 0; i < RT_NBINS; ++i) spectrum_out[i] = 0.0f;
  }
  return r;
}

// What a wave appends at the end of a bounce goes to up to three sharded lists (emitter hits for emit_kernel, and
// the continuing paths in one or two classes). Their places are reserved with ONE atomic instruction: lane j of the
// wave adds list j's count to that list's fill counter of region chunk % kShards, so the wave waits for one round
// trip to the L2 instead of three in a row. Must be reached by all lanes of the wave.
struct Appends {
  unsigned long long emit_mask, first_mask, second_mask;   // the lanes that append to each list
  uint32_t emit_start, first_start, second_start;          // where the wave's run begins inside the region
};

__device__ __forceinline__ Appends reserve_appends(uint32_t* emit_counts, uint32_t* first_counts, uint32_t* second_counts,
                                                   uint32_t region_entries, uint32_t chunk, bool emits, bool first, bool second) {
  Appends a;
  a.emit_mask = __ballot(emits); a.first_mask = __ballot(first); a.second_mask = __ballot(second);
  a.emit_start = a.first_start = a.second_start = 0;
  if (!(a.emit_mask | a.first_mask | a.second_mask)) return a;
  const uint32_t lane = __lane_id();
  const unsigned long long mine = lane == 0 ? a.emit_mask : (lane == 1 ? a.first_mask : a.second_mask);
  uint32_t* counter = (lane == 0 ? emit_counts : (lane == 1 ? first_counts : second_counts)) + (chunk % kShards) * kCountStride;
  const uint32_t want = uint32_t(__popcll(mine));
  uint32_t start = 0;
  if (lane < 3 && want) {
    start = atomicAdd(counter, want);
    // the queue regions are sized so that this cannot happen; the emit list has its own way out (emit_or_append)
    if (lane != 0 && start + want > region_entries) counter[kOverflowWord] = 1u;
  }
  a.emit_start = __shfl(start, 0); a.first_start = __shfl(start, 1); a.second_start = __shfl(start, 2);
  return a;
}

__device__ __forceinline__ uint32_t rank_in(unsigned long long mask) {
  return uint32_t(__popcll(mask & ((1ull << __lane_id()) - 1ull)));
}

// Emitter hits go to the emit queue. A wave that finds its region full marks what it reserved as unused and forms
// its spectra in place instead.
__device__ __forceinline__ void emit_or_append(const DevScene& sc, const TraceOutputs& o, uint32_t chunk, bool deep, const Appends& a,
                                               bool emits, uint32_t emit_info, uint32_t slot, const Chain& chain) {
  if (!emits) return;
  const uint32_t want = uint32_t(__popcll(a.emit_mask)), rank = rank_in(a.emit_mask);
  const uint32_t at = (chunk % kShards) * o.emit_region_entries + a.emit_start + rank;
  if (a.emit_start + want <= o.emit_region_entries) {
    o.emit.slot[at] = slot; o.emit.info[at] = emit_info; o.emit.chain0[at] = chain.a;
    if (deep) o.emit.chain1[at] = chain.b;
  } else {
    if (a.emit_start + rank < o.emit_region_entries) o.emit.slot[at] = RT_MISS;
    float xyz[3];
    emit_sample(sc.spectra, emit_info & 0xFFu, chain, emit_info >> 8, xyz, o.spectra ? o.spectra + size_t(slot) * RT_NBINS : nullptr);
    o.contrib_x[slot] = xyz[0]; o.contrib_y[slot] = xyz[1]; o.contrib_z[slot] = xyz[2];
  }
}

// Position inside its region of a lane that appends to a list whose run starts at `start`; RT_MISS past the region's end,
// so nothing is ever written outside a region.
__device__ __forceinline__ uint32_t place_in(unsigned long long mask, uint32_t start, uint32_t region_entries) {
  const uint32_t within = start + rank_in(mask);
  return within < region_entries ? within : RT_MISS;
}

// Wave-level compaction of the lanes with `take` set into region chunk % kShards of a sharded
// list: one ballot, one atomic per wave, takers packed in lane order. Returns the position the
// lane's element goes to inside its region (RT_MISS for a lane that does not take part). The regions
// are sized so that a position past the region's end cannot occur; should it, the wave flags the
// region and the lanes past the end are told RT_MISS, so nothing is ever written outside a region.
// Must be reached by all lanes of the wave.
__device__ __forceinline__ uint32_t reserve_offset(uint32_t* counts, uint32_t region_entries, uint32_t chunk, bool take) {
  const unsigned long long mask = __ballot(take);
  if (!mask) return RT_MISS;
  const uint32_t lane = __lane_id();
  const uint32_t leader = uint32_t(__ffsll((long long)mask)) - 1u;
  const uint32_t region = chunk % kShards;
  uint32_t start = 0;
  if (lane == leader) {
    const uint32_t want = uint32_t(__popcll(mask));
    start = atomicAdd(counts + region * kCountStride, want);
    if (start + want > region_entries) counts[region * kCountStride + kOverflowWord] = 1u;
  }
  start = __shfl(start, int(leader));
  const uint32_t within = start + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
  return take && within < region_entries ? within : RT_MISS;
}

__device__ __forceinline__ uint32_t reserve_in_region(uint32_t* counts, uint32_t region_entries, uint32_t chunk, bool take) {
  const uint32_t within = reserve_offset(counts, region_entries, chunk, take);
  return within == RT_MISS ? RT_MISS : (chunk % kShards) * region_entries + within;
}

// COUNT builds: adds 64 to `w` once per wave-level execution of the place it is called from
__device__ __forceinline__ void count_slot(uint32_t& w) {
  const unsigned long long m = __ballot(true);
  if (__lane_id() == uint32_t(__ffsll((long long)m)) - 1u) w += 64u;
}

// chunk -> position of its first entry's lane in the input queue; false for a lane past the fill
__device__ __forceinline__ bool queue_position(const uint32_t* s_prefix, const uint32_t* s_fill, uint32_t region_entries,
                                               uint32_t chunk, uint32_t& idx) {
  const uint32_t lo = find_region(s_prefix, chunk);
  const uint32_t within = (chunk - s_prefix[lo]) * kBlock + threadIdx.x;
  idx = lo * region_entries + within;
  return within < s_fill[lo];
}

// Two-kernel pipeline: a queue level keeps its paths in two classes per region. Class "mesh" (the new ray meets a
// mesh's box, so traverse_kernel has work on it) fills a region from its front, class "other" from its back; each
// has its own fill counter. traverse_kernel then reads its entries contiguously without a list, and shade_kernel's
// waves are all-mesh or all-other, which keeps the lanes of a wave on the same shading branches.
//   pseudo-region j < kShards : mesh class of region j, entry w at j * cap + w
//   pseudo-region j >= kShards: other class of region j - kShards, entry w at (j - kShards) * cap + cap - 1 - w
__device__ __forceinline__ uint32_t number_chunks2(const uint32_t* __restrict__ mesh_counts, const uint32_t* __restrict__ other_counts,
                                                   uint32_t region_entries, uint32_t* s_prefix, uint32_t* s_fill) {
  // one wave per class scans its kShards counters
  if (threadIdx.x < 2 * kShards) {
    const uint32_t cls = threadIdx.x / kShards, r = threadIdx.x % kShards;
    uint32_t fill = (cls ? other_counts : mesh_counts)[r * kCountStride];
    fill = fill < region_entries ? fill : region_entries;
    uint32_t scan = (fill + kBlock - 1) / kBlock;
    for (int off = 1; off < int(kShards); off <<= 1) {
      const uint32_t up = __shfl_up(scan, off);
      if (int(r) >= off) scan += up;
    }
    s_fill[threadIdx.x] = fill;
    s_prefix[threadIdx.x + 1] = scan;     // per class for now
    if (threadIdx.x == 0) s_prefix[0] = 0;
  }
  __syncthreads();
  const uint32_t mesh_chunks = s_prefix[kShards];
  __syncthreads();
  if (threadIdx.x >= kShards && threadIdx.x < 2 * kShards) s_prefix[threadIdx.x + 1] += mesh_chunks;
  __syncthreads();
  return s_prefix[2 * kShards];
}

// The 2 * kShards prefix and fill values, two of each per lane of a wave: the pseudo-region of a chunk is then found
// with two wave-wide comparisons instead of a binary search through LDS.
struct ChunkMap { uint32_t prefix_lo, prefix_hi, fill_lo, fill_hi; };

__device__ __forceinline__ ChunkMap load_chunk_map(const uint32_t* s_prefix, const uint32_t* s_fill) {
  const uint32_t lane = threadIdx.x % 64;
  ChunkMap m;
  m.prefix_lo = s_prefix[lane]; m.prefix_hi = s_prefix[kShards + lane];
  m.fill_lo = s_fill[lane]; m.fill_hi = s_fill[kShards + lane];
  return m;
}

// chunk (the same for the whole wave, all of whose lanes must be here) -> the lane's queue entry
__device__ __forceinline__ bool queue_position2(const ChunkMap& m, uint32_t region_entries, uint32_t chunk, uint32_t& idx) {
  static_assert(kShards == 64, "one prefix value of each class per lane");
  // the last pseudo-region whose first chunk is <= chunk: prefixes never decrease and the first is 0
  const uint32_t at_or_before = uint32_t(__popcll(__ballot(m.prefix_lo <= chunk))) + uint32_t(__popcll(__ballot(m.prefix_hi <= chunk)));
  const int lo = __builtin_amdgcn_readfirstlane(int(at_or_before) - 1);
  const bool other = lo >= int(kShards);
  const int holder = other ? lo - int(kShards) : lo;
  const uint32_t first = uint32_t(__builtin_amdgcn_readlane(int(other ? m.prefix_hi : m.prefix_lo), holder));
  const uint32_t fill = uint32_t(__builtin_amdgcn_readlane(int(other ? m.fill_hi : m.fill_lo), holder));
  const uint32_t within = (chunk - first) * kBlock + threadIdx.x;
  idx = other ? uint32_t(holder) * region_entries + (region_entries - 1u - within) : uint32_t(holder) * region_entries + within;
  return within < fill;
}

// ------------------------------------------------------------------------------- the fused bounce kernel

// The dense kernels take their arguments as one struct and read its fields through the kernel-argument pointer
// where they are used (scalar loads that hit the constant cache) instead of as separate by-value parameters,
// which the compiler loads once at entry and then has to keep in -- and spill from -- the 100-odd scalar registers.
struct BounceArgs {
  DevScene sc; BatchDesc b; PathQueue in; const uint32_t* in_counts; uint32_t region_entries; PathQueue out;
  uint32_t* out_counts; TraceOutputs o; TraversalCounters* counters; uint32_t level;
};
typedef const __attribute__((address_space(4))) BounceArgs* BounceArgsPtr;

// One bounce for every path of a queue: closest hit, shading, append of the continuing paths.
// PRIMARY: the path starts here (camera ray from the Philox jitter, or a caller ray); otherwise
// its state comes from `in`. Used for scenes without meshes, where every lane does the same work.
// Held to 128 registers (four waves per SIMD): the standard materials need a few more than that, the extension
// materials about 40 more, and the spilled values cost less than the fourth wave brings (measured +9 % and +6 %).
template <bool PRIMARY, int RULE, bool COUNT, bool EXT, bool MESH = true>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
bounce_kernel(BounceArgs a) {
  const BounceArgsPtr ap = (BounceArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
  const DevScene& sc = *(const DevScene*)&ap->sc;
  const BatchDesc& b = *(const BatchDesc*)&ap->b;
  const PathQueue& in = *(const PathQueue*)&ap->in;
  const PathQueue& out = *(const PathQueue*)&ap->out;
  const TraceOutputs& o = *(const TraceOutputs*)&ap->o;
  const uint32_t* __restrict__ in_counts = ap->in_counts;
  uint32_t* __restrict__ out_counts = ap->out_counts;
  const uint32_t region_entries = ap->region_entries;
  TraversalCounters* counters = ap->counters;
  const uint32_t level = PRIMARY ? 0u : ap->level;   // ray casts the paths of `in` have made
  extern __shared__ uint4 lds_dynamic[];
  __shared__ uint32_t s_chunk_prefix[kShards + 1];  // chunks of kBlock paths before each input region
  __shared__ uint32_t s_region_fill[kShards];
  uint32_t* stack;
  const uint4* node_words = stage_nodes<false>(sc, lds_dynamic, stack);
  const bool deep = b.max_depth > 16;
  const uint32_t n_start = b.rays ? b.n_rays : b.n_local_pixels * b.n_samples;
  const uint32_t n_chunks = PRIMARY ? (n_start + kBlock - 1) / kBlock
                                    : number_chunks(in_counts, region_entries, s_chunk_prefix, s_region_fill);
  uint32_t n_nodes = 0, n_prims = 0;

  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    uint32_t idx;
    bool active;
    if (PRIMARY) { idx = chunk * kBlock + threadIdx.x; active = idx < n_start; }
    else active = queue_position(s_chunk_prefix, s_region_fill, region_entries, chunk, idx);
    PathState ps;
    ps.o = mk(0, 0, 0); ps.d = mk(1, 0, 0); ps.pixel = ps.sample = ps.slot = 0;
    ps.chain.a = make_uint4(0, 0, 0, 0); ps.chain.b = make_uint4(0, 0, 0, 0);
    if (active) {
      if (PRIMARY) active = start_path(b, idx, ps);
      else load_path(in, idx, chain_words(level), ps);
    }
    Shaded sh;
    sh.alive = false; sh.emits = false; sh.emit_info = 0; sh.no = ps.o; sh.nd = ps.d;
    if (active) {
      Hit hit;
      const bool found = closest_hit<RULE, COUNT, MESH>(sc, node_words, ps.o, ps.d, stack, hit, n_nodes, n_prims);
      sh = shade_hit<EXT, PRIMARY>(sc, b, o, found, hit, level, ps);
    }
    const Appends ap3 = reserve_appends(o.emit_counts, out_counts, out_counts, region_entries, chunk, sh.emits, sh.alive, false);
    emit_or_append(sc, o, chunk, deep, ap3, sh.emits, sh.emit_info, ps.slot, ps.chain);
    const uint32_t within = sh.alive ? place_in(ap3.first_mask, ap3.first_start, region_entries) : RT_MISS;
    if (within != RT_MISS)
      queue_store(out, (chunk % kShards) * region_entries + within, sh.no, sh.nd, ps.slot, ps.chain, chain_words(level + 1));
  }

  if (COUNT) {
    unsigned long long a = n_nodes, c = n_prims;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); c += __shfl_down(c, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters->nodes, a); atomicAdd(&counters->prims, c); }
  }
}

// ------------------------------------------------------------------------------- the two-kernel pipeline

// Level 0 of a batch needs no queue: path idx can be recreated from idx alone (start_path). This kernel
// only finds the primary rays that meet a mesh's box, stores their direction (and origin, for caller-supplied
// rays) at position idx of `q` for traverse_kernel, lists idx, and clears every path's hit record.
__global__ void __launch_bounds__(kBlock)
generate_kernel(DevScene sc, BatchDesc b, PathQueue q, uint32_t region_entries, MeshList mesh) {
  const uint32_t n_start = b.rays ? b.n_rays : b.n_local_pixels * b.n_samples;
  const uint32_t n_chunks = (n_start + kBlock - 1) / kBlock;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t idx = chunk * kBlock + threadIdx.x;
    bool meets = false;
    if (idx < n_start) {
      q.hit_obj()[idx] = RT_MISS;
      PathState ps;
      if (start_path(b, idx, ps) && meets_any_mesh(sc, ps.o, ps.d)) {
        meets = true;
        q.oz_dx()[idx] = make_double2(ps.o.z, ps.d.x); q.d_yz()[idx] = make_double2(ps.d.y, ps.d.z);
        if (b.rays) q.o_xy()[idx] = make_double2(ps.o.x, ps.o.y);
      }
    }
    const uint32_t listed = reserve_in_region(mesh.counts, region_entries, chunk, meets);
    if (listed != RT_MISS) mesh.entries[listed] = idx;
  }
}

// Can any ray through pixel (x, y) -- whatever its jitter -- meet the bounding sphere of a mesh? The rays of a pixel
// lie in a cone about the ray through the pixel's centre whose half-angle is at most (half the pixel's diagonal on
// the screen) / (distance from the focus to the pixel); a mesh is out of reach when its centre direction is further
// from the axis than that plus the sphere's angular radius. Angles are compared through sines and cosines with a
// 50 % allowance on the cone and 1e-6 on the comparison, so the answer errs towards "yes".
__device__ __forceinline__ bool pixel_may_see_mesh(const DevScene& sc, const DevCamera& cam, uint32_t x, uint32_t y) {
  const V3 p1 = ld3(cam.p1), f = ld3(cam.focus);
  const V3 ex = (ld3(cam.p2) - p1) * (1.0 / double(cam.width)), ey = (ld3(cam.p3) - p1) * (1.0 / double(cam.height));
  const V3 axis = ((p1 + ex * double(x)) + ey * double(y)) - f;
  const double axis_len = sqrt(dot(axis, axis));
  const double half_diag = 0.5 * (sqrt(dot(ex, ex)) + sqrt(dot(ey, ey)));
  const double sin_a = 1.5 * half_diag / axis_len;           // >= sin of the cone's half-angle
  if (!(sin_a < 0.1)) return true;                            // enormous pixels: no culling
  const double cos_a = 1.0 - sin_a * sin_a;                   // <= cos of it
  for (uint32_t k = 0; k < sc.n_meshes; ++k) {
    const DevObject ob = load_uniform(sc.objects + load_uniform(sc.mesh_objects + k));
    const V3 v = ld3(ob.center) - f;
    const double dist = sqrt(dot(v, v));
    const double sin_b = ob.radius * 1.000001 / dist;         // sine of the sphere's angular radius
    if (!(sin_b < 0.999)) return true;                        // the focus is inside or next to the sphere
    const double cos_b = sqrt(1.0 - sin_b * sin_b);
    const double cos_t = dot(axis, v) / (axis_len * dist);    // cosine of the angle between axis and centre
    if (cos_t >= (cos_a * cos_b - sin_a * sin_b) - 1e-6) return true;   // angle <= a + b
  }
  return false;
}

// The same for the camera rays of a batch, one lane per pixel looping over the pixel's samples: pixels that cannot see
// a mesh (most of them, as a rule) only clear their hit records and never generate a ray here.
__global__ void __launch_bounds__(kBlock)
generate_camera_kernel(DevScene sc, BatchDesc b, PathQueue q, uint32_t region_entries, MeshList mesh) {
  const uint32_t n_chunks = (b.n_local_pixels + kBlock - 1) / kBlock;
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t lp = chunk * kBlock + threadIdx.x;
    uint32_t x = 0, y = 0;
    const bool inside = lp < b.n_local_pixels && local_pixel_to_xy(b, lp, x, y);
    const bool may_see = inside && pixel_may_see_mesh(sc, b.cam, x, y);
    for (uint32_t smp = 0; smp < b.n_samples; ++smp) {
      const uint32_t idx = smp * b.n_local_pixels + lp;
      bool meets = false;
      if (inside) q.hit_obj()[idx] = RT_MISS;
      if (may_see) {
        PathState ps;
        if (start_path(b, idx, ps) && meets_any_mesh(sc, ps.o, ps.d)) {
          meets = true;
          q.oz_dx()[idx] = make_double2(ps.o.z, ps.d.x); q.d_yz()[idx] = make_double2(ps.d.y, ps.d.z);
        }
      }
      // chunk * n_samples + smp runs over as many values as the batch has chunks of paths, which is what the regions are sized for
      const uint32_t listed = reserve_in_region(mesh.counts, region_entries, chunk * b.n_samples + smp, meets);
      if (listed != RT_MISS) mesh.entries[listed] = idx;
    }
  }
}

// Dense half of a bounce, for the paths of level k (START: level 0, recreated from their index; otherwise read
// from `in`): the exact sphere / lone-triangle tests of the path's ray, folded with what traverse_kernel found
// in the meshes (smaller reported t wins, the earlier object on a tie -- the order the reference's sequential
// strict-'<' scan produces), hit point and normal for the winner only, shading, and the append of the
// continuing paths to `out`. A new ray that meets a mesh's box is listed for the next traverse_kernel.
#ifndef RT_PREFETCH
#define RT_PREFETCH 1
#endif

struct ShadeArgs {
  DevScene sc; BatchDesc b; PathQueue in; LevelCounts in_counts; uint32_t region_entries; PathQueue out;
  LevelCounts out_counts; TraceOutputs o; uint32_t level;
};
typedef const __attribute__((address_space(4))) ShadeArgs* ShadeArgsPtr;

template <bool START, bool EXT>
__device__ __forceinline__ void shade_paths(ShadeArgsPtr ap) {
  __shared__ uint32_t s_chunk_prefix[2 * kShards + 1];
  __shared__ uint32_t s_region_fill[2 * kShards];
  uint32_t n_chunks;
  {
    const BatchDesc& b0 = *(const BatchDesc*)&ap->b;
    const uint32_t n_start0 = b0.rays ? b0.n_rays : b0.n_local_pixels * b0.n_samples;
    n_chunks = START ? (n_start0 + kBlock - 1) / kBlock
                     : number_chunks2(ap->in_counts.mesh, ap->in_counts.other, ap->region_entries, s_chunk_prefix, s_region_fill);
  }

  const ChunkMap chunk_map = START ? ChunkMap{0, 0, 0, 0} : load_chunk_map(s_chunk_prefix, s_region_fill);

  // Levels >= 1 read their paths one chunk ahead: the record of the chunk after this one is requested just before the
  // wave reserves its appends -- when little else is live in registers -- so its trip from memory overlaps the atomic's
  // round trip, the stores and the top of the next iteration instead of standing at the head of it.
  constexpr bool kAhead = !START && RT_PREFETCH;
  const uint32_t level = START ? 0u : ap->level;        // ray casts the paths of `in` have made
  const uint32_t words_in = chain_words(level);
  PathState ps_ahead;
  ps_ahead.o = mk(0, 0, 0); ps_ahead.d = mk(1, 0, 0); ps_ahead.pixel = ps_ahead.sample = ps_ahead.slot = 0;
  ps_ahead.chain.a = make_uint4(0, 0, 0, 0); ps_ahead.chain.b = make_uint4(0, 0, 0, 0);
  uint32_t idx_ahead = 0, obj_ahead = RT_MISS;
  bool active_ahead = false;
  if (kAhead && blockIdx.x < n_chunks) {
    const PathQueue& in0 = *(const PathQueue*)&ap->in;
    active_ahead = queue_position2(chunk_map, ap->region_entries, blockIdx.x, idx_ahead);
    if (active_ahead) { load_path(in0, idx_ahead, words_in, ps_ahead); obj_ahead = in0.hit_obj()[idx_ahead]; }
  }

  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const DevScene& sc = *(const DevScene*)&ap->sc;
    const BatchDesc& b = *(const BatchDesc*)&ap->b;
    const PathQueue& in = *(const PathQueue*)&ap->in;
    const PathQueue& out = *(const PathQueue*)&ap->out;
    const TraceOutputs& o = *(const TraceOutputs*)&ap->o;
    const LevelCounts out_counts = *(const LevelCounts*)&ap->out_counts;
    const uint32_t region_entries = ap->region_entries;
    const bool deep = b.max_depth > 16;
    const uint32_t n_start = b.rays ? b.n_rays : b.n_local_pixels * b.n_samples;
    uint32_t idx;
    bool active;
    if (START) { idx = chunk * kBlock + threadIdx.x; active = idx < n_start; }
    else if (kAhead) { idx = idx_ahead; active = active_ahead; }
    else active = queue_position2(chunk_map, region_entries, chunk, idx);
    PathState ps;
    ps.o = mk(0, 0, 0); ps.d = mk(1, 0, 0); ps.pixel = ps.sample = ps.slot = 0;
    ps.chain.a = make_uint4(0, 0, 0, 0); ps.chain.b = make_uint4(0, 0, 0, 0);
    Shaded sh;
    sh.alive = false; sh.emits = false; sh.emit_info = 0; sh.no = ps.o; sh.nd = ps.d;
    const uint32_t ahead_obj = obj_ahead;
    if (kAhead) {
      if (active) ps = ps_ahead;
    } else if (active) {
      if (START) active = start_path(b, idx, ps);
      else load_path(in, idx, words_in, ps);
    }
    if (active) {
      const V3 dn = normalize(ps.d);   // src/geometry/ray.rs:45
      Prelim pre;
      simple_objects(sc, ps.o, ps.d, dn, pre);
      Hit hit;
      hit.obj = pre.obj; hit.prim = 0; hit.t = pre.t;
      double t_raw = pre.t_raw;
      uint32_t tri_slot = 0;
      const uint32_t mesh_obj = kAhead ? ahead_obj : (sc.n_meshes ? in.hit_obj()[idx] : RT_MISS);
      if (mesh_obj != RT_MISS) {
        const double2 mesh_t = in.hit_t_raw()[idx];
        if (pre.obj == RT_MISS || mesh_t.x < pre.t || (mesh_t.x == pre.t && mesh_obj < pre.obj)) {
          const uint2 which = in.hit_prim_slot()[idx];
          hit.obj = mesh_obj; hit.t = mesh_t.x; hit.prim = which.x;
          t_raw = mesh_t.y; tri_slot = which.y;
        }
      }
      const bool found = hit.obj != RT_MISS;
      if (found) hit_surface(sc, hit.obj, ps.o, ps.d, dn, hit.t, t_raw, tri_slot, hit.p, hit.n, hit.mat);
      sh = shade_hit<EXT, START>(sc, b, o, found, hit, level, ps);
    }
    // a level keeps its paths in two classes per region: those whose ray meets a mesh's box fill it from the front,
    // the others from the back (see number_chunks2)
    const bool meets_mesh = sh.alive && meets_any_mesh(sc, sh.no, sh.nd);
    if (kAhead) {
      const uint32_t next_chunk = chunk + gridDim.x;
      active_ahead = false;
      if (next_chunk < n_chunks) {
        active_ahead = queue_position2(chunk_map, region_entries, next_chunk, idx_ahead);
        if (active_ahead) { load_path(in, idx_ahead, words_in, ps_ahead); obj_ahead = in.hit_obj()[idx_ahead]; }
      }
    }
    const Appends ap3 = reserve_appends(o.emit_counts, out_counts.mesh, out_counts.other, region_entries, chunk, sh.emits, meets_mesh,
                                        sh.alive && !meets_mesh);
    emit_or_append(sc, o, chunk, deep, ap3, sh.emits, sh.emit_info, ps.slot, ps.chain);
    uint32_t at = RT_MISS;
    if (sh.alive) {
      const uint32_t within = meets_mesh ? place_in(ap3.first_mask, ap3.first_start, region_entries)
                                         : place_in(ap3.second_mask, ap3.second_start, region_entries);
      if (within != RT_MISS) at = (chunk % kShards) * region_entries + (meets_mesh ? within : region_entries - 1u - within);
    }
    if (at != RT_MISS) {
      queue_store(out, at, sh.no, sh.nd, ps.slot, ps.chain, chain_words(level + 1));
      out.hit_obj()[at] = RT_MISS;
    }
  }
}

// The reference's three material types fit in 128 registers, i.e. four waves per SIMD (measured +4 % over the three
// the allocator settles for by itself); the variant with the extension materials needs about 170 and is left alone.
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4
#endif
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, RT_SHADE_WAVES)))
shade_kernel(ShadeArgs a) {
  shade_paths<false, false>((ShadeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr());
}

#ifndef RT_START_WAVES
#define RT_START_WAVES 4
#endif
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_START_WAVES, RT_START_WAVES)))
shade_start_kernel(ShadeArgs a) {
  shade_paths<true, false>((ShadeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr());
}

template <bool START>
__global__ void __launch_bounds__(kBlock)
shade_extended_kernel(ShadeArgs a) {
  shade_paths<START, true>((ShadeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr());
}

// Sparse half of a bounce: BVH traversal for the listed queue entries, by persistent waves. Each
// wave owns a contiguous share of the list; a lane that has finished its ray takes the next entry
// of the share, so a wave keeps working at near-full width while some of its rays take ten times
// as many steps as others. Lanes are serviced (retired, moved to the next mesh, refilled) in groups
// of at least kRefillLanes to keep that bookkeeping off the traversal's critical path.
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 40
#endif
#ifndef RT_MIN_WAVE_SHARE
#define RT_MIN_WAVE_SHARE 512
#endif
constexpr uint32_t kRefillLanes = RT_REFILL_LANES;
constexpr uint32_t kMinWaveShare = RT_MIN_WAVE_SHARE;   // entries; a smaller share leaves too little to refill from
#ifndef RT_MAX_ROUNDS
#define RT_MAX_ROUNDS 4
#endif
// Box-test steps a lane takes in a row once the wave has chosen the node phase: the choice itself (three ballots and
// their bookkeeping) costs half as much as a step, and a lane that reaches a leaf early just sits the rest out.
// Measured on the traversal launches of the Suzanne benchmark: 1.047 ms at 1, 0.982 at 2, 0.968 at 3, 0.977 at 4, 1.014 at 6.
#ifndef RT_NODE_UNROLL
#define RT_NODE_UNROLL 3
#endif
constexpr uint32_t kMaxRounds = RT_MAX_ROUNDS;

#ifndef RT_TRAVERSE_WAVES
#define RT_TRAVERSE_WAVES 5
#endif


template <int FMT, int RULE, bool COUNT, bool STAGED>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_TRAVERSE_WAVES, RT_TRAVERSE_WAVES)))
traverse_kernel(DevScene sc, PathQueue q, MeshList mesh, uint32_t region_entries, FixedOrigin origin, uint32_t resident_waves,
                TraversalCounters* counters) {
  typedef NodeFmt<FMT> F;
  extern __shared__ uint4 lds_dynamic[];
  __shared__ uint32_t s_prefix[kShards + 1];   // list entries before each region
  if (threadIdx.x < kShards) {
    uint32_t fill = mesh.counts[threadIdx.x * kCountStride];
    fill = fill < region_entries ? fill : region_entries;
    uint32_t scan = fill;
    for (int off = 1; off < int(kShards); off <<= 1) {
      const uint32_t up = __shfl_up(scan, off);
      if (int(threadIdx.x) >= off) scan += up;
    }
    s_prefix[threadIdx.x + 1] = scan;
    if (threadIdx.x == 0) s_prefix[0] = 0;
  }
  __syncthreads();
  const uint32_t total = s_prefix[kShards];
  // The list is cut into whole rounds of the waves the chip holds at once (a partial last round leaves most of the
  // chip idle while it drains): as many rounds as keep a share at kMinWaveShare entries or more, at most kMaxRounds.
  uint32_t rounds = total / (resident_waves * kMinWaveShare);
  rounds = rounds < 1 ? 1 : (rounds > kMaxRounds ? kMaxRounds : rounds);
  const uint32_t n_waves = resident_waves * rounds;
  uint32_t share = (total + n_waves - 1) / n_waves;
  share = share < 64u ? 64u : share;
  share = (share + 63u) & ~63u;
  // a short list leaves the later workgroups of the grid without a share: they leave before copying the tree
  if (uint64_t(blockIdx.x) * (kBlock / 64) * share >= total) return;
  uint32_t* stack32;
  const uint4* node_words = stage_nodes<STAGED>(sc, lds_dynamic, stack32);
  typename F::stack_t* stack = reinterpret_cast<typename F::stack_t*>(stack32 - threadIdx.x) + threadIdx.x;
  const uint64_t begin = uint64_t(blockIdx.x * (kBlock / 64) + threadIdx.x / 64) * share;
  uint32_t next = begin < total ? uint32_t(begin) : total;
  const uint32_t end = begin + share < total ? uint32_t(begin + share) : total;

  constexpr uint32_t kDone = F::kNone;     // has the leaf bit set, so it also ends the node loop
  const uint32_t lane = threadIdx.x & 63u;
  static_assert(kShards == 64, "one region's prefix per lane");
  const uint32_t my_prefix = s_prefix[lane], my_prefix_end = s_prefix[lane + 1];   // list entries before / through region `lane`
  uint32_t n_nodes = 0, n_prims = 0;
  uint32_t w_nodes = 0, w_prims = 0, w_service = 0, n_service = 0;   // COUNT only: wave-level iterations per phase
  uint32_t n_idle[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // COUNT only: see TraversalCounters::idle
  uint32_t entry = RT_MISS;                // queue entry whose ray this lane is working on
  V3 o = mk(0, 0, 0), dn = mk(1, 0, 0);
  QuantRay qr = {};
  uint32_t m = 0;                          // which mesh of the scene the lane is in
  uint32_t root = 0, first_tri = 0;
  MeshHit mh;                              // best inside the current mesh
  mh.t = INFINITY; mh.t_raw = 0.0; mh.index = RT_MISS; mh.slot = 0;
  double best_t = 0.0, best_raw = 0.0;     // best over the meshes done so far
  uint32_t best_obj = RT_MISS, best_prim = 0, best_slot = 0;
  uint32_t cur = kDone;
  constexpr uint32_t kNoLeaf = 0u;         // a leaf reference always has its flag bit set
  uint32_t lf = kNoLeaf;                   // the leaf whose triangles the lane is testing, counting itself down
  int sp = 0;

  for (;;) {
    // ---- service: every lane whose traversal has ended
    if (COUNT) { count_slot(w_service); if (cur == kDone && lf == kNoLeaf) ++n_service; }
    if (cur == kDone && lf == kNoLeaf && entry != RT_MISS) {
      if (RULE == RT_HIT_FIRST_INDEX && mh.index != RT_MISS) {
        mh.t = mh.t_raw / sc.mesh_frames[first_tri + mh.slot].len;
      }
      // meshes are visited in scene order, so strict '<' keeps the earlier object on a tie
      if (mh.index != RT_MISS && (best_obj == RT_MISS || mh.t < best_t)) {
        best_t = mh.t; best_raw = mh.t_raw; best_obj = sc.mesh_objects[m]; best_prim = mh.index; best_slot = mh.slot;
      }
      if (++m == sc.n_meshes) {
        if (best_obj != RT_MISS) {   // shade_kernel folds this with the spheres and lone triangles
          q.hit_t_raw()[entry] = make_double2(best_t, best_raw); q.hit_obj()[entry] = best_obj;
          q.hit_prim_slot()[entry] = make_uint2(best_prim, best_slot);
        }
        entry = RT_MISS;
      }
    }
    const unsigned long long idle = __ballot(entry == RT_MISS);
    if (idle && next < end) {
      const uint32_t g = next + uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
      // the idle lanes take consecutive list positions, nearly always inside one region: the region of the first is found
      // with one wave-wide comparison against the prefix value each lane holds, and only a run that crosses into the next
      // region searches
      const uint32_t first_region = uint32_t(__popcll(__ballot(my_prefix <= next))) - 1u;
      const uint32_t first_region_end = uint32_t(__builtin_amdgcn_readlane(int(my_prefix_end), int(first_region)));
      if (entry == RT_MISS && g < end) {
        const uint32_t region = g < first_region_end ? first_region : find_region(s_prefix, g);
        entry = region * region_entries + (g - s_prefix[region]);   // the mesh class of a level sits at the front of its regions
        if (mesh.entries) entry = mesh.entries[entry];               // level 0: positions of the listed primary rays
        const double2 oz_dx = q.oz_dx()[entry], d_yz = q.d_yz()[entry];
        if (origin.fixed) o = mk(origin.x, origin.y, origin.z);
        else { const double2 o_xy = q.o_xy()[entry]; o = mk(o_xy.x, o_xy.y, oz_dx.x); }
        dn = normalize(mk(oz_dx.y, d_yz.x, d_yz.y));  // src/geometry/ray.rs:45
        m = 0; best_obj = RT_MISS;
      }
      const uint32_t taken = uint32_t(__popcll(idle));
      next = end - next < taken ? end : next + taken;
    }
    if (cur == kDone && lf == kNoLeaf && entry != RT_MISS) {   // enter mesh m
      auto enter = [&](const DevObject& ob) {
        const BoxRay br = make_box_ray(o, dn, ob);
        qr = make_quant_ray(br, ob);
        root = ob.root; first_tri = ob.first;
        mh.t = INFINITY; mh.t_raw = 0.0; mh.index = RT_MISS; mh.slot = 0;
        cur = 0; sp = 0;
        if (!model_gate_certain(ob, br)) {
          // the reference's own box test of the model, on the ray as it was built (the direction before re-normalisation)
          const double2 oz_dx = q.oz_dx()[entry], d_yz = q.d_yz()[entry];
          if (!model_gate_exact(ob.gate_lo, ob.gate_hi, o, mk(oz_dx.y, d_yz.x, d_yz.y))) cur = kDone;
        }
      };
      // a scene with one mesh (the usual case): its record is the same for every lane and comes through scalar loads
      if (sc.n_meshes == 1) { const DevObject ob = load_uniform(sc.objects + load_uniform(sc.mesh_objects)); enter(ob); }
      else enter(sc.objects[sc.mesh_objects[m]]);
    }
    if (__ballot(entry != RT_MISS) == 0) break;
    const bool refillable = next < end;

    // ---- traverse until enough lanes can be given new work. A lane with a ray may need a box-test step (cur is an
    // inner node), a triangle test (it holds a leaf in `lf`), or both: a lane that reaches a leaf parks it in `lf` and
    // walks on, so it only stalls when it reaches a second leaf before the first is done. Each wave-level iteration
    // runs the step that more lanes are waiting for. (Letting every lane finish its run of nodes before any leaf was
    // touched left the node step at 27 % lane occupancy and the triangle step at 40 %.) `lf` counts itself down: after
    // each tested triangle its first-triangle field goes up and its count field down.
    for (;;) {
      if ((cur & F::kLeaf) && cur != kDone && lf == kNoLeaf) {   // park the leaf, keep walking
        lf = F::leaf(cur);
        cur = sp ? stack[(--sp) * kBlock] : kDone;
      }
      const bool walking = !(cur & F::kLeaf);
      const bool has_leaf = lf != kNoLeaf;
      const uint32_t n_walking = uint32_t(__popcll(__ballot(walking))), n_leaf = uint32_t(__popcll(__ballot(has_leaf)));
      if (n_walking + n_leaf == 0) break;
      const bool finished = cur == kDone && !has_leaf;
      const bool serviceable = finished && (refillable || (entry != RT_MISS && m + 1 < sc.n_meshes));
      if (uint32_t(__popcll(__ballot(serviceable))) >= kRefillLanes) break;
      if (n_walking >= n_leaf) {
#pragma unroll
        for (int rep = 0; rep < RT_NODE_UNROLL; ++rep) {
          if (COUNT && __ballot(!(cur & F::kLeaf)) != 0 && (cur & F::kLeaf))
            ++n_idle[entry == RT_MISS ? 1 : (cur != kDone ? (lf != kNoLeaf ? 2 : 7) : (lf != kNoLeaf ? 3 : 0))];
          if (!(cur & F::kLeaf)) {
          if (COUNT) count_slot(w_nodes);
          if (COUNT) ++n_nodes;
          const float tmax = RULE == RT_HIT_NEAREST ? __double2float_ru(mh.t) * 1.000002f : INFINITY;
          if (FMT == RT_NODES_QUAD) {
            // every reference was checked when the tree was built; the bound keeps a damaged one from becoming a wild fetch
            if (root + cur >= sc.n_nodes) { cur = kDone; sp = 0; continue; }
            const Quad c = test_children4(node_words + (size_t(root) + cur) * F::kWords4, qr, tmax);
            const int cap = int(sc.stack_entries);
            if (c.n > 3 && sp < cap) stack[(sp++) * kBlock] = typename F::stack_t(c.ref[3]);
            if (c.n > 2 && sp < cap) stack[(sp++) * kBlock] = typename F::stack_t(c.ref[2]);
            if (c.n > 1 && sp < cap) stack[(sp++) * kBlock] = typename F::stack_t(c.ref[1]);
            cur = c.n ? c.ref[0] : (sp ? stack[(--sp) * kBlock] : kDone);
            continue;
          }
          const Children c = test_children<FMT, RULE == RT_HIT_NEAREST>(node_words + (size_t(root) + cur) * F::kWords4, qr, tmax);
          const uint32_t ref0 = c.ref0, ref1 = c.ref1, min0 = c.min0, min1 = c.min1;
          const float tn0 = c.tn0, tn1 = c.tn1;
          bool h0 = c.h0, h1 = c.h1;
          if (RULE == RT_HIT_FIRST_INDEX) { h0 = h0 && min0 < mh.index; h1 = h1 && min1 < mh.index; }
          if (h0 && h1) {
            const bool swap = RULE == RT_HIT_FIRST_INDEX ? (min1 < min0) : (tn1 < tn0);
            stack[(sp++) * kBlock] = typename F::stack_t(swap ? ref0 : ref1);
            cur = swap ? ref1 : ref0;
          } else if (h0) {
            cur = ref0;
          } else if (h1) {
            cur = ref1;
          } else {
            cur = sp ? stack[(--sp) * kBlock] : kDone;
          }
          }
        }
      } else {
        if (COUNT && !has_leaf) ++n_idle[entry == RT_MISS ? 6 : (walking ? 4 : 5)];
        if (has_leaf) {
        if (COUNT) count_slot(w_prims);
        const uint32_t slot = (lf & ~RT_LEAF_FLAG) >> 4;
        const DevMeshTri* tp = sc.mesh_tris + first_tri + slot;
        const uint32_t index = tp->index;
        const double2* v = reinterpret_cast<const double2*>(tp->v);
        // First-index rule: the vertices are requested together with the index that decides whether they are needed -- one
        // trip to memory per triangle step instead of two in a row (a lane whose leaf turns out to be beaten wastes 72
        // bytes of an L2 hit; measured -2 % on the step). The nearest rule always needs them and waits for them one by one.
        double2 a = v[0], b = v[1], c = v[2], e = v[3];
        double last = tp->v[8];
        if (RULE == RT_HIT_FIRST_INDEX)
          asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(e.x), "+v"(e.y), "+v"(last));
        bool leaf_done = RULE == RT_HIT_FIRST_INDEX && index >= mh.index;   // a leaf is sorted by index
        if (!leaf_done) {
          if (COUNT) ++n_prims;
          double t_raw, t;
          if (RULE == RT_HIT_FIRST_INDEX) {
            // the winner is decided by the index alone: its reported t is formed once, when the lane is serviced
            if (moller_trumbore(mk(a.x, a.y, b.x), mk(b.y, c.x, c.y), mk(e.x, e.y, last), o, dn, t_raw)) {
              mh.t_raw = t_raw; mh.index = index; mh.slot = slot;
              leaf_done = true;
            }
          } else if (triangle_hit(mk(a.x, a.y, b.x), mk(b.y, c.x, c.y), mk(e.x, e.y, last), o, dn, sc.mesh_frames[first_tri + slot].len, t_raw, t)) {
            if (t < mh.t || (t == mh.t && index < mh.index)) {
              mh.t = t; mh.t_raw = t_raw; mh.index = index; mh.slot = slot;
            }
          }
          lf += 15u;                                   // first + 1 (bit 4 up), count - 1
          leaf_done = leaf_done || (lf & 15u) == 0u;
        }
        if (leaf_done) lf = kNoLeaf;
        }
      }
    }
  }

  if (COUNT) {
    unsigned long long v[14] = {n_nodes, n_prims, w_nodes, w_prims, w_service, n_service, n_idle[0], n_idle[1], n_idle[2], n_idle[3],
                                n_idle[4], n_idle[5], n_idle[6], n_idle[7]};
    unsigned long long* dst[14] = {&counters->nodes, &counters->prims, &counters->node_slots, &counters->prim_slots,
                                   &counters->service_slots, &counters->service_lanes, &counters->idle[0], &counters->idle[1],
                                   &counters->idle[2], &counters->idle[3], &counters->idle[4], &counters->idle[5], &counters->idle[6],
                                   &counters->idle[7]};
    for (int k = 0; k < 14; ++k) {
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
      if (lane == 0) atomicAdd(dst[k], v[k]);
    }
  }
}

// One thread per queued emitter hit: Le * (chain product) in the reference's order, then to_XYZ.
__global__ void __launch_bounds__(kBlock) emit_kernel(DevScene sc, TraceOutputs o, bool deep) {
  __shared__ uint32_t s_chunk_prefix[kShards + 1];
  __shared__ uint32_t s_region_fill[kShards];
  const uint32_t n_chunks = number_chunks(o.emit_counts, o.emit_region_entries, s_chunk_prefix, s_region_fill);
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint32_t region = find_region(s_chunk_prefix, chunk);
    const uint32_t within = (chunk - s_chunk_prefix[region]) * kBlock + threadIdx.x;
    if (within >= s_region_fill[region]) continue;
    const uint32_t at = region * o.emit_region_entries + within;
    const uint32_t slot = o.emit.slot[at];
    if (slot == RT_MISS) continue;
    const uint32_t info = o.emit.info[at];
    Chain chain;
    chain.a = o.emit.chain0[at];
    chain.b = deep ? o.emit.chain1[at] : make_uint4(0, 0, 0, 0);
    float xyz[3];
    emit_sample(sc.spectra, info & 0xFFu, chain, info >> 8, xyz, o.spectra ? o.spectra + size_t(slot) * RT_NBINS : nullptr);
    o.contrib_x[slot] = xyz[0]; o.contrib_y[slot] = xyz[1]; o.contrib_z[slot] = xyz[2];
  }
}

// Scene::intersect_ray over a batch of caller rays; writes rtmi_hit records.
struct HitOut { uint32_t obj, prim; double t; double p[3]; double n[3]; };
static_assert(sizeof(HitOut) == 64, "hit record must be 64 bytes");

template <int RULE, bool COUNT, bool STAGED>
__global__ void __launch_bounds__(kBlock)
intersect_kernel(DevScene sc, const double* __restrict__ rays, size_t n, HitOut* __restrict__ hits, TraversalCounters* counters) {
  extern __shared__ uint4 lds_dynamic[];
  uint32_t* stack;
  const uint4* node_words = stage_nodes<STAGED>(sc, lds_dynamic, stack);
  uint32_t n_nodes = 0, n_prims = 0;
  for (size_t idx = size_t(blockIdx.x) * kBlock + threadIdx.x; idx < n; idx += size_t(gridDim.x) * kBlock) {
    const V3 o = ld3(rays + idx * 6), d = ld3(rays + idx * 6 + 3);
    Hit hit;
    HitOut h;
    if (closest_hit<RULE, COUNT>(sc, node_words, o, d, stack, hit, n_nodes, n_prims)) {
      h.obj = hit.obj; h.prim = hit.prim; h.t = hit.t;
      h.p[0] = hit.p.x; h.p[1] = hit.p.y; h.p[2] = hit.p.z;
      h.n[0] = hit.n.x; h.n[1] = hit.n.y; h.n[2] = hit.n.z;
    } else {
      h.obj = RT_MISS; h.prim = RT_MISS; h.t = 0.0;
      h.p[0] = h.p[1] = h.p[2] = 0.0; h.n[0] = h.n[1] = h.n[2] = 0.0;
    }
    hits[idx] = h;
  }
  if (COUNT) {
    unsigned long long a = n_nodes, c = n_prims;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); c += __shfl_down(c, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters->nodes, a); atomicAdd(&counters->prims, c); }
  }
}

// Per pixel of the batch: sum the samples' XYZ in sample order onto the running per-pixel sum.
__global__ void __launch_bounds__(kBlock) resolve_kernel(BatchDesc b, TraceOutputs o, float* __restrict__ xyz_sum) {
  const uint32_t lp = blockIdx.x * kBlock + threadIdx.x;
  if (lp >= b.n_local_pixels) return;
  uint32_t x, y;
  if (!local_pixel_to_xy(b, lp, x, y)) return;
  const size_t pix = size_t(y) * b.cam.width + x;
  float sx = xyz_sum[pix * 3], sy = xyz_sum[pix * 3 + 1], sz = xyz_sum[pix * 3 + 2];
  for (uint32_t s = 0; s < b.n_samples; ++s) {
    const size_t slot = size_t(s) * b.n_local_pixels + lp;
    sx += o.contrib_x[slot]; sy += o.contrib_y[slot]; sz += o.contrib_z[slot];
  }
  xyz_sum[pix * 3] = sx; xyz_sum[pix * 3 + 1] = sy; xyz_sum[pix * 3 + 2] = sz;
}

__global__ void sum_counts_kernel(const uint32_t* counts, uint32_t n_levels, unsigned long long* total) {
  unsigned long long s = 0;
  for (uint32_t level = 0; level < n_levels; ++level) s += counts[(level * kShards + threadIdx.x) * kCountStride];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (threadIdx.x == 0) *total += s;
}

__global__ void __launch_bounds__(kBlock) fill_tiles_kernel(uint32_t* __restrict__ tiles, uint32_t n, uint32_t first, uint32_t stride) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < n) tiles[j] = first + j * stride;
}

__global__ void __launch_bounds__(kBlock) add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  for (size_t i = size_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += size_t(gridDim.x) * kBlock) dst[i] = dst[i] + src[i];
}

// One wave: lane r looks at region r of every level.
__global__ void check_overflow_kernel(const uint32_t* counts, uint32_t n_levels, uint32_t pair_level0, uint32_t n_pairs,
                                      uint32_t region_entries, uint32_t* flag) {
  bool bad = false;
  for (uint32_t level = 0; level < n_levels; ++level) bad = bad || counts[(level * kShards + threadIdx.x) * kCountStride + kOverflowWord] != 0u;
  for (uint32_t k = 0; k < n_pairs; ++k) {
    const unsigned long long both = (unsigned long long)counts[(k * kShards + threadIdx.x) * kCountStride] +
                                    counts[((pair_level0 + k) * kShards + threadIdx.x) * kCountStride];
    bad = bad || both > region_entries;
  }
  if (bad) *flag = 1u;
}

// mean, XYZ -> RGB, clamp, gamma 1/2.2, `(256 * v) as u8` (src/integrators/integrator.rs:79-90,
// src/spectrum/mod.rs:630-636). A Rust float->u8 cast saturates and maps NaN to 0.
__global__ void __launch_bounds__(kBlock) tonemap_kernel(const float* __restrict__ xyz_sum, float spp, uint32_t n_pixels,
                                                         uint8_t* __restrict__ rgb8) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pixels) return;
  const float x = (xyz_sum[i * 3] / spp), y = (xyz_sum[i * 3 + 1] / spp), z = (xyz_sum[i * 3 + 2] / spp);
  float rgb[3];
  rgb[0] = 3.2404790f * x - 1.537150f * y - 0.498535f * z;
  rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
  rgb[2] = 0.0556480f * x - 0.204043f * y + 1.057311f * z;
  for (int k = 0; k < 3; ++k) {
    float c = rgb[k];
    if (c < 0.0f) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    const float v = 256.0f * powf(c, 1.0f / 2.2f);
    uint8_t q;
    if (!(v == v) || v <= 0.0f) q = 0; else if (v >= 255.0f) q = 255; else q = uint8_t(v);
    rgb8[i * 3 + k] = q;
  }
}

inline uint32_t grid_for(size_t n) {
  const size_t blocks = (n + kBlock - 1) / kBlock;
  const size_t cap = 256 * 8;  // 256 CUs, up to 8 resident blocks each; longer queues are grid-strided
  return uint32_t(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

// LDS of a kernel that walks trees: the staged nodes (if any) and one stack column per lane. traverse_kernel keeps
// 16-bit entries under the small node format (`narrow`); the fused kernels always keep 32-bit ones.
inline size_t stack_bytes(const DevScene& sc, bool narrow = false) {
  const size_t entry = narrow && sc.node_format == RT_NODES_SMALL ? sizeof(uint16_t) : sizeof(uint32_t);
  return size_t(sc.lds_nodes) * sizeof(uint4) + size_t(sc.stack_entries) * kBlock * entry;
}

}  // namespace

// bounce_kernel<PRIM, RULE, COUNT, EXT> chosen from the scene's hit rule, whether traversal counters were asked
// for, and whether the scene has extension materials (the fused kernel reads BVH nodes from memory, never LDS)
#define RTMI_DISPATCH(KERNEL, PRIM, ...)                                                                    \
  do {                                                                                                      \
    if (sc.n_meshes == 0) {   /* no traversal code, no hit rule, nothing to count */                        \
      if (sc.extension_materials) hipLaunchKernelGGL((KERNEL<PRIM, 0, false, true, false>), grid, kBlock, lds, s, __VA_ARGS__); \
      else hipLaunchKernelGGL((KERNEL<PRIM, 0, false, false, false>), grid, kBlock, lds, s, __VA_ARGS__);   \
      break;                                                                                                \
    }                                                                                                       \
    const int sel_ = (sc.hit_rule == RT_HIT_NEAREST ? 4 : 0) | (counters ? 2 : 0) | (sc.extension_materials ? 1 : 0); \
    switch (sel_) {                                                                                         \
      case 0: hipLaunchKernelGGL((KERNEL<PRIM, 0, false, false>), grid, kBlock, lds, s, __VA_ARGS__); break; \
      case 1: hipLaunchKernelGGL((KERNEL<PRIM, 0, false, true>), grid, kBlock, lds, s, __VA_ARGS__); break;  \
      case 2: hipLaunchKernelGGL((KERNEL<PRIM, 0, true, false>), grid, kBlock, lds, s, __VA_ARGS__); break;  \
      case 3: hipLaunchKernelGGL((KERNEL<PRIM, 0, true, true>), grid, kBlock, lds, s, __VA_ARGS__); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<PRIM, 1, false, false>), grid, kBlock, lds, s, __VA_ARGS__); break; \
      case 5: hipLaunchKernelGGL((KERNEL<PRIM, 1, false, true>), grid, kBlock, lds, s, __VA_ARGS__); break;  \
      case 6: hipLaunchKernelGGL((KERNEL<PRIM, 1, true, false>), grid, kBlock, lds, s, __VA_ARGS__); break;  \
      default: hipLaunchKernelGGL((KERNEL<PRIM, 1, true, true>), grid, kBlock, lds, s, __VA_ARGS__); break;  \
    }                                                                                                       \
  } while (0)

void launch_primary(const DevScene& sc, const BatchDesc& b, PathQueue out, uint32_t* out_counts, uint32_t region_entries,
                    TraceOutputs o, TraversalCounters* counters, hipStream_t s) {
  const size_t n = b.rays ? size_t(b.n_rays) : size_t(b.n_local_pixels) * b.n_samples;
  if (n == 0) return;
  const dim3 grid(grid_for(n));
  const size_t lds = size_t(sc.stack_entries) * kBlock * sizeof(uint32_t);
  PathQueue none{};
  const BounceArgs a{sc, b, none, nullptr, region_entries, out, out_counts, o, counters, 0};
  RTMI_DISPATCH(bounce_kernel, true, a);
}

void launch_bounce(const DevScene& sc, const BatchDesc& b, uint32_t level, PathQueue in, const uint32_t* in_counts, uint32_t max_in,
                   PathQueue out, uint32_t* out_counts, uint32_t region_entries, TraceOutputs o,
                   TraversalCounters* counters, hipStream_t s) {
  if (max_in == 0) return;
  const dim3 grid(grid_for(size_t(max_in) + size_t(kShards) * kBlock));
  const size_t lds = size_t(sc.stack_entries) * kBlock * sizeof(uint32_t);
  const BounceArgs a{sc, b, in, in_counts, region_entries, out, out_counts, o, counters, level};
  RTMI_DISPATCH(bounce_kernel, false, a);
}

void launch_generate(const DevScene& sc, const BatchDesc& b, PathQueue q, uint32_t region_entries, MeshList mesh, hipStream_t s) {
  const size_t n = b.rays ? size_t(b.n_rays) : size_t(b.n_local_pixels) * b.n_samples;
  if (n == 0) return;
  if (b.rays) hipLaunchKernelGGL(generate_kernel, dim3(grid_for(n)), kBlock, 0, s, sc, b, q, region_entries, mesh);
  else hipLaunchKernelGGL(generate_camera_kernel, dim3(grid_for(b.n_local_pixels)), kBlock, 0, s, sc, b, q, region_entries, mesh);
}

void launch_shade(const DevScene& sc, const BatchDesc& b, uint32_t level, PathQueue in, LevelCounts in_counts, uint32_t max_in,
                  PathQueue out, LevelCounts out_counts, uint32_t region_entries, TraceOutputs o, hipStream_t s) {
  if (max_in == 0) return;
  const bool start = level == 0;
  const dim3 grid(grid_for(size_t(max_in) + (start ? 0 : 2 * size_t(kShards) * kBlock)));
  const int sel = (start ? 2 : 0) | (sc.extension_materials ? 1 : 0);
  const ShadeArgs a{sc, b, in, in_counts, region_entries, out, out_counts, o, level};
  switch (sel) {
    case 0: hipLaunchKernelGGL(shade_kernel, grid, kBlock, 0, s, a); break;
    case 1: hipLaunchKernelGGL((shade_extended_kernel<false>), grid, kBlock, 0, s, a); break;
    case 2: hipLaunchKernelGGL(shade_start_kernel, grid, kBlock, 0, s, a); break;
    default: hipLaunchKernelGGL((shade_extended_kernel<true>), grid, kBlock, 0, s, a); break;
  }
}

void launch_traverse(const DevScene& sc, PathQueue q, MeshList mesh, uint32_t max_in, uint32_t region_entries, FixedOrigin origin,
                     TraversalCounters* counters, hipStream_t s) {
  if (max_in == 0 || sc.n_meshes == 0) return;
  // the kernel cuts the list into whole rounds of `resident` waves (5 per SIMD at its 96 registers, unless the LDS of
  // the traversal stacks and staged nodes allows fewer workgroups per CU); launch enough workgroups for the most rounds
  const size_t lds = stack_bytes(sc, true);
  const uint32_t blocks_per_cu = uint32_t(std::min<size_t>(RT_TRAVERSE_WAVES, lds ? (160 * 1024) / (lds + 1024) : RT_TRAVERSE_WAVES));
  const uint32_t resident = 256u * std::max(1u, blocks_per_cu) * (kBlock / 64);
  const size_t waves = std::min<size_t>((size_t(max_in) + kMinWaveShare - 1) / kMinWaveShare, size_t(resident) * kMaxRounds);
  const dim3 grid(uint32_t(std::max<size_t>(1, (std::max<size_t>(waves, resident) + kBlock / 64 - 1) / (kBlock / 64))));
  // node format x hit rule (the nearest-only format never meets the first-index rule) x counters x staged nodes
#define RTMI_TRAVERSE(FMT, RULE)                                                                                                   \
  do {                                                                                                                             \
    const int sel_ = (counters ? 2 : 0) | (sc.lds_nodes ? 1 : 0);                                                                  \
    switch (sel_) {                                                                                                                \
      case 0: hipLaunchKernelGGL((traverse_kernel<FMT, RULE, false, false>), grid, kBlock, lds, s, sc, q, mesh, region_entries, origin, resident, counters); break; \
      case 1: hipLaunchKernelGGL((traverse_kernel<FMT, RULE, false, true>), grid, kBlock, lds, s, sc, q, mesh, region_entries, origin, resident, counters); break;  \
      case 2: hipLaunchKernelGGL((traverse_kernel<FMT, RULE, true, false>), grid, kBlock, lds, s, sc, q, mesh, region_entries, origin, resident, counters); break;  \
      default: hipLaunchKernelGGL((traverse_kernel<FMT, RULE, true, true>), grid, kBlock, lds, s, sc, q, mesh, region_entries, origin, resident, counters); break;  \
    }                                                                                                                              \
  } while (0)
  const bool nearest = sc.hit_rule == RT_HIT_NEAREST;
  if (sc.node_format == RT_NODES_SMALL) { if (nearest) RTMI_TRAVERSE(RT_NODES_SMALL, 1); else RTMI_TRAVERSE(RT_NODES_SMALL, 0); }
  else if (nearest && sc.node_format == RT_NODES_QUAD) RTMI_TRAVERSE(RT_NODES_QUAD, 1);
  else if (nearest) RTMI_TRAVERSE(RT_NODES_NEAREST, 1);
  else RTMI_TRAVERSE(RT_NODES_WIDE, 0);
#undef RTMI_TRAVERSE
}

void launch_intersect(const DevScene& sc, const void* rays_, size_t n, void* hits_, TraversalCounters* counters, hipStream_t s) {
  if (n == 0) return;
  const dim3 grid(grid_for(n));
  const size_t lds = stack_bytes(sc);
  const double* rays = static_cast<const double*>(rays_);
  HitOut* hits = static_cast<HitOut*>(hits_);
  const int sel = (sc.hit_rule == RT_HIT_NEAREST ? 4 : 0) | (counters ? 2 : 0) | (sc.lds_nodes ? 1 : 0);
  switch (sel) {
    case 0: hipLaunchKernelGGL((intersect_kernel<0, false, false>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 1: hipLaunchKernelGGL((intersect_kernel<0, false, true>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 2: hipLaunchKernelGGL((intersect_kernel<0, true, false>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 3: hipLaunchKernelGGL((intersect_kernel<0, true, true>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 4: hipLaunchKernelGGL((intersect_kernel<1, false, false>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 5: hipLaunchKernelGGL((intersect_kernel<1, false, true>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    case 6: hipLaunchKernelGGL((intersect_kernel<1, true, false>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
    default: hipLaunchKernelGGL((intersect_kernel<1, true, true>), grid, kBlock, lds, s, sc, rays, n, hits, counters); break;
  }
}

void launch_emit(const DevScene& sc, TraceOutputs o, bool deep, uint32_t max_entries, hipStream_t s) {
  if (!o.emit_counts || max_entries == 0) return;
  hipLaunchKernelGGL(emit_kernel, dim3(grid_for(size_t(max_entries) + size_t(kShards) * kBlock)), kBlock, 0, s, sc, o, deep);
}

void launch_resolve(const BatchDesc& b, TraceOutputs o, float* xyz_sum, hipStream_t s) {
  if (b.n_local_pixels == 0) return;
  hipLaunchKernelGGL(resolve_kernel, dim3((b.n_local_pixels + kBlock - 1) / kBlock), kBlock, 0, s, b, o, xyz_sum);
}

void launch_sum_counts(const uint32_t* counts, uint32_t n_levels, unsigned long long* total, hipStream_t s) {
  static_assert(kShards == 64, "sum_counts_kernel reduces one wave of kShards lanes");
  hipLaunchKernelGGL(sum_counts_kernel, dim3(1), 64, 0, s, counts, n_levels, total);
}

void launch_fill_tiles(uint32_t* tiles, uint32_t n, uint32_t first, uint32_t stride, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_tiles_kernel, dim3((n + kBlock - 1) / kBlock), kBlock, 0, s, tiles, n, first, stride);
}

void launch_add(float* dst, const float* src, size_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), kBlock, 0, s, dst, src, n);
}

void launch_check_overflow(const uint32_t* counts, uint32_t n_levels, uint32_t pair_level0, uint32_t n_pairs,
                           uint32_t region_entries, uint32_t* flag, hipStream_t s) {
  hipLaunchKernelGGL(check_overflow_kernel, dim3(1), 64, 0, s, counts, n_levels, pair_level0, n_pairs, region_entries, flag);
}

void launch_tonemap(const float* xyz_sum, uint32_t spp, uint32_t n_pixels, uint8_t* rgb8, hipStream_t s) {
  if (n_pixels == 0) return;
  hipLaunchKernelGGL(tonemap_kernel, dim3((n_pixels + kBlock - 1) / kBlock), kBlock, 0, s, xyz_sum, float(spp), n_pixels, rgb8);
}

}  // namespace rtmi
