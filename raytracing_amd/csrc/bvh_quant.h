// bvh_quant.h -- the quantisation grid of a mesh's node boxes and the outward rounding onto it, shared by the host
// builder (bvh.cpp) and the device builder (lbvh.hip) so that both put a box on exactly the same levels.
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RTMI_HD __host__ __device__ inline
#else
#define RTMI_HD inline
#endif

namespace rtmi {

// largest level whose plane (grid_lo + level * step) is at least one step below v; levels are 0 .. 65535
RTMI_HD uint32_t quant_level_lo(double v, float grid_lo, float grid_step) {
  const double g = double(grid_lo), s = double(grid_step);
  long t = long(floor((v - g) / s)) - 1;
  while (t > 0 && g + double(t) * s > v - s) --t;
  return uint32_t(t < 0 ? 0 : (t > 65535 ? 65535 : t));
}

// smallest level whose plane is at least one step above v
RTMI_HD uint32_t quant_level_hi(double v, float grid_lo, float grid_step) {
  const double g = double(grid_lo), s = double(grid_step);
  long t = long(ceil((v - g) / s)) + 1;
  while (t < 65535 && g + double(t) * s < v + s) ++t;
  return uint32_t(t < 0 ? 0 : (t > 65535 ? 65535 : t));
}

// The grid for boxes that all lie within [lo, hi] (binary32 box coordinates, relative to the mesh centre) on axis k:
// 65 535 levels of which the boxes use 4 .. 65 531, so the outward safety steps never leave the range; an axis the mesh
// is flat along still gets a positive step. `widest` is the largest extent over the three axes.
inline void quant_grid_axis(double lo, double hi, double widest, float* grid_lo, float* grid_step) {
  if (!(hi >= lo)) { lo = hi = 0.0; }
  double step = (hi - lo) / 65520.0;
  if (step < widest * 0x1.0p-30) step = widest * 0x1.0p-30;
  if (step < 1e-30) step = 1e-30;
  float fs = float(step);
  if (double(fs) < step) fs = nextafterf(fs, FLT_MAX);
  float fl = float(lo - 6.0 * double(fs));
  if (double(fl) > lo - 6.0 * double(fs)) fl = nextafterf(fl, -FLT_MAX);
  *grid_lo = fl; *grid_step = fs;
}

// ---- binary16 planes (the RT_NODES_SMALL format). A plane is stored as coordinate / grid_step in binary16, grid_step a
// power of two that brings the mesh's largest coordinate (relative to its centre) to about 1000, so neither the
// binary16 range nor its subnormals come into play; grid_lo is 0. The kernels convert a plane inside the fused
// multiply-add that places it along the ray (v_fma_mix_f32), where the 16-bit integer levels need a conversion
// instruction each. A binary16 has 11 significant bits: a plane moves outward by at most 2^-10 of its distance from the
// mesh centre, which is little against the boxes of a small mesh (what this format is for).

RTMI_HD float half_bits_to_float(uint32_t h) {
  const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
  const float a = e == 0 ? ldexpf(float(m), -24) : ldexpf(float(1024u + m), int(e) - 25);
  return (h & 0x8000u) ? -a : a;
}

// nearest binary16 (bit pattern) of a finite binary32 value, clamped to the largest finite binary16
RTMI_HD uint32_t half_bits_nearest(float v) {
  const uint32_t sign = v < 0.0f || (v == 0.0f && 1.0f / v < 0.0f) ? 0x8000u : 0u;
  const float a = fabsf(v);
  if (a == 0.0f) return sign;
  if (!(a < 65504.0f)) return sign | 0x7BFFu;
  int e;
  const float m = frexpf(a, &e);                                   // a = m * 2^e, 0.5 <= m < 1
  if (e - 1 < -14) return sign | uint32_t(floorf(ldexpf(a, 24) + 0.5f));   // subnormal: units of 2^-24 (1024 of them is the first normal)
  uint32_t q = uint32_t(floorf(m * 2048.0f + 0.5f));               // 1024 .. 2048
  if (q == 2048u) { q = 1024u; ++e; }
  const uint32_t biased = uint32_t(e - 1 + 15);
  if (biased >= 31u) return sign | 0x7BFFu;
  return sign | biased << 10 | (q - 1024u);
}

RTMI_HD uint32_t half_bits_up(uint32_t h) {      // the next binary16 towards +inf (finite ones only)
  if (h & 0x8000u) return (h & 0x7FFFu) == 0u ? 0x0001u : h - 1u;
  return h >= 0x7BFFu ? 0x7BFFu : h + 1u;
}
RTMI_HD uint32_t half_bits_down(uint32_t h) {    // the next binary16 towards -inf
  if (h & 0x8000u) return h >= 0xFBFFu ? 0xFBFFu : h + 1u;
  return h == 0u ? 0x8001u : h - 1u;
}

// the binary16 plane at or below (lo) / at or above (hi) v / grid_step, then one more binary16 outward: the slack the
// kernels' binary32 slab arithmetic is allowed (the integer formats step one grid level outward for the same reason)
RTMI_HD uint32_t half_plane_lo(double v, float grid_step) {
  const float x = float(v / double(grid_step));
  uint32_t h = half_bits_nearest(x);
  while (double(half_bits_to_float(h)) * double(grid_step) > v && h != 0xFBFFu) h = half_bits_down(h);
  return half_bits_down(h);
}
RTMI_HD uint32_t half_plane_hi(double v, float grid_step) {
  const float x = float(v / double(grid_step));
  uint32_t h = half_bits_nearest(x);
  while (double(half_bits_to_float(h)) * double(grid_step) < v && h != 0x7BFFu) h = half_bits_up(h);
  return half_bits_up(h);
}

// grid of the binary16 planes on one axis: grid_lo = 0, grid_step = the power of two that maps the largest |coordinate|
// of the mesh (over all three axes: `reach`) into [512, 1024]
inline void half_grid_axis(double reach, float* grid_lo, float* grid_step) {
  if (!(reach > 1e-30)) reach = 1e-30;
  int e;
  frexp(reach, &e);                       // reach = m * 2^e, m in [0.5, 1): reach / 2^(e - 10) in [512, 1024)
  *grid_lo = 0.0f;
  *grid_step = float(ldexp(1.0, e - 10));
}

// 16-bit child reference of the RT_NODES_SMALL format from the common 32-bit form
RTMI_HD uint32_t small_child_ref(uint32_t ref) {
  if (ref == 0xFFFFFFFFu) return 0xFFFFu;
  if (ref & 0x80000000u) {
    const uint32_t first = (ref & 0x7FFFFFFFu) >> 4, count = ref & 15u;   // count 1..4
    return 0x8000u | (first << 2) | (count - 1u);
  }
  return ref;   // node index < 0x8000
}

// One node in the device's form (rt_types.h: RT_NODES_*): b = the 12 box levels lo0[3], hi0[3], lo1[3], hi1[3].
template <class Word4>
RTMI_HD void pack_node(Word4* w, int format, const uint32_t b[12], uint32_t ref0, uint32_t ref1, uint32_t min0, uint32_t min1) {
  w[0].x = b[0] | b[1] << 16; w[0].y = b[2] | b[3] << 16; w[0].z = b[4] | b[5] << 16; w[0].w = b[6] | b[7] << 16;
  w[1].x = b[8] | b[9] << 16; w[1].y = b[10] | b[11] << 16;
  if (format == 0 /* RT_NODES_SMALL */) {
    w[1].z = small_child_ref(ref0) | small_child_ref(ref1) << 16;
    w[1].w = (min0 < 0xFFFFu ? min0 : 0xFFFFu) | (min1 < 0xFFFFu ? min1 : 0xFFFFu) << 16;
  } else {
    w[1].z = ref0; w[1].w = ref1;
    if (format == 2 /* RT_NODES_WIDE */) { w[2].x = min0; w[2].y = min1; w[2].z = 0; w[2].w = 0; }
  }
}

}  // namespace rtmi
