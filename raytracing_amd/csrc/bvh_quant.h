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
