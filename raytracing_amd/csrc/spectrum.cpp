// spectrum.cpp -- resampling of the colorimetric tables into 60 bins of 5 nm and the
// RGB <-> spectrum conversions a scene loader needs. All arithmetic is binary32 in the
// reference's operation order so that material spectra come out bit-identical.
// Build with -ffp-contract=off.
#include "spectrum.h"

#include <algorithm>
#include <mutex>

#include "spectral_tables.inc"

namespace rtmi {
namespace {

inline float mix(float a, float b, float t) { return (1.0f - t) * a + t * b; }  // src/auxiliar.rs:13

// Mean over [l0, l1] of the piecewise-linear curve through (lambda[i], value[i]), constant
// outside the sampled range (src/spectrum/mod.rs:320-383).
float band_average(const float* lambda, const float* value, int n, float l0, float l1) {
  if (n == 0) return 0.0f;
  if (n == 1) return value[0];
  const float first = lambda[0], last = lambda[n - 1];
  if (l1 <= first) return value[0];
  if (l0 >= last) return value[n - 1];
  float area = 0.0f;
  if (l0 < first) area += value[0] * (first - l0);
  if (l1 > last) area += value[n - 1] * (l1 - last);
  int seg = 0;
  while (l0 > lambda[seg + 1]) ++seg;
  for (; seg + 1 < n && l1 >= lambda[seg]; ++seg) {
    const float a = std::max(lambda[seg], l0), b = std::min(lambda[seg + 1], l1);
    const float width = lambda[seg + 1] - lambda[seg];
    const float va = mix(value[seg], value[seg + 1], (a - lambda[seg]) / width);
    const float vb = mix(value[seg], value[seg + 1], (b - lambda[seg]) / width);
    area += 0.5f * (va + vb) * (b - a);
  }
  return area / (l1 - l0);
}

SpectrumTables g_tables;
std::once_flag g_once;

void build_tables() {
  const float* refl[7] = {kSmitsReflWhite, kSmitsReflCyan, kSmitsReflMagenta, kSmitsReflYellow,
                          kSmitsReflRed, kSmitsReflGreen, kSmitsReflBlue};
  const float* illum[7] = {kSmitsIllumWhite, kSmitsIllumCyan, kSmitsIllumMagenta, kSmitsIllumYellow,
                           kSmitsIllumRed, kSmitsIllumGreen, kSmitsIllumBlue};
  const float* cie[3] = {kCieX, kCieY, kCieZ};
  for (int bin = 0; bin < RT_NBINS; ++bin) {
    const float l0 = mix(400.0f, 700.0f, float(bin) / float(RT_NBINS));
    const float l1 = mix(400.0f, 700.0f, float(bin + 1) / float(RT_NBINS));
    for (int c = 0; c < 3; ++c) g_tables.cie[c][bin] = band_average(kCieLambda, cie[c], RT_N_CIE_SAMPLES, l0, l1);
    for (int k = 0; k < 7; ++k) {
      g_tables.smits[0][k][bin] = band_average(kSmitsLambda, refl[k], RT_N_SMITS_SAMPLES, l0, l1);
      g_tables.smits[1][k][bin] = band_average(kSmitsLambda, illum[k], RT_N_SMITS_SAMPLES, l0, l1);
    }
  }
  g_tables.xyz_scale = (700.0f - 400.0f) / (kCieYIntegral * float(RT_NBINS));
}

inline void axpy(float* acc, const float* basis, float k) {
  for (int i = 0; i < RT_NBINS; ++i) acc[i] += basis[i] * k;
}

}  // namespace

const SpectrumTables& spectrum_tables() {
  std::call_once(g_once, build_tables);
  return g_tables;
}

// Smits (1999) decomposition: white * min + secondary * mid + primary * max
// (src/spectrum/mod.rs:525-581; strict '<' comparisons as there).
void spectrum_from_rgb(float r, float g, float b, int type, float out[RT_NBINS]) {
  const float(*s)[RT_NBINS] = spectrum_tables().smits[type ? 1 : 0];
  enum { W, C, M, Y, R, G, B };
  std::fill(out, out + RT_NBINS, 0.0f);
  if (r < g && r < b) {
    axpy(out, s[W], r);
    if (g < b) { axpy(out, s[C], g - r); axpy(out, s[B], b - g); }
    else       { axpy(out, s[C], b - r); axpy(out, s[G], g - b); }
  } else if (g < r && g < b) {
    axpy(out, s[W], g);
    if (r < b) { axpy(out, s[M], r - g); axpy(out, s[B], b - r); }
    else       { axpy(out, s[M], b - g); axpy(out, s[R], r - b); }
  } else {
    axpy(out, s[W], b);
    if (r < g) { axpy(out, s[Y], r - b); axpy(out, s[G], g - r); }
    else       { axpy(out, s[Y], g - b); axpy(out, s[R], r - g); }
  }
}

void spectrum_to_xyz(const float spec[RT_NBINS], float xyz[3]) {  // src/spectrum/mod.rs:590-609
  const SpectrumTables& t = spectrum_tables();
  float x = 0.0f, y = 0.0f, z = 0.0f;
  for (int i = 0; i < RT_NBINS; ++i) {
    x += t.cie[0][i] * spec[i];
    y += t.cie[1][i] * spec[i];
    z += t.cie[2][i] * spec[i];
  }
  xyz[0] = x * t.xyz_scale; xyz[1] = y * t.xyz_scale; xyz[2] = z * t.xyz_scale;
}

void xyz_to_rgb(const float xyz[3], float rgb[3]) {  // src/spectrum/mod.rs:630-636
  const float x = xyz[0], y = xyz[1], z = xyz[2];
  rgb[0] = 3.2404790f * x - 1.537150f * y - 0.498535f * z;
  rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
  rgb[2] = 0.0556480f * x - 0.204043f * y + 1.057311f * z;
}

}  // namespace rtmi
