// spectrum.h -- host side of the 60-bin sampled spectrum (load-time work only).
#pragma once
#include "rt_types.h"

namespace rtmi {

// The 17 resampled tables SampledSpectrum::init() fills (src/spectrum/mod.rs:386-522).
struct SpectrumTables {
  float cie[3][RT_NBINS];        // X, Y, Z matching curves
  float smits[2][7][RT_NBINS];   // [reflectance|illuminant][white,cyan,magenta,yellow,red,green,blue]
  float xyz_scale;               // (700-400) / (CIE_Y_INTEGRAL * 60), src/spectrum/mod.rs:601-602
};

const SpectrumTables& spectrum_tables();
void spectrum_from_rgb(float r, float g, float b, int type, float out[RT_NBINS]);
void spectrum_to_xyz(const float spec[RT_NBINS], float xyz[3]);
void xyz_to_rgb(const float xyz[3], float rgb[3]);

}  // namespace rtmi
