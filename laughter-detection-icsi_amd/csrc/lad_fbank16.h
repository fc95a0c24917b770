// Interface between the fbank plan (fbank.hip) and the 16-lanes-per-frame fast kernel (fbank16.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lad_hip.h"

namespace lad_fb16 {

struct Fast;  // device tables of the fast kernel; nullptr when the configuration is not eligible

// *out = nullptr (and LAD_OK) when the configuration has to take the general kernel
int build(const lad_fbank_cfg &cfg, const float *window, const float *melbank, Fast **out);
void destroy(Fast *f);
bool eligible(const Fast *f, int64_t n_clips, int64_t samples_per_clip, const float *pcm);
int launch(const Fast *f, const lad_fbank_cfg &cfg, int left_off, const float *pcm, int64_t n_clips, int64_t samples_per_clip,
           int64_t T, float *out, hipStream_t stream);

}  // namespace lad_fb16
