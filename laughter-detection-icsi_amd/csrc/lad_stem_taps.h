// Tap table of the stem convolutions (1 input channel, 3x3): shared by stem.hip (f32) and conv_f16.hip (half output).
#pragma once
#include "lad_device.h"

namespace lad {

// The 9 taps of a row are the same for the 16 channel-quad threads that share it: ONE thread per row decodes the row
// and gathers its taps into LDS (tap_s[row][0..8], tap_s[row][9] = 1.0 for an interior row, 0.0 otherwise), the
// others read them back as broadcasts -- 16x fewer address decodes and global loads than every thread for itself.
constexpr int STEM_TM = 128;  // rows per stem tile
constexpr int TAPW = 12;  // floats per row of the table (9 taps + flag, padded to 48 bytes)
__device__ __forceinline__ void fill_taps(const float *__restrict__ feat, const Geom &g, int H, int W, int64_t q0,
                                          int64_t frame_stride, int64_t frames_avail, float *tap_s /*[STEM_TM][TAPW]*/) {
    const int r = threadIdx.x;
    if (r < STEM_TM) {
        const int64_t q = q0 + r;
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = 0.0f;
        float flag = 0.0f;
        if (q < g.body) {
            const int64_t b = q / g.img;
            const int rr = (int)(q - b * g.img);
            const int yp = rr / g.Wp, xp = rr - yp * g.Wp;
            if (yp >= 1 && xp >= 1) {
                flag = 1.0f;
                const int y = yp - 1, x = xp - 1;
                const int64_t f0 = b * frame_stride;  // first frame of image b (frame_stride = H: back-to-back images)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W && (f0 + yy) < frames_avail;
                        v[ky * 3 + kx] = ok ? feat[(f0 + yy) * W + xx] : 0.0f;
                    }
            }
        }
        float4 *dst = reinterpret_cast<float4 *>(tap_s + r * TAPW);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        dst[2] = make_float4(v[8], flag, 0.f, 0.f);
    }
}
__device__ __forceinline__ bool read_taps(const float *tap_s, int r, float (&v)[9]) {
    const float4 *src = reinterpret_cast<const float4 *>(tap_s + r * TAPW);
    const float4 a = src[0], b = src[1], c = src[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x;
    return c.y != 0.0f;
}


}  // namespace lad
