// Device-side helpers shared by the ResNet kernels (gfx950 only).
//
// Activation layout ("PNHWC"): float A[batch][H+2][W+2][C], channels innermost, one ring of border
// positions around every image.  A *row* is one spatial position (C contiguous floats); rows are numbered
// flat: q = (b*(H+2) + yp)*(W+2) + xp.  A 3x3 stride-1 convolution is then a sum of 9 row-shifted GEMMs,
// out[q] = sum_tap in[q + (ky-1)*(W+2) + (kx-1)] * W_tap, with no per-tap bounds logic.
// INVARIANT: border rows hold 0.0f in HBM in every activation and gradient tensor.  Every kernel that writes such a
// tensor writes zeros there (conv epilogues via the row mask, element-wise passes via their row geometry), so the
// MFMA kernels stage operands with plain 16-byte loads and only guard the two ends of the tensor.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lad {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Geom {
    int64_t rows;  // batch * Hp * Wp
    int Hp, Wp;    // padded height / width
    int img;       // Hp * Wp
};

__device__ __forceinline__ bool interior_row(int64_t q, const Geom &g) {
    if (q < 0 || q >= g.rows) return false;
    const int rr = (int)(q % g.img);
    const int yp = rr / g.Wp;
    const int xp = rr - yp * g.Wp;
    return (yp >= 1) & (yp <= g.Hp - 2) & (xp >= 1) & (xp <= g.Wp - 2);
}

// D(32x32) += A(32x2) * B(2x32), exact f32.  Lane l supplies A[l&31][l>>5] and B[l>>5][l&31];
// D register r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_sum64d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace lad
