// Device-side helpers shared by the ResNet kernels (gfx950 only).
//
// Activation layout ("PNHWC"): channels innermost, a *row* is one spatial position (C contiguous floats), and every
// image row / image is preceded by ONE border position / border row that it SHARES with its predecessor:
//     float A[batch][H+1][W+1][C]  followed by a tail of (W+1)+1 border rows,
// padded coordinates (yp, xp) = (y+1, x+1), flat row q = (b*(H+1) + yp)*(W+1) + xp.  The right neighbour of the last
// column is the next image row's border position, the row below the last image row is the next image's border row
// (or the tail).  A 3x3 stride-1 convolution is then a sum of 9 row-shifted GEMMs,
// out[q] = sum_tap in[q + (ky-1)*(W+1) + (kx-1)] * W_tap, with no per-tap bounds logic, at (H+1)(W+1)/(HW) = 1.033x the
// rows of the unpadded tensor at 100x44 (a private ring per image would cost 1.066x).
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
    int64_t rows;  // body + tail: every row a kernel may touch
    int64_t body;  // batch * Hp * Wp
    int Hp, Wp;    // H + 1, W + 1
    int img;       // Hp * Wp
};

__host__ __device__ inline Geom make_geom(int64_t batch, int H, int W) {
    Geom g;
    g.Hp = H + 1;
    g.Wp = W + 1;
    g.img = g.Hp * g.Wp;
    g.body = batch * g.img;
    g.rows = g.body + g.Wp + 1;
    return g;
}

__device__ __forceinline__ bool interior_row(int64_t q, const Geom &g) {
    if (q < 0 || q >= g.body) return false;
    const int rr = (int)(q % g.img);
    const int yp = rr / g.Wp;
    const int xp = rr - yp * g.Wp;
    return (yp >= 1) & (xp >= 1);
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

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// D(32x32) += A(32x16) * B(16x32), f16 inputs, f32 accumulate (v_mfma_f32_32x32x16_f16).  Lane l (r = l&31, h = l>>5)
// supplies A[r][8h + j] and B[8h + j][r], j = 0..7; D layout as mfma32.
__device__ __forceinline__ f32x16 mfma32_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// One 16-byte LDS-DMA per lane (global_load_lds_dwordx4): LDS destination = M0 (wave-uniform byte address) + lane*16.
// Issued through inline asm on purpose: hipcc orders a builtin LDS-DMA against every later ds_read with
// s_waitcnt vmcnt(0), which serialises the weight stream behind the MFMAs it should overlap; an asm statement is
// invisible to that pass, so the wait is placed by hand (dma_wait_all) in front of the barrier that publishes
// the chunk.  M0 is saved/restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}

}  // namespace lad
