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
    // host-computed reciprocals for the division-free row decode of the MFMA kernels (interior_row32)
    uint32_t wp_magic;  // floor(2^32 / Wp) + 1: __umulhi(n, wp_magic) == n / Wp for n < 2^32 / Wp
    double inv_img;     // 1.0 / img
};

__host__ __device__ inline Geom make_geom(int64_t batch, int H, int W) {
    Geom g;
    g.Hp = H + 1;
    g.Wp = W + 1;
    g.img = g.Hp * g.Wp;
    g.body = batch * g.img;
    g.rows = g.body + g.Wp + 1;
    g.wp_magic = (uint32_t)((1ull << 32) / (uint64_t)g.Wp) + 1u;
    g.inv_img = 1.0 / (double)g.img;
    return g;
}

__device__ __forceinline__ bool interior_row(int64_t q, const Geom &g) {
    if (q < 0 || q >= g.body) return false;
    const int rr = (int)(q % g.img);
    const int yp = rr / g.Wp;
    const int xp = rr - yp * g.Wp;
    return (yp >= 1) & (xp >= 1);
}

// Same predicate for rows below 2^31 of images below 2^20 positions (launchers check both), without integer division:
// the quotient by img comes from one f64 multiply (at most one too small, only when q is an exact multiple: fixed up),
// the quotient by Wp from a multiply-high with the host-computed reciprocal.  ~14 VALU instructions instead of the
// ~300 of two 64-bit divisions -- these run once per workgroup in the prologue of the MFMA kernels, where a wave
// competes for issue slots with the MFMA waves of its neighbours.
__device__ __forceinline__ bool interior_row32(uint32_t q, const Geom &g) {
    const uint32_t img = (uint32_t)g.img;
    uint32_t rr = q - (uint32_t)((double)q * g.inv_img) * img;
    rr = rr >= img ? rr - img : rr;
    const uint32_t yp = __umulhi(rr, g.wp_magic);
    const uint32_t xp = rr - yp * (uint32_t)g.Wp;
    return ((int64_t)q < g.body) & (yp >= 1u) & (xp >= 1u);
}

// Raw buffer resource over [base, base + bytes): loads outside it return 0, stores outside it are dropped, so the tile
// kernels need no per-row bounds logic at the two ends of a tensor.  Offsets are unsigned 32-bit: a "negative" row
// (before the tensor) wraps to a huge offset and is out of range like one past the end.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int64_t bytes) {
    const uint32_t n = bytes > 0x7fffffffll ? 0x7fffffffu : (bytes < 0 ? 0u : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(u32x4 v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0);
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 buf_load8(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store8(u32x2 v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, 0);
}
__device__ __forceinline__ float4 as_f4(u32x4 v) { return __builtin_bit_cast(float4, v); }
__device__ __forceinline__ u32x4 as_u4(float4 v) { return __builtin_bit_cast(u32x4, v); }

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


// Largest value of a wavefront, the same in every lane, without the LDS: four DPP steps inside each row of 16 lanes (xor 1,
// xor 2, mirror of the half row, mirror of the row), then the four rows' values through scalar registers.
__device__ __forceinline__ float wave_max64_dpp(float v) {
    int x = __builtin_bit_cast(int, v);
    x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true))));
    x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true))));
    x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, true))));
    x = __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, true))));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
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
// The same with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset: one VGPR of address instead of two.
__device__ __forceinline__ void dma16s(const void *uniform_base, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    const unsigned long long b = (unsigned long long)(size_t)uniform_base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(sb), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}

}  // namespace lad
