// Pieces of the split-operand tile kernels shared by conv_b3.hip (three bf16 planes) and conv_h2.hip (two f16 planes): the
// tile constants, the epilogue (bias / addend / sign-bit gate / BatchNorm partial sums / scatter) and the counted waits of
// the LDS-DMA weight ring.  Device code only; everything is a template or force-inlined.
#pragma once
#include "lad_common.h"
#include "lad_device.h"

namespace lad {
namespace b3t {

constexpr int TAPS = 9;
constexpr int TM = 128;        // output rows per epilogue sub-tile
constexpr int THREADS = 256;   // 4 wavefronts x 32 rows

// Epilogue = the training variant (EPI_PLAIN) of s1_epilogue in conv_mfma.hip for 64 output channels: each wave
// transposes its 32 x 64 tile through LDS, then whole rows: + bias (+ addend), border rows times 0, 16-byte stores,
// per-128-row (sum, sum of squares) partials for the BatchNorm statistics.
// abits (with addend): sign bits of the activation whose ReLU gates the addend (one uint64 per row, lad_bn_math.h); the
// addend then is addend * [bit] -- the identity-shortcut gradient dy * [y > 0] of a residual block, taken from dy itself.
// `out` may be the addend's own buffer: a thread reads the 16 bytes it later writes, nobody else touches them.
//
// STAT (data-gradient launches): the partials become the first pass of the BatchNorm backward that consumes `out` --
// (sum d, sum d * xhat) per 128-row tile with d = out * [that BatchNorm's ReLU passed], xhat from its input x and saved
// statistics -- so lad_bn_bwd(pre_partials) skips its own pass over two tensors (what EPI_BNSTAT is to conv_mfma.hip).
// The ReLU decision comes from sign bits (B3Stat::bits, a residual block's output) or is recomputed from x (bits = NULL).
struct B3Stat {
    const float *x;                   // input of the consuming BatchNorm, geometry of `out`
    const unsigned long long *bits;   // sign bits of its (residual) output, or nullptr: mask = (x * scale + shift > 0)
    const float *coef;                // float[6][64]: scale, shift, mean, invstd, mean_lo, invstd_lo
};

// SCATTER (the stride-2 data gradient): row j of the 128-row sub-tile goes to tensor row base_row + rowoff[j] instead of q0 + j
// (rowoff[j] < 0: the position has no row -- nothing is stored, nothing is summed); the mask is not used.
struct B3Scatter {
    const int *rowoff;      // LDS, [128]: tensor row of every tile row relative to base_row, or -1
    int64_t base_row;       // first tensor row this tile can touch
    int64_t span_rows;      // rows of the tensor from base_row on that it may address
    int64_t part_tile;      // index of this sub-tile's partials
};

template <int C, bool STAT, bool SCATTER = false, bool ASMBAR = false, class StoreAcc, class MaskT>
__device__ __forceinline__ void b3_epilogue(StoreAcc store_acc, const float *__restrict__ bias, const float *addend,
                                            const unsigned long long *__restrict__ abits, float *out, float *__restrict__ partials,
                                            const MaskT *mask_tile, float *out_s, int64_t q0, int64_t rows, const B3Stat &bst,
                                            const B3Scatter &sct = B3Scatter{nullptr, 0, 0, 0}, int tid_in = -1) {
    constexpr int LDO = C + 4, LPR = C / 4, RPI = 64 / LPR, ITER = 32 / RPI, STEP = RPI * C * 4;
    // (tid_in: the caller's own copy of threadIdx.x -- a persistent kernel passes one the compiler cannot see through, so that what is
    // derived from it here is recomputed per tile instead of being kept, and spilled, across the caller's MFMA loop)
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *my = out_s + wave * 32 * LDO;
    store_acc(my);   // the wave's 32 x C tile, row-major with leading dimension LDO
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int64_t row0 = SCATTER ? sct.base_row : q0;
    const int64_t tile_rows = SCATTER ? sct.span_rows : rows - q0;
    const int64_t tile_bytes = tile_rows * (C * 4);
    const int voff = ((wave * 32 + rsub) * C + c4 * 4) * 4;
    const int woff = (wave * 32 + rsub) * 8;   // sign-bit words: 8 bytes per row
    int ro[SCATTER ? ITER : 1];
    if (SCATTER) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) ro[it] = sct.rowoff[wave * 32 + it * RPI + rsub];
    }
    auto off16 = [&](int it) { return SCATTER ? (ro[SCATTER ? it : 0] < 0 ? -1 : ro[SCATTER ? it : 0] * (C * 4) + c4 * 16) : voff + it * STEP; };
    auto off8 = [&](int it) { return SCATTER ? (ro[SCATTER ? it : 0] < 0 ? -1 : ro[SCATTER ? it : 0] * 8) : woff + it * RPI * 8; };
    const __amdgpu_buffer_rsrc_t out_r = make_rsrc(out + row0 * C, tile_bytes);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = *reinterpret_cast<const f32x4 *>(bias + c4 * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    f32x4 fsc = s1, fsh = s1, mu = s1, is = s1, mul = s1, isl = s1;
    if (STAT) {
        fsc = *reinterpret_cast<const f32x4 *>(bst.coef + 0 * C + c4 * 4);
        fsh = *reinterpret_cast<const f32x4 *>(bst.coef + 1 * C + c4 * 4);
        mu = *reinterpret_cast<const f32x4 *>(bst.coef + 2 * C + c4 * 4);
        is = *reinterpret_cast<const f32x4 *>(bst.coef + 3 * C + c4 * 4);
        mul = *reinterpret_cast<const f32x4 *>(bst.coef + 4 * C + c4 * 4);
        isl = *reinterpret_cast<const f32x4 *>(bst.coef + 5 * C + c4 * 4);
    }
    const bool from_bits = STAT && bst.bits != nullptr;
    auto gate = [&](u32x4 v, u32x2 w) {   // v * [sign bit of its channel] (lad_bn_math.h: mask_from_bits)
        const unsigned lo = w.x >> c4, hi = w.y >> c4;
        v.x = (lo & 1u) ? v.x : 0u;
        v.y = (lo & 0x10000u) ? v.y : 0u;
        v.z = (hi & 1u) ? v.z : 0u;
        v.w = (hi & 0x10000u) ? v.w : 0u;
        return v;
    };
    // The rows go in chunks of NI wave-instructions: every tensor the chunk needs is requested first, then consumed.  The
    // STAT variant reads two tensors more and takes two chunks so that it stays within the 168 registers of three
    // workgroups per CU.
    constexpr int NI = STAT ? ITER / 2 : ITER;
#pragma unroll
    for (int i0 = 0; i0 < ITER; i0 += NI) {
        u32x4 adv[NI], bx[STAT ? NI : 1];
        u32x2 wv[NI], bw[STAT ? NI : 1];
        if (addend != nullptr) {
            const __amdgpu_buffer_rsrc_t add_r = make_rsrc(addend + row0 * C, tile_bytes);
#pragma unroll
            for (int u = 0; u < NI; ++u) adv[u] = buf_load16(add_r, off16(i0 + u));
            if (abits != nullptr) {
                const __amdgpu_buffer_rsrc_t bits_r = make_rsrc(abits + row0, tile_rows * 8);
#pragma unroll
                for (int u = 0; u < NI; ++u) wv[u] = buf_load8(bits_r, off8(i0 + u));
            }
        }
        if (STAT) {
            const __amdgpu_buffer_rsrc_t x_r = make_rsrc(bst.x + row0 * C, tile_bytes);
#pragma unroll
            for (int u = 0; u < NI; ++u) bx[u] = buf_load16(x_r, off16(i0 + u));
            if (from_bits) {
                const __amdgpu_buffer_rsrc_t w_r = make_rsrc(bst.bits + row0, tile_rows * 8);
#pragma unroll
                for (int u = 0; u < NI; ++u) bw[u] = buf_load8(w_r, off8(i0 + u));
            }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int it = i0 + u;
            const int row = it * RPI + rsub;
            const float keep = SCATTER ? (ro[SCATTER ? it : 0] < 0 ? 0.f : 1.f) : (float)mask_tile[wave * 32 + row];   // 1 on interior rows, 0 on border rows
            f32x4 t = *reinterpret_cast<const f32x4 *>(my + row * LDO + c4 * 4);
            t += bv;
            if (addend != nullptr) t += __builtin_bit_cast(f32x4, abits != nullptr ? gate(adv[u], wv[u]) : adv[u]);
            t *= keep;
            buf_store16(__builtin_bit_cast(u32x4, t), out_r, off16(it));
            if (!STAT) {   // (sum, sum of squares) of the output: the train-mode BatchNorm that follows a forward convolution
                s1 += t;
                s2 = __builtin_elementwise_fma(t, t, s2);
            } else {       // the arithmetic of bn_bwd_reduce_kernel (bn.hip)
                const f32x4 xv = __builtin_bit_cast(f32x4, bx[u]);
                f32x4 d;
                if (from_bits) {
                    d = __builtin_bit_cast(f32x4, gate(__builtin_bit_cast(u32x4, t), bw[u]));
                } else {
                    const f32x4 yv = __builtin_elementwise_fma(xv, fsc, fsh);   // the fmaf the forward pass evaluated (mask_from_x)
                    d.x = yv.x > 0.f ? t.x : 0.f;
                    d.y = yv.y > 0.f ? t.y : 0.f;
                    d.z = yv.z > 0.f ? t.z : 0.f;
                    d.w = yv.w > 0.f ? t.w : 0.f;
                }
                const f32x4 tx = (xv - mu) - mul;
                const f32x4 xh = __builtin_elementwise_fma(tx, is, tx * isl);   // xhat1 (lad_bn_math.h)
                s1 += d;
                s2 = __builtin_elementwise_fma(d, xh, s2);
            }
        }
    }
    if (partials == nullptr) return;
    *reinterpret_cast<f32x4 *>(my + (rsub * 2 + 0) * C + c4 * 4) = s1;
    *reinterpret_cast<f32x4 *>(my + (rsub * 2 + 1) * C + c4 * 4) = s2;
    // (ASMBAR: called by the first four waves of a larger workgroup whose other waves execute a matching bare barrier)
    if (ASMBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
    if (tid < 2 * C) {
        const int k = tid / C, co = tid - k * C;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int rs = 0; rs < RPI; ++rs) s += out_s[w * 32 * LDO + (rs * 2 + k) * C + co];
        partials[((SCATTER ? sct.part_tile : q0 / TM) * 2 + k) * C + co] = s;
    }
}

// LDS-DMA wave-instructions this wave issues per tap (issue_tap below): wave-uniform
template <int TAP_BYTES>
__device__ __forceinline__ int dma_per_tap(int wave) {
    int n = 0;
#pragma unroll
    for (int r = 0; r * THREADS * 16 < TAP_BYTES; ++r) n += ((r * THREADS + wave * 64) * 16 < TAP_BYTES) ? 1 : 0;
    return n;
}

// wait until all but this wave's `PER_TAP_YOUNGER x (its DMAs per tap) + EXTRA` youngest vector-memory operations are done
template <int TAPS_YOUNGER, int EXTRA>
__device__ __forceinline__ void wait_dma(int nw_tap) {
    static_assert(2 * TAPS_YOUNGER + EXTRA <= 63, "vmcnt is a 6-bit field");
    if (nw_tap == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TAPS_YOUNGER + EXTRA) : "memory");
    else if (nw_tap == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TAPS_YOUNGER + EXTRA) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXTRA) : "memory");
}

}  // namespace b3t
}  // namespace lad
