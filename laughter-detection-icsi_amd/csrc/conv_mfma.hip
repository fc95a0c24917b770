// Implicit-GEMM convolutions on the f32 matrix cores (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the nn.Conv2d calls of the reference's ResidualBlock / ResNetBigger (models.py:86-106,
// 110-115, 186-189) in forward and in the data-gradient direction.  Layout and the "row-shifted GEMM"
// formulation: lad_device.h.  GEMM view of one launch: M = rows (spatial positions of the whole batch),
// N = output channels, K = taps * input channels.
//
// conv_s1_kernel  stride 1 (3x3 pad 1, or 1x1): a workgroup owns 128 consecutive output rows and all
//                 output channels; the input rows it needs (128 + a halo of W+3 rows either side) are one
//                 contiguous span of HBM, staged into LDS 32 channels at a time with 16-byte coalesced loads
//                 and border rows zeroed; every wavefront runs 32 rows x COUT through 9 * CIN/2 MFMAs.  A
//                 operands are ds_read_b128 (4 consecutive input channels feed 4 MFMAs); B operands come from
//                 a weight image pre-packed as [tap][CIN/4][COUT][4] (a lane's 16 bytes are exactly its four
//                 K-slices), streamed chunk by chunk (one tap x 32 channels, <= 8 KB) through a two-slot LDS
//                 ring by LDS-DMA one chunk ahead of the MFMAs.  51 KB of LDS -> three workgroups per CU.
// conv_s2_kernel  stride 2 (3x3 pad 1, or 1x1): same MFMA core, A operands gathered per lane from HBM/L2
//                 (these layers are 3 % of the model's FLOPs).
// Epilogue (both): + bias, optional residual addend, border rows forced to zero, per-channel sum / sum of
//                 squares of the tile written as a partial for the train-mode BatchNorm statistics.
#include "lad_common.h"
#include "lad_device.h"

namespace {
using namespace lad;

constexpr int TM = 128;      // output rows per workgroup
constexpr int THREADS = 256; // 4 wavefronts x 32 rows

template <int COUT>
struct NTiles {
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int COUTP = NT * 32;
};

// Shared epilogue.  acc[n][r] holds out[row = acc_row(r)][co = n*32 + (lane&31)] of this wave's 32 rows: a lane owns
// one column of 16 scattered rows, the wrong shape for memory.  Each wave therefore transposes its 32 x COUT tile
// through LDS (out_s, private to the wave: LDS operations of one wave execute in order) and then works on whole
// rows: + bias, + residual addend, border rows forced to zero, 16-byte stores that cover 1 KB of consecutive HBM
// per wave instruction, and the per-channel (sum, sum of squares) of the tile for the train-mode BatchNorm.
// Optional fusion for data-gradient launches: the tensor this launch produces is the `dy` of a BatchNorm backward, whose
// first pass (sum dz, sum dz*xhat per channel; dz = dy * ReLU mask) only needs dy row by row -- exactly what the epilogue
// holds.  With `x` set, the per-tile partials become those two sums (instead of sum, sum of squares of the output),
// computed against that BatchNorm's input x (and its output y for the mask when it has a residual branch; otherwise
// the mask is recomputed from x), and lad_bn_bwd skips its reduce pass over the tensors.
struct BnStat {
    const float *x;     // BatchNorm input (pre-normalisation), same geometry and channels as `out`; nullptr = off
    const float *y;     // BatchNorm(+residual)+ReLU output, or nullptr: mask = (x*scale + shift > 0)
    const float *coef;  // float[6][C]: scale, shift, mean, invstd, mean_lo, invstd_lo
};

template <int COUT>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[NTiles<COUT>::NT], const float *__restrict__ bias,
                                              const float *__restrict__ addend, float *__restrict__ out,
                                              float *__restrict__ partials, const float *mask_tile /*[TM]*/,
                                              float *out_s /*[TM][COUT+4]*/, float *red_s /*[4][2][COUT]*/, int64_t q0,
                                              int64_t rows, const float *__restrict__ scale = nullptr, int relu = 0,
                                              BnStat bst = BnStat{nullptr, nullptr, nullptr}) {
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int LDO = COUT + 4;
    constexpr int LPR = COUT / 4;   // lanes per output row
    constexpr int RPI = 64 / LPR;   // rows per wave instruction
    constexpr int ITER = 32 / RPI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31;
    float *my = out_s + wave * 32 * LDO;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        if (co < COUT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, lane) * LDO + co] = acc[n][r];
        }
    }
    const int c4 = lane % LPR, rsub = lane / LPR;
    // eval mode folds the BatchNorm into the convolution: out = relu(acc * scale + bias' + addend)
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (bias != nullptr) bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
    if (scale != nullptr) sv = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float4 v[ITER], ad[ITER];
    bool ok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int grow = wave * 32 + it * RPI + rsub;
        const int64_t q = q0 + grow;
        ok[it] = q < rows;
        ad[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (addend != nullptr && ok[it]) ad[it] = *reinterpret_cast<const float4 *>(addend + q * COUT + c4 * 4);
    }
    float4 bx[ITER], by[ITER];
    if (bst.x != nullptr) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int64_t q = q0 + wave * 32 + it * RPI + rsub;
            bx[it] = by[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[it]) {
                bx[it] = *reinterpret_cast<const float4 *>(bst.x + q * COUT + c4 * 4);
                if (bst.y != nullptr) by[it] = *reinterpret_cast<const float4 *>(bst.y + q * COUT + c4 * 4);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = it * RPI + rsub;
        const float keep = mask_tile[wave * 32 + row];
        float4 t = *reinterpret_cast<const float4 *>(my + row * LDO + c4 * 4);
        if (scale != nullptr) {
            t.x = fmaf(t.x, sv.x, bv.x) + ad[it].x; t.y = fmaf(t.y, sv.y, bv.y) + ad[it].y;
            t.z = fmaf(t.z, sv.z, bv.z) + ad[it].z; t.w = fmaf(t.w, sv.w, bv.w) + ad[it].w;
        } else {
            t.x += bv.x + ad[it].x; t.y += bv.y + ad[it].y; t.z += bv.z + ad[it].z; t.w += bv.w + ad[it].w;
        }
        if (relu) {
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
        }
        if (keep == 0.0f) t = make_float4(0.f, 0.f, 0.f, 0.f);
        v[it] = t;
    }
    if (bst.x == nullptr) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            if (ok[it]) {
                const int64_t q = q0 + wave * 32 + it * RPI + rsub;
                *reinterpret_cast<float4 *>(out + q * COUT + c4 * 4) = v[it];
                s1.x += v[it].x; s1.y += v[it].y; s1.z += v[it].z; s1.w += v[it].w;
                s2.x = fmaf(v[it].x, v[it].x, s2.x); s2.y = fmaf(v[it].y, v[it].y, s2.y);
                s2.z = fmaf(v[it].z, v[it].z, s2.z); s2.w = fmaf(v[it].w, v[it].w, s2.w);
            }
        }
    } else {
        // BatchNorm-backward sums of the consumer of this gradient (same arithmetic as bn_bwd_reduce_kernel, bn.hip)
        const float4 fsc = *reinterpret_cast<const float4 *>(bst.coef + 0 * COUT + c4 * 4);
        const float4 fsh = *reinterpret_cast<const float4 *>(bst.coef + 1 * COUT + c4 * 4);
        const float4 mu = *reinterpret_cast<const float4 *>(bst.coef + 2 * COUT + c4 * 4);
        const float4 is = *reinterpret_cast<const float4 *>(bst.coef + 3 * COUT + c4 * 4);
        const float4 mul = *reinterpret_cast<const float4 *>(bst.coef + 4 * COUT + c4 * 4);
        const float4 isl = *reinterpret_cast<const float4 *>(bst.coef + 5 * COUT + c4 * 4);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            if (ok[it]) {
                const int64_t q = q0 + wave * 32 + it * RPI + rsub;
                *reinterpret_cast<float4 *>(out + q * COUT + c4 * 4) = v[it];
                float4 d = v[it];
                const float4 xv = bx[it];
                if (bst.y != nullptr) {
                    d.x = by[it].x > 0.f ? d.x : 0.f; d.y = by[it].y > 0.f ? d.y : 0.f;
                    d.z = by[it].z > 0.f ? d.z : 0.f; d.w = by[it].w > 0.f ? d.w : 0.f;
                } else {
                    d.x = fmaf(xv.x, fsc.x, fsh.x) > 0.f ? d.x : 0.f; d.y = fmaf(xv.y, fsc.y, fsh.y) > 0.f ? d.y : 0.f;
                    d.z = fmaf(xv.z, fsc.z, fsh.z) > 0.f ? d.z : 0.f; d.w = fmaf(xv.w, fsc.w, fsh.w) > 0.f ? d.w : 0.f;
                }
                float tx = (xv.x - mu.x) - mul.x, ty = (xv.y - mu.y) - mul.y, tz = (xv.z - mu.z) - mul.z, tw = (xv.w - mu.w) - mul.w;
                tx = fmaf(tx, is.x, tx * isl.x); ty = fmaf(ty, is.y, ty * isl.y);
                tz = fmaf(tz, is.z, tz * isl.z); tw = fmaf(tw, is.w, tw * isl.w);
                s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
                s2.x = fmaf(d.x, tx, s2.x); s2.y = fmaf(d.y, ty, s2.y); s2.z = fmaf(d.z, tz, s2.z); s2.w = fmaf(d.w, tw, s2.w);
            }
        }
    }
    if (partials != nullptr) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            s1.x += __shfl_xor(s1.x, off, 64); s1.y += __shfl_xor(s1.y, off, 64);
            s1.z += __shfl_xor(s1.z, off, 64); s1.w += __shfl_xor(s1.w, off, 64);
            s2.x += __shfl_xor(s2.x, off, 64); s2.y += __shfl_xor(s2.y, off, 64);
            s2.z += __shfl_xor(s2.z, off, 64); s2.w += __shfl_xor(s2.w, off, 64);
        }
        if (lane < LPR) {
            *reinterpret_cast<float4 *>(red_s + (wave * 2 + 0) * COUT + c4 * 4) = s1;
            *reinterpret_cast<float4 *>(red_s + (wave * 2 + 1) * COUT + c4 * 4) = s2;
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int k = tid / COUT, co = tid - k * COUT;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red_s[(w * 2 + k) * COUT + co];
            partials[((q0 / TM) * 2 + k) * COUT + co] = s;  // (indexed by tile, not by blockIdx: callers may reorder tiles)
        }
    }
}

// Weight chunks of the stride-1 kernel: the packed image wt[tap][CIN/4][COUTP][4] is consumed in chunks of KC input
// channels of one tap (contiguous in the image), streamed global -> LDS by LDS-DMA into a two-deep ring.
template <int CIN, int COUT, int TAPS>
struct S1Cfg {
    static constexpr int COUTP = NTiles<COUT>::COUTP;
    static constexpr int KC = CIN >= 32 ? 32 : CIN;          // input channels per chunk
    static constexpr int CPT = CIN / KC;                     // chunks per tap
    static constexpr int NCHUNK = TAPS * CPT;
    static constexpr int CHUNK_FLOATS = KC * COUTP;          // 8 KB (64-wide) ... 2 KB (16 -> 32-wide)
    static constexpr int ROUNDS = (CHUNK_FLOATS * 4 + THREADS * 16 - 1) / (THREADS * 16);
};

template <int CIN, int COUT, int TAPS>
__device__ __forceinline__ void issue_chunk(const float *__restrict__ wt, float *b_buf, int chunk, int tid, int wave) {
    using C = S1Cfg<CIN, COUT, TAPS>;
    const float *src = wt + (int64_t)chunk * C::CHUNK_FLOATS;
#pragma unroll
    for (int r = 0; r < C::ROUNDS; ++r) {
        const int f = (r * THREADS + tid) * 4;  // first float this lane moves
        if ((r * THREADS + wave * 64) * 4 < C::CHUNK_FLOATS)  // wave-uniform
            dma16(src + f, lds_addr(b_buf + (r * THREADS + wave * 64) * 4));
    }
}

#ifdef LAD_STAMP
// diagnostic build only (tools/stamp_conv.py): shader-clock stamps of wave 0 per workgroup, never part of the product
__device__ unsigned long long lad_dbg[8 * 32768];
#define LAD_STAMP_AT(k)                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 32768) lad_dbg[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LAD_STAMP_AT(k)
#endif

constexpr int S1_PRE = 8;  // 16-byte registers per thread that carry one input stage (bounds the tile width)
template <int RB> struct S1Pre { static constexpr int N = RB == 1 ? S1_PRE : 11; };  // 256 + 2*46 rows x 8 / 256 threads

// Epilogue of the stride-1 kernel.  Same arithmetic as conv_epilogue above, organised for the place it runs in: a
// wave leaving its MFMA loop shares its SIMD with two waves (of the neighbouring workgroups) that are still in
// theirs, and every VALU / SALU instruction and every branch it issues waits its turn behind their MFMAs -- in-kernel
// stamps (tools/stamp_conv.py) showed the old, general epilogue and prologue (1,500 + 1,200 instructions, ~200
// branches from per-row bounds tests and run-time options) taking 40% of a workgroup's lifetime while the memory
// they wait on answers in 3,000 cycles.  Here the variant (EPI) is a template parameter, tensors are reached through
// buffer resources whose range check replaces every per-row `q < rows` test, and what is left is straight-line code.
enum { EPI_PLAIN = 0, EPI_EVAL = 1, EPI_BNSTAT = 2 };

template <int COUT, int EPI>
__device__ __forceinline__ void s1_epilogue(f32x16 (&acc)[NTiles<COUT>::NT], const float *__restrict__ bias,
                                            const float *__restrict__ addend, float *__restrict__ out,
                                            float *__restrict__ partials, const float *mask_tile /*[TM]*/,
                                            float *out_s /*[TM][COUT+4]*/, float *red_s /*[4][2][COUT]*/, int64_t q0,
                                            int64_t rows, const float *__restrict__ scale, int relu, BnStat bst) {
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int LDO = COUT + 4;
    constexpr int LPR = COUT / 4;   // lanes per output row
    constexpr int RPI = 64 / LPR;   // rows per wave instruction
    constexpr int ITER = 32 / RPI;
    constexpr int STEP = RPI * COUT * 4;  // bytes between the rows of consecutive iterations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31;
    float *my = out_s + wave * 32 * LDO;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        if (co < COUT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, lane) * LDO + co] = acc[n][r];
        }
    }
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int64_t tile_bytes = (rows - q0) * (COUT * 4);  // to the end of the tensor (wave-uniform)
    const int voff = ((wave * 32 + rsub) * COUT + c4 * 4) * 4;
    const __amdgpu_buffer_rsrc_t out_r = make_rsrc(out + q0 * COUT, tile_bytes);
    if (EPI == EPI_PLAIN) {
        // The training path (forward convolutions and data gradients).  While this wave runs, its SIMD neighbour is in
        // its MFMA loop and a VALU instruction gets an issue slot about once per MFMA (64 cycles; LDS, memory and
        // scalar instructions issue beside the MFMAs): stamps put the two 305-VALU passes of the general form at 39 k
        // of a workgroup's 169 k cycles.  Hence: 4-wide vector arithmetic (packed f32 adds / multiplies / FMAs), the
        // border mask as a multiplication, and NO cross-lane shuffles -- the per-lane statistics go to LDS (this wave's
        // own, by then consumed, slice of the output tile) and are summed by the 2*COUT threads that write the partial.
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) bv = *reinterpret_cast<const f32x4 *>(bias + c4 * 4);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (addend != nullptr) {
            const __amdgpu_buffer_rsrc_t add_r = make_rsrc(addend + q0 * COUT, tile_bytes);
            u32x4 adv[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) adv[it] = buf_load16(add_r, voff + it * STEP);
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int row = it * RPI + rsub;
                const float keep = mask_tile[wave * 32 + row];  // 1.0f inside the image, 0.0f on border rows / past the tensor
                f32x4 t = *reinterpret_cast<const f32x4 *>(my + row * LDO + c4 * 4);
                t = (t + bv + __builtin_bit_cast(f32x4, adv[it])) * keep;
                buf_store16(__builtin_bit_cast(u32x4, t), out_r, voff + it * STEP);
                s1 += t;
                s2 = __builtin_elementwise_fma(t, t, s2);
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int row = it * RPI + rsub;
                const float keep = mask_tile[wave * 32 + row];
                f32x4 t = *reinterpret_cast<const f32x4 *>(my + row * LDO + c4 * 4);
                t = (t + bv) * keep;
                buf_store16(__builtin_bit_cast(u32x4, t), out_r, voff + it * STEP);
                s1 += t;
                s2 = __builtin_elementwise_fma(t, t, s2);
            }
        }
        if (partials == nullptr) return;
        // `my` (32 x LDO floats, private to this wave, fully read above -- LDS is in order within a wave) now takes the
        // wave's RPI x 2 x COUT partial sums
        *reinterpret_cast<f32x4 *>(my + (rsub * 2 + 0) * COUT + c4 * 4) = s1;
        *reinterpret_cast<f32x4 *>(my + (rsub * 2 + 1) * COUT + c4 * 4) = s2;
        __syncthreads();
        if (tid < 2 * COUT) {
            const int k = tid / COUT, co = tid - k * COUT;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int rs = 0; rs < RPI; ++rs) s += out_s[w * 32 * LDO + (rs * 2 + k) * COUT + co];
            partials[((q0 / TM) * 2 + k) * COUT + co] = s;  // one partial per 128-row sub-tile (lad_conv_num_tiles)
        }
        return;
    }
    float4 ad[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) ad[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (addend != nullptr) {
        const __amdgpu_buffer_rsrc_t add_r = make_rsrc(addend + q0 * COUT, tile_bytes);
#pragma unroll
        for (int it = 0; it < ITER; ++it) ad[it] = as_f4(buf_load16(add_r, voff + it * STEP));
    }
    float4 bx[ITER], by[ITER];
    if (EPI == EPI_BNSTAT) {
        const __amdgpu_buffer_rsrc_t x_r = make_rsrc(bst.x + q0 * COUT, tile_bytes);
#pragma unroll
        for (int it = 0; it < ITER; ++it) bx[it] = as_f4(buf_load16(x_r, voff + it * STEP));
        if (bst.y != nullptr) {
            const __amdgpu_buffer_rsrc_t y_r = make_rsrc(bst.y + q0 * COUT, tile_bytes);
#pragma unroll
            for (int it = 0; it < ITER; ++it) by[it] = as_f4(buf_load16(y_r, voff + it * STEP));
        }
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (bias != nullptr) bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
    if (EPI == EPI_EVAL) sv = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    float4 v[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = it * RPI + rsub;
        const bool keep = mask_tile[wave * 32 + row] != 0.0f;  // border rows and rows past the tensor: exact zero
        float4 t = *reinterpret_cast<const float4 *>(my + row * LDO + c4 * 4);
        if (EPI == EPI_EVAL) {  // BatchNorm folded into the convolution: relu(acc * scale + shift + addend)
            t.x = fmaf(t.x, sv.x, bv.x) + ad[it].x; t.y = fmaf(t.y, sv.y, bv.y) + ad[it].y;
            t.z = fmaf(t.z, sv.z, bv.z) + ad[it].z; t.w = fmaf(t.w, sv.w, bv.w) + ad[it].w;
            t.x = relu ? fmaxf(t.x, 0.f) : t.x; t.y = relu ? fmaxf(t.y, 0.f) : t.y;
            t.z = relu ? fmaxf(t.z, 0.f) : t.z; t.w = relu ? fmaxf(t.w, 0.f) : t.w;
        } else {
            t.x += bv.x + ad[it].x; t.y += bv.y + ad[it].y; t.z += bv.z + ad[it].z; t.w += bv.w + ad[it].w;
        }
        t.x = keep ? t.x : 0.f; t.y = keep ? t.y : 0.f; t.z = keep ? t.z : 0.f; t.w = keep ? t.w : 0.f;
        v[it] = t;
        buf_store16(as_u4(t), out_r, voff + it * STEP);
    }
    if (EPI == EPI_EVAL) return;
    if (partials == nullptr) return;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    {   // EPI_BNSTAT (EPI_PLAIN and EPI_EVAL returned above)
        // BatchNorm-backward sums of the consumer of this gradient (same arithmetic as bn_bwd_reduce_kernel, bn.hip)
        const float4 fsc = *reinterpret_cast<const float4 *>(bst.coef + 0 * COUT + c4 * 4);
        const float4 fsh = *reinterpret_cast<const float4 *>(bst.coef + 1 * COUT + c4 * 4);
        const float4 mu = *reinterpret_cast<const float4 *>(bst.coef + 2 * COUT + c4 * 4);
        const float4 is = *reinterpret_cast<const float4 *>(bst.coef + 3 * COUT + c4 * 4);
        const float4 mul = *reinterpret_cast<const float4 *>(bst.coef + 4 * COUT + c4 * 4);
        const float4 isl = *reinterpret_cast<const float4 *>(bst.coef + 5 * COUT + c4 * 4);
        const bool from_y = bst.y != nullptr;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            float4 d = v[it];
            const float4 xv = bx[it];
            const float4 yv = from_y ? by[it] : make_float4(fmaf(xv.x, fsc.x, fsh.x), fmaf(xv.y, fsc.y, fsh.y),
                                                            fmaf(xv.z, fsc.z, fsh.z), fmaf(xv.w, fsc.w, fsh.w));
            d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f;
            d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
            float tx = (xv.x - mu.x) - mul.x, ty = (xv.y - mu.y) - mul.y, tz = (xv.z - mu.z) - mul.z, tw = (xv.w - mu.w) - mul.w;
            tx = fmaf(tx, is.x, tx * isl.x); ty = fmaf(ty, is.y, ty * isl.y);
            tz = fmaf(tz, is.z, tz * isl.z); tw = fmaf(tw, is.w, tw * isl.w);
            s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
            s2.x = fmaf(d.x, tx, s2.x); s2.y = fmaf(d.y, ty, s2.y); s2.z = fmaf(d.z, tz, s2.z); s2.w = fmaf(d.w, tw, s2.w);
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        s1.x += __shfl_xor(s1.x, off, 64); s1.y += __shfl_xor(s1.y, off, 64);
        s1.z += __shfl_xor(s1.z, off, 64); s1.w += __shfl_xor(s1.w, off, 64);
        s2.x += __shfl_xor(s2.x, off, 64); s2.y += __shfl_xor(s2.y, off, 64);
        s2.z += __shfl_xor(s2.z, off, 64); s2.w += __shfl_xor(s2.w, off, 64);
    }
    if (lane < LPR) {
        *reinterpret_cast<float4 *>(red_s + (wave * 2 + 0) * COUT + c4 * 4) = s1;
        *reinterpret_cast<float4 *>(red_s + (wave * 2 + 1) * COUT + c4 * 4) = s2;
    }
    __syncthreads();
    if (tid < 2 * COUT) {
        const int k = tid / COUT, co = tid - k * COUT;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red_s[(w * 2 + k) * COUT + co];
        partials[((q0 / TM) * 2 + k) * COUT + co] = s;  // one partial per 128-row sub-tile (lad_conv_num_tiles)
    }
}

template <int CIN, int COUT, int TAPS, int EPI, int RB>
__global__ __launch_bounds__(THREADS, RB == 1 ? 3 : 2) void conv_s1_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ wt,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ addend,
                                                             float *__restrict__ out, float *__restrict__ partials,
                                                             Geom g, const float *__restrict__ scale, int relu, BnStat bst) {
    // K = TAPS * CIN is walked stage by stage: a stage is KC input channels (all taps).  Per stage the input rows of
    // the tile (+halo) sit in LDS, KC channels wide; per (stage, tap) one weight chunk comes through the DMA ring.
    // Keeping only KC = 32 channels of the 64 resident halves the tile (51 KB with the ring), so THREE workgroups
    // share a CU and one of them is (nearly) always in its MFMA phase while the others stage or store.
    // RB = row blocks per wavefront: with RB = 2 a workgroup owns 256 rows (wave w: rows w*32.. of each 128-row half),
    // every weight fragment read from LDS feeds two row blocks, a barrier interval holds 64 MFMAs per wave instead of
    // 32, and the halo is amortised over twice the rows; 69 KB of LDS -> two workgroups per CU.
    using C = S1Cfg<CIN, COUT, TAPS>;
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int COUTP = NTiles<COUT>::COUTP;
    constexpr int KC = C::KC;
    constexpr int NSTAGE = C::CPT;
    constexpr int LDA = KC + 4;  // padded LDS row: conflict-free ds_read_b128 across 32 rows
    constexpr int A4 = KC / 4;
    constexpr int RPU = THREADS / A4;        // input rows one register (one 16-byte load per thread) covers
    constexpr int USTEP = RPU * CIN * 4;     // bytes between the rows of consecutive registers
    constexpr int PRE = S1Pre<RB>::N;
    constexpr int TMW = TM * RB;             // rows per workgroup
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMW + 2 * halo;
    // LDS: [ weight ring | input rows ]  (re-used as the output tile by the epilogue)  | row mask | stat scratch
    // RB = 2 rounds the staged rows up to whole register batches: the LDS writes of a batch then need no per-row guard
    const int arows = RB == 1 ? nrows : ((nrows + PRE * RPU - 1) / (PRE * RPU)) * (PRE * RPU);
    const int main_floats = max(2 * C::CHUNK_FLOATS + arows * LDA, TM * (COUT + 4));
    float *b_s = smem;                             // [2][CHUNK_FLOATS]   (first: LDS-DMA wants 16-byte alignment)
    float *a_s = b_s + 2 * C::CHUNK_FLOATS;        // [nrows][LDA]
    float *mask_s = smem + main_floats;            // [TMW]
    float *red_s = mask_s + TMW;                   // [4][2][COUT]
    const int64_t q0 = (int64_t)blockIdx.x * TMW;

    LAD_STAMP_AT(0)
#ifdef LAD_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 32768)
        lad_dbg[blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((32 - 1) << 11 | (0 << 6) | 20) << 32) |
                                      __builtin_amdgcn_s_getreg((32 - 1) << 11 | (0 << 6) | 4);  // XCC_ID, HW_ID (tools/stamp_overlap.py)
#endif
    issue_chunk<CIN, COUT, TAPS>(wt, b_s, 0, tid, wave);
    // the row mask is only needed for the OUTPUT rows (the epilogue zeroes border positions); input border rows are
    // zero in HBM already (layout invariant), and the two ends of the tensor are the buffer resource's range check
    for (int j = tid; j < TMW; j += THREADS) mask_s[j] = interior_row32((uint32_t)q0 + (uint32_t)j, g) ? 1.0f : 0.0f;
    LAD_STAMP_AT(4)
    const int64_t start = q0 - halo;                 // first staged row; negative in the first tile(s)
    const int64_t first = start < 0 ? 0 : start;
    const int row_lo = (int)(first - start);
    // the resource ends with the tile's span (or the tensor): registers past the tile cost no memory traffic
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + first * CIN, min(g.rows - first, (int64_t)(nrows - row_lo)) * (CIN * 4));
    const int r0 = tid / A4, c4 = tid - r0 * A4;
    const int voff = ((r0 - row_lo) * CIN + c4 * 4) * 4;   // rows before the tensor: negative = out of range = 0.0f
    float *lds0 = a_s + r0 * LDA + c4 * 4;
    float *dummy = a_s + r0 * LDA + KC;              // this row's padding: sink for registers past the tile
    u32x4 pre[PRE];
    // first stage of input rows: all loads in flight together, then the LDS writes
    for (int base = 0; base < nrows; base += PRE * RPU) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = buf_load16(in_r, voff + (base / RPU + u) * USTEP);
        LAD_STAMP_AT(5)
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int row = base + u * RPU + r0;
            *reinterpret_cast<u32x4 *>((RB > 1 || row < nrows) ? lds0 + (base + u * RPU) * LDA : dummy) = pre[u];
        }
        LAD_STAMP_AT(6)
    }

    f32x16 acc[RB][NT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][n][r] = 0.0f;

    const int i = lane & 31, gk = lane >> 5;
    const float *a_base = a_s + (wave * 32 + i + halo) * LDA + 4 * gk;  // row block rb: + rb * TM rows
    const int b_off = (gk * COUTP + i) * 4;
    int seq = 0;
    LAD_STAMP_AT(1)
#pragma unroll 1
    for (int stage = 0; stage < NSTAGE; ++stage) {
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap, ++seq) {
            // this wave's DMA of the current chunk (issued one iteration ago) has landed; after the barrier every
            // wave's has, the staged input rows are visible, and everyone is done with the other ring slot
            dma_wait_all();
            __syncthreads();
            if (seq + 1 < C::NCHUNK) {
                const int nseq = seq + 1, ntap = nseq % TAPS, nstage = nseq / TAPS;
                issue_chunk<CIN, COUT, TAPS>(wt, b_s + (nseq & 1) * C::CHUNK_FLOATS, ntap * NSTAGE + nstage, tid, wave);
            }
            if (NSTAGE > 1 && tap == TAPS - 1 && stage + 1 < NSTAGE) {
                // next stage's input rows: global -> registers now, registers -> LDS after this tap's MFMAs
                const int voff2 = voff + (stage + 1) * KC * 4;
#pragma unroll
                for (int u = 0; u < PRE; ++u) pre[u] = buf_load16(in_r, voff2 + u * USTEP);
            }
            const int off = (TAPS == 9) ? ((tap / 3 - 1) * g.Wp + (tap % 3 - 1)) : 0;
            const float *ap = a_base + off * LDA;
            const float *bp = b_s + (seq & 1) * C::CHUNK_FLOATS + b_off;
            // operand fragments one 8-channel group ahead of the MFMAs that use them: the LDS latency of group c8+1
            // sits behind the 8 MFMAs of group c8 (sched_barrier fences keep the scheduler from sinking the reads)
            float4 af[2][RB], bf[2][NT];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) af[0][rb] = *reinterpret_cast<const float4 *>(ap + rb * TM * LDA);
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[0][n] = *reinterpret_cast<const float4 *>(bp + n * 32 * 4);
#pragma unroll
            for (int c8 = 0; c8 < KC / 8; ++c8) {
                const int cur = c8 & 1, nxt = cur ^ 1;
                __builtin_amdgcn_sched_barrier(0);
                if (c8 + 1 < KC / 8) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        af[nxt][rb] = *reinterpret_cast<const float4 *>(ap + rb * TM * LDA + (c8 + 1) * 8);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        bf[nxt][n] = *reinterpret_cast<const float4 *>(bp + ((c8 + 1) * 2 * COUTP + n * 32) * 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[rb][n] = mfma32(af[cur][rb].x, bf[cur][n].x, acc[rb][n]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[rb][n] = mfma32(af[cur][rb].y, bf[cur][n].y, acc[rb][n]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[rb][n] = mfma32(af[cur][rb].z, bf[cur][n].z, acc[rb][n]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[rb][n] = mfma32(af[cur][rb].w, bf[cur][n].w, acc[rb][n]);
            }
        }
        if (NSTAGE > 1 && stage + 1 < NSTAGE) {
            __syncthreads();  // every wave has finished reading this stage's rows
#pragma unroll
            for (int u = 0; u < PRE; ++u) {
                const int row = u * RPU + r0;
                *reinterpret_cast<u32x4 *>((RB > 1 || row < nrows) ? lds0 + u * RPU * LDA : dummy) = pre[u];
            }
        }
    }
    LAD_STAMP_AT(2)
    __syncthreads();  // every wave is out of the MFMA loop: the ring + input rows become the output tile
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int64_t qs = q0 + rb * TM;  // 128-row sub-tile: the unit of the epilogue and of the stat partials
        if (rb > 0) {
            if (qs >= g.rows) break;      // (workgroup-uniform) the tensor ended inside the first half
            __syncthreads();              // stat scratch of the previous half has been read
        }
        s1_epilogue<COUT, EPI>(acc[rb], bias, addend, out, partials, mask_s + rb * TM, smem, red_s, qs, g.rows, scale, relu, bst);
    }
    LAD_STAMP_AT(3)
}

// SC (3x3 only): the block's 1x1 stride-2 shortcut convolution (no bias) reads exactly the rows the 3x3's centre tap
// gathers, so its product rides on that tap's A fragments into a second accumulator and leaves through a second
// epilogue (out_sc, partials_sc) -- one launch and one pass over the input instead of two.
template <int CIN, int COUT, int TAPS, bool SC = false>
__global__ __launch_bounds__(THREADS, 2) void conv_s2_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ wt,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out, float *__restrict__ partials,
                                                             Geom gi, Geom go, const float *__restrict__ scale, int relu,
                                                             const float *__restrict__ wt_sc = nullptr,
                                                             float *__restrict__ out_sc = nullptr,
                                                             float *__restrict__ partials_sc = nullptr) {
    static_assert(!SC || TAPS == 9, "the shortcut rides with the 3x3 convolution");
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int COUTP = NTiles<COUT>::COUTP;
    constexpr int C4 = CIN / 4;
    __shared__ float mask_s[TM];
    __shared__ float red_s[4 * 2 * COUT];
    __shared__ __attribute__((aligned(16))) float out_s[TM * (COUT + 4)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    // (XCD-aware tile ranges, as in conv_b3.hip, were measured here: same time, same traffic -- the taps' reuse is inside a tile)
    const int64_t q0 = (int64_t)blockIdx.x * TM;
    const int64_t qo = q0 + wave * 32 + i;
    const bool inter = interior_row(qo, go);
    int yo = 0, xo = 0;
    int64_t base_row = 0;
    if (inter) {
        const int64_t b = qo / go.img;
        const int rr = (int)(qo - b * go.img);
        const int ypo = rr / go.Wp;
        yo = ypo - 1;
        xo = rr - ypo * go.Wp - 1;
        base_row = b * gi.img + (int64_t)(2 * yo) * gi.Wp + 2 * xo;  // padded input coords of tap (0,0)
    }
    if (gk == 0) mask_s[wave * 32 + i] = inter ? 1.0f : 0.0f;

    f32x16 acc[NT], acc_sc[SC ? NT : 1];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[n][r] = 0.0f;
            if (SC) acc_sc[n][r] = 0.0f;
        }

    const float *w_base = wt + (gk * COUTP + i) * 4;
    // Operands come straight from HBM/L2 (gathered rows) and L1/L2 (weights): a whole tap's fragments are requested
    // together, so their latencies overlap each other; co-resident waves cover the rest.
    // Lanes of non-interior output rows gather from the first image (base_row = 0: in-bounds); their accumulators are
    // discarded by the row mask in the epilogue, so the loop carries no predicate.
    constexpr int G = CIN / 8;
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = (TAPS == 9) ? tap / 3 : 1, kx = (TAPS == 9) ? tap % 3 : 1;
        const float *ap = in + (base_row + (int64_t)ky * gi.Wp + kx) * CIN + 4 * gk;
        const float *wp = w_base + tap * (C4 * COUTP * 4);
        float4 av[G], bv[G][NT];  // the whole tap's fragments requested together, then its MFMAs
#pragma unroll
        for (int c8 = 0; c8 < G; ++c8) {
            av[c8] = *reinterpret_cast<const float4 *>(ap + c8 * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[c8][n] = *reinterpret_cast<const float4 *>(wp + (c8 * 2 * COUTP + n * 32) * 4);
        }
#pragma unroll
        for (int c8 = 0; c8 < G; ++c8) {
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32(av[c8].x, bv[c8][n].x, acc[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32(av[c8].y, bv[c8][n].y, acc[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32(av[c8].z, bv[c8][n].z, acc[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32(av[c8].w, bv[c8][n].w, acc[n]);
        }
        if (SC && tap == 4) {   // centre tap: the 1x1 shortcut on the same A fragments (1x1 image = one tap slot)
            const float *wp2 = wt_sc + (gk * COUTP + i) * 4;
#pragma unroll
            for (int c8 = 0; c8 < G; ++c8)
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[c8][n] = *reinterpret_cast<const float4 *>(wp2 + (c8 * 2 * COUTP + n * 32) * 4);
#pragma unroll
            for (int c8 = 0; c8 < G; ++c8) {
#pragma unroll
                for (int n = 0; n < NT; ++n) acc_sc[n] = mfma32(av[c8].x, bv[c8][n].x, acc_sc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc_sc[n] = mfma32(av[c8].y, bv[c8][n].y, acc_sc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc_sc[n] = mfma32(av[c8].z, bv[c8][n].z, acc_sc[n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc_sc[n] = mfma32(av[c8].w, bv[c8][n].w, acc_sc[n]);
            }
        }
    }
    __syncthreads();
    conv_epilogue<COUT>(acc, bias, nullptr, out, partials, mask_s, out_s, red_s, q0, go.rows, scale, relu);
    if (SC) {
        __syncthreads();   // the first epilogue's use of out_s / red_s is over
        conv_epilogue<COUT>(acc_sc, nullptr, nullptr, out_sc, partials_sc, mask_s, out_s, red_s, q0, go.rows, nullptr, 0);
    }
}

// weight image for the MFMA kernels: wt[tap][K/4][NP][4] with K = GEMM-K channels, N = GEMM-N channels.
// mode 0 (forward):  K = cin,  N = cout, wt[tap][ci/4][co][ci%4] = w[co][ci][tap]
// mode 1 (dgrad):    K = cout, N = cin,  wt[tap][co/4][ci][co%4] = w[co][ci][taps-1-tap]   (flipped kernel)
__global__ void repack_kernel(const float *__restrict__ w, float *__restrict__ wt, int cout, int cin, int taps,
                              int mode) {
    const int K = mode == 0 ? cin : cout;
    const int N = mode == 0 ? cout : cin;
    const int NP = ((N + 31) / 32) * 32;
    const int total = taps * K * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3;
        int t = idx >> 2;
        const int n = t % NP;
        t /= NP;
        const int k4 = t % (K / 4);
        const int tap = t / (K / 4);
        const int k = k4 * 4 + e;
        float v = 0.0f;
        if (n < N) {
            const int co = mode == 0 ? n : k;
            const int ci = mode == 0 ? k : n;
            const int src_tap = mode == 0 ? tap : taps - 1 - tap;
            v = w[((int64_t)co * cin + ci) * taps + src_tap];
        }
        wt[idx] = v;
    }
}

// every convolution of the model in ONE launch: blockIdx.y selects the (layer, mode) descriptor
struct PackDesc {
    const float *w;
    float *wt;
    int cout, cin, taps, mode;
};
__global__ void repack_multi_kernel(const PackDesc *__restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const int K = d.mode == 0 ? d.cin : d.cout;
    const int N = d.mode == 0 ? d.cout : d.cin;
    const int NP = ((N + 31) / 32) * 32;
    const int total = d.taps * K * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 3;
        int t = idx >> 2;
        const int n = t % NP;
        t /= NP;
        const int k4 = t % (K / 4);
        const int tap = t / (K / 4);
        const int k = k4 * 4 + e;
        float v = 0.0f;
        if (n < N) {
            const int co = d.mode == 0 ? n : k;
            const int ci = d.mode == 0 ? k : n;
            const int src_tap = d.mode == 0 ? tap : d.taps - 1 - tap;
            v = d.w[((int64_t)co * d.cin + ci) * d.taps + src_tap];
        }
        d.wt[idx] = v;
    }
}

// up[b][2yo+1][2xo+1][:] = src[b][yo+1][xo+1][:] (padded coords), everything else zero: turns the gradient of a
// stride-2 convolution into the input of the stride-1 dgrad / wgrad kernels.
__global__ void upsample2_kernel(const float *__restrict__ src, float *__restrict__ up, Geom gs, Geom gu, int c4n) {
    const int64_t total = gu.rows * c4n;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = idx / c4n;
        const int c4 = (int)(idx - q * c4n);
        const int64_t b = q / gu.img;
        const int rr = (int)(q - b * gu.img);
        const int yp = rr / gu.Wp, xp = rr - yp * gu.Wp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < gu.body && (yp & 1) && (xp & 1)) {
            const int yo = (yp - 1) >> 1, xo = (xp - 1) >> 1;
            if (yo <= gs.Hp - 2 && xo <= gs.Wp - 2) {
                const int64_t qs = b * gs.img + (int64_t)(yo + 1) * gs.Wp + (xo + 1);
                v = reinterpret_cast<const float4 *>(src)[qs * c4n + c4];
            }
        }
        reinterpret_cast<float4 *>(up)[idx] = v;
    }
}

template <int CIN, int COUT, int TAPS, int EPI, int RB>
int launch_s1_rb(const float *in, const float *wt, const float *bias, const float *addend, float *out, float *partials,
                 const Geom &g, hipStream_t st, const float *scale, int relu, BnStat bst, bool probe_only) {
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TM * RB + 2 * halo;
    using C = S1Cfg<CIN, COUT, TAPS>;
    constexpr int PRE = S1Pre<RB>::N;
    const int batch_rows = PRE * (THREADS / (C::KC / 4));
    const int arows = RB == 1 ? nrows : ((nrows + batch_rows - 1) / batch_rows) * batch_rows;  // as in the kernel
    const size_t main_floats = std::max<size_t>(2 * (size_t)C::CHUNK_FLOATS + (size_t)arows * (C::KC + 4), (size_t)TM * (COUT + 4));
    const size_t lds = (main_floats + TM * RB + 8 * COUT) * sizeof(float);
    const bool fits = lds <= (RB == 1 ? 160 : 80) * 1024 &&
                      ((C::CPT == 1 && RB == 1) || (int64_t)nrows * (C::KC / 4) <= (int64_t)PRE * THREADS);  // later stages: one register batch
    if (probe_only) return fits ? LAD_OK : LAD_ERR_INVALID;
    if (!fits) return lad::fail(LAD_ERR_INVALID, "conv_s1: image too wide for the LDS tile (W = %d)", g.Wp - 1);
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_s1_kernel<CIN, COUT, TAPS, EPI, RB>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const unsigned grid = (unsigned)lad::ceil_div(g.rows, TM * RB);
    hipLaunchKernelGGL((conv_s1_kernel<CIN, COUT, TAPS, EPI, RB>), dim3(grid), dim3(THREADS), lds, st, in, wt, bias, addend, out,
                       partials, g, scale, relu, bst);
    return lad::check_launch("conv_s1_kernel");
}

template <int CIN, int COUT, int TAPS, int EPI = EPI_PLAIN>
int launch_s1(const float *in, const float *wt, const float *bias, const float *addend, float *out, float *partials,
              const Geom &g, hipStream_t st, const float *scale = nullptr, int relu = 0,
              BnStat bst = BnStat{nullptr, nullptr, nullptr}) {
    if (g.rows >= (1ll << 31) || g.img >= (1 << 20))
        return lad::fail(LAD_ERR_INVALID, "conv_s1: tensor of %lld rows / image of %d positions exceeds the 32-bit row decode",
                         (long long)g.rows, g.img);
    // 256-row workgroups (two row blocks per wavefront) where the tile fits and the launch still fills the chip
    constexpr bool WIDE = (CIN == 64 && COUT == 64 && TAPS == 9);
    if (WIDE && g.rows >= 512ll * 256 &&
        launch_s1_rb<CIN, COUT, TAPS, EPI, WIDE ? 2 : 1>(in, wt, bias, addend, out, partials, g, st, scale, relu, bst, true) == LAD_OK)
        return launch_s1_rb<CIN, COUT, TAPS, EPI, WIDE ? 2 : 1>(in, wt, bias, addend, out, partials, g, st, scale, relu, bst, false);
    return launch_s1_rb<CIN, COUT, TAPS, EPI, 1>(in, wt, bias, addend, out, partials, g, st, scale, relu, bst, false);
}

template <int CIN, int COUT, int TAPS>
int launch_s2(const float *in, const float *wt, const float *bias, float *out, float *partials, const Geom &gi,
              const Geom &go, hipStream_t st, const float *scale = nullptr, int relu = 0) {
    const unsigned grid = (unsigned)lad::ceil_div(go.rows, TM);
    hipLaunchKernelGGL((conv_s2_kernel<CIN, COUT, TAPS>), dim3(grid), dim3(THREADS), 0, st, in, wt, bias, out, partials,
                       gi, go, scale, relu);
    return lad::check_launch("conv_s2_kernel");
}

}  // namespace

#ifdef LAD_STAMP
extern "C" int lad_debug_read_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int64_t lad_conv_num_tiles(int64_t batch, int32_t H, int32_t W) {
    if (batch < 0 || H < 1 || W < 1) return -1;
    return lad::ceil_div(lad::make_geom(batch, H, W).rows, TM);
}

extern "C" int64_t lad_act_rows(int64_t batch, int32_t H, int32_t W) {
    if (batch < 0 || H < 1 || W < 1) return -1;
    return lad::make_geom(batch, H, W).rows;
}

extern "C" int64_t lad_conv_packed_weight_floats(int32_t cout, int32_t cin, int32_t taps, int32_t mode) {
    const int K = mode == 0 ? cin : cout, N = mode == 0 ? cout : cin;
    return (int64_t)taps * K * (((N + 31) / 32) * 32);
}

extern "C" int lad_conv_pack_weights(const float *w, int32_t cout, int32_t cin, int32_t taps, int32_t mode, float *wt,
                                     void *stream) {
    using namespace lad;
    LAD_REQUIRE(w && wt, "lad_conv_pack_weights: null buffer");
    LAD_REQUIRE((taps == 9 || taps == 1) && (mode == 0 || mode == 1), "lad_conv_pack_weights: bad taps/mode");
    const int K = mode == 0 ? cin : cout;
    LAD_REQUIRE(K % 8 == 0 && cin > 0 && cout > 0, "lad_conv_pack_weights: GEMM-K channels (%d) must be a multiple of 8", K);
    const int64_t total = lad_conv_packed_weight_floats(cout, cin, taps, mode);
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wt,
                       cout, cin, taps, mode);
    return check_launch("repack_kernel");
}

extern "C" int lad_conv_pack_weights_multi(const void *descs, int32_t n_desc, void *stream) {
    using namespace lad;
    LAD_REQUIRE(descs && n_desc >= 1, "lad_conv_pack_weights_multi: bad argument");
    static_assert(sizeof(PackDesc) == 32, "PackDesc is { const float*, float*, int32 x 4 }: 32 bytes, as the host writes it");
    hipLaunchKernelGGL(repack_multi_kernel, dim3(64, (unsigned)n_desc), dim3(256), 0, (hipStream_t)stream, (const PackDesc *)descs);
    return check_launch("repack_multi_kernel");
}

#define LAD_S1_CASE(CI, CO, T)                                                                         \
    if (cin == CI && cout == CO && taps == T)                                                          \
        return launch_s1<CI, CO, T>(in, wt, bias, addend, out, stat_partials, g, (hipStream_t)stream);

extern "C" int lad_conv_fwd(const float *in, const float *wt, const float *bias, const float *addend, float *out,
                            float *stat_partials, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout,
                            int32_t taps, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out, "lad_conv_fwd: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = make_geom(batch, H, W);
    LAD_S1_CASE(64, 64, 9)
    LAD_S1_CASE(32, 32, 9)
    LAD_S1_CASE(16, 16, 9)
    LAD_S1_CASE(32, 64, 9)
    LAD_S1_CASE(16, 32, 9)
    LAD_S1_CASE(32, 64, 1)
    LAD_S1_CASE(16, 32, 1)
    LAD_S1_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_fwd: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

namespace {
template <int CIN, int COUT>
int launch_s2_sc(const float *in, const float *wt, const float *bias, const float *wt_sc, float *out, float *partials, float *out_sc,
                 float *partials_sc, const Geom &gi, const Geom &go, hipStream_t st) {
    const unsigned grid = (unsigned)lad::ceil_div(go.rows, TM);
    hipLaunchKernelGGL((conv_s2_kernel<CIN, COUT, 9, true>), dim3(grid), dim3(THREADS), 0, st, in, wt, bias, out, partials, gi, go,
                       nullptr, 0, wt_sc, out_sc, partials_sc);
    return lad::check_launch("conv_s2_kernel<SC>");
}
}  // namespace

#define LAD_S2_CASE(CI, CO, T)                                                                     \
    if (cin == CI && cout == CO && taps == T)                                                      \
        return launch_s2<CI, CO, T>(in, wt, bias, out, stat_partials, gi, go, (hipStream_t)stream);

extern "C" int lad_conv_s2_fwd(const float *in, const float *wt, const float *bias, float *out, float *stat_partials,
                               int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                               void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out, "lad_conv_s2_fwd: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_s2_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom gi = make_geom(batch, H, W);
    const Geom go = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_S2_CASE(64, 32, 9)
    LAD_S2_CASE(32, 16, 9)
    LAD_S2_CASE(16, 16, 9)
    LAD_S2_CASE(64, 32, 1)
    LAD_S2_CASE(32, 16, 1)
    LAD_S2_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_fwd: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

// 3x3 stride-2 convolution (bias) and the 1x1 stride-2 shortcut convolution (no bias) of the same block in one launch
#define LAD_S2F_CASE(CI, CO)     \
    if (cin == CI && cout == CO) \
        return launch_s2_sc<CI, CO>(in, wt, bias, wt_sc, out, stat_partials, out_sc, stat_partials_sc, gi, go, (hipStream_t)stream);

extern "C" int lad_conv_s2_fwd_fused(const float *in, const float *wt, const float *bias, const float *wt_sc, float *out,
                                     float *stat_partials, float *out_sc, float *stat_partials_sc, int64_t batch, int32_t H,
                                     int32_t W, int32_t cin, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && wt_sc && out && out_sc && stat_partials && stat_partials_sc, "lad_conv_s2_fwd_fused: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_s2_fwd_fused: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom gi = make_geom(batch, H, W);
    const Geom go = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_S2F_CASE(64, 32)
    LAD_S2F_CASE(32, 16)
    LAD_S2F_CASE(16, 16)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_fwd_fused: unsupported (cin=%d, cout=%d)", cin, cout);
}

// Data-gradient launch fused with the first pass of the BatchNorm backward that consumes its output (see BnStat).
#define LAD_S1B_CASE(CI, CO, T)                                                                        \
    if (cin == CI && cout == CO && taps == T)                                                          \
        return launch_s1<CI, CO, T, EPI_BNSTAT>(in, wt, nullptr, addend, out, stat_partials, g, (hipStream_t)stream, nullptr, 0, bst);

extern "C" int lad_conv_fwd_bnstat(const float *in, const float *wt, const float *addend, float *out, float *stat_partials,
                                   const float *bn_x, const float *bn_y, const float *bn_coef, int64_t batch, int32_t H,
                                   int32_t W, int32_t cin, int32_t cout, int32_t taps, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out && stat_partials && bn_x && bn_coef, "lad_conv_fwd_bnstat: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_fwd_bnstat: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = make_geom(batch, H, W);
    const BnStat bst{bn_x, bn_y, bn_coef};
    LAD_S1B_CASE(64, 64, 9)
    LAD_S1B_CASE(32, 32, 9)
    LAD_S1B_CASE(16, 16, 9)
    return fail(LAD_ERR_INVALID, "lad_conv_fwd_bnstat: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

// Eval-mode convolutions with the following BatchNorm (running statistics) folded in:
// out = [relu](conv(in) * scale + shift [+ addend]); scale/shift from lad_bn_fold.
#define LAD_S1E_CASE(CI, CO, T)                                                                       \
    if (cin == CI && cout == CO && taps == T)                                                         \
        return launch_s1<CI, CO, T, EPI_EVAL>(in, wt, shift, addend, out, nullptr, g, (hipStream_t)stream, scale, relu);

extern "C" int lad_conv_fwd_eval(const float *in, const float *wt, const float *scale, const float *shift,
                                 const float *addend, float *out, int64_t batch, int32_t H, int32_t W, int32_t cin,
                                 int32_t cout, int32_t taps, int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out && scale && shift, "lad_conv_fwd_eval: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_fwd_eval: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = make_geom(batch, H, W);
    LAD_S1E_CASE(64, 64, 9)
    LAD_S1E_CASE(32, 32, 9)
    LAD_S1E_CASE(16, 16, 9)
    return fail(LAD_ERR_INVALID, "lad_conv_fwd_eval: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

#define LAD_S2E_CASE(CI, CO, T)                                                                    \
    if (cin == CI && cout == CO && taps == T)                                                      \
        return launch_s2<CI, CO, T>(in, wt, shift, out, nullptr, gi, go, (hipStream_t)stream, scale, relu);

extern "C" int lad_conv_s2_fwd_eval(const float *in, const float *wt, const float *scale, const float *shift, float *out,
                                    int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                                    int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out && scale && shift, "lad_conv_s2_fwd_eval: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_s2_fwd_eval: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom gi = make_geom(batch, H, W);
    const Geom go = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_S2E_CASE(64, 32, 9)
    LAD_S2E_CASE(32, 16, 9)
    LAD_S2E_CASE(16, 16, 9)
    LAD_S2E_CASE(64, 32, 1)
    LAD_S2E_CASE(32, 16, 1)
    LAD_S2E_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_fwd_eval: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

extern "C" int lad_upsample2(const float *src, float *up, int64_t batch, int32_t H, int32_t W, int32_t channels,
                             void *stream) {
    using namespace lad;
    LAD_REQUIRE(src && up, "lad_upsample2: null buffer");
    LAD_REQUIRE(channels % 4 == 0 && channels > 0, "lad_upsample2: channels must be a multiple of 4");
    if (batch == 0) return LAD_OK;
    const Geom gu = make_geom(batch, H, W);
    const Geom gs = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    const int64_t total = gu.rows * (channels / 4);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(upsample2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, up, gs, gu, channels / 4);
    return check_launch("upsample2_kernel");
}
