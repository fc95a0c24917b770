// 64 -> 64 3x3 stride-1 convolution on the 16-bit matrix cores with fp32-equivalent arithmetic ("bf16 x 3").
//
// Replaces the four nn.Conv2d(64, 64, 3, padding=1) of block1 (models.py:86-95,110-115) in forward and in the
// data-gradient direction: 8 launches = 43 % of the training step on the exact-f32 MFMA (conv_mfma.hip, 1.31 ms each at
// 80 % of the 157 TFLOP/s fp32-matrix peak -- that pipe is 16x slower than the bf16 one).
//
// Arithmetic.  Every fp32 operand is the EXACT sum of three bf16 numbers, x = x1 + x2 + x3 with x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (8 + 8 + 8 significant bits; bf16 has the exponent range of fp32, so no
// scaling is needed for gradients of 1e-8).  A product is then a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1 (+ terms
// below 2^-24 of it, dropped): six v_mfma_f32_32x32x16_bf16 with f32 accumulation, each product exact in f32.  Measured
// against float64 this is as accurate as the fmaf chain of the f32 MFMA (2.9e-7 vs 4.4e-7 of the largest output over
// K = 576), at 6 / 16 of its matrix-core time.  Tensors arrive pre-split ("split3": [C/16 groups][rows][plane][16 bf16],
// 6 bytes per element, lad_split3), weights are split when they are packed.
//
// Structure = conv_s1_kernel (conv_mfma.hip): a workgroup owns 128 consecutive output rows and all 64 output channels;
// the input rows it needs (+ a halo of W+2 either side) are staged into LDS 32 channels (x 3 planes) at a time through a
// buffer resource; the packed weight image streams chunk by chunk (one tap x 32 channels x 3 planes = 12 KB) through a
// two-slot LDS ring by LDS-DMA, one chunk ahead of the MFMAs.  Epilogue: bias, optional addend, zero border rows,
// per-tile (sum, sum of squares) for the train-mode BatchNorm -- the same f32 output tensor as the f32 kernel.
#include "lad_common.h"
#include "lad_device.h"
#include "lad_b3.h"
#include "lad_b3_tile.h"

#include <algorithm>
#include <cstdlib>

namespace {
using namespace lad;

// C = input = output channels: 64 (block1, the kernel this file was written for) or 32 (block2's stride-1 convolutions) -- a
// template parameter of everything below; the comments quote the 64-channel figures.
using namespace lad::b3t;   // TAPS, TM, THREADS, B3Stat, B3Scatter, b3_epilogue, dma_per_tap, wait_dma: lad_b3_tile.h
constexpr int GROW_B = 3 * 16 * 2;          // bytes of one row of one 16-channel group in HBM: [plane][16 bf16]
template <int C>
struct Ch {
    static constexpr int NT = C / 32;                          // 32-column output tiles per wavefront
    static constexpr int G16_BYTES = 3 * NT * 2 * 32 * 16;     // one 16-channel group of a tap: [plane][ntile][k half][n][8 bf16] = 6144
    static constexpr int IMG_BYTES = TAPS * (C / 16) * G16_BYTES;   // packed weight image: [tap][c16 group][...]
};

// KC = input channels per stage (resident in LDS at a time) = channels per weight chunk.  16: 37 KB of LDS, four
// workgroups per CU; 32: 73 KB, two.
template <int C, int KC, int RB = 1>
struct Cfg {
    static constexpr int NSTAGE = C / KC;
    static constexpr int NG = KC / 16;                    // 16-channel groups per chunk
    static constexpr int ROWB_L = 3 * KC * 2 + 16;        // bytes per staged row in LDS: 3 planes x KC bf16 + 16 (conflict-free b128)
    static constexpr int PIECES = 3 * KC * 2 / 16;        // 16-byte pieces per staged row
    static constexpr int CHUNK_BYTES = NG * Ch<C>::G16_BYTES;
    static constexpr int NCHUNK = TAPS * NSTAGE;
    static constexpr int TMW = TM * RB;                   // output rows per workgroup: RB row blocks of 32 per wavefront
    static constexpr int PRE = ((TMW + 2 * 47) * PIECES + THREADS - 1) / THREADS;   // registers for one stage at the widest image (W = 46)
};

// ---- weights: (cout, cin, 3, 3) fp32 -> [tap][16-channel group][plane][ntile][k half g][n][8 bf16] ----------------------
// (a chunk of the kernel = KC / 16 consecutive groups of one tap.)
// mode 0: forward, GEMM K = cin, N = cout.  mode 1: data gradient, K = cout, N = cin, taps flipped (as repack_kernel).
template <int C>
__global__ void pack_b3_kernel(const float *__restrict__ w, unsigned short *__restrict__ wt, int mode) {
    constexpr int NT = Ch<C>::NT;
    const int total = Ch<C>::IMG_BYTES / 2;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int n = t & 31; t >>= 5;
        const int g = t & 1; t >>= 1;
        const int nt = t % NT; t /= NT;
        const int plane = t % 3; t /= 3;
        const int c16 = t % (C / 16);
        const int tap = t / (C / 16);
        const int k = c16 * 16 + g * 8 + e;
        const int nn = nt * 32 + n;
        const int co = mode == 0 ? nn : k, ci = mode == 0 ? k : nn;
        const int src_tap = mode == 0 ? tap : TAPS - 1 - tap;
        const float v = w[((int64_t)co * C + ci) * TAPS + src_tap];
        const __bf16 b1 = (__bf16)v;
        const float r1 = v - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const __bf16 b3 = (__bf16)(r1 - (float)b2);
        const __bf16 pick = plane == 0 ? b1 : (plane == 1 ? b2 : b3);
        wt[idx] = __builtin_bit_cast(unsigned short, pick);
    }
}

// ---- the convolution ----------------------------------------------------------------------------------------------
template <int C, int KC>
__device__ __forceinline__ void issue_chunk(const unsigned char *__restrict__ wt, unsigned char *slot, int tap, int stage, int tid, int wave) {
    using K = Cfg<C, KC>;
    const unsigned char *src = wt + (int64_t)(tap * (C / 16) + stage * K::NG) * Ch<C>::G16_BYTES;
    static_assert(K::CHUNK_BYTES % (64 * 16) == 0, "whole wave-instructions");
#pragma unroll
    for (int r = 0; r * THREADS * 16 < K::CHUNK_BYTES; ++r)
        if ((r * THREADS + wave * 64) * 16 < K::CHUNK_BYTES)   // wave-uniform
            dma16(src + (r * THREADS + tid) * 16, lds_addr(slot + (r * THREADS + wave * 64) * 16));
}

// (The round-2 kernel conv_b3_kernel -- v_mfma_f32_32x32x16_bf16, rolled taps, 16-channel stages -- and the ring / chunk variants of
// conv_b3x_kernel below other than (1 tap per chunk, 2 slots) are in tools/experiments/retired/conv_b3_variants.hip: round 5.)
template <int C, int TPC, int NSLOT>
struct CfgX {
    static constexpr int RB = 2, KC = 16;
    static constexpr int NSTAGE = C / KC;
    static constexpr int NCT = C / 16;                        // 16-column tiles
    static constexpr int ROWB = 3 * KC * 2;                   // bytes per staged row: [plane][16 bf16]
    static constexpr int PIECES = ROWB / 16;
    static constexpr int TAP_BYTES = Ch<C>::G16_BYTES;        // weights of one tap x 16 channels: [plane][ntile][k half][n][8 bf16]
    static constexpr int PLANE_B = Ch<C>::NT * 1024;          // bytes per plane of it
    static constexpr int CHUNK_BYTES = TPC * TAP_BYTES;
    static constexpr int CPS = TAPS / TPC;                    // chunks per stage
    static constexpr int NCH = NSTAGE * CPS;
    static constexpr int TMW = TM * RB;
    static constexpr int KP = (CPS - NSLOT) < CPS / 2 ? (CPS - NSLOT) : CPS / 2;   // chunk boundary after which the next stage's rows are requested
    static_assert(TAPS % TPC == 0 && NSLOT >= 2 && NSLOT <= CPS && KP >= 0, "ring geometry");
    static constexpr int FPIECES = KC * 4 / 16;
    static constexpr int PRE = ((TMW + 2 * 47) * PIECES + THREADS - 1) / THREADS;
    static constexpr int PREF = ((TMW + 2 * 47) * FPIECES + THREADS - 1) / THREADS;
};

#ifdef LAD_STAMP
// diagnostic build only (tools/stamp_b3x.py): shader-clock stamps of wave 0 of every workgroup; never the product
__device__ unsigned long long lad_dbg_b3x[16 * 16384];
#define LAD_B3X_STAMP(k)                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_b3x[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LAD_B3X_STAMP(k)
#endif

// (a chunk of three taps needs 70 KB of LDS = two workgroups per CU anyway: that variant may use 256 registers)
template <int C, bool F32IN, bool STAT, bool INBN, int TPC, int NSLOT>
__global__ __launch_bounds__(THREADS, TPC == 3 ? 2 : 3) void conv_b3x_kernel(const unsigned char *__restrict__ in, const unsigned char *__restrict__ wt,
                                                              const float *__restrict__ bias, const float *addend,
                                                              const unsigned long long *__restrict__ abits, float *out,
                                                              float *__restrict__ partials, Geom g, B3Stat bst,
                                                              const float *__restrict__ in_coef) {
    static_assert(!INBN || F32IN, "the input BatchNorm is applied to fp32 rows");
    using K = CfgX<C, TPC, NSLOT>;
    constexpr int ROWB = K::ROWB, PIECES = K::PIECES, NSTAGE = K::NSTAGE, NCT = K::NCT, KC = K::KC, TMW = K::TMW;
    constexpr int TAP_BYTES = K::TAP_BYTES, CHUNK_BYTES = K::CHUNK_BYTES, CPS = K::CPS, PLANE_B = K::PLANE_B;
    constexpr int FPIECES = K::FPIECES, NPRE = F32IN ? K::PREF : K::PRE;
    constexpr bool STATIC_SLOT = (CPS % NSLOT) == 0;   // a stage starts in slot 0: every slot index is a compile-time number
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = g.Wp + 1;
    const int nrows = TMW + 2 * halo;
    const int main_bytes = max(NSLOT * CHUNK_BYTES + nrows * ROWB, TM * (C + 4) * 4);
    unsigned char *b_s = smem_b;                          // [NSLOT][CHUNK_BYTES]
    unsigned char *a_s = b_s + NSLOT * CHUNK_BYTES;       // [nrows][ROWB]
    unsigned char *mask_s = smem_b + main_bytes;          // [TMW] (bytes: with a float mask the third ring slot would cost the third workgroup per CU)
    // XCD-aware tile order (see conv_b3_kernel)
    const unsigned per_x = (gridDim.x + 7u) / 8u;
    const unsigned tile_id = (blockIdx.x % 8u) * per_x + blockIdx.x / 8u;
    const int64_t q0 = (int64_t)tile_id * TMW;
    if (q0 >= g.rows) return;

    LAD_B3X_STAMP(0)
    const int nw_tap = dma_per_tap<TAP_BYTES>(wave);
    // weights of (tap, stage) -> ring slot `slot`, sub-chunk tap % TPC
    auto issue_tap = [&](int tap, int stage, int slot) {
        const unsigned char *src = wt + (int64_t)(tap * (C / 16) + stage) * TAP_BYTES;
        unsigned char *dst = b_s + slot * CHUNK_BYTES + (tap % TPC) * TAP_BYTES;
#pragma unroll
        for (int r = 0; r * THREADS * 16 < TAP_BYTES; ++r)
            if ((r * THREADS + wave * 64) * 16 < TAP_BYTES)   // wave-uniform
                dma16(src + (r * THREADS + tid) * 16, lds_addr(dst + (r * THREADS + wave * 64) * 16));
    };

    // ---- staging of the input rows (as conv_b3_kernel; rows of 96 bytes) ------------------------------------------------
    const int64_t start = q0 - halo;
    const int64_t first = start < 0 ? 0 : start;
    const int row_lo = (int)(first - start);
    const int64_t span_rows = min(g.rows - first, (int64_t)(nrows - row_lo));
    const int64_t span_bytes = span_rows * GROW_B;
    const int64_t group_bytes = g.rows * GROW_B;
    const int vbase = tid * 16 - row_lo * GROW_B;
    auto voff = [&](int u) { return (u * THREADS + tid) < nrows * PIECES ? vbase + u * THREADS * 16 : -1; };
    auto voff_f = [&](int u) {
        const int idx = u * THREADS + tid;
        return idx < nrows * FPIECES ? ((idx >> 2) - row_lo) * (C * 4) + (idx & 3) * 16 : -1;
    };
    auto stage_rsrc = [&](int stage) {
        return F32IN ? make_rsrc(in + first * (C * 4) + stage * (KC * 4), span_rows * (C * 4) - stage * (KC * 4))
                     : make_rsrc(in + stage * group_bytes + first * GROW_B, span_bytes);
    };
    unsigned keep_bits = 0;
    f32x4 bn_sc = {0.f, 0.f, 0.f, 0.f}, bn_sh = bn_sc;
    auto load_in_coef = [&](int stage) {
        if (INBN) {
            bn_sc = *reinterpret_cast<const f32x4 *>(in_coef + stage * KC + (tid & 3) * 4);
            bn_sh = *reinterpret_cast<const f32x4 *>(in_coef + C + stage * KC + (tid & 3) * 4);
        }
    };
    auto put = [&](int u, u32x4 v) {
        const int idx = u * THREADS + tid;
        if (F32IN) {
            if (idx < nrows * FPIECES) {
                unsigned char *dst = a_s + (idx >> 2) * ROWB + (idx & 3) * 8;
                unsigned a1, a2, a3, b1, b2, b3;
                float4 f = as_f4(v);
                if (INBN) {
                    const bool keep = (keep_bits >> u) & 1u;
                    f.x = keep ? fmaxf(fmaf(f.x, bn_sc.x, bn_sh.x), 0.f) : 0.f;
                    f.y = keep ? fmaxf(fmaf(f.y, bn_sc.y, bn_sh.y), 0.f) : 0.f;
                    f.z = keep ? fmaxf(fmaf(f.z, bn_sc.z, bn_sh.z), 0.f) : 0.f;
                    f.w = keep ? fmaxf(fmaf(f.w, bn_sc.w, bn_sh.w), 0.f) : 0.f;
                }
                split_pair(f.x, f.y, a1, a2, a3);
                split_pair(f.z, f.w, b1, b2, b3);
                *reinterpret_cast<u32x2 *>(dst + 0 * (KC * 2)) = u32x2{a1, b1};
                *reinterpret_cast<u32x2 *>(dst + 1 * (KC * 2)) = u32x2{a2, b2};
                *reinterpret_cast<u32x2 *>(dst + 2 * (KC * 2)) = u32x2{a3, b3};
            }
        } else {
            if (idx < nrows * PIECES) *reinterpret_cast<u32x4 *>(a_s + idx * 16) = v;   // (unpadded rows: the span is copied as it is)
        }
    };

    // ---- prologue: the first stage's rows are requested first (HBM), then the first NSLOT - 1 weight chunks (L2) ---------
    u32x4 pre[NPRE];
    {
        const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(0);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = buf_load16(in_r, F32IN ? voff_f(u) : voff(u));
        load_in_coef(0);
    }
#pragma unroll
    for (int k = 0; k < NSLOT - 1; ++k)
#pragma unroll
        for (int t = 0; t < TPC; ++t) issue_tap(k * TPC + t, 0, k);
    for (int j = tid; j < TMW; j += THREADS) mask_s[j] = interior_row32((uint32_t)q0 + (uint32_t)j, g) ? 1 : 0;
    if (INBN) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u)
            keep_bits |= (interior_row32((uint32_t)(start + ((u * THREADS + tid) >> 2)), g) ? 1u : 0u) << u;
    }
    LAD_B3X_STAMP(1)
#pragma unroll
    for (int u = 0; u < NPRE; ++u) put(u, pre[u]);
    LAD_B3X_STAMP(2)

    // ---- fragments ------------------------------------------------------------------------------------------------------
    // A: lane (m = lane & 15, k half kg, upper hi) reads row (tile r: sub-tile r >> 1, rows wave * 32 + (r & 1) * 16 + m),
    //    16 bytes at plane * 32 + kg * 16 with plane = P: hi ? a2 : a1, Q: hi ? a1 : a3.
    // B: 16 bytes at plane * PLANE_B + (column tile c: (c >> 1) * 1024 + (c & 1) * 256) + kg * 512 + m * 16 with
    //    plane = U: hi ? b2 : b3, V: b1, W: hi ? b2 : b1.
    const int m = lane & 15, kg = (lane >> 4) & 1, hi = lane >> 5;
    const unsigned char *a_lane = a_s + (wave * 32 + m + halo - 1) * ROWB + kg * 16;   // (- 1: tap column offsets 0, 1, 2)
    const unsigned char *aP = a_lane + (hi ? 32 : 0);
    const unsigned char *aQ = a_lane + (hi ? 0 : 64);
    const unsigned char *b_lane = b_s + kg * 512 + m * 16;
    const unsigned char *bU = b_lane + (hi ? 1 : 2) * PLANE_B;
    const unsigned char *bV = b_lane;
    const unsigned char *bW = b_lane + (hi ? 1 : 0) * PLANE_B;
    const int wrow = g.Wp * ROWB;

    f32x4 acc[4][NCT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int slot0 = 0;   // ring slot of the current stage's first chunk (always 0 when STATIC_SLOT)
#pragma unroll 1
    for (int stage = 0; stage < NSTAGE; ++stage) {
        const bool last = stage + 1 == NSTAGE;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kc = tap / TPC;   // chunk of the stage
            const int slot = STATIC_SLOT ? kc % NSLOT : (slot0 + kc) % NSLOT;
            if (tap % TPC == 0) {
                // chunk kc has landed (this wave's part), then everybody's; the slot of chunk kc - 1 is free
                constexpr int YOUNGER = NSLOT - 2;
                if (!last) {
                    if (kc > K::KP && kc <= K::KP + NSLOT - 1) wait_dma<YOUNGER * TPC, NPRE>(nw_tap);   // + the requested rows
                    else wait_dma<YOUNGER * TPC, 0>(nw_tap);
                } else {
                    if (CPS - 1 - kc >= YOUNGER) wait_dma<YOUNGER * TPC, 0>(nw_tap);
                    else wait_dma<0, 0>(nw_tap);   // (NSLOT <= 3: at most one chunk is missing at the end)
                }
                __syncthreads();
                const int kn = kc + NSLOT - 1;   // the chunk to request now
                const int slot_n = STATIC_SLOT ? kn % NSLOT : (slot0 + kn) % NSLOT;
                if (kn < CPS) {
#pragma unroll
                    for (int t = 0; t < TPC; ++t) issue_tap(kn * TPC + t, stage, slot_n);
                } else if (!last) {
#pragma unroll
                    for (int t = 0; t < TPC; ++t) issue_tap((kn - CPS) * TPC + t, stage + 1, slot_n);
                }
                if (kc == K::KP && !last) {
                    const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(stage + 1);
#pragma unroll
                    for (int u = 0; u < NPRE; ++u) pre[u] = buf_load16(in_r, F32IN ? voff_f(u) : voff(u));
                }
            }
            const int off = (tap / 3 - 1) * wrow + (tap % 3) * ROWB;
            const int boff = slot * CHUNK_BYTES + (tap % TPC) * TAP_BYTES;
            bf16x8 ap[4], aq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int roff = ((r >> 1) * TM + (r & 1) * 16) * ROWB;
                ap[r] = *reinterpret_cast<const bf16x8 *>(aP + off + roff);
                aq[r] = *reinterpret_cast<const bf16x8 *>(aQ + off + roff);
            }
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int coff = boff + (c >> 1) * 1024 + (c & 1) * 256;
                const bf16x8 bu = *reinterpret_cast<const bf16x8 *>(bU + coff);
                const bf16x8 bw = *reinterpret_cast<const bf16x8 *>(bW + coff);
                const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(bV + coff);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bu, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[r], bw, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bv, acc[r][c], 0, 0, 0);
            }
        }
        if (!STATIC_SLOT) slot0 = (slot0 + CPS) % NSLOT;
#ifdef LAD_STAMP
        if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_b3x[blockIdx.x * 16 + 3 + 2 * stage] = __builtin_amdgcn_s_memtime();
#endif
        if (!last) {
            load_in_coef(stage + 1);
            __syncthreads();  // every wave has finished reading this stage's rows
#pragma unroll
            for (int u = 0; u < NPRE; ++u) put(u, pre[u]);
        }
#ifdef LAD_STAMP
        if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_b3x[blockIdx.x * 16 + 4 + 2 * stage] = __builtin_amdgcn_s_memtime();
#endif
    }
    __syncthreads();  // every wave is out of the MFMA loop: ring + input rows become the output tile
    LAD_B3X_STAMP(11)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int64_t qs = q0 + rb * TM;
        if (rb > 0) {
            if (qs >= g.rows) break;
            __syncthreads();
        }
        auto store_acc = [&](float *my) {   // D register j of lane l of tile (r, c): row 4 (l >> 4) + j, column l & 15
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) my[(rr * 16 + (lane >> 4) * 4 + j) * (C + 4) + c * 16 + m] = acc[rb * 2 + rr][c][j];
        };
        b3_epilogue<C, STAT>(store_acc, bias, addend, abits, out, partials, mask_s + rb * TM, reinterpret_cast<float *>(smem_b), qs, g.rows, bst);
    }
    LAD_B3X_STAMP(12)
}

template <int C, int TPC, int NSLOT>
size_t b3x_lds_bytes(const Geom &g) {
    using K = CfgX<C, TPC, NSLOT>;
    const int nrows = K::TMW + 2 * (g.Wp + 1);
    const size_t main_bytes = std::max<size_t>(NSLOT * K::CHUNK_BYTES + (size_t)nrows * K::ROWB, (size_t)TM * (C + 4) * 4);
    return main_bytes + K::TMW;   // + the row mask (bytes)
}

// =====================================================================================================================
// Round 3: the stride-2 transition 64 -> 32 (block2.0: 3x3 stride-2 convolution + its 1x1 stride-2 shortcut, models.py:98-106)
// on the split-operand path, "conv_s2b3".
//
// Round 2 ran these on the f32 matrix pipe with per-lane gathers from L2 (conv_s2_kernel: 377 us for 1.6 % of the step's
// FLOPs, bound by the rate of 16-byte-per-line gathers).  A stride-2 3x3 convolution is a stride-1 convolution over the
// space-to-depth view of its input: with z_{py,px}[I][J] = x[2I + py][2J + px] (four "parity classes" of 64 channels each,
// living in the OUTPUT's padded geometry, borders zero),
//     out[i][j] = sum over (ky, kx) of x[2i + ky - 1][2j + kx - 1] W[ky][kx] = sum over classes and their taps of z_class[i + dI][j + dJ] W[ky][kx]
// with ky = 0 -> (py 1, dI -1), ky = 1 -> (py 0, dI 0), ky = 2 -> (py 1, dI 0) and the same for columns: class (1,1) carries
// four taps, (1,0) and (0,1) two, (0,0) one -- and the 1x1 shortcut, which reads exactly class (0,0).  No tensor changes
// layout for this: the view is formed WHILE STAGING -- LDS row r of a stage holds the 16 channels of the x row that
// output position q0 - halo + r maps to in that class -- so every x element is fetched and split once per tile (+ 9 % halo,
// which is one-sided here: dI, dJ <= 0), and the MFMA loop is conv_b3x's: row-shifted 16-byte fragment reads.
// 16 stages (class x 16-channel group); a stage's K blocks (its taps, + the shortcut in class (0,0)) are one LDS-DMA chunk.
namespace s2b3 {
constexpr int CIN = 64, COUT = 32, NBLOCKS = 40;               // K blocks of 16 channels: 36 of the 3x3 + 4 of the shortcut
constexpr int BLOCK_BYTES = Ch<COUT>::G16_BYTES;                // [plane][k half][n 32][8 bf16] = 3072
constexpr int IMG_BYTES = NBLOCKS * BLOCK_BYTES;
constexpr int SLOT_BYTES = 4 * BLOCK_BYTES;
constexpr int ROWB = 96;
constexpr int TMW = 2 * TM;
constexpr int MAXROWS = TMW + 47;                               // W_out <= 45
constexpr int NPRE = (MAXROWS * 4 + THREADS - 1) / THREADS;     // 16-byte pieces per thread and stage
__host__ __device__ constexpr int py_of(int cls) { return cls < 2 ? 1 : 0; }
__host__ __device__ constexpr int px_of(int cls) { return (cls == 0 || cls == 2) ? 1 : 0; }
__host__ __device__ constexpr int ntaps(int cls) { return cls == 0 ? 4 : (cls == 3 ? 1 : 2); }
__host__ __device__ constexpr int nblocks(int cls) { return cls == 0 ? 4 : 2; }          // K blocks per stage (class 3: tap + shortcut)
__host__ __device__ constexpr int first_block(int cls) { return cls == 0 ? 0 : (cls == 1 ? 16 : (cls == 2 ? 24 : 32)); }
// tap t of a class: its (dI, dJ) in the class tensor and its (ky, kx) in the 3x3 kernel
__host__ __device__ constexpr int dI_of(int cls, int t) { return cls == 0 ? ((t >> 1) ? 0 : -1) : (cls == 1 ? (t ? 0 : -1) : 0); }
__host__ __device__ constexpr int dJ_of(int cls, int t) { return cls == 0 ? ((t & 1) ? 0 : -1) : (cls == 2 ? (t ? 0 : -1) : 0); }
__host__ __device__ constexpr int ky_of(int cls, int t) { return py_of(cls) ? (dI_of(cls, t) ? 0 : 2) : 1; }
__host__ __device__ constexpr int kx_of(int cls, int t) { return px_of(cls) ? (dJ_of(cls, t) ? 0 : 2) : 1; }
}  // namespace s2b3

// (cout 32, cin 64, 3, 3) + (32, 64, 1, 1) -> the forward image: blocks in stage order, class -> channel group -> tap (+ shortcut)
__device__ __forceinline__ void pack_s2b3_fwd_image(const float *__restrict__ w, const float *__restrict__ w_sc, unsigned short *__restrict__ wt,
                                                    int blk, int nblk) {
    using namespace s2b3;
    const int total = IMG_BYTES / 2;
    for (int idx = blk * blockDim.x + threadIdx.x; idx < total; idx += nblk * blockDim.x) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int n = t & 31; t >>= 5;
        const int gh = t & 1; t >>= 1;
        const int plane = t % 3;
        const int kb = t / 3;
        const int cls = kb < 16 ? 0 : (kb < 24 ? 1 : (kb < 32 ? 2 : 3));
        const int rel = kb - first_block(cls);
        const int cg = rel / nblocks(cls), tb = rel % nblocks(cls);
        const int ci = cg * 16 + gh * 8 + e;
        float v;
        if (cls == 3 && tb == 1) v = w_sc[n * CIN + ci];
        else v = w[((int64_t)n * CIN + ci) * 9 + ky_of(cls, tb) * 3 + kx_of(cls, tb)];
        const __bf16 b1 = (__bf16)v;
        const float r1 = v - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const __bf16 b3 = (__bf16)(r1 - (float)b2);
        const __bf16 pick = plane == 0 ? b1 : (plane == 1 ? b2 : b3);
        wt[idx] = __builtin_bit_cast(unsigned short, pick);
    }
}

__global__ void pack_s2b3_fwd_kernel(const float *__restrict__ w, const float *__restrict__ w_sc, unsigned short *__restrict__ wt) {
    pack_s2b3_fwd_image(w, w_sc, wt, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(THREADS, 3) void conv_s2b3_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wt,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               float *__restrict__ partials, float *__restrict__ out_sc,
                                                               float *__restrict__ partials_sc, Geom gi, Geom go) {
    using namespace s2b3;
    constexpr int NCT = COUT / 16, PLANE_B = Ch<COUT>::NT * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = go.Wp + 1;                         // one-sided: every tap looks up / left
    const int nrows = TMW + halo;
    const int main_bytes = max(2 * SLOT_BYTES + nrows * ROWB, TM * (COUT + 4) * 4);
    unsigned char *b_s = smem_b;                        // [2][SLOT_BYTES]
    unsigned char *a_s = b_s + 2 * SLOT_BYTES;          // [nrows][ROWB]
    unsigned char *mask_s = smem_b + main_bytes;        // [TMW]
    const unsigned per_x = (gridDim.x + 7u) / 8u;
    const unsigned tile_id = (blockIdx.x % 8u) * per_x + blockIdx.x / 8u;
    const int64_t q0 = (int64_t)tile_id * TMW;
    if (q0 >= go.rows) return;

    // weights of `nb` consecutive K blocks starting at block `kb` -> ring slot `slot`
    auto issue_chunk = [&](int kb, int nb, int slot) {
        const unsigned char *src = wt + (int64_t)kb * BLOCK_BYTES;
        unsigned char *dst = b_s + slot * SLOT_BYTES;
        const int bytes = nb * BLOCK_BYTES;
#pragma unroll
        for (int r = 0; r * THREADS * 16 < SLOT_BYTES; ++r)
            if ((r * THREADS + wave * 64) * 16 < bytes)   // wave-uniform
                dma16(src + (r * THREADS + tid) * 16, lds_addr(dst + (r * THREADS + wave * 64) * 16));
    };
    issue_chunk(0, nblocks(0), 0);

    // ---- where this thread's pieces come from: LDS row lr = idx >> 2 <-> output-geometry position q0 - halo + lr ------------
    // byte offset of the class-(0,0) x row of that position relative to the tile's first x row, or -1 (border / outside)
    const int64_t qa = max(q0 - halo, (int64_t)0);
    int64_t tile_row0;
    {
        const int64_t ba = qa / go.img;
        const int ipa = (int)((qa - ba * go.img) / go.Wp);
        tile_row0 = (ba * gi.Hp + max(2 * (ipa - 1), 0)) * gi.Wp;
    }
    int xoff[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int idx = u * THREADS + tid;
        const int64_t q = q0 - halo + (idx >> 2);
        xoff[u] = -1;
        if (idx < nrows * 4 && q >= 0 && interior_row32((uint32_t)q, go)) {
            const uint32_t qq = (uint32_t)q;
            uint32_t b = (uint32_t)((double)qq * go.inv_img);
            uint32_t rr = qq - b * (uint32_t)go.img;
            if (rr >= (uint32_t)go.img) { rr -= (uint32_t)go.img; ++b; }
            const uint32_t ip = __umulhi(rr, go.wp_magic), jp = rr - ip * (uint32_t)go.Wp;
            const int64_t xr = ((int64_t)b * gi.Hp + 2 * (ip - 1) + 1) * gi.Wp + 2 * (jp - 1) + 1;
            xoff[u] = (int)((xr - tile_row0) * (CIN * 4)) + (idx & 3) * 16;
        }
    }
    const int64_t tile_bytes = (gi.rows - tile_row0) * (CIN * 4);
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + tile_row0 * CIN, tile_bytes);
    u32x4 pre[NPRE];
    auto fetch = [&](int cls, int cg) {
        const int add = (py_of(cls) * gi.Wp + px_of(cls)) * (CIN * 4) + cg * 64;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = buf_load16(in_r, xoff[u] < 0 ? -1 : xoff[u] + add);
    };
    auto put = [&](int u) {
        const int idx = u * THREADS + tid;
        if (idx < nrows * 4) {
            unsigned char *dst = a_s + (idx >> 2) * ROWB + (idx & 3) * 8;
            unsigned a1, a2, a3, b1, b2, b3;
            const float4 f = as_f4(pre[u]);
            split_pair(f.x, f.y, a1, a2, a3);
            split_pair(f.z, f.w, b1, b2, b3);
            *reinterpret_cast<u32x2 *>(dst + 0 * 32) = u32x2{a1, b1};
            *reinterpret_cast<u32x2 *>(dst + 1 * 32) = u32x2{a2, b2};
            *reinterpret_cast<u32x2 *>(dst + 2 * 32) = u32x2{a3, b3};
        }
    };
    fetch(0, 0);
    for (int j = tid; j < TMW; j += THREADS) mask_s[j] = interior_row32((uint32_t)q0 + (uint32_t)j, go) ? 1 : 0;

    const int m = lane & 15, kg = (lane >> 4) & 1, hi = lane >> 5;
    const unsigned char *a_lane = a_s + (wave * 32 + m + halo) * ROWB + kg * 16;
    const unsigned char *aP = a_lane + (hi ? 32 : 0);
    const unsigned char *aQ = a_lane + (hi ? 0 : 64);
    const unsigned char *b_lane = b_s + kg * 512 + m * 16;
    const unsigned char *bU = b_lane + (hi ? 1 : 2) * PLANE_B;
    const unsigned char *bV = b_lane;
    const unsigned char *bW = b_lane + (hi ? 1 : 0) * PLANE_B;
    const int wrow = go.Wp * ROWB;

    f32x4 acc[4][NCT], acc_sc[4][NCT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = acc_sc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int stage = 0;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
#pragma unroll 1
        for (int cg = 0; cg < 4; ++cg, ++stage) {
            dma_wait_all();     // this stage's weights (this wave's part) and rows have arrived
            __syncthreads();    // ... everybody's; every wave is out of the previous stage's MFMAs
#pragma unroll
            for (int u = 0; u < NPRE; ++u) put(u);
            if (stage + 1 < 16) {   // the next stage's weights -> the other slot, its rows -> registers (in flight during the MFMAs)
                const int ncls = cg < 3 ? cls : cls + 1, ncg = cg < 3 ? cg + 1 : 0;
                issue_chunk(first_block(ncls) + ncg * nblocks(ncls), nblocks(ncls), (stage + 1) & 1);
                fetch(ncls, ncg);
            }
            __syncthreads();    // the rows are visible
            const int slot_off = (stage & 1) * SLOT_BYTES;
#pragma unroll
            for (int tb = 0; tb < nblocks(cls); ++tb) {
                const bool is_sc = cls == 3 && tb == 1;
                const int t = is_sc ? 0 : tb;
                const int off = dI_of(cls, t) * wrow + dJ_of(cls, t) * ROWB;
                const int boff = slot_off + tb * BLOCK_BYTES;
                bf16x8 ap[4], aq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int roff = ((r >> 1) * TM + (r & 1) * 16) * ROWB;
                    ap[r] = *reinterpret_cast<const bf16x8 *>(aP + off + roff);
                    aq[r] = *reinterpret_cast<const bf16x8 *>(aQ + off + roff);
                }
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    const int coff = boff + c * 256;
                    const bf16x8 bu = *reinterpret_cast<const bf16x8 *>(bU + coff);
                    const bf16x8 bw = *reinterpret_cast<const bf16x8 *>(bW + coff);
                    const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(bV + coff);
                    if (is_sc) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc_sc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bu, acc_sc[r][c], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc_sc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[r], bw, acc_sc[r][c], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc_sc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bv, acc_sc[r][c], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bu, acc[r][c], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[r], bw, acc[r][c], 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bv, acc[r][c], 0, 0, 0);
                    }
                }
            }
        }
    }
    const B3Stat none{nullptr, nullptr, nullptr};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {   // the 3x3 convolution's output, then the shortcut's
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int64_t qs = q0 + rb * TM;
            if (qs >= go.rows) break;
            __syncthreads();   // ring + rows (first round) / the previous use of the output tile are over
            auto store_acc = [&](float *my) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            my[(rr * 16 + (lane >> 4) * 4 + j) * (COUT + 4) + c * 16 + m] = pass == 0 ? acc[rb * 2 + rr][c][j] : acc_sc[rb * 2 + rr][c][j];
            };
            b3_epilogue<COUT, false>(store_acc, pass == 0 ? bias : nullptr, nullptr, nullptr, pass == 0 ? out : out_sc,
                                     pass == 0 ? partials : partials_sc, mask_s + rb * TM, reinterpret_cast<float *>(smem_b), qs, go.rows, none);
        }
    }
}

// ---- the data gradient of the same transition: dx = dgrad3x3(dout) + dgrad1x1(dout_sc) -------------------------------------------
// Class by class (blockIdx.y): the rows of x in parity class (py, px) receive
//     dz_class[q'] = sum over the class's taps of dout[q' - dI Wp' - dJ] W_tap^T   (+ dout_sc[q'] W_sc^T in class (0,0))
// -- a stride-1 row-shifted GEMM over the OUTPUT geometry with K = taps x 32 channels and N = 64, whose result rows are
// SCATTERED to x's rows (every other row of every other image row: whole 256-byte rows, so the stores stay coalesced).
// Border rows of dx are not written: they are zero by the layout invariant (as lad_conv_s2_dgrad).  The dout rows are staged
// contiguously, 16 channels at a time; stages: dout[0:16], dout[16:32], and in class (0,0) dout_sc[0:16], dout_sc[16:32].
// STAT: the epilogue also forms the first pass of the BatchNorm backward that consumes dx (lad_conv_s2_dgrad_fused_bnstat).
namespace s2b3 {
constexpr int DG_BLOCK_BYTES = Ch<CIN>::G16_BYTES;    // 16 dout channels x 64 dx channels: [plane][ntile 2][k half][n 32][8 bf16] = 6144
constexpr int DG_NBLOCKS = 20;                        // class (1,1): 4 taps x 2 groups; (1,0), (0,1): 2 x 2; (0,0): 1 x 2 + shortcut x 2
constexpr int DG_IMG_BYTES = DG_NBLOCKS * DG_BLOCK_BYTES;
__host__ __device__ constexpr int dg_first_block(int cls) { return cls == 0 ? 0 : (cls == 1 ? 8 : (cls == 2 ? 12 : 16)); }
}  // namespace s2b3

// block order: class -> stage (dout group 0, dout group 1[, shortcut group 0, shortcut group 1]) -> tap
__device__ __forceinline__ void pack_s2b3_dgrad_image(const float *__restrict__ w, const float *__restrict__ w_sc, unsigned short *__restrict__ wt,
                                                      int blk, int nblk) {
    using namespace s2b3;
    const int total = DG_IMG_BYTES / 2;
    for (int idx = blk * blockDim.x + threadIdx.x; idx < total; idx += nblk * blockDim.x) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int n = t & 31; t >>= 5;
        const int gh = t & 1; t >>= 1;
        const int ntile = t & 1; t >>= 1;
        const int plane = t % 3;
        const int kb = t / 3;
        const int cls = kb < 8 ? 0 : (kb < 12 ? 1 : (kb < 16 ? 2 : 3));
        const int rel = kb - dg_first_block(cls);
        const int nt_c = ntaps(cls);
        int stage, tap;
        if (rel < 2 * nt_c) { stage = rel / nt_c; tap = rel % nt_c; }
        else { stage = 2 + (rel - 2 * nt_c); tap = 0; }
        const int co = (stage & 1) * 16 + gh * 8 + e;     // K index: the dout channel
        const int ci = ntile * 32 + n;                    // N index: the dx channel
        float v;
        if (stage >= 2) v = w_sc[co * CIN + ci];
        else v = w[((int64_t)co * CIN + ci) * 9 + ky_of(cls, tap) * 3 + kx_of(cls, tap)];
        const __bf16 b1 = (__bf16)v;
        const float r1 = v - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const __bf16 b3 = (__bf16)(r1 - (float)b2);
        const __bf16 pick = plane == 0 ? b1 : (plane == 1 ? b2 : b3);
        wt[idx] = __builtin_bit_cast(unsigned short, pick);
    }
}
__global__ void pack_s2b3_dgrad_kernel(const float *__restrict__ w, const float *__restrict__ w_sc, unsigned short *__restrict__ wt) {
    pack_s2b3_dgrad_image(w, w_sc, wt, blockIdx.x, gridDim.x);
}
// both images of a training step in one launch: the first half of the grid packs the forward image, the second the data gradient's
__global__ void pack_s2b3_pair_kernel(const float *__restrict__ w, const float *__restrict__ w_sc, unsigned short *__restrict__ wt_fwd,
                                      unsigned short *__restrict__ wt_dgrad) {
    const int half = gridDim.x / 2;
    if ((int)blockIdx.x < half) pack_s2b3_fwd_image(w, w_sc, wt_fwd, blockIdx.x, half);
    else pack_s2b3_dgrad_image(w, w_sc, wt_dgrad, blockIdx.x - half, half);
}


template <bool STAT>
__global__ __launch_bounds__(THREADS, 3) void dgrad_s2b3_kernel(const float *__restrict__ dout, const float *__restrict__ dout_sc,
                                                                const unsigned char *__restrict__ wt, float *dx,
                                                                float *__restrict__ partials, Geom gi, Geom go, B3Stat bst,
                                                                int64_t tiles_x) {
    using namespace s2b3;
    constexpr int C = CIN, NCT = C / 16, PLANE_B = Ch<C>::NT * 1024, DCH = COUT;
    constexpr int DG_NPRE = ((TMW + 47) * 4 + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cls = blockIdx.y;
    const int py = py_of(cls), px = px_of(cls), nt_c = ntaps(cls);
    const int nstage = cls == 3 ? 4 : 2;
    const int nkb = 2 * nt_c + (cls == 3 ? 2 : 0);
    const int halo = go.Wp + 1;                         // one-sided, towards larger rows: the taps look down / right
    const int nrows = TMW + halo;
    const int main_bytes = max(2 * DG_BLOCK_BYTES + nrows * ROWB, TM * (C + 4) * 4);
    unsigned char *b_s = smem_b;                        // [2][DG_BLOCK_BYTES]
    unsigned char *a_s = b_s + 2 * DG_BLOCK_BYTES;      // [nrows][ROWB]
    int *rowoff_s = reinterpret_cast<int *>(smem_b + main_bytes);   // [TMW]
    const unsigned per_x = (gridDim.x + 7u) / 8u;
    const unsigned tile_id = (blockIdx.x % 8u) * per_x + blockIdx.x / 8u;
    const int64_t q0 = (int64_t)tile_id * TMW;
    // (STAT) a half tile past the tensor leaves ZERO sums: every row of `partials` is written by the launch (it was a memset in front of it)
    auto zero_sums = [&](int rb) {
        if (STAT && tid < 2 * C) partials[(((int64_t)cls * tiles_x + tile_id) * 2 + rb) * 2 * C + tid] = 0.0f;
    };
    if (q0 >= go.rows) {
        zero_sums(0);
        zero_sums(1);
        return;
    }

    const unsigned char *wcls = wt + (int64_t)dg_first_block(cls) * DG_BLOCK_BYTES;
    auto issue_block = [&](int kb) {
        const unsigned char *src = wcls + (int64_t)kb * DG_BLOCK_BYTES;
        unsigned char *dst = b_s + (kb & 1) * DG_BLOCK_BYTES;
#pragma unroll
        for (int r = 0; r * THREADS * 16 < DG_BLOCK_BYTES; ++r)
            if ((r * THREADS + wave * 64) * 16 < DG_BLOCK_BYTES)   // wave-uniform
                dma16(src + (r * THREADS + tid) * 16, lds_addr(dst + (r * THREADS + wave * 64) * 16));
    };
    issue_block(0);

    // where the tile's rows go: x row of (class, position q0 + j) relative to the tile's first x row, or -1
    int64_t tile_row0;
    {
        const int64_t ba = q0 / go.img;
        const int ipa = (int)((q0 - ba * go.img) / go.Wp);
        tile_row0 = (ba * gi.Hp + max(2 * (ipa - 1), 0)) * gi.Wp;
    }
    for (int j = tid; j < TMW; j += THREADS) {
        const uint32_t qq = (uint32_t)q0 + (uint32_t)j;
        int ro = -1;
        if (interior_row32(qq, go)) {
            uint32_t b = (uint32_t)((double)qq * go.inv_img);
            uint32_t rr = qq - b * (uint32_t)go.img;
            if (rr >= (uint32_t)go.img) { rr -= (uint32_t)go.img; ++b; }
            const uint32_t ip = __umulhi(rr, go.wp_magic), jp = rr - ip * (uint32_t)go.Wp;
            const int y = 2 * ((int)ip - 1) + py, x = 2 * ((int)jp - 1) + px;
            if (y < gi.Hp - 1 && x < gi.Wp - 1)   // (odd sizes: the last class row / column lies outside the image)
                ro = (int)(((int64_t)b * gi.Hp + y + 1) * gi.Wp + x + 1 - tile_row0);
        }
        rowoff_s[j] = ro;
    }

    // staging: rows [q0, q0 + nrows) of dout / dout_sc, 16 channels (64 bytes = 4 pieces) per row and stage
    const int64_t span_bytes = (go.rows - q0) * (DCH * 4);
    auto stage_rsrc = [&](int stage) {
        const float *t = stage < 2 ? dout : dout_sc;
        return make_rsrc(t + q0 * DCH + (stage & 1) * 16, span_bytes - (stage & 1) * 64);
    };
    auto voff_f = [&](int u) {
        const int idx = u * THREADS + tid;
        return idx < nrows * 4 ? (idx >> 2) * (DCH * 4) + (idx & 3) * 16 : -1;
    };
    u32x4 pre[DG_NPRE];
    auto put = [&](int u) {
        const int idx = u * THREADS + tid;
        if (idx < nrows * 4) {
            unsigned char *dst = a_s + (idx >> 2) * ROWB + (idx & 3) * 8;
            unsigned a1, a2, a3, b1, b2, b3;
            const float4 f = as_f4(pre[u]);
            split_pair(f.x, f.y, a1, a2, a3);
            split_pair(f.z, f.w, b1, b2, b3);
            *reinterpret_cast<u32x2 *>(dst + 0 * 32) = u32x2{a1, b1};
            *reinterpret_cast<u32x2 *>(dst + 1 * 32) = u32x2{a2, b2};
            *reinterpret_cast<u32x2 *>(dst + 2 * 32) = u32x2{a3, b3};
        }
    };
    {
        const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(0);
#pragma unroll
        for (int u = 0; u < DG_NPRE; ++u) pre[u] = buf_load16(in_r, voff_f(u));
    }
#pragma unroll
    for (int u = 0; u < DG_NPRE; ++u) put(u);

    const int m = lane & 15, kg = (lane >> 4) & 1, hi = lane >> 5;
    const unsigned char *a_lane = a_s + (wave * 32 + m) * ROWB + kg * 16;
    const unsigned char *aP = a_lane + (hi ? 32 : 0);
    const unsigned char *aQ = a_lane + (hi ? 0 : 64);
    const unsigned char *b_lane = b_s + kg * 512 + m * 16;
    const unsigned char *bU = b_lane + (hi ? 1 : 2) * PLANE_B;
    const unsigned char *bV = b_lane;
    const unsigned char *bW = b_lane + (hi ? 1 : 0) * PLANE_B;

    f32x4 acc[4][NCT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kb = 0;
#pragma unroll 1
    for (int stage = 0; stage < nstage; ++stage) {
        const int nt_s = stage < 2 ? nt_c : 1;
#pragma unroll 1
        for (int t = 0; t < nt_s; ++t, ++kb) {
            dma_wait_all();
            __syncthreads();
            if (kb + 1 < nkb) issue_block(kb + 1);
            if (t == nt_s - 1 && stage + 1 < nstage) {
                const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(stage + 1);
#pragma unroll
                for (int u = 0; u < DG_NPRE; ++u) pre[u] = buf_load16(in_r, voff_f(u));
            }
            // forward tap (dI, dJ) of the class reads z[q + dI Wp' + dJ]: its transpose reads dout[q - dI Wp' - dJ]
            int off = 0;
            if (stage < 2) {
                const int dI = cls == 0 ? ((t >> 1) ? 0 : -1) : (cls == 1 ? (t ? 0 : -1) : 0);
                const int dJ = cls == 0 ? ((t & 1) ? 0 : -1) : (cls == 2 ? (t ? 0 : -1) : 0);
                off = (-dI * go.Wp - dJ) * ROWB;
            }
            const int boff = (kb & 1) * DG_BLOCK_BYTES;
            bf16x8 ap[4], aq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int roff = ((r >> 1) * TM + (r & 1) * 16) * ROWB;
                ap[r] = *reinterpret_cast<const bf16x8 *>(aP + off + roff);
                aq[r] = *reinterpret_cast<const bf16x8 *>(aQ + off + roff);
            }
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int coff = boff + (c >> 1) * 1024 + (c & 1) * 256;
                const bf16x8 bu = *reinterpret_cast<const bf16x8 *>(bU + coff);
                const bf16x8 bw = *reinterpret_cast<const bf16x8 *>(bW + coff);
                const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(bV + coff);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bu, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[r], bw, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[r], bv, acc[r][c], 0, 0, 0);
            }
        }
        if (stage + 1 < nstage) {
            __syncthreads();  // every wave has finished reading this stage's rows
#pragma unroll
            for (int u = 0; u < DG_NPRE; ++u) put(u);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int64_t qs = q0 + rb * TM;
        if (qs >= go.rows) {
            zero_sums(rb);
            break;
        }
        __syncthreads();   // ring + rows (first round) / the previous half's output tile are free
        auto store_acc = [&](float *my) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) my[(rr * 16 + (lane >> 4) * 4 + j) * (C + 4) + c * 16 + m] = acc[rb * 2 + rr][c][j];
        };
        const B3Scatter sct{rowoff_s + rb * TM, tile_row0, gi.rows - tile_row0, ((int64_t)cls * tiles_x + tile_id) * 2 + rb};
        b3_epilogue<C, STAT, true>(store_acc, nullptr, nullptr, nullptr, dx, partials, (const unsigned char *)nullptr,
                                   reinterpret_cast<float *>(smem_b), qs, go.rows, bst, sct);
    }
}


}  // namespace

#ifdef LAD_STAMP
extern "C" int lad_debug_read_b3x_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_b3x), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int64_t lad_conv_b3_packed_weight_bytes(void) { return Ch<64>::IMG_BYTES; }
extern "C" int64_t lad_conv_b3c_packed_weight_bytes(int32_t channels) {
    return channels == 64 ? Ch<64>::IMG_BYTES : channels == 32 ? Ch<32>::IMG_BYTES : -1;
}

namespace {
int pack_b3(const float *w, int32_t mode, void *wt, int32_t channels, void *stream, const char *who) {
    using namespace lad;
    LAD_REQUIRE(w && wt && (mode == 0 || mode == 1), "%s: bad argument", who);
    LAD_REQUIRE(channels == 64 || channels == 32, "%s: 64 or 32 channels (got %d)", who, channels);
    if (channels == 64)
        hipLaunchKernelGGL(pack_b3_kernel<64>, dim3(216), dim3(256), 0, (hipStream_t)stream, w, (unsigned short *)wt, mode);
    else
        hipLaunchKernelGGL(pack_b3_kernel<32>, dim3(54), dim3(256), 0, (hipStream_t)stream, w, (unsigned short *)wt, mode);
    return check_launch("pack_b3_kernel");
}
}  // namespace

extern "C" int lad_conv_b3_pack_weights(const float *w, int32_t mode, void *wt, void *stream) {
    return pack_b3(w, mode, wt, 64, stream, "lad_conv_b3_pack_weights");
}
extern "C" int lad_conv_b3c_pack_weights(const float *w, int32_t mode, void *wt, int32_t channels, void *stream) {
    return pack_b3(w, mode, wt, channels, stream, "lad_conv_b3c_pack_weights");
}

namespace {
template <int C, bool F32IN, bool STAT = false, bool INBN = false>
int launch_b3(const void *in, const void *wt, const float *bias, const float *addend, const uint64_t *abits, float *out,
              float *partials, int64_t batch, int32_t H, int32_t W, void *stream, const char *who, B3Stat bst = B3Stat{nullptr, nullptr, nullptr},
              const float *in_coef = nullptr) {
    using namespace lad;
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "%s: bad geometry", who);
    LAD_REQUIRE(in && wt && out, "%s: null buffer", who);
    LAD_REQUIRE(abits == nullptr || addend != nullptr, "%s: sign bits without an addend", who);
    LAD_REQUIRE(C == 64 || (abits == nullptr && bst.bits == nullptr), "%s: sign bits are kept for 64-channel activations only", who);
    LAD_REQUIRE((const void *)in != (const void *)out, "%s: the convolution cannot run in place", who);
    const Geom g = make_geom(batch, H, W);
    // offsets are relative to the workgroup's tile (64-bit tile bases); only ROW numbers must fit 32 bits (interior_row32)
    LAD_REQUIRE(g.rows < ((int64_t)1 << 31) && g.img < (1 << 20), "%s: more than 2^31 rows, or an image of more than 2^20 positions", who);
    LAD_REQUIRE(W <= 46, "%s: image too wide for the tile (W = %d)", who, W);
    const int64_t tiles = ceil_div(g.rows, TM * 2);
    const dim3 grid((unsigned)(ceil_div(tiles, 8) * 8));
    constexpr int TPC = 1, NSLOT = 2;   // one tap per ring chunk, two slots (the other combinations: tools/experiments/retired/)
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_b3x_kernel<C, F32IN, STAT, INBN, TPC, NSLOT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_b3x_kernel<C, F32IN, STAT, INBN, TPC, NSLOT>), grid, dim3(THREADS), (b3x_lds_bytes<C, TPC, NSLOT>(g)),
                       (hipStream_t)stream, (const unsigned char *)in, (const unsigned char *)wt, bias, addend,
                       (const unsigned long long *)abits, out, partials, g, bst, in_coef);
    return check_launch("conv_b3x_kernel");
}
}  // namespace

extern "C" int lad_conv_b3_fwd_f32(const float *in, const void *wt, const float *bias, const float *addend, float *out,
                                   float *partials, int64_t batch, int32_t H, int32_t W, void *stream) {
    return launch_b3<64, true>(in, wt, bias, addend, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3_fwd_f32");
}

// the same for `channels` = 64 or 32 input = output channels (32: block2's stride-1 convolutions)
extern "C" int lad_conv_b3c_fwd_f32(const float *in, const void *wt, const float *bias, const float *addend, float *out,
                                    float *partials, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    if (channels == 64) return launch_b3<64, true>(in, wt, bias, addend, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3c_fwd_f32");
    if (channels == 32) return launch_b3<32, true>(in, wt, bias, addend, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3c_fwd_f32");
    return fail(LAD_ERR_INVALID, "lad_conv_b3c_fwd_f32: 64 or 32 channels (got %d)", channels);
}

// out = conv(in) + bias + addend * [addend_bits]: the data gradient of the first convolution of an identity-shortcut block,
// with the shortcut's share dy * [y > 0] formed from dy and the sign bits of y (lad_bn_act_bits).  out may be addend.
extern "C" int lad_conv_b3_fwd_f32_gated(const float *in, const void *wt, const float *bias, const float *addend,
                                         const uint64_t *addend_bits, float *out, float *partials, int64_t batch, int32_t H,
                                         int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(addend && addend_bits, "lad_conv_b3_fwd_f32_gated: null addend / sign bits");
    return launch_b3<64, true>(in, wt, bias, addend, addend_bits, out, partials, batch, H, W, stream, "lad_conv_b3_fwd_f32_gated");
}

// Data gradient (no bias) fused with the first pass of the BatchNorm backward that consumes it: stat_partials receives, per
// 128-row tile, (sum dz, sum dz * xhat) of that BatchNorm (input bn_x, coefficients bn_coef; ReLU decisions from bn_bits
// or, bn_bits = NULL, recomputed from bn_x) -- hand them to lad_bn_bwd / lad_bn_bwd_bits as pre_partials.  addend /
// addend_bits as in lad_conv_b3_fwd_f32_gated (both may be NULL).
extern "C" int lad_conv_b3_dgrad_bnstat(const float *in, const void *wt, const float *addend, const uint64_t *addend_bits,
                                        float *out, float *stat_partials, const float *bn_x, const uint64_t *bn_bits,
                                        const float *bn_coef, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stat_partials && bn_x && bn_coef, "lad_conv_b3_dgrad_bnstat: null buffer");
    return launch_b3<64, true, true>(in, wt, nullptr, addend, addend_bits, out, stat_partials, batch, H, W, stream,
                                     "lad_conv_b3_dgrad_bnstat", B3Stat{bn_x, (const unsigned long long *)bn_bits, bn_coef});
}

// the same for 64 or 32 channels, ReLU decisions recomputed from bn_x (no sign bits), optional plain addend
extern "C" int lad_conv_b3c_dgrad_bnstat(const float *in, const void *wt, const float *addend, float *out, float *stat_partials,
                                         const float *bn_x, const float *bn_coef, int64_t batch, int32_t H, int32_t W,
                                         int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stat_partials && bn_x && bn_coef, "lad_conv_b3c_dgrad_bnstat: null buffer");
    const B3Stat bst{bn_x, nullptr, bn_coef};
    if (channels == 64) return launch_b3<64, true, true>(in, wt, nullptr, addend, nullptr, out, stat_partials, batch, H, W, stream, "lad_conv_b3c_dgrad_bnstat", bst);
    if (channels == 32) return launch_b3<32, true, true>(in, wt, nullptr, addend, nullptr, out, stat_partials, batch, H, W, stream, "lad_conv_b3c_dgrad_bnstat", bst);
    return fail(LAD_ERR_INVALID, "lad_conv_b3c_dgrad_bnstat: 64 or 32 channels (got %d)", channels);
}

// Forward convolution whose input is relu(BatchNorm(in)) with in_coef = that BatchNorm's coefficients (lad_bn_finalize):
// the second convolution of a residual block reading the first one's raw output (models.py:110-112 in one launch; the
// activation between them is never written).  Bit-identical to lad_bn_act followed by lad_conv_b3_fwd_f32.
extern "C" int lad_conv_b3_fwd_f32_bnrelu(const float *in, const float *in_coef, const void *wt, const float *bias, float *out,
                                          float *partials, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in_coef, "lad_conv_b3_fwd_f32_bnrelu: null coefficients");
    return launch_b3<64, true, false, true>(in, wt, bias, nullptr, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3_fwd_f32_bnrelu",
                                            B3Stat{nullptr, nullptr, nullptr}, in_coef);
}

// the same for 64 or 32 channels
extern "C" int lad_conv_b3c_fwd_f32_bnrelu(const float *in, const float *in_coef, const void *wt, const float *bias, float *out,
                                           float *partials, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in_coef, "lad_conv_b3c_fwd_f32_bnrelu: null coefficients");
    const B3Stat none{nullptr, nullptr, nullptr};
    if (channels == 64)
        return launch_b3<64, true, false, true>(in, wt, bias, nullptr, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3c_fwd_f32_bnrelu", none, in_coef);
    if (channels == 32)
        return launch_b3<32, true, false, true>(in, wt, bias, nullptr, nullptr, out, partials, batch, H, W, stream, "lad_conv_b3c_fwd_f32_bnrelu", none, in_coef);
    return fail(LAD_ERR_INVALID, "lad_conv_b3c_fwd_f32_bnrelu: 64 or 32 channels (got %d)", channels);
}

// ---- stride-2 transition 64 -> 32 on the split-operand path (conv_s2b3_kernel) ----------------------------------------------
extern "C" int64_t lad_conv_s2b3_packed_weight_bytes(void) { return s2b3::IMG_BYTES; }

extern "C" int lad_conv_s2b3_pack_weights(const float *w, const float *w_sc, void *wt, void *stream) {
    using namespace lad;
    LAD_REQUIRE(w && w_sc && wt, "lad_conv_s2b3_pack_weights: null buffer");
    hipLaunchKernelGGL(pack_s2b3_fwd_kernel, dim3(120), dim3(256), 0, (hipStream_t)stream, w, w_sc, (unsigned short *)wt);
    return check_launch("pack_s2b3_fwd_kernel");
}

extern "C" int lad_conv_s2b3_pack_weights_pair(const float *w, const float *w_sc, void *wt_fwd, void *wt_dgrad, void *stream) {
    using namespace lad;
    LAD_REQUIRE(w && w_sc && wt_fwd && wt_dgrad, "lad_conv_s2b3_pack_weights_pair: null buffer");
    hipLaunchKernelGGL(pack_s2b3_pair_kernel, dim3(240), dim3(256), 0, (hipStream_t)stream, w, w_sc, (unsigned short *)wt_fwd, (unsigned short *)wt_dgrad);
    return check_launch("pack_s2b3_pair_kernel");
}

extern "C" int lad_conv_s2b3_fwd(const float *in, const void *wt, const float *bias, float *out, float *stat_partials, float *out_sc,
                                 float *stat_partials_sc, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && out && out_sc, "lad_conv_s2b3_fwd: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 2 && W >= 2, "lad_conv_s2b3_fwd: bad geometry");
    const Geom gi = make_geom(batch, H, W), go = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_REQUIRE(go.Wp <= 46, "lad_conv_s2b3_fwd: output too wide for the tile (W = %d)", W);
    LAD_REQUIRE(gi.rows < ((int64_t)1 << 31) && gi.img < (1 << 20), "lad_conv_s2b3_fwd: more than 2^31 rows, or an image of more than 2^20 positions");
    const int nrows = s2b3::TMW + go.Wp + 1;
    const size_t lds = std::max<size_t>(2 * s2b3::SLOT_BYTES + (size_t)nrows * s2b3::ROWB, (size_t)TM * (s2b3::COUT + 4) * 4) + s2b3::TMW;
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_s2b3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    const int64_t tiles = ceil_div(go.rows, s2b3::TMW);
    hipLaunchKernelGGL(conv_s2b3_kernel, dim3((unsigned)(ceil_div(tiles, 8) * 8)), dim3(THREADS), lds, (hipStream_t)stream, in,
                       (const unsigned char *)wt, bias, out, stat_partials, out_sc, stat_partials_sc, gi, go);
    return check_launch("conv_s2b3_kernel");
}

extern "C" int64_t lad_conv_s2b3_dgrad_packed_weight_bytes(void) { return s2b3::DG_IMG_BYTES; }

extern "C" int lad_conv_s2b3_dgrad_pack_weights(const float *w, const float *w_sc, void *wt, void *stream) {
    using namespace lad;
    LAD_REQUIRE(w && w_sc && wt, "lad_conv_s2b3_dgrad_pack_weights: null buffer");
    hipLaunchKernelGGL(pack_s2b3_dgrad_kernel, dim3(120), dim3(256), 0, (hipStream_t)stream, w, w_sc, (unsigned short *)wt);
    return check_launch("pack_s2b3_dgrad_kernel");
}

// partial rows (of [2][64] floats) lad_conv_s2b3_dgrad writes when it carries the BatchNorm sums: 4 classes x 128-row tiles
extern "C" int64_t lad_conv_s2b3_dgrad_partials(int64_t batch, int32_t H, int32_t W) {
    if (batch < 1 || H < 2 || W < 2) return -1;
    const lad::Geom go = lad::make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    return 4 * lad::ceil_div(lad::ceil_div(go.rows, s2b3::TMW), 8) * 8 * 2;
}

extern "C" int lad_conv_s2b3_dgrad(const float *dout, const float *dout_sc, const void *wt, float *dx, float *stat_partials,
                                   const float *bn_x, const uint64_t *bn_bits, const float *bn_coef, int64_t batch, int32_t H, int32_t W,
                                   void *stream) {
    using namespace lad;
    LAD_REQUIRE(dout && dout_sc && wt && dx, "lad_conv_s2b3_dgrad: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 2 && W >= 2, "lad_conv_s2b3_dgrad: bad geometry");
    const bool stat = stat_partials != nullptr;
    LAD_REQUIRE(!stat || (bn_x && bn_bits && bn_coef), "lad_conv_s2b3_dgrad: the BatchNorm sums need bn_x, bn_bits and bn_coef");
    const Geom gi = make_geom(batch, H, W), go = make_geom(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_REQUIRE(go.Wp <= 46, "lad_conv_s2b3_dgrad: output too wide for the tile (W = %d)", W);
    LAD_REQUIRE(gi.rows < ((int64_t)1 << 31) && gi.img < (1 << 20), "lad_conv_s2b3_dgrad: more than 2^31 rows, or an image of more than 2^20 positions");
    const int nrows = s2b3::TMW + go.Wp + 1;
    const size_t lds = std::max<size_t>(2 * s2b3::DG_BLOCK_BYTES + (size_t)nrows * s2b3::ROWB, (size_t)TM * (s2b3::CIN + 4) * 4) + s2b3::TMW * sizeof(int);
    const int64_t tiles_x = ceil_div(ceil_div(go.rows, s2b3::TMW), 8) * 8;
    const dim3 grid((unsigned)tiles_x, 4);
    const B3Stat bst{bn_x, (const unsigned long long *)bn_bits, bn_coef};
    if (stat) {   // (workgroups past the tensor -- the grid is rounded up to a multiple of 8 -- write zero sums)
        hipLaunchKernelGGL(dgrad_s2b3_kernel<true>, grid, dim3(THREADS), lds, (hipStream_t)stream, dout, dout_sc, (const unsigned char *)wt, dx,
                           stat_partials, gi, go, bst, tiles_x);
    } else {
        hipLaunchKernelGGL(dgrad_s2b3_kernel<false>, grid, dim3(THREADS), lds, (hipStream_t)stream, dout, dout_sc, (const unsigned char *)wt, dx,
                           (float *)nullptr, gi, go, bst, tiles_x);
    }
    return check_launch("dgrad_s2b3_kernel");
}
