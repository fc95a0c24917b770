// 64 -> 64 (and 32 -> 32) 3x3 stride-1 convolution on the 16-bit matrix cores with TWO f16 planes per operand ("f16 x 2").
//
// Same layers, same tensors, same epilogue as conv_b3.hip (nn.Conv2d(64, 64, 3, padding=1) of block1, models.py:86-95,110-115,
// forward and data gradient) -- at HALF the matrix instructions: three plane products per fp32-equivalent product instead of
// six.  VERDICT r3 item 1 ("cut the executed MFMA work of block1: the part is power-bound, not schedule-bound") asked for a
// Winograd F(2x2,3x3) prototype; profiles/r04_conv_arith_numerics.log prices both against float64 at K = 576: this form has the
// operand error of that Winograd form (7.5e-8 vs 7.0e-8 of the rms output; the fp32 accumulation error all forms share is
// 2.2e-7) with 2x instead of 2.25x fewer MFMAs -- and none of Winograd's structural costs here (16/9 more weight bytes =
// 393 KB of split planes against 160 KB of LDS, 4x the accumulators per output, 16 transformed values to split per 4 inputs).
//
// Arithmetic.  An operand tile is held as x * 2^k = h1 + h2 + e with h1 = f16(x 2^k), h2 = f16(x 2^k - h1) (both round to
// nearest even; the subtraction is exact in f32) and k chosen PER STAGED TILE so that its largest magnitude lands in
// [2^14, 2^15): |e| <= 2^-22 |x 2^k| (measured 2^-23; 2^-24.6 rms -- an fp32 rounding is 2^-24) for every element within 2^17 of
// the tile's maximum, and <= 2^-25 absolute (2^-40 of the tile's maximum) below that, where h2 leaves the normal range of
// f16.  A product is a1 b2 + a2 b1 + a1 b1 (the dropped a2 b2 is below 2^-22 of it): three v_mfma_f32_16x16x32_f16 with f32
// accumulation, each plane product exact in f32.  The power-of-two scales are exact to apply and to undo:
//   * activations: one scale per (workgroup tile, 32-channel stage), from the maximum the workgroup finds while it stages the
//     rows (a block floating point; nothing outside the kernel knows about it);
//   * weights: one scale per 32-channel stage of the packed image, found by the pack kernel and carried in the image's tail;
//   * the f32 accumulators are re-based when the stage changes (ldexp by the difference, exact) and scaled back in the
//     epilogue; a stage may raise the exponent by at most 8 over its predecessor (what it then loses lies below 2^-40 of
//     the sum so far), so nothing can overflow.
//
// Structure = conv_b3x_kernel: a workgroup owns RB x 128 consecutive output rows and all output channels; the fp32 input rows
// (+ halo) are staged 32 channels at a time (one whole 128-byte line per row), BatchNorm + ReLU of the previous layer applied
// on the way when asked (INBN), split into the two planes: LDS rows of 128 bytes [h1: 32 f16][h2: 32 f16] with the 16-byte
// slots XOR-swizzled by (row & 6) -- ds_read_b128 of the A fragments of v_mfma_f32_16x16x32_f16 (16 rows x 4 k-octets) is then
// conflict-free for every tap shift (brute-forced over all row alignments and both planes).  The packed weights stream tap by
// tap (8 KB = [plane][column tile][k octet][16][8 f16]) through an LDS-DMA ring awaited with counted vmcnt.
#include "lad_common.h"
#include "lad_device.h"
#include "lad_b3_tile.h"

#include <algorithm>
#include <cstdlib>

namespace {
using namespace lad;
using namespace lad::b3t;

constexpr int KC = 32;         // input channels per stage
constexpr int ROWB = 128;      // bytes per staged row: two planes of 32 f16
constexpr int FPIECES = KC * 4 / 16;   // 16-byte pieces (4 fp32 channels) per row and stage

template <int C>
struct H2 {
    static constexpr int NSTAGE = C / KC;
    static constexpr int NCT = C / 16;                       // 16-column output tiles
    static constexpr int PLANE_B = NCT * 1024;               // one plane of a tap: [column tile][k octet 4][n 16][8 f16]
    static constexpr int TAP_BYTES = 2 * PLANE_B;
    static constexpr int IMG_BYTES = TAPS * NSTAGE * TAP_BYTES;   // [tap][stage][plane][...]
    static constexpr int TAIL_BYTES = 16;                    // int32 scale exponent per stage
};

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h16x2));   // round to nearest even
}
// two (scaled) fp32 values -> their two f16 planes, packed pairs
__device__ __forceinline__ void split2_pair(float a, float b, unsigned &p1, unsigned &p2) {
    const f32x2 v = {a, b};
    const h16x2 h = __builtin_convertvector(v, h16x2);
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = pack_f16(a - (float)h.x, b - (float)h.y);   // (the differences are exact in f32)
}

// scale exponent for a tile / an image stage whose largest magnitude is `amax` (>= 0, finite): amax * 2^k in [2^14, 2^15)
__host__ __device__ __forceinline__ int scale_exp(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
    const int k = 141 - e;
    return k > 100 ? 100 : k;       // (zero / denormal maxima: 2^100 keeps every product finite)
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }   // -126 <= k <= 127

__device__ __forceinline__ float wave_max64(float v) { return wave_max64_dpp(v); }

// ---- weights: (cout, cin, 3, 3) fp32 -> [tap][stage][plane][column tile][k octet][n 16][8 f16] + the stages' exponents ------
// mode 0: forward, GEMM K = cin, N = cout.  mode 1: data gradient, K = cout, N = cin, taps flipped.  One workgroup per image.
struct PackRec {
    const float *w;
    unsigned char *wt;
    int mode;
    int channels;   // read by the mixed launch only (lad_conv_h2_pack_weights_multi with channels = 0)
};
constexpr int PACK_SPLIT = 16;   // workgroups per image: each finds the stages' maxima itself (36,864 weights: 9 float4 per thread), then writes 1/16 of the planes
template <int C>
__device__ __forceinline__ void pack_h2_image(const PackRec &rec, int part) {
    using K = H2<C>;
    __shared__ float smax[K::NSTAGE][16];
    __shared__ int kexp[K::NSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pass 1: the largest magnitude per K stage (every workgroup of the image computes the same numbers)
    float m[K::NSTAGE];
#pragma unroll
    for (int s = 0; s < K::NSTAGE; ++s) m[s] = 0.f;
    static_assert((C * TAPS) % 4 == 0, "float4 groups stay inside one (co, ci range)");
    for (int idx4 = tid; idx4 < C * C * TAPS / 4; idx4 += 1024) {
        const float4 v = reinterpret_cast<const float4 *>(rec.w)[idx4];
        const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = idx4 * 4 + j;
            const int co = idx / (C * TAPS), ci = (idx / TAPS) % C;
            const int k = rec.mode == 0 ? ci : co;
#pragma unroll
            for (int s = 0; s < K::NSTAGE; ++s) m[s] = (k / KC == s) ? fmaxf(m[s], a[j]) : m[s];
        }
    }
#pragma unroll
    for (int s = 0; s < K::NSTAGE; ++s) {
        const float v = wave_max64(m[s]);
        if (lane == 0) smax[s][wave] = v;
    }
    __syncthreads();
    if (tid < K::NSTAGE) {
        float v = 0.f;
        for (int w = 0; w < 16; ++w) v = fmaxf(v, smax[tid][w]);
        kexp[tid] = scale_exp(v);
        if (part == 0) reinterpret_cast<int *>(rec.wt + K::IMG_BYTES)[tid] = kexp[tid];
    }
    __syncthreads();
    // pass 2: this workgroup's share of the planes
    unsigned short *out = reinterpret_cast<unsigned short *>(rec.wt);
    constexpr int SHARE = K::IMG_BYTES / 2 / PACK_SPLIT;
    for (int idx = part * SHARE + tid; idx < (part + 1) * SHARE; idx += 1024) {
        int t = idx;
        const int e = t & 7; t >>= 3;
        const int n = t & 15; t >>= 4;
        const int kq = t & 3; t >>= 2;
        const int ct = t % K::NCT; t /= K::NCT;
        const int plane = t & 1; t >>= 1;
        const int stage = t % K::NSTAGE;
        const int tap = t / K::NSTAGE;
        const int k = stage * KC + kq * 8 + e;
        const int nn = ct * 16 + n;
        const int co = rec.mode == 0 ? nn : k, ci = rec.mode == 0 ? k : nn;
        const int src_tap = rec.mode == 0 ? tap : TAPS - 1 - tap;
        const float v = rec.w[((int64_t)co * C + ci) * TAPS + src_tap] * pow2f(kexp[stage]);
        const _Float16 h1 = (_Float16)v;
        const _Float16 h2 = (_Float16)(v - (float)h1);
        out[idx] = __builtin_bit_cast(unsigned short, plane == 0 ? h1 : h2);
    }
}
template <int C>   // C > 0: every record is a C-channel layer; C = 0: a record names its own channel count (64 or 32)
__global__ __launch_bounds__(1024) void pack_h2_kernel(const PackRec *__restrict__ recs) {
    const PackRec rec = recs[blockIdx.x / PACK_SPLIT];
    const int part = blockIdx.x % PACK_SPLIT;
    if (C == 64 || (C == 0 && rec.channels == 64)) pack_h2_image<64>(rec, part);
    else pack_h2_image<32>(rec, part);
}

#ifdef LAD_STAMP
__device__ unsigned long long lad_dbg_h2[16 * 16384];
#define LAD_H2_STAMP(k) \
    if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_h2[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LAD_H2_STAMP(k)
#endif

// ---- the convolution ----------------------------------------------------------------------------------------------------------
// `in`: fp32 rows [rows][C] (layout: lad_device.h).  INBN: `in` is the previous convolution's raw output and in_coef that
// BatchNorm's float[6][C] (scale, shift, ...): relu(in * scale + shift) on interior rows, 0 on border rows, is formed while a
// stage is staged -- the same fmaf / max as bn_act_kernel.  STAT / addend / abits / partials: b3_epilogue (lad_b3_tile.h).
// RB = 2: 256-row tiles, two workgroups per CU (every launch of two dispatch rounds or more); RB = 1: 128-row tiles, three per CU
// (small launches).  Every other structure that was built and measured -- 384-row tiles, four ring slots, persistent workgroups,
// no weight ring, eight waves per workgroup, and the diagnostic builds without MFMAs / loads / epilogue / split -- lives in
// tools/experiments/retired/conv_h2_variants.hip (tools/exp_h2.sh builds it into a library of its own); all of them run in
// 0.49-0.57 ms per launch at batch 512 (profiles/r04_conv_h2_experiments.log, profiles/r05_conv_h2_8waves.log).
constexpr int NSLOT = 3;   // ring slots: the LDS-DMA runs two taps ahead of the MFMAs
template <int C, int RB, bool STAT, bool INBN>
__global__ __launch_bounds__(THREADS, RB == 1 ? 3 : 2) void conv_h2_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wt,
                                                             const float *__restrict__ bias, const float *addend,
                                                             const unsigned long long *__restrict__ abits, float *out,
                                                             float *__restrict__ partials, Geom g, B3Stat bst,
                                                             const float *__restrict__ in_coef) {
    using K = H2<C>;
    constexpr int NSTAGE = K::NSTAGE, NCT = K::NCT, PLANE_B = K::PLANE_B, TAP_BYTES = K::TAP_BYTES;
    constexpr int TMW = TM * RB, NRT = 2 * RB;                       // output rows per workgroup; 16-row tiles per wavefront
    constexpr int NPRE = ((TMW + 2 * 47) * FPIECES + THREADS - 1) / THREADS;   // registers for one stage at the widest image (W = 46)
    constexpr int CPS = TAPS;                                          // one tap per ring chunk
    constexpr int KP = (CPS - NSLOT) < CPS / 2 ? (CPS - NSLOT) : CPS / 2;
    static_assert(CPS % NSLOT == 0 && NPRE <= 32, "every stage starts in ring slot 0; staging geometry");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = g.Wp + 1;
    const int nrows = TMW + 2 * halo;
    const int main_bytes = max(NSLOT * TAP_BYTES + nrows * ROWB, TM * (C + 4) * 4);
    unsigned char *b_s = smem_b;                          // [NSLOT][TAP_BYTES]
    unsigned char *a_s = b_s + NSLOT * TAP_BYTES;         // [nrows][ROWB]
    unsigned char *mask_s = smem_b + main_bytes;          // [TMW]
    float *smax = reinterpret_cast<float *>(mask_s + TMW);   // [4]: the waves' maxima of the stage being staged
    // XCD-aware tile order (conv_b3_kernel): XCD x takes a contiguous range of tiles
    const unsigned per_x = (gridDim.x + 7u) / 8u;
    const unsigned tile_id = (blockIdx.x % 8u) * per_x + blockIdx.x / 8u;
    const int64_t q0 = (int64_t)tile_id * TMW;
    if (q0 >= g.rows) return;

    LAD_H2_STAMP(0)
    const int nw_tap = dma_per_tap<TAP_BYTES>(wave);
    auto issue_tap = [&](int tap, int stage, int slot) {
        const unsigned char *src = wt + (int64_t)(tap * NSTAGE + stage) * TAP_BYTES;
        unsigned char *dst = b_s + slot * TAP_BYTES;
#pragma unroll
        for (int r = 0; r * THREADS * 16 < TAP_BYTES; ++r)
            if ((r * THREADS + wave * 64) * 16 < TAP_BYTES)   // wave-uniform
                dma16(src + (r * THREADS + tid) * 16, lds_addr(dst + (r * THREADS + wave * 64) * 16));
    };
    const int *wexp = reinterpret_cast<const int *>(wt + K::IMG_BYTES);

    // ---- staging: piece idx = u * THREADS + tid = 16 bytes (4 channels) of row idx >> 3 of the stage's 128-byte line -------------
    const int64_t start = q0 - halo;
    const int64_t first = start < 0 ? 0 : start;
    const int row_lo = (int)(first - start);
    const int64_t span_rows = min(g.rows - first, (int64_t)(nrows - row_lo));
    auto voff = [&](int u) {
        const int idx = u * THREADS + tid;
        return idx < nrows * FPIECES ? ((idx >> 3) - row_lo) * (C * 4) + (idx & 7) * 16 : -1;
    };
    auto stage_rsrc = [&](int stage) {
        return make_rsrc(reinterpret_cast<const unsigned char *>(in) + first * (C * 4) + stage * (KC * 4), span_rows * (C * 4) - stage * (KC * 4));
    };
    unsigned keep_bits = 0;
    u32x4 pre[NPRE];
    // the loaded pieces -> the values the convolution sees (INBN) and their largest magnitude over this thread
    auto activate = [&](int stage) {
        float m = 0.f;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
        if (INBN) {
            sc = *reinterpret_cast<const f32x4 *>(in_coef + stage * KC + (tid & 7) * 4);
            sh = *reinterpret_cast<const f32x4 *>(in_coef + C + stage * KC + (tid & 7) * 4);
        }
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            float4 f = as_f4(pre[u]);
            if (INBN) {
                const bool keep = (keep_bits >> u) & 1u;
                f.x = keep ? fmaxf(fmaf(f.x, sc.x, sh.x), 0.f) : 0.f;
                f.y = keep ? fmaxf(fmaf(f.y, sc.y, sh.y), 0.f) : 0.f;
                f.z = keep ? fmaxf(fmaf(f.z, sc.z, sh.z), 0.f) : 0.f;
                f.w = keep ? fmaxf(fmaf(f.w, sc.w, sh.w), 0.f) : 0.f;
                pre[u] = as_u4(f);
            }
            m = fmaxf(fmaxf(m, fmaxf(fabsf(f.x), fabsf(f.y))), fmaxf(fabsf(f.z), fabsf(f.w)));
        }
        m = wave_max64(m);
        if (lane == 0) smax[wave] = m;
    };
    auto tile_exp = [&]() { return scale_exp(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))); };
    // registers -> LDS: 4 channels become 8 bytes in each plane; slot (4 plane + k octet) ^ (row & 6) of the row
    auto put_all = [&](float scl) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = u * THREADS + tid;
            if (idx < nrows * FPIECES) {
                const int row = idx >> 3, piece = idx & 7;
                const float4 f = as_f4(pre[u]);
                unsigned a1, a2, b1, b2;
                split2_pair(f.x * scl, f.y * scl, a1, a2);
                split2_pair(f.z * scl, f.w * scl, b1, b2);
                // (offsets, not pointer bits: an XOR on the pointer would lose the LDS address space and turn the store into a flat one)
                const unsigned off = (unsigned)row * ROWB + ((((piece >> 1) ^ (row & 6)) << 4) | ((piece & 1) << 3));
                *reinterpret_cast<u32x2 *>(a_s + off) = u32x2{a1, b1};
                *reinterpret_cast<u32x2 *>(a_s + (off ^ 64u)) = u32x2{a2, b2};
            }
        }
    };

    // ---- prologue: the first stage's rows are requested first (HBM), then the first NSLOT - 1 taps of weights (L2) ---------------
    {
        const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(0);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = buf_load16(in_r, voff(u));
    }
#pragma unroll
    for (int k = 0; k < NSLOT - 1; ++k) issue_tap(k, 0, k);
    for (int j = tid; j < TMW; j += THREADS) mask_s[j] = interior_row32((uint32_t)q0 + (uint32_t)j, g) ? 1 : 0;
    if (INBN) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u)
            keep_bits |= (interior_row32((uint32_t)(start + ((u * THREADS + tid) >> 3)), g) ? 1u : 0u) << u;
    }
    LAD_H2_STAMP(1)
    activate(0);
    __syncthreads();
    int ktot = tile_exp() + wexp[0];   // exponent of the accumulators: acc = 2^ktot x (sum so far)
    put_all(pow2f(ktot - wexp[0]));
    LAD_H2_STAMP(2)

    // ---- fragments ---------------------------------------------------------------------------------------------------------------
    // A (plane p): lane (m = lane & 15, kq = lane >> 4) reads 16 bytes of row r at slot (4 p + kq) ^ (r & 6); row tile t covers rows
    //    (t >> 1) * 128 + wave * 32 + (t & 1) * 16 + m of the tile (multiples of 16 apart: the swizzle term is the same for all).
    // B (plane p): 16 bytes at p * PLANE_B + column tile * 1024 + kq * 256 + m * 16.
    const int m = lane & 15, kq = lane >> 4;
    const int rl = wave * 32 + m + halo - 1;                           // (- 1: tap column offsets 0, 1, 2)
    const unsigned char *b_lane = b_s + kq * 256 + m * 16;

    f32x4 acc[NRT][NCT];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int stage = 0; stage < NSTAGE; ++stage) {
        const bool last = stage + 1 == NSTAGE;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kc = tap;
            const int slot = kc % NSLOT;
            {
                // tap kc has landed (this wave's part), then everybody's; the slot of tap kc - 1 is free
                constexpr int YOUNGER = NSLOT - 2;
                if (!last) {
                    if (kc > KP && kc <= KP + NSLOT - 1) wait_dma<YOUNGER, NPRE>(nw_tap);   // + the requested rows
                    else wait_dma<YOUNGER, 0>(nw_tap);
                } else {
                    if (CPS - 1 - kc >= YOUNGER) wait_dma<YOUNGER, 0>(nw_tap);
                    else wait_dma<0, 0>(nw_tap);
                }
                __syncthreads();
                const int kn = kc + NSLOT - 1;
                if (kn < CPS) issue_tap(kn, stage, kn % NSLOT);
                else if (!last) issue_tap(kn - CPS, stage + 1, kn % NSLOT);
                if (kc == KP && !last) {
                    const __amdgpu_buffer_rsrc_t in_r = stage_rsrc(stage + 1);
#pragma unroll
                    for (int u = 0; u < NPRE; ++u) pre[u] = buf_load16(in_r, voff(u));
                }
            }
            const int rt = rl + (tap / 3 - 1) * g.Wp + (tap % 3);
            const unsigned a1o = (unsigned)rt * ROWB + ((unsigned)(kq ^ (rt & 6)) << 4);
            const unsigned a2o = a1o ^ 64u;
            const int boff = slot * TAP_BYTES;
            f16x8 a1[NRT], a2[NRT];
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                const int roff = ((r >> 1) * TM + (r & 1) * 16) * ROWB;
                a1[r] = *reinterpret_cast<const f16x8 *>(a_s + a1o + roff);
                a2[r] = *reinterpret_cast<const f16x8 *>(a_s + a2o + roff);
            }
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const f16x8 b1 = *reinterpret_cast<const f16x8 *>(b_lane + boff + c * 1024);
                const f16x8 b2 = *reinterpret_cast<const f16x8 *>(b_lane + boff + PLANE_B + c * 1024);
                // smallest terms first: a1 b2, a2 b1, then a1 b1
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[r], b2, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[r], b1, acc[r][c], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[r], b1, acc[r][c], 0, 0, 0);
            }
        }
#ifdef LAD_STAMP
        if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_h2[blockIdx.x * 16 + 3 + 2 * stage] = __builtin_amdgcn_s_memtime();
#endif
        if (!last) {
            activate(stage + 1);
            __syncthreads();  // every wave has finished reading this stage's rows; the next stage's maxima are in smax
            const int kw = wexp[stage + 1];
            const int kn = min(tile_exp() + kw, ktot + 8);
            const int d = kn - ktot;
            if (d != 0) {
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[r][c][j] = __builtin_ldexpf(acc[r][c][j], d);
            }
            ktot = kn;
            const int ka = kn - kw;                      // <= the exponent the stage's maximum allows: no f16 overflow
            put_all(ka >= -126 ? pow2f(ka) : 0.f);       // (more than 2^126 below its predecessor: the stage contributes nothing)
        }
#ifdef LAD_STAMP
        if (threadIdx.x == 0 && blockIdx.x < 16384) lad_dbg_h2[blockIdx.x * 16 + 4 + 2 * stage] = __builtin_amdgcn_s_memtime();
#endif
    }
    __syncthreads();  // every wave is out of the MFMA loop: ring + input rows become the output tile
    LAD_H2_STAMP(11)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
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
                    for (int j = 0; j < 4; ++j)
                        my[(rr * 16 + (lane >> 4) * 4 + j) * (C + 4) + c * 16 + m] = __builtin_ldexpf(acc[rb * 2 + rr][c][j], -ktot);
        };
        b3_epilogue<C, STAT>(store_acc, bias, addend, abits, out, partials, mask_s + rb * TM, reinterpret_cast<float *>(smem_b), qs, g.rows, bst);
    }
    LAD_H2_STAMP(12)
}

#ifdef LAD_H2_WS_BUILD
// The wave-specialised persistent form of this kernel (four MFMA waves + four helper waves per CU, round 5) was built, verified
// bit-identical and measured: on a par in the micro-benchmark, slower in the step.  It lives in tools/experiments/conv_h2w.inc and is
// compiled in by tools/exp_ws.sh only (profiles/r05_conv_h2w.log has the ablations that say where its time goes).
#include "../../tools/experiments/conv_h2w.inc"
#endif

template <int C, int RB>
size_t h2_lds_bytes(const Geom &g) {
    using K = H2<C>;
    const int nrows = TM * RB + 2 * (g.Wp + 1);
    const size_t main_bytes = std::max<size_t>(NSLOT * K::TAP_BYTES + (size_t)nrows * ROWB, (size_t)TM * (C + 4) * 4);
    return main_bytes + TM * RB + 16;   // + the row mask (bytes) + the waves' maxima
}

// launches of fewer than two dispatch rounds of 256-row tiles (two workgroups per CU) take 128-row tiles, three per CU
int h2_two_rounds() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? 4 * prop.multiProcessorCount : 1024;
    }
    return n;
}

#ifdef LAD_H2_WS_BUILD
bool h2_ws_from_env() {
    const char *e = getenv("LAD_H2_WS");
    return e ? e[0] != '0' : true;
}
bool g_h2_ws = h2_ws_from_env();   // (experiment builds: LAD_H2_WS=0 keeps the 256-row launches on conv_h2_kernel)

#endif

template <int C, bool STAT, bool INBN>
int launch_h2(const float *in, const float *in_coef, const void *wt, const float *bias, const float *addend, const uint64_t *abits,
              float *out, float *partials, const B3Stat &bst, int64_t batch, int32_t H, int32_t W, void *stream, const char *who) {
    const Geom g = make_geom(batch, H, W);
    LAD_REQUIRE(g.rows < ((int64_t)1 << 31) && g.img < (1 << 20), "%s: more than 2^31 rows, or an image of more than 2^20 positions", who);
    LAD_REQUIRE(W <= 46, "%s: image too wide for the tile (W = %d)", who, W);
#define LAD_H2_LAUNCH(RB)                                                                                                          \
    {                                                                                                                              \
        static lad::DeviceOnce attr_set;                                                                                              \
        if (!attr_set) {                                                                                                           \
            LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_h2_kernel<C, RB, STAT, INBN>,                                     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                             \
            attr_set = true;                                                                                                       \
        }                                                                                                                          \
        const int64_t tiles = ceil_div(g.rows, TM * RB);                                                                           \
        hipLaunchKernelGGL((conv_h2_kernel<C, RB, STAT, INBN>), dim3((unsigned)(ceil_div(tiles, 8) * 8)), dim3(THREADS),           \
                           (h2_lds_bytes<C, RB>(g)), (hipStream_t)stream, in, (const unsigned char *)wt, bias, addend,             \
                           (const unsigned long long *)abits, out, partials, g, bst, in_coef);                                    \
        return check_launch("conv_h2_kernel");                                                                                     \
    }
    if (ceil_div(g.rows, TM * 2) < h2_two_rounds()) LAD_H2_LAUNCH(1)
#ifdef LAD_H2_WS_BUILD
    if constexpr (C == 64) {
        if (g_h2_ws) {   // wave-specialised persistent workgroups, one per CU (conv_h2w_kernel)
            static bool attr_ws = false;
            if (!attr_ws) {
                LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_h2w_kernel<C, STAT, INBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_ws = true;
            }
            const int groups = h2_two_rounds() / 4 / 8 * 8;   // one workgroup per CU, a multiple of the 8 XCDs
            hipLaunchKernelGGL((conv_h2w_kernel<C, STAT, INBN>), dim3((unsigned)groups), dim3(WS_THREADS), (h2w_lds_bytes<C>(g)),
                               (hipStream_t)stream, in, (const unsigned char *)wt, bias, addend, (const unsigned long long *)abits, out,
                               partials, g, bst, in_coef);
            return check_launch("conv_h2w_kernel");
        }
    }
#endif
    LAD_H2_LAUNCH(2)
#undef LAD_H2_LAUNCH
}
}  // namespace

extern "C" int64_t lad_conv_h2_packed_weight_bytes(int32_t channels) {
    if (channels == 64) return H2<64>::IMG_BYTES + H2<64>::TAIL_BYTES;
    if (channels == 32) return H2<32>::IMG_BYTES + H2<32>::TAIL_BYTES;
    return -1;
}

// `table`: device array of `n` records {const float *w; void *wt; int32 mode; int32 channels} (24 bytes each) of `channels`-channel
// convolutions -- or, with channels = 0, of the channel count each record names (64 or 32) --; one launch packs them all.
extern "C" int lad_conv_h2_pack_weights_multi(const void *table, int32_t n, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(table && n >= 1, "lad_conv_h2_pack_weights_multi: empty table");
    static_assert(sizeof(PackRec) == 24, "record layout");
    if (channels == 64) hipLaunchKernelGGL(pack_h2_kernel<64>, dim3(n * PACK_SPLIT), dim3(1024), 0, (hipStream_t)stream, (const PackRec *)table);
    else if (channels == 32) hipLaunchKernelGGL(pack_h2_kernel<32>, dim3(n * PACK_SPLIT), dim3(1024), 0, (hipStream_t)stream, (const PackRec *)table);
    else if (channels == 0) hipLaunchKernelGGL(pack_h2_kernel<0>, dim3(n * PACK_SPLIT), dim3(1024), 0, (hipStream_t)stream, (const PackRec *)table);
    else return fail(LAD_ERR_INVALID, "lad_conv_h2_pack_weights_multi: 64 or 32 channels, or 0 = per record (got %d)", channels);
    return check_launch("pack_h2_kernel");
}

// out = conv3x3(act(in)) + bias + addend * [addend_bits], with the options of the bf16 x 3 entry points in one signature:
//   in_coef != NULL: act = relu(BatchNorm(in)) formed while staging (lad_conv_b3c_fwd_f32_bnrelu), else act = identity;
//   addend / addend_bits: lad_conv_b3_fwd_f32_gated (both optional; bits need an addend; out may be addend);
//   bn_x != NULL: `partials` receives the sums of the BatchNorm backward that consumes out (lad_conv_b3_dgrad_bnstat: bn_x,
//   bn_coef, optional bn_bits), else the (sum, sum of squares) of out per 128-row tile (or nothing when partials is NULL).
// wt: an image of lad_conv_h2_pack_weights_multi (mode 0 forward, mode 1 data gradient).
extern "C" int lad_conv_h2(const float *in, const float *in_coef, const void *wt, const float *bias, const float *addend,
                           const uint64_t *addend_bits, float *out, float *partials, const float *bn_x, const uint64_t *bn_bits,
                           const float *bn_coef, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    const char *who = "lad_conv_h2";
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "%s: bad geometry", who);
    LAD_REQUIRE(in && wt && out, "%s: null buffer", who);
    LAD_REQUIRE(addend_bits == nullptr || addend != nullptr, "%s: sign bits without an addend", who);
    LAD_REQUIRE((const void *)in != (const void *)out, "%s: the convolution cannot run in place", who);
    const bool stat = bn_x != nullptr;
    LAD_REQUIRE(!stat || (partials && bn_coef), "%s: the BatchNorm sums need partials and bn_coef", who);
    LAD_REQUIRE(stat || bn_bits == nullptr, "%s: bn_bits without bn_x", who);
    LAD_REQUIRE(!(stat && in_coef), "%s: an input BatchNorm and the backward sums are not combined", who);
    LAD_REQUIRE(channels == 64 || (addend_bits == nullptr && bn_bits == nullptr), "%s: sign bits are kept for 64-channel activations only", who);
    const B3Stat bst{bn_x, (const unsigned long long *)bn_bits, bn_coef};
    if (channels == 64) {
        if (stat) return launch_h2<64, true, false>(in, nullptr, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
        if (in_coef) return launch_h2<64, false, true>(in, in_coef, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
        return launch_h2<64, false, false>(in, nullptr, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
    }
    if (channels == 32) {
        if (stat) return launch_h2<32, true, false>(in, nullptr, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
        if (in_coef) return launch_h2<32, false, true>(in, in_coef, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
        return launch_h2<32, false, false>(in, nullptr, wt, bias, addend, addend_bits, out, partials, bst, batch, H, W, stream, who);
    }
    return fail(LAD_ERR_INVALID, "%s: 64 or 32 channels (got %d)", who, channels);
}

#ifdef LAD_STAMP
extern "C" int lad_debug_read_h2_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_h2), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
