// fp16 sliding-window inference: EVERYTHING BEHIND RESOLUTION LEVEL 2 in one launch (round 6).
//
// Replaces, per window of segment_laughter.py:90-101 (models.py:226-239 in eval mode), the launches that engine._eval_level2_shared
// issued after the shared level-2 streams and strips exist:
//     block3.0: conv1 3x3 stride 2 (32 -> 16, read through the window map) | 1x1 stride-2 shortcut | conv2 + shortcut
//     block3.1: identity block        block4.0: conv1 3x3 stride 2 + 1x1 shortcut | conv2 + shortcut        block4.1: identity block
//     AvgPool2d(4) -> BatchNorm1d -> Linear(F, 32) -> BatchNorm1d -> ReLU -> Linear(32, 1) -> sigmoid
// = seven kernels in nine launches per group of 8,192 windows, 66 GFLOP in 460 us (6 % of the f16 matrix peak: every one of them a
// launch-sized kernel that moves its small tensors through HBM / L2).  A window's level-2 image is 50 x 22 x 32 halfs = 70 KB and the ten
// weight images 42 KB: both fit a CU's LDS, so here ONE 1024-thread workgroup per CU walks windows and keeps a window on chip from the
// level-2 rows to its probability -- one float out.
//
// Bit-identical to those launches: the same v_mfma_f32_32x32x16_f16, the same order per output element (taps 0..8, 16 channels per
// k-step), the same epilogue arithmetic (conv_f16.hip: epilogue_f16 for the stride-2 layers -- incl. its `+ 0.0f` --, block_f16_small_kernel
// for the others), the intermediates rounded to half where those kernels round their outputs, pool_f16_kernel's and head_fwd_eval_kernel's
// operation order.  Matrix roles as block_f16_small_kernel: D[channel][position] = W-fragment x X-fragment (channels 16..31 of the
// 32-row tile are padding: the weight images are kept COMPACT in LDS, 16 output channels, and lanes 16..31 re-read lanes 0..15's rows).
//
// Stride 2 without gathers: the level-2 window arrives by LDS-DMA DE-INTERLEAVED into its four parity classes (space-to-depth; the
// permutation is free: a DMA lane's source address is its own) z[py][px][I][J] = x[2 I + py][2 J + px] (padded coordinates), each class an
// image of pitch W3 + 1 -- the pitch of the level-3 tensors -- so that tap (ky, kx) of output position q reads class (ky & 1, kx & 1) at row
// q + (ky >> 1)(W3 + 1) + (kx >> 1) - (W3 + 2): a stride-1 row-shifted GEMM like every other layer here.  Level 3's last layer writes its
// output in the same class form for level 4's stride-2 layer (odd sizes there: the class rows / columns that stand for the bottom / right
// border are never written and stay zero).  Rows are 64 B (32 channels) / 32 B (16 channels) with the 16-byte slots XOR-swizzled as in
// block_f16_small_kernel (conflict-free ds_read_b128 for the 16-lane groups of MI355X_MICROARCH.md).
//
// Schedule per window (barriers between steps; a step's 32-position tiles go to waves 0, 1, ...: ten at level 3, four at level 4):
//   wait for the window's DMA | b3.0 conv1 + shortcut (X) -- and wave 15 pools + classifies the PREVIOUS window meanwhile | request the
//   NEXT window's rows (X is free from here on: the DMA runs under the eight remaining layers) | b3.0 conv2 | b3.1 | b4.0 | b4.1.
#include "lad_common.h"

#include <algorithm>
#include <type_traits>

#include "lad_device.h"

namespace {
using namespace lad;

constexpr int TL_THREADS = 1024, TL_WAVES = TL_THREADS / 64;
constexpr int TL_HID = 32, TL_MAXF = 128;
constexpr float TL_BN_EPS = 1e-5f;
constexpr int TL_NCONV = 10;

struct TailConv {
    const _Float16 *wt;            // lad_f16_pack_weights image: [tap][cin / 16][2][32][8]
    const float *scale, *shift;    // the BatchNorm behind it, folded (16 values each)
};
struct TailArgs {
    const _Float16 *act;           // the level-2 strips + phase streams (engine: cat2), 32 channels
    float *probs;
    int B;                         // windows
    int band, strip_rows, img_t;   // window map (conv_f16.hip, WinMap): rows taken from the strips, rows / positions of a strip image
    long long bot_img0, stream_row0, phase_img;
    TailConv cv[TL_NCONV];         // b3.0 conv1, b3.0 shortcut, b3.0 conv2, b3.1 conv1, b3.1 conv2, b4.0 conv1, b4.0 shortcut, b4.0 conv2, b4.1 conv1, b4.1 conv2
    const float *g2, *b2, *rm2, *rv2, *lin1, *bias1, *g3, *b3, *rm3, *rv3, *lin2, *bias2;   // the classifier (head.hip, HeadArgs)
};

// geometry + LDS map, the same on the host (size check) and on the device
struct TailGeo {
    int Wp2, H3, W3, Wp3, n3, nt3, H4, W4, Wp4, n4, nt4, PH, PW, F;
    int x0[4], nX;       // first row of the level-2 parity classes in X; rows of X
    int y0[4], nY;       // the same for level 3's output (the input of level 4's stride-2 layer) in R2
    int l4b;             // first row of a level-4 tensor inside R1 / R3 (so that its zero tail is the region's)
    int w_off[TL_NCONV];
    int x_off, r1_off, r3_off, r2_off, s3_off, t_off, total;
    // tables behind t_off
    int mask3, mask4, cls3, coef, zs, zt, us, ut, w2s, b1s, pooled, hid;
};
__host__ __device__ constexpr int tl_up(int v, int a) { return (v + a - 1) / a * a; }
__host__ __device__ constexpr TailGeo tail_geo(int H2, int W2) {
    TailGeo g{};
    g.Wp2 = W2 + 1;
    g.H3 = H2 / 2; g.W3 = W2 / 2; g.Wp3 = g.W3 + 1; g.n3 = (g.H3 + 1) * g.Wp3; g.nt3 = (g.n3 + 31) / 32;
    g.H4 = (g.H3 + 1) / 2; g.W4 = (g.W3 + 1) / 2; g.Wp4 = g.W4 + 1; g.n4 = (g.H4 + 1) * g.Wp4; g.nt4 = (g.n4 + 31) / 32;
    g.PH = g.H4 / 4; g.PW = g.W4 / 4; g.F = 16 * g.PH * g.PW;
    const int c0 = (g.H3 + 1) * g.Wp3, c1 = g.H3 * g.Wp3;          // classes with py = 0 hold rows I = 0 .. H3, py = 1: I = 0 .. H3 - 1
    g.x0[0] = 0; g.x0[1] = c0; g.x0[2] = 2 * c0; g.x0[3] = 2 * c0 + c1; g.nX = 2 * c0 + 2 * c1;
    const int d0 = (g.H4 + 1) * g.Wp4, d1 = g.H4 * g.Wp4;
    g.y0[0] = 0; g.y0[1] = d0; g.y0[2] = 2 * d0; g.y0[3] = 2 * d0 + d1; g.nY = 2 * d0 + 2 * d1;
    g.l4b = g.n3 - g.n4;
    int o = 0;
    for (int k = 0; k < TL_NCONV; ++k) {
        const int cin = k <= 1 ? 32 : 16, taps = (k == 1 || k == 6) ? 1 : 9;
        g.w_off[k] = o;
        o += taps * (cin / 16) * 512;
    }
    g.x_off = tl_up(o, 256);
    g.r1_off = tl_up(g.x_off + g.nX * 64, 256);
    const int r_bytes = tl_up((g.n3 + g.Wp3 + 1) * 32, 256);
    g.r3_off = g.r1_off + r_bytes;
    g.r2_off = g.r3_off + r_bytes;
    g.s3_off = tl_up(g.r2_off + (g.nY + 32 + g.Wp4 + 1) * 32, 256);   // (+ what the dropped lanes of the last level-4 tile read below the classes)
    g.t_off = tl_up(g.s3_off + g.nt4 * 32 * 32, 256);
    o = g.t_off;
    g.mask3 = o; o += g.nt3 * 32;
    g.mask4 = o; o += g.nt4 * 32;
    o = tl_up(o, 4);
    g.cls3 = o; o += g.nt3 * 32 * 2;
    o = tl_up(o, 16);
    g.coef = o; o += TL_NCONV * 32 * 4;
    g.zs = o; o += g.F * 4;
    g.zt = o; o += g.F * 4;
    g.us = o; o += TL_HID * 4;
    g.ut = o; o += TL_HID * 4;
    g.w2s = o; o += TL_HID * 4;
    g.b1s = o; o += TL_HID * 4;
    g.pooled = o; o += g.F * 4;
    g.hid = o; o += TL_HID * 4;
    g.total = tl_up(o, 16);
    return g;
}

// byte offset of 16-byte slot `slot` of row `row` of a tensor with RB-byte rows (64: 32 channels, 32: 16 channels)
template <int RB>
__device__ __forceinline__ int tl_off(int row, int slot) {
    return row * RB + ((slot ^ (RB == 64 ? (row >> 2) & 3 : (row >> 3) & 1)) << 4);
}

__device__ __forceinline__ float tl_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }   // (head.hip: sigmoidf)

typedef _Float16 tf16x2 __attribute__((ext_vector_type(2)));

// H2, W2: rows / columns of a window at level 2 -- compile-time, so that every LDS offset and tap shift is an immediate (with the geometry
// at run time the kernel needed 106 SGPRs + 65 spilled and spilled 40 VGPRs at the 128 a 1024-thread workgroup gets)
template <int H2, int W2>
__global__ __launch_bounds__(TL_THREADS, 1) void tail_f16_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    constexpr TailGeo g = tail_geo(H2, W2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int i = lane & 31, h = lane >> 5;

    // ---- once per workgroup: weights (compact), zeroed tensors, tables --------------------------------------------------------------
#pragma unroll   // (constant k: a run-time index into the kernel arguments would copy them to scratch)
    for (int k = 0; k < TL_NCONV; ++k) {
        const int ks_n = k <= 1 ? 2 : 1, taps = (k == 1 || k == 6) ? 1 : 9;
        const int n16 = taps * ks_n * 2 * 16;   // 16-byte pieces: (tap, k-step, half, output channel)
        for (int p = tid; p < n16; p += TL_THREADS) {
            const int co = p & 15, rest = p >> 4;   // rest = (tap * ks_n + ks) * 2 + half
            *reinterpret_cast<u32x4 *>(lds + g.w_off[k] + p * 16) = *reinterpret_cast<const u32x4 *>(a.cv[k].wt + (rest * 32 + co) * 8);
        }
    }
    // (not X: every slot of it is rewritten by each window's DMA, which other waves' zero stores could overtake)
    for (int p = g.r1_off / 16 + tid; p < g.t_off / 16; p += TL_THREADS) reinterpret_cast<u32x4 *>(lds)[p] = u32x4{0u, 0u, 0u, 0u};
    for (int q = tid; q < g.nt3 * 32; q += TL_THREADS) {
        const int yp = q / g.Wp3, xp = q - yp * g.Wp3;
        lds[g.mask3 + q] = (q < g.n3 && yp >= 1 && xp >= 1) ? 1 : 0;
        // where position q of level 3's output lies in the parity classes that level 4's stride-2 layer reads
        reinterpret_cast<unsigned short *>(lds + g.cls3)[q] = (unsigned short)(g.y0[(yp & 1) * 2 + (xp & 1)] + (yp >> 1) * g.Wp4 + (xp >> 1));
    }
    for (int q = tid; q < g.nt4 * 32; q += TL_THREADS) {
        const int yp = q / g.Wp4, xp = q - yp * g.Wp4;
        lds[g.mask4 + q] = (q < g.n4 && yp >= 1 && xp >= 1) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < TL_NCONV; ++k)
        if ((tid >> 5) == k) reinterpret_cast<float *>(lds + g.coef)[tid] = (tid & 31) < 16 ? a.cv[k].scale[tid & 15] : a.cv[k].shift[tid & 15];
    if (tid < g.F) {   // the classifier's BatchNorms, folded as head_fwd_eval_kernel folds them
        const float s = a.g2[tid] / sqrtf(a.rv2[tid] + TL_BN_EPS);
        reinterpret_cast<float *>(lds + g.zs)[tid] = s;
        reinterpret_cast<float *>(lds + g.zt)[tid] = a.b2[tid] - a.rm2[tid] * s;
    }
    if (tid < TL_HID) {
        const float s = a.g3[tid] / sqrtf(a.rv3[tid] + TL_BN_EPS);
        reinterpret_cast<float *>(lds + g.us)[tid] = s;
        reinterpret_cast<float *>(lds + g.ut)[tid] = a.b3[tid] - a.rm3[tid] * s;
        reinterpret_cast<float *>(lds + g.w2s)[tid] = a.lin2[tid];
        reinterpret_cast<float *>(lds + g.b1s)[tid] = a.bias1[tid];
    }

    // ---- the window's rows -> X: per lane and 1 KB chunk, which of the three places a 16-byte piece comes from and where in it -------
    // kind 0: the window's top strip (and its border row: the zeros of row yp = 0 and of the pad column), 1: its bottom strip, 2: its phase
    // stream; source = act + 64 * base[kind](window) + off
    constexpr int TL_NCH = 6;
    constexpr int n_slot = g.nX * 4, n_chunk = (n_slot + 63) >> 6;
    unsigned dsc[TL_NCH];
#pragma unroll
    for (int c = 0; c < TL_NCH; ++c) {
        const int s = ((wave + c * TL_WAVES) << 6) + lane;
        unsigned d = 0xffffffffu;
        if (s < n_slot) {
            const int ra = s >> 2, sp = s & 3;
            const int cls = ra >= g.x0[3] ? 3 : ra >= g.x0[2] ? 2 : ra >= g.x0[1] ? 1 : 0;
            const int rr = ra - g.x0[cls];
            const int I = rr / g.Wp3, J = rr - I * g.Wp3;
            const int yp = 2 * I + (cls >> 1), xp = 2 * J + (cls & 1);
            const int piece = sp ^ ((ra >> 2) & 3);
            int kind, pos;
            if (xp > W2) kind = 0, pos = 0;                                                   // the pad column of the odd classes: zeros
            else if (yp <= a.band) kind = 0, pos = yp * g.Wp2 + xp;                            // (yp = 0: the strip image's border row)
            else if (yp - 1 >= H2 - a.band) kind = 1, pos = (yp - (H2 - a.strip_rows)) * g.Wp2 + xp;
            else kind = 2, pos = (yp - 1) * g.Wp2 + xp;
            d = ((unsigned)kind << 30) | (unsigned)(pos * 64 + (xp > W2 ? 0 : piece * 16));
        }
        dsc[c] = d;
    }
    auto stage_in = [&](int win) __attribute__((always_inline)) {
        const long long b_top = (long long)win * a.img_t, b_bot = (a.bot_img0 + win) * (long long)a.img_t;
        const long long b_str = a.stream_row0 + (win & 1) * a.phase_img + ((win >> 1) + 1) * (long long)g.Wp2;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.act);
#pragma unroll
        for (int c = 0; c < TL_NCH; ++c) {
            const int ch = wave + c * TL_WAVES;
            if (ch < n_chunk && dsc[c] != 0xffffffffu) {
                const unsigned kind = dsc[c] >> 30;
                const long long base = kind == 0 ? b_top : kind == 1 ? b_bot : b_str;
                dma16(src + base * 64 + (dsc[c] & 0x3fffffffu), lds_addr(lds + g.x_off + ch * 1024));
            }
        }
    };

    // ---- one 32-position tile of one convolution: acc[channel][position] += W[tap] x X[position + shift(tap)] -------------------------
    // RB: bytes per source row (64: two k-steps per tap, 32: one); toff(tap): the source row of position 0 for that tap (class base and
    // region base included); SC: a second accumulator (the block's 1x1 shortcut) fed by the centre tap's fragments
    auto conv_tile = [&](auto RBc, auto TAPSc, auto SCc, int src_off, auto toff, int w_off, int w2_off, int q, f32x16 &acc, f32x16 &acc2) {
        constexpr int RB = decltype(RBc)::value, KS = RB / 32, TAPS = decltype(TAPSc)::value;
        constexpr bool SC = decltype(SCc)::value;
        const unsigned char *w_lane = lds + w_off + h * 256 + (i & 15) * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (SC) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int row = q + toff(tap);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 wf = *reinterpret_cast<const f16x8 *>(w_lane + (tap * KS + ks) * 512);
                const f16x8 xf = *reinterpret_cast<const f16x8 *>(lds + src_off + tl_off<RB>(row, ks * 2 + h));
                acc = mfma32_f16(wf, xf, acc);
                if (SC && tap == 4) {
                    const f16x8 wf2 = *reinterpret_cast<const f16x8 *>(lds + w2_off + h * 256 + (i & 15) * 16 + ks * 512);
                    acc2 = mfma32_f16(wf2, xf, acc2);
                }
            }
        }
    };
    // ---- a tile's epilogue: register 4 qd + j of lane (i, h) is channel 8 qd + 4 h + j of position i (qd < 2; the rest is padding) ----
    // MODE 0: conv_f16.hip's epilogue_f16 (fma, + 0.0f, optional ReLU); 1: block_f16_small_kernel's first convolution (fma, ReLU);
    // 2: its second one (fma, + residual, ReLU).  dst_row / res_row: absolute rows of the 32-byte-row tensors at dst_off / res_off.
    float fzero = 0.0f;
    asm volatile("" : "+v"(fzero));   // (an opaque + 0.0f: epilogue_f16 adds its absent residual, which turns a -0.0 into +0.0)
    auto epilogue = [&](auto MODEc, auto RELUc, const f32x16 &acc, int k, bool valid, bool keep_b, int dst_off, int dst_row, int res_off, int res_row) {
        constexpr int MODE = decltype(MODEc)::value;
        constexpr bool RELU = decltype(RELUc)::value;
        const float *cf = reinterpret_cast<const float *>(lds + g.coef) + k * 32 + 4 * h;
        const unsigned keep = keep_b ? 0xffffffffu : 0u;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(cf + 8 * qd);
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(cf + 16 + 8 * qd);
            f32x4 t = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
            t = __builtin_elementwise_fma(t, sv, bv);
            if (MODE == 0) t = t + f32x4{fzero, fzero, fzero, fzero};
            if (MODE == 2) {
                const f16x4 a4 = *reinterpret_cast<const f16x4 *>(lds + res_off + tl_off<32>(res_row, qd) + h * 8);
                t += f32x4{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
            }
            if (RELU) t = __builtin_elementwise_max(t, f32x4{0.f, 0.f, 0.f, 0.f});
            asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));   // (no fma + conversion contraction: conv_f16.hip, block_f16_strip_kernel)
            const tf16x2 lo = {(_Float16)t[0], (_Float16)t[1]}, hi = {(_Float16)t[2], (_Float16)t[3]};
            const u32x2 o = {__builtin_bit_cast(unsigned, lo) & keep, __builtin_bit_cast(unsigned, hi) & keep};
            if (valid) *reinterpret_cast<u32x2 *>(lds + dst_off + tl_off<32>(dst_row, qd) + h * 8) = o;
        }
    };
    // AvgPool2d(4) + the classifier of one window whose last activation lies in S3: wave 15 only (pool_f16_kernel, head_fwd_eval_kernel)
    auto classify = [&](int win) __attribute__((always_inline)) {
        float *pooled = reinterpret_cast<float *>(lds + g.pooled), *hid = reinterpret_cast<float *>(lds + g.hid);
        if (lane < g.F) {
            const int ppw = g.PH * g.PW;
            const int c = lane / ppw, ph = (lane / g.PW) % g.PH, pw = lane % g.PW;
            float s = 0.f;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int row = (1 + 4 * ph + dy) * g.Wp4 + (1 + 4 * pw + dx);
                    s += (float)*reinterpret_cast<const _Float16 *>(lds + g.s3_off + tl_off<32>(row, c >> 3) + (c & 7) * 2);
                }
            pooled[lane] = s * 0.0625f;
        }
        for (int f = lane + 64; f < g.F; f += 64) {   // (F > 64: not the product's geometry, kept correct)
            const int ppw = g.PH * g.PW;
            const int c = f / ppw, ph = (f / g.PW) % g.PH, pw = f % g.PW;
            float s = 0.f;
            for (int dy = 0; dy < 4; ++dy)
                for (int dx = 0; dx < 4; ++dx) {
                    const int row = (1 + 4 * ph + dy) * g.Wp4 + (1 + 4 * pw + dx);
                    s += (float)*reinterpret_cast<const _Float16 *>(lds + g.s3_off + tl_off<32>(row, c >> 3) + (c & 7) * 2);
                }
            pooled[f] = s * 0.0625f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < TL_HID) {
            const float *zs = reinterpret_cast<const float *>(lds + g.zs), *zt = reinterpret_cast<const float *>(lds + g.zt);
            float acc = reinterpret_cast<const float *>(lds + g.b1s)[lane];
            const float *w1 = a.lin1 + lane * g.F;
#pragma unroll 8
            for (int f = 0; f < g.F; ++f) {
                const float z = fmaf(pooled[f], zs[f], zt[f]);
                acc = fmaf(w1[f], z, acc);
            }
            hid[lane] = fmaxf(fmaf(acc, reinterpret_cast<const float *>(lds + g.us)[lane], reinterpret_cast<const float *>(lds + g.ut)[lane]), 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            const float *w2s = reinterpret_cast<const float *>(lds + g.w2s);
            float logit = a.bias2[0];
#pragma unroll 8
            for (int j = 0; j < TL_HID; ++j) logit = fmaf(w2s[j], hid[j], logit);
            a.probs[win] = tl_sigmoid(logit);
        }
    };

    using I64 = std::integral_constant<int, 64>;
    using I32 = std::integral_constant<int, 32>;
    using T9 = std::integral_constant<int, 9>;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    constexpr int Wp3 = g.Wp3, Wp4 = g.Wp4;
    // shifts of the nine taps: stride-1 layers on a pitch-Wp tensor whose first row is `base`; stride-2 layers on parity classes
    auto s1_3 = [&](int base) { return [=](int tap) { return base + (tap / 3 - 1) * Wp3 + (tap % 3 - 1); }; };
    auto s1_4 = [&](int base) { return [=](int tap) { return base + (tap / 3 - 1) * Wp4 + (tap % 3 - 1); }; };
    constexpr int x00 = g.x0[0], x01 = g.x0[1], x10 = g.x0[2], x11 = g.x0[3];
    auto s2_3 = [=](int tap) {
        const int ky = tap / 3, kx = tap % 3;
        const int cb = (ky & 1) ? ((kx & 1) ? x11 : x10) : ((kx & 1) ? x01 : x00);
        return cb + (ky >> 1) * Wp3 + (kx >> 1) - Wp3 - 1;
    };
    constexpr int y00 = g.y0[0], y01 = g.y0[1], y10 = g.y0[2], y11 = g.y0[3];
    auto s2_4 = [=](int tap) {
        const int ky = tap / 3, kx = tap % 3;
        const int cb = (ky & 1) ? ((kx & 1) ? y11 : y10) : ((kx & 1) ? y01 : y00);
        return cb + (ky >> 1) * Wp4 + (kx >> 1) - Wp4 - 1;
    };

    int win = (int)blockIdx.x;
    if (win < a.B) stage_in(win);
    int prev = -1;
    f32x16 acc, acc2;
    for (; win < a.B; win += (int)gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // the window's rows are in X (first window: and the tables are written)
        {   // per-lane addresses are recomputed per window, not hoisted out of the loop into more registers than a 1024-thread workgroup has
            int ll = lane;
            asm volatile("" : "+v"(ll));
            i = ll & 31, h = ll >> 5;
        }
        const int q3 = wave * 32 + i, q4 = wave * 32 + i;
        const bool t3 = wave < g.nt3, t4 = wave < g.nt4;
        const bool v3 = q3 < g.n3, v4 = q4 < g.n4;
        const bool k3 = t3 && lds[g.mask3 + (t3 ? q3 : 0)] != 0, k4 = t4 && lds[g.mask4 + (t4 ? q4 : 0)] != 0;
        // block3.0 conv1 (stride 2, from the classes) + its 1x1 shortcut: a1 -> R1, shortcut -> R3
        if (t3) {
            conv_tile(I64{}, T9{}, std::true_type{}, g.x_off, s2_3, g.w_off[0], g.w_off[1], q3, acc, acc2);
            epilogue(M0{}, std::true_type{}, acc, 0, v3, k3, g.r1_off, q3, 0, 0);
            epilogue(M0{}, std::false_type{}, acc2, 1, v3, k3, g.r3_off, q3, 0, 0);
        } else if (wave == TL_WAVES - 1 && prev >= 0) {
            classify(prev);
        }
        __syncthreads();
        if (win + (int)gridDim.x < a.B) stage_in(win + (int)gridDim.x);   // X is free: the next window's rows travel under the rest
        // block3.0 conv2 + shortcut -> y3a, in place over the shortcut (R3)
        if (t3) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_3(0), g.w_off[2], 0, q3, acc, acc2);
            epilogue(M2{}, std::true_type{}, acc, 2, v3, k3, g.r3_off, q3, g.r3_off, q3);
        }
        __syncthreads();
        // block3.1 conv1: y3a (R3) -> R1
        if (t3) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r3_off, s1_3(0), g.w_off[3], 0, q3, acc, acc2);
            epilogue(M1{}, std::true_type{}, acc, 3, v3, k3, g.r1_off, q3, 0, 0);
        }
        __syncthreads();
        // block3.1 conv2 + y3a -> the parity classes of level 4's stride-2 layer (R2)
        if (t3) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_3(0), g.w_off[4], 0, q3, acc, acc2);
            const int dst = reinterpret_cast<const unsigned short *>(lds + g.cls3)[q3];
            epilogue(M2{}, std::true_type{}, acc, 4, v3, k3, g.r2_off, dst, g.r3_off, q3);
        }
        __syncthreads();
        // block4.0 conv1 (stride 2) + shortcut: -> R1 / R3 at their level-4 rows
        if (t4) {
            conv_tile(I32{}, T9{}, std::true_type{}, g.r2_off, s2_4, g.w_off[5], g.w_off[6], q4, acc, acc2);
            epilogue(M0{}, std::true_type{}, acc, 5, v4, k4, g.r1_off, g.l4b + q4, 0, 0);
            epilogue(M0{}, std::false_type{}, acc2, 6, v4, k4, g.r3_off, g.l4b + q4, 0, 0);
        }
        __syncthreads();
        if (t4) {   // block4.0 conv2 + shortcut, in place (R3)
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_4(g.l4b), g.w_off[7], 0, q4, acc, acc2);
            epilogue(M2{}, std::true_type{}, acc, 7, v4, k4, g.r3_off, g.l4b + q4, g.r3_off, g.l4b + q4);
        }
        __syncthreads();
        if (t4) {   // block4.1 conv1: R3 -> R1
            conv_tile(I32{}, T9{}, std::false_type{}, g.r3_off, s1_4(g.l4b), g.w_off[8], 0, q4, acc, acc2);
            epilogue(M1{}, std::true_type{}, acc, 8, v4, k4, g.r1_off, g.l4b + q4, 0, 0);
        }
        __syncthreads();
        if (t4) {   // block4.1 conv2 + its input -> S3 (what the pool reads)
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_4(g.l4b), g.w_off[9], 0, q4, acc, acc2);
            epilogue(M2{}, std::true_type{}, acc, 9, v4, k4, g.s3_off, q4, g.r3_off, g.l4b + q4);
        }
        prev = win;
    }
    __syncthreads();
    if (wave == TL_WAVES - 1 && prev >= 0) classify(prev);
}

}  // namespace

// Everything behind the shared level 2 of the fp16 sliding-window path in one launch: windows [0, n_windows) of the buffer `act` that
// lad_f16_conv_s2_fwd_mapped reads with the same (H, W, band, strip_rows, bot_img0, stream_row0, phases = 2, phase_img) -- the level-2
// strips followed by the two phase streams, 32 channels -- through block3 and block4 (16 channels), AvgPool2d(4) and the classifier
// to probs[n_windows].  conv_params: HOST array of 30 device pointers, {packed weights (lad_f16_pack_weights), folded scale, folded shift}
// for block3.0 conv1, block3.0 shortcut, block3.0 conv2, block3.1 conv1, block3.1 conv2 and the same five of block4; head_params: as
// lad_head_fwd_eval.  Bit-identical to the launches it replaces (lad_f16_conv_s2_fwd_mapped(_sc), lad_f16_conv_fwd, lad_f16_block_fwd,
// lad_f16_conv_s2_fwd_sc, lad_f16_pool_fwd, lad_head_fwd_eval).  LAD_NOT_COVERED (nothing launched) for geometries it does not hold
// in a CU's LDS or whose sizes are odd at level 2: the caller issues those launches.
// Replaces segment_laughter.py:90-101 -> models.py:226-239 (block3, block4, pooling, classifier; eval mode).
extern "C" int lad_f16_tail_fwd(const void *act, int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows, int64_t bot_img0,
                                int64_t stream_row0, int32_t phases, int64_t phase_img, int64_t act_rows, const void *const *conv_params,
                                const float *const *head_params, int32_t F, float *probs, void *stream) {
    using namespace lad;
    LAD_REQUIRE(act && conv_params && head_params && probs, "lad_f16_tail_fwd: null buffer");
    LAD_REQUIRE(n_windows >= 0 && H >= 2 && W >= 2 && band >= 1 && strip_rows >= band, "lad_f16_tail_fwd: bad geometry");
    if (n_windows == 0) return LAD_OK;
    if (phases != 2 || (H & 1) || (W & 1) || 2 * band > H || n_windows >= (1 << 30)) return LAD_NOT_COVERED;
    if (H != 50 || W != 22) return LAD_NOT_COVERED;   // the instantiated geometry: 100 x 44 windows (config.FEAT), two stride-2 levels down
    const TailGeo g = tail_geo(H, W);
    if (g.PH < 1 || g.PW < 1 || g.F != F || g.F > TL_MAXF || g.nt3 > TL_WAVES - 1 || g.nt4 > TL_WAVES - 1 || g.total > 160 * 1024 ||
        (g.nX * 4 + 63) / 64 > 6 * TL_WAVES || g.Wp4 + 1 > g.Wp3 + 1 || g.nt3 * 32 + g.nY > 65535)
        return LAD_NOT_COVERED;
    // the furthest position a window reads must lie inside the buffer
    const int64_t last = n_windows - 1;
    const int64_t far_bot = (bot_img0 + last) * ((int64_t)(strip_rows + 1) * (W + 1)) + (int64_t)(strip_rows + 1) * (W + 1);
    const int64_t far_str = stream_row0 + phase_img + ((last >> 1) + 1 + H) * (int64_t)(W + 1);
    LAD_REQUIRE(far_bot <= act_rows && far_str <= act_rows, "lad_f16_tail_fwd: the window map reaches past the buffer (%lld rows)", (long long)act_rows);
    TailArgs a;
    a.act = (const _Float16 *)act;
    a.probs = probs;
    a.B = (int)n_windows; a.band = band; a.strip_rows = strip_rows; a.img_t = (strip_rows + 1) * (W + 1);
    a.bot_img0 = bot_img0; a.stream_row0 = stream_row0; a.phase_img = phase_img;
    for (int k = 0; k < TL_NCONV; ++k) {
        LAD_REQUIRE(conv_params[3 * k] && conv_params[3 * k + 1] && conv_params[3 * k + 2], "lad_f16_tail_fwd: null convolution parameter %d", k);
        a.cv[k] = TailConv{(const _Float16 *)conv_params[3 * k], (const float *)conv_params[3 * k + 1], (const float *)conv_params[3 * k + 2]};
    }
    for (int k = 0; k < 12; ++k) LAD_REQUIRE(head_params[k], "lad_f16_tail_fwd: null classifier parameter %d", k);
    a.g2 = head_params[0]; a.b2 = head_params[1]; a.rm2 = head_params[2]; a.rv2 = head_params[3]; a.lin1 = head_params[4]; a.bias1 = head_params[5];
    a.g3 = head_params[6]; a.b3 = head_params[7]; a.rm3 = head_params[8]; a.rv3 = head_params[9]; a.lin2 = head_params[10]; a.bias2 = head_params[11];
    static lad::DeviceOnce attr_set;
    static int n_cu = 256;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)tail_f16_kernel<50, 22>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    hipLaunchKernelGGL((tail_f16_kernel<50, 22>), dim3((unsigned)std::min<int64_t>(n_windows, n_cu)), dim3(TL_THREADS), (size_t)g.total, (hipStream_t)stream, a);
    return check_launch("tail_f16_kernel");
}
