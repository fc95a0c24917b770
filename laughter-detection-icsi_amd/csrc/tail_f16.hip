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
// Schedule: THREE windows in flight per workgroup, four steps (= four barriers) per window.  Waves 0-9 run block3 of window n (one
// 32-position tile each: conv1 + shortcut from X | conv2 | block3.1 conv1 | conv2 -> parity classes), waves 10-13 block4 of window n - 1 in
// the same four steps, wave 15 pooling + classifier of window n - 2 in the first three; the next window's rows travel by LDS-DMA during the
// last three (X is read in the first step only), two 1 KB chunks per wave and step.  tools/stamp_tail.py + profiles/r06_tail_stamps.log: the
// first version -- nine steps in a row -- took 21,500 cycles per window for 5,700 cycles of MFMA on the busiest SIMD (block4: four waves,
// one per SIMD, 6,300 of them); this one 14,100, which is what the SIMDs' issue slots allow: ~180 MFMAs (half of each is the padding of 16
// channels to the 32-row tile), ~2,200 vector and ~450 LDS instructions per SIMD and window, and on one SIMD those do not overlap.
// 8,192 windows: 460 us in nine launches -> ~200 us.
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
    int w_off[TL_NCONV];
    int x_off, r1_off, r3_off, r2_off, e1_off, s3_off, t_off, total;
    // tables behind t_off
    int coef, zs, zt, us, ut, w2s, b1s, pooled, hid;
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
    int o = 0;
    for (int k = 0; k < TL_NCONV; ++k) {
        const int cin = k <= 1 ? 32 : 16, taps = (k == 1 || k == 6) ? 1 : 9;
        g.w_off[k] = o;
        o += taps * (cin / 16) * 512;
    }
    // (regions are 32-byte aligned, not 256: a constant offset only rotates the banks a swizzled row's slots fall on)
    g.x_off = o; o += g.nX * 64;
    g.r1_off = o; o += (g.n3 + g.Wp3 + 1) * 32;        // + the zero rows a tensor's last image row reads below itself
    g.r3_off = o; o += (g.n3 + g.Wp3 + 1) * 32;
    g.r2_off = o; o += g.nY * 32;
    g.e1_off = o; o += (g.n4 + g.Wp4 + 1) * 32;
    g.s3_off = o; o += (g.n4 + g.Wp4 + 1) * 32;
    g.t_off = o;
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

#ifdef LAD_STAMP
// diagnostic build only (tools/stamp_tail.py): shader-clock time per phase (work / wait at the barrier behind it), summed over a
// workgroup's windows, of waves 0, 10, 14 and 15
__device__ unsigned long long lad_dbg_tail[256 * 4 * 24];
#define LAD_TL_T(k)                                                   \
    {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph_[k] += now_ - last_;                                       \
        last_ = now_;                                                 \
    }
#else
#define LAD_TL_T(k)
#endif

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

    // ---- the window's rows -> X, 1 KB (64 pieces of 16 bytes) per DMA instruction; a piece's place in X says where it comes from: its
    // parity class and (I, J) give the window's padded (yp, xp), and that one of three places of the buffer -- the window's top strip (kind 0;
    // also its border row: the zeros of row yp = 0 and of the odd classes' pad column), its bottom strip (1), or its phase stream (2).  One
    // wave-uniform base (the top strip) + a 32-bit offset per lane: the other two lie behind it in the buffer (the launcher checks its size).
    // Wave w owns chunks w, w + 16, ...: their (kind, offset) descriptors are formed once and live in five registers.  (Formed per chunk
    // by the two waves without a tile -- ~70 instructions each -- the descriptors cost more than the convolutions: those waves share their
    // SIMDs' issue slots with the tile waves; tools/stamp_tail.py, third version: 650 cycles per chunk.)
    constexpr int n_slot = g.nX * 4, n_chunk = (n_slot + 63) >> 6, TL_NCH = (n_chunk + TL_WAVES - 1) / TL_WAVES;
    static_assert(TL_NCH <= 6, "two chunks per wave in each of the three steps behind a window's first");
    unsigned dsc[TL_NCH];
#pragma unroll
    for (int c = 0; c < TL_NCH; ++c) {
        const int sl = ((wave + c * TL_WAVES) << 6) + lane;
        unsigned d = 0xffffffffu;
        if (sl < n_slot) {
            const int ra = sl >> 2, sp = sl & 3;
            const int cls = ra >= g.x0[3] ? 3 : ra >= g.x0[2] ? 2 : ra >= g.x0[1] ? 1 : 0;
            const int rr = ra - (cls == 3 ? g.x0[3] : cls == 2 ? g.x0[2] : cls == 1 ? g.x0[1] : 0);
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
    struct WinBase { const unsigned char *src; unsigned d_bot, d_str; };
    auto win_base = [&](int win) __attribute__((always_inline)) {
        const long long b_top = (long long)win * a.img_t;
        WinBase wb;
        wb.src = reinterpret_cast<const unsigned char *>(a.act) + b_top * 64;
        wb.d_bot = (unsigned)(a.bot_img0 * (long long)a.img_t) * 64u;
        wb.d_str = (unsigned)(a.stream_row0 + (win & 1) * a.phase_img + ((win >> 1) + 1) * (long long)g.Wp2 - b_top) * 64u;
        return wb;
    };
    // step st (0, 1, 2: the three steps behind a window's first, in which X is no longer read) issues this wave's chunks 2 st and 2 st + 1
    auto dma_step = [&](int st, const WinBase &wb) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < TL_NCH; ++c) {
            if (c / 2 != st) continue;
            const int ch = wave + c * TL_WAVES;
            if (ch < n_chunk && dsc[c] != 0xffffffffu) {
                const unsigned kind = dsc[c] >> 30;
                const unsigned off = (dsc[c] & 0x3fffffffu) + (kind == 1 ? wb.d_bot : kind == 2 ? wb.d_str : 0u);
                dma16s(wb.src, off, lds_addr(lds + g.x_off + ch * 1024));
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
        // The fragments of a GROUP of taps are requested together, then multiplied: left to itself the compiler reads one or two MFMAs
        // ahead, and with one to three waves per SIMD every MFMA then waits out most of an LDS round trip (tools/stamp_tail.py: 1,250-2,100
        // cycles for the nine MFMAs of a level-4 tile).  Nine taps of a 16-channel layer = 72 registers; the 32-channel layer goes three
        // taps (six k-steps) at a time next to its second accumulator.
        constexpr int TG = KS == 1 ? 5 : 2, NG = (TAPS + TG - 1) / TG;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            f16x8 wf[TG * KS], xf[TG * KS], wf2[SC ? KS : 1];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tap = grp * TG + t;
                if (tap < TAPS) {
                    const int row = q + toff(tap);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        wf[t * KS + ks] = *reinterpret_cast<const f16x8 *>(w_lane + (tap * KS + ks) * 512);
                        xf[t * KS + ks] = *reinterpret_cast<const f16x8 *>(lds + src_off + tl_off<RB>(row, ks * 2 + h));
                        if (SC && tap == 4) wf2[ks] = *reinterpret_cast<const f16x8 *>(lds + w2_off + h * 256 + (i & 15) * 16 + ks * 512);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tap = grp * TG + t;
                if (tap < TAPS) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        acc = mfma32_f16(wf[t * KS + ks], xf[t * KS + ks], acc);
                        if (SC && tap == 4) acc2 = mfma32_f16(wf2[ks], xf[t * KS + ks], acc2);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- a tile's epilogue: register 4 qd + j of lane (i, h) is channel 8 qd + 4 h + j of position i (qd < 2; the rest is padding) ----
    // MODE 0: conv_f16.hip's epilogue_f16 (fma, + 0.0f, optional ReLU); 1: block_f16_small_kernel's first convolution (fma, ReLU);
    // 2: its second one (fma, + residual, ReLU).  dst_row / res_row: absolute rows of the 32-byte-row tensors at dst_off / res_off.
    float fzero = 0.0f;
    asm volatile("" : "+v"(fzero));   // (an opaque + 0.0f: epilogue_f16 adds its absent residual, which turns a -0.0 into +0.0)
    // 3: mode 2 with the residual taken from `regs` (the packed halves another epilogue left there: regs != nullptr with mode 0 = "store to regs").
    auto epilogue = [&](auto MODEc, auto RELUc, const f32x16 &acc, int k, bool valid, bool keep_b, int dst_off, int dst_row, int res_off, int res_row,
                        u32x2 *regs) __attribute__((always_inline)) {
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
            if (MODE == 2 || MODE == 3) {
                const f16x4 a4 = MODE == 3 ? __builtin_bit_cast(f16x4, regs[qd])
                                           : *reinterpret_cast<const f16x4 *>(lds + res_off + tl_off<32>(res_row, qd) + h * 8);
                t += f32x4{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
            }
            if (RELU) t = __builtin_elementwise_max(t, f32x4{0.f, 0.f, 0.f, 0.f});
            asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));   // (no fma + conversion contraction: conv_f16.hip, block_f16_strip_kernel)
            const tf16x2 lo = {(_Float16)t[0], (_Float16)t[1]}, hi = {(_Float16)t[2], (_Float16)t[3]};
            const u32x2 o = {__builtin_bit_cast(unsigned, lo) & keep, __builtin_bit_cast(unsigned, hi) & keep};
            if (MODE == 0 && regs != nullptr) regs[qd] = o;
            else if (valid) *reinterpret_cast<u32x2 *>(lds + dst_off + tl_off<32>(dst_row, qd) + h * 8) = o;
        }
    };
    // AvgPool2d(4) + the classifier of the window whose last activation lies in S3 (pool_f16_kernel, head_fwd_eval_kernel): wave 15, in
    // three phases that ride in three consecutive steps (a latency chain of ~4,500 cycles as one piece: twice a step of the waves beside it).
    static_assert(g.F % 4 == 0 && g.F <= 64, "classifier: one pooled feature per lane");
    auto classify_a = [&]() __attribute__((always_inline)) {   // pool + BatchNorm1d -> z[F] (LDS)
        int cl = lane;
        asm volatile("" : "+v"(cl));   // (addresses are formed here, not carried through the window loop in registers)
        if (cl < g.F) {
            constexpr int ppw = g.PH * g.PW;
            const int c = cl / ppw, ph = (cl / g.PW) % g.PH, pw = cl % g.PW;
            _Float16 v[16];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int row = (1 + 4 * ph + dy) * g.Wp4 + (1 + 4 * pw + dx);
                    v[dy * 4 + dx] = *reinterpret_cast<const _Float16 *>(lds + g.s3_off + tl_off<32>(row, c >> 3) + (c & 7) * 2);
                }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += (float)v[k];
            // z = BatchNorm1d(pooled), once per feature (head_fwd_eval_kernel computes the same fmaf in every thread)
            reinterpret_cast<float *>(lds + g.pooled)[cl] =
                fmaf(sum * 0.0625f, reinterpret_cast<const float *>(lds + g.zs)[cl], reinterpret_cast<const float *>(lds + g.zt)[cl]);
        }
    };
    auto classify_b = [&]() __attribute__((always_inline)) {   // Linear(F, 32) -> BatchNorm1d -> ReLU -> hid[32] (LDS)
        int cl = lane;
        asm volatile("" : "+v"(cl));
        if (cl < TL_HID) {
            f32x4 w1r[g.F / 4], z[4];   // this lane's row of linear1: requested first, the chain below consumes it
            const f32x4 *w1 = reinterpret_cast<const f32x4 *>(a.lin1 + cl * g.F);
#pragma unroll
            for (int k = 0; k < g.F / 4; ++k) w1r[k] = w1[k];
            float acc1 = reinterpret_cast<const float *>(lds + g.b1s)[cl];
#pragma unroll
            for (int k0 = 0; k0 < g.F / 4; k0 += 4) {   // (z four quads at a time: all twelve next to the row are 96 registers)
#pragma unroll
                for (int k = k0; k < k0 + 4 && k < g.F / 4; ++k) z[k - k0] = reinterpret_cast<const f32x4 *>(lds + g.pooled)[k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = k0; k < k0 + 4 && k < g.F / 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1 = fmaf(w1r[k][e], z[k - k0][e], acc1);
                __builtin_amdgcn_sched_barrier(0);
            }
            reinterpret_cast<float *>(lds + g.hid)[cl] =
                fmaxf(fmaf(acc1, reinterpret_cast<const float *>(lds + g.us)[cl], reinterpret_cast<const float *>(lds + g.ut)[cl]), 0.f);
        }
    };
    auto classify_c = [&](int win) __attribute__((always_inline)) {   // Linear(32, 1) -> sigmoid
        int cl = lane;
        asm volatile("" : "+v"(cl));
        if (cl == 0) {
            f32x4 hv[TL_HID / 4], wv[TL_HID / 4];
#pragma unroll
            for (int k = 0; k < TL_HID / 4; ++k) {
                hv[k] = reinterpret_cast<const f32x4 *>(lds + g.hid)[k];
                wv[k] = reinterpret_cast<const f32x4 *>(lds + g.w2s)[k];
            }
            float logit = a.bias2[0];
#pragma unroll
            for (int k = 0; k < TL_HID / 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) logit = fmaf(wv[k][e], hv[k][e], logit);
            a.probs[win] = tl_sigmoid(logit);
        }
    };

    using I64 = std::integral_constant<int, 64>;
    using I32 = std::integral_constant<int, 32>;
    using T9 = std::integral_constant<int, 9>;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
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

    // ---- the window loop: three windows in flight per workgroup ------------------------------------------------------------------------
    //   waves 0-9   (A): block3 of window `it`        -- four steps, one 32-position tile per wave
    //   waves 10-13 (B): block4 of window `it - 1`    -- the same four steps (its input, level 3's output, was written in the last one)
    //   wave 15     (C): pooling + classifier of window `it - 2`, in the first three steps;  wave 14 (+ everybody, a little): the DMA of window
    //   `it + 1`, in the last three (X is read in the first step only).  Four barriers per window instead of nine, and block4 -- four waves
    //   at one per SIMD, 6,300 of the first version's 20,400 cycles per window -- no longer holds the other twelve up.
    // Buffers: R1 a1 / a1' (A), R3 shortcut -> y3a in place (A), R2 level 3's output as parity classes (A step 4 -> B step 1), E1 a4 / a4' (B),
    // S3 y4a (B step 2 -> 3), then block4's output in place (B step 4 -> C step 1); the block4.0 shortcut waits in registers (B step 1 -> 2).
    const int n_mine = (int)blockIdx.x < a.B ? (a.B - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int stride = (int)gridDim.x;
    if (n_mine > 0) {
        const WinBase wb = win_base((int)blockIdx.x);
#pragma unroll
        for (int st = 0; st < 3; ++st) dma_step(st, wb);
    }
    f32x16 acc, acc2;
    u32x2 cs4[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};   // block4.0's shortcut of this wave's tile, as the halves epilogue_f16 would have stored
    // (block4 on the OLDEST waves of the SIMDs instead -- waves 0-3, block3 on 4-13 -- is the same trade as priority below: measured 14,700
    // cycles per window against 14,100)
    const bool roleA = wave < 10, roleB = wave >= 10 && wave < 14, roleC = wave == 15;
    static_assert(g.nt3 <= 10 && g.nt4 <= 4, "wave roles");
    // The classifier wave is the YOUNGEST wave of its SIMD, behind two block3 waves and a block4 wave, and issue arbitration goes by
    // priority, then age (MI355X_MICROARCH.md, "Two waves per SIMD"): its three short dependent chains took 2,500-3,200 cycles each and two of
    // the four steps waited for them.  It gets priority -- a few hundred instructions per window.  (Priority for the four block4 waves as
    // well was zero-sum: their steps 3,500 / 2,900 / 2,300 / 2,800 -> 2,300 / 2,100 / 1,850 / 2,000 cycles, block3's 2,900 / 2,100 / 1,750 /
    // 2,000 -> 3,400 / 2,500 / 2,100 / 2,400, the window 14,300 -> 14,500: tools/stamp_tail.py, profiles/r06_tail_stamps.log.)
    if (wave == 15) __builtin_amdgcn_s_setprio(3);
#ifdef LAD_STAMP
    unsigned long long ph_[24], last_ = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 24; ++k) ph_[k] = 0;
#endif
#pragma unroll 1
    for (int it = 0; it < n_mine + 2; ++it) {
        const bool doA = roleA && it < n_mine && wave < g.nt3, doB = roleB && it >= 1 && it <= n_mine && wave - 10 < g.nt4;
        const bool doC = roleC && it >= 2, more = it + 1 < n_mine;
        const int win_c = (int)blockIdx.x + (it - 2) * stride;
        LAD_TL_T(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LAD_TL_T(1)
        __syncthreads();   // window `it`'s rows are in X; the previous iteration's last step is complete (first: the tables are written)
        LAD_TL_T(2)
        {   // per-lane addresses are recomputed per window, not hoisted out of the loop into more registers than a 1024-thread workgroup has
            int ll = lane;
            asm volatile("" : "+v"(ll));
            i = ll & 31, h = ll >> 5;
        }
        const WinBase wb = win_base((int)blockIdx.x + (it + 1) * stride);
        // this lane's position: level 3 (A) or level 4 (B); is it inside the tensor, is it an interior (non-border) position
        const int q = (roleA ? wave : wave - 10) * 32 + i;
        const int qy = roleA ? q / g.Wp3 : q / g.Wp4, qx = q - qy * (roleA ? g.Wp3 : g.Wp4);   // (constant divisors)
        const bool valid = q < (roleA ? g.n3 : g.n4), keep = valid && qy >= 1 && qx >= 1;
        // ---- step 1: A block3.0 conv1 (stride 2, from the classes in X) + its 1x1 shortcut: a1 -> R1, shortcut -> R3
        //              B block4.0 conv1 (stride 2, from the classes in R2) + shortcut: a4 -> E1, shortcut -> registers;  C pooling
        if (doA) {
            conv_tile(I64{}, T9{}, std::true_type{}, g.x_off, s2_3, g.w_off[0], g.w_off[1], q, acc, acc2);
            epilogue(M0{}, std::true_type{}, acc, 0, valid, keep, g.r1_off, q, 0, 0, nullptr);
            epilogue(M0{}, std::false_type{}, acc2, 1, valid, keep, g.r3_off, q, 0, 0, nullptr);
        } else if (doB) {
            conv_tile(I32{}, T9{}, std::true_type{}, g.r2_off, s2_4, g.w_off[5], g.w_off[6], q, acc, acc2);
            epilogue(M0{}, std::true_type{}, acc, 5, valid, keep, g.e1_off, q, 0, 0, nullptr);
            epilogue(M0{}, std::false_type{}, acc2, 6, false, keep, 0, 0, 0, 0, cs4);
        } else if (doC) {
            classify_a();
        }
        LAD_TL_T(3)
        __syncthreads();
        LAD_TL_T(4)
        // ---- step 2: A block3.0 conv2 + shortcut -> y3a, in place over the shortcut (R3);  B block4.0 conv2 + shortcut -> y4a (S3);
        //              C hidden layer;  the next window's rows start to travel (X is free from here on)
        if (doA) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_3(0), g.w_off[2], 0, q, acc, acc2);
            epilogue(M2{}, std::true_type{}, acc, 2, valid, keep, g.r3_off, q, g.r3_off, q, nullptr);
        } else if (doB) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.e1_off, s1_4(0), g.w_off[7], 0, q, acc, acc2);
            epilogue(M3{}, std::true_type{}, acc, 7, valid, keep, g.s3_off, q, 0, 0, cs4);
        } else if (doC) {
            classify_b();
        }
        if (more) dma_step(0, wb);
        LAD_TL_T(5)
        __syncthreads();
        LAD_TL_T(6)
        // ---- step 3: A block3.1 conv1: y3a (R3) -> R1;  B block4.1 conv1: y4a (S3) -> E1;  C output layer
        if (doA) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r3_off, s1_3(0), g.w_off[3], 0, q, acc, acc2);
            epilogue(M1{}, std::true_type{}, acc, 3, valid, keep, g.r1_off, q, 0, 0, nullptr);
        } else if (doB) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.s3_off, s1_4(0), g.w_off[8], 0, q, acc, acc2);
            epilogue(M1{}, std::true_type{}, acc, 8, valid, keep, g.e1_off, q, 0, 0, nullptr);
        } else if (doC) {
            classify_c(win_c);
        }
        if (more) dma_step(1, wb);
        LAD_TL_T(7)
        __syncthreads();
        LAD_TL_T(8)
        // ---- step 4: A block3.1 conv2 + y3a -> the parity classes of block4.0's stride-2 layer (R2);  B block4.1 conv2 + y4a -> in place (S3)
        if (doA) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.r1_off, s1_3(0), g.w_off[4], 0, q, acc, acc2);
            const int dst = ((qy & 1) ? ((qx & 1) ? y11 : y10) : ((qx & 1) ? y01 : y00)) + (qy >> 1) * g.Wp4 + (qx >> 1);
            epilogue(M2{}, std::true_type{}, acc, 4, valid, keep, g.r2_off, dst, g.r3_off, q, nullptr);
        } else if (doB) {
            conv_tile(I32{}, T9{}, std::false_type{}, g.e1_off, s1_4(0), g.w_off[9], 0, q, acc, acc2);
            epilogue(M2{}, std::true_type{}, acc, 9, valid, keep, g.s3_off, q, g.s3_off, q, nullptr);
        }
        if (more) dma_step(2, wb);
        LAD_TL_T(9)
    }
#ifdef LAD_STAMP
    if ((wave == 0 || wave == 10 || wave == 14 || wave == 15) && lane == 0 && blockIdx.x < 256)
        for (int j = 0; j < 24; ++j) lad_dbg_tail[(blockIdx.x * 4 + (wave == 0 ? 0 : wave == 10 ? 1 : wave == 14 ? 2 : 3)) * 24 + j] = ph_[j];
#endif
}

}  // namespace

#ifdef LAD_STAMP
extern "C" int lad_debug_read_tail_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_tail), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

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
    if (g.PH < 1 || g.PW < 1 || g.F != F || g.F > TL_MAXF || g.nt3 > 10 || g.nt4 > 4 || g.total > 160 * 1024 ||
        (g.nX * 4 + 63) / 64 > 3 * 26)
        return LAD_NOT_COVERED;
    // the furthest position a window reads must lie inside the buffer
    const int64_t last = n_windows - 1;
    const int64_t far_bot = (bot_img0 + last) * ((int64_t)(strip_rows + 1) * (W + 1)) + (int64_t)(strip_rows + 1) * (W + 1);
    const int64_t far_str = stream_row0 + phase_img + ((last >> 1) + 1 + H) * (int64_t)(W + 1);
    if (act_rows >= (1ll << 31) / 64) return LAD_NOT_COVERED;   // (32-bit byte offsets from a window's top strip to its other rows)
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
