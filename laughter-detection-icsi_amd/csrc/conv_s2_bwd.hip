// Backward of the stride-2 convolutions (3x3 pad 1 and the 1x1 shortcut) at their true cost.
//
// Replaces what autograd computes for nn.Conv2d(stride=2) in loss.backward() (train.py:289; layers models.py:86-89
// with stride 2 and the shortcut models.py:102-105).  The generic route (zero-stuff the output gradient to the input
// resolution, then run the stride-1 data-/weight-gradient kernels) spends 4x the necessary MFMA work on rows that
// are zero by construction; these two kernels touch only the non-zero terms:
//
//   dgrad_s2   dx[b,y,x,:] = sum over taps (ky,kx) with (y+1-ky), (x+1-kx) even of dout[b,(y+1-ky)/2,(x+1-kx)/2,:] * W[ky,kx]
//              Input positions fall into four parity classes (y&1, x&1) that use 1, 2, 2 or 4 taps.  A workgroup owns
//              128 positions of ONE class (so every row of an MFMA tile uses the same taps / the same B operand),
//              gathers its A fragments (rows of the low-resolution dout) per lane from HBM/L2 and scatters whole
//              256-byte rows of dx through the LDS transpose.  blockIdx.y = class.
//   wgrad_s2   dW[co,ci,ky,kx] = sum over output positions of in[b,2yo+ky-1,2xo+kx-1,ci] * dout[b,yo,xo,co]
//              GEMM K = low-resolution rows, split over persistent workgroups like wgrad_mfma.hip; a tile is one row of
//              output positions, whose dout rows and three input image rows are contiguous spans staged in LDS.
//
// Layout, zero-border invariant: lad_device.h.  Weight images: the mode-1 ("dgrad") packing of conv_mfma.hip.
#include <type_traits>

#include "lad_common.h"
#include "lad_device.h"
#include "lad_bn_math.h"

namespace {
using namespace lad;

constexpr int THREADS = 256;
constexpr int TM = 128;

template <int N>
struct NTl {
    static constexpr int NT = (N + 31) / 32;
    static constexpr int NP = NT * 32;
};

// class-local index m of class (py, px) -> flat full-resolution row; false if m is past the end of the class
struct ClassGeom {
    int A, Bx;          // positions per image along y / x in this class
    int py, px;
    int64_t total;      // batch * A * Bx
};

// n / d for n < 2^31 through one f64 multiply with inv = 1.0 / d (quotient at most one short: fixed up); the 64-bit
// integer divisions this replaces cost ~150 instructions each and ran 9 times per lane per tile
__device__ __forceinline__ uint32_t udiv_f64(uint32_t n, uint32_t d, double inv, uint32_t &rem) {
    uint32_t qt = (uint32_t)((double)n * inv);
    uint32_t r = n - qt * d;
    if (r >= d) { r -= d; ++qt; }
    rem = r;
    return qt;
}

// KC = channels of dout (the conv's cout), NC = channels of dx (the conv's cin)
// SC (3x3 only): the 1x1 stride-2 shortcut's data gradient dx[2yo, 2xo] += dout_sc[yo, xo] * W_sc lands exactly on parity
// class (0, 0), whose positions take ONE tap of the 3x3 from the same low-resolution position: one more tap for that class
// (A rows from dout_sc, B image wt_sc) instead of a second launch that re-reads and re-writes a quarter of dx.
// STAT (with SC, 64 input channels): dx is final when this launch writes it, so the first pass of the BatchNorm backward that
// consumes it -- the bn2 of the block below, whose ReLU decisions are its sign bits -- rides in the epilogue as it does in
// conv_b3.hip: per workgroup (sum dz, sum dz * xhat) -> stat_partials[class][workgroup][2][NC] for lad_bn_bwd_bits.
struct S2Stat {
    const float *x;                   // input of the consuming BatchNorm (geometry of dx)
    const unsigned long long *bits;   // sign bits of its output
    const float *coef;                // float[6][NC]
    float *partials;                  // float[4 * gridDim.x][2][NC]
};

template <int KC, int NC, int TAPS, bool SC = false, bool STAT = false>
__global__ __launch_bounds__(THREADS, 2) void dgrad_s2_kernel(const float *__restrict__ dout, const float *__restrict__ wt,
                                                              float *__restrict__ dx, Geom glo, Geom ghi, int H, int W,
                                                              int64_t batch, int accumulate,
                                                              const float *__restrict__ dout_sc = nullptr,
                                                              const float *__restrict__ wt_sc = nullptr,
                                                              S2Stat bst = S2Stat{nullptr, nullptr, nullptr, nullptr}) {
    static_assert(!SC || TAPS == 9, "the shortcut rides with the 3x3 convolution");
    static_assert(!STAT || (SC && NC == 64), "the fused sums are written for the 64-channel transition with the shortcut fused");
    constexpr int NT = NTl<NC>::NT;
    constexpr int NP = NTl<NC>::NP;
    constexpr int K4 = KC / 4;
    constexpr int G = KC / 8;
    constexpr int LDO = NC + 4;
    __shared__ __attribute__((aligned(16))) float out_s[TM * LDO];
    __shared__ int qrow_s[TM];  // full-resolution row of each of the tile's outputs, -1 past the end of the class
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    ClassGeom c;
    c.py = (TAPS == 9) ? (blockIdx.y >> 1) : 0;
    c.px = (TAPS == 9) ? (blockIdx.y & 1) : 0;
    c.A = (H - c.py + 1) / 2;
    c.Bx = (W - c.px + 1) / 2;
    c.total = batch * c.A * c.Bx;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    if (m0 >= c.total) {  // classes differ in size; the grid is sized for the largest
        if (STAT && tid < 2 * NC) bst.partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * NC) + tid] = 0.0f;
        return;
    }

    // this lane's output position (class-local index -> image, a, b), 32-bit arithmetic (launcher: rows < 2^31)
    const uint32_t per_img = (uint32_t)(c.A * c.Bx);
    const uint32_t m = (uint32_t)m0 + wave * 32 + i;
    const bool ok = (int64_t)m < c.total;
    uint32_t r, bq;
    const uint32_t img = udiv_f64(ok ? m : 0u, per_img, 1.0 / (double)per_img, r);
    const uint32_t aq = udiv_f64(r, (uint32_t)c.Bx, 1.0 / (double)c.Bx, bq);
    const int a = (int)aq, b = (int)bq;
    const int64_t lo_base = (int64_t)img * glo.img;
    if (gk == 0) qrow_s[wave * 32 + i] = ok ? (int)(img * (uint32_t)ghi.img + (uint32_t)(2 * a + c.py + 1) * (uint32_t)ghi.Wp + (uint32_t)(2 * b + c.px + 1)) : -1;

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[n][rr] = 0.0f;

    const float *w_base = wt + (gk * NP + i) * 4;
    // taps of this class: ky in {1} (py = 0) or {0, 2} (py = 1); the same for kx.  Lanes past the end of the class gather
    // position 0 of image 0 (in-bounds, never stored), so the loop carries no predicate.
    const int nky = (TAPS == 9 && c.py) ? 2 : 1, nkx = (TAPS == 9 && c.px) ? 2 : 1;
    const int ntap = nky * nkx;
    const int nloop = ntap + ((SC && blockIdx.y == 0) ? 1 : 0);   // class (0, 0): + the shortcut
#pragma unroll 1
    for (int t = 0; t < nloop; ++t) {
        const bool sc_tap = SC && t == ntap;   // (class (0,0): same low-resolution position as its centre tap)
        const int ky = (TAPS == 9) ? (c.py ? 2 * (t / nkx) : 1) : 0;
        const int kx = (TAPS == 9) ? (c.px ? 2 * (t % nkx) : 1) : 0;
        // y = 2a + py, yo = (y + 1 - ky) / 2 (3x3 pad 1)  |  yo = y / 2 (1x1 pad 0);  padded low-res coordinate yo + 1
        const int ypo = (TAPS == 9) ? (2 * a + c.py + 1 - ky) / 2 + 1 : a + 1;
        const int xpo = (TAPS == 9) ? (2 * b + c.px + 1 - kx) / 2 + 1 : b + 1;
        const float *ap = (sc_tap ? dout_sc : dout) + (lo_base + (int64_t)ypo * glo.Wp + xpo) * KC + 4 * gk;
        // mode-1 image: tap slot t' holds w[.., taps-1-t'], so the unflipped tap (ky,kx) sits at slot 8 - (3 ky + kx)
        const int slot = (TAPS == 9) ? 8 - (3 * ky + kx) : 0;
        const float *wp = sc_tap ? wt_sc + (gk * NP + i) * 4 : w_base + slot * (K4 * NP * 4);
        float4 av[G], bv[G][NT];  // the whole tap's fragments requested together, then its MFMAs
#pragma unroll
        for (int c8 = 0; c8 < G; ++c8) {
            av[c8] = *reinterpret_cast<const float4 *>(ap + c8 * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[c8][n] = *reinterpret_cast<const float4 *>(wp + (c8 * 2 * NP + n * 32) * 4);
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
    }
    // ---- transpose through LDS (wave-private region), then whole rows of dx ----------------------------------------
    float *my = out_s + wave * 32 * LDO;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        if (co < NC) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) my[acc_row(rr, lane) * LDO + co] = acc[n][rr];
        }
    }
    constexpr int LPR = NC / 4, RPI = 64 / LPR, ITER = 32 / RPI;
    const int c4 = lane % LPR, rsub = lane / LPR;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    Norm4 nm;
    float4 xs[STAT ? ITER : 1];
    unsigned long long wds[STAT ? ITER : 1];
    if (STAT) {
        nm = load_norm(bst.coef, NC, c4 * 4);
        // all rows' BatchNorm inputs and sign words requested together (rows past the class read row 0: never used)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int q2 = max(qrow_s[wave * 32 + it * RPI + rsub], 0);
            xs[it] = *reinterpret_cast<const float4 *>(bst.x + (int64_t)q2 * NC + c4 * 4);
            wds[it] = bst.bits[q2];
        }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = it * RPI + rsub;
        const int q2 = qrow_s[wave * 32 + row];  // written by this wave's own lanes (LDS is in order within a wave)
        if (q2 >= 0) {
            float4 v = *reinterpret_cast<const float4 *>(my + row * LDO + c4 * 4);
            float4 *dst = reinterpret_cast<float4 *>(dx + (int64_t)q2 * NC + c4 * 4);
            if (accumulate) {
                const float4 o = *dst;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *dst = v;
            if (STAT) {   // the arithmetic of bn_bwd_reduce_kernel (bn.hip), relu = 3
                const float4 xv = xs[it];
                const float4 d = mask_from_bits(v, wds[it], c4);
                const float4 xh = xhat4(xv, nm);
                s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
                s2.x = fmaf(d.x, xh.x, s2.x); s2.y = fmaf(d.y, xh.y, s2.y);
                s2.z = fmaf(d.z, xh.z, s2.z); s2.w = fmaf(d.w, xh.w, s2.w);
            }
        }
    }
    if (STAT) {
        // this wave's (now consumed) slice of the output tile takes its RPI x 2 x NC partial sums; 2 NC threads add the 16
        *reinterpret_cast<float4 *>(my + (rsub * 2 + 0) * NC + c4 * 4) = s1;
        *reinterpret_cast<float4 *>(my + (rsub * 2 + 1) * NC + c4 * 4) = s2;
        __syncthreads();
        if (tid < 2 * NC) {
            const int k = tid / NC, co = tid - k * NC;
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int rs = 0; rs < RPI; ++rs) t += out_s[w * 32 * LDO + (rs * 2 + k) * NC + co];
            bst.partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * NC) + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// A tile is ONE row of output positions (b, yo): its K dimension is the padded low-resolution image row (Wp_lo
// consecutive rows of dout, the border position is zero) and the input it needs is the three padded input image rows
// 2yo, 2yo+1, 2yo+2 plus the border position after them -- one contiguous span of 3*Wp_hi + 1 rows.  Both are staged in LDS (the span of tile t+1 travels
// HBM -> registers while tile t's MFMAs run); the A operand of tap (ky,kx) at column xo is then an LDS read at row
// ky*Wp_hi + 2*xo + kx.
constexpr int MAX_GROUPS = 768;   // persistent workgroups (38 KB of LDS at 64 channels x 46 columns: 3 per CU)
constexpr int W2_PRE = 10;        // float4 registers per thread for the next tile's input span (3*Wp_hi*CIN/4 <= 2560)

template <int CIN, int COUT, int TAPS>
struct W2Cfg {
    static constexpr int MT = (CIN + 31) / 32, NT = (COUT + 31) / 32, MN = MT * NT;
    static constexpr int TSTRIDE = (MN >= 4) ? 1 : 4 / MN;
    static constexpr int TPW = (TAPS + TSTRIDE - 1) / TSTRIDE;
};

// SC (3x3 only): the 1x1 stride-2 shortcut convolution of the same block reads the same input -- its taps are the
// centre tap's rows (2yo+1, 2xo+1) -- so its weight gradient dW_sc[ci][co] = sum in_centre[ci] * dout_sc[co] rides along as a
// TENTH tap with its own output-gradient rows (dout_sc: the gradient into the shortcut BatchNorm's input): the wave that owns
// the odd taps has a free accumulator slot (9 taps over 2 or 4 tap-owners).  On its own that gradient was a launch that
// staged three image rows per tile to use one (173 us at 64 -> 32, batch 512, for 1/9 of the 3x3's arithmetic).
template <int CIN, int COUT, int TAPS, bool SC = false>
__global__ __launch_bounds__(THREADS, 3) void wgrad_s2_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                              float *__restrict__ slabs, float *__restrict__ bias_slabs, Geom ghi,
                                                              Geom glo, int64_t n_tiles, int Ho,
                                                              const float *__restrict__ dout_sc = nullptr,
                                                              float *__restrict__ slabs_sc = nullptr) {
    static_assert(!SC || TAPS == 9, "the shortcut rides with the 3x3 convolution");
    using C = W2Cfg<CIN, COUT, TAPS>;
    constexpr int CI4 = CIN / 4, CO4 = COUT / 4;
    constexpr int BPARTS = THREADS / COUT;
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    const int KR = (glo.Wp + 1) & ~1;                 // k rows per tile (even); row Wp_lo, if present, is a zero row
    const int span = 3 * ghi.Wp + 1;                  // 3 image rows + the border position that follows them (shared borders)
    const int nin = span * CI4;                       // float4 of the input span
    float *in_s = smem;                               // [span][CIN]
    float *do_s = in_s + span * CIN;                  // [KR][COUT] (+32 slack for the padded MFMA columns)
    float *bred_s = do_s + KR * COUT + 32;            // [BPARTS][COUT]
    float *do2_s = bred_s + THREADS;                  // SC: [KR][COUT] (+32 slack)
    const int mn = wave % C::MN;
    const int mt = mn / C::NT, nt = mn % C::NT;
    const int tap0 = wave / C::MN;

    f32x16 acc[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    int toff[C::TPW];  // LDS float offset of tap j relative to column 2*xo of input row 2*yo
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int tap = tap0 + j * C::TSTRIDE;
        // 3x3 pad 1: padded input (2yo + ky, 2xo + kx);  1x1 pad 0: padded input (2yo + 1, 2xo + 1)
        toff[j] = ((TAPS == 9 && tap < TAPS) ? (tap / 3) * ghi.Wp + (tap % 3) : ghi.Wp + 1) * CIN;
    }
    float bsum = 0.0f;
    const int bco = tid % COUT, bpart = tid / COUT;

    // the wave whose last slot is "tap 9": it owns the shortcut (toff of a slot past the 3x3 taps is the centre tap's)
    const bool sc_wave = SC && (tap0 + (C::TPW - 1) * C::TSTRIDE == TAPS);
    float4 pin[W2_PRE], pdo, pdo2 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int64_t tile) {
        const int64_t b = tile / Ho;
        const int yo = (int)(tile - b * Ho);
        const float4 *src = reinterpret_cast<const float4 *>(in + ((b * ghi.Hp + 2 * yo) * (int64_t)ghi.Wp) * CIN);
#pragma unroll
        for (int u = 0; u < W2_PRE; ++u) {
            const int f = u * THREADS + tid;
            pin[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < nin) pin[u] = src[f];
        }
        const float4 *dsrc = reinterpret_cast<const float4 *>(dout + ((b * glo.Hp + yo + 1) * (int64_t)glo.Wp) * COUT);
        pdo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < glo.Wp * CO4) pdo = dsrc[tid];
        if (SC) {
            const float4 *dsrc2 = reinterpret_cast<const float4 *>(dout_sc + ((b * glo.Hp + yo + 1) * (int64_t)glo.Wp) * COUT);
            pdo2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < glo.Wp * CO4) pdo2 = dsrc2[tid];
        }
    };
    const bool last_tap_live = tap0 + (C::TPW - 1) * C::TSTRIDE < TAPS;
    const float *b_base = do_s + nt * 32 + i;
    const float *a_base = in_s + mt * 32 + i;
    const float *b2_base = do2_s + nt * 32 + i;
    auto mfma_loop = [&](auto ntaps_c, auto with_sc_c) {
        constexpr int NTAPS = decltype(ntaps_c)::value;
        constexpr bool WITH_SC = decltype(with_sc_c)::value;   // slot NTAPS: centre-tap rows x the shortcut's output gradient
#pragma unroll 4
        for (int k = 0; k < KR; k += 2) {
            const int xpo = k + gk;
            const float b = b_base[xpo * COUT];
            // column 2*xo of the input row; clamped for the (zero) border / padding k rows so the read stays inside the span
            const int c0 = min(max(2 * (xpo - 1), 0), ghi.Wp - 2);
            const float *arow = a_base + c0 * CIN;
#pragma unroll
            for (int j = 0; j < NTAPS; ++j) acc[j] = mfma32(arow[toff[j]], b, acc[j]);
            if constexpr (WITH_SC) acc[NTAPS] = mfma32(arow[toff[NTAPS]], b2_base[xpo * COUT], acc[NTAPS]);
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();  // previous tile's readers are done
#pragma unroll
        for (int u = 0; u < W2_PRE; ++u) {
            const int f = u * THREADS + tid;
            if (f < nin) reinterpret_cast<float4 *>(in_s)[f] = pin[u];
        }
        if (tid < KR * CO4) reinterpret_cast<float4 *>(do_s)[tid] = pdo;  // rows >= Wp_lo were fetched as zero
        if (SC && tid < KR * CO4) reinterpret_cast<float4 *>(do2_s)[tid] = pdo2;
        __syncthreads();
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        if (bias_slabs != nullptr) {
            for (int r = bpart; r < glo.Wp; r += BPARTS) bsum += do_s[r * COUT + bco];
        }
        // the number of taps this wave owns is wave-uniform: choose the loop body once, so that the body itself is
        // branch-free and the compiler can batch the operand reads of several k steps ahead of their MFMAs
        constexpr int FEWER = C::TPW > 1 ? C::TPW - 1 : 1;
        if (last_tap_live) mfma_loop(std::integral_constant<int, C::TPW>{}, std::false_type{});
        else if (sc_wave) mfma_loop(std::integral_constant<int, FEWER>{}, std::integral_constant<bool, SC && (FEWER < C::TPW)>{});
        else mfma_loop(std::integral_constant<int, FEWER>{}, std::false_type{});
    }
    float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CIN * COUT);
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int tap = tap0 + j * C::TSTRIDE;
        if (tap < TAPS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = mt * 32 + acc_row(r, lane);
                const int co = nt * 32 + i;
                if (ci < CIN && co < COUT) slab[(tap * CIN + ci) * COUT + co] = acc[j][r];
            }
        }
    }
    if (sc_wave) {   // slab_sc[wg][ci][co], one tap
        float *slab2 = slabs_sc + (int64_t)blockIdx.x * (CIN * COUT);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = mt * 32 + acc_row(r, lane);
            const int co = nt * 32 + i;
            if (ci < CIN && co < COUT) slab2[ci * COUT + co] = acc[C::TPW - 1][r];
        }
    }
    if (bias_slabs != nullptr) {
        __syncthreads();
        bred_s[bpart * COUT + bco] = bsum;
        __syncthreads();
        if (tid < COUT) {
            float s = 0.0f;
            for (int p = 0; p < BPARTS; ++p) s += bred_s[p * COUT + tid];
            bias_slabs[(int64_t)blockIdx.x * COUT + tid] = s;
        }
    }
}

Geom mk(int64_t batch, int H, int W) { return make_geom(batch, H, W); }

template <int KC, int NC, int TAPS, bool SC = false, bool STAT = false>
int launch_dgrad(const float *dout, const float *wt, float *dx, int64_t batch, int H, int W, int accumulate, hipStream_t st,
                 const float *dout_sc = nullptr, const float *wt_sc = nullptr, S2Stat bst = S2Stat{nullptr, nullptr, nullptr, nullptr}) {
    const Geom ghi = mk(batch, H, W), glo = mk(batch, (H + 1) / 2, (W + 1) / 2);
    const int64_t biggest = batch * ((H + 1) / 2) * ((W + 1) / 2);  // class (0,0)
    if (ghi.rows >= (1ll << 31)) return lad::fail(LAD_ERR_INVALID, "dgrad_s2: %lld rows exceed the 32-bit row decode", (long long)ghi.rows);
    const dim3 grid((unsigned)lad::ceil_div(biggest, TM), TAPS == 9 ? 4 : 1);
    hipLaunchKernelGGL((dgrad_s2_kernel<KC, NC, TAPS, SC, STAT>), grid, dim3(THREADS), 0, st, dout, wt, dx, glo, ghi, H, W, batch, accumulate,
                       dout_sc, wt_sc, bst);
    return lad::check_launch("dgrad_s2_kernel");
}

template <int CIN, int COUT, int TAPS, bool SC = false>
int launch_wgrad(const float *in, const float *dout, float *ws, float *dw, float *dbias, int64_t batch, int H, int W, hipStream_t st,
                 const float *dout_sc = nullptr, float *dw_sc = nullptr) {
    const Geom ghi = mk(batch, H, W), glo = mk(batch, (H + 1) / 2, (W + 1) / 2);
    const int Ho = (H + 1) / 2;
    const int64_t n_tiles = batch * Ho;
    const int groups = (int)std::min<int64_t>(MAX_GROUPS, n_tiles);
    const int KR = (glo.Wp + 1) & ~1;
    if ((3 * ghi.Wp + 1) * (CIN / 4) > W2_PRE * THREADS || KR * (COUT / 4) > THREADS)
        return lad::fail(LAD_ERR_INVALID, "wgrad_s2: image too wide for the tile (W = %d)", W);
    const size_t lds = ((size_t)(3 * ghi.Wp + 1) * CIN + (size_t)KR * COUT + 32 + THREADS + (SC ? (size_t)KR * COUT + 32 : 0)) * sizeof(float);
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_s2_kernel<CIN, COUT, TAPS, SC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        attr_set = true;
    }
    float *bias_slabs = ws + (int64_t)MAX_GROUPS * TAPS * CIN * COUT;
    float *slabs_sc = bias_slabs + (int64_t)MAX_GROUPS * COUT;   // SC: [groups][CIN][COUT]
    hipLaunchKernelGGL((wgrad_s2_kernel<CIN, COUT, TAPS, SC>), dim3(groups), dim3(THREADS), lds, st, in, dout, ws,
                       dbias ? bias_slabs : nullptr, ghi, glo, n_tiles, Ho, dout_sc, slabs_sc);
    int rc = lad::check_launch("wgrad_s2_kernel");
    if (rc) return rc;
    rc = lad::reduce_slabs(lad::SlabReduce{ws, dbias ? bias_slabs : nullptr, dw, dbias, groups, CIN, COUT, TAPS}, st);
    if (rc || !SC) return rc;
    return lad::reduce_slabs(lad::SlabReduce{slabs_sc, nullptr, dw_sc, nullptr, groups, CIN, COUT, 1}, st);
}

}  // namespace

#define LAD_DG_CASE(CI, CO, T)                \
    if (cin == CI && cout == CO && taps == T) \
        return launch_dgrad<CO, CI, T>(dout, wt, dx, batch, H, W, accumulate, (hipStream_t)stream);

extern "C" int lad_conv_s2_dgrad(const float *dout, const float *wt, float *dx, int64_t batch, int32_t H, int32_t W, int32_t cin,
                                 int32_t cout, int32_t taps, int32_t accumulate, void *stream) {
    using namespace lad;
    LAD_REQUIRE(dout && wt && dx, "lad_conv_s2_dgrad: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_conv_s2_dgrad: bad geometry");
    LAD_REQUIRE(taps == 9 || accumulate, "lad_conv_s2_dgrad: the 1x1 shortcut writes one parity class only, so it accumulates");
    if (batch == 0) return LAD_OK;
    LAD_DG_CASE(64, 32, 9)
    LAD_DG_CASE(32, 16, 9)
    LAD_DG_CASE(16, 16, 9)
    LAD_DG_CASE(64, 32, 1)
    LAD_DG_CASE(32, 16, 1)
    LAD_DG_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_dgrad: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

extern "C" int64_t lad_conv_s2_wgrad_workspace_floats(int32_t cin, int32_t cout, int32_t taps) {
    if (cin <= 0 || cout <= 0 || (taps != 1 && taps != 9)) return -1;
    return (int64_t)MAX_GROUPS * ((int64_t)taps * cin * cout + cout);
}

#define LAD_WG2_CASE(CI, CO, T)               \
    if (cin == CI && cout == CO && taps == T) \
        return launch_wgrad<CI, CO, T>(in, dout, workspace, dw, dbias, batch, H, W, (hipStream_t)stream);

extern "C" int lad_conv_s2_wgrad(const float *in, const float *dout, float *workspace, float *dw, float *dbias, int64_t batch,
                                 int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_s2_wgrad: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_s2_wgrad: bad geometry");
    LAD_WG2_CASE(64, 32, 9)
    LAD_WG2_CASE(32, 16, 9)
    LAD_WG2_CASE(16, 16, 9)
    LAD_WG2_CASE(64, 32, 1)
    LAD_WG2_CASE(32, 16, 1)
    LAD_WG2_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_wgrad: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

// The 3x3 stride-2 convolution's weight gradient AND the 1x1 stride-2 shortcut's (same input, output gradients dout and
// dout_sc) in one launch: see wgrad_s2_kernel<.., SC>.  Bit-identical to the two separate launches.
extern "C" int64_t lad_conv_s2_wgrad_fused_workspace_floats(int32_t cin, int32_t cout) {
    if (cin <= 0 || cout <= 0) return -1;
    return (int64_t)MAX_GROUPS * ((int64_t)9 * cin * cout + cout + (int64_t)cin * cout);
}

#define LAD_WG2F_CASE(CI, CO)      \
    if (cin == CI && cout == CO) \
        return launch_wgrad<CI, CO, 9, true>(in, dout, workspace, dw, dbias, batch, H, W, (hipStream_t)stream, dout_sc, dw_sc);

extern "C" int lad_conv_s2_wgrad_fused(const float *in, const float *dout, const float *dout_sc, float *workspace, float *dw,
                                       float *dbias, float *dw_sc, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout,
                                       void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && dout_sc && workspace && dw && dw_sc, "lad_conv_s2_wgrad_fused: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_s2_wgrad_fused: bad geometry");
    LAD_WG2F_CASE(64, 32)
    LAD_WG2F_CASE(32, 16)
    LAD_WG2F_CASE(16, 16)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_wgrad_fused: unsupported (cin=%d, cout=%d)", cin, cout);
}

// Data gradient of a stride-2 block's input: 3x3 convolution (dout, wt) + 1x1 shortcut (dout_sc, wt_sc), one launch, dx
// written once (not accumulated).  cin / cout name the convolutions' channels, as in lad_conv_s2_dgrad.
#define LAD_DGF_CASE(CI, CO)     \
    if (cin == CI && cout == CO) \
        return launch_dgrad<CO, CI, 9, true>(dout, wt, dx, batch, H, W, 0, (hipStream_t)stream, dout_sc, wt_sc);

extern "C" int lad_conv_s2_dgrad_fused(const float *dout, const float *wt, const float *dout_sc, const float *wt_sc, float *dx,
                                       int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(dout && wt && dout_sc && wt_sc && dx, "lad_conv_s2_dgrad_fused: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_s2_dgrad_fused: bad geometry");
    LAD_DGF_CASE(64, 32)
    LAD_DGF_CASE(32, 16)
    LAD_DGF_CASE(16, 16)
    return fail(LAD_ERR_INVALID, "lad_conv_s2_dgrad_fused: unsupported (cin=%d, cout=%d)", cin, cout);
}

// lad_conv_s2_dgrad_fused for the 64 <- 32 transition with the sums of the BatchNorm that consumes dx (the bn2 of the
// 64-channel block below; ReLU decisions from its sign bits): stat_partials float[lad_conv_s2_dgrad_partials(...)][2][64],
// to be handed to lad_bn_bwd_bits as pre_partials / pre_tiles.
extern "C" int64_t lad_conv_s2_dgrad_partials(int64_t batch, int32_t H, int32_t W) {
    if (batch < 1 || H < 1 || W < 1) return -1;
    return 4 * lad::ceil_div(batch * ((H + 1) / 2) * ((W + 1) / 2), TM);
}

extern "C" int lad_conv_s2_dgrad_fused_bnstat(const float *dout, const float *wt, const float *dout_sc, const float *wt_sc, float *dx,
                                              float *stat_partials, const float *bn_x, const uint64_t *bn_bits, const float *bn_coef,
                                              int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(dout && wt && dout_sc && wt_sc && dx && stat_partials && bn_x && bn_bits && bn_coef, "lad_conv_s2_dgrad_fused_bnstat: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_s2_dgrad_fused_bnstat: bad geometry");
    LAD_REQUIRE(cin == 64 && cout == 32, "lad_conv_s2_dgrad_fused_bnstat: the 64 <- 32 transition only (got cin=%d, cout=%d)", cin, cout);
    return launch_dgrad<32, 64, 9, true, true>(dout, wt, dx, batch, H, W, 0, (hipStream_t)stream, dout_sc, wt_sc,
                                               S2Stat{bn_x, (const unsigned long long *)bn_bits, bn_coef, stat_partials});
}
