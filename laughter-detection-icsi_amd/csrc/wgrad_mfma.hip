// Weight-gradient of the 3x3 / 1x1 convolutions on the f32 matrix cores (gfx950).
//
// Replaces the weight/bias gradient autograd computes for nn.Conv2d in loss.backward() (train.py:289).
//   dW[tap][ci][co] = sum over rows q of  in[q + shift(tap)][ci] * dout[q][co]      (layout: lad_device.h)
// GEMM view: M = ci, N = co, K = rows of the whole batch (millions) -> split-K over persistent workgroups.
// A workgroup walks tiles of TMW rows; per tile it stages the input rows (+halo) and the dout rows into
// LDS once (border rows zeroed) and every wavefront accumulates its share of the 9*MT*NT 32x32 output
// tiles in registers across ALL its tiles.  At the end each workgroup writes one partial slab; a second
// small kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics) and emits the
// gradient in the reference's (cout, cin, kh, kw) parameter layout, plus the bias gradient (column sums of
// dout, accumulated on the side while the tile is in LDS).
#include "lad_common.h"
#include "lad_device.h"

namespace {
using namespace lad;

constexpr int THREADS = 256;
constexpr int TMW = 64;        // rows per tile
constexpr int MAX_GROUPS = 512;  // persistent workgroups (2 per CU)

template <int CIN, int COUT, int TAPS>
struct WgCfg {
    static constexpr int MT = (CIN + 31) / 32;
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int MN = MT * NT;          // 1, 2 or 4 distinct (mt, nt) pairs
    static constexpr int TSTRIDE = 4 / MN;      // wavefronts that share one (mt, nt) split the taps
    static constexpr int TPW = (TAPS + TSTRIDE - 1) / TSTRIDE;  // accumulator tiles per wavefront
};

template <int CIN, int COUT, int TAPS>
__global__ __launch_bounds__(THREADS) void wgrad_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                        float *__restrict__ slabs, float *__restrict__ bias_slabs,
                                                        Geom g, int64_t n_tiles) {
    using C = WgCfg<CIN, COUT, TAPS>;
    constexpr int CI4 = CIN / 4, CO4 = COUT / 4;
    constexpr int BPARTS = THREADS / COUT;
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMW + 2 * halo;
    float *in_s = smem;                                  // [nrows][CIN] (+32 slack)
    float *do_s = in_s + nrows * CIN + 32;               // [TMW][COUT]  (+32 slack)
    float *mask_s = do_s + TMW * COUT + 32;              // [nrows]
    float *bred_s = mask_s + ((nrows + 3) & ~3);         // [BPARTS][COUT]

    const int mn = wave % C::MN;
    const int mt = mn / C::NT, nt = mn % C::NT;
    const int tap0 = wave / C::MN;

    f32x16 acc[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    int aoff[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int tap = tap0 + j * C::TSTRIDE;
        const int sh = (TAPS == 9 && tap < TAPS) ? ((tap / 3 - 1) * g.Wp + (tap % 3 - 1)) : 0;
        aoff[j] = (gk + halo + sh) * CIN + mt * 32 + i;
    }
    const int boff = gk * COUT + nt * 32 + i;
    float bsum = 0.0f;
    const int bco = tid % COUT, bpart = tid / COUT;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t q0 = tile * TMW;
        __syncthreads();  // previous tile's readers are done
        for (int j = tid; j < nrows; j += THREADS) mask_s[j] = interior_row(q0 - halo + j, g) ? 1.0f : 0.0f;
        __syncthreads();
        const float *src = in + (q0 - halo) * CIN;
        for (int f = tid; f < nrows * CI4; f += THREADS) {
            const int row = f / CI4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mask_s[row] != 0.0f) v = reinterpret_cast<const float4 *>(src)[f];
            reinterpret_cast<float4 *>(in_s)[f] = v;
        }
        const float *dsrc = dout + q0 * COUT;
        for (int f = tid; f < TMW * CO4; f += THREADS) {
            const int row = f / CO4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mask_s[halo + row] != 0.0f) v = reinterpret_cast<const float4 *>(dsrc)[f];
            reinterpret_cast<float4 *>(do_s)[f] = v;
        }
        __syncthreads();
        if (bias_slabs != nullptr) {
#pragma unroll 4
            for (int r = bpart; r < TMW; r += BPARTS) bsum += do_s[r * COUT + bco];
        }
#pragma unroll 2
        for (int k = 0; k < TMW; k += 2) {
            const float b = do_s[k * COUT + boff];
#pragma unroll
            for (int j = 0; j < C::TPW; ++j) {
                if (tap0 + j * C::TSTRIDE < TAPS) {
                    const float a = in_s[k * CIN + aoff[j]];
                    acc[j] = mfma32(a, b, acc[j]);
                }
            }
        }
    }

    // ---- write this workgroup's partial slab: slab[wg][tap][ci][co] ------------------------------
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

// dw[co][ci][tap] = sum_wg slab[wg][tap][ci][co];  dbias[co] = sum_wg bias_slab[wg][co]
__global__ void wgrad_reduce_kernel(const float *__restrict__ slabs, const float *__restrict__ bias_slabs,
                                    float *__restrict__ dw, float *__restrict__ dbias, int groups, int cin, int cout,
                                    int taps) {
    const int n = taps * cin * cout;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) {
        double s = 0.0;
        for (int w = 0; w < groups; ++w) s += (double)slabs[(int64_t)w * n + idx];
        const int co = idx % cout;
        const int t = idx / cout;
        const int ci = t % cin, tap = t / cin;
        dw[((int64_t)co * cin + ci) * taps + tap] = (float)s;
    } else if (dbias != nullptr && idx < n + cout) {
        const int co = idx - n;
        double s = 0.0;
        for (int w = 0; w < groups; ++w) s += (double)bias_slabs[(int64_t)w * cout + co];
        dbias[co] = (float)s;
    }
}

int groups_for(int64_t n_tiles) { return (int)std::min<int64_t>(MAX_GROUPS, n_tiles); }

template <int CIN, int COUT, int TAPS>
int launch_wgrad(const float *in, const float *dout, float *ws, float *dw, float *dbias, const Geom &g, hipStream_t st) {
    const int64_t n_tiles = lad::ceil_div(g.rows, TMW);
    const int groups = groups_for(n_tiles);
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMW + 2 * halo;
    const size_t lds = ((size_t)nrows * CIN + 32 + TMW * COUT + 32 + ((nrows + 3) & ~3) + THREADS) * sizeof(float);
    if (lds > 160 * 1024) return lad::fail(LAD_ERR_INVALID, "wgrad: image too wide for the LDS tile (%zu B)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_kernel<CIN, COUT, TAPS>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    float *slabs = ws;
    float *bias_slabs = ws + (int64_t)MAX_GROUPS * TAPS * CIN * COUT;
    hipLaunchKernelGGL((wgrad_kernel<CIN, COUT, TAPS>), dim3(groups), dim3(THREADS), lds, st, in, dout, slabs,
                       dbias ? bias_slabs : nullptr, g, n_tiles);
    int rc = lad::check_launch("wgrad_kernel");
    if (rc) return rc;
    const int n = TAPS * CIN * COUT + (dbias ? COUT : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)lad::ceil_div(n, 256)), dim3(256), 0, st, slabs, bias_slabs,
                       dw, dbias, groups, CIN, COUT, TAPS);
    return lad::check_launch("wgrad_reduce_kernel");
}

}  // namespace

extern "C" int64_t lad_conv_wgrad_workspace_floats(int32_t cin, int32_t cout, int32_t taps) {
    if (cin <= 0 || cout <= 0 || (taps != 1 && taps != 9)) return -1;
    return (int64_t)MAX_GROUPS * ((int64_t)taps * cin * cout + cout);
}

#define LAD_WG_CASE(CI, CO, T)                  \
    if (cin == CI && cout == CO && taps == T)   \
        return launch_wgrad<CI, CO, T>(in, dout, workspace, dw, dbias, g, (hipStream_t)stream);

extern "C" int lad_conv_wgrad(const float *in, const float *dout, float *workspace, float *dw, float *dbias,
                              int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                              void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_wgrad: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad: bad geometry");
    Geom g;
    g.Hp = H + 2;
    g.Wp = W + 2;
    g.img = g.Hp * g.Wp;
    g.rows = batch * g.img;
    LAD_WG_CASE(64, 64, 9)
    LAD_WG_CASE(32, 32, 9)
    LAD_WG_CASE(16, 16, 9)
    LAD_WG_CASE(64, 32, 9)
    LAD_WG_CASE(32, 16, 9)
    LAD_WG_CASE(64, 32, 1)
    LAD_WG_CASE(32, 16, 1)
    LAD_WG_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_wgrad: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}
