// Implicit-GEMM convolutions on the f32 matrix cores (v_mfma_f32_32x32x2_f32) for gfx950.
//
// Replaces the nn.Conv2d calls of the reference's ResidualBlock / ResNetBigger (models.py:86-106,
// 110-115, 186-189) in forward and in the data-gradient direction.  Layout and the "row-shifted GEMM"
// formulation: lad_device.h.  GEMM view of one launch: M = rows (spatial positions of the whole batch),
// N = output channels, K = taps * input channels.
//
// conv_s1_kernel  stride 1 (3x3 pad 1, or 1x1): a workgroup owns 128 consecutive output rows and all
//                 output channels; the input rows it needs (128 + a halo of W+3 rows either side) are one
//                 contiguous span of HBM, staged once into LDS with 16-byte coalesced loads and border rows
//                 zeroed; every wavefront then runs 32 rows x COUT through 9 * CIN/2 MFMAs.  A operands are
//                 ds_read_b128 (4 consecutive input channels feed 4 MFMAs), B operands come from a weight
//                 image pre-packed as [tap][CIN/4][COUT][4] so that a lane's 16-byte load is exactly its
//                 four K-slices; the image is small (<=147 KB), shared by every workgroup and L2 resident.
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

// Shared epilogue: acc[n][r] holds out[row = acc_row(r)][co = n*32 + (lane&31)] of this wave's 32 rows.
template <int COUT>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[NTiles<COUT>::NT], const float *__restrict__ bias,
                                              const float *__restrict__ addend, float *__restrict__ out,
                                              float *__restrict__ partials, const float *mask_tile /*[TM]*/,
                                              float *red_s /*[4][2][COUTP]*/, int64_t q0, int64_t rows) {
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int COUTP = NTiles<COUT>::COUTP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        const bool co_ok = co < COUT;
        const float bv = (bias != nullptr && co_ok) ? bias[co] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + acc_row(r, lane);
            const int64_t q = q0 + row;
            const bool keep = mask_tile[row] != 0.0f;
            float v = acc[n][r] + bv;
            if (co_ok && q < rows) {
                if (addend != nullptr) v += addend[q * COUT + co];
                v = keep ? v : 0.0f;
                out[q * COUT + co] = v;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
        }
        if (partials != nullptr) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                red_s[(wave * 2 + 0) * COUTP + co] = s1;
                red_s[(wave * 2 + 1) * COUTP + co] = s2;
            }
        }
    }
    if (partials != nullptr) {
        __syncthreads();
        if (tid < 2 * COUT) {
            const int k = tid / COUT, co = tid - k * COUT;
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red_s[(w * 2 + k) * COUTP + co];
            partials[((int64_t)blockIdx.x * 2 + k) * COUT + co] = s;
        }
    }
}

template <int CIN, int COUT, int TAPS>
__global__ __launch_bounds__(THREADS, 2) void conv_s1_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ wt,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ addend,
                                                             float *__restrict__ out, float *__restrict__ partials,
                                                             Geom g) {
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int COUTP = NTiles<COUT>::COUTP;
    constexpr int LDA = CIN + 4;  // padded LDS row: conflict-free ds_read_b128 across 32 rows
    constexpr int C4 = CIN / 4;
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TM + 2 * halo;
    float *a_s = smem;                             // [nrows][LDA]
    float *mask_s = a_s + nrows * LDA;             // [nrows]
    float *red_s = mask_s + ((nrows + 3) & ~3);    // [4][2][COUTP]
    const int64_t q0 = (int64_t)blockIdx.x * TM;

    for (int j = tid; j < nrows; j += THREADS) mask_s[j] = interior_row(q0 - halo + j, g) ? 1.0f : 0.0f;
    __syncthreads();
    const float *src = in + (q0 - halo) * CIN;
    for (int f = tid; f < nrows * C4; f += THREADS) {
        const int row = f / C4, c4 = f - row * C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask_s[row] != 0.0f) v = *reinterpret_cast<const float4 *>(src + (int64_t)row * CIN + c4 * 4);
        *reinterpret_cast<float4 *>(a_s + row * LDA + c4 * 4) = v;
    }
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int i = lane & 31, gk = lane >> 5;
    const float *a_base = a_s + (wave * 32 + i + halo) * LDA + 4 * gk;
    const float *w_base = wt + (gk * COUTP + i) * 4;
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int off = (TAPS == 9) ? ((tap / 3 - 1) * g.Wp + (tap % 3 - 1)) : 0;
        const float *ap = a_base + off * LDA;
        const float *wp = w_base + tap * (C4 * COUTP * 4);
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) {
            const float4 a = *reinterpret_cast<const float4 *>(ap + c8 * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b = *reinterpret_cast<const float4 *>(wp + (c8 * 2 * COUTP + n * 32) * 4);
                acc[n] = mfma32(a.x, b.x, acc[n]);
                acc[n] = mfma32(a.y, b.y, acc[n]);
                acc[n] = mfma32(a.z, b.z, acc[n]);
                acc[n] = mfma32(a.w, b.w, acc[n]);
            }
        }
    }
    conv_epilogue<COUT>(acc, bias, addend, out, partials, mask_s + halo, red_s, q0, g.rows);
}

template <int CIN, int COUT, int TAPS>
__global__ __launch_bounds__(THREADS, 2) void conv_s2_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ wt,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out, float *__restrict__ partials,
                                                             Geom gi, Geom go) {
    constexpr int NT = NTiles<COUT>::NT;
    constexpr int COUTP = NTiles<COUT>::COUTP;
    constexpr int C4 = CIN / 4;
    __shared__ float mask_s[TM];
    __shared__ float red_s[4 * 2 * COUTP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
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

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const float *w_base = wt + (gk * COUTP + i) * 4;
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        const int ky = (TAPS == 9) ? tap / 3 : 1, kx = (TAPS == 9) ? tap % 3 : 1;
        const int ypi = 2 * yo + ky, xpi = 2 * xo + kx;
        const bool valid = inter && ypi >= 1 && ypi <= gi.Hp - 2 && xpi >= 1 && xpi <= gi.Wp - 2;
        const float *ap = in + (base_row + (int64_t)ky * gi.Wp + kx) * CIN + 4 * gk;
        const float *wp = w_base + tap * (C4 * COUTP * 4);
#pragma unroll
        for (int c8 = 0; c8 < CIN / 8; ++c8) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) a = *reinterpret_cast<const float4 *>(ap + c8 * 8);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 b = *reinterpret_cast<const float4 *>(wp + (c8 * 2 * COUTP + n * 32) * 4);
                acc[n] = mfma32(a.x, b.x, acc[n]);
                acc[n] = mfma32(a.y, b.y, acc[n]);
                acc[n] = mfma32(a.z, b.z, acc[n]);
                acc[n] = mfma32(a.w, b.w, acc[n]);
            }
        }
    }
    __syncthreads();
    conv_epilogue<COUT>(acc, bias, nullptr, out, partials, mask_s, red_s, q0, go.rows);
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
        if ((yp & 1) && (xp & 1)) {
            const int yo = (yp - 1) >> 1, xo = (xp - 1) >> 1;
            if (yo <= gs.Hp - 3 && xo <= gs.Wp - 3) {
                const int64_t qs = b * gs.img + (int64_t)(yo + 1) * gs.Wp + (xo + 1);
                v = reinterpret_cast<const float4 *>(src)[qs * c4n + c4];
            }
        }
        reinterpret_cast<float4 *>(up)[idx] = v;
    }
}

Geom make_geom(int64_t batch, int H, int W) {
    Geom g;
    g.Hp = H + 2;
    g.Wp = W + 2;
    g.img = g.Hp * g.Wp;
    g.rows = batch * g.img;
    return g;
}

template <int CIN, int COUT, int TAPS>
int launch_s1(const float *in, const float *wt, const float *bias, const float *addend, float *out, float *partials,
              const Geom &g, hipStream_t st) {
    constexpr int COUTP = NTiles<COUT>::COUTP;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TM + 2 * halo;
    const size_t lds = ((size_t)nrows * (CIN + 4) + ((nrows + 3) & ~3) + 8 * COUTP) * sizeof(float);
    if (lds > 160 * 1024) return lad::fail(LAD_ERR_INVALID, "conv_s1: image too wide for the LDS tile (%zu B)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_s1_kernel<CIN, COUT, TAPS>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const unsigned grid = (unsigned)lad::ceil_div(g.rows, TM);
    hipLaunchKernelGGL((conv_s1_kernel<CIN, COUT, TAPS>), dim3(grid), dim3(THREADS), lds, st, in, wt, bias, addend, out,
                       partials, g);
    return lad::check_launch("conv_s1_kernel");
}

template <int CIN, int COUT, int TAPS>
int launch_s2(const float *in, const float *wt, const float *bias, float *out, float *partials, const Geom &gi,
              const Geom &go, hipStream_t st) {
    const unsigned grid = (unsigned)lad::ceil_div(go.rows, TM);
    hipLaunchKernelGGL((conv_s2_kernel<CIN, COUT, TAPS>), dim3(grid), dim3(THREADS), 0, st, in, wt, bias, out, partials,
                       gi, go);
    return lad::check_launch("conv_s2_kernel");
}

}  // namespace

extern "C" int64_t lad_conv_num_tiles(int64_t batch, int32_t H, int32_t W) {
    if (batch < 0 || H < 1 || W < 1) return -1;
    return lad::ceil_div(batch * (int64_t)(H + 2) * (W + 2), TM);
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
