// Stem convolution 3x3, 1 -> COUT channels (models.py:186-189, 224: conv1, no bias), forward and weight gradient.
//
// K = 9 is far too small for the matrix cores to matter and the op is bound by the HBM write of its
// 64-channel output, so this is a direct convolution: a thread owns 4 output channels (36 weights in
// registers) and walks rows; 16 threads cover the 64 channels of one row with one coalesced 256-byte store.
// Input is the un-padded feature map (B, H, W) exactly as the reference passes it (B,1,100,44); output is
// PNHWC (lad_device.h) with border rows written as zero, plus the per-tile BatchNorm partial sums.
#include "lad_common.h"
#include "lad_device.h"
#include "lad_stem_taps.h"
#include "lad_bn_math.h"

namespace {
using namespace lad;

constexpr int THREADS = 256;
constexpr int TM = lad::STEM_TM;
constexpr int COUT = 64;
constexpr int CQ = COUT / 4;          // 16 channel quads
constexpr int RL = THREADS / CQ;      // 16 row lanes
constexpr int MAX_GROUPS = 1536;  // persistent workgroups of the weight-gradient pass: 6 per CU (80 VGPRs, 11 KB LDS)

template <bool STORE>  // false: statistics only (the training path recomputes the convolution where it needs it)
__global__ __launch_bounds__(THREADS) void stem_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ w /*[64][9]*/,
                                                           float *__restrict__ out, float *__restrict__ partials, Geom g,
                                                           int H, int W) {
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const int64_t q0 = (int64_t)blockIdx.x * TM;
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    fill_taps(feat, g, H, W, q0, H, (int64_t)1 << 62, tap_s);
    __syncthreads();
    for (int r = rl; r < TM; r += RL) {
        const int64_t q = q0 + r;
        if (q >= g.rows) break;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        float v[9];
        if (read_taps(tap_s, r, v)) {
            float acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[t], wr[c][t], acc[c]);
            o = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += acc[c];
                s2[c] = fmaf(acc[c], acc[c], s2[c]);
            }
        }
        if (STORE) *reinterpret_cast<float4 *>(out + q * COUT + cq * 4) = o;
    }
    if (partials != nullptr) {
        __shared__ float red[RL][2][COUT];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[rl][0][cq * 4 + c] = s1[c];
            red[rl][1][cq * 4 + c] = s2[c];
        }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int k = tid / COUT, c = tid - k * COUT;
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < RL; ++p) s += red[p][k][c];
            partials[((int64_t)blockIdx.x * 2 + k) * COUT + c] = s;
        }
    }
}

// Eval-mode stem with bn1 (running statistics) + ReLU folded in, reading images as WINDOWS of a (frames, W) feature
// matrix: image b = frames [b*frame_stride, b*frame_stride + H); frames >= frames_avail read as 0.0 (the zero right-pad
// of datasets.py:86-93).  frame_stride = H gives ordinary back-to-back images, frame_stride = 1 the stride-one-frame
// sliding windows of InferenceDataset -- straight out of the feature matrix, no (T,100,44) materialisation.
__global__ __launch_bounds__(THREADS) void stem_fwd_eval_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                                const float *__restrict__ scale, const float *__restrict__ shift,
                                                                float *__restrict__ out, Geom g, int H, int W,
                                                                int64_t frame_stride, int64_t frames_avail) {
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    const float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
    const int64_t q0 = (int64_t)blockIdx.x * TM;
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    fill_taps(feat, g, H, W, q0, frame_stride, frames_avail, tap_s);
    __syncthreads();
    for (int r = rl; r < TM; r += RL) {
        const int64_t q = q0 + r;
        if (q >= g.rows) break;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        float v[9];
        if (read_taps(tap_s, r, v)) {
            float acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[t], wr[c][t], acc[c]);
            o = make_float4(fmaxf(fmaf(acc[0], sc.x, sh.x), 0.f), fmaxf(fmaf(acc[1], sc.y, sh.y), 0.f),
                            fmaxf(fmaf(acc[2], sc.z, sh.z), 0.f), fmaxf(fmaf(acc[3], sc.w, sh.w), 0.f));
        }
        *reinterpret_cast<float4 *>(out + q * COUT + cq * 4) = o;
    }
}

// dW[co][tap] = sum over interior rows of feat(row, tap) * dout[row][co]; slab[wg][co*9 + tap]
// MODE >= 1: `dout` is the gradient wrt the stem BatchNorm's OUTPUT (after its ReLU) and the BatchNorm backward is applied
// on the fly from x (the convolution output kept by the forward pass), coef and bcoef (lad_bn_bwd with dx = NULL): the
// stem needs no data gradient, so dz is consumed here and never written (one 596 MB write and two reads less per step).
// MODE 2: x is not read either but recomputed from the taps and the stem weights w (the same fmaf chain as the forward).
template <int MODE>
__global__ __launch_bounds__(THREADS) void stem_wgrad_kernel(const float *__restrict__ feat, const float *__restrict__ dout,
                                                             const float *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ coef,
                                                             const float *__restrict__ bcoef, float *__restrict__ slabs,
                                                             Geom g, int H, int W, int64_t n_tiles) {
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
    float acc[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    float4 fsc = make_float4(0.f, 0.f, 0.f, 0.f), fsh = fsc, k1 = fsc, k2 = fsc, k3 = fsc, k2l = fsc, k3l = fsc;
    Norm4 nm{fsc, fsc, fsc, fsc};
    constexpr bool BN = MODE >= 1;
    float wr[4][9];
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    }
    if (BN) {  // per-channel coefficients of this thread's channel quad (bn.hip: coef float[6][C], bcoef float[8][C])
        const int c = cq * 4;
        fsc = *reinterpret_cast<const float4 *>(coef + c);
        fsh = *reinterpret_cast<const float4 *>(coef + COUT + c);
        nm = load_norm(coef, COUT, c);
        k1 = *reinterpret_cast<const float4 *>(bcoef + 0 * COUT + c);
        k2 = *reinterpret_cast<const float4 *>(bcoef + 1 * COUT + c);
        k3 = *reinterpret_cast<const float4 *>(bcoef + 2 * COUT + c);
        k2l = *reinterpret_cast<const float4 *>(bcoef + 4 * COUT + c);
        k3l = *reinterpret_cast<const float4 *>(bcoef + 6 * COUT + c);
    }
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    constexpr int NR = TM / RL;   // rows of a tile per thread
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t q0 = tile * TM;
        // this thread's NR gradient rows are requested before the tap table is built: with one 16-byte load in flight per
        // thread the pass ran at 3.5 TB/s (latency x occupancy), not at the rate of the stream
        float4 dv[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int64_t q = q0 + rl + k * RL;
            dv[k] = q < g.rows ? *reinterpret_cast<const float4 *>(dout + q * COUT + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();  // the previous tile's table has been read
        fill_taps(feat, g, H, W, q0, H, (int64_t)1 << 62, tap_s);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rl + k * RL;
            const int64_t q = q0 + r;
            if (q >= g.rows) break;
            float v[9];
            if (read_taps(tap_s, r, v)) {
                float4 d = dv[k];
                if (BN) {
                    float4 xv;
                    if (MODE == 2) {
                        float xa[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int t = 0; t < 9; ++t)
#pragma unroll
                            for (int c = 0; c < 4; ++c) xa[c] = fmaf(v[t], wr[c][t], xa[c]);
                        xv = make_float4(xa[0], xa[1], xa[2], xa[3]);
                    } else {
                        xv = *reinterpret_cast<const float4 *>(x + q * COUT + cq * 4);
                    }
                    d = mask_from_x(d, xv, fsc, fsh);
                    const float4 xh = xhat4(xv, nm);
                    d = make_float4(bn_dx1(d.x, xh.x, k1.x, k2.x, k2l.x, k3.x, k3l.x), bn_dx1(d.y, xh.y, k1.y, k2.y, k2l.y, k3.y, k3l.y),
                                    bn_dx1(d.z, xh.z, k1.z, k2.z, k2l.z, k3.z, k3l.z), bn_dx1(d.w, xh.w, k1.w, k2.w, k2l.w, k3.w, k3l.w));
                }
                const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(v[t], dd[c], acc[c][t]);
            }
        }
    }
    __shared__ float red[RL][COUT * 9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[rl][(cq * 4 + c) * 9 + t] = acc[c][t];
    __syncthreads();
    for (int e = tid; e < COUT * 9; e += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < RL; ++p) s += red[p][e];
        slabs[(int64_t)blockIdx.x * (COUT * 9) + e] = s;
    }
}

// First pass of the stem BatchNorm's backward with x recomputed: per workgroup (sum dz, sum dz * xhat) per channel,
// dz = dy * [x*scale + shift > 0]; the rows of lad_bn_bwd's pre_partials.
__global__ __launch_bounds__(THREADS) void stem_bn_sums_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                               const float *__restrict__ dy, const float *__restrict__ coef,
                                                               float *__restrict__ partials, Geom g, int H, int W, int64_t n_tiles) {
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    const float4 fsc = *reinterpret_cast<const float4 *>(coef + cq * 4);
    const float4 fsh = *reinterpret_cast<const float4 *>(coef + COUT + cq * 4);
    const Norm4 nm = load_norm(coef, COUT, cq * 4);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    constexpr int NR = TM / RL;   // rows of a tile per thread
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t q0 = tile * TM;
        float4 dv[NR];   // requested before the tap table is built (see stem_wgrad_kernel)
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int64_t q = q0 + rl + k * RL;
            dv[k] = q < g.rows ? *reinterpret_cast<const float4 *>(dy + q * COUT + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        fill_taps(feat, g, H, W, q0, H, (int64_t)1 << 62, tap_s);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rl + k * RL;
            const int64_t q = q0 + r;
            if (q >= g.rows) break;
            float v[9];
            if (read_taps(tap_s, r, v)) {
                float xa[4] = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xa[c] = fmaf(v[t], wr[c][t], xa[c]);
                const float4 xv = make_float4(xa[0], xa[1], xa[2], xa[3]);
                const float4 d = mask_from_x(dv[k], xv, fsc, fsh);
                const float4 xh = xhat4(xv, nm);
                a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
                a1.x = fmaf(d.x, xh.x, a1.x); a1.y = fmaf(d.y, xh.y, a1.y); a1.z = fmaf(d.z, xh.z, a1.z); a1.w = fmaf(d.w, xh.w, a1.w);
            }
        }
    }
    __shared__ float red[RL][2][COUT];
    *reinterpret_cast<float4 *>(&red[rl][0][cq * 4]) = a0;
    *reinterpret_cast<float4 *>(&red[rl][1][cq * 4]) = a1;
    __syncthreads();
    if (tid < 2 * COUT) {
        const int k = tid / COUT, c = tid - k * COUT;
        float sacc = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < RL; ++p2) sacc += red[p2][k][c];
        partials[((int64_t)blockIdx.x * 2 + k) * COUT + c] = sacc;
    }
}

// out[idx] = sum over workgroups of slabs[wg][idx]: a block owns 64 outputs, its 16 wavefronts each take a sixteenth of
// the slabs, four loads in flight (fixed order, double accumulation: bit-reproducible)
constexpr int CS_PARTS = 16;
__global__ __launch_bounds__(64 * CS_PARTS) void colsum_kernel(const float *__restrict__ slabs, float *__restrict__ out, int groups, int n) {
    const int o = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    __shared__ double red[CS_PARTS][64];
    double s = 0.0;
    if (idx < n) {
        const float *src = slabs + idx;
        int w = part;
        for (; w + 3 * CS_PARTS < groups; w += 4 * CS_PARTS) {
            const float a = src[(int64_t)w * n], b = src[(int64_t)(w + CS_PARTS) * n], c = src[(int64_t)(w + 2 * CS_PARTS) * n],
                        d = src[(int64_t)(w + 3 * CS_PARTS) * n];
            s += (double)a; s += (double)b; s += (double)c; s += (double)d;
        }
        for (; w < groups; w += CS_PARTS) s += (double)src[(int64_t)w * n];
    }
    red[part][o] = s;
    __syncthreads();
    if (part == 0 && idx < n) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < CS_PARTS; ++p) t += red[p][o];
        out[idx] = (float)t;
    }
}

}  // namespace

extern "C" int lad_stem_fwd(const float *feat, const float *weight, float *out, float *stat_partials, int64_t batch,
                            int32_t H, int32_t W, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && (out || stat_partials), "lad_stem_fwd: null buffer");
    LAD_REQUIRE(cout == COUT, "lad_stem_fwd: cout must be %d", COUT);
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_stem_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = make_geom(batch, H, W);
    if (out != nullptr)
        hipLaunchKernelGGL(stem_fwd_kernel<true>, dim3((unsigned)ceil_div(g.rows, TM)), dim3(THREADS), 0, (hipStream_t)stream, feat,
                           weight, out, stat_partials, g, H, W);
    else
        hipLaunchKernelGGL(stem_fwd_kernel<false>, dim3((unsigned)ceil_div(g.rows, TM)), dim3(THREADS), 0, (hipStream_t)stream, feat,
                           weight, out, stat_partials, g, H, W);
    return check_launch("stem_fwd_kernel");
}

extern "C" int lad_stem_fwd_eval(const float *feat, const float *weight, const float *scale, const float *shift, float *out,
                                 int64_t batch, int32_t H, int32_t W, int32_t cout, int64_t frame_stride, int64_t frames_avail,
                                 void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && scale && shift && out, "lad_stem_fwd_eval: null buffer");
    LAD_REQUIRE(cout == COUT, "lad_stem_fwd_eval: cout must be %d", COUT);
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1 && frame_stride >= 1 && frames_avail >= 0, "lad_stem_fwd_eval: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = make_geom(batch, H, W);
    hipLaunchKernelGGL(stem_fwd_eval_kernel, dim3((unsigned)ceil_div(g.rows, TM)), dim3(THREADS), 0, (hipStream_t)stream, feat,
                       weight, scale, shift, out, g, H, W, frame_stride, frames_avail);
    return check_launch("stem_fwd_eval_kernel");
}

extern "C" int64_t lad_stem_wgrad_workspace_floats(void) { return (int64_t)MAX_GROUPS * COUT * 9; }

static int stem_wgrad_launch(const float *feat, const float *dout, const float *x, const float *weight, const float *coef,
                             const float *bcoef, float *workspace, float *dw, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    const Geom g = make_geom(batch, H, W);
    const int64_t n_tiles = ceil_div(g.rows, TM);
    const int groups = (int)std::min<int64_t>(MAX_GROUPS, n_tiles);
    const dim3 grid(groups), block(THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (coef == nullptr)
        hipLaunchKernelGGL(stem_wgrad_kernel<0>, grid, block, 0, st, feat, dout, nullptr, nullptr, nullptr, nullptr, workspace, g, H, W, n_tiles);
    else if (x != nullptr)
        hipLaunchKernelGGL(stem_wgrad_kernel<1>, grid, block, 0, st, feat, dout, x, nullptr, coef, bcoef, workspace, g, H, W, n_tiles);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<2>, grid, block, 0, st, feat, dout, nullptr, weight, coef, bcoef, workspace, g, H, W, n_tiles);
    int rc = check_launch("stem_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(COUT * 9, 64)), dim3(64 * CS_PARTS), 0, st, workspace, dw, groups, COUT * 9);
    return check_launch("colsum_kernel");
}

extern "C" int lad_stem_wgrad(const float *feat, const float *dout, float *workspace, float *dw, int64_t batch, int32_t H,
                              int32_t W, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && dout && workspace && dw, "lad_stem_wgrad: null buffer");
    LAD_REQUIRE(cout == COUT && batch >= 1, "lad_stem_wgrad: bad arguments");
    return stem_wgrad_launch(feat, dout, nullptr, nullptr, nullptr, nullptr, workspace, dw, batch, H, W, stream);
}

extern "C" int lad_stem_wgrad_bn(const float *feat, const float *dy, const float *x, const float *weight, const float *coef,
                                 const float *bcoef, float *workspace, float *dw, int64_t batch, int32_t H, int32_t W, int32_t cout,
                                 void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && dy && (x || weight) && coef && bcoef && workspace && dw, "lad_stem_wgrad_bn: null buffer");
    LAD_REQUIRE(cout == COUT && batch >= 1, "lad_stem_wgrad_bn: bad arguments");
    return stem_wgrad_launch(feat, dy, x, weight, coef, bcoef, workspace, dw, batch, H, W, stream);
}

extern "C" int64_t lad_stem_bn_bwd_groups(int64_t batch, int32_t H, int32_t W) {
    if (batch < 1 || H < 1 || W < 1) return -1;
    return std::min<int64_t>(MAX_GROUPS, lad::ceil_div(lad::make_geom(batch, H, W).rows, TM));
}

extern "C" int lad_stem_bn_bwd_sums(const float *feat, const float *weight, const float *dy, const float *coef, float *partials,
                                    int64_t batch, int32_t H, int32_t W, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && dy && coef && partials, "lad_stem_bn_bwd_sums: null buffer");
    LAD_REQUIRE(cout == COUT && batch >= 1 && H >= 1 && W >= 1, "lad_stem_bn_bwd_sums: bad arguments");
    const Geom g = make_geom(batch, H, W);
    const int64_t n_tiles = ceil_div(g.rows, TM);
    const int groups = (int)std::min<int64_t>(MAX_GROUPS, n_tiles);
    hipLaunchKernelGGL(stem_bn_sums_kernel, dim3(groups), dim3(THREADS), 0, (hipStream_t)stream, feat, weight, dy, coef, partials, g, H, W,
                       n_tiles);
    return check_launch("stem_bn_sums_kernel");
}
