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


// ---- round 6: the stem's batch statistics and its BatchNorm + weight-gradient backward from MOMENTS of the input ----------------------
// The stem convolution has ONE input channel: x_c = sum_t w[c][t] f_t is linear in the nine taps f_t of a position, so every sum over
// positions that the BatchNorm needs is a combination of F1[t] = sum f_t and F2[t][u] = sum f_t f_u (54 numbers, channel-independent):
//   sum x_c = sum_t w[c][t] F1[t],   sum x_c^2 = sum_t sum_u w[c][t] w[c][u] F2[t][u].
// Forward: one pass over the 9 MB of features instead of a 64-channel convolution pass (stem_fwd_kernel<false>, 62 us + the two-level sum).
// Backward: dW[c][t] = sum f_t dx_c with dx_c = k1 (dz - k2 - xhat k3) needs k2, k3 = means of (dz, dz xhat) BEFORE the sum can be formed --
// two passes over the 577 MB of dy (stem_bn_sums_kernel, stem_wgrad_kernel<2>).  With g_t = f_t - m (m = the features' mean: sum dx = 0, so
// the shift is free and takes the common part out of the sums)
//   dW[c][t] = k1 (A[c][t] - k2 G1[t] - k3 Gx[c][t]),  A = sum g_t dz,  G1[t] = sum g_t = F1[t] - m N,
//   Gx[c][t] = sum g_t xhat_c = invstd_c (sum_u w[c][u] (F2[t][u] - m F1[u]) - mean_c G1[t]):
// ONE pass over dy that leaves A, sum dz and sum dz xhat per workgroup; the rest is 576 double-precision expressions.
constexpr int MOM_N = 9 + 45;        // F1[t], then F2[t][u] for u >= t, row by row
constexpr int MOM_GROUPS = 1024, MOM_THREADS = 256;
constexpr int MOM_DOUBLES = 64;      // the saved record: the 54 totals, [54] = m, [55] = N
__host__ __device__ constexpr int mom_idx(int t, int u) { return t <= u ? 9 + t * 9 - t * (t - 1) / 2 + (u - t) : 9 + u * 9 - u * (u - 1) / 2 + (t - u); }
// A workgroup takes whole images: the image with a zero ring in LDS (coalesced loads), then a thread per position with the 54 sums in its
// registers -- nine LDS reads and 54 double multiply-adds per position; the threads' sums meet once per workgroup.  (Measured on the way:
// a thread per row with its nine taps gathered from global memory, 140 us -- one exposed L2 round trip per row at two waves per CU; a
// thread per (sum, row quarter) reading two taps per multiply-add from LDS, 109 us, 44 us with eight positions in flight -- 108 LDS reads
// per position instead of nine.)
__global__ __launch_bounds__(MOM_THREADS) void stem_moments_kernel(const float *__restrict__ feat, double *__restrict__ partials /*[groups][MOM_N]*/,
                                                                   int64_t batch, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mom_lds[];
    float *img_s = reinterpret_cast<float *>(mom_lds);   // [(H + 2)][(W + 2)]
    const int tid = threadIdx.x, Wq = W + 2, n_pad = (H + 2) * Wq;
    double acc[MOM_N];
#pragma unroll
    for (int k = 0; k < MOM_N; ++k) acc[k] = 0.0;
    for (int i = tid; i < n_pad; i += MOM_THREADS) img_s[i] = 0.0f;   // the ring stays zero: only the interior is rewritten per image
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        __syncthreads();   // the previous image has been read (first image: the zeros are written)
        const float *src = feat + b * H * W;
        constexpr int LD = 18;   // loads in flight per thread (a load and its LDS store per iteration exposed 18 L2 round trips per image)
        for (int i0 = tid; i0 < H * W; i0 += LD * MOM_THREADS) {
            float ld[LD];
#pragma unroll
            for (int k = 0; k < LD; ++k) ld[k] = i0 + k * MOM_THREADS < H * W ? src[i0 + k * MOM_THREADS] : 0.0f;
#pragma unroll
            for (int k = 0; k < LD; ++k) {
                const int i = i0 + k * MOM_THREADS, y = i / W, x = i - y * W;
                if (i < H * W) img_s[(y + 1) * Wq + x + 1] = ld[k];
            }
        }
        __syncthreads();
        for (int i = tid; i < H * W; i += MOM_THREADS) {
            const int y = i / W, x = i - y * W;
            const float *c = img_s + (y + 1) * Wq + x + 1;
            double v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = (double)c[(t / 3 - 1) * Wq + (t % 3 - 1)];
            int k = 9;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                acc[t] += v[t];
#pragma unroll
                for (int u = t; u < 9; ++u) acc[k + (u - t)] = fma(v[t], v[u], acc[k + (u - t)]);
                k += 9 - t;
            }
        }
    }
    // the threads' sums meet through LDS, 18 sums at a time in the image's place (54 butterflies of shuffles took 20 us: twelve dependent
    // LDS-crossbar round trips each); thread k of a batch adds its column over the threads in order
    constexpr int RB = 18;
    double *red = reinterpret_cast<double *>(mom_lds);   // [RB][MOM_THREADS]
    __shared__ double red2[MOM_THREADS / RB][RB];
#pragma unroll
    for (int k0 = 0; k0 < MOM_N; k0 += RB) {
        __syncthreads();   // the image / the previous batch has been read
#pragma unroll
        for (int k = 0; k < RB; ++k) red[k * MOM_THREADS + tid] = acc[k0 + k];
        __syncthreads();
        constexpr int NP = MOM_THREADS / RB, SPAN = (MOM_THREADS + NP - 1) / NP;   // 14 parts of up to 19 threads' sums per column
        const int kk = tid % RB, part = tid / RB;
        if (part < NP) {
            const double *col = red + kk * MOM_THREADS;
            double t = 0.0;
            for (int i = part * SPAN; i < min(MOM_THREADS, (part + 1) * SPAN); ++i) t += col[i];
            red2[part][kk] = t;
        }
        __syncthreads();
        if (tid < RB) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < NP; ++q) t += red2[q][tid];
            partials[(int64_t)blockIdx.x * MOM_N + k0 + tid] = t;
        }
    }
}
// one workgroup: the totals (fixed order), the saved record, and per channel the coefficients of bn_finalize_kernel (same formulas)
constexpr int MOMF_THREADS = 1024;
__global__ __launch_bounds__(MOMF_THREADS) void stem_stats_finalize_kernel(const double *__restrict__ partials, int groups, const float *__restrict__ w,
                                                                           double count, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                           float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                           float momentum, float *__restrict__ coef, double *__restrict__ mom) {
    constexpr float BN_EPS = 1e-5f;
    const int tid = threadIdx.x, j = tid & 63, part = tid >> 6, nparts = MOMF_THREADS / 64;
    __shared__ double red[MOMF_THREADS / 64][64];
    __shared__ double tot[64];
    double s = 0.0;
    if (j < MOM_N) {
        int gi = part;
        for (; gi + 7 * nparts < groups; gi += 8 * nparts) {   // eight loads in flight, added in order
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = partials[(int64_t)(gi + q * nparts) * MOM_N + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; gi < groups; gi += nparts) s += partials[(int64_t)gi * MOM_N + j];
    }
    red[part][j] = s;
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
        for (int p2 = 0; p2 < nparts; ++p2) t += red[p2][tid];
        tot[tid] = t;
    }
    __syncthreads();
    if (tid < MOM_N) mom[tid] = tot[tid];
    if (tid == MOM_N) mom[MOM_N] = tot[4] / count;   // m: the mean of the centre tap = of the features
    if (tid == MOM_N + 1) mom[MOM_N + 1] = count;
    if (tid < COUT) {
        const int c = tid;
        double wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = (double)w[c * 9 + t];
        double sx = 0.0, sxx = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            sx = fma(wv[t], tot[t], sx);
#pragma unroll
            for (int u = 0; u < 9; ++u) sxx = fma(wv[t] * wv[u], tot[mom_idx(t, u)], sxx);
        }
        const double mean = sx / count;
        double var = sxx / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd_d = 1.0 / sqrt(var + (double)BN_EPS);
        const float invstd = (float)invstd_d;
        const float scale = gamma[c] * invstd;
        coef[0 * COUT + c] = scale;
        coef[1 * COUT + c] = beta[c] - (float)mean * scale;
        coef[2 * COUT + c] = (float)mean;
        coef[3 * COUT + c] = invstd;
        coef[4 * COUT + c] = (float)(mean - (double)(float)mean);
        coef[5 * COUT + c] = (float)(invstd_d - (double)invstd);
        if (running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// the ONE pass over dy: slab[wg] = A[c][t] (576) | sum dz [c] (64) | sum dz xhat [c] (64)
constexpr int BWD1_COLS = COUT * 9 + 2 * COUT;
__global__ __launch_bounds__(THREADS) void stem_bwd_onepass_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                                   const float *__restrict__ dy, const float *__restrict__ coef,
                                                                   const double *__restrict__ mom, float *__restrict__ slabs, Geom g, int H,
                                                                   int W, int64_t n_tiles) {
    const int tid = threadIdx.x, cq = tid % CQ, rl = tid / CQ;
    float acc[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    const float4 fsc = *reinterpret_cast<const float4 *>(coef + cq * 4);
    const float4 fsh = *reinterpret_cast<const float4 *>(coef + COUT + cq * 4);
    const Norm4 nm = load_norm(coef, COUT, cq * 4);
    const float m = (float)mom[MOM_N];
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
                const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float gt = v[t] - m;
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c][t] = fmaf(gt, dd[c], acc[c][t]);
                }
            }
        }
    }
    __shared__ float red[RL][BWD1_COLS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[rl][(cq * 4 + c) * 9 + t] = acc[c][t];
    *reinterpret_cast<float4 *>(&red[rl][COUT * 9 + cq * 4]) = a0;
    *reinterpret_cast<float4 *>(&red[rl][COUT * 10 + cq * 4]) = a1;
    __syncthreads();
    for (int e = tid; e < BWD1_COLS; e += THREADS) {
        float sacc = 0.f;
#pragma unroll
        for (int p = 0; p < RL; ++p) sacc += red[p][e];
        slabs[(int64_t)blockIdx.x * BWD1_COLS + e] = sacc;
    }
}
// the 704 column sums (colsum_kernel's walk: a block owns 64 columns, its 16 wavefronts a sixteenth of the slabs each, fixed order) and,
// in the workgroup that takes the last ticket (common.hip; the sums travel by device-scope stores and loads), dw, dgamma, dbeta from them
// in double arithmetic
__global__ __launch_bounds__(64 * CS_PARTS) void stem_bwd_finish_kernel(const float *__restrict__ slabs, float *sums, int groups,
                                                                        const double *__restrict__ mom, const float *__restrict__ w,
                                                                        const float *__restrict__ coef, const float *__restrict__ gamma,
                                                                        float *__restrict__ dw, float *__restrict__ dgamma,
                                                                        float *__restrict__ dbeta, unsigned int *ticket) {
    constexpr int n = BWD1_COLS;
    const int o = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    __shared__ double red[CS_PARTS][64];
    __shared__ float tot[BWD1_COLS];
    __shared__ int last_s;
    double s = 0.0;
    if (idx < n) {
        const float *src = slabs + idx;
        int wg = part;
        for (; wg + 3 * CS_PARTS < groups; wg += 4 * CS_PARTS) {
            const float a = src[(int64_t)wg * n], b = src[(int64_t)(wg + CS_PARTS) * n], c = src[(int64_t)(wg + 2 * CS_PARTS) * n],
                        d = src[(int64_t)(wg + 3 * CS_PARTS) * n];
            s += (double)a; s += (double)b; s += (double)c; s += (double)d;
        }
        for (; wg < groups; wg += CS_PARTS) s += (double)src[(int64_t)wg * n];
    }
    red[part][o] = s;
    __syncthreads();
    if (part == 0 && idx < n) {
        double t = 0.0;
#pragma unroll
        for (int p = 0; p < CS_PARTS; ++p) t += red[p][o];
        __hip_atomic_store(sums + idx, (float)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this block's sums have arrived before its ticket is taken (see bn_slice_sum_kernel)
    __syncthreads();
    if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last_s) return;
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int e = threadIdx.x;
    if (e < n) tot[e] = __hip_atomic_load(sums + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (e >= n) return;
    if (e >= COUT * 9) {
        const int c = (e - COUT * 9) & (COUT - 1);
        if (e < COUT * 10) dbeta[c] = tot[e];
        else dgamma[c] = tot[e];
        return;
    }
    const int c = e / 9, t = e - c * 9;
    const double cnt = mom[MOM_N + 1], m = mom[MOM_N];
    const double mean = (double)coef[2 * COUT + c] + (double)coef[4 * COUT + c], invstd = (double)coef[3 * COUT + c] + (double)coef[5 * COUT + c];
    const double k1 = (double)gamma[c] * invstd, k2 = (double)tot[COUT * 9 + c] / cnt, k3 = (double)tot[COUT * 10 + c] / cnt;
    const double g1 = mom[t] - m * cnt;
    double sgx = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) sgx = fma((double)w[c * 9 + u], mom[mom_idx(t, u)] - m * mom[u], sgx);
    const double gx = invstd * (sgx - mean * g1);
    dw[e] = (float)(k1 * ((double)tot[e] - k2 * g1 - k3 * gx));
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

// ---- round 6: statistics and backward of the stem from moments of the input (kernels above) ------------------------------------------
extern "C" int64_t lad_stem_moments_workspace_doubles(void) { return (int64_t)MOM_GROUPS * MOM_N; }
extern "C" int64_t lad_stem_moments_doubles(void) { return MOM_DOUBLES; }

extern "C" int lad_stem_bn_stats(const float *feat, const float *weight, const float *gamma, const float *beta, float *running_mean,
                                 float *running_var, float momentum, float *coef, double *moments, double *workspace, int64_t batch,
                                 int32_t H, int32_t W, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && gamma && beta && coef && moments && workspace, "lad_stem_bn_stats: null buffer");
    LAD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "lad_stem_bn_stats: running stats must come in pairs");
    LAD_REQUIRE(cout == COUT && batch >= 1 && H >= 1 && W >= 1, "lad_stem_bn_stats: bad arguments");
    const int groups = (int)std::min<int64_t>(MOM_GROUPS, batch);
    const size_t lds = std::max<size_t>((size_t)(H + 2) * (W + 2) * sizeof(float), (size_t)18 * MOM_THREADS * sizeof(double));
    LAD_REQUIRE(lds <= 60 * 1024, "lad_stem_bn_stats: an image of %d x %d does not fit the workgroup's LDS", H, W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_moments_kernel, dim3(groups), dim3(MOM_THREADS), lds, st, feat, workspace, batch, H, W);
    int rc = check_launch("stem_moments_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(stem_stats_finalize_kernel, dim3(1), dim3(MOMF_THREADS), 0, st, (const double *)workspace, groups, weight,
                       (double)(batch * (int64_t)H * W), gamma, beta, running_mean, running_var, momentum, coef, moments);
    return check_launch("stem_stats_finalize_kernel");
}

extern "C" int64_t lad_stem_bwd_onepass_workspace_floats(void) { return (int64_t)(MAX_GROUPS + 1) * BWD1_COLS; }

extern "C" int lad_stem_bwd_onepass(const float *feat, const float *weight, const float *dy, const float *coef, const float *gamma,
                                    const double *moments, float *workspace, float *dw, float *dgamma, float *dbeta, int64_t batch,
                                    int32_t H, int32_t W, int32_t cout, void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && dy && coef && gamma && moments && workspace && dw && dgamma && dbeta, "lad_stem_bwd_onepass: null buffer");
    LAD_REQUIRE(cout == COUT && batch >= 1 && H >= 1 && W >= 1, "lad_stem_bwd_onepass: bad arguments");
    const Geom g = make_geom(batch, H, W);
    const int64_t n_tiles = ceil_div(g.rows, TM);
    const int groups = (int)std::min<int64_t>(MAX_GROUPS, n_tiles);
    hipStream_t st = (hipStream_t)stream;
    float *sums = workspace + (int64_t)MAX_GROUPS * BWD1_COLS;
    hipLaunchKernelGGL(stem_bwd_onepass_kernel, dim3(groups), dim3(THREADS), 0, st, feat, weight, dy, coef, moments, workspace, g, H, W, n_tiles);
    int rc = check_launch("stem_bwd_onepass_kernel");
    if (rc) return rc;
    unsigned int *ticket = launch_ticket();
    LAD_REQUIRE(ticket, "lad_stem_bwd_onepass: no ticket");
    static_assert(BWD1_COLS <= 64 * CS_PARTS, "the last workgroup takes a column per thread");
    hipLaunchKernelGGL(stem_bwd_finish_kernel, dim3((unsigned)ceil_div(BWD1_COLS, 64)), dim3(64 * CS_PARTS), 0, st, (const float *)workspace, sums,
                       groups, moments, weight, coef, gamma, dw, dgamma, dbeta, ticket);
    return check_launch("stem_bwd_finish_kernel");
}
