// Classifier head of ResNetBigger (models.py:229-238): AvgPool2d(4) -> flatten -> BatchNorm1d -> dropout ->
// Linear(F,32) -> BatchNorm1d -> dropout -> ReLU -> Linear(32,1) -> sigmoid, fused with nn.BCELoss and the
// counters behind _calc_metrics (train.py:203-224, 279-285), forward and backward.
//
// The head is ~0.1 % of the model's work (B x 48 x 32 MACs); what matters is that it costs a handful of
// launches and no host synchronisation.  Train mode needs batch statistics over B between the stages, so the
// train kernels run as ONE workgroup of 1024 threads that walks the stages with workgroup barriers; eval mode
// has no cross-sample dependency and runs one thread per sample across the whole chip.
// Dropout masks (already scaled by 1/(1-p)) are supplied by the caller so the host keeps control of the RNG.
#include "lad_common.h"
#include "lad_device.h"

namespace {
using namespace lad;

constexpr float BN_EPS = 1e-5f;
constexpr int HID = 32;
constexpr int HEAD_THREADS = 1024;
constexpr int MAX_F = 128;
constexpr int CH = 64;  // samples per LDS chunk in the train kernels

struct HeadArgs {
    int B, F;
    const float *g2, *b2;
    float *rm2, *rv2;
    const float *W1, *bias1;  // [HID][F], [HID]
    const float *g3, *b3;
    float *rm3, *rv3;
    const float *W2, *bias2;  // [HID], [1]
    const float *m1, *m2;     // dropout masks [B][F], [B][HID] or nullptr
    const int *labels;        // [B] or nullptr
    float momentum;
};

// pooled[b][c*PH*PW + ph*PW + pw] = mean of the 4x4 window (ph, pw) of channel c
__global__ void pool_fwd_kernel(const float *__restrict__ x, float *__restrict__ pooled, int64_t batch, int Hp, int Wp, int C,
                                int PH, int PW) {
    const int F = C * PH * PW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * F) return;
    const int64_t b = idx / F;
    const int f = (int)(idx - b * F);
    const int c = f / (PH * PW), ph = (f / PW) % PH, pw = f % PW;
    const float *img = x + b * (int64_t)Hp * Wp * C;
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) s += img[((1 + 4 * ph + dy) * Wp + (1 + 4 * pw + dx)) * C + c];
    pooled[idx] = s * 0.0625f;
}

// dx[b][yp][xp][c] = dpooled[b][f(c, ph, pw)] / 16 inside the pooled region, 0 elsewhere (incl. border rows)
__global__ void pool_bwd_kernel(const float *__restrict__ dpooled, float *__restrict__ dx, int64_t batch, int Hp, int Wp, int C,
                                int PH, int PW) {
    const int F = C * PH * PW;
    const int64_t total = batch * Hp * Wp * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        int64_t t = idx / C;
        const int xp = (int)(t % Wp);
        t /= Wp;
        const int yp = (int)(t % Hp);
        const int64_t b = t / Hp;
        float v = 0.f;
        const int y = yp - 1, xx = xp - 1;
        if (y >= 0 && y < 4 * PH && xx >= 0 && xx < 4 * PW) v = dpooled[b * F + c * PH * PW + (y >> 2) * PW + (xx >> 2)] * 0.0625f;
        dx[idx] = v;
    }
}

// Column reductions over the batch, coalesced: thread t owns column t % ncol and every nph-th row (nph = threads / ncol),
// so a wavefront reads consecutive floats; the per-phase partials (double) meet in `scratch` (>= 2 * blockDim.x doubles,
// workgroup-shared) and column c's result is summed over the phases in a fixed order.  fn(b, c, s0, s1) adds row b's
// contributions.  Contains workgroup barriers: every thread of the workgroup must call it.
template <typename Fn>
__device__ __forceinline__ void col_reduce(int B, int ncol, double *scratch, double &r0, double &r1, Fn fn) {
    const int nt = blockDim.x, nph = nt / ncol;
    const int c = threadIdx.x % ncol, ph = threadIdx.x / ncol;
    double s0 = 0.0, s1 = 0.0;
    if (ph < nph)
        for (int b = ph; b < B; b += nph) fn(b, c, s0, s1);
    __syncthreads();  // scratch may still be in use by the caller's previous phase
    scratch[threadIdx.x] = s0;
    scratch[nt + threadIdx.x] = s1;
    __syncthreads();
    r0 = r1 = 0.0;
    if (threadIdx.x < ncol) {
        for (int p = 0; p < nph; ++p) {
            r0 += scratch[p * ncol + threadIdx.x];
            r1 += scratch[nt + p * ncol + threadIdx.x];
        }
    }
    __syncthreads();
}

// column mean / invstd of X[B][ncol]; optional running update
__device__ void col_stats(const float *__restrict__ X, int B, int ncol, float *mean_s, float *istd_s, float *rmean, float *rvar,
                          float momentum, double *scratch) {
    double s1, s2;
    col_reduce(B, ncol, scratch, s1, s2, [&](int b, int c, double &a0, double &a1) {
        const double v = (double)X[(int64_t)b * ncol + c];
        a0 += v;
        a1 += v * v;
    });
    const int c = threadIdx.x;
    if (c < ncol) {
        const double mean = s1 / B;
        double var = s2 / B - mean * mean;
        if (var < 0.0) var = 0.0;
        mean_s[c] = (float)mean;
        istd_s[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
        if (rmean != nullptr) {
            const double unb = B > 1 ? var * B / (B - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
}

// out0[c] = sum_b X[b][c];  out1[c] = sum_b X[b][c] * (Hm[b][c] - mean[c]) * istd[c]   (out1 / Hm optional)
__device__ void col_dot2(const float *__restrict__ X, const float *__restrict__ Hm, const float *__restrict__ mean,
                         const float *__restrict__ istd, int B, int ncol, float *out0, float *out1, double *scratch) {
    double s0, s1;
    col_reduce(B, ncol, scratch, s0, s1, [&](int b, int c, double &a0, double &a1) {
        const int64_t i = (int64_t)b * ncol + c;
        const double x = (double)X[i];
        a0 += x;
        if (Hm != nullptr) a1 += x * (double)((Hm[i] - mean[c]) * istd[c]);
    });
    if (threadIdx.x < ncol) {
        out0[threadIdx.x] = (float)s0;
        if (out1 != nullptr) out1[threadIdx.x] = (float)s1;
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// stats layout (saved for backward): float[2*F + 2*HID] = mean2[F], istd2[F], mean3[HID], istd3[HID]
// metrics: float[8] = mean BCE, #correct, #pred positive, #true positive, #target positive, B, 0, 0
__global__ __launch_bounds__(HEAD_THREADS) void head_fwd_train_kernel(HeadArgs a, const float *__restrict__ pooled,
                                                                      float *__restrict__ h, float *__restrict__ stats,
                                                                      float *__restrict__ probs, float *__restrict__ metrics) {
    __shared__ float zs[MAX_F], zt[MAX_F];       // z = pooled*zs + zt
    __shared__ float w1t[MAX_F * HID];           // [F][HID]
    __shared__ float us[HID], ut[HID], w2s[HID];
    __shared__ float red[5][HEAD_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float z_s[CH * (MAX_F + 1)];  // also the scratch of the column reductions
    double *scratch = reinterpret_cast<double *>(z_s);
    static_assert(sizeof(float) * CH * (MAX_F + 1) >= sizeof(double) * 2 * HEAD_THREADS, "scratch too small");
    const int tid = threadIdx.x, nt = blockDim.x;
    const int B = a.B, F = a.F;
    float *mean2 = stats, *istd2 = stats + F, *mean3 = stats + 2 * F, *istd3 = stats + 2 * F + HID;

    col_stats(pooled, B, F, mean2, istd2, a.rm2, a.rv2, a.momentum, scratch);
    for (int i = tid; i < F * HID; i += nt) {
        const int j = i / F, f = i - j * F;
        w1t[f * HID + j] = a.W1[i];
    }
    __syncthreads();
    for (int f = tid; f < F; f += nt) {
        const float s = istd2[f] * a.g2[f];
        zs[f] = s;
        zt[f] = a.b2[f] - mean2[f] * s;
    }
    __syncthreads();
    // h = linear1(dropout(bn2(pooled))): CH samples at a time, their normalised rows staged in LDS
    for (int c0 = 0; c0 < B; c0 += CH) {
        const int nb = min(CH, B - c0);
        for (int idx = tid; idx < nb * F; idx += nt) {
            const int b = idx / F, f = idx - b * F;
            float z = fmaf(pooled[(int64_t)(c0 + b) * F + f], zs[f], zt[f]);
            if (a.m1) z *= a.m1[(int64_t)(c0 + b) * F + f];
            z_s[b * (MAX_F + 1) + f] = z;
        }
        __syncthreads();
        for (int idx = tid; idx < nb * HID; idx += nt) {
            const int b = idx / HID, j = idx - b * HID;
            float acc = a.bias1[j];
            const float *zr = z_s + b * (MAX_F + 1);
            for (int f = 0; f < F; ++f) acc = fmaf(w1t[f * HID + j], zr[f], acc);
            h[(int64_t)(c0 + b) * HID + j] = acc;
        }
        __syncthreads();
    }
    col_stats(h, B, HID, mean3, istd3, a.rm3, a.rv3, a.momentum, scratch);
    __syncthreads();
    if (tid < HID) {
        const float s = istd3[tid] * a.g3[tid];
        us[tid] = s;
        ut[tid] = a.b3[tid] - mean3[tid] * s;
        w2s[tid] = a.W2[tid];
    }
    __syncthreads();
    float loss = 0.f, n_corr = 0.f, n_pp = 0.f, n_tp = 0.f, n_t = 0.f;
    // one (sample, hidden unit) per thread: coalesced reads of h / the mask, the 32 units of a sample meet by shuffles
    static_assert(HID == 32, "a sample's hidden units are one half-wavefront");
    for (int idx = tid; idx < B * HID; idx += nt) {
        const int b = idx >> 5, j = idx & 31;
        float u = fmaf(h[idx], us[j], ut[j]);
        if (a.m2) u *= a.m2[idx];
        float part = w2s[j] * fmaxf(u, 0.f);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (j == 0) {
            const float p = sigmoidf(part + a.bias2[0]);
            probs[b] = p;
            if (a.labels != nullptr) {
                const float t = (float)a.labels[b];
                const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
                loss -= t * lp + (1.f - t) * l1p;
                const float pred = rintf(p);
                n_corr += (pred == t) ? 1.f : 0.f;
                n_pp += pred;
                n_tp += (pred == 1.f && t == 1.f) ? 1.f : 0.f;
                n_t += t;
            }
        }
    }
    float vals[5] = {loss, n_corr, n_pp, n_tp, n_t};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = wave_sum64(vals[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 5) {
        float s = 0.f;
        for (int w = 0; w < nt / 64; ++w) s += red[tid][w];
        metrics[tid] = (tid == 0) ? s / (float)B : s;
    }
    if (tid == 5) metrics[5] = (float)B;
}

// eval: running statistics, one thread per sample
__global__ __launch_bounds__(256) void head_fwd_eval_kernel(HeadArgs a, const float *__restrict__ pooled, float *__restrict__ probs) {
    __shared__ float zs[MAX_F], zt[MAX_F];
    __shared__ float w1t[MAX_F * HID];
    __shared__ float us[HID], ut[HID], w2s[HID], b1s[HID];
    const int tid = threadIdx.x, F = a.F;
    for (int i = tid; i < F * HID; i += blockDim.x) {
        const int j = i / F, f = i - j * F;
        w1t[f * HID + j] = a.W1[i];
    }
    for (int f = tid; f < F; f += blockDim.x) {
        const float s = a.g2[f] / sqrtf(a.rv2[f] + BN_EPS);
        zs[f] = s;
        zt[f] = a.b2[f] - a.rm2[f] * s;
    }
    if (tid < HID) {
        const float s = a.g3[tid] / sqrtf(a.rv3[tid] + BN_EPS);
        us[tid] = s;
        ut[tid] = a.b3[tid] - a.rm3[tid] * s;
        w2s[tid] = a.W2[tid];
        b1s[tid] = a.bias1[tid];
    }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + tid;
    if (b >= a.B) return;
    float acc[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) acc[j] = b1s[j];
    const float *pr = pooled + b * F;
    for (int f = 0; f < F; ++f) {
        const float z = fmaf(pr[f], zs[f], zt[f]);
#pragma unroll
        for (int j = 0; j < HID; ++j) acc[j] = fmaf(w1t[f * HID + j], z, acc[j]);
    }
    float logit = a.bias2[0];
#pragma unroll
    for (int j = 0; j < HID; ++j) logit = fmaf(w2s[j], fmaxf(fmaf(acc[j], us[j], ut[j]), 0.f), logit);
    probs[b] = sigmoidf(logit);
}

// metrics[0..5] = mean BCE, #correct, #pred positive, #true positive, #target positive, n for a probability vector
// (the evaluation side of train.py:226-259: BCELoss + _calc_metrics on eval-mode outputs), one workgroup
__global__ __launch_bounds__(HEAD_THREADS) void bce_metrics_kernel(const float *__restrict__ probs, const int *__restrict__ labels,
                                                                   int n, float *__restrict__ metrics) {
    __shared__ float red[5][HEAD_THREADS / 64];
    const int tid = threadIdx.x, nt = blockDim.x;
    float vals[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < n; b += nt) {
        const float p = probs[b], t = (float)labels[b];
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
        vals[0] -= t * lp + (1.f - t) * l1p;
        const float pred = rintf(p);
        vals[1] += (pred == t) ? 1.f : 0.f;
        vals[2] += pred;
        vals[3] += (pred == 1.f && t == 1.f) ? 1.f : 0.f;
        vals[4] += t;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = wave_sum64(vals[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 5) {
        float s = 0.f;
        for (int w = 0; w < nt / 64; ++w) s += red[tid][w];
        metrics[tid] = (tid == 0) ? s / (float)n : s;
    }
    if (tid == 5) metrics[5] = (float)n;
}

struct HeadGrads {
    float *dg2, *db2, *dW1, *dbias1, *dg3, *db3, *dW2, *dbias2;
};

// workspace: du[B][HID], gr[B][HID], dh[B][HID], dz[B][F]
__global__ __launch_bounds__(HEAD_THREADS) void head_bwd_kernel(HeadArgs a, HeadGrads gr_out, const float *__restrict__ pooled,
                                                                const float *__restrict__ h, const float *__restrict__ stats,
                                                                const float *__restrict__ probs, const float *__restrict__ dprobs,
                                                                float *__restrict__ ws, float *__restrict__ dpooled) {
    __shared__ float zs[MAX_F], zt[MAX_F];
    __shared__ float w1t[MAX_F * (HID + 1)];  // [F][HID+1]: stage 4 reads it with f across lanes -- 33 floats apart, no bank conflict
    __shared__ float us[HID], ut[HID], w2s[HID];
    __shared__ float ca[MAX_F], cb[MAX_F];
    __shared__ float redw[HEAD_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float z_s[CH * (MAX_F + 1)];  // also the scratch of the column reductions
    __shared__ float dh_s[CH * (HID + 1)];
    double *scratch = reinterpret_cast<double *>(z_s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int B = a.B, F = a.F;
    const float *mean2 = stats, *istd2 = stats + F, *mean3 = stats + 2 * F, *istd3 = stats + 2 * F + HID;
    float *du = ws, *gr = du + (int64_t)B * HID, *dh = gr + (int64_t)B * HID, *dz = dh + (int64_t)B * HID;

    for (int i = tid; i < F * HID; i += nt) {
        const int j = i / F, f = i - j * F;
        w1t[f * (HID + 1) + j] = a.W1[i];
    }
    for (int f = tid; f < F; f += nt) {
        const float s = istd2[f] * a.g2[f];
        zs[f] = s;
        zt[f] = a.b2[f] - mean2[f] * s;
    }
    if (tid < HID) {
        const float s = istd3[tid] * a.g3[tid];
        us[tid] = s;
        ut[tid] = a.b3[tid] - mean3[tid] * s;
        w2s[tid] = a.W2[tid];
    }
    __syncthreads();
    // ---- stage 1: dlogit, du (grad wrt bn3 output), gr = dlogit * relu(.) for dW2 -----------------------------
    float dl_sum = 0.f;
    for (int idx = tid; idx < B * HID; idx += nt) {  // one (sample, hidden unit) per thread: coalesced
        const int b = idx >> 5, j = idx & 31;
        const float p = probs[b];
        float dlogit;
        if (dprobs != nullptr) dlogit = dprobs[b] * p * (1.f - p);
        else dlogit = (p - (float)a.labels[b]) / (float)B;
        if (j == 0) dl_sum += dlogit;
        float u = fmaf(h[idx], us[j], ut[j]);
        const float m = a.m2 ? a.m2[idx] : 1.f;
        u *= m;
        gr[idx] = dlogit * fmaxf(u, 0.f);
        du[idx] = (u > 0.f) ? dlogit * w2s[j] * m : 0.f;
    }
    dl_sum = wave_sum64(dl_sum);
    if ((tid & 63) == 0) redw[tid >> 6] = dl_sum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < nt / 64; ++w) s += redw[w];
        gr_out.dbias2[0] = s;
    }
    col_dot2(gr, nullptr, nullptr, nullptr, B, HID, gr_out.dW2, nullptr, scratch);
    col_dot2(du, h, mean3, istd3, B, HID, ca, cb, scratch);  // ca = sum du (dbeta3), cb = sum du * xhat3 (dgamma3)
    __syncthreads();
    if (tid < HID) {
        gr_out.db3[tid] = ca[tid];
        gr_out.dg3[tid] = cb[tid];
    }
    // ---- stage 2: dh = g3*istd3*(du - mean(du) - xhat3*mean(du*xhat3)) ---------------------------------------
    for (int idx = tid; idx < B * HID; idx += nt) {
        const int j = idx % HID;
        const float xh = (h[idx] - mean3[j]) * istd3[j];
        dh[idx] = a.g3[j] * istd3[j] * (du[idx] - ca[j] / (float)B - xh * cb[j] / (float)B);
    }
    __syncthreads();
    // ---- stages 3 + 4, CH samples at a time with dh and the normalised inputs staged in LDS:
    //      dW1[j][f] = sum_b dh[b][j] * z[b][f];   dz[b][f] = m1[b][f] * sum_j W1[j][f] * dh[b][j];   dbias1 = colsum(dh)
    float w1acc[2] = {0.f, 0.f};  // this thread's (j, f) pairs: idx = tid and tid + nt (HID * F <= 2 * nt)
    for (int c0 = 0; c0 < B; c0 += CH) {
        const int nb = min(CH, B - c0);
        for (int idx = tid; idx < nb * F; idx += nt) {
            const int b = idx / F, f = idx - b * F;
            float z = fmaf(pooled[(int64_t)(c0 + b) * F + f], zs[f], zt[f]);
            if (a.m1) z *= a.m1[(int64_t)(c0 + b) * F + f];
            z_s[b * (MAX_F + 1) + f] = z;
        }
        for (int idx = tid; idx < nb * HID; idx += nt) dh_s[(idx / HID) * (HID + 1) + (idx % HID)] = dh[(int64_t)c0 * HID + idx];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + u * nt;
            if (idx < HID * F) {
                const int j = idx / F, f = idx - j * F;
                float sacc = w1acc[u];
                for (int b = 0; b < nb; ++b) sacc = fmaf(dh_s[b * (HID + 1) + j], z_s[b * (MAX_F + 1) + f], sacc);
                w1acc[u] = sacc;
            }
        }
        for (int idx = tid; idx < nb * F; idx += nt) {
            const int b = idx / F, f = idx - b * F;
            float sacc = 0.f;
#pragma unroll 8
            for (int j = 0; j < HID; ++j) sacc = fmaf(w1t[f * (HID + 1) + j], dh_s[b * (HID + 1) + j], sacc);
            if (a.m1) sacc *= a.m1[(int64_t)(c0 + b) * F + f];
            dz[(int64_t)(c0 + b) * F + f] = sacc;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * nt;
        if (idx < HID * F) gr_out.dW1[idx] = w1acc[u];
    }
    col_dot2(dh, nullptr, nullptr, nullptr, B, HID, gr_out.dbias1, nullptr, scratch);
    // ---- stage 5: bn2 backward -------------------------------------------------------------------------------------
    col_dot2(dz, pooled, mean2, istd2, B, F, ca, cb, scratch);
    __syncthreads();
    for (int f = tid; f < F; f += nt) {
        gr_out.db2[f] = ca[f];
        gr_out.dg2[f] = cb[f];
    }
    for (int idx = tid; idx < B * F; idx += nt) {
        const int f = idx % F;
        const float xh = (pooled[idx] - mean2[f]) * istd2[f];
        dpooled[idx] = a.g2[f] * istd2[f] * (dz[idx] - ca[f] / (float)B - xh * cb[f] / (float)B);
    }
}

int check_head(const HeadArgs &a) {
    using namespace lad;
    LAD_REQUIRE(a.B >= 1 && a.F >= 1 && a.F <= MAX_F && a.F * HID <= 2 * HEAD_THREADS, "head: F must be 1..%d (got %d), B >= 1",
                HEAD_THREADS * 2 / HID, a.F);
    LAD_REQUIRE(a.g2 && a.b2 && a.rm2 && a.rv2 && a.W1 && a.bias1 && a.g3 && a.b3 && a.rm3 && a.rv3 && a.W2 && a.bias2,
                "head: null parameter pointer");
    return LAD_OK;
}

}  // namespace

extern "C" int lad_pool_fwd(const float *x, float *pooled, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(x && pooled, "lad_pool_fwd: null buffer");
    LAD_REQUIRE(H >= 4 && W >= 4 && channels >= 1, "lad_pool_fwd: AvgPool2d(4) needs H, W >= 4");
    if (batch == 0) return LAD_OK;
    const int PH = H / 4, PW = W / 4;
    const int64_t n = batch * channels * PH * PW;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, pooled, batch, H + 1,
                       W + 1, channels, PH, PW);
    return check_launch("pool_fwd_kernel");
}

extern "C" int lad_pool_bwd(const float *dpooled, float *dx, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(dpooled && dx, "lad_pool_bwd: null buffer");
    LAD_REQUIRE(H >= 4 && W >= 4 && channels >= 1, "lad_pool_bwd: AvgPool2d(4) needs H, W >= 4");
    if (batch == 0) return LAD_OK;
    const int64_t n = batch * (H + 1) * (W + 1) * channels;  // the body; the tail of the layout stays zero
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       dpooled, dx, batch, H + 1, W + 1, channels, H / 4, W / 4);
    return check_launch("pool_bwd_kernel");
}

// params: HOST array of 12 device pointers: bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, linear1.weight,
// linear1.bias, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var, linear2.weight, linear2.bias
static HeadArgs make_args(const float *const *p, int64_t B, int32_t F, const float *m1, const float *m2, const int32_t *labels,
                          float momentum) {
    HeadArgs a;
    a.B = (int)B; a.F = F;
    a.g2 = p[0]; a.b2 = p[1]; a.rm2 = (float *)p[2]; a.rv2 = (float *)p[3];
    a.W1 = p[4]; a.bias1 = p[5];
    a.g3 = p[6]; a.b3 = p[7]; a.rm3 = (float *)p[8]; a.rv3 = (float *)p[9];
    a.W2 = p[10]; a.bias2 = p[11];
    a.m1 = m1; a.m2 = m2; a.labels = (const int *)labels; a.momentum = momentum;
    return a;
}

extern "C" int64_t lad_head_workspace_floats(int64_t batch, int32_t F) { return batch * (3 * HID + F); }

extern "C" int lad_head_fwd_train(const float *const *params, const float *pooled, int64_t batch, int32_t F, const float *drop1,
                                  const float *drop2, const int32_t *labels, float momentum, float *h, float *stats, float *probs,
                                  float *metrics, void *stream) {
    using namespace lad;
    LAD_REQUIRE(params && pooled && h && stats && probs && metrics, "lad_head_fwd_train: null buffer");
    LAD_REQUIRE(batch < (1 << 24), "lad_head_fwd_train: batch too large for the single-workgroup train head");
    HeadArgs a = make_args(params, batch, F, drop1, drop2, labels, momentum);
    int rc = check_head(a);
    if (rc) return rc;
    hipLaunchKernelGGL(head_fwd_train_kernel, dim3(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, a, pooled, h, stats, probs, metrics);
    return check_launch("head_fwd_train_kernel");
}

extern "C" int lad_head_fwd_eval(const float *const *params, const float *pooled, int64_t batch, int32_t F, float *probs, void *stream) {
    using namespace lad;
    LAD_REQUIRE(params && pooled && probs, "lad_head_fwd_eval: null buffer");
    if (batch == 0) return LAD_OK;
    HeadArgs a = make_args(params, batch, F, nullptr, nullptr, nullptr, 0.f);
    int rc = check_head(a);
    if (rc) return rc;
    hipLaunchKernelGGL(head_fwd_eval_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, (hipStream_t)stream, a, pooled, probs);
    return check_launch("head_fwd_eval_kernel");
}

extern "C" int lad_bce_metrics(const float *probs, const int32_t *labels, int64_t n, float *metrics, void *stream) {
    using namespace lad;
    LAD_REQUIRE(probs && labels && metrics && n >= 1 && n < (1 << 30), "lad_bce_metrics: bad argument");
    hipLaunchKernelGGL(bce_metrics_kernel, dim3(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, probs, (const int *)labels, (int)n,
                       metrics);
    return check_launch("bce_metrics_kernel");
}

// grads: HOST array of 8 device pointers: d bn2.weight, d bn2.bias, d linear1.weight, d linear1.bias, d bn3.weight,
// d bn3.bias, d linear2.weight, d linear2.bias.  dprobs == NULL means "loss is mean BCE against labels".
extern "C" int lad_head_bwd(const float *const *params, float *const *grads, const float *pooled, const float *h, const float *stats,
                            const float *probs, const float *dprobs, int64_t batch, int32_t F, const float *drop1, const float *drop2,
                            const int32_t *labels, float *workspace, float *dpooled, void *stream) {
    using namespace lad;
    LAD_REQUIRE(params && grads && pooled && h && stats && probs && workspace && dpooled, "lad_head_bwd: null buffer");
    LAD_REQUIRE(dprobs || labels, "lad_head_bwd: need dprobs or labels");
    HeadArgs a = make_args(params, batch, F, drop1, drop2, labels, 0.f);
    int rc = check_head(a);
    if (rc) return rc;
    HeadGrads g;
    g.dg2 = grads[0]; g.db2 = grads[1]; g.dW1 = grads[2]; g.dbias1 = grads[3];
    g.dg3 = grads[4]; g.db3 = grads[5]; g.dW2 = grads[6]; g.dbias2 = grads[7];
    hipLaunchKernelGGL(head_bwd_kernel, dim3(1), dim3(HEAD_THREADS), 0, (hipStream_t)stream, a, g, pooled, h, stats, probs, dprobs,
                       workspace, dpooled);
    return check_launch("head_bwd_kernel");
}
