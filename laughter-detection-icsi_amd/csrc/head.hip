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

#ifdef LAD_STAMP
__device__ unsigned long long lad_dbg_head[64];   // diagnostic build only (tools/stamp_head.py): s_memtime at the phase boundaries
#define LAD_HEAD_STAMP(k) if (threadIdx.x == 0) lad_dbg_head[k] = __builtin_amdgcn_s_memtime();
#else
#define LAD_HEAD_STAMP(k)
#endif

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
    // lad_head_fwd_train_rng: the kernel draws the dropout masks itself (into m1 / m2, for the backward pass) and counts the step
    float keep;               // 1 - p; masks hold 0 or 1 / keep
    unsigned long long seed;
    long long *rng_counter;   // device counter of mask draws (nullptr: masks are the caller's, or none)
    long long *nbt;           // num_batches_tracked counters of the model's BatchNorms, incremented by one (or nullptr)
    int n_nbt;
};

// Philox4x32-10 (Salmon et al., SC'11): four 32-bit words from a 128-bit counter and a 64-bit key -- a counter-based generator needs no
// state, so a mask element's value is a pure function of (seed, draw number, element index) and a hipGraph replay draws fresh masks
struct U4 {
    unsigned x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// inverted-dropout mask values (0 or 1 / keep) for elements 4 g .. 4 g + 3 of mask `which` at draw `draw`
__device__ __forceinline__ float4 dropout_mask4(unsigned long long seed, long long draw, int which, unsigned g, float keep) {
    const U4 r = philox4x32_10(U4{g, (unsigned)which, (unsigned)draw, (unsigned)((unsigned long long)draw >> 32)}, (unsigned)seed,
                               (unsigned)(seed >> 32));
    const float inv = 1.0f / keep, s = 1.0f / 16777216.0f;
    return make_float4((float)(r.x >> 8) * s < keep ? inv : 0.f, (float)(r.y >> 8) * s < keep ? inv : 0.f,
                       (float)(r.z >> 8) * s < keep ? inv : 0.f, (float)(r.w >> 8) * s < keep ? inv : 0.f);
}

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
// The single workgroup of the train kernels is latency-bound, not throughput-bound: a loop "load, then use" pays one L2 round
// trip (~1 us) per iteration.  batched<UN>: UN iterations' loads are issued before the first is consumed; consumption stays in
// ascending index order, so every sum keeps its order (round 4, with the per-sample linear layers and the register-tiled dW1 below: head_fwd_train 109 -> ~50 us, head_bwd 139 -> ~100 us at batch 512; tools/stamp_head.py).
template <int UN, typename V, typename Load, typename Use>
__device__ __forceinline__ void batched(int first, int n, int step, Load load, Use use) {
    for (int base = first; base < n; base += step * UN) {
        V v[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int idx = base + k * step;
            if (idx < n) v[k] = load(idx);
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int idx = base + k * step;
            if (idx < n) use(idx, v[k]);
        }
    }
}
struct F2 {
    float a, b;
};

template <typename Load, typename Acc>
__device__ __forceinline__ void col_reduce(int B, int ncol, double *scratch, double &r0, double &r1, Load load, Acc acc) {
    const int nt = blockDim.x, nph = nt / ncol;
    const int c = threadIdx.x % ncol, ph = threadIdx.x / ncol;
    double s0 = 0.0, s1 = 0.0;
    if (ph < nph)
        batched<8, F2>(ph, B, nph, [&](int b) { return load(b, c); }, [&](int b, const F2 &v) { acc(v, c, s0, s1); });
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
    col_reduce(B, ncol, scratch, s1, s2, [&](int b, int c) { return F2{X[(int64_t)b * ncol + c], 0.f}; },
               [&](const F2 &x, int, double &a0, double &a1) {
                   const double v = (double)x.a;
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
    const int cc = threadIdx.x % ncol;
    const float mc = Hm != nullptr ? mean[cc] : 0.f, ic = Hm != nullptr ? istd[cc] : 0.f;   // (this thread's column: read once)
    col_reduce(B, ncol, scratch, s0, s1,
               [&](int b, int c) {
                   const int64_t i = (int64_t)b * ncol + c;
                   return F2{X[i], Hm != nullptr ? Hm[i] : 0.f};
               },
               [&](const F2 &v, int, double &a0, double &a1) {
                   const double x = (double)v.a;
                   a0 += x;
                   if (Hm != nullptr) a1 += x * (double)((v.b - mc) * ic);
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
    __shared__ __attribute__((aligned(16))) float w1t[MAX_F * HID];   // [F][HID]
    __shared__ float us[HID], ut[HID], w2s[HID], b1s[HID];
    __shared__ float red[5][HEAD_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float z_s[CH * (MAX_F + 1)];  // also the scratch of the column reductions
    double *scratch = reinterpret_cast<double *>(z_s);
    static_assert(sizeof(float) * CH * (MAX_F + 1) >= sizeof(double) * 2 * HEAD_THREADS, "scratch too small");
    const int tid = threadIdx.x, nt = blockDim.x;
    const int B = a.B, F = a.F;
    float *mean2 = stats, *istd2 = stats + F, *mean3 = stats + 2 * F, *istd3 = stats + 2 * F + HID;

    LAD_HEAD_STAMP(0)
    if (a.rng_counter != nullptr) {   // the two dropout masks of models.py:232,235, drawn here (one launch instead of torch's four)
        const long long draw = *a.rng_counter;
        float4 *o1 = reinterpret_cast<float4 *>(const_cast<float *>(a.m1)), *o2 = reinterpret_cast<float4 *>(const_cast<float *>(a.m2));
        for (int g = tid; g < B * F / 4; g += nt) o1[g] = dropout_mask4(a.seed, draw, 0, (unsigned)g, a.keep);
        for (int g = tid; g < B * HID / 4; g += nt) o2[g] = dropout_mask4(a.seed, draw, 1, (unsigned)g, a.keep);
        // (read back below by this same workgroup, behind the barriers of col_stats)
    }
    col_stats(pooled, B, F, mean2, istd2, a.rm2, a.rv2, a.momentum, scratch);
    LAD_HEAD_STAMP(1)
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
    if (tid < HID) b1s[tid] = a.bias1[tid];
    __syncthreads();
    LAD_HEAD_STAMP(2)
    // h = linear1(dropout(bn2(pooled))): one sample per thread, its 32 hidden units in registers; a weight row is one LDS
    // broadcast for the whole wave (round 4; before: 64 samples at a time through LDS with two LDS reads per multiply-add, two
    // barriers and one exposed load latency per chunk).  Per hidden unit the sum runs over f in ascending order from the bias, as
    // before.  (Two threads per sample, 16 units each, measured SLOWER -- 34 us against 20: what is left is the row-strided
    // global reads of `pooled`, one cache line per lane, and the second thread doubles them.)
#pragma unroll 1
    for (int b = tid; b < B; b += nt) {
        float acc[HID];
#pragma unroll
        for (int j = 0; j < HID; ++j) acc[j] = b1s[j];
        const float *pr = pooled + (int64_t)b * F;
        const float *mr = a.m1 ? a.m1 + (int64_t)b * F : nullptr;
        for (int f0 = 0; f0 < F; f0 += 8) {
            float pv[8], mv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = f0 + k;
                pv[k] = f < F ? pr[f] : 0.f;
                mv[k] = (mr != nullptr && f < F) ? mr[f] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = f0 + k;
                if (f < F) {
                    float z = fmaf(pv[k], zs[f], zt[f]);
                    if (mr != nullptr) z *= mv[k];
#pragma unroll
                    for (int j = 0; j < HID; ++j) acc[j] = fmaf(w1t[f * HID + j], z, acc[j]);
                }
            }
        }
        float4 *hr = reinterpret_cast<float4 *>(h + (int64_t)b * HID);
#pragma unroll
        for (int j = 0; j < HID; j += 4) hr[j >> 2] = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    }
    __syncthreads();   // h is complete (written and read by this one workgroup)
    LAD_HEAD_STAMP(3)
    col_stats(h, B, HID, mean3, istd3, a.rm3, a.rv3, a.momentum, scratch);
    LAD_HEAD_STAMP(4)
    __syncthreads();
    if (tid < HID) {
        const float s = istd3[tid] * a.g3[tid];
        us[tid] = s;
        ut[tid] = a.b3[tid] - mean3[tid] * s;
        w2s[tid] = a.W2[tid];
    }
    __syncthreads();
    float loss = 0.f, n_corr = 0.f, n_pp = 0.f, n_tp = 0.f, n_t = 0.f;
    static_assert(HID == 32, "the in-register sum below mirrors a 32-lane butterfly");
    const float bias2 = a.bias2[0];
    // one sample per thread: its 32 hidden units are summed in registers in the order of the xor-butterfly that used to join the 32
    // lanes of a sample (p[j] + p[j + 16], then + 8, + 4, + 2, + 1); sigmoid and the logarithms run with every lane busy.
    #pragma unroll 1
    for (int b = tid; b < B; b += nt) {
        asm volatile("" ::: "memory");   // (the 96 coefficients in LDS are read per sample: hoisted out of this loop they do not fit)
        float part[HID], mk[HID];
        const float4 *hr = reinterpret_cast<const float4 *>(h + (int64_t)b * HID);
        const float4 *mr = a.m2 ? reinterpret_cast<const float4 *>(a.m2 + (int64_t)b * HID) : nullptr;
#pragma unroll
        for (int q = 0; q < HID / 4; ++q) {
            const float4 hv = hr[q];
            part[4 * q] = hv.x; part[4 * q + 1] = hv.y; part[4 * q + 2] = hv.z; part[4 * q + 3] = hv.w;
            if (mr != nullptr) {
                const float4 mv = mr[q];
                mk[4 * q] = mv.x; mk[4 * q + 1] = mv.y; mk[4 * q + 2] = mv.z; mk[4 * q + 3] = mv.w;
            }
        }
        const float t = a.labels != nullptr ? (float)a.labels[b] : 0.f;
#pragma unroll
        for (int j = 0; j < HID; ++j) {
            float u = fmaf(part[j], us[j], ut[j]);
            if (mr != nullptr) u *= mk[j];
            part[j] = w2s[j] * fmaxf(u, 0.f);
        }
#pragma unroll
        for (int off = HID / 2; off > 0; off >>= 1)
#pragma unroll
            for (int j = 0; j < off; ++j) part[j] += part[j + off];
        const float p = sigmoidf(part[0] + bias2);
        probs[b] = p;
        if (a.labels != nullptr) {
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
            loss -= t * lp + (1.f - t) * l1p;
            const float pred = rintf(p);
            n_corr += (pred == t) ? 1.f : 0.f;
            n_pp += pred;
            n_tp += (pred == 1.f && t == 1.f) ? 1.f : 0.f;
            n_t += t;
        }
    }
    LAD_HEAD_STAMP(5)
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
    if (a.rng_counter != nullptr && tid == 6) *a.rng_counter += 1;   // (every thread read it before the first barrier)
    if (a.nbt != nullptr && tid >= 64 && tid - 64 < a.n_nbt) a.nbt[tid - 64] += 1;
    LAD_HEAD_STAMP(6)
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
    __shared__ __attribute__((aligned(16))) float w1t[MAX_F * HID];  // [F][HID]: stage 4 reads a row as a wave-wide broadcast
    __shared__ float us[HID], ut[HID], w2s[HID];
    __shared__ float ca[MAX_F], cb[MAX_F];
    __shared__ float mu2[MAX_F], is2[MAX_F], kk2[MAX_F];   // bn2: mean, invstd, gamma * invstd
    __shared__ float redw[HEAD_THREADS / 64];
    __shared__ __attribute__((aligned(16))) float z_s[CH * (MAX_F + 1)];  // also the scratch of the column reductions
    __shared__ __attribute__((aligned(16))) float dh_s[CH * (HID + 4)];
    double *scratch = reinterpret_cast<double *>(z_s);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int B = a.B, F = a.F;
    const float *mean2 = stats, *istd2 = stats + F, *mean3 = stats + 2 * F, *istd3 = stats + 2 * F + HID;
    float *du = ws, *gr = du + (int64_t)B * HID, *dh = gr + (int64_t)B * HID, *dz = dh + (int64_t)B * HID;

    LAD_HEAD_STAMP(16)
    for (int i = tid; i < F * HID; i += nt) {
        const int j = i / F, f = i - j * F;
        w1t[f * HID + j] = a.W1[i];
    }
    for (int f = tid; f < F; f += nt) {
        const float s = istd2[f] * a.g2[f];
        zs[f] = s;
        zt[f] = a.b2[f] - mean2[f] * s;
        mu2[f] = mean2[f];
        is2[f] = istd2[f];
        kk2[f] = a.g2[f] * istd2[f];
    }
    if (tid < HID) {
        const float s = istd3[tid] * a.g3[tid];
        us[tid] = s;
        ut[tid] = a.b3[tid] - mean3[tid] * s;
        w2s[tid] = a.W2[tid];
    }
    const int jt = tid & (HID - 1);   // this thread's hidden unit in every (sample, unit) loop: nt is a multiple of HID
    const float mean3_j = mean3[jt], istd3_j = istd3[jt], k3_j = a.g3[jt] * istd3[jt];
    __syncthreads();
    LAD_HEAD_STAMP(17)
    // ---- stage 1: dlogit, du (grad wrt bn3 output), gr = dlogit * relu(.) for dW2 -----------------------------
    float dl_sum = 0.f;
    struct S1 {
        float p, t, h, m;
    };
    batched<8, S1>(tid, B * HID, nt,   // one (sample, hidden unit) per thread: coalesced
                   [&](int idx) {
                       const int b = idx >> 5;
                       return S1{probs[b], dprobs != nullptr ? dprobs[b] : (float)a.labels[b], h[idx], a.m2 ? a.m2[idx] : 1.f};
                   },
                   [&](int idx, const S1 &v) {
                       const int j = idx & 31;
                       const float p = v.p;
                       float dlogit;
                       if (dprobs != nullptr) dlogit = v.t * p * (1.f - p);
                       else dlogit = (p - v.t) / (float)B;
                       if (j == 0) dl_sum += dlogit;
                       float u = fmaf(v.h, us[j], ut[j]);
                       const float m = v.m;
                       u *= m;
                       gr[idx] = dlogit * fmaxf(u, 0.f);
                       du[idx] = (u > 0.f) ? dlogit * w2s[j] * m : 0.f;
                   });
    dl_sum = wave_sum64(dl_sum);
    if ((tid & 63) == 0) redw[tid >> 6] = dl_sum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < nt / 64; ++w) s += redw[w];
        gr_out.dbias2[0] = s;
    }
    LAD_HEAD_STAMP(18)
    col_dot2(gr, nullptr, nullptr, nullptr, B, HID, gr_out.dW2, nullptr, scratch);
    col_dot2(du, h, mean3, istd3, B, HID, ca, cb, scratch);  // ca = sum du (dbeta3), cb = sum du * xhat3 (dgamma3)
    __syncthreads();
    if (tid < HID) {
        gr_out.db3[tid] = ca[tid];
        gr_out.dg3[tid] = cb[tid];
    }
    LAD_HEAD_STAMP(19)
    // ---- stage 2: dh = g3*istd3*(du - mean(du) - xhat3*mean(du*xhat3)) ---------------------------------------
    {
        const float ca_j = ca[jt] / (float)B, cb_j = cb[jt];
        batched<16, F2>(tid, B * HID, nt, [&](int idx) { return F2{h[idx], du[idx]}; },
                       [&](int idx, const F2 &v) {
                           const float xh = (v.a - mean3_j) * istd3_j;
                           dh[idx] = k3_j * (v.b - ca_j - xh * cb_j / (float)B);
                       });
    }
    __syncthreads();
    LAD_HEAD_STAMP(20)
    // ---- stage 4: dz[b][f] = m1[b][f] * sum_j W1[j][f] * dh[b][j] -- two threads per sample (one half of the features each), the
    //      sample's dh row in registers, a weight row one LDS broadcast (round 4; j ascending from zero, as before)
    {
        const int fh = (F + 1) / 2;
        #pragma unroll 1
        for (int t = tid; t < 2 * B; t += nt) {
            const int b = t >> 1, fa = (t & 1) * fh, fb = min(F, fa + fh);
            float dhr[HID];
            const float4 *dr = reinterpret_cast<const float4 *>(dh + (int64_t)b * HID);
#pragma unroll
            for (int j = 0; j < HID; j += 4) {
                const float4 v = dr[j >> 2];
                dhr[j] = v.x; dhr[j + 1] = v.y; dhr[j + 2] = v.z; dhr[j + 3] = v.w;
            }
            const float *mr = a.m1 ? a.m1 + (int64_t)b * F : nullptr;
            float *zr = dz + (int64_t)b * F;
            for (int f0 = fa; f0 < fb; f0 += 8) {
                float mv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) mv[k] = (mr != nullptr && f0 + k < fb) ? mr[f0 + k] : 1.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int f = f0 + k;
                    if (f < fb) {
                        float sacc = 0.f;
#pragma unroll
                        for (int j = 0; j < HID; ++j) sacc = fmaf(w1t[f * HID + j], dhr[j], sacc);
                        if (mr != nullptr) sacc *= mv[k];
                        zr[f] = sacc;
                    }
                }
            }
        }
    }
    LAD_HEAD_STAMP(21)
    // ---- stage 3: dW1[j][f] = sum_b dh[b][j] * z[b][f].  786 k multiply-adds fed from ONE CU's LDS: with an output per thread
    //      (two LDS reads per multiply-add) the LDS pipe alone took 40 us.  Now a thread owns a 4 (j) x 3 (f) tile of outputs and one of
    //      KS residue classes of the samples (b = g mod KS): 7 LDS values per 12 multiply-adds, a wave's dh reads are broadcasts.
    //      CH samples at a time are staged (dh, and z = the normalised, masked inputs), the next chunk's values are requested before
    //      the current one is summed.  Per output: each class sums its samples in ascending order, the classes are added in order
    //      g = 0 .. KS - 1 (a fixed order; before round 4 it was one ascending sum).
    constexpr int TJ = 4, TF = 3, DHS = HID + 4;                  // (dh_s row stride: 16-byte aligned tiles of four)
    const int ftiles = (F + TF - 1) / TF, ntile = (HID / TJ) * ftiles;
    int KS = 8;   // the largest power of two that the threads (ntile * KS <= nt) and the meeting place (KS * HID * F floats in z_s) allow
    while (KS > 1 && (ntile * KS > nt || KS * HID * F > CH * (MAX_F + 1))) KS >>= 1;
    const bool worker = tid < ntile * KS;
    const int g = tid / ntile, tile = tid - g * ntile, tj = tile / ftiles, tf = tile - tj * ftiles;   // lanes: f tile fastest
    float wacc[TJ][TF];
#pragma unroll
    for (int x = 0; x < TJ; ++x)
#pragma unroll
        for (int y = 0; y < TF; ++y) wacc[x][y] = 0.f;
    constexpr int NZ = (CH * MAX_F + HEAD_THREADS - 1) / HEAD_THREADS, ND = (CH * HID + HEAD_THREADS - 1) / HEAD_THREADS;
    F2 pz[NZ];
    float pd[ND];
    auto fetch = [&](int c0) {
        const int nb = min(CH, B - c0);
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
            const int idx = tid + k * nt;
            if (idx < nb * F) pz[k] = F2{pooled[(int64_t)c0 * F + idx], a.m1 ? a.m1[(int64_t)c0 * F + idx] : 1.f};
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + k * nt;
            if (idx < nb * HID) pd[k] = dh[(int64_t)c0 * HID + idx];
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < B; c0 += CH) {
        const int nb = min(CH, B - c0);
#pragma unroll
        for (int k = 0; k < NZ; ++k) {
            const int idx = tid + k * nt;
            if (idx < nb * F) {
                const int b = idx / F, f = idx - b * F;
                float z = fmaf(pz[k].a, zs[f], zt[f]);
                if (a.m1) z *= pz[k].b;
                z_s[b * (MAX_F + 1) + f] = z;
            }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + k * nt;
            if (idx < nb * HID) dh_s[(idx / HID) * DHS + (idx % HID)] = pd[k];
        }
        __syncthreads();
        if (c0 + CH < B) fetch(c0 + CH);
        if (worker) {
            for (int b = g; b < nb; b += KS) {   // (c0 is a multiple of CH and CH of KS: b = g mod KS over the whole batch)
                const float4 dv = *reinterpret_cast<const float4 *>(dh_s + b * DHS + tj * TJ);
                const float dd[TJ] = {dv.x, dv.y, dv.z, dv.w};
                float zz[TF];
#pragma unroll
                for (int y = 0; y < TF; ++y) zz[y] = z_s[b * (MAX_F + 1) + min(tf * TF + y, MAX_F - 1)];
#pragma unroll
                for (int x = 0; x < TJ; ++x)
#pragma unroll
                    for (int y = 0; y < TF; ++y) wacc[x][y] = fmaf(dd[x], zz[y], wacc[x][y]);
            }
        }
        __syncthreads();
    }
    // the KS classes meet in LDS (z_s is free now): part[g][j][f], summed in class order
    {
        float *part = z_s;   // (KS * HID * F <= CH * (MAX_F + 1) floats by the choice of KS; HID * MAX_F = 4096 fits for KS = 1)
        if (worker) {
#pragma unroll
            for (int x = 0; x < TJ; ++x)
#pragma unroll
                for (int y = 0; y < TF; ++y) {
                    const int f = tf * TF + y;
                    if (f < F) part[(g * HID + tj * TJ + x) * F + f] = wacc[x][y];
                }
        }
        __syncthreads();
        for (int idx = tid; idx < HID * F; idx += nt) {
            float sacc = part[idx];
            for (int q = 1; q < KS; ++q) sacc += part[q * HID * F + idx];
            gr_out.dW1[idx] = sacc;
        }
        __syncthreads();   // (z_s is the scratch of the column reductions below)
    }
    LAD_HEAD_STAMP(22)
    col_dot2(dh, nullptr, nullptr, nullptr, B, HID, gr_out.dbias1, nullptr, scratch);
    LAD_HEAD_STAMP(23)
    // ---- stage 5: bn2 backward -------------------------------------------------------------------------------------
    col_dot2(dz, pooled, mean2, istd2, B, F, ca, cb, scratch);
    __syncthreads();
    for (int f = tid; f < F; f += nt) {
        gr_out.db2[f] = ca[f];
        gr_out.dg2[f] = cb[f];
    }
    LAD_HEAD_STAMP(24)
    batched<12, F2>(tid, B * F, nt, [&](int idx) { return F2{pooled[idx], dz[idx]}; },
                   [&](int idx, const F2 &v) {
                       const int f = idx % F;
                       const float xh = (v.a - mu2[f]) * is2[f];
                       dpooled[idx] = kk2[f] * (v.b - ca[f] / (float)B - xh * cb[f] / (float)B);
                   });
    LAD_HEAD_STAMP(25)
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
    a.keep = 1.f; a.seed = 0; a.rng_counter = nullptr; a.nbt = nullptr; a.n_nbt = 0;
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

// lad_head_fwd_train that draws the dropout masks itself: drop1 / drop2 are OUTPUT buffers here ([batch][F] and [batch][32] floats, the
// masks lad_head_bwd then takes; F must be a multiple of 4), element values 0 or 1 / keep from Philox4x32-10 keyed by `seed` at draw
// number *rng_counter (a device int64 the kernel increments: a hipGraph replay draws fresh masks).  keep >= 1: no dropout, the buffers
// are not touched and may be NULL.  nbt (or NULL): n_nbt device int64 counters incremented by one -- the num_batches_tracked buffers
// of the model's BatchNorm layers, which a train-mode forward advances (torch.nn.BatchNorm*.forward).  Replaces the four launches of
// nn.Dropout's mask generation (models.py:232,235: bernoulli_ + div_ twice) and the counter increment by none.
extern "C" int lad_head_fwd_train_rng(const float *const *params, const float *pooled, int64_t batch, int32_t F, float *drop1, float *drop2,
                                      float keep, uint64_t seed, int64_t *rng_counter, int64_t *nbt, int32_t n_nbt, const int32_t *labels,
                                      float momentum, float *h, float *stats, float *probs, float *metrics, void *stream) {
    using namespace lad;
    LAD_REQUIRE(params && pooled && h && stats && probs && metrics, "lad_head_fwd_train_rng: null buffer");
    LAD_REQUIRE(batch >= 2, "lad_head_fwd_train_rng: batch statistics need at least two samples");
    const bool drop = keep < 1.0f;
    LAD_REQUIRE(!drop || (keep > 0.f && drop1 && drop2 && rng_counter && F % 4 == 0), "lad_head_fwd_train_rng: dropout needs mask buffers, a counter and F %% 4 == 0");
    LAD_REQUIRE(n_nbt >= 0 && n_nbt <= HEAD_THREADS - 64 && (n_nbt == 0 || nbt), "lad_head_fwd_train_rng: bad counter list");
    HeadArgs a = make_args(params, batch, F, drop ? drop1 : nullptr, drop ? drop2 : nullptr, labels, momentum);
    int rc = check_head(a);
    if (rc) return rc;
    a.keep = keep; a.seed = seed; a.rng_counter = drop ? (long long *)rng_counter : nullptr; a.nbt = (long long *)nbt; a.n_nbt = n_nbt;
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

#ifdef LAD_STAMP
extern "C" int lad_debug_read_head_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_head), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif
