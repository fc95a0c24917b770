// BatchNorm2d (train + eval) and the fused normalise / residual / ReLU element-wise passes, forward and
// backward, for PNHWC activations (layout: lad_device.h).  HBM-bound: every kernel moves 16 bytes per lane
// per access and touches each tensor exactly once.
//
// Replaces nn.BatchNorm2d + nn.ReLU + the residual add of ResidualBlock.forward (models.py:110-115) and of
// ResNetBigger's stem (models.py:224), and their autograd backward.
//   forward (train): conv epilogue wrote per-tile (sum, sumsq) partials -> bn_finalize (double accumulation:
//                    mean, biased var, invstd, scale/shift, running-stat update with momentum 0.1 and the
//                    unbiased variance, as torch does) -> bn_act (y = relu(x*scale+shift [+ residual]))
//   forward (eval):  bn_fold (scale/shift from running stats) applied inside the convolution epilogues
//   backward:        bn_bwd_reduce (sum dz, sum dz*xhat [, sum dz*xhat_shortcut]; dz = dy * (y > 0))
//                    -> bn_bwd_finalize (dgamma, dbeta, per-channel coefficients)
//                    -> bn_bwd_apply (dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) [, dz, dx_shortcut])
// Precision: BatchNorm backward is structurally ill-conditioned (sum dx = 0 and sum dx*xhat = 0 hold only if the
// per-channel constants are exact), so a constant rounded to fp32 becomes a *systematic* bias of every element of a
// channel, which the next layer's per-channel sums amplify by the element count.  mean, invstd, mean(dz) and
// mean(dz*xhat) are therefore produced in double and carried as (hi, lo) float pairs into the element-wise pass.
// Border positions: gradients arriving there are zero by construction, so they drop out of every sum; the
// element-wise passes write zeros there (RowGeom), keeping the zero-border invariant of lad_device.h.
#include "lad_common.h"
#include "lad_device.h"
#include "lad_bn_math.h"

namespace {
using namespace lad;

constexpr float BN_EPS = 1e-5f;
constexpr int THREADS = 256;

// coef layout per BN layer: float[6][C] = scale, shift, mean, invstd, mean_lo, invstd_lo  (x_hi + x_lo = double value)
// blockIdx.y picks the layer: a stride-2 block's conv1 and its shortcut convolution leave their sums in one launch, and their two
// BatchNorm layers are finalized in one (lad_bn_finalize_pair).
struct FinSet {
    const float *partials, *gamma, *beta;
    float *running_mean, *running_var, *coef;
};
struct FinSets {
    FinSet s[2];
};
__global__ void bn_finalize_kernel(FinSets sets, int64_t n_tiles, int C, double count, float momentum) {
    const FinSet f = sets.s[blockIdx.y];
    const float *__restrict__ partials = f.partials;
    float *__restrict__ coef = f.coef;
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    // A lane's walk is a chain of L2 round trips (a row of partials per lane and step, 4 bytes of it): sixteen / eight loads in flight
    // instead of the two the plain loop gets, added in the same order (4,692 tiles at 50x22, batch 512: 18 round trips -> 4).
    int64_t t = threadIdx.x;
    const int64_t bd = blockDim.x;
    for (; t + 7 * bd < n_tiles; t += 8 * bd) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a[u] = partials[((t + u * bd) * 2 + 0) * C + c];
            b[u] = partials[((t + u * bd) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s1 += (double)a[u];
            s2 += (double)b[u];
        }
    }
    for (; t + 3 * bd < n_tiles; t += 4 * bd) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = partials[((t + u * bd) * 2 + 0) * C + c];
            b[u] = partials[((t + u * bd) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s1 += (double)a[u];
            s2 += (double)b[u];
        }
    }
    for (; t < n_tiles; t += bd) {
        s1 += (double)partials[(t * 2 + 0) * C + c];
        s2 += (double)partials[(t * 2 + 1) * C + c];
    }
    __shared__ double red[2][THREADS / 64];
    s1 = wave_sum64d(s1);
    s2 = wave_sum64d(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd_d = 1.0 / sqrt(var + (double)BN_EPS);
        const float invstd = (float)invstd_d;
        const float scale = f.gamma[c] * invstd;
        coef[0 * C + c] = scale;
        coef[1 * C + c] = f.beta[c] - (float)mean * scale;
        coef[2 * C + c] = (float)mean;
        coef[3 * C + c] = invstd;
        coef[4 * C + c] = (float)(mean - (double)(float)mean);
        coef[5 * C + c] = (float)(invstd_d - (double)invstd);
        if (f.running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            f.running_mean[c] = (1.0f - momentum) * f.running_mean[c] + momentum * (float)mean;
            f.running_var[c] = (1.0f - momentum) * f.running_var[c] + momentum * (float)unbiased;
        }
    }
}

// Two-level form for the large layers (18,182 tiles at 100x44, batch 512): the one-block-per-channel walk above reads
// 4 bytes out of every 512-byte row (50 us); here level 1 sums row slices with whole rows per wave instruction and leaves
// each slice's 2C double sums IN PLACE, in the first two rows of its own slice (which no other workgroup reads);
// level 2 -- one workgroup over the <= 64 slice results -- is run by the workgroup that FINISHES LAST, inside the same launch
// (round 6: it was a launch of its own, 13 per step).  Fixed summation order at both levels: bit-reproducible, and the same
// bits as the two-launch form.
constexpr int FIN_SLICES = 64;
// From how many per-tile partials on.  Round 6, second session: 2,048 (was 8,192).  The one-level kernel's workgroup-per-channel walk
// fetches a 64-byte line per lane and load for 4 bytes of it, so its time is the partials' LINES through one CU's address path, per
// channel: 17 us for the 4,692 x 2 x 32 sums of a 50x22 layer at batch 512, whatever the loads in flight (measured with 16).
constexpr int64_t TWO_LEVEL_MIN_TILES = 2048;
// Which workgroup is last: a ticket per launch (common.hip: launch_ticket).
struct FinTail {
    int backward;                      // 0: forward statistics -> coef (+ running statistics); 1: backward sums -> dgamma, dbeta, bcoef
    double count;
    const float *gamma, *beta, *coef;  // beta: forward; coef: backward (the layer's forward coefficients)
    float *running_mean, *running_var;
    float momentum;
    float *out;                        // forward: coef float[6][C]; backward: bcoef float[8][C]
    float *dgamma, *dbeta;
    unsigned int *ticket;
};
__global__ __launch_bounds__(THREADS) void bn_slice_sum_kernel(float *__restrict__ partials, int64_t n_tiles, int C, int64_t rows_per_slice,
                                                               FinTail ft) {
    const int cols = 2 * C, q4 = cols / 4, phases = THREADS / q4;  // a thread owns 4 consecutive columns of every phases-th row
    const int cq = threadIdx.x % q4, ph = threadIdx.x / q4;
    const int64_t lo = (int64_t)blockIdx.x * rows_per_slice;
    const int64_t hi = blockIdx.x == gridDim.x - 1 ? n_tiles : lo + rows_per_slice;  // the last slice takes the remainder
    const float4 *src = reinterpret_cast<const float4 *>(partials) + cq;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t t = lo + ph;
    for (; t + 3 * phases < hi; t += 4 * phases) {  // four loads in flight, summed in row order
        const float4 a = src[t * q4], b = src[(t + phases) * q4], c = src[(t + 2 * phases) * q4], d = src[(t + 3 * phases) * q4];
        s0 += (double)a.x; s1 += (double)a.y; s2 += (double)a.z; s3 += (double)a.w;
        s0 += (double)b.x; s1 += (double)b.y; s2 += (double)b.z; s3 += (double)b.w;
        s0 += (double)c.x; s1 += (double)c.y; s2 += (double)c.z; s3 += (double)c.w;
        s0 += (double)d.x; s1 += (double)d.y; s2 += (double)d.z; s3 += (double)d.w;
    }
    for (; t < hi; t += phases) {
        const float4 a = src[t * q4];
        s0 += (double)a.x; s1 += (double)a.y; s2 += (double)a.z; s3 += (double)a.w;
    }
    __shared__ double red[THREADS][4];
    __shared__ int last_s;
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2; red[threadIdx.x][3] = s3;
    __syncthreads();  // also: every row of the slice has been read before its head is overwritten
    if (threadIdx.x < cols) {
        const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
        double tsum = 0.0;
        for (int p = 0; p < phases; ++p) tsum += red[p * q4 + q][e];
        // device-scope store: written through to where every XCD sees it.  (A __threadfence() here instead writes back and invalidates the
        // XCD's whole L2 -- eight L2s on this part -- in every workgroup: the launch took 30 us instead of the 16 us of the two it replaces.)
        __hip_atomic_store(reinterpret_cast<double *>(partials + lo * cols) + threadIdx.x, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- level 2, by the workgroup that takes the last ticket: a slice's sums have arrived (vmcnt) before its ticket is taken
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_s = __hip_atomic_fetch_add(ft.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last_s) return;
    if (threadIdx.x == 0) __hip_atomic_store(ft.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int slices = (int)gridDim.x;
    double *tot = &red[0][0];   // (free again: every thread is past the sums above)
    if (threadIdx.x < cols) {
        double tt = 0.0;
        double *hd = reinterpret_cast<double *>(partials) + threadIdx.x;   // (device-scope loads: the other XCDs' stores, not this L2's copy)
        const int64_t step = rows_per_slice * cols / 2;  // doubles between slice heads
        int b = 0;
        // device-scope loads are round trips past this XCD's L2: 32 in flight (two trips for 64 slices instead of eight), summed in slice order
        for (; b + 32 <= slices; b += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = __hip_atomic_load(hd + (b + u) * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 32; ++u) tt += v[u];
        }
        for (; b + 8 <= slices; b += 8) {  // eight loads in flight, summed in slice order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(hd + (b + u) * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) tt += v[u];
        }
        for (; b < slices; ++b) tt += __hip_atomic_load(hd + b * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[threadIdx.x] = tt;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= C) return;
    if (!ft.backward) {
        float *coef = ft.out;
        const double count = ft.count;
        const double mean = tot[c] / count;
        double var = tot[C + c] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd_d = 1.0 / sqrt(var + (double)BN_EPS);
        const float invstd = (float)invstd_d;
        const float scale = ft.gamma[c] * invstd;
        coef[0 * C + c] = scale;
        coef[1 * C + c] = ft.beta[c] - (float)mean * scale;
        coef[2 * C + c] = (float)mean;
        coef[3 * C + c] = invstd;
        coef[4 * C + c] = (float)(mean - (double)(float)mean);
        coef[5 * C + c] = (float)(invstd_d - (double)invstd);
        if (ft.running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            ft.running_mean[c] = (1.0f - ft.momentum) * ft.running_mean[c] + ft.momentum * (float)mean;
            ft.running_var[c] = (1.0f - ft.momentum) * ft.running_var[c] + ft.momentum * (float)unbiased;
        }
    } else {
        // bcoef layout: see bn_bwd_finalize_kernel
        float *bcoef = ft.out;
        const double t0 = tot[c], t1 = tot[C + c];
        ft.dbeta[c] = (float)t0;
        ft.dgamma[c] = (float)t1;
        const double k2 = t0 / ft.count, k3 = t1 / ft.count;
        bcoef[0 * C + c] = ft.gamma[c] * ft.coef[3 * C + c];
        bcoef[1 * C + c] = (float)k2;
        bcoef[2 * C + c] = (float)k3;
        bcoef[4 * C + c] = (float)(k2 - (double)(float)k2);
        bcoef[6 * C + c] = (float)(k3 - (double)(float)k3);
    }
}

// Geometry of an element-wise pass: a workgroup walks whole padded image rows (b, yp) so that "is this a border
// position" costs one wave-uniform test per row plus a shift per element.  Border positions are WRITTEN AS ZERO:
// every activation / gradient tensor in HBM has a zero border ring, which is what lets the MFMA kernels stage their
// operands without any per-row bounds logic (lad_device.h).
struct RowGeom {
    int64_t n_img_rows;  // batch * Hp
    int Hp, Wp;
};

// Rows of an element-wise pass per workgroup.  A padded image row of a small layer is a fraction of a workgroup (48 float4 at 16
// channels x 11 columns, 28 at 16 x 6): with a row per workgroup four of five lanes idled.  When two or more rows fit, lane t works
// on row blockIdx * rows + t / per_row at float4 t % per_row of it -- per_row is a multiple of C / 4, so a lane's channel quad (and
// its place among the 16 lanes of a 64-channel pixel row) is what it was.  Same arithmetic per element: same bits.
struct RowPack {
    int rows, sub, f0, fstep;   // rows per workgroup pass; this lane's sub-row (>= rows: idle), first float4 and stride inside a row
};
__device__ __forceinline__ RowPack row_pack(int per_row) {
    RowPack p{1, 0, (int)threadIdx.x, (int)blockDim.x};
    if (2 * per_row <= (int)blockDim.x) {
        p.rows = (int)blockDim.x / per_row;
        p.sub = (int)threadIdx.x / per_row;
        p.f0 = (int)threadIdx.x - p.sub * per_row;
        p.fstep = per_row;
    }
    return p;
}

// y = act(x*scale + shift + residual), residual = none | res | res*rscale + rshift
// BITS (C = 64 only): also leave the sign bits of y, one uint64 per pixel row (lad_bn_math.h), for the backward pass.
template <int RES, bool BITS = false>  // 0 none, 1 identity, 2 affine (shortcut BatchNorm)
__global__ void bn_act_kernel(const float4 *__restrict__ x, const float *__restrict__ coef,
                              const float4 *__restrict__ res, const float *__restrict__ rcoef, float4 *__restrict__ y,
                              RowGeom g, int C, int c4shift, int relu, unsigned long long *__restrict__ bits = nullptr) {
    // a thread's channel quad never changes: f advances by blockDim.x float4, a multiple of C/4 (launcher) -> the
    // per-channel coefficients live in registers, the loop issues only the tensor loads and the store
    const int c = (threadIdx.x * 4) & (C - 1);
    const int per_row = g.Wp << c4shift;  // float4 per padded image row
    const float4 sc = *reinterpret_cast<const float4 *>(coef + c);
    const float4 sh = *reinterpret_cast<const float4 *>(coef + C + c);
    float4 rs = sc, rh = sh;
    if (RES == 2) {
        rs = *reinterpret_cast<const float4 *>(rcoef + c);
        rh = *reinterpret_cast<const float4 *>(rcoef + C + c);
    }
    const RowPack rp = row_pack(per_row);
    for (int64_t r = (int64_t)blockIdx.x * rp.rows + rp.sub; rp.sub < rp.rows && r < g.n_img_rows; r += (int64_t)gridDim.x * rp.rows) {
        const int yp = (int)(r % g.Hp);
        const bool border_row = (yp == 0);
        const int64_t base = r * per_row;
        for (int f = rp.f0; f < per_row; f += rp.fstep) {
            const int64_t idx = base + f;
            const int xp = f >> c4shift;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(border_row | (xp == 0))) {
                const float4 v = x[idx];
                o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
                if (RES == 1) {
                    const float4 rr = res[idx];
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                } else if (RES == 2) {
                    const float4 rr = res[idx];
                    o.x += fmaf(rr.x, rs.x, rh.x); o.y += fmaf(rr.y, rs.y, rh.y);
                    o.z += fmaf(rr.z, rs.z, rh.z); o.w += fmaf(rr.w, rs.w, rh.w);
                }
                if (relu) {
                    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                }
            }
            y[idx] = o;
            if (BITS) {  // the 16 lanes of a pixel row enter and leave the loop together (per_row is a multiple of 16)
                const unsigned long long word = pack_sign_bits(o, threadIdx.x & 63);
                if ((threadIdx.x & 15) == 0) bits[idx >> 4] = word;
            }
        }
    }
}

// partial sums over a slice of rows: out[blk][k][c], k: 0 = sum dz, 1 = sum dz*xhat, 2 = sum dz*xhat_s
template <int C, bool SHORT>
__global__ __launch_bounds__(THREADS) void bn_bwd_reduce_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y,
                                                                const float4 *__restrict__ x, const float *__restrict__ coef,
                                                                const float4 *__restrict__ xs, const float *__restrict__ scoef,
                                                                float *__restrict__ partials, int64_t rows, int relu,
                                                                const unsigned long long *__restrict__ bits) {
    constexpr int C4 = C / 4;
    constexpr int RP = THREADS / C4;  // row-parts per block
    constexpr int K = SHORT ? 3 : 2;
    const int c4 = threadIdx.x % C4, rp = threadIdx.x / C4;
    const Norm4 nm = load_norm(coef, C, c4 * 4);
    Norm4 sn = nm;
    if (SHORT) sn = load_norm(scoef, C, c4 * 4);
    const float4 fscale = *reinterpret_cast<const float4 *>(coef + c4 * 4);
    const float4 fshift = *reinterpret_cast<const float4 *>(coef + C + c4 * 4);
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0;
    for (int64_t row = (int64_t)blockIdx.x * RP + rp; row < rows; row += (int64_t)gridDim.x * RP) {
        const int64_t idx = row * C4 + c4;
        float4 d = dy[idx];
        const float4 xv = x[idx];
        if (relu == 1) {
            const float4 yy = y[idx];
            d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
            d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
        } else if (relu == 2) {
            d = mask_from_x(d, xv, fscale, fshift);
        } else if (relu == 3) {
            d = mask_from_bits(d, bits[row], c4);
        }
        a0.x += d.x; a0.y += d.y; a0.z += d.z; a0.w += d.w;
        const float4 xh = xhat4(xv, nm);
        a1.x = fmaf(d.x, xh.x, a1.x); a1.y = fmaf(d.y, xh.y, a1.y);
        a1.z = fmaf(d.z, xh.z, a1.z); a1.w = fmaf(d.w, xh.w, a1.w);
        if (SHORT) {
            const float4 sh = xhat4(xs[idx], sn);
            a2.x = fmaf(d.x, sh.x, a2.x); a2.y = fmaf(d.y, sh.y, a2.y);
            a2.z = fmaf(d.z, sh.z, a2.z); a2.w = fmaf(d.w, sh.w, a2.w);
        }
    }
    __shared__ float red[RP][K][C];
    *reinterpret_cast<float4 *>(&red[rp][0][c4 * 4]) = a0;
    *reinterpret_cast<float4 *>(&red[rp][1][c4 * 4]) = a1;
    if (SHORT) *reinterpret_cast<float4 *>(&red[rp][2][c4 * 4]) = a2;
    __syncthreads();
    for (int t = threadIdx.x; t < K * C; t += THREADS) {
        const int k = t / C, c = t - k * C;
        float s = 0.f;
#pragma unroll 4
        for (int p = 0; p < RP; ++p) s += red[p][k][c];
        partials[((int64_t)blockIdx.x * K + k) * C + c] = s;
    }
}

// bcoef layout: float[8][C] = k1 (gamma*invstd), k2 (mean dz), k3 (mean dz*xhat), j1, k2_lo, j3, k3_lo, j3_lo
// (j1, j3: the same quantities for the shortcut BatchNorm; *_lo: low halves of the double values)
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ partials, int groups, int K, int C, double count,
                                       const float *__restrict__ gamma, const float *__restrict__ coef,
                                       const float *__restrict__ sgamma, const float *__restrict__ scoef,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta,
                                       float *__restrict__ dsgamma, float *__restrict__ dsbeta, float *__restrict__ bcoef) {
    const int c = blockIdx.x;
    double s[3] = {0.0, 0.0, 0.0};
    // (as in bn_finalize_kernel: twelve loads in flight instead of one, added in the same order.  K = 2: the third column re-reads the
    // second and its sum is not used -- a load that is always issued instead of a branch around it)
    int g = threadIdx.x;
    const int bd = blockDim.x;
    const int k2 = K - 1;   // 1 or 2
    for (; g + 3 * bd < groups; g += 4 * bd) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *row = partials + (int64_t)(g + u * bd) * K * C + c;
            v[u][0] = row[0];
            v[u][1] = row[C];
            v[u][2] = row[k2 * C];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[0] += (double)v[u][0];
            s[1] += (double)v[u][1];
            s[2] += (double)v[u][2];
        }
    }
    for (; g < groups; g += bd) {
        const float *row = partials + (int64_t)g * K * C + c;
        const float v0 = row[0], v1 = row[C], v2 = row[k2 * C];
        s[0] += (double)v0;
        s[1] += (double)v1;
        s[2] += (double)v2;
    }
    __shared__ double red[3][THREADS / 64];
    for (int k = 0; k < 3; ++k) {
        const double v = wave_sum64d(s[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < 3; ++k)
            for (int w = 0; w < THREADS / 64; ++w) t[k] += red[k][w];
        dbeta[c] = (float)t[0];
        dgamma[c] = (float)t[1];
        const double k2 = t[0] / count, k3 = t[1] / count;
        bcoef[0 * C + c] = gamma[c] * coef[3 * C + c];
        bcoef[1 * C + c] = (float)k2;
        bcoef[2 * C + c] = (float)k3;
        bcoef[4 * C + c] = (float)(k2 - (double)(float)k2);
        bcoef[6 * C + c] = (float)(k3 - (double)(float)k3);
        if (K == 3) {
            const double j3 = t[2] / count;
            dsbeta[c] = (float)t[0];
            dsgamma[c] = (float)t[2];
            bcoef[3 * C + c] = sgamma[c] * scoef[3 * C + c];
            bcoef[5 * C + c] = (float)j3;
            bcoef[7 * C + c] = (float)(j3 - (double)(float)j3);
        }
    }
}

// dx = k1*(dz - k2 - xhat*k3); optional dz_out (identity shortcut) or dxs (shortcut BatchNorm input gradient).
// Border positions are written as zero (see RowGeom).
template <int MODE>  // 0: dx only, 1: dx + dz_out, 2: dx + dxs
__global__ void bn_bwd_apply_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ y, const float4 *__restrict__ x,
                                    const float *__restrict__ coef, const float *__restrict__ bcoef,
                                    const float4 *__restrict__ xs, const float *__restrict__ scoef, float4 *__restrict__ dx,
                                    float4 *__restrict__ aux, RowGeom g, int C, int c4shift, int relu,
                                    const unsigned long long *__restrict__ bits) {
    const int c = (threadIdx.x * 4) & (C - 1);  // invariant per thread (see bn_act_kernel): coefficients in registers
    const int per_row = g.Wp << c4shift;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 fsc = *reinterpret_cast<const float4 *>(coef + c), fsh = *reinterpret_cast<const float4 *>(coef + C + c);
    const Norm4 nm = load_norm(coef, C, c);
    const float4 k1 = *reinterpret_cast<const float4 *>(bcoef + 0 * C + c);
    const float4 k2 = *reinterpret_cast<const float4 *>(bcoef + 1 * C + c);
    const float4 k3 = *reinterpret_cast<const float4 *>(bcoef + 2 * C + c);
    const float4 k2l = *reinterpret_cast<const float4 *>(bcoef + 4 * C + c);
    const float4 k3l = *reinterpret_cast<const float4 *>(bcoef + 6 * C + c);
    Norm4 sn = nm;
    float4 j1 = k1, j3 = k3, j3l = k3l;
    if (MODE == 2) {
        sn = load_norm(scoef, C, c);
        j1 = *reinterpret_cast<const float4 *>(bcoef + 3 * C + c);
        j3 = *reinterpret_cast<const float4 *>(bcoef + 5 * C + c);
        j3l = *reinterpret_cast<const float4 *>(bcoef + 7 * C + c);
    }
    const RowPack rp = row_pack(per_row);
    for (int64_t r = (int64_t)blockIdx.x * rp.rows + rp.sub; rp.sub < rp.rows && r < g.n_img_rows; r += (int64_t)gridDim.x * rp.rows) {
        const int yp = (int)(r % g.Hp);
        const bool border_row = (yp == 0);
        const int64_t base = r * per_row;
        for (int f = rp.f0; f < per_row; f += rp.fstep) {
            const int64_t idx = base + f;
            const int xp = f >> c4shift;
            if (border_row | (xp == 0)) {
                dx[idx] = zero;
                if (MODE != 0) aux[idx] = zero;
                continue;
            }
            float4 d = dy[idx];
            const float4 xv = x[idx];
            if (relu == 1) {
                const float4 yy = y[idx];
                d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
                d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
            } else if (relu == 2) {
                d = mask_from_x(d, xv, fsc, fsh);
            } else if (relu == 3) {
                d = mask_from_bits(d, bits[idx >> 4], threadIdx.x & 15);
            }
            const float4 xh = xhat4(xv, nm);
            float4 o;
            o.x = bn_dx1(d.x, xh.x, k1.x, k2.x, k2l.x, k3.x, k3l.x);
            o.y = bn_dx1(d.y, xh.y, k1.y, k2.y, k2l.y, k3.y, k3l.y);
            o.z = bn_dx1(d.z, xh.z, k1.z, k2.z, k2l.z, k3.z, k3l.z);
            o.w = bn_dx1(d.w, xh.w, k1.w, k2.w, k2l.w, k3.w, k3l.w);
            dx[idx] = o;
            if (MODE == 1) {
                aux[idx] = d;
            } else if (MODE == 2) {
                const float4 sh = xhat4(xs[idx], sn);
                float4 sv;
                sv.x = bn_dx1(d.x, sh.x, j1.x, k2.x, k2l.x, j3.x, j3l.x);
                sv.y = bn_dx1(d.y, sh.y, j1.y, k2.y, k2l.y, j3.y, j3l.y);
                sv.z = bn_dx1(d.z, sh.z, j1.z, k2.z, k2l.z, j3.z, j3l.z);
                sv.w = bn_dx1(d.w, sh.w, j1.w, k2.w, k2l.w, j3.w, j3l.w);
                aux[idx] = sv;
            }
        }
    }
}

// Workgroups of an element-wise pass.  Streams of this size run fastest as MANY short workgroups (tools/experiments/
// stream_rate.hip: 6.0 TB/s with one trip per thread against 5.4 grid-striding over 8 k workgroups), so a large layer gets one
// workgroup per padded image row (720 float4 at 64 x 44: three trips per thread); rows shorter than two trips keep the
// grid-stride form, where a workgroup per row would leave most of its threads idle.
unsigned row_grid(int64_t n_img_rows, int float4_per_row) {
    const int rows = 2 * float4_per_row <= THREADS ? THREADS / float4_per_row : 1;   // row_pack
    return (unsigned)std::min<int64_t>(lad::ceil_div(n_img_rows, (int64_t)rows), float4_per_row >= 2 * THREADS ? 256 * 256 : 256 * 32);
}
int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}
RowGeom make_row_geom(int64_t batch, int H, int W) {
    RowGeom g;
    g.Hp = H + 1;  // shared-border layout (lad_device.h): border row 0 / border column 0 only; the tail is never touched here
    g.Wp = W + 1;
    g.n_img_rows = batch * g.Hp;
    return g;
}
constexpr int BWD_GROUPS = 1024;
}  // namespace

extern "C" int lad_bn_finalize(float *stat_partials, int64_t n_tiles, int32_t channels, int64_t count,
                               const float *gamma, const float *beta, float *running_mean, float *running_var,
                               float momentum, float *coef, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stat_partials && gamma && beta && coef, "lad_bn_finalize: null buffer");
    LAD_REQUIRE(channels > 0 && n_tiles > 0 && count > 0, "lad_bn_finalize: bad sizes");
    LAD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "lad_bn_finalize: running stats must come in pairs");
    if (n_tiles >= TWO_LEVEL_MIN_TILES && 2 * channels <= 128 && THREADS % (channels / 2) == 0) {
        const int64_t rps = ceil_div(n_tiles, FIN_SLICES);  // >= 16 rows: room for the slice's 2C doubles (two rows)
        const int slices = (int)(n_tiles / rps);            // every slice holds >= rps rows
        const FinTail ft{0, (double)count, gamma, beta, nullptr, running_mean, running_var, momentum, coef, nullptr, nullptr, launch_ticket()};
        LAD_REQUIRE(ft.ticket, "lad_bn_finalize: no ticket for the two-level sum");
        hipLaunchKernelGGL(bn_slice_sum_kernel, dim3(slices), dim3(THREADS), 0, (hipStream_t)stream, stat_partials, n_tiles, channels, rps, ft);
        return check_launch("bn_slice_sum_kernel");
    }
    const FinSets sets{{{stat_partials, gamma, beta, running_mean, running_var, coef}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}}};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(channels, 1), dim3(THREADS), 0, (hipStream_t)stream, sets, n_tiles, channels, (double)count, momentum);
    return check_launch("bn_finalize_kernel");
}

// lad_bn_finalize for TWO layers whose sums have the same shape (a stride-2 block's bn1 and its shortcut BatchNorm) in one launch;
// the same results as two calls.
extern "C" int lad_bn_finalize_pair(float *stat_partials_a, float *stat_partials_b, int64_t n_tiles, int32_t channels, int64_t count,
                                    const float *gamma_a, const float *beta_a, float *running_mean_a, float *running_var_a, float *coef_a,
                                    const float *gamma_b, const float *beta_b, float *running_mean_b, float *running_var_b, float *coef_b,
                                    float momentum, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stat_partials_a && stat_partials_b && gamma_a && beta_a && coef_a && gamma_b && beta_b && coef_b, "lad_bn_finalize_pair: null buffer");
    LAD_REQUIRE(channels > 0 && n_tiles > 0 && count > 0, "lad_bn_finalize_pair: bad sizes");
    LAD_REQUIRE((running_mean_a == nullptr) == (running_var_a == nullptr) && (running_mean_b == nullptr) == (running_var_b == nullptr),
                "lad_bn_finalize_pair: running stats must come in pairs");
    if (n_tiles >= TWO_LEVEL_MIN_TILES && 2 * channels <= 128 && THREADS % (channels / 2) == 0) {   // the two-level form: one launch each
        const int rc = lad_bn_finalize(stat_partials_a, n_tiles, channels, count, gamma_a, beta_a, running_mean_a, running_var_a, momentum, coef_a, stream);
        return rc ? rc : lad_bn_finalize(stat_partials_b, n_tiles, channels, count, gamma_b, beta_b, running_mean_b, running_var_b, momentum, coef_b, stream);
    }
    const FinSets sets{{{stat_partials_a, gamma_a, beta_a, running_mean_a, running_var_a, coef_a},
                        {stat_partials_b, gamma_b, beta_b, running_mean_b, running_var_b, coef_b}}};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(channels, 2), dim3(THREADS), 0, (hipStream_t)stream, sets, n_tiles, channels, (double)count, momentum);
    return check_launch("bn_finalize_kernel");
}

// eval-mode fold of BatchNorm(conv(x) + conv_bias) into a per-channel affine on the bare convolution
__global__ void bn_fold_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ running_mean, const float *__restrict__ running_var,
                               const float *__restrict__ conv_bias, int C, float *__restrict__ scale, float *__restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float s = gamma[c] / sqrtf(running_var[c] + BN_EPS);
        const float b = conv_bias != nullptr ? conv_bias[c] : 0.0f;
        scale[c] = s;
        shift[c] = fmaf(b - running_mean[c], s, beta[c]);
    }
}

extern "C" int lad_bn_fold(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                           const float *conv_bias, int32_t channels, float *scale, float *shift, void *stream) {
    using namespace lad;
    LAD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && channels > 0, "lad_bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)ceil_div(channels, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, conv_bias, channels, scale, shift);
    return check_launch("bn_fold_kernel");
}

namespace {
int bn_act_impl(const float *x, const float *coef, const float *res, const float *res_coef, float *y, unsigned long long *bits,
                int64_t batch, int32_t H, int32_t W, int32_t channels, int32_t relu, void *stream, const char *who) {
    using namespace lad;
    LAD_REQUIRE(x && coef && y, "%s: null buffer", who);
    LAD_REQUIRE(channels >= 4 && channels <= 4 * THREADS && (channels & (channels - 1)) == 0, "%s: channels must be a power of two in 4..1024", who);
    LAD_REQUIRE(res != nullptr || res_coef == nullptr, "%s: res_coef without res", who);
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "%s: bad geometry", who);
    LAD_REQUIRE(bits == nullptr || channels == 64, "%s: sign bits are kept for 64-channel activations only", who);
    if (batch == 0) return LAD_OK;
    const RowGeom g = make_row_geom(batch, H, W);
    const int sh = log2_exact(channels / 4);
    const dim3 grid(row_grid(g.n_img_rows, g.Wp << sh)), block(THREADS);
    hipStream_t st = (hipStream_t)stream;
#define LAD_ACT(RES, R, RC)                                                                                                   \
    do {                                                                                                                      \
        if (bits)                                                                                                             \
            hipLaunchKernelGGL((bn_act_kernel<RES, true>), grid, block, 0, st, (const float4 *)x, coef, (const float4 *)(R), RC, \
                               (float4 *)y, g, channels, sh, relu, bits);                                                     \
        else                                                                                                                  \
            hipLaunchKernelGGL((bn_act_kernel<RES, false>), grid, block, 0, st, (const float4 *)x, coef, (const float4 *)(R), RC, \
                               (float4 *)y, g, channels, sh, relu, nullptr);                                                  \
    } while (0)
    if (res == nullptr)
        LAD_ACT(0, nullptr, nullptr);
    else if (res_coef == nullptr)
        LAD_ACT(1, res, nullptr);
    else
        LAD_ACT(2, res, res_coef);
#undef LAD_ACT
    return check_launch("bn_act_kernel");
}
}  // namespace

extern "C" int lad_bn_act(const float *x, const float *coef, const float *res, const float *res_coef, float *y,
                          int64_t batch, int32_t H, int32_t W, int32_t channels, int32_t relu, void *stream) {
    return bn_act_impl(x, coef, res, res_coef, y, nullptr, batch, H, W, channels, relu, stream, "lad_bn_act");
}

extern "C" int lad_bn_act_bits(const float *x, const float *coef, const float *res, const float *res_coef, float *y,
                               uint64_t *y_bits, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(y_bits, "lad_bn_act_bits: null sign-bit buffer");
    return bn_act_impl(x, coef, res, res_coef, y, (unsigned long long *)y_bits, batch, H, W, channels, 1, stream, "lad_bn_act_bits");
}

extern "C" int64_t lad_bn_bwd_workspace_floats(int32_t channels) { return (int64_t)BWD_GROUPS * 3 * channels; }

// Full BatchNorm(+ReLU)(+shortcut) backward.  mode: 0 dx only; 1 also write dz to `aux` (identity shortcut);
// 2 also write the shortcut-BN input gradient to `aux` (needs xs, scoef, sgamma, dsgamma, dsbeta).
namespace {
int bn_bwd_impl(const float *dy, const float *y, const unsigned long long *bits, const float *x, const float *coef, const float *gamma,
                const float *xs, const float *scoef, const float *sgamma, float *dx, float *aux,
                float *dgamma, float *dbeta, float *dsgamma, float *dsbeta, float *workspace, float *bcoef,
                float *pre_partials, int64_t pre_tiles, int64_t batch, int32_t H, int32_t W, int32_t channels,
                int32_t relu, int32_t mode, void *stream) {
    using namespace lad;
    LAD_REQUIRE(relu != 3 || (bits && channels == 64), "lad_bn_bwd_bits: sign bits are kept for 64-channel activations only");
    LAD_REQUIRE(dy && coef && gamma && dgamma && dbeta && workspace && bcoef, "lad_bn_bwd: null buffer");
    LAD_REQUIRE(x || (pre_partials && !dx), "lad_bn_bwd: x may only be omitted when the sums are given and nothing is applied");
    LAD_REQUIRE(dx || mode == 0, "lad_bn_bwd: dx may only be omitted (sums and coefficients only) in mode 0");
    LAD_REQUIRE(relu >= 0 && relu <= 3, "lad_bn_bwd: relu must be 0 (none), 1 (mask from y), 2 (mask recomputed from x) or 3 (sign bits: lad_bn_bwd_bits)");
    LAD_REQUIRE(relu != 1 || y, "lad_bn_bwd: relu = 1 needs y");
    LAD_REQUIRE(relu != 2 || mode == 0, "lad_bn_bwd: relu = 2 (mask recomputed from x) is for the residual-free BatchNorm only");
    LAD_REQUIRE(mode >= 0 && mode <= 2, "lad_bn_bwd: bad mode");
    LAD_REQUIRE(mode == 0 || aux, "lad_bn_bwd: mode needs aux");
    LAD_REQUIRE(mode != 2 || (xs && scoef && sgamma && dsgamma && dsbeta), "lad_bn_bwd: mode 2 needs the shortcut tensors");
    LAD_REQUIRE(channels == 16 || channels == 32 || channels == 64, "lad_bn_bwd: channels must be 16, 32 or 64");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_bn_bwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const RowGeom rg = make_row_geom(batch, H, W);
    const int64_t rows = rg.n_img_rows * rg.Wp;
    const int64_t count = batch * H * W;
    hipStream_t st = (hipStream_t)stream;
    const int rp = THREADS / (channels / 4);
    int groups = (int)std::min<int64_t>(BWD_GROUPS, ceil_div(rows, rp));
    const bool sh = mode == 2;
    const float *sums = workspace;
    if (pre_partials != nullptr) {
        // the producer of dy already left the per-tile sums (lad_conv_fwd_bnstat): no pass over the tensors
        LAD_REQUIRE(!sh && pre_tiles >= 1 && pre_tiles < (1 << 30), "lad_bn_bwd: pre-computed sums are for modes 0 and 1");
        sums = pre_partials;
        groups = (int)pre_tiles;
    } else {
#define LAD_RED(CC)                                                                                                  \
    if (channels == CC) {                                                                                            \
        if (sh)                                                                                                      \
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<CC, true>), dim3(groups), dim3(THREADS), 0, st, (const float4 *)dy, \
                               (const float4 *)y, (const float4 *)x, coef, (const float4 *)xs, scoef, workspace, rows, relu, bits); \
        else                                                                                                         \
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<CC, false>), dim3(groups), dim3(THREADS), 0, st, (const float4 *)dy, \
                               (const float4 *)y, (const float4 *)x, coef, nullptr, nullptr, workspace, rows, relu, bits); \
    }
    LAD_RED(16) LAD_RED(32) LAD_RED(64)
#undef LAD_RED
    }
    int rc = check_launch("bn_bwd_reduce_kernel");
    if (rc) return rc;
    if (pre_partials != nullptr && groups >= TWO_LEVEL_MIN_TILES && 2 * channels <= 128 && THREADS % (channels / 2) == 0) {
        // per-tile partials of a large layer: two levels, as lad_bn_finalize does (pre_partials is CONSUMED)
        const int64_t rps = ceil_div((int64_t)groups, FIN_SLICES);
        const int slices = (int)(groups / rps);
        const FinTail ft{1, (double)count, gamma, nullptr, coef, nullptr, nullptr, 0.0f, bcoef, dgamma, dbeta, launch_ticket()};
        LAD_REQUIRE(ft.ticket, "lad_bn_bwd: no ticket for the two-level sum");
        hipLaunchKernelGGL(bn_slice_sum_kernel, dim3(slices), dim3(THREADS), 0, st, pre_partials, (int64_t)groups, channels, rps, ft);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(channels), dim3(THREADS), 0, st, sums, groups, sh ? 3 : 2, channels,
                           (double)count, gamma, coef, sgamma, scoef, dgamma, dbeta, dsgamma, dsbeta, bcoef);
    }
    rc = check_launch("bn_bwd_finalize_kernel");
    if (rc) return rc;
    if (dx == nullptr) return LAD_OK;  // the consumer applies bcoef itself (lad_stem_wgrad_bn)
    const int c4s = log2_exact(channels / 4);
    const dim3 grid(row_grid(rg.n_img_rows, rg.Wp << c4s)), block(THREADS);
    if (mode == 0)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, grid, block, 0, st, (const float4 *)dy, (const float4 *)y, (const float4 *)x,
                           coef, bcoef, nullptr, nullptr, (float4 *)dx, nullptr, rg, channels, c4s, relu, bits);
    else if (mode == 1)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, grid, block, 0, st, (const float4 *)dy, (const float4 *)y, (const float4 *)x,
                           coef, bcoef, nullptr, nullptr, (float4 *)dx, (float4 *)aux, rg, channels, c4s, relu, bits);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<2>, grid, block, 0, st, (const float4 *)dy, (const float4 *)y, (const float4 *)x,
                           coef, bcoef, (const float4 *)xs, scoef, (float4 *)dx, (float4 *)aux, rg, channels, c4s, relu, bits);
    return check_launch("bn_bwd_apply_kernel");
}
}  // namespace

extern "C" int lad_bn_bwd(const float *dy, const float *y, const float *x, const float *coef, const float *gamma,
                          const float *xs, const float *scoef, const float *sgamma, float *dx, float *aux,
                          float *dgamma, float *dbeta, float *dsgamma, float *dsbeta, float *workspace, float *bcoef,
                          float *pre_partials, int64_t pre_tiles, int64_t batch, int32_t H, int32_t W, int32_t channels,
                          int32_t relu, int32_t mode, void *stream) {
    using namespace lad;
    LAD_REQUIRE(relu >= 0 && relu <= 2, "lad_bn_bwd: relu must be 0, 1 or 2");
    return bn_bwd_impl(dy, y, nullptr, x, coef, gamma, xs, scoef, sgamma, dx, aux, dgamma, dbeta, dsgamma, dsbeta, workspace, bcoef,
                       pre_partials, pre_tiles, batch, H, W, channels, relu, mode, stream);
}

// BatchNorm + residual ReLU backward with the ReLU decisions taken from the sign bits lad_bn_act_bits left: dx only (the
// masked gradient dy * [y > 0] that the identity shortcut carries is re-derived by its consumer from dy and the same bits,
// lad_conv_b3_fwd_f32_gated, instead of being written here).
extern "C" int lad_bn_bwd_bits(const float *dy, const uint64_t *y_bits, const float *x, const float *coef, const float *gamma,
                               float *dx, float *dgamma, float *dbeta, float *workspace, float *bcoef, float *pre_partials,
                               int64_t pre_tiles, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(y_bits && x, "lad_bn_bwd_bits: null buffer");   // (dx = NULL: sums and coefficients only -- lad_conv_wgrad_h2_bnbwd applies them)
    return bn_bwd_impl(dy, nullptr, (const unsigned long long *)y_bits, x, coef, gamma, nullptr, nullptr, nullptr, dx, nullptr, dgamma,
                       dbeta, nullptr, nullptr, workspace, bcoef, pre_partials, pre_tiles, batch, H, W, channels, 3, 0, stream);
}
