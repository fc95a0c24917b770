// Gradient clipping + Adam on one flat parameter buffer.
//
// Replaces torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0) + optim.Adam(...).step() +
// model.zero_grad() of train.py:291-295 (84 small tensors -> 2 launches, no host synchronisation):
//   lad_grad_sumsq : per-block partial sums of g^2 (fixed order, reproducible)
//   (both take an optional DEVICE step counter: lad_grad_sumsq increments it, lad_adam_step derives the bias corrections
//    from it -- a captured hipGraph replays with constant host arguments, so the step number has to live on the device)
//   lad_adam_step  : every block re-reduces the partials (tiny), derives
//                    clip_coef = min(1, max_norm / (||g * grad_scale|| + 1e-6)) like torch, then applies the
//                    bias-corrected Adam update m,v,p in place and (optionally) zeroes the gradient.
// grad_scale folds the 1/world_size of the data-parallel mean into the same pass.
#include "lad_common.h"
#include "lad_device.h"

namespace {
using namespace lad;
constexpr int THREADS = 256;
constexpr int NORM_BLOCKS = 256;

__global__ __launch_bounds__(THREADS) void sumsq_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ partials,
                                                        int64_t *__restrict__ step_counter) {
    // device-side step count (hipGraph replays cannot take a new host argument per step): bumped once per optimiser step,
    // ahead of the Adam kernel that reads it (same stream)
    if (step_counter != nullptr && blockIdx.x == 0 && threadIdx.x == 0) step_counter[0] += 1;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const double v = (double)g[i];
        s += v * v;
    }
    __shared__ double red[THREADS / 64];
    s = wave_sum64d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < THREADS / 64; ++w) t += red[w];
        partials[blockIdx.x] = (float)t;
    }
}

__global__ __launch_bounds__(THREADS) void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                       float *__restrict__ v, int64_t n, const float *__restrict__ partials,
                                                       int n_partials, float grad_scale, float max_norm, float lr, float beta1,
                                                       float beta2, float eps, float bc1, float bc2_sqrt, int zero_grad,
                                                       float *__restrict__ norm_out, const int64_t *__restrict__ step_counter) {
    __shared__ float coef_s, bc_s[2];
    if (step_counter != nullptr && threadIdx.x == 64) {  // bias corrections from the device-side step count
        const double t = (double)step_counter[0];
        bc_s[0] = (float)(1.0 - pow((double)beta1, t));
        bc_s[1] = (float)sqrt(1.0 - pow((double)beta2, t));
    }
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 64) s += (double)partials[i];
        s = wave_sum64d(s);
        if (threadIdx.x == 0) {
            const float norm = (float)sqrt(s) * fabsf(grad_scale);
            float c = 1.0f;
            if (max_norm > 0.f) c = fminf(1.0f, max_norm / (norm + 1e-6f));
            coef_s = c * grad_scale;
            if (blockIdx.x == 0 && norm_out != nullptr) norm_out[0] = norm;
        }
    }
    __syncthreads();
    const float coef = coef_s;
    if (step_counter != nullptr) {
        bc1 = bc_s[0];
        bc2_sqrt = bc_s[1];
    }
    const float step = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        if (zero_grad) g[i] = 0.0f;
    }
}

// gradient accumulation over batches (train.py:287-289: loss / gradient_accumulation_steps, .grad accumulates)
__global__ __launch_bounds__(THREADS) void accumulate_kernel(float *__restrict__ acc, const float *__restrict__ g, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * THREADS) acc[i] = fmaf(scale, g[i], acc[i]);
}

}  // namespace

extern "C" int lad_grad_accumulate(float *acc, const float *grad, int64_t n, double scale, void *stream) {
    using namespace lad;
    LAD_REQUIRE(acc && grad && n >= 0, "lad_grad_accumulate: bad argument");
    if (n == 0) return LAD_OK;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, THREADS), 1024));
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, acc, grad, n, (float)scale);
    return check_launch("accumulate_kernel");
}

extern "C" int32_t lad_grad_sumsq_partials(void) { return NORM_BLOCKS; }

extern "C" int lad_grad_sumsq(const float *grad, int64_t n, float *partials, int64_t *step_counter, void *stream) {
    using namespace lad;
    LAD_REQUIRE(grad && partials && n >= 0, "lad_grad_sumsq: bad argument");
    hipLaunchKernelGGL(sumsq_kernel, dim3(NORM_BLOCKS), dim3(THREADS), 0, (hipStream_t)stream, grad, n, partials, step_counter);
    return check_launch("sumsq_kernel");
}

extern "C" int lad_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *sumsq_partials,
                             double grad_scale, double max_norm, double lr, double beta1, double beta2, double eps, int64_t step,
                             const int64_t *step_counter, int32_t zero_grad, float *norm_out, void *stream) {
    using namespace lad;
    LAD_REQUIRE(param && grad && exp_avg && exp_avg_sq && sumsq_partials, "lad_adam_step: null buffer");
    LAD_REQUIRE((step >= 1 || step_counter) && n >= 0, "lad_adam_step: step counts from 1");
    const float bc1 = step_counter ? 1.0f : (float)(1.0 - pow(beta1, (double)step));
    const float bc2_sqrt = step_counter ? 1.0f : (float)sqrt(1.0 - pow(beta2, (double)step));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, THREADS), 1024));
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       sumsq_partials, NORM_BLOCKS, (float)grad_scale, (float)max_norm, (float)lr, (float)beta1, (float)beta2, (float)eps,
                       bc1, bc2_sqrt, zero_grad, norm_out, step_counter);
    return check_launch("adam_kernel");
}
