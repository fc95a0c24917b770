// Reproducer attempt for the round-3 finding (build.py: stem.hip is compiled with -fno-slp-vectorize): with a SECOND PROCESS on the
// same GPU, stem_wgrad_kernel<2> -- whose scalar fmaf chains hipcc packs into v_pk_fma_f32 with op_sel forms -- returned another
// value in ONE accumulator register of whole workgroups in 1-5 % of identical launches.  This program isolates that code shape
// (36 accumulators acc[4][9] += v[t] * d[c] per thread over many rows, the taps broadcast from LDS, the same launch bounds) and
// checks run-to-run bit equality; run two instances at once (tools/experiments/run_pk_fma_repro.sh).
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/pk_fma_repro.hip -o tools/experiments/pk_fma_repro          (SLP on: packed)
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize ... -o tools/experiments/pk_fma_repro_noslp                (plain v_fma_f32)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>

constexpr int THREADS = 256, COUT = 64, CQ = COUT / 4, RL = THREADS / CQ, TM = 128, TAPW = 12;

__global__ __launch_bounds__(THREADS) void chain_kernel(const float *__restrict__ feat, const float *__restrict__ dout, const float *__restrict__ w,
                                                        float *__restrict__ slabs, long rows, long n_tiles) {
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
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    constexpr int NR = TM / RL;
    for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long q0 = tile * TM;
        float4 dv[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const long q = q0 + rl + k * RL;
            dv[k] = q < rows ? *reinterpret_cast<const float4 *>(dout + q * COUT + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (tid < TM) {
            const long q = q0 + tid;
            float4 *dst = reinterpret_cast<float4 *>(tap_s + tid * TAPW);
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = q < rows ? feat[(q * 9 + t) % (rows * 3)] : 0.f;
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            dst[2] = make_float4(v[8], (q % 45 != 0) ? 1.f : 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int r = rl + k * RL;
            if (q0 + r >= rows) break;
            const float4 *src = reinterpret_cast<const float4 *>(tap_s + r * TAPW);
            const float4 a = src[0], b = src[1], c2 = src[2];
            const float v[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c2.x};
            if (c2.y != 0.0f) {
                float xa[4] = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xa[c] = fmaf(v[t], wr[c][t], xa[c]);
                const float dd[4] = {xa[0] > 0.1f ? dv[k].x : 0.f, xa[1] > 0.1f ? dv[k].y : 0.f, xa[2] > 0.1f ? dv[k].z : 0.f, xa[3] > 0.1f ? dv[k].w : 0.f};
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
        slabs[(long)blockIdx.x * (COUT * 9) + e] = s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 1500;
    const long rows = 64L * 101 * 45, n_tiles = (rows + TM - 1) / TM;
    const int groups = 1536 < n_tiles ? 1536 : (int)n_tiles;
    std::vector<float> hf(rows * 3), hd(rows * COUT), hw(COUT * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hf) v = rnd();
    for (auto &v : hd) v = rnd() * 1e-3f;
    for (auto &v : hw) v = rnd() * 0.3f;
    float *f, *d, *w, *slabs;
    CK(hipMalloc(&f, hf.size() * 4)); CK(hipMalloc(&d, hd.size() * 4)); CK(hipMalloc(&w, hw.size() * 4));
    CK(hipMalloc(&slabs, (size_t)groups * COUT * 9 * 4));
    CK(hipMemcpy(f, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    // K launches back to back per pass into K buffers, ONE synchronisation per pass: with two such processes the queues stay
    // full and the hardware time-slices them inside kernels (a launch + copy per pass lets them alternate at kernel boundaries,
    // where nothing went wrong in 2 x 1500 passes)
    const int K = 40;
    const size_t n = (size_t)groups * COUT * 9;
    CK(hipFree(slabs));
    CK(hipMalloc(&slabs, K * n * 4));
    std::vector<float> ref(n), got(K * n);
    int bad = 0;
    hipLaunchKernelGGL(chain_kernel, dim3(groups), dim3(THREADS), 0, 0, f, d, w, slabs, rows, n_tiles);
    CK(hipMemcpy(ref.data(), slabs, n * 4, hipMemcpyDeviceToHost));
    for (int p = 0; p < passes / K; ++p) {
        for (int k = 0; k < K; ++k)
            hipLaunchKernelGGL(chain_kernel, dim3(groups), dim3(THREADS), 0, 0, f, d, w, slabs + k * n, rows, n_tiles);
        CK(hipMemcpy(got.data(), slabs, K * n * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k) {
            if (memcmp(ref.data(), got.data() + k * n, n * 4) != 0) {
                int first = -1, cnt = 0;
                for (size_t i = 0; i < n; ++i)
                    if (memcmp(&ref[i], &got[k * n + i], 4) != 0) { if (first < 0) first = (int)i; ++cnt; }
                if (bad < 5) printf("launch %d differs: %d values, first at workgroup %d element %d (co %d, tap %d)\n", p * K + k, cnt, first / (COUT * 9), first % (COUT * 9), (first % (COUT * 9)) / 9, first % 9);
                ++bad;
            }
        }
    }
    printf("pid %d: %d of %d launches differ from the first\n", (int)getpid(), bad, passes / K * K);
    return bad ? 1 : 0;
}
