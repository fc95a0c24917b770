// Do the matrix pipe and the vector ALU of a SIMD overlap when they are fed by DIFFERENT wavefronts?  (Round 4: conv_h2 and wgrad_h2
// behave as if a SIMD's MFMA cycles and its other instructions' issue cycles simply add up.)
// A workgroup = 8 waves = 2 per SIMD: waves 0-3 run a chain-free stream of v_mfma_f32_16x16x32_f16 (16 accumulators), waves 4-7 a
// stream of plain / packed FMAs or LDS reads.  Timed: each half alone (the other half exits at once), then both.
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/issue_overlap.hip -o tools/experiments/issue_overlap && tools/experiments/issue_overlap
#include <hip/hip_runtime.h>

#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // what waves 4-7 do: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = ds_read_b128
__global__ __launch_bounds__(512, 1) void k(float *out, int do_mfma, int do_other, int iters, unsigned long long *clk) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i * 1e-3f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (wave < 4) {
        if (!do_mfma) return;
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (_Float16)(0.01f * (lane + j));
            b[j] = (_Float16)(0.02f * (lane - j));
        }
        f32x4 acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
        for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (blockIdx.x == 7 && threadIdx.x == 0) {   // shader clock over this wave's life: s_memtime ticks per 100 MHz tick
            const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            clk[0] = c1 - c0;
            clk[1] = r1 - r0;
        }
    } else {
        if (!do_other) return;
        if (MODE == 0) {
            float v[16];
            for (int t = 0; t < 16; ++t) v[t] = 0.001f * (lane + t);
            const float m = 1.0001f, c = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)   // 64 v_fma_f32 per iteration (= the issue cycles of 16 MFMAs at 4 cycles each)
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = __builtin_fmaf(v[t], m, c);
            }
            float s = 0.f;
            for (int t = 0; t < 16; ++t) s += v[t];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else if (MODE == 1) {
            f32x2 v[16];
            for (int t = 0; t < 16; ++t) v[t] = f32x2{0.001f * (lane + t), 0.002f * t};
            const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 16; ++t) v[t] = __builtin_elementwise_fma(v[t], m, c);
            }
            float s = 0.f;
            for (int t = 0; t < 16; ++t) s += v[t].x + v[t].y;
            out[blockIdx.x * 512 + threadIdx.x] = s;
        } else {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 *p = reinterpret_cast<const f32x4 *>(lds) + lane;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int t = 0; t < 16; ++t) s4 += p[(t * 64 + it) & 1023];
            }
            out[blockIdx.x * 512 + threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
        }
    }
}

unsigned long long *g_clk;
template <int MODE>
float run(float *out, int a, int b, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 2), dim3(512), 0, 0, out, a, b, iters, g_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256 * 2), dim3(512), 0, 0, out, a, b, iters, g_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (a) {
        unsigned long long h[2];
        hipMemcpy(h, g_clk, 16, hipMemcpyDeviceToHost);
        printf("    [mfma %d other %d] shader clock of an MFMA wave's life: %.0f MHz (%llu cycles)\n", a, b, 100.0 * (double)h[0] / (double)h[1], h[0]);
    }
    return ms / 5;
}

int main() {
    float *out;
    hipMalloc(&out, 512 * 512 * 4);
    hipMalloc(&g_clk, 16);
    const int iters = 4000;   // 64,000 MFMAs per wave = 1.02 M matrix-pipe cycles
    const char *names[3] = {"v_fma_f32 x64/iter", "v_pk_fma_f32 x64/iter", "ds_read_b128 x16/iter"};
    for (int mode = 0; mode < 3; ++mode) {
        float m, o, both;
        if (mode == 0) { m = run<0>(out, 1, 0, iters); o = run<0>(out, 0, 1, iters); both = run<0>(out, 1, 1, iters); }
        else if (mode == 1) { m = run<1>(out, 1, 0, iters); o = run<1>(out, 0, 1, iters); both = run<1>(out, 1, 1, iters); }
        else { m = run<2>(out, 1, 0, iters); o = run<2>(out, 0, 1, iters); both = run<2>(out, 1, 1, iters); }
        printf("%-24s MFMA waves alone %.3f ms, other waves alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", names[mode], m, o, both, m + o,
               m > o ? m : o);
    }
    return 0;
}
