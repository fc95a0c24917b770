// Micro-benchmark: issue rate of packed vs plain f32 VALU instructions on gfx950, 1..8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 2048;

template <int MODE>
__global__ void k(float *out, float s) {
    v2f a[8];
    float b[16];
    for (int j = 0; j < 8; ++j) a[j] = v2f{(float)threadIdx.x + j, 1.0f};
    for (int j = 0; j < 16; ++j) b[j] = threadIdx.x + j;
    v2f m = {s, s};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(m));
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(b[j]) : "v"(s));
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(b[j]) : "v"(s));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int j = 0; j < 8; ++j) r += a[j].x + a[j].y;
    for (int j = 0; j < 16; ++j) r += b[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}

int main() {
    float *d;
    hipMalloc(&d, (1 << 23) + 64);
    const char *names[5] = {"v_pk_fma_f32 x8", "v_fma_f32 x16", "v_pk_add_f32 x8", "v_add_f32 x16", "v_pk_mul_f32 x8"};
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        for (int mode = 0; mode < 5; ++mode) {
            dim3 grid(256 * waves_per_simd), block(256);   // 4 waves per workgroup: one per SIMD
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, 1.0001f); break;
                    case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, 1.0001f); break;
                    case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, 1.0001f); break;
                    case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, 1.0001f); break;
                    default: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, 1.0001f); break;
                }
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long cyc;
            hipMemcpy(&cyc, (char *)d + (1 << 23), 8, hipMemcpyDeviceToHost);
            const int n_instr = (mode == 1 || mode == 3) ? 16 * ITER : 8 * ITER;
            printf("%d wave(s)/SIMD  %-18s %7.1f us   wave 0: %6.2f cycles per instruction  -> SIMD: %5.2f cycles per instruction, %5.1f flop-lanes/clk\n",
                   waves_per_simd, names[mode], ms * 1e3, (double)cyc / n_instr, (double)cyc / n_instr / waves_per_simd,
                   64.0 * ((mode == 1 || mode == 3) ? 1 : 2) / ((double)cyc / n_instr / waves_per_simd));
        }
    }
    return 0;
}
