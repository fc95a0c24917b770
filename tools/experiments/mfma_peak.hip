// Calibration: how close does a loop of nothing but v_mfma_f32_32x32x2_f32 come to the nominal fp32-matrix peak
// (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz) on this MI355X?  Diagnostic only, never in the product library.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libmfma_peak.so tools/experiments/mfma_peak.hip
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// the conv_s1 inner loop without its staging, barriers and epilogue: per 16 MFMAs on 4 accumulators, 2 A and 2 B
// ds_read_b128 (one group ahead), operands really consumed -- what LDS operand delivery alone costs
__global__ __launch_bounds__(256) void mfma_lds_loop(float *out, int iters, int look) {
    __shared__ __attribute__((aligned(16))) float a_s[384 * 36];
    __shared__ __attribute__((aligned(16))) float b_s[2 * 8 * 64 * 4];
    for (int i = threadIdx.x; i < 384 * 36; i += 256) a_s[i] = 1e-3f * (i & 15);
    for (int i = threadIdx.x; i < 2 * 8 * 64 * 4; i += 256) b_s[i] = 1e-3f * (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, gk = lane >> 5;
    const float *ap = a_s + (wave * 32 + i + 46) * 36 + 4 * gk;
    const float *bp = b_s + (gk * 64 + i) * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
    float4 af[2][2], bf[2][2];
    af[0][0] = *(const float4 *)(ap); af[0][1] = *(const float4 *)(ap + 128 * 36);
    bf[0][0] = *(const float4 *)(bp); bf[0][1] = *(const float4 *)(bp + 128);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            const int cur = c8 & 1, nxt = cur ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            const int off = ((it + c8) & 7) * 36 * look;  // a different row shift per group, like the taps
            af[nxt][0] = *(const float4 *)(ap + off + ((c8 + 1) & 3) * 8);
            af[nxt][1] = *(const float4 *)(ap + off + 128 * 36 + ((c8 + 1) & 3) * 8);
            bf[nxt][0] = *(const float4 *)(bp + (((c8 + 1) & 3) * 2 * 64) * 4);
            bf[nxt][1] = *(const float4 *)(bp + (((c8 + 1) & 3) * 2 * 64 + 32) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][x].x, bf[cur][y].x, acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][x].y, bf[cur][y].y, acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][x].z, bf[cur][y].z, acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][x].w, bf[cur][y].w, acc[x][y], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[x][y][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" float mfma_lds_run(int blocks, int iters, void *stream) {
    float *out;
    (void)hipMalloc(&out, 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipStream_t st = (hipStream_t)stream;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(mfma_lds_loop, dim3(blocks), dim3(256), 0, st, out, iters, 1);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}

extern "C" float mfma_peak_run(int nacc, int blocks, int iters, void *stream) {
    float *out;
    hipMalloc(&out, 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStream_t st = (hipStream_t)stream;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, st);
        if (nacc == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 2.0f);
        else if (nacc == 4) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 2.0f);
        else hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
