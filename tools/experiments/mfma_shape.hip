// Calibration (round 3): which bf16 MFMA shape should the split-operand ("bf16 x 3") kernels use?
//
// MI355X_MICROARCH.md, "DVFS give-back" item 7: on random data the chip holds a higher clock under v_mfma_f32_16x16x32_bf16
// than under v_mfma_f32_32x32x16_bf16 at equal cycles per FLOP.  conv_b3 / wgrad_b3 run at ~1.6 GHz, i.e. in that regime.
// This file runs the MFMA loop of conv_b3 alone -- a wave owns 64 rows x 64 columns, operands re-read from LDS for every
// (tap, 16-channel group), the LDS images filled with three-way split random numbers, no staging, no barriers, no epilogue --
// in both shapes:
//   shape 0: 24 x 32x32x16 per step (2 x 2 accumulators x 6 plane products), A rows padded to 112 bytes
//   shape 1: 48 x 16x16x32 per step (4 x 4 accumulators x 3): two plane products share one MFMA through the K dimension,
//            k = 0..15 -> (plane x, channels 0..15), k = 16..31 -> (plane y, channels 0..15):
//            [a1|a2] x [b3;b2] = a1 b3 + a2 b2,  [a3|a1] x [b1;b2] = a3 b1 + a1 b2,  [a1|a2] x [b1;b1] = a1 b1 + a2 b1
//            (lanes 32..63 simply read another plane of the same LDS images); A rows unpadded (96 bytes: conflict-free
//            for this read pattern).
// Each workgroup also stamps s_memtime / s_memrealtime around its loop: the in-kernel clock (item 6 of the same section).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libmfma_shape.so tools/experiments/mfma_shape.hip
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int NROWS = 256 + 2 * 46;

__device__ __forceinline__ float rnd(unsigned &s) {
    s = s * 1664525u + 1013904223u;
    return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23));   // uniform [-1, 1)
}
__device__ __forceinline__ void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
    p1 = (__bf16)x;
    const float r1 = x - (float)p1;
    p2 = (__bf16)r1;
    p3 = (__bf16)(r1 - (float)p2);
}

template <int SHAPE, int ROWB>
__global__ __launch_bounds__(256, 3) void shape_loop(float *out, unsigned long long *stamps, int iters, int zero) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *a_s = smem;                  // [NROWS][ROWB]: [plane][16 bf16] (+ pad)
    unsigned char *b_s = smem + NROWS * ROWB;   // [plane][ntile 2][k half 2][n 32][8 bf16] = 6144
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int e = threadIdx.x; e < NROWS * 16; e += 256) {
        const int row = e >> 4, c = e & 15;
        __bf16 p1, p2, p3;
        split3(zero ? 0.f : rnd(s), p1, p2, p3);
        *reinterpret_cast<__bf16 *>(a_s + row * ROWB + 0 * 32 + c * 2) = p1;
        *reinterpret_cast<__bf16 *>(a_s + row * ROWB + 1 * 32 + c * 2) = p2;
        *reinterpret_cast<__bf16 *>(a_s + row * ROWB + 2 * 32 + c * 2) = p3;
    }
    for (int e = threadIdx.x; e < 1024; e += 256) {
        __bf16 p1, p2, p3;
        split3(zero ? 0.f : 0.05f * rnd(s), p1, p2, p3);
        *reinterpret_cast<__bf16 *>(b_s + 0 * 2048 + e * 2) = p1;
        *reinterpret_cast<__bf16 *>(b_s + 1 * 2048 + e * 2) = p2;
        *reinterpret_cast<__bf16 *>(b_s + 2 * 2048 + e * 2) = p3;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        const int i = lane & 31, gk = lane >> 5;
        const unsigned char *a_base = a_s + (wave * 32 + i + 46) * ROWB + gk * 16;
        const unsigned char *bp = b_s + (gk * 32 + i) * 16;
        f32x16 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_sched_barrier(0);   // (keep the taps apart: hoisting every tap's reads spills)
                const int off = (tap / 3 - 1) * 45 + (tap % 3 - 1);
                const unsigned char *ap = a_base + off * ROWB;
                bf16x8 a[2][3], b[3][2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[rb][p] = *reinterpret_cast<const bf16x8 *>(ap + rb * 128 * ROWB + p * 32);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int n = 0; n < 2; ++n) b[p][n] = *reinterpret_cast<const bf16x8 *>(bp + (p * 2 + n) * 1024);
#define TERM(pa, pb)                                  \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)  \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) \
            acc[rb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb][pa], b[pb][n], acc[rb][n], 0, 0, 0);
                TERM(0, 2) TERM(1, 1) TERM(2, 0) TERM(0, 1) TERM(1, 0) TERM(0, 0)
#undef TERM
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) total += acc[x][y][r];
    } else {
        const int m = lane & 15, kg = (lane >> 4) & 1, hi = lane >> 5;
        // A fragments: P = [a1|a2], Q = [a3|a1]; B fragments: U = [b3;b2], V = [b1;b1], W = [b1;b2]
        const unsigned char *a_row = a_s + (wave * 64 + m + 46) * ROWB + kg * 16;   // row tile r: + r * 16 rows
        const int aP = (hi ? 1 : 0) * 32, aQ = (hi ? 0 : 2) * 32;
        const unsigned char *b_col = b_s + kg * 512 + m * 16;                         // column tile c: + (c >> 1) * 1024 + (c & 1) * 256
        const int bU = (hi ? 1 : 2) * 2048, bV = 0, bW = (hi ? 1 : 0) * 2048;
        f32x4 acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_sched_barrier(0);
                const int off = (tap / 3 - 1) * 45 + (tap % 3 - 1);
                const unsigned char *ap = a_row + off * ROWB;
                bf16x8 bu[4], bv[4], bw[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned char *bc = b_col + (c >> 1) * 1024 + (c & 1) * 256;
                    bu[c] = *reinterpret_cast<const bf16x8 *>(bc + bU);
                    bv[c] = *reinterpret_cast<const bf16x8 *>(bc + bV);
                    bw[c] = *reinterpret_cast<const bf16x8 *>(bc + bW);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16x8 p = *reinterpret_cast<const bf16x8 *>(ap + r * 16 * ROWB + aP);
                    const bf16x8 q = *reinterpret_cast<const bf16x8 *>(ap + r * 16 * ROWB + aQ);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, bu[c], acc[r][c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, bw[c], acc[r][c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, bv[c], acc[r][c], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) total += acc[x][y].x + acc[x][y].y + acc[x][y].z + acc[x][y].w;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2 + 0] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
    if (total == 12345.678f) out[threadIdx.x] = total;
}
}  // namespace

// returns ms of the last of `reps` launches; stamps_host (2 x blocks u64) receives (shader cycles, 100 MHz ticks) per workgroup
extern "C" float mfma_shape_run(int shape, int blocks, int iters, int reps, int zero, unsigned long long *stamps_host) {
    float *out;
    unsigned long long *stamps;
    (void)hipMalloc(&out, 1024);
    (void)hipMalloc(&stamps, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t lds0 = NROWS * 112 + 6144, lds1 = NROWS * 96 + 6144;
    // pad the dynamic LDS so that exactly three workgroups fit a CU in both shapes (as conv_b3)
    const size_t lds = 52 * 1024;
    (void)lds0;
    (void)lds1;
    (void)hipFuncSetAttribute((const void *)shape_loop<0, 112>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute((const void *)shape_loop<1, 96>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (shape == 0) hipLaunchKernelGGL((shape_loop<0, 112>), dim3(blocks), dim3(256), lds, 0, out, stamps, iters, zero);
        else hipLaunchKernelGGL((shape_loop<1, 96>), dim3(blocks), dim3(256), lds, 0, out, stamps, iters, zero);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (stamps_host) (void)hipMemcpy(stamps_host, stamps, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    (void)hipFree(out);
    (void)hipFree(stamps);
    return ms;
}
