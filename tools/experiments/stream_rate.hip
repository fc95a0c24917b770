// Micro-benchmark: what rate does an element-wise pass over 596 MB tensors reach on gfx950, by loop shape?
// (copy and "axpy" = two inputs, one output; 16 bytes per lane per access)
//   hipcc -O3 --offload-arch=gfx950 -o stream_rate stream_rate.hip && ./stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: one float4 per thread per trip, grid-stride.  1: UNR trips' loads issued before their stores.  2: as 1 with
// nontemporal loads and stores.
template <int NIN, int UNR, bool NT>
__global__ __launch_bounds__(256) void k(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ o, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        f4 va[UNR], vb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            va[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
            if (NIN == 2) vb[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f4 r = va[u] * 1.5f + 0.25f;
            if (NIN == 2) r += vb[u];
            if (NT) __builtin_nontemporal_store(r, o + i + u * stride);
            else o[i + u * stride] = r;
        }
    }
    for (; i < n; i += stride) o[i] = a[i] * 1.5f + 0.25f + (NIN == 2 ? b[i] : f4{0, 0, 0, 0});
}

template <int NIN, int UNR, bool NT>
void run(const char *name, const f4 *a, const f4 *b, f4 *o, long n, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NIN, UNR, NT>), dim3(wgs), dim3(256), 0, 0, a, b, o, n);
    hipEventRecord(e0);
    const int it = 10;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k<NIN, UNR, NT>), dim3(wgs), dim3(256), 0, 0, a, b, o, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    printf("%-28s wgs %6d: %7.1f us  %5.2f TB/s\n", name, wgs, ms * 1e3, (NIN + 1) * n * 16.0 / ms / 1e9);
}

int main() {
    const long n = 512L * 101 * 45 * 64 / 4;   // float4 per tensor
    f4 *a, *b, *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&o, n * 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    for (int wgs : {2048, 4096, 8192, 16384, 65536}) {
        run<1, 1, false>("copy unr1", a, b, o, n, wgs);
        run<1, 2, false>("copy unr2", a, b, o, n, wgs);
        run<1, 4, false>("copy unr4", a, b, o, n, wgs);
        run<1, 4, true>("copy unr4 nontemporal", a, b, o, n, wgs);
        run<2, 1, false>("axpy unr1", a, b, o, n, wgs);
        run<2, 2, false>("axpy unr2", a, b, o, n, wgs);
        run<2, 4, false>("axpy unr4", a, b, o, n, wgs);
        run<2, 4, true>("axpy unr4 nontemporal", a, b, o, n, wgs);
    }
    return 0;
}
