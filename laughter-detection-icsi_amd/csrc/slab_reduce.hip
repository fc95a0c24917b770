// Sum of the per-workgroup partial slabs the weight-gradient kernels leave (wgrad_mfma.hip, conv_s2_bwd.hip):
//   dw[co][ci][tap] = sum_wg slab[wg][tap][ci][co],   dbias[co] = sum_wg bias_slab[wg][co]
// in fixed order with double accumulation (bitwise reproducible, no float atomics).
//
// A training step has 19 weight-gradient launches; summed one by one that is 19 more launches of 5-18 us, each a full
// dependency boundary on the stream (0.3 ms per step, plus the gaps).  With deferral on, the weight-gradient entry points
// queue a descriptor instead, and lad_wgrad_defer_flush sums every queued layer in ONE launch (blockIdx.y = layer; the
// descriptors travel by value as kernel arguments, so the launch is graph-capturable and needs no device-side table).
// Each deferred layer must have been given its OWN workspace: the slabs stay there until the flush.
#include "lad_common.h"

#include <vector>

namespace {
using namespace lad;

constexpr int RED_THREADS = 256;
constexpr int MAX_PACK = 24;
struct Pack {
    SlabReduce d[MAX_PACK];
};

__global__ __launch_bounds__(RED_THREADS) void slab_reduce_multi_kernel(Pack pack) {
    const SlabReduce d = pack.d[blockIdx.y];
    const int n = d.taps * d.cin * d.cout;
    const int o = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;
    if (blockIdx.x * 64 >= n + (d.dbias != nullptr ? d.cout : 0)) return;   // (workgroup-uniform) past this layer's outputs
    __shared__ double red[4][64];
    double s = 0.0;
    const bool is_w = idx < n, is_b = !is_w && d.dbias != nullptr && idx < n + d.cout;
    const float *src = is_w ? d.slabs + idx : (is_b ? d.bias_slabs + (idx - n) : nullptr);
    const int64_t stride = is_w ? n : d.cout;
    if (src != nullptr) {
        int w = part;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains: the loads of a step are in flight together
        for (; w + 12 < d.groups; w += 16) {
            s0 += (double)src[(int64_t)w * stride];
            s1 += (double)src[(int64_t)(w + 4) * stride];
            s2 += (double)src[(int64_t)(w + 8) * stride];
            s3 += (double)src[(int64_t)(w + 12) * stride];
        }
        for (; w < d.groups; w += 4) s0 += (double)src[(int64_t)w * stride];
        s = (s0 + s1) + (s2 + s3);
    }
    red[part][o] = s;
    __syncthreads();
    if (part == 0) {
        const double t = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        if (is_w) {
            const int co = idx % d.cout;
            const int q = idx / d.cout;
            const int ci = q % d.cin, tap = q / d.cin;
            d.dw[((int64_t)co * d.cin + ci) * d.taps + tap] = (float)t;
        } else if (is_b) {
            d.dbias[idx - n] = (float)t;
        }
    }
}

bool g_defer = false;
std::vector<SlabReduce> g_pending;   // one engine per process (one process per GPU): not thread-safe, documented in lad_hip.h

int launch(const SlabReduce *d, int count, hipStream_t st) {
    for (int base = 0; base < count; base += MAX_PACK) {
        Pack pack;
        const int m = std::min(MAX_PACK, count - base);
        int64_t max_n = 0;
        for (int k = 0; k < m; ++k) {
            pack.d[k] = d[base + k];
            max_n = std::max<int64_t>(max_n, (int64_t)d[base + k].taps * d[base + k].cin * d[base + k].cout + d[base + k].cout);
        }
        hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)ceil_div(max_n, 64), (unsigned)m), dim3(RED_THREADS), 0, st, pack);
        const int rc = check_launch("slab_reduce_multi_kernel");
        if (rc) return rc;
    }
    return LAD_OK;
}
}  // namespace

namespace lad {
int reduce_slabs(const SlabReduce &d, hipStream_t st) {
    if (g_defer) {
        for (const SlabReduce &p : g_pending)
            if (p.slabs == d.slabs)
                return fail(LAD_ERR_INVALID, "deferred weight-gradient sums: two layers share one workspace (%p)", (const void *)d.slabs);
        g_pending.push_back(d);
        return LAD_OK;
    }
    return launch(&d, 1, st);
}
}  // namespace lad

extern "C" int lad_wgrad_defer_begin(void) {
    g_pending.clear();
    g_defer = true;
    return LAD_OK;
}

extern "C" int lad_wgrad_defer_flush(void *stream) {
    g_defer = false;
    const int rc = g_pending.empty() ? LAD_OK : launch(g_pending.data(), (int)g_pending.size(), (hipStream_t)stream);
    g_pending.clear();
    return rc;
}
