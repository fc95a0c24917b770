// Batch assembly straight from HBM-resident whole-channel feature matrices.
//
// Replaces the per-cut lilcom read + pad + stack of Lhotse's PrecomputedFeatures behind LadDataset.__getitem__
// (datasets.py:49-68, load_data.py:12-34) and the per-window slicing + zero right-pad of InferenceDataset
// (datasets.py:85-93).  Segment b = frames [first[b], first[b] + count[b]) of channel matrix `chan[b]`, right-padded
// to n_frames rows with `pad` (log-eps for training cuts, 0.0 for inference windows).  Pure copy: HBM-bound, one
// 16-byte (4-filter) element per lane, coalesced on both sides (F = 44 -> 11 float4 per frame).
#include "lad_common.h"

namespace {
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void gather_kernel(const float *const *__restrict__ chan_ptr,
                                                         const int64_t *__restrict__ chan_frames,
                                                         const int32_t *__restrict__ chan, const int64_t *__restrict__ first,
                                                         const int32_t *__restrict__ count, int64_t n_seg, int n_frames,
                                                         int F4, float pad, float4 *__restrict__ out) {
    const int per_seg = n_frames * F4;
    const int64_t total = n_seg * per_seg;
    for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * THREADS) {
        const int64_t b = idx / per_seg;
        const int r = (int)(idx - b * per_seg);
        const int t = r / F4, f4 = r - t * F4;
        const int c = chan[b];
        const int64_t src_t = first[b] + t;
        float4 v = make_float4(pad, pad, pad, pad);
        if (t < count[b] && src_t >= 0 && src_t < chan_frames[c])
            v = reinterpret_cast<const float4 *>(chan_ptr[c])[src_t * F4 + f4];
        out[idx] = v;
    }
}
}  // namespace

extern "C" int lad_gather_segments(const float *const *chan_ptr, const int64_t *chan_frames, const int32_t *chan,
                                   const int64_t *first, const int32_t *count, int64_t n_seg, int32_t n_frames, int32_t F,
                                   float pad, float *out, void *stream) {
    using namespace lad;
    LAD_REQUIRE(chan_ptr && chan_frames && chan && first && count && out, "lad_gather_segments: null buffer");
    LAD_REQUIRE(n_seg >= 0 && n_frames >= 1 && F >= 4 && F % 4 == 0, "lad_gather_segments: F must be a multiple of 4");
    if (n_seg == 0) return LAD_OK;
    const int64_t total = n_seg * n_frames * (F / 4);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, THREADS), 256 * 16);
    hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, chan_ptr, chan_frames, chan, first, count,
                       n_seg, n_frames, F / 4, pad, (float4 *)out);
    return check_launch("gather_kernel");
}
