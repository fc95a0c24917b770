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
// ---- sliding-window inference: a window's level-1 activation from a shared stream + its own boundary strips ----------------
// Windows at a stride of one frame (segment_laughter.py:90-101 over InferenceDataset, datasets.py:72-93) overlap by 99 %; the
// stem and the stride-1 blocks at full resolution see that overlap unchanged: row y of window w of their output equals row
// w + y of the same layers run over the whole feature stream, EXCEPT within `band` rows of the window's top and bottom, where
// the window's own zero padding reaches (band = number of 3x3 convolutions on the way: 5 for ResNetBigger).  So those layers
// run ONCE over the stream (one tall image) and on STRIPS of 2 * band rows: strip s = frames [s, s + 2 band) as an image of its
// own, zero-padded above and below.  Its upper `band` rows see the padding above and not the one below (band layers cannot
// reach further): they are the top rows of the window that starts at frame s.  Its lower `band` rows see only the padding
// below: they are the bottom rows of the window that ENDS at frame s + 2 band, i.e. of window s - (H - 2 band).  One strip per
// frame offset therefore serves two windows (n_windows + H - 2 band strips for n_windows windows), and this kernel
// assembles each window's activation: rows [0, band) from strip w, [H - band, H) from strip w + H - 2 band, the rest from the
// stream.  Same kernels, same summation order per output: bit-identical to running them on every window in full, at a tenth of
// the arithmetic.  Layout on all sides: the shared-border PNHWC of lad_device.h, `row_bytes` bytes per position.
__global__ __launch_bounds__(THREADS) void assemble_windows_kernel(const uint4 *__restrict__ stream_act, const uint4 *__restrict__ strips,
                                                                   uint4 *__restrict__ out, int64_t n_win, int H, int Wp, int band,
                                                                   int row16, int64_t total) {
    const int Hp = H + 1, Hs = 2 * band + 1;             // padded heights of a window and of a strip
    const int64_t per_win = (int64_t)Hp * Wp * row16;    // 16-byte pieces per window
    for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * THREADS) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int64_t w = idx / per_win;
        if (w < n_win) {                                  // (past the last window: the tail rows, zero)
            const int r = (int)(idx - w * per_win);
            const int pos = r / row16, piece = r - pos * row16;
            const int yp = pos / Wp, xp = pos - yp * Wp;
            if (yp >= 1 && xp >= 1) {
                const int y = yp - 1;
                if (y < band) v = strips[((w * Hs + yp) * Wp + xp) * row16 + piece];
                else if (y >= H - band) v = strips[(((w + H - 2 * band) * Hs + (y - (H - 2 * band)) + 1) * Wp + xp) * row16 + piece];
                else v = stream_act[((w + y + 1) * (int64_t)Wp + xp) * row16 + piece];
            }
        }
        out[idx] = v;
    }
}
}  // namespace

extern "C" int lad_assemble_windows(const void *stream_act, const void *strips, void *out, int64_t n_windows, int32_t H, int32_t W,
                                    int32_t band, int32_t row_bytes, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stream_act && strips && out, "lad_assemble_windows: null buffer");
    LAD_REQUIRE(n_windows >= 1 && W >= 1 && band >= 1 && H > 2 * band && row_bytes >= 16 && row_bytes % 16 == 0,
                "lad_assemble_windows: bad geometry (H = %d, band = %d, row_bytes = %d)", H, band, row_bytes);
    const int Wp = W + 1, row16 = row_bytes / 16;
    const int64_t total = (n_windows * (H + 1) * Wp + Wp + 1) * row16;   // body + tail
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, THREADS), 256 * 64);
    hipLaunchKernelGGL(assemble_windows_kernel, dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, (const uint4 *)stream_act,
                       (const uint4 *)strips, (uint4 *)out, n_windows, H, Wp, band, row16, total);
    return check_launch("assemble_windows_kernel");
}

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
