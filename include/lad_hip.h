/*
 * lad_hip.h -- C ABI of liblad_hip.so: the MI355X (gfx950) hot path of laughter-detection-icsi.
 *
 * The reference (LasseWolter/laughter-detection-icsi) is pure Python and has no FFI layer; its seams
 * for this path are Python call signatures.  Each entry point below names the reference interface it
 * replaces (file:line in the reference tree).  The reference-side binding is a ctypes stub
 * (INTEGRATION.md); the host-side mirror of the reference modules lives in laughter-detection-icsi_amd/.
 *
 * Conventions
 *   - every function returns 0 on success, a negative lad_status otherwise; lad_last_error() gives the
 *     thread-local message.  Nothing throws across the ABI.
 *   - all data pointers are caller-owned DEVICE pointers to contiguous buffers unless a parameter is
 *     documented as host memory; the library never frees caller memory.
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous, no hidden device sync.
 *   - scratch memory is passed in by the caller (size from the matching *_workspace_bytes query).
 *   - plans own small immutable device tables (window, twiddles, filterbank) and may be shared by
 *     threads; launches on distinct streams may run concurrently.
 */
#ifndef LAD_HIP_H
#define LAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAD_VERSION 100 /* major*10000 + minor*100 + patch */

enum lad_status {
    LAD_OK = 0,
    LAD_ERR_INVALID = -1, /* bad argument / unsupported shape */
    LAD_ERR_HIP = -2,     /* a HIP runtime call failed */
    LAD_ERR_NOMEM = -3,
    /* not an error: a try-and-fall-back entry point (lad_f16_block_fwd) does not cover this geometry; nothing was launched,
     * lad_last_error() is untouched, the caller takes the general path */
    LAD_NOT_COVERED = 1
};

int lad_version(void);
const char *lad_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Feature extraction: framed PCM -> window -> rFFT-512 -> power -> mel filterbank -> log [-> DCT].
 * Replaces the object returned by get_feat_extractor (utils/utils.py:6-26; Lhotse
 * Fbank(FbankConfig(num_filters=44, frame_shift=0.01)), config.py:28-31) and its .extract() as
 * reached from load_data.py:47-49 and compute_features.py:84,105-109.
 * ---------------------------------------------------------------------------------------------- */
enum lad_pad_mode {
    LAD_PAD_KALDI_MIRROR = 0,   /* snip_edges=False: T=(N+hop/2)/hop, edge-inclusive mirror padding  */
    LAD_PAD_CENTER_REFLECT = 1, /* librosa center=True, pad_mode="reflect": T=1+N/hop                 */
    LAD_PAD_CENTER_ZERO = 2     /* librosa center=True, pad_mode="constant"                           */
};
enum lad_log_mode {
    LAD_LOG_LN = 0,   /* ln(max(mel, log_floor))        (Kaldi/Lhotse)                */
    LAD_LOG_DB = 1,   /* 10*log10(max(mel, log_floor))  (librosa power_to_db, no top_db) */
    LAD_LOG_NONE = 2  /* raw mel power                                                 */
};

typedef struct lad_fbank_cfg {
    int32_t n_fft;      /* must be 512                                                           */
    int32_t frame_len;  /* samples per frame that enter DC removal / pre-emphasis (<= n_fft)     */
    int32_t hop;        /* frame shift in samples                                                */
    int32_t n_mels;     /* 1..64                                                                 */
    int32_t n_mfcc;     /* 0 = output log-mel; 1..64 = output first n_mfcc DCT coefficients      */
    int32_t pad_mode;   /* lad_pad_mode                                                          */
    int32_t log_mode;   /* lad_log_mode                                                          */
    int32_t remove_dc;  /* subtract the frame mean before pre-emphasis                           */
    float preemph;      /* 0 disables                                                            */
    float log_floor;    /* 1.1920929e-07 for Lhotse, 1e-10 for librosa                           */
} lad_fbank_cfg;

/* window:  HOST float[n_fft]   (analysis window, zero beyond frame_len / centred as the convention needs)
 * melbank: HOST float[(n_fft/2+1) * n_mels], row-major (bin, filter)
 * dct:     HOST float[n_mels * n_mfcc] row-major (filter, coefficient), or NULL when n_mfcc == 0      */
int lad_fbank_plan_create(const lad_fbank_cfg *cfg, const float *window, const float *melbank,
                          const float *dct, void **plan_out);
int lad_fbank_plan_destroy(void *plan);
/* Two kernels implement the same arithmetic: the general one (any hop / frame length / DCT) and a fast one for
 * configurations with hop % 16 == 0, frame_len % 16 == 0, no DCT (the reference's: 400 / 160 / 44 filters), chosen
 * automatically.  which = 1 pins the general kernel (tests compare the two), 0 restores the automatic choice. */
int lad_fbank_plan_set_kernel(void *plan, int32_t which);
int lad_fbank_plan_has_fast_kernel(const void *plan);
/* frames produced for a clip of `samples_per_clip` samples (same formula the kernel uses) */
int64_t lad_fbank_num_frames(const void *plan, int64_t samples_per_clip);
/* pcm: float[n_clips][samples_per_clip] in [-1,1];  out: float[n_clips][T][n_out], n_out = n_mfcc ? n_mfcc : n_mels */
int lad_fbank_forward(void *plan, const float *pcm, int64_t n_clips, int64_t samples_per_clip,
                      float *out, void *stream);
/* one long channel (load_data.py:44-49: whole file as a single cut): out float[T][n_out] */
int lad_fbank_forward_long(void *plan, const float *pcm, int64_t n_samples, float *out, void *stream);

/* Batch assembly from HBM-resident whole-channel feature matrices: segment b = frames [first[b], first[b]+count[b])
 * of matrix chan[b], right-padded to n_frames rows with `pad`.  Replaces PrecomputedFeatures()(cuts) inside
 * LadDataset.__getitem__ (datasets.py:49-68; pad = log(eps) of the training cuts) and InferenceDataset.__getitem__
 * (datasets.py:85-93; pad = 0.0).  chan_ptr: DEVICE array of device pointers to float[chan_frames[c]][F];
 * chan_frames, chan, first, count: device arrays.  out: float[n_seg][n_frames][F]. */
int lad_gather_segments(const float *const *chan_ptr, const int64_t *chan_frames, const int32_t *chan,
                        const int64_t *first, const int32_t *count, int64_t n_seg, int32_t n_frames, int32_t F, float pad,
                        float *out, void *stream);

/* Sliding-window inference (segment_laughter.py:90-101): windows at a stride of one frame share 99 % of their input, and the
 * full-resolution layers (stem + the stride-1 blocks) see that overlap unchanged outside `band` rows of a window's top and
 * bottom (band = 3x3 convolutions on the way).  lad_assemble_windows builds the activation of n_windows windows of H rows from
 *   stream_act  the same layers run once over the stream: ONE image of n_windows + H - 1 rows (frame f of the chunk = row f),
 *   strips      the same layers run on n_windows + H - 2 band images of 2 * band rows: strip s = frames [s, s + 2 band) of the
 *               chunk, zero-padded above and below like a window.  Its upper band rows are the top rows of the window that
 *               STARTS at frame s; its lower band rows are the bottom rows of the window that ENDS at frame s + 2 band;
 * rows [0, band) of window w come from strip w, [H - band, H) from strip w + H - 2 band, the rest from the stream.  All three
 * in the shared-border PNHWC layout with row_bytes bytes per position (channels x element size, a multiple of 16).
 * Bit-identical to running the layers on every window in full (engine.predict_windows, tests/test_fullsize_gpu.py). */
int lad_assemble_windows(const void *stream_act, const void *strips, void *out, int64_t n_windows, int32_t H, int32_t W,
                         int32_t band, int32_t row_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * ResNetBigger forward / backward (models.py:82-115 ResidualBlock, :181-239 ResNetBigger; the autograd
 * backward of loss.backward() at train.py:289).  Activations are "PNHWC" with shared borders:
 * float[batch][H+1][W+1][C] followed by a tail of W+2 border rows, channels innermost; padded coordinates
 * (yp, xp) = (y+1, x+1); a "row" is one spatial position, lad_act_rows(batch, H, W) of them.  H, W always name the
 * UNPADDED image size.  INVARIANT: border positions (yp = 0, xp = 0, the tail) hold 0.0f in every tensor an entry
 * point reads; every entry point writes 0.0f there (or leaves them untouched where documented).
 * ---------------------------------------------------------------------------------------------- */

/* rows (spatial positions incl. borders and tail) of a (batch, H, W) activation tensor: allocate rows * channels elements */
int64_t lad_act_rows(int64_t batch, int32_t H, int32_t W);

/* Packed weight image consumed by the MFMA kernels.  w is the reference parameter (cout, cin, kh, kw)
 * with taps = kh*kw in {9, 1}.  mode 0: forward operand; mode 1: data-gradient operand (transposed,
 * spatially flipped).  nn.Conv2d.weight -> models.py:86-106,186-189. */
int64_t lad_conv_packed_weight_floats(int32_t cout, int32_t cin, int32_t taps, int32_t mode);
int lad_conv_pack_weights(const float *w, int32_t cout, int32_t cin, int32_t taps, int32_t mode, float *wt,
                          void *stream);
/* The same packing for many layers in one launch.  descs: DEVICE array of n_desc records
 * { const float *w; float *wt; int32_t cout, cin, taps, mode; } (32 bytes each). */
int lad_conv_pack_weights_multi(const void *descs, int32_t n_desc, void *stream);
/* number of 128-row tiles == rows of the (tile, 2, cout) BatchNorm partial-sum buffer a conv launch writes */
int64_t lad_conv_num_tiles(int64_t batch, int32_t H, int32_t W);
/* stride-1 convolution, 3x3 pad 1 (taps 9) or 1x1 (taps 1): out = conv(in, wt) + bias [+ addend], border rows
 * zeroed.  Used for nn.Conv2d forward (models.py:111-112) with a mode-0 image and for its data gradient with a
 * mode-1 image (then cin/cout are the GEMM K/N channel counts, i.e. swapped).  bias, addend, stat_partials
 * may be NULL.  stat_partials: float[num_tiles][2][cout] per-tile (sum, sum of squares) of the output. */
int lad_conv_fwd(const float *in, const float *wt, const float *bias, const float *addend, float *out,
                 float *stat_partials, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                 void *stream);
/* stride-2 convolution (3x3 pad 1 or 1x1 pad 0), input HxW -> output ceil(H/2) x ceil(W/2)
 * (models.py:86-89 with stride=2, :102-105 shortcut).  stat_partials sized by the OUTPUT geometry. */
int lad_conv_s2_fwd(const float *in, const float *wt, const float *bias, float *out, float *stat_partials,
                    int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, void *stream);
/* Eval mode: the BatchNorm that follows a convolution (running statistics) folded into its epilogue.
 * lad_bn_fold: scale = gamma / sqrt(running_var + 1e-5), shift = beta + (conv_bias - running_mean) * scale
 * (nn.BatchNorm2d in eval mode after nn.Conv2d, models.py:110-115,224; conv_bias may be NULL).
 * lad_conv_fwd_eval / lad_conv_s2_fwd_eval: out = [relu](conv(in, wt) * scale + shift [+ addend]), mode-0 image. */
int lad_bn_fold(const float *gamma, const float *beta, const float *running_mean, const float *running_var,
                const float *conv_bias, int32_t channels, float *scale, float *shift, void *stream);
int lad_conv_fwd_eval(const float *in, const float *wt, const float *scale, const float *shift, const float *addend,
                      float *out, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                      int32_t relu, void *stream);
int lad_conv_s2_fwd_eval(const float *in, const float *wt, const float *scale, const float *shift, float *out,
                         int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t relu,
                         void *stream);
/* zero-stuffing: up (HxW) <- src (ceil(H/2) x ceil(W/2)); turns a stride-2 conv's output gradient into the
 * operand of the stride-1 data-/weight-gradient kernels */
int lad_upsample2(const float *src, float *up, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* Backward of a stride-2 convolution (3x3 pad 1 or 1x1 pad 0; models.py:86-89 with stride 2, :102-105) at its true
 * cost, without zero-stuffing.  H, W: INPUT size; dout has the ceil(H/2) x ceil(W/2) geometry.
 * lad_conv_s2_dgrad: dx (input geometry, cin channels) from dout (cout channels) and the mode-1 packed image; every
 *   interior position of dx is written (accumulate = 0) or added to (accumulate = 1; required for the 1x1 shortcut,
 *   which only reaches the even/even positions); border positions are left as they are (zero by the layout invariant).
 * lad_conv_s2_wgrad: dw in the reference layout (cout, cin, kh, kw) (+ dbias, may be NULL). */
int lad_conv_s2_dgrad(const float *dout, const float *wt, float *dx, int64_t batch, int32_t H, int32_t W, int32_t cin,
                      int32_t cout, int32_t taps, int32_t accumulate, void *stream);
int64_t lad_conv_s2_wgrad_workspace_floats(int32_t cin, int32_t cout, int32_t taps);
int lad_conv_s2_wgrad(const float *in, const float *dout, float *workspace, float *dw, float *dbias, int64_t batch, int32_t H,
                      int32_t W, int32_t cin, int32_t cout, int32_t taps, void *stream);
/* The weight gradients of a stride-2 block's 3x3 convolution (dout) and of its 1x1 stride-2 shortcut (dout_sc, no bias)
 * in ONE launch: both convolutions read the same input and the shortcut's taps are the 3x3's centre-tap rows, so dw_sc
 * (cout, cin, 1, 1) rides along as a tenth tap.  Bit-identical to lad_conv_s2_wgrad(taps 9) + lad_conv_s2_wgrad(taps 1). */
int64_t lad_conv_s2_wgrad_fused_workspace_floats(int32_t cin, int32_t cout);
int lad_conv_s2_wgrad_fused(const float *in, const float *dout, const float *dout_sc, float *workspace, float *dw,
                            float *dbias, float *dw_sc, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout,
                            void *stream);
/* The same fusion in the other two directions.  lad_conv_s2_fwd_fused: the 3x3 stride-2 convolution (bias) and the 1x1
 * stride-2 shortcut (no bias) of one block in one launch, each with its own output and BatchNorm partials; the 3x3 part is
 * bit-identical to lad_conv_s2_fwd, the shortcut to lad_conv_s2_fwd(taps 1).  lad_conv_s2_dgrad_fused: the block's input
 * gradient dx = dgrad3x3(dout) + dgrad1x1(dout_sc), written once (the 1x1 lands on parity class (0,0) as one more tap);
 * equals lad_conv_s2_dgrad(taps 9, accumulate 0) + lad_conv_s2_dgrad(taps 1, accumulate 1) up to the order of one sum. */
int lad_conv_s2_fwd_fused(const float *in, const float *wt, const float *bias, const float *wt_sc, float *out,
                          float *stat_partials, float *out_sc, float *stat_partials_sc, int64_t batch, int32_t H, int32_t W,
                          int32_t cin, int32_t cout, void *stream);
int lad_conv_s2_dgrad_fused(const float *dout, const float *wt, const float *dout_sc, const float *wt_sc, float *dx,
                            int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, void *stream);
/* The 64 -> 32 stride-2 transition on the bf16 matrix cores with three-way split operands (csrc/conv_b3.hip, conv_s2b3:
 * the space-to-depth view of the input is formed while the rows are staged; fp32-equivalent arithmetic as lad_conv_b3*).
 * lad_conv_s2b3_pack_weights: w (32, 64, 3, 3) and w_sc (32, 64, 1, 1) -> one split image of
 * lad_conv_s2b3_packed_weight_bytes() bytes.  lad_conv_s2b3_fwd: same outputs and BatchNorm partials as
 * lad_conv_s2_fwd_fused(cin 64, cout 32). */
int64_t lad_conv_s2b3_packed_weight_bytes(void);
int lad_conv_s2b3_pack_weights(const float *w, const float *w_sc, void *wt, void *stream);
int lad_conv_s2b3_fwd(const float *in, const void *wt, const float *bias, float *out, float *stat_partials, float *out_sc,
                      float *stat_partials_sc, int64_t batch, int32_t H, int32_t W, void *stream);
/* Its data gradient dx = dgrad3x3(dout) + dgrad1x1(dout_sc) (= lad_conv_s2_dgrad_fused for 64 <- 32), parity class by parity
 * class with the result rows scattered to dx; border rows of dx are not written (zero by the layout invariant).  With
 * stat_partials != NULL also the first pass of the BatchNorm backward that consumes dx (as lad_conv_s2_dgrad_fused_bnstat:
 * input bn_x, sign bits bn_bits, coefficients bn_coef): float[lad_conv_s2b3_dgrad_partials(batch, H, W)][2][64]. */
int64_t lad_conv_s2b3_dgrad_packed_weight_bytes(void);
int lad_conv_s2b3_dgrad_pack_weights(const float *w, const float *w_sc, void *wt, void *stream);
/* lad_conv_s2b3_pack_weights + lad_conv_s2b3_dgrad_pack_weights in one launch (a training step needs both images). */
int lad_conv_s2b3_pack_weights_pair(const float *w, const float *w_sc, void *wt_fwd, void *wt_dgrad, void *stream);
int64_t lad_conv_s2b3_dgrad_partials(int64_t batch, int32_t H, int32_t W);
int lad_conv_s2b3_dgrad(const float *dout, const float *dout_sc, const void *wt, float *dx, float *stat_partials, const float *bn_x,
                        const uint64_t *bn_bits, const float *bn_coef, int64_t batch, int32_t H, int32_t W, void *stream);
/* ... and, for the 64 <- 32 transition, with the first pass of the BatchNorm backward that consumes dx (the bn2 of the
 * 64-channel block below: input bn_x, ReLU decisions from its sign bits bn_bits, coefficients bn_coef) in the epilogue:
 * stat_partials float[lad_conv_s2_dgrad_partials(batch, H, W)][2][64] -> lad_bn_bwd_bits pre_partials / pre_tiles. */
int64_t lad_conv_s2_dgrad_partials(int64_t batch, int32_t H, int32_t W);
int lad_conv_s2_dgrad_fused_bnstat(const float *dout, const float *wt, const float *dout_sc, const float *wt_sc, float *dx,
                                   float *stat_partials, const float *bn_x, const uint64_t *bn_bits, const float *bn_coef,
                                   int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, void *stream);
/* weight (+bias) gradient of a stride-1 conv: dw in the reference layout (cout, cin, kh, kw); dbias may be NULL.
 * GEOMETRY LIMIT: a tile stages 64 rows plus a halo of W+2 rows on either side in a fixed register/LDS budget, so for
 * cin = cout = 64 with 3x3 taps the image may be at most 46 columns wide (the model's widest map is 44, config.py:28-31);
 * wider images are refused with LAD_ERR_INVALID ("image too wide for the tile"), never computed wrongly
 * (tests/test_resnet_gpu.py::test_conv_s1_other_large_geometries). */
int64_t lad_conv_wgrad_workspace_floats(int32_t cin, int32_t cout, int32_t taps);
int lad_conv_wgrad(const float *in, const float *dout, float *workspace, float *dw, float *dbias, int64_t batch,
                   int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, void *stream);

/* 64 -> 64 3x3 stride-1 convolution (the four convolutions of block1, models.py:86-95, forward and data gradient) on the
 * bf16 matrix cores with fp32-equivalent arithmetic: every fp32 operand is the exact sum of three bf16 numbers and a
 * product keeps the six partial products above 2^-24 of it (csrc/conv_b3.hip).  Same f32 output tensor, bias / addend /
 * border / BatchNorm-partial semantics as lad_conv_fwd.  Since round 4 this is the FALLBACK generation of the split-operand
 * layers (engine.f16x2 = False / bench.py --no-h2; the product path is lad_conv_h2 below).
 *   lad_conv_b3_pack_weights: w (64, 64, 3, 3) fp32 -> split, MFMA-ordered image; mode 0 forward, 1 data gradient. */
int64_t lad_conv_b3_packed_weight_bytes(void);
int lad_conv_b3_pack_weights(const float *w, int32_t mode, void *wt, void *stream);
/* the convolution on the ordinary fp32 tensor (float[rows][64]): the three-way split happens while a stage of input
 * rows is staged into LDS, so producers and the other consumers of the tensor are untouched */
int lad_conv_b3_fwd_f32(const float *in, const void *wt, const float *bias, const float *addend, float *out,
                        float *partials, int64_t batch, int32_t H, int32_t W, void *stream);
/* out = conv(in) + bias + addend * [addend_bits]: lad_conv_b3_fwd_f32 whose addend is gated by the sign bits of an
 * activation (lad_bn_act_bits) -- the data gradient of a residual block's first convolution plus the identity shortcut's
 * share dy * [y > 0], taken from dy itself.  out may be the addend's own buffer (each 16 bytes are read, then written, by
 * the same thread). */
int lad_conv_b3_fwd_f32_gated(const float *in, const void *wt, const float *bias, const float *addend,
                              const uint64_t *addend_bits, float *out, float *partials, int64_t batch, int32_t H,
                              int32_t W, void *stream);
/* the data gradient (no bias; addend / addend_bits optional, as above) fused with the first pass of the BatchNorm
 * backward that consumes it: stat_partials float[lad_conv_num_tiles][2][64] = per 128-row tile (sum dz, sum dz*xhat) of
 * that BatchNorm (input bn_x, coefficients bn_coef float[6][64]; its ReLU decisions from the sign bits bn_bits or,
 * bn_bits = NULL, recomputed from bn_x as lad_bn_bwd relu = 2 does) -> lad_bn_bwd / lad_bn_bwd_bits pre_partials. */
int lad_conv_b3_dgrad_bnstat(const float *in, const void *wt, const float *addend, const uint64_t *addend_bits,
                             float *out, float *stat_partials, const float *bn_x, const uint64_t *bn_bits,
                             const float *bn_coef, int64_t batch, int32_t H, int32_t W, void *stream);
/* The split-operand convolution for `channels` = 64 or 32 (32: block2's three stride-1 3x3 convolutions, forward and data
 * gradient): packed image size / packing / forward (fp32 input, optional bias and addend, BatchNorm partials) / data
 * gradient with the consuming BatchNorm's sums (ReLU decisions recomputed from bn_x), as the 64-channel entry points
 * above.  Sign bits exist for 64 channels only. */
int64_t lad_conv_b3c_packed_weight_bytes(int32_t channels);
int lad_conv_b3c_pack_weights(const float *w, int32_t mode, void *wt, int32_t channels, void *stream);
int lad_conv_b3c_fwd_f32(const float *in, const void *wt, const float *bias, const float *addend, float *out,
                         float *partials, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
int lad_conv_b3c_dgrad_bnstat(const float *in, const void *wt, const float *addend, float *out, float *stat_partials,
                              const float *bn_x, const float *bn_coef, int64_t batch, int32_t H, int32_t W,
                              int32_t channels, void *stream);
int lad_conv_b3c_fwd_f32_bnrelu(const float *in, const float *in_coef, const void *wt, const float *bias, float *out,
                                float *partials, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* ---- "f16 x 2": the same 64 -> 64 / 32 -> 32 3x3 stride-1 convolutions (models.py:86-96,110-115) on TWO f16 planes per
 * operand, three plane products per fp32-equivalent product (csrc/conv_h2.hip: block floating point per staged tile, the
 * power-of-two scales live inside the kernels and the packed image).  Round 4; replaces lad_conv_b3c_* on the training path.
 * lad_conv_h2_pack_weights_multi: `table` = device array of n records {const float *w; void *wt; int32_t mode; int32_t channels}
 * (mode 0 forward, 1 data gradient; wt holds lad_conv_h2_packed_weight_bytes(channels) bytes; the record's `channels` is read only
 * when the call's `channels` is 0: then the table may mix 64- and 32-channel layers -- one launch per step instead of two).
 * lad_conv_h2: out = conv3x3(act(in)) + bias + addend * [addend_bits];  in_coef != NULL: act = relu(BatchNorm(in)) formed
 * while staging (lad_conv_b3c_fwd_f32_bnrelu);  bn_x != NULL: `partials` receives the sums of the BatchNorm backward that
 * consumes out (lad_conv_b3_dgrad_bnstat), else (sum, sum of squares) of out per 128-row tile, or nothing when NULL.
 * Tile: 256 rows at two workgroups per CU, or 128 rows at three for launches of fewer than two dispatch rounds (chosen by the
 * launcher; the other structures that were measured are built from tools/experiments/retired/ by tools/exp_h2.sh). */
int64_t lad_conv_h2_packed_weight_bytes(int32_t channels);
int lad_conv_h2_pack_weights_multi(const void *table, int32_t n, int32_t channels, void *stream);
/* weight (+ bias) gradient of the 64-channel convolutions on the same arithmetic; arguments as lad_conv_wgrad_b3c. */
int lad_conv_wgrad_h2(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw, float *dbias,
                      int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* The same weight gradient with the BatchNorm backward of this convolution's output on its gradient side: dy = the gradient
 * arriving at the BatchNorm(+ReLU) that follows the convolution, bn_x = that BatchNorm's input (the convolution's output),
 * bn_coef / bcoef = its forward / backward coefficients (lad_bn_finalize; lad_bn_bwd or lad_bn_bwd_bits called with dx = NULL),
 * bn_bits = sign bits of the block output (lad_bn_act_bits) or NULL (ReLU decisions recomputed from bn_x).  Writes
 * dc = the BatchNorm's input gradient -- what lad_bn_bwd would have written to dx, bit for bit -- for the data-gradient launch,
 * and accumulates dw / dbias from it.  Replaces the reference's autograd nodes NativeBatchNormBackward + ConvolutionBackward
 * (weight) of models.py:86-96 in loss.backward() (train.py:289). */
int lad_conv_wgrad_h2_bnbwd(const float *in, const float *in_coef, const float *dy, const float *bn_x, const uint64_t *bn_bits,
                            const float *bn_coef, const float *bcoef, float *dc, float *workspace, float *dw, float *dbias,
                            int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
int lad_conv_h2(const float *in, const float *in_coef, const void *wt, const float *bias, const float *addend,
                const uint64_t *addend_bits, float *out, float *partials, const float *bn_x, const uint64_t *bn_bits,
                const float *bn_coef, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* weight (+ bias) gradient of the same convolutions for 64 or 32 channels; in_coef = NULL: `in` is the stored activation,
 * otherwise relu(BatchNorm(in)) is formed while staging (lad_conv_wgrad_b3_bnrelu).  32 channels: W <= 30. */
int64_t lad_conv_wgrad_b3c_workspace_floats(int32_t channels);
int lad_conv_wgrad_b3c(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw, float *dbias,
                       int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* Forward convolution / weight gradient whose input is relu(BatchNorm(in)), in_coef = that BatchNorm's float[6][64] from
 * lad_bn_finalize: the second convolution of a residual block (models.py:110-112) reading the FIRST one's raw output;
 * scale, shift, ReLU and the zero border are applied while the rows are staged into LDS, with the fmaf / max of lad_bn_act,
 * so the results are bit-identical to lad_bn_act + lad_conv_b3_fwd_f32 / lad_conv_wgrad_b3 and the activation between
 * the two convolutions is never written or read (596 MB each way at batch 512). */
int lad_conv_b3_fwd_f32_bnrelu(const float *in, const float *in_coef, const void *wt, const float *bias, float *out,
                               float *partials, int64_t batch, int32_t H, int32_t W, void *stream);
int lad_conv_wgrad_b3_bnrelu(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw,
                             float *dbias, int64_t batch, int32_t H, int32_t W, void *stream);
/* Deferred sums.  Every weight-gradient entry point (lad_conv_wgrad, lad_conv_wgrad_b3[_bnrelu], lad_conv_s2_wgrad) ends
 * with a small launch that sums its per-workgroup partial slabs (in `workspace`) into dw / dbias.  Between
 * lad_wgrad_defer_begin() and lad_wgrad_defer_flush(stream) those launches are queued instead, and the flush sums all
 * queued layers in ONE launch on `stream` (which must be ordered after the weight-gradient launches): 19 dependent
 * launches per training step become one.  Each deferred call must have been given its OWN workspace (checked) -- the
 * slabs stay there until the flush; dw / dbias are undefined until then.  Process-wide state, not thread-safe: one
 * engine per process, as the data-parallel design has it. */
int lad_wgrad_defer_begin(void);
int lad_wgrad_defer_flush(void *stream);
/* weight (+bias) gradient of the same 64 -> 64 3x3 convolution with the same split arithmetic (csrc/wgrad_mfma.hip:
 * K = rows, so both operands come out of LDS through transposing reads); arguments and workspace
 * (lad_conv_wgrad_workspace_floats(64, 64, 9)) as lad_conv_wgrad; images up to 46 columns wide */
int lad_conv_wgrad_b3(const float *in, const float *dout, float *workspace, float *dw, float *dbias, int64_t batch,
                      int32_t H, int32_t W, void *stream);

/* stem conv3x3 1->64, no bias (models.py:186-189,224).  feat: float[batch][H][W] (the (B,1,100,44) input). */
int lad_stem_fwd(const float *feat, const float *weight, float *out, float *stat_partials, int64_t batch, int32_t H,
                 int32_t W, int32_t cout, void *stream);
/* eval-mode stem with bn1 + ReLU folded in, reading image b as frames [b*frame_stride, b*frame_stride + H) of a
 * (frames, W) feature matrix; frames >= frames_avail read as 0.  frame_stride = 1: the stride-one-frame windows of
 * InferenceDataset (datasets.py:72-93) taken straight from the whole-file features (load_data.py:49-53);
 * frame_stride = H: ordinary (batch, H, W) input. */
int lad_stem_fwd_eval(const float *feat, const float *weight, const float *scale, const float *shift, float *out,
                      int64_t batch, int32_t H, int32_t W, int32_t cout, int64_t frame_stride, int64_t frames_avail,
                      void *stream);
/* Round 6: the stem's batch statistics and its backward from MOMENTS of the input.  The stem convolution has one input channel, so
 * sum x_c and sum x_c^2 over the batch are combinations of F1[t] = sum f_t and F2[t][u] = sum f_t f_u of the nine taps (54 numbers).
 * lad_stem_bn_stats = lad_stem_fwd(out = NULL) + lad_bn_finalize from ONE pass over the features (no 64-channel pass): coef as
 * lad_bn_finalize writes it (the sums are those of the unrounded convolution: the coefficients agree to ~1e-7 relative), running
 * statistics updated; `moments` (lad_stem_moments_doubles() doubles) keeps the totals for the backward pass, `workspace` holds
 * lad_stem_moments_workspace_doubles() doubles.
 * lad_stem_bwd_onepass = lad_stem_bn_bwd_sums + lad_bn_bwd(dx = NULL, relu = 2) + lad_stem_wgrad_bn(x = NULL) with ONE pass over dy
 * instead of two: dw (64, 1, 3, 3), dgamma, dbeta of models.py:186-190,224; workspace: lad_stem_bwd_onepass_workspace_floats(). */
int64_t lad_stem_moments_workspace_doubles(void);
int64_t lad_stem_moments_doubles(void);
int lad_stem_bn_stats(const float *feat, const float *weight, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float momentum, float *coef, double *moments, double *workspace, int64_t batch, int32_t H,
                      int32_t W, int32_t cout, void *stream);
int64_t lad_stem_bwd_onepass_workspace_floats(void);
int lad_stem_bwd_onepass(const float *feat, const float *weight, const float *dy, const float *coef, const float *gamma,
                         const double *moments, float *workspace, float *dw, float *dgamma, float *dbeta, int64_t batch, int32_t H,
                         int32_t W, int32_t cout, void *stream);
int64_t lad_stem_wgrad_workspace_floats(void);
int lad_stem_wgrad(const float *feat, const float *dout, float *workspace, float *dw, int64_t batch, int32_t H,
                   int32_t W, int32_t cout, void *stream);
/* The same with the stem BatchNorm's backward applied on the fly: dy is the gradient wrt the BatchNorm(+ReLU) output, coef
 * the BatchNorm's float[6][C] forward coefficients and bcoef the float[8][C] backward coefficients left by
 * lad_bn_bwd(..., dx = NULL, ..., relu = 2, mode = 0).  x is the stem convolution's output (the BatchNorm input) if the
 * caller kept it, or NULL: then it is recomputed from feat and weight (the forward's own fmaf chain, bit-identical).
 * The stem needs no data gradient (models.py:224 is the first layer), so the BatchNorm's input gradient is never
 * materialised -- and with x = NULL neither is the convolution output: lad_stem_fwd(out = NULL) gives the batch statistics,
 * lad_stem_fwd_eval(scale = coef, shift = coef + C) the activated output, lad_stem_bn_bwd_sums the backward sums. */
int lad_stem_wgrad_bn(const float *feat, const float *dy, const float *x, const float *weight, const float *coef,
                      const float *bcoef, float *workspace, float *dw, int64_t batch, int32_t H, int32_t W, int32_t cout,
                      void *stream);
/* per workgroup (sum dz, sum dz*xhat) per channel of the stem BatchNorm's backward, x recomputed from feat and weight:
 * partials float[lad_stem_bn_bwd_groups(batch,H,W)][2][C], to be handed to lad_bn_bwd as pre_partials / pre_tiles */
int64_t lad_stem_bn_bwd_groups(int64_t batch, int32_t H, int32_t W);
int lad_stem_bn_bwd_sums(const float *feat, const float *weight, const float *dy, const float *coef, float *partials,
                         int64_t batch, int32_t H, int32_t W, int32_t cout, void *stream);

/* BatchNorm2d (+ residual + ReLU), train and eval (models.py:90,98,106,190; eps 1e-5, momentum 0.1).
 * coef: float[6][C] = scale, shift, mean, invstd, mean_lo, invstd_lo (hi + lo = the double-precision value; the
 * backward pass needs the extra bits, csrc/bn.hip).  count = batch*H*W (positions per channel).
 * stat_partials (float[n_tiles][2][C] from the convolution / stem epilogues) is CONSUMED: large layers reduce it in
 * place in two levels. */
int lad_bn_finalize(float *stat_partials, int64_t n_tiles, int32_t channels, int64_t count, const float *gamma,
                    const float *beta, float *running_mean, float *running_var, float momentum, float *coef,
                    void *stream);
/* Round 6: lad_bn_finalize for the TWO BatchNorm layers behind a stride-2 block's fused conv1 + shortcut launch (bn1 and the
 * shortcut's BatchNorm, models.py:98-106: sums of the same shape) in one launch; the same results as two calls. */
int lad_bn_finalize_pair(float *stat_partials_a, float *stat_partials_b, int64_t n_tiles, int32_t channels, int64_t count,
                         const float *gamma_a, const float *beta_a, float *running_mean_a, float *running_var_a, float *coef_a,
                         const float *gamma_b, const float *beta_b, float *running_mean_b, float *running_var_b, float *coef_b,
                         float momentum, void *stream);
/* y = act(x*scale + shift [+ res | + res*rscale + rshift]) on the interior of a (batch, H, W, channels) PNHWC tensor;
 * border positions of y are written as zero (the layout invariant the MFMA kernels rely on) */
int lad_bn_act(const float *x, const float *coef, const float *res, const float *res_coef, float *y, int64_t batch,
               int32_t H, int32_t W, int32_t channels, int32_t relu, void *stream);
/* backward of the above; relu 0: none, 1: mask = (y > 0), 2 (mode 0 only): mask recomputed as (x*scale+shift > 0), y
 * not read; mode 0: dx; 1: dx and aux = dz (identity shortcut); 2: dx and aux = gradient into the
 * shortcut BatchNorm's input.  bcoef: float[8][C] scratch, workspace: lad_bn_bwd_workspace_floats(C) floats.
 * dx = NULL (mode 0 only): only dgamma, dbeta and bcoef are produced -- for a consumer that applies bcoef itself
 * (lad_stem_wgrad_bn); with pre_partials given as well, x may be NULL (it is not read).
 * pre_partials (float[pre_tiles][2][C], from a producer that reduced while it wrote dy) is CONSUMED: 8192 tiles or more
 * are summed in place in two levels, as lad_bn_finalize does. */
int64_t lad_bn_bwd_workspace_floats(int32_t channels);
int lad_bn_bwd(const float *dy, const float *y, const float *x, const float *coef, const float *gamma,
               const float *xs, const float *scoef, const float *sgamma, float *dx, float *aux, float *dgamma,
               float *dbeta, float *dsgamma, float *dsbeta, float *workspace, float *bcoef, float *pre_partials,
               int64_t pre_tiles, int64_t batch, int32_t H, int32_t W, int32_t channels, int32_t relu, int32_t mode,
               void *stream);
/* Sign-bit flavour for the 64-channel residual blocks (replaces the `out = F.relu(out + shortcut)` mask that autograd
 * keeps for models.py:113-114).  lad_bn_act_bits = lad_bn_act(relu = 1) that also leaves y_bits: one uint64 per row of
 * y (lad_act_rows(batch, H, W) words; bit k*16 + j <-> channel 4*j + k is set iff y > 0).  lad_bn_bwd_bits = lad_bn_bwd
 * (relu = 1, mode 0) that takes the ReLU decisions from y_bits instead of reading y, and writes dx only: the masked
 * gradient dy * [y > 0] the identity shortcut carries is formed by its consumer, lad_conv_b3_fwd_f32_gated, from dy and
 * the same bits.  Per residual block the backward pass moves three activation-sized tensors less. */
int lad_bn_act_bits(const float *x, const float *coef, const float *res, const float *res_coef, float *y,
                    uint64_t *y_bits, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
int lad_bn_bwd_bits(const float *dy, const uint64_t *y_bits, const float *x, const float *coef, const float *gamma,
                    float *dx, float *dgamma, float *dbeta, float *workspace, float *bcoef, float *pre_partials,
                    int64_t pre_tiles, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
/* Data gradient of a stride-1 3x3 convolution (mode-1 image; cin/cout = GEMM K/N channels) fused with the FIRST pass of
 * the BatchNorm backward that consumes it: stat_partials receives, per 128-row tile, (sum dz, sum dz*xhat) of that
 * BatchNorm (input bn_x, output bn_y or NULL = mask recomputed from bn_x, coefficients bn_coef float[6][C]); pass it to
 * lad_bn_bwd as pre_partials with pre_tiles = lad_conv_num_tiles(...) (modes 0 and 1) and the reduce pass is skipped. */
int lad_conv_fwd_bnstat(const float *in, const float *wt, const float *addend, float *out, float *stat_partials,
                        const float *bn_x, const float *bn_y, const float *bn_coef, int64_t batch, int32_t H, int32_t W,
                        int32_t cin, int32_t cout, int32_t taps, void *stream);

/* Head: AvgPool2d(4) -> flatten -> bn2 -> dropout -> linear1 -> bn3 -> dropout -> ReLU -> linear2 -> sigmoid
 * (models.py:229-238) fused with nn.BCELoss and the _calc_metrics counters (train.py:203-224,279-285).
 * params: HOST array of 12 device pointers (bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
 * linear1.weight, linear1.bias, bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var, linear2.weight,
 * linear2.bias).  drop1/drop2: masks already scaled by 1/(1-p), or NULL.  stats: float[2F+64] saved for
 * backward; metrics: float[8] = mean BCE, #correct, #pred positive, #true positive, #target positive, B. */
int lad_pool_fwd(const float *x, float *pooled, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
int lad_pool_bwd(const float *dpooled, float *dx, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);
int64_t lad_head_workspace_floats(int64_t batch, int32_t F);
int lad_head_fwd_train(const float *const *params, const float *pooled, int64_t batch, int32_t F, const float *drop1,
                       const float *drop2, const int32_t *labels, float momentum, float *h, float *stats,
                       float *probs, float *metrics, void *stream);
/* The same with the dropout masks drawn INSIDE the launch (round 6): drop1 / drop2 are output buffers here (what lad_head_bwd then takes),
 * values 0 or 1 / keep from Philox4x32-10 keyed by `seed` at draw number *rng_counter (device int64, incremented by the kernel: graph
 * replays draw fresh masks); keep >= 1 = no dropout.  nbt: n_nbt device int64 counters incremented by one (the BatchNorm layers'
 * num_batches_tracked, which a train-mode forward advances), or NULL.  Replaces nn.Dropout's mask generation of models.py:232,235
 * (four torch launches per step) and the counter increment (one more). */
int lad_head_fwd_train_rng(const float *const *params, const float *pooled, int64_t batch, int32_t F, float *drop1, float *drop2, float keep,
                           uint64_t seed, int64_t *rng_counter, int64_t *nbt, int32_t n_nbt, const int32_t *labels, float momentum,
                           float *h, float *stats, float *probs, float *metrics, void *stream);
int lad_head_fwd_eval(const float *const *params, const float *pooled, int64_t batch, int32_t F, float *probs,
                      void *stream);
/* nn.BCELoss (mean, log clamped at -100) + the _calc_metrics counters for eval-mode probabilities (train.py:226-259):
 * metrics float[8] = mean BCE, #correct, #pred positive, #true positive, #target positive, n */
int lad_bce_metrics(const float *probs, const int32_t *labels, int64_t n, float *metrics, void *stream);
/* grads: HOST array of 8 device pointers (d bn2.weight, d bn2.bias, d linear1.weight, d linear1.bias, d bn3.weight,
 * d bn3.bias, d linear2.weight, d linear2.bias).  dprobs NULL = loss is the mean BCE against labels. */
int lad_head_bwd(const float *const *params, float *const *grads, const float *pooled, const float *h,
                 const float *stats, const float *probs, const float *dprobs, int64_t batch, int32_t F,
                 const float *drop1, const float *drop2, const int32_t *labels, float *workspace, float *dpooled,
                 void *stream);

/* ------------------------------------------------------------------------------------------------
 * fp16 inference path (eval mode; BASELINE configs[4]): half activations (PNHWC, zero border ring) and half weights,
 * f32 accumulation on the 16-bit matrix cores, BatchNorm folded in (scale/shift from lad_bn_fold).  Same layers as
 * lad_stem_fwd_eval / lad_conv_fwd_eval / lad_conv_s2_fwd_eval / lad_pool_fwd; `void*` activations are _Float16.
 * ---------------------------------------------------------------------------------------------- */
int64_t lad_f16_packed_weight_halfs(int32_t cout, int32_t cin, int32_t taps);
int lad_f16_pack_weights(const float *w, int32_t cout, int32_t cin, int32_t taps, void *wt, void *stream);
int lad_f16_stem_fwd(const float *feat, const float *weight, const float *scale, const float *shift, void *out,
                     int64_t batch, int32_t H, int32_t W, int32_t cout, int64_t frame_stride, int64_t frames_avail,
                     void *stream);
/* (16 channels with an addend, relu and >= 512 images small enough for a CU's LDS: the call runs block_f16_small_kernel's
 * one-convolution form -- several images per workgroup, weights resident -- instead of the tiled kernel; identical results.) */
int lad_f16_conv_fwd(const void *in, const void *wt, const float *scale, const float *shift, const void *addend, void *out,
                     int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t relu,
                     void *stream);
int lad_f16_conv_s2_fwd(const void *in, const void *wt, const float *scale, const float *shift, void *out, int64_t batch,
                        int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t relu, void *stream);
/* One residual block with the identity shortcut, 16 / 32 / 64 channels, in ONE launch with each image resident in a CU's LDS (round 5):
 *   y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)     -- models.py:110-115 (ResidualBlock.forward) in eval mode, BatchNorms folded.
 * wt1 / wt2: lad_f16_pack_weights images of the two 3x3 convolutions; x, y: shared-border half tensors of `batch` H x W images, y != x.
 * channels 64: covers (H + 1)(W + 1) <= 512 positions with (H + 1)(W + 1) + W <= 562 (LDS) and batch >= 256 (the boundary strips of the
 * sliding-window path: segment_laughter.py:90-101 through engine._forward_eval_stream).  channels 16 / 32: several images per workgroup,
 * both weight images resident in LDS; covers batch >= 512 images small enough that two of their tensors fit 160 KB next to the weights
 * (the windows at resolution levels 3 and 4, the strips of level 2).  Any other geometry returns LAD_NOT_COVERED with nothing launched
 * (no error string) and the caller runs the two convolutions by lad_f16_conv_fwd.  Results are bit-identical to that pair of calls.
 * Trailing rows: like every shared-border tensor, x ends in W + 2 zero rows behind its last image (the rows below the last image's
 * last row: the block reads them as that image's lower border) -- the caller provides them.  The 64-channel form does NOT write
 * y's trailing rows (lad_f16_conv_fwd does): a y that a later launch reads as an image sequence must have had them zeroed once. */
int lad_f16_block_fwd(const void *x, const void *wt1, const float *scale1, const float *shift1, const void *wt2,
                      const float *scale2, const float *shift2, void *y, int64_t batch, int32_t H, int32_t W, int32_t channels,
                      void *stream);
/* Round 6: lad_f16_stem_fwd + lad_f16_block_fwd (64 channels) for the FIRST residual block of the sliding-window strips in ONE launch
 * (models.py:224 + :110-115 in eval mode): strip b = frames [b, b + H) of `feat` ((frames, W) float32, zero-padded above and below, zeros
 * from frames_avail on).  The stem of a strip's rows 1 .. H - 2 is not recomputed: it is read from rows stream_row0 + b + 1 .. of
 * `stream_act` = lad_f16_stem_fwd over the whole frame stream as ONE image of stream_rows rows (frame 0 of `feat` is its frame
 * stream_row0); rows 0 and H - 1 -- the ones that see the strip's own padding -- are computed inside the launch from stem_weight
 * (conv1.weight) and the folded bn1 (stem_scale, stem_shift).  No strip-sized stem tensor is written or read.  Results identical to the
 * two calls; LAD_NOT_COVERED (nothing launched) outside lad_f16_block_fwd's 64-channel coverage or when W > 64 (a lane per column). */
int lad_f16_block_fwd_stem_rows(const void *stream_act, int64_t stream_rows, int64_t stream_row0, const float *feat, int64_t frames_avail,
                                const float *stem_weight, const float *stem_scale, const float *stem_shift, const void *wt1,
                                const float *scale1, const float *shift1, const void *wt2, const float *scale2, const float *shift2, void *y,
                                int64_t batch, int32_t H, int32_t W, void *stream);
/* Round 6: lad_f16_conv_s2_fwd_mapped_sc for the LEVEL-2 STRIPS (phases = 1, cin = 64, cout = 32, relu = 1, out_rows > 0: block2.0's
 * 3x3 stride-2 convolution -> out and its 1x1 shortcut -> out_sc, models.py:98-106, on the strip images of engine._eval_level2_shared)
 * with the input rows resident in LDS as parity classes (filled by LDS-DMA, a third of a strip image at a time, double-buffered) instead
 * of gathered per lane.  Same arguments; identical results on every output a window uses.  LAD_NOT_COVERED (nothing launched) for
 * other geometries than W = 44, out_rows = 12: the caller issues lad_f16_conv_s2_fwd_mapped_sc. */
int lad_f16_conv_s2_strips_fwd(const void *act, const void *wt, const float *scale, const float *shift, void *out, const void *wt_sc,
                               const float *scale_sc, const float *shift_sc, void *out_sc, int64_t n_windows, int32_t H, int32_t W,
                               int32_t band, int32_t strip_rows, int64_t bottom_image0, int64_t stream_row0, int64_t act_rows,
                               int32_t out_rows, void *stream);
/* Round 6: EVERYTHING BEHIND THE SHARED LEVEL 2 of the fp16 sliding-window path in ONE launch -- block3.0 (3x3 stride 2 through the
 * window map + 1x1 shortcut + conv2), block3.1, block4.0, block4.1, AvgPool2d(4) and the classifier, one probability per window:
 * models.py:226-239 in eval mode for the windows of segment_laughter.py:90-101.  `act`, H, W, band, strip_rows, bot_img0, stream_row0,
 * phases (2), phase_img, act_rows: exactly what lad_f16_conv_s2_fwd_mapped takes for the level-2 buffer (32 channels).  conv_params: HOST
 * array of 30 device pointers = {lad_f16_pack_weights image, folded scale, folded shift} of block3.0 conv1, block3.0 shortcut, block3.0
 * conv2, block3.1 conv1, block3.1 conv2 and the same five of block4 (16 channels); head_params, F: as lad_head_fwd_eval.  One
 * 1024-thread workgroup per CU keeps a window in LDS from its level-2 rows to its probability.  Bit-identical to the launches it replaces
 * (lad_f16_conv_s2_fwd_mapped(_sc), lad_f16_conv_fwd, lad_f16_block_fwd, lad_f16_conv_s2_fwd_sc, lad_f16_pool_fwd, lad_head_fwd_eval).
 * Returns LAD_NOT_COVERED (nothing launched) when a window does not fit a CU's LDS or H / W are odd: the caller issues those launches. */
int lad_f16_tail_fwd(const void *act, int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows, int64_t bot_img0,
                     int64_t stream_row0, int32_t phases, int64_t phase_img, int64_t act_rows, const void *const *conv_params,
                     const float *const *head_params, int32_t F, float *probs, void *stream);
/* The stride-2 convolutions behind the level-1 layers in the sliding-window path (engine._forward_eval_stream), reading every
 * window's rows from where they lie -- `act` = the n_windows + H - 2 band strip images of 2 * band rows followed by the stream image
 * of n_windows + H - 1 rows (the operands of lad_assemble_windows, in ONE buffer) -- instead of from an assembled copy: identical
 * results, one 1.2 GB write + read per 2048 windows less.  (cin, cout, taps) = (64, 32, 9) or (64, 32, 1). */
int lad_f16_conv_s2_fwd_windows(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                                int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t cin, int32_t cout, int32_t taps,
                                int32_t relu, void *stream);
/* The general form (round 3: the SECOND resolution level is shared between the windows too).  `act` (act_rows rows of cin halfs)
 * holds strip images of strip_rows rows -- window b's rows [0, strip_rows) in image b, its rows [H - strip_rows, H) in image
 * bottom_image0 + b (n_windows for strips of their own, H - strip_rows where one strip serves two windows as in
 * lad_assemble_windows) -- of which the first / last `band` rows are used, and, from row stream_row0 on, the stream image(s)
 * every other row comes from:
 * phases 1: row y of window b is stream row b + y; phases 2 (the input level lies behind a stride-2 layer, so windows of even
 * and odd b see different samplings): two stream images phase_rows rows apart, row y of window b = row (b >> 1) + y of image
 * b & 1.  out_rows 0: whole windows out (n_windows images of (H + 1) / 2 rows).  out_rows > 0 (even; H even; phases 1; paired
 * input strips): the NEXT level's strips out, paired the same way -- n_windows + 2 (H / 2 - out_rows) images of out_rows rows,
 * image s = the first out_rows / 2 output rows of window s over the last out_rows / 2 of window s - 2 (H / 2 - out_rows).
 * (cin, cout) = (64, 32) or (32, 16), taps 9 or 1. */
int lad_f16_conv_s2_fwd_mapped(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                               int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows, int64_t bottom_image0,
                               int64_t stream_row0, int32_t phases, int64_t phase_rows, int64_t act_rows, int32_t out_rows,
                               int32_t cin, int32_t cout, int32_t taps, int32_t relu, void *stream);
/* A down-sampling block's conv1 (3x3 stride 2 + BatchNorm + ReLU -> out) and its 1x1 stride-2 shortcut (+ BatchNorm -> out_sc) in ONE
 * launch (round 5): the shortcut reads exactly the rows the 3x3's centre tap gathers (models.py:98-106, eval mode).  Identical bits to
 * lad_f16_conv_s2_fwd(taps 9, relu) + lad_f16_conv_s2_fwd(taps 1, relu 0) / the two lad_f16_conv_s2_fwd_mapped calls on the same input.
 * (cin, cout) = (64, 32), (32, 16), and for the plain form (16, 16); wt_sc: lad_f16_pack_weights(taps = 1); out_sc != out. */
int lad_f16_conv_s2_fwd_sc(const void *in, const void *wt, const float *scale, const float *shift, void *out, const void *wt_sc,
                           const float *scale_sc, const float *shift_sc, void *out_sc, int64_t batch, int32_t H, int32_t W,
                           int32_t cin, int32_t cout, int32_t relu, void *stream);
int lad_f16_conv_s2_fwd_mapped_sc(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                                  const void *wt_sc, const float *scale_sc, const float *shift_sc, void *out_sc, int64_t n_windows,
                                  int32_t H, int32_t W, int32_t band, int32_t strip_rows, int64_t bottom_image0, int64_t stream_row0,
                                  int32_t phases, int64_t phase_rows, int64_t act_rows, int32_t out_rows, int32_t cin, int32_t cout,
                                  int32_t relu, void *stream);
int lad_f16_pool_fwd(const void *x, float *pooled, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream);

/* clip_grad_norm_ + Adam + zero_grad on a flat buffer (train.py:291-295) */
int32_t lad_grad_sumsq_partials(void);
/* step_counter (DEVICE int64, may be NULL): lad_grad_sumsq increments it, lad_adam_step then takes the step number
 * for the bias corrections from it instead of the host argument `step` -- what a captured hipGraph needs. */
int lad_grad_sumsq(const float *grad, int64_t n, float *partials, int64_t *step_counter, void *stream);
/* acc += scale * grad: gradient accumulation over batches (train.py:287-289, loss / gradient_accumulation_steps). */
int lad_grad_accumulate(float *acc, const float *grad, int64_t n, double scale, void *stream);
int lad_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                  const float *sumsq_partials, double grad_scale, double max_norm, double lr, double beta1,
                  double beta2, double eps, int64_t step, const int64_t *step_counter, int32_t zero_grad,
                  float *norm_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LAD_HIP_H */
