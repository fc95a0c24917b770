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
    LAD_ERR_NOMEM = -3
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
/* frames produced for a clip of `samples_per_clip` samples (same formula the kernel uses) */
int64_t lad_fbank_num_frames(const void *plan, int64_t samples_per_clip);
/* pcm: float[n_clips][samples_per_clip] in [-1,1];  out: float[n_clips][T][n_out], n_out = n_mfcc ? n_mfcc : n_mels */
int lad_fbank_forward(void *plan, const float *pcm, int64_t n_clips, int64_t samples_per_clip,
                      float *out, void *stream);
/* one long channel (load_data.py:44-49: whole file as a single cut): out float[T][n_out] */
int lad_fbank_forward_long(void *plan, const float *pcm, int64_t n_samples, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LAD_HIP_H */
