// Fused feature extractor for gfx950: framed PCM -> DC removal -> pre-emphasis -> window -> real FFT-512
// -> power -> mel filterbank -> log [-> DCT-II].  One kernel, one HBM read of the PCM, one HBM write of
// the features.
//
// Replaces Lhotse `Fbank.extract` as configured by the reference (utils/utils.py:25, config.py:28-31;
// reached from load_data.py:49 and compute_features.py:105-109).  Algorithm statement: DESIGN.md section 3.
//
// Mapping to the hardware
//   * a workgroup (8 wavefronts) owns FRAMES_PER_WG consecutive frames of one clip; the PCM span those
//     frames cover is staged once into LDS with coalesced loads (each sample is reused by 2.5 frames);
//   * one wavefront computes one frame at a time: a 512-point real FFT done as a 256-point complex
//     radix-4 Stockham FFT, 4 points per lane, exchanging through a per-wave LDS buffer (in-order LDS
//     within a wave, so no workgroup barrier inside the frame loop);
//   * the frame mean is a wavefront shuffle reduction; window and twiddle factors are frame-invariant
//     per lane and live in registers; the banded mel filterbank (and the DCT matrix) live in LDS;
//   * the F x n_out output tile is collected in LDS and written back with coalesced stores.
#include "lad_common.h"
#include "lad_fbank16.h"

#include <cmath>
#include <vector>

namespace {

constexpr int NFFT = 512;
constexpr int NCPLX = 256;          // complex FFT length
constexpr int NBINS = NFFT / 2 + 1; // 257
constexpr int WAVES = 8;            // measured: 4 waves 0.149 ms, 8 waves 0.131 ms per 1024 clips (2 WGs/CU either way useful; forcing 6 waves/SIMD: 0.135)
constexpr int THREADS = WAVES * 64;
constexpr int FRAMES_PER_WG = 25;
constexpr int FFT_PAD = NCPLX + NCPLX / 32;  // padded index i + (i>>5)
constexpr int PW_STRIDE = 264;

struct FbankParams {
    int frame_len, hop, n_mels, n_mfcc, pad_mode, log_mode, remove_dc;
    float preemph, log_floor;
    int left_off;  // samples before t*hop where frame t starts
    int maxlen;    // longest filter support (bins)
    int64_t n_samples;  // per clip
    int64_t n_frames;   // per clip
    int chunks_per_clip;
    const float *window;    // [NFFT]
    const float2 *tw256;    // [256]  exp(-2 pi i m / 256)
    const float2 *tw512;    // [257]  exp(-2 pi i k / 512)
    const int *mel_start;   // [64]
    const int *mel_len;     // [64]
    const float *mel_w;     // [maxlen][64]
    const float *dct;       // [n_mels][64] or nullptr
};

__device__ __forceinline__ int fpad(int i) { return i + (i >> 5); }

// LDS traffic inside one wavefront is issued in order; this keeps the compiler from moving accesses
// across the exchange points and retires the outstanding LDS operations.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

__device__ __forceinline__ void radix4(float2 (&v)[4]) {
    float2 t0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
    float2 t1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
    float2 t2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
    float2 t3 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
    v[0] = make_float2(t0.x + t2.x, t0.y + t2.y);
    v[2] = make_float2(t0.x - t2.x, t0.y - t2.y);
    v[1] = make_float2(t1.x + t3.y, t1.y - t3.x);  // t1 - i t3
    v[3] = make_float2(t1.x - t3.y, t1.y + t3.x);  // t1 + i t3
}

__device__ __forceinline__ int64_t map_sample(int64_t s, int64_t n, int pad_mode) {
    // returns the source index of padded position s, or -1 for an implicit zero
    if (s >= 0 && s < n) return s;
    if (pad_mode == LAD_PAD_KALDI_MIRROR) {
        s = (s < 0) ? (-1 - s) : (2 * n - 1 - s);
    } else if (pad_mode == LAD_PAD_CENTER_REFLECT) {
        s = (s < 0) ? (-s) : (2 * n - 2 - s);
    } else {
        return -1;
    }
    return (s >= 0 && s < n) ? s : -1;
}

__global__ __launch_bounds__(THREADS) void fbank_kernel(FbankParams p, const float *__restrict__ pcm,
                                                        float *__restrict__ out) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_out = p.n_mfcc > 0 ? p.n_mfcc : p.n_mels;

    const int64_t clip = blockIdx.x / p.chunks_per_clip;
    const int chunk = blockIdx.x % p.chunks_per_clip;
    const int64_t t0 = (int64_t)chunk * FRAMES_PER_WG;
    const int nfr = (int)min((int64_t)FRAMES_PER_WG, p.n_frames - t0);
    const int span_max = (FRAMES_PER_WG - 1) * p.hop + NFFT;
    const int span = (nfr - 1) * p.hop + NFFT;

    // ---- LDS carve-up -------------------------------------------------------------------------
    float *pcm_s = smem;                                  // [span_max]
    float *fft_re = pcm_s + ((span_max + 3) & ~3);        // [WAVES][FFT_PAD]
    float *fft_im = fft_re + WAVES * FFT_PAD;             // [WAVES][FFT_PAD]
    float *pw = fft_im + WAVES * FFT_PAD;                 // [WAVES][PW_STRIDE]
    float *melw = pw + WAVES * PW_STRIDE;                 // [maxlen][64]
    float *outbuf = melw + p.maxlen * 64;                 // [FRAMES_PER_WG][n_out]
    float *dct_s = outbuf + ((FRAMES_PER_WG * n_out + 3) & ~3);  // [n_mels][64] (only if n_mfcc)

    // ---- stage PCM span (coalesced), filterbank and DCT tables ------------------------------------
    const float *clip_pcm = pcm + clip * p.n_samples;
    const int64_t s_base = t0 * p.hop - p.left_off;
    for (int i = tid; i < span; i += THREADS) {
        int64_t src = map_sample(s_base + i, p.n_samples, p.pad_mode);
        pcm_s[i] = (src >= 0) ? clip_pcm[src] : 0.0f;
    }
    for (int i = tid; i < p.maxlen * 64; i += THREADS) melw[i] = p.mel_w[i];
    if (p.n_mfcc > 0)
        for (int i = tid; i < p.n_mels * 64; i += THREADS) dct_s[i] = p.dct[i];
    __syncthreads();

    // ---- frame-invariant per-lane constants -----------------------------------------------------
    float win[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        win[2 * r] = p.window[2 * lane + 128 * r];
        win[2 * r + 1] = p.window[2 * lane + 128 * r + 1];
    }
    float2 tw[3][3];  // [stage 1..3][r-1]
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int ns = 4 << (2 * s);          // 4, 16, 64
        const int k = lane & (ns - 1);
        const int mul = 64 / ns;
#pragma unroll
        for (int r = 1; r < 4; ++r) tw[s][r - 1] = p.tw256[r * k * mul];
    }
    float2 twp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) twp[r] = p.tw512[lane + 64 * r];
    const int m_start = (lane < p.n_mels) ? p.mel_start[lane] : 0;
    const int m_len = (lane < p.n_mels) ? p.mel_len[lane] : 0;
    const float inv_len = 1.0f / (float)p.frame_len;

    float *wre = fft_re + wave * FFT_PAD;
    float *wim = fft_im + wave * FFT_PAD;
    float *wpw = pw + wave * PW_STRIDE;

    for (int f = wave; f < nfr; f += WAVES) {
        const float *fr = pcm_s + f * p.hop;
        // -- load 8 samples per lane (+ the predecessor of each even sample for the pre-emphasis) --
        float x[8], xm[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i0 = 2 * lane + 128 * r;
            x[2 * r] = fr[i0];
            x[2 * r + 1] = fr[i0 + 1];
            xm[r] = fr[i0 > 0 ? i0 - 1 : 0];
        }
        float mu = 0.0f;
        if (p.remove_dc) {
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i0 = 2 * lane + 128 * r;
                s += (i0 < p.frame_len ? x[2 * r] : 0.0f) + (i0 + 1 < p.frame_len ? x[2 * r + 1] : 0.0f);
            }
            mu = wave_sum(s) * inv_len;
        }
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = x[2 * r] - mu, b = x[2 * r + 1] - mu, am = xm[r] - mu;
            v[r].x = (a - p.preemph * am) * win[2 * r];
            v[r].y = (b - p.preemph * a) * win[2 * r + 1];
        }

        // -- 256-point complex FFT: radix-4 Stockham, Ns = 1, 4, 16, 64 ----------------------------
        radix4(v);
        {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = fpad(4 * lane + r);
                wre[d] = v[r].x;
                wim[d] = v[r].y;
            }
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = fpad(lane + 64 * r);
                v[r] = make_float2(wre[s], wim[s]);
            }
            wave_lds_sync();
        }
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int ns = 4 << (2 * st);
#pragma unroll
            for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], tw[st][r - 1]);
            radix4(v);
            if (st < 2) {
                const int k = lane & (ns - 1);
                const int base = (lane / ns) * ns * 4 + k;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = fpad(base + r * ns);
                    wre[d] = v[r].x;
                    wim[d] = v[r].y;
                }
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = fpad(lane + 64 * r);
                    v[r] = make_float2(wre[s], wim[s]);
                }
                wave_lds_sync();
            }
        }
        // lane now holds Z[lane + 64 r]; publish for the conjugate-partner reads
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = fpad(lane + 64 * r);
            wre[d] = v[r].x;
            wim[d] = v[r].y;
        }
        wave_lds_sync();
        // -- real-FFT split + power spectrum ---------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const int kp = fpad((NCPLX - k) & (NCPLX - 1));
            const float pr = wre[kp], pi = wim[kp];
            const float er = 0.5f * (v[r].x + pr), ei = 0.5f * (v[r].y - pi);
            const float orr = 0.5f * (v[r].y + pi), oi = -0.5f * (v[r].x - pr);
            const float xr = er + (twp[r].x * orr - twp[r].y * oi);
            const float xi = ei + (twp[r].x * oi + twp[r].y * orr);
            wpw[k] = xr * xr + xi * xi;
        }
        if (lane == 0) {
            const float d = v[0].x - v[0].y;  // X[256] = Re Z[0] - Im Z[0]
            wpw[NCPLX] = d * d;
        }
        wave_lds_sync();
        // -- banded mel filterbank, log ------------------------------------------------------------------
        // Uniform trip count (maxlen, a multiple of 4; table rows past a filter's support hold 0.0f and re-read the
        // filter's own last bin, so they add +0.0f: bit-identical to stopping at m_len), four iterations' LDS reads in
        // flight per wait -- lane-dependent trip counts made this loop one LDS round trip per tap.
        float acc = 0.0f;
        const int m_last = max(m_len - 1, 0);
        for (int i0 = 0; i0 < p.maxlen; i0 += 4) {
            float pv[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pv[u] = wpw[m_start + min(i0 + u, m_last)];
                wv[u] = melw[(i0 + u) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(pv[u], wv[u], acc);
        }
        float val;
        if (p.log_mode == LAD_LOG_LN) val = logf(fmaxf(acc, p.log_floor));
        else if (p.log_mode == LAD_LOG_DB) val = 10.0f * log10f(fmaxf(acc, p.log_floor));
        else val = acc;
        wave_lds_sync();
        if (p.n_mfcc > 0) {
            wpw[lane] = (lane < p.n_mels) ? val : 0.0f;
            wave_lds_sync();
            float c = 0.0f;
            if (lane < p.n_mfcc) {
                int m = 0;
                for (; m + 3 < p.n_mels; m += 4) {  // four iterations' LDS reads in flight per wait, same summation order
                    float pv[4], dv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        pv[u] = wpw[m + u];
                        dv[u] = dct_s[(m + u) * 64 + lane];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) c = fmaf(pv[u], dv[u], c);
                }
                for (; m < p.n_mels; ++m) c = fmaf(wpw[m], dct_s[m * 64 + lane], c);
            }
            wave_lds_sync();
            if (lane < p.n_mfcc) outbuf[f * n_out + lane] = c;
        } else if (lane < p.n_mels) {
            outbuf[f * n_out + lane] = val;
        }
    }
    __syncthreads();
    float *dst = out + (clip * p.n_frames + t0) * n_out;
    for (int i = tid; i < nfr * n_out; i += THREADS) dst[i] = outbuf[i];
}

struct FbankPlan {
    lad_fbank_cfg cfg;
    int left_off = 0;
    int maxlen = 0;
    size_t lds_bytes = 0;
    float *d_window = nullptr;
    float2 *d_tw256 = nullptr;
    float2 *d_tw512 = nullptr;
    int *d_mel_start = nullptr;
    int *d_mel_len = nullptr;
    float *d_mel_w = nullptr;
    float *d_dct = nullptr;
    lad_fb16::Fast *fast = nullptr;  // tables of the 16-lanes-per-frame kernel (fbank16.hip), or nullptr
    bool force_general = false;      // lad_fbank_plan_set_kernel: tests compare the two kernels
};

int64_t num_frames(const lad_fbank_cfg &c, int64_t n) {
    if (c.pad_mode == LAD_PAD_KALDI_MIRROR) return (n + c.hop / 2) / c.hop;
    return 1 + n / c.hop;
}

template <typename T>
int upload(T **dst, const std::vector<T> &src) {
    LAD_HIP_CHECK(hipMalloc((void **)dst, src.size() * sizeof(T)));
    LAD_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LAD_OK;
}

}  // namespace

extern "C" int lad_fbank_plan_create(const lad_fbank_cfg *cfg, const float *window, const float *melbank,
                                     const float *dct, void **plan_out) {
    using namespace lad;
    LAD_REQUIRE(cfg && window && melbank && plan_out, "lad_fbank_plan_create: null argument");
    LAD_REQUIRE(cfg->n_fft == NFFT, "lad_fbank_plan_create: n_fft must be %d (got %d)", NFFT, cfg->n_fft);
    LAD_REQUIRE(cfg->frame_len >= 2 && cfg->frame_len <= NFFT, "frame_len %d out of range", cfg->frame_len);
    LAD_REQUIRE(cfg->hop >= 1 && cfg->hop <= NFFT, "hop %d out of range", cfg->hop);
    LAD_REQUIRE(cfg->n_mels >= 1 && cfg->n_mels <= 64, "n_mels %d out of range 1..64", cfg->n_mels);
    LAD_REQUIRE(cfg->n_mfcc >= 0 && cfg->n_mfcc <= 64, "n_mfcc %d out of range 0..64", cfg->n_mfcc);
    LAD_REQUIRE(cfg->pad_mode >= 0 && cfg->pad_mode <= 2, "bad pad_mode %d", cfg->pad_mode);
    LAD_REQUIRE(cfg->log_mode >= 0 && cfg->log_mode <= 2, "bad log_mode %d", cfg->log_mode);
    LAD_REQUIRE(cfg->n_mfcc == 0 || dct, "n_mfcc > 0 needs a dct matrix");

    FbankPlan *pl = new FbankPlan();
    pl->cfg = *cfg;
    pl->left_off = (cfg->pad_mode == LAD_PAD_KALDI_MIRROR) ? (cfg->frame_len - cfg->hop) / 2 : NFFT / 2;

    // banded form of the dense filterbank
    std::vector<int> start(64, 0), len(64, 0);
    int maxlen = 1;
    for (int m = 0; m < cfg->n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int b = 0; b < NBINS; ++b)
            if (melbank[(size_t)b * cfg->n_mels + m] != 0.0f) {
                if (lo < 0) lo = b;
                hi = b;
            }
        if (lo >= 0) {
            start[m] = lo;
            len[m] = hi - lo + 1;
            if (len[m] > maxlen) maxlen = len[m];
        }
    }
    maxlen = (maxlen + 3) & ~3;  // the kernel walks the band four taps at a time; the padding rows stay 0.0f
    pl->maxlen = maxlen;
    std::vector<float> w((size_t)maxlen * 64, 0.0f);
    for (int m = 0; m < cfg->n_mels; ++m)
        for (int i = 0; i < len[m]; ++i) w[(size_t)i * 64 + m] = melbank[(size_t)(start[m] + i) * cfg->n_mels + m];
    std::vector<float> win(window, window + NFFT);
    std::vector<float2> tw256(256), tw512(NBINS);
    for (int m = 0; m < 256; ++m) {
        double a = -2.0 * M_PI * m / 256.0;
        tw256[m] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int k = 0; k < NBINS; ++k) {
        double a = -2.0 * M_PI * k / 512.0;
        tw512[k] = make_float2((float)cos(a), (float)sin(a));
    }
    int rc;
    if ((rc = upload(&pl->d_window, win)) || (rc = upload(&pl->d_tw256, tw256)) || (rc = upload(&pl->d_tw512, tw512)) ||
        (rc = upload(&pl->d_mel_start, start)) || (rc = upload(&pl->d_mel_len, len)) || (rc = upload(&pl->d_mel_w, w))) {
        lad_fbank_plan_destroy(pl);
        return rc;
    }
    if (cfg->n_mfcc > 0) {
        std::vector<float> d((size_t)cfg->n_mels * 64, 0.0f);
        for (int m = 0; m < cfg->n_mels; ++m)
            for (int c = 0; c < cfg->n_mfcc; ++c) d[(size_t)m * 64 + c] = dct[(size_t)m * cfg->n_mfcc + c];
        if ((rc = upload(&pl->d_dct, d))) {
            lad_fbank_plan_destroy(pl);
            return rc;
        }
    }
    const int span_max = (FRAMES_PER_WG - 1) * cfg->hop + NFFT;
    size_t floats = ((span_max + 3) & ~3) + 2 * WAVES * FFT_PAD + WAVES * PW_STRIDE + (size_t)maxlen * 64 +
                    ((FRAMES_PER_WG * (cfg->n_mfcc > 0 ? cfg->n_mfcc : cfg->n_mels) + 3) & ~3) +
                    (cfg->n_mfcc > 0 ? (size_t)cfg->n_mels * 64 : 0);
    pl->lds_bytes = floats * sizeof(float);
    if (pl->lds_bytes > 160 * 1024) {
        lad_fbank_plan_destroy(pl);
        return fail(LAD_ERR_INVALID, "fbank plan needs %zu B of LDS (> 160 KiB)", pl->lds_bytes);
    }
    if (pl->lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fbank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)pl->lds_bytes);
        if (e != hipSuccess) {
            lad_fbank_plan_destroy(pl);
            return fail(LAD_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        }
    }
    if ((rc = lad_fb16::build(pl->cfg, window, melbank, &pl->fast))) {
        lad_fbank_plan_destroy(pl);
        return rc;
    }
    *plan_out = pl;
    return LAD_OK;
}

extern "C" int lad_fbank_plan_destroy(void *plan) {
    if (!plan) return LAD_OK;
    FbankPlan *pl = (FbankPlan *)plan;
    (void)hipFree(pl->d_window);
    (void)hipFree(pl->d_tw256);
    (void)hipFree(pl->d_tw512);
    (void)hipFree(pl->d_mel_start);
    (void)hipFree(pl->d_mel_len);
    (void)hipFree(pl->d_mel_w);
    (void)hipFree(pl->d_dct);
    lad_fb16::destroy(pl->fast);
    delete pl;
    return LAD_OK;
}

extern "C" int lad_fbank_plan_set_kernel(void *plan, int32_t which) {
    using namespace lad;
    LAD_REQUIRE(plan, "lad_fbank_plan_set_kernel: null plan");
    LAD_REQUIRE(which == 0 || which == 1, "lad_fbank_plan_set_kernel: which must be 0 (automatic) or 1 (general kernel)");
    ((FbankPlan *)plan)->force_general = which == 1;
    return LAD_OK;
}

extern "C" int lad_fbank_plan_has_fast_kernel(const void *plan) {
    return plan != nullptr && ((const FbankPlan *)plan)->fast != nullptr;
}

extern "C" int64_t lad_fbank_num_frames(const void *plan, int64_t samples_per_clip) {
    if (!plan || samples_per_clip < 0) return -1;
    return num_frames(((const FbankPlan *)plan)->cfg, samples_per_clip);
}

extern "C" int lad_fbank_forward(void *plan, const float *pcm, int64_t n_clips, int64_t samples_per_clip, float *out,
                                 void *stream) {
    using namespace lad;
    LAD_REQUIRE(plan, "lad_fbank_forward: null plan");
    FbankPlan *pl = (FbankPlan *)plan;
    LAD_REQUIRE(n_clips >= 0 && samples_per_clip >= 0, "lad_fbank_forward: negative size");
    if (n_clips == 0) return LAD_OK;
    // mirror / reflect padding reads at most NFFT samples past either end of the clip
    LAD_REQUIRE(samples_per_clip >= NFFT, "lad_fbank_forward: clips shorter than %d samples are not supported (got %lld)",
                NFFT, (long long)samples_per_clip);
    LAD_REQUIRE(pcm && out, "lad_fbank_forward: null buffer");
    const int64_t T = num_frames(pl->cfg, samples_per_clip);
    if (T == 0) return LAD_OK;
    if (!pl->force_general && lad_fb16::eligible(pl->fast, n_clips, samples_per_clip, pcm))
        return lad_fb16::launch(pl->fast, pl->cfg, pl->left_off, pcm, n_clips, samples_per_clip, T, out, (hipStream_t)stream);
    const int64_t chunks = ceil_div(T, FRAMES_PER_WG);
    LAD_REQUIRE(n_clips * chunks < (int64_t)1 << 31, "lad_fbank_forward: grid too large");
    FbankParams p;
    p.frame_len = pl->cfg.frame_len;
    p.hop = pl->cfg.hop;
    p.n_mels = pl->cfg.n_mels;
    p.n_mfcc = pl->cfg.n_mfcc;
    p.pad_mode = pl->cfg.pad_mode;
    p.log_mode = pl->cfg.log_mode;
    p.remove_dc = pl->cfg.remove_dc;
    p.preemph = pl->cfg.preemph;
    p.log_floor = pl->cfg.log_floor;
    p.left_off = pl->left_off;
    p.maxlen = pl->maxlen;
    p.n_samples = samples_per_clip;
    p.n_frames = T;
    p.chunks_per_clip = (int)chunks;
    p.window = pl->d_window;
    p.tw256 = pl->d_tw256;
    p.tw512 = pl->d_tw512;
    p.mel_start = pl->d_mel_start;
    p.mel_len = pl->d_mel_len;
    p.mel_w = pl->d_mel_w;
    p.dct = pl->d_dct;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)(n_clips * chunks)), dim3(THREADS), pl->lds_bytes,
                       (hipStream_t)stream, p, pcm, out);
    return check_launch("fbank_kernel");
}

extern "C" int lad_fbank_forward_long(void *plan, const float *pcm, int64_t n_samples, float *out, void *stream) {
    return lad_fbank_forward(plan, pcm, 1, n_samples, out, stream);
}
