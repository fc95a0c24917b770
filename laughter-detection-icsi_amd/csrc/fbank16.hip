// Fast path of the fused feature extractor for gfx950 (csrc/fbank.hip keeps the general kernel and the plan).
//
// Same arithmetic as fbank.hip (framed PCM -> DC removal -> pre-emphasis -> window -> real FFT-512 -> power -> mel ->
// log; Lhotse `Fbank.extract` as configured by utils/utils.py:25, config.py:28-31), laid out so that a frame costs
// ~160 vector instructions per wavefront instead of ~450 and a third of the LDS operations
// (profiles/r02_fbank_pmc_before.json vs profiles/r02_fbank_pmc.json; the round-1 kernel spent its time in VALU issue and
// in three dependent LDS exchanges per FFT, not in HBM):
//
//   * 16 lanes own one frame, a wavefront owns FOUR frames: the 256-point complex FFT behind the 512-point real FFT is a
//     16 x 16 decomposition with SIXTEEN points per lane, so both passes are 16-point FFTs entirely in registers
//     (radix-4 x radix-4, packed-f32 v_pk_add/mul/fma) and ONE transposition through LDS sits between them (row stride 17
//     dwords: conflict-free).  Pass 1 works on (re, im) pairs as the samples arrive from LDS, pass 2 on planar pairs of
//     two elements as the transposed values arrive: no register moves on either side;
//   * the real-FFT split pairs bin k with 256-k; lane i holds bins i + 16 j, its partner lane 16-i holds 256-k, and each
//     lane handles 8 pairs: one (S, T) per pair gives |X[k]|^2 = |S+T|^2/4 and |X[256-k]|^2 = |S-T|^2/4, so only 8
//     complex values cross lanes (ds_bpermute) and no work is done twice;
//   * wavefronts are independent and persistent: each stages the PCM span of ITS four frames privately (16-byte loads,
//     pre-emphasis e[j] = x[j] - p x[j-1] and 16-sample sums for the frame mean once per staged sample:
//     (x[j]-mu) - p (x[j-1]-mu) = e[j] - (1-p) mu), the span of the next group travels HBM -> registers while the current
//     one is transformed, and no workgroup barrier follows the table load;
//   * the power spectra of the four frames are written as one float4 per bin, so a mel tap is one 16-byte LDS read feeding
//     two packed FMAs for four frames; the triangular filters are cut into <= 64 parts of equal maximum length (host
//     table) = one part per lane; a filter is the sum of its (one or two) parts in fixed order.
//
// Measured (1024 clips, one MI355X): 0.131 ms -> 0.059 ms = 1.4 TB/s of algorithmic traffic.  What bounds it now
// (tools/experiments/valu_rate.hip, PMC): a wavefront issues one vector instruction per 5.8 cycles whatever its kind, a
// SIMD retires a plain one per 2.5 and a packed one per 4.1 cycles (VALU 38 % busy), the LDS pipe is 53 % busy
// (transposition 27 % of it, spectra + mel taps 31 %), and 12 or 16 wavefronts per CU make no difference: the frame pass
// is a chain of ~9 LDS round trips whose latency grows with the load on the LDS pipe.
//
// Eligibility (else lad_fbank_forward takes the general kernel): hop % 16 == 0 and <= 160, frame_len % 16 == 0, no DCT,
// the filter parts fit 64 lanes with at most LMAX_CAP taps, clips start on 16-byte boundaries.
#include "lad_common.h"
#include "lad_fbank16.h"

#include <cmath>
#include <vector>

namespace lad_fb16 {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NFFT = 512;
constexpr int NBINS = 257;
constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int FPW = 4;                    // frames per wavefront
constexpr int TR_ROW = 17;                // dwords per transposition row (16 + 1: conflict-free reads)
constexpr int TR_FRAME = 16 * TR_ROW;     // 272
constexpr int TR_WAVE = FPW * TR_FRAME;   // 1088 dwords: also holds the wave's power spectra [272 bins][4 frames]
constexpr int LMAX_CAP = 32;

struct Params {
    int hop, hop4, hb, nblk, n_mels, pad_mode, log_mode, left_off, lmax, np_max, groups_per_clip;
    float preemph, dc_scale, log_floor, log_scale;
    int64_t n_samples, n_frames;
    const v2f *win;       // [16 n1][16 i]   window at samples (32 n1 + 2 i, 32 n1 + 2 i + 1)
    const v2f *tw1;       // [16 k1][16 i]   W256^(i k1)
    const v4f *tw2;       // [4 c][16 i]     (re a, re b | im a, im b) of -i W512^k for k = i + 16 c, i + 16 (c + 8)
    const float *melw;    // [LT][64]        0.25 * filter weight of tap t of the lane's part (zero past the part)
    const int *part_bin0; // [64]
    const int *filt_part; // [64]  first part of filter m
    const int *filt_np;   // [64]  number of parts of filter m (0 for m >= n_mels)
};

// ---- pair helpers.  Rounds 2-3 wrote these as single VOP3P instructions (v_pk_add_f32 / v_pk_fma_f32 with op_sel and neg
// modifiers); round 4 found that packed-f32 instructions are what a wave loses a row of 16 lanes of when the GPU switches it out and
// back in next to a second process (profiles/r04_slp_nondeterminism.md), and that a packed instruction issues in the time of two plain
// ones anyway: the library is built without them (build.py), and these are plain C.
__device__ __forceinline__ v2f lohi_addsub(v2f a, v2f b) { return v2f{a.x + b.y, a.x - b.y}; }   // (a.lo + b.hi, a.lo - b.hi)
__device__ __forceinline__ v2f lohi_subadd(v2f a, v2f b) { return v2f{a.x - b.y, a.x + b.y}; }   // (a.lo - b.hi, a.lo + b.hi)

__device__ __forceinline__ void wave_fence() {
    // cross-lane hand-over through LDS inside one wavefront: the hardware executes a wave's LDS operations in order;
    // this only keeps the compiler from moving them across the hand-over point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float s) {  // sum over the 16 lanes of a DPP row, same bits in every lane
    s += dpp<0x128>(s);  // row_ror:8
    s += dpp<0x124>(s);  // row_ror:4
    s += dpp<0x122>(s);  // row_ror:2
    s += dpp<0x121>(s);  // row_ror:1
    return s;
}
__device__ __forceinline__ float bperm(int addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

__device__ __forceinline__ v2f add_mi(v2f a, v2f b) { return v2f{a.x + b.y, a.y - b.x}; }   // a + (-i) b on an (re, im) pair
__device__ __forceinline__ v2f add_pi(v2f a, v2f b) { return v2f{a.x - b.y, a.y + b.x}; }   // a + i b
// (re, im) pair times w = (w.x, w.y): a.xx * w + a.yy * (-w.y, w.x); a table holds each twiddle once (8-byte LDS reads)
__device__ __forceinline__ v2f cmul(v2f a, v2f w) {
    v2f d = a.xx * w;
    d.x = fmaf(-a.y, w.y, d.x);
    d.y = fmaf(a.y, w.x, d.y);
    return d;
}
__device__ __forceinline__ v2f cmulc(v2f a, float wx, float wy) { return cmul(a, v2f{wx, wy}); }

#define LAD_R4(a, b, c, d)                                        \
    {                                                             \
        const v2f t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d; \
        a = t0 + t2;                                              \
        c = t0 - t2;                                              \
        b = add_mi(t1, t3);                                       \
        d = add_pi(t1, t3);                                       \
    }

// 16-point complex FFT on (re, im) pairs -- the form in which a frame's samples arrive from LDS (x[2n], x[2n+1]).
// Natural order in and out, all indices compile-time (x[n], n = 4a + b; X[c + 4d]).
__device__ __forceinline__ void fft16_interleaved(v2f (&x)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
    LAD_R4(x[0], x[4], x[8], x[12]);
    LAD_R4(x[1], x[5], x[9], x[13]);
    LAD_R4(x[2], x[6], x[10], x[14]);
    LAD_R4(x[3], x[7], x[11], x[15]);
    // x[4c + b] *= W16^(b c)
    x[5] = cmulc(x[5], C1, -S1);    // m = 1
    x[6] = cmulc(x[6], H, -H);      // m = 2
    x[7] = cmulc(x[7], S1, -C1);    // m = 3
    x[9] = cmulc(x[9], H, -H);      // m = 2
    x[10] = add_mi(v2f{0.0f, 0.0f}, x[10]);  // m = 4: * (-i)
    x[11] = cmulc(x[11], -H, -H);   // m = 6
    x[13] = cmulc(x[13], S1, -C1);  // m = 3
    x[14] = cmulc(x[14], -H, -H);   // m = 6
    x[15] = cmulc(x[15], -C1, S1);  // m = 9
    LAD_R4(x[0], x[1], x[2], x[3]);
    LAD_R4(x[4], x[5], x[6], x[7]);
    LAD_R4(x[8], x[9], x[10], x[11]);
    LAD_R4(x[12], x[13], x[14], x[15]);
    // result X[c + 4d] sits in x[4c + d]: transpose the 4 x 4 index grid (register renaming)
    v2f t;
#define LAD_SWAP(i, j) t = x[i], x[i] = x[j], x[j] = t
    LAD_SWAP(1, 4); LAD_SWAP(2, 8); LAD_SWAP(3, 12); LAD_SWAP(6, 9); LAD_SWAP(7, 13); LAD_SWAP(11, 14);
#undef LAD_SWAP
}
#undef LAD_R4

// After the transposition complex data is held "planar in pairs": R[m] = (re x[e0], re x[e1]), I[m] = (im x[e0], im x[e1]) for two elements of the
// sequence, so that every butterfly is a packed instruction on two elements and LDS reads of two neighbouring real (or
// imaginary) parts land directly in a register pair.
//
// 16-point complex FFT on planar pairs.  In:  pair m = elements (2m, 2m+1).  Out: pair 2c = (X[c], X[c+8]), pair 2c+1 = (X[c+4], X[c+12]).
// n = 4a + b: four radix-4 butterflies over a (two per packed instruction), twiddle W16^(bc), four over b (inside pairs).
__device__ __forceinline__ void fft16(v2f (&R)[8], v2f (&I)[8]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const v2f t0r = R[p] + R[4 + p], t0i = I[p] + I[4 + p], t1r = R[p] - R[4 + p], t1i = I[p] - I[4 + p];
        const v2f t2r = R[2 + p] + R[6 + p], t2i = I[2 + p] + I[6 + p], t3r = R[2 + p] - R[6 + p], t3i = I[2 + p] - I[6 + p];
        R[p] = t0r + t2r, I[p] = t0i + t2i;          // c = 0
        R[4 + p] = t0r - t2r, I[4 + p] = t0i - t2i;  // c = 2
        R[2 + p] = t1r + t3i, I[2 + p] = t1i - t3r;  // c = 1: t1 - i t3
        R[6 + p] = t1r - t3i, I[6 + p] = t1i + t3r;  // c = 3: t1 + i t3
    }
    // pair 2c + p holds Y[c] for b = (2p, 2p+1): multiply by W16^(bc) = (cos, -sin)(bc pi / 8)
#define LAD_TW(q, wr0, wi0, wr1, wi1)                                    \
    {                                                                     \
        const v2f wr = {wr0, wr1}, wi = {wi0, wi1};                       \
        const v2f nr = R[q] * wr - I[q] * wi, ni = R[q] * wi + I[q] * wr; \
        R[q] = nr, I[q] = ni;                                             \
    }
    LAD_TW(2, 1.0f, 0.0f, C1, -S1);   // c = 1: (W^0, W^1)
    LAD_TW(3, H, -H, S1, -C1);        //        (W^2, W^3)
    LAD_TW(4, 1.0f, 0.0f, H, -H);     // c = 2: (W^0, W^2)
    LAD_TW(5, 0.0f, -1.0f, -H, -H);   //        (W^4, W^6)
    LAD_TW(6, 1.0f, 0.0f, S1, -C1);   // c = 3: (W^0, W^3)
    LAD_TW(7, -H, -H, -C1, S1);       //        (W^6, W^9)
#undef LAD_TW
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const v2f ur = R[2 * c] + R[2 * c + 1], ui = I[2 * c] + I[2 * c + 1];  // (t0, t2)
        const v2f vr = R[2 * c] - R[2 * c + 1], vi = I[2 * c] - I[2 * c + 1];  // (t1, t3)
        R[2 * c] = lohi_addsub(ur, ur), I[2 * c] = lohi_addsub(ui, ui);        // (X0, X2) = (t0 + t2, t0 - t2)
        R[2 * c + 1] = lohi_addsub(vr, vi), I[2 * c + 1] = lohi_subadd(vi, vr);  // (X1, X3) = (t1 - i t3, t1 + i t3)
    }
}

__device__ __forceinline__ int map_sample(int s, int n, int pad_mode) {
    // source index of padded position s, or -1 for an implicit zero (same rule as fbank.hip)
    if (s >= 0 && s < n) return s;
    if (pad_mode == LAD_PAD_KALDI_MIRROR) s = (s < 0) ? (-1 - s) : (2 * n - 1 - s);
    else if (pad_mode == LAD_PAD_CENTER_REFLECT) s = (s < 0) ? (-s) : (2 * n - 2 - s);
    else return -1;
    return (s >= 0 && s < n) ? s : -1;
}
__device__ __forceinline__ float load_padded(const float *clip, int s, int n, int pad_mode) {
    const int src = map_sample(s, n, pad_mode);
    return src >= 0 ? clip[src] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wavefronts are independent: each one owns groups of FPW = 4 consecutive frames of one clip, stages the PCM span of ITS
// group privately (3 hops + 512 samples; neighbouring groups re-read the overlap from L2), and never meets a workgroup
// barrier after the tables are in LDS -- the first version staged 20 frames per workgroup behind two barriers per chunk
// and kept all five waves of a workgroup in the same phase at the same time (VALU busy 34 %).
// A wavefront is persistent: group gw, gw + total waves, ...; the span of the next group travels HBM -> registers while
// the current group is transformed.
constexpr int SLOTS = 4;                   // float4 slots per lane: (3 hop + 512) / 4 <= SLOTS * 64
constexpr int ES_WAVE = SLOTS * 64 * 4;    // 1024 floats of staged (pre-emphasised, padded) samples per wave
constexpr int BS_WAVE = ES_WAVE / 16;      // sums of 16 padded samples
constexpr int WAVE_LDS = ES_WAVE + BS_WAVE + FPW + TR_WAVE;   // + first-sample corrections + transposition / spectra

struct Staged {   // what a lane carries from the global loads of a span to their commit into LDS
    v4f x[SLOTS];
    float xp[SLOTS];
};

#ifdef LAD_STAMP
// diagnostic build only (tools/stamp_fbank.py): shader-clock stamps of wave 0 of a workgroup's SECOND group; never the product
__device__ unsigned long long lad_dbg_fb[16 * 4096];
#define LAD_FB_STAMP(k)                                                                                      \
    if (threadIdx.x == 0 && iter == 1 && blockIdx.x < 4096) lad_dbg_fb[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define LAD_FB_STAMP(k)
#endif

struct Group {
    const float *clip_pcm;
    float *out;      // first output row of the group
    int s_base;      // padded-signal position of the first staged sample
    int nfr;         // frames of this group (<= FPW)
    int interior;    // the whole span (and the sample before it) lies inside the clip: no padding rule needed
};

__device__ __forceinline__ Group locate(const Params &p, const float *pcm, float *out, unsigned gw) {
    const unsigned clip = gw / (unsigned)p.groups_per_clip;
    const int t0 = (int)(gw - clip * (unsigned)p.groups_per_clip) * FPW;
    Group G;
    G.clip_pcm = pcm + (int64_t)clip * p.n_samples;
    G.out = out + ((int64_t)clip * p.n_frames + t0) * p.n_mels;
    G.s_base = t0 * p.hop - p.left_off;   // t0 * hop < n_samples + hop < 2^31 (checked by launch())
    G.nfr = (int)min((int64_t)FPW, p.n_frames - t0);
    G.interior = (G.s_base >= 1 && G.s_base + ES_WAVE + 3 < (int)p.n_samples) ? 1 : 0;
    return G;
}

// Every lane issues the SAME eight loads whatever its position (addresses are clamped into the clip; edge groups are
// redone by stage_padded()): the number of loads in flight is a compile-time constant and nothing here selects between or
// copies values that are still in flight, so no wait is placed here and the wait in front of commit() can be "all but the
// four result stores issued after them".  (A first version branched per thread between 16-byte loads and padded scalar
// loads: the register moves at the merge made the compiler wait for each load where it was issued, 8 k cycles per chunk.)
__device__ __forceinline__ void issue_loads(const Params &p, const Group &G, int lane, Staged &st) {
    const int n = (int)p.n_samples;
    const int last4 = (n - 4) & ~3;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int s = G.s_base + 4 * (lane + 64 * k);
        st.x[k] = *reinterpret_cast<const v4f *>(G.clip_pcm + min(max(s, 0), last4));
        st.xp[k] = G.clip_pcm[min(max(s - 1, 0), n - 1)];
    }
}

// pre-emphasis e[j] = x[j] - p x[j-1] and the sums of 16 samples, once per staged sample
__device__ __forceinline__ void commit_slot(const Params &p, int q, v4f x, float xp, float *es, float *bsum) {
    v4f e;
    e.x = fmaf(-p.preemph, xp, x.x);
    e.y = fmaf(-p.preemph, x.x, x.y);
    e.z = fmaf(-p.preemph, x.y, x.z);
    e.w = fmaf(-p.preemph, x.z, x.w);
    *reinterpret_cast<v4f *>(es + 4 * q) = e;
    float qs = (x.x + x.y) + (x.z + x.w);
    qs += dpp<0xB1>(qs);  // quad_perm [1,0,3,2]
    qs += dpp<0x4E>(qs);  // quad_perm [2,3,0,1]
    if ((q & 3) == 0) bsum[q >> 2] = qs;
}

__device__ __forceinline__ void commit(const Params &p, int lane, const Staged &st, float *es, float *bsum, float *fix) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) commit_slot(p, lane + 64 * k, st.x[k], st.xp[k], es, bsum);
    // replicate padding of the pre-emphasis at the first sample of a frame: x[0] - p x[0] instead of e[0].  Frame r starts at
    // float4 slot r * hop / 4 of the span: the lane that staged it holds both samples.
#pragma unroll
    for (int r = 0; r < FPW; ++r) {
        const int q = r * p.hop4;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if ((q >> 6) == k && (q & 63) == lane) fix[r] = p.preemph * (st.xp[k] - st.x[k].x);
    }
}

// groups that touch a clip edge: the same staging through the padding rule, sample by sample (a few groups per clip)
__device__ __forceinline__ void stage_padded(const Params &p, const Group &G, int lane, float *es, float *bsum, float *fix) {
    const int n = (int)p.n_samples;
#pragma unroll 1
    for (int k = 0; k < SLOTS; ++k) {
        const int q = lane + 64 * k;
        const int s = G.s_base + 4 * q;
        v4f x;
        const float xp = load_padded(G.clip_pcm, s - 1, n, p.pad_mode);
        x.x = load_padded(G.clip_pcm, s, n, p.pad_mode);
        x.y = load_padded(G.clip_pcm, s + 1, n, p.pad_mode);
        x.z = load_padded(G.clip_pcm, s + 2, n, p.pad_mode);
        x.w = load_padded(G.clip_pcm, s + 3, n, p.pad_mode);
        commit_slot(p, q, x, xp, es, bsum);
        for (int r = 0; r < FPW; ++r)
            if (q == r * p.hop4) fix[r] = p.preemph * (xp - x.x);
    }
}

// NZ: rows of 32 samples of the 512-point frame that hold samples (32 NZ >= frame_len); LT: mel taps per lane.
template <int NZ, int LT>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void fbank16_kernel(
    Params p, const float *__restrict__ pcm, float *__restrict__ out, unsigned total_groups) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;   // frame slot of the lane's 16-lane row
    const int i = lane & 15;

    v2f *win_s = reinterpret_cast<v2f *>(smem);            // [256]
    v2f *tw1_s = win_s + 256;                              // [256]
    v4f *tw2_s = reinterpret_cast<v4f *>(tw1_s + 256);     // [64]
    float *mw_s = reinterpret_cast<float *>(tw2_s + 64);   // [LT][64]
    float *mine = mw_s + LT * 64 + wave * WAVE_LDS;        // this wave's private part
    float *es = mine;                   // [ES_WAVE]   pre-emphasised padded samples of the group's span
    float *bsum = es + ES_WAVE;         // [BS_WAVE]   sums of 16 padded samples
    float *fix = bsum + BS_WAVE;        // [FPW]       first-sample correction p (x[s-1] - x[s])
    float *tr = fix + FPW;              // [TR_WAVE]   transposition buffer, then power spectra, then part sums

    // ---- once per workgroup: tables into LDS (the only workgroup barrier), per-lane mel constants into registers -----------
    for (int k = tid; k < 256; k += THREADS) {
        win_s[k] = p.win[k];
        tw1_s[k] = p.tw1[k];
    }
    if (tid < 64) tw2_s[tid] = p.tw2[tid];
    for (int k = tid; k < LT * 64; k += THREADS) mw_s[k] = p.melw[k];
    int bin0 = p.part_bin0[lane];
    int np = p.filt_np[lane];
    int fp = p.filt_part[lane];
    // have these three arrive HERE: their first use sits in the loop below, and a wait for them there would also wait, in
    // every iteration, for the prefetch that was just issued
    asm volatile("" : "+v"(bin0), "+v"(np), "+v"(fp));
    __syncthreads();

    const unsigned stride = gridDim.x * WAVES;
    unsigned gw = blockIdx.x * WAVES + wave;
    if (gw >= total_groups) return;
    Group G = locate(p, pcm, out, gw);
    Staged st;
    issue_loads(p, G, lane, st);
    // Results are stored one group late, right AFTER the next prefetch has been issued.  The wait in front of commit() is a
    // full vmcnt(0) whenever loads and stores are in flight together (they may retire out of order with respect to each
    // other): stores issued at the end of a frame pass would make every commit wait for their acknowledgement (measured:
    // 5 k cycles); stores issued one whole frame pass earlier have long retired.  After the loads, not before: a load that
    // re-used an address register of a store still in flight would have to wait for that store.
    v4f held = {0.0f, 0.0f, 0.0f, 0.0f};
    float *held_dst = nullptr;
    int held_n = 0;
    int iter = 0;
    (void)iter;
    for (;;) {
    LAD_FB_STAMP(0)
    commit(p, lane, st, es, bsum, fix);
    LAD_FB_STAMP(1)
    if (!G.interior) stage_padded(p, G, lane, es, bsum, fix);
    wave_fence();
    LAD_FB_STAMP(2)
    const unsigned gw_next = gw + stride;
    const bool has_next = gw_next < total_groups;
    const Group G_next = locate(p, pcm, out, has_next ? gw_next : gw);
    issue_loads(p, G_next, lane, st);   // in flight during the frame pass below
    if (lane < p.n_mels) {
#pragma unroll
        for (int r = 0; r < FPW; ++r)
            if (r < held_n) held_dst[(int64_t)r * p.n_mels] = held[r];
    }
    held_n = 0;
    LAD_FB_STAMP(3)
    const int nfr = G.nfr;
    {
    // ---- this lane's frame ------------------------------------------------------------------------------------------
    const int f = min(g, nfr - 1);  // slots past the last frame recompute it (results go to the scrap area)
    const float *ef = es + f * p.hop + 2 * i;
    float bs = 0.0f;
    {
        const float *b0 = bsum + f * p.hb;
        if (i < p.nblk) bs = b0[i];
        if (i + 16 < p.nblk) bs += b0[i + 16];
    }
    const float cm = p.dc_scale * row16_sum(bs);  // (1 - preemph) * mean, or 0 without DC removal
    const float fx = (i == 0) ? fix[f] : 0.0f;

    // ---- pass 1 on (re, im) pairs as they come from LDS: z[n1] = x[32 n1 + 2 i] + i x[32 n1 + 2 i + 1] -----------------------
    v2f z[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        if (n1 < NZ) {
            v2f e2 = *reinterpret_cast<const v2f *>(ef + 32 * n1);
            if (n1 == 0) e2.x += fx;
            z[n1] = (e2 - cm) * win_s[n1 * 16 + i];
        } else {
            z[n1] = v2f{0.0f, 0.0f};
        }
    }
    LAD_FB_STAMP(4)
    fft16_interleaved(z);  // over n1 (lane = n2 = i) -> z[k1]
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {  // twiddle W256^(n2 k1)
        z[k1] = cmul(z[k1], tw1_s[k1 * 16 + i]);
    }
    LAD_FB_STAMP(5)
    // ---- transposition inside the 16-lane row: lane n2 holds A[k1] -> lane k1 holds A[n2], now as planar pairs ---------
    // Both planes cross in ONE round trip (round 3): the real parts through `tr`, the imaginary parts through the staged-sample
    // area es | bsum | fix (1092 floats >= TR_WAVE), which is dead once pass 1 has read its samples -- a wave's LDS operations
    // execute in order, and the next group's samples are committed only after this pass.  (Round 2 sent the two planes through
    // the same buffer one after the other: two of the ~9 LDS round trips of a frame pass.)
    static_assert(ES_WAVE + BS_WAVE + FPW >= TR_WAVE, "the imaginary plane borrows the staged-sample area");
    float *trw = tr + g * TR_FRAME + i;                 // column i
    const float *trr = tr + g * TR_FRAME + i * TR_ROW;  // row i
    float *trw_i = es + g * TR_FRAME + i;
    const float *trr_i = es + g * TR_FRAME + i * TR_ROW;
    v2f R[8], I[8];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        trw[k1 * TR_ROW] = z[k1].x;
        trw_i[k1 * TR_ROW] = z[k1].y;
    }
    wave_fence();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        R[m] = v2f{trr[2 * m], trr[2 * m + 1]};
        I[m] = v2f{trr_i[2 * m], trr_i[2 * m + 1]};
    }
    wave_fence();
    LAD_FB_STAMP(6)
    // ---- pass 2: FFT over n2 (lane = k1 = i): pair 2c = (Z[i + 16c], Z[i + 16(c+8)]), pair 2c+1 = (Z[i + 16(c+4)], Z[i + 16(c+12)])
    fft16(R, I);

    LAD_FB_STAMP(7)
    // ---- real-FFT split.  Bin k = i + 16 j pairs with 256 - k = (16 - i) % 16 + 16 (15 - j): partner lane (16 - i) % 16.
    // A lane treats the pairs of its registers j in {c, c + 8 : c < 4} (the even pairs); the partners' values are the odd
    // pairs of the partner lane: j = c <-> 15 - c (high half of pair 2(3-c)+1), j = c + 8 <-> 7 - c (its low half).
    // Lane 0 pairs with itself, one register further: j <-> (16 - j) % 16.
    // One (S, T) per pair yields both bins: X[k] = (S + T) / 2, X[256 - k] = conj(S - T) / 2.
    const int src_addr = ((lane & 48) | ((16 - i) & 15)) << 2;
    float *pw = tr;  // power spectra of the wave's four frames: pw[bin][frame slot] (the transposition buffer is free now)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int og = 2 * (3 - cc) + 1;                    // generic: odd pair 3 - cc, halves swapped
        const int o0 = cc == 0 ? 0 : 2 * (4 - cc) + 1;      // lane 0: cc = 0 -> its own pair 0; else odd pair 4 - cc, swapped
        v2f sr, si;
        if (cc == 0) {
            sr = (i == 0) ? R[0] : v2f{R[og].y, R[og].x};
            si = (i == 0) ? I[0] : v2f{I[og].y, I[og].x};
        } else {
            sr = (i == 0) ? v2f{R[o0].y, R[o0].x} : v2f{R[og].y, R[og].x};
            si = (i == 0) ? v2f{I[o0].y, I[o0].x} : v2f{I[og].y, I[og].x};
        }
        const v2f pr = {bperm(src_addr, sr.x), bperm(src_addr, sr.y)};
        const v2f pi = {bperm(src_addr, si.x), bperm(src_addr, si.y)};
        const v2f Sr = R[2 * cc] + pr, Si = I[2 * cc] - pi;   // S = Z + conj(P)
        const v2f Dr = R[2 * cc] - pr, Di = I[2 * cc] + pi;   // D = Z - conj(P)
        const v4f w = tw2_s[cc * 16 + i];                     // W' = -i W512^k for k = i + 16 cc, i + 16 (cc + 8)
        const v2f Tr = Dr * w.xy - Di * w.zw, Ti = Dr * w.zw + Di * w.xy;
        const v2f Ar = Sr + Tr, Ai = Si + Ti, Br = Sr - Tr, Bi = Si - Ti;
        const v2f pk = Ar * Ar + Ai * Ai;   // 4 |X[k]|^2 for k = ka, ka + 128   (the 1/4 is folded into the mel weights)
        const v2f pp = Br * Br + Bi * Bi;   // 4 |X[256 - k]|^2
        const int ka = i + 16 * cc;
        pw[ka * 4 + g] = pk.x;
        pw[(ka + 128) * 4 + g] = pk.y;
        pw[(256 - ka) * 4 + g] = pp.x;
        pw[(128 - ka) * 4 + g] = pp.y;
    }
    {   // lane 0 only: its registers 4 and 12 (bins 64 and 192) pair with each other; W' = -i W512^64 = (-H, -H)
        constexpr float H = 0.70710678118654752f;
        const float Sr = R[1].x + R[1].y, Si = I[1].x - I[1].y, Dr = R[1].x - R[1].y, Di = I[1].x + I[1].y;
        const float Tr = H * (Di - Dr), Ti = -H * (Dr + Di);
        const float Ar = Sr + Tr, Ai = Si + Ti, Br = Sr - Tr, Bi = Si - Ti;
        if (i == 0) {
            pw[64 * 4 + g] = fmaf(Ar, Ar, Ai * Ai);
            pw[192 * 4 + g] = fmaf(Br, Br, Bi * Bi);
        }
    }
    wave_fence();

    LAD_FB_STAMP(8)
    // ---- mel: lane = one part of one filter, all four frames at once ----------------------------------------------------
    {
        const v4f *src = reinterpret_cast<const v4f *>(pw) + bin0;
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < LT; ++t) acc += src[t] * mw_s[t * 64 + lane];
        // part sums [64 lanes][4 frames] + one zero slot go where the power spectra were: every lane of the wave has executed
        // its last read of them (one instruction stream, LDS operations in order) before the first write below executes
        float *part = tr;
        *reinterpret_cast<v4f *>(part + lane * 4) = acc;
        if (lane == 0) *reinterpret_cast<v4f *>(part + 256) = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        wave_fence();
        v4f m = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r = 0; r < p.np_max; ++r) m += *reinterpret_cast<const v4f *>(part + (r < np ? (fp + r) * 4 : 256));
#pragma unroll
        for (int r = 0; r < FPW; ++r)  // v_log_f32 (log2, ~1 ulp, no denormal handling: the floor keeps the argument normal) times a constant
            held[r] = (p.log_mode != LAD_LOG_NONE) ? p.log_scale * __builtin_amdgcn_logf(fmaxf(m[r], p.log_floor)) : m[r];
        held_dst = G.out + lane;
        held_n = nfr;
    }
    }  // frame pass
    LAD_FB_STAMP(9)
    if (!has_next) {
        if (lane < p.n_mels) {
#pragma unroll
            for (int r = 0; r < FPW; ++r)
                if (r < held_n) held_dst[(int64_t)r * p.n_mels] = held[r];
        }
        break;
    }
    wave_fence();   // (compiler only) the next commit overwrites what this pass has read
    LAD_FB_STAMP(10)
    ++iter;
    gw = gw_next;
    G = G_next;
    }
}

template <typename T>
int upload(T **dst, const std::vector<T> &src) {
    LAD_HIP_CHECK(hipMalloc((void **)dst, src.size() * sizeof(T)));
    LAD_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LAD_OK;
}

}  // namespace

struct Fast {
    int lmax = 0, lt = 16, np_max = 0, nz = 16;
    int resident_wgs = 512;  // persistent grid: workgroups the device holds at once (CUs x 2, 8 waves each)
    v2f *d_win = nullptr;
    v2f *d_tw1 = nullptr;
    v4f *d_tw2 = nullptr;
    float *d_melw = nullptr;
    int *d_bin0 = nullptr, *d_fpart = nullptr, *d_fnp = nullptr;
};

void destroy(Fast *f) {
    if (!f) return;
    (void)hipFree(f->d_win);
    (void)hipFree(f->d_tw1);
    (void)hipFree(f->d_tw2);
    (void)hipFree(f->d_melw);
    (void)hipFree(f->d_bin0);
    (void)hipFree(f->d_fpart);
    (void)hipFree(f->d_fnp);
    delete f;
}

int build(const lad_fbank_cfg &cfg, const float *window, const float *melbank, Fast **out) {
    *out = nullptr;
    if (cfg.n_fft != NFFT || cfg.n_mfcc != 0 || cfg.hop % 16 != 0 || cfg.frame_len % 16 != 0 || cfg.n_mels > 64) return LAD_OK;
    if ((FPW - 1) * cfg.hop + NFFT > ES_WAVE) return LAD_OK;  // the span of four frames must fit a wave's staging area
    // banded filters -> parts of at most lmax taps, one part per lane
    std::vector<int> start(cfg.n_mels, 0), len(cfg.n_mels, 0);
    for (int m = 0; m < cfg.n_mels; ++m) {
        int lo = -1, hi = -1;
        for (int b = 0; b < NBINS; ++b)
            if (melbank[(size_t)b * cfg.n_mels + m] != 0.0f) {
                if (lo < 0) lo = b;
                hi = b;
            }
        if (lo >= 0) start[m] = lo, len[m] = hi - lo + 1;
    }
    int lmax = 0;
    for (int L = 2; L <= LMAX_CAP; L += 2) {
        int parts = 0;
        for (int m = 0; m < cfg.n_mels; ++m) parts += std::max(1, (len[m] + L - 1) / L);
        if (parts <= 64) {
            lmax = L;
            break;
        }
    }
    if (lmax == 0) return LAD_OK;  // too many / too long filters for one part per lane: general kernel
    Fast *f = new Fast();
    f->lmax = lmax;
    f->lt = lmax <= 16 ? 16 : 32;
    f->nz = (cfg.frame_len <= 13 * 32) ? 13 : 16;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            f->resident_wgs = prop.multiProcessorCount * 2;
    }
    std::vector<float> melw((size_t)f->lt * 64, 0.0f);
    std::vector<int> bin0(64, 0), fpart(64, 0), fnp(64, 0);
    int lane = 0;
    for (int m = 0; m < cfg.n_mels; ++m) {
        const int np = std::max(1, (len[m] + lmax - 1) / lmax);
        fpart[m] = lane;
        fnp[m] = np;
        f->np_max = std::max(f->np_max, np);
        for (int r = 0; r < np; ++r, ++lane) {
            // keep bin0 + lt - 1 <= 256: taps past the part carry zero weights but must read WRITTEN bins (0 * NaN = NaN)
            bin0[lane] = std::min(start[m] + r * lmax, NBINS - f->lt);
            for (int t = 0; t < f->lt; ++t) {
                const int b = bin0[lane] + t;
                if (b >= start[m] + r * lmax && b < start[m] + std::min(len[m], (r + 1) * lmax) && b < NBINS)
                    melw[(size_t)t * 64 + lane] = 0.25f * melbank[(size_t)b * cfg.n_mels + m];
            }
        }
    }
    std::vector<v2f> win(256);
    std::vector<v2f> tw1(256);
    std::vector<v4f> tw2(64);
    for (int n1 = 0; n1 < 16; ++n1)
        for (int i = 0; i < 16; ++i) win[n1 * 16 + i] = v2f{window[32 * n1 + 2 * i], window[32 * n1 + 2 * i + 1]};
    for (int k1 = 0; k1 < 16; ++k1)
        for (int i = 0; i < 16; ++i) {
            const double a = 2.0 * M_PI * (double)(i * k1) / 256.0;
            tw1[k1 * 16 + i] = v2f{(float)cos(a), (float)-sin(a)};
        }
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 16; ++i) {
            const double a = 2.0 * M_PI * (double)(i + 16 * c) / 512.0, b = 2.0 * M_PI * (double)(i + 16 * (c + 8)) / 512.0;
            tw2[c * 16 + i] = v4f{(float)-sin(a), (float)-sin(b), (float)-cos(a), (float)-cos(b)};  // -i exp(-i a)
        }
    int rc;
    if ((rc = upload(&f->d_win, win)) || (rc = upload(&f->d_tw1, tw1)) || (rc = upload(&f->d_tw2, tw2)) ||
        (rc = upload(&f->d_melw, melw)) || (rc = upload(&f->d_bin0, bin0)) || (rc = upload(&f->d_fpart, fpart)) ||
        (rc = upload(&f->d_fnp, fnp))) {
        destroy(f);
        return rc;
    }
    *out = f;
    return LAD_OK;
}

#ifdef LAD_STAMP
extern "C" int lad_debug_read_fbank_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_fb), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

bool eligible(const Fast *f, int64_t n_clips, int64_t samples_per_clip, const float *pcm) {
    // 16-byte loads of the PCM: every clip must start on a 16-byte boundary; 32-bit sample indices inside a clip
    return f != nullptr && samples_per_clip >= NFFT && (n_clips == 1 || samples_per_clip % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && samples_per_clip < ((int64_t)1 << 30);
}

int launch(const Fast *f, const lad_fbank_cfg &cfg, int left_off, const float *pcm, int64_t n_clips, int64_t samples_per_clip,
           int64_t T, float *out, hipStream_t stream) {
    using namespace lad;
    const int64_t groups = ceil_div(T, FPW);
    LAD_REQUIRE(n_clips * groups < (int64_t)1 << 31, "lad_fbank_forward: too many frame groups");
    LAD_REQUIRE(left_off % 4 == 0, "lad_fbank_forward: frame offset %d is not a multiple of 4", left_off);
    Params p;
    p.hop = cfg.hop;
    p.hop4 = cfg.hop / 4;
    p.hb = cfg.hop / 16;
    p.nblk = cfg.frame_len / 16;
    p.n_mels = cfg.n_mels;
    p.pad_mode = cfg.pad_mode;
    p.log_mode = cfg.log_mode;
    p.left_off = left_off;
    p.lmax = f->lmax;
    p.np_max = f->np_max;
    p.groups_per_clip = (int)groups;
    p.preemph = cfg.preemph;
    p.dc_scale = cfg.remove_dc ? (1.0f - cfg.preemph) / (float)cfg.frame_len : 0.0f;
    p.log_floor = std::max(cfg.log_floor, 1.1754944e-38f);  // keep v_log_f32's argument normal
    p.log_scale = cfg.log_mode == LAD_LOG_DB ? 3.0102999566398120f : 0.69314718055994531f;  // 10 log10(2) | ln 2
    p.n_samples = samples_per_clip;
    p.n_frames = T;
    p.win = f->d_win;
    p.tw1 = f->d_tw1;
    p.tw2 = f->d_tw2;
    p.melw = f->d_melw;
    p.part_bin0 = f->d_bin0;
    p.filt_part = f->d_fpart;
    p.filt_np = f->d_fnp;
    const size_t floats = 2 * 256 + 2 * 256 + 4 * 64 + (size_t)f->lt * 64 + (size_t)WAVES * WAVE_LDS;
    const size_t lds = floats * sizeof(float);
    const int64_t total = n_clips * groups;
    const int64_t wgs = ceil_div(total, WAVES);
    const dim3 grid((unsigned)std::min<int64_t>(wgs, (int64_t)f->resident_wgs)), block(THREADS);
    const unsigned tg = (unsigned)total;
    if (f->nz == 13 && f->lt == 16) hipLaunchKernelGGL((fbank16_kernel<13, 16>), grid, block, lds, stream, p, pcm, out, tg);
    else if (f->nz == 13) hipLaunchKernelGGL((fbank16_kernel<13, 32>), grid, block, lds, stream, p, pcm, out, tg);
    else if (f->lt == 16) hipLaunchKernelGGL((fbank16_kernel<16, 16>), grid, block, lds, stream, p, pcm, out, tg);
    else hipLaunchKernelGGL((fbank16_kernel<16, 32>), grid, block, lds, stream, p, pcm, out, tg);
    return check_launch("fbank16_kernel");
}

}  // namespace lad_fb16
