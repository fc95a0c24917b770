// fp16 inference path (eval mode only): activations and weights in IEEE half, f32 accumulation on the 16-bit matrix
// cores (v_mfma_f32_32x32x16_f16), every BatchNorm folded into the epilogue of the convolution in front of it.
//
// Replaces, for segment_laughter.py's window loop (segment_laughter.py:90-101 -> models.py:222-239 in eval mode), the
// same nn.Conv2d + nn.BatchNorm2d(+residual)+ReLU chains as conv_mfma.hip / bn.hip, at 16x the matrix rate: with
// 1.4 GFLOP per window the fp32 path is MFMA-bound (~77 k windows/s); in half the convolutions drop to a few hundred
// MFMA cycles per tile and the path becomes HBM-bound, so what matters here is bytes: half-width activations,
// 16-byte accesses everywhere, and enough workgroups per CU (3) to keep loads, MFMAs and stores of different tiles
// in flight together.
//
// Layout: shared-border PNHWC exactly as in lad_device.h, with _Float16 elements; zero border positions (invariant).
// Kernel structure = conv_s1_kernel of conv_mfma.hip (128 output rows x all channels per workgroup, input rows +halo
// staged in padded LDS rows, weights streamed one tap ahead through a two-slot LDS ring by LDS-DMA, accumulators
// transposed through LDS for whole-row stores); differences: a tile row is CIN halfs (all channels resident), a weight
// chunk is one tap = CIN x COUTP halfs packed [CIN/16][2][COUTP][8] so that a lane's 16-byte read is exactly its
// A[r][8h..8h+7] / B[8h..8h+7][r] fragment, and the epilogue writes 8-byte (4-channel) pieces of half rows.
#include "lad_common.h"
#include <type_traits>
#include "lad_device.h"
#include "lad_stem_taps.h"

namespace {
using namespace lad;

constexpr int TM = 128;
constexpr int THREADS = 256;

template <int COUT>
struct NTilesH {
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int COUTP = NT * 32;
};

// weight image (halfs): wt[tap][CIN/16][2][COUTP][8];  element (tap, s, h, co, j) = w[co][ci = 16 s + 8 h + j][tap]
__global__ void pack_f16_kernel(const float *__restrict__ w, _Float16 *__restrict__ wt, int cout, int cin, int taps) {
    const int NP = ((cout + 31) / 32) * 32;
    const int total = taps * cin * NP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7;
        int t = idx >> 3;
        const int co = t % NP;
        t /= NP;
        const int h = t & 1;
        t >>= 1;
        const int s = t % (cin / 16);
        const int tap = t / (cin / 16);
        const int ci = 16 * s + 8 * h + j;
        float v = 0.0f;
        if (co < cout) v = w[((int64_t)co * cin + ci) * taps + tap];
        wt[idx] = (_Float16)v;
    }
}

// ---- epilogue: acc -> (scale, shift, addend, relu, border mask) -> half rows -------------------------------------
template <int COUT>
__device__ __forceinline__ void epilogue_f16(f32x16 (&acc)[NTilesH<COUT>::NT], const float *__restrict__ scale,
                                             const float *__restrict__ shift, const _Float16 *__restrict__ addend,
                                             _Float16 *__restrict__ out, const float *mask_tile, float *out_s, int64_t q0,
                                             int64_t rows, int relu) {
    constexpr int NT = NTilesH<COUT>::NT;
    constexpr int LDO = COUT + 4;
    constexpr int LPR = COUT / 4;   // lanes per output row (4 channels = 8 bytes each)
    constexpr int RPI = 64 / LPR;
    constexpr int ITER = 32 / RPI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31;
    float *my = out_s + wave * 32 * LDO;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        if (co < COUT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, lane) * LDO + co] = acc[n][r];
        }
    }
    // (hipcc 7.2, COUT = 16, second of two calls in one kernel: the first ds_read below was emitted INSIDE the EXEC region of the
    // `co < COUT` stores above -- half the lanes kept the previous call's values.  A volatile asm after the region pins the order.)
    asm volatile("" ::: "memory");
    const int c4 = lane % LPR, rsub = lane / LPR;
    const float4 sv = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    const float4 bv = *reinterpret_cast<const float4 *>(shift + c4 * 4);
    f16x4 ad[ITER];
    bool ok[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int64_t q = q0 + wave * 32 + it * RPI + rsub;
        ok[it] = q < rows;
        ad[it] = f16x4{0, 0, 0, 0};
        if (addend != nullptr && ok[it]) ad[it] = *reinterpret_cast<const f16x4 *>(addend + q * COUT + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = it * RPI + rsub;
        const float keep = mask_tile[wave * 32 + row];
        float4 t = *reinterpret_cast<const float4 *>(my + row * LDO + c4 * 4);
        t.x = fmaf(t.x, sv.x, bv.x) + (float)ad[it][0];
        t.y = fmaf(t.y, sv.y, bv.y) + (float)ad[it][1];
        t.z = fmaf(t.z, sv.z, bv.z) + (float)ad[it][2];
        t.w = fmaf(t.w, sv.w, bv.w) + (float)ad[it][3];
        if (relu) {
            t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
        }
        if (keep == 0.0f) t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[it]) {
            const int64_t q = q0 + wave * 32 + row;
            const f16x4 o = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
            *reinterpret_cast<f16x4 *>(out + q * COUT + c4 * 4) = o;
        }
    }
}

// Stride-1 epilogue, branch-free (cf. s1_epilogue in conv_mfma.hip): tensors are reached through buffer resources whose
// range check replaces the per-row `q < rows` tests, the residual addend is a compile-time variant.
template <int COUT, bool ADD>
__device__ __forceinline__ void epilogue_f16_lean(f32x16 (&acc)[NTilesH<COUT>::NT], const float *__restrict__ scale,
                                                  const float *__restrict__ shift, const _Float16 *__restrict__ addend,
                                                  _Float16 *__restrict__ out, const float *mask_tile, float *out_s, int64_t q0,
                                                  int64_t rows, int relu) {
    constexpr int NT = NTilesH<COUT>::NT;
    constexpr int LDO = COUT + 4;
    constexpr int LPR = COUT / 4;   // lanes per output row (4 channels = 8 bytes each)
    constexpr int RPI = 64 / LPR;
    constexpr int ITER = 32 / RPI;
    constexpr int STEP = RPI * COUT * 2;  // bytes between the rows of consecutive iterations
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31;
    float *my = out_s + wave * 32 * LDO;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = n * 32 + i;
        if (co < COUT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) my[acc_row(r, lane) * LDO + co] = acc[n][r];
        }
    }
    asm volatile("" ::: "memory");   // (see epilogue_f16: the reads below stay outside the stores' EXEC region)
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int64_t tile_bytes = (rows - q0) * (COUT * 2);
    const int voff = ((wave * 32 + rsub) * COUT + c4 * 4) * 2;
    const __amdgpu_buffer_rsrc_t out_r = make_rsrc(out + q0 * COUT, tile_bytes);
    u32x2 ad[ITER];
    if (ADD) {
        const __amdgpu_buffer_rsrc_t add_r = make_rsrc(addend + q0 * COUT, tile_bytes);
#pragma unroll
        for (int it = 0; it < ITER; ++it) ad[it] = buf_load8(add_r, voff + it * STEP);
    }
    const float4 sv = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    const float4 bv = *reinterpret_cast<const float4 *>(shift + c4 * 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int row = it * RPI + rsub;
        const bool keep = mask_tile[wave * 32 + row] != 0.0f;
        float4 t = *reinterpret_cast<const float4 *>(my + row * LDO + c4 * 4);
        t.x = fmaf(t.x, sv.x, bv.x); t.y = fmaf(t.y, sv.y, bv.y); t.z = fmaf(t.z, sv.z, bv.z); t.w = fmaf(t.w, sv.w, bv.w);
        if (ADD) {
            const f16x4 av = __builtin_bit_cast(f16x4, ad[it]);
            t.x += (float)av[0]; t.y += (float)av[1]; t.z += (float)av[2]; t.w += (float)av[3];
        }
        t.x = relu ? fmaxf(t.x, 0.f) : t.x; t.y = relu ? fmaxf(t.y, 0.f) : t.y;
        t.z = relu ? fmaxf(t.z, 0.f) : t.z; t.w = relu ? fmaxf(t.w, 0.f) : t.w;
        t.x = keep ? t.x : 0.f; t.y = keep ? t.y : 0.f; t.z = keep ? t.z : 0.f; t.w = keep ? t.w : 0.f;
        const f16x4 o = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
        buf_store8(__builtin_bit_cast(u32x2, o), out_r, voff + it * STEP);
    }
}

template <int CIN, int COUT, int TAPS>
struct HCfg {
    static constexpr int COUTP = NTilesH<COUT>::COUTP;
    static constexpr int KS = CIN / 16;                      // MFMA k-steps per tap
    static constexpr int CHUNK_HALFS = CIN * COUTP;          // one tap: 8 KB at 64x64
    static constexpr int ROUNDS = (CHUNK_HALFS * 2 + THREADS * 16 - 1) / (THREADS * 16);
    static constexpr int LDA = CIN + 8;                      // halfs; (CIN+8)*2 bytes per row: conflict-free ds_read_b128
    static constexpr int A8 = CIN / 8;                       // 16-byte pieces per row
};

template <int CIN, int COUT, int TAPS, int NTHR = THREADS>
__device__ __forceinline__ void issue_tap(const _Float16 *__restrict__ wt, _Float16 *b_buf, int tap, int tid, int wave) {
    using C = HCfg<CIN, COUT, TAPS>;
    constexpr int ROUNDS = (C::CHUNK_HALFS * 2 + NTHR * 16 - 1) / (NTHR * 16);
    const _Float16 *src = wt + (int64_t)tap * C::CHUNK_HALFS;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if ((r * NTHR + wave * 64) * 8 < C::CHUNK_HALFS)  // wave-uniform
            dma16(src + (r * NTHR + tid) * 8, lds_addr(b_buf + (r * NTHR + wave * 64) * 8));
    }
}

constexpr int H_PRE = 8;  // 16-byte registers per thread for the stage-in (bounds the tile: nrows * CIN/8 <= 2048 per batch)

// WAVES = 4: 128-row workgroups, 3 per CU.  WAVES = 8: 256-row workgroups of 512 threads, 2 per CU -- a tile's time is
// mostly latency (9 weight chunks each waited for, barriers), so twice the rows per tile at the same lifetime is nearly
// twice the rows per second, with half the weight traffic from L2 and a smaller halo share.
template <int CIN, int COUT, int TAPS, bool ADD, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 2) void conv_f16_s1_kernel(const _Float16 *__restrict__ in,
                                                                 const _Float16 *__restrict__ wt,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 const _Float16 *__restrict__ addend,
                                                                 _Float16 *__restrict__ out, Geom g, int relu) {
    using C = HCfg<CIN, COUT, TAPS>;
    constexpr int NT = NTilesH<COUT>::NT;
    constexpr int COUTP = C::COUTP;
    constexpr int LDA = C::LDA;
    constexpr int A8 = C::A8;
    constexpr int NTHR = 64 * WAVES, TMV = 32 * WAVES;  // threads / rows per workgroup
    constexpr int PRE = WAVES == 4 ? H_PRE : 6;         // (256 + 2*46) rows x 8 pieces / 512 threads
    constexpr int SLOTS = 2;  // weight ring; a third slot (two chunks in flight) measured 834 vs 832 us: not the limiter
    constexpr int RPU = NTHR / A8;        // input rows one register (one 16-byte load per thread) covers
    constexpr int USTEP = RPU * CIN * 2;     // bytes between the rows of consecutive registers
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMV + 2 * halo;
    // LDS: [ weight ring (2 taps) | input rows ] re-used as the f32 output tile | row mask
    const int main_bytes = max(SLOTS * C::CHUNK_HALFS * 2 + nrows * LDA * 2, TMV * (COUT + 4) * 4);
    _Float16 *b_s = reinterpret_cast<_Float16 *>(smem);
    _Float16 *a_s = b_s + SLOTS * C::CHUNK_HALFS;
    float *mask_s = smem + (main_bytes + 3) / 4;
    const int64_t q0 = (int64_t)blockIdx.x * TMV;

    // With the matrix work of a tile down to a few thousand cycles, this kernel's time is its instruction count and
    // its memory latency: staging is straight-line (buffer-resource range checks instead of per-row bounds tests, a
    // division-free row mask), all loads of the tile are in flight together.
    issue_tap<CIN, COUT, TAPS, NTHR>(wt, b_s, 0, tid, wave);
    if (tid < TMV) mask_s[tid] = interior_row32((uint32_t)q0 + (uint32_t)tid, g) ? 1.0f : 0.0f;
    const int64_t start = q0 - halo;
    const int64_t first = start < 0 ? 0 : start;
    const int row_lo = (int)(first - start);
    // the resource ends with the tile's span (or the tensor): registers past the tile cost no memory traffic
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + first * CIN, min(g.rows - first, (int64_t)(nrows - row_lo)) * (CIN * 2));
    const int r0 = tid / A8, c8 = tid - r0 * A8;
    const int voff = ((r0 - row_lo) * CIN + c8 * 8) * 2;   // rows before the tensor: negative = out of range = 0
    _Float16 *lds0 = a_s + r0 * LDA + c8 * 8;
    _Float16 *dummy = a_s + r0 * LDA + CIN;                 // this row's padding: sink for registers past the tile
    u32x4 pre[PRE];
    for (int base = 0; base < nrows; base += PRE * RPU) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = buf_load16(in_r, voff + (base / RPU + u) * USTEP);
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int row = base + u * RPU + r0;
            *reinterpret_cast<u32x4 *>(row < nrows ? lds0 + (base + u * RPU) * LDA : dummy) = pre[u];
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    const int i = lane & 31, h = lane >> 5;
    const _Float16 *a_base = a_s + (wave * 32 + i + halo) * LDA + 8 * h;
    const int b_off = (h * COUTP + i) * 8;
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
        dma_wait_all();
        __syncthreads();
        if (tap + SLOTS - 1 < TAPS)
            issue_tap<CIN, COUT, TAPS, NTHR>(wt, b_s + ((tap + SLOTS - 1) % SLOTS) * C::CHUNK_HALFS, tap + SLOTS - 1, tid, wave);
        const int off = (TAPS == 9) ? ((tap / 3 - 1) * g.Wp + (tap % 3 - 1)) : 0;
        const _Float16 *ap = a_base + off * LDA;
        const _Float16 *bp = b_s + (tap % SLOTS) * C::CHUNK_HALFS + b_off;
        f16x8 av[C::KS], bv[C::KS][NT];  // the whole tap's fragments requested together, then its MFMAs
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            av[s] = *reinterpret_cast<const f16x8 *>(ap + s * 16);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[s][n] = *reinterpret_cast<const f16x8 *>(bp + (s * 2 * COUTP + n * 32) * 8);
        }
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32_f16(av[s], bv[s][n], acc[n]);
    }
    __syncthreads();
    epilogue_f16_lean<COUT, ADD>(acc, scale, shift, addend, out, mask_s, smem, q0, g.rows, relu);
}

// Persistent form for the 64->64 3x3 layers of large launches.  A tile's MFMAs take ~2,000 cycles but a weight chunk
// takes about as long to arrive from L2, so the ring above spends most of a tile's lifetime waiting for its nine chunks
// (PMC: waves issue VALU 9 % and LDS 4 % of their cycles, and wait 47 %).  Here the whole weight image of the layer
// (72 KB) is loaded into LDS ONCE per workgroup; a workgroup (8 waves, one per CU) then walks tiles of 256 rows: the next
// tile's input rows travel HBM -> registers while the current tile computes, the MFMA loop has no barrier and no wait on
// memory, and each wave writes its 32 x 64 outputs in two 16-row passes through a small private transposition buffer
// that overlays the (consumed) input rows.
#ifdef LAD_STAMP
// diagnostic build only (tools/stamp_f16.py): shader-clock time per phase, summed over a workgroup's tiles, of waves 0 and 7
__device__ unsigned long long lad_dbg_f16p[256 * 16];
#define LAD_F16P_T(k)                                          \
    {                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ph_[k] += now_ - last_;                                \
        last_ = now_;                                          \
    }
#else
#define LAD_F16P_T(k)
#endif

template <bool ADD>
__global__ __launch_bounds__(512, 2) void conv_f16_s1p_kernel(const _Float16 *__restrict__ in, const _Float16 *__restrict__ wt,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const _Float16 *__restrict__ addend, _Float16 *__restrict__ out,
                                                              Geom g, int relu, int n_tiles) {
    constexpr int CIN = 64, COUT = 64, TAPS = 9;
    using C = HCfg<CIN, COUT, TAPS>;
    constexpr int NT = 2, COUTP = 64, LDA = C::LDA, A8 = C::A8;
    constexpr int NTHR = 512, TMV = 256, PRE = 6;
    constexpr int RPU = NTHR / A8;           // 64 rows per register
    constexpr int USTEP = RPU * CIN * 2;
    constexpr int LDO = COUT + 4;            // floats per row of the transposition buffer
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = g.Wp + 1;
    const int nrows = TMV + 2 * halo;        // <= PRE * RPU = 384 (launcher)
    _Float16 *w_s = reinterpret_cast<_Float16 *>(smem);              // [9][CHUNK_HALFS]: 72 KB, resident
    _Float16 *a_s = w_s + TAPS * C::CHUNK_HALFS;                     // [PRE * RPU][LDA] halfs = 55 KB
    float *t_s = reinterpret_cast<float *>(a_s);                     // overlay: [8 waves][16][LDO] floats = 34.8 KB
    float *mask_s = reinterpret_cast<float *>(a_s + PRE * RPU * LDA);  // [TMV]

    // the layer's weights, once
#pragma unroll
    for (int t = 0; t < TAPS; ++t) issue_tap<CIN, COUT, TAPS, NTHR>(wt, w_s + t * C::CHUNK_HALFS, t, tid, wave);

    const int r0 = tid / A8, c8 = tid - r0 * A8;
    _Float16 *lds0 = a_s + r0 * LDA + c8 * 8;
    const int i = lane & 31, h = lane >> 5;
    const _Float16 *a_base = a_s + (wave * 32 + i + halo) * LDA + 8 * h;
    const _Float16 *b_base = w_s + (h * COUTP + i) * 8;
    float *my = t_s + wave * 16 * LDO;
    constexpr int LPR = COUT / 4, RPI = 64 / LPR, ITER = 16 / RPI;  // 16 lanes per row, 4 rows per instruction, 4 per pass
    const int c4 = lane % LPR, rsub = lane / LPR;
    const f32x4 sv = *reinterpret_cast<const f32x4 *>(scale + c4 * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(shift + c4 * 4);

    auto window = [&](int64_t q0, int &voff) {
        const int64_t start = q0 - halo;
        const int64_t first = start < 0 ? 0 : start;
        const int row_lo = (int)(first - start);
        voff = ((r0 - row_lo) * CIN + c8 * 8) * 2;
        return make_rsrc(in + first * CIN, min(g.rows - first, (int64_t)(nrows - row_lo)) * (CIN * 2));
    };
    // Tile order: workgroups are dealt to the 8 XCDs round-robin (workgroup w runs on XCD w % 8), and neighbouring tiles share
    // their halo rows (2 x (W + 2) of 256 + 2 x (W + 2)).  With tile = w, w + grid, ... neighbours always sit on different XCDs
    // and every halo row is fetched from HBM twice (PMC: 3.35 GB per launch for 2.4 GB of tensors).  Here XCD x owns the
    // contiguous range [x * per_x, (x + 1) * per_x) and its grid / 8 workgroups walk it side by side, so a halo row is
    // fetched once into that XCD's L2.
    const int nx = (gridDim.x % 8 == 0) ? 8 : 1;
    const int per_x = (n_tiles + nx - 1) / nx;
    const int tile_hi = min(n_tiles, ((int)blockIdx.x % nx + 1) * per_x);
    const int tile_step = (int)gridDim.x / nx;
    int tile = ((int)blockIdx.x % nx) * per_x + (int)blockIdx.x / nx;
    n_tiles = tile_hi;   // (this workgroup's range ends here)
    u32x4 pre[PRE];
    {
        int voff;
        const __amdgpu_buffer_rsrc_t in_r = window((int64_t)tile * TMV, voff);
#pragma unroll
        for (int u = 0; u < PRE; ++u) pre[u] = buf_load16(in_r, voff + u * USTEP);
    }
#ifdef LAD_STAMP
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t q0 = (int64_t)tile * TMV;
        __syncthreads();  // the previous tile's readers of the rows / the transposition overlay are done
        LAD_F16P_T(0)
#pragma unroll
        for (int u = 0; u < PRE; ++u) *reinterpret_cast<u32x4 *>(lds0 + u * RPU * LDA) = pre[u];
        if (tid < TMV) mask_s[tid] = interior_row32((uint32_t)q0 + (uint32_t)tid, g) ? 1.0f : 0.0f;
        dma_wait_all();   // (first tile) the weights have landed
        __syncthreads();
        LAD_F16P_T(1)
        // The next tile's rows and this tile's residual rows are requested INSIDE the MFMA loop, a couple per tap.  Requested in
        // one burst in front of the loop they cost 930 cycles of a 9,450-cycle tile (2,800 of 11,200 with the residual:
        // profiles/r03_conv_f16_stamps.log) -- the vector-memory queue is full, a wave sits in the issue of its loads, and both
        // waves of a SIMD do so at the same time.  Spread over the taps, one wave's blocked issue is the other one's MFMA time.
        const int next = tile + tile_step;
        int voff_n = 0;
        const __amdgpu_buffer_rsrc_t nxt_r = next < n_tiles ? window((int64_t)next * TMV, voff_n) : make_rsrc(in, 0);   // (no next tile: every load out of range)
        const int64_t tile_bytes = (g.rows - q0) * (COUT * 2);
        const __amdgpu_buffer_rsrc_t add_r = make_rsrc(ADD ? addend + q0 * COUT : out + q0 * COUT, tile_bytes);
        u32x2 ad[2][ITER];
        static_assert(PRE <= TAPS && 2 * ITER <= TAPS, "one row load and one residual load per tap");
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        LAD_F16P_T(2)
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap < PRE) pre[tap] = buf_load16(nxt_r, voff_n + tap * USTEP);
            if (ADD && tap < 2 * ITER)
                ad[tap / ITER][tap % ITER] = buf_load8(add_r, ((wave * 32 + 16 * (tap / ITER) + rsub + (tap % ITER) * RPI) * COUT + c4 * 4) * 2);
            __builtin_amdgcn_sched_barrier(0);
            const int off = (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
            const _Float16 *ap = a_base + off * LDA;
            const _Float16 *bp = b_base + tap * C::CHUNK_HALFS;
            f16x8 av[C::KS], bw[C::KS][NT];
#pragma unroll
            for (int s2 = 0; s2 < C::KS; ++s2) {
                av[s2] = *reinterpret_cast<const f16x8 *>(ap + s2 * 16);
#pragma unroll
                for (int n = 0; n < NT; ++n) bw[s2][n] = *reinterpret_cast<const f16x8 *>(bp + (s2 * 2 * COUTP + n * 32) * 8);
            }
#pragma unroll
            for (int s2 = 0; s2 < C::KS; ++s2)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n] = mfma32_f16(av[s2], bw[s2][n], acc[n]);
        }
        LAD_F16P_T(3)
        __syncthreads();  // every wave is done with the input rows: they become the transposition buffers
        LAD_F16P_T(4)
        const __amdgpu_buffer_rsrc_t out_r = make_rsrc(out + q0 * COUT, tile_bytes);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            // accumulator registers 8*pass .. 8*pass+7 of a lane are rows 16*pass + {0..3, 8..11} (+4 for the upper half-wave)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 8; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * h) * LDO + n * 32 + i] = acc[n][8 * pass + r];
            const int voff_o = ((wave * 32 + 16 * pass + rsub) * COUT + c4 * 4) * 2;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int row = it * RPI + rsub;
                const float keep = mask_s[wave * 32 + 16 * pass + row];
                f32x4 t = __builtin_elementwise_fma(*reinterpret_cast<const f32x4 *>(my + row * LDO + c4 * 4), sv, bv);
                if (ADD) {
                    const f16x4 a4 = __builtin_bit_cast(f16x4, ad[pass][it]);
                    t += f32x4{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
                }
                if (relu) t = __builtin_elementwise_max(t, f32x4{0.f, 0.f, 0.f, 0.f});
                t = t * keep;
                const f16x4 o = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
                buf_store8(__builtin_bit_cast(u32x2, o), out_r, voff_o + it * RPI * COUT * 2);
            }
        }
        LAD_F16P_T(5)
    }
#ifdef LAD_STAMP
    if ((wave == 0 || wave == 7) && lane == 0 && blockIdx.x < 256)
        for (int j = 0; j < 6; ++j) lad_dbg_f16p[blockIdx.x * 16 + (wave ? 8 : 0) + j] = ph_[j];
#endif
}

// (Round 3 measured two restructurings of this kernel and kept neither -- profiles/r03_conv_f16_variants_ab.log, commits fa2a22c
// and the one after it: 512-row tiles with 64 x 64 wave tiles (a quarter fewer LDS reads) and one 12-wave workgroup whose two
// halves alternate between the MFMA loop and the epilogue.  Both gave identical results in the same time: the launches move
// their 1.05-1.57 GB of tensors at 3.2-4.6 TB/s, and at 192-288 FLOP/B against a ridge of 312 that -- HBM -- is the roof.)

// ---- one residual block (two 64 -> 64 3x3 convolutions + identity shortcut) on SMALL IMAGES, fused (round 5) ---------------------
// The sliding-window path (engine._forward_eval_stream) runs block1 on one 10-row "strip" image per frame offset: 8,282 images of
// 11 x 45 = 495 positions per group of 8,192 windows, 4 launches of conv_f16_s1p_kernel that move the 525 MB strip tensor ten times
// (profiles/r04_infer16_60min_kernel_stats.csv: 40 % of the path).  A whole strip fits the LDS of a CU -- 495 rows x 128 B = 62 KB --
// so here ONE workgroup keeps a strip on chip through the block: conv1 + BatchNorm + ReLU into a second LDS image, conv2 + BatchNorm +
// the residual (still in LDS) + ReLU in place, one read and one write of HBM per strip instead of five.  Same instructions in the same
// order per output element as conv_f16_s1_kernel (taps 0..8, four k-steps each, v_mfma_f32_32x32x16_f16; the same epilogue
// arithmetic; the intermediate is rounded to half exactly where that kernel rounds its output): bit-identical results
// (tests/test_resnet_gpu.py compare the streaming path with the per-window one by torch.equal).
//
// Matrix roles are swapped against conv_f16_s1_kernel -- D[channel][position] = W^T-fragment x X-fragment, the same two ds_read_b128
// per lane -- so that a lane ends up with FOUR CONSECUTIVE CHANNELS of one position per accumulator quad: epilogues are
// ds_write_b64 / ds_read_b64 (16 per wave) instead of 64 two-byte accesses.  (Products commute exactly and every element's k-order
// is the same: the bits are.)
//
// LDS (128-byte rows, 16-byte slots XOR-ed with (row >> 1) & 7: the 16-lane groups of a ds_read_b128 -- lanes {0-3, 12-15, 20-27},
// ... of consecutive rows, MI355X_MICROARCH.md "LDS" -- land on 16 different slots of the 256-byte bank line, for even and odd tap
// offsets alike; a padded row as conv_f16_s1_kernel's does not fit twice):
//   [weight ring: 3 taps x 8 KB][P: IMG rows][Q: IMG rows][zeros: up to row 512 + W + 2 of Q][row mask][BatchNorm coefficients]
// Even images of a workgroup have their input in Q and the intermediate in P, odd ones the other way round: an image's input
// arrives by LDS-DMA (the swizzle applied on the global side: a lane's source address is free) in the buffer its predecessor's
// intermediate has just left, while the predecessor's output -- written in place over ITS input -- leaves for HBM during the first
// convolution, two pieces per tap.  (A first version staged through registers and one buffer: 64 VGPRs, 63 KB of ds_write_b128
// per image at 79 B/clk and the store, all with the matrix pipe idle -- 0.11 of 0.71 ms, profiles/r05_strip_block.log.)
// What a buffer's last image row reads below itself is the next buffer's first W + 2 rows: border positions, zero in every
// tensor (P -> Q: the input's or the previous output's border row; Q -> the zero region).  Rows above an image (its border row's
// upper taps) are whatever lies there, always finite halves: they only reach outputs at border positions, which are written as
// zero by SELECT.  Tiles are 512 rows (8 waves x 64); rows past IMG are computed and dropped.
//
// Schedule: a wave's fragments of tap t + 1 are read k-step by k-step INTO THE REGISTERS tap t's k-step has just been issued from
// (a first version read 16 fragments, then issued 16 MFMAs, per tap and behind the tap's barrier: all 8 waves in the LDS phase,
// then all in the MFMA phase, 2,600 cycles per tap for 1,024 of MFMA).  The weight ring is three deep for that: tap t + 2 is
// written while t + 1 is read and t multiplied.  A tap's 8 KB of weights travel L2 -> registers (waves 0-3, 32 bytes per thread) FOUR taps ahead
// and registers -> ring TWO taps ahead, so no barrier ever waits for memory: as LDS-DMA issued two taps ahead (all the ring allows
// with the read-ahead) every tap waited out the rest of the ~1,500-cycle DMA latency -- MFMA pipe busy 45 % of cycles at 2.08 GHz
// (profiles/r05_block_f16_pmc.json before / after).  Waves 4-7 also issue the output's stores; nobody waits for those.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int BLK_THREADS = 512, BLK_ROWB = 128, BLK_NSLOT = 3, BLK_TAPB = 64 * 64 * 2;
static inline int blk_zero_rows(int img, int Wp) { return 512 + Wp + 1 - img; }   // what the dropped rows of the 512-row tile read
static inline size_t blk_lds_bytes(int img, int Wp) {
    return (size_t)BLK_NSLOT * BLK_TAPB + (2 * (size_t)img + blk_zero_rows(img, Wp)) * BLK_ROWB + 512 + 4 * 64 * 4;
}
__device__ __forceinline__ unsigned blk_off(int row, int col8) {   // byte offset of the 16-byte slot `col8` (8 channels) of row `row`
    return (unsigned)row * BLK_ROWB + (unsigned)((col8 ^ ((row >> 1) & 7)) << 4);
}
// MAPPED (round 6): the input image is not a tensor of its own.  For the FIRST block of the sliding-window strips the input is the stem's
// output, and rows 1 .. H - 2 of strip s are rows s + 1 .. s + H - 2 of the stem run over the frame stream (their three input frames lie
// inside the strip either way); only its first and last row see the strip's own zero padding.  So stage_in takes the inner rows from the
// stream's stem output (a DMA lane's source address is its own; L2-resident: consecutive strips share all but one of those rows) and the
// workgroup computes the two edge rows itself from the features (2 x W positions x 64 channels x 9 taps, stem_f16_kernel's arithmetic; a
// lane per column -- W <= 64: launcher -- and two channel quads per wave: edge_load / edge_store below) straight into the LDS image.  The strips' stem was a launch of 133 us per group of 8,192 windows and a
// 525 MB tensor written and read once: both gone.  Same values -> same bits.
struct StripMap {
    const _Float16 *xs;             // the stream's stem output (one image)
    long long srow0;                // strip s, padded row yp = the stream's padded row srow0 + s + yp
    unsigned wp_magic;              // floor(2^32 / Wp) + 1
    const float *feat;              // frame 0 of strip 0: (frames, W) float32 features
    long long frames_avail;         // frames from there to the end of the file (zeros behind it)
    const float *stem_w, *stem_sc, *stem_sh;   // conv1.weight (64, 1, 3, 3), bn1 folded
};
template <bool MAPPED>
__global__ __launch_bounds__(BLK_THREADS, 2) void block_f16_strip_kernel(const _Float16 *__restrict__ x, _Float16 *__restrict__ y,
                                                                       const _Float16 *__restrict__ wt1, const float *__restrict__ sc1,
                                                                       const float *__restrict__ sh1, const _Float16 *__restrict__ wt2,
                                                                       const float *__restrict__ sc2, const float *__restrict__ sh2,
                                                                       int n_img, int Hp, int Wp, StripMap sm) {
    constexpr int CIN = 64, COUT = 64, TAPS = 9, KS = 4, NT = 2, COUTP = 64;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int IMG = Hp * Wp;                                   // positions of one image (<= 512: launcher)
    const int zrows = 512 + Wp + 1 - IMG;                      // (blk_zero_rows)
    const int n_piece = IMG * 8, n_chunk = (n_piece + 63) >> 6;   // 16-byte pieces / 1 KB chunks (one DMA instruction of a wave) of an image
    unsigned char *ring = lds_b;                               // [3][8 KB]
    unsigned char *p_s = ring + BLK_NSLOT * BLK_TAPB;          // [IMG][128]
    unsigned char *q_s = p_s + IMG * BLK_ROWB;                 // [IMG][128]
    unsigned char *z_s = q_s + IMG * BLK_ROWB;                 // [zrows][128] zeros
    unsigned char *mask_s = z_s + zrows * BLK_ROWB;            // [512]
    float *coef_s = reinterpret_cast<float *>(mask_s + 512);   // scale1 | shift1 | scale2 | shift2
    for (int j = tid; j < zrows * (BLK_ROWB / 16); j += BLK_THREADS) reinterpret_cast<u32x4 *>(z_s)[j] = u32x4{0u, 0u, 0u, 0u};
    {
        const int yp = tid / Wp, xp = tid - yp * Wp;
        mask_s[tid] = (tid < IMG && yp >= 1 && xp >= 1) ? 1 : 0;
        if (tid < 256) coef_s[tid] = (tid < 64 ? sc1 : tid < 128 ? sh1 : tid < 192 ? sc2 : sh2)[tid & 63];
    }
    const int n_mine = blockIdx.x < (unsigned)n_img ? (n_img - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (n_mine == 0) return;
    int grid = (int)gridDim.x;
    asm volatile("" : "+s"(grid));   // (kept in a register: see y_prev)
    // weights of tap g (0..8: conv1, 9..17: conv2, 18..21: the next image's first four), 32 bytes per thread of waves 0-3: L2 -> registers
    // FOUR taps ahead, registers -> ring slot g % 3 two taps ahead (see the schedule note).  Loads, stores and their waits are inline
    // asm: with builtins the compiler's wait-count pass merges the two exclusive roles (waves 0-3 load weights, waves 4-7 store the
    // output from the same registers) into vmcnt(0) / vmcnt(1) in front of nearly every tap.
    struct W2 { u32x4 a, b; };
    auto w_load = [&](int g, W2 &w) {
        const int t = g >= 2 * TAPS ? g - 2 * TAPS : g;
        const _Float16 *base = t < TAPS ? wt1 : wt2;
        const unsigned o = (unsigned)((t < TAPS ? t : t - TAPS) * BLK_TAPB + tid * 16);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w.a) : "v"(o), "s"(base) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(w.b) : "v"(o + BLK_TAPB / 2), "s"(base) : "memory");   // (immediate offsets end at 4095)
    };
    auto w_put = [&](int slot, const W2 &w) {
        *reinterpret_cast<u32x4 *>(ring + slot * BLK_TAPB + tid * 16) = w.a;
        *reinterpret_cast<u32x4 *>(ring + slot * BLK_TAPB + BLK_TAPB / 2 + tid * 16) = w.b;
    };
    // image `im` -> buffer `dst`, all waves: chunk c of the LDS image = pieces 64 c .. 64 c + 63 in LDS order; piece (row, slot') holds
    // the row's channels 8 (slot' ^ ((row >> 1) & 7)) ..
    auto stage_in = [&](int im, unsigned char *dst) {
        if (!MAPPED) {
            const _Float16 *src = x + (int64_t)im * IMG * CIN;
            for (int c = wave; c < n_chunk; c += 8) {
                const int pc = c * 64 + lane, row = pc >> 3;
                if (pc < n_piece) dma16(src + row * CIN + (((pc & 7) ^ ((row >> 1) & 7)) << 3), lds_addr(dst + c * 1024));
            }
        } else {
            const _Float16 *strm = sm.xs + (sm.srow0 + im) * Wp * CIN;
            for (int c = wave; c < n_chunk; c += 8) {
                const int pc = c * 64 + lane, row = pc >> 3;
                if (pc < n_piece) {
                    const int yp = (int)__umulhi((unsigned)row, sm.wp_magic), xp = row - yp * Wp;
                    const bool edge = yp == 1 || yp == Hp - 1;   // (computed below; its border position comes from the zeros)
                    const _Float16 *src = (yp == 0 || edge) ? sm.xs + (yp == 0 ? xp : 0) * CIN   // the stream image's border row: zeros
                                                             : strm + (yp * Wp + xp) * CIN;      // the rows that are the stream's
                    if (!edge || xp == 0) dma16(src + (((pc & 7) ^ ((row >> 1) & 7)) << 3), lds_addr(dst + c * 1024));
                }
            }
        }
    };
    // (MAPPED) the strip's first and last row: stem_f16_kernel's arithmetic -- acc = fmaf(tap, w, acc) over the nine taps in order, zeros
    // where the strip / the frame / the file ends, then relu(fmaf(acc, scale, shift)) rounded to half -- written into the LDS image.
    // Lane l < W of EVERY wave: column l of both rows; wave w: channel quads w and w + 8, so a quad's 36 weights and its folds are
    // wave-uniform and come through the scalar cache (s_load), not through LDS.  The lane's own column of each row's two frames inside the
    // strip (the third kernel row is the strip's zero padding) is REQUESTED by edge_load a tap early; edge_store gets the two neighbouring
    // columns from the neighbouring lanes.  (Earlier versions, all bit-identical: per-thread (position, 4 quads) items with the weights in
    // an LDS table -- 44 serialised LDS round trips per image under the tap loop's register pressure, +115 us per launch; the taps
    // requested BEHIND the image's DMA -- the compiler's wait for them, which cannot see the DMA instructions, waits for the image.)
    typedef const __attribute__((address_space(4))) float cfloat;
    float ev[4] = {0.f, 0.f, 0.f, 0.f};   // [row][frame]
    auto edge_load = [&](int im) {
        const int W = Wp - 1;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));   // (addresses formed HERE, not kept in registers through the image's taps)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long f = (long long)im + (k >> 1) * (Hp - 3) + (k & 1);   // (relative to sm.feat) top row: the strip's frames 0, 1; bottom: H - 2, H - 1
            ev[k] = (lane < W && f < sm.frames_avail) ? sm.feat[f * W + lane] : 0.0f;
        }
    };
    auto edge_store = [&](unsigned char *dst) {
        const int W = Wp - 1;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        float lf[4], rt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = __builtin_bit_cast(int, ev[k]);
            const float l_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, b));
            const float r_ = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, b));
            lf[k] = lane > 0 ? l_ : 0.0f;
            rt[k] = lane < W - 1 ? r_ : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cq = wave + 8 * q;   // (wave-uniform)
            cfloat *wq = (cfloat *)(sm.stem_w + 36 * cq), *scq = (cfloat *)(sm.stem_sc + 4 * cq), *shq = (cfloat *)(sm.stem_sh + 4 * cq);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                // top row: kernel rows 1, 2 are the strip's frames 0, 1; bottom row: kernel rows 0, 1 are its frames H - 2, H - 1
                const float z = 0.0f;
                const float v[9] = {r ? lf[2] : z, r ? ev[2] : z, r ? rt[2] : z, r ? lf[3] : lf[0], r ? ev[3] : ev[0], r ? rt[3] : rt[0],
                                    r ? z : lf[1], r ? z : ev[1], r ? z : rt[1]};
                float o4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(v[t], wq[c * 9 + t], acc);
                    o4[c] = fmaxf(fmaf(acc, scq[c], shq[c]), 0.f);
                }
                const f16x4 o = {(_Float16)o4[0], (_Float16)o4[1], (_Float16)o4[2], (_Float16)o4[3]};
                const int row = (r ? Hp - 1 : 1) * Wp + lane + 1;
                if (lane < W) *reinterpret_cast<f16x4 *>(dst + blk_off(row, cq >> 1) + (cq & 1) * 8) = o;
            }
        }
    };
    const int lt = tid - 256;   // waves 4-7: thread lt of 256 moves pieces lt, lt + 256, ... of the output
    int img = (int)blockIdx.x;
    stage_in(img, q_s);
    if (MAPPED) {
        edge_load(img);
        edge_store(q_s);
    }
    W2 wreg[3];   // waves 0-3: tap t's weights wait in wreg[t % 3] from tap t - 4 to tap t - 2; waves 4-7: output pieces on their way out
#pragma unroll
    for (int k = 0; k < 3; ++k) wreg[k].a = wreg[k].b = u32x4{0u, 0u, 0u, 0u};
    if (wave < 4) {
        W2 w0, w1;
        w_load(0, w0);
        w_load(1, w1);
        w_load(2, wreg[2]);
        w_load(3, wreg[0]);
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(w0.a), "+v"(w0.b), "+v"(w1.a), "+v"(w1.b)::"memory");
        w_put(0, w0);
        w_put(1, w1);
    }
    unsigned char *out_prev = q_s;   // (where the previous image's output lies; unused for the first image)
#ifdef LAD_STAMP
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();   // (tools/stamp_block.py)
#endif
    for (int it = 0; it < n_mine; ++it, img += grid) {
        const bool more = it + 1 < n_mine;
        unsigned char *x_s = (it & 1) ? p_s : q_s, *a1_s = (it & 1) ? q_s : p_s;
        // where the previous image's output goes.  Formed HERE and made opaque: inside the taps the compiler re-read gridDim.x with an
        // s_load + s_waitcnt lgkmcnt(0) -- which also waits for the sixteen fragment reads just issued: 820 cycles per tap of waves 4-7
        const _Float16 *y_prev = y + (int64_t)(img - grid) * IMG * COUT;
        asm volatile("" : "+s"(y_prev));
        // (ring slots of this image's taps: g = 18 * it + ..., and 18 % 3 == 0)
#pragma unroll 1
        for (int conv = 0; conv < 2; ++conv) {
            const unsigned char *src_s = conv == 0 ? x_s : a1_s;
            int ll = lane;
            asm volatile("" : "+v"(ll));   // addresses are recomputed per convolution, not hoisted out of the image loop into 100+ registers
            const int i = ll & 31, h = ll >> 5;
            const unsigned w_lane = (unsigned)((h * COUTP + i) * 16);
            const int row0 = wave * 64 + i;
            f16x8 xf[2][KS], wf[KS][NT];
            auto read_k = [&](int tap, int ks) {   // the fragments of k-step ks of tap `tap` of this convolution
                const int off = (tap / 3 - 1) * Wp + (tap % 3 - 1);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = row0 + rt * 32 + off;   // (may be negative or past the image: see the layout note)
                    xf[rt][ks] = *reinterpret_cast<const f16x8 *>(src_s + (int)blk_off(row, ks * 2 + h));
                }
                const unsigned char *wp = ring + ((conv * TAPS + tap) % BLK_NSLOT) * BLK_TAPB + w_lane;
#pragma unroll
                for (int n = 0; n < NT; ++n) wf[ks][n] = *reinterpret_cast<const f16x8 *>(wp + (ks * 2 * COUTP + n * 32) * 16);
            };
            f32x16 acc[NT][2];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][rt][r] = 0.0f;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                // every wave is past the previous tap (tap 0: past the epilogue that wrote the image this convolution reads; before conv1 the
                // image's DMA has landed); the weights written in the previous tap are in LDS
                if (tap > 0) {   // a tap's body (its MFMAs issued, the next tap's fragments requested): conv1 / conv2
                    if (conv == 0) { LAD_F16P_T(3) } else { LAD_F16P_T(5) }
                }
                if (conv == 0 && tap == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const bool last = conv == 1 && tap == TAPS - 1;   // behind this barrier nobody reads the intermediate any more
                if (tap == 0 || last) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(11)\n\ts_barrier" ::: "memory");   // (the previous tap's weight store is older than 12 later
                                                                                       // fragment reads: it has landed; the last reads stay in flight)
                if (tap == 0) { LAD_F16P_T(0) } else { LAD_F16P_T(2) }   // waiting at a convolution's first barrier / at a tap barrier
                const bool storing = conv == 0 && wave >= 4 && tap < 8, storing_prev = conv == 0 && wave >= 4 && tap >= 1;
                if (tap == 0) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) read_k(0, ks);
#ifdef LAD_STAMP
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    LAD_F16P_T(1)   // the exposed fragment reads of a convolution's first tap
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) acc[n][rt] = mfma32_f16(wf[ks][n], xf[rt][ks], acc[n][rt]);
                    if (tap + 1 < TAPS) read_k(tap + 1, ks);
                    if (ks == 0) {
                        // the tap's housekeeping sits BEHIND its first MFMAs (in front of them, right after the barrier, it was ~150 cycles of
                        // an idle matrix pipe per tap)
                        {
                            // ring slot (tap + 2) % 3 held tap t - 1: multiplied before the barrier above.  Its new content is read behind the NEXT barrier.
                            const int gw = conv * TAPS + tap + 2;
                            if (wave < 4) {   // in flight: tap t + 2's pair, then tap t + 3's (past the last image too: the waits count on them)
                                W2 &w = wreg[(tap + 2) % 3];
                                asm volatile("s_waitcnt vmcnt(2)" : "+v"(w.a), "+v"(w.b)::"memory");
                                w_put((tap + 2) % BLK_NSLOT, w);
                                w_load(gw + 2, wreg[(tap + 4) % 3]);
                            }
                        }
                        // (MAPPED) the next image's edge rows: their taps are requested one tap early and used HERE, in front of the image's DMA --
                        // behind it, the compiler's wait for them (it does not see the DMA instructions) is a vmcnt(0) that waits for the whole
                        // image: 3,900 cycles per image, measured
                        if (MAPPED && conv == 1 && tap == TAPS - 2 && more) edge_load(img + grid);
                        if (last && more) {
                            if (MAPPED) asm volatile("" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]));   // (the compiler's wait for the taps: HERE, in front of the DMA)
                            stage_in(img + grid, a1_s);
                            if (MAPPED) edge_store(a1_s);
                        }
                        // the previous image's output, two pieces per tap of conv1: LDS -> registers here, registers -> HBM one tap later.
                        // The registers are the weight registers, which waves 4-7 do not use (three sets in rotation)
                        if (storing_prev && it > 0) {   // the pieces read ONE TAP AGO leave here, behind this tap's first MFMAs (issued at the end of
                            // their own tap, in front of the barrier, the two stores held everybody up by ~400 cycles: tools/exp_blk.sh NOSTORE)
                            const int pa = (2 * (tap - 1)) * 256 + lt;
                            const W2 &wo = wreg[(tap + 2) % 3];   // = (tap - 1) % 3
                            if (pa < n_piece) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"((unsigned)pa * 16), "v"(wo.a), "s"(y_prev) : "memory");
                            if (pa + 256 < n_piece)
                                asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"((unsigned)(pa + 256) * 16), "v"(wo.b), "s"(y_prev) : "memory");
                        }
                        if (storing) {
                            const int pa = (2 * tap) * 256 + lt, pb = pa + 256;   // (past the image: in LDS, not stored)
                            W2 &w = wreg[tap % 3];   // (a store reads its data registers when it issues -- no wait before they are loaded again: a
                            // vmcnt(4) here, "the stores of three taps ago are done", cost these waves 880 cycles per tap: profiles/r05_strip_block.log)
                            w.a = *reinterpret_cast<const u32x4 *>(out_prev + blk_off(pa >> 3, pa & 7));
                            w.b = *reinterpret_cast<const u32x4 *>(out_prev + blk_off(pb >> 3, pb & 7));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (conv == 0) { LAD_F16P_T(3) } else { LAD_F16P_T(5) }
            // epilogue: register 4 q + j of lane (i, h) of tile (n, rt) is channel n * 32 + 8 q + 4 h + j of position rt * 32 + i.
            // ReLU and the border mask act on the PACKED halves (v_pk_max_f16, v_and_b32: one instruction per two elements instead of
            // three -- rounding to half is monotonic and keeps the sign, so max(half(t), 0) is half(max(t, 0)) bit for bit).
            const float *cf = coef_s + conv * 128 + 4 * h;
            unsigned char *dst_s = conv == 0 ? a1_s : x_s;
            unsigned keep[2];
            unsigned char *prow[2];
            int sw[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int row = row0 + rt * 32;
                keep[rt] = mask_s[row] ? 0xffffffffu : 0u;
                // rows past the image (the tile has 512) go to the zero region's rows W + 2 .., which only dropped rows ever read: no
                // branch in the epilogue (an EXEC region per quad kept the compiler from moving the coefficient reads ahead)
                prow[rt] = row < IMG ? dst_s + row * BLK_ROWB : z_s + (row - IMG + Wp + 1) * BLK_ROWB;
                sw[rt] = (row >> 1) & 7;
            }
            auto epilogue = [&](auto RES) {   // RES: + the block's input (conv2), overwritten in place by the thread that read it
                // Half the channels at a time: first ALL their LDS reads (4 coefficient pairs, 8 residual pieces), then the arithmetic -- left
                // to itself at 250 registers the compiler read, waited and computed quad by quad: sixteen exposed LDS round trips per epilogue.
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    f32x4 sv[4], bv[4];
                    u32x2 a4[4][2];
                    unsigned char *pa[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sv[q] = *reinterpret_cast<const f32x4 *>(cf + n * 32 + 8 * q);
                        bv[q] = *reinterpret_cast<const f32x4 *>(cf + 64 + n * 32 + 8 * q);
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            pa[q][rt] = prow[rt] + (((n * 4 + q) ^ sw[rt]) << 4) + h * 8;
                            if (decltype(RES)::value) a4[q][rt] = *reinterpret_cast<const u32x2 *>(pa[q][rt]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            f32x4 t = {acc[n][rt][4 * q], acc[n][rt][4 * q + 1], acc[n][rt][4 * q + 2], acc[n][rt][4 * q + 3]};
                            t = __builtin_elementwise_fma(t, sv[q], bv[q]);
                            if (decltype(RES)::value) {
                                // t + float(half): v_fma_mix_f32 (half * 1.0 + t, one rounding = convert, then add) instead of two instructions
                                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(t[0]) : "v"(a4[q][rt][0]));
                                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t[1]) : "v"(a4[q][rt][0]));
                                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(t[2]) : "v"(a4[q][rt][1]));
                                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t[3]) : "v"(a4[q][rt][1]));
                            }
                            // (opaque: fma + conversion must not contract into v_fma_mixlo_f16, which rounds once -- the separate convolution
                            // kernels round to f32, then to half)
                            asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
                            const f16x2 lo = {(_Float16)t[0], (_Float16)t[1]}, hi = {(_Float16)t[2], (_Float16)t[3]};
                            const f16x2 zero = {(_Float16)0.f, (_Float16)0.f};
                            u32x2 o = {__builtin_bit_cast(unsigned, __builtin_elementwise_max(lo, zero)) & keep[rt],
                                       __builtin_bit_cast(unsigned, __builtin_elementwise_max(hi, zero)) & keep[rt]};
                            *reinterpret_cast<u32x2 *>(pa[q][rt]) = o;
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (conv == 0) epilogue(std::false_type{});
            else epilogue(std::true_type{});
            LAD_F16P_T(4)   // epilogue
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(wreg[0].a), "+v"(wreg[0].b), "+v"(wreg[1].a), "+v"(wreg[1].b), "+v"(wreg[2].a), "+v"(wreg[2].b)::"memory");
            LAD_F16P_T(4)   // (+ the drain in front of the loop back-edge: ~35 cycles)
        }
        out_prev = x_s;
    }
#ifdef LAD_STAMP
    if ((wave == 0 || wave == 7) && lane == 0 && blockIdx.x < 256)
        for (int j = 0; j < 6; ++j) lad_dbg_f16p[blockIdx.x * 16 + (wave ? 8 : 0) + j] = ph_[j];
#endif
    __syncthreads();   // the last image's output is complete
    if (wave >= 4) {
        const __amdgpu_buffer_rsrc_t r = make_rsrc(y + (int64_t)(img - grid) * IMG * COUT, (int64_t)IMG * BLK_ROWB);
        for (int pc = lt; pc < n_piece; pc += 256) buf_store16(*reinterpret_cast<const u32x4 *>(out_prev + blk_off(pc >> 3, pc & 7)), r, pc * 16);
    }
}

// ---- the same residual block for 16 and 32 channels: several small images per workgroup, both weight images resident ----------------
// Levels 2-4 of the sliding-window path run their identity blocks on images of 98-312 positions (per window at levels 3 and 4, per
// boundary strip at level 2): two launches each moving the tensor, with a tile's MFMAs in the hundreds of cycles.  Here a 256-thread
// workgroup takes G consecutive images (the shared-border layout makes them one contiguous row range whose image boundaries are zero
// rows), keeps them in LDS through conv1 -> BN -> ReLU -> conv2 -> BN -> + input -> ReLU, and both packed weight images (9 or 18 KB
// each) stay resident for its lifetime: no weight ring, no barrier inside a convolution.  Same MFMA, tap and k order per output element
// as conv_f16_s1_kernel<C, C, 9>, same epilogue arithmetic: identical bits.  LDS rows are C halfs, 16-byte slots XOR-ed with
// (row >> 2) & 3 (C = 32) / (row >> 3) & 1 (C = 16): the 16-lane groups of a ds_read_b128 land on 16 different slots of a bank line.
//   [w1: 9 taps][w2: 9 taps][P: R rows (the intermediate)][Q: R rows (input, then output)][zeros][row mask][BatchNorm coefficients]
// R = G * IMG rows; what P's last image row reads below itself is Q's first rows (a border row: zeros), Q's is the zero region; rows
// above P are weights (finite), reaching border outputs only, which are written as zero.
// (256 threads = 256-row tiles at 16 channels, two workgroups per CU; 512 at 32 channels, where two images with their intermediates
// fill the LDS: eight waves then hide each other's LDS latency)
template <int C>
struct Smb {
    static constexpr int THREADS = C == 32 ? 512 : 256, TILE = THREADS;
};
template <int C>
__device__ __forceinline__ unsigned smb_off(int row, int slot) {
    return (unsigned)(row * (C * 2)) + (unsigned)((slot ^ (C == 32 ? (row >> 2) & 3 : (row >> 3) & 1)) << 4);
}
static inline int smb_tiles(int rows, int tile) { return (rows + tile - 1) / tile; }
static inline size_t smb_lds_bytes(int C, int rows, int Wp) {
    const int tile = C == 32 ? 512 : 256;
    const int zrows = smb_tiles(rows, tile) * tile - rows + Wp + 2;
    return (size_t)18 * C * 64 + (size_t)(2 * rows + zrows) * (C * 2) + (size_t)smb_tiles(rows, tile) * tile + 4 * C * 4;
}
// SINGLE: ONE convolution with a residual from a second tensor -- y = relu(bn(conv(x)) + addend), the conv2 of a down-sampling block,
// whose shortcut is the addend (models.py:110-115) -- on the same machinery: x -> Q, addend -> P, the result in place over the addend
// (wt2 / sc2 / sh2 unused, `addend` = nullptr otherwise).  About twice as fast as conv_f16_s1_kernel on these shapes (weights resident,
// several images per workgroup, no per-tap barrier): 8,192 images of 26 x 12 x 16 in ~30 us against 56.
template <int C, bool SINGLE = false>
__global__ __launch_bounds__(Smb<C>::THREADS, 2) void block_f16_small_kernel(const _Float16 *__restrict__ x, _Float16 *__restrict__ y,
                                                                         const _Float16 *__restrict__ wt1, const float *__restrict__ sc1,
                                                                         const float *__restrict__ sh1, const _Float16 *__restrict__ wt2,
                                                                         const float *__restrict__ sc2, const float *__restrict__ sh2,
                                                                         const _Float16 *__restrict__ addend, int n_img, int Hp, int Wp,
                                                                         int G) {
    constexpr int KS = C / 16, RB = C * 2, SL = RB / 16, W_TAP = C * 64, TAPS = 9, QV = C / 8;
    constexpr int SMB_THREADS = Smb<C>::THREADS, SMB_TILE = Smb<C>::TILE;   // QV: valid channel quads per lane half
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int IMG = Hp * Wp, R = G * IMG, NTL = (R + SMB_TILE - 1) / SMB_TILE;
    const int zrows = NTL * SMB_TILE - R + Wp + 2;
    unsigned char *w1_s = lds_b, *w2_s = w1_s + TAPS * W_TAP;
    unsigned char *p_s = w2_s + TAPS * W_TAP, *q_s = p_s + R * RB, *z_s = q_s + R * RB;
    unsigned char *mask_s = z_s + zrows * RB;
    float *coef_s = reinterpret_cast<float *>(mask_s + NTL * SMB_TILE);
    for (int j = tid; j < TAPS * W_TAP / 16; j += SMB_THREADS) {
        reinterpret_cast<u32x4 *>(w1_s)[j] = reinterpret_cast<const u32x4 *>(wt1)[j];
        if (!SINGLE) reinterpret_cast<u32x4 *>(w2_s)[j] = reinterpret_cast<const u32x4 *>(wt2)[j];
    }
    for (int j = tid; j < zrows * (RB / 16); j += SMB_THREADS) reinterpret_cast<u32x4 *>(z_s)[j] = u32x4{0u, 0u, 0u, 0u};
    for (int j = tid; j < NTL * SMB_TILE; j += SMB_THREADS) {
        const int r = j % IMG, yp = r / Wp, xp = r - yp * Wp;
        mask_s[j] = (j < R && yp >= 1 && xp >= 1) ? 1 : 0;
    }
    if (tid < (SINGLE ? 2 : 4) * C) coef_s[tid] = (tid < C ? sc1 : tid < 2 * C ? sh1 : tid < 3 * C ? sc2 : sh2)[tid % C];
    const int i = lane & 31, h = lane >> 5;
    const int n_groups = (n_img + G - 1) / G;
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int img0 = grp * G, rows_here = min(G, n_img - img0) * IMG;
        // (a short last group also fetches the W + 2 zero rows behind the tensor's last image: what that image's last row reads below itself)
        const int n_piece = rows_here * SL, n_piece_in = (rows_here + (rows_here < R ? Wp + 1 : 0)) * SL;
        __syncthreads();   // the previous group's output has left Q (first group: the tables above are written)
        {   // the group's images -> Q by LDS-DMA, the swizzle applied on the global side; 1 KB (64 pieces) per instruction
            const _Float16 *src = x + (int64_t)img0 * IMG * C;
            for (int c = wave; c * 64 < n_piece_in; c += SMB_THREADS / 64) {
                const int pc = c * 64 + lane, row = pc / SL, sl = pc % SL;
                const int gs = sl ^ (C == 32 ? (row >> 2) & 3 : (row >> 3) & 1);
                if (pc < n_piece_in) dma16(src + row * C + gs * 8, lds_addr(q_s + c * 1024));
                if (SINGLE && pc < n_piece) dma16(addend + (int64_t)img0 * IMG * C + row * C + gs * 8, lds_addr(p_s + c * 1024));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto run_conv = [&](auto RES) {   // RES: conv2 (+ the block's input, read from the rows it overwrites)
            constexpr int conv = decltype(RES)::value ? 1 : 0;
            // (SINGLE: the one convolution reads Q and has its residual and its output in P)
            const unsigned char *src_s = (conv == 0 || SINGLE) ? q_s : p_s;
            const unsigned char *w_s = ((conv == 0 || SINGLE) ? w1_s : w2_s) + (h * 32 + i) * 16;
            unsigned char *dst_s = (conv == 0 || SINGLE) ? p_s : q_s;
            const float *cf = coef_s + (SINGLE ? 0 : conv * 2 * C) + 4 * h;
#pragma unroll 1
            for (int tile = 0; tile < NTL; ++tile) {
                const int row0 = tile * SMB_TILE + wave * 64 + i;
                f32x16 acc[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.0f;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int off = (tap / 3 - 1) * Wp + (tap % 3 - 1);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const f16x8 wf = *reinterpret_cast<const f16x8 *>(w_s + tap * W_TAP + ks * 2 * 32 * 16);
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            const int row = row0 + rt * 32 + off;   // (negative / past the group: see the layout note)
                            const f16x8 xf = *reinterpret_cast<const f16x8 *>(src_s + (int)smb_off<C>(row, ks * 2 + h));
                            acc[rt] = mfma32_f16(wf, xf, acc[rt]);
                        }
                    }
                }
                // register 4 q + j of lane (i, h) is channel 8 q + 4 h + j of position i (q < C / 8; the rest is the weight image's padding)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int row = row0 + rt * 32;
                    const unsigned keep = mask_s[row] ? 0xffffffffu : 0u;
                    // rows past the group go to the zero region's rows behind what valid outputs read (no EXEC region around the stores)
                    unsigned char *prow = row < R ? dst_s + row * RB : z_s + (row - R + Wp + 2) * RB;
                    const int sw = C == 32 ? (row >> 2) & 3 : (row >> 3) & 1;
#pragma unroll
                    for (int q = 0; q < QV; ++q) {
                        const f32x4 sv = *reinterpret_cast<const f32x4 *>(cf + 8 * q);
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(cf + C + 8 * q);
                        unsigned char *pa = prow + ((q ^ sw) << 4) + h * 8;
                        f32x4 t = {acc[rt][4 * q], acc[rt][4 * q + 1], acc[rt][4 * q + 2], acc[rt][4 * q + 3]};
                        t = __builtin_elementwise_fma(t, sv, bv);
                        if (conv == 1) {
                            const f16x4 a4 = *reinterpret_cast<const f16x4 *>(pa);
                            t += f32x4{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
                        }
                        t = __builtin_elementwise_max(t, f32x4{0.f, 0.f, 0.f, 0.f});
                        asm("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));   // (no fma + conversion contraction: block_f16_strip_kernel)
                        const f16x2 lo = {(_Float16)t[0], (_Float16)t[1]}, hi = {(_Float16)t[2], (_Float16)t[3]};
                        u32x2 o = {__builtin_bit_cast(unsigned, lo) & keep, __builtin_bit_cast(unsigned, hi) & keep};
                        *reinterpret_cast<u32x2 *>(pa) = o;
                    }
                }
            }
            __syncthreads();   // the intermediate / the output is complete
        };
        if (!SINGLE) run_conv(std::false_type{});
        run_conv(std::true_type{});
        {   // the output -> HBM, 16-byte pieces in global order
            _Float16 *dst = y + (int64_t)img0 * IMG * C;
            const unsigned char *out_s = SINGLE ? p_s : q_s;
            for (int pc = tid; pc < n_piece; pc += SMB_THREADS) {
                const int row = pc / SL, sl = pc % SL;
                *reinterpret_cast<u32x4 *>(dst + (int64_t)pc * 8) = *reinterpret_cast<const u32x4 *>(out_s + smb_off<C>(row, sl));
            }
            // (lad_f16_conv_fwd leaves the W + 2 rows behind the last image zero, as conv_f16_s1_kernel does)
            if (SINGLE && grp == n_groups - 1)
                for (int pc = tid; pc < (Wp + 1) * SL; pc += SMB_THREADS) *reinterpret_cast<u32x4 *>(dst + (int64_t)(n_piece + pc) * 8) = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

// stride 2 (3x3 pad 1 or 1x1).  The 32 input rows a wave needs for one tap are scattered (stride-2 positions): read
// in MFMA-fragment order (lane = row) every load instruction touches 32 different cache lines, and the texture
// addresser -- not HBM -- sets the pace (607 us per 2048-window chunk at 64->32).  Here a row is read by CIN/8
// neighbouring lanes (whole 16*CIN/8-byte rows, 8 lines per instruction at 64 channels), parked in a wave-private LDS
// stage and re-read in fragment order; LDS is in order within a wave, so the kernel's only barrier is before the
// epilogue.  Weights per lane from the packed image (L1/L2).
// WMAP (sliding-window inference, engine._forward_eval_stream): image b's input is not a tensor of its own but window b of the
// level-1 activation that lad_assemble_windows would build -- its rows come from three places of ONE buffer `in`:
// rows [0, band) from the top strip (image b of the strip images at the start of the buffer), rows [H - band, H) from the
// bottom strip (image bot_img0 + b), the rest from the shared stream image that follows the strips (row b + y of it).  The gather already goes through a per-row base index; with the map there is one base per kernel row ky.
struct WinMap {
    int B, H, band;        // windows, rows per window (input level), boundary rows taken from the strips
    int strip_rows;        // rows of a strip image: the top strip holds window rows [0, strip_rows), the bottom one [H - strip_rows, H)
    int img_t;             // rows (positions) of one strip image: (strip_rows + 1) * Wp
    int bot_img0;          // strip image that holds the bottom rows of window 0 (window b: bot_img0 + b)
    int stream_row0;       // first row of the (first) stream image in the buffer
    // phases 1: row y of window b is stream row b + y.  phases 2 (the input level is itself behind a stride-2 layer): the windows
    // of even and odd b have a stream image each (phase_img rows apart), and row y of window b is row (b >> 1) + y of image b & 1
    int phases, phase_img;
    // the launch produces whole windows (out_half 0: output image b = window b) or the next level's STRIPS (gather.hip: one image
    // of 2 out_half rows per window offset s: its upper half = the first rows of window s, its lower half = the last rows of
    // window s - out_win_shift, i.e. that window's rows from out_row_shift + out_half on)
    int out_half, out_win_shift, out_row_shift;
};

// The block's 1x1 stride-2 shortcut (models.py:98-106) reads exactly the rows the 3x3's centre tap gathers.  SC (round 5): it rides in
// the 3x3 launch -- a second accumulator fed by the centre tap's input fragments, its own weight chunk (behind the nine in LDS),
// BatchNorm fold and output -- instead of gathering them again in a launch of its own (64 -> 32: 72 us next to 314, per group of
// 8,192 windows).  Same MFMAs in the same order as the separate launch: identical bits.
struct ScOut {
    const _Float16 *wt;          // packed 1x1 weights (lad_f16_pack_weights, taps = 1)
    const float *scale, *shift;  // the shortcut BatchNorm, folded
    _Float16 *out;
};

template <int CIN, int COUT, int TAPS, bool WMAP = false, bool SC = false>
__global__ __launch_bounds__(THREADS, 2) void conv_f16_s2_kernel(const _Float16 *__restrict__ in,
                                                                 const _Float16 *__restrict__ wt,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 _Float16 *__restrict__ out, Geom gi, Geom go, int relu,
                                                                 int n_tiles, WinMap wm, ScOut sc) {
    static_assert(!SC || TAPS == 9, "the shortcut rides in the 3x3 launch");
    using C = HCfg<CIN, COUT, TAPS>;
    constexpr int NT = NTilesH<COUT>::NT;
    constexpr int COUTP = C::COUTP;
    constexpr int A8 = C::A8;            // 16-byte pieces per input row
    constexpr int LDA = C::LDA;          // padded row of the stage (halfs)
    constexpr int RPI = 64 / A8;         // rows one load instruction covers
    constexpr int NLD = 32 / RPI;        // load instructions per tap
    constexpr int NKY = WMAP ? 3 : 1;   // row bases per output row: one per kernel row with the window map
    __shared__ float mask_s[TM];
    __shared__ int rowbase_s[NKY * TM];
    __shared__ __attribute__((aligned(16))) float out_s[TM * (COUT + 4)];
    __shared__ __attribute__((aligned(16))) _Float16 stage_s[(THREADS / 64) * 32 * LDA];
    // The layer's weights, resident in LDS for the workgroup's lifetime (it walks tiles blockIdx.x, + gridDim.x, ...).  Read per
    // lane from the packed image in global memory they were as many vector-memory instructions as the gathered rows -- 36 KB
    // through the L1 per wave and tile at 64 -> 32 -- and the address path, not HBM, sets this kernel's pace.
    extern __shared__ __attribute__((aligned(16))) _Float16 w_s[];   // [TAPS][CHUNK_HALFS] (+ [CHUNK_HALFS] of the shortcut)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    for (int p = tid; p < TAPS * C::CHUNK_HALFS / 8; p += THREADS)
        *reinterpret_cast<u32x4 *>(w_s + p * 8) = *reinterpret_cast<const u32x4 *>(wt + p * 8);
    if (SC)
        for (int p = tid; p < C::CHUNK_HALFS / 8; p += THREADS)
            *reinterpret_cast<u32x4 *>(w_s + TAPS * C::CHUNK_HALFS + p * 8) = *reinterpret_cast<const u32x4 *>(sc.wt + p * 8);
    __syncthreads();
    const float *scale_p = scale, *shift_p = shift;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t q0 = (int64_t)tile * TM;
    const int64_t qo = q0 + wave * 32 + i;
    const bool inter = interior_row(qo, go);
    int64_t base_row = 0;  // non-interior output rows gather image 0 (in-bounds) and are discarded by the row mask
    int base_ky[3] = {0, 0, 0};
    if (inter) {
        const int64_t b = qo / go.img;
        const int rr = (int)(qo - b * go.img);
        const int ypo = rr / go.Wp;
        const int yo = ypo - 1, xo = rr - ypo * go.Wp - 1;
        base_row = b * gi.img + (int64_t)(2 * yo) * gi.Wp + 2 * xo;
        if (WMAP) {
            const bool bot = wm.out_half > 0 && yo >= wm.out_half;
            const int64_t win = bot ? b - wm.out_win_shift : b;   // (windows outside [0, B) read in-bounds rows; nobody uses the result)
            const int yw = yo + (bot ? wm.out_row_shift : 0);      // the window's output row
            const int64_t srow = wm.phases == 2 ? wm.stream_row0 + (win & 1) * (int64_t)wm.phase_img + ((win >> 1) + 1) * (int64_t)gi.Wp
                                                : wm.stream_row0 + (win + 1) * (int64_t)gi.Wp;   // row 0 of the window in its stream image
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int y = 2 * yw + ky - 1;   // row of the window this kernel row reads (-1 / H: the window's zero padding)
                int64_t R;
                if (y < 0 || y >= wm.H) R = win * wm.img_t;                                                  // a border row: zeros
                else if (y < wm.band) R = win * wm.img_t + (int64_t)(y + 1) * gi.Wp;                         // top strip
                else if (y >= wm.H - wm.band) R = (wm.bot_img0 + win) * (int64_t)wm.img_t + (int64_t)(y - (wm.H - wm.strip_rows) + 1) * gi.Wp;   // bottom strip
                else R = srow + (int64_t)y * gi.Wp;                                                          // the shared stream
                base_ky[ky] = (int)(R + 2 * xo);
            }
        }
    }
    if (h == 0) {
        mask_s[wave * 32 + i] = inter ? 1.0f : 0.0f;
        if (WMAP) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) rowbase_s[ky * TM + wave * 32 + i] = base_ky[ky];
        } else {
            rowbase_s[wave * 32 + i] = (int)base_row;  // launcher: input rows < 2^31
        }
    }
    // cooperative mapping: lane -> (row lane / A8 + RPI * j, piece lane % A8)
    const int piece = lane % A8, rsub = lane / A8;
    int rb[NKY][NLD];
#pragma unroll
    for (int k = 0; k < NKY; ++k)
#pragma unroll
        for (int j = 0; j < NLD; ++j) rb[k][j] = rowbase_s[k * TM + wave * 32 + rsub + RPI * j];  // same wave wrote it: in order
    _Float16 *stage = stage_s + wave * 32 * LDA;

    f32x16 acc[NT], acc2[SC ? NT : 1];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[n][r] = 0.0f;
            if (SC) acc2[n][r] = 0.0f;
        }
    const _Float16 *w_base = w_s + (h * COUTP + i) * 8;
    u32x4 pre[NLD];
    auto fetch = [&](int tap) {
        const int ky = (TAPS == 9) ? tap / 3 : 1, kx = (TAPS == 9) ? tap % 3 : 1;
        const int shift_rows = (WMAP ? 0 : ky * gi.Wp) + kx;  // input border rows are zero in HBM: no per-tap test
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            pre[j] = *reinterpret_cast<const u32x4 *>(in + ((int64_t)rb[WMAP ? ky : 0][j] + shift_rows) * CIN + piece * 8);
    };
    fetch(0);
#pragma unroll 1
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) *reinterpret_cast<u32x4 *>(stage + (rsub + RPI * j) * LDA + piece * 8) = pre[j];
        if (tap + 1 < TAPS) fetch(tap + 1);  // the next tap's rows travel while this tap's MFMAs run
        const _Float16 *wp = w_base + tap * C::CHUNK_HALFS;
        f16x8 av[C::KS], bv[C::KS][NT];
#pragma unroll
        for (int s2 = 0; s2 < C::KS; ++s2) {
            av[s2] = *reinterpret_cast<const f16x8 *>(stage + i * LDA + s2 * 16 + 8 * h);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[s2][n] = *reinterpret_cast<const f16x8 *>(wp + (s2 * 2 * COUTP + n * 32) * 8);
        }
#pragma unroll
        for (int s2 = 0; s2 < C::KS; ++s2)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = mfma32_f16(av[s2], bv[s2][n], acc[n]);
        if (SC && tap == 4) {   // the centre tap's rows are the 1x1's rows
            const _Float16 *wp2 = w_base + TAPS * C::CHUNK_HALFS;
#pragma unroll
            for (int s2 = 0; s2 < C::KS; ++s2)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc2[n] = mfma32_f16(av[s2], *reinterpret_cast<const f16x8 *>(wp2 + (s2 * 2 * COUTP + n * 32) * 8), acc2[n]);
        }
    }
    epilogue_f16<COUT>(acc, scale_p, shift_p, nullptr, out, mask_s, out_s, q0, go.rows, relu);   // (wave-private LDS throughout: no barrier)
    if constexpr (SC) epilogue_f16<COUT>(acc2, sc.scale, sc.shift, nullptr, sc.out, mask_s, out_s, q0, go.rows, 0);
    }
}

// Stem in eval mode, f32 features in -> half activations out (bn1 + ReLU folded); window addressing as lad_stem_fwd_eval.
constexpr int SCOUT = 64, SCQ = SCOUT / 4, SRL = THREADS / SCQ;
__global__ __launch_bounds__(THREADS) void stem_f16_kernel(const float *__restrict__ feat, const float *__restrict__ w,
                                                           const float *__restrict__ scale, const float *__restrict__ shift,
                                                           _Float16 *__restrict__ out, Geom g, int H, int W,
                                                           int64_t frame_stride, int64_t frames_avail) {
    const int tid = threadIdx.x, cq = tid % SCQ, rl = tid / SCQ;
    float wr[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[c][t] = w[(cq * 4 + c) * 9 + t];
    const float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
    const int64_t q0 = (int64_t)blockIdx.x * TM;
    __shared__ __attribute__((aligned(16))) float tap_s[TM * TAPW];
    static_assert(TM == STEM_TM, "tap table is sized for the stem tile");
    fill_taps(feat, g, H, W, q0, frame_stride, frames_avail, tap_s);  // one decode + gather per row, shared by its 16 threads
    __syncthreads();
    for (int r = rl; r < TM; r += SRL) {
        const int64_t q = q0 + r;
        if (q >= g.rows) break;
        f16x4 o = {0, 0, 0, 0};
        float v[9];
        if (read_taps(tap_s, r, v)) {
            float acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(v[t], wr[c][t], acc[c]);
            o = f16x4{(_Float16)fmaxf(fmaf(acc[0], sc.x, sh.x), 0.f), (_Float16)fmaxf(fmaf(acc[1], sc.y, sh.y), 0.f),
                      (_Float16)fmaxf(fmaf(acc[2], sc.z, sh.z), 0.f), (_Float16)fmaxf(fmaf(acc[3], sc.w, sh.w), 0.f)};
        }
        *reinterpret_cast<f16x4 *>(out + q * SCOUT + cq * 4) = o;
    }
}

// AvgPool2d(4) over half activations -> f32 pooled features (feeds the f32 head)
__global__ void pool_f16_kernel(const _Float16 *__restrict__ x, float *__restrict__ pooled, int64_t batch, int Hp, int Wp, int C,
                                int PH, int PW) {
    const int F = C * PH * PW;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * F) return;
    const int64_t b = idx / F;
    const int f = (int)(idx - b * F);
    const int c = f / (PH * PW), ph = (f / PW) % PH, pw = f % PW;
    const _Float16 *img = x + b * (int64_t)Hp * Wp * C;
    float s = 0.f;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) s += (float)img[((1 + 4 * ph + dy) * Wp + (1 + 4 * pw + dx)) * C + c];
    pooled[idx] = s * 0.0625f;
}

Geom geom_of(int64_t batch, int H, int W) { return make_geom(batch, H, W); }

template <int CIN, int COUT, int TAPS, int WAVES>
int launch_h1w(const _Float16 *in, const _Float16 *wt, const float *scale, const float *shift, const _Float16 *addend,
               _Float16 *out, const Geom &g, int relu, hipStream_t st, bool probe_only) {
    using C = HCfg<CIN, COUT, TAPS>;
    constexpr int TMV = 32 * WAVES;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMV + 2 * halo;
    constexpr int SLOTS = 2;
    const size_t main_bytes = std::max<size_t>(SLOTS * (size_t)C::CHUNK_HALFS * 2 + (size_t)nrows * C::LDA * 2, (size_t)TMV * (COUT + 4) * 4);
    const size_t lds = ((main_bytes + 3) / 4) * 4 + TMV * sizeof(float);
    const bool fits = lds <= (WAVES == 4 ? 160 : 80) * 1024;
    if (probe_only) return fits ? LAD_OK : LAD_ERR_INVALID;
    if (!fits) return lad::fail(LAD_ERR_INVALID, "conv_f16: image too wide for the LDS tile (W = %d)", g.Wp - 1);
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s1_kernel<CIN, COUT, TAPS, false, WAVES>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s1_kernel<CIN, COUT, TAPS, true, WAVES>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const dim3 grid((unsigned)lad::ceil_div(g.rows, TMV)), block(64 * WAVES);
    if (addend != nullptr)
        hipLaunchKernelGGL((conv_f16_s1_kernel<CIN, COUT, TAPS, true, WAVES>), grid, block, lds, st, in, wt, scale, shift, addend, out, g, relu);
    else
        hipLaunchKernelGGL((conv_f16_s1_kernel<CIN, COUT, TAPS, false, WAVES>), grid, block, lds, st, in, wt, scale, shift, addend, out, g, relu);
    return lad::check_launch("conv_f16_s1_kernel");
}

template <int CIN, int COUT, int TAPS>
int launch_h1(const _Float16 *in, const _Float16 *wt, const float *scale, const float *shift, const _Float16 *addend,
              _Float16 *out, const Geom &g, int relu, hipStream_t st) {
    // interior_row32: 32-bit row numbers, and the multiply-high division by Wp is exact for positions below 2^32 / Wp (the run-long
    // stream image of the sliding-window path is 16.2 M rows of 45: round 5)
    if (g.rows >= (1ll << 31) || (uint64_t)g.img * (uint64_t)g.Wp >= (1ull << 32))
        return lad::fail(LAD_ERR_INVALID, "conv_f16: tensor of %lld rows exceeds the 32-bit row decode", (long long)g.rows);
    constexpr bool WIDE = (CIN == 64 && COUT == 64 && TAPS == 9);
    if (WIDE && g.rows >= 4096ll * 256) {  // enough 256-row tiles for a persistent workgroup per CU
        const int nrows = 256 + 2 * (g.Wp + 1);
        const size_t lds = (size_t)9 * 4096 * 2 + (size_t)384 * 72 * 2 + 256 * 4;
        if (nrows <= 384) {
            static bool attr_p = false;
            if (!attr_p) {
                LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s1p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s1p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_p = true;
            }
            const int n_tiles = (int)lad::ceil_div(g.rows, 256);
            const dim3 grid((unsigned)std::min(n_tiles, 256)), block(512);
            if (addend != nullptr)
                hipLaunchKernelGGL(conv_f16_s1p_kernel<true>, grid, block, lds, st, in, wt, scale, shift, addend, out, g, relu, n_tiles);
            else
                hipLaunchKernelGGL(conv_f16_s1p_kernel<false>, grid, block, lds, st, in, wt, scale, shift, addend, out, g, relu, n_tiles);
            return lad::check_launch("conv_f16_s1p_kernel");
        }
    }
    if (WIDE && g.rows >= 512ll * 256 && launch_h1w<CIN, COUT, TAPS, WIDE ? 8 : 4>(in, wt, scale, shift, addend, out, g, relu, st, true) == LAD_OK)
        return launch_h1w<CIN, COUT, TAPS, WIDE ? 8 : 4>(in, wt, scale, shift, addend, out, g, relu, st, false);
    return launch_h1w<CIN, COUT, TAPS, 4>(in, wt, scale, shift, addend, out, g, relu, st, false);
}

template <int CIN, int COUT, int TAPS, bool SC = false>
int launch_h2(const _Float16 *in, const _Float16 *wt, const float *scale, const float *shift, _Float16 *out, const Geom &gi,
              const Geom &go, int relu, hipStream_t st, const WinMap *wm = nullptr, const ScOut *sc = nullptr) {
    using C = HCfg<CIN, COUT, TAPS>;
    if (gi.rows >= (1ll << 31)) return lad::fail(LAD_ERR_INVALID, "conv_f16_s2: %lld input rows exceed 32-bit row indices", (long long)gi.rows);
    constexpr size_t W_BYTES = (size_t)(TAPS + (SC ? 1 : 0)) * C::CHUNK_HALFS * 2;
    constexpr size_t STATIC_BYTES = TM * 4 + 3 * TM * 4 + (size_t)TM * (COUT + 4) * 4 + (size_t)(THREADS / 64) * 32 * C::LDA * 2;
    constexpr int PER_CU = (int)std::min<size_t>(4, (160 * 1024) / (W_BYTES + STATIC_BYTES));
    static_assert(PER_CU >= 2, "conv_f16_s2: two workgroups per CU");
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s2_kernel<CIN, COUT, TAPS, true, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_BYTES));
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_f16_s2_kernel<CIN, COUT, TAPS, false, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_BYTES));
        attr_set = true;
    }
    const int64_t n_tiles = lad::ceil_div(go.rows, TM);
    const dim3 grid((unsigned)std::min<int64_t>(n_tiles, 256 * PER_CU)), block(THREADS);
    const ScOut sco = SC ? *sc : ScOut{nullptr, nullptr, nullptr, nullptr};
    if (wm != nullptr)
        hipLaunchKernelGGL((conv_f16_s2_kernel<CIN, COUT, TAPS, true, SC>), grid, block, W_BYTES, st, in, wt, scale, shift, out, gi, go, relu,
                           (int)n_tiles, *wm, sco);
    else
        hipLaunchKernelGGL((conv_f16_s2_kernel<CIN, COUT, TAPS, false, SC>), grid, block, W_BYTES, st, in, wt, scale, shift, out, gi, go, relu,
                           (int)n_tiles, WinMap{0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0}, sco);
    return lad::check_launch("conv_f16_s2_kernel");
}

}  // namespace

#ifdef LAD_STAMP
extern "C" int lad_debug_read_f16p_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_dbg_f16p), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int64_t lad_f16_packed_weight_halfs(int32_t cout, int32_t cin, int32_t taps) {
    return (int64_t)taps * cin * (((cout + 31) / 32) * 32);
}

extern "C" int lad_f16_pack_weights(const float *w, int32_t cout, int32_t cin, int32_t taps, void *wt, void *stream) {
    using namespace lad;
    LAD_REQUIRE(w && wt, "lad_f16_pack_weights: null buffer");
    LAD_REQUIRE((taps == 9 || taps == 1) && cin % 16 == 0 && cin > 0 && cout > 0, "lad_f16_pack_weights: cin must be a multiple of 16");
    const int64_t total = lad_f16_packed_weight_halfs(cout, cin, taps);
    hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16 *)wt,
                       cout, cin, taps);
    return check_launch("pack_f16_kernel");
}

static int launch_small_block(const void *x, const void *wt1, const float *scale1, const float *shift1, const void *wt2, const float *scale2,
                              const float *shift2, const void *addend, void *y, int64_t batch, int32_t H, int32_t W, int32_t channels,
                              void *stream, const char *who);

#define LAD_H1_CASE(CI, CO, T)                \
    if (cin == CI && cout == CO && taps == T) \
        return launch_h1<CI, CO, T>((const _Float16 *)in, (const _Float16 *)wt, scale, shift, (const _Float16 *)addend, (_Float16 *)out, g, relu, (hipStream_t)stream);

extern "C" int lad_f16_conv_fwd(const void *in, const void *wt, const float *scale, const float *shift, const void *addend,
                                void *out, int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                                int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && scale && shift && out, "lad_f16_conv_fwd: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_f16_conv_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    // conv2 of a down-sampling block at 16 channels on many small images (the windows at levels 3 and 4): block_f16_small_kernel's
    // one-convolution form, same bits.  (At 32 channels -- the 13 x 23 strips of level 2 -- that form takes 122 us against this
    // path's 104: not routed; profiles/r05_small_blocks.log.)
    if (addend != nullptr && relu && taps == 9 && cin == cout && cin == 16 && batch >= 512 && addend != out && in != out) {
        const int rc = launch_small_block(in, wt, scale, shift, nullptr, nullptr, nullptr, addend, out, batch, H, W, cin, stream, "lad_f16_conv_fwd");
        if (rc != LAD_NOT_COVERED) return rc;
    }
    const Geom g = geom_of(batch, H, W);
    LAD_H1_CASE(64, 64, 9)
    LAD_H1_CASE(32, 32, 9)
    LAD_H1_CASE(16, 16, 9)
    return fail(LAD_ERR_INVALID, "lad_f16_conv_fwd: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

#define LAD_H2_CASE(CI, CO, T)                \
    if (cin == CI && cout == CO && taps == T) \
        return launch_h2<CI, CO, T>((const _Float16 *)in, (const _Float16 *)wt, scale, shift, (_Float16 *)out, gi, go, relu, (hipStream_t)stream);

extern "C" int lad_f16_conv_s2_fwd(const void *in, const void *wt, const float *scale, const float *shift, void *out,
                                   int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps, int32_t relu,
                                   void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && scale && shift && out, "lad_f16_conv_s2_fwd: null buffer");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_f16_conv_s2_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom gi = geom_of(batch, H, W);
    const Geom go = geom_of(batch, (H + 1) / 2, (W + 1) / 2);
    LAD_H2_CASE(64, 32, 9)
    LAD_H2_CASE(32, 16, 9)
    LAD_H2_CASE(16, 16, 9)
    LAD_H2_CASE(64, 32, 1)
    LAD_H2_CASE(32, 16, 1)
    LAD_H2_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_f16_conv_s2_fwd: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

// conv1 (3x3 stride 2 + BatchNorm + ReLU) and the 1x1 stride-2 shortcut (+ BatchNorm) of a down-sampling block in ONE launch:
// = lad_f16_conv_s2_fwd(taps 9, relu) + lad_f16_conv_s2_fwd(taps 1, no relu) on the same input, identical bits (models.py:98-106).
extern "C" int lad_f16_conv_s2_fwd_sc(const void *in, const void *wt, const float *scale, const float *shift, void *out, const void *wt_sc,
                                      const float *scale_sc, const float *shift_sc, void *out_sc, int64_t batch, int32_t H, int32_t W,
                                      int32_t cin, int32_t cout, int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && wt && scale && shift && out && wt_sc && scale_sc && shift_sc && out_sc, "lad_f16_conv_s2_fwd_sc: null buffer");
    LAD_REQUIRE(out != out_sc, "lad_f16_conv_s2_fwd_sc: the two outputs must be different tensors");
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1, "lad_f16_conv_s2_fwd_sc: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom gi = geom_of(batch, H, W);
    const Geom go = geom_of(batch, (H + 1) / 2, (W + 1) / 2);
    const ScOut sc{(const _Float16 *)wt_sc, scale_sc, shift_sc, (_Float16 *)out_sc};
#define LAD_H2SC_CASE(CI, CO)     \
    if (cin == CI && cout == CO) \
        return launch_h2<CI, CO, 9, true>((const _Float16 *)in, (const _Float16 *)wt, scale, shift, (_Float16 *)out, gi, go, relu, (hipStream_t)stream, nullptr, &sc);
    LAD_H2SC_CASE(64, 32)
    LAD_H2SC_CASE(32, 16)
    LAD_H2SC_CASE(16, 16)
#undef LAD_H2SC_CASE
    return fail(LAD_ERR_INVALID, "lad_f16_conv_s2_fwd_sc: unsupported (cin=%d, cout=%d)", cin, cout);
}

extern "C" int lad_f16_stem_fwd(const float *feat, const float *weight, const float *scale, const float *shift, void *out,
                                int64_t batch, int32_t H, int32_t W, int32_t cout, int64_t frame_stride, int64_t frames_avail,
                                void *stream) {
    using namespace lad;
    LAD_REQUIRE(feat && weight && scale && shift && out, "lad_f16_stem_fwd: null buffer");
    LAD_REQUIRE(cout == SCOUT, "lad_f16_stem_fwd: cout must be %d", SCOUT);
    LAD_REQUIRE(batch >= 0 && H >= 1 && W >= 1 && frame_stride >= 1 && frames_avail >= 0, "lad_f16_stem_fwd: bad geometry");
    if (batch == 0) return LAD_OK;
    const Geom g = geom_of(batch, H, W);
    hipLaunchKernelGGL(stem_f16_kernel, dim3((unsigned)ceil_div(g.rows, TM)), dim3(THREADS), 0, (hipStream_t)stream, feat, weight,
                       scale, shift, (_Float16 *)out, g, H, W, frame_stride, frames_avail);
    return check_launch("stem_f16_kernel");
}

extern "C" int lad_f16_pool_fwd(const void *x, float *pooled, int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(x && pooled, "lad_f16_pool_fwd: null buffer");
    LAD_REQUIRE(H >= 4 && W >= 4 && channels >= 1, "lad_f16_pool_fwd: AvgPool2d(4) needs H, W >= 4");
    if (batch == 0) return LAD_OK;
    const int PH = H / 4, PW = W / 4;
    const int64_t n = batch * channels * PH * PW;
    hipLaunchKernelGGL(pool_f16_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x,
                       pooled, batch, H + 1, W + 1, channels, PH, PW);
    return check_launch("pool_f16_kernel");
}

// The stride-2 convolutions that FOLLOW shared layers in the sliding-window path, reading each window's activation from where
// it lies instead of from an assembled copy (WinMap above; same arithmetic and summation order as lad_f16_conv_s2_fwd on the
// tensor lad_assemble_windows would have written).  `act`: the strip images of strip_rows rows (window b: its top rows in image
// b, its bottom rows in image bottom_image0 + b), followed -- at row stream_row0 -- by the stream image(s).
static int conv_s2_mapped(const void *act, const void *wt, const float *scale, const float *shift, void *out, const ScOut *sc,
                          int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows, int64_t bottom_image0,
                          int64_t stream_row0, int32_t phases, int64_t phase_rows, int64_t act_rows, int32_t out_rows, int32_t cin,
                          int32_t cout, int32_t taps, int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(act && wt && scale && shift && out, "lad_f16_conv_s2_fwd_mapped: null buffer");
    LAD_REQUIRE(n_windows >= 1 && H >= 1 && W >= 1 && band >= 1 && strip_rows >= band && H >= 2 * band,
                "lad_f16_conv_s2_fwd_mapped: bad geometry");
    LAD_REQUIRE(phases == 1 || phases == 2, "lad_f16_conv_s2_fwd_mapped: phases must be 1 or 2");
    const int Ho = (H + 1) / 2;
    LAD_REQUIRE(out_rows >= 0 && out_rows <= Ho, "lad_f16_conv_s2_fwd_mapped: out_rows must be in [0, %d]", Ho);
    const Geom gi = geom_of(n_windows, H, W);   // (only Wp and the per-image decode of the OUTPUT rows are used)
    LAD_REQUIRE(out_rows % 2 == 0 && 2 * out_rows <= 2 * Ho, "lad_f16_conv_s2_fwd_mapped: out_rows must be even and at most %d", Ho);
    const int64_t out_shift = 2 * (int64_t)(Ho - out_rows);   // windows between the two users of one output strip
    const Geom go = out_rows > 0 ? geom_of(n_windows + out_shift, out_rows, (W + 1) / 2) : geom_of(n_windows, Ho, (W + 1) / 2);
    const int64_t Wp = W + 1;
    const int64_t img_t = (int64_t)(strip_rows + 1) * Wp;
    // every row the map can produce lies inside the caller's buffer of act_rows rows (+ the 3x3 kernel's reach of 2 columns)
    // (strips out: window offsets up to n_windows + out_shift - 1 are read for their first rows, offsets down to -out_shift for
    // their last rows -- inside the strips and the stream as long as the strips of THIS level pair windows no closer)
    LAD_REQUIRE(out_rows == 0 || (H % 2 == 0 && out_rows + band + 1 <= H), "lad_f16_conv_s2_fwd_mapped: strips out need an even H >= out_rows + band + 1");
    LAD_REQUIRE(out_rows == 0 || (phases == 1 && out_shift <= bottom_image0 && bottom_image0 <= H - strip_rows),
                "lad_f16_conv_s2_fwd_mapped: strips out need paired input strips (bottom_image0 = %lld) at least %lld windows apart",
                (long long)bottom_image0, (long long)out_shift);
    const int64_t win_hi = n_windows - 1;
    const int64_t stream_hi = phases == 2 ? stream_row0 + phase_rows + ((win_hi >> 1) + H) * Wp + Wp
                                          : stream_row0 + (win_hi + H) * Wp + Wp;
    LAD_REQUIRE(bottom_image0 >= 0 && stream_row0 >= (bottom_image0 + n_windows) * img_t && stream_hi + 2 <= act_rows,
                "lad_f16_conv_s2_fwd_mapped: the stream image(s) do not fit the buffer (%lld rows needed, %lld given)",
                (long long)(stream_hi + 2), (long long)act_rows);
    LAD_REQUIRE(act_rows < ((int64_t)1 << 31), "lad_f16_conv_s2_fwd_mapped: more than 2^31 rows");
    const WinMap wm{(int)n_windows, H, band, strip_rows, (int)img_t, (int)bottom_image0, (int)stream_row0, phases, (int)phase_rows,
                    out_rows / 2, (int)out_shift, Ho - out_rows};
#define LAD_H2W_CASE(CI, CO, T)               \
    if (cin == CI && cout == CO && taps == T) \
        return launch_h2<CI, CO, T>((const _Float16 *)act, (const _Float16 *)wt, scale, shift, (_Float16 *)out, gi, go, relu, (hipStream_t)stream, &wm);
#define LAD_H2WSC_CASE(CI, CO)                \
    if (cin == CI && cout == CO && taps == 9) \
        return launch_h2<CI, CO, 9, true>((const _Float16 *)act, (const _Float16 *)wt, scale, shift, (_Float16 *)out, gi, go, relu, (hipStream_t)stream, &wm, sc);
    if (sc != nullptr) {
        LAD_H2WSC_CASE(64, 32)
        LAD_H2WSC_CASE(32, 16)
        return fail(LAD_ERR_INVALID, "lad_f16_conv_s2_fwd_mapped_sc: unsupported (cin=%d, cout=%d)", cin, cout);
    }
    LAD_H2W_CASE(64, 32, 9)
    LAD_H2W_CASE(64, 32, 1)
    LAD_H2W_CASE(32, 16, 9)
    LAD_H2W_CASE(32, 16, 1)
#undef LAD_H2W_CASE
#undef LAD_H2WSC_CASE
    return fail(LAD_ERR_INVALID, "lad_f16_conv_s2_fwd_mapped: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}

extern "C" int lad_f16_conv_s2_fwd_mapped(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                                          int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows,
                                          int64_t bottom_image0, int64_t stream_row0, int32_t phases, int64_t phase_rows, int64_t act_rows,
                                          int32_t out_rows, int32_t cin, int32_t cout, int32_t taps, int32_t relu, void *stream) {
    return conv_s2_mapped(act, wt, scale, shift, out, nullptr, n_windows, H, W, band, strip_rows, bottom_image0, stream_row0, phases,
                          phase_rows, act_rows, out_rows, cin, cout, taps, relu, stream);
}

// ... with the block's 1x1 shortcut in the same launch (lad_f16_conv_s2_fwd_sc over the window map): taps = 9 + 1
extern "C" int lad_f16_conv_s2_fwd_mapped_sc(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                                             const void *wt_sc, const float *scale_sc, const float *shift_sc, void *out_sc,
                                             int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t strip_rows,
                                             int64_t bottom_image0, int64_t stream_row0, int32_t phases, int64_t phase_rows,
                                             int64_t act_rows, int32_t out_rows, int32_t cin, int32_t cout, int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(wt_sc && scale_sc && shift_sc && out_sc && out_sc != out, "lad_f16_conv_s2_fwd_mapped_sc: null or aliased shortcut buffer");
    const ScOut sc{(const _Float16 *)wt_sc, scale_sc, shift_sc, (_Float16 *)out_sc};
    return conv_s2_mapped(act, wt, scale, shift, out, &sc, n_windows, H, W, band, strip_rows, bottom_image0, stream_row0, phases,
                          phase_rows, act_rows, out_rows, cin, cout, 9, relu, stream);
}

// the level-1 case of it: one strip of 2 * band rows per frame offset (lad_assemble_windows), one stream image of
// n_windows + H - 1 rows right behind them
extern "C" int lad_f16_conv_s2_fwd_windows(const void *act, const void *wt, const float *scale, const float *shift, void *out,
                                           int64_t n_windows, int32_t H, int32_t W, int32_t band, int32_t cin, int32_t cout,
                                           int32_t taps, int32_t relu, void *stream) {
    using namespace lad;
    LAD_REQUIRE(n_windows >= 1 && H >= 1 && W >= 1 && band >= 1 && H >= 2 * band, "lad_f16_conv_s2_fwd_windows: bad geometry");
    LAD_REQUIRE(cin == 64 && cout == 32, "lad_f16_conv_s2_fwd_windows: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
    const int64_t img_t = (int64_t)(2 * band + 1) * (W + 1);
    const int64_t bottom_image0 = H - 2 * band;
    const int64_t stream_row0 = (bottom_image0 + n_windows) * img_t;
    const int64_t act_rows = stream_row0 + (n_windows + H) * (int64_t)(W + 1) + W + 2;
    return lad_f16_conv_s2_fwd_mapped(act, wt, scale, shift, out, n_windows, H, W, band, 2 * band, bottom_image0, stream_row0, 1, 0,
                                      act_rows, 0, cin, cout, taps, relu, stream);
}

// block_f16_small_kernel: the block (addend == nullptr) or one convolution with a residual (SINGLE).  Picks the group size with the
// fullest tiles among those that leave room for two workgroups per CU, else for one.  LAD_NOT_COVERED: nothing launched, no error string.
static int launch_small_block(const void *x, const void *wt1, const float *scale1, const float *shift1, const void *wt2, const float *scale2,
                              const float *shift2, const void *addend, void *y, int64_t batch, int32_t H, int32_t W, int32_t channels,
                              void *stream, const char *who) {
    using namespace lad;
    const int Wp = W + 1, img = (H + 1) * Wp;
    if (batch < 512 || batch >= (1 << 30) || W + 2 > 280 || (addend != nullptr && channels != 16))
        return LAD_NOT_COVERED;
    const int tile = channels == 32 ? 512 : 256;
    int best = 0;
    double best_eff = 0.0;
    for (int pass = 0; pass < 2 && best == 0; ++pass)
        for (int g = 1; g <= 64 && (int64_t)g * img <= 8192; ++g) {
            const int rows = g * img;
            if (smb_lds_bytes(channels, rows, Wp) > (size_t)(pass == 0 ? 80 * 1024 : 160 * 1024)) break;
            const double eff = (double)rows / (smb_tiles(rows, tile) * tile);
            if (eff >= best_eff) best_eff = eff, best = g;
        }
    if (best == 0) return LAD_NOT_COVERED;
    const size_t lds = smb_lds_bytes(channels, best * img, Wp);
    const int64_t n_groups = ceil_div(batch, (int64_t)best);
    const dim3 grid((unsigned)std::min<int64_t>(n_groups, 256 * (lds <= 80 * 1024 ? 2 : 1)));
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)block_f16_small_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)block_f16_small_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)block_f16_small_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
#define LAD_SMB_LAUNCH(C, S)                                                                                                                    \
    hipLaunchKernelGGL((block_f16_small_kernel<C, S>), grid, dim3(Smb<C>::THREADS), lds, (hipStream_t)stream, (const _Float16 *)x, (_Float16 *)y, \
                       (const _Float16 *)wt1, scale1, shift1, (const _Float16 *)wt2, scale2, shift2, (const _Float16 *)addend, (int)batch, H + 1, \
                       Wp, best)
    if (channels == 16 && addend == nullptr) LAD_SMB_LAUNCH(16, false);
    else if (channels == 16) LAD_SMB_LAUNCH(16, true);
    else LAD_SMB_LAUNCH(32, false);
#undef LAD_SMB_LAUNCH
    return check_launch("block_f16_small_kernel");
}

// One residual block (identity shortcut, 64 channels) on images small enough for a CU's LDS: y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// with both BatchNorms folded (scale / shift as lad_f16_conv_fwd takes them), x and y shared-border half tensors of `batch` images
// (y may not be x).  Returns LAD_NOT_COVERED (nothing launched, lad_last_error untouched) when the geometry does not fit -- (H + 1)(W + 1) <= 512 positions and
// (H + 1)(W + 1) + W <= 562 (160 KB of LDS) --
// or the launch is too small to fill the chip (batch < 256): the caller then runs the two convolutions by lad_f16_conv_fwd.
// Bit-identical to that pair of calls.  Replaces models.py:110-115 (ResidualBlock.forward, eval mode) for block1 on the boundary
// strips of the sliding-window path (segment_laughter.py:90-101).
extern "C" int lad_f16_block_fwd(const void *x, const void *wt1, const float *scale1, const float *shift1, const void *wt2,
                                 const float *scale2, const float *shift2, void *y, int64_t batch, int32_t H, int32_t W,
                                 int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(x && y && wt1 && wt2 && scale1 && shift1 && scale2 && shift2, "lad_f16_block_fwd: null buffer");
    LAD_REQUIRE(x != y, "lad_f16_block_fwd: the block cannot run in place");
    LAD_REQUIRE((channels == 64 || channels == 32 || channels == 16) && H >= 1 && W >= 1, "lad_f16_block_fwd: 16, 32 or 64 channels");
    const int Hp = H + 1, Wp = W + 1, img = Hp * Wp;
    if (channels != 64)
        return launch_small_block(x, wt1, scale1, shift1, wt2, scale2, shift2, nullptr, y, batch, H, W, channels, stream, "lad_f16_block_fwd");
    if (img > 512 || blk_lds_bytes(img, Wp) > 160 * 1024 || batch < 256 || batch >= (1 << 30))
        return LAD_NOT_COVERED;
    const size_t lds = blk_lds_bytes(img, Wp);   // 157.1 KB for the product's 11 x 45 strips; 160 KB at 512 positions and W = 99
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)block_f16_strip_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(block_f16_strip_kernel<false>, dim3((unsigned)std::min<int64_t>(batch, 256)), dim3(BLK_THREADS), lds, (hipStream_t)stream,
                       (const _Float16 *)x, (_Float16 *)y, (const _Float16 *)wt1, scale1, shift1, (const _Float16 *)wt2, scale2, shift2,
                       (int)batch, Hp, Wp, StripMap{nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr});
    return check_launch("block_f16_strip_kernel");
}

// lad_f16_stem_fwd + lad_f16_block_fwd (64 channels) for the FIRST block of the sliding-window strips in one launch: strip b = frames
// [b, b + H) of `feat` (zero-padded above and below, zeros from frames_avail on), stem (conv1 + folded bn1 + ReLU, models.py:224) and
// the residual block (models.py:110-115, eval mode).  The stem of the strip's rows 1 .. H - 2 is not recomputed: it is rows
// stream_row0 + b + 1 .. of `stream_act`, lad_f16_stem_fwd run over the whole frame stream as ONE image of stream_rows rows whose frame 0
// is frame -stream_row0 of `feat`; rows 0 and H - 1 are computed in the launch.  Same coverage rules and the same results as the two calls;
// LAD_NOT_COVERED (nothing launched) otherwise.
extern "C" int lad_f16_block_fwd_stem_rows(const void *stream_act, int64_t stream_rows, int64_t stream_row0, const float *feat,
                                           int64_t frames_avail, const float *stem_weight, const float *stem_scale, const float *stem_shift,
                                           const void *wt1, const float *scale1, const float *shift1, const void *wt2, const float *scale2,
                                           const float *shift2, void *y, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(stream_act && feat && stem_weight && stem_scale && stem_shift && y && wt1 && wt2 && scale1 && shift1 && scale2 && shift2,
                "lad_f16_block_fwd_stem_rows: null buffer");
    LAD_REQUIRE(H >= 3 && W >= 1 && batch >= 1 && stream_row0 >= 0 && stream_row0 + batch - 1 + H <= stream_rows,
                "lad_f16_block_fwd_stem_rows: the strips reach past the stream (%lld rows)", (long long)stream_rows);
    const int Hp = H + 1, Wp = W + 1, img = Hp * Wp;
    if (img > 512 || W > 64 || blk_lds_bytes(img, Wp) > 160 * 1024 || batch < 256 || batch >= (1 << 30)) return LAD_NOT_COVERED;   // (W: a lane per column)
    const size_t lds = blk_lds_bytes(img, Wp);
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)block_f16_strip_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const StripMap sm{(const _Float16 *)stream_act, (long long)stream_row0, (unsigned)((1ull << 32) / (unsigned long long)Wp) + 1u, feat,
                      (long long)std::max<int64_t>(frames_avail, 0), stem_weight, stem_scale, stem_shift};
    hipLaunchKernelGGL(block_f16_strip_kernel<true>, dim3((unsigned)std::min<int64_t>(batch, 256)), dim3(BLK_THREADS), lds, (hipStream_t)stream,
                       (const _Float16 *)nullptr, (_Float16 *)y, (const _Float16 *)wt1, scale1, shift1, (const _Float16 *)wt2, scale2, shift2,
                       (int)batch, Hp, Wp, sm);
    return check_launch("block_f16_strip_kernel<mapped>");
}

