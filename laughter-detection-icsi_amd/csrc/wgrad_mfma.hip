// Weight-gradient of the 3x3 / 1x1 convolutions on the f32 matrix cores (gfx950).
//
// Replaces the weight/bias gradient autograd computes for nn.Conv2d in loss.backward() (train.py:289).
//   dW[tap][ci][co] = sum over rows q of  in[q + shift(tap)][ci] * dout[q][co]      (layout: lad_device.h)
// GEMM view: M = ci, N = co, K = rows of the whole batch (millions) -> split-K over persistent workgroups
// (two 256-thread workgroups per CU; the 4 wavefronts split the (ci-tile, co-tile, tap) output tiles).
// A workgroup walks tiles of TMW rows; per tile it stages the input rows (+halo) and the dout rows into
// LDS once (border rows are zero in HBM: layout invariant) and every wavefront accumulates its share of the 9*MT*NT 32x32 output
// tiles in registers across ALL its tiles.  At the end each workgroup writes one partial slab; a second
// small kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics) and emits the
// gradient in the reference's (cout, cin, kh, kw) parameter layout, plus the bias gradient (column sums of
// dout, accumulated on the side while the tile is in LDS).
#include "lad_common.h"
#include "lad_device.h"
#include "lad_b3.h"
#include "lad_bn_math.h"

#include <cstdlib>

namespace {
using namespace lad;

// Measured alternatives at 64->64, bs 512: 256 threads x 64-row tiles x 2 WGs/CU (this) 1.62 ms;
// 512 threads x 128-row tiles x 1 WG/CU 1.66 ms and slower on the small layers (fewer, longer-lived workgroups).
constexpr int THREADS = 256;     // 4 wavefronts; two workgroups per CU (2 waves per SIMD)
constexpr int NWAVES = THREADS / 64;
constexpr int TMW = 64;          // rows per tile
constexpr int MAX_GROUPS = 512;  // persistent workgroups (2 per CU)

template <int CIN, int COUT, int TAPS>
struct WgCfg {
    static constexpr int MT = (CIN + 31) / 32;
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int MN = MT * NT;          // 1, 2 or 4 distinct (mt, nt) pairs
    static constexpr int TSTRIDE = NWAVES / MN; // wavefronts that share one (mt, nt) split the taps
    static constexpr int TPW = (TAPS + TSTRIDE - 1) / TSTRIDE;  // accumulator tiles per wavefront
};

#ifdef LAD_STAMP
__device__ unsigned long long lad_wg_dbg[8 * 1024];  // diagnostic build only (tools/stamp_conv.py wgrad)
#define WG_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#else
#define WG_T(var)
#endif

constexpr int WG_PRE_IN = 10;  // float4 registers per thread carrying the next tile's input rows (bounds the image width)

template <int CIN, int COUT, int TAPS>
__global__ __launch_bounds__(THREADS, 2) void wgrad_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                           float *__restrict__ slabs, float *__restrict__ bias_slabs,
                                                           Geom g, int64_t n_tiles) {
    using C = WgCfg<CIN, COUT, TAPS>;
    constexpr int CI4 = CIN / 4, CO4 = COUT / 4;
    constexpr int BPARTS = THREADS / COUT;
    constexpr int NPD = (TMW * CO4 + THREADS - 1) / THREADS;
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMW + 2 * halo;
    float *in_s = smem;                                  // [nrows][CIN] (+32 slack)
    float *do_s = in_s + nrows * CIN + 32;               // [TMW][COUT]  (+32 slack)
    float *bred_s = do_s + TMW * COUT + 32;              // [BPARTS][COUT]

    const int mn = wave % C::MN;
    const int mt = mn / C::NT, nt = mn % C::NT;
    const int tap0 = (C::TSTRIDE == 1) ? 0 : wave / C::MN;  // one wave per (mt, nt): it owns every tap

    f32x16 acc[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    int aoff[C::TPW];
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int tap = tap0 + j * C::TSTRIDE;
        const int sh = (TAPS == 9 && tap < TAPS) ? ((tap / 3 - 1) * g.Wp + (tap % 3 - 1)) : 0;
        aoff[j] = (gk + halo + sh) * CIN + mt * 32 + i;
    }
    const int boff = gk * COUT + nt * 32 + i;
    float bsum = 0.0f;
    const int bco = tid % COUT, bpart = tid / COUT;
    const int nfi = nrows * CI4;

    // Software pipeline over this workgroup's tiles: the rows of tile t+1 travel HBM -> registers while the MFMAs of
    // tile t run from LDS, and move registers -> LDS between the two barriers that separate the tiles.
    u32x4 pin[WG_PRE_IN], pdo[NPD];
    // The prefetch is straight-line: per tile two buffer resources (wave-uniform, SGPRs) whose range check returns zeros
    // for rows before the first / after the last row of the tensor, and one 32-bit lane offset per load.  It must not
    // cost registers the 144 accumulator registers need, nor instructions: it is issued between the MFMAs.
    auto fetch = [&](int64_t tile) {
        const int64_t q0 = tile * TMW;
        const int64_t qb = q0 - halo;  // tensor row of staged row 0 (negative in the first tiles)
        const int64_t first = qb < 0 ? 0 : qb;
        const int skip = (int)(first - qb) * (CIN * 4);
        const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + first * CIN, (g.rows - first) * (CIN * 4));
#pragma unroll
        for (int u = 0; u < WG_PRE_IN; ++u) pin[u] = buf_load16(in_r, (u * THREADS + tid) * 16 - skip);
        const __amdgpu_buffer_rsrc_t do_r = make_rsrc(dout + q0 * COUT, (g.rows - q0) * (COUT * 4));
#pragma unroll
        for (int u = 0; u < NPD; ++u) pdo[u] = buf_load16(do_r, (u * THREADS + tid) * 16);
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) fetch(tile);
#ifdef LAD_STAMP
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
#endif
    for (; tile < n_tiles; tile += gridDim.x) {
        WG_T(t0)
        __syncthreads();  // previous tile's readers are done
        WG_T(t1)
#pragma unroll
        for (int u = 0; u < WG_PRE_IN; ++u) {
            const int f = u * THREADS + tid;  // registers past the tile land in the slack behind it
            reinterpret_cast<u32x4 *>(in_s)[f < nfi ? f : nfi + (tid & 7)] = pin[u];
        }
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const int f = u * THREADS + tid;
            if ((TMW * CO4) % THREADS == 0 || f < TMW * CO4) reinterpret_cast<u32x4 *>(do_s)[f] = pdo[u];
        }
        WG_T(t2)
        __syncthreads();
        WG_T(t3)
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        WG_T(t4)
        if (bias_slabs != nullptr) {
#pragma unroll 4
            for (int r = bpart; r < TMW; r += BPARTS) bsum += do_s[r * COUT + bco];
        }
#pragma unroll
        for (int k = 0; k < TMW; k += 2) {  // fully unrolled: every LDS address is a per-kernel base + an immediate
            const float b = do_s[k * COUT + boff];
#pragma unroll
            for (int j = 0; j < C::TPW; ++j) {
                if (C::TSTRIDE == 1 || tap0 + j * C::TSTRIDE < TAPS) {
                    const float a = in_s[k * CIN + aoff[j]];
                    acc[j] = mfma32(a, b, acc[j]);
                }
            }
        }
#ifdef LAD_STAMP
        {
            const unsigned long long t5 = __builtin_amdgcn_s_memtime();
            acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4;
        }
#endif
    }
#ifdef LAD_STAMP
    if (tid == 0 && blockIdx.x < 1024)
        for (int k = 0; k < 5; ++k) lad_wg_dbg[blockIdx.x * 8 + k] = acc_t[k];
#endif

    // ---- write this workgroup's partial slab: slab[wg][tap][ci][co] ------------------------------
    float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CIN * COUT);
#pragma unroll
    for (int j = 0; j < C::TPW; ++j) {
        const int tap = tap0 + j * C::TSTRIDE;
        if (tap < TAPS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = mt * 32 + acc_row(r, lane);
                const int co = nt * 32 + i;
                if (ci < CIN && co < COUT) slab[(tap * CIN + ci) * COUT + co] = acc[j][r];
            }
        }
    }
    if (bias_slabs != nullptr) {
        __syncthreads();
        bred_s[bpart * COUT + bco] = bsum;
        __syncthreads();
        if (tid < COUT) {
            float s = 0.0f;
            for (int p = 0; p < BPARTS; ++p) s += bred_s[p * COUT + tid];
            bias_slabs[(int64_t)blockIdx.x * COUT + tid] = s;
        }
    }
}

// Workgroups of a split-K launch: as many as there are tiles, up to MAX_GROUPS.  (Fewer, larger shares for small launches -- every
// workgroup writes a 147 KB slab that the slab reduction reads back -- measured slower at every setting, batch 32: 20.7 k -> 17.6 k
// segments/s at 12 tiles each: a workgroup's tile loop is latency-bound.  profiles/r04_conv_h2_experiments.log)
int groups_for(int64_t n_tiles) { return (int)std::max<int64_t>(1, std::min<int64_t>(MAX_GROUPS, n_tiles)); }

// ---------------------------------------------------------------------------------------------------------------------
// 64 x 64 x 9 weight gradient on the bf16 matrix cores with three-way split operands (lad_b3.h): the same sum
//   dW[tap][ci][co] = sum_q in[q + shift(tap)][ci] * dout[q][co]
// with K = rows.  Both MFMA operands are indexed [k][...] here (A[m = ci][k = row], B[k = row][n = co]) while the tensors
// are [row][channel]: the fragments come out of LDS through the transposing read ds_read_b64_tr_b16 (a group of 16 lanes
// reads 4 rows x 16 columns of 16-bit elements and receives them column-major: lane i gets column i of the 4 rows).
//
// A workgroup owns a CONTIGUOUS range of 32-row tiles and keeps a circular window of 128 input rows (32 + 2 x 46 halo
// fits) in LDS as three bf16 planes: per tile only the 32 new rows of `in` and the 32 rows of `dout` are loaded (fp32,
// 16-byte loads, one tile ahead, in registers during the MFMAs), split on the way to LDS -- the f32 kernel above re-stages
// 156 rows per 64.  Wave w owns the (ci tile, co tile) pair w and all nine taps: 9 accumulators of 32 x 32 in registers
// across all its tiles; per 16-row chunk a dout fragment (3 planes) feeds 9 x 6 MFMAs.  Slabs and their reduction are the
// f32 kernel's.
constexpr int B3_WIN = 128;   // rows of the circular input window (>= rows per tile + 2 * (W + 2))
// CH = 64: a tile is 32 rows and the four waves are the 2 x 2 tiles of 32 x 32 (ci, co) outputs.  CH = 32 (block2): one
// 32 x 32 output tile, so the four waves split K instead -- a tile is 64 rows, wave w takes rows 16 w .. 16 w + 15 -- and
// the four partial sums meet in LDS at the end (one slab per workgroup either way).
template <int CH>
struct WB3 {
    static constexpr int TK = CH == 64 ? 32 : 64;          // rows per tile
    static constexpr int ROWB = CH * 2;                    // bytes per row of a bf16 plane
    static constexpr int KPARTS = CH == 64 ? 1 : 4;        // waves that split the rows of a tile
    static constexpr int PLANE_IN = B3_WIN * ROWB;         // bytes of one plane of the window: [row][CH bf16]
    static constexpr int PLANE_DO = TK * ROWB;
    static constexpr int LPR = CH / 4;                     // threads per fp32 row (16-byte pieces)
    static constexpr int RPP = THREADS / LPR;              // rows one pass of the workgroup's threads covers
};

// The transposing read has no builtin: it is issued through inline asm, which the compiler's wait-count pass does not see.
// Reads are therefore issued in groups (read_frags) and waited for by hand (wait_frags: LDS operations of a wave return in
// order, so "at most N outstanding" retires everything issued before the last N); the wait statement takes the fragment
// registers as in/out operands so that no consumer can be scheduled above it.
// Bank swizzle of the [row][64 bf16] planes: a transposing read takes, per half-wave, 64 bytes of each of 4 consecutive rows;
// at 128 bytes per row, rows r and r + 2 sit on the same 64 banks (PMC: 45% of the LDS cycles of the first version were bank
// conflicts).  Swapping the two 64-byte halves of every other PAIR of rows puts any 4 consecutive rows on four different
// quarters of the banks; writers and readers apply the same XOR, pieces of 8 bytes stay whole.
// (64-byte rows -- 32 channels -- need none: four consecutive rows are 256 contiguous bytes.)
template <int CH>
__device__ __forceinline__ unsigned b3_swz(unsigned row) { return CH == 64 ? (row & 2u) << 5 : 0u; }

struct Frags {
    u32x2 lo[3], hi[3];   // planes 0..2: k = 8h + 0..3 | 8h + 4..7
};
__device__ __forceinline__ void read_frags(Frags &f, unsigned addr_lo, unsigned addr_hi, int plane_bytes) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo[p]) : "v"(addr_lo + p * plane_bytes));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi[p]) : "v"(addr_hi + p * plane_bytes));
    }
}
template <int N>
__device__ __forceinline__ void wait_frags(Frags &f) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]), "+v"(f.lo[2]), "+v"(f.hi[2])
                 : "n"(N));
}
__device__ __forceinline__ bf16x8 frag_of(const Frags &f, int p) {
    const u32x4 v = {f.lo[p].x, f.lo[p].y, f.hi[p].x, f.hi[p].y};
    return __builtin_bit_cast(bf16x8, v);
}

// INBN: `in` is the raw output of the previous convolution and in_coef the coefficients of the BatchNorm between the two;
// the activation relu(in * scale + shift) (0 on border rows) is formed while the rows are staged, bit for bit what
// bn_act_kernel would have written (conv_b3.hip applies the same to the forward convolution: the activation tensor
// never exists).
template <int CH, bool INBN>
__global__ __launch_bounds__(THREADS, 2) void wgrad_b3_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                             float *__restrict__ slabs, float *__restrict__ bias_slabs, Geom g,
                                                             int64_t n_tiles, int tiles_per_wg, const float *__restrict__ in_coef) {
    constexpr int TAPS = 9;
    using K = WB3<CH>;
    constexpr int B3_TK = K::TK, ROWB = K::ROWB, B3_PLANE_IN = K::PLANE_IN, B3_PLANE_DO = K::PLANE_DO, LPR = K::LPR, RPP = K::RPP;
    static_assert(2 * RPP == B3_TK, "a thread stages two pieces of each tensor per tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    unsigned char *in_s = smem_w;                        // [3 planes][B3_WIN rows][CH bf16]
    unsigned char *do_s = in_s + 3 * B3_PLANE_IN;        // [3 planes][B3_TK rows][CH bf16]
    float *bred_s = reinterpret_cast<float *>(do_s + 3 * B3_PLANE_DO);   // [RPP][CH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int halo = g.Wp + 1;
    const int mt = CH == 64 ? wave >> 1 : 0, nt = CH == 64 ? wave & 1 : 0;
    const int kpart = CH == 64 ? 0 : wave;               // which 16 rows of the tile this wave multiplies (CH = 32)

    f32x16 acc[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    const int64_t t_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t t_end = min(t_begin + tiles_per_wg, n_tiles);

    // registers -> LDS: 4 channels of one row become 8 bytes in each of the three planes
    auto put = [&](unsigned char *plane0, int plane_bytes, int slot, int c4, u32x4 v) {
        const float4 f = as_f4(v);
        unsigned a1, a2, a3, b1, b2, b3;
        split_pair(f.x, f.y, a1, a2, a3);
        split_pair(f.z, f.w, b1, b2, b3);
        unsigned char *dst = plane0 + slot * ROWB + ((c4 * 8) ^ b3_swz<CH>(slot));
        *reinterpret_cast<u32x2 *>(dst) = u32x2{a1, b1};
        *reinterpret_cast<u32x2 *>(dst + plane_bytes) = u32x2{a2, b2};
        *reinterpret_cast<u32x2 *>(dst + 2 * plane_bytes) = u32x2{a3, b3};
    };
    // a thread's two 16-byte pieces of a TK-row x CH-channel fp32 block: rows prow and RPP + prow, channels 4 * pc4
    const int prow = tid / LPR, pc4 = tid % LPR;
    f32x4 bn_sc = {0.f, 0.f, 0.f, 0.f}, bn_sh = bn_sc;
    if (INBN) {
        bn_sc = *reinterpret_cast<const f32x4 *>(in_coef + pc4 * 4);
        bn_sh = *reinterpret_cast<const f32x4 *>(in_coef + CH + pc4 * 4);
    }
    auto activate = [&](u32x4 v, int64_t row) {   // input piece of tensor row `row` -> the activation the convolution saw
        if (!INBN) return v;
        const bool keep = interior_row32((uint32_t)row, g);   // (a row before the tensor wraps to a huge index: not interior)
        float4 f = as_f4(v);
        f.x = keep ? fmaxf(fmaf(f.x, bn_sc.x, bn_sh.x), 0.f) : 0.f;
        f.y = keep ? fmaxf(fmaf(f.y, bn_sc.y, bn_sh.y), 0.f) : 0.f;
        f.z = keep ? fmaxf(fmaf(f.z, bn_sc.z, bn_sh.z), 0.f) : 0.f;
        f.w = keep ? fmaxf(fmaf(f.w, bn_sc.w, bn_sh.w), 0.f) : 0.f;
        return as_u4(f);
    };
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in, g.rows * (CH * 4));
    const __amdgpu_buffer_rsrc_t do_r = make_rsrc(dout, g.rows * (CH * 4));
    auto row_off = [&](int64_t row) {   // byte offset of this thread's piece of tensor row `row`; rows outside the tensor read as 0
        return (row >= 0 && row < g.rows) ? (int)(row * (CH * 4)) + pc4 * 16 : -1;
    };

    if (t_begin < t_end) {
        // ---- the window of the first tile: rows [q0 - halo, q0 + 32 + halo) ---------------------------------------------
        const int64_t q0 = t_begin * B3_TK;
        for (int r = prow; r < B3_TK + 2 * halo; r += RPP) {
            const int64_t row = q0 - halo + r;
            put(in_s, B3_PLANE_IN, (int)(row & (B3_WIN - 1)), pc4, activate(buf_load16(in_r, row_off(row)), row));
        }
    }
    u32x4 pin[2], pdo[2];
    auto fetch = [&](int64_t tile) {   // the 32 NEW input rows of `tile` (the top of its window) and its 32 dout rows
        const int64_t q0 = tile * B3_TK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pin[u] = buf_load16(in_r, row_off(q0 + halo + prow + RPP * u));
            pdo[u] = buf_load16(do_r, row_off(q0 + prow + RPP * u));
        }
    };
    if (t_begin < t_end) {
        const int64_t q0 = t_begin * B3_TK;   // the first tile's dout rows (its input rows are in the window already)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pin[u] = buf_load16(in_r, row_off(q0 + halo + prow + RPP * u));   // rewritten below: same values as the window's
            pdo[u] = buf_load16(do_r, row_off(q0 + prow + RPP * u));
        }
    }

    // per-lane pieces of the fragment addresses.  Transposing read: lane 4q + p of a 16-lane group supplies the address
    // of row q, columns 4p .. 4p+3 of the group's 4 x 16 block; the group (lane >> 4) = (k half h, column half).
    const int grp = lane >> 4, w16 = lane & 15;
    const int kh = grp >> 1;                                   // k = 8 kh + (0..3 | 4..7)
    const int qrow = w16 >> 2, colb = (w16 & 3) * 8;
    const unsigned a_col = (mt * 32 + (grp & 1) * 16) * 2 + colb;     // byte within a row of ROWB bytes
    const unsigned b_col = (nt * 32 + (grp & 1) * 16) * 2 + colb;
    // (dout rows of a fragment: kc * 16 + 8 kh + 4 blk + qrow -- the swizzle bit is bit 1 of qrow)
    const unsigned b_lane = lds_addr(do_s) + (8 * kh + qrow) * ROWB + (b_col ^ b3_swz<CH>(qrow));
    const unsigned a_plane0 = lds_addr(in_s);

    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        const int64_t q0 = tile * B3_TK;
        __syncthreads();  // previous tile's readers are done
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t row = q0 + halo + prow + RPP * u;
            put(in_s, B3_PLANE_IN, (int)(row & (B3_WIN - 1)), pc4, activate(pin[u], row));
            put(do_s, B3_PLANE_DO, prow + RPP * u, pc4, pdo[u]);
            bsum += __builtin_bit_cast(f32x4, pdo[u]);
        }
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
        // row slot (x 128 bytes) of this lane's first A row for tap offset 0 and chunk 0: q0 + 8 kh + qrow
        const unsigned a_row0 = (unsigned)((q0 + 8 * kh + qrow) & (B3_WIN - 1));
        // 16-row chunks of the tile this wave multiplies: all of them (CH = 64), or chunk `kpart` only (CH = 32)
        constexpr int NKC = (B3_TK / 16) / K::KPARTS;
#pragma unroll
        for (int kci = 0; kci < NKC; ++kci) {
            const int kc = K::KPARTS == 1 ? kci : kpart;
            auto a_addr = [&](int tap, int blk) {   // LDS byte address of this lane's piece of rows (chunk, block of 4) shifted by the tap
                const int sh = (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
                const unsigned slot = (a_row0 + (unsigned)(kc * 16 + 4 * blk + sh + B3_WIN)) & (B3_WIN - 1);
                return a_plane0 + slot * ROWB + (a_col ^ b3_swz<CH>(slot));
            };
            Frags fb, fa[2];
            read_frags(fb, b_lane + (kc * 16) * ROWB, b_lane + (kc * 16 + 4) * ROWB, B3_PLANE_DO);
            read_frags(fa[0], a_addr(0, 0), a_addr(0, 1), B3_PLANE_IN);
            wait_frags<6>(fb);   // the dout fragments (issued first) have arrived
            const bf16x8 b0 = frag_of(fb, 0), b1 = frag_of(fb, 1), b2 = frag_of(fb, 2);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                Frags &cur = fa[tap & 1];
                if (tap + 1 < TAPS) {
                    read_frags(fa[(tap + 1) & 1], a_addr(tap + 1, 0), a_addr(tap + 1, 1), B3_PLANE_IN);   // one tap ahead of the MFMAs
                    wait_frags<6>(cur);
                } else {
                    wait_frags<0>(cur);
                }
                const bf16x8 a0 = frag_of(cur, 0), a1 = frag_of(cur, 1), a2 = frag_of(cur, 2);
                // smallest terms first
                acc[tap] = mfma_bf16(a0, b2, acc[tap]);
                acc[tap] = mfma_bf16(a1, b1, acc[tap]);
                acc[tap] = mfma_bf16(a2, b0, acc[tap]);
                acc[tap] = mfma_bf16(a0, b1, acc[tap]);
                acc[tap] = mfma_bf16(a1, b0, acc[tap]);
                acc[tap] = mfma_bf16(a0, b0, acc[tap]);
            }
        }
    }

    // ---- this workgroup's partial slab: slab[wg][tap][ci][co] (layout of wgrad_kernel; summed by slab_reduce.hip) -----
    float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CH * CH);
    const int i = lane & 31;
    if constexpr (K::KPARTS == 1) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(tap * CH + mt * 32 + acc_row(r, lane)) * CH + nt * 32 + i] = acc[tap][r];
    } else {
        // CH = 32: the four waves hold partial sums of the SAME 32 x 32 tiles (they split K): add them through LDS (the
        // window is free now), two taps per round, in wave order -- one slab per workgroup, as for 64 channels
        static_assert(CH == 32 && K::KPARTS == 4, "the in-workgroup sum is written for 4 waves x 32 x 32");
        float *sum_s = reinterpret_cast<float *>(smem_w);   // [2 taps][4 waves][32 x 32]
#pragma unroll
        for (int t0 = 0; t0 < TAPS; t0 += 2) {
            __syncthreads();   // the window (first round) / the previous round's readers are done
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (t0 + u < TAPS) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum_s[((u * 4 + wave) * 32 + acc_row(r, lane)) * 32 + i] = acc[t0 + u][r];
                }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (t0 + u < TAPS) {
                    const f32x4 *src = reinterpret_cast<const f32x4 *>(sum_s + u * 4 * 1024) + tid;   // 256 threads x 4 floats = one tile
                    const f32x4 v = ((src[0] + src[256]) + src[512]) + src[768];
                    *reinterpret_cast<f32x4 *>(slab + (t0 + u) * (CH * CH) + tid * 4) = v;
                }
        }
    }
    if (bias_slabs != nullptr) {
        __syncthreads();
        *reinterpret_cast<f32x4 *>(bred_s + prow * CH + pc4 * 4) = bsum;
        __syncthreads();
        if (tid < CH) {
            float s = 0.0f;
            for (int pp = 0; pp < RPP; ++pp) s += bred_s[pp * CH + tid];
            bias_slabs[(int64_t)blockIdx.x * CH + tid] = s;
        }
    }
}

// =====================================================================================================================
// Round 3: the 64 x 64 x 9 split-operand weight gradient on v_mfma_f32_16x16x32_bf16, "wgrad_b3x".
//
// Same sum, same circular window, same slabs as wgrad_b3_kernel.  What changed, and why:
//   * MFMA shape 16x16x32 (conv_b3.hip, tools/experiments/mfma_shape.hip: the power-bound chip holds a higher clock under
//     it).  K = 32 = the WHOLE 32-row tile per MFMA; a wave's 32 x 32 (ci, co) share of a tap is 2 x 2 tiles x 6 plane
//     products = 24 MFMAs.
//   * Fragments through the builtin __builtin_amdgcn_ds_read_tr16_b64_v4i16 instead of inline asm: the round-2 kernel
//     executed 2 v_mov per MFMA to assemble 128-bit operands out of separately allocated 64-bit asm outputs, carried its
//     own wait counts and could not use the offset field; with the builtin the compiler allocates the two halves of a
//     fragment in adjacent registers, places the waits itself and folds planes into immediates (4.6 -> ~1 vector
//     instructions per MFMA).
//   * The rows of the tile are dealt to the K index so that every half-wave reads EIGHT CONSECUTIVE rows:
//     k = 8 g + j  <->  row 4 g + j (j < 4), 16 + 4 g + (j - 4) (j >= 4), g = lane >> 4 -- for both operands, a sum over K
//     does not care -- and the planes are swizzled by ((row >> 1) & 3) x 32 bytes: eight consecutive rows x 32 bytes then
//     tile the 64 banks for any window position (the round-2 swizzle served 4 + 4 rows eight apart).
//   * Offsets are relative to the workgroup's own first row (64-bit base, 32-bit lane offsets): no 2 GiB limit.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned b3x_swz(unsigned row) { return (row & 6u) << 4; }   // ((row >> 1) & 3) * 32 bytes

__device__ __forceinline__ bf16x8 read_tr_frag(unsigned addr_lo, unsigned addr_hi, int imm) {
    typedef __attribute__((address_space(3))) s16x4 *lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr_lo + imm));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr_hi + imm));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <bool INBN>
__global__ __launch_bounds__(THREADS, 2) void wgrad_b3x_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                              float *__restrict__ slabs, float *__restrict__ bias_slabs, Geom g,
                                                              int64_t n_tiles, int tiles_per_wg, const float *__restrict__ in_coef) {
    constexpr int TAPS = 9, CH = 64, TK = 32, ROWB = CH * 2;
    constexpr int PLANE_IN = B3_WIN * ROWB, PLANE_DO = TK * ROWB, LPR = CH / 4, RPP = THREADS / LPR;
    static_assert(2 * RPP == TK, "a thread stages two pieces of each tensor per tile");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem_w[];
    unsigned char *in_s = smem_w;                        // [3 planes][B3_WIN rows][64 bf16]   (128-byte aligned: addresses are OR-ed)
    unsigned char *do_s = in_s + 3 * PLANE_IN;           // [3 planes][32 rows][64 bf16]
    float *bred_s = reinterpret_cast<float *>(do_s + 3 * PLANE_DO);   // [RPP][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = g.Wp + 1;
    const int mtw = wave >> 1, ntw = wave & 1;           // the wave's 32 x 32 (ci, co) share

    f32x4 acc[TAPS][2][2];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    const int64_t t_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t t_end = min(t_begin + tiles_per_wg, n_tiles);
    if (t_begin >= t_end) {   // (whole-workgroup) nothing to do: an all-zero slab
        float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CH * CH);
        for (int e = tid; e < TAPS * CH * CH; e += THREADS) slab[e] = 0.f;
        if (bias_slabs != nullptr && tid < CH) bias_slabs[(int64_t)blockIdx.x * CH + tid] = 0.f;
        return;
    }

    auto put = [&](unsigned char *plane0, int plane_bytes, int slot, int c4, u32x4 v) {
        const float4 f = as_f4(v);
        unsigned a1, a2, a3, b1, b2, b3;
        split_pair(f.x, f.y, a1, a2, a3);
        split_pair(f.z, f.w, b1, b2, b3);
        unsigned char *dst = plane0 + slot * ROWB + ((c4 * 8) ^ b3x_swz(slot));
        *reinterpret_cast<u32x2 *>(dst) = u32x2{a1, b1};
        *reinterpret_cast<u32x2 *>(dst + plane_bytes) = u32x2{a2, b2};
        *reinterpret_cast<u32x2 *>(dst + 2 * plane_bytes) = u32x2{a3, b3};
    };
    const int prow = tid / LPR, pc4 = tid % LPR;
    f32x4 bn_sc = {0.f, 0.f, 0.f, 0.f}, bn_sh = bn_sc;
    if (INBN) {
        bn_sc = *reinterpret_cast<const f32x4 *>(in_coef + pc4 * 4);
        bn_sh = *reinterpret_cast<const f32x4 *>(in_coef + CH + pc4 * 4);
    }
    auto activate = [&](u32x4 v, int64_t row) {
        if (!INBN) return v;
        const bool keep = row >= 0 && interior_row32((uint32_t)row, g);
        float4 f = as_f4(v);
        f.x = keep ? fmaxf(fmaf(f.x, bn_sc.x, bn_sh.x), 0.f) : 0.f;
        f.y = keep ? fmaxf(fmaf(f.y, bn_sc.y, bn_sh.y), 0.f) : 0.f;
        f.z = keep ? fmaxf(fmaf(f.z, bn_sc.z, bn_sh.z), 0.f) : 0.f;
        f.w = keep ? fmaxf(fmaf(f.w, bn_sc.w, bn_sh.w), 0.f) : 0.f;
        return as_u4(f);
    };
    // buffer resources over the rows this workgroup can touch, [r_first, r_last): offsets relative to r_first stay far below 2^31
    const int64_t r_first = max((int64_t)0, t_begin * TK - halo);
    const int64_t r_last = min(g.rows, t_end * TK + halo);
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + r_first * CH, (r_last - r_first) * (CH * 4));
    const __amdgpu_buffer_rsrc_t do_r = make_rsrc(dout + r_first * CH, (r_last - r_first) * (CH * 4));
    auto row_off = [&](int64_t row) {   // rows outside [r_first, r_last) read as 0 (before the tensor / past its end)
        return (row >= r_first && row < r_last) ? (int)((row - r_first) * (CH * 4)) + pc4 * 16 : -1;
    };

    {   // the window of the first tile: rows [q0 - halo, q0 + 32 + halo)
        const int64_t q0 = t_begin * TK;
        for (int r = prow; r < TK + 2 * halo; r += RPP) {
            const int64_t row = q0 - halo + r;
            put(in_s, PLANE_IN, (int)(row & (B3_WIN - 1)), pc4, activate(buf_load16(in_r, row_off(row)), row));
        }
    }
    u32x4 pin[2], pdo[2];
    auto fetch = [&](int64_t tile) {
        const int64_t q0 = tile * TK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pin[u] = buf_load16(in_r, row_off(q0 + halo + prow + RPP * u));
            pdo[u] = buf_load16(do_r, row_off(q0 + prow + RPP * u));
        }
    };
    fetch(t_begin);

    // fragment addressing: lane 4 q + p of the 16-lane group gr supplies row 4 gr + q (lo) / 16 + 4 gr + q (hi), bytes 8 p .. 8 p + 7
    // of the tile's 32-byte column block; the transposed result is column (lane & 15) of those four rows.
    const int gr = lane >> 4, w16 = lane & 15;
    const unsigned lrow = 4 * gr + (w16 >> 2), colb = (w16 & 3) * 8;
    const unsigned a_col = mtw * 64 + colb, b_col = ntw * 64 + colb;   // first 16-column tile; the second is the address ^ 32
    const unsigned a_base = lds_addr(in_s), b_base = lds_addr(do_s);
    const unsigned b_lo = b_base + lrow * ROWB + (b_col ^ b3x_swz(lrow));
    const unsigned b_hi = b_base + (lrow + 16) * ROWB + (b_col ^ b3x_swz(lrow + 16));

    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        const int64_t q0 = tile * TK;
        __syncthreads();  // previous tile's readers are done
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t row = q0 + halo + prow + RPP * u;
            put(in_s, PLANE_IN, (int)(row & (B3_WIN - 1)), pc4, activate(pin[u], row));
            put(do_s, PLANE_DO, prow + RPP * u, pc4, pdo[u]);
            bsum += __builtin_bit_cast(f32x4, pdo[u]);
        }
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
        bf16x8 b[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[nt][p] = read_tr_frag(b_lo ^ (nt * 32), b_hi ^ (nt * 32), p * PLANE_DO);
        const unsigned s0 = (unsigned)q0 + lrow + B3_WIN;   // (window slot of this lane's first row) + B3_WIN, before the tap shift
        // The taps are software-pipelined by halves: while the 12 MFMAs of the wave's first 16 input channels (mt = 0) run,
        // the fragments of its second 16 (mt = 1) are in flight, and during those 12 the mt = 0 fragments of the NEXT tap
        // (round-3 PMC of the unpipelined form: every tap began with ~130 cycles of LDS latency in front of 384 cycles of
        // MFMAs, two waves per SIMD).  Per accumulator the six plane products still come smallest first.
        auto a_frags = [&](int tap, int mt, bf16x8 (&a)[3]) {
            const int sh = (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
            const unsigned slot_lo = (s0 + (unsigned)sh) & (B3_WIN - 1), slot_hi = (slot_lo + 16) & (B3_WIN - 1);
            const unsigned cs = (a_col ^ b3x_swz(slot_lo)) ^ (mt * 32);   // (slot_hi has the same bits 1..2)
            const unsigned a_lo = a_base + ((slot_lo << 7) | cs), a_hi = a_base + ((slot_hi << 7) | cs);
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = read_tr_frag(a_lo, a_hi, p * PLANE_IN);
        };
        auto mfmas = [&](int tap, int mt, const bf16x8 (&a)[3]) {
#define LAD_WB3X_TERM(pa, pb) \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) acc[tap][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa], b[nt][pb], acc[tap][mt][nt], 0, 0, 0);
            LAD_WB3X_TERM(0, 2)
            LAD_WB3X_TERM(1, 1)
            LAD_WB3X_TERM(2, 0)
            LAD_WB3X_TERM(0, 1)
            LAD_WB3X_TERM(1, 0)
            LAD_WB3X_TERM(0, 0)
#undef LAD_WB3X_TERM
        };
        bf16x8 a0[3], a1[3], an[3];
        a_frags(0, 0, a0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            a_frags(tap, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(tap, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) a_frags(tap + 1, 0, an);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(tap, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 3; ++p) a0[p] = an[p];
        }
    }

    // ---- this workgroup's partial slab: slab[wg][tap][ci][co]; D register j of lane l of a tile: row (ci) 4 (l >> 4) + j, column (co) l & 15
    float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CH * CH);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    slab[(tap * CH + mtw * 32 + mt * 16 + 4 * gr + j) * CH + ntw * 32 + nt * 16 + w16] = acc[tap][mt][nt][j];
    if (bias_slabs != nullptr) {
        __syncthreads();
        *reinterpret_cast<f32x4 *>(bred_s + prow * CH + pc4 * 4) = bsum;
        __syncthreads();
        if (tid < CH) {
            float s = 0.0f;
            for (int pp = 0; pp < RPP; ++pp) s += bred_s[pp * CH + tid];
            bias_slabs[(int64_t)blockIdx.x * CH + tid] = s;
        }
    }
}


// =====================================================================================================================
// Round 4: the 64 x 64 x 9 weight gradient on TWO f16 planes per operand, "wgrad_h2" (arithmetic: conv_h2.hip).
//
// wgrad_b3x_kernel with half the matrix instructions: K = 32 = the 32-row tile, a wave's 32 x 32 (ci, co) share of a tap is
// 2 x 2 tiles x 3 plane products (a1 b2, a2 b1, a1 b1) = 12 MFMAs.  Same circular window, same transposing reads, same row
// dealing and swizzle, same slabs.
//
// Scales.  K is the ROW index here and the input window persists from tile to tile, so an operand's power-of-two scale must
// hold for every row of a wave's MFMA: a workgroup keeps ONE running exponent per operand over its contiguous row range
// (acc = 2^(e_in + e_do) x sum).  dout is split afresh per tile: when a tile's maximum no longer fits, e_do drops (with one
// binade of headroom) and the accumulators are re-based (ldexp, exact).  For `in`, whose earlier rows are still in the
// window, the same event also re-stages the window's 2 x halo older rows from HBM with the new exponent (three binades of
// headroom make it rare: an activation tensor's tile maxima vary by far less than 8x).  Exponents only ever decrease along a
// workgroup's range, so what a later, much smaller tile loses is below 2^-37 of the running maximum -- and of the sums it joins.
typedef short s16x8w __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 read_tr_frag_h(unsigned addr_lo, unsigned addr_hi, int imm) {
    typedef __attribute__((address_space(3))) s16x4 *lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr_lo + imm));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(size_t)(addr_hi + imm));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
typedef _Float16 wh16x2 __attribute__((ext_vector_type(2)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair_w(float a, float b, unsigned &p1, unsigned &p2) {
    const wf32x2 v = {a, b};
    const wh16x2 h = __builtin_convertvector(v, wh16x2);   // round to nearest even
    p1 = __builtin_bit_cast(unsigned, h);
    const wf32x2 r = {a - (float)h.x, b - (float)h.y};     // exact
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, wh16x2));
}
__device__ __forceinline__ int scale_exp_w(float amax) {   // amax * 2^k in [2^14, 2^15) (conv_h2.hip: scale_exp)
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
    const int k = 141 - e;
    return k > 100 ? 100 : k;
}
__device__ __forceinline__ float pow2f_w(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }
__device__ __forceinline__ float wave_max64_w(float v) { return wave_max64_dpp(v); }
__device__ __forceinline__ float amax4(u32x4 v) {
    const float4 f = as_f4(v);
    return fmaxf(fmaxf(fabsf(f.x), fabsf(f.y)), fmaxf(fabsf(f.z), fabsf(f.w)));
}

// DOBN != 0: `dout` is the gradient dy arriving at a BatchNorm (+ReLU) and the rows the weight gradient pairs with the input
// are that BatchNorm's INPUT gradient dc = k1 (dz - k2 - xhat k3) (bn_bwd_apply_kernel<0>, same arithmetic: lad_bn_math.h), formed
// while a tile's rows are staged -- dz = dy x [ReLU decision], recomputed from the BatchNorm's input x (DOBN = 2) or taken from
// the sign bits of the block's output (DOBN = 3).  Every dy row belongs to exactly one tile of one workgroup, so the kernel also
// WRITES dc (zeros on border and tail rows) for the data-gradient launch that follows: the element-wise pass over three
// activation-sized tensors that bn_bwd_apply_kernel was (0.31 ms per 64-channel BatchNorm at batch 512) rides under these MFMAs.
struct WgBnBwd {
    const float *x;                   // the BatchNorm's input (the convolution output it normalised)
    const unsigned long long *bits;   // sign bits of the block output (DOBN = 3)
    const float *coef;                // float[6][64]: forward coefficients (lad_bn_finalize)
    const float *bcoef;               // float[8][64]: backward coefficients (bn_bwd_finalize_kernel)
    float *dc;                        // out: the BatchNorm's input gradient
};

template <bool INBN, int DOBN>
__global__ __launch_bounds__(THREADS, 2) void wgrad_h2_kernel(const float *__restrict__ in, const float *__restrict__ dout,
                                                             float *__restrict__ slabs, float *__restrict__ bias_slabs, Geom g,
                                                             int64_t n_tiles, int tiles_per_wg, const float *__restrict__ in_coef,
                                                             WgBnBwd bb) {
    constexpr int TAPS = 9, CH = 64, TK = 32, ROWB = CH * 2;
    constexpr int NCF = 11;   // coefficient rows kept in LDS: scale, shift, mean, istd, mean_lo, istd_lo, k1, k2, k3, k2_lo, k3_lo
    constexpr int PLANE_IN = B3_WIN * ROWB, PLANE_DO = TK * ROWB, LPR = CH / 4, RPP = THREADS / LPR;
    constexpr int HEAD_IN = 3, HEAD_DO = 1;                                 // binades of headroom kept when an exponent is (re)chosen
    constexpr int NWR = (2 * 47 + RPP - 1) / RPP;                           // passes that cover the window's 2 x halo older rows (W <= 46)
    static_assert(2 * RPP == TK, "a thread stages two pieces of each tensor per tile");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem_w[];
    unsigned char *in_s = smem_w;                        // [2 planes][B3_WIN rows][64 f16]
    unsigned char *do_s = in_s + 2 * PLANE_IN;           // [2 planes][32 rows][64 f16]
    float *bred_s = reinterpret_cast<float *>(do_s + 2 * PLANE_DO);   // [RPP][64]
    float *smx = bred_s + RPP * CH;                      // [3][4]: the waves' maxima (new input rows, dout rows, window rows)
    float *ci_s = smx + 12;                              // [2][64]: scale, shift of the input BatchNorm (INBN)
    float *cf_s = ci_s + 2 * CH;                         // [NCF][64] (DOBN)
    // (DOBN) [dy: 8 KB][x: 8 KB][bits: 4 waves x 256 B], 256-byte aligned
    unsigned char *st_s = smem_w + ((reinterpret_cast<unsigned char *>(cf_s + NCF * CH) - smem_w + 255) & ~255);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo = g.Wp + 1;
    const int mtw = wave >> 1, ntw = wave & 1;
    if (INBN && tid < 2 * CH) ci_s[tid] = in_coef[tid];
    if (DOBN != 0) {
        for (int i = tid; i < NCF * CH; i += THREADS) {
            const int k = i / CH, c = i - k * CH;
            cf_s[i] = k < 6 ? bb.coef[k * CH + c] : bb.bcoef[(k == 6 ? 0 : k == 7 ? 1 : k == 8 ? 2 : k == 9 ? 4 : 6) * CH + c];
        }
    }

    f32x4 acc[TAPS][2][2];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // (row and tile indices fit 32 bits: the launcher checks rows < 2^31; per-lane 64-bit arithmetic costs register pairs)
    const int t_begin = (int)blockIdx.x * tiles_per_wg;
    const int t_end = (int)min((int64_t)t_begin + tiles_per_wg, n_tiles);
    if (t_begin >= t_end) {
        float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CH * CH);
        for (int e = tid; e < TAPS * CH * CH; e += THREADS) slab[e] = 0.f;
        if (bias_slabs != nullptr && tid < CH) bias_slabs[(int64_t)blockIdx.x * CH + tid] = 0.f;
        return;
    }

    auto put = [&](unsigned char *plane0, int plane_bytes, int slot, int c4, u32x4 v, float scl) {
        const float4 f = as_f4(v);
        unsigned a1, a2, b1, b2;
        split2_pair_w(f.x * scl, f.y * scl, a1, a2);
        split2_pair_w(f.z * scl, f.w * scl, b1, b2);
        unsigned char *dst = plane0 + slot * ROWB + ((c4 * 8) ^ b3x_swz(slot));
        *reinterpret_cast<u32x2 *>(dst) = u32x2{a1, b1};
        *reinterpret_cast<u32x2 *>(dst + plane_bytes) = u32x2{a2, b2};
    };
    const int prow = tid / LPR, pc4 = tid % LPR;
    auto activate = [&](u32x4 v, int row) {
        if (!INBN) return v;
        const bool keep = row >= 0 && interior_row32((uint32_t)row, g);
        float4 f = as_f4(v);
        const f32x4 bn_sc = *reinterpret_cast<const f32x4 *>(ci_s + pc4 * 4), bn_sh = *reinterpret_cast<const f32x4 *>(ci_s + CH + pc4 * 4);
        f.x = keep ? fmaxf(fmaf(f.x, bn_sc.x, bn_sh.x), 0.f) : 0.f;
        f.y = keep ? fmaxf(fmaf(f.y, bn_sc.y, bn_sh.y), 0.f) : 0.f;
        f.z = keep ? fmaxf(fmaf(f.z, bn_sc.z, bn_sh.z), 0.f) : 0.f;
        f.w = keep ? fmaxf(fmaf(f.w, bn_sc.w, bn_sh.w), 0.f) : 0.f;
        return as_u4(f);
    };
    const int n_rows = (int)g.rows;
    const int r_first = max(0, t_begin * TK - halo);
    const int r_last = (int)min(g.rows, (int64_t)t_end * TK + halo);
    const int64_t span_bytes = (int64_t)(r_last - r_first) * (CH * 4);
    const __amdgpu_buffer_rsrc_t in_r = make_rsrc(in + (int64_t)r_first * CH, span_bytes);
    const __amdgpu_buffer_rsrc_t do_r = make_rsrc(dout + (int64_t)r_first * CH, span_bytes);
    auto row_off = [&](int row) {
        return (row >= r_first && row < r_last) ? (row - r_first) * (CH * 4) + pc4 * 16 : -1;
    };

    const __amdgpu_buffer_rsrc_t dc_r = make_rsrc(DOBN != 0 ? bb.dc + (int64_t)r_first * CH : nullptr, DOBN != 0 ? span_bytes : 0);

    // The next tile's rows travel in registers (pin, pdo) while the MFMAs run -- except with DOBN, whose three extra streams
    // (dy, the BatchNorm input, the sign bits) would push the kernel past its 256 registers: those go by LDS-DMA into st_s, each
    // thread's own two pieces of dy and x and, per wave, the tile's 32 sign-bit words, and are read back at the top of the tile.
    // Rows past the tensor's end are clamped to its last row (a tail row: its result is discarded by the interior test).
    u32x4 pin[2], pdo[2];
    auto fetch = [&](int tile) {
        const int q0 = tile * TK;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pin[u] = buf_load16(in_r, row_off(q0 + halo + prow + RPP * u));
            const int row = q0 + prow + RPP * u;
            if (DOBN == 0) {
                pdo[u] = buf_load16(do_r, row_off(row));
            } else {
                const unsigned off = (unsigned)(min(row, n_rows - 1) - r_first) * (CH * 4) + pc4 * 16;
                dma16s(dout + (int64_t)r_first * CH, off, lds_addr(st_s + (u * 4 + wave) * 1024));
                dma16s(bb.x + (int64_t)r_first * CH, off, lds_addr(st_s + 8192 + (u * 4 + wave) * 1024));
            }
        }
        if (DOBN == 3 && lane < 16) {
            const int pair = min(q0 / 2 + lane, (n_rows - 2) / 2);   // (>= r_first / 2: the offset below is not negative)
            dma16s(bb.bits + r_first, (unsigned)(2 * pair - r_first) * 8, lds_addr(st_s + 16384 + wave * 256));
        }
    };
    fetch(t_begin);
    if (INBN || DOBN != 0) __syncthreads();   // the coefficients are in ci_s / cf_s
    // dy rows -> the BatchNorm's input gradient (bn_bwd_apply_kernel<0>): written out, and what the tile is staged from.  Both
    // of the thread's pieces at once, coefficient by coefficient (a row of cf_s is read, used for both and dropped: held all at
    // once the eleven float4 do not fit next to the 144 accumulators); per element the operations and their order are those of
    // mask_from_x / mask_from_bits, xhat1 and bn_dx1 (lad_bn_math.h).
    auto bn_backward2 = [&](int q0) {
        auto cf = [&](int k) { return *reinterpret_cast<const float4 *>(cf_s + k * CH + pc4 * 4); };
        float4 d[2], t[2];
        bool keep[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            d[u] = *reinterpret_cast<const float4 *>(st_s + (u * 4 + wave) * 1024 + lane * 16);
            t[u] = *reinterpret_cast<const float4 *>(st_s + 8192 + (u * 4 + wave) * 1024 + lane * 16);
            keep[u] = interior_row32((uint32_t)(q0 + prow + RPP * u), g);
        }
        if (DOBN == 2) {
            const float4 sc = cf(0), sh = cf(1);
#pragma unroll
            for (int u = 0; u < 2; ++u) d[u] = mask_from_x(d[u], t[u], sc, sh);
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                d[u] = mask_from_bits(d[u], *reinterpret_cast<const unsigned long long *>(st_s + 16384 + wave * 256 + (prow + RPP * u) * 8), pc4);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 m = cf(2), ml = cf(4);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                t[u].x = (t[u].x - m.x) - ml.x; t[u].y = (t[u].y - m.y) - ml.y;
                t[u].z = (t[u].z - m.z) - ml.z; t[u].w = (t[u].w - m.w) - ml.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 is = cf(3), isl = cf(5);
#pragma unroll
            for (int u = 0; u < 2; ++u) {   // xhat
                t[u].x = fmaf(t[u].x, is.x, t[u].x * isl.x); t[u].y = fmaf(t[u].y, is.y, t[u].y * isl.y);
                t[u].z = fmaf(t[u].z, is.z, t[u].z * isl.z); t[u].w = fmaf(t[u].w, is.w, t[u].w * isl.w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 k2 = cf(7), k2l = cf(9);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d[u].x = (d[u].x - k2.x) - k2l.x; d[u].y = (d[u].y - k2.y) - k2l.y;
                d[u].z = (d[u].z - k2.z) - k2l.z; d[u].w = (d[u].w - k2.w) - k2l.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 k3 = cf(8), k3l = cf(10);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d[u].x = fmaf(-t[u].x, k3l.x, fmaf(-t[u].x, k3.x, d[u].x)); d[u].y = fmaf(-t[u].y, k3l.y, fmaf(-t[u].y, k3.y, d[u].y));
                d[u].z = fmaf(-t[u].z, k3l.z, fmaf(-t[u].z, k3.z, d[u].z)); d[u].w = fmaf(-t[u].w, k3l.w, fmaf(-t[u].w, k3.w, d[u].w));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const float4 k1 = cf(6);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d[u].x = keep[u] ? k1.x * d[u].x : 0.f; d[u].y = keep[u] ? k1.y * d[u].y : 0.f;
                d[u].z = keep[u] ? k1.z * d[u].z : 0.f; d[u].w = keep[u] ? k1.w * d[u].w : 0.f;
                pdo[u] = as_u4(d[u]);
                buf_store16(pdo[u], dc_r, row_off(q0 + prow + RPP * u));
            }
        }
    };

    const int gr = lane >> 4, w16 = lane & 15;
    const unsigned lrow = 4 * gr + (w16 >> 2), colb = (w16 & 3) * 8;
    const unsigned a_col = mtw * 64 + colb, b_col = ntw * 64 + colb;
    const unsigned a_base = lds_addr(in_s), b_base = lds_addr(do_s);
    const unsigned b_lo = b_base + lrow * ROWB + (b_col ^ b3x_swz(lrow));
    const unsigned b_hi = b_base + (lrow + 16) * ROWB + (b_col ^ b3x_swz(lrow + 16));

    int e_in = 0, e_do = 0;      // exponents of the planes in LDS: plane value = tensor value x 2^e
    bool have = false;           // (false until the first tile has chosen them)
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int q0 = tile * TK;
        // the new rows as the convolution sees them, and their largest magnitudes
        float m_in = 0.f, m_do = 0.f;
        if (DOBN != 0) {
            dma_wait_all();   // this tile's rows have landed in st_s (every thread reads what its own wave requested)
            bn_backward2(q0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pin[u] = activate(pin[u], q0 + halo + prow + RPP * u);
            m_in = fmaxf(m_in, amax4(pin[u]));
            m_do = fmaxf(m_do, amax4(pdo[u]));
        }
        m_in = wave_max64_w(m_in);
        m_do = wave_max64_w(m_do);
        if (lane == 0) {
            smx[wave] = m_in;
            smx[4 + wave] = m_do;
        }
        __syncthreads();  // previous tile's readers are done; the maxima are visible
        m_in = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        m_do = fmaxf(fmaxf(smx[4], smx[5]), fmaxf(smx[6], smx[7]));
        int d = 0;
        if (!have || scale_exp_w(m_in) < e_in) {   // (workgroup-uniform) the window's older rows are re-staged with the new exponent
            u32x4 wr[NWR];
            float m_w = 0.f;
#pragma unroll
            for (int u = 0; u < NWR; ++u) {
                const int r = prow + RPP * u;
                const int row = q0 - halo + r;
                wr[u] = activate(buf_load16(in_r, r < 2 * halo ? row_off(row) : -1), row);
                if (r >= 2 * halo) wr[u] = u32x4{0u, 0u, 0u, 0u};   // (not a window row: an activated zero is relu(shift), not 0)
                m_w = fmaxf(m_w, amax4(wr[u]));
            }
            m_w = wave_max64_w(m_w);
            if (lane == 0) smx[8 + wave] = m_w;
            __syncthreads();
            m_w = fmaxf(fmaxf(smx[8], smx[9]), fmaxf(smx[10], smx[11]));
            const int e_new = scale_exp_w(fmaxf(m_in, m_w)) - HEAD_IN;
            d += have ? e_new - e_in : 0;
            e_in = e_new;
            const float scl = pow2f_w(e_in);
#pragma unroll
            for (int u = 0; u < NWR; ++u) {
                const int r = prow + RPP * u;
                if (r < 2 * halo) put(in_s, PLANE_IN, (int)((q0 - halo + r) & (B3_WIN - 1)), pc4, wr[u], scl);
            }
        }
        if (!have || scale_exp_w(m_do) < e_do) {
            const int e_new = scale_exp_w(m_do) - HEAD_DO;
            d += have ? e_new - e_do : 0;
            e_do = e_new;
        }
        if (d != 0) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[t][a][b][j] = __builtin_ldexpf(acc[t][a][b][j], d);
        }
        have = true;
        {
            const float s_in = pow2f_w(e_in), s_do = pow2f_w(e_do);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = q0 + halo + prow + RPP * u;
                put(in_s, PLANE_IN, (int)(row & (B3_WIN - 1)), pc4, pin[u], s_in);
                put(do_s, PLANE_DO, prow + RPP * u, pc4, pdo[u], s_do);
                bsum += __builtin_bit_cast(f32x4, pdo[u]);
            }
        }
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
        f16x8 b[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[nt][p] = read_tr_frag_h(b_lo ^ (nt * 32), b_hi ^ (nt * 32), p * PLANE_DO);
        const unsigned s0 = (unsigned)q0 + lrow + B3_WIN;
        auto a_frags = [&](int tap, int mt, f16x8 (&a)[2]) {
            const int sh = (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
            const unsigned slot_lo = (s0 + (unsigned)sh) & (B3_WIN - 1), slot_hi = (slot_lo + 16) & (B3_WIN - 1);
            const unsigned cs = (a_col ^ b3x_swz(slot_lo)) ^ (mt * 32);
            const unsigned a_lo = a_base + ((slot_lo << 7) | cs), a_hi = a_base + ((slot_hi << 7) | cs);
#pragma unroll
            for (int p = 0; p < 2; ++p) a[p] = read_tr_frag_h(a_lo, a_hi, p * PLANE_IN);
        };
        auto mfmas = [&](int tap, int mt, const f16x8 (&a)[2]) {   // smallest terms first: a1 b2, a2 b1, a1 b1
#define LAD_WH2_TERM(pa, pb) \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) acc[tap][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pa], b[nt][pb], acc[tap][mt][nt], 0, 0, 0);
            LAD_WH2_TERM(0, 1)
            LAD_WH2_TERM(1, 0)
            LAD_WH2_TERM(0, 0)
#undef LAD_WH2_TERM
        };
        // software pipeline by halves, as wgrad_b3x_kernel
        f16x8 a0[2], a1[2], an[DOBN != 0 ? 1 : 2];
        a_frags(0, 0, a0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            a_frags(tap, 1, a1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(tap, 0, a0);
            __builtin_amdgcn_sched_barrier(0);
            if (DOBN != 0) {   // (registers: the next half-tap's fragments replace the ones the MFMAs above were issued with)
                if (tap + 1 < TAPS) a_frags(tap + 1, 0, a0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(tap, 1, a1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if (tap + 1 < TAPS) a_frags(tap + 1, 0, *reinterpret_cast<f16x8(*)[2]>(&an[0]));
                __builtin_amdgcn_sched_barrier(0);
                mfmas(tap, 1, a1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2; ++p) a0[p] = an[DOBN != 0 ? 0 : p];
            }
        }
    }

    float *slab = slabs + (int64_t)blockIdx.x * (TAPS * CH * CH);
    const int e_tot = e_in + e_do;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    slab[(tap * CH + mtw * 32 + mt * 16 + 4 * gr + j) * CH + ntw * 32 + nt * 16 + w16] = __builtin_ldexpf(acc[tap][mt][nt][j], -e_tot);
    if (bias_slabs != nullptr) {
        __syncthreads();
        *reinterpret_cast<f32x4 *>(bred_s + prow * CH + pc4 * 4) = bsum;
        __syncthreads();
        if (tid < CH) {
            float s = 0.0f;
            for (int pp = 0; pp < RPP; ++pp) s += bred_s[pp * CH + tid];
            bias_slabs[(int64_t)blockIdx.x * CH + tid] = s;
        }
    }
}

template <bool INBN, int DOBN>
int launch_wgrad_h2(const float *in, const float *in_coef, const float *dout, float *ws, float *dw, float *dbias, const Geom &g, hipStream_t st,
                    const WgBnBwd &bb = WgBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    constexpr int TAPS = 9, CH = 64, TK = 32, ROWB = CH * 2;
    if (TK + 2 * (g.Wp + 1) > B3_WIN) return lad::fail(LAD_ERR_INVALID, "wgrad (f16 x 2): image too wide for the window (W = %d)", g.Wp - 1);
    if (g.rows >= ((int64_t)1 << 31) - 65536 || g.img >= (1 << 20))
        return lad::fail(LAD_ERR_INVALID, "wgrad (f16 x 2): tensor too large for 32-bit row arithmetic");
    const int64_t n_tiles = lad::ceil_div(g.rows, TK);
    const int groups = groups_for(n_tiles);
    const int tiles_per_wg = (int)lad::ceil_div(n_tiles, groups);
    const size_t lds = 2 * B3_WIN * ROWB + 2 * TK * ROWB + (THREADS / (CH / 4)) * CH * sizeof(float) + 12 * sizeof(float) +
                       (2 + (DOBN != 0 ? 11 : 0)) * CH * sizeof(float) + (DOBN != 0 ? 256 + 2 * 8192 + 4 * 256 : 0);
    float *slabs = ws;
    float *bias_slabs = ws + (int64_t)MAX_GROUPS * TAPS * CH * CH;
    static lad::DeviceOnce attr_set;   // (the DOBN variants ask for 66 KB of dynamic LDS: past the 64 KB a kernel gets without opting in)
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_h2_kernel<INBN, DOBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_h2_kernel<INBN, DOBN>), dim3(groups), dim3(THREADS), lds, st, in, dout, slabs, dbias ? bias_slabs : nullptr, g, n_tiles,
                       tiles_per_wg, in_coef, bb);
    int rc = lad::check_launch("wgrad_h2_kernel");
    if (rc) return rc;
    return lad::reduce_slabs(lad::SlabReduce{slabs, dbias ? bias_slabs : nullptr, dw, dbias, groups, CH, CH, TAPS}, st);
}

template <int CH, bool INBN>
int launch_wgrad_b3(const float *in, const float *in_coef, const float *dout, float *ws, float *dw, float *dbias, const Geom &g, hipStream_t st) {
    constexpr int TAPS = 9;
    using K = WB3<CH>;
    if (K::TK + 2 * (g.Wp + 1) > B3_WIN) return lad::fail(LAD_ERR_INVALID, "wgrad (bf16 x 3): image too wide for the window (W = %d)", g.Wp - 1);
    if (g.rows >= ((int64_t)1 << 31) || g.img >= (1 << 20))
        return lad::fail(LAD_ERR_INVALID, "wgrad (bf16 x 3): tensor too large for 32-bit row arithmetic");
    const int64_t n_tiles = lad::ceil_div(g.rows, K::TK);
    const int groups = groups_for(n_tiles);
    const int tiles_per_wg = (int)lad::ceil_div(n_tiles, groups);
    const size_t lds = 3 * K::PLANE_IN + 3 * K::PLANE_DO + K::RPP * CH * sizeof(float);
    float *slabs = ws;
    float *bias_slabs = ws + (int64_t)MAX_GROUPS * TAPS * CH * CH;
    if constexpr (CH == 64) {   // wgrad_b3x_kernel (16x16x32); the round-2 form of the 64-channel kernel: tools/experiments/retired/
        static bool attr_x = false;
        if (!attr_x) {
            LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_b3x_kernel<INBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_x = true;
        }
        hipLaunchKernelGGL((wgrad_b3x_kernel<INBN>), dim3(groups), dim3(THREADS), lds, st, in, dout, slabs, dbias ? bias_slabs : nullptr, g,
                           n_tiles, tiles_per_wg, in_coef);
        int rcx = lad::check_launch("wgrad_b3x_kernel");
        if (rcx) return rcx;
        return lad::reduce_slabs(lad::SlabReduce{slabs, dbias ? bias_slabs : nullptr, dw, dbias, groups, CH, CH, TAPS}, st);
    } else {
        static lad::DeviceOnce attr_set;
        if (!attr_set) {
            LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_b3_kernel<CH, INBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_set = true;
        }
        if (g.rows >= ((int64_t)1 << 31) / (CH * 4)) return lad::fail(LAD_ERR_INVALID, "wgrad (bf16 x 3, 32x32x16 kernel): tensor too large for 32-bit offsets");
        hipLaunchKernelGGL((wgrad_b3_kernel<CH, INBN>), dim3(groups), dim3(THREADS), lds, st, in, dout, slabs, dbias ? bias_slabs : nullptr, g, n_tiles,
                           tiles_per_wg, in_coef);
        int rc = lad::check_launch("wgrad_b3_kernel");
        if (rc) return rc;
        return lad::reduce_slabs(lad::SlabReduce{slabs, dbias ? bias_slabs : nullptr, dw, dbias, groups, CH, CH, TAPS}, st);
    }
}



template <int CIN, int COUT, int TAPS>
int launch_wgrad(const float *in, const float *dout, float *ws, float *dw, float *dbias, const Geom &g, hipStream_t st) {
    const int64_t n_tiles = lad::ceil_div(g.rows, TMW);
    const int groups = groups_for(n_tiles);
    const int halo = (TAPS == 9) ? g.Wp + 1 : 0;
    const int nrows = TMW + 2 * halo;
    const size_t lds = ((size_t)nrows * CIN + 32 + TMW * COUT + 32 + THREADS) * sizeof(float);
    if (lds > 160 * 1024 || (int64_t)nrows * (CIN / 4) > (int64_t)WG_PRE_IN * THREADS)
        return lad::fail(LAD_ERR_INVALID, "wgrad: image too wide for the tile (W = %d)", g.Wp - 1);
    static lad::DeviceOnce attr_set;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)wgrad_kernel<CIN, COUT, TAPS>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    float *slabs = ws;
    float *bias_slabs = ws + (int64_t)MAX_GROUPS * TAPS * CIN * COUT;
    hipLaunchKernelGGL((wgrad_kernel<CIN, COUT, TAPS>), dim3(groups), dim3(THREADS), lds, st, in, dout, slabs,
                       dbias ? bias_slabs : nullptr, g, n_tiles);
    int rc = lad::check_launch("wgrad_kernel");
    if (rc) return rc;
    return lad::reduce_slabs(lad::SlabReduce{slabs, dbias ? bias_slabs : nullptr, dw, dbias, groups, CIN, COUT, TAPS}, st);
}

}  // namespace

#ifdef LAD_STAMP
extern "C" int lad_debug_read_wgrad_stamps(unsigned long long *host_dst, int64_t n) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lad_wg_dbg), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int64_t lad_conv_wgrad_workspace_floats(int32_t cin, int32_t cout, int32_t taps) {
    if (cin <= 0 || cout <= 0 || (taps != 1 && taps != 9)) return -1;
    return (int64_t)MAX_GROUPS * ((int64_t)taps * cin * cout + cout);
}

#define LAD_WG_CASE(CI, CO, T)                  \
    if (cin == CI && cout == CO && taps == T)   \
        return launch_wgrad<CI, CO, T>(in, dout, workspace, dw, dbias, g, (hipStream_t)stream);

extern "C" int lad_conv_wgrad_b3(const float *in, const float *dout, float *workspace, float *dw, float *dbias, int64_t batch,
                                 int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_wgrad_b3: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad_b3: bad geometry");
    return launch_wgrad_b3<64, false>(in, nullptr, dout, workspace, dw, dbias, make_geom(batch, H, W), (hipStream_t)stream);
}

// The same with in := relu(BatchNorm(in)) formed while the rows are staged (in_coef: lad_bn_finalize's float[6][64] of the
// BatchNorm between the previous convolution and this one): the weight gradient of a block's second convolution from
// the first one's raw output.  Bit-identical to lad_bn_act followed by lad_conv_wgrad_b3.
extern "C" int lad_conv_wgrad_b3_bnrelu(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw,
                                        float *dbias, int64_t batch, int32_t H, int32_t W, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && in_coef && dout && workspace && dw, "lad_conv_wgrad_b3_bnrelu: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad_b3_bnrelu: bad geometry");
    const Geom g = make_geom(batch, H, W);
    LAD_REQUIRE(g.img < (1 << 20), "lad_conv_wgrad_b3_bnrelu: image too large for 32-bit row arithmetic");
    return launch_wgrad_b3<64, true>(in, in_coef, dout, workspace, dw, dbias, g, (hipStream_t)stream);
}

// The same for `channels` = 64 or 32 (32: block2's stride-1 convolutions; four waves split the rows of a 64-row tile).  in_coef may be NULL
// (the input is the stored activation) or the BatchNorm coefficients of lad_conv_wgrad_b3_bnrelu.
extern "C" int64_t lad_conv_wgrad_b3c_workspace_floats(int32_t channels) {
    if (channels == 64) return (int64_t)MAX_GROUPS * (9 * 64 * 64 + 64);
    if (channels == 32) return (int64_t)MAX_GROUPS * (9 * 32 * 32 + 32);
    return -1;
}

extern "C" int lad_conv_wgrad_b3c(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw, float *dbias,
                                  int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_wgrad_b3c: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad_b3c: bad geometry");
    const Geom g = make_geom(batch, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (channels == 64)
        return in_coef ? launch_wgrad_b3<64, true>(in, in_coef, dout, workspace, dw, dbias, g, st)
                       : launch_wgrad_b3<64, false>(in, nullptr, dout, workspace, dw, dbias, g, st);
    if (channels == 32)
        return in_coef ? launch_wgrad_b3<32, true>(in, in_coef, dout, workspace, dw, dbias, g, st)
                       : launch_wgrad_b3<32, false>(in, nullptr, dout, workspace, dw, dbias, g, st);
    return fail(LAD_ERR_INVALID, "lad_conv_wgrad_b3c: 64 or 32 channels (got %d)", channels);
}


// The 64-channel weight (+ bias) gradient on two f16 planes per operand (wgrad_h2_kernel); arguments as lad_conv_wgrad_b3c
// (workspace: lad_conv_wgrad_b3c_workspace_floats(64)).
extern "C" int lad_conv_wgrad_h2(const float *in, const float *in_coef, const float *dout, float *workspace, float *dw, float *dbias,
                                 int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_wgrad_h2: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad_h2: bad geometry");
    LAD_REQUIRE(channels == 64, "lad_conv_wgrad_h2: 64 channels (got %d)", channels);
    const Geom g = make_geom(batch, H, W);
    hipStream_t st = (hipStream_t)stream;
    return in_coef ? launch_wgrad_h2<true, 0>(in, in_coef, dout, workspace, dw, dbias, g, st)
                   : launch_wgrad_h2<false, 0>(in, nullptr, dout, workspace, dw, dbias, g, st);
}

// The same launch with a BatchNorm backward on its `dout` side (wgrad_h2_kernel, DOBN): `dy` is the gradient arriving at the
// BatchNorm(+ReLU) that follows this convolution, bn_x that BatchNorm's input (this convolution's output), bn_coef / bcoef its
// forward and backward coefficients (lad_bn_finalize; lad_bn_bwd / lad_bn_bwd_bits called with dx = NULL).  The ReLU decisions come
// from bn_bits (the block output's sign bits, lad_bn_act_bits) or, when bn_bits is NULL, are recomputed from bn_x.  Writes
// dc = the BatchNorm's input gradient (what lad_bn_bwd would have written to dx, bit for bit) and accumulates dw / dbias from it.
extern "C" int lad_conv_wgrad_h2_bnbwd(const float *in, const float *in_coef, const float *dy, const float *bn_x, const uint64_t *bn_bits,
                                       const float *bn_coef, const float *bcoef, float *dc, float *workspace, float *dw, float *dbias,
                                       int64_t batch, int32_t H, int32_t W, int32_t channels, void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dy && bn_x && bn_coef && bcoef && dc && workspace && dw, "lad_conv_wgrad_h2_bnbwd: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad_h2_bnbwd: bad geometry");
    LAD_REQUIRE(channels == 64, "lad_conv_wgrad_h2_bnbwd: 64 channels (got %d)", channels);
    LAD_REQUIRE(dc != dy && dc != bn_x && dc != in, "lad_conv_wgrad_h2_bnbwd: dc must be a buffer of its own");
    const Geom g = make_geom(batch, H, W);
    hipStream_t st = (hipStream_t)stream;
    const WgBnBwd bb{bn_x, (const unsigned long long *)bn_bits, bn_coef, bcoef, dc};
    if (bn_bits != nullptr)
        return in_coef ? launch_wgrad_h2<true, 3>(in, in_coef, dy, workspace, dw, dbias, g, st, bb)
                       : launch_wgrad_h2<false, 3>(in, nullptr, dy, workspace, dw, dbias, g, st, bb);
    return in_coef ? launch_wgrad_h2<true, 2>(in, in_coef, dy, workspace, dw, dbias, g, st, bb)
                   : launch_wgrad_h2<false, 2>(in, nullptr, dy, workspace, dw, dbias, g, st, bb);
}

extern "C" int lad_conv_wgrad(const float *in, const float *dout, float *workspace, float *dw, float *dbias,
                              int64_t batch, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t taps,
                              void *stream) {
    using namespace lad;
    LAD_REQUIRE(in && dout && workspace && dw, "lad_conv_wgrad: null buffer");
    LAD_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "lad_conv_wgrad: bad geometry");
    const Geom g = make_geom(batch, H, W);
    LAD_WG_CASE(64, 64, 9)
    LAD_WG_CASE(32, 32, 9)
    LAD_WG_CASE(16, 16, 9)
    LAD_WG_CASE(64, 32, 9)
    LAD_WG_CASE(32, 16, 9)
    LAD_WG_CASE(64, 32, 1)
    LAD_WG_CASE(32, 16, 1)
    LAD_WG_CASE(16, 16, 1)
    return fail(LAD_ERR_INVALID, "lad_conv_wgrad: unsupported (cin=%d, cout=%d, taps=%d)", cin, cout, taps);
}
