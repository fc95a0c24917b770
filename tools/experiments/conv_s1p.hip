// Persistent, fully pipelined variant of the stride-1 3x3 implicit-GEMM convolution (f32 MFMA) for gfx950.
//
// Same operator as conv_s1_kernel of conv_mfma.hip (forward of nn.Conv2d 3x3 stride 1 -- models.py:86-98,110-115 -- and,
// with the mode-1 weight image, its data gradient); what changes is the schedule.  In conv_s1_kernel a workgroup
// stages its input rows, computes, stores and exits: its own load and store phases leave the matrix cores to the two
// other workgroups on the CU.  Here a workgroup is persistent (two per CU, each walks tiles blockIdx, blockIdx+grid, ...)
// and keeps its matrix cores busy by itself:
//   * the input rows of a tile are staged 32 channels at a time ("stage") into one of TWO LDS buffers by LDS-DMA
//     (global_load_lds_dwordx4: no registers, no instructions on the compute path).  While the MFMAs of stage g run
//     from buffer g&1, the DMA of stage g+1 -- the other channel half, or the first half of the NEXT tile -- fills the
//     other buffer.  This is possible because border rows are zero in HBM (layout invariant, lad_device.h): the rows
//     go from HBM to LDS untouched.
//   * rows in LDS are unpadded 128-byte rows; the bank conflicts of a 32-row x 16-byte fragment read are removed by an
//     XOR swizzle of the 16-byte slot with bits of the row number, applied on the SOURCE address of the DMA (the LDS
//     image of a DMA instruction is lane-linear) and on the ds_read_b128 address.
//   * weights: the same two-slot LDS ring, one (tap x 32-channel) chunk ahead, as conv_s1_kernel.
//   * epilogue: accumulators -> LDS transpose (16 rows per wave at a time, aliased onto the input buffer the tile has
//     finished with) -> bias / folded BatchNorm / residual / border mask -> 1 KB-contiguous float4 stores, per-tile
//     BatchNorm partial sums.  Only this phase is not overlapped inside the workgroup; the second workgroup on the CU
//     covers it.
#include "lad_common.h"
#include "lad_device.h"

namespace {
using namespace lad;

constexpr int TM = 128;
constexpr int THREADS = 256;
constexpr int KC = 32;  // channels per stage and per weight chunk

template <int COUT>
struct NTp {
    static constexpr int NT = (COUT + 31) / 32;
    static constexpr int COUTP = NT * 32;
};

// s_waitcnt vmcnt(n) needs an immediate: n is wave-uniform and small
__device__ __forceinline__ void dma_wait_keep(int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// 16-byte slot of (row, chunk) inside a 128-byte LDS row: two rows share a 256-byte bank row, so rows r, r+2, r+4, ...
// would collide; XOR with (row >> 1) & 7 spreads any 16 consecutive rows over all 16 slots
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int CIN, int COUT>
__global__ __launch_bounds__(THREADS, 2) void conv_s1p_kernel(const float *__restrict__ in, const float *__restrict__ wt,
                                                              const float *__restrict__ bias, const float *__restrict__ addend,
                                                              float *__restrict__ out, float *__restrict__ partials, Geom g,
                                                              const float *__restrict__ scale, int relu, int64_t n_tiles) {
    constexpr int NT = NTp<COUT>::NT;
    constexpr int COUTP = NTp<COUT>::COUTP;
    constexpr int NSTAGE = CIN / KC;
    constexpr int CHUNK = KC * COUTP;               // floats per weight chunk (one tap, 32 input channels)
    constexpr int CROUNDS = (CHUNK * 4 + THREADS * 16 - 1) / (THREADS * 16);
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, gk = lane >> 5;
    const int halo = g.Wp + 1;
    const int nrows = TM + 2 * halo;
    const int n_slots = nrows * (KC / 4);           // 16-byte slots per stage
    const int stage_iters = (n_slots + THREADS - 1) / THREADS;  // DMA instructions per wave and stage (the same for every wave)
    const int a_floats = stage_iters * THREADS * 4;   // one stage buffer in whole 4 KB rounds (the tail is padding)
    float *b_s = smem;                              // [2][CHUNK]
    float *a_s = b_s + 2 * CHUNK;                   // [2][nrows][KC], swizzled 16-byte slots
    float *mask_s = a_s + 2 * a_floats;             // [TM]
    float *red_s = mask_s + TM;                     // [4][2][COUT]

    // ---- DMA issue helpers ---------------------------------------------------------------------------------------
    auto issue_stage = [&](int64_t tile, int stage, int buf) {
        const int64_t qb = tile * TM - halo;        // tensor row of staged row 0
        for (int base = wave * 64; base < stage_iters * THREADS; base += THREADS) {   // a wave moves 64 slots = 1 KB per step
            const int L = base + lane;              // LDS slot index (lane-linear destination)
            const int row = min(L >> 3, nrows - 1), slot = L & 7;
            const int64_t q = min(max(qb + row, (int64_t)0), g.rows - 1);  // rows outside the tensor only feed border outputs
            const float *src = in + q * CIN + stage * KC + swz(row, slot) * 4;
            dma16(src, lds_addr(a_s + buf * a_floats + base * 4));
        }
    };
    auto issue_chunk = [&](int tap, int stage, int slot) {
        const float *src = wt + (int64_t)(tap * NSTAGE + stage) * CHUNK;
#pragma unroll
        for (int r = 0; r < CROUNDS; ++r)
            if ((r * THREADS + wave * 64) * 4 < CHUNK)
                dma16(src + (r * THREADS + tid) * 4, lds_addr(b_s + slot * CHUNK + (r * THREADS + wave * 64) * 4));
    };

    int64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    int gstage = 0;                                 // running stage counter: buffer = gstage & 1
    int gchunk = 0;                                 // running chunk counter: ring slot = gchunk & 1
    issue_stage(tile, 0, 0);
    issue_chunk(0, 0, 0);

    const int b_off = (gk * COUTP + i) * 4;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t q0 = tile * TM;
        f32x16 acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

#pragma unroll 1
        for (int stage = 0; stage < NSTAGE; ++stage, ++gstage) {
            const float *abuf = a_s + (gstage & 1) * a_floats;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap, ++gchunk) {
                // every DMA this wave has issued (this chunk, and the stage it belongs to) has landed; after the barrier
                // every wave's has, and nobody still reads the ring slot / stage buffer about to be refilled
                // (at tap 1 the input-row DMAs of the NEXT stage, issued during tap 0 after this tap's weight chunk, stay in
                //  flight: vmcnt counts in issue order, so leaving the youngest `stage_iters` outstanding is exactly that)
                if (tap == 1) dma_wait_keep(stage_iters);
                else dma_wait_all();
                __syncthreads();
                // next weight chunk
                if (tap + 1 < 9) issue_chunk(tap + 1, stage, (gchunk + 1) & 1);
                else if (stage + 1 < NSTAGE) issue_chunk(0, stage + 1, (gchunk + 1) & 1);
                else if (tile + gridDim.x < n_tiles) issue_chunk(0, 0, (gchunk + 1) & 1);
                if (tap == 0) {
                    // next stage of input rows into the buffer the previous stage (and the previous epilogue) released
                    if (stage + 1 < NSTAGE) issue_stage(tile, stage + 1, (gstage + 1) & 1);
                    else if (tile + gridDim.x < n_tiles) issue_stage(tile + gridDim.x, 0, (gstage + 1) & 1);
                    if (stage == 0 && tid < TM) mask_s[tid] = interior_row(q0 + tid, g) ? 1.0f : 0.0f;
                }
                const int row = wave * 32 + i + halo + (tap / 3 - 1) * g.Wp + (tap % 3 - 1);
                const float *ap = abuf + row * KC;
                const int rs = (row >> 1) & 7;
                const float *bp = b_s + (gchunk & 1) * CHUNK + b_off;
#pragma unroll
                for (int c8 = 0; c8 < KC / 8; ++c8) {
                    const float4 a = *reinterpret_cast<const float4 *>(ap + (((c8 * 2 + gk) ^ rs) << 2));
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float4 b = *reinterpret_cast<const float4 *>(bp + (c8 * 2 * COUTP + n * 32) * 4);
                        acc[n] = mfma32(a.x, b.x, acc[n]);
                        acc[n] = mfma32(a.y, b.y, acc[n]);
                        acc[n] = mfma32(a.z, b.z, acc[n]);
                        acc[n] = mfma32(a.w, b.w, acc[n]);
                    }
                }
            }
        }
        // ---- epilogue: the buffer of the tile's last stage is free once every wave has left the MFMA loop ----------------
        __syncthreads();
        float *my = a_s + ((gstage - 1) & 1) * a_floats + wave * (16 * COUT);   // 16 rows x COUT per wave
        constexpr int LPR = COUT / 4, RPI = 64 / LPR, ITER = 16 / RPI;
        const int c4 = lane % LPR, rsub = lane / LPR;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (bias != nullptr) bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
        if (scale != nullptr) sv = *reinterpret_cast<const float4 *>(scale + c4 * 4);
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 32 + i;
                if (co < COUT) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int lr = (r & 3) + 8 * (r >> 2) + 4 * gk;  // local row 0..15 of accumulator register half*8 + r
                        my[lr * COUT + co] = acc[n][half * 8 + r];
                    }
                }
            }
            float4 ad[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int64_t q = q0 + wave * 32 + half * 16 + it * RPI + rsub;
                ad[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (addend != nullptr && q < g.rows) ad[it] = *reinterpret_cast<const float4 *>(addend + q * COUT + c4 * 4);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int lrow = it * RPI + rsub;
                const int trow = wave * 32 + half * 16 + lrow;
                const int64_t q = q0 + trow;
                float4 t = *reinterpret_cast<const float4 *>(my + lrow * COUT + c4 * 4);
                if (scale != nullptr) {
                    t.x = fmaf(t.x, sv.x, bv.x) + ad[it].x; t.y = fmaf(t.y, sv.y, bv.y) + ad[it].y;
                    t.z = fmaf(t.z, sv.z, bv.z) + ad[it].z; t.w = fmaf(t.w, sv.w, bv.w) + ad[it].w;
                } else {
                    t.x += bv.x + ad[it].x; t.y += bv.y + ad[it].y; t.z += bv.z + ad[it].z; t.w += bv.w + ad[it].w;
                }
                if (relu) {
                    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                }
                if (mask_s[trow] == 0.0f) t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < g.rows) {
                    *reinterpret_cast<float4 *>(out + q * COUT + c4 * 4) = t;
                    s1.x += t.x; s1.y += t.y; s1.z += t.z; s1.w += t.w;
                    s2.x = fmaf(t.x, t.x, s2.x); s2.y = fmaf(t.y, t.y, s2.y);
                    s2.z = fmaf(t.z, t.z, s2.z); s2.w = fmaf(t.w, t.w, s2.w);
                }
            }
        }
        if (partials != nullptr) {
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) {
                s1.x += __shfl_xor(s1.x, off, 64); s1.y += __shfl_xor(s1.y, off, 64);
                s1.z += __shfl_xor(s1.z, off, 64); s1.w += __shfl_xor(s1.w, off, 64);
                s2.x += __shfl_xor(s2.x, off, 64); s2.y += __shfl_xor(s2.y, off, 64);
                s2.z += __shfl_xor(s2.z, off, 64); s2.w += __shfl_xor(s2.w, off, 64);
            }
            if (lane < LPR) {
                *reinterpret_cast<float4 *>(red_s + (wave * 2 + 0) * COUT + c4 * 4) = s1;
                *reinterpret_cast<float4 *>(red_s + (wave * 2 + 1) * COUT + c4 * 4) = s2;
            }
            __syncthreads();
            if (tid < 2 * COUT) {
                const int k = tid / COUT, co = tid - k * COUT;
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) s += red_s[(w * 2 + k) * COUT + co];
                partials[(tile * 2 + k) * COUT + co] = s;
            }
        }
        // the next iteration starts with dma_wait_all + __syncthreads: the epilogue's LDS traffic is retired before the
        // stage buffer it borrowed is refilled (that DMA is issued after that barrier)
    }
}

template <int CIN, int COUT>
int launch_p(const float *in, const float *wt, const float *bias, const float *addend, float *out, float *partials, const Geom &g,
             const float *scale, int relu, hipStream_t st) {
    constexpr int COUTP = NTp<COUT>::COUTP;
    const int halo = g.Wp + 1;
    const int nrows = TM + 2 * halo;
    const size_t a_floats = (((size_t)nrows * (KC / 4) + THREADS - 1) / THREADS) * THREADS * 4;
    const size_t lds = (2 * (size_t)KC * COUTP + 2 * a_floats + TM + 8 * COUT) * sizeof(float);
    if (lds > 160 * 1024 || (size_t)nrows * KC < (size_t)4 * 16 * COUT)
        return lad::fail(LAD_ERR_INVALID, "conv_s1p: unsupported tile geometry (W = %d)", g.Wp - 2);
    static bool attr_set = false;
    static int wgs_per_cu = 1;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)conv_s1p_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          160 * 1024));
        attr_set = true;
    }
    wgs_per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
    const int64_t n_tiles = lad::ceil_div(g.rows, TM);
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, (int64_t)256 * wgs_per_cu);
    hipLaunchKernelGGL((conv_s1p_kernel<CIN, COUT>), dim3(grid), dim3(THREADS), lds, st, in, wt, bias, addend, out, partials, g, scale,
                       relu, n_tiles);
    return lad::check_launch("conv_s1p_kernel");
}

}  // namespace

// Entry used by lad_conv_fwd / lad_conv_fwd_eval (conv_mfma.hip) for the shapes instantiated here; returns 1 if the
// shape is not covered (the caller then uses conv_s1_kernel).
namespace lad {
int conv_s1p_dispatch(const float *in, const float *wt, const float *bias, const float *addend, float *out, float *partials,
                      int64_t batch, int H, int W, int cin, int cout, int taps, const float *scale, int relu, hipStream_t st) {
    if (taps != 9) return 1;
    Geom g;
    g.Hp = H + 2;
    g.Wp = W + 2;
    g.img = g.Hp * g.Wp;
    g.rows = batch * g.img;
    if (cin == 64 && cout == 64) return launch_p<64, 64>(in, wt, bias, addend, out, partials, g, scale, relu, st);
    return 1;
}
}  // namespace lad
