// fp16 sliding-window inference: the 64 -> 32 stride-2 block entry of the LEVEL-2 STRIPS, input resident in LDS (round 6).
//
// Replaces lad_f16_conv_s2_fwd_mapped_sc (conv_f16_s2_kernel<64, 32, 9, WMAP, SC>: block2.0 conv1 3x3 stride 2 + its 1x1 stride-2 shortcut,
// models.py:98-106, read through the window map) for the strips of the second resolution level in engine._eval_level2_shared: 8,268
// images of 12 x 22 per group of 8,192 windows, 93 GFLOP in 343 us = 270 TFLOP/s -- that kernel gathers every output row's 9 x 64 input
// channels from L2 per lane (2.4 GB through the L1 per launch: the texture addresser, not the matrix cores, sets its pace).
//
// Here, as in tail_f16.hip, the stride-2 layer runs on PARITY CLASSES filled by LDS-DMA (the de-interleave is free: a DMA lane's source
// address is its own) -- class (py, px) holds x[2 I + py][2 J + px] at pitch W_out + 1, the pitch of the OUTPUT tensor, so tap (ky, kx) of
// output position q reads class (ky & 1, kx & 1) at row q + (ky >> 1)(W_out + 1) + (kx >> 1) - 1 (relative to the unit's first row): nine
// row-shifted GEMMs from LDS, no gather.  A strip image (12 output rows) takes 25 input rows of 45 x 64 halfs = 144 KB -- too much at once
// -- so a workgroup walks its images in THIRDS: four output rows = 92 positions = three 32-position tiles (4 % padding) from nine input
// rows (53 KB as classes), double-buffered: waves 0-2 multiply unit u (40 MFMAs each: nine taps x four k-steps + the shortcut's four on
// the centre tap's fragments) while waves 3-15 bring unit u + 1 in, 52 chunks of 1 KB, four per wave.  Weights (41 KB) stay resident.
// Measured (profiles/r06_s2strip.log): 343 -> ~250 us per group of 8,192 windows; ablations: without its DMA the launch takes ~200 us,
// without its MFMA waves ~150 us, without both ~0 -- a unit's tile costs its wave ~4,300 cycles (40 MFMAs = 1,280, 76 fragment reads, ~300
// vector instructions, 16 stores: one wave per SIMD, so they add up) and only two units fit LDS next to the weights.
//
// Bit-identical to the kernel it replaces on every output a window uses: same v_mfma_f32_32x32x16_f16 and order per output element
// (taps 0..8, k-steps 0..3), epilogue_f16's arithmetic (fma, + 0.0f, ReLU for conv1 / none for the shortcut, border positions zero).
// (Outputs no window uses -- the lower halves of the first out_shift strip images, whose window lies before the group -- are computed from
// in-bounds rows here; the replaced kernel read them from before the buffer.)
#include "lad_common.h"

#include <algorithm>
#include <type_traits>

#include "lad_device.h"

namespace {
using namespace lad;

constexpr int S2_THREADS = 1024, S2_WAVES = S2_THREADS / 64;
constexpr int S2_CIN = 64, S2_COUT = 32, S2_KS = S2_CIN / 16;
constexpr int S2_CWAVES = 3;                       // compute waves = tiles of a unit
constexpr int S2_DWAVES = S2_WAVES - S2_CWAVES;    // DMA waves

struct S2Args {
    const _Float16 *act;                 // level-1 strips followed by the stream (64 channels)
    _Float16 *out, *out_sc;              // [n_img][out_rows + 1][W_out + 1][32] each (shared-border layout) + tail
    const _Float16 *wt, *wt_sc;          // lad_f16_pack_weights images (9 taps / 1 tap)
    const float *scale, *shift, *scale_sc, *shift_sc;
    int n_img;                           // output strip images = windows + out_shift
    int H, band, strip_rows;             // rows of a window at the input level; rows taken from the strips; rows of a strip image
    long long img_t, bot_img, stream_row0;   // positions of a strip image; bottom-strip image of output image 0; first row of the stream
    int out_shift, row_shift2;           // windows between the two users of an output strip; 2 x (Ho - out_rows): the lower half's row offset
};

constexpr int tl2_up(int v, int a) { return (v + a - 1) / a * a; }
// W: columns of a window at the input level (even); OR: rows of an output strip image (a multiple of 4: thirds / quarters of 4 rows)
template <int W, int OR>
struct S2Geo {
    static constexpr int Wp = W + 1, Wo = W / 2, Wpo = Wo + 1;
    static constexpr int UR = 4;                                   // output rows per unit
    static constexpr int NU = OR / UR;                             // units per image
    static constexpr int IR = 2 * UR + 1;                          // input rows per unit
    static constexpr int C0 = (UR + 1) * Wpo, C1 = UR * Wpo;       // rows of the classes with py = 0 / py = 1
    static constexpr int X0[4] = {0, C0, 2 * C0, 2 * C0 + C1};
    static constexpr int NX = 2 * C0 + 2 * C1;                     // rows (128 B) of one X buffer
    static constexpr int NPOS = UR * Wpo;                          // output positions of a unit
    static constexpr int NSLOT = NX * 8, NCHUNK = (NSLOT + 63) / 64, NCH = (NCHUNK + S2_DWAVES - 1) / S2_DWAVES;
    static constexpr int IMG_O = (OR + 1) * Wpo;                   // positions of an output image
    static constexpr int W_BYTES = 9 * S2_KS * 1024, WSC_BYTES = S2_KS * 1024;
    static constexpr int W_OFF = 0, WSC_OFF = W_BYTES, X_OFF = W_BYTES + WSC_BYTES, X_BYTES = NX * 128;
    // (+ 4 KB behind the tables: what the four dropped lanes of a unit's last tile read below the second buffer)
    static constexpr int COEF_OFF = X_OFF + 2 * X_BYTES, DSC_OFF = COEF_OFF + 4 * S2_COUT * 4, TOTAL = DSC_OFF + tl2_up(NU * NX * 2, 16) + 4096;
    static_assert(NPOS <= S2_CWAVES * 32 && OR % UR == 0 && W % 2 == 0, "unit geometry");
};

// byte offset of 16-byte slot `slot` (8 channels) of 128-byte row `row`: slots XOR-ed with (row >> 1) & 7 (block_f16_strip_kernel's layout:
// conflict-free ds_read_b128 for the 16-lane groups of MI355X_MICROARCH.md at this pitch)
__device__ __forceinline__ int s2_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

typedef _Float16 sf16x2 __attribute__((ext_vector_type(2)));

template <int W, int OR>
__global__ __launch_bounds__(S2_THREADS, 1) void s2strip_f16_kernel(S2Args a) {
    using G = S2Geo<W, OR>;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- once per workgroup: weights, BatchNorm folds ---------------------------------------------------------------------------------
    for (int p = tid; p < G::W_BYTES / 16; p += S2_THREADS) reinterpret_cast<u32x4 *>(lds + G::W_OFF)[p] = reinterpret_cast<const u32x4 *>(a.wt)[p];
    for (int p = tid; p < G::WSC_BYTES / 16; p += S2_THREADS) reinterpret_cast<u32x4 *>(lds + G::WSC_OFF)[p] = reinterpret_cast<const u32x4 *>(a.wt_sc)[p];
    if (tid < 4 * S2_COUT) {
        const int k = tid / S2_COUT, c = tid % S2_COUT;
        reinterpret_cast<float *>(lds + G::COEF_OFF)[tid] = (k == 0 ? a.scale : k == 1 ? a.shift : k == 2 ? a.scale_sc : a.shift_sc)[c];
    }

    // ---- DMA descriptors, one 16-bit word per unit of an image (its thirds) and ROW of X, in LDS: kind (bits 14-15: 0 the image's top strip --
    // also the zeros: its border row, and position 0 for the pad column --, 1 the bottom strip, 2 the stream) and position within that place.
    // (Per lane and chunk in registers, as tail_f16.hip keeps them, they were twelve registers that the compute waves' fragment sets need.)
    unsigned short *rowdsc = reinterpret_cast<unsigned short *>(lds + G::DSC_OFF);
    for (int e = tid; e < G::NU * G::NX; e += S2_THREADS) {
        const int t = e / G::NX, ra = e - t * G::NX;
        const int cls = ra >= G::X0[3] ? 3 : ra >= G::X0[2] ? 2 : ra >= G::X0[1] ? 1 : 0;
        const int rr = ra - (cls == 3 ? G::X0[3] : cls == 2 ? G::X0[2] : cls == 1 ? G::X0[1] : 0);
        const int I = rr / G::Wpo, J = rr - I * G::Wpo;
        const int ypl = 2 * G::UR * t + 2 * I + (cls >> 1), xp = 2 * J + (cls & 1);   // padded input row within the image's 2 OR + 1
        int kind = 0, pos = 0;
        if (xp <= W) {
            if (ypl <= OR) {                              // the upper half's rows: row y = ypl - 1 of the window at this offset
                const int y = ypl - 1;
                if (y < a.band) kind = 0, pos = ypl * G::Wp + xp;          // (y = -1: the strip image's border row, zeros)
                else kind = 2, pos = y * G::Wp + xp;
            } else {                                     // the lower half's: row y of the window out_shift offsets earlier
                const int y = ypl - 1 + a.row_shift2;
                if (y >= a.H) kind = 0, pos = 0;
                else if (y >= a.H - a.band) kind = 1, pos = (y - (a.H - a.strip_rows) + 1) * G::Wp + xp;
                else kind = 2, pos = (y - a.out_shift) * G::Wp + xp;
            }
        }
        rowdsc[e] = (unsigned short)((kind << 14) | pos);
    }
    const int dw = wave - S2_CWAVES;   // DMA wave number (negative: a compute wave)
    auto stage_unit = [&](int img, int t, int buf) __attribute__((always_inline)) {
        const unsigned char *b_top = reinterpret_cast<const unsigned char *>(a.act) + (long long)img * a.img_t * 128;
        const unsigned char *b_bot = reinterpret_cast<const unsigned char *>(a.act) + (a.bot_img + img) * a.img_t * 128;
        const unsigned char *b_str = reinterpret_cast<const unsigned char *>(a.act) + (a.stream_row0 + (long long)(img + 1) * G::Wp) * 128;
        int ll = lane;
        asm volatile("" : "+v"(ll));
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const int ch = dw + c * S2_DWAVES;
            const int sl = (ch << 6) + ll;
            if (ch < G::NCHUNK && sl < G::NSLOT) {
                const int ra = sl >> 3;
                const unsigned d = rowdsc[t * G::NX + ra];
                const unsigned kind = d >> 14, pos = d & 0x3fffu;
                const unsigned piece = (unsigned)((sl & 7) ^ ((ra >> 1) & 7));
                const unsigned char *base = kind == 0 ? b_top : kind == 1 ? b_bot : b_str;
                dma16(base + pos * 128u + piece * 16u, lds_addr(lds + G::X_OFF + buf * G::X_BYTES + ch * 1024));
            }
        }
    };


    float fzero = 0.0f;
    asm volatile("" : "+v"(fzero));   // (epilogue_f16 adds its absent residual: + 0.0f turns a -0.0 into +0.0)

    const int n_mine = (int)blockIdx.x < a.n_img ? (a.n_img - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_units = n_mine * G::NU;
    __syncthreads();   // the descriptor table is written
    if (n_units > 0 && dw >= 0) stage_unit((int)blockIdx.x, 0, 0);
#pragma unroll 1
    for (int u = 0; u < n_units; ++u) {
        const int k = u / G::NU, t = u - k * G::NU;
        const int img = (int)blockIdx.x + k * (int)gridDim.x;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // unit u's rows are in buffer u & 1; the other buffer's readers (unit u - 1) are done
        if (dw >= 0) {
            if (u + 1 < n_units) {
                const int k1 = (u + 1) / G::NU, t1 = (u + 1) - k1 * G::NU;
                stage_unit((int)blockIdx.x + k1 * (int)gridDim.x, t1, (u + 1) & 1);
            }
            if (t == 0 && dw >= S2_DWAVES - 2) {   // the image's border row (zeros) in both outputs: two of the DMA waves
                _Float16 *dst = (dw == S2_DWAVES - 1 ? a.out : a.out_sc) + (long long)img * G::IMG_O * S2_COUT;
                for (int p = lane; p < G::Wpo * S2_COUT * 2 / 16; p += 64) reinterpret_cast<u32x4 *>(dst)[p] = u32x4{0u, 0u, 0u, 0u};
                if (img == a.n_img - 1) {          // ... and the Wpo + 1 zero rows behind the last image
                    _Float16 *tail = dst + (long long)G::IMG_O * S2_COUT;
                    for (int p = lane; p < (G::Wpo + 1) * S2_COUT * 2 / 16; p += 64) reinterpret_cast<u32x4 *>(tail)[p] = u32x4{0u, 0u, 0u, 0u};
                }
            }
            continue;
        }
        // ---- a compute wave: its 32-position tile of the unit ------------------------------------------------------------------------
        int ll = lane;
        asm volatile("" : "+v"(ll));   // (per-lane addresses are formed per unit, not carried through the loop in registers)
        const int i = ll & 31, h = ll >> 5;
        const int qr = wave * 32 + i;                                  // position within the unit
        const int xpo = qr % G::Wpo;
        const bool valid = qr < G::NPOS, keep_b = valid && xpo >= 1;   // (every row of a unit is an interior row: ypo >= 1)
        const unsigned char *xb = lds + G::X_OFF + (u & 1) * G::X_BYTES;
        const unsigned char *w_lane = lds + G::W_OFF + (h * 32 + i) * 16, *w2_lane = lds + G::WSC_OFF + (h * 32 + i) * 16;
        f32x16 acc, acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc2[r] = 0.f;
        // Software pipeline by taps: the eight fragments of tap t + 1 are requested before the MFMAs of tap t are issued, so an LDS round
        // trip hides under four to eight MFMAs (one wave per SIMD here: left to the compiler, which reads one or two MFMAs ahead, every
        // MFMA waited for most of it -- the first version of this kernel took 5,400 cycles per unit for 1,280 of matrix pipe).
        f16x8 wf[2][S2_KS], xf[2][S2_KS], wf2[S2_KS];
        auto frags = [&](int tap, int b) __attribute__((always_inline)) {
            const int ky = tap / 3, kx = tap % 3;
            const int row = qr + G::X0[(ky & 1) * 2 + (kx & 1)] + (ky >> 1) * G::Wpo + (kx >> 1) - 1;
#pragma unroll
            for (int ks = 0; ks < S2_KS; ++ks) {
                wf[b][ks] = *reinterpret_cast<const f16x8 *>(w_lane + (tap * S2_KS + ks) * 1024);
                xf[b][ks] = *reinterpret_cast<const f16x8 *>(xb + s2_off(row, ks * 2 + h));
                if (tap == 4) wf2[ks] = *reinterpret_cast<const f16x8 *>(w2_lane + ks * 1024);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) frags(tap + 1, (tap + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < S2_KS; ++ks) {
                acc = mfma32_f16(wf[tap & 1][ks], xf[tap & 1][ks], acc);
                if (tap == 4) acc2 = mfma32_f16(wf2[ks], xf[tap & 1][ks], acc2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue (epilogue_f16): register 4 qd + j of lane (i, h) is channel 8 qd + 4 h + j of position i
        const long long orow = (long long)img * G::IMG_O + (G::UR * t + 1) * G::Wpo + qr;
        const unsigned keep = keep_b ? 0xffffffffu : 0u;
        const float *cf = reinterpret_cast<const float *>(lds + G::COEF_OFF) + 4 * h;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            _Float16 *dst = (which == 0 ? a.out : a.out_sc) + orow * S2_COUT + 4 * h;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 sv = *reinterpret_cast<const f32x4 *>(cf + which * 2 * S2_COUT + 8 * qd);
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(cf + which * 2 * S2_COUT + S2_COUT + 8 * qd);
                f32x4 tv = which == 0 ? f32x4{acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]}
                                      : f32x4{acc2[4 * qd], acc2[4 * qd + 1], acc2[4 * qd + 2], acc2[4 * qd + 3]};
                tv = __builtin_elementwise_fma(tv, sv, bv);
                tv = tv + f32x4{fzero, fzero, fzero, fzero};
                if (which == 0) tv = __builtin_elementwise_max(tv, f32x4{0.f, 0.f, 0.f, 0.f});
                asm("" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]));   // (no fma + conversion contraction)
                const sf16x2 lo = {(_Float16)tv[0], (_Float16)tv[1]}, hi = {(_Float16)tv[2], (_Float16)tv[3]};
                const u32x2 o = {__builtin_bit_cast(unsigned, lo) & keep, __builtin_bit_cast(unsigned, hi) & keep};
                if (valid) *reinterpret_cast<u32x2 *>(dst + 8 * qd) = o;
            }
        }
    }
}

}  // namespace

// block2.0's entry (3x3 stride-2 convolution + ReLU -> out, 1x1 stride-2 shortcut -> out_sc, both with their BatchNorm folded) on the
// LEVEL-2 STRIPS of the fp16 sliding-window path: the arguments of lad_f16_conv_s2_fwd_mapped_sc with phases = 1, cin = 64, cout = 32,
// relu = 1 and out_rows > 0, the input image rows resident in LDS as parity classes instead of gathered per lane (csrc/s2strip_f16.hip).
// Identical results on every output a window uses.  LAD_NOT_COVERED (nothing launched) for other geometries: the caller issues
// lad_f16_conv_s2_fwd_mapped_sc.  Replaces models.py:98-106 (ResidualBlock.__init__ / forward: conv1 + shortcut of a down-sampling
// block, eval mode) for those strips (segment_laughter.py:90-101 through engine._eval_level2_shared).
extern "C" int lad_f16_conv_s2_strips_fwd(const void *act, const void *wt, const float *scale, const float *shift, void *out, const void *wt_sc,
                                          const float *scale_sc, const float *shift_sc, void *out_sc, int64_t n_windows, int32_t H, int32_t W,
                                          int32_t band, int32_t strip_rows, int64_t bottom_image0, int64_t stream_row0, int64_t act_rows,
                                          int32_t out_rows, void *stream) {
    using namespace lad;
    LAD_REQUIRE(act && wt && scale && shift && out && wt_sc && scale_sc && shift_sc && out_sc && out_sc != out, "lad_f16_conv_s2_strips_fwd: null or aliased buffer");
    LAD_REQUIRE(n_windows >= 1 && H >= 2 && W >= 2 && band >= 1 && strip_rows >= band && H >= 2 * band, "lad_f16_conv_s2_strips_fwd: bad geometry");
    if (W != 44 || out_rows != 12 || (H & 1)) return LAD_NOT_COVERED;   // the instantiated geometry (100 x 44 windows, 12-row level-2 strips)
    using G = S2Geo<44, 12>;
    const int Ho = H / 2;
    const int64_t out_shift = 2 * (int64_t)(Ho - out_rows);
    if (out_rows + band + 1 > H || out_shift > bottom_image0 || bottom_image0 > H - strip_rows || 2 * out_rows + 1 > 2 * (strip_rows + 1) + 64)
        return LAD_NOT_COVERED;
    const int64_t n_img = n_windows + out_shift;
    const int64_t img_t = (int64_t)(strip_rows + 1) * (W + 1);
    // every row the descriptors can name lies inside the buffer (as conv_s2_mapped checks)
    const int64_t stream_hi = stream_row0 + (n_img + 1 + (int64_t)2 * out_rows) * (W + 1);
    LAD_REQUIRE(stream_row0 >= (bottom_image0 - out_shift + n_img) * img_t && stream_hi <= act_rows && act_rows < ((int64_t)1 << 31),
                "lad_f16_conv_s2_strips_fwd: the strips / the stream do not fit the buffer (%lld rows given)", (long long)act_rows);
    if (n_img >= (1 << 30) || G::TOTAL > 160 * 1024) return LAD_NOT_COVERED;
    S2Args a;
    a.act = (const _Float16 *)act; a.out = (_Float16 *)out; a.out_sc = (_Float16 *)out_sc;
    a.wt = (const _Float16 *)wt; a.wt_sc = (const _Float16 *)wt_sc;
    a.scale = scale; a.shift = shift; a.scale_sc = scale_sc; a.shift_sc = shift_sc;
    a.n_img = (int)n_img; a.H = H; a.band = band; a.strip_rows = strip_rows;
    a.img_t = img_t; a.bot_img = bottom_image0 - out_shift; a.stream_row0 = stream_row0;
    a.out_shift = (int)out_shift; a.row_shift2 = 2 * (Ho - out_rows);
    static lad::DeviceOnce attr_set;
    static int n_cu = 256;
    if (!attr_set) {
        LAD_HIP_CHECK(hipFuncSetAttribute((const void *)s2strip_f16_kernel<44, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    hipLaunchKernelGGL((s2strip_f16_kernel<44, 12>), dim3((unsigned)std::min<int64_t>(n_img, n_cu)), dim3(S2_THREADS), (size_t)G::TOTAL, (hipStream_t)stream, a);
    return check_launch("s2strip_f16_kernel");
}
