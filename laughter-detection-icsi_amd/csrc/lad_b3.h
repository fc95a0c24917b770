// Three-way bf16 split of fp32 numbers ("bf16 x 3"): shared by conv_b3.hip and the weight-gradient kernel.
//
// Every fp32 operand is the EXACT sum of three bf16 numbers, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2) (8 + 8 + 8 significant bits; bf16 has the exponent range of fp32, |x| < 3.39e38).  A product
// a b is then a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1 up to terms below 2^-24 of it: six bf16 MFMAs with f32
// accumulation, each partial product exact in f32 -- as accurate against float64 as the fmaf chain of the f32 MFMA.
#pragma once
#include "lad_device.h"

namespace lad {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two fp32 values -> their three bf16 planes (packed pairs); x == p1 + p2 + p3 exactly
__device__ __forceinline__ void split_pair(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = pack_bf16(a, b);
    const float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);
    p2 = pack_bf16(ra, rb);
    p3 = pack_bf16(ra - bf16_lo(p2), rb - bf16_hi(p2));
}

}  // namespace lad
