// Element-wise arithmetic of the BatchNorm backward pass, shared by bn.hip and the kernels that apply it on the fly
// (stem.hip: the stem's weight gradient consumes dz without it ever being written).
#pragma once
#include <hip/hip_runtime.h>

namespace lad {

// (mean, invstd) of 4 consecutive channels as (hi, lo) pairs; xhat = ((x - mean_hi) - mean_lo) * (istd_hi + istd_lo)
struct Norm4 {
    float4 mean, mean_lo, istd, istd_lo;
};
__device__ __forceinline__ Norm4 load_norm(const float *__restrict__ coef, int C, int c) {
    Norm4 n;
    n.mean = *reinterpret_cast<const float4 *>(coef + 2 * C + c);
    n.istd = *reinterpret_cast<const float4 *>(coef + 3 * C + c);
    n.mean_lo = *reinterpret_cast<const float4 *>(coef + 4 * C + c);
    n.istd_lo = *reinterpret_cast<const float4 *>(coef + 5 * C + c);
    return n;
}
__device__ __forceinline__ float xhat1(float x, float m, float ml, float s, float sl) {
    const float t = (x - m) - ml;
    return fmaf(t, s, t * sl);
}
__device__ __forceinline__ float4 xhat4(const float4 x, const Norm4 &n) {
    return make_float4(xhat1(x.x, n.mean.x, n.mean_lo.x, n.istd.x, n.istd_lo.x), xhat1(x.y, n.mean.y, n.mean_lo.y, n.istd.y, n.istd_lo.y),
                       xhat1(x.z, n.mean.z, n.mean_lo.z, n.istd.z, n.istd_lo.z), xhat1(x.w, n.mean.w, n.mean_lo.w, n.istd.w, n.istd_lo.w));
}
// ReLU mask of y = relu(x*scale + shift) recomputed from x: the same fmaf the forward pass evaluated, so the decision is
// bit-identical to testing the stored y > 0, and the pass reads one tensor less (no residual branch: relu == 2)
__device__ __forceinline__ float4 mask_from_x(float4 d, const float4 x, const float4 sc, const float4 sh) {
    d.x = fmaf(x.x, sc.x, sh.x) > 0.f ? d.x : 0.f;
    d.y = fmaf(x.y, sc.y, sh.y) > 0.f ? d.y : 0.f;
    d.z = fmaf(x.z, sc.z, sh.z) > 0.f ? d.z : 0.f;
    d.w = fmaf(x.w, sc.w, sh.w) > 0.f ? d.w : 0.f;
    return d;
}
// Sign bits of a 64-channel activation y (one uint64 per pixel row, written by bn_act_kernel<RES, true>): bit k*16 + j
// <-> channel 4*j + k, i.e. the 16 lanes that hold a row as float4 each own bit j of the four 16-bit fields.  A backward
// pass that only needs "was y > 0" reads 8 bytes per row instead of the 256-byte row of y.
__device__ __forceinline__ unsigned long long pack_sign_bits(const float4 o, int lane) {
    const unsigned long long bx = __ballot(o.x > 0.f), by = __ballot(o.y > 0.f), bz = __ballot(o.z > 0.f), bw = __ballot(o.w > 0.f);
    const int sh = lane & 48;  // first lane of this row's 16
    return ((bx >> sh) & 0xFFFFull) | (((by >> sh) & 0xFFFFull) << 16) | (((bz >> sh) & 0xFFFFull) << 32) | (((bw >> sh) & 0xFFFFull) << 48);
}
__device__ __forceinline__ float4 mask_from_bits(float4 d, unsigned long long word, int j /* channel quad 0..15 */) {
    const unsigned lo = (unsigned)word >> j, hi = (unsigned)(word >> 32) >> j;
    d.x = (lo & 1u) ? d.x : 0.f;
    d.y = (lo & 0x10000u) ? d.y : 0.f;
    d.z = (hi & 1u) ? d.z : 0.f;
    d.w = (hi & 0x10000u) ? d.w : 0.f;
    return d;
}
// dx = k1 * (((d - k2_hi) - k2_lo) - xhat * k3_hi - xhat * k3_lo)
__device__ __forceinline__ float bn_dx1(float d, float xh, float k1, float k2, float k2l, float k3, float k3l) {
    float t = (d - k2) - k2l;
    t = fmaf(-xh, k3, t);
    t = fmaf(-xh, k3l, t);
    return k1 * t;
}

}  // namespace lad
