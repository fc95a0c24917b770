#!/usr/bin/env python3
"""Numerics of candidate arithmetics for the 64->64 3x3 stride-1 convolutions of block1 (models.py:86-96, K = 9 * 64 = 576),
against float64.  CPU only (numpy); writes a table to stdout (kept as profiles/r04_conv_arith_numerics.log).

VERDICT r3 item 1 asked for this report before any kernel: does a Winograd F(2x2, 3x3) form of the split-operand
convolution hold the operator bar (2e-4 of max) and how does it compare with what runs today?  It also prices the
alternative that needs no transform at all: TWO f16 planes per operand (round-to-nearest residual split, x = h1 + h2 up to
2^-24 |x| for |x| within 2^17 of the tensor's maximum after a power-of-two scale), three plane products a1 b1 + a1 b2 + a2 b1.

Arithmetics compared (every one accumulates in f32, as the MFMA does; `*_f64acc` rows accumulate in float64 to isolate the
OPERAND error from the accumulation error, which is common to all of them and to the reference's own fp32 convolution):

  f32            plain fp32 operands (what torch / the exact-f32 MFMA kernel compute)
  bf16x3         three bf16 planes per operand, six products (the kernels of rounds 2-3)
  f16x2          two f16 planes per operand (RNE), three products, per-tensor power-of-two scale
  f16x2+h2h2     the same with the fourth product
  wino_f32       Winograd F(2x2,3x3), transforms and products in fp32
  wino_bf16x3    Winograd with the transformed operands split into three bf16 planes, six products
  wino_f16x2     Winograd with two f16 planes (scale from the transformed tensors' maxima)

Data sets: (a) forward-like: relu(BatchNorm-like) activations, weights ~ N(0, 0.05); (b) gradient-like: heavy-tailed
(normal x log-normal with sigma 2: 6+ decades of dynamic range), channel scales spread over 2^10.
"""
import sys

import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split_bf16x3(x):
    x = x.astype(np.float32)
    p1 = bf16_rne(x)
    r = (x - p1).astype(np.float32)
    p2 = bf16_rne(r)
    p3 = bf16_rne((r - p2).astype(np.float32))
    return p1, p2, p3


def pow2_scale(x):
    """s = 2^e with max|x| * s in [2^14, 2^15): the largest f16 magnitudes stay finite, everything else as high as possible."""
    m = float(np.abs(x).max())
    if m == 0.0:
        return 1.0
    return 2.0 ** (14 - int(np.floor(np.log2(m))))


def split_f16x2(x, s):
    xs = (x.astype(np.float32) * np.float32(s)).astype(np.float32)      # exact: power of two
    h1 = xs.astype(np.float16)
    r = (xs - h1.astype(np.float32)).astype(np.float32)                 # exact in f32
    h2 = r.astype(np.float16)
    return h1.astype(np.float32), h2.astype(np.float32)


def mm(a, b, acc):
    if acc == "f32":
        return a.astype(np.float32) @ b.astype(np.float32)
    return a.astype(np.float64) @ b.astype(np.float64)


def gemm_f32(A, B, acc):
    return mm(A, B, acc)


def gemm_bf16x3(A, B, acc):
    a1, a2, a3 = split_bf16x3(A)
    b1, b2, b3 = split_bf16x3(B)
    out = mm(a1, b3, acc) + mm(a2, b2, acc) + mm(a3, b1, acc)      # smallest terms first, as the kernel orders them
    out = out + mm(a1, b2, acc) + mm(a2, b1, acc)
    return out + mm(a1, b1, acc)


def gemm_f16x2(A, B, acc, fourth=False, sa=None, sb=None):
    sa = pow2_scale(A) if sa is None else sa
    sb = pow2_scale(B) if sb is None else sb
    a1, a2 = split_f16x2(A, sa)
    b1, b2 = split_f16x2(B, sb)
    out = mm(a1, b2, acc) + mm(a2, b1, acc)
    if fourth:
        out = out + mm(a2, b2, acc)
    out = out + mm(a1, b1, acc)
    return out * (1.0 / (sa * sb))


def im2col(x):
    """x (N, H, W, C) -> (N*H*W, 9*C), zero padding 1."""
    N, H, W, C = x.shape
    xp = np.zeros((N, H + 2, W + 2, C), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    cols = [xp[:, ky:ky + H, kx:kx + W].reshape(N * H * W, C) for ky in range(3) for kx in range(3)]
    return np.concatenate(cols, axis=1)


BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd(x, w, gemm, acc, tdtype=np.float32):
    """x (N, H, W, C), w (3, 3, C, K); H, W even.  Transforms in `tdtype` (fp32: what the VALU would do), the 16 frequency
    GEMMs through `gemm`."""
    N, H, W, C = x.shape
    K = w.shape[3]
    xp = np.zeros((N, H + 2, W + 2, C), tdtype)
    xp[:, 1:-1, 1:-1] = x
    th, tw = H // 2, W // 2
    # input tiles d[n, ty, tx, i, j, c] = xp[n, 2ty + i, 2tx + j, c]
    d = np.stack([np.stack([xp[:, i:i + 2 * th:2, j:j + 2 * tw:2] for j in range(4)], axis=3) for i in range(4)], axis=3)
    bt = BT.astype(tdtype)
    v = np.einsum("ai,ntsijc->ntsajc", bt, d).astype(tdtype)        # rows (each output = one add/sub of two inputs: exact order)
    v = np.einsum("bj,ntsajc->ntsabc", bt, v).astype(tdtype)
    g = G.astype(tdtype)
    u = np.einsum("ai,ijck->ajck", g, w.astype(tdtype)).astype(tdtype)
    u = np.einsum("bj,ajck->abck", g, u).astype(tdtype)
    m = np.empty((N, th, tw, 4, 4, K), np.float64 if acc == "f64" else np.float32)
    for a in range(4):
        for b in range(4):
            m[:, :, :, a, b] = gemm(v[:, :, :, a, b].reshape(-1, C), u[a, b], acc).reshape(N, th, tw, K)
    at = AT.astype(m.dtype)
    y = np.einsum("ia,ntsabk->ntsibk", at, m).astype(m.dtype)
    y = np.einsum("jb,ntsibk->ntsijk", at, y).astype(m.dtype)
    return y.transpose(0, 1, 3, 2, 4, 5).reshape(N, H, W, K)


def report(name, got, ref):
    e = got.astype(np.float64) - ref
    mx = np.abs(e).max() / np.abs(ref).max()
    rms = np.sqrt((e ** 2).mean()) / np.sqrt((ref ** 2).mean())
    print(f"  {name:<22} max|err|/max|ref| = {mx:9.3e}   rms err / rms ref = {rms:9.3e}")
    return mx, rms


def datasets(rng, N, H, W, C, K):
    gam = rng.uniform(0.5, 1.5, C)
    bet = rng.normal(0, 0.5, C)
    act = np.maximum(rng.standard_normal((N, H, W, C)) * gam + bet, 0).astype(np.float32)
    wgt = (rng.standard_normal((3, 3, C, K)) * 0.05).astype(np.float32)
    yield "forward-like (relu(bn) activations, N(0,0.05) weights)", act, wgt
    chan = 2.0 ** rng.uniform(-10, 0, C)
    grad = (rng.standard_normal((N, H, W, C)) * np.exp(2.0 * rng.standard_normal((N, H, W, C))) * chan * 1e-4).astype(np.float32)
    wgt2 = (rng.standard_normal((3, 3, C, K)) * 0.05 * 2.0 ** rng.uniform(-4, 0, (1, 1, C, 1))).astype(np.float32)
    yield "gradient-like (heavy tails, channel scales over 2^10)", grad, wgt2


def main():
    rng = np.random.default_rng(2024)
    N, H, W, C, K = 4, 100, 44, 64, 64
    rows = {}
    for title, x, w in datasets(rng, N, H, W, C, K):
        print(f"\n== {title}: x {x.shape}, w {w.shape}, K = {9 * C}")
        A = im2col(x)
        Bm = w.reshape(9 * C, K)
        ref = A.astype(np.float64) @ Bm.astype(np.float64)
        ref4 = ref.reshape(N, H, W, K)
        for acc in ("f32", "f64"):
            print(f" accumulate in {acc}:")
            tag = "" if acc == "f32" else "_f64acc"
            rows[(title, "f32" + tag)] = report("f32" + tag, gemm_f32(A, Bm, acc), ref)
            rows[(title, "bf16x3" + tag)] = report("bf16x3" + tag, gemm_bf16x3(A, Bm, acc), ref)
            rows[(title, "f16x2" + tag)] = report("f16x2" + tag, gemm_f16x2(A, Bm, acc), ref)
            rows[(title, "f16x2+h2h2" + tag)] = report("f16x2+h2h2" + tag, gemm_f16x2(A, Bm, acc, fourth=True), ref)
            rows[(title, "wino_f32" + tag)] = report("wino_f32" + tag, winograd(x, w, gemm_f32, acc), ref4)
            rows[(title, "wino_bf16x3" + tag)] = report("wino_bf16x3" + tag, winograd(x, w, gemm_bf16x3, acc), ref4)
            rows[(title, "wino_f16x2" + tag)] = report("wino_f16x2" + tag, winograd(x, w, gemm_f16x2, acc), ref4)
    # the split itself
    print("\n== operand representation error |x - sum of planes| / |x| (forward-like activations, nonzero elements)")
    x = next(datasets(np.random.default_rng(7), 2, 100, 44, 64, 64))[1].reshape(-1)
    x = x[x != 0]
    p1, p2, p3 = split_bf16x3(x)
    e3 = np.abs(x.astype(np.float64) - (p1.astype(np.float64) + p2 + p3)) / np.abs(x)
    s = pow2_scale(x)
    h1, h2 = split_f16x2(x, s)
    e2 = np.abs(x.astype(np.float64) * s - (h1.astype(np.float64) + h2)) / np.abs(x * s)
    print(f"  bf16x3: max {e3.max():.3e}   f16x2: max {e2.max():.3e} (2^-24 = {2.0 ** -24:.3e}), "
          f"max over |x| >= 2^-17 max|x|: {e2[np.abs(x) >= np.abs(x).max() * 2.0 ** -17].max():.3e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
