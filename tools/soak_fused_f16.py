"""Soak of the round-5 fused fp16 inference launches: N back-to-back launches each of lad_f16_block_fwd (64 / 32 / 16 channels), the
one-convolution form inside lad_f16_conv_fwd and lad_f16_conv_s2_fwd_sc, every output compared bit for bit with the first one and with
the separate launches it replaces (the hand-placed waits of these kernels have no compiler behind them).
    python tools/soak_fused_f16.py [--n 200]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "laughter-detection-icsi_amd"))
import _hip as h  # noqa: E402


def pnhwc(x):
    B, C, H, W = x.shape
    buf = torch.zeros((B * (H + 1) * (W + 1) + W + 2) * C)
    buf[:B * (H + 1) * (W + 1) * C].view(B, H + 1, W + 1, C)[:, 1:, 1:, :] = x.permute(0, 2, 3, 1)
    return buf.half().cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    n = ap.parse_args().n
    lib, st = h.lib(), h.stream_handle()
    g = torch.Generator().manual_seed(5)
    bad = 0
    for C, B, H, W in ((64, 8282, 10, 44), (32, 8268, 12, 22), (16, 8192, 25, 11), (16, 8192, 13, 6)):
        x = pnhwc(torch.randn(B, C, H, W, generator=g))
        wts = []
        for _ in range(2):
            wt = torch.zeros(int(lib.lad_f16_packed_weight_halfs(C, C, 9)), device="cuda", dtype=torch.float16)
            h.check(lib.lad_f16_pack_weights(h.ptr((torch.randn(C, C, 3, 3, generator=g) * (0.5 / C ** 0.5)).cuda()), C, C, 9, h.ptr(wt), st))
            wts.append(wt)
        sc = [(torch.rand(C, generator=g) + 0.5).cuda() for _ in range(2)]
        sh = [(torch.randn(C, generator=g) * 0.2).cuda() for _ in range(2)]
        a1, ref, y = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
        h.check(lib.lad_f16_conv_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(sc[0]), h.ptr(sh[0]), None, h.ptr(a1), B, H, W, C, C, 9, 1, st))
        h.check(lib.lad_f16_conv_fwd(h.ptr(a1), h.ptr(wts[1]), h.ptr(sc[1]), h.ptr(sh[1]), h.ptr(x), h.ptr(ref), B, H, W, C, C, 9, 1, st))
        for k in range(n):
            y.fill_(1.0)
            y[B * (H + 1) * (W + 1) * C:] = 0
            h.check(lib.lad_f16_block_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(sc[0]), h.ptr(sh[0]), h.ptr(wts[1]), h.ptr(sc[1]), h.ptr(sh[1]),
                                          h.ptr(y), B, H, W, C, st), "lad_f16_block_fwd")
            if not torch.equal(y, ref):
                bad += 1
                print(f"block C={C}: launch {k} differs from the pair by {float((y.float() - ref.float()).abs().max())}", flush=True)
        print(f"lad_f16_block_fwd C={C} B={B} {H}x{W}: {n} launches compared", flush=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
