"""The fused residual block on the sliding-window path's strips (lad_f16_block_fwd) against the two lad_f16_conv_fwd launches it
replaces: time per launch over the product's group of 8,282 strips of 10 x 44, HIP events around `reps` launches.
    [LAD_HIP_LIB=tools/libexp_<tag>.so] python tools/bench_block.py [--images 8282] [--reps 20] [--rounds 3] [--only fused]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "laughter-detection-icsi_amd"))
import _hip as h  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=8282)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default=None)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (no switching power: the schedule alone)")
    args = ap.parse_args()
    lib = h.lib()
    st = h.stream_handle()
    B, H, W, C = args.images, 10, 44, 64
    rows = B * (H + 1) * (W + 1) + W + 2
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(rows * C, dtype=torch.float16)
    if not args.zeros:
        x[:B * (H + 1) * (W + 1) * C].view(B, H + 1, W + 1, C)[:, 1:, 1:, :] = torch.randn(B, H, W, C, generator=g).half()
    x = x.cuda()
    wts = []
    for k in range(2):
        w = (torch.zeros if args.zeros else torch.randn)(C, C, 3, 3) * 0.06
        wt = torch.zeros(int(lib.lad_f16_packed_weight_halfs(C, C, 9)), device="cuda", dtype=torch.float16)
        h.check(lib.lad_f16_pack_weights(h.ptr(w.cuda()), C, C, 9, h.ptr(wt), st))
        wts.append(wt)
    sc = [torch.ones(C, device="cuda") for _ in range(2)]
    sh = [torch.zeros(C, device="cuda") for _ in range(2)]
    a1 = torch.zeros_like(x)
    y = torch.zeros_like(x)
    y2 = torch.zeros_like(x)

    def fused():
        h.check(lib.lad_f16_block_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(sc[0]), h.ptr(sh[0]), h.ptr(wts[1]), h.ptr(sc[1]), h.ptr(sh[1]),
                                      h.ptr(y), B, H, W, C, st), "lad_f16_block_fwd")

    def pair():
        h.check(lib.lad_f16_conv_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(sc[0]), h.ptr(sh[0]), None, h.ptr(a1), B, H, W, C, C, 9, 1, st))
        h.check(lib.lad_f16_conv_fwd(h.ptr(a1), h.ptr(wts[1]), h.ptr(sc[1]), h.ptr(sh[1]), h.ptr(x), h.ptr(y2), B, H, W, C, C, 9, 1, st))

    flop = 2 * 2 * B * H * W * C * C * 9
    for r in range(args.rounds):
        for name, fn in (("fused", fused), ("pair", pair)):
            if args.only and name != args.only:
                continue
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            print(f"round {r} {name:6s} {ms:.4f} ms  {flop / ms / 1e9:.1f} TFLOP/s algorithmic", flush=True)
    if not args.only:
        print("identical:", bool(torch.equal(y, y2)))


if __name__ == "__main__":
    main()
