"""A/B of an engine switch of the fp16 inference path (default engine.strip_block_fused; --flag f16_s2_shortcut_fused, stream_level2, ...)
on the 60-minute workload: wall time per pass, and whether the probabilities are identical.
    python tools/ab_strip_block.py [--flag strip_block_fused] [--minutes 60] [--rounds 3]"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "laughter-detection-icsi_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=60.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--flag", default="strip_block_fused")
    ap.add_argument("--chunk", type=int, default=None, help="windows per group (default: engine.PREDICT_CHUNK)")
    args = ap.parse_args()
    import bench
    m = bench._make_model(0.0, torch.device("cuda"), degenerate_ok=False)
    m.eval()
    eng = m.engine
    T = int(args.minutes * 60 * 100)
    g = torch.Generator().manual_seed(1)
    feats = (torch.randn(T, 44, generator=g) * 2.0 - 8.0).cuda()
    outs = {}
    for r in range(args.rounds):
        for fused in (False, True):
            setattr(eng, args.flag, fused)
            eng.predict_windows(feats, precision="fp16", stop=20000 if args.chunk is None else min(T, 2 * args.chunk + 100), chunk=args.chunk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            p = eng.predict_windows(feats, precision="fp16", chunk=args.chunk)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            outs[fused] = p.clone()
            print(f"round {r} {args.flag}={fused}: {dt * 1e3:.2f} ms  (RTF {dt / (args.minutes * 60):.3e})", flush=True)
    print("identical:", bool(torch.equal(outs[False], outs[True])))


if __name__ == "__main__":
    main()
