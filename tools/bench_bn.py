"""Micro-benchmark of the element-wise BatchNorm passes at the level-1 shape (64 x 100 x 44, batch 512) through the C ABI:
achieved HBM rate = algorithmic bytes (tensors read + written) / time.   python tools/bench_bn.py [--iters 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = h.lib(); st = h.stream_handle()
B, H, W, C = a.batch, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x, res, dy = (torch.randn(rows * C, device="cuda", generator=g) for _ in range(3))
y, dx = torch.zeros(rows * C, device="cuda"), torch.zeros(rows * C, device="cuda")
bits = torch.zeros(rows, device="cuda", dtype=torch.int64)
coef = torch.rand(6 * C, device="cuda") + 0.5
gam = torch.rand(C, device="cuda") + 0.5
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
ws = torch.zeros(int(lib.lad_bn_bwd_workspace_floats(C)), device="cuda")
bc = torch.zeros(8 * C, device="cuda")
T = B * (H + 1) * (W + 1) * C * 4
cases = {
    "bn_act (x -> y), 2 T": (2, lambda: lib.lad_bn_act(h.ptr(x), h.ptr(coef), None, None, h.ptr(y), B, H, W, C, 1, st)),
    "bn_act_bits (x, res -> y, bits), 3 T": (3, lambda: lib.lad_bn_act_bits(h.ptr(x), h.ptr(coef), h.ptr(res), None, h.ptr(y), h.ptr(bits), B, H, W, C, st)),
    "bn_bwd relu=2 (reduce 2 T + apply 3 T)": (5, lambda: lib.lad_bn_bwd(h.ptr(dy), None, h.ptr(x), h.ptr(coef), h.ptr(gam), None, None, None, h.ptr(dx), None,
                                                                     h.ptr(dg), h.ptr(db), None, None, h.ptr(ws), h.ptr(bc), None, 0, B, H, W, C, 2, 0, st)),
    "bn_bwd_bits (reduce 2 T + apply 3 T)": (5, lambda: lib.lad_bn_bwd_bits(h.ptr(dy), h.ptr(bits), h.ptr(x), h.ptr(coef), h.ptr(gam), h.ptr(dx), h.ptr(dg), h.ptr(db),
                                                                        h.ptr(ws), h.ptr(bc), None, 0, B, H, W, C, st)),
}
for name, (nt, fn) in cases.items():
    for _ in range(3): h.check(fn())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): h.check(fn())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name}: {ms * 1e3:.1f} us, {nt * T / ms / 1e9:.2f} TB/s", flush=True)
