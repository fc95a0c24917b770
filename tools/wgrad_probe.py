"""What bounds wgrad_h2 (the 64 x 64 x 9 weight gradient on two f16 planes, with and without the BatchNorm backward inside)?
Times the four forms the step launches at batch 512 on random data and on zeros (power: see tools/power_probe.py); with
LAD_HIP_LIB pointing at an ablation build (tools/exp_wgrad.sh NOMFMA | NOLOAD | NOBN) the differences say where the time goes.
    python tools/wgrad_probe.py [--batch 512] [--iters 20]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()
lib = h.lib(); st = h.stream_handle()
B, H, W, C = a.batch, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(scale=1.0, shift=0.0): return torch.randn(rows * C, device="cuda", generator=g) * scale + shift
data = {"random": (torch.relu(rnd(1.2, 0.3)), rnd(1e-3), rnd(2.0, 1.0)), "zeros": (torch.zeros(rows * C, device="cuda"),) * 3}
coef = torch.zeros(6 * C, device="cuda"); coef[:C] = 1.0; coef[3 * C:4 * C] = 1.0
in_coef = coef.clone()
bcoef = torch.zeros(8 * C, device="cuda"); bcoef[:C] = 1.0
bits = torch.randint(-2**62, 2**62, (rows,), device="cuda", dtype=torch.int64)
dc = torch.empty(rows * C, device="cuda")
ws = torch.zeros(int(lib.lad_conv_wgrad_b3c_workspace_floats(C)), device="cuda")
dw = torch.zeros(C * C * 9, device="cuda"); db = torch.zeros(C, device="cuda")
def arms(x, dy, cx):
    return {
        "wgrad_h2 plain                ": lambda: h.check(lib.lad_conv_wgrad_h2(h.ptr(x), None, h.ptr(dy), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st)),
        "wgrad_h2 input BatchNorm      ": lambda: h.check(lib.lad_conv_wgrad_h2(h.ptr(x), h.ptr(in_coef), h.ptr(dy), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st)),
        "wgrad_h2<false,2> bn backward ": lambda: h.check(lib.lad_conv_wgrad_h2_bnbwd(h.ptr(x), None, h.ptr(dy), h.ptr(cx), None, h.ptr(coef), h.ptr(bcoef), h.ptr(dc), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st)),
        "wgrad_h2<true,3> in BN + bits ": lambda: h.check(lib.lad_conv_wgrad_h2_bnbwd(h.ptr(x), h.ptr(in_coef), h.ptr(dy), h.ptr(cx), h.ptr(bits), h.ptr(coef), h.ptr(bcoef), h.ptr(dc), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st)),
    }
def timed(run):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
for rnd_ in range(a.rounds):
    for dname, (x, dy, cx) in data.items():
        for name, run in arms(x, dy, cx).items():
            print(f"round {rnd_} {dname:6s} {name} {timed(run):.4f} ms", flush=True)
