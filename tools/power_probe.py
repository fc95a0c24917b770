"""Is conv_h2 / wgrad_h2 held back by POWER or by its schedule?  (Round 5)
The matrix pipe takes the same cycles on any operands (MI355X_MICROARCH.md, Matrix cores) but draws far less power on zeros, so the
chip holds a higher clock: the same launch timed on the step-like random data, on an all-zero input (weights random) and on all-zero
input + weights.  If the launch gets much faster on zeros, its pace on real data is set by power; if not, by its schedule.
    python tools/power_probe.py [--batch 512] [--iters 30]"""
import argparse, os, struct, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
lib = h.lib(); st = h.stream_handle()
B, H, W, C = a.batch, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x_r = torch.relu(torch.randn(rows * C, device="cuda", generator=g) * 1.2 + 0.3)
d_r = torch.randn(rows * C, device="cuda", generator=g) * 1e-3
w_r = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
x_0, d_0, w_0 = torch.zeros_like(x_r), torch.zeros_like(d_r), torch.zeros_like(w_r)
bias = torch.randn(C, device="cuda", generator=g)
out = torch.empty(rows * C, device="cuda")
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * C, device="cuda")
ws = torch.zeros(int(lib.lad_conv_wgrad_b3c_workspace_floats(C)), device="cuda")
dw = torch.zeros(C, C, 3, 3, device="cuda"); db = torch.zeros(C, device="cuda")
def packed(w):
    wt = torch.zeros(int(lib.lad_conv_h2_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
    table = torch.frombuffer(bytearray(struct.pack("<QQii", w.data_ptr(), wt.data_ptr(), 0, 0)), dtype=torch.uint8).cuda()
    h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 1, C, st))
    torch.cuda.synchronize()
    return wt
wt_r, wt_0 = packed(w_r), packed(w_0)
def conv(x, wt):
    return lambda: h.check(lib.lad_conv_h2(h.ptr(x), None, h.ptr(wt), h.ptr(bias), None, None, h.ptr(out), h.ptr(part), None, None, None, B, H, W, C, st))
def wgrad(x, d):
    return lambda: h.check(lib.lad_conv_wgrad_h2(h.ptr(x), None, h.ptr(d), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st))
def copy():   # a pure HBM pass of the same two tensors, for scale
    return lambda: out.copy_(x_r)
arms = {"conv  random x, random w": conv(x_r, wt_r), "conv  zero x,   random w": conv(x_0, wt_r), "conv  zero x,   zero w  ": conv(x_0, wt_0),
        "wgrad random x, random dy": wgrad(x_r, d_r), "wgrad zero x,   random dy": wgrad(x_0, d_r), "wgrad zero x,   zero dy  ": wgrad(x_0, d_0),
        "copy of one tensor       ": copy()}
def timed(run):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
for rnd in range(a.rounds):
    for name, run in arms.items():
        print(f"round {rnd} {name} {timed(run):.4f} ms", flush=True)
