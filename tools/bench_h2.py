"""A/B of the 64 -> 64 3x3 convolution kernels in one process, interleaved rounds (boxes and processes differ by 2-4 %):
bf16 x 3 (conv_b3x, round 3) against f16 x 2 (conv_h2, variants 0 = 384-row tiles, 1 = 256-row tiles), plain forward launch,
and optionally the weight gradients.   python tools/bench_h2.py [--batch 512] [--iters 20] [--rounds 3] [--wgrad] [--channels 64]
Round 5: the variants live in tools/experiments/retired/ -- build them first and point the loader at that library:
    tools/exp_h2.sh VARIANTS && LAD_HIP_LIB=tools/libexp_h2_VARIANTS.so python tools/bench_h2.py
(with the product library only the arms b3x / h2 / wb3x / wh2 exist)."""
import argparse, os, struct, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--channels", type=int, default=64)
ap.add_argument("--H", type=int, default=100)
ap.add_argument("--W", type=int, default=44)
ap.add_argument("--wgrad", action="store_true")
ap.add_argument("--only", type=str, default=None, help="run just this arm (for rocprofv3): b3x | h2v0 | h2v1 | wb3x | wh2")
a = ap.parse_args()
lib = h.lib(); st = h.stream_handle()
B, H, W, C = a.batch, a.H, a.W, a.channels
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.relu(torch.randn(rows * C, device="cuda", generator=g) * 1.2 + 0.3)     # post-BatchNorm-ReLU statistics
dout = torch.randn(rows * C, device="cuda", generator=g) * 1e-3
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
bias = torch.randn(C, device="cuda", generator=g)
out = torch.empty(rows * C, device="cuda"); out2 = torch.empty(rows * C, device="cuda")
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * C, device="cuda")
wt3 = torch.zeros(int(lib.lad_conv_b3c_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
h.check(lib.lad_conv_b3c_pack_weights(h.ptr(w), 0, h.ptr(wt3), C, st))
wth = torch.zeros(int(lib.lad_conv_h2_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
table = torch.frombuffer(bytearray(struct.pack("<QQii", w.data_ptr(), wth.data_ptr(), 0, 0)), dtype=torch.uint8).cuda()
h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 1, C, st))
ws = torch.zeros(int(lib.lad_conv_wgrad_b3c_workspace_floats(C)), device="cuda")
dw = torch.zeros(C, C, 3, 3, device="cuda"); db = torch.zeros(C, device="cuda")
dw2 = torch.zeros(C, C, 3, 3, device="cuda"); db2 = torch.zeros(C, device="cuda")

def b3x():
    h.check(lib.lad_conv_b3c_fwd_f32(h.ptr(x), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, C, st))
def h2():
    h.check(lib.lad_conv_h2(h.ptr(x), None, h.ptr(wth), h.ptr(bias), None, None, h.ptr(out2), h.ptr(part), None, None, None, B, H, W, C, st))
HAVE_VARIANTS = hasattr(lib, "lad_conv_h2_set_variant")
def h2v(v):
    def f():
        h.check(lib.lad_conv_h2_set_variant(v)); h2()
    return f
def wb3x():
    h.check(lib.lad_conv_wgrad_b3c(h.ptr(x), None, h.ptr(dout), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, C, st))
def wh2():
    h.check(lib.lad_conv_wgrad_h2(h.ptr(x), None, h.ptr(dout), h.ptr(ws), h.ptr(dw2), h.ptr(db2), B, H, W, C, st))
arms = {"b3x": b3x, "h2": h2}
if HAVE_VARIANTS:
    arms = {"b3x": b3x, "h2v0": h2v(0), "h2v1": h2v(1), "h2v2": h2v(2), "h2v3": h2v(3), "h2v4": h2v(4), "h2v5": h2v(5), "h2v6": h2v(6)}
if a.wgrad:
    arms = {"wb3x": wb3x, "wh2": wh2}
if a.only:
    arms = {a.only: dict(b3x=b3x, h2=h2, h2v0=h2v(0), h2v1=h2v(1), h2v2=h2v(2), h2v3=h2v(3), h2v4=h2v(4), h2v5=h2v(5), h2v6=h2v(6), wb3x=wb3x, wh2=wh2)[a.only]}
flop = 2.0 * B * H * W * C * C * 9
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
        ms = timed(run)
        print(f"round {rnd} {name:5s} {ms:.4f} ms  {flop / ms * 1e-9:.1f} TFLOP/s algorithmic", flush=True)
if not a.only:
    if a.wgrad:
        print("max |dw_h2 - dw_b3x| / max|dw| =", float((dw2 - dw).abs().max() / dw.abs().max()))
    else:
        print("max |out_h2 - out_b3x| / max|out| =", float((out2 - out).abs().max() / out.abs().max()))
