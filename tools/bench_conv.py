"""Micro-benchmark of single kernels through the C ABI (used under rocprofv3 for PMC collection).
    python tools/bench_conv.py [conv|convb3|convb3f|convb3c|split3|wgrad|wgradb3|wgradb3c] [--batch 512] [--iters 10] [--cin 32 --cout 32 --H 50 --W 22]"""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="?", default="conv")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--H", type=int, default=100)
ap.add_argument("--W", type=int, default=44)
ap.add_argument("--variant", type=int, nargs="*", default=None, help="lad_conv_b3_set_variant values to compare (interleaved rounds)")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
lib = h.lib(); st = h.stream_handle()
B, H, W, cin, cout = a.batch, a.H, a.W, a.cin, a.cout
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows * cin, device="cuda", generator=g)
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
# Arms that need entry points retired from the product library in round 5 (lad_split3*, lad_conv_b3_fwd, *_set_variant) run only
# against an experiment library that still exports them (tools/exp_retired.sh; LAD_HIP_LIB=...): their signatures are registered
# here, since _hip.SIGNATURES lists the product ABI only (unregistered ctypes calls would truncate int64 arguments to 32 bits).
_c, _v, _i64, _i32 = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
_RETIRED = {"lad_split3_bytes": (_i64, [_i64, _i32]), "lad_split3": (_c, [_v, _v, _i64, _i32, _v]),
            "lad_conv_b3_fwd": (_c, [_v] * 6 + [_i64, _i32, _i32, _v]),
            "lad_conv_b3_set_variant": (_c, [_c]), "lad_conv_wgrad_b3_set_variant": (_c, [_c])}
_NEEDS = {"convb3": ["lad_split3_bytes", "lad_split3", "lad_conv_b3_fwd"], "split3": ["lad_split3_bytes", "lad_split3"],
          "convb3f": ["lad_split3_bytes", "lad_split3"]}
_need = list(_NEEDS.get(a.what, [])) + ((["lad_conv_wgrad_b3_set_variant"] if a.what.startswith("wgrad") else ["lad_conv_b3_set_variant"]) if a.variant else [])
for _name in _need:
    if not hasattr(lib, _name):
        sys.exit(f"bench_conv.py {a.what}{' --variant' if a.variant else ''}: {_name} is not in {h.LIB_PATH} -- it was retired from the "
                 "product library in round 5; build tools/exp_retired.sh and set LAD_HIP_LIB to that library")
    getattr(lib, _name).restype, getattr(lib, _name).argtypes = _RETIRED[_name]
bias = torch.randn(cout, device="cuda", generator=g)
out = torch.empty(rows * cout, device="cuda")
wt = torch.zeros(int(lib.lad_conv_packed_weight_floats(cout, cin, 9, 0)), device="cuda")
h.check(lib.lad_conv_pack_weights(h.ptr(w), cout, cin, 9, 0, h.ptr(wt), st))
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * cout, device="cuda")
ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(cin, cout, 9)), device="cuda")
dw = torch.zeros(cout, cin, 3, 3, device="cuda"); db = torch.zeros(cout, device="cuda")
dout = torch.randn(rows * cout, device="cuda", generator=g)
if a.what == "convb3c":
    wt3 = torch.zeros(int(lib.lad_conv_b3c_packed_weight_bytes(cin)), device="cuda", dtype=torch.uint8)
    h.check(lib.lad_conv_b3c_pack_weights(h.ptr(w), 0, h.ptr(wt3), cin, st))
if a.what in ("convb3", "convb3f", "split3"):
    xs = torch.zeros(int(lib.lad_split3_bytes(rows, cin)), device="cuda", dtype=torch.uint8)
    h.check(lib.lad_split3(h.ptr(x), h.ptr(xs), rows, cin, st))
    wt3 = torch.zeros(int(lib.lad_conv_b3_packed_weight_bytes()), device="cuda", dtype=torch.uint8)
    h.check(lib.lad_conv_b3_pack_weights(h.ptr(w), 0, h.ptr(wt3), st))
if a.what == "convb3c":
    wt3 = torch.zeros(int(lib.lad_conv_b3c_packed_weight_bytes(cin)), device="cuda", dtype=torch.uint8)
    h.check(lib.lad_conv_b3c_pack_weights(h.ptr(w), 0, h.ptr(wt3), cin, st))
if a.what == "wgradb3c":
    wsc = torch.zeros(int(lib.lad_conv_wgrad_b3c_workspace_floats(cin)), device="cuda")
def run():
    if a.what == "convb3":
        h.check(lib.lad_conv_b3_fwd(h.ptr(xs), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, st))
    elif a.what == "convb3f":
        h.check(lib.lad_conv_b3_fwd_f32(h.ptr(x), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, st))
    elif a.what == "convb3c":
        h.check(lib.lad_conv_b3c_fwd_f32(h.ptr(x), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, cin, st))
    elif a.what == "convb3c":
        h.check(lib.lad_conv_b3c_fwd_f32(h.ptr(x), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, cin, st))
    elif a.what == "split3":
        h.check(lib.lad_split3(h.ptr(x), h.ptr(xs), rows, cin, st))
    elif a.what == "conv":
        h.check(lib.lad_conv_fwd(h.ptr(x), h.ptr(wt), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, cin, cout, 9, st))
    elif a.what == "wgradb3c":
        h.check(lib.lad_conv_wgrad_b3c(h.ptr(x), None, h.ptr(dout), h.ptr(wsc), h.ptr(dw), h.ptr(db), B, H, W, cin, st))
    elif a.what == "wgradb3":
        h.check(lib.lad_conv_wgrad_b3(h.ptr(x), h.ptr(dout), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, st))
    elif a.what == "wgrad":
        h.check(lib.lad_conv_wgrad(h.ptr(x), h.ptr(dout), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, cin, cout, 9, st))
flop = 2.0 * B * H * W * cin * cout * 9
def timed():
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
if a.variant:   # interleaved rounds in one process (boxes and processes differ by 2-3 %)
    ref = None
    for rnd in range(a.rounds):
        for v in a.variant:
            h.check(lib.lad_conv_wgrad_b3_set_variant(v) if a.what.startswith("wgrad") else lib.lad_conv_b3_set_variant(v))
            ms = timed()
            if rnd == 0:   # every variant computes the same convolution: compare with the first one
                cur = (dw if a.what.startswith("wgrad") else out).clone()
                if ref is None: ref = cur
                err = (cur - ref).abs().max().item() / ref.abs().max().item()
                print(f"variant {v}: max |out - out(variant {a.variant[0]})| / max |out| = {err:.2e}")
            print(f"{a.what} variant {v} round {rnd} B={B} {cin}->{cout} {H}x{W}: {ms:.4f} ms/launch, {flop / ms / 1e9:.2f} TFLOP/s", flush=True)
else:
    ms = timed()
    print(f"{a.what} B={B} {cin}->{cout} {H}x{W}: {ms:.4f} ms/launch, {flop / ms / 1e9:.2f} TFLOP/s")
