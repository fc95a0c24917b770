"""Does the 4-convolution chain of the level-1 strips run faster per strip when its tensors fit the 256 MB memory-side cache?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import torch
import _hip as h
lib = h.lib(); st = h.stream_handle()
H, W, C = 10, 44, 64
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.04
wt = torch.zeros(int(lib.lad_f16_packed_weight_halfs(C, C, 9)), device="cuda", dtype=torch.float16)
h.check(lib.lad_f16_pack_weights(h.ptr(w), C, C, 9, h.ptr(wt), st))
scale = torch.ones(C, device="cuda"); shift = torch.zeros(C, device="cuda")
for n in (256, 512, 1024, 2048, 4096, 8282):
    rows = int(lib.lad_act_rows(n, H, W))
    bufs = [torch.zeros(rows * C, device="cuda", dtype=torch.float16) for _ in range(4)]
    bufs[0].normal_(generator=g)
    def chain():
        a0, a1, a2, a3 = bufs
        h.check(lib.lad_f16_conv_fwd(h.ptr(a0), h.ptr(wt), h.ptr(scale), h.ptr(shift), None, h.ptr(a1), n, H, W, C, C, 9, 1, st))
        h.check(lib.lad_f16_conv_fwd(h.ptr(a1), h.ptr(wt), h.ptr(scale), h.ptr(shift), h.ptr(a0), h.ptr(a2), n, H, W, C, C, 9, 1, st))
        h.check(lib.lad_f16_conv_fwd(h.ptr(a2), h.ptr(wt), h.ptr(scale), h.ptr(shift), None, h.ptr(a3), n, H, W, C, C, 9, 1, st))
        h.check(lib.lad_f16_conv_fwd(h.ptr(a3), h.ptr(wt), h.ptr(scale), h.ptr(shift), h.ptr(a2), h.ptr(a1), n, H, W, C, C, 9, 1, st))
    for _ in range(3): chain()
    torch.cuda.synchronize()
    reps = max(4, 40000 // n)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): chain()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"n={n:5d} strips ({rows * C * 2 / 1e6:7.1f} MB per tensor): chain of 4 convs {ms * 1e3:8.1f} us = {ms * 1e6 / n / 4:6.1f} ns per strip and conv", flush=True)
