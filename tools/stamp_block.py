"""Diagnostic: where the waves of block_f16_strip_kernel (the fused residual block of the inference strips) spend their time -- in-kernel
s_memtime sums per phase of waves 0 and 7 (-DLAD_STAMP build into tools/liblad_stamp_f16.so, never the product).
    python tools/stamp_f16.py --build     (build container)        python tools/stamp_block.py     (GPU box)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
os.environ["LAD_HIP_LIB"] = os.environ.get("LAD_STAMP_LIB") or os.path.join(ROOT, "tools", "liblad_stamp_f16.so")
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import _hip as h
lib = h.lib(); st = h.stream_handle()
lib.lad_debug_read_f16p_stamps.restype = ctypes.c_int
lib.lad_debug_read_f16p_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
n, H, W, C = 8282, 10, 44, 64
g = torch.Generator().manual_seed(1)
wts = []
for k in range(2):
    w = torch.randn(C, C, 3, 3, generator=g) * 0.06
    wt = torch.zeros(int(lib.lad_f16_packed_weight_halfs(C, C, 9)), device="cuda", dtype=torch.float16)
    h.check(lib.lad_f16_pack_weights(h.ptr(w.cuda()), C, C, 9, h.ptr(wt), st))
    wts.append(wt)
scale = torch.ones(C, device="cuda"); shift = torch.zeros(C, device="cuda")
rows = n * (H + 1) * (W + 1) + W + 2
x = torch.zeros(rows * C, dtype=torch.float16)
x[:n * (H + 1) * (W + 1) * C].view(n, H + 1, W + 1, C)[:, 1:, 1:, :] = torch.randn(n, H, W, C, generator=g).half()
x = x.cuda(); y = torch.zeros_like(x)
names = ["first barrier of a convolution (+ the image's DMA before conv1)", "exposed fragment reads of a convolution's first tap",
         "tap barriers (8 per convolution)", "tap bodies of conv1 (9; waves 4-7 move the previous output in 8 of them)", "epilogues",
         "tap bodies of conv2 (9)"]
for _ in range(5):
    h.check(lib.lad_f16_block_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(scale), h.ptr(shift), h.ptr(wts[1]), h.ptr(scale), h.ptr(shift),
                                  h.ptr(y), n, H, W, C, st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    h.check(lib.lad_f16_block_fwd(h.ptr(x), h.ptr(wts[0]), h.ptr(scale), h.ptr(shift), h.ptr(wts[1]), h.ptr(scale), h.ptr(shift),
                                  h.ptr(y), n, H, W, C, st))
e1.record()
torch.cuda.synchronize()
print(f"{os.path.basename(os.environ['LAD_HIP_LIB'])}: {e0.elapsed_time(e1) / 10:.4f} ms per launch (stamped build)")
buf = np.zeros(256 * 16, np.uint64)
assert lib.lad_debug_read_f16p_stamps(buf.ctypes.data, 256 * 16) == 0
t = buf.reshape(256, 2, 8)[:, :, :6].astype(np.float64)
imgs = n / 256.0
print(f"{imgs:.2f} images per workgroup; s_memtime ticks per IMAGE (two convolutions), mean over workgroups")
for wv, nm in ((0, "wave 0 (weights)"), (1, "wave 7 (output stores)")):
    tot = t[:, wv].sum(axis=1).mean() / imgs
    print(f"  {nm}: {tot:8.0f} per image")
    for j, ph in enumerate(names):
        v = t[:, wv, j].mean() / imgs
        print(f"      {ph:66s} {v:8.0f}  {100 * v / tot:5.1f} %")
