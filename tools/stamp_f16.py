"""Diagnostic: where the waves of conv_f16_s1p_kernel (the 64->64 half-precision convolution of the inference strips) spend their
time (in-kernel s_memtime sums per phase of waves 0 and 7; -DLAD_STAMP build into tools/liblad_stamp_f16.so, never the product).
    python tools/stamp_f16.py --build     (build container)        python tools/stamp_f16.py     (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_f16.so")
if "--build" in sys.argv:
    srcs = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DLAD_STAMP", "-shared", "-I",
                           os.path.join(ROOT, "include"), "-o", LIB] + srcs)
    print("built", LIB); sys.exit(0)
os.environ["LAD_HIP_LIB"] = LIB
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import _hip as h
lib = h.lib(); st = h.stream_handle()
lib.lad_debug_read_f16p_stamps.restype = ctypes.c_int
lib.lad_debug_read_f16p_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
n, H, W, C = 8282, 10, 44, 64
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.04
wt = torch.zeros(int(lib.lad_f16_packed_weight_halfs(C, C, 9)), device="cuda", dtype=torch.float16)
h.check(lib.lad_f16_pack_weights(h.ptr(w), C, C, 9, h.ptr(wt), st))
scale = torch.ones(C, device="cuda"); shift = torch.zeros(C, device="cuda")
rows = int(lib.lad_act_rows(n, H, W))
a0 = torch.zeros(rows * C, device="cuda", dtype=torch.float16).normal_(generator=g)
a1 = torch.zeros_like(a0); a2 = torch.zeros_like(a0)
names = ["wait at the top barrier", "rows regs -> LDS, mask, barrier", "issue next tile's loads + residual loads", "MFMA loop (9 taps)",
         "barrier after the loop", "epilogue (2 passes)"]
for add in (None, a0):
    for _ in range(5):
        h.check(lib.lad_f16_conv_fwd(h.ptr(a1), h.ptr(wt), h.ptr(scale), h.ptr(shift), h.ptr(add), h.ptr(a2), n, H, W, C, C, 9, 1, st))
    torch.cuda.synchronize()
    buf = np.zeros(256 * 16, np.uint64)
    assert lib.lad_debug_read_f16p_stamps(buf.ctypes.data, 256 * 16) == 0
    t = buf.reshape(256, 2, 8)[:, :, :6].astype(np.float64)
    tiles = (rows + 255) // 256 / 256.0
    print(f"residual {'yes' if add is not None else 'no'}: {tiles:.1f} tiles per workgroup; cycles per tile (s_memtime ticks), mean over workgroups")
    for wv, nm in ((0, "wave 0"), (1, "wave 7")):
        tot = t[:, wv].sum(axis=1).mean() / tiles
        print(f"  {nm}: {tot:8.0f} per tile")
        for j, ph in enumerate(names):
            v = t[:, wv, j].mean() / tiles
            print(f"      {ph:42s} {v:8.0f}  {100 * v / tot:5.1f} %")
