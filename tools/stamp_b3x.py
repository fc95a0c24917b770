"""Diagnostic: where a conv_b3x workgroup spends its lifetime (in-kernel s_memtime stamps of wave 0; -DLAD_STAMP build into
tools/liblad_stamp_b3x.so, never the product library).
    python tools/stamp_b3x.py --build     (build container)        python tools/stamp_b3x.py     (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_b3x.so")
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
lib.lad_debug_read_b3x_stamps.restype = ctypes.c_int
lib.lad_debug_read_b3x_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
B, H, W, C = 512, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows * C, device="cuda", generator=g).relu_()
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
bias = torch.randn(C, device="cuda", generator=g)
out = torch.empty(rows * C, device="cuda")
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * C, device="cuda")
wt3 = torch.zeros(int(lib.lad_conv_b3_packed_weight_bytes()), device="cuda", dtype=torch.uint8)
h.check(lib.lad_conv_b3_pack_weights(h.ptr(w), 0, h.ptr(wt3), st))
for _ in range(20):
    h.check(lib.lad_conv_b3_fwd_f32(h.ptr(x), h.ptr(wt3), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, st))
torch.cuda.synchronize()
n = 9096
buf = np.zeros(16 * n, np.uint64)
assert lib.lad_debug_read_b3x_stamps(buf.ctypes.data, 16 * n) == 0
t = buf.reshape(n, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 12] > t[:, 0])]
names = [("prologue: issue loads, mask", 0, 1), ("wait first rows + split into LDS", 1, 2), ("stage 0 MFMA (9 taps)", 2, 3), ("transition 0->1", 3, 4),
         ("stage 1 MFMA", 4, 5), ("transition 1->2", 5, 6), ("stage 2 MFMA", 6, 7), ("transition 2->3", 7, 8), ("stage 3 MFMA", 8, 9),
         ("final barrier", 10, 11), ("epilogue (2 x 128 rows)", 11, 12)]
total = np.median(t[:, 12] - t[:, 0])
print(f"workgroups with stamps: {len(t)}; median lifetime {total:.0f} cycles")
for nm, a, b in names:
    d = t[:, b] - t[:, a]
    print(f"  {nm:36s} median {np.median(d):8.0f}  mean {d.mean():8.0f} cycles  = {100 * np.median(d) / total:5.1f} %")
