"""Diagnostic: who waits for whom in conv_h2w_kernel (wave-specialised conv_h2; -DLAD_STAMP build of conv_h2.hip into
tools/liblad_stamp_h2w.so, never the product).   python tools/stamp_h2w.py --build (build container);  python tools/stamp_h2w.py (GPU box)"""
import ctypes, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_h2w.so")
if "--build" in sys.argv:
    obj = "/tmp/conv_h2_stamp.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DLAD_STAMP", "-Xclang", "-target-feature",
                           "-Xclang", "-packed-fp32-ops", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(PKG, "csrc", "conv_h2.hip"), "-o", obj])
    objs = [os.path.join(PKG, "csrc", "build", f) for f in os.listdir(os.path.join(PKG, "csrc", "build")) if f.endswith(".o") and f != "conv_h2.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [obj])
    print("built", LIB); sys.exit(0)
os.environ["LAD_HIP_LIB"] = LIB
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import _hip as h
lib = h.lib(); st = h.stream_handle()
lib.lad_debug_read_h2_stamps.restype = ctypes.c_int
lib.lad_debug_read_h2_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
B, H, W, C = 512, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows * C, device="cuda", generator=g).relu_()
if "--zeros" in sys.argv: x.zero_()
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
bias = torch.randn(C, device="cuda", generator=g)
out = torch.empty(rows * C, device="cuda")
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * C, device="cuda")
wt = torch.zeros(int(lib.lad_conv_h2_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
table = torch.frombuffer(bytearray(struct.pack("<QQii", w.data_ptr(), wt.data_ptr(), 0, 0)), dtype=torch.uint8).cuda()
h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 1, C, st))
for _ in range(10):
    h.check(lib.lad_conv_h2(h.ptr(x), None, h.ptr(wt), h.ptr(bias), None, None, h.ptr(out), h.ptr(part), None, None, None, B, H, W, C, st))
torch.cuda.synchronize()
n = 256
buf = np.zeros(16 * n, np.uint64)
assert lib.lad_debug_read_h2_stamps(buf.ctypes.data, 16 * n) == 0
t = buf.reshape(n, 16).astype(np.float64)
t = t[t[:, 0] > 0]
nt = t[:, 3]
steps = nt * 18
print(f"workgroups {len(t)}, tiles per workgroup {np.median(nt):.0f}; per TAP (median over workgroups), in cycles; a tap's own MFMAs are 768")
print(f"  M wave 0: lifetime / step {np.median(t[:, 0] / steps):7.0f}   in barriers {np.median(t[:, 1] / steps):7.0f}   epilogue / step {np.median(t[:, 2] / steps):7.0f}")
print(f"  H wave 4: lifetime / step {np.median(t[:, 4] / steps):7.0f}   vmcnt waits {np.median(t[:, 5] / steps):7.0f}   in barriers {np.median(t[:, 6] / steps):7.0f}   epilogue barriers {np.median(t[:, 7] / steps):7.0f}")
