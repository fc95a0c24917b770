"""Diagnostic: where a conv_h2 workgroup spends its lifetime (in-kernel s_memtime stamps of wave 0; -DLAD_STAMP build into
tools/liblad_stamp_h2.so, never the product library).
    python tools/stamp_h2.py --build     (build container)        python tools/stamp_h2.py [--variant 1]    (GPU box)"""
import ctypes, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_h2.so")
if "--build" in sys.argv:
    srcs = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DLAD_STAMP", "-shared", "-I",
                           os.path.join(ROOT, "include"), "-o", LIB] + srcs)
    print("built", LIB); sys.exit(0)
if "--lib" in sys.argv:
    LIB = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
os.environ["LAD_HIP_LIB"] = LIB
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import _hip as h
lib = h.lib(); st = h.stream_handle()
lib.lad_debug_read_h2_stamps.restype = ctypes.c_int
lib.lad_debug_read_h2_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 1
if hasattr(lib, "lad_conv_h2_set_variant"):   # (only the retired-variants experiment library has the knob: tools/exp_h2.sh)
    lib.lad_conv_h2_set_variant.restype, lib.lad_conv_h2_set_variant.argtypes = ctypes.c_int, [ctypes.c_int]
    h.check(lib.lad_conv_h2_set_variant(variant))
elif "--variant" in sys.argv:
    sys.exit("--variant needs a library with lad_conv_h2_set_variant (tools/exp_h2.sh VARIANTS; --lib tools/libexp_h2_VARIANTS.so)")
B, H, W, C = 512, 100, 44, 64
rows = int(lib.lad_act_rows(B, H, W))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows * C, device="cuda", generator=g).relu_()
w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
bias = torch.randn(C, device="cuda", generator=g)
out = torch.empty(rows * C, device="cuda")
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * C, device="cuda")
wt = torch.zeros(int(lib.lad_conv_h2_packed_weight_bytes(C)), device="cuda", dtype=torch.uint8)
table = torch.frombuffer(bytearray(struct.pack("<QQii", w.data_ptr(), wt.data_ptr(), 0, 0)), dtype=torch.uint8).cuda()
h.check(lib.lad_conv_h2_pack_weights_multi(h.ptr(table), 1, C, st))
for _ in range(20):
    h.check(lib.lad_conv_h2(h.ptr(x), None, h.ptr(wt), h.ptr(bias), None, None, h.ptr(out), h.ptr(part), None, None, None, B, H, W, C, st))
torch.cuda.synchronize()
n = 9096
buf = np.zeros(16 * n, np.uint64)
assert lib.lad_debug_read_h2_stamps(buf.ctypes.data, 16 * n) == 0
t = buf.reshape(n, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 12] > t[:, 0])]
names = [("prologue: issue loads, mask", 0, 1), ("wait first rows + max + split into LDS", 1, 2), ("stage 0 MFMA (9 taps)", 2, 3), ("transition 0->1", 3, 4),
         ("stage 1 MFMA", 4, 5), ("final barrier", 5, 11), ("epilogue", 11, 12)]
total = np.median(t[:, 12] - t[:, 0])
print(f"variant {variant}: workgroups with stamps: {len(t)}; median lifetime {total:.0f} cycles (s_memtime: 100 MHz? see ratio to MFMA cycles)")
for nm, a, b in names:
    d = t[:, b] - t[:, a]
    print(f"  {nm:42s} median {np.median(d):8.0f}  mean {d.mean():8.0f} cycles  = {100 * np.median(d) / total:5.1f} %")
