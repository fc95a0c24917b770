"""Diagnostic: where the waves of tail_f16_kernel (block3 + block4 + classifier of the fp16 sliding-window path, one window resident in LDS)
spend their time -- in-kernel s_memtime sums per step (work / wait at the barrier behind it) of waves 0, 10, 14 and 15
(-DLAD_STAMP build of the whole library into tools/liblad_stamp_tail.so, never the product).
    python tools/stamp_tail.py --build     (build container)        python tools/stamp_tail.py     (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_tail.so")
if "--build" in sys.argv:
    import concurrent.futures as cf
    srcs = [f for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip")]
    os.makedirs("/tmp/stamp_tail", exist_ok=True)
    def one(f):
        o = os.path.join("/tmp/stamp_tail", f[:-4] + ".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Xclang", "-target-feature", "-Xclang",
                               "-packed-fp32-ops", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(PKG, "csrc", f), "-o", o]
                              + (["-DLAD_STAMP"] if f == "tail_f16.hip" else []), stderr=subprocess.DEVNULL)
        return o
    with cf.ThreadPoolExecutor(6) as ex:
        objs = list(ex.map(one, srcs))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB); sys.exit(0)
os.environ["LAD_HIP_LIB"] = LIB
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import _hip as h
import bench
lib = h.lib()
lib.lad_debug_read_tail_stamps.restype = ctypes.c_int
lib.lad_debug_read_tail_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
dev = torch.device("cuda", 0)
model = bench._make_model(0.5, dev, False)
model.eval()
eng = model.engine
T = 8192
g = torch.Generator(device="cuda").manual_seed(3)
feats = torch.randn(T + 99, 44, device=dev, generator=g) * 2.0 - 8.0
for _ in range(3):
    eng.predict_windows(feats, precision="fp16", stop=T)      # ONE group of 8192 windows: the stamps are those of the last launch
torch.cuda.synchronize()
buf = np.zeros(256 * 4 * 24, np.uint64)
assert lib.lad_debug_read_tail_stamps(buf.ctypes.data, buf.size) == 0
t = buf.reshape(256, 4, 24).astype(np.float64)
wins = T / 256.0
names = ["loop top", "wait for the window's DMA", "top barrier",
         "step 1: A b3.0 conv1 + shortcut | B b4.0 conv1 + shortcut | C pooling", "   barrier behind it",
         "step 2: A b3.0 conv2 | B b4.0 conv2 | C hidden layer | DMA share", "   barrier behind it",
         "step 3: A b3.1 conv1 | B b4.1 conv1 | C output | DMA share", "   barrier behind it",
         "step 4: A b3.1 conv2 -> classes | B b4.1 conv2 | DMA share"]
print(f"{wins:.0f} windows per workgroup; s_memtime ticks (100 MHz: x ~21 = shader cycles at 2.1 GHz) per WINDOW, mean over 256 workgroups")
for wv, nm in enumerate(["wave 0 (A)", "wave 10 (B)", "wave 14 (DMA)", "wave 15 (C + DMA)"]):
    tot = t[:, wv, :10].sum(axis=1).mean() / wins
    print(f"  {nm}: {tot:8.1f} per window")
    for j, ph in enumerate(names):
        v = t[:, wv, j].mean() / wins
        print(f"      {ph:86s} {v:8.1f}  {100 * v / tot:5.1f} %")
