"""Diagnostic: where the single workgroup of the train-mode head kernels spends its time (in-kernel s_memtime stamps of thread 0;
-DLAD_STAMP build into tools/liblad_stamp_head.so, never the product library).
    python tools/stamp_head.py --build     (build container)        python tools/stamp_head.py [--batch 512]    (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_head.so")
if "--build" in sys.argv:
    srcs = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DLAD_STAMP", "-Xclang", "-target-feature",
                           "-Xclang", "-packed-fp32-ops", "-shared", "-I", os.path.join(ROOT, "include"), "-o", LIB] + srcs)
    print("built", LIB); sys.exit(0)
os.environ["LAD_HIP_LIB"] = LIB
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import numpy as np, torch
import bench, synth, config, _hip as h
from utils import get_feat_extractor
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 512
lib = h.lib()
lib.lad_debug_read_head_stamps.restype = ctypes.c_int
lib.lad_debug_read_head_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
dev = torch.device("cuda", 0)
ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
m = bench._make_model(0.0, dev, degenerate_ok=True)
m.train(); m.engine.reset_optimizer()
feats = torch.empty((B, 100, 44), device=dev)
ex.extract_batch(synth.make_clips(B, seed=1234, device=dev), out=feats)
labels = synth.make_labels(B, seed=4321, device=dev)
for _ in range(5):
    m.engine.forward(feats, train=True, labels=labels)
    m.engine.backward(None)
torch.cuda.synchronize()
buf = np.zeros(64, np.uint64)
assert lib.lad_debug_read_head_stamps(buf.ctypes.data, 64) == 0
t = buf.astype(np.int64)
fwd = [("column statistics of pooled (bn2)", 0, 1), ("weights -> LDS, bn2 coefficients", 1, 2), ("linear1, one sample per thread", 2, 3),
       ("column statistics of h (bn3)", 3, 4), ("bn3 + relu + linear2 + sigmoid + BCE", 4, 5), ("metric sums", 5, 6)]
bwd = [("weights / coefficients -> LDS", 16, 17), ("stage 1: dlogit, du, gr", 17, 18), ("column sums: dW2, dbeta3 / dgamma3", 18, 19),
       ("stage 2: dh", 19, 20), ("stage 4: dz, one sample per thread", 20, 21), ("stage 3: dW1 (chunks through LDS)", 21, 22),
       ("column sums: dbias1", 22, 23), ("column sums: dbeta2 / dgamma2", 23, 24), ("stage 5: dpooled", 24, 25)]
for title, ph, a, b in (("head_fwd_train_kernel", fwd, 0, 6), ("head_bwd_kernel", bwd, 16, 25)):
    tot = t[b] - t[a]
    print(f"{title} at batch {B}: {tot} ticks of s_memtime (100 MHz: {tot / 100.0:.1f} us) between the first and the last stamp")
    for nm, i, j in ph:
        print(f"  {nm:48s} {t[j] - t[i]:7d} ticks = {100.0 * (t[j] - t[i]) / tot:5.1f} %")
