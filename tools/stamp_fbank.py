"""Diagnostic: phase shares of fbank16_kernel from in-kernel s_memtime stamps.  Builds a SEPARATE library with -DLAD_STAMP
(tools/liblad_stamp_fb.so, never the product library) when hipcc is available, then runs on the GPU:

    python tools/stamp_fbank.py --build        (build container: cross-compile)
    python tools/stamp_fbank.py                (GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
LIB = os.path.join(ROOT, "tools", "liblad_stamp_fb.so")
if "--build" in sys.argv:
    srcs = [os.path.join(PKG, "csrc", f) for f in sorted(os.listdir(os.path.join(PKG, "csrc"))) if f.endswith(".hip")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-DLAD_STAMP", "-shared", "-I",
           os.path.join(ROOT, "include"), "-o", LIB] + srcs
    subprocess.check_call(cmd)
    print("built", LIB)
    sys.exit(0)
sys.path[:0] = [PKG, ROOT]
import numpy as np
import torch
import _hip as h
h.LIB_PATH = LIB
import feats
import synth
lib = h.lib()
lib.lad_debug_read_fbank_stamps.restype = ctypes.c_int
lib.lad_debug_read_fbank_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
ex = feats.HipFbank(feats.HipFbankConfig(num_filters=44, frame_shift=0.01))
pcm = synth.make_clips(1024, seed=1, device="cuda")
out = torch.empty((1024, 100, 44), device="cuda")
for _ in range(3):
    ex.extract_batch(pcm, out=out)
torch.cuda.synchronize()
n = 768
buf = np.zeros(16 * n, np.uint64)
assert lib.lad_debug_read_fbank_stamps(buf.ctypes.data, 16 * n) == 0
t = buf.reshape(n, 16).astype(np.int64)[:, :11]
t = t[t[:, 0] > 0]
names = ["commit", "barrier", "issue next loads", "front (mean, window)", "fft 1 + twiddle", "transposition", "fft 2",
         "split + power", "mel + log + store", "barrier 2"]
d = np.diff(t, axis=1)
print("workgroups with stamps:", len(t))
for k, nm in enumerate(names):
    print("  %-24s median %6d  mean %6d cycles" % (nm, np.median(d[:, k]), d[:, k].mean()))
print("  %-24s median %6d" % ("whole chunk", np.median(t[:, 10] - t[:, 0])))
