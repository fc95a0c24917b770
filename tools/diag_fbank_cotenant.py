"""Diagnostic: is the featuriser bit-reproducible while a second process keeps the same GPU busy?  (Round 4: the extended co-tenant
watch of tests/test_resnet_gpu.py found 8 of 300 featurisations of 256 clips differing from the first.)
    python tools/diag_fbank_cotenant.py [--passes 600] [--clips 256] [--kernel fast|general] [--no-peer]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=600)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--kernel", default="fast")
ap.add_argument("--no-peer", action="store_true")
a = ap.parse_args()
peer = None
if not a.no_peer:
    peer = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "diag_determinism.py"), "--inproc", "60000", "--batch", "64"],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
import time
time.sleep(12.0 if peer else 0.0)   # let the peer get busy before the first featurisation
import torch
import config, synth, _hip
from utils import get_feat_extractor
ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
if a.kernel == "general":
    ex.use_general_kernel(True)
pcm = synth.make_clips(a.clips, seed=77, device=torch.device("cuda"))
ref = ex.extract_batch(pcm).clone()
bad = 0
for i in range(a.passes):
    out = ex.extract_batch(pcm)
    if not torch.equal(out, ref):
        bad += 1
        d = (out != ref)
        idx = torch.nonzero(d)
        if bad <= 6:
            clips = sorted(set(idx[:, 0].tolist()))
            frames = sorted(set((idx[:, 0] * 100 + idx[:, 1]).tolist()))
            bins = sorted(set(idx[:, 2].tolist()))
            dv = (out - ref)[d]
            print(f"pass {i}: {int(d.sum())} values differ; clips {clips[:8]} ({len(clips)}); frames {len(frames)}: {[ (f // 100, f % 100) for f in frames[:10]]}; "
                  f"bins {bins[:50]}; max |delta| {float(dv.abs().max()):.3e}; ref there {ref[d][:4].tolist()} got {out[d][:4].tolist()}", flush=True)
print(f"[{a.kernel}{'' if peer else ', alone'}] {bad} of {a.passes} featurisations of {a.clips} clips differ from the first")
if peer:
    peer.kill()    # (our own child, by handle)
    peer.wait()
