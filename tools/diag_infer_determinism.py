"""Diagnostic: sliding-window inference (fp16 and fp32, streaming path) and the feature kernel, call to call, next to a second process."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
for p in (os.path.join(PKG, "utils"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench, config, synth
from utils import get_feat_extractor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SECONDS = int(sys.argv[2]) if len(sys.argv) > 2 else 40   # 300: several groups of 8,192 windows, the run-long streams of levels 1 and 2
dev = torch.device("cuda", 0)
ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
pcm = synth.make_clips(64, seed=7, device=dev)
f0 = ex.extract_batch(pcm).clone()
bad = sum(0 if torch.equal(ex.extract_batch(pcm), f0) else 1 for _ in range(N * 5))
print(f"fbank: {bad} of {N * 5} calls differ")
m = bench._make_model(0.0, dev, degenerate_ok=False)
m.eval()
feats = ex.extract_long(synth.make_clips(SECONDS, seed=9, device=dev).view(-1))
for prec in ("fp16", "fp32"):
    ref = m.engine.predict_windows(feats, precision=prec).clone()
    bad = sum(0 if torch.equal(m.engine.predict_windows(feats, precision=prec), ref) else 1 for _ in range(N))
    print(f"predict_windows {prec} ({feats.shape[0]} windows): {bad} of {N} calls differ")
