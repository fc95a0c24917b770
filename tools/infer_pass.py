"""The fp16 sliding-window pass over a synthetic 60-minute channel, N times (for rocprofv3 --kernel-trace --stats):
    python tools/infer_pass.py [--minutes 60] [--passes 3] [--no-tail]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=60.0)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--no-tail", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = bench._make_model(0.5, dev, False)
model.eval()
eng = model.engine
eng.tail_fused = not a.no_tail
T = int(a.minutes * 6000)
g = torch.Generator(device="cuda").manual_seed(3)
feats = torch.randn(T, 44, device=dev, generator=g) * 2.0 - 8.0
eng.predict_windows(feats, precision="fp16")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.passes):
    eng.predict_windows(feats, precision="fp16")
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / a.passes:.4f} s per {a.minutes:g}-minute pass ({a.passes} passes after one warm-up pass)")
