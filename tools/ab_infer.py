"""A/B of the fp16 sliding-window pass (60-minute channel) under engine switches, in one process, interleaved rounds:
    python tools/ab_infer.py [--minutes 60] [--rounds 3] [--switch tail_fused]      (GPU box)
prints seconds per pass with the switch on / off and checks that the probabilities are identical."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=60.0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--switch", type=str, default="tail_fused")
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = bench._make_model(0.5, dev, False)
model.eval()
eng = model.engine
T = int(a.minutes * 6000)
g = torch.Generator(device="cuda").manual_seed(3)
feats = torch.randn(T, 44, device=dev, generator=g) * 2.0 - 8.0
out = {}
for on in (True, False):
    setattr(eng, a.switch, on)
    out[on] = eng.predict_windows(feats, precision="fp16").clone()
torch.cuda.synchronize()
print(f"{a.switch}: identical probabilities: {torch.equal(out[True], out[False])}  max |dp| = {float((out[True] - out[False]).abs().max()):.3e}", flush=True)
for rnd in range(a.rounds):
    for on in (True, False):
        setattr(eng, a.switch, on)
        eng.predict_windows(feats, precision="fp16")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.predict_windows(feats, precision="fp16")
        torch.cuda.synchronize()
        print(f"round {rnd} {a.switch}={on}: {(time.perf_counter() - t0) / 3:.4f} s per {a.minutes:g}-minute pass", flush=True)
setattr(eng, a.switch, True)
