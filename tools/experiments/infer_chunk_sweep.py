import sys, os, time
ROOT="/root/repo"
sys.path[:0]=[os.path.join(ROOT,"laughter-detection-icsi_amd","utils"), os.path.join(ROOT,"laughter-detection-icsi_amd"), ROOT]
import torch, bench, config, synth, engine
from utils import get_feat_extractor
dev=torch.device("cuda",0)
ex=get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
model=bench._make_model(0.0, dev, degenerate_ok=False); model.eval()
pcm=synth.make_clips(3600, seed=9876, device=dev).view(-1)
feats=ex.extract_long(pcm)
ref=None
for chunk in (4096, 8192, 16384, 32768):
    try:
        p=model.engine.predict_windows(feats, chunk=chunk, precision="fp16"); torch.cuda.synchronize()
        t0=time.perf_counter(); p=model.engine.predict_windows(feats, chunk=chunk, precision="fp16"); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        if ref is None: ref=p.clone()
        print(f"chunk {chunk}: {dt*1e3:.1f} ms, max |p - p(first chunk size)| = {float((p-ref).abs().max()):.2e}", flush=True)
    except Exception as e:
        print("chunk", chunk, "failed:", str(e)[:200])
