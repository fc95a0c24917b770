"""Diagnostic: host-side enqueue time per training step (Python + ctypes + launch) vs GPU time."""
import contextlib, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import config, synth
from utils import get_feat_extractor
cfg = config.MODEL_MAP["resnet_base"]
with contextlib.redirect_stdout(io.StringIO()):
    model = cfg["model"](dropout_rate=0.5, linear_layer_size=48, filter_sizes=cfg["filter_sizes"])
model.set_device("cuda"); model.train(); model.engine.reset_optimizer()
ex = get_feat_extractor(100, 44)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pcm = synth.make_clips(B); labels = synth.make_labels(B); feats = torch.empty((B, 100, 44), device="cuda")
def step():
    ex.extract_batch(pcm, out=feats); return model.train_step(feats, labels)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, total {1e3 * (t2 - t0) / 20:.2f} ms/step")

# the same step replayed from a hipGraph (features already extracted: graph includes the fbank launch)
model.engine.reset_optimizer()
gstep = model.make_graphed_train_step(B, extractor=ex)
for _ in range(3): gstep(pcm, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): gstep(pcm, labels)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"B={B}: graphed {1e3 * (t1 - t0) / 20:.2f} ms/step = {B * 20 / (t1 - t0):.0f} segments/s")
