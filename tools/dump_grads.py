"""Diagnostic: dump the HIP gradients of the golden train step to gpurun_out/grads.npz (analysed on CPU vs float64)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT, os.path.join(ROOT, "tests")]
from oracle import recipe
from test_resnet_gpu import build_model
g = np.load(os.path.join(ROOT, "tests/golden/resnet_train.npz"))
B = int(g["batch"])
m, sd = build_model(int(g["state_seed"]))
m.train()
x = torch.from_numpy(recipe.make_features(int(g["feat_seed"]), B)).cuda()
t = torch.from_numpy(recipe.make_labels(int(g["label_seed"]), B)).cuda()
eng = m.engine
eng.forward(x, train=True, labels=t)
eng.backward(None)
out = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "grads.npz"), **out)
print("dumped", len(out))
