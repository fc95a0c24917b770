"""Masked-gradient parity at batch 512 under engine option sets (diagnostic): relative L2 per tensor, worst five.
    python tools/diag_parity512.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"),
                os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
from oracle import recipe, resnet_oracle as ro
from test_resnet_gpu import build_model, noise_grad
torch.set_num_threads(16)
B = 512
xf, tl = recipe.make_features(72, B), recipe.make_labels(73, B)
ref64 = None
for name, opts in (("default", {}), ("no fused sums", {"fuse_bn_bwd_b3": False}), ("exact f32 kernels", {"bf16x3": False})):
    m, sd = build_model(71)
    m.train()
    for k, v in opts.items():
        setattr(m.engine, k, v)
    eng = m.engine
    eng.forward(torch.from_numpy(xf).cuda(), train=True, labels=torch.from_numpy(tl).cuda())
    eng.backward(None)
    masks = eng.export_relu_masks()
    rm = ro.train_step(sd, torch.from_numpy(xf), torch.from_numpy(tl), relu_masks=masks)
    rows = []
    for k, gv in eng.grad_views().items():
        if noise_grad(k):
            continue
        ref = rm["grads"][k].double().numpy()
        rows.append((np.linalg.norm(gv.cpu().double().numpy() - ref) / np.linalg.norm(ref), k))
    rows.sort(reverse=True)
    print(name, " ".join(f"{k}:{v:.2e}" for v, k in rows[:5]), flush=True)
