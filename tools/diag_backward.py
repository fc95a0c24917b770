"""Diagnostic (GPU box): per-block backward intermediates of the HIP path vs a float64 CPU autograd reference."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT, os.path.join(ROOT, "tests")]
from oracle import recipe, resnet_oracle as ro
from test_resnet_gpu import build_model, from_pnhwc

def ref_run(sd, x, t, dtype):
    sd = {k: v.to(dtype).clone().requires_grad_(v.dtype.is_floating_point and not ('running' in k)) for k, v in sd.items()}
    inter = {}
    def keep(name, v):
        v.retain_grad(); inter[name] = v; return v
    def bn(v, p):
        return F.batch_norm(v, None, None, sd[p + ".weight"], sd[p + ".bias"], training=True, eps=1e-5)
    out = keep("stem.c", F.conv2d(x.to(dtype), sd["conv1.weight"], None, padding=1))
    out = keep("stem.a", F.relu(bn(out, "bn1")))
    for bi in range(1, 5):
        for j in range(2):
            p = f"block{bi}.{j}"; stride = 2 if (bi > 1 and j == 0) else 1
            c1 = keep(p + ".c1", F.conv2d(out, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=stride, padding=1))
            a1 = keep(p + ".a1", F.relu(bn(c1, p + ".bn1")))
            c2 = keep(p + ".c2", F.conv2d(a1, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1))
            z = bn(c2, p + ".bn2")
            if (p + ".shortcut.0.weight") in sd:
                cs = keep(p + ".cs", F.conv2d(out, sd[p + ".shortcut.0.weight"], None, stride=stride))
                z = z + bn(cs, p + ".shortcut.1")
            else:
                z = z + out
            out = keep(p + ".y", F.relu(z))
    o = F.avg_pool2d(out, 4).reshape(x.shape[0], -1)
    o = bn(o, "bn2"); o = F.linear(o, sd["linear1.weight"], sd["linear1.bias"]); o = F.relu(bn(o, "bn3"))
    probs = torch.sigmoid(F.linear(o, sd["linear2.weight"], sd["linear2.bias"])).squeeze(-1)
    loss = ro.bce_mean(probs, t.to(dtype)); loss.backward()
    return inter

g = np.load(os.path.join(ROOT, "tests/golden/resnet_train.npz")); B = int(g["batch"])
m, sd = build_model(int(g["state_seed"])); m.train()
xc = torch.from_numpy(recipe.make_features(int(g["feat_seed"]), B)); tc = torch.from_numpy(recipe.make_labels(int(g["label_seed"]), B))
eng = m.engine; eng.debug_capture = {}
eng.forward(xc.cuda(), train=True, labels=tc.cuda()); eng.backward(None)
r64 = ref_run(sd, xc, tc, torch.float64); r32 = ref_run(sd, xc, tc, torch.float32)
plan = eng._last_train_plan
def rel(a, ref): return float((a.double() - ref).abs().max() / ref.abs().max())
print("%-14s %-5s %10s %10s" % ("block", "what", "gpu-vs-64", "cpu32-vs-64"))
for b, a in zip(plan["blocks"], plan["acts"]):
    co, ho, wo = b.conv1.cout, b.conv1.h_out, b.conv1.w_out
    for nm, key in (("c1", "c1"), ("a1", "a1"), ("c2", "c2"), ("y", "y")):
        print("%-14s %-5s %10.2e %10.2e" % (b.name, nm, rel(from_pnhwc(a[key], B, co, ho, wo), r64[b.name + "." + nm].detach()),
              rel(r32[b.name + "." + nm].detach(), r64[b.name + "." + nm].detach())))
    cap = eng.debug_capture[b.name]
    for nm, key in (("dy", "y"), ("dc2", "c2"), ("da1", "a1"), ("dc1", "c1")):
        print("%-14s %-5s %10.2e %10.2e" % (b.name, nm, rel(from_pnhwc(cap[nm], B, co, ho, wo), r64[b.name + "." + key].grad),
              rel(r32[b.name + "." + key].grad, r64[b.name + "." + key].grad)))

# ---- isolate block1.0.bn2 backward: recompute it in float64 from the GPU's own inputs
print("\nblock1.0.bn2 backward recomputed in float64 from the GPU's inputs")
b, a = plan["blocks"][0], plan["acts"][0]
cap = eng.debug_capture["block1.0"]
co, ho, wo = 64, 100, 44
dy = from_pnhwc(cap["dy"], B, co, ho, wo).double(); y = from_pnhwc(a["y"], B, co, ho, wo).double()
c2 = from_pnhwc(a["c2"], B, co, ho, wo).double()
dz = dy * (y > 0)
N = B * ho * wo
mean = c2.mean((0, 2, 3), keepdim=True); var = c2.var((0, 2, 3), unbiased=False, keepdim=True)
xh = (c2 - mean) / torch.sqrt(var + 1e-5)
s0 = dz.sum((0, 2, 3)); s1 = (dz * xh).sum((0, 2, 3))
gam = sd["block1.0.bn2.weight"].double().view(1, -1, 1, 1)
dx = gam / torch.sqrt(var + 1e-5) * (dz - s0.view(1, -1, 1, 1) / N - xh * s1.view(1, -1, 1, 1) / N)
gv = eng.grad_views()
print("dbeta  gpu vs recomputed:", rel(gv["block1.0.bn2.bias"].cpu(), s0), " recomputed vs ref64:", rel(s0, r64["block1.0.c2"].grad.sum((0,2,3)) if False else s0))
print("dgamma gpu vs recomputed:", rel(gv["block1.0.bn2.weight"].cpu(), s1))
print("dc2    gpu vs recomputed:", rel(from_pnhwc(cap["dc2"], B, co, ho, wo), dx), " recomputed vs ref64:", rel(dx, r64["block1.0.c2"].grad))
coef = a["coef2"].view(6, 64).cpu().double()
print("mean  gpu vs recomputed:", rel(coef[2] + coef[4], mean.flatten()), " istd:", rel(coef[3] + coef[5], (1 / torch.sqrt(var + 1e-5)).flatten()))
print("max|mean|*istd:", float((mean.abs() / torch.sqrt(var + 1e-5)).max()), "max|dz|", float(dz.abs().max()), "max|dx|", float(dx.abs().max()), "max|xh|", float(xh.abs().max()))
# border rows of dy
full = cap["dy"].view(B, ho + 2, wo + 2, co)
print("dy border max:", float(full[:, 0].abs().max()), float(full[:, -1].abs().max()), float(full[:, :, 0].abs().max()), float(full[:, :, -1].abs().max()))
