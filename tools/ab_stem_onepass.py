"""A/B of engine.stem_onepass (the stem's statistics and backward from moments of the input) against the convolution passes it replaces:
coefficients, running statistics, the stem's gradients, the whole flat gradient, and the step time.  python tools/ab_stem_onepass.py"""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import bench

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, 100, 44, device=dev, generator=g) * 2.0 - 8.0
t = (torch.rand(B, device=dev, generator=g) > 0.5).int()
res = {}
for flag in (False, True):
    torch.manual_seed(11)
    model = bench._make_model(0.5, dev, False); model.train(); eng = model.engine
    eng.stem_onepass = flag
    eng.forward(x, train=True, labels=t, drop_masks=None)
    eng.backward(None)
    torch.cuda.synchronize()
    p = eng._last_train_plan
    res[flag] = dict(coef=p["stem_coef"].clone().double(), rm=eng.stem_bn.rm.clone().double(), rv=eng.stem_bn.rv.clone().double(),
                     gw=eng.stem_gw.clone().double(), gg=eng.stem_bn.gg.clone().double(), gb=eng.stem_bn.gb.clone().double(),
                     flat=eng.flat_grad().clone().double(), probs=p["probs"].clone().double())
a, b = res[False], res[True]
for k in a:
    d = (a[k] - b[k]).abs().max().item()
    rel = ((a[k] - b[k]).norm() / a[k].norm().clamp_min(1e-30)).item()
    print(f"{k:6s} max |diff| {d:.3e}   rel-L2 {rel:.3e}   (|ref| max {a[k].abs().max().item():.3e})", flush=True)
for flag in (False, True, False, True):
    model.engine.stem_onepass = flag
    for _ in range(5): model.train_step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): model.train_step(x, t)
    torch.cuda.synchronize()
    print("stem_onepass", flag, round((time.perf_counter() - t0) / 60 * 1e3, 3), "ms per step", flush=True)
