import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/laughter-detection-icsi_amd")
import torch
import bench, _hip
dev = torch.device("cuda:0")
model = bench._make_model(0.0, dev, degenerate_ok=False)
model.eval()
eng = model.engine
lib = eng.lib()
T = 360000
g = torch.Generator(device="cpu").manual_seed(1)
feats = (torch.randn(T, 44, generator=g) * 2 - 8).to(dev)
res = {}
for rnd in range(2):
    for wide in (0, 1):
        lib.lad_f16_set_dual_groups(wide)
        eng.predict_windows(feats, precision="fp16", stop=8192)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = eng.predict_windows(feats, precision="fp16")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[wide] = p.clone()
        print(f"wide={wide}: {dt*1e3:.1f} ms  {T/dt/1e6:.3f} M windows/s", flush=True)
print("identical:", bool(torch.equal(res[0], res[1])), float((res[0]-res[1]).abs().max()))
lib.lad_f16_set_dual_groups(1)
