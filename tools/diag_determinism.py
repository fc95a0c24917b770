"""Diagnostic: are N fused train steps bit-reproducible from run to run -- also with another process on the same GPU?
    python tools/diag_determinism.py [--procs 2] [--steps 4] [--batch 64] [--repeat 3]
Every process runs the same seeded steps and prints one checksum line per (step, parameter tensor) of the gradient and
of the parameters; the parent compares the lines of all processes and repetitions and reports the first difference."""
import argparse, hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
for p in (os.path.join(PKG, "utils"), PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def child(steps, batch):
    import torch
    import bench, config, synth
    from utils import get_feat_extractor
    dev = torch.device("cuda", 0)
    ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
    m = bench._make_model(0.0, dev, degenerate_ok=True)
    m.train(); m.engine.reset_optimizer()
    feats = torch.empty((batch, 100, 44), device=dev)
    ex.extract_batch(synth.make_clips(batch, seed=1234, device=dev), out=feats)
    labels = synth.make_labels(batch, seed=4321, device=dev)
    names = [n for n, _ in m.named_parameters()]
    for s in range(steps):
        probs = m.engine.forward(feats, train=True, labels=labels)
        plan = m.engine._last_train_plan
        for bi, a in enumerate(plan["acts"]):
            for k in ("c1", "c2", "cs", "y", "coef1", "coef2", "ybits"):
                if k in a and a[k] is not None:
                    print(f"step {s} fwd block{bi} {k} {hashlib.md5(a[k].detach().cpu().numpy().tobytes()).hexdigest()}")
        print(f"step {s} fwd probs {hashlib.md5(probs.detach().cpu().numpy().tobytes()).hexdigest()}")
        m.engine.backward(None)
        for n, (_, gv) in zip(names, m.engine.grad_views().items()):
            print(f"step {s} grad {n} {hashlib.md5(gv.detach().cpu().numpy().tobytes()).hexdigest()}")
        m.engine.clip_and_step()
        m.global_step += 1
        print(f"step {s} params {hashlib.md5(m.engine.flat_param().detach().cpu().numpy().tobytes()).hexdigest()}", flush=True)


def inproc(n, batch, opts=""):
    """one forward, n backward passes over it: which gradient tensors change from pass to pass?"""
    import torch
    import bench, config, synth
    from utils import get_feat_extractor
    dev = torch.device("cuda", 0)
    ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
    m = bench._make_model(0.0, dev, degenerate_ok=True)
    m.train(); m.engine.reset_optimizer()
    for o in [o for o in opts.split(",") if o]:
        k, v = o.split("=")
        setattr(m.engine, k, v == "1")
    feats = torch.empty((batch, 100, 44), device=dev)
    ex.extract_batch(synth.make_clips(batch, seed=1234, device=dev), out=feats)
    labels = synth.make_labels(batch, seed=4321, device=dev)
    m.engine.forward(feats, train=True, labels=labels)
    ref, bad = None, {}
    plan = m.engine._last_train_plan
    ref_extra = None
    for i in range(n):
        m.engine.backward(None)
        cur = {k: v.clone() for k, v in m.engine.grad_views().items()}
        extra = {"ws": plan["wgrad_ws"][:1536 * 576].clone(), "bcoef": plan["bcoef"].clone(), "dy": plan["g"][(100, 44)][0].clone(),
                 "partials": plan["partials"][:1536 * 128].clone()}
        if ref is None:
            ref, ref_extra = cur, extra
            continue
        if not torch.equal(cur["conv1.weight"], ref["conv1.weight"]):
            for k, v in extra.items():
                d = (v != ref_extra[k])
                msg = f"   {k}: {int(d.sum())} elements differ"
                if k == "ws" and d.any():
                    rows_bad = torch.nonzero(d.reshape(-1, 576).any(1)).reshape(-1).tolist()
                    msg += f"; slabs {rows_bad[:12]} ({len(rows_bad)}); elements of the first: {torch.nonzero(d.reshape(-1, 576)[rows_bad[0]]).reshape(-1).tolist()[:30]}"
                print(msg)
        for k in cur:
            if not torch.equal(cur[k], ref[k]):
                bad[k] = bad.get(k, 0) + 1
                if bad[k] <= 0:
                    d = (cur[k] - ref[k]).abs()
                    print(f"pass {i}: {k} differs in {int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3e} (|ref| max {float(ref[k].abs().max()):.3e})")
                    idx = torch.nonzero(d.reshape(-1) > 0).reshape(-1).tolist()
                    print("   flat indices:", idx[:40], "...", idx[-10:])
                    print("   delta / ref at those:", [(round(float(cur[k].reshape(-1)[j] - ref[k].reshape(-1)[j]) * 1e9, 2)) for j in idx[:16]])
    print(f"[{opts}] {n} backward passes at batch {batch}: tensors that changed: {bad}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--inproc", type=int, default=0)
    ap.add_argument("--opts", default="")
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.inproc:
        inproc(a.inproc, a.batch, a.opts)
        sys.exit(0)
    if a.child:
        child(a.steps, a.batch)
        sys.exit(0)
    ref = None
    bad = 0
    for rep in range(a.repeat):
        ps = [subprocess.Popen([sys.executable, __file__, "--child", "--steps", str(a.steps), "--batch", str(a.batch)],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(a.procs)]
        outs = [p.communicate()[0].strip().splitlines() for p in ps]
        for k, o in enumerate(outs):
            if ref is None:
                ref = o
                continue
            diff = [(la, lb) for la, lb in zip(ref, o) if la != lb]
            if diff:
                bad += 1
                print(f"repetition {rep} process {k}: {len(diff)} of {len(ref)} lines differ; first: {diff[0][0]}  vs  {diff[0][1]}")
                print("   differing (first 12):", [" ".join(d[0].split()[:4]) for d in diff[:12]])
                same_after = [la for la, lb in zip(ref, o) if la == lb and la.split()[1] == diff[0][0].split()[1]]
                print("   identical lines of the same step:", len(same_after))
    print(f"{a.repeat} x {a.procs} runs of {a.steps} steps at batch {a.batch}: {bad} differ from the first")
