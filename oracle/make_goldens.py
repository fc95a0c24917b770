#!/usr/bin/env python3
"""Generate tests/golden/* by importing the reference itself (runs in the build container only).

    python oracle/make_goldens.py [--ref /root/reference]

The reference cannot travel to the GPU box, so the outputs are committed as
small fixtures; inputs and weights are re-derived from oracle/recipe.py seeds.
Importable reference modules: models.py, utils/torch_utils.py,
laugh_segmenter.py (needs an empty stand-in module named `librosa` for its
unused top-level import).  Feature extraction (lhotse) is NOT importable here,
so no feature golden exists: see oracle/fbank_oracle.py ("parity unpinned").
"""
import argparse
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import recipe  # noqa: E402

FULL_GRAD_KEYS = ["conv1.weight", "block1.0.conv1.weight", "block1.1.conv2.bias", "block2.0.shortcut.0.weight",
                  "block2.0.conv1.weight", "block3.0.conv1.weight", "block4.1.conv2.weight",
                  "bn1.weight", "bn1.bias", "block1.0.bn2.weight", "block2.0.shortcut.1.weight",
                  "bn2.weight", "bn3.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"]


def load_reference(ref):
    sys.path.insert(0, ref)
    sys.path.insert(0, os.path.join(ref, "utils"))
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    with contextlib.redirect_stdout(io.StringIO()):
        import models  # noqa
        import torch_utils  # noqa
        import laugh_segmenter  # noqa
    return models, torch_utils, laugh_segmenter


def build_model(models, seed, dropout=0.0):
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.ResNetBigger(dropout_rate=dropout, **recipe.RESNET_BASE)
    sd = recipe.make_state(seed)
    full = m.state_dict()
    for k, v in sd.items():
        assert tuple(full[k].shape) == v.shape, k
        full[k] = torch.from_numpy(v.copy())
    m.load_state_dict(full)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    models, torch_utils, seg = load_reference(args.ref)

    # ---- state_dict layout (keys + shapes) --------------------------------------------------
    m = build_model(models, 101)
    layout = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    with open(os.path.join(args.out, "state_dict_layout.json"), "w") as f:
        json.dump({"n_params": sum(p.numel() for p in m.parameters()), "entries": layout,
                   "param_order": [k for k, _ in m.named_parameters()]}, f, indent=0)

    # ---- G1: eval-mode forward -----------------------------------------------------------------
    m.eval()
    x = torch.from_numpy(recipe.make_features(202, 8))
    feats = {}
    hooks = []
    for name in ["bn1", "block1", "block2", "block3", "block4"]:
        mod = getattr(m, name)
        hooks.append(mod.register_forward_hook(lambda _m, _i, o, name=name: feats.__setitem__(name, o.detach())))
    with torch.no_grad():
        probs = m(x)
    for h in hooks:
        h.remove()
    np.savez(os.path.join(args.out, "resnet_eval.npz"),
             state_seed=101, feat_seed=202, batch=8,
             probs=probs.numpy(),
             block4=feats["block4"].numpy(),               # (8,16,13,6)
             block1_sample=feats["block1"][0, :4].numpy(),  # (4,100,44) of sample 0
             block2_sum=feats["block2"].double().sum(dim=(2, 3)).numpy(),
             block3_sum=feats["block3"].double().sum(dim=(2, 3)).numpy())

    # ---- G2 + G3: train-mode forward/backward, clip, Adam --------------------------------------
    B = 8
    m = build_model(models, 101)
    m.train()
    x = torch.from_numpy(recipe.make_features(303, B))
    t = torch.from_numpy(recipe.make_labels(404, B)).float()
    opt = torch.optim.Adam(m.parameters())
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    out = m(x).squeeze()
    loss = torch.nn.BCELoss()(out, t)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    after = {k: v.detach().clone() for k, v in m.named_parameters()}
    sd_after = m.state_dict()
    save = dict(state_seed=101, feat_seed=303, label_seed=404, batch=B,
                probs=out.detach().numpy(), loss=np.float64(loss.item()),
                total_norm=np.float64(float(total_norm)),
                grad_keys=np.array(list(grads.keys())),
                grad_l2=np.array([float(g.double().norm()) for g in grads.values()]),
                grad_sum=np.array([float(g.double().sum()) for g in grads.values()]),
                delta_l2=np.array([float((after[k] - before[k]).double().norm()) for k in grads]),
                delta_sum=np.array([float((after[k] - before[k]).double().sum()) for k in grads]))
    for k in FULL_GRAD_KEYS:
        save["grad::" + k] = grads[k].numpy()
        save["delta::" + k] = (after[k] - before[k]).numpy()
    for k, v in sd_after.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            save["stat::" + k] = v.numpy()
    np.savez(os.path.join(args.out, "resnet_train.npz"), **save)

    # second step from the updated state (exercises Adam moments at step 2)
    x2 = torch.from_numpy(recipe.make_features(304, B))
    t2 = torch.from_numpy(recipe.make_labels(405, B)).float()
    m.zero_grad()
    out2 = m(x2).squeeze()
    loss2 = torch.nn.BCELoss()(out2, t2)
    loss2.backward()
    tn2 = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    after2 = {k: v.detach().clone() for k, v in m.named_parameters()}
    np.savez(os.path.join(args.out, "resnet_train_step2.npz"),
             feat_seed=304, label_seed=405, probs=out2.detach().numpy(), loss=np.float64(loss2.item()),
             total_norm=np.float64(float(tn2)),
             delta_l2=np.array([float((after2[k] - after[k]).double().norm()) for k in grads]),
             **{"delta::" + k: (after2[k] - after[k]).numpy() for k in ["linear2.weight", "bn1.weight", "conv1.weight"]})

    # ---- G6: init_weights statistics + degenerate constant output -----------------------------
    with contextlib.redirect_stdout(io.StringIO()):
        mi = models.ResNetBigger(dropout_rate=0.0, **recipe.RESNET_BASE)
    torch.manual_seed(1234)
    mi.apply(torch_utils.init_weights)
    stds = {k: float(p.detach().std()) for k, p in mi.named_parameters() if p.numel() >= 256}
    mi.eval()
    with torch.no_grad():
        const = mi(torch.from_numpy(recipe.make_features(202, 4))).numpy()
    with open(os.path.join(args.out, "init_weights.json"), "w") as f:
        json.dump({"std_large_tensors": stds, "eval_output": const.reshape(-1).tolist()}, f, indent=0)

    # ---- G5: get_laughter_instances -----------------------------------------------------------
    cases = []
    for seed, n, fps in [(1, 600, 100.0), (2, 777, 99.7), (3, 50, 100.0)]:
        p = recipe.make_prob_track(seed, n)
        thr = [0.0, 0.5, 0.6, 1.0]
        mls = [0.0, 0.1, 0.2]
        with contextlib.redirect_stdout(io.StringIO()):
            d = seg.get_laughter_instances(p, thresholds=thr, min_lengths=mls, fps=fps)
        cases.append({"seed": seed, "n": n, "fps": fps, "thresholds": thr, "min_lengths": mls,
                      "result": [[list(k), [list(map(float, s)) for s in v]] for k, v in d.items()]})
    # all-below and all-above tracks, empty track
    for name, p in [("zeros", np.zeros(40)), ("ones", np.ones(40)), ("empty", np.zeros(0))]:
        with contextlib.redirect_stdout(io.StringIO()):
            d = seg.get_laughter_instances(p, thresholds=[0.0, 0.5], min_lengths=[0.0, 0.2], fps=100.0)
        cases.append({"name": name, "probs": p.tolist(), "fps": 100.0, "thresholds": [0.0, 0.5],
                      "min_lengths": [0.0, 0.2],
                      "result": [[list(k), [list(map(float, s)) for s in v]] for k, v in d.items()]})
    with open(os.path.join(args.out, "segmenter.json"), "w") as f:
        json.dump(cases, f)
    print("goldens written to", args.out)
    for fn in sorted(os.listdir(args.out)):
        print(f"  {fn:32s} {os.path.getsize(os.path.join(args.out, fn)):>9d} B")


if __name__ == "__main__":
    main()
