"""Pin the CPU oracle (oracle/*.py) against golden vectors produced by the reference itself.

Goldens: tests/golden/*.npz|json, written by oracle/make_goldens.py which imports
/root/reference/{models.py,utils/torch_utils.py,laugh_segmenter.py}.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import recipe, resnet_oracle as ro, segmenter_oracle as so


@pytest.fixture(scope="module")
def sd():
    return ro.to_torch_state(recipe.make_state(101))


def test_state_layout_matches_reference(golden_dir):
    lay = json.load(open(os.path.join(golden_dir, "state_dict_layout.json")))
    ref = [(k, tuple(s)) for k, s, dt in lay["entries"] if dt == "float32"]
    ours = recipe.resnet_state_shapes()
    assert ours == ref
    assert lay["n_params"] == 221217
    sd = recipe.make_state(1)
    assert ro.param_keys(sd) == lay["param_order"]
    assert sum(int(np.prod(sd[k].shape)) for k in ro.param_keys(sd)) == 221217


def test_eval_forward_matches_reference(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "resnet_eval.npz"))
    x = torch.from_numpy(recipe.make_features(int(g["feat_seed"]), int(g["batch"])))
    with torch.no_grad():
        probs, inter = ro.forward(sd, x, train=False, return_intermediates=True)
    np.testing.assert_allclose(probs.numpy(), g["probs"], rtol=0, atol=2e-6)
    for ours, ref in [(inter["block4"].numpy(), g["block4"]), (inter["block1"][0, :4].numpy(), g["block1_sample"])]:
        np.testing.assert_allclose(ours, ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    np.testing.assert_allclose(inter["block2"].double().sum(dim=(2, 3)).numpy(), g["block2_sum"], rtol=1e-5, atol=1e-3)


def test_train_step_matches_reference(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "resnet_train.npz"))
    B = int(g["batch"])
    x = torch.from_numpy(recipe.make_features(int(g["feat_seed"]), B))
    t = torch.from_numpy(recipe.make_labels(int(g["label_seed"]), B))
    r = ro.train_step(sd, x, t)
    np.testing.assert_allclose(r["probs"].numpy(), g["probs"], atol=2e-6)
    assert abs(r["loss"] - float(g["loss"])) < 2e-6
    assert abs(r["grad_norm"] - float(g["total_norm"])) < 1e-4 * float(g["total_norm"])
    keys = [str(k) for k in g["grad_keys"]]
    assert keys == ro.param_keys(sd)
    for k, l2 in zip(keys, g["grad_l2"]):
        ours = float(r["grads"][k].double().norm())
        # conv biases feeding a BatchNorm have an analytically zero gradient: pure rounding noise
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            assert ours < 1e-5 and l2 < 1e-5
        else:
            assert abs(ours - l2) <= 2e-3 * l2 + 1e-7, k
    for k in g.files:
        if k.startswith("grad::"):
            name = k[6:]
            ref = g[k]
            if name.endswith("conv1.bias") or name.endswith("conv2.bias"):
                assert np.abs(r["grads"][name].numpy()).max() < 1e-4 and np.abs(ref).max() < 1e-4
                continue
            tol = 1e-3 * np.abs(ref).max() + 1e-7
            np.testing.assert_allclose(r["grads"][name].numpy(), ref, rtol=0, atol=tol, err_msg=name)
        if k.startswith("stat::"):
            np.testing.assert_allclose(r["new_sd"][k[6:]].numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)
        if k.startswith("delta::"):
            name = k[7:]
            if name.endswith("conv1.bias") or name.endswith("conv2.bias"):
                continue  # Adam normalises rounding-noise gradients: not comparable
            ours = (r["new_sd"][name] - sd[name]).numpy()
            # Adam's update is lr*sign-like: compare where the gradient is well above noise
            gref = g["grad::" + name]
            big = np.abs(gref) > 1e-3 * np.abs(gref).max()
            np.testing.assert_allclose(ours[big], g[k][big], rtol=0, atol=2e-5, err_msg=name)


def test_second_step_matches_reference(sd, golden_dir):
    g1 = np.load(os.path.join(golden_dir, "resnet_train.npz"))
    g2 = np.load(os.path.join(golden_dir, "resnet_train_step2.npz"))
    B = int(g1["batch"])
    r1 = ro.train_step(sd, torch.from_numpy(recipe.make_features(303, B)), torch.from_numpy(recipe.make_labels(404, B)))
    r2 = ro.train_step(r1["new_sd"], torch.from_numpy(recipe.make_features(int(g2["feat_seed"]), B)),
                       torch.from_numpy(recipe.make_labels(int(g2["label_seed"]), B)),
                       adam_state=r1["adam_state"], step=r1["step"])
    np.testing.assert_allclose(r2["probs"].numpy(), g2["probs"], atol=5e-5)
    assert abs(r2["loss"] - float(g2["loss"])) < 5e-5
    d = (r2["new_sd"]["linear2.weight"] - r1["new_sd"]["linear2.weight"]).numpy()
    np.testing.assert_allclose(d, g2["delta::linear2.weight"], atol=3e-5)


def test_calc_metrics_hand_cases():
    # train.py:203-224: precision is 1.0 when nothing is predicted positive; recall NaN without positive targets
    acc, prec, rec = ro.calc_metrics([1, 0, 1, 0], [0, 0, 0, 0])
    assert (acc, prec, rec) == (0.5, 1.0, 0.0)
    acc, prec, rec = ro.calc_metrics([0, 0, 0], [0, 1, 0])
    assert acc == pytest.approx(2 / 3) and prec == 0.0 and np.isnan(rec)
    acc, prec, rec = ro.calc_metrics([1, 1, 0, 0], [1, 0, 1, 0])
    assert (acc, prec, rec) == (0.5, 0.5, 0.5)


def test_init_weights_degenerate_output(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "init_weights.json")))
    for k, s in j["std_large_tensors"].items():
        assert 0.008 < s < 0.012, k
    out = np.array(j["eval_output"])
    assert np.ptp(out) < 1e-6 and abs(out[0] - 0.5) < 0.01


def test_segmenter_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "segmenter.json")))
    for c in cases:
        p = recipe.make_prob_track(c["seed"], c["n"]) if "seed" in c else np.array(c["probs"])
        d = so.laughter_instances(p, c["thresholds"], c["min_lengths"], c["fps"])
        ref = {tuple(k): [tuple(s) for s in v] for k, v in c["result"]}
        assert list(d.keys()) == list(ref.keys())
        for k in ref:
            assert d[k] == ref[k], (c.get("seed", c.get("name")), k)


def test_fbank_oracle_float32_matches_float64_on_real_audio(golden_dir):
    """Self-consistency of the (unpinned) feature oracle on the reference's own demo recordings."""
    from oracle import fbank_oracle as fo
    z = np.load(os.path.join(golden_dir, "demo_clips.npz"))
    clip = z["clip0"].astype(np.float32) / 32768.0
    f64 = fo.fbank(clip, num_filters=44, dtype=np.float64)
    f32 = fo.fbank(clip, num_filters=44, dtype=np.float32)
    assert f64.shape == (100, 44) and np.abs(f64 - f32).max() < 1e-4
