"""Seeded recipes shared by the golden-vector generator and the parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing under the product
package imports this module.

The golden fixtures under tests/golden/ store *outputs only*; weights and
inputs are regenerated from these recipes so the fixtures stay small
(SURVEY.md section 8(c), rows G1-G6).
"""
import numpy as np

RESNET_BASE = dict(linear_layer_size=48, filter_sizes=[64, 32, 16, 16])


def resnet_state_shapes(filter_sizes=(64, 32, 16, 16), linear_layer_size=48):
    """Ordered (key, shape) list of ResNetBigger's float state_dict entries.

    Mirrors the reference module tree (models.py:82-115, 181-220): conv1/bn1,
    four stages of two ResidualBlocks (conv bias=True, shortcut conv1x1 bias=False
    + BN when stride!=1 or channels change), bn2, bn3, linear1, linear2.
    `num_batches_tracked` entries are integer counters and are not listed here.
    """
    out = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
                    (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,))])

    out.append(("conv1.weight", (64, 1, 3, 3)))
    bn("bn1", 64)
    cin = 64
    for bi, cout in enumerate(filter_sizes, start=1):
        stride = 1 if bi == 1 else 2
        for j in range(2):
            p = f"block{bi}.{j}"
            s = stride if j == 0 else 1
            ci = cin if j == 0 else cout
            out.append((p + ".conv1.weight", (cout, ci, 3, 3)))
            out.append((p + ".conv1.bias", (cout,)))
            bn(p + ".bn1", cout)
            out.append((p + ".conv2.weight", (cout, cout, 3, 3)))
            out.append((p + ".conv2.bias", (cout,)))
            bn(p + ".bn2", cout)
            if s != 1 or ci != cout:
                out.append((p + ".shortcut.0.weight", (cout, ci, 1, 1)))
                bn(p + ".shortcut.1", cout)
        cin = cout
    bn("bn2", linear_layer_size)
    bn("bn3", 32)
    out.append(("linear1.weight", (32, linear_layer_size)))
    out.append(("linear1.bias", (32,)))
    out.append(("linear2.weight", (1, 32)))
    out.append(("linear2.bias", (1,)))
    return out


def make_state(seed, filter_sizes=(64, 32, 16, 16), linear_layer_size=48):
    """Non-degenerate weights + BN running stats from default_rng(seed).

    init_weights (utils/torch_utils.py:22-24) draws *every* parameter from
    N(0, 0.01) including BN gamma/beta, which makes an eval-mode model output
    a constant; goldens therefore use conv weights
    ~ N(0, 0.9^2/fan_in) (stem: 0.1^2/fan_in, linear: 2^2/fan_in) so eval-mode activations
    stay O(1..10) and the sigmoid is not saturated, gamma ~ U(0.5,1.5),
    beta ~ N(0,0.1), running_mean ~ N(0,0.2), running_var ~ U(0.5,1.5).
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in resnet_state_shapes(filter_sizes, linear_layer_size):
        leaf = key.rsplit(".", 1)[1]
        is_bn = (".bn" in "." + key) or key.startswith("bn") or ".shortcut.1" in key
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.2, shape)
        elif leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif is_bn and leaf == "weight":
            v = rng.uniform(0.5, 1.5, shape)
        elif is_bn and leaf == "bias":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            gain = 0.1 if key == "conv1.weight" else (2.0 if key.startswith("linear") else 0.9)
            v = rng.normal(0.0, gain * np.sqrt(1.0 / fan_in), shape)
        else:  # conv / linear bias
            v = rng.normal(0.0, 0.05, shape)
        sd[key] = v.astype(np.float32)
    return sd


def make_features(seed, batch, n_frames=100, n_filters=44):
    """Log-mel-like inputs (B,1,T,F): per-sample level/spread + a smooth time-frequency pattern,
    clipped to [-16, 6] (so samples differ structurally and eval outputs spread out)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames)[:, None] / n_frames
    f = np.arange(n_filters)[None, :] / n_filters
    out = np.empty((batch, 1, n_frames, n_filters), np.float32)
    for b in range(batch):
        mu = rng.uniform(-10, -2)
        sg = rng.uniform(1, 5)
        pat = (rng.uniform(1, 6) * np.sin(2 * np.pi * (rng.uniform(0.5, 4) * t + rng.uniform(0, 1)))
               * np.cos(2 * np.pi * (rng.uniform(0.5, 3) * f + rng.uniform(0, 1))))
        out[b, 0] = np.clip(mu + sg * rng.standard_normal((n_frames, n_filters)) + pat, -16.0, 6.0)
    return out


def make_labels(seed, batch):
    rng = np.random.default_rng(seed)
    return (rng.random(batch) < 0.5).astype(np.int32)


def make_clips(seed, n_clips, n_samples=16000, sr=16000):
    """Synthetic 16 kHz mono clips, recipe of SURVEY.md section 8(d).

    0.05*N(0,1) noise + 3 sinusoids (log-uniform f in [60,7000] Hz, amplitude
    U(0.02,0.3), random phase) times a slow (~4 Hz) random envelope, clipped
    to [-1,1], float32.
    """
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((n_clips, n_samples), np.float32)
    for i in range(n_clips):
        x = 0.05 * rng.standard_normal(n_samples)
        f = np.exp(rng.uniform(np.log(60.0), np.log(7000.0), 3))
        a = rng.uniform(0.02, 0.3, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        tone = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(0)
        env = 0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2.0, 6.0) * t + rng.uniform(0, 2 * np.pi))
        x = x + tone * env
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out


def edge_case_clips(n_samples=16000, sr=16000):
    """Parity-only clips: zeros, DC, square, impulses, loud noise."""
    z = np.zeros(n_samples, np.float32)
    dc = np.full(n_samples, 0.5, np.float32)
    t = np.arange(n_samples)
    sq = np.where((t // 8) % 2 == 0, 1.0, -1.0).astype(np.float32)  # 1 kHz full-scale square
    i0 = z.copy(); i0[0] = 1.0
    i1 = z.copy(); i1[-1] = 1.0
    rng = np.random.default_rng(77)
    wn = np.clip(rng.standard_normal(n_samples), -1, 1).astype(np.float32)
    return np.stack([z, dc, sq, i0, i1, wn])


def make_prob_track(seed, n):
    """Smooth-ish probability track with plateaus, exact 0s, >1 values and a run touching the end."""
    rng = np.random.default_rng(seed)
    base = rng.random(n)
    k = np.ones(9) / 9.0
    p = np.convolve(base, k, mode="same")
    p = (p - p.min()) / (p.max() - p.min())
    p[: n // 50] = 0.0              # exact zeros at the start
    p[n // 3: n // 3 + 7] = 1.5     # overflow values (fix_over_underflow -> 1)
    p[n // 2] = -0.25               # underflow value
    p[-5:] = 0.99                   # run touching the last frame
    return p.astype(np.float64)
