"""Synthetic 16 kHz clips and labels for benchmarking (recipe of SURVEY.md section 8(d)).

Generated directly on the GPU with torch's generator: bench data only needs the right shape and
statistics (noise floor + tones under a slow envelope, clipped to [-1,1]), not bit-reproducibility with
the numpy recipe the parity tests use.
"""
import math

import torch


def make_clips(n_clips, n_samples=16000, sr=16000, seed=1234, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / sr
    x = 0.05 * torch.randn((n_clips, n_samples), generator=g, device=device)
    lo, hi = math.log(60.0), math.log(7000.0)
    f = torch.exp(lo + (hi - lo) * torch.rand((n_clips, 3, 1), generator=g, device=device))
    a = 0.02 + 0.28 * torch.rand((n_clips, 3, 1), generator=g, device=device)
    ph = 2 * math.pi * torch.rand((n_clips, 3, 1), generator=g, device=device)
    tone = (a * torch.sin(2 * math.pi * f * t[None, None, :] + ph)).sum(1)
    fe = 2.0 + 4.0 * torch.rand((n_clips, 1), generator=g, device=device)
    pe = 2 * math.pi * torch.rand((n_clips, 1), generator=g, device=device)
    env = 0.6 + 0.4 * torch.sin(2 * math.pi * fe * t[None, :] + pe)
    return torch.clamp(x + tone * env, -1.0, 1.0).contiguous()


def make_labels(n, seed=4321, device="cuda"):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (torch.rand(n, generator=g, device=device) < 0.5).to(torch.int32)
