import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "laughter-detection-icsi_amd"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import feats
from oracle import fbank_oracle as fo, recipe
cfg = feats.HipFbankConfig(num_filters=44, frame_shift=0.01)
fast = feats.HipFbank(cfg); gen = feats.HipFbank(cfg).use_general_kernel()
for seed, N in ((32, 4_000_000), (5, 1_600_000)):
    clips = recipe.make_clips(seed, 1, n_samples=N)
    x = torch.from_numpy(clips).cuda()
    a = fast.extract_batch(x).cpu().numpy()[0]; b = gen.extract_batch(x).cpu().numpy()[0]
    ref = fo.fbank(clips[0], num_filters=44, dtype=np.float64)
    ea, eb = np.abs(a - ref), np.abs(b - ref)
    print(N, "fast vs oracle max %.2e rms %.2e | general max %.2e rms %.2e | fast-general max %.2e" % (ea.max(), np.sqrt((ea**2).mean()), eb.max(), np.sqrt((eb**2).mean()), np.abs(a-b).max()))
    for e, nm in ((ea, "fast"), (eb, "gen")):
        idx = np.unravel_index(e.argmax(), e.shape)
        print("  worst", nm, idx, "ref", ref[idx], "row range", ref[idx[0]].min(), ref[idx[0]].max())
    print("  count > 1e-4: fast", (ea > 1e-4).sum(), "gen", (eb > 1e-4).sum(), "of", ea.size)
