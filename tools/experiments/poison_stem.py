"""Does stem_wgrad_kernel<2> (SLP build: LAD_HIP_LIB=tools/libexp_stem_slp.so) read registers it never wrote?  ONE process: the
vector register files are filled with a fresh pattern before every backward pass; a kernel that only reads what it wrote gives the
same gradient every time.    LAD_HIP_LIB=tools/libexp_stem_slp.so python tools/experiments/poison_stem.py [passes]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "laughter-detection-icsi_amd")
sys.path[:0] = [os.path.join(PKG, "utils"), PKG, ROOT]
import torch
import bench, config, synth, _hip
from utils import get_feat_extractor
poison = ctypes.CDLL(os.path.join(ROOT, "tools", "experiments", "libvgpr_poison.so"))
poison.lad_poison_vgprs.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
ex = get_feat_extractor(config.FEAT["num_samples"], config.FEAT["num_filters"])
m = bench._make_model(0.0, dev, degenerate_ok=True)
m.train(); m.engine.reset_optimizer()
B = 64
feats = torch.empty((B, 100, 44), device=dev)
ex.extract_batch(synth.make_clips(B, seed=1234, device=dev), out=feats)
labels = synth.make_labels(B, seed=4321, device=dev)
m.engine.forward(feats, train=True, labels=labels)
sink = torch.zeros(4, device=dev, dtype=torch.int32)
lib = _hip.lib()
# poison right before the stem weight gradient: wrap the library entry
orig = lib.lad_stem_wgrad_bn
count = [0]
def wrapped(*a):
    count[0] += 1
    assert poison.lad_poison_vgprs(0x7fc00000 + 7919 * count[0], sink.data_ptr(), a[-1]) == 0
    return orig(*a)
for mode in ("plain", "poisoned"):
    if mode == "poisoned":
        m.engine._lib = type("L", (), {"__getattr__": lambda s, k: wrapped if k == "lad_stem_wgrad_bn" else getattr(lib, k)})()
    ref, bad = None, 0
    for i in range(n):
        m.engine.backward(None)
        g = m.engine.grad_views()["conv1.weight"].clone()
        if ref is None:
            ref = g
        elif not torch.equal(g, ref):
            bad += 1
    print(f"{mode}: {bad} of {n - 1} passes give another conv1.weight gradient than the first ({os.environ.get('LAD_HIP_LIB', 'product library')})")
