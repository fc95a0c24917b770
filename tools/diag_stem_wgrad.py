"""Diagnostic: is lad_stem_wgrad_bn / lad_stem_wgrad bit-reproducible call to call (also next to another process)?"""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd", "utils"), os.path.join(ROOT, "laughter-detection-icsi_amd"), ROOT]
import _hip as h
lib = h.lib(); st = h.stream_handle()
B, H, W, C = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 100, 44, 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
g = torch.Generator(device="cuda").manual_seed(5)
rows = int(lib.lad_act_rows(B, H, W))
feat = torch.randn(B * H * W, device="cuda", generator=g)
dy = torch.randn(rows * C, device="cuda", generator=g)
w = torch.randn(C * 9, device="cuda", generator=g) * 0.3
coef = torch.zeros(6 * C, device="cuda"); coef[:C] = 1.0; coef[3 * C:4 * C] = 1.0
bcoef = torch.randn(8 * C, device="cuda", generator=g) * 0.1
ws = torch.zeros(int(lib.lad_stem_wgrad_workspace_floats()), device="cuda")
dw = torch.zeros(C * 9, device="cuda")
def run(mode):
    if mode == 2:
        h.check(lib.lad_stem_wgrad_bn(h.ptr(feat), h.ptr(dy), None, h.ptr(w), h.ptr(coef), h.ptr(bcoef), h.ptr(ws), h.ptr(dw), B, H, W, C, st))
    else:
        h.check(lib.lad_stem_wgrad(h.ptr(feat), h.ptr(dy), h.ptr(ws), h.ptr(dw), B, H, W, C, st))
for mode in (2, 0):
    run(mode); ref = dw.clone(); ref_ws = ws.clone(); bad = 0
    for i in range(N):
        run(mode)
        if not torch.equal(dw, ref):
            bad += 1
            if bad <= 2:
                dws = (ws != ref_ws).reshape(-1, C * 9)[:1536]
                rowsbad = torch.nonzero(dws.any(1)).reshape(-1).tolist()
                print(f"mode {mode} call {i}: dw differs in {int((dw != ref).sum())} elements; slabs that differ: {rowsbad[:10]} ({len(rowsbad)}); "
                      f"elements in the first: {torch.nonzero(dws[rowsbad[0]]).reshape(-1).tolist()[:24] if rowsbad else None}")
    print(f"mode {mode}: {bad} of {N} calls differ")
