"""Diagnostic: phase shares of wgrad_kernel<64,64,9> from in-kernel stamps (-DLAD_STAMP build, never the product)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd")]
import _hip as h
h.LIB_PATH = sys.argv[1]
lib = h.lib()
lib.lad_debug_read_wgrad_stamps.restype = ctypes.c_int
lib.lad_debug_read_wgrad_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
st = h.stream_handle()
B, H, W, cin, cout = 512, 100, 44, 64, 64
rows = int(lib.lad_act_rows(B, H, W))
x = torch.randn(rows * cin, device="cuda"); dout = torch.randn(rows * cout, device="cuda")
ws = torch.zeros(int(lib.lad_conv_wgrad_workspace_floats(cin, cout, 9)), device="cuda")
dw = torch.zeros(cout, cin, 3, 3, device="cuda"); db = torch.zeros(cout, device="cuda")
for _ in range(3):
    h.check(lib.lad_conv_wgrad(h.ptr(x), h.ptr(dout), h.ptr(ws), h.ptr(dw), h.ptr(db), B, H, W, cin, cout, 9, st))
torch.cuda.synchronize()
buf = np.zeros(8 * 512, np.uint64)
assert lib.lad_debug_read_wgrad_stamps(buf.ctypes.data, 8 * 512) == 0
t = buf.reshape(512, 8)[:, :5].astype(np.float64)
tiles = rows / 64 / 512
names = ["barrier A (wait for readers)", "regs->LDS (incl. vmcnt wait)", "barrier B", "fetch issue", "bias sum + MFMA loop"]
tot = t.sum(1).mean()
for k, n in enumerate(names):
    print(f"{n:32s} {t[:, k].mean() / tiles:10.0f} cycles/tile  {100 * t[:, k].mean() / tot:5.1f} %")
print(f"total per tile {tot / tiles:.0f} cycles; MFMA issue per tile per wave = 288*64 = {288 * 64}")
