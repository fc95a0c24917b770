"""Diagnostic: phase shares of conv_s1<64,64,9> from in-kernel s_memtime stamps (build with -DLAD_STAMP into a
separate library; never the product library).  Usage: python tools/stamp_conv.py /path/to/liblad_stamp.so"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "laughter-detection-icsi_amd")]
import _hip as h
h.LIB_PATH = sys.argv[1]
lib = h.lib()
lib.lad_debug_read_stamps.restype = ctypes.c_int
lib.lad_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
st = h.stream_handle()
B, H, W, cin, cout = 512, 100, 44, 64, 64
rows = int(lib.lad_act_rows(B, H, W))
x = torch.randn(rows * cin, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
bias = torch.randn(cout, device="cuda"); out = torch.empty(rows * cout, device="cuda")
wt = torch.zeros(int(lib.lad_conv_packed_weight_floats(cout, cin, 9, 0)), device="cuda")
h.check(lib.lad_conv_pack_weights(h.ptr(w), cout, cin, 9, 0, h.ptr(wt), st))
part = torch.zeros(int(lib.lad_conv_num_tiles(B, H, W)) * 2 * cout, device="cuda")
for _ in range(3):
    h.check(lib.lad_conv_fwd(h.ptr(x), h.ptr(wt), h.ptr(bias), None, h.ptr(out), h.ptr(part), B, H, W, cin, cout, 9, st))
torch.cuda.synchronize()
n = int(lib.lad_conv_num_tiles(B, H, W))
buf = np.zeros(8 * n, np.uint64)
assert lib.lad_debug_read_stamps(buf.ctypes.data, 8 * n) == 0
t8 = buf.reshape(n, 8).astype(np.int64)
t = t8[:, :4]
d = np.diff(t, axis=1)
fine = np.c_[t8[:, 4] - t8[:, 0], t8[:, 5] - t8[:, 4], t8[:, 6] - t8[:, 5], t8[:, 1] - t8[:, 6]]
print('stage-in split (median cycles): dma+mask %d  loads issued %d  loads landed + LDS writes %d  acc init %d' % tuple(np.median(fine, axis=0)))
print("blocks", n, "median cycles: stage-in %d  main loop %d  epilogue %d  total %d" % tuple(np.median(np.c_[d, t[:, 3] - t[:, 0]], axis=0)))
print("mean   cycles: stage-in %d  main loop %d  epilogue %d  total %d" % tuple(np.mean(np.c_[d, t[:, 3] - t[:, 0]], axis=0)))
span = t[:, 3].max() - t[:, 0].min()
print("kernel span (cycles):", span, " ideal MFMA cycles per block pair-slot: 576*64 =", 576 * 64)
