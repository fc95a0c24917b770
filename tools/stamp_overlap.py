"""Diagnostic: per-CU overlap of the MFMA phases of co-resident conv_s1<64,64,9> workgroups (-DLAD_STAMP build).
Usage: python tools/stamp_overlap.py /path/to/liblad_stamp.so"""
import collections, ctypes, os, sys
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
n = (rows + 255) // 256
buf = np.zeros(8 * n, np.uint64)
assert lib.lad_debug_read_stamps(buf.ctypes.data, 8 * n) == 0
t = buf.reshape(n, 8)
ids = t[:, 7]
hw = (ids & np.uint64(0xffffffff)).astype(np.int64); xcc = (ids >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7; slot = hw & 0xf
key = xcc * 1000 + se * 100 + sh * 20 + cu
tt = t[:, :4].astype(np.int64)
groups = collections.defaultdict(list)
for i in range(n):
    groups[int(key[i])].append(i)
print("blocks", n, "distinct CUs seen", len(groups), "wave slots seen", sorted(set(slot.tolist())))
busy = collections.Counter()
tot = 0
for k, idx in groups.items():
    ev = []
    for i in idx:
        ev.append((tt[i, 1], +1)); ev.append((tt[i, 2], -1))   # MFMA loop = stamps 1..2
    ev.sort()
    lo, hi = min(tt[i, 0] for i in idx), max(tt[i, 3] for i in idx)
    cur, last = 0, lo
    for time, d in ev:
        busy[cur] += time - last
        last = time; cur += d
    busy[0] += hi - last
    tot += hi - lo
print("share of CU time with k workgroups inside their MFMA loop:", {k: round(v / tot, 3) for k, v in sorted(busy.items())})
d = np.diff(tt, axis=1)
print("median cycles: staging %d  MFMA loop %d  epilogue %d  lifetime %d" % (tuple(np.median(d, axis=0)) + (np.median(tt[:, 3] - tt[:, 0]),)))
first = [min(idx, key=lambda i: tt[i, 0]) for idx in groups.values()]
print("blockIdx of the first workgroup per CU (sample):", sorted(first)[:12], "... spread:", min(first), max(first))
