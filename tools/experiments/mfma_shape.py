"""Run tools/experiments/mfma_shape.hip: the conv_b3 MFMA loop alone in both bf16 MFMA shapes, random and zero operands."""
import ctypes, os
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmfma_shape.so"))
lib.mfma_shape_run.restype = ctypes.c_float
lib.mfma_shape_run.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
for zero in (0, 1):
    for wgs in (1, 2, 3):
        for rnd in range(2):
            for shape in (0, 1):
                blocks, iters = 256 * wgs, 1500 // wgs
                st = np.zeros((blocks, 2), dtype=np.uint64)
                ms = lib.mfma_shape_run(shape, blocks, iters, 6, zero, st.ctypes.data)
                # per wave and step (tap): 64 x 64 x 16 x 6 products x 2 FLOP executed
                flop = blocks * 4.0 * iters * 9 * 64 * 64 * 16 * 6 * 2
                clk = np.median(st[:, 0].astype(np.float64) / st[:, 1].astype(np.float64)) * 0.1
                cyc_per_step = np.median(st[:, 0].astype(np.float64)) / (iters * 9)
                print(f"{'zeros ' if zero else 'random'} {wgs} WG/CU shape {'16x16x32' if shape else '32x32x16'}: {ms:8.3f} ms  "
                      f"{flop / ms / 1e9:7.1f} TFLOP/s executed = {flop / ms / 1e9 / 2500:.3f} of 2.5 PF;  in-kernel clock {clk:.2f} GHz, "
                      f"{cyc_per_step:.0f} cycles per (tap, 16 ch) step per workgroup (ideal {768 * wgs})", flush=True)
