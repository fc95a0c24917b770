"""Run the pure-MFMA calibration loop (mfma_peak.hip) and print the achieved fraction of the nominal fp32-matrix peak."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmfma_peak.so"))
lib.mfma_peak_run.restype = ctypes.c_float
lib.mfma_peak_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.zeros(1, device="cuda")
for wgs_per_cu in (1, 2, 3):
    for nacc in (2, 4, 8):
        blocks, iters = 256 * wgs_per_cu, 4000
        ms = lib.mfma_peak_run(nacc, blocks, iters, None)
        flops = blocks * 4 * iters * 8 * nacc * 4096.0
        print(f"{wgs_per_cu} WG/CU x 4 waves, {nacc} independent accumulators: {ms:.3f} ms, {flops / ms / 1e9:.1f} TFLOP/s = {flops / ms / 1e9 / 157.3:.3f} of 157.3")
lib.mfma_lds_run.restype = ctypes.c_float
lib.mfma_lds_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for wgs_per_cu in (1, 2):
    blocks, iters = 256 * wgs_per_cu, 4000
    ms = lib.mfma_lds_run(blocks, iters, None)
    flops = blocks * 4 * iters * 64 * 4096.0
    print(f"LDS-fed loop (conv_s1 operand pattern), {wgs_per_cu} WG/CU: {ms:.3f} ms, {flops / ms / 1e9:.1f} TFLOP/s = {flops / ms / 1e9 / 157.3:.3f} of 157.3")
