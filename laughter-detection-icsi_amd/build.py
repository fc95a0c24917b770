#!/usr/bin/env python3
"""Build liblad_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python laughter-detection-icsi_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "liblad_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(HERE, "..", "include")]


# Per-file extra flags.
# stem.hip: no SLP vectorisation.  Round 3 found the stem weight gradient (stem_wgrad_kernel<2>) NOT reproducible from run to
# run when a second process shares the GPU (the two-rank rehearsal of tests/test_bench_gpu.py; tools/diag_determinism.py
# --inproc: 20-70 of 1500 identical backward passes gave another conv1.weight gradient, always ONE accumulator register
# acc[odd channel][odd tap] of whole workgroups, inputs bit-identical).  hipcc had packed the scalar fmaf chains of that kernel
# into v_pk_fma_f32 with op_sel forms over register pairs of which only one half is live; compiled without SLP (plain
# v_fma_f32) 0 of 3000 passes differ.  The kernel is bound by its loads, not by VALU issue: same time either way.
EXTRA = {"stem.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "lad_hip.h"))
    if _newer(obj, deps):
        return obj, ""
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = sources()
    try:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
            res = list(ex.map(_compile, srcs))
    except Exception:
        # never leave a library behind that no longer matches the header / the ctypes table
        if os.path.exists(LIB):
            os.remove(LIB)
        raise
    objs = [o for o, _ in res]
    for _, warn in res:
        if warn.strip() and verbose:
            sys.stderr.write(warn)
    if not _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB)} bytes) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
