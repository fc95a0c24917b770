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

# No packed-f32 vector instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the library: the target feature is
# switched off for every source (the host pass of hipcc says it does not know the feature: that line is filtered below).
# Round 4 (profiles/r04_slp_nondeterminism.md): kernels whose waves the GPU switches out and back in -- a second process with
# CU-filling kernels on the same device -- lose one row of 16 lanes of a packed-f32 result now and then (the fast featuriser: 7-8 %
# of 256-clip launches; the stem weight gradient, where round 3 met it and fenced stem.hip with -fno-slp-vectorize: 1-10 % of
# backward passes); the same kernels without those instructions are clean (0 of 8000 launches next to the same peer), and they are
# not slower: a plain v_fma_f32 issues in 2 cycles, a packed one in 4 (tools/experiments/issue_overlap.hip) -- the step measured
# 43.5-43.7 k segments/s without them against 42.8-43.0 k with them on the same box.  csrc/fbank16.hip's hand-written packed forms
# went the same way (plain C now).
FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
NOISE = "is not a recognized feature for this target (ignoring feature)"

# Per-file extra flags (none at present).
EXTRA = {}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src, obj_dir=None):
    obj = os.path.join(obj_dir or OBJ, src[:-4] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "lad_hip.h"))
    if _newer(obj, deps):
        return obj, ""
    cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force=False, verbose=True, out_dir=None, workers=6):
    """out_dir: objects and the library go there instead of in-tree (a from-scratch build next to the shipped one:
    tests/test_zz_build_on_box_gpu.py compiles the sources on the GPU box that way)."""
    obj_dir = os.path.join(out_dir, "obj") if out_dir else OBJ
    lib_path = os.path.join(out_dir, "liblad_hip.so") if out_dir else LIB
    os.makedirs(obj_dir, exist_ok=True)
    if force:
        for f in os.listdir(obj_dir):
            os.remove(os.path.join(obj_dir, f))
    srcs = sources()
    try:
        with cf.ThreadPoolExecutor(max_workers=min(workers, len(srcs))) as ex:
            res = list(ex.map(lambda s: _compile(s, obj_dir), srcs))
    except Exception:
        # never leave a library behind that no longer matches the header / the ctypes table
        if os.path.exists(lib_path):
            os.remove(lib_path)
        raise
    objs = [o for o, _ in res]
    for _, warn in res:
        warn = "\n".join(l for l in warn.splitlines() if NOISE not in l)
        if warn.strip() and verbose:
            sys.stderr.write(warn + "\n")
    if not _newer(lib_path, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {lib_path} ({os.path.getsize(lib_path)} bytes) from {len(srcs)} sources")
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv)
