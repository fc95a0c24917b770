"""The C-ABI library builds, loads on a CPU-only host and exports exactly what include/lad_hip.h declares.
No compute call is made here (no GPU); the parity tests proper are the `-m gpu` ones."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lad_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lad_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("lad_build", os.path.join(ROOT, "laughter-detection-icsi_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return _hip


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "lad_hip.h"\nint main(void){ lad_fbank_cfg c; (void)c; return LAD_VERSION > 0 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "t.o")], check=True)


def test_binding_table_covers_header(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    assert sorted(built_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert built_lib.lib().lad_version() == 100
    assert built_lib.lib().lad_last_error() is not None


def test_size_queries_need_no_gpu(built_lib):
    lib = built_lib.lib()
    assert lib.lad_conv_packed_weight_floats(64, 64, 9, 0) == 9 * 64 * 64
    assert lib.lad_conv_packed_weight_floats(32, 64, 1, 1) == 32 * 64      # dgrad image: K = cout = 32, N = cin = 64
    assert lib.lad_conv_packed_weight_floats(16, 16, 9, 0) == 9 * 16 * 32  # N padded to a 32-wide MFMA tile
    assert lib.lad_act_rows(512, 100, 44) == 512 * 101 * 45 + 46  # shared-border layout: body + tail
    assert lib.lad_conv_num_tiles(512, 100, 44) == (512 * 101 * 45 + 46 + 127) // 128
    assert lib.lad_conv_wgrad_workspace_floats(64, 64, 9) > 0
    assert lib.lad_grad_sumsq_partials() > 0
    assert lib.lad_head_workspace_floats(512, 48) == 512 * (96 + 48)
    assert lib.lad_bn_bwd_workspace_floats(64) > 0


def test_product_path_has_no_cpu_fallback(built_lib):
    import contextlib
    import io

    import torch

    import models
    with contextlib.redirect_stdout(io.StringIO()):
        m = models.ResNetBigger(dropout_rate=0.0, linear_layer_size=48, filter_sizes=[64, 32, 16, 16])
    with pytest.raises(built_lib.LadHipError):
        m(torch.zeros(2, 1, 100, 44))
    with pytest.raises(built_lib.LadHipError):
        built_lib.require_cuda(torch.zeros(3), "x")
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "laughter-detection-icsi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("oracle/", ""), f


def test_library_has_no_packed_f32_instruction(built_lib, tmp_path):
    """Round 4: packed-f32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) are what a wave loses a row of 16 lanes of
    when the GPU switches it out and back in next to a second process (profiles/r04_slp_nondeterminism.md); build.py switches the
    target feature off for every source.  Disassemble the gfx950 code object of every object file the library was linked from."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    objdir = os.path.join(ROOT, "laughter-detection-icsi_amd", "csrc", "build")
    objs = sorted(f for f in os.listdir(objdir) if f.endswith(".o")) if os.path.isdir(objdir) else []
    if not (os.path.exists(objdump) and os.path.exists(bundler) and objs):
        pytest.skip("no llvm-objdump / clang-offload-bundler / object files in this tree")
    srcs = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "laughter-detection-icsi_amd", "csrc")) if f.endswith(".hip"))
    assert [o[:-2] for o in objs] == srcs                      # one object per source, nothing stale
    n_mfma = 0
    HOST_ONLY = ("common.o",)
    for o in objs:
        fb, co = tmp_path / (o + ".fatbin"), tmp_path / (o + ".co")
        r = subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fb}", os.path.join(objdir, o)], capture_output=True, text=True)
        if r.returncode != 0:                                  # host-only source (common.hip): no device section
            assert "not found" in r.stderr and o in HOST_ONLY, r.stderr
            continue
        subprocess.run([bundler, "--type=o", "--unbundle", f"--input={fb}", f"--output={co}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True, check=True).stdout
        n_mfma += dis.count("v_mfma_")
        packed = re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", dis)
        assert not packed, f"{len(packed)} packed-f32 instructions in {o}"
    assert n_mfma > 1000                                        # (it was the device code we looked at)
