"""The shipped liblad_hip.so against a library compiled from the sources ON THE GPU BOX (runs last: the file name sorts behind the
parity tests, so an environment problem here cannot stop them under `pytest -x`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_compiled_from_source_on_this_box_gives_the_shipped_librarys_results(tmp_path):
    """The in-tree liblad_hip.so travels to the GPU box prebuilt (build.py is incremental), so nothing there proves that the
    SOURCES give it.  Compile all of csrc/*.hip from scratch on this box into a scratch directory, run
    smoke() -- fbank + one ResNetBigger training step against the oracle -- once on each library in a child process (LAD_HIP_LIB
    selects the library; the child says which file it mapped) and ask for the same figures to the last printed digit."""
    import importlib.util
    pkg = os.path.join(ROOT, "laughter-detection-icsi_amd")
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc on this box: nothing to compile with")
    spec = importlib.util.spec_from_file_location("lad_build_fresh", os.path.join(pkg, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.build(force=True, verbose=False, out_dir=str(tmp_path), workers=min(16, os.cpu_count() or 4))
    assert os.path.dirname(fresh) == str(tmp_path) and os.path.getsize(fresh) > 1 << 20
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as g\n"
            "g.smoke()\n"
            "import _hip\n"
            "maps = open('/proc/self/maps').read()\n"
            "print('MAPPED', [l.split()[-1] for l in maps.splitlines() if 'liblad_hip' in l][0])\n" % ROOT)
    out = {}
    for name, path in (("shipped", None), ("fresh", fresh)):
        env = dict(os.environ)
        env.pop("LAD_HIP_LIB", None)
        if path:
            env["LAD_HIP_LIB"] = path
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        mapped = [l for l in lines if l.startswith("MAPPED")][0].split(None, 1)[1]
        assert os.path.realpath(mapped) == os.path.realpath(path or os.path.join(pkg, "liblad_hip.so")), (name, mapped)
        out[name] = [l for l in lines if l.startswith("smoke:")]
        assert len(out[name]) == 2, r.stdout
    assert out["fresh"] == out["shipped"], out
