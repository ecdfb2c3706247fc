"""The C-ABI library loads on a CPU-only box, exports every symbol include/gsplat_hip.h
declares, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gsplat_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+|gsplat_sort_host)\s*\(", src)))


def test_header_symbols_are_exported():
    import gsplat_hip as gh
    lib = gh.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libgsplat_hip.so does not export %s" % n
    assert sorted(gh.EXPORTS) == names


def test_library_holds_gfx950_code_objects():
    import gsplat_hip as gh
    out = subprocess.run(["strings", "-a", gh.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for k in ("k_project_key", "k_scatter", "k_bin_scatter", "k_blend"):
        assert k in out


def test_no_oracle_in_product():
    # the product path must not link, load or import anything under oracle/
    import gsplat_hip as gh
    ldd = subprocess.run(["ldd", gh.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
    pkg = os.path.join(ROOT, "gsplat.js_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".js", ".cpp", ".hip", ".h", ".cc", ".ts")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, os.path.join(dp, f)


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gsplat_hip as gh
    with pytest.raises(gh.GsplatError) as ei:
        gh.HIPRenderer(64, 64)
    assert "no HIP device" in str(ei.value) or "failed" in str(ei.value)
    lib = gh.load_library()
    ctx = ctypes.c_void_p()
    assert lib.gsr_create(ctypes.byref(ctx), None) < 0
    assert not ctx.value


def test_cpp_caller_builds_and_fails_loudly_without_gpu():
    # the third caller of the C ABI is a plain C++ program (tools/bench_cabi.cpp, built by csrc/Makefile)
    import torch
    exe = os.path.join(ROOT, "gsplat.js_amd", "lib", "bench_cabi")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_parity.py::test_cpp_caller_matches_python_host")
    r = subprocess.run([exe, "--config", "C1", "--frames", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU path" in r.stderr
