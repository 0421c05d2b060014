"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sm_c_api.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from surfelmapping_amd import capi
    L = capi.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sm_c_api.h but not exported"
    assert sorted(capi.SYMBOLS) == declared
    assert L.sm_api_version() == 4


def test_config_struct_matches_header_defaults():
    from surfelmapping_amd import capi
    c = capi.make_config(1242, 375, 718.856, 718.856, 607.1928, 185.2157)
    assert (c.width, c.height) == (1242, 375)
    assert c.near_clip == 1.0 and c.far_clip == 30.0 and c.fuse_thresh == 0.0   # src/Config.cpp:33-35
    assert c.max_sqrt_vertices == 5000 and c.time_delta == 200 and c.stereo_border == 80.0
    assert c.conflict_cap == 1 and c.device == 0


def test_ctypes_mirrors_have_the_header_layout(tmp_path):
    """The ctypes structures of surfelmapping_amd/capi.py against the C compiler's view of include/sm_c_api.h:
    same size and the same offset for every field."""
    import subprocess
    from surfelmapping_amd import capi
    mirrors = {"sm_config": capi.SmConfig, "sm_counts": capi.SmCounts, "sm_timings": capi.SmTimings}
    lines = []
    for cname, cls in mirrors.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append('printf("sm_frame_log %zu\\n", sizeof(sm_frame_log));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sm_c_api.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in mirrors.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
    assert int(got["sm_frame_log"]) == capi.FRAME_LOG_DTYPE.itemsize


def test_null_arguments_are_rejected_without_touching_the_gpu():
    from surfelmapping_amd import capi
    L = capi.load()
    assert L.sm_process_frame(None, None, None, None, None) == capi.SM_E_ARG
    assert L.sm_sync(None) == capi.SM_E_ARG
    assert L.sm_get_counts(None, None) == capi.SM_E_ARG
    assert L.sm_create(None) is None


def test_no_cpu_fallback_without_device():
    """On a box without a GPU sm_create must fail loudly (SM_E_NO_DEVICE), never compute on CPU."""
    from surfelmapping_amd import capi
    from backends import gpu_available
    if gpu_available():
        pytest.skip("GPU present")
    cfg = capi.make_config(64, 64, 50.0, 50.0, 31.5, 31.5, max_sqrt_vertices=64)
    with pytest.raises(capi.SurfelMapError) as e:
        capi.SurfelMap(cfg)
    assert "no HIP device" in str(e.value) or "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "surfelmapping_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in text and "libsmo" not in text and "smo_" not in text, os.path.join(dp, f)
