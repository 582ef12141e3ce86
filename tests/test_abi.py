"""The C-ABI shared library: loads, exports every symbol include/pt_api.h declares, and its
host-only entry points behave.  No GPU needed (no kernels are launched)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(pkg.API_SYMBOLS) == syms


def test_struct_layouts_match_header(pkg):
    # pt_params: 5 x 4 bytes + 6 + 2 + 9 + 3 + 2 floats + 2 u32 -> 4-byte fields only
    assert C.sizeof(pkg.PtParams) == 4 * (4 + 1 + 3 + 3 + 1 + 1 + 3 + 3 + 3 + 3 + 1 + 1 + 1 + 1)
    assert C.sizeof(pkg.PtCounters) == 7 * 8
    assert C.sizeof(pkg.PtTiming) == 2 * 8 + 3 * 8


def test_default_params_are_the_reference_constants(pkg):
    p = pkg.default_params()
    assert (p.width, p.height, p.rays_per_pixel, p.max_bounce_count) == (1200, 600, 20, 10)   # shader.wgsl:33-35
    assert p.fov_degrees == 60.0
    assert [round(x, 6) for x in p.tri_color] == [0.65, 0.65, 1.0] and round(p.tri_smoothness, 6) == 0.9
    assert list(p.sun_dir) == [0.0, np.float32(-0.4), 0.5] and p.sun_intensity == 3.0 and p.sun_focus == 200.0
    assert p.use_environment_lighting == 1 and p.flags == 0


def test_null_arguments_are_rejected_not_fatal(pkg):
    lib = pkg.load_library()
    assert lib.pt_default_params(None) < 0
    assert lib.pt_create(0, None, None) < 0
    ctx = C.c_void_p()
    assert lib.pt_create(0, None, C.byref(ctx)) < 0 and not ctx
    assert b"NULL" in lib.pt_last_error(None)
    assert lib.pt_render_frames(None, 0, 1) < 0
    assert lib.pt_read_accum(None, None, 4) < 0
    lib.pt_destroy(None)
    lib.pt_free(None)


def test_create_without_gpu_reports_an_error(pkg):
    """On a host without a gfx950 GPU pt_create must fail with a message (never fall back to a CPU path)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtError) as e:
        pkg.Renderer(pkg.default_params(width=16, height=16), device=0)
    assert e.value.code < 0 and str(e.value)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg, "_lib", None)
    monkeypatch.setattr(pkg, "LIB_PATH", "/nonexistent/libpt_hip.so")
    with pytest.raises(FileNotFoundError):
        pkg.load_library()
