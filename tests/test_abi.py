"""The C-ABI shared library: loads, exports every symbol include/pt_api.h declares, and its
host-only entry points behave.  No GPU needed (no kernels are launched)."""
import ctypes as C
import os
import re
import subprocess

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


def test_header_is_plain_c99():
    """The header is meant for bindgen / cgo / ctypes: it must compile as strict C99 on its own."""
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(ROOT, "include", "pt_api.h")], check=True)


def test_struct_layouts_match_header(pkg, tmp_path):
    """sizeof / offsetof of every structure field as a C compiler sees the header == the ctypes mirror."""
    structs = {"pt_params": pkg.PtParams, "pt_counters": pkg.PtCounters, "pt_timing": pkg.PtTiming}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "pt_api.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src)], check=True)
    seen = dict(l.split() for l in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(seen[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(seen["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
    # the fields the reference fixes as WGSL constants (shader.wgsl:24-37, 48-54, 142, 283-286) are all there
    names = [f for f, _ in pkg.PtParams._fields_]
    for want in ("width", "height", "rays_per_pixel", "max_bounce_count", "bvh_max_depth", "fov_degrees", "far_distance",
                 "tri_color", "tri_emission_color", "tri_emission_strength", "tri_smoothness", "sky_horizon", "sky_zenith",
                 "ground", "sun_dir", "sun_intensity", "sun_focus", "use_environment_lighting"):
        assert want in names


def test_default_params_are_the_reference_constants(pkg):
    p = pkg.default_params()
    assert (p.width, p.height, p.rays_per_pixel, p.max_bounce_count) == (1200, 600, 20, 10)   # shader.wgsl:33-35
    assert p.fov_degrees == 60.0
    assert [round(x, 6) for x in p.tri_color] == [0.65, 0.65, 1.0] and round(p.tri_smoothness, 6) == 0.9
    assert list(p.sun_dir) == [0.0, np.float32(-0.4), 0.5] and p.sun_intensity == 3.0 and p.sun_focus == 200.0
    assert p.use_environment_lighting == 1 and p.flags == 0
    assert p.far_distance == 1000000.0 and p.bvh_max_depth == 0          # shader.wgsl:142; 0 = depth of the node file


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


def _rust_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return "\n".join(re.findall(r"```rust\n(.*?)```", text, flags=re.S))


def test_integration_document_params_struct_matches_the_abi(pkg):
    """The Rust FFI stub a maintainer would paste from the integration guide: its `pt_params` must have the header's fields,
    in the header's order and with the header's types, or the library reads a struct of another size."""
    rust = _rust_blocks()
    m = re.search(r"#\[repr\(C\)\]\s*pub struct pt_params\s*\{(.*?)\}", rust, flags=re.S)
    assert m, "INTEGRATION.md has no #[repr(C)] struct pt_params"
    fields = re.findall(r"pub\s+([a-z_0-9]+)\s*:\s*(\[f32;\s*3\]|u32|f32)", m.group(1))
    ctype = {"u32": C.c_uint32, "f32": C.c_float}
    want = []
    for name, t in pkg.PtParams._fields_:
        want.append((name, "[f32;3]" if t is C.c_float * 3 else {C.c_uint32: "u32", C.c_float: "f32"}[t]))
    got = [(n, t.replace(" ", "")) for n, t in fields]
    assert got == want
    size = sum(12 if t == "[f32;3]" else C.sizeof(ctype[t]) for _, t in got)
    assert size == C.sizeof(pkg.PtParams)


def test_integration_document_functions_match_the_header():
    """Every function the Rust stub declares exists in include/pt_api.h with the same number of arguments."""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pt_api.h")).read(), flags=re.S)
    decls = {}
    for name, args in re.findall(r"\b(pt_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = args.strip()
        decls[name] = 0 if args in ("", "void") else args.count(",") + 1
    stub = re.findall(r"pub fn (pt_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", _rust_blocks(), flags=re.S)
    assert len(stub) >= 15
    for name, args in stub:
        assert name in decls, name
        n = len(re.findall(r"(?<!:):(?!:)", args))          # one `name: type` per argument (`::` is a path)
        assert n == decls[name], (name, n, decls[name])
