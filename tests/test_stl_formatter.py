"""STL formatter: oracle vs. outputs of the reference script (golden), product vs. both.

The golden .bin files were written by prepare_data/data_formatter.js itself under node
(tests/golden/make_formatter_golden.sh); assets/objects/small_knight.bin is the reference's
own committed output for small_knight.stl."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import stl_format_oracle as so  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OBJ = os.path.join(ROOT, "assets", "objects")

CASES = [
    ("small_knight.stl", os.path.join(GOLD, "small_knight_from_node.bin")),
    ("small_knight.stl", os.path.join(OBJ, "small_knight.bin")),
    ("small_teapot.stl", os.path.join(GOLD, "small_teapot_from_node.bin")),
]

def test_awkward_text_matches_reference_script_output(tmp_path):
    """awkward.stl (ragged spacing, CRLF, non-numbers, ends of the f32 range) went through the
    reference script under node; NaNs included, the bytes must match."""
    pt = load_package()
    text = open(os.path.join(GOLD, "awkward.stl"), "rb").read()
    want = np.fromfile(os.path.join(GOLD, "awkward_from_node.bin"), dtype="<f4")
    assert want.size == 43 and np.isnan(want).sum() == 5
    assert _same(so.stl_format(text, 1.0), want)
    assert _same(pt.stl_format(text, 1.0), want)
    out = tmp_path / "o.bin"
    pt.stl_format_file(os.path.join(GOLD, "awkward.stl"), str(out), 1.0)
    assert open(out, "rb").read() == want.tobytes()


TRICKY = [
    b"",
    b"solid x\nendsolid x\n",
    b"  vertex 1 2 3",                                   # no trailing newline
    b"  vertex 1 2 3\r\n  vertex 4.5e-1 -6E+2 .5\r\n",    # CRLF
    b"  vertex 1  2 3\n",                                # double space -> extra 0
    b"  vertex 1 2 3 \n",                                # trailing space -> extra 0
    b"  vertex 1 2 3 vertex 7 8 9\n",                    # only the first group counts
    b"  vertex abc 0x10 Infinity\n  vertex -Infinity 1e 5.\n",
    b"  vertex 1e39 -1e39 1e-46\n",                      # f32 overflow / underflow
    b"  vertex 0.1 0.2 0.3\n  vertex 16777217 3.4028235677973366e38 1.401298464324817e-45\n",
    b"solid vertex \n",                                  # "vertex " with nothing after it -> one 0
    b"  vertex\t1 2 3 vertex 4 5 6\n",                   # first "vertex " is the second word
]


def _same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("stl,gold", CASES)
def test_oracle_reproduces_reference_script_output(stl, gold):
    text = open(os.path.join(OBJ, stl), "rb").read()
    want = np.fromfile(gold, dtype="<f4")
    assert _same(so.stl_format(text, 1.0), want)


@pytest.mark.parametrize("stl,gold", CASES)
def test_product_reproduces_reference_script_output(stl, gold, tmp_path):
    pt = load_package()
    text = open(os.path.join(OBJ, stl), "rb").read()
    want = np.fromfile(gold, dtype="<f4")
    assert _same(pt.stl_format(text, 1.0), want)
    out = tmp_path / "o.bin"
    n = pt.stl_format_file(os.path.join(OBJ, stl), str(out), 1.0)
    assert n == want.size
    assert open(out, "rb").read() == open(gold, "rb").read()


@pytest.mark.parametrize("scale", [1.0, 0.1, 3.0, 1e-3])
def test_product_matches_oracle_with_scale(scale):
    pt = load_package()
    text = open(os.path.join(OBJ, "small_teapot.stl"), "rb").read()
    assert _same(pt.stl_format(text, scale), so.stl_format(text, scale))


@pytest.mark.parametrize("i", range(len(TRICKY)))
def test_product_matches_oracle_on_awkward_text(i):
    pt = load_package()
    a, b = pt.stl_format(TRICKY[i], 1.0), so.stl_format(TRICKY[i], 1.0)
    assert a.shape == b.shape
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b)
    assert np.array_equal(a[~nan_a].view(np.uint32), b[~nan_b].view(np.uint32))


def test_expected_counts_of_awkward_text():
    assert so.stl_format(TRICKY[4]).tolist() == [1.0, 0.0, 2.0, 3.0]
    assert so.stl_format(TRICKY[5]).tolist() == [1.0, 2.0, 3.0, 0.0]
    assert so.stl_format(TRICKY[6]).tolist() == [1.0, 2.0, 3.0, 0.0]
    assert so.stl_format(TRICKY[10]).tolist() == [0.0]
    assert so.stl_format(TRICKY[11]).tolist() == [4.0, 5.0, 6.0]


def test_vertex_without_following_space_is_an_error():
    pt = load_package()
    with pytest.raises(ValueError):
        so.stl_format(b"endvertex\n")
    with pytest.raises(pt.PtError) as e:
        pt.stl_format(b"ok\nendvertex\n")
    assert e.value.code == -1 and "line 2" in str(e.value)


def test_missing_input_file(tmp_path):
    pt = load_package()
    with pytest.raises(pt.PtError) as e:
        pt.stl_format_file(str(tmp_path / "nope.stl"), str(tmp_path / "o.bin"))
    assert e.value.code == -4


def test_formatter_output_feeds_the_builder():
    """formatter -> builder chain of the reference's prepare step (data_formatter.js then
    prepare_data/src/main.rs): the result is a valid scene."""
    pt = load_package()
    tris = pt.stl_format(open(os.path.join(OBJ, "small_teapot.stl"), "rb").read(), 1.0)
    assert tris.size % 9 == 0
    out_tr, nodes = pt.bvh_build(tris.reshape(-1, 9), 8)
    assert nodes.shape[1] == 9 and out_tr.shape == (tris.size // 9, 9)
    assert sorted(map(tuple, out_tr.tolist())) == sorted(map(tuple, tris.reshape(-1, 9).tolist()))


def test_pt_prepare_tool_runs_both_preparation_steps(tmp_path):
    """pt_prepare = data_formatter.js followed by prepare_data/src/main.rs: its files equal what the
    library calls give, and its --bin mode (step 2 only) reproduces the reference's teapot pair from
    the triangle file the reference's formatter wrote."""
    import subprocess
    pt = load_package()
    tool = os.path.join(ROOT, "path-tracing_amd", "pt_prepare")
    assert os.path.exists(tool), "build pt_prepare with make -C path-tracing_amd"
    b, v = str(tmp_path / "m.bin"), str(tmp_path / "m_bvh.bin")
    res = subprocess.run([tool, os.path.join(OBJ, "small_teapot.stl"), b, v, "--scale", "0.5", "--max-depth", "9"],
                         capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    tris = pt.stl_format(open(os.path.join(OBJ, "small_teapot.stl"), "rb").read(), 0.5)
    out_tr, nodes = pt.bvh_build(tris.reshape(-1, 9), 9)
    assert open(b, "rb").read() == out_tr.tobytes() and open(v, "rb").read() == nodes.tobytes()
    # step 2 alone (the Rust tool's job), default depth limit
    import shutil
    shutil.copy(os.path.join(OBJ, "teapot.bin"), tmp_path / "t.bin")
    res = subprocess.run([tool, "--bin", str(tmp_path / "t.bin"), str(tmp_path / "t_bvh.bin")],       # default depth 16
                         capture_output=True, text=True, timeout=60)
    assert res.returncode == 0, res.stderr
    t2, n2 = pt.bvh_build(np.fromfile(os.path.join(OBJ, "teapot.bin"), "<f4").reshape(-1, 9), 16)
    assert open(tmp_path / "t.bin", "rb").read() == t2.tobytes() and open(tmp_path / "t_bvh.bin", "rb").read() == n2.tobytes()
    assert subprocess.run([tool, "nope.stl", b, v], capture_output=True).returncode == 1
    assert subprocess.run([tool, "only_one_argument"], capture_output=True).returncode == 2
