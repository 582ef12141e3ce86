"""Host camera motion (State::update_camera / rotate_vector, src/main.rs:45-142): the C entry
points against the numpy-f32 restatement and against hand-computed cases.  The trigonometric
values may differ in the last bit between libms, so comparisons that involve a rotated vector
allow a few ulps; cases with rotation (0, 0, 0) or multiples of 90 degrees are checked exactly
where the arithmetic is exact."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import camera_oracle as co  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def _close(a, b, ulps=4):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), np.float32(1.0))
    return bool(np.all(np.abs(a - b) <= ulps * np.finfo(np.float32).eps * scale))


def test_symbols_and_key_table():
    pt = load_package()
    assert pt.KEYS == co.KEYS
    assert pt.CAMERA_SPEED == 1.0 and pt.CAMERA_ROT_SPEED == 1.0


def test_no_key_changes_nothing_and_does_not_reset():
    pt = load_package()
    pos, rot, any_key = pt.update_camera([], (200.0, 0.0, 200.0), (0.0, 90.0, 0.0))
    assert pos == [200.0, 0.0, 200.0] and rot == [0.0, 90.0, 0.0] and any_key is False


@pytest.mark.parametrize("key,dpos,drot", [
    ("W", (0, 0, -1), (0, 0, 0)), ("S", (0, 0, 1), (0, 0, 0)), ("D", (1, 0, 0), (0, 0, 0)), ("A", (-1, 0, 0), (0, 0, 0)),
    ("Space", (0, 1, 0), (0, 0, 0)), ("Shift", (0, -1, 0), (0, 0, 0)),
    ("I", (0, 0, 0), (1, 0, 0)), ("K", (0, 0, 0), (-1, 0, 0)), ("L", (0, 0, 0), (0, 1, 0)), ("J", (0, 0, 0), (0, -1, 0)),
    ("O", (0, 0, 0), (0, 0, 1)), ("U", (0, 0, 0), (0, 0, -1)),
])
def test_each_key_with_identity_rotation_is_exact(key, dpos, drot):
    """rotation (0,0,0): cos = 1, sin = 0 exactly, so local axes are the global ones (main.rs:54-102)."""
    pt = load_package()
    pos, rot, any_key = pt.update_camera([key], (5.0, 35.0, 150.0), (0.0, 0.0, 0.0))
    assert any_key
    assert pos == [5.0 + dpos[0], 35.0 + dpos[1], 150.0 + dpos[2]]
    assert rot == [float(drot[0]), float(drot[1]), float(drot[2])]


def test_opposite_keys_cancel_but_still_reset():
    pt = load_package()
    pos, rot, any_key = pt.update_camera(["W", "S", "I", "K"], (1.0, 2.0, 3.0), (10.0, 20.0, 30.0))
    assert any_key and pos == [1.0, 2.0, 3.0] and rot == [10.0, 20.0, 30.0]


def test_forward_with_the_reference_start_pose():
    """Start pose of the reference (main.rs:382-383): rotation (0, 90, 0).  Host rotate_vector of
    local -z: x_rot = z*cos_c*sin_b*cos_a = -1*1*sin(90 deg)*1, z_rot = z*cos_b*cos_a ~ 4e-8."""
    pt = load_package()
    pos, rot, _ = pt.update_camera(["W"], (200.0, 0.0, 200.0), (0.0, 90.0, 0.0))
    assert _close(pos, [199.0, 0.0, 200.0]) and rot == [0.0, 90.0, 0.0]


def test_against_numpy_restatement_random_poses():
    pt = load_package()
    rng = np.random.default_rng(11)
    for _ in range(300):
        keys = [k for k in co.KEYS if rng.random() < 0.3]
        pos = rng.uniform(-300, 300, 3).astype(np.float32)
        rot = rng.uniform(-400, 400, 3).astype(np.float32)
        p1, r1, a1 = pt.update_camera(keys, pos, rot)
        p2, r2, a2 = co.update_camera(set(keys), pos, rot)
        assert a1 == a2 == bool(keys)
        assert _close(p1, p2) and _close(r1, r2)
        v = rng.uniform(-2, 2, 3).astype(np.float32)
        assert _close(pt.host_rotate_vector(v, rot), co.rotate_vector(v, rot))


def test_restatement_with_double_trig_is_within_ulps():
    """Independent of any f32 libm: cos/sin evaluated in double and rounded once."""
    pt = load_package()
    rng = np.random.default_rng(5)
    c64 = lambda x: np.float32(np.cos(np.float64(x)))      # noqa: E731
    s64 = lambda x: np.float32(np.sin(np.float64(x)))      # noqa: E731
    for _ in range(100):
        rot = rng.uniform(-180, 180, 3).astype(np.float32)
        v = rng.uniform(-1, 1, 3).astype(np.float32)
        assert _close(pt.host_rotate_vector(v, rot), co.rotate_vector(v, rot, c64, s64), ulps=8)


def test_rotation_uses_the_pose_before_the_tick():
    """main.rs:105 and :111 both read camera_rotation before :112-114 change it."""
    pt = load_package()
    pos, rot, _ = pt.update_camera(["W", "L"], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    assert pos == [0.0, 0.0, -1.0] and rot == [0.0, 1.0, 0.0]
    pos2, rot2, _ = pt.update_camera(["W"], pos, rot)              # now one degree of yaw applies
    want = co.update_camera({"W"}, pos, rot)
    assert _close(pos2, want[0]) and pos2[0] != 0.0 and rot2 == [0.0, 1.0, 0.0]


def test_a_walk_stays_close_to_the_restatement():
    pt = load_package()
    script = [["W"]] * 10 + [["W", "L"]] * 20 + [["D", "I"]] * 15 + [["Space", "O"]] * 5 + [[]] * 3
    p1, r1 = [200.0, 0.0, 200.0], [0.0, 90.0, 0.0]
    p2, r2 = list(p1), list(r1)
    for keys in script:
        p1, r1, _ = pt.update_camera(keys, p1, r1)
        p2, r2, _ = co.update_camera(set(keys), p2, r2)
    assert _close(p1, p2, ulps=64) and _close(r1, r2, ulps=64)


def test_null_arguments_are_rejected():
    pt = load_package()
    L = pt.load_library()
    assert L.pt_update_camera(None, 1.0, 1.0, None, None, None) == -1
    assert L.pt_host_rotate_vector(None, None, None) == -1
