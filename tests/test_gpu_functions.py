"""The small shader functions evaluated on the device (pt_probe_functions) against the CPU
restatement, one by one: random_value, random_normal, random_direction,
random_hemisphere_direction, get_environment_light, rotate_vector (shader.wgsl:327-404).
Bit-exact in exact-math mode; in the default mode (hardware log/cos/exp) within the stated
tolerance."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

ROT = (12.5, -100.0, 33.0)
NORMAL = (0.3, -0.2, 0.9)


def _inputs(n=4000):
    rng = np.random.default_rng(99)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    seeds[:6] = [0, 1, 2, 0xFFFFFFFF, 0x80000000, 12345]
    dirs = rng.normal(0, 1, (n, 3)).astype(np.float32)
    dirs[:4] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, -0.005, 1]]        # zenith, nadir, horizon, inside the ground blend
    dirs[4] = [0.0, 0.4, -0.5]                                            # straight into the sun (-sun_dir)
    return seeds, dirs


def _oracle_values(oracle, params, seeds, dirs):
    out = {k: [] for k in ("random_value", "random_normal", "random_direction", "random_hemisphere_direction",
                           "environment_light", "rotate_vector")}
    for s, d in zip(seeds, dirs):
        s = int(s)
        out["random_value"].append(oracle.random_value(s))
        out["random_normal"].append(oracle.random_normal(s))
        out["random_direction"].append(oracle.random_direction(s))
        out["random_hemisphere_direction"].append(oracle.random_hemisphere_direction(NORMAL, s))
        out["environment_light"].append(oracle.environment_light(params, d))
        out["rotate_vector"].append(oracle.rotate_vector(d, ROT))
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def test_functions_bit_exact_in_exact_math_mode(pkg, oracle_shared):
    seeds, dirs = _inputs()
    r = pkg.Renderer(pkg.default_params(width=16, height=16, flags=pkg.PT_FLAG_EXACT_MATH), device=0)
    with r:
        r.set_camera((0, 0, 0), ROT)
        got = r.probe_functions(seeds, dirs, NORMAL)
    want = _oracle_values(oracle_shared, oracle_shared.default_params(width=16, height=16), seeds, dirs)
    for k in want:
        same = bits(got[k]) == bits(want[k])
        both_nan = np.isnan(got[k]) & np.isnan(want[k])          # random_normal(0) is sqrt(-2 log 0) * cos(0): inf
        assert (same | both_nan).all(), k
    assert want["random_value"][0] == 0.0 and np.isinf(want["random_normal"][0])
    assert (want["random_value"] <= 1.0).all()


def test_functions_default_math_within_tolerance(pkg, oracle):
    """Hardware v_log/v_cos/v_exp against libm.  random_value and rotate_vector use no
    transcendental on the device and stay bit-exact; the others: 1e-5 relative (+1e-6 absolute)
    for the Box-Muller values away from the pole of log at 0, 2e-4 for the two pow() of the sky."""
    seeds, dirs = _inputs()
    r = pkg.Renderer(pkg.default_params(width=16, height=16), device=0)
    with r:
        r.set_camera((0, 0, 0), ROT)
        got = r.probe_functions(seeds, dirs, NORMAL)
    want = _oracle_values(oracle, oracle.default_params(width=16, height=16), seeds, dirs)
    assert np.array_equal(bits(got["random_value"]), bits(want["random_value"]))
    assert np.array_equal(bits(got["rotate_vector"]), bits(want["rotate_vector"]))
    ok = np.isfinite(want["random_normal"])
    np.testing.assert_allclose(got["random_normal"][ok], want["random_normal"][ok], rtol=1e-5, atol=2e-6)
    fin = np.isfinite(want["random_direction"]).all(axis=1)
    np.testing.assert_allclose(got["random_direction"][fin], want["random_direction"][fin], rtol=0, atol=2e-5)
    np.testing.assert_allclose(got["random_hemisphere_direction"][fin], want["random_hemisphere_direction"][fin], rtol=0, atol=2e-5)
    np.testing.assert_allclose(got["environment_light"], want["environment_light"], rtol=2e-4, atol=1e-6)
