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


def _intersection_inputs(n=20000):
    rng = np.random.default_rng(7)
    o = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    lo = rng.uniform(-15, 10, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 12, (n, 3)).astype(np.float32)).astype(np.float32)
    tri = rng.uniform(-10, 10, (n, 9)).astype(np.float32)
    sph = np.concatenate([rng.uniform(-10, 10, (n, 3)), rng.uniform(0.1, 8, (n, 1))], axis=1).astype(np.float32)
    k = n // 10
    # rays aimed at a point of their triangle (inside, on an edge, at a vertex) and at their sphere / box
    w = rng.dirichlet((1, 1, 1), k).astype(np.float32)
    w[: k // 3, 2] = 0
    w[: k // 6, 1] = 0
    w /= w.sum(axis=1, keepdims=True)
    t3 = tri[:k].reshape(k, 3, 3)
    d[:k] = (np.einsum("ki,kij->kj", w, t3) - o[:k]).astype(np.float32)
    d[k:2 * k] = (sph[k:2 * k, :3] + rng.normal(0, 0.7, (k, 3)) * sph[k:2 * k, 3:4] - o[k:2 * k]).astype(np.float32)
    d[2 * k:3 * k] = ((lo[2 * k:3 * k] + hi[2 * k:3 * k]) / 2 + rng.normal(0, 3, (k, 3)) - o[2 * k:3 * k]).astype(np.float32)
    # awkward numbers: zero / negative-zero / tiny / huge direction components, origins on box planes and
    # inside boxes, flat boxes, significands that the reciprocal form must refuse
    s = slice(3 * k, 4 * k)
    d[s, rng.integers(0, 3, k)] = rng.choice(np.asarray([0.0, -0.0, 1e-30, -1e-30, 1e30, 1e-8], np.float32), k)
    s = slice(4 * k, 5 * k)
    ax = rng.integers(0, 3, k)
    o[s][np.arange(k), ax] = lo[s][np.arange(k), ax]                     # origin exactly on the min plane
    d[4 * k:4 * k + k // 2][np.arange(k // 2), ax[: k // 2]] = 0.0       # ... and moving within it: 0/0
    s = slice(5 * k, 6 * k)
    o[s] = ((lo[s] + hi[s]) / 2).astype(np.float32)                      # origin inside the box
    s = slice(6 * k, 7 * k)
    hi[s][np.arange(k), rng.integers(0, 3, k)] = lo[s][np.arange(k), rng.integers(0, 3, k)]
    s = slice(7 * k, 8 * k)
    d[s] = (d[s].view(np.uint32) | np.uint32(0x7FFF00)).view(np.float32)   # top of the significand range
    return np.concatenate([o, d], axis=1), np.concatenate([lo, hi], axis=1), tri, sph


def test_intersection_functions_bit_exact(pkg, oracle):
    """ray_box (both division back ends), ray_triangle and ray_sphere (shader.wgsl:237-325) on 20000
    random and awkward inputs: did_hit flags equal, distances bit-exact (NaN compares equal to NaN)."""
    rays, boxes, tris, sph = _intersection_inputs()
    r = pkg.Renderer(pkg.default_params(width=16, height=16), device=0)
    with r:
        got = r.probe_intersections(rays, boxes, tris, sph)
    p = oracle.default_params(width=16, height=16)
    n = len(rays)
    want_box = np.empty(n, np.float32)
    want_th, want_td = np.zeros(n, bool), np.zeros(n, np.float32)
    want_sh, want_sd = np.zeros(n, bool), np.zeros(n, np.float32)
    for i in range(n):
        o, d = rays[i, :3], rays[i, 3:]
        want_box[i] = oracle.ray_box(o, d, boxes[i])
        h = oracle.ray_triangle(p, o, d, tris[i])
        want_th[i], want_td[i] = bool(h.did_hit), h.distance
        s12 = np.zeros(12, np.float32)
        s12[:4] = sph[i]
        h = oracle.ray_sphere(o, d, s12)
        want_sh[i], want_sd[i] = bool(h.did_hit), h.distance

    def same(a, b):
        return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))

    assert same(got["box"], want_box).all()
    used = ~np.isnan(got["box_reciprocal"]) | np.isnan(got["box"])
    assert 0.5 < used.mean() < 0.95                       # both the accepted and the refused inputs are exercised
    assert same(got["box_reciprocal"][used], want_box[used]).all()
    assert 0.1 < (want_box < 1e8).mean() < 0.9
    assert np.array_equal(got["tri_hit"], want_th) and want_th.sum() > 200
    assert same(got["tri_distance"][want_th], want_td[want_th]).all()
    assert np.array_equal(got["sphere_hit"], want_sh) and want_sh.sum() > 500
    assert same(got["sphere_distance"][want_sh], want_sd[want_sh]).all()
