"""Known-answer tests of the CPU oracle's pieces against hand-computed values and against
independent restatements written here in Python integers / float64.  CPU only.

The reference has no tests or golden vectors for its shader (parity unpinned), so these
pins come from reading src/shaders/shader.wgsl; each test cites the lines it checks."""
import math

import numpy as np
import pytest

F = np.float32


def xorshift_py(seed):
    """shader.wgsl:351-358 in Python integers."""
    s = seed & 0xFFFFFFFF
    s ^= (s << 13) & 0xFFFFFFFF
    s ^= s >> 17
    s ^= (s << 5) & 0xFFFFFFFF
    return s


@pytest.mark.parametrize("seed", [0, 1, 2, 12345, 248135, 393939123, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 2072640 + 960])
def test_random_value_matches_integer_restatement(oracle, seed):
    want = F(F(xorshift_py(seed)) / F(4294967296.0))
    assert F(oracle.random_value(seed)) == want


def test_random_value_hand_computed(oracle):
    # seed 1: 1 ^ (1<<13) = 0x2001; >>17 adds nothing; ^ (<<5) = 0x2001 ^ 0x40020 = 0x42021 = 270369
    assert oracle.random_value(1) == 270369.0 / 4294967296.0
    assert oracle.random_value(0) == 0.0
    # f32(u32) rounds to nearest, so a state >= 2^32 - 128 maps to exactly 1.0
    top = [s for s in range(1, 400000) if xorshift_py(s) >= 0xFFFFFF80]
    for s in top[:3]:
        assert oracle.random_value(s) == 1.0


def test_random_value_range_and_mean(oracle):
    vals = np.array([oracle.random_value(s) for s in range(1, 20001)])
    assert vals.min() >= 0.0 and vals.max() <= 1.0


def test_random_normal_formula(oracle):
    # shader.wgsl:361-366: sqrt(-2 log u1) * cos(2 * 3.14159 * u2), u2 from seed*7462 (u32 wrap)
    for seed in [1, 77, 99991, 0xDEADBEEF, 4000000000]:
        u1 = float(oracle.random_value(seed))
        u2 = float(oracle.random_value((seed * 7462) & 0xFFFFFFFF))
        want = math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * 3.14159 * u2)
        assert oracle.random_normal(seed) == pytest.approx(want, rel=2e-5, abs=2e-6)


def test_random_direction_is_unit_and_uses_seed_multiples(oracle):
    for seed in [3, 123456, 987654321]:
        d = oracle.random_direction(seed)
        assert abs(float(np.linalg.norm(d.astype(np.float64))) - 1.0) < 1e-6
        comp = np.array([oracle.random_normal(seed), oracle.random_normal((seed * 379) & 0xFFFFFFFF),
                         oracle.random_normal((seed * 123) & 0xFFFFFFFF)], dtype=np.float64)
        np.testing.assert_allclose(d, comp / np.linalg.norm(comp), rtol=1e-5, atol=1e-6)


def test_seed_zero_gives_nan_direction(oracle):
    # random_value(0) = 0 -> log(0) = -inf -> sqrt(inf) * cos(0) = inf in all three
    # components -> normalize(inf, inf, inf) = NaN (shader.wgsl:351-375)
    assert math.isinf(oracle.random_normal(0))
    assert np.isnan(oracle.random_direction(0)).all()


# ---- ray_box, shader.wgsl:237-256 --------------------------------------------
def test_ray_box_axis_aligned_with_zero_direction_components(oracle):
    box = [1, -1, -1, 2, 1, 1]
    assert oracle.ray_box([0, 0, 0], [1, 0, 0], box) == 1.0          # +-inf on the zero axes
    assert oracle.ray_box([1.5, 0, 0], [1, 0, 0], box) == -0.5       # origin inside: negative t_near
    assert oracle.ray_box([3, 0, 0], [1, 0, 0], box) == F(999999999.0)  # behind: t_far < 0
    assert oracle.ray_box([0, 5, 0], [1, 0, 0], box) == F(999999999.0)  # slab miss
    assert F(999999999.0) == 1e9                                      # the constant rounds to 1e9 in f32


def test_ray_box_nan_from_zero_over_zero_is_ignored_by_min_max(oracle):
    # origin on the y = -1 plane, dir.y = 0: (bmin.y - o.y)/0 = 0/0 = NaN, (bmax.y - o.y)/0 = +inf.
    # minNum/maxNum drop the NaN: t1.y = t2.y = +inf -> t_near = inf > t_far -> miss.
    assert oracle.ray_box([0, -1, 0], [1, 0, 0], [1, -1, -1, 2, 1, 1]) == F(999999999.0)


def test_ray_box_general_matches_float64(oracle):
    rng = np.random.default_rng(5)
    for _ in range(200):
        o = rng.uniform(-3, 3, 3).astype(F)
        d = rng.uniform(-1, 1, 3).astype(F)
        lo = rng.uniform(-2, 0, 3).astype(F)
        hi = (lo + rng.uniform(0.1, 2, 3)).astype(F)
        t0 = (lo.astype(np.float64) - o) / d
        t1 = (hi.astype(np.float64) - o) / d
        tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
        got = oracle.ray_box(o, d, list(lo) + list(hi))
        if tn <= tf - 1e-5 and tf >= 1e-5:
            assert got == pytest.approx(tn, rel=1e-5, abs=1e-5)
        elif tn > tf + 1e-5 or tf < -1e-5:
            assert got == F(999999999.0)


# ---- ray_triangle, shader.wgsl:258-290 ---------------------------------------
TRI = [0, 0, 0, 1, 0, 0, 0, 1, 0]   # normal (0,0,1)


def test_ray_triangle_front_face_hit(oracle):
    p = oracle.default_params()
    h = oracle.ray_triangle(p, [0.25, 0.25, 1], [0, 0, -1], TRI)
    assert h.did_hit == 1 and h.distance == 1.0
    assert list(h.position) == [0.25, 0.25, 0.0]
    assert list(h.normal) == [0.0, 0.0, 1.0]
    np.testing.assert_array_equal(np.array(h.color[:], F), np.array([0.65, 0.65, 1.0], F))   # :283
    assert h.emission_strength == 0.0 and h.smoothness == F(0.9)                             # :285-286


def test_ray_triangle_back_face_and_outside(oracle):
    p = oracle.default_params()
    assert oracle.ray_triangle(p, [0.25, 0.25, -1], [0, 0, 1], TRI).did_hit == 0   # determinant < 0.0001
    assert oracle.ray_triangle(p, [0.75, 0.75, 1], [0, 0, -1], TRI).did_hit == 0   # w < 0
    assert oracle.ray_triangle(p, [0.25, 0.25, -1], [0, 0, -1], TRI).did_hit == 0  # dst < 0
    # edge and vertex hits count (>= 0 tests)
    assert oracle.ray_triangle(p, [0.5, 0.0, 1], [0, 0, -1], TRI).did_hit == 1
    assert oracle.ray_triangle(p, [0.0, 0.0, 1], [0, 0, -1], TRI).did_hit == 1


def test_ray_triangle_distance_in_units_of_dir_length(oracle):
    p = oracle.default_params()
    h = oracle.ray_triangle(p, [0.25, 0.25, 1], [0, 0, -2], TRI)
    assert h.did_hit == 1 and h.distance == 0.5


def test_ray_triangle_determinant_threshold(oracle):
    p = oracle.default_params()
    # grazing ray: determinant = -dot(dir, n) = 0.00005 < 0.0001 -> rejected although geometrically a hit
    assert oracle.ray_triangle(p, [0.25, 0.25, 0.00005], [0.0, 0.0, -0.00005], TRI).did_hit == 0
    assert oracle.ray_triangle(p, [0.25, 0.25, 0.0002], [0.0, 0.0, -0.0002], TRI).did_hit == 1


# ---- ray_sphere, shader.wgsl:292-325 -----------------------------------------
SPH = [0, 0, 0, 1, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 7.0, 0.25]


def test_ray_sphere_hits_and_material(oracle):
    h = oracle.ray_sphere([0, 0, 5], [0, 0, -1], SPH)
    assert h.did_hit == 1 and h.distance == 4.0
    assert list(h.position) == [0.0, 0.0, 1.0] and list(h.normal) == [0.0, 0.0, 1.0]
    np.testing.assert_array_equal(np.array(h.color[:], F), np.array([0.1, 0.2, 0.3], F))
    np.testing.assert_array_equal(np.array(h.emission_color[:], F), np.array([0.4, 0.5, 0.6], F))
    assert h.emission_strength == 7.0 and h.smoothness == 0.25


def test_ray_sphere_non_unit_direction_inside_and_behind(oracle):
    assert oracle.ray_sphere([0, 0, 5], [0, 0, -2], SPH).distance == 2.0   # a = dot(dir,dir) = 4
    assert oracle.ray_sphere([0, 0, 0], [0, 0, -1], SPH).did_hit == 0     # near root only: none from inside
    assert oracle.ray_sphere([0, 0, 5], [0, 0, 1], SPH).did_hit == 0      # behind
    assert oracle.ray_sphere([0, 2, 5], [0, 0, -1], SPH).did_hit == 0     # discriminant < 0
    h = oracle.ray_sphere([0, 1, 5], [0, 0, -1], SPH)                      # tangent: discriminant == 0
    assert h.did_hit == 1 and h.distance == 5.0


# ---- camera, shader.wgsl:56-95, 327-348 --------------------------------------
def test_rotate_vector_matches_float64_with_pi_3_14159(oracle):
    rng = np.random.default_rng(11)
    for _ in range(50):
        v = rng.uniform(-1, 1, 3)
        ang = rng.uniform(-180, 180, 3)
        a, b, c = ang * 3.14159 / 180.0
        ca, sa, cb, sb, cc, sc = math.cos(a), math.sin(a), math.cos(b), math.sin(b), math.cos(c), math.sin(c)
        x, y, z = v
        want = [x * cc * cb + y * (cc * sb * sa - sc * ca) + z * (cc * sb * ca + sc * sa),
                x * sc * cb + y * (sc * sb * sa + cc * ca) + z * (sc * sb * ca - cc * sa),
                -x * sb + y * cb * sa + z * cb * ca]
        np.testing.assert_allclose(oracle.rotate_vector(v, ang), want, rtol=0, atol=3e-6)


def test_rotation_conventions(oracle):
    # rotation (0,0,0) looks along -z; (0,90,0) looks along -x (to ~1e-6: pi is 3.14159); (90,0,0) along +y
    np.testing.assert_array_equal(oracle.rotate_vector([0, 0, -1], [0, 0, 0]), [0, 0, -1])
    np.testing.assert_allclose(oracle.rotate_vector([0, 0, -1], [0, 90, 0]), [-1, 0, 0], atol=2e-6)
    np.testing.assert_allclose(oracle.rotate_vector([0, 0, -1], [90, 0, 0]), [0, 1, 0], atol=2e-6)
    np.testing.assert_allclose(oracle.rotate_vector([0, 1, 0], [90, 0, 0]), [0, 0, 1], atol=2e-6)


def test_screen_size_and_pixel_index(oracle):
    p = oracle.default_params(width=1920, height=1080)
    sw, sh = oracle.screen_size(p)
    fov = np.float32(60.0 * 3.14159 / 180.0)        # const-expression: f64, rounded once
    assert sw == pytest.approx(2.0 * math.tan(float(fov) * 0.5), rel=1e-6)
    assert sh == pytest.approx(sw / (1920.0 / 1080.0), rel=1e-6)
    # pixel_index = u32(pos.x + pos.y * W) in f32 = x + y*W + W/2 for even W (shader.wgsl:83)
    for (x, y) in [(0, 0), (1919, 0), (0, 1079), (1919, 1079), (960, 540)]:
        _, _, idx = oracle.primary_ray(p, [0, 0, 0], [0, 0, 0], x, y)
        assert idx == x + y * 1920 + 960
    p4k = oracle.default_params(width=3840, height=2160)
    _, _, idx = oracle.primary_ray(p4k, [0, 0, 0], [0, 0, 0], 3839, 2159)
    assert idx == 3839 + 2159 * 3840 + 1920


def test_primary_ray_direction(oracle):
    p = oracle.default_params(width=640, height=360)
    sw, sh = oracle.screen_size(p)
    o, d, _ = oracle.primary_ray(p, [1, 2, 3], [0, 0, 0], 0, 0)
    np.testing.assert_array_equal(o, [1, 2, 3])
    u = (2.0 * 0.5 / 640.0 - 1.0) * sw / 2.0
    v = (1.0 - 2.0 * 0.5 / 360.0) * sh / 2.0
    want = np.array([u, v, -1.0]) / math.sqrt(u * u + v * v + 1.0)
    np.testing.assert_allclose(d, want, rtol=1e-6, atol=1e-7)
    assert d[0] < 0 < d[1]        # pixel (0,0) is the TOP left: y down


# ---- environment, shader.wgsl:391-404 ----------------------------------------
def env_f64(p, d):
    def smooth(e0, e1, x):
        t = min(max((x - e0) / (e1 - e0), 0.0), 1.0)
        return t * t * (3 - 2 * t)
    hor, zen, gr = np.array(p.sky_horizon[:]), np.array(p.sky_zenith[:]), np.array(p.ground[:])
    t = smooth(0.0, 0.4, d[1]) ** 0.35
    sky = hor * (1 - t) + zen * t
    s = np.array(p.sun_dir[:], dtype=np.float64)
    s = -s / np.linalg.norm(s)
    sun = max(0.0, float(np.dot(d, s))) ** p.sun_focus * p.sun_intensity
    g = smooth(-0.01, 0.0, d[1])
    return gr * (1 - g) + sky * g + (sun if g >= 1 else 0.0)


def test_environment_light(oracle):
    p = oracle.default_params()
    np.testing.assert_array_equal(oracle.environment_light(p, [0, -1, 0]), np.array([0.2, 0.2, 0.2], F))
    rng = np.random.default_rng(3)
    for _ in range(100):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        d = d.astype(F)
        np.testing.assert_allclose(oracle.environment_light(p, d), env_f64(p, d.astype(np.float64)), rtol=2e-4, atol=1e-6)
    # looking into the sun: the pow(.., 200) * 3 lobe is added only above the horizon
    s = np.array([0, 0.4, -0.5]) / math.sqrt(0.41)
    assert oracle.environment_light(p, s.astype(F))[0] > 2.9
    below = np.array([0.0, -0.005, -1.0])
    assert oracle.environment_light(p, below.astype(F)).max() < 1.01   # ground_to_sky_t < 1: no sun term


def test_random_hemisphere_direction_is_the_direction_flipped_into_the_hemisphere(oracle):
    """shader.wgsl:378-382 (dead code in the shader): dir * sign(dot(normal, dir))."""
    for seed in (1, 2, 77, 123456789, 0xFFFFFFFF):
        d = oracle.random_direction(seed)
        for normal in ((0.0, 1.0, 0.0), (0.3, -0.2, 0.9), (-1.0, 0.0, 0.0)):
            h = oracle.random_hemisphere_direction(normal, seed)
            dot = np.float32(normal[0]) * d[0] + np.float32(normal[1]) * d[1] + np.float32(normal[2]) * d[2]
            want = d if dot > 0 else (-d if dot < 0 else d * 0)
            assert np.array_equal(h, want)
            assert float(np.dot(np.asarray(normal, np.float32), h)) >= 0.0
    # a direction perpendicular to the normal is multiplied by sign(0) = 0
    d = oracle.random_direction(5)
    n = np.cross(d, np.asarray([1, 0, 0], np.float32)).astype(np.float32)
    n64 = np.cross(d.astype(np.float64), [1.0, 0.0, 0.0])
    if np.float32(n[0]) * d[0] + np.float32(n[1]) * d[1] + np.float32(n[2]) * d[2] == 0:
        assert np.array_equal(oracle.random_hemisphere_direction(n, 5), d * 0) and n64 is not None
