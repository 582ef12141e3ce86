"""Property tests of the CPU oracle on the reference's committed scene files.  CPU only."""
import numpy as np
import pytest

from conftest import bits

F = np.float32


def _rays_at_mesh(tris, n, seed):
    rng = np.random.default_rng(seed)
    v = tris.reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    centre, ext = (lo + hi) / 2, (hi - lo).max()
    o = (centre + rng.normal(size=(n, 3)) * ext * 1.2).astype(F)
    target = (lo + rng.random((n, 3)) * (hi - lo)).astype(F)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    return o, d


def _covered(bvh, ntris):
    """Triangles referenced by at least one reachable leaf of the node file."""
    cov = np.zeros(ntris, bool)
    stack = [0]
    while stack:
        n = stack.pop()
        node = bvh[n]
        if node[8] == 0:
            cov[int(node[6]):int(node[6] + node[7])] = True
        else:
            stack += [int(node[8]), int(node[8]) + 1]
    return cov


@pytest.mark.parametrize("name", ["teapot", "knight", "dragon_8k"])
def test_bvh_traversal_equals_brute_force(oracle, scenes, name):
    """Closest hit through the BVH == closest hit over all triangles in index order: distance
    bit-exact; ids equal except among exactly equal distances (the BVH visits triangles in a
    different order, and strict `<` keeps the first one tested, shader.wgsl:192)."""
    tris, bvh = scenes.mesh(name)
    p = oracle.default_params()
    sc = oracle.scene(np.zeros((0, 12), F), tris, bvh)
    covered = _covered(bvh, len(tris))
    o, d = _rays_at_mesh(tris, 400, seed={"teapot": 1, "knight": 2, "dragon_8k": 3}[name])
    hits = 0
    for i in range(len(o)):
        hb = oracle.closest_hit(p, sc, o[i], d[i], bvh_only=True)
        hf = oracle.closest_hit(p, sc, o[i], d[i], brute=True)
        if hf.did_hit and not covered[hf.prim_id]:
            continue     # dragon_8k's legacy file leaves one triangle in no leaf: the shader cannot see it either
        assert hb.did_hit == hf.did_hit
        if hf.did_hit:
            hits += 1
            assert bits(F(hb.distance)) == bits(F(hf.distance))
            if hb.prim_id != hf.prim_id:
                other = oracle.ray_triangle(p, o[i], d[i], tris[hb.prim_id])
                assert other.did_hit and bits(F(other.distance)) == bits(F(hf.distance))
            np.testing.assert_array_equal(np.array(hb.position[:]), np.array(hf.position[:]))
    assert hits > 50


def test_spheres_win_ties_and_lower_index_wins(oracle, scenes):
    """calculate_ray_collision (shader.wgsl:139-163): strict `<`, spheres first in index order."""
    tris = np.array([[-5, -5, 0, 5, -5, 0, 0, 5, 0]], F)        # a triangle in the z = 0 plane, facing +z
    bvh = np.array([[-5, -5, 0, 5, 5, 0, 0, 1, 0]], F)           # single leaf = root
    sph = np.array([[0, 0, -1, 1, 1, 0, 0, 0, 0, 0, 0, 0],        # touches z = 0 from below at the origin
                    [0, 0, -1, 1, 0, 1, 0, 0, 0, 0, 0, 0]], F)    # identical second sphere
    p = oracle.default_params()
    sc = oracle.scene(sph, tris, bvh)
    h = oracle.closest_hit(p, sc, [0, 0, 3], [0, 0, -1])
    assert h.did_hit and h.distance == 3.0
    assert h.prim_id == -1                                         # sphere 0, not sphere 1, not the triangle
    sc2 = oracle.scene(np.zeros((0, 12), F), tris, bvh)
    assert oracle.closest_hit(p, sc2, [0, 0, 3], [0, 0, -1]).prim_id == 0


def test_far_clip_is_one_million(oracle):
    """Hits at distance >= 1e6 are discarded (initial closest distance, shader.wgsl:142,177)."""
    tris = np.array([[-5, -5, 0, 5, -5, 0, 0, 5, 0]], F)
    bvh = np.array([[-5, -5, 0, 5, 5, 0, 0, 1, 0]], F)
    p = oracle.default_params()
    sc = oracle.scene(np.zeros((0, 12), F), tris, bvh)
    assert oracle.closest_hit(p, sc, [0, 0, 999999.0], [0, 0, -1]).did_hit == 1
    assert oracle.closest_hit(p, sc, [0, 0, 1000000.0], [0, 0, -1]).did_hit == 0


def _trace_py(oracle, p, sc, o, d, seed, frame_count):
    """trace (shader.wgsl:111-137) re-stated with numpy float32 on top of the oracle's
    intersection / RNG / environment pieces: checks the loop logic independently."""
    ro, rd = np.array(o, F), np.array(d, F)
    light, color = np.zeros(3, F), np.ones(3, F)
    for i in range(p.max_bounce_count + 1):               # inclusive loop bound
        h = oracle.closest_hit(p, sc, ro, rd)
        if h.did_hit:
            n = np.array(h.normal[:], F)
            ro = np.array(h.position[:], F)
            rs = (seed + i * 12345 + frame_count * 393939123) & 0xFFFFFFFF
            v = n + oracle.random_direction(rs)
            inv = F(1.0) / np.sqrt(F(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=F)
            diffuse = v * inv
            dn = F(F(n[0] * rd[0] + n[1] * rd[1]) + n[2] * rd[2])
            spec = rd - n * F(F(2.0) * dn)
            sm = F(h.smoothness)
            rd = diffuse * F(F(1.0) - sm) + spec * sm      # not re-normalised
            emitted = np.array(h.emission_color[:], F) * F(h.emission_strength)
            light = light + emitted * color
            color = color * np.array(h.color[:], F)
        else:
            if p.use_environment_lighting:
                light = light + oracle.environment_light(p, rd) * color
            break
    return light


def test_trace_loop_logic(oracle, scenes):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    sph = np.vstack([scenes.spheres_for("cfg1"), scenes.spheres_for("red_light")])
    p = oracle.default_params(width=64, height=36, max_bounce_count=4)
    sc = oracle.scene(sph, tris, bvh)
    checked = 0
    for (x, y) in [(32, 18), (20, 20), (40, 12), (10, 30), (50, 25), (32, 30), (5, 5)]:
        o, d, idx = oracle.primary_ray(p, cfg["camera_position"], cfg["camera_rotation"], x, y)
        for s in range(3):
            for frame in (0, 1, 11):
                seed = (idx + s * 248135) & 0xFFFFFFFF
                got = oracle.trace(p, sc, o, d, seed, frame)
                want = _trace_py(oracle, p, sc, o, d, seed, frame)
                np.testing.assert_array_equal(bits(got), bits(want))
                checked += 1
    assert checked == 63


def test_exhausted_bounces_add_no_sky(oracle):
    """Two facing mirrors: a path that is still bouncing when the loop ends returns only emission."""
    # floor facing +z at z = 0 and ceiling facing -z at z = 2, both far larger than the ray's walk
    tris = np.array([[-100, -100, 0, 100, -100, 0, 0, 100, 0],
                     [-100, -100, 2, 0, 100, 2, 100, -100, 2]], F)
    bvh = np.array([[-100, -100, 0, 100, 100, 2, 0, 2, 0]], F)
    p = oracle.default_params(max_bounce_count=3, tri_smoothness=1.0)
    sc = oracle.scene(np.zeros((0, 12), F), tris, bvh)
    out = oracle.trace(p, sc, [0, 0, 1], [0, 0, -1], 5, 0)
    np.testing.assert_array_equal(out, [0, 0, 0])
    p2 = oracle.default_params(max_bounce_count=3, tri_smoothness=1.0, tri_emission_color=(1, 1, 1),
                               tri_emission_strength=2.0, tri_color=(0.5, 0.5, 0.5))
    out2 = oracle.trace(p2, sc, [0, 0, 1], [0, 0, -1], 5, 0)
    np.testing.assert_array_equal(out2, np.array([2 + 1 + 0.5 + 0.25] * 3, F))    # 4 casts


def test_running_mean_equals_arithmetic_mean(oracle, scenes):
    """shader.wgsl:104-106: after frames 0..n-1 the buffer holds the mean of the per-frame means."""
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=32, height=18, rays_per_pixel=2, max_bounce_count=2)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    n = 5
    acc = oracle.render_frames(p, sc, cam, rot, 0, n)
    frames = [oracle.render_frames(p, sc, cam, rot, f, 1, accum=np.zeros((18, 32, 4), F) if f == 0 else None)
              for f in range(1)]
    per_frame = []
    for f in range(n):
        buf = np.zeros((18, 32, 4), F)
        oracle.render_frames_sum(p, sc, cam, rot, f, 1, 1, accum=buf)
        per_frame.append(buf.astype(np.float64))
    np.testing.assert_allclose(acc[..., :3], np.mean(per_frame, axis=0)[..., :3], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(bits(frames[0]), bits(per_frame[0].astype(F)))   # frame 0: weight 1
    # frame_count = 0 ignores what the buffer held
    junk = np.full((18, 32, 4), 7.0, F)
    a = oracle.render_frames(p, sc, cam, rot, 0, 1, accum=junk)
    np.testing.assert_array_equal(bits(a[..., :3]), bits(frames[0][..., :3]))


def test_samples_of_a_pixel_share_the_primary_ray_and_differ_by_seed(oracle, scenes):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=64, height=36, rays_per_pixel=4, max_bounce_count=4)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    o, d, idx = oracle.primary_ray(p, cfg["camera_position"], cfg["camera_rotation"], 30, 20)
    buf = np.zeros((36, 64, 4), F)
    oracle.render_frame_region(p, sc, cfg["camera_position"], cfg["camera_rotation"], 3, buf, 30, 31, 1, 20, 21, 1)
    s = np.zeros(3, F)
    for k in range(4):
        s = s + oracle.trace(p, sc, o, d, (idx + k * 248135) & 0xFFFFFFFF, 3)      # shader.wgsl:100
    mean = s / F(4.0)
    want = np.zeros(3, F) * F(1.0 - 0.25) + mean * F(0.25)                          # weight 1/(3+1)
    np.testing.assert_array_equal(bits(buf[20, 30, :3]), bits(want))


def test_bvh_depth_walk(oracle, scenes):
    assert oracle.bvh_depth(scenes.mesh("teapot")[1]) == 16
    assert oracle.bvh_depth(scenes.mesh("knight")[1]) == 10
    assert oracle.bvh_depth(scenes.mesh("dragon_8k")[1]) == 21      # deeper than the shader's constant 16
    bad = np.zeros((3, 9), F)
    bad[0, 8] = 5
    assert oracle.bvh_depth(bad) == -1
