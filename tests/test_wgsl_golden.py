"""The oracle (and, on a GPU, the HIP kernels) against what the reference shader itself returns.

tests/golden/wgsl_golden.npz holds outputs of the reference's shader.wgsl, produced by executing its source text with
the interpreter in tests/golden/wgsl_interp.py (tests/golden/make_wgsl_golden.py; only numbers are stored).  Every
comparison here is bit for bit, with the portable log / cos / pow on both sides.  Where /root/reference exists part of
the file is computed again from the shader's text and must reproduce the committed values.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.environ.get("PT_WGSL_GOLDEN", os.path.join(ROOT, "tests", "golden", "wgsl_golden.npz"))
SHADER = "/root/reference/src/shaders/shader.wgsl"


@pytest.fixture(scope="module")
def G():
    return dict(np.load(GOLDEN))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(a, b):
    """bit-equal, any NaN equal to any NaN (a NaN's payload is not the shader's to define)"""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def hit_row(h):
    return np.array([1.0 if h.did_hit else 0.0, h.distance] + list(h.position) + list(h.normal) + list(h.color) +
                    list(h.emission_color) + [h.emission_strength, h.smoothness], dtype=np.float32)


def check_hits(got_rows, want_rows, what):
    """did_hit everywhere; the other fields where the shader's did_hit is set (a miss leaves them undefined for the caller)"""
    got_rows, want_rows = np.stack(got_rows), np.asarray(want_rows)
    assert np.array_equal(got_rows[:, 0], want_rows[:, 0]), what + ": did_hit"
    hit = want_rows[:, 0] == 1.0
    assert hit.any() and (~hit).any(), what + ": the vectors must hold hits and misses"
    assert same(got_rows[hit], want_rows[hit]), what


def test_random_functions(G, oracle_shared):
    o = oracle_shared
    seeds = [int(s) for s in G["rng_seeds"]]
    assert same([o.random_value(s) for s in seeds], G["rng_value"])
    assert same([o.random_normal(s) for s in seeds], G["rng_normal"])
    assert same(np.stack([o.random_direction(s) for s in seeds]), G["rng_direction"])
    n = G["rng_hemisphere_normal"]
    assert same(np.stack([o.random_hemisphere_direction(n, s) for s in seeds]), G["rng_hemisphere"])
    assert np.isnan(G["rng_direction"][0]).all()          # seed 0: log(0) * cos(0) -> the shader's NaN direction


def test_rotate_vector_and_environment_light(G, oracle_shared):
    o = oracle_shared
    assert same(np.stack([o.rotate_vector(v, a) for v, a in zip(G["rot_vectors"], G["rot_angles"])]), G["rot_result"])
    p = o.default_params()
    assert same(np.stack([o.environment_light(p, d) for d in G["env_dirs"]]), G["env_light"])


def test_ray_box(G, oracle_shared):
    got = [oracle_shared.ray_box(o, d, b) for o, d, b in zip(G["box_origins"], G["box_dirs"], G["box_bounds"])]
    assert same(got, G["box_result"])
    assert (G["box_result"] == np.float32(999999999.0)).any() and (G["box_result"] < 1e6).any()


def test_ray_triangle_and_ray_sphere(G, oracle_shared):
    o = oracle_shared
    p = o.default_params()
    check_hits([hit_row(o.ray_triangle(p, a, d, t)) for a, d, t in zip(G["tri_origins"], G["tri_dirs"], G["tri_vertices"])],
               G["tri_hits"], "ray_triangle")
    check_hits([hit_row(o.ray_sphere(a, d, s)) for a, d, s in zip(G["sph_origins"], G["sph_dirs"], G["sph_records"])],
               G["sph_hits"], "ray_sphere")


def _scene(G, o, scenes):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    w, h, rpp, bounces, frames, depth = [int(x) for x in G["scene_width_height_rpp_bounces_frames_depth"]]
    p = o.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=bounces)
    cam = G["scene_camera"]
    assert np.array_equal(cam, np.array(list(cfg["camera_position"]) + list(cfg["camera_rotation"]), dtype=np.float32))
    return p, o.scene(spheres, tris, bvh), cam[:3], cam[3:], (w, h, rpp, bounces, frames), (spheres, tris, bvh)


def test_calculate_ray_collision_and_trace(G, oracle_shared, scenes):
    o = oracle_shared
    p, sc, _, _, _, _ = _scene(G, o, scenes)
    check_hits([hit_row(o.closest_hit(p, sc, a, d)) for a, d in zip(G["cast_origins"], G["cast_dirs"])], G["cast_hits"],
               "calculate_ray_collision")
    got = [o.trace(p, sc, a, d, int(s), int(f)) for a, d, s, f in
           zip(G["cast_origins"], G["cast_dirs"], G["trace_seeds"], G["trace_frame_counts"])]
    assert same(np.stack(got), G["trace_result"])


def test_whole_frames(G, oracle_shared, scenes):
    """fs_main over every pixel, draw after draw: the accumulation buffer after each frame"""
    o = oracle_shared
    p, sc, cam, rot, (w, h, rpp, bounces, frames), _ = _scene(G, o, scenes)
    acc = np.zeros((h, w, 4), dtype=np.float32)
    for f in range(frames):
        o.render_frames(p, sc, cam, rot, f, 1, accum=acc)
        assert same(acc[..., :3], G["frames_accum"][f]), "frame %d" % f
    assert (G["frames_accum"][-1] > 0).any()


@pytest.mark.skipif(not os.path.exists(SHADER), reason="the reference is not on this machine")
def test_the_committed_vectors_come_from_the_reference_shader(G, tmp_path):
    """Runs the reference's shader text through the interpreter again on inputs of the committed file and finds the
    committed outputs: the file is what the generator says it is."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_wgsl_golden as mk
    import wgsl_interp as W
    src = open(SHADER).read()
    shim = mk.build_shim()
    I = mk.make_interpreter(src, shim)
    seeds = G["rng_seeds"][:11]
    assert same([I.call("random_value", [W.U32(int(s))]) for s in seeds], G["rng_value"][:11])
    assert same(np.stack([mk.fvec(I.call("random_direction", [W.U32(int(s))])) for s in seeds]), G["rng_direction"][:11])
    k = slice(0, 40)
    got = [I.call("ray_box", [mk.ray_struct(I, o, d), W.Arr(W.F32(x) for x in b)])
           for o, d, b in zip(G["box_origins"][k], G["box_dirs"][k], G["box_bounds"][k])]
    assert same(got, G["box_result"][k])
    got = [mk.fvec(I.call("get_environment_light", [mk.ray_struct(I, (0, 0, 0), d)])) for d in G["env_dirs"][:8]]
    assert same(np.stack(got), G["env_light"][:8])
    # and the libm flavour of the interpreter against the libm oracle: the choice of log / cos / pow is the only difference
    Il = W.Interpreter(src, W.Builtins(W.LibmMath()))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    ol = entry.load_oracle_module().Oracle(shared_math=False)
    assert same([Il.call("random_normal", [W.U32(int(s))]) for s in seeds], [ol.random_normal(int(s)) for s in seeds])


@pytest.mark.gpu
def test_gpu_frames_equal_the_shader(G, pkg, scenes):
    """the HIP kernels in exact-math mode: frame after frame the same bits as fs_main"""
    w, h, rpp, bounces, frames, depth = [int(x) for x in G["scene_width_height_rpp_bounces_frames_depth"]]
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    p = pkg.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=bounces, flags=pkg.PT_FLAG_EXACT_MATH)
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        for f in range(frames):
            r.render_frames(f, 1)
            assert same(r.read_accum(4)[..., :3], G["frames_accum"][f]), "frame %d" % f
        r.render_frames(0, frames)                      # and as one batch
        assert same(r.read_accum(4)[..., :3], G["frames_accum"][frames - 1])


@pytest.mark.gpu
def test_gpu_functions_equal_the_shader(G, pkg, scenes):
    """the device functions one by one (pt_probe_functions, pt_probe_intersections) against the shader's values"""
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    p = pkg.default_params(width=64, height=36, rays_per_pixel=1, max_bounce_count=1, flags=pkg.PT_FLAG_EXACT_MATH)
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        seeds = G["rng_seeds"]
        f = r.probe_functions(seeds, np.tile(np.float32([0, 1, 0]), (len(seeds), 1)), G["rng_hemisphere_normal"])
        assert same(f["random_value"], G["rng_value"]) and same(f["random_normal"], G["rng_normal"])
        assert same(f["random_direction"], G["rng_direction"])
        assert same(f["random_hemisphere_direction"], G["rng_hemisphere"])
        dirs = G["env_dirs"]
        f = r.probe_functions(np.ones(len(dirs), np.uint32), dirs, [0, 1, 0])
        assert same(f["environment_light"], G["env_light"])
        for v, a, want in zip(G["rot_vectors"], G["rot_angles"], G["rot_result"]):
            r.set_camera([0, 0, 0], a)
            assert same(r.probe_functions([1], [v], [0, 1, 0])["rotate_vector"][0], want)
        n = len(G["box_result"])
        rays = np.concatenate([G["box_origins"], G["box_dirs"]], axis=1)
        x = r.probe_intersections(rays, G["box_bounds"], G["tri_vertices"][:n], G["sph_records"][:n, :4])
        assert same(x["box"], G["box_result"])
        ok = ~np.isnan(x["box_reciprocal"])             # rays and boxes that qualify for the reciprocal form
        assert ok.sum() > n // 2 and same(x["box_reciprocal"][ok], G["box_result"][ok])
        rays = np.concatenate([G["tri_origins"], G["tri_dirs"]], axis=1)
        x = r.probe_intersections(rays, G["box_bounds"], G["tri_vertices"], G["sph_records"][:, :4])
        hit = G["tri_hits"][:, 0] == 1.0
        assert np.array_equal(x["tri_hit"], hit) and same(x["tri_distance"][hit], G["tri_hits"][hit, 1])
        rays = np.concatenate([G["sph_origins"], G["sph_dirs"]], axis=1)
        x = r.probe_intersections(rays, G["box_bounds"], G["tri_vertices"], G["sph_records"][:, :4])
        hit = G["sph_hits"][:, 0] == 1.0
        assert np.array_equal(x["sphere_hit"], hit) and same(x["sphere_distance"][hit], G["sph_hits"][hit, 1])


# ---- the other benchmark scenes: tests/golden/wgsl_golden_scenes.npz (make_wgsl_golden_scenes.py) ----
SCENES_GOLDEN = os.path.join(ROOT, "tests", "golden", "wgsl_golden_scenes.npz")


@pytest.fixture(scope="module")
def GS():
    return dict(np.load(SCENES_GOLDEN))


def _scene_case(GS, n, scenes):
    cfg = scenes.config(n)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    w, h, rpp, bounces = [int(x) for x in GS["cfg%d_size_rpp_bounces" % n]]
    assert bounces == cfg["max_bounce_count"]
    return cfg, tris, bvh, spheres, w, h, rpp, bounces


@pytest.mark.parametrize("n", [2, 3, 4, 5])
def test_whole_frames_of_the_other_scenes(GS, oracle_shared, scenes, n):
    """fs_main of the reference shader over every pixel for frame_count 0, 1 and 13 (wrapped seed term) on the legacy
    dragon_8k file (21 levels, a corrupt link), the 87 k-triangle dragon, the 784 k-triangle benchmark mesh and the knight with 16
    spheres and 12 bounces"""
    o = oracle_shared
    cfg, tris, bvh, spheres, w, h, rpp, bounces = _scene_case(GS, n, scenes)
    p = o.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=bounces)
    sc = o.scene(spheres, tris, bvh)
    acc = np.zeros((h, w, 4), dtype=np.float32)
    want = GS["cfg%d_frames_accum" % n]
    for k, fc in enumerate(GS["frame_counts"]):
        o.render_frames(p, sc, cfg["camera_position"], cfg["camera_rotation"], int(fc), 1, accum=acc)
        assert same(acc[..., :3], want[k]), "configuration %d, frame_count %d" % (n, fc)
    assert (want[-1] > 0).any() and len(np.unique(want[-1])) > 20


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 4, 5])
def test_gpu_frames_of_the_other_scenes_equal_the_shader(GS, pkg, scenes, n):
    cfg, tris, bvh, spheres, w, h, rpp, bounces = _scene_case(GS, n, scenes)
    p = pkg.default_params(width=w, height=h, rays_per_pixel=rpp, max_bounce_count=bounces, flags=pkg.PT_FLAG_EXACT_MATH)
    want = GS["cfg%d_frames_accum" % n]
    with pkg.Renderer(p) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        for k, fc in enumerate(GS["frame_counts"]):
            r.render_frames(int(fc), 1)
            assert same(r.read_accum(4)[..., :3], want[k]), "configuration %d, frame_count %d" % (n, fc)
