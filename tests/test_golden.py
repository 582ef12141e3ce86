"""Oracle vs the committed golden fixtures (tests/golden/oracle_golden.npz, made by
tests/golden/make_golden.py).  The fixtures are outputs of the oracle itself - the reference holds
none for its shader - so this guards the restatement against accidental change.  CPU only; the
GPU variant at the bottom checks the HIP path against the same file."""
import os

import numpy as np
import pytest

from conftest import bits

F = np.float32


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_golden.npz"))


def test_rng_golden(oracle, oracle_shared, gold):
    seeds = [int(s) for s in gold["seeds"]]
    got = np.array([oracle.random_value(s) for s in seeds], F)
    assert np.array_equal(bits(got), bits(gold["random_value"]))
    gotn = np.array([oracle_shared.random_normal(s) for s in seeds], F)
    assert np.array_equal(bits(gotn), bits(gold["random_normal_portable"]))      # pure IEEE: bit-exact
    gotd = np.array([oracle_shared.random_direction(s) for s in seeds], F)
    assert np.array_equal(bits(gotd), bits(gold["random_direction_portable"]))
    gotl = np.array([oracle.random_normal(s) for s in seeds], F)                 # libm: allow 2 ulp drift
    np.testing.assert_allclose(gotl[1:], gold["random_normal_libm"][1:], rtol=3e-7, atol=1e-7)


def test_primary_hits_golden(oracle, scenes, gold):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=64, height=36, rays_per_pixel=1, max_bounce_count=4)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    ids, dist = oracle.primary_hits(p, sc, cfg["camera_position"], cfg["camera_rotation"])
    assert np.array_equal(ids, gold["teapot_64x36_prim_id"])
    assert np.array_equal(bits(dist), bits(gold["teapot_64x36_distance"]))
    assert (ids >= 0).mean() > 0.3 and ((ids < 0) & (ids > -4)).any()

    cfg2 = scenes.config(2)
    t8, b8 = scenes.mesh("dragon_8k")
    p3 = oracle.default_params(width=48, height=27, rays_per_pixel=1, max_bounce_count=8)
    ids8, dist8 = oracle.primary_hits(p3, oracle.scene(scenes.spheres_for("white_light"), t8, b8),
                                      cfg2["camera_position"], cfg2["camera_rotation"])
    assert np.array_equal(ids8, gold["dragon8k_48x27_prim_id"])
    assert np.array_equal(bits(dist8), bits(gold["dragon8k_48x27_distance"]))


def test_radiance_golden(oracle, oracle_shared, scenes, gold):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    sph = scenes.spheres_for("cfg1")
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    kw = dict(width=16, height=9, rays_per_pixel=4, max_bounce_count=4)
    img = oracle_shared.render_frames(oracle_shared.default_params(**kw), oracle_shared.scene(sph, tris, bvh), cam, rot, 0, 2)
    assert np.array_equal(bits(img), bits(gold["teapot_16x9_4spp_2frames_portable"]))
    img2 = oracle.render_frames(oracle.default_params(**kw), oracle.scene(sph, tris, bvh), cam, rot, 0, 2)
    np.testing.assert_allclose(img2, gold["teapot_16x9_4spp_2frames_libm"], rtol=1e-5, atol=1e-6)
    # the two math back ends agree to well within the GPU tolerance
    np.testing.assert_allclose(img[..., :3], img2[..., :3], rtol=2e-3, atol=1e-4)


def test_ray_box_golden(oracle, gold):
    got = np.array([oracle.ray_box(r[:3], r[3:], [1, -1, -1, 2, 1, 1]) for r in gold["ray_box_rays"]], F)
    assert np.array_equal(bits(got), bits(gold["ray_box_cases"]))


@pytest.mark.gpu
def test_hip_path_against_golden_file(pkg, scenes, gold):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    sph = scenes.spheres_for("cfg1")
    params = pkg.default_params(width=64, height=36, rays_per_pixel=1, max_bounce_count=4)
    with pkg.Renderer(params) as r:
        r.upload_scene(sph, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        ids, dist = r.primary_hits()
    assert np.array_equal(ids, gold["teapot_64x36_prim_id"])
    assert np.array_equal(bits(dist), bits(gold["teapot_64x36_distance"]))
    params = pkg.default_params(width=16, height=9, rays_per_pixel=4, max_bounce_count=4, flags=pkg.PT_FLAG_EXACT_MATH)
    with pkg.Renderer(params) as r:
        r.upload_scene(sph, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        r.render(2)
        img = r.read_accum(4)
    assert np.array_equal(bits(img[..., :3]), bits(gold["teapot_16x9_4spp_2frames_portable"][..., :3]))
