"""The path kernel's alternative schedules, which the library keeps behind environment switches read by pt_create:
PT_QUAD_FETCH=1 (the four lanes of a quad fetch each other's inner records through LDS) and PT_TWO_PATHS=1 (two paths
per lane, one walked at a time, state in a per-lane buffer pair).  Neither is faster than the default (see
profiles/r03_block_packing.txt); both must still give the oracle's bits: radiance, work counters, deep trees whose
stacks overflow LDS, the wrapped-seed frame, the default math mode against the default schedule."""
import numpy as np
import pytest

from conftest import bits
from test_gpu_parity import _comb_scene, _setup

pytestmark = pytest.mark.gpu

SWITCHES = [{"PT_QUAD_FETCH": "1"}, {"PT_TWO_PATHS": "1"}, {"PT_TWO_PATHS": "1", "PT_QUAD_FETCH": "1"},
            {"PT_TWO_PATHS": "1", "PT_QUAD_FETCH": "1", "PT_REFILL_MIN": "40", "PT_STARVE_MIN": "4"}]


def _env(monkeypatch, env):
    for k in ("PT_QUAD_FETCH", "PT_TWO_PATHS", "PT_REFILL_MIN", "PT_STARVE_MIN"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: "+".join(sorted(e)))
@pytest.mark.parametrize("n,w,h,rpp,frames,first", [(1, 160, 90, 3, 2, 0), (2, 128, 72, 4, 2, 13), (5, 128, 72, 2, 3, 0), (4, 96, 54, 2, 2, 0)])
def test_schedules_render_the_oracles_bits(pkg, scenes, oracle_shared, orc_mod, monkeypatch, env, n, w, h, rpp, frames, first):
    _env(monkeypatch, env)
    cfg, r, op, sc = _setup(pkg, scenes, oracle_shared, n, flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_COUNT_WORK,
                            width=w, height=h, rays_per_pixel=rpp)
    with r:
        seed_img = np.full((h, w, 4), 0.125, np.float32)
        if first:
            r.write_accum(seed_img)
        r.render_frames(first, frames)
        img = r.read_accum(4)
        got = r.counters()
    c = orc_mod.OrcCounters()
    oimg = oracle_shared.render_frames(op, sc, cfg["camera_position"], cfg["camera_rotation"], first, frames,
                                       accum=seed_img.copy() if first else None, counters=c)
    assert np.array_equal(bits(img[..., :3]), bits(oimg[..., :3]))
    want = c.as_dict()
    for k in ("ray_casts", "box_tests", "tri_tests", "paths"):
        assert got[k] == want[k], k


@pytest.mark.parametrize("env", SWITCHES[:3], ids=lambda e: "+".join(sorted(e)))
def test_schedules_on_a_tree_deeper_than_the_lds_stack(pkg, oracle_shared, orc_mod, monkeypatch, env):
    _env(monkeypatch, env)
    n = 64
    tris, bvh = _comb_scene(n)
    cam, rot = (0.3, 0.2, 0.5 * n + 12.0), (1.5, -2.0, 0.0)
    spheres = np.asarray([[0.0, 0.0, 0.5 * n + 80.0, 50.0, 0.9, 0.8, 0.7, 0.0, 0.0, 0.0, 0.0, 0.95]], np.float32)
    kw = dict(width=96, height=54, rays_per_pixel=2, max_bounce_count=3)
    with pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_COUNT_WORK, **kw), device=0) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cam, rot)
        r.render(2)
        img = r.read_accum(4)
        got = r.counters()
    op = oracle_shared.default_params(**kw)
    sc = oracle_shared.scene(spheres, tris, bvh)
    c = orc_mod.OrcCounters()
    oimg = oracle_shared.render_frames(op, sc, cam, rot, 0, 2, counters=c)
    assert c.max_stack >= n - 1
    assert np.array_equal(bits(img[..., :3]), bits(oimg[..., :3]))
    for k in ("ray_casts", "box_tests", "tri_tests", "paths"):
        assert got[k] == c.as_dict()[k], k


def test_schedules_agree_with_the_default_one_in_default_math(pkg, scenes, oracle, monkeypatch):
    """Hardware transcendentals, full benchmark mesh at a quarter of its resolution: a path's arithmetic does not
    depend on when its lane runs it, so every schedule gives the same bits as the default one."""
    imgs = []
    for env in [{}] + SWITCHES[:3]:
        _env(monkeypatch, env)
        cfg, r, op, sc = _setup(pkg, scenes, oracle, 4, width=480, height=270, rays_per_pixel=4)
        with r:
            r.render(3)
            imgs.append(r.read_accum(4))
    assert imgs[0][..., :3].any()
    for img in imgs[1:]:
        assert np.array_equal(bits(img), bits(imgs[0]))
