"""PT_FLAG_WIDE_BVH: the four-wide tree folded from the node file at upload.

The reference walks a binary tree (shader.wgsl:169-235); the wide walk has no counterpart there, so
its bar is stated against the binary oracle:
  - wherever both report the same primitive the distance is the oracle's, bit for bit;
  - where they differ, the wide result is verified to be a real hit: the oracle's ray_triangle on that
    very triangle confirms it with the same bits; the difference is then an exact-distance tie (same
    distance, other id), a triangle the binary walk lost behind a folded-away box (wide is closer), or,
    never seen so far, the reverse;
  - the counts of each kind are printed (pytest -s) and bounded;
  - radiance agrees with the binary GPU render except on the few pixels whose paths meet such a case.
"""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _ctx(pkg, cfg, tris, bvh, spheres, flags, **over):
    kw = dict(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
              max_bounce_count=cfg["max_bounce_count"], flags=flags)
    kw.update(over)
    r = pkg.Renderer(pkg.default_params(**kw), device=0)
    r.upload_scene(spheres, tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    return r


def _classify(pkg, oracle, op, sc, cfg, tris, ids_w, dist_w, ids_b, dist_b):
    """Returns counts {tie, wide_closer, wide_farther}; raises when a wide hit is not a real hit."""
    same = ids_w == ids_b
    assert np.array_equal(bits(dist_w[same]), bits(dist_b[same]))
    counts = {"tie": 0, "wide_closer": 0, "wide_farther": 0}
    for y, x in zip(*np.nonzero(~same)):
        o, d, _ = oracle.primary_ray(op, cfg["camera_position"], cfg["camera_rotation"], int(x), int(y))
        iw = int(ids_w[y, x])
        if iw >= 0:         # a triangle: the oracle's ray_triangle must confirm it with the same distance
            h = oracle.ray_triangle(op, o, d, tris[iw])
            assert h.did_hit and np.float32(h.distance).view(np.uint32) == dist_w[y, x].view(np.uint32), (x, y, iw)
        else:               # a sphere or a miss can only replace a triangle the binary walk found
            assert int(ids_b[y, x]) >= 0
        if dist_w[y, x] == dist_b[y, x]:
            counts["tie"] += 1
        elif dist_w[y, x] < dist_b[y, x]:
            counts["wide_closer"] += 1
        else:
            counts["wide_farther"] += 1
    return counts


@pytest.mark.parametrize("n,w,h", [(1, 640, 360), (2, 960, 540), (3, 960, 540), (4, 960, 540), (5, 960, 540)])
def test_wide_primary_hits_against_the_binary_oracle(pkg, scenes, oracle, n, w, h):
    cfg = scenes.config(n, width=w, height=h, rays_per_pixel=1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    with _ctx(pkg, cfg, tris, bvh, spheres, pkg.PT_FLAG_WIDE_BVH) as r:
        ids_w, dist_w = r.primary_hits()
    op = oracle.default_params(width=w, height=h, rays_per_pixel=1, max_bounce_count=cfg["max_bounce_count"])
    sc = oracle.scene(spheres, tris, bvh)
    ids_b, dist_b = oracle.primary_hits(op, sc, cfg["camera_position"], cfg["camera_rotation"])
    assert (ids_w != pkg.MISS_ID).mean() > 0.05
    c = _classify(pkg, oracle, op, sc, cfg, tris, ids_w, dist_w, ids_b, dist_b)
    n_diff = int((ids_w != ids_b).sum())
    print("config %d (%dx%d): %d of %d pixels differ from the binary walk: %s" % (n, w, h, n_diff, w * h, c))
    assert n_diff <= 2e-4 * w * h
    assert c["wide_farther"] == 0


@pytest.mark.parametrize("n,w,h,rpp,frames", [(1, 320, 180, 4, 2), (2, 320, 180, 4, 2), (4, 480, 270, 4, 2), (5, 320, 180, 4, 2)])
def test_wide_radiance_matches_the_binary_render(pkg, scenes, n, w, h, rpp, frames):
    """Exact-math mode on both sides, so every difference comes from the traversal: at least 99.5 % of the pixels
    are equal bit for bit and the image means agree to 1e-3."""
    cfg = scenes.config(n, width=w, height=h, rays_per_pixel=rpp)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    imgs = []
    for flags in (pkg.PT_FLAG_EXACT_MATH, pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_WIDE_BVH):
        with _ctx(pkg, cfg, tris, bvh, spheres, flags) as r:
            r.render(frames)
            imgs.append(r.read_accum(4)[..., :3])
    a, b = imgs
    assert np.isfinite(b).all() and (b > 0).any()
    equal = (bits(a) == bits(b)).all(axis=-1).mean()
    print("config %d: %.5f of the pixels bit-equal, image means %.6g / %.6g" % (n, equal, a.mean(), b.mean()))
    assert equal > 0.995
    assert abs(a.mean() - b.mean()) <= 1e-3 * a.mean()


def _leaf_only(tris):
    v = tris.reshape(-1, 3)
    return np.asarray([list(v.min(axis=0)) + list(v.max(axis=0)) + [0, len(tris), 0]], np.float32)


def test_wide_degenerate_trees(pkg, oracle_shared):
    """Root leaf (nothing to fold), a root with two leaf children, and a built tree with odd child counts:
    no ties in these scenes, so ids, distances and radiance are the binary oracle's bit for bit."""
    rng = np.random.default_rng(5)
    n = 23
    c = rng.uniform(-4, 4, (n, 2)).astype(np.float32)
    z = (np.arange(n) * 0.37).astype(np.float32)            # distinct depths: no distance ties
    s = rng.uniform(0.5, 1.5, n).astype(np.float32)
    t = np.zeros((n, 9), np.float32)
    t[:, 0], t[:, 1], t[:, 2] = c[:, 0] - s, c[:, 1] - s, z
    t[:, 3], t[:, 4], t[:, 5] = c[:, 0] + s, c[:, 1] - s, z
    t[:, 6], t[:, 7], t[:, 8] = c[:, 0], c[:, 1] + s, z
    cam, rot = (0.3, 0.1, 14.0), (0, 0, 0)
    kw = dict(width=96, height=72, rays_per_pixel=2, max_bounce_count=3)
    scenes_ = [(t[:1], _leaf_only(t[:1]))]
    for depth in (1, 2, 5):
        scenes_.append(pkg.bvh_build(t, depth))
    for tris, bvh in scenes_:
        with pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_WIDE_BVH, **kw), device=0) as r:
            r.upload_scene(np.zeros((0, 12), np.float32), tris, bvh)
            r.set_camera(cam, rot)
            ids, dist = r.primary_hits()
            r.render(2)
            img = r.read_accum(4)
        op = oracle_shared.default_params(**kw)
        sc = oracle_shared.scene(np.zeros((0, 12), np.float32), tris, bvh)
        oids, odist = oracle_shared.primary_hits(op, sc, cam, rot)
        oimg = oracle_shared.render_frames(op, sc, cam, rot, 0, 2)
        assert (ids >= 0).any()
        assert np.array_equal(ids, oids) and np.array_equal(bits(dist), bits(odist))
        assert np.array_equal(bits(img[..., :3]), bits(oimg[..., :3]))


def test_wide_flag_limits(pkg):
    """Not combinable with the work counters; a tree whose folded form would need more than 64 stack entries is refused."""
    with pytest.raises(pkg.PtError):
        pkg.Renderer(pkg.default_params(width=16, height=16, flags=pkg.PT_FLAG_WIDE_BVH | pkg.PT_FLAG_COUNT_WORK), device=0)
    from test_gpu_parity import _comb_scene
    tris, bvh = _comb_scene(64)                  # 63 binary levels -> 32 wide levels -> 96 entries
    with pkg.Renderer(pkg.default_params(width=16, height=16, flags=pkg.PT_FLAG_WIDE_BVH), device=0) as r:
        with pytest.raises(pkg.PtError) as e:
            r.upload_scene(np.zeros((0, 12), np.float32), tris, bvh)
        assert "four-wide" in str(e.value)
    tris, bvh = _comb_scene(40)                  # 39 levels -> 20 wide levels -> 60 entries: the overflow area is used
    with pkg.Renderer(pkg.default_params(width=64, height=48, rays_per_pixel=1, max_bounce_count=2,
                                         flags=pkg.PT_FLAG_WIDE_BVH | pkg.PT_FLAG_EXACT_MATH), device=0) as r:
        r.upload_scene(np.zeros((0, 12), np.float32), tris, bvh)
        r.set_camera((0.3, 0.2, 32.0), (1.5, -2.0, 0.0))
        ids, _ = r.primary_hits()
        r.render(1)
        assert np.isfinite(r.read_accum(4)).all() and len(np.unique(ids)) >= 3
