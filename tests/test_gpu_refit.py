"""pt_refit_scene: moved triangles, same tree.  The resident records after a refit must equal what an upload of
the moved triangles with a refitted node file gives; the refitted node file is made here with numpy by the node
file's own rule (a node's box = tight bounds of the triangles in its range, prepare_data/src/main.rs:16-20,
:108-117; checked on the reference's teapot pair below)."""
import numpy as np
import pytest

from conftest import bits

F = np.float32
pytestmark = pytest.mark.gpu


def refit_nodes(tris, nodes):
    out = nodes.copy()
    v = tris.reshape(-1, 3, 3)
    lo, hi = v.min(1), v.max(1)
    for i, nd in enumerate(nodes):
        first, cnt = int(nd[6]), int(nd[7])
        out[i, 0:3] = lo[first:first + cnt].min(0)
        out[i, 3:6] = hi[first:first + cnt].max(0)
    return out


def deform(tris, amount):
    """moves every vertex by a smooth function of its position, so shared vertices stay shared"""
    v = tris.reshape(-1, 3).astype(np.float64)
    d = np.stack([np.sin(v[:, 1] * 0.11 + 0.3), np.cos(v[:, 2] * 0.07), np.sin(v[:, 0] * 0.05 + v[:, 1] * 0.03)], 1)
    return (v + amount * d).astype(F).reshape(-1, 9)


def render_all(pkg, r, frames):
    ids, dist = r.primary_hits()
    r.reset_accum() if hasattr(r, "reset_accum") else None
    r.render_frames(0, frames)
    return ids, dist, r.read_accum(4).copy(), r.counters()


def test_numpy_refit_of_the_reference_teapot_pair_is_the_identity(scenes):
    tris, bvh = scenes.mesh("teapot")
    assert np.array_equal(bits(refit_nodes(tris, bvh)), bits(bvh))


@pytest.mark.parametrize("mesh,amount", [("teapot", 4.0), ("dragon_80k", 1.5)])
def test_refit_equals_upload_of_the_refitted_file(pkg, oracle_shared, scenes, mesh, amount):
    tris, bvh = scenes.mesh(mesh)
    cfg = scenes.config(1 if mesh == "teapot" else 3)
    spheres = scenes.spheres_for(cfg["spheres"])
    moved = deform(tris, amount)
    assert not np.array_equal(moved, tris)
    bvh_moved = refit_nodes(moved, bvh)
    W, H, frames = 128, 72, 2
    p = pkg.default_params(width=W, height=H, rays_per_pixel=2, max_bounce_count=4,
                           flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_COUNT_WORK)
    with pkg.Renderer(p) as a, pkg.Renderer(p) as b:
        a.upload_scene(spheres, tris, bvh)
        a.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        a.render_frames(0, 1)                      # the context has rendered the old scene before the refit
        ids0, _ = a.primary_hits()
        a.refit_scene(moved)
        a.reset_counters()
        b.upload_scene(spheres, moved, bvh_moved)
        b.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        ia, da, ra, ca = render_all(pkg, a, frames)
        ib, db, rb, cb = render_all(pkg, b, frames)
        assert np.array_equal(ia, ib)
        assert np.array_equal(bits(da), bits(db))
        assert np.array_equal(bits(ra), bits(rb))
        assert ca == cb                              # same box and triangle tests: the same boxes, not just the same image
        assert not np.array_equal(ia, ids0)          # the scene did move
        # back to where it was: the original file's boxes are tight, so this is the original upload again
        a.refit_scene(tris)
        a.reset_counters()
        b.upload_scene(spheres, tris, bvh)
        b.reset_counters()
        ia, da, ra, ca = render_all(pkg, a, frames)
        ib, db, rb, cb = render_all(pkg, b, frames)
        assert np.array_equal(ia, ib)
        assert np.array_equal(bits(da), bits(db))
        assert np.array_equal(bits(ra), bits(rb))
        assert ca == cb
    # and the refitted pair against the CPU oracle
    op = oracle_shared.default_params(width=W, height=H, rays_per_pixel=2, max_bounce_count=4)
    sc = oracle_shared.scene(spheres, moved, bvh_moved)
    oid, odist = oracle_shared.primary_hits(op, sc, cfg["camera_position"], cfg["camera_rotation"])
    with pkg.Renderer(p) as a:
        a.upload_scene(spheres, tris, bvh)
        a.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        a.refit_scene(moved)
        ids, dist = a.primary_hits()
    assert np.array_equal(ids, oid)
    assert np.array_equal(bits(dist), bits(odist))
    assert (ids >= 0).sum() > 200


def test_refit_switches_the_division_form_with_the_bounds(pkg, oracle_shared, scenes):
    """Bounds outside the range of the reciprocal slab test (pt_exact_div.h) after a refit: the plain divider
    takes over and results stay the oracle's."""
    tris, bvh = scenes.mesh("teapot")
    huge = (tris.astype(np.float64) * 1e-30).astype(F)      # bounds around 1e-28 < 2^-56
    bvh_h = refit_nodes(huge, bvh)
    W, H = 64, 36
    p = pkg.default_params(width=W, height=H, rays_per_pixel=1, max_bounce_count=1, flags=pkg.PT_FLAG_EXACT_MATH)
    cam, rot = [0.0, 0.0, 2e-28], [0.0, 0.0, 0.0]
    with pkg.Renderer(p) as a:
        a.upload_scene([], tris, bvh)
        a.set_camera(cam, rot)
        a.refit_scene(huge)
        ids, dist = a.primary_hits()
    op = oracle_shared.default_params(width=W, height=H, rays_per_pixel=1, max_bounce_count=1)
    sc = oracle_shared.scene([], huge, bvh_h)
    oid, odist = oracle_shared.primary_hits(op, sc, cam, rot)
    assert np.array_equal(ids, oid) and np.array_equal(bits(dist), bits(odist))


def test_refit_refusals(pkg, scenes):
    tris, bvh = scenes.mesh("dragon_8k")                    # legacy file: a corrupt link, ranges that disagree
    p = pkg.default_params(width=32, height=18, rays_per_pixel=1)
    with pkg.Renderer(p) as r:
        with pytest.raises(pkg.PtError, match="no scene"):
            r._check(r._L.pt_refit_scene(r._ctx, tris.ctypes.data_as(pkg.C.POINTER(pkg.C.c_float))))
        r.upload_scene([], tris, bvh)
        with pytest.raises(pkg.PtError, match="cannot be refitted"):
            r.refit_scene(tris)
        t2, b2 = scenes.mesh("teapot")
        r.upload_scene([], t2, b2)
        with pytest.raises(ValueError):
            r.refit_scene(t2[:10])
        bad = t2.copy()
        bad[5, 2] = np.inf
        with pytest.raises(pkg.PtError, match="not finite"):
            r.refit_scene(bad)
        r.refit_scene(t2)
    pw = pkg.default_params(width=32, height=18, rays_per_pixel=1, flags=pkg.PT_FLAG_WIDE_BVH)
    with pkg.Renderer(pw) as r:
        r.upload_scene([], t2, b2)
        with pytest.raises(pkg.PtError, match="four-wide"):
            r.refit_scene(t2)


def _state(pkg, r, frames=2):
    ids, dist = r.primary_hits()
    r.reset_counters()
    r.render_frames(0, frames)
    return ids, dist, r.read_accum(4).copy(), r.counters(), r.scene_info()


@pytest.mark.parametrize("mesh,depth", [("dragon_8k", 24), ("teapot", 16), ("teapot", 3), ("small_knight", 0), ("dragon_80k", 32)])
def test_rebuild_on_the_gpu_equals_build_files_and_upload(pkg, scenes, mesh, depth):
    """pt_rebuild_scene(triangles) == pt_upload_scene(pt_bvh_build(triangles)): images, ids, distances, work counters and
    tree statistics, from shuffled triangles; depth 3 gives leaves of more than 127 triangles, depth 0 a single leaf."""
    tris = scenes.load_bin(mesh + ".bin")
    tris = tris[np.random.default_rng(3).permutation(len(tris))]
    cfg = scenes.config({"dragon_8k": 2, "teapot": 1, "small_knight": 1, "dragon_80k": 3}[mesh])
    spheres = scenes.spheres_for(cfg["spheres"])
    p = pkg.default_params(width=128, height=72, rays_per_pixel=2, max_bounce_count=4,
                           flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_COUNT_WORK)
    bt, bn = pkg.bvh_build(tris, depth)
    with pkg.Renderer(p) as a, pkg.Renderer(p) as b:
        a.upload_scene(spheres, *scenes.mesh("knight"))        # some other scene first: its spheres stay
        a.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        ms = a.rebuild_scene(tris, depth)
        assert ms > 0
        b.upload_scene(spheres, bt, bn)
        b.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        sa, sb = _state(pkg, a), _state(pkg, b)
        assert np.array_equal(sa[0], sb[0])
        assert np.array_equal(bits(sa[1]), bits(sb[1]))
        assert np.array_equal(bits(sa[2]), bits(sb[2]))
        assert sa[3] == sb[3]
        assert sa[4] == sb[4]
        if mesh != "small_knight":
            assert (sa[0] >= 0).sum() > 100
        # the rebuilt scene can be refitted like an uploaded one
        moved = deform(bt, 1.0)
        a.refit_scene(moved)
        b.upload_scene(spheres, moved, refit_nodes(moved, bn))
        sa, sb = _state(pkg, a), _state(pkg, b)
        assert np.array_equal(sa[0], sb[0])
        assert np.array_equal(bits(sa[1]), bits(sb[1]))
        assert np.array_equal(bits(sa[2]), bits(sb[2]))
        assert sa[3] == sb[3]


def test_rebuild_into_an_empty_context_and_refusals(pkg, scenes):
    tris = scenes.load_bin("teapot.bin")
    p = pkg.default_params(width=64, height=36, rays_per_pixel=1, flags=pkg.PT_FLAG_EXACT_MATH)
    cfg = scenes.config(1)
    with pkg.Renderer(p) as a, pkg.Renderer(p) as b:
        a.rebuild_scene(tris, 16)                                 # no scene before: no spheres
        a.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        b.upload_scene([], *pkg.bvh_build(tris, 16))
        b.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        ia, da = a.primary_hits()
        ib, db = b.primary_hits()
        assert np.array_equal(ia, ib) and np.array_equal(bits(da), bits(db))
        with pytest.raises(pkg.PtError, match="at least one triangle"):
            a.rebuild_scene(np.zeros((0, 9), F), 16)
        bad = tris.copy()
        bad[0, 0] = np.nan
        with pytest.raises(pkg.PtError, match="not finite"):
            a.rebuild_scene(bad, 16)
        ia2, _ = a.primary_hits()                                 # a refused rebuild leaves the scene as it was
        assert np.array_equal(ia, ia2)
    pd = pkg.default_params(width=64, height=36, rays_per_pixel=1, bvh_max_depth=4)
    with pkg.Renderer(pd) as r:
        with pytest.raises(pkg.PtError, match="levels deep"):
            r.rebuild_scene(tris, 16)
    pw = pkg.default_params(width=64, height=36, rays_per_pixel=1, flags=pkg.PT_FLAG_WIDE_BVH)
    with pkg.Renderer(pw) as r:
        with pytest.raises(pkg.PtError, match="four-wide"):
            r.rebuild_scene(tris, 16)
