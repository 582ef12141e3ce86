"""Edge cases of the scene and parameter space, GPU vs oracle (bit-exact in exact-math mode):
degenerate trees, big leaves, tiny and odd images, many spheres, many samples, awkward geometry."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

NO_SPHERES = np.zeros((0, 12), np.float32)


def _check(pkg, oracle_shared, spheres, tris, bvh, cam, rot, frames=2, **kw):
    """primary hits + radiance + counters of a small render against the oracle"""
    r = pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH, **kw), device=0)
    with r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cam, rot)
        ids, dist = r.primary_hits()
        r.render(frames)
        img = r.read_accum(4)
    op = oracle_shared.default_params(**kw)
    sc = oracle_shared.scene(spheres, tris, bvh)
    oids, odist = oracle_shared.primary_hits(op, sc, cam, rot)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(dist), bits(odist))
    oimg = oracle_shared.render_frames(op, sc, cam, rot, 0, frames)
    assert np.array_equal(bits(img[..., :3]), bits(oimg[..., :3]))
    return ids, img


def _leaf_only(tris):
    """a node file whose root is a leaf holding every triangle (child_index 0)"""
    v = tris.reshape(-1, 3)
    return np.asarray([list(v.min(axis=0)) + list(v.max(axis=0)) + [0, len(tris), 0]], np.float32)


def _wall(n, seed=3):
    """n small triangles facing +z, scattered over a 10 x 10 wall at z in [0, 1)"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-5, 5, (n, 2)).astype(np.float32)
    z = rng.uniform(0, 1, n).astype(np.float32)
    s = rng.uniform(0.3, 1.0, n).astype(np.float32)
    t = np.zeros((n, 9), np.float32)
    t[:, 0], t[:, 1], t[:, 2] = c[:, 0] - s, c[:, 1] - s, z
    t[:, 3], t[:, 4], t[:, 5] = c[:, 0] + s, c[:, 1] - s, z
    t[:, 6], t[:, 7], t[:, 8] = c[:, 0], c[:, 1] + s, z
    return t


def test_single_triangle_root_leaf(pkg, oracle_shared):
    tris = np.asarray([[-3, -2, 0, 3, -2, 0, 0, 3, 0]], np.float32)
    ids, _ = _check(pkg, oracle_shared, NO_SPHERES, tris, _leaf_only(tris), (0, 0, 8), (0, 0, 0),
                    width=64, height=48, rays_per_pixel=2, max_bounce_count=2)
    assert set(np.unique(ids)) == {pkg.MISS_ID, 0}


@pytest.mark.parametrize("n", [127, 128, 300])
def test_root_leaf_with_many_triangles(pkg, oracle_shared, n):
    """Leaves above 126 triangles use the side table for their count (127 is the escape value)."""
    tris = _wall(n)
    ids, _ = _check(pkg, oracle_shared, NO_SPHERES, tris, _leaf_only(tris), (0.5, -0.3, 9), (2, -3, 0),
                    width=80, height=60, rays_per_pixel=1, max_bounce_count=2, frames=1)
    assert len(np.unique(ids)) > 20


def test_built_tree_with_big_leaves(pkg, oracle_shared):
    """The builder stopped early (max_depth 2): four leaves of ~200 triangles each."""
    tris = _wall(800, seed=9)
    out_tr, nodes = pkg.bvh_build(tris, 2)
    assert (nodes[nodes[:, 8] == 0][:, 7] > 127).any()
    _check(pkg, oracle_shared, NO_SPHERES, out_tr, nodes, (0.0, 0.0, 12), (0, 0, 0),
           width=64, height=64, rays_per_pixel=1, max_bounce_count=1, frames=1)


@pytest.mark.parametrize("w,h", [(1, 1), (1, 37), (63, 1), (17, 9), (130, 65)])
def test_tiny_and_odd_image_sizes(pkg, scenes, oracle_shared, w, h):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    _check(pkg, oracle_shared, scenes.spheres_for(cfg["spheres"]), tris, bvh, cfg["camera_position"],
           cfg["camera_rotation"], width=w, height=h, rays_per_pixel=2, max_bounce_count=3)


def test_many_samples_per_pixel_and_many_frames(pkg, scenes, oracle_shared):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    _check(pkg, oracle_shared, scenes.spheres_for(cfg["spheres"]), tris, bvh, cfg["camera_position"],
           cfg["camera_rotation"], width=40, height=24, rays_per_pixel=67, max_bounce_count=5, frames=9)


def test_forty_spheres_with_emitters_and_mirrors(pkg, oracle_shared):
    rng = np.random.default_rng(21)
    sp = np.zeros((40, 12), np.float32)
    sp[:, 0:3] = rng.uniform(-12, 12, (40, 3))
    sp[:, 2] -= 30
    sp[:, 3] = rng.uniform(0.5, 4, 40)
    sp[:, 4:7] = rng.uniform(0.1, 1, (40, 3))
    sp[::5, 7:10] = 1.0
    sp[::5, 10] = rng.uniform(1, 5, 8)
    sp[:, 11] = rng.choice([0.0, 0.3, 1.0], 40)
    tris = np.asarray([[-40, -14, 10, 40, -14, 10, 0, -14, -90]], np.float32)          # a floor, facing up
    ids, img = _check(pkg, oracle_shared, sp, tris, _leaf_only(tris), (0, 0, 5), (0, 0, 0),
                      width=96, height=64, rays_per_pixel=3, max_bounce_count=6)
    on_sphere = (ids < 0) & (ids != pkg.MISS_ID)                       # spheres have ids -1 - index
    assert on_sphere.sum() > 300 and (ids == 0).sum() > 200 and len(np.unique(ids[on_sphere])) > 10


def test_camera_inside_the_mesh_and_axis_aligned_rays(pkg, scenes, oracle_shared):
    """Origin inside the bounding boxes (negative entry distances) and a camera whose centre column
    and row have direction components of exactly zero."""
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    _check(pkg, oracle_shared, NO_SPHERES, tris, bvh, (0.0, 30.0, 0.0), (0, 0, 0),
           width=65, height=33, rays_per_pixel=2, max_bounce_count=4)
    _check(pkg, oracle_shared, NO_SPHERES, tris, bvh, (0.0, 30.0, 0.0), (90, 0, 0),
           width=33, height=33, rays_per_pixel=1, max_bounce_count=2)


def test_degenerate_and_huge_triangles(pkg, oracle_shared):
    """Zero-area triangles, a triangle far outside the f32 range the reciprocal division accepts
    (the scene then runs on the plain divider), duplicated triangles (exact distance ties: the
    lower index wins, as in the shader's strict <)."""
    tris = np.asarray([
        [-3, -2, 0, 3, -2, 0, 0, 3, 0],
        [-3, -2, 0, 3, -2, 0, 0, 3, 0],                 # exact duplicate of triangle 0
        [1, 1, 1, 1, 1, 1, 1, 1, 1],                    # a point
        [-1, 0, 2, 1, 0, 2, 0, 0, 2],                   # a segment
        [-1e30, -1e30, -50, 1e30, -1e30, -50, 0, 1e30, -50],
        [-2, -1, 3, 2, -1, 3, 0, 2, 3],
    ], np.float32)
    out_tr, nodes = pkg.bvh_build(tris, 8)
    ids, _ = _check(pkg, oracle_shared, NO_SPHERES, out_tr, nodes, (0.2, 0.1, 9), (0, 0, 0),
                    width=72, height=54, rays_per_pixel=2, max_bounce_count=3)
    assert len(np.unique(ids)) >= 3


def test_zero_triangles_spheres_only(pkg, oracle_shared):
    """A node file is required, but it may describe an empty leaf."""
    sp = np.asarray([[0, 0, -10, 3, 0.8, 0.3, 0.3, 0, 0, 0, 0, 0.2],
                     [5, 4, -14, 2, 1, 1, 1, 1, 0.9, 0.7, 3, 0]], np.float32)
    bvh = np.asarray([[0, 0, 0, 0, 0, 0, 0, 0, 0]], np.float32)
    tris = np.zeros((0, 9), np.float32)
    ids, _ = _check(pkg, oracle_shared, sp, tris, bvh, (0, 0, 0), (0, 0, 0),
                    width=64, height=40, rays_per_pixel=4, max_bounce_count=4)
    assert set(np.unique(ids)) == {pkg.MISS_ID, -1, -2}
