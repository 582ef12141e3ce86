"""Differential test on random scenes: random triangle clusters (tree from the builder with a
random depth limit), random spheres, camera, image size, sample / bounce / frame counts, material
and sky constants.  GPU vs oracle: ids and distances, radiance (exact-math mode) and work counters,
all bit-exact."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _scene(seed):
    rng = np.random.default_rng(1000 + seed)
    n_clusters = int(rng.integers(1, 6))
    tris = []
    for _ in range(n_clusters):
        centre = rng.uniform(-20, 20, 3)
        spread = rng.uniform(0.5, 12)
        size = rng.uniform(0.05, 4)
        n = int(rng.integers(10, 700))
        a = centre + rng.normal(0, spread, (n, 3))
        tris.append(np.concatenate([a, a + rng.normal(0, size, (n, 3)), a + rng.normal(0, size, (n, 3))], axis=1))
    tris = np.concatenate(tris).astype(np.float32)
    if seed % 5 == 0:                                    # a big ground triangle facing up
        tris = np.concatenate([tris, np.asarray([[-90, -25, 90, 90, -25, 90, 0, -25, -120]], np.float32)])
    n_sph = int(rng.integers(0, 7))
    sp = np.zeros((n_sph, 12), np.float32)
    if n_sph:
        sp[:, 0:3] = rng.uniform(-25, 25, (n_sph, 3))
        sp[:, 3] = rng.uniform(0.5, 8, n_sph)
        sp[:, 4:7] = rng.uniform(0, 1, (n_sph, 3))
        sp[:, 7:10] = rng.uniform(0, 1, (n_sph, 3))
        sp[:, 10] = rng.choice([0.0, 0.0, 2.0, 10.0], n_sph)
        sp[:, 11] = rng.choice([0.0, 0.5, 1.0], n_sph)
    d = rng.normal(0, 1, 3)
    cam = (d / np.linalg.norm(d) * rng.uniform(20, 70)).astype(np.float32)
    # aim roughly at the origin: yaw about y, pitch about x (degrees), plus some roll
    yaw = np.degrees(np.arctan2(cam[0], cam[2]))
    pitch = -np.degrees(np.arctan2(cam[1], np.hypot(cam[0], cam[2])))
    rot = (float(pitch + rng.uniform(-10, 10)), float(yaw + rng.uniform(-10, 10)), float(rng.uniform(-30, 30)))
    kw = dict(width=int(rng.integers(16, 160)), height=int(rng.integers(16, 100)),
              rays_per_pixel=int(rng.integers(1, 6)), max_bounce_count=int(rng.integers(0, 7)),
              fov_degrees=float(rng.uniform(30, 100)),
              tri_color=tuple(rng.uniform(0.2, 1, 3)), tri_emission_color=tuple(rng.uniform(0, 1, 3)),
              tri_emission_strength=float(rng.choice([0.0, 0.0, 0.5])), tri_smoothness=float(rng.choice([0.0, 0.4, 0.9, 1.0])),
              sky_horizon=tuple(rng.uniform(0, 1, 3)), sky_zenith=tuple(rng.uniform(0, 1, 3)), ground=tuple(rng.uniform(0, 1, 3)),
              sun_dir=tuple(rng.normal(0, 1, 3)), sun_intensity=float(rng.uniform(0, 5)), sun_focus=float(rng.uniform(1, 300)),
              use_environment_lighting=int(rng.integers(0, 2)))
    return tris, sp, tuple(map(float, cam)), rot, kw, int(rng.integers(1, 4)), int(rng.integers(1, 24))


@pytest.mark.parametrize("seed", range(24))
def test_random_scene(pkg, oracle_shared, orc_mod, seed):
    tris, sp, cam, rot, kw, frames, depth = _scene(seed)
    tris, nodes = pkg.bvh_build(tris, depth)
    r = pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH | pkg.PT_FLAG_COUNT_WORK, **kw), device=0)
    with r:
        r.upload_scene(sp, tris, nodes)
        r.set_camera(cam, rot)
        ids, dist = r.primary_hits()
        r.reset_counters()
        r.render(frames)
        img = r.read_accum(4)
        got = r.counters()
    op = oracle_shared.default_params(**kw)
    sc = oracle_shared.scene(sp, tris, nodes)
    oids, odist = oracle_shared.primary_hits(op, sc, cam, rot)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(dist), bits(odist))
    c = orc_mod.OrcCounters()
    oimg = oracle_shared.render_frames(op, sc, cam, rot, 0, frames, counters=c)
    assert np.array_equal(bits(img[..., :3]), bits(oimg[..., :3]))
    want = c.as_dict()
    for k in ("ray_casts", "box_tests", "tri_tests", "paths"):
        assert got[k] == want[k], k
