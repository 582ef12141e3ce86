"""pt_render_frames_static, the entry point that allocates nothing (its per-batch working state lives
in a static device array of the library), against pt_render_frames through the C ABI: bit for bit,
default (hardware transcendental) math, on two calls in a row of which the second is smaller, so
that stale contents of the shared array would show."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _both(pkg, scenes, config, width, height, rpp, frames, first=0):
    cfg = scenes.config(config)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    out = []
    for static in (False, True):
        params = pkg.default_params(width=width, height=height, rays_per_pixel=rpp,
                                    max_bounce_count=cfg["max_bounce_count"])
        with pkg.Renderer(params, device=0) as r:
            r.upload_scene(spheres, tris, bvh)
            r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
            if first:       # continue a running mean: both start from the same non-zero buffer
                seed_img = np.full((height, width, 4), 0.25, np.float32)
                r.write_accum(seed_img)
            if static:
                r.render_frames_static(first, frames)
            else:
                r.render_frames(first, frames)
            out.append(r.read_accum(4))
    return out


def test_static_scratch_render_matches_allocating_render_twice(pkg, scenes):
    L = pkg.load_library()
    assert L.pt_static_scratch_bytes() == 64 << 20
    # first call: several frames per batch, most of the array in use (cfg 1: teapot + 3 spheres)
    a, b = _both(pkg, scenes, 1, 320, 180, 4, 5)
    assert np.isfinite(a).all() and (a[..., :3] > 0).any()
    assert np.array_equal(bits(a), bits(b))
    # second call: smaller image, fewer samples, other scene (cfg 2: dragon_8k, depth-21 tree), frames 3..4
    a, b = _both(pkg, scenes, 2, 96, 54, 2, 2, first=3)
    assert np.array_equal(bits(a), bits(b))
    # third: a one-pixel image right after (everything but a few bytes of the array is stale)
    a, b = _both(pkg, scenes, 1, 1, 1, 1, 1)
    assert np.array_equal(bits(a), bits(b))


def test_static_scratch_rejects_what_does_not_fit(pkg, scenes):
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    params = pkg.default_params(width=1200, height=600, rays_per_pixel=20, max_bounce_count=2)   # 230 MB of samples per frame
    with pkg.Renderer(params, device=0) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        with pytest.raises(pkg.PtError) as e:
            r.render_frames_static(0, 1)
        assert e.value.code == -1 and "static array holds 67108864" in str(e.value)
        before = r.read_accum(4)
        assert not before.any()          # nothing was rendered
