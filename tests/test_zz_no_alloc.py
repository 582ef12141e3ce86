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


def test_static_render_allocates_no_device_memory(pkg, scenes):
    """Free device memory (hipMemGetInfo) is the same before and after the call, on a context whose other buffers were
    set up by an allocating render first; the allocating render of the same size does move it."""
    import torch
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    torch.cuda.init()
    torch.cuda.synchronize()
    params = pkg.default_params(width=320, height=180, rays_per_pixel=4, max_bounce_count=cfg["max_bounce_count"])
    with pkg.Renderer(params, device=0) as warm, pkg.Renderer(params, device=0) as r:
        for x in (warm, r):
            x.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
            x.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        warm.render_frames_static(0, 1)       # first use of the kernels (code objects are loaded lazily)
        warm.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        r.render_frames_static(0, 5)
        r.synchronize()
        free1, _ = torch.cuda.mem_get_info(0)
        assert free1 == free0, "pt_render_frames_static changed free device memory by %d bytes" % (free0 - free1)
        img = r.read_accum(4)
        # the allocating entry point, same context, same render: it takes its sample buffer now
        r.render_frames(0, 5)
        r.synchronize()
        free2, _ = torch.cuda.mem_get_info(0)
        assert free2 < free1
        assert np.array_equal(bits(img), bits(r.read_accum(4)))


def test_overlapping_static_renders_of_two_contexts_are_refused(pkg, scenes):
    """Two contexts on one GPU share the array: the second call, made while the first context's kernels still run,
    gets PT_ESTATE and renders nothing; after a synchronisation it goes through."""
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh(cfg["mesh"])
    params = pkg.default_params(width=320, height=180, rays_per_pixel=4, max_bounce_count=cfg["max_bounce_count"])
    with pkg.Renderer(params, device=0) as a, pkg.Renderer(params, device=0) as b:
        for x in (a, b):
            x.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
            x.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        a.render_frames_static(0, 1)
        a.synchronize()
        a.render_frames_static(0, 3000)       # a few hundred milliseconds of kernels, enqueued asynchronously
        with pytest.raises(pkg.PtError) as e:
            b.render_frames_static(0, 2)
        assert e.value.code == pkg.PT_ESTATE and "still running" in str(e.value)
        a.render_frames_static(3000, 1)       # the owner itself may go on: its stream orders the calls
        assert not b.read_accum(4).any()
        a.synchronize()
        b.render_frames_static(0, 2)
        b.synchronize()
        want = pkg.Renderer(params, device=0)
        try:
            want.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
            want.set_camera(cfg["camera_position"], cfg["camera_rotation"])
            want.render_frames(0, 2)
            assert np.array_equal(bits(b.read_accum(4)), bits(want.read_accum(4)))
        finally:
            want.close()
