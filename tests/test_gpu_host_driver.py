"""Headless host side on the GPU: the camera-motion protocol of the reference's timer loop
(src/main.rs:674-676, 45-115, 444-523) through Renderer.update_camera, and the C++ driver
pt_render (image output with the sRGB transfer the reference's surface applies)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PT_RENDER = os.path.join(ROOT, "path-tracing_amd", "pt_render")
OBJ = os.path.join(ROOT, "assets", "objects")


def _srgb8(img):
    v = np.clip(np.nan_to_num(img, nan=0.0), 0.0, 1.0).astype(np.float32)
    e = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, np.float32(1.0 / 2.4)) - 0.055)
    return np.rint(e * 255.0).astype(np.int32)


def _read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255"
    w, h = (int(x) for x in parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3).astype(np.int32)


def test_camera_keys_restart_the_accumulation(pkg, scenes, oracle_shared):
    """Ticks without keys accumulate (frame_count 0,1,2); a tick with a key held moves the camera,
    restarts at frame_count 0 and the next frame overwrites the buffer.  The final buffer equals
    the oracle's frame 0 at the pose the host arithmetic arrived at."""
    cfg = scenes.config(1, width=96, height=54, rays_per_pixel=2)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    kw = dict(width=96, height=54, rays_per_pixel=2, max_bounce_count=cfg["max_bounce_count"])
    r = pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH, **kw), device=0)
    with r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        for _ in range(3):
            assert r.update_camera([]) is False
            r.render()
        assert r.frame_count == 3
        still = r.read_accum(4)
        for keys in (["W", "L"], ["W", "L"], ["D"]):
            assert r.update_camera(keys) is True
            assert r.frame_count == 0
            r.render()
            assert r.frame_count == 1
        moved = r.read_accum(4)
        pos, rot = r.camera_position, r.camera_rotation
    op = oracle_shared.default_params(**kw)
    sc = oracle_shared.scene(spheres, tris, bvh)
    want_still = oracle_shared.render_frames(op, sc, cfg["camera_position"], cfg["camera_rotation"], 0, 3)
    assert np.array_equal(bits(still[..., :3]), bits(want_still[..., :3]))
    assert pos != list(map(float, cfg["camera_position"])) and rot[1] != float(cfg["camera_rotation"][1])
    want_moved = oracle_shared.render_frames(op, sc, pos, rot, 0, 1)
    assert np.array_equal(bits(moved[..., :3]), bits(want_moved[..., :3]))
    assert not np.array_equal(bits(moved), bits(still))


def test_pt_render_writes_the_srgb_image(pkg, oracle_shared, tmp_path):
    """pt_render (C++ over the C ABI) against the oracle's linear image pushed through the sRGB
    transfer function: 8-bit levels equal, up to 1 level where rounding of the power function
    may differ between libms."""
    assert os.path.exists(PT_RENDER), "build pt_render with make -C path-tracing_amd"
    out = tmp_path / "t.ppm"
    sphere = "40,60,-20,15,0.9,0.2,0.2,0,0,0,0,0.5"
    cmd = [PT_RENDER, "--tris", os.path.join(OBJ, "teapot.bin"), "--bvh", os.path.join(OBJ, "teapot_bvh.bin"),
           "--sphere", sphere, "--cam", "5,35,150", "--rot", "0,0,0", "--size", "160x90", "--rpp", "3",
           "--frames", "2", "--bounces", "4", "--exact", "--out", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = _read_ppm(out)
    tris = np.fromfile(os.path.join(OBJ, "teapot.bin"), "<f4").reshape(-1, 9)
    bvh = np.fromfile(os.path.join(OBJ, "teapot_bvh.bin"), "<f4").reshape(-1, 9)
    spheres = np.asarray([[float(x) for x in sphere.split(",")]], np.float32)
    op = oracle_shared.default_params(width=160, height=90, rays_per_pixel=3, max_bounce_count=4)
    sc = oracle_shared.scene(spheres, tris, bvh)
    lin = oracle_shared.render_frames(op, sc, (5, 35, 150), (0, 0, 0), 0, 2)[..., :3]
    want = _srgb8(lin)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1
    assert (got != want).mean() < 0.01
    assert got.std() > 10          # an actual picture


def test_pt_render_walk_matches_the_renderer(pkg, oracle_shared, tmp_path):
    """--walk plays update_camera + render per tick; the final pose and image equal what the
    Python mirror of State produces with the same key script."""
    out = tmp_path / "w.ppm"
    cmd = [PT_RENDER, "--tris", os.path.join(OBJ, "teapot.bin"), "--bvh", os.path.join(OBJ, "teapot_bvh.bin"),
           "--cam", "5,35,150", "--rot", "0,0,0", "--size", "128x72", "--rpp", "2", "--bounces", "3", "--exact",
           "--walk", "W*3,AJ*4,^*1,-*3", "--out", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = _read_ppm(out)
    tris = np.fromfile(os.path.join(OBJ, "teapot.bin"), "<f4").reshape(-1, 9)
    bvh = np.fromfile(os.path.join(OBJ, "teapot_bvh.bin"), "<f4").reshape(-1, 9)
    r = pkg.Renderer(pkg.default_params(width=128, height=72, rays_per_pixel=2, max_bounce_count=3,
                                        flags=pkg.PT_FLAG_EXACT_MATH), device=0)
    with r:
        r.upload_scene(np.zeros((0, 12), np.float32), tris, bvh)
        r.set_camera((5, 35, 150), (0, 0, 0))
        for keys, n in ((["W"], 3), (["A", "J"], 4), (["Space"], 1), ([], 3)):
            for _ in range(n):
                r.update_camera(keys)
                r.render()
        assert r.frame_count == 4          # the Space tick restarted at 0, then three still ticks
        img = r.read_accum(3)
        pos, rot = r.camera_position, r.camera_rotation
    assert "4 accumulated frame(s)" in res.stdout
    line = [l for l in res.stdout.splitlines() if l.startswith("camera after the walk")][0]
    nums = line.split("position ")[1].split(" rotation ")
    assert [np.float32(x) for x in nums[0].split(",")] == [np.float32(x) for x in pos]
    assert [np.float32(x) for x in nums[1].split(",")[:3:1][0:2]] == [np.float32(x) for x in rot[:2]]
    want = _srgb8(img)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01


def test_two_contexts_interleaved_and_from_two_threads(pkg, scenes, oracle_shared):
    """Contexts are independent: two of them with different scenes and sizes, driven alternately
    from one thread and then each from its own thread, give the oracle's images."""
    import threading
    setups = []
    for n, (w, h) in ((1, (96, 54)), (2, (80, 60))):
        cfg = scenes.config(n, width=w, height=h, rays_per_pixel=2)
        tris, bvh = scenes.mesh(cfg["mesh"])
        sp = scenes.spheres_for(cfg["spheres"])
        kw = dict(width=w, height=h, rays_per_pixel=2, max_bounce_count=cfg["max_bounce_count"])
        r = pkg.Renderer(pkg.default_params(flags=pkg.PT_FLAG_EXACT_MATH, **kw), device=0)
        r.upload_scene(sp, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        op = oracle_shared.default_params(**kw)
        want = oracle_shared.render_frames(op, oracle_shared.scene(sp, tris, bvh), cfg["camera_position"],
                                           cfg["camera_rotation"], 0, 4)
        setups.append((r, want))
    try:
        for _ in range(4):                        # alternate: one frame each
            for r, _w in setups:
                r.render(1)
        for r, want in setups:
            assert np.array_equal(bits(r.read_accum(4)[..., :3]), bits(want[..., :3]))

        def work(r):
            r.set_camera(r.camera_position, r.camera_rotation)      # restart at frame 0
            for _ in range(4):
                r.render(1)
            r.synchronize()
        threads = [threading.Thread(target=work, args=(r,)) for r, _w in setups]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for r, want in setups:
            assert np.array_equal(bits(r.read_accum(4)[..., :3]), bits(want[..., :3]))
    finally:
        for r, _w in setups:
            r.close()


def test_create_destroy_does_not_leak_device_memory(pkg, scenes):
    import torch
    cfg = scenes.config(1, width=320, height=200, rays_per_pixel=4)
    tris, bvh = scenes.mesh(cfg["mesh"])
    sp = scenes.spheres_for(cfg["spheres"])

    def cycle():
        r = pkg.Renderer(pkg.default_params(width=320, height=200, rays_per_pixel=4), device=0)
        with r:
            r.upload_scene(sp, tris, bvh)
            r.upload_scene(sp, tris, bvh)            # re-upload replaces the scene buffers
            r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
            r.render(2)
            r.synchronize()

    cycle()
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info(0)
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 64 << 20, "device memory shrank by %d MiB over 20 create/destroy cycles" % ((free0 - free1) >> 20)
