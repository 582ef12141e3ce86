"""Batches of frames in flight on the GPU: the kept primary pass, the two batch slots (one stream each, resolve
kernels chained in frame order), the grid sized to the work.  None
of these may change a bit of the image: everything here is compared with the CPU oracle on the same frames
(shader.wgsl:99-106 is the accumulation whose order matters), through the C ABI."""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

KW = dict(width=96, height=54, rays_per_pixel=3, max_bounce_count=4)


def _setup(scenes, n=1):
    cfg = scenes.config(n, width=KW["width"], height=KW["height"], rays_per_pixel=KW["rays_per_pixel"])
    tris, bvh = scenes.mesh(cfg["mesh"])
    return cfg, tris, bvh, scenes.spheres_for(cfg["spheres"])


def _renderer(pkg, env=None, **flags):
    """A context created under an environment (the switches are read by pt_create)."""
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return pkg.Renderer(pkg.default_params(flags=flags.get("flags", pkg.PT_FLAG_EXACT_MATH), **KW), device=0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, first, n):
    op = oracle_shared.default_params(**KW)
    sc = oracle_shared.scene(spheres, tris, bvh)
    return oracle_shared.render_frames(op, sc, pos, rot, first, n)


@pytest.mark.parametrize("env", [{}, {"PT_PIPELINE": "0"}, {"PT_MAX_BATCH": "1"}, {"PT_MAX_BATCH": "2", "PT_PIPELINE_MB": "0"},
                                 {"PT_ADAPT_GRID": "0"}],
                         ids=["default", "one-slot", "one-frame-batches", "no-room-to-pipeline", "full-grid"])
def test_calls_in_a_row_accumulate_like_the_oracle(pkg, scenes, oracle_shared, env):
    """Seven frames over four calls, enqueued without a synchronisation in between (the batches alternate between
    the two slots and their resolve kernels must still fold the frames in order), with the hit count known from
    the third call on (the grid is then sized to the few paths of this small image)."""
    cfg, tris, bvh, spheres = _setup(scenes)
    with _renderer(pkg, env) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        r.render_frames(0, 2)
        r.render_frames(2, 1)
        r.synchronize()
        r.render_frames(3, 3)
        r.render_frames(6, 1)
        got = r.read_accum(4)
        t = r.timing()
    want = _oracle_frames(oracle_shared, spheres, tris, bvh, cfg["camera_position"], cfg["camera_rotation"], 0, 7)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    if env.get("PT_MAX_BATCH") is None:
        assert t["batches"] == 4


def test_primary_pass_is_kept_until_the_view_changes(pkg, scenes, oracle_shared):
    """The primary pass runs once per slot for a still camera (its kernel time stops growing), again after a move,
    not again when the same pose is set once more; every image equals the oracle's for its pose."""
    cfg, tris, bvh, spheres = _setup(scenes)
    pos, rot = list(map(float, cfg["camera_position"])), list(map(float, cfg["camera_rotation"]))
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        for _ in range(2):
            r.render_frames(0, 1)       # one per slot
        warm = r.timing()["primary_ms"]
        assert warm > 0
        for _ in range(4):
            r.render_frames(0, 1)
        r.set_camera(pos, rot)          # the same pose: nothing to redo
        r.render_frames(0, 2)
        assert r.timing()["primary_ms"] == warm
        still = r.read_accum(4)
        moved_pos = [pos[0] + 3.0, pos[1], pos[2] - 2.0]
        r.set_camera(moved_pos, rot)
        r.render_frames(0, 2)
        assert r.timing()["primary_ms"] > warm
        moved = r.read_accum(4)
    want_still = _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, 0, 2)
    want_moved = _oracle_frames(oracle_shared, spheres, tris, bvh, moved_pos, rot, 0, 2)
    assert np.array_equal(bits(still[..., :3]), bits(want_still[..., :3]))
    assert np.array_equal(bits(moved[..., :3]), bits(want_moved[..., :3]))
    assert not np.array_equal(bits(still), bits(moved))


def test_a_new_scene_or_a_refit_drops_the_kept_primary_pass(pkg, scenes, oracle_shared):
    """upload_scene and refit_scene change what the primary rays hit: renders after them equal a fresh context's."""
    cfg, tris, bvh, spheres = _setup(scenes)
    other_tris, other_bvh = scenes.mesh("dragon_8k")
    pos, rot = cfg["camera_position"], cfg["camera_rotation"]
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        r.render_frames(0, 1); r.render_frames(1, 1); r.render_frames(2, 1)
        r.upload_scene(spheres, other_tris, other_bvh)
        r.render_frames(0, 2)
        after_upload = r.read_accum(4)
        r.upload_scene(spheres, tris, bvh)
        r.render_frames(0, 1); r.render_frames(1, 1)
        squashed = tris.copy()
        squashed[:, 1::3] *= np.float32(0.5)            # y of every vertex
        r.refit_scene(squashed)
        r.render_frames(0, 2)
        after_refit = r.read_accum(4)
    with _renderer(pkg) as f:
        f.upload_scene(spheres, other_tris, other_bvh)
        f.set_camera(pos, rot)
        f.render_frames(0, 2)
        fresh_other = f.read_accum(4)
        f.upload_scene(spheres, tris, bvh)
        f.refit_scene(squashed)
        f.render_frames(0, 2)
        fresh_refit = f.read_accum(4)
    assert np.array_equal(bits(after_upload), bits(fresh_other))
    assert np.array_equal(bits(after_refit), bits(fresh_refit))
    want = _oracle_frames(oracle_shared, spheres, other_tris, other_bvh, pos, rot, 0, 2)
    assert np.array_equal(bits(after_upload[..., :3]), bits(want[..., :3]))


def test_operations_on_the_accumulation_buffer_keep_their_place_between_batches(pkg, scenes, oracle_shared):
    """reset, scale and the sum mode are enqueued on the first slot's stream while batches run on both: the order
    of the calls is the order of the operations on the buffer."""
    cfg, tris, bvh, spheres = _setup(scenes)
    pos, rot = cfg["camera_position"], cfg["camera_rotation"]
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        r.render_frames(0, 1); r.render_frames(1, 1); r.render_frames(2, 1)
        r.reset_accum()
        r.render_frames_sum(0, 2, 1)
        r.render_frames_sum(2, 1, 1)
        r.scale_accum(0.25)
        r.render_frames_sum(3, 1, 1)
        got = r.read_accum(4)
    op = oracle_shared.default_params(**KW)
    sc = oracle_shared.scene(spheres, tris, bvh)
    want = oracle_shared.render_frames_sum(op, sc, pos, rot, 0, 2, 1)
    want = oracle_shared.render_frames_sum(op, sc, pos, rot, 2, 1, 1, accum=want)
    want[..., :3] *= np.float32(0.25)
    want = oracle_shared.render_frames_sum(op, sc, pos, rot, 3, 1, 1, accum=want)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_static_render_after_pipelined_batches(pkg, scenes):
    """The entry point that allocates nothing runs on the first slot's stream with its own buffers: it continues
    the running mean the pipelined batches have left, and they continue its."""
    cfg, tris, bvh, spheres = _setup(scenes)
    with _renderer(pkg) as r, _renderer(pkg, {"PT_PIPELINE": "0"}) as plain:
        for x in (r, plain):
            x.upload_scene(spheres, tris, bvh)
            x.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        r.render_frames(0, 1); r.render_frames(1, 1); r.render_frames(2, 1)
        r.render_frames_static(3, 2)
        r.render_frames(5, 1); r.render_frames(6, 1)
        plain.render_frames(0, 7)
        assert np.array_equal(bits(r.read_accum(4)), bits(plain.read_accum(4)))


def test_pipelined_full_size_equals_one_slot(pkg, scenes):
    """Configuration 2 at full size in default math, eight renders in a row on two slots against the same on one."""
    cfg = scenes.config(2)
    tris, bvh = scenes.mesh(cfg["mesh"])
    images = []
    for env in ({}, {"PT_PIPELINE": "0", "PT_ADAPT_GRID": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = pkg.Renderer(pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                                                max_bounce_count=cfg["max_bounce_count"]), device=0)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        with r:
            r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
            r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
            for k in range(8):
                r.render_frames(k, 1)
            images.append(r.read_accum(4))
    assert np.array_equal(bits(images[0]), bits(images[1]))


@pytest.mark.parametrize("env", [{}, {"PT_PIPELINE": "0"}], ids=["two-slots", "one-slot"])
def test_frame_pacing_with_one_frame_in_flight(pkg, scenes, oracle_shared, env):
    """The tick loop of an interactive host: move the camera, enqueue a frame, wait for the frame BEFORE it.  The
    camera moves every second tick (a move restarts the running mean, src/main.rs:47-49); the buffer after the last
    tick equals the oracle's for the last pose and the frames rendered at it."""
    cfg, tris, bvh, spheres = _setup(scenes)
    rot = list(map(float, cfg["camera_rotation"]))
    pos = list(map(float, cfg["camera_position"]))
    with _renderer(pkg, env) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        for tick in range(9):
            if tick % 2 == 0 and tick < 8:
                pos = [pos[0] + 0.5, pos[1], pos[2]]
                r.set_camera(pos, rot)          # frame_count back to 0
            r.render(1)
            r.wait_in_flight(1)
        assert r.frame_count == 3               # ticks 6, 7, 8 at the last pose
        r.wait_in_flight(0)
        got = r.read_accum(4)
    want = _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, 0, 3)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_overlapping_calls_give_the_same_bits_every_time(pkg, scenes, oracle_shared):
    """Forty repetitions of: calls of 1, 2, 1, 3 frames, a camera move and back, the same calls again - all enqueued
    without waiting.  Which batch overlaps which, the order of the hit list and who takes which path vary from run
    to run; the image may not (and it equals the oracle's seven frames)."""
    import zlib
    cfg, tris, bvh, spheres = _setup(scenes)
    pos, rot = list(map(float, cfg["camera_position"])), list(map(float, cfg["camera_rotation"]))
    away = [pos[0] + 2.0, pos[1], pos[2]]
    sums = set()
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        for k in range(40):
            for where in (pos, away, pos):
                r.set_camera(where, rot)
                f = 0
                for n in ((1,) if where is away else (1, 2, 1, 3)):
                    r.render_frames(f, n)
                    f += n
                    if k % 3 == 0:
                        r.wait_in_flight(1)
            img = r.read_accum(4)
            sums.add(zlib.crc32(img.tobytes()))
    assert len(sums) == 1
    want = _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, 0, 7)
    assert np.array_equal(bits(img[..., :3]), bits(want[..., :3]))


@pytest.mark.parametrize("distance", [1500.0, 6000.0, 1.0e5], ids=["some-hits", "few-hits", "no-hits"])
def test_fewer_hit_pixels_than_queue_parts(pkg, scenes, oracle_shared, distance):
    """The camera far from the mesh: a handful of hit pixels (or none), so most of the queue's parts are empty and
    the waves that start on them must find the others; three calls so that the kept primary pass and the small grid
    are used too."""
    cfg, tris, bvh, spheres = _setup(scenes)
    rot = list(map(float, cfg["camera_rotation"]))
    pos = [float(cfg["camera_position"][0]), float(cfg["camera_position"][1]), distance]
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        ids, _ = r.primary_hits()
        r.render_frames(0, 2)
        r.synchronize()
        r.render_frames(2, 3)
        r.render_frames(5, 1)
        got = r.read_accum(4)
    hits = int((ids != np.iinfo(np.int32).min).sum())
    assert hits < 2000 if distance < 1e5 else hits == 0 or hits < 64
    want = _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, 0, 6)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_random_call_sequences_with_and_without_the_batch_slots():
    """tools/fuzz_calls.py: 800 random API calls (renders, sums, camera moves, resets, scales, waits, static renders,
    uploads, refits, shard changes) applied to a context with the two batch slots and to one with a single stream and
    the full grid; the accumulation buffers are compared after every read."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_calls.py"), "11", "800"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "all equal" in res.stdout


def test_accumulation_buffer_handed_to_a_caller_stream_and_back(pkg, scenes, oracle_shared):
    """pt_accum_release / pt_accum_acquire: renders, copies of their images on a stream of the caller's and (every
    other time) a clearing of the buffer by that stream, enqueued without one wait on the host.  Every copy must hold
    exactly the image of its render: a copy that ran early would miss frames, a frame fold that ran before the copy
    (or the clearing) of the render before it would spoil that copy or its own image."""
    import torch
    cfg, tris, bvh, spheres = _setup(scenes)
    pos, rot = cfg["camera_position"], cfg["camera_rotation"]
    counts = [1, 3, 2, 3, 1, 2, 3, 1]
    want = {n: _oracle_frames(oracle_shared, spheres, tris, bvh, pos, rot, 0, n) for n in set(counts)}
    op = oracle_shared.default_params(**KW)
    sc = oracle_shared.scene(spheres, tris, bvh)
    want_sum = {n: oracle_shared.render_frames_sum(op, sc, pos, rot, 0, n, 1) for n in set(counts)}
    with _renderer(pkg) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        bound = torch.zeros(KW["height"] * KW["width"] * 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.bind_accum(bound.data_ptr())
        side = torch.cuda.Stream()
        copies = [torch.empty_like(bound) for _ in counts]
        # running mean: frame 0 overwrites, so no clearing is needed between renders
        for n, out in zip(counts, copies):
            r.render_frames(0, n)
            r.accum_release(side.cuda_stream)
            with torch.cuda.stream(side):
                out.copy_(bound)
            r.accum_acquire(side.cuda_stream)
        r.synchronize()             # waits for the caller's stream up to the last acquire as well
        for n, out in zip(counts, copies):
            got = out.cpu().numpy().reshape(KW["height"], KW["width"], 4)
            assert np.array_equal(bits(got[..., :3]), bits(want[n][..., :3])), n
        # sums: every render adds to a buffer the caller's stream has cleared behind its copy
        with torch.cuda.stream(side):
            bound.zero_()
        r.accum_acquire(side.cuda_stream)
        for n, out in zip(counts, copies):
            r.render_frames_sum(0, n, 1)
            r.accum_release(side.cuda_stream)
            with torch.cuda.stream(side):
                out.copy_(bound)
                bound.zero_()
            r.accum_acquire(side.cuda_stream)
        r.synchronize()
        for n, out in zip(counts, copies):
            got = out.cpu().numpy().reshape(KW["height"], KW["width"], 4)
            assert np.array_equal(bits(got[..., :3]), bits(want_sum[n][..., :3])), n
        assert float(bound.abs().max().item()) == 0.0
        r.bind_accum(None)


@pytest.mark.parametrize("env", [{}, {"PT_MAX_BATCH": "1"}], ids=["one-batch", "one-frame-batches"])
def test_second_output_of_a_render_call(pkg, scenes, oracle_shared, env):
    """pt_next_render_output: the call's last frame fold writes the image to a buffer of the caller's too.  Copy: both
    buffers hold the running mean.  Move (sum renders): the second buffer holds the sum, the accumulation buffer zeros,
    so the next sum needs no clearing; the request holds for one call.  Pixels of other ranks' shards come out as
    zeros whatever the buffer held."""
    import torch
    cfg, tris, bvh, spheres = _setup(scenes)
    pos, rot = cfg["camera_position"], cfg["camera_rotation"]
    op = oracle_shared.default_params(**KW)
    sc = oracle_shared.scene(spheres, tris, bvh)
    mean3 = oracle_shared.render_frames(op, sc, pos, rot, 0, 3)
    sum3 = oracle_shared.render_frames_sum(op, sc, pos, rot, 0, 3, 1)
    sum2 = oracle_shared.render_frames_sum(op, sc, pos, rot, 1, 2, 2)
    shape = (KW["height"], KW["width"], 4)
    with _renderer(pkg, env) as r:
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(pos, rot)
        out = torch.full((KW["height"] * KW["width"] * 4,), 7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.next_render_output(out.data_ptr())
        r.render_frames(0, 3)
        r.synchronize()
        got = out.cpu().numpy().reshape(shape)
        assert np.array_equal(bits(got[..., :3]), bits(mean3[..., :3]))
        assert np.array_equal(bits(r.read_accum(4)[..., :3]), bits(mean3[..., :3]))
        # the request was for that call only
        out.fill_(7.0); torch.cuda.synchronize()
        r.render_frames(0, 1)
        r.synchronize()
        assert float(out.min().item()) == 7.0
        # the entry point that allocates nothing honours the request as well
        r.next_render_output(out.data_ptr())
        r.render_frames_static(0, 3)
        r.synchronize()
        assert np.array_equal(bits(out.cpu().numpy().reshape(shape)[..., :3]), bits(mean3[..., :3]))
        # sums that move: two in a row without a clearing in between
        r.reset_accum()
        for want, (first, n, stride) in ((sum3, (0, 3, 1)), (sum2, (1, 2, 2))):
            r.next_render_output(out.data_ptr(), move=True)
            r.render_frames_sum(first, n, stride)
            r.synchronize()
            got = out.cpu().numpy().reshape(shape)
            assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
            assert not r.read_accum(4).any()
        # a shard: the other ranks' pixels are written as zeros
        from path_tracing_amd.distributed import owner_mask
        r.set_pixel_shard(1, 3)
        out.fill_(7.0); torch.cuda.synchronize()
        r.next_render_output(out.data_ptr())
        r.render_frames(0, 3)
        r.synchronize()
        got = out.cpu().numpy().reshape(shape)
        mine = owner_mask(KW["width"], KW["height"], 1, 3)
        assert np.array_equal(bits(got[mine][:, :3]), bits(mean3[mine][:, :3]))
        assert not got[~mine].any()
        # refused: the accumulation buffer itself, a host pointer
        with pytest.raises(pkg.PtError):
            r.next_render_output(r.accum_device_ptr())
        with pytest.raises(pkg.PtError):
            r.next_render_output(np.zeros(16, np.float32).ctypes.data)
