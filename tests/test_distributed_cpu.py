"""Frame sharding across ranks (path-tracing_amd/distributed.py) on CPU: two `gloo` processes, each
rendering its share of the frames with the CPU oracle standing in for the GPU renderer, one
all-reduce, scale by 1/n.  Checks the host logic that bench.py --gpus N runs with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def test_shard_frames_partitions_every_frame_once(pkg):
    from path_tracing_amd.distributed import shard_frames
    for n_frames in [1, 2, 3, 4, 7, 16, 64]:
        for world in [1, 2, 3, 4, 8]:
            seen = []
            for rank in range(world):
                first, count, stride = shard_frames(10, n_frames, rank, world)
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(10, 10 + n_frames))
    with pytest.raises(ValueError):
        shard_frames(0, 4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import render_sharded
    oracle = entry.load_oracle_module().Oracle(shared_math=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    buf = torch.zeros(27, 48, 4, dtype=torch.float32)
    arr = buf.numpy()

    def render_sum(first, count, stride):
        oracle.render_frames_sum(p, sc, cfg["camera_position"], cfg["camera_rotation"], first, count, stride,
                                 accum=arr, nthreads=2)

    def scale(s):
        buf.mul_(s)

    render_sharded(render_sum, lambda: dist.all_reduce(buf), scale, 0, n_frames, rank, world)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), arr)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_ownership_partitions_the_image(pkg):
    from path_tracing_amd.distributed import owner_mask, tile_owner
    for w, h in [(1, 1), (17, 9), (96, 54), (130, 65)]:
        for world in [1, 2, 3, 8]:
            masks = [owner_mask(w, h, r, world) for r in range(world)]
            assert np.array_equal(sum(m.astype(int) for m in masks), np.ones((h, w), int))
            assert masks[tile_owner(w - 1, h - 1, world)][h - 1, w - 1]
    # neighbouring tiles go to different ranks, so every rank sees every part of the image
    m = owner_mask(1920, 1080, 3, 8)
    assert abs(m.mean() - 1 / 8) < 0.002 and m[:540].mean() > 0.12 and m[:, :960].mean() > 0.12


def _pixel_worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import owner_mask, render_pixel_sharded
    oracle = entry.load_oracle_module().Oracle(shared_math=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    buf = torch.zeros(27, 48, 4, dtype=torch.float32)
    arr = buf.numpy()

    def render_owned(first, count):
        # the oracle renders whole images; keeping only the rank's tiles is what the device
        # does when pt_set_pixel_shard is in force
        full = oracle.render_frames(p, sc, cfg["camera_position"], cfg["camera_rotation"], first, count, nthreads=2)
        arr[owner_mask(48, 27, rank, world)] = full[owner_mask(48, 27, rank, world)]

    render_pixel_sharded(render_owned, lambda: dist.all_reduce(buf), 0, n_frames)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), arr)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pixel_sharded_render_is_the_single_rank_image(tmp_path, oracle_shared, scenes):
    import torch.multiprocessing as mp
    n_frames = 3
    mp.spawn(_pixel_worker, args=(2, _free_port(), n_frames, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    mean = oracle_shared.render_frames(p, sc, cfg["camera_position"], cfg["camera_rotation"], 0, n_frames)
    assert np.array_equal(r0.view(np.uint32), mean.view(np.uint32))          # adding zeros is exact


@pytest.mark.parametrize("n_frames", [4, 3])
def test_two_rank_sharded_render_matches_single_rank(tmp_path, oracle_shared, scenes, n_frames):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)                                   # all-reduce: every rank holds the image
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    # exact expectation: (sum over rank-0 frames) + (sum over rank-1 frames), times 1/n, all in f32
    parts = [oracle_shared.render_frames_sum(p, sc, cam, rot, rank, len(range(rank, n_frames, 2)), 2) for rank in range(2)]
    want = (parts[0] + parts[1]) * F(1.0 / n_frames)
    assert np.array_equal(r0.view(np.uint32), want.view(np.uint32))
    # and it is the single-GPU running mean up to f32 rounding (tolerance: 2e-6 relative)
    mean = oracle_shared.render_frames(p, sc, cam, rot, 0, n_frames)
    np.testing.assert_allclose(r0[..., :3], mean[..., :3], rtol=2e-6, atol=1e-7)


def _combine_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    from path_tracing_amd.distributed import combine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    # two INDEPENDENT one-rank renders inside a two-rank job: no collective, images stay apart.  Rank 1 does
    # not even call combine() a second time, which would hang rank 0 if a collective were issued.
    mine = torch.full((4,), float(rank + 1))
    res["alone_ran"] = combine(mine, 1)
    res["alone"] = mine.tolist()
    if rank == 0:
        extra = torch.full((4,), 7.0)
        combine(extra, 1, collective="reduce")
        res["extra"] = extra.tolist()
    # a render sharded over both ranks: one sum
    both = torch.full((4,), float(rank + 1))
    res["both_ran"] = combine(both, 2)
    res["both"] = both.tolist()
    # world size that disagrees with the group
    try:
        combine(torch.zeros(4), 3)
        res["mismatch"] = "no error"
    except ValueError as e:
        res["mismatch"] = str(e)
    # a one-rank SUBGROUP with the collective forced (what the one-GPU RCCL rehearsal does)
    groups = [dist.new_group([r]) for r in range(world)]
    solo = torch.full((4,), float(10 * (rank + 1)))
    res["solo_ran"] = combine(solo, 1, group=groups[rank], collective="reduce", force_collective=True)
    res["solo"] = solo.tolist()
    np.save(os.path.join(out_dir, "combine%d.npy" % rank), np.array([repr(res)]))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_renderers_in_a_larger_job_do_not_talk(tmp_path, pkg):
    """ShardedRenderer's collective step (distributed.combine): world_size 1 issues no collective on the job's default
    group; world_size must equal the group's size; a forced one-rank collective stays inside its own group."""
    import torch.multiprocessing as mp
    mp.spawn(_combine_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [eval(str(np.load(tmp_path / ("combine%d.npy" % k))[0])) for k in range(2)]
    for k in range(2):
        assert r[k]["alone_ran"] is False and r[k]["alone"] == [float(k + 1)] * 4
        assert r[k]["both_ran"] is True and r[k]["both"] == [3.0] * 4
        assert "sharded over 3 ranks but the process group has 2" in r[k]["mismatch"]
        assert r[k]["solo_ran"] is True and r[k]["solo"] == [10.0 * (k + 1)] * 4
    assert r[0]["extra"] == [7.0] * 4


def test_steps_in_flight_settle_a_buffer_before_it_is_written_again(pkg):
    """StepsInFlight on its own: buffers alternate, the collective on a buffer is waited for (and its result scaled)
    right before the render that writes that buffer again, finish() settles the older one first."""
    from path_tracing_amd.distributed import StepsInFlight
    log = []
    steps = StepsInFlight(lambda j: log.append(("render", j)), lambda j: log.append(("start", j)) or "h%d" % len(log),
                          lambda h: log.append(("wait", h)), lambda j, s: log.append(("scale", j, s)))
    assert steps.finish() is None
    assert [steps.submit(0.5), steps.submit(0.25), steps.submit(None)] == [0, 1, 0]
    assert log == [("render", 0), ("start", 0), ("render", 1), ("start", 1),
                   ("wait", "h2"), ("scale", 0, 0.5), ("render", 0), ("start", 0)]
    del log[:]
    assert steps.finish() == 0
    assert log == [("wait", "h4"), ("scale", 1, 0.25), ("wait", "h8")]          # the last call had no scale
    assert steps.finish() == 0 and len(log) == 3                                # nothing left to settle


def _flight_worker(rank, world, port, n_frames, out_dir):
    """Frame-sharded renders in flight on CPU: the oracle stands in for the GPU renderer, gloo all-reduces run as
    asynchronous operations and are waited for two calls later (StepsInFlight, the driver of
    ShardedRenderer.render_async)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import StepsInFlight, shard_frames
    oracle = entry.load_oracle_module().Oracle(shared_math=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle.scene(scenes.spheres_for("cfg1"), tris, bvh)
    staging = [torch.zeros(27, 48, 4, dtype=torch.float32) for _ in range(2)]
    call = {}

    def render_into(j):
        first, count, stride = shard_frames(0, call["n"], rank, world)
        staging[j].zero_()                  # the device moves its sum here and starts the next one from zeros
        if count:
            oracle.render_frames_sum(p, sc, cfg["camera_position"], cfg["camera_rotation"], first, count, stride,
                                     accum=staging[j].numpy(), nthreads=2)

    steps = StepsInFlight(render_into, lambda j: dist.all_reduce(staging[j], async_op=True), lambda w: w.wait(),
                          lambda j, s: staging[j].mul_(s))
    images = []
    for n in (1, n_frames, 2, n_frames):    # rank 1 renders nothing in the first call (more ranks than frames)
        call["n"] = n
        j = steps.submit(1.0 / n)
        images.append((n, j))
    last = steps.finish()
    assert last == images[-1][1]
    np.save(os.path.join(out_dir, "flight%d.npy" % rank), np.stack([staging[last].numpy(), staging[last ^ 1].numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_renders_in_flight_match_the_waiting_protocol(tmp_path, oracle_shared, scenes):
    import torch.multiprocessing as mp
    n_frames = 5
    mp.spawn(_flight_worker, args=(2, _free_port(), n_frames, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "flight0.npy"), np.load(tmp_path / "flight1.npy")
    assert np.array_equal(r0, r1)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=48, height=27, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]

    def want(n):
        parts = [oracle_shared.render_frames_sum(p, sc, cam, rot, rank, len(range(rank, n, 2)), 2) if n > rank
                 else np.zeros((27, 48, 4), F) for rank in range(2)]
        return (parts[0] + parts[1]) * F(1.0 / n)

    assert np.array_equal(r0[0].view(np.uint32), want(n_frames).view(np.uint32))      # the last call
    assert np.array_equal(r0[1].view(np.uint32), want(2).view(np.uint32))             # the one before it, in the other buffer
