"""Two processes, one MI355X: the sharded render paths (ShardedRenderer: torch tensor bound as
accumulation buffer; pixel tiles with pt_set_pixel_shard + pt_render_frames, or frames with
pt_render_frames_sum + scale; one all-reduce) end to end on the GPU.  The box has a
single GPU, so the two ranks share it and the process group uses gloo (RCCL refuses two ranks on one
device); bench.py --gpus N runs the same code with backend nccl, one rank per GPU."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, out_dir, mode):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg = entry.load_package()
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)     # a rank that hangs says where, and ends (armed after the imports)
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import ShardedRenderer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = pkg.default_params(width=96, height=54, rays_per_pixel=2, max_bounce_count=3, flags=pkg.PT_FLAG_EXACT_MATH)
    with pkg.Renderer(p, device=0) as r:
        r.upload_scene(scenes.spheres_for("cfg1"), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        sh = ShardedRenderer(r, rank, world, mode=mode, collective="reduce" if mode == "pixels" else "all_reduce")
        sh.render(0, n_frames)
        if mode == "pixels":            # a second render restarts at frame 0 (weight 1) and is continued by a third call
            sh.render(0, 2)
            sh.render(2, n_frames - 2)
        else:                           # a second render must start from a clean buffer
            sh.render(0, n_frames)
            try:
                sh.render(1, 2)
                raise AssertionError("frame sharding accepted first_frame != 0")
            except ValueError:
                pass
        img = sh.image().cpu().numpy()
        # the same renders without a wait on the host in between (render_async / finish), then a plain render again
        if mode == "pixels":
            sh.render_async(0, n_frames); sh.render_async(0, 2); sh.render_async(2, n_frames - 2)
        else:
            sh.render_async(0, 2); sh.render_async(0, n_frames); sh.render_async(0, n_frames)
        sh.finish()
        img_async = sh.image().cpu().numpy()
        sh.render(0, n_frames)
        img_after = sh.image().cpu().numpy()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), img)
    np.save(os.path.join(out_dir, "rank%d_async.npy" % rank), img_async)
    np.save(os.path.join(out_dir, "rank%d_after.npy" % rank), img_after)
    dist.barrier()
    dist.destroy_process_group()


def _rccl_worker(rank, port, n_frames, out_dir):
    """One rank, backend nccl: the calls bench.py --gpus N makes (group creation bound to the device, reduce and
    all_reduce on the tensor the library renders into, barrier, all_gather_object) on RCCL itself."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg = entry.load_package()
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import ShardedRenderer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = pkg.default_params(width=96, height=54, rays_per_pixel=2, max_bounce_count=3, flags=pkg.PT_FLAG_EXACT_MATH)
    for mode, collective in (("pixels", "reduce"), ("pixels", "all_reduce"), ("frames", "all_reduce")):
        with pkg.Renderer(p, device=0) as r:
            r.upload_scene(scenes.spheres_for("cfg1"), tris, bvh)
            r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
            sh = ShardedRenderer(r, 0, 1, mode=mode, collective=collective, force_collective=True)
            sh.render(0, n_frames)
            np.save(os.path.join(out_dir, "%s_%s.npy" % (mode, collective)), sh.image().cpu().numpy())
            # asynchronous collectives on a side stream, as bench.py --gpus N issues them
            sh.render_async(0, 1); sh.render_async(0, n_frames); sh.render_async(0, n_frames)
            sh.finish()
            np.save(os.path.join(out_dir, "%s_%s_async.npy" % (mode, collective)), sh.image().cpu().numpy())
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None]
    dist.all_gather_object(gathered, {"rank": 0, "ms": float(t.item())})
    assert gathered == [{"rank": 0, "ms": 1.5}]
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_one_rank_process_group_on_rccl(tmp_path, oracle_shared, scenes):
    """The box has one GPU, so RCCL can only be exercised with a group of one rank: same calls, same tensors."""
    import torch.multiprocessing as mp
    n_frames = 3
    mp.spawn(_rccl_worker, args=(_free_port(), n_frames, str(tmp_path)), nprocs=1, join=True)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=96, height=54, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    mean = oracle_shared.render_frames(p, sc, cam, rot, 0, n_frames)
    for name in ("pixels_reduce", "pixels_all_reduce", "pixels_reduce_async", "pixels_all_reduce_async"):
        img = np.load(tmp_path / (name + ".npy"))
        assert np.array_equal(img[..., :3].view(np.uint32), mean[..., :3].view(np.uint32)), name
    want = oracle_shared.render_frames_sum(p, sc, cam, rot, 0, n_frames, 1) * np.float32(1.0 / n_frames)
    for name in ("frames_all_reduce", "frames_all_reduce_async"):
        img = np.load(tmp_path / (name + ".npy"))
        assert np.array_equal(img[..., :3].view(np.uint32), want[..., :3].view(np.uint32)), name


def test_two_ranks_one_gpu_pixel_shards(tmp_path, oracle_shared, scenes):
    """Pixel sharding: every rank ends with the single-GPU running mean, bit for bit."""
    import torch.multiprocessing as mp
    n_frames = 5
    mp.spawn(_worker, args=(2, _free_port(), n_frames, str(tmp_path), "pixels"), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npy")            # collective "reduce": the image is on rank 0
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=96, height=54, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    mean = oracle_shared.render_frames(p, sc, cfg["camera_position"], cfg["camera_rotation"], 0, n_frames)
    assert np.array_equal(r0[..., :3].view(np.uint32), mean[..., :3].view(np.uint32))
    # renders in flight (render_async / finish) and a plain render after them: the same image
    for name in ("rank0_async.npy", "rank0_after.npy"):
        assert np.array_equal(np.load(tmp_path / name).view(np.uint32), r0.view(np.uint32)), name


def test_two_ranks_one_gpu_frame_shards(tmp_path, oracle_shared, scenes):
    import torch.multiprocessing as mp
    n_frames = 5
    mp.spawn(_worker, args=(2, _free_port(), n_frames, str(tmp_path), "frames"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)
    cfg = scenes.config(1)
    tris, bvh = scenes.mesh("teapot")
    p = oracle_shared.default_params(width=96, height=54, rays_per_pixel=2, max_bounce_count=3)
    sc = oracle_shared.scene(scenes.spheres_for("cfg1"), tris, bvh)
    cam, rot = cfg["camera_position"], cfg["camera_rotation"]
    parts = [oracle_shared.render_frames_sum(p, sc, cam, rot, rank, len(range(rank, n_frames, 2)), 2) for rank in range(2)]
    want = (parts[0] + parts[1]) * np.float32(1.0 / n_frames)
    assert np.array_equal(r0[..., :3].view(np.uint32), want[..., :3].view(np.uint32))
    mean = oracle_shared.render_frames(p, sc, cam, rot, 0, n_frames)
    np.testing.assert_allclose(r0[..., :3], mean[..., :3], rtol=2e-6, atol=1e-7)
    for name in ("rank0_async.npy", "rank0_after.npy", "rank1_async.npy", "rank1_after.npy"):
        assert np.array_equal(np.load(tmp_path / name).view(np.uint32), r0.view(np.uint32)), name
