#!/usr/bin/env python3
"""Wall time of one rank's share of a render under both splits, run alone on a GPU, without the collective: what
bench.py --gpus N would see per rank.  Two protocols: every render followed by the wait a synchronous collective needs
(ShardedRenderer.render), and the steps in flight (ShardedRenderer.render_async: the image written to a staging tensor by the
render's last frame fold, a side stream made to wait for it, no wait on the host).
usage: tools/shard_cost.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
cfg = scenes.config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
K = 6
n = cfg["n_frames"]
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    for split in ("pixels", "frames"):
        base = None
        for world in (1, 2, 4, 8):
            if split == "frames" and n % world:
                continue
            r.set_pixel_shard(0, world if split == "pixels" else 1)
            def once():
                if split == "pixels":
                    r.render_frames(0, n)
                else:
                    r.render_frames_sum(0, n // world, world)
                r.synchronize()
            once(); once()
            t0 = time.perf_counter()
            for _ in range(K):
                once()
            ms = (time.perf_counter() - t0) / K * 1e3
            base = base or ms
            # the same share with the steps in flight
            import torch
            bound = torch.zeros(cfg["width"] * cfg["height"] * 4, dtype=torch.float32, device="cuda")
            staging = [torch.empty_like(bound), torch.empty_like(bound)]
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            r.bind_accum(bound.data_ptr())
            def flight(k):
                r.accum_acquire(side.cuda_stream)
                if split == "pixels":
                    r.next_render_output(staging[k & 1].data_ptr())
                    r.render_frames(0, n)
                else:
                    r.next_render_output(staging[k & 1].data_ptr(), move=True)
                    r.render_frames_sum(0, n // world, world)
                r.accum_release(side.cuda_stream)
            for k in range(2):
                flight(k)
            r.synchronize()
            t0 = time.perf_counter()
            for k in range(3 * K):
                flight(k)
            r.synchronize()
            ms_f = (time.perf_counter() - t0) / (3 * K) * 1e3
            base_f = ms_f if world == 1 else base_f
            r.bind_accum(None)
            print("%s split, world %d: waiting after every render %.3f ms -> efficiency %.1f %%; steps in flight %.3f ms -> %.1f %% "
                  "(render time only)" % (split, world, ms, 100 * base / world / ms, ms_f, 100 * base_f / world / ms_f), flush=True)
