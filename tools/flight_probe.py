#!/usr/bin/env python3
"""One rank's share of a sharded render (pixel tiles), run alone on a GPU: wall time per render when the host waits after
every render, when renders are enqueued back to back into the library's own buffer, and when every render is followed
by the hand-over of a bound buffer to a side stream that copies it, or writes its image to a second output that a side
stream waits for (the steps of ShardedRenderer.render_async without the collective).  usage: tools/flight_probe.py [--frames] [--modes=a,b] [config] [world ...]   (--frames: the frame split, sums that move)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
split = "frames" if "--frames" in sys.argv else "pixels"
only = [a[len("--modes="):].split(",") for a in sys.argv if a.startswith("--modes=")]
argv = [a for a in sys.argv[1:] if a != "--frames" and not a.startswith("--modes=")]
cfg = scenes.config(int(argv[0]) if argv else 4)
worlds = [int(a) for a in argv[1:]] or [1, 2, 4, 8]
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
n = cfg["n_frames"]
K = 12
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    bound = torch.zeros(cfg["width"] * cfg["height"] * 4, dtype=torch.float32, device="cuda")
    staging = [torch.empty_like(bound), torch.empty_like(bound)]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for world in worlds:
        if split == "frames" and n % world:
            continue
        r.set_pixel_shard(0, world if split == "pixels" else 1)
        def render():
            if split == "pixels":
                r.render_frames(0, n)
            else:
                r.render_frames_sum(0, n // world, world)
        out = []
        for name in ("wait", "flight", "flight+copy", "flight+kernelcopy", "flight+handover-only", "flight+second-output"):
            if only and name not in only[0]:
                continue
            r.bind_accum(bound.data_ptr() if name != "wait" and name != "flight" else None)
            def step(k):
                if name == "flight+second-output":      # what ShardedRenderer.render_async enqueues, minus the collective
                    r.accum_acquire(side.cuda_stream)
                    r.next_render_output(staging[k & 1].data_ptr(), move=split == "frames")
                    render()
                    r.accum_release(side.cuda_stream)
                    return
                render()
                if name == "wait":
                    r.synchronize()
                elif name != "flight":
                    r.accum_release(side.cuda_stream)
                    with torch.cuda.stream(side):
                        if name == "flight+copy":
                            staging[k & 1].copy_(bound)
                        elif name == "flight+kernelcopy":
                            torch.mul(bound, 1.0, out=staging[k & 1])      # an elementwise kernel instead of a memcpy
                    r.accum_acquire(side.cuda_stream)
            for k in range(3):
                step(k)
            r.synchronize(); r.reset_timing()
            t0 = time.perf_counter()
            for k in range(K):
                step(k)
            r.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            t = r.timing()
            out.append("%s %.3f ms (path kernel %.3f)" % (name, ms, t["path_ms"] / max(1, t["batches"])))
        r.bind_accum(None)
        print("%s split, world %d: %s" % (split, world, "; ".join(out)), flush=True)
