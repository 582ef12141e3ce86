#!/usr/bin/env python3
"""Cost of PT_FLAG_EXACT_MATH (the mode whose whole images equal the CPU restatement bit for bit): renders enqueued back
to back with and without the flag, wall time per render (needs a GPU).  usage: tools/exact_math_cost.py [config] [renders]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = scenes.config(n)
tris, bvh = scenes.mesh(cfg["mesh"])
samples = cfg["width"] * cfg["height"] * cfg["rays_per_pixel"] * cfg["n_frames"]
for flags in (0, 1, 0, 1):
    p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                           max_bounce_count=cfg["max_bounce_count"], flags=flags)
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        for _ in range(2):
            r.render_frames(0, cfg["n_frames"])
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            r.render_frames(0, cfg["n_frames"])
        r.synchronize()
        dt = time.perf_counter() - t0
    print("cfg %d flags=%d (%s math): %.3f ms per render, %.1f Msamples/s" % (
        n, flags, "exact" if flags & 1 else "hardware", dt / K * 1e3, samples * K / dt / 1e6), flush=True)
