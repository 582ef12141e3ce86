#!/usr/bin/env python3
"""Can a small kernel of another stream run while the path kernel's full grid holds every CU?  Context A renders configuration 4
(a 40 ms path kernel, six blocks per CU); meanwhile context B (a 64x64 image) clears its accumulation buffer (a fill kernel of 8
registers) and waits for it, again and again: the latencies tell when such a kernel gets onto the GPU (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PT_PIPELINE"] = "0"
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
cfg = scenes.config(4)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
small = pkg.default_params(width=64, height=64, rays_per_pixel=1, max_bounce_count=1)
with pkg.Renderer(p) as a, pkg.Renderer(small) as b:
    a.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    a.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    a.render_frames(0, cfg["n_frames"]); a.synchronize()
    b.reset_accum(); b.synchronize()
    idle = []
    for _ in range(20):
        t0 = time.perf_counter(); b.reset_accum(); b.synchronize(); idle.append((time.perf_counter() - t0) * 1e3)
    a.render_frames(0, cfg["n_frames"])
    t_start = time.perf_counter()
    busy = []
    while (time.perf_counter() - t_start) < 0.035:
        t0 = time.perf_counter(); b.reset_accum(); b.synchronize(); busy.append(((t0 - t_start) * 1e3, (time.perf_counter() - t0) * 1e3))
    a.synchronize()
    total = (time.perf_counter() - t_start) * 1e3
    print("fill kernel + wait on an idle GPU: median %.3f ms" % sorted(idle)[len(idle) // 2])
    print("while the path kernel runs (%.1f ms): %d fills, latencies ms (at ms): %s" % (
        total, len(busy), " ".join("%.2f(@%.1f)" % (l, t) for t, l in busy[:12])))
