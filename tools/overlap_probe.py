#!/usr/bin/env python3
"""Do two renders on two streams hide each other's tails?  Renders configuration 4 K times on one context, then K/2
times on each of two contexts whose launches are enqueued alternately (needs a GPU).  Analysis only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = scenes.config(n)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
rs = []
for _ in range(2):
    r = pkg.Renderer(p)
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    r.render_frames(0, cfg["n_frames"]); r.synchronize()
    rs.append(r)
samples = cfg["width"] * cfg["height"] * cfg["rays_per_pixel"] * cfg["n_frames"]
t0 = time.perf_counter()
for _ in range(K):
    rs[0].render_frames(0, cfg["n_frames"])
rs[0].synchronize()
one = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(K // 2):
    rs[0].render_frames(0, cfg["n_frames"])
    rs[1].render_frames(0, cfg["n_frames"])
rs[0].synchronize(); rs[1].synchronize()
two = time.perf_counter() - t0
print("one stream: %.1f Msamples/s (%.2f ms per render); two streams alternating: %.1f Msamples/s (%.2f ms per render)" % (
    samples * K / one / 1e6, one / K * 1e3, samples * K / two / 1e6, two / K * 1e3))
for r in rs:
    r.close()
