#!/usr/bin/env python3
"""How much of the path kernel's drain can N renders in flight hide?  N contexts (one stream each, PT_PIPELINE=0) render the
same configuration in turn, enqueued back to back; wall time per render (needs a GPU).
usage: tools/multi_context_probe.py [config] [renders] [max contexts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PT_PIPELINE"] = "0"
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = scenes.config(n)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
rs = []
for _ in range(M):
    r = pkg.Renderer(p)
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    r.render_frames(0, cfg["n_frames"]); r.render_frames(0, cfg["n_frames"]); r.synchronize()
    rs.append(r)
for m in range(1, M + 1):
    t0 = time.perf_counter()
    for k in range(K):
        rs[k % m].render_frames(0, cfg["n_frames"])
    for r in rs[:m]:
        r.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%d in flight (PT_BLOCKS_PER_CU=%s): %.3f ms per render" % (m, os.environ.get("PT_BLOCKS_PER_CU", "6"), dt * 1e3))
for r in rs:
    r.close()
