#!/usr/bin/env python3
"""Back-to-back renders with and without batch pipelining (PT_PIPELINE): wall time per render and equality of the images
(needs a GPU).  usage: tools/pipeline_probe.py [config] [renders]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = scenes.config(n)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
samples = cfg["width"] * cfg["height"] * cfg["rays_per_pixel"] * cfg["n_frames"]
images = {}
for mode in ("0", "1"):
    os.environ["PT_PIPELINE"] = mode
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        for _ in range(3):
            r.render_frames(0, cfg["n_frames"])
        r.synchronize(); r.reset_timing()
        t0 = time.perf_counter()
        for _ in range(K):
            r.render_frames(0, cfg["n_frames"])
        r.synchronize()
        dt = time.perf_counter() - t0
        t = r.timing()
        # a running mean over several calls, with a camera move in between
        r.render_frames(0, 2); r.render_frames(2, 1); r.render_frames(3, 2)
        a = r.read_accum().copy()
        pos = list(cfg["camera_position"]); pos[0] += 1.5
        r.set_camera(pos, cfg["camera_rotation"])
        r.render_frames(0, 1); r.render_frames(1, 1); r.render_frames(2, 1)
        b = r.read_accum().copy()
        images[mode] = (a, b)
        print("PT_PIPELINE=%s: %.3f ms per render, %.1f Msamples/s; kernels per batch: primary %.3f path %.3f resolve %.3f ms (%d batches)" % (
            mode, dt / K * 1e3, samples * K / dt / 1e6, t["primary_ms"] / t["batches"], t["path_ms"] / t["batches"], t["resolve_ms"] / t["batches"], t["batches"]))
same = all(np.array_equal(images["0"][k].view(np.uint32), images["1"][k].view(np.uint32)) for k in range(2))
print("images bit-equal:", same)
sys.exit(0 if same else 1)
