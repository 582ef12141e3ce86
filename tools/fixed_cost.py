#!/usr/bin/env python3
"""Wall time of a render call against the frames it holds, for renders enqueued back to back: separates the cost per
frame from the fixed cost of a launch (the drain of the persistent path kernel, launch gaps).  Once with every batch
on one stream (PT_PIPELINE=0), once with the two batch slots (needs a GPU).  usage: tools/fixed_cost.py [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
cfg = scenes.config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
tris, bvh = scenes.mesh(cfg["mesh"])
K = 12
for pipeline in ("0", "1"):
    os.environ["PT_PIPELINE"] = pipeline
    p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=cfg["max_bounce_count"])
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        res = []
        for frames in (1, 2, 4, 8, 16):
            for _ in range(3):
                r.render_frames(0, frames)
            r.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                r.render_frames(0, frames)
            r.synchronize()
            res.append((frames, (time.perf_counter() - t0) / K * 1e3))
        slope = (res[-1][1] - res[-2][1]) / (res[-1][0] - res[-2][0])
        print("PT_PIPELINE=%s: " % pipeline + "  ".join("%d:%.2f" % x for x in res) +
              "  -> %.3f ms per frame + %.2f ms per render call (one-frame call: %.2f ms over the per-frame cost)" % (
                  slope, res[-1][1] - slope * res[-1][0], res[0][1] - slope))
