#!/usr/bin/env python3
"""Path-kernel time against frames per launch, for several bounce limits: separates the per-frame
cost from the fixed cost of a launch (needs a GPU).  usage: tools/fixed_cost.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
cfg = scenes.config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
tris, bvh = scenes.mesh(cfg["mesh"])
for bounces in (cfg["max_bounce_count"], 2, 1):
    p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"], max_bounce_count=bounces)
    with pkg.Renderer(p) as r:
        r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        res = []
        for frames in (1, 2, 4, 8, 16):
            r.render_frames(0, frames); r.synchronize(); r.reset_timing()
            for _ in range(3):
                r.render_frames(0, frames)
            r.synchronize()
            t = r.timing()
            res.append((frames, t["path_ms"] / t["batches"]))
        slope = (res[-1][1] - res[-2][1]) / (res[-1][0] - res[-2][0])
        print("max_bounce_count %d: " % bounces + "  ".join("%d:%.2f" % x for x in res) +
              "  -> %.3f ms per frame + %.2f ms per launch" % (slope, res[-1][1] - slope * res[-1][0]))
