#!/usr/bin/env python3
"""Prints the path kernel's lane/time diagnostics for one frame of a config (needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
from path_tracing_amd import scenes
import json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
cfg = scenes.config(n)
tris, bvh = scenes.mesh(cfg["mesh"])
p = pkg.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                       max_bounce_count=cfg["max_bounce_count"], flags=pkg.PT_FLAG_COUNT_WORK)
with pkg.Renderer(p) as r:
    r.upload_scene(scenes.spheres_for(cfg["spheres"]), tris, bvh)
    r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
    r.render_frames(0, int(os.environ.get("FRAMES", cfg["n_frames"])))
    c, d, t = r.counters(), r.debug_stats(), r.timing()
print(c); print(d); print(t)
print("trav share of wave time %.3f" % (d["trav_clocks"] / d["wave_clocks"]))
print("inner lanes/step %.1f  leaf runs/step %.3f  leaf lanes/run %.1f" % (
    d["inner_lanes"] / d["trav_steps"], d["leaf_runs"] / d["trav_steps"], d["leaf_lanes"] / max(1, d["leaf_runs"])))
print("outer iters %d  shade lanes/iter %.1f  steps/outer %.1f" % (d["outer_iters"], d["shade_lanes"] / d["outer_iters"], d["trav_steps"] / d["outer_iters"]))
print("inner visits/cast %.1f (boxes/2=%.1f)" % (d["inner_lanes"] / c["path_kernel_ray_casts"], c["path_kernel_box_tests"] / 2 / c["path_kernel_ray_casts"]))
print("tail (after a wave found the queue empty): longest %.3f of the longest wave lifetime, summed %.4f of all wave time" % (
    d["longest_tail"] / d["longest_wave"], d["tail_sum"] / d["wave_clocks"]))
print("kernel %.3f ms by events; longest wave %d clocks -> %.1f MHz timestamp clock" % (t["path_ms"], d["longest_wave"], d["longest_wave"] / t["path_ms"] / 1e3))
print("wall clock (100 MHz ticks): first wave sees the queue empty after %.3f ms, last wave ends after %.3f ms (kernel by events %.3f ms): tail %.3f ms" % (
    d["first_drained_ticks"] / 1e5, d["last_end_ticks"] / 1e5, t["path_ms"], (d["last_end_ticks"] - d["first_drained_ticks"]) / 1e5))
if json_out:
    with open(json_out, "w") as f:
        json.dump({"config": n, "frames": int(os.environ.get("FRAMES", cfg["n_frames"])),
                   "inner_lanes_per_step": d["inner_lanes"] / d["trav_steps"], "wave_steps": d["trav_steps"],
                   "shading_rounds": d["outer_iters"], "lanes_per_round": d["shade_lanes"] / d["outer_iters"],
                   "inner_visits_per_cast": d["inner_lanes"] / c["path_kernel_ray_casts"],
                   "traversal_share_of_wave_time": d["trav_clocks"] / d["wave_clocks"]}, f, indent=1)
