#!/usr/bin/env python3
"""Drives tools/sim/liblane_sim.so: profile real paths of config 4 with the oracle, then compare
lane-scheduling policies of the path kernel.  Analysis only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

om = entry.load_oracle_module()
entry.load_package()
from path_tracing_amd import scenes  # noqa: E402


class Policy(C.Structure):
    _fields_ = [("waves", C.c_int), ("slots", C.c_int), ("refill_min", C.c_int), ("swap_min", C.c_int),
                ("leaf_min", C.c_int), ("leaf_shift", C.c_int), ("leaf_period", C.c_int), ("steps_per_check", C.c_int),
                ("cost_inner", C.c_double), ("cost_leaf", C.c_double), ("cost_shade", C.c_double), ("cost_swap", C.c_double)]


class Result(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("time", "steps", "inner_lane_steps", "leaf_runs", "leaf_lane_steps",
                                          "shade_rounds", "shade_lanes", "swap_rounds", "swap_lanes")]


def main():
    lib = C.CDLL(os.path.join(ROOT, "tools", "sim", "liblane_sim.so"))
    o = om.Oracle(shared_math=False)
    cfg = scenes.config(int(os.environ.get("CFG", "4")))
    tris, bvh = scenes.mesh(cfg["mesh"])
    sp = scenes.spheres_for(cfg["spheres"])
    op = o.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                          max_bounce_count=cfg["max_bounce_count"])
    sc = o.scene(sp, tris, bvh)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    x0, y0, w, h = [int(v) for v in os.environ.get("REGION", "640,300,384,160").split(",")]
    n = lib.sim_profile(C.byref(op), C.byref(sc), f3(cfg["camera_position"]), f3(cfg["camera_rotation"]),
                        x0, y0, w, h, 0)
    sizes = (C.c_uint64 * 3)()
    lib.sim_sizes(sizes)
    print("paths %d casts %d visits %d  (%.1f casts/path, %.1f visits/cast)" % (sizes[0], sizes[1], sizes[2],
          sizes[1] / max(1, sizes[0]), sizes[2] / max(1, sizes[1])))
    waves = int(os.environ.get("WAVES", "40"))

    def run(**kw):
        base = dict(waves=waves, slots=1, refill_min=60, swap_min=16, leaf_min=8, leaf_shift=2, leaf_period=3, steps_per_check=6,
                    cost_inner=1.0, cost_leaf=1.0, cost_shade=6.0, cost_swap=1.0)
        base.update(kw)
        p = Policy(**base)
        r = Result()
        lib.sim_run(C.byref(p), C.byref(r))
        return r

    def show(tag, r):
        print("%-34s time %9.0f  steps %8.0f inner lanes/step %.1f  leaf runs/step %.2f lanes/run %.1f  shade rounds %6.0f (%.1f lanes)  swaps %6.0f (%.1f lanes)" % (
            tag, r.time, r.steps, r.inner_lane_steps / max(1, r.steps), r.leaf_runs / max(1, r.steps),
            r.leaf_lane_steps / max(1, r.leaf_runs), r.shade_rounds, r.shade_lanes / max(1, r.shade_rounds),
            r.swap_rounds, r.swap_lanes / max(1, r.swap_rounds)))

    for rm in (52, 56, 60, 62):
        show("1 slot refill_min=%d" % rm, run(refill_min=rm, cost_leaf=0.8))
    for sm in (4, 8, 16, 24):
        for rm in (32, 48, 56, 60):
            show("2 slots swap_min=%d refill_min=%d" % (sm, rm), run(slots=2, swap_min=sm, refill_min=rm, cost_leaf=0.8, cost_shade=5.0, cost_swap=1.0))


if __name__ == "__main__":
    main()
