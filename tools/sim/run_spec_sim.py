#!/usr/bin/env python3
"""Drives tools/sim/libspec_sim.so: real paths of a configuration traced by the oracle, replayed under the
path kernel's schedule with and without deferred leaves.  Analysis only."""
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
    _fields_ = [("waves", C.c_int), ("refill_min", C.c_int), ("leaf_period", C.c_int), ("steps_per_check", C.c_int),
                ("spec", C.c_int), ("defer_free", C.c_int), ("chain", C.c_int),
                ("cost_inner", C.c_double), ("cost_leaf", C.c_double), ("cost_shade", C.c_double), ("cost_chain", C.c_double)]


class Result(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("time", "steps", "inner_lane_steps", "wasted_lane_steps", "leaf_runs", "leaf_lane_steps",
                                          "shade_rounds", "shade_lanes", "stalls_leaf", "stalls_floor", "rollbacks")]


def main():
    lib = C.CDLL(os.path.join(ROOT, "tools", "sim", "libspec_sim.so"))
    o = om.Oracle(shared_math=False)
    cfg = scenes.config(int(os.environ.get("CFG", "4")))
    tris, bvh = scenes.mesh(cfg["mesh"])
    sp = scenes.spheres_for(cfg["spheres"])
    op = o.default_params(width=cfg["width"], height=cfg["height"], rays_per_pixel=cfg["rays_per_pixel"],
                          max_bounce_count=cfg["max_bounce_count"])
    sc = o.scene(sp, tris, bvh)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    x0, y0, w, h = [int(v) for v in os.environ.get("REGION", "640,300,384,160").split(",")]
    lib.sim_profile(C.byref(op), C.byref(sc), f3(cfg["camera_position"]), f3(cfg["camera_rotation"]), x0, y0, w, h, 0)
    sizes = (C.c_uint64 * 4)()
    lib.sim_sizes(sizes)
    print("paths %d casts %d visits %d improving leaves %d (%.1f casts/path, %.1f visits/cast)" % (
        sizes[0], sizes[1], sizes[2], sizes[3], sizes[1] / max(1, sizes[0]), sizes[2] / max(1, sizes[1])))
    waves = int(os.environ.get("WAVES", "40"))

    def run(**kw):
        base = dict(waves=waves, refill_min=60, leaf_period=3, steps_per_check=9, spec=0, defer_free=0, chain=0,
                    cost_inner=1.0, cost_leaf=0.8, cost_shade=6.0, cost_chain=0.6)
        base.update(kw)
        p = Policy(**base)
        r = Result()
        lib.sim_run(C.byref(p), C.byref(r))
        return r

    def show(tag, r, ref=None):
        print("%-44s time %9.0f %s steps %8.0f live lanes/step %.1f (wasted %.1f)  leaf runs/step %.2f lanes/run %.1f  rounds %6.0f (%.1f lanes)  stalls leaf %.0f floor %.0f rollbacks %.0f" % (
            tag, r.time, ("(%+.1f%%)" % (100 * (r.time / ref - 1))) if ref else "        ", r.steps, r.inner_lane_steps / max(1, r.steps),
            r.wasted_lane_steps / max(1, r.steps), r.leaf_runs / max(1, r.steps), r.leaf_lane_steps / max(1, r.leaf_runs),
            r.shade_rounds, r.shade_lanes / max(1, r.shade_rounds), r.stalls_leaf, r.stalls_floor, r.rollbacks))

    base = run()
    show("today: period 3, refill 60", base)
    for ch in (1, 2):
        show("chain %d" % ch, run(chain=ch), base.time)
    for df in (0, 1):
        for lp in (3, 4, 5, 6, 8):
            for rm in (56, 60):
                show("deferred leaves free=%d period %d refill %d" % (df, lp, rm), run(spec=1, defer_free=df, leaf_period=lp, refill_min=rm, cost_inner=1.05), base.time)


if __name__ == "__main__":
    main()
