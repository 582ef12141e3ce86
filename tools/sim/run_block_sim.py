#!/usr/bin/env python3
"""Drives tools/sim/libblock_sim.so: real paths of a configuration traced by the oracle, replayed under
block-level schedules of the path kernel (traversal waves served by a shading wave; paths packed into
full waves column by column).  Analysis only.
Build: gcc -O2 -shared -fPIC -fopenmp -ffp-contract=off -o libblock_sim.so block_sim.c -lm"""
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
    _fields_ = [(n, C.c_int) for n in ("buffers", "latency", "service", "leaf_period", "leaf_first",
                                       "slots_per_column", "waves", "min_lanes")]


class Result(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lane_steps", "inner_lane_steps", "leaf_lane_steps", "idle_no_ray", "wave_steps", "casts")]


class OldPolicy(C.Structure):
    _fields_ = [("waves", C.c_int), ("slots", C.c_int), ("refill_min", C.c_int), ("swap_min", C.c_int),
                ("leaf_min", C.c_int), ("leaf_shift", C.c_int), ("leaf_period", C.c_int), ("steps_per_check", C.c_int),
                ("cost_inner", C.c_double), ("cost_leaf", C.c_double), ("cost_shade", C.c_double), ("cost_swap", C.c_double)]


class OldResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("time", "steps", "inner_lane_steps", "leaf_runs", "leaf_lane_steps",
                                          "shade_rounds", "shade_lanes", "swap_rounds", "swap_lanes")]


def main():
    lib = C.CDLL(os.path.join(ROOT, "tools", "sim", "libblock_sim.so"))
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
    sizes = (C.c_uint64 * 3)()
    lib.sim_sizes(sizes)
    print("paths %d casts %d visits %d  (%.2f casts/path, %.1f visits/cast)" % (sizes[0], sizes[1], sizes[2],
          sizes[1] / max(1, sizes[0]), sizes[2] / max(1, sizes[1])))
    cpp = (C.c_uint64 * 16)(); vpc = (C.c_uint64 * 32)()
    lib.sim_histograms(cpp, vpc)
    print("casts per path:", list(cpp))
    print("visits per cast (bins of 16):", list(vpc))

    p = OldPolicy(waves=40, slots=1, refill_min=60, swap_min=16, leaf_min=8, leaf_shift=2, leaf_period=3, steps_per_check=9,
                  cost_inner=1.0, cost_leaf=0.8, cost_shade=6.0, cost_swap=1.0)
    r = OldResult()
    lib.sim_run(C.byref(p), C.byref(r))
    print("today (refill at 60 waiting lanes, leaves every 3rd step): %.1f inner lanes per step, %.0f steps" % (r.inner_lane_steps / r.steps, r.steps))

    if os.environ.get("ONLY") == "multibuf":
        multibuf(lib)
        return

    def run_async(**kw):
        base = dict(buffers=2, latency=8, service=3, leaf_period=3, leaf_first=0, slots_per_column=0, waves=0, min_lanes=0)
        base.update(kw)
        pol = Policy(**base); res = Result()
        lib.sim_async(C.byref(pol), 64 * 40, C.byref(res))
        return res

    print("\ntraversal waves served by a shading wave: inner lanes per step (idle for want of a ray)")
    for lf in (0, 1):
        for lp in (1, 2, 3):
            for buffers in (1, 2, 3):
                row = []
                for lat in (4, 8, 16, 32):
                    for svc in (1, 3):
                        res = run_async(buffers=buffers, latency=lat, service=svc, leaf_period=lp, leaf_first=lf)
                        row.append("%.1f(%.1f)" % (64 * res.inner_lane_steps / res.lane_steps, 64 * res.idle_no_ray / res.lane_steps))
                print("leaf_first %d leaf_period %d buffers %d | lat 4/8/16/32 x service 1/3: %s" % (lf, lp, buffers, " ".join(row)))

    def run_packed(**kw):
        base = dict(buffers=0, latency=8, service=1, leaf_period=3, leaf_first=0, slots_per_column=4, waves=4, min_lanes=1)
        base.update(kw)
        pol = Policy(**base); res = Result()
        lib.sim_packed(C.byref(pol), 10, C.byref(res))
        return res

    print("\npaths packed into waves column by column: inner lanes per wave step run, wave steps per 1000 inner visits")
    for S in (4, 5, 6, 8):
        for W in (3, 4):
            if W > S:
                continue
            for ml in (1, 16, 32, 48):
                for lat in (8, 16):
                    res = run_packed(slots_per_column=S, waves=W, min_lanes=ml, latency=lat)
                    print("slots/column %d waves %d min_lanes %2d latency %2d: %.1f lanes per wave step, %.1f wave steps per 1000 visits"
                          % (S, W, ml, lat, res.inner_lane_steps / res.wave_steps, 1000 * res.wave_steps / res.inner_lane_steps))


class MBPolicy(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("buffers", "service", "shade_min", "starve_min", "leaf_period", "leaf_first", "waves")] + \
               [(n, C.c_double) for n in ("cost_step", "cost_shade", "cost_service")]


class MBResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("time", "steps", "inner_lane_steps", "shade_rounds", "shade_lanes", "services", "idle_lane_steps")]


def multibuf(lib):
    print("\nseveral paths per lane, one walked at a time, the wave shades its own waiting buffers")
    for B in (2, 3, 4):
        for svc in (3, 6, 9):
            for sm in (32, 48, 56, 60):
                for st in (8, 16, 64):
                    pol = MBPolicy(buffers=B, service=svc, shade_min=sm, starve_min=st, leaf_period=3, leaf_first=1, waves=40,
                                   cost_step=1.0, cost_shade=7.0, cost_service=0.3)
                    res = MBResult()
                    lib.sim_multibuf(C.byref(pol), C.byref(res))
                    print("buffers %d service %d shade_min %2d starve_min %2d: time %8.0f steps %8.0f lanes/step %.1f idle %.1f rounds %6.0f (%.1f lanes) services %6.0f" % (
                        B, svc, sm, st, res.time, res.steps, res.inner_lane_steps / res.steps, res.idle_lane_steps / res.steps,
                        res.shade_rounds, res.shade_lanes / max(1, res.shade_rounds), res.services))


if __name__ == "__main__":
    main()
