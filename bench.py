#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--shard frames|pixels|both]

A "step" is one complete render of the configuration (default: benchmark configuration 4,
dragon_800k SYNTHETIC STAND-IN, 1920x1080, 256 spp = 16 frames x 16 rays/pixel,
max_bounce_count 8), scene already resident in HBM.  For N > 1 there is one process per GPU
(started by torch.distributed.run; a plain `python bench.py --gpus N` starts that launcher itself,
as a child process, before anything in this process has touched the GPU).  The render is sharded
over ranks by frames (the samples of a pixel are split: rank r renders frames r, r+N, ... into a
sum; the reference's own accumulation unit) or in 8x8-pixel tiles (every rank renders all frames
of its tiles, running mean as on one GPU, result equal to the single-GPU image bit for bit), and
the buffers are combined with ONE RCCL sum reduce to rank 0 per step (strong scaling: total
samples fixed).  The default for N > 1 times both splits; `value` is the frame split's, the
other one is listed under `splits`.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     - the path (bounce) kernel, 98 % of the time.  It streams nothing: its working set
                 sits in L2 / Infinity Cache and what it asks of the chip is 64-lane GATHER
                 instructions (four per BVH node visit).  `achieved` is the rate at which it issues
                 them (SQ_INSTS_VMEM_RD of the committed counter pass of the same workload and kernel
                 source, profiles/path_kernel_pmc_cfgN.json, over the launch time measured here with
                 HIP events); `peak` is the rate at which the chip takes such gathers when nothing
                 else is asked of it, measured by tools/ubench/gather_rate at the kernel's number of
                 cache lines per instruction (profiles/r03_ubench_gather_ceiling.jsonl).  Address-unit
                 busy share, HBM, Infinity Cache, L2 and VALU utilisation (busy and busy x live lanes)
                 are listed under `other_bounds`; the algorithmic bytes (reference layout: 36 B per
                 box test, 36 B per triangle test, 48 B per sphere per cast, counted by an
                 instrumented run of the same launches) are reported as `algorithmic_over_actual`
                 against the bytes that really left the L2s.
  exact_math   - the timed region once more under PT_FLAG_EXACT_MATH, the mode whose whole images equal
                 the CPU restatement bit for bit (N = 1 only), with a one-frame bit comparison.
  cpu_baseline - the CPU oracle (oracle/, a C restatement of the reference shader;
                 the reference itself cannot run here) on a bounded sample of the
                 same workload, rank 0 at N = 1 only.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
L2_PEAK_GBS = 34500.0        # aggregate L2, measured figure of the microarchitecture guide
MALL_PEAK_GBS = 8600.0       # Infinity Cache random-row gather figure of the same guide
# share of linear scaling predicted from single-GPU runs of one rank's share of configuration 4, render time only
# (tools/flight_probe.py, profiles/r03c_steps_in_flight.txt): the persistent kernel's start and tail do not shrink with
# the image; steps in flight hide part of them (and the collective), steps that wait pay them (and the collective) in full
PREDICTED_EFFICIENCY = {"steps_in_flight": {"frames": {"2": 0.97, "4": 0.93, "8": 0.86}, "pixels": {"2": 0.98, "4": 0.97, "8": 0.92}},
                        "steps_that_wait": {"frames": {"2": 0.96, "4": 0.90, "8": 0.81}, "pixels": {"2": 0.97, "4": 0.94, "8": 0.86}}}


ABSOLUTE_GATHER_GINSTR = 256 * 2.4 / 16.0        # one gather per 16 cycles and CU


def gather_ceiling(lines_per_instruction):
    """Peak rate of the traversal's gathers for a given number of cache lines per instruction, from the committed run of
    tools/ubench/gather_rate ceiling (profiles/r03_ubench_gather_ceiling.jsonl): rows with one lane per line, linear
    interpolation of the time per instruction between the two neighbouring line counts."""
    path = os.path.join(ROOT, "profiles", "r03_ubench_gather_ceiling.jsonl")
    if not os.path.exists(path):
        return None
    rows = sorted((r["lines_per_instruction"], r["ginstr_per_s"]) for r in map(json.loads, open(path)) if r["share"] == 1)
    if not rows:
        return None
    x = min(max(lines_per_instruction, rows[0][0]), rows[-1][0])
    for (l0, g0), (l1, g1) in zip(rows, rows[1:] + rows[-1:]):
        if l0 <= x <= l1 or l1 == l0:
            t = 0.0 if l1 == l0 else (x - l0) / (l1 - l0)
            per_instr = (1 - t) / g0 + t / g1
            return {"file": "profiles/r03_ubench_gather_ceiling.jsonl", "lines_per_instruction": lines_per_instruction,
                    "between_rows": [l0, l1], "ginstr_per_s": 1.0 / per_instr}
    return None


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("pt_kernels.hip", "pt_internal.h", "pt_exact_div.h", "pt_portable_math.h"):
        h.update(open(os.path.join(ROOT, "path-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def metric_name(cfg):
    res = {(1920, 1080): "1080p", (3840, 2160): "4K"}.get((cfg["width"], cfg["height"]), "%dx%d" % (cfg["width"], cfg["height"]))
    mesh = {"dragon_800k_synth": "dragon_800k"}.get(cfg["mesh"], cfg["mesh"])
    return "Msamples/s at %s %s, %d bounces" % (res, mesh, cfg["max_bounce_count"])


def host_cores():
    """Threads for the CPU baseline: every core this process may run on - the affinity mask, cut down to the
    cgroup's CPU quota where one is set.  Returns (count, how it was found)."""
    try:
        mask = len(os.sched_getaffinity(0))
    except AttributeError:
        mask = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = max(1, int(float(q) / float(period) + 0.999))
    except (OSError, ValueError):
        pass
    n = max(1, min(mask, quota) if quota else mask)
    return n, "affinity mask %d cores, cgroup quota %s, OMP_NUM_THREADS %s" % (
        mask, quota if quota else "none", os.environ.get("OMP_NUM_THREADS", "unset"))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process
    has not imported torch or touched the GPU), pass its output through and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the line of rank 0 goes to stdout; whatever else the ranks' libraries write there (gloo announces its
    # connections on stdout) goes to stderr, so that stdout holds the one JSON line
    with subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) as child:
        for line in child.stdout:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--shard", choices=("frames", "pixels", "both"), default="both",
                    help="N > 1: the frames of the render (samples per pixel split, sum + scale), 8x8-pixel tiles "
                         "(bit-identical image), or both one after the other (value = the frame split)")
    ap.add_argument("--wide", action="store_true",
                    help="PT_FLAG_WIDE_BVH: four-wide tree (not the reference's visit order; reported under its own workload name)")
    ap.add_argument("--sync-steps", action="store_true",
                    help="N > 1: quote the steps that wait for their kernels and their collective (default: steps in flight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented counting pass")
    ap.add_argument("--no-exact", action="store_true", help="skip the second timed region under PT_FLAG_EXACT_MATH")
    args = ap.parse_args()
    if args.wide:
        args.no_count = True          # the work counters follow the reference's binary walk
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    if os.environ.get("PT_BENCH_WATCHDOG_S"):
        # a rank that has not finished by then prints the Python stack of every thread and exits (tests of the N > 1
        # path set this: a hang of two ranks then names its place instead of running into the caller's time limit);
        # started after the imports, which can take minutes in the first process of a fresh machine
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PT_BENCH_WATCHDOG_S"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and args.gpus > 1:
        sys.exit("bench.py --gpus %d runs under a launcher that started %d ranks" % (args.gpus, world))
    # PT_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo, to rehearse the N > 1 code path
    # on a one-GPU box (RCCL refuses two ranks on one device); its numbers mean nothing
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        os.environ.setdefault("PT_SAMPLE_BUDGET_MB", "8192")      # N contexts share one GPU's memory
    torch.cuda.set_device(local_rank)
    # PT_BENCH_FORCE_GROUP=1 (started with torch.distributed.run --nproc-per-node 1): a process group of ONE rank
    # over RCCL, so that everything the N > 1 path calls can be run on a one-GPU box
    grouped = world > 1 or os.environ.get("PT_BENCH_FORCE_GROUP") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    from path_tracing_amd import scenes
    from path_tracing_amd.distributed import ShardedRenderer

    cfg = scenes.config(args.config)
    tris, bvh = scenes.mesh(cfg["mesh"])
    spheres = scenes.spheres_for(cfg["spheres"])
    W, H, rpp, n_frames = cfg["width"], cfg["height"], cfg["rays_per_pixel"], cfg["n_frames"]
    spp = rpp * n_frames
    samples_per_step = W * H * spp

    def make_renderer(flags=0):
        params = pkg.default_params(width=W, height=H, rays_per_pixel=rpp,
                                    max_bounce_count=cfg["max_bounce_count"], flags=flags)
        r = pkg.Renderer(params, device=local_rank)
        r.upload_scene(spheres, tris, bvh)
        r.set_camera(cfg["camera_position"], cfg["camera_rotation"])
        return r

    r = make_renderer(pkg.PT_FLAG_WIDE_BVH if args.wide else 0)
    info = r.scene_info()

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
            torch.cuda.synchronize()

    def run_split(mode, pipelined):
        """W warm-up steps, then K timed steps of one way to split the render; returns the time (maximum over ranks),
        the library's event timing of this rank and, for N > 1, every rank's figures.  pipelined (N > 1): the steps are
        enqueued without a wait on the host, collectives included (ShardedRenderer.render_async), as the steps of the
        single-GPU run are; otherwise every step waits for its kernels and then for its collective."""
        # the image ends on rank 0
        sharded = ShardedRenderer(r, rank, world, mode=mode, collective="reduce",
                                  force_collective=world == 1) if grouped else None
        phase = {"render_s": 0.0, "collective_s": 0.0}

        def step(timed=False):
            if sharded is not None and pipelined:
                sharded.render_async(0, n_frames)
            elif sharded is not None:
                sharded.render(0, n_frames)
                if timed:
                    for k in phase:
                        phase[k] += sharded.last_timing[k]
            else:
                # the reference protocol: running mean over frames.  Like State::render the call only enqueues: the steps
                # of the timed region run back to back on the GPU (the next render starts while this one drains) and the
                # region ends with a synchronisation.
                r.render_frames(0, n_frames)

        def drain():
            if sharded is not None:
                sharded.finish()
            r.synchronize()

        for _ in range(args.warmup):
            step()
        drain()
        r.reset_timing()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=True)
        drain()
        fence()
        elapsed = time.perf_counter() - t0
        timing = r.timing()
        per_rank = None
        if grouped:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            mine = {"rank": rank, "path_ms_per_step": timing["path_ms"] / args.steps,
                    "primary_ms_per_step": timing["primary_ms"] / args.steps,
                    "resolve_ms_per_step": timing["resolve_ms"] / args.steps,
                    "frames_per_step": timing["frames"] / args.steps}
            if not pipelined:      # the host sees the two phases of a step only when it waits between them
                mine["render_ms_per_step"] = phase["render_s"] / args.steps * 1e3
                mine["collective_ms_per_step"] = phase["collective_s"] / args.steps * 1e3
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        checksum = None
        if sharded is not None and rank == 0:
            checksum = float(sharded.image().double().sum().item())
        return elapsed, timing, per_rank, checksum

    # the split `value` is quoted on comes first: frames (samples per pixel split over the GPUs) unless --shard pixels
    modes = ["single"] if not grouped else (["frames", "pixels"] if args.shard == "both" else [args.shard])
    # N > 1: every split is timed twice, with a wait after every step's kernels and collective (which shows the two
    # phases) and with the steps in flight; `value` is the latter unless --sync-steps
    runs_sync = {m: run_split(m, False) for m in modes} if grouped else {}
    runs = runs_sync if grouped and args.sync_steps else {m: run_split(m, grouped) for m in modes}
    shard_mode = modes[0]
    elapsed, timing, per_rank, _ = runs[shard_mode]
    parallelism_of = {
        "single": "single GPU, running mean",
        "pixels": "8x8-pixel tiles sharded over %d GPUs, one RCCL sum reduce to rank 0 per step" % world,
        "frames": "samples per pixel split over %d GPUs by frame (frames f = rank mod %d summed per GPU), one RCCL sum "
                  "reduce to rank 0 per step, scaled by 1/frames" % (world, world),
    }

    result = None
    if rank == 0:
        value = samples_per_step * args.steps / elapsed / 1e6
        result = {
            "metric": metric_name(cfg),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": cfg["name"] + ("+wide4" if args.wide else ""),
                "note": "dragon_800k is absent from the reference; SYNTHETIC STAND-IN = dragon_80k with every "
                        "triangle split 1->9, BVH from this repo's builder (max_depth %d)" % scenes.REGEN_MAX_DEPTH
                        if cfg["mesh"] == "dragon_800k_synth" else "",
                "resolution": [W, H], "spp": spp, "frames": n_frames, "rays_per_pixel": rpp,
                "max_bounce_count": cfg["max_bounce_count"], "triangles": int(len(tris)),
                "bvh_nodes": int(len(bvh)), "bvh_depth": info["bvh_depth"], "spheres": int(len(spheres)),
                "parallelism": parallelism_of[shard_mode],
            },
            "kernel_ms_per_frame": {k: timing[k] / max(1, timing["frames"])
                                    for k in ("primary_ms", "path_ms", "resolve_ms")},
            "kernel_launches_per_step": timing["batches"] / max(1, args.steps),
        }
        if grouped:
            def split_entry(m):
                e, _, pr, chk = runs[m]
                es, _, prs, chks = runs_sync[m]
                return {"value": samples_per_step * args.steps / e / 1e6, "ms_per_step": e / args.steps * 1e3,
                        "parallelism": parallelism_of[m], "per_rank": pr,
                        "collective_ms_per_step": max(p["collective_ms_per_step"] for p in prs),
                        "synchronous": {"value": samples_per_step * args.steps / es / 1e6, "ms_per_step": es / args.steps * 1e3,
                                        "per_rank": prs,
                                        "note": "the same steps with a wait after every step's kernels and after its collective"},
                        "image_checksum_equal": chk == chks}
            result["step_protocol"] = ("synchronous: every step waits for its kernels, then for its collective" if args.sync_steps else
                                       "steps in flight: the render of every step (its last frame fold writes the image to one of two staging "
                                       "tensors) and the asynchronous collective on that tensor are enqueued without a wait on the host, as "
                                       "the steps of the single-GPU run are; the timed region ends with a wait for all of them")
            result["per_rank"] = per_rank
            result["splits"] = {m: split_entry(m) for m in modes}
            result["collective_ms_per_step"] = result["splits"][shard_mode]["collective_ms_per_step"]
            result["predicted_efficiency"] = dict(PREDICTED_EFFICIENCY, note="render time of one rank's share of configuration 4 run alone on one GPU, "
                                                  "without the collective (tools/flight_probe.py); no N > 1 run on hardware has been "
                                                  "compared with it yet")

    # ---- roofline of the dominant kernel (rank 0) -------------------------
    if rank == 0:
        avg_ms = timing["path_ms"] / max(1, timing["batches"])
        roof = {"kernel": "path_kernel", "bound": "gather issue of the CUs' vector-memory address path", "achieved": None,
                "peak": None, "unit": "G gather instructions/s", "frac": None, "traffic": None,
                "avg_launch_ms": avg_ms, "launches_timed": int(timing["batches"])}
        launches_per_step = timing["batches"] / args.steps     # frames are rendered in batches
        # Consecutive launches run on two streams: the next one starts while this one drains, and short ones run side by
        # side for most of their lives.  Their own durations (avg_launch_ms: HIP events around each launch, what a
        # profiler's kernel trace shows) then add up to more than the wall time, and "work of one launch / its duration"
        # would describe half a GPU.  The rates below are therefore taken over rate_launch_ms: the launch's own duration
        # when launches do not overlap, else the wall time of the timed region per launch (which also holds the resolve
        # kernels and launch gaps, so the rate is on the low side).
        roof["wall_ms_per_launch"] = elapsed / args.steps * 1e3 / launches_per_step if launches_per_step else None
        roof["launches_overlap"] = bool(launches_per_step and avg_ms > roof["wall_ms_per_launch"])
        rate_ms = roof["wall_ms_per_launch"] if roof["launches_overlap"] else avg_ms
        roof["rate_launch_ms"] = rate_ms
        roof["rate_basis"] = ("wall time of the timed region per launch (launches overlap on two streams)" if roof["launches_overlap"]
                              else "mean duration of a launch (HIP events on its stream)")
        pfile = "profiles/path_kernel_pmc_cfg%d.json" % args.config
        ppath = os.path.join(ROOT, pfile)
        pj = None
        if os.path.exists(ppath):
            with open(ppath) as f:
                pj = json.load(f)
        if pj and pj.get("workload") == cfg["name"] and world == 1 and "path_kernel" in pj and timing["batches"] and not args.wide:
            d = pj["path_kernel"]
            live_s = rate_ms * 1e-3
            roof["counter_pass"] = {"file": pfile, "launch_ms_in_pass": d["trace_mean_ms"],
                                    "clock_ghz_in_pass": d["clock_ghz"],
                                    "kernel_source_matches": pj.get("source_hash") == kernel_source_hash(),
                                    "note": "counter values are those of this committed pass of the same workload, not of this run; "
                                            "the launch time is this run's"}
            lines = d["l1_tag_lookups"] / d["vmem_read_instructions"]
            ceiling = gather_ceiling(lines)
            if ceiling:
                # achieved: vector-memory read instructions the kernel issued (SQ_INSTS_VMEM_RD of the pass) over the launch
                # time measured here.  peak: the rate at which the chip takes such instructions when nothing else is asked
                # of it, measured by tools/ubench/gather_rate for gathers that touch as many cache lines per instruction
                # as the kernel's do (TCP_TOTAL_CACHE_ACCESSES / instruction of the pass).
                roof["achieved"] = d["vmem_read_instructions"] / live_s / 1e9
                roof["peak"] = ceiling["ginstr_per_s"]
                roof["frac"] = roof["achieved"] / roof["peak"]
                roof["peak_source"] = ceiling
                roof["gather_instructions_per_launch"] = d["vmem_read_instructions"]
                roof["cache_lines_per_gather"] = lines
                roof["absolute_ceiling"] = {"peak": ABSOLUTE_GATHER_GINSTR, "frac": roof["achieved"] / ABSOLUTE_GATHER_GINSTR,
                                            "note": "one 64-lane 16-byte gather per 16 cycles and CU, 256 CUs at 2.4 GHz"}
            if "lane_stats" in d:
                roof["inner_lanes_per_step"] = d["lane_stats"]["inner_lanes_per_step"]
            other = {}
            scale = d["trace_mean_ms"] / rate_ms
            if "ta_busy_frac" in d:
                other["address_unit_busy"] = {"frac": min(1.0, d["ta_busy_frac"] * scale),
                                              "note": "TA_TA_BUSY / (256 x cycles): a busy share, not a throughput"}
            if "hbm_side_bytes" in d:
                roof["traffic"] = d["hbm_side_bytes"]
                gbs = d["hbm_side_bytes"] / live_s / 1e9
                other["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "note": "bytes on the memory side of the L2s (Infinity Cache hits included)"}
                other["infinity_cache"] = {"achieved": gbs, "peak": MALL_PEAK_GBS, "unit": "GB/s", "frac": gbs / MALL_PEAK_GBS}
            if "l2_read_bytes" in d:
                gbs = d["l2_read_bytes"] / live_s / 1e9
                other["l2"] = {"achieved": gbs, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": gbs / L2_PEAK_GBS,
                               "l1_hit_rate": d["l1_hit_rate"], "l2_hit_rate": d.get("l2_hit_rate")}
            if "valu_busy_low" in d:
                other["valu"] = {"frac_low": min(1.0, d["valu_busy_low"] * scale), "frac_high": min(1.0, d["valu_busy_high"] * scale),
                                 "lane_utilisation": d["valu_lane_utilisation"],
                                 "useful": min(1.0, d["valu_busy_high"] * scale) * d["valu_lane_utilisation"],
                                 "note": "issue cycles / SIMD cycles; high = SQ_ACTIVE_INST_VALU x 4 cycles (the counter's own busy "
                                         "time), low = SQ_INSTS_VALU x 2 (as if every instruction were a 2-cycle v_fma/v_mul/v_add; "
                                         "packed f32, min/max, compares and selects take 4: tools/ubench/valu_ops); useful = high x "
                                         "the share of an instruction's 64 lanes that are live"}
            # which memory level the scene sits in, and what binds the kernel: no level's bandwidth comes near its peak
            scene_mb = (info["inner_nodes"] * 64 + len(tris) * 48) / 1e6 if "inner_nodes" in info else None
            roof["binding_unit"] = {
                "scene_resident_in": None if scene_mb is None else ("L2 (4 MiB per XCD)" if scene_mb <= 4.0 else "Infinity Cache (256 MiB)"),
                "scene_megabytes": scene_mb,
                "binding": "%s (address unit busy %.2f, %.2f of the gather ceiling; vector ALUs busy %.2f); bandwidth of that "
                           "level: L2 %.3f, Infinity Cache / HBM %.3f of peak"
                           % ("the vector ALUs (shading and the sphere tests of every cast) together with the gather issue of the record fetches"
                              if other.get("valu", {}).get("frac_high", 0) >= other.get("address_unit_busy", {}).get("frac", 0)
                              else "gather issue and dependent latency of the record fetches in the CUs",
                              other.get("address_unit_busy", {}).get("frac", float("nan")), roof["frac"] or float("nan"),
                              other.get("valu", {}).get("frac_high", float("nan")),
                              other.get("l2", {}).get("frac", float("nan")), other.get("hbm", {}).get("frac", float("nan")))}
            roof["other_bounds"] = other
        elif world > 1:
            roof["note"] = "the counter pass describes the single-GPU launch shape; see per_rank for this run"
        if not args.no_count and timing["frames"]:
            rc = make_renderer(pkg.PT_FLAG_COUNT_WORK)
            if world > 1 and shard_mode == "pixels":
                rc.set_pixel_shard(rank, world)          # the launches this rank timed
                rc.render_frames(0, n_frames)
                frames_local = n_frames
            elif world > 1:
                frames_local = len(range(rank, n_frames, world))
                rc.render_frames_sum(rank, frames_local, world)
            else:
                rc.render_frames(0, n_frames)
                frames_local = n_frames
            cnt = rc.counters()
            rc.close()
            S = len(spheres)
            path_bytes = 36 * cnt["path_kernel_box_tests"] + 36 * cnt["path_kernel_tri_tests"] \
                + 48 * S * cnt["path_kernel_ray_casts"]
            per_launch = path_bytes / launches_per_step
            pixels_local = W * H // world if (world > 1 and shard_mode == "pixels") else W * H
            # casts that ran: the path kernel's, plus one primary cast per pixel and batch of frames
            executed = cnt["path_kernel_ray_casts"] + pixels_local * launches_per_step
            roof.update({
                "algorithmic_bytes_per_launch": per_launch,
                "algorithmic_GBps": per_launch / (rate_ms * 1e-3) / 1e9,
                "algorithmic_bytes_note": "reference layout and visit order: 36 B per box test + 36 B per triangle test + "
                                          "48 B per sphere per cast; served almost entirely by the vector L1s, so this is "
                                          "not a traffic figure",
                "algorithmic_over_actual": per_launch / roof["traffic"] if roof["traffic"] else None,
                "frames_per_launch": frames_local / launches_per_step,
                "ray_casts_per_launch": cnt["path_kernel_ray_casts"] / launches_per_step,
                "box_tests_per_cast": cnt["path_kernel_box_tests"] / max(1, cnt["path_kernel_ray_casts"]),
                "tri_tests_per_cast": cnt["path_kernel_tri_tests"] / max(1, cnt["path_kernel_ray_casts"]),
                "executed_ray_casts_per_step": executed * world,         # rank 0's share x ranks
                "rays_per_second": executed * world * args.steps / elapsed,
                "reference_ray_casts_per_step": cnt["ray_casts"] * world,
                "reference_ray_casts_note": "casts the reference shader would make for the same image (it repeats the "
                                            "primary cast for every sample); not used for rays_per_second",
            })
        result["roofline"] = roof

    # ---- CPU baseline (rank 0, N = 1 only) --------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        om = entry.load_oracle_module()
        oracle = om.Oracle(shared_math=False)
        cores, cores_note = host_cores()
        op = oracle.default_params(width=W, height=H, rays_per_pixel=rpp, max_bounce_count=cfg["max_bounce_count"])
        sc = oracle.scene(spheres, tris, bvh)
        cpu_frames = min(n_frames, 8)      # 10-30 s of CPU work on 16 cores
        acc = np.zeros((H, W, 4), np.float32)
        t0 = time.perf_counter()
        oracle.render_frames(op, sc, cfg["camera_position"], cfg["camera_rotation"], 0, cpu_frames,
                             accum=acc, nthreads=cores)
        cpu_s = time.perf_counter() - t0
        cpu_samples = W * H * rpp * cpu_frames
        result["cpu_baseline"] = {
            "value": cpu_samples / cpu_s / 1e6, "unit": "Msamples/s", "cores": cores, "cores_note": cores_note, "kind": "port",
            "sample": "frames 0..%d of %d of the same workload, all pixels (%.1f M of %.1f M samples), %.1f s; "
                      "C restatement of shader.wgsl with OpenMP over rows (the reference needs wgpu and cannot run here)"
                      % (cpu_frames - 1, n_frames, cpu_samples / 1e6, samples_per_step / 1e6, cpu_s),
        }
        # parity spot check of what was just timed: GPU running mean over the same frames
        r.render_frames(0, cpu_frames)
        gimg = r.read_accum(4)[..., :3]
        err = np.abs(gimg - acc[..., :3])
        ok = (err <= 2e-3 * np.abs(acc[..., :3]) + 1e-4).all(axis=-1).mean()
        result["parity_spot_check"] = {"frames": cpu_frames, "pixels_within_2e-3_rel": float(ok),
                                       "rmse_over_mean": float(np.sqrt(np.mean((gimg.astype(np.float64) - acc[..., :3]) ** 2))
                                                               / acc[..., :3].mean()),
                                       "mean_rel_err_of_image_mean": float(abs(gimg.mean() - acc[..., :3].mean()) / acc[..., :3].mean())}

    # ---- the same steps in the mode whose images equal the CPU restatement bit for bit (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not args.wide and not args.no_exact:
        rx = make_renderer(pkg.PT_FLAG_EXACT_MATH)
        for _ in range(max(1, args.warmup)):
            rx.render_frames(0, n_frames)
        rx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rx.render_frames(0, n_frames)
        rx.synchronize()
        ex_s = time.perf_counter() - t0
        ex_value = samples_per_step * args.steps / ex_s / 1e6
        result["exact_math"] = {
            "value": ex_value, "unit": "Msamples/s", "ms_per_step": ex_s / args.steps * 1e3,
            "relative_to_value": ex_value / result["value"],
            "note": "the same timed region under PT_FLAG_EXACT_MATH (log / cos / pow of the shading code from IEEE-only "
                    "polynomials instead of v_log / v_cos / v_exp): every pixel of the image then equals the CPU restatement "
                    "bit for bit; `value` above is the default mode, whose hit decisions are the same and whose radiance is "
                    "within the tolerance of parity_spot_check",
        }
        if not args.no_cpu_baseline:
            om = entry.load_oracle_module()
            ox = om.Oracle(shared_math=True)
            cores, _ = host_cores()
            op = ox.default_params(width=W, height=H, rays_per_pixel=rpp, max_bounce_count=cfg["max_bounce_count"])
            acc = np.zeros((H, W, 4), np.float32)
            ox.render_frames(op, ox.scene(spheres, tris, bvh), cfg["camera_position"], cfg["camera_rotation"], 0, 1,
                             accum=acc, nthreads=cores)
            rx.render_frames(0, 1)
            gimg = rx.read_accum(4)[..., :3]
            same = (gimg.view(np.uint32) == acc[..., :3].view(np.uint32)).all(axis=-1)
            result["exact_math"]["bit_equal_spot_check"] = {"frames": 1, "pixels": int(same.size),
                                                            "pixels_bit_equal": int(same.sum())}
        rx.close()

    r.close()
    if rank == 0:
        print(json.dumps(result))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
