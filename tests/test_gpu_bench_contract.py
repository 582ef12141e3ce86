"""bench.py prints exactly one JSON line with the agreed fields (run on the smallest configuration)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "1", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
              "executed_ray_casts_per_step", "rays_per_second", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["frac"] is None or 0 < r["frac"] <= 1          # no counter pass is committed for configuration 1
    # rays/s counts casts that ran: at most one primary cast per pixel plus max_bounce_count casts per sample
    assert 0 < r["executed_ray_casts_per_step"] <= 640 * 360 * (1 + 4)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "Msamples/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["parity_spot_check"]["pixels_within_2e-3_rel"] > 0.99
    # the timed region once more in the mode whose images equal the CPU restatement: every pixel of the spot check
    x = d["exact_math"]
    assert x["unit"] == "Msamples/s" and x["value"] > 0 and x["ms_per_step"] > 0
    assert abs(x["relative_to_value"] - x["value"] / d["value"]) < 1e-9
    assert x["bit_equal_spot_check"]["pixels"] == 640 * 360 == x["bit_equal_spot_check"]["pixels_bit_equal"]


def _ceiling_rows():
    with open(os.path.join(ROOT, "profiles", "r03_ubench_gather_ceiling.jsonl")) as f:
        return sorted((r["lines_per_instruction"], r["ginstr_per_s"]) for r in map(json.loads, f) if r["share"] == 1)


def _check_roofline(d, cfg_no):
    """`frac` = achieved / peak with the two sides from different measurements: achieved = instruction count of the
    committed counter pass over this run's launch time, peak = the gather microbenchmark's rate at the kernel's number
    of cache lines per instruction.  Recomputed here from the files under profiles/."""
    r = d["roofline"]
    pmc = json.load(open(os.path.join(ROOT, "profiles", "path_kernel_pmc_cfg%d.json" % cfg_no)))["path_kernel"]
    assert r["counter_pass"]["file"] == "profiles/path_kernel_pmc_cfg%d.json" % cfg_no
    assert r["peak_source"]["file"] == "profiles/r03_ubench_gather_ceiling.jsonl"
    lines = pmc["l1_tag_lookups"] / pmc["vmem_read_instructions"]
    assert abs(r["cache_lines_per_gather"] - lines) < 1e-9
    rows = _ceiling_rows()
    x = min(max(lines, rows[0][0]), rows[-1][0])
    lo = max(k for k in range(len(rows)) if rows[k][0] <= x)
    hi = min(lo + 1, len(rows) - 1)
    t = 0.0 if hi == lo else (x - rows[lo][0]) / (rows[hi][0] - rows[lo][0])
    peak = 1.0 / ((1 - t) / rows[lo][1] + t / rows[hi][1])
    assert abs(r["peak"] - peak) < 1e-6 * peak
    # over the launch's own duration, or over the wall time per launch when consecutive launches overlap
    assert r["rate_launch_ms"] == (r["wall_ms_per_launch"] if r["launches_overlap"] else r["avg_launch_ms"])
    assert r["launches_overlap"] == (r["avg_launch_ms"] > r["wall_ms_per_launch"])
    achieved = pmc["vmem_read_instructions"] / (r["rate_launch_ms"] * 1e-3) / 1e9
    assert abs(r["achieved"] - achieved) < 1e-6 * achieved
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["peak"] <= r["absolute_ceiling"]["peak"] * 1.001          # the tighter of the two ceilings is the one quoted
    ob = r["other_bounds"]
    assert 0 < ob["hbm"]["frac"] < 0.2 and 0 < ob["l2"]["frac"] < 0.5 and 0 < ob["valu"]["frac_low"] < ob["valu"]["frac_high"] <= 1
    assert abs(ob["valu"]["useful"] - ob["valu"]["frac_high"] * ob["valu"]["lane_utilisation"]) < 1e-9
    assert 0 < ob["address_unit_busy"]["frac"] <= 1
    assert r["binding_unit"]["scene_resident_in"] in ("L2 (4 MiB per XCD)", "Infinity Cache (256 MiB)")


def test_headline_roofline_is_a_physical_fraction():
    """The default (headline) configuration: `peak` comes from the microbenchmark file, `frac` is in (0, 1);
    HBM-side traffic is far below the algorithmic bytes (the tree is cache resident)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert d["metric"] == "Msamples/s at 1080p dragon_800k, 8 bounces" and d["config"]["workload"].startswith("dragon_800k_synth")
    _check_roofline(d, 4)
    r = d["roofline"]
    assert r["traffic"] > 0 and r["algorithmic_over_actual"] > 10
    assert r["binding_unit"]["scene_resident_in"] == "Infinity Cache (256 MiB)"
    # executed casts, not the reference's count: below 2 casts per sample on this scene
    assert r["executed_ray_casts_per_step"] < r["reference_ray_casts_per_step"]


@pytest.mark.parametrize("cfg_no,level", [(2, "L2 (4 MiB per XCD)"), (3, "Infinity Cache (256 MiB)"), (5, "L2 (4 MiB per XCD)")])
def test_other_configurations_have_a_roofline_from_their_own_counter_pass(cfg_no, level):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg_no), "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-count"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    _check_roofline(d, cfg_no)
    assert d["roofline"]["binding_unit"]["scene_resident_in"] == level


@pytest.mark.parametrize("shard", ["pixels", "frames"])
def test_bench_under_the_launcher_with_a_one_rank_rccl_group(shard):
    """What the driver does for N > 1 (torch.distributed.run, backend nccl, ShardedRenderer, max over ranks, per-rank
    report), with the only group a one-GPU box allows: one rank (PT_BENCH_FORCE_GROUP=1)."""
    env = dict(os.environ, PT_BENCH_FORCE_GROUP="1", PT_BENCH_WATCHDOG_S="240")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--config", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--shard", shard],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0 and d["per_rank"][0]["path_ms_per_step"] > 0
    assert d["collective_ms_per_step"] >= 0 and "predicted_efficiency" in d
    assert d["splits"][shard]["image_checksum_equal"] is True and d["splits"][shard]["synchronous"]["value"] > 0


def test_plain_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself (as a child,
    before it has touched the GPU) and passes the one JSON line through.  A one-GPU box cannot hold two RCCL ranks, so
    both ranks share GPU 0 and talk over gloo here (PT_BENCH_REHEARSAL=1: the code path, not the numbers)."""
    env = dict(os.environ, PT_BENCH_REHEARSAL="1", PT_BENCH_WATCHDOG_S="240")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # nothing but the JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["per_rank"]) == 2
    assert sorted(p["rank"] for p in d["per_rank"]) == [0, 1]
    # the default for N > 1 times both splits; the line's value is the frame split's (samples per pixel split over GPUs)
    assert set(d["splits"]) == {"frames", "pixels"} and d["value"] == d["splits"]["frames"]["value"]
    assert "samples per pixel split" in d["config"]["parallelism"]
    assert all(len(s["per_rank"]) == 2 for s in d["splits"].values())
    # every split is timed with its steps in flight (the value) and with a wait after every step; same image either way
    assert d["step_protocol"].startswith("steps in flight")
    for s in d["splits"].values():
        assert s["synchronous"]["value"] > 0 and len(s["synchronous"]["per_rank"]) == 2
        assert all(p["collective_ms_per_step"] >= 0 and p["render_ms_per_step"] > 0 for p in s["synchronous"]["per_rank"])
        assert s["image_checksum_equal"] is True
