"""The bench line the driver parses: every committed line under profiles/ (produced by `python bench.py` on an
MI355X) must carry the contract's fields with sane values -- metric/unit from BASELINE.json, whole-job value,
the roofline object (algorithmic bytes / measured kernel time, PMC traffic) and the CPU baseline with its core
count -- and must report parity of the rows the CPU baseline rendered."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_*.json")))
LEGACY = [p for p in LINES if os.sep + "r01_" in p]  # round 1 priced the kernel against SURVEY 8(d)'s per-ray HBM model


@pytest.mark.parametrize("path", LINES, ids=[os.path.relpath(p, ROOT) for p in LINES])
def test_committed_bench_line_follows_the_contract(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["metric"].startswith("Mrays/s")
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rays = d["config"]["image"][0] * d["config"]["image"][1] * d["config"]["spp"]
    # value is whole-job rays over the timed wall time
    assert d["value"] == pytest.approx(rays * d["n_gpus"] / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-6)
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    if path in LEGACY:
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
        assert r["achieved"] == pytest.approx(r["rays_per_launch"] * r["bytes_per_ray"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    elif "necessary" in r:
        # round 3: `achieved` is the frame's ALGORITHMIC lane-operations (necessary work under the cut walk) over the
        # kernel's time -- distance from necessary work; the executed instructions (x64, active lanes) are under `issue`
        assert r["bound"] == "fp64_valu_issue" and r["unit"] == "T lane-ops/s"
        assert r["peak"] == pytest.approx(256 * 4 * 16 * 2.4e9 / 1e12)
        n = r["necessary"]
        assert n["lane_ops_per_launch"] == pytest.approx(sum(n["by_part"].values()))
        assert r["achieved"] == pytest.approx(n["lane_ops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12, rel=1e-9)
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] <= 1.0
        i = r["issue"]
        if i["executed_lane_ops_x64"] is not None:  # null (with counters_from saying why) when the PMC summary is stale
            assert r["efficiency"] == pytest.approx(n["lane_ops_per_launch"] / i["executed_lane_ops_x64"])
            assert 0.0 < r["efficiency"] <= 1.0 and r["frac"] <= i["frac_active_lanes"] <= i["frac_x64"] <= 1.0
            assert 1.0 <= i["active_lanes_per_instruction"] <= 64.0
        else:
            assert i["counters_from"]
        w, v = r["hbm_per_wave"], r["reference_visit_rule"]
        assert "nodes_per_ray" not in r and "hbm_algorithmic" not in r  # root-walk figures live under reference_visit_rule
        assert w["box_tests_per_ray"] == n["box_tests_per_ray"] <= v["nodes_per_ray"]
        assert w["frac"] == pytest.approx(w["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0, rel=1e-6)
        if r["traffic"] is not None:
            assert r["hbm_measured_frac"] == pytest.approx(r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0, rel=1e-6)
            assert r["hbm_measured_frac"] <= 1.0
        assert d["config"]["host_threads_per_rank"] >= 1
    else:
        # round 2: executed instructions x 64 against the fp64 VALU issue peak (a busy-ness figure, superseded)
        assert r["bound"] == "fp64_valu_issue" and r["unit"] == "T lane-ops/s"
        assert r["peak"] == pytest.approx(256 * 4 * 16 * 2.4e9 / 1e12)
        if r["achieved"] is not None:  # null (with counters_from saying why) when the PMC summary is stale
            assert r["achieved"] == pytest.approx(r["lane_ops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12, rel=1e-9)
            assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.0 < r["frac"] <= 1.0
        else:
            assert r["frac"] is None and r["counters_from"]
        a, w = r["hbm_algorithmic"], r["hbm_per_wave"]
        assert a["achieved_gbs"] == pytest.approx(r["rays_per_launch"] * a["bytes_per_ray"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
        assert a["peak_gbs"] == 8000.0 and w["peak_gbs"] == 8000.0
        assert w["frac"] == pytest.approx(w["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0, rel=1e-6)
        if r["traffic"] is not None:
            assert r["hbm_measured_frac"] == pytest.approx(r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9 / 8000.0, rel=1e-6)
            assert r["hbm_measured_frac"] <= 1.0
        assert d["config"]["host_threads_per_rank"] >= 1
    c = d["cpu_baseline"]
    if c is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
        if c.get("gpu_vs_oracle"):
            assert c["gpu_vs_oracle"]["hit_id_mismatches"] == 0
            assert c["gpu_vs_oracle"]["max_abs_intensity_diff"] <= 1e-6


def test_bench_lines_exist():
    assert LINES, "no committed bench lines under profiles/"


def _clean_env():
    return {k: v for k, v in os.environ.items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                         "OMP_NUM_THREADS", "RT_HOST_THREADS")}


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must run two ranks (fresh child processes under
    torch.distributed.run), never one rank printing n_gpus: 1.  --rehearse keeps the GPU out of it: the ranks
    rendezvous over gloo on 127.0.0.1 and report their shard schedules and host thread counts."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["rehearsal"] is True and d["n_gpus"] == 2 and [x["rank"] for x in d["ranks"]] == [0, 1]
    # frame sharding by timestep: rank 0 gets the even timesteps' two cameras, rank 1 the odd ones'
    assert d["ranks"][0]["frames"][:4] == [0, 1, 4, 5] and d["ranks"][1]["frames"][:4] == [2, 3, 6, 7]
    assert d["ranks"][0]["band"] == [0, 2048] and d["ranks"][1]["band"] == [2048, 4096]
    # torchrun exported OMP_NUM_THREADS=1 to the ranks; the build pool ignores it and shares the cores
    cores = len(os.sched_getaffinity(0))
    for x in d["ranks"]:
        assert x["omp_num_threads_env"] == "1"
        assert x["host_threads"] == max(1, min(cores // 2, 16))


def test_gpus_flag_that_contradicts_the_launcher_is_an_error():
    import subprocess
    import sys
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearse"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
