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
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["achieved"] == pytest.approx(r["rays_per_launch"] * r["bytes_per_ray"] / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
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
