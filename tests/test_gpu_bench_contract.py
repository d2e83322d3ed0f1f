"""bench.py prints ONE JSON line with the driver's contract keys (plus `roofline` and `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_the_contract_line():
    env = dict(os.environ, RT_BENCH_CPU_SPP="1")            # bounded CPU sample: 1 spp of the frame (1.6 s)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Msamples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1000.0 and abs(d["value"] - 96.0e6 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.0 < rf["frac"] <= 1.0       # executed flops: a fraction of the peak by construction
    assert rf["effective_frac"] > rf["frac"]                                                      # culling: executed work < the brute-force scan's work
    assert 5.0 < rf["executed_per_ray"]["sphere_tests"] < 488.0
    # every other BASELINE.json config rides in the same line
    oc = d["other_configs"]
    assert set(oc) == {"C3", "C4", "C5_one_gpu"}
    for k in oc:
        assert oc[k]["unit"] == "Msamples/s" and oc[k]["value"] > 100.0 and "workload" in oc[k] and "roofline" in oc[k], k
        assert 0.0 < oc[k]["roofline"]["frac"] <= 1.0, (k, oc[k]["roofline"]["frac"])              # every roofline of the line is a roofline
    assert "1000spp" in oc["C3"]["workload"] and "1920x1080 256spp" in oc["C4"]["workload"] and "3840x2160 4096spp" in oc["C5_one_gpu"]["workload"]
    r4 = oc["C4"]["roofline"]
    assert r4["bound"] == "l2-gather" and r4["unit"] == "TB/s" and abs(r4["frac"] - r4["achieved"] / r4["peak"]) < 1e-9 and 0.0 < r4["frac"] < 1.0
    assert 100.0 < r4["per_sample"]["node_visits"] < 400.0 and 2.0 < r4["per_sample"]["rays"] < 8.0
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and 0.05 < cb["value"] < 50.0
