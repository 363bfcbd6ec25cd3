"""bench.py's contract pieces that do not need a GPU: the CPU-baseline worker runs and reports, the
JSON field list is what the driver expects, and the workload constants match BASELINE.json."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_worker_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "0.3"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["kind"] in ("reference", "port")
    assert r["sps"] > 1e6                       # > 1 Msample/s on any x86 core


def test_bench_emits_the_contract_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert f'"{key}"' in src, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert f'"{key}"' in src, key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert f'"{key}"' in src, key
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "Msamples/s" in base["metric"] and '"unit": "Msamples/s"' in src
    assert base["published"] == {} and '"vs_baseline": None' in src
    assert re.search(r"BLOCK = 8192", src)
