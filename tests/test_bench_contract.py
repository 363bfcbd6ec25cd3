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


def _canned_full_result():
    """A full result as a real run produced it: round 5's 23.5 KB object (profiles/r05_bench_full.json), the one whose single line
    the driver could not parse, plus the objects round 6 added."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_full.json")))
    full["host_streamed"]["link_roofline"]["config3_firResampler_65536_float_blocks_memcpy"] = {
        "Melements_per_s": 5522.2, "us_per_push": 11.87, "link_GBps": 28.72, "link_ceiling_GBps": 55.6, "frac": 0.516}
    full["host_streamed"]["link_roofline"]["config3_firResampler_65536_float_blocks_zero_copy"] = {
        "Melements_per_s": 6604.5, "us_per_push": 9.92, "link_GBps": 34.34, "link_ceiling_GBps": 55.6, "frac": 0.618}
    return full


def test_last_stdout_line_is_a_small_contract_line(tmp_path, monkeypatch):
    """the single 23.5 KB line could not be parsed by the driver (BENCH_r05 parsed = null).  The LAST stdout line must be
    the contract line alone, below 4 KB, and no other stdout line may start with `{`."""
    import io
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))                # the extras file goes to <ROOT>/gpurun_out
    for n_gpus in (1, 8):
        full = _canned_full_result()
        assert len(json.dumps(full)) > 20000
        if n_gpus == 8:                                                # what an 8-rank run adds
            full["n_gpus"] = 8
            full["cpu_baseline"] = None
            full["per_rank_ms_per_pass"] = [0.98] * 8
            full["scaling_efficiency"] = 0.97
            full["without_halo_exchange"] = {"what": "x" * 200, "value": 4.4e6, "unit": "Msamples/s",
                                             "shard_1M_samples_per_gpu": {"value": 7.4e5, "us_per_pass": 11.3}}
            full["shard_1M_samples_per_gpu"].update({"scaling_efficiency": 0.4,
                "passes_per_exchange_1": {"value": 3.0e5, "us_per_pass": 28.0, "exchange_us_per_pass": 17.0, "scaling_efficiency": 0.4},
                "passes_per_exchange_16": {"value": 6.5e5, "us_per_pass": 12.9, "exchange_us_per_pass": 1.4, "scaling_efficiency": 0.88}})
        buf = io.StringIO()
        bench.emit(full, None, out=buf)
        lines = buf.getvalue().splitlines()
        last = lines[-1]
        assert len(last) < 4096, len(last)
        assert sum(1 for ln in lines if ln.startswith("{")) == 1 and last.startswith("{")
        assert all(ln.startswith("extras ") for ln in lines[:-1])
        r = json.loads(last)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in r, key
        assert r["config"]["workload"] and "model" not in r["config"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
            assert key in r["roofline"], key
        assert abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-3
        assert r["roofline_config1"]["read_only_frac"] > 0 and r["roofline_config1"]["ms"] > 0
        if n_gpus == 1:
            for key in ("value", "unit", "cores", "kind", "sample"):
                assert key in r["cpu_baseline"], key
            assert r["cpu_baseline"]["value"] > 0 and len(r["cpu_baseline"]["sample"]) <= 300
            assert r["host_streamed"]["config3_firResampler_65536_float_blocks_memcpy"]["frac"] > 0       # BASELINE configs[3]
            assert r["launch_size_sweep"]["config1_cfloat_decimator"]["worst_auto_over_best"] >= 1.0
        else:
            s1 = r["shard_1M_samples_per_gpu"]                          # BASELINE configs[4] at K = 1 and at the K the design recommends
            assert s1["passes_per_exchange_1"]["scaling_efficiency"] and s1["passes_per_exchange_16"]["exchange_us_per_pass"]
            assert r["scaling_efficiency"] and len(r["per_rank_ms_per_pass"]) == 8
        # the full result is beside it, as a file and as the extras lines
        kept = json.load(open(os.path.join(str(tmp_path), r["extras_file"])))
        assert kept["launch_size_sweep"] and kept["power"] and kept["host_streamed"]
        names = [ln.split(" ", 2)[1] for ln in lines[:-1]]
        assert "launch_size_sweep" in names and "host_streamed" in names and "roofline_config1_cfloat_decimate" in names


def test_contract_line_sheds_companions_rather_than_grow():
    sys.path.insert(0, ROOT)
    import bench
    full = _canned_full_result()
    full["stage_ms"] = {f"stage_{i}": 0.123456 for i in range(400)}     # a later round's mistake
    line = bench.contract_line(full)
    assert len(json.dumps(line)) < bench.CONTRACT_LINE_LIMIT
    assert line["roofline"]["frac"] and line["cpu_baseline"]["value"] and "stage_ms" not in line


def test_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must run as TWO ranks (it ran as one and printed
    n_gpus 1).  BENCH_PLUMBING=1 leaves out the device work only: launch, rendezvous, the library's shard plans, the halo
    exchange over gloo and the max-over-ranks timing all run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BENCH_PLUMBING"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2
    assert r["halo_ok_on_every_rank"] and r["owned_outputs_tile_the_stream"]
    assert r["plumbing_only"] is True and r["value"] is None  # never mistaken for a measurement


def test_eight_ranks_at_the_shard_size_of_baseline_config_4():
    """BASELINE configs[4] as far as a box without GPUs goes (VERDICT r03 "next" #7): `python bench.py --gpus 8` starts eight ranks,
    each owning a 2^20-sample shard; the library's plans tile the stream, every rank receives its right neighbour's head, and most
    shards start in the middle of a polyphase cycle of the 3/10 resampler."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BENCH_PLUMBING"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--blocks", "128", "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["samples_per_rank"] == 1 << 20
    assert r["halo_ok_on_every_rank"] and r["owned_outputs_tile_the_stream"]
    assert r["passes_per_exchange_checked"] == [1, 16]        # configs[4]'s shard is answered at K = 1 and K = 16 in one run
    assert len(r["seconds_per_rank"]) == 8
    assert sum(1 for g in r["resampler_group_of_first_output_per_rank"] if g != 0) >= 4, r["resampler_group_of_first_output_per_rank"]


def test_eight_ranks_sixteen_super_blocks_per_exchange():
    """--passes-per-exchange 16 (VERDICT r04 "next" 7): the halos of 16 consecutive super-blocks travel in ONE message pair per rank;
    every row's halo region must hold the right neighbour's head of THAT super-block (rows differ by construction)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["BENCH_PLUMBING"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--blocks", "128", "--steps", "2", "--warmup", "1", "--no-extras",
                          "--passes-per-exchange", "16"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["ranks_seen"] == 8 and r["passes_per_exchange"] == 16
    assert r["halo_ok_on_every_rank"] and r["owned_outputs_tile_the_stream"]


def test_world_size_that_disagrees_with_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", BENCH_PLUMBING="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-extras"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "refusing" in out.stderr
