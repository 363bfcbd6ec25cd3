#!/usr/bin/env python3
"""bench.py -- Msamples/s of input IQ through the FM pipeline on MI355X.

Workload (BASELINE.json configs[2]; configs[4] for N > 1): the full FM chain of
examples/fm/fm.hs:34-41 -- u8 IQ -> cfloat (fused) -> 127(->128)-tap complex FIR
decimate-by-8 -> fmDemod -> polyphase resample 3/10 (191 taps) -> 128-tap (64
half-tap) symmetric FIR -> *0.2 -- with the reference Pipes' 8192-sample block
seams reproduced bit-exactly.  One "pass" = that chain over one batch of `--blocks`
8192-sample blocks per GPU (default 65536 blocks = 2^29 samples = 1 GiB of u8 IQ), inputs
already resident in HBM; one "step" = `passes_per_step` passes (auto: ~50 ms of GPU time,
so the default 20 steps keep the GPU busy for about a second).

N > 1: one process per GPU.  The sample stream is sharded contiguously, rank r owning
samples [r*S, (r+1)*S) of each super-block; every pass each rank receives the head of its
right neighbour's shard (the composed ntaps-1 overlap of all four stages, ~4.4k samples =
8.7 KB of u8) through the LIBRARY's RCCL point-to-point (sdrhip_fm_chain_halo_exchange:
ncclSend/ncclRecv on the compute stream) and processes shard+halo; torch.distributed (gloo)
is only the control plane.  Per-GPU work is fixed: weak scaling.  The same run also reports
BASELINE configs[4]'s shard size (2^20 samples per GPU per pass).

The LAST line of stdout (rank 0) is ONE JSON object of < 4 KB: the driver's contract fields plus
`roofline` (dominant kernel = the fused convert+decimate kernel, timed with HIP events on its
own stream inside the timed region) and `cpu_baseline` (the reference's own C kernels,
oracle/_ref, timed on this host's cores; N=1 only) and a few short companions (configs[1]'s
roofline, stage times, configs[4]'s shard).  Everything else the run measures (ceilings, launch-size
sweep, host-streamed rates, power) is printed BEFORE it, one object per line prefixed `extras <name> `,
and written to gpurun_out/bench_extras_<N>gpu.json (contract_line / emit below).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK = 8192
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 78.6        # f32 VALU, UNFUSED mul+add (parity forbids FMA): 256 CU x 4 SIMD x 32 lanes x 2.4 GHz


# --------------------------------------------------------------------------------------
# CPU baseline (the checker's libraries, used here only as the reported baseline)
# --------------------------------------------------------------------------------------
def _cpu_harness():
    """oracle/cpu_chain_bench (+ a taps file): the reference's receiver loop as a COMPILED caller -- the reference's own C
    kernels (oracle/_ref) under the restated Pipes state machines, no Python in the loop.  Built by oracle/Makefile."""
    import tempfile
    import numpy as np
    import signals as S
    exe = os.path.join(ROOT, "oracle", "cpu_chain_bench")
    if not os.path.exists(exe) or not os.path.exists(os.path.join(ROOT, "oracle", "libsdr_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    fd, taps = tempfile.mkstemp(suffix=".taps")
    with os.fdopen(fd, "wb") as f:
        d, r, h = S.taps_decim127(), S.taps_resamp191(), S.taps_audio_half64()
        np.array([d.size, r.size, h.size], np.int32).tofile(f)
        for a in (d, r, h):
            np.ascontiguousarray(a, np.float32).tofile(f)
    return exe, taps


def _cpu_run(exe, taps, *args):
    out = subprocess.run([exe, taps] + [str(a) for a in args], capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_topology():
    """(physical cores, hardware threads) this process may run on."""
    threads = len(os.sched_getaffinity(0))
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (min(len(cores), threads) if cores else threads), threads


def cpu_baseline(seconds_single=6.0, seconds_all=8.0):
    """Single-thread (how the reference actually runs: one pipeline thread) and all-hardware-threads (one independent
    receiver per thread) rates of the compiled receiver loop."""
    exe, taps = _cpu_harness()
    try:
        cores, threads = cpu_topology()
        single = _cpu_run(exe, taps, seconds_single, 1)
        allc = _cpu_run(exe, taps, seconds_all, threads)
        stages = _cpu_run(exe, taps, "--stages", 0.4)
    finally:
        os.unlink(taps)
    return {
        "value": round(allc["sps_total"] / 1e6, 2), "unit": "Msamples/s", "cores": threads, "kind": allc["kind"],
        "physical_cores": cores, "hardware_threads": threads,
        "single_thread_value": round(single["sps_total"] / 1e6, 2),
        "isolated_kernels_Melements_per_s": {k[:-len("_elements_per_s")]: round(v / 1e6, 1) for k, v in stages.items() if k.endswith("_elements_per_s")},
        "harmonic_sum_of_isolated_kernels": round(stages["harmonic_sum_sps"] / 1e6, 2),
        # `sample` is on the contract line (<= 300 characters there); `sample_detail` goes with the extras
        "sample": (f"oracle/cpu_chain_bench: the FM receiver of examples/fm/fm.hs:34-41 over 64 x 8192-sample u8 IQ blocks, looped {seconds_all:.0f} s on "
                   f"{threads} threads ({cores} cores; one receiver per thread) and {seconds_single:.0f} s single-thread; kernels = the reference's own C "
                   "(oracle/_ref, -O2 -mavx2 -msse4); fmDemod + Pipes bookkeeping from the restatement"),
        "sample_detail": ("compiled C caller, no Python in the loop; kind=reference: convertCAVX / decimateAVXRC / resampleAVXRR / filterAVXSymmetricRR / "
                          "scaleAVX and the scalar kernels of the seam outputs are the reference's own C, fmDemod and the Pipes' block bookkeeping "
                          "(Haskell in the reference) come from the restatement -- fmDemod's atanf there is the fdlibm f32 MODEL (oracle/sdr_oracle.c: "
                          "orc_atanf_model, the spec since round 5; == glibc 2.35 atanf on every float), not this host's libm; "
                          "harmonic_sum_of_isolated_kernels = the five SIMD kernels alone on one cache-resident block (no seam outputs, no re-blocking, "
                          "no data movement between stages), the ceiling of what the loop can reach"),
    }


# --------------------------------------------------------------------------------------
def self_launch(n):
    """Re-run this command line under torch.distributed.run with n ranks on this node (free rendezvous port on
    127.0.0.1 -- the container's host name may not resolve).  Returns the launcher's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def plumbing_check(args, rank, world):
    """BENCH_PLUMBING=1: everything of an N-rank run EXCEPT the device work -- launch, rendezvous, shard plans from the
    library, the halo exchange (over gloo on host tensors), barrier + max-over-ranks timing -- so that the launch logic of
    `python bench.py --gpus N` can be tested where there is no GPU.  The line it prints is marked plumbing_only and carries
    no throughput."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import sdr_amd.lib as L
    import signals as S
    from sdr_amd import sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
    S_len = min(args.blocks, 128) * BLOCK                            # up to BASELINE configs[4]'s shard: 2^20 samples per rank
    plan = sharding.ShardPlan(chain, rank, world, S_len)
    stream = S.iq_u8(world * S_len + plan.halo_cap)                # the same global stream on every rank
    K = max(1, args.passes_per_exchange)
    right0 = ((rank + 1) % world) * S_len
    t0 = time.perf_counter()
    oks = {}
    # the run's own K and, as the measured run does for BASELINE configs[4]'s shard, K = 1 and K = 16 (one exchange each is enough here)
    for kk in sorted({K, 1, 16}):
        # row k: this rank's shard of super-block k (the stream with k added to every byte, so that a halo in the wrong row shows)
        rows = torch.zeros(kk, 2 * plan.n_in, dtype=torch.uint8)
        for k in range(kk):
            rows[k, :2 * S_len] = torch.from_numpy((stream[2 * plan.s0:2 * plan.s1] + np.uint8(k)).copy())
        for _ in range((args.warmup + args.steps) if kk == K else 1):
            if kk == 1:
                sharding.halo_exchange(rows[0], plan, dist)
            else:
                sharding.halo_exchange_batch(rows, plan, dist)
        oks[kk] = world == 1 or all(np.array_equal(rows[k, 2 * S_len:].numpy(), stream[2 * right0:2 * (right0 + plan.halo_cap)] + np.uint8(k)) for k in range(kk))
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ok = all(oks.values())
    mine = (plan.q0, plan.q1, ok, float(el.item()))
    plans = [mine]
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        plans = [None] * world
        dist.all_gather_object(plans, mine)
    if rank == 0:
        tiles = all(a[1] == b[0] for a, b in zip(plans[:-1], plans[1:]))
        print(json.dumps({"plumbing_only": True, "metric": "none (BENCH_PLUMBING=1: launch, plans and halo exchange only, no device work)",
                          "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks_seen": len(plans),
                          "halo_ok_on_every_rank": all(p[2] for p in plans), "owned_outputs_tile_the_stream": tiles,
                          "halo_transport": "host memory through gloo", "seconds": round(float(el.item()), 4), "passes_per_exchange": K,
                          "passes_per_exchange_checked": sorted({K, 1, 16}),
                          "samples_per_rank": S_len, "halo_samples": plan.halo_cap,
                          # a shard's first audio output rarely starts a polyphase cycle: the resampler's group (phase) of q0
                          "resampler_group_of_first_output_per_rank": [p[0] % 3 for p in plans],
                          "seconds_per_rank": [round(p[3], 4) for p in plans]}))
    if world > 1:
        dist.destroy_process_group()


def k2_source_sha256():
    """Digest of the dominant kernel's sources (the systolic kernel, the tile kernel + their launcher) with comments and white space stripped:
    profiles/k2_traffic.json is only quoted while it was measured on this version of the code (tools/summarize_profile.py
    writes the same digest)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("decimate_tile.hpp", "kernels_fast.hip", "kernels_systolic.hip"):
        text = open(os.path.join(ROOT, "sdr_amd", "csrc", name), "r").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()



# --------------------------------------------------------------------------------------
# Output: ONE small contract line as the LAST line of stdout; everything else beside it
# --------------------------------------------------------------------------------------
CONTRACT_LINE_LIMIT = 4096       # the driver keeps an 8 KB tail of stdout and parses its last line (round 5's 23.5 KB line: parsed = null)


def _short(text, limit):
    text = " ".join(str(text).split())
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def contract_line(result):
    """The contract line of a full result: metric / value / config / roofline / cpu_baseline and a handful of short companions,
    nothing a reader has to scroll for.  The shape follows the reference's own benchmark report -- one short record per
    benchmark (benchmarks/Benchmarks.hs:79-156) -- not one object holding every experiment of the run.  Returns a dict whose
    JSON is below CONTRACT_LINE_LIMIT bytes (tests/test_bench_contract.py holds it to that)."""
    line = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                          "vs_baseline", "dtype", "data", "audio_crc32_per_rank"))
    cfg = result.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "blocks_per_gpu_per_pass", "samples_per_gpu_per_pass", "passes_per_step", "ms_per_pass",
                                 "sharding", "ranks_seen_by_rccl", "passes_per_exchange", "exchange_us_per_pass", "devices_visible",
                                 "ranks_share_devices", "order"))
    line["config"]["workload"] = _short(cfg.get("workload", ""), 200)
    if cfg.get("halo_transport"):
        line["config"]["halo_transport"] = _short(cfg["halo_transport"], 60)
    if cfg.get("overlap"):
        line["config"]["overlap"] = _short(cfg["overlap"], 90)
    rf = result.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "algorithmic_flops_per_launch",
                                  "algorithmic_bytes_per_launch"))
    line["roofline"]["kernel"] = _short(rf.get("kernel", ""), 120)
    if rf.get("traffic_note"):
        line["roofline"]["traffic_note"] = _short(rf["traffic_note"], 120)
    if isinstance(rf.get("hbm"), dict):
        line["roofline"]["hbm"] = _pick(rf["hbm"], ("achieved", "peak", "unit", "frac"))
    c1 = result.get("roofline_config1_cfloat_decimate")
    if isinstance(c1, dict):
        line["roofline_config1"] = {"kernel": _short(c1.get("kernel", ""), 60), "samples_per_launch": c1.get("samples_per_launch"), "bound": "hbm",
                                    "ms": c1.get("avg_launch_ms"), "read_only_frac": c1.get("read_only_frac"), "frac": c1.get("frac"),
                                    "peak": c1.get("peak"), "unit": c1.get("unit")}
        ce = c1.get("ceilings_same_process")
        if isinstance(ce, dict):
            line["roofline_config1"]["over_best_no_arithmetic_stream"] = ce.get("kernel_over_nt_stream")
    if result.get("stage_ms"):
        line["stage_ms"] = result["stage_ms"]
    for key in ("one_pass_at_a_time", "two_passes_in_flight", "fm_carrier_input", "example_taps_chain"):
        v = result.get(key)
        if isinstance(v, dict):
            line[key] = _pick(v, ("value", "ms_per_pass"))
    s1 = result.get("shard_1M_samples_per_gpu")
    if isinstance(s1, dict):            # BASELINE configs[4]'s shard, at every passes-per-exchange measured
        line["shard_1M_samples_per_gpu"] = _pick(s1, ("value", "us_per_pass", "passes_per_exchange", "exchange_us_per_pass", "scaling_efficiency"))
        for k, v in s1.items():
            if k.startswith("passes_per_exchange_") and isinstance(v, dict):
                line["shard_1M_samples_per_gpu"][k] = _pick(v, ("value", "us_per_pass", "exchange_us_per_pass", "scaling_efficiency"))
    wh = result.get("without_halo_exchange")
    if isinstance(wh, dict):
        line["without_halo_exchange"] = {"value": wh.get("value")}
        if isinstance(wh.get("shard_1M_samples_per_gpu"), dict):
            line["without_halo_exchange"]["shard_1M_samples_per_gpu"] = _pick(wh["shard_1M_samples_per_gpu"], ("value", "us_per_pass"))
    for key in ("scaling_efficiency", "per_rank_ms_per_pass"):
        if key in result:
            line[key] = result[key]
    hs = result.get("host_streamed")
    if isinstance(hs, dict) and isinstance(hs.get("link_roofline"), dict):   # PCIe-inclusive, never `value`: the two ends + configs[3]
        lr = hs["link_roofline"]
        line["host_streamed"] = {k: _pick(lr[k], ("Msamples_per_s", "Melements_per_s", "link_GBps", "frac")) for k in
                                 ("fm_stream_1_block_per_push", "fm_stream_4096_blocks_per_push_memcpy",
                                  "config3_firResampler_65536_float_blocks_memcpy", "config3_firResampler_65536_float_blocks_zero_copy") if k in lr}
    sw = result.get("launch_size_sweep")
    if isinstance(sw, dict):
        worst = {}
        for name, rows in sw.items():
            if isinstance(rows, list):
                vals = [(r.get("auto_over_best"), r.get("blocks_per_launch")) for r in rows if isinstance(r, dict) and r.get("auto_over_best")]
                if vals:
                    w = max(vals)
                    worst[name] = {"worst_auto_over_best": w[0], "at_blocks": w[1]}
        if worst:
            line["launch_size_sweep"] = worst
    cpu = result.get("cpu_baseline")
    if isinstance(cpu, dict):
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "physical_cores", "single_thread_value"))
        line["cpu_baseline"]["sample"] = _short(cpu.get("sample", ""), 300)
    else:
        line["cpu_baseline"] = None
    if result.get("extras_file"):
        line["extras_file"] = result["extras_file"]
    # never over the limit, whatever a later round adds: drop the companions, least important first
    for key in ("launch_size_sweep", "host_streamed", "example_taps_chain", "fm_carrier_input", "two_passes_in_flight", "one_pass_at_a_time", "per_rank_ms_per_pass",
                "stage_ms", "roofline_config1"):
        if len(json.dumps(line)) < CONTRACT_LINE_LIMIT:
            break
        line.pop(key, None)
    return line


def emit(result, args, out=None):
    """Print the run: first the extras (every top-level object of the full result that is not on the contract line), one per
    stdout line, each prefixed `extras ` so that no line but the last starts with `{`; the full result also goes to
    gpurun_out/bench_extras_<N>gpu.json (what comes back from a gpurun call; copied to profiles/ by hand).  LAST: the contract line."""
    out = out or sys.stdout
    path = None
    try:
        if getattr(args, "no_extras", False):        # a profiling run (--no-extras) must not overwrite the full result of the run before it
            raise OSError("no extras collected")
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"bench_extras_{result.get('n_gpus', 1)}gpu.json")
        with open(path, "w") as f:
            json.dump(result, f, indent=1)
        result["extras_file"] = os.path.relpath(path, ROOT)
    except OSError:
        path = None
    line = contract_line(result)
    # the most-asked-for extras last (the driver keeps a tail): config1 ceilings, sweep, host-streamed
    order = [k for k in result if k not in ("roofline_config1_cfloat_decimate", "launch_size_sweep", "host_streamed")] + \
            ["power", "host_streamed", "launch_size_sweep", "roofline_config1_cfloat_decimate"]
    seen = set()
    for k in order:
        if k in seen or k not in result or result[k] is None or k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "extras_file"):
            seen.add(k)
            continue
        seen.add(k)
        if isinstance(result[k], (dict, list, str)):
            out.write("extras " + k + " " + json.dumps(result[k]) + "\n")
    out.write(json.dumps(line) + "\n")
    out.flush()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=65536, help="8192-sample blocks per GPU per pass")
    ap.add_argument("--passes-per-step", type=int, default=0,
                    help="passes of the chain over the batch that make one step (0 = auto: ~50 ms of GPU time per step, so that "
                         "the default 20 steps keep the GPU busy for ~1 s and an outside utilisation sampler can see the run)")
    ap.add_argument("--passes-per-exchange", type=int, default=1,
                    help="N > 1 ranks: exchange the halos of this many consecutive super-blocks in ONE send/recv pair, then run that many "
                         "passes (sdrhip_fm_chain_halo_exchange_batch): a launch-bound shard cannot hide a point-to-point round trip per pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the contract line (profiling runs)")
    ap.add_argument("--cpu-worker", type=float, default=None, help="only the compiled CPU receiver loop, single thread, for this many seconds")
    args = ap.parse_args()

    if args.cpu_worker is not None:
        exe, taps = _cpu_harness()
        try:
            r = _cpu_run(exe, taps, args.cpu_worker, 1)
        finally:
            os.unlink(taps)
        print(json.dumps({"sps": r["sps_total"], "kind": r["kind"]}))
        return

    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU), exactly
        # the command line the driver uses, and hand back its exit code -- a plain `--gpus 8` must never run as one rank.
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report n_gpus != --gpus\n")
        sys.exit(2)
    if os.environ.get("BENCH_PLUMBING") == "1":
        plumbing_check(args, rank, world)
        return

    # CPU baseline first (rank 0, N=1 only), before the GPU is touched
    cpu = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch
    import torch.distributed as dist
    from sdr_amd import build as _build
    if not os.path.exists(_build.LIB) and local_rank == 0:
        _build.build()               # a fresh checkout: compile the HIP library (hipcc is in the image)
    for _ in range(600):
        if os.path.exists(_build.LIB):
            break
        time.sleep(0.5)
    import sdr_amd.lib as L          # raises if libsdr_hip.so is missing: there is no CPU fallback
    import signals as S
    from sdr_amd import sharding

    # Control plane (rendezvous of the RCCL id, barriers, max-over-ranks of the timing): torch.distributed over gloo.
    # Data plane (the halo exchange): the library's own RCCL point-to-point, sdrhip_fm_chain_halo_exchange (comm.cpp).
    # BENCH_TRANSPORT=host is a plumbing check only (several ranks may then share one GPU; the halo travels through gloo).
    transport = os.environ.get("BENCH_TRANSPORT", "rccl")
    ndev = torch.cuda.device_count()
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    L.check(L.lib.sdrhip_set_device(dev), "sdrhip_set_device")
    comm = None
    rccl_hung = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if transport == "rccl":
            ok = 1
            try:
                box = [L.comm_unique_id() if rank == 0 else None]
            except Exception as e:                   # noqa: BLE001
                sys.stderr.write(f"bench: RCCL unavailable on rank 0 ({e!r})\n")
                box, ok = [None], 0
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                # ncclCommInitRank is a collective that cannot be cancelled: run it beside a watchdog, so that a rendezvous that
                # never completes (a mis-set environment on the box) costs two minutes and the host-memory fallback, not the run
                import threading
                res = {}

                def _init():
                    try:
                        L.check(L.lib.sdrhip_set_device(dev), "sdrhip_set_device")
                        res["comm"] = L.Comm(world, rank, box[0])
                    except Exception as e:           # noqa: BLE001 -- SURVEY 8(e) fallback: same halos through host memory
                        res["err"] = e

                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(float(os.environ.get("BENCH_RCCL_INIT_TIMEOUT", "120")))
                if th.is_alive():
                    sys.stderr.write(f"bench: sdrhip_comm_init_rank did not return on rank {rank} within the time limit\n")
                    ok = 0
                    rccl_hung = True
                elif "err" in res:
                    sys.stderr.write(f"bench: sdrhip_comm_init_rank failed on rank {rank} ({res['err']!r})\n")
                    ok = 0
                else:
                    comm = res["comm"]
            flag = torch.tensor([ok if box[0] is not None else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not bool(flag.item()):
                comm, transport = None, "host"
                if rank == 0:
                    sys.stderr.write("bench: falling back to the halo exchange through host memory (gloo)\n")

    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    def measure(blocks, steps, warmup, passes, ramp_s, timing, graph=False, do_exchange=True, data="uniform", lib_overlap=False, ppe=1):
        """One configuration: `blocks` 8192-sample blocks per GPU per pass.  Returns the max-over-ranks wall time of
        `steps` steps of `passes` passes each, the per-stage HIP-event times and the plan.  graph: the chain's kernels of
        one pass replayed from a hipGraph captured once (sdrhip_fm_chain_graph_*): one launch per pass instead of one per kernel."""
        S_len = blocks * BLOCK
        plan = sharding.ShardPlan(chain, rank, world, S_len)          # owned outputs + halo for this rank
        gen = torch.Generator(device="cuda").manual_seed(S.SEED_IQ + rank)
        if data == "fm":
            # what an FM receiver is fed: a frequency-modulated carrier (1 kHz tone, 25 kHz deviation at 1.28 MS/s, tests/
            # signals.py:iq_u8_fm) + a little noise, quantised to u8 IQ -- generated on the device in slices
            buf = torch.empty(2 * (S_len + plan.halo_cap), dtype=torch.uint8, device="cuda")
            nall, step = S_len + plan.halo_cap, 1 << 24
            for a in range(0, nall, step):
                b = min(nall, a + step)
                t = torch.arange(a, b, device="cuda", dtype=torch.float64) / 1.28e6
                ph = (25.0 * torch.sin(2 * 3.141592653589793 * 1e3 * t)).to(torch.float32)
                nz = 0.02 * torch.randn(2, b - a, device="cuda", generator=gen)
                buf[2 * a:2 * b:2] = torch.clamp(torch.round((0.8 * torch.cos(ph) + nz[0]) * 127.5 + 127.5), 0, 255).to(torch.uint8)
                buf[2 * a + 1:2 * b:2] = torch.clamp(torch.round((0.8 * torch.sin(ph) + nz[1]) * 127.5 + 127.5), 0, 255).to(torch.uint8)
            del t, ph, nz
        else:
            buf = torch.randint(0, 256, (2 * (S_len + plan.halo_cap),), dtype=torch.uint8, device="cuda", generator=gen)
        # ppe > 1 (N > 1): the rank's shards of ppe consecutive super-blocks as rows of one buffer; their halos travel in one message pair
        ppe = ppe if (world > 1 and not graph and data == "uniform") else 1
        rows = staging = None
        row_bytes = 2 * (S_len + plan.halo_cap)
        if ppe > 1:
            rows = torch.randint(0, 256, (ppe, row_bytes), dtype=torch.uint8, device="cuda", generator=gen)
            staging = torch.empty(max(1, chain.halo_staging_bytes(ppe)), dtype=torch.uint8, device="cuda")
            buf = rows[0]
        pidx = [0]
        audio = torch.empty(plan.q1 - plan.q0, dtype=torch.float32, device="cuda")
        # lib_overlap (N = 1): two passes in flight INSIDE the library (sdrhip_fm_chain_set_overlap): consecutive runs alternate
        # between two internal streams and workspace halves; the audio is double-buffered as that contract asks
        lib_overlap = lib_overlap and world == 1 and not graph
        if lib_overlap:
            chain.set_overlap(True)
            audio_b = torch.empty_like(audio)
            # the contract of two runs in flight: input double-buffered like the audio (sdr_hip.h) -- a second batch of its own
            buf_b = buf.clone()         # (measured: no different from both runs reading one buffer, 558.9 against 558.4 Gsample/s)
        ws_bytes = chain.workspace_bytes(S_len + plan.halo_cap)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        flip = [0]
        # N > 1: the halo (the right neighbour's first ~4k samples) travels while this rank already computes the outputs
        # that need only its own samples, [q0, q_mid); the few that reach into the halo, [q_mid, q1), run on a second stream
        # as soon as it has landed.  Two chain objects: each keeps its own timing events.
        overlap = world > 1 and plan.q_mid > plan.q0 and os.environ.get("BENCH_NO_OVERLAP") != "1"
        if overlap:
            chain_b = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
            aux = torch.cuda.Stream()
            ws_b = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")

        def exchange(on_stream):
            if not do_exchange:          # "replicas only": the same passes with the halo left as it is (upper bound of the scaling)
                return
            if ppe > 1:
                if pidx[0] % ppe != 0:   # the halos of this super-block arrived with the batch
                    return
                if comm is not None:
                    comm.chain_halo_exchange_batch(chain, rows.data_ptr(), S_len, row_bytes, ppe, staging.data_ptr(), stream=on_stream.cuda_stream)
                else:
                    sharding.halo_exchange_batch(rows, plan, dist, via_host=True)
                return
            if comm is not None:
                comm.chain_halo_exchange(chain, buf.data_ptr(), S_len, stream=on_stream.cuda_stream)   # ncclSend/ncclRecv on that stream
            else:
                sharding.halo_exchange(buf, plan, dist, via_host=True)

        g_all = g_a = g_b = None
        if graph:
            if not overlap:
                g_all = L.FmGraph(chain, buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q1, ws.data_ptr(), ws_bytes)
            else:
                g_a = L.FmGraph(chain, buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q_mid, ws.data_ptr(), ws_bytes)
                if plan.q1 > plan.q_mid:
                    g_b = L.FmGraph(chain_b, buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr() + 4 * (plan.q_mid - plan.q0), plan.q_mid,
                                    plan.q1, ws_b.data_ptr(), ws_bytes)

        def one_pass():
            in_ptr = buf.data_ptr() if ppe == 1 else rows.data_ptr() + (pidx[0] % ppe) * row_bytes
            try:
                _one_pass(in_ptr)
            finally:
                pidx[0] += 1

        def _one_pass(in_ptr):
            if lib_overlap:
                flip[0] ^= 1
                chain.run((buf_b if flip[0] else buf).data_ptr(), plan.s0, plan.n_in, (audio_b if flip[0] else audio).data_ptr(), plan.q0, plan.q1, ws.data_ptr(), ws_bytes, stream=sptr)
                return
            if not overlap:
                if world > 1:
                    exchange(stream)
                if g_all is not None:
                    g_all.launch(sptr)
                else:
                    chain.run(in_ptr, plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q1, ws.data_ptr(), ws_bytes, stream=sptr)
                return
            aux.wait_stream(stream)                      # the previous pass's readers of the halo region are done
            with torch.cuda.stream(aux):
                exchange(aux)
                if plan.q1 > plan.q_mid:
                    if g_b is not None:
                        g_b.launch(aux.cuda_stream)
                    else:
                        chain_b.run(in_ptr, plan.s0, plan.n_in, audio.data_ptr() + 4 * (plan.q_mid - plan.q0), plan.q_mid, plan.q1,
                                    ws_b.data_ptr(), ws_bytes, stream=aux.cuda_stream)
            if g_a is not None:
                g_a.launch(sptr)
            else:
                chain.run(in_ptr, plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q_mid, ws.data_ptr(), ws_bytes, stream=sptr)
            stream.wait_stream(aux)

        # Clock / power-state ramp: the first ~15 ms of sustained work of a fresh process run 10 % slow; spin for ramp_s first.
        # With several ranks every pass is a send/recv with the neighbours, so all ranks must run the SAME number of passes: the
        # decision to go on is taken collectively.
        t_ramp = time.perf_counter()
        while True:
            go = time.perf_counter() - t_ramp < ramp_s
            if world > 1:
                flag = torch.tensor([1 if go else 0], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                go = bool(flag.item())
            if not go:
                break
            one_pass()
            torch.cuda.synchronize()
        if passes <= 0:
            # auto: ~50 ms of GPU time per step, measured on this configuration (the same on every rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                one_pass()
            torch.cuda.synchronize()
            per = (time.perf_counter() - t0) / 4
            pt = torch.tensor([max(1, min(2000, int(round(0.05 / max(per, 1e-6)))))], dtype=torch.int32)
            if world > 1:
                dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            passes = int(pt.item())
        for _ in range(warmup):
            for _ in range(passes):
                one_pass()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if timing:
            chain.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            for _ in range(passes):
                one_pass()
        if lib_overlap:
            chain.join(sptr)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0           # this rank's own work, before it waits for the others
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        same_audio = None
        if lib_overlap:
            same_audio = bool(torch.equal(audio.view(torch.int32), audio_b.view(torch.int32)))
            chain.set_overlap(False)
            del audio_b, buf_b
        stage_ms, runs = ({}, 0)
        if timing:
            stage_ms, runs = chain.read_timing()
            chain.enable_timing(False)
        rank_elapsed = [own_elapsed]
        if world > 1:
            rank_elapsed = [None] * world
            dist.all_gather_object(rank_elapsed, own_elapsed)
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        crc = None
        if os.environ.get("BENCH_CHECKSUM") == "1":      # plumbing checks: the same audio whichever way the pass is scheduled
            import zlib
            crc = [zlib.crc32(audio.cpu().numpy().tobytes())]
            if world > 1:
                gathered = [None] * world
                dist.all_gather_object(gathered, crc[0])
                crc = gathered
        # what one exchange costs on its own: back-to-back exchanges on the compute stream, nothing else queued (N > 1, RCCL only)
        exch_us = None
        if world > 1 and do_exchange and comm is not None:
            pidx[0] = 0
            torch.cuda.synchronize()
            dist.barrier()
            nex = 50
            tq = time.perf_counter()
            for _ in range(nex):
                pidx[0] = 0
                exchange(stream)
            torch.cuda.synchronize()
            te = torch.tensor([(time.perf_counter() - tq) / nex * 1e6], dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            exch_us = float(te.item())
        del g_all, g_a, g_b
        del buf, audio, ws, rows, staging
        return {"elapsed": elapsed, "passes": passes, "stage_ms": stage_ms, "plan": plan, "S_len": S_len, "overlap": overlap, "crc": crc,
                "lib_overlap": lib_overlap, "same_audio": same_audio, "rank_elapsed": rank_elapsed, "ppe": ppe, "exchange_us": exch_us}

    def measure_in_flight(blocks, steps, warmup, nflight, do_exchange=True):
        """Launch-bound shards: `nflight` passes in flight, pass i on HIP stream i % nflight with its own input / audio
        buffers and chain object.  Consecutive passes of a sharded stream are independent of each other -- what a rank needs
        from elsewhere is RAW input (the right neighbour's head), never a result -- so a host may queue the next super-block
        while the GPU still drains the previous one; on one stream the ~4 us between the end of one kernel and the start of
        the next are idle.  Each pass = halo exchange (N > 1) + the chain on that pass's stream.  Returns
        (max-over-ranks seconds, passes per step)."""
        S_len = blocks * BLOCK
        plan = sharding.ShardPlan(chain, rank, world, S_len)
        gen = torch.Generator(device="cuda").manual_seed(S.SEED_IQ + rank)
        ws_bytes = chain.workspace_bytes(S_len + plan.halo_cap)
        lanes = []
        buf0 = torch.randint(0, 256, (2 * (S_len + plan.halo_cap),), dtype=torch.uint8, device="cuda", generator=gen)
        for _ in range(nflight):
            lanes.append({
                "chain": L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK),
                "stream": torch.cuda.Stream(),
                "buf": buf0.clone(),
                "audio": torch.empty(plan.q1 - plan.q0, dtype=torch.float32, device="cuda"),
                "ws": torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")})
        torch.cuda.synchronize()

        def one_pass(i):
            ln = lanes[i % nflight]
            if world > 1 and do_exchange:
                if comm is not None:
                    comm.chain_halo_exchange(ln["chain"], ln["buf"].data_ptr(), S_len, stream=ln["stream"].cuda_stream)
                else:
                    ln["stream"].synchronize()
                    sharding.halo_exchange(ln["buf"], plan, dist, via_host=True)
            ln["chain"].run(ln["buf"].data_ptr(), plan.s0, plan.n_in, ln["audio"].data_ptr(), plan.q0, plan.q1, ln["ws"].data_ptr(), ws_bytes,
                            stream=ln["stream"].cuda_stream)

        for i in range(64):
            one_pass(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(256):
            one_pass(i)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 256
        pt = torch.tensor([max(1, min(4000, int(round(0.03 / max(per, 1e-6)))))], dtype=torch.int32)
        if world > 1:
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        passes = int(pt.item())
        for i in range(warmup * passes):
            one_pass(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps * passes):
            one_pass(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        same = all(torch.equal(ln["audio"], lanes[0]["audio"]) for ln in lanes) if world == 1 else None
        del lanes
        return elapsed, passes, S_len, same

    dbg = (lambda m: sys.stderr.write(f"bench[{rank}]: {m}\n")) if os.environ.get("BENCH_DEBUG") else (lambda m: None)
    extras = not args.no_extras

    def k2c_time(x, reps=10, warm=5):
        """The cfloat-in decimate-by-8 kernel (+ seam fix-up) on 2^27 samples: mean launch time by HIP events on its stream."""
        n1 = x.numel() // 2
        k1 = (n1 - 128) // 8 + 1
        dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
        o1 = torch.empty(2 * k1 + 64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warm):
            dec.run(x.data_ptr(), 0, o1.data_ptr(), 0, k1, BLOCK, stream=sptr)
        e0.record(stream)
        for _ in range(reps):
            dec.run(x.data_ptr(), 0, o1.data_ptr(), 0, k1, BLOCK, stream=sptr)
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    # BASELINE configs[1] (the north_star's roofline kernel): the same decimate-by-8 kernel fed cfloat IQ (8 B read + 1 B
    # written per input sample), device-resident, 8192-sample seams, as a workload of its own: warmed up by ~0.1 s of its
    # own launches (the first ~100 launches of a fresh process run 15-25 % slow while the clocks ramp), then timed over
    # ~0.25 s of back-to-back launches -- next to what the memory system of THIS box delivers in THIS process, in the same
    # state, for the same traffic shape.  (The same kernel is timed again right after the sustained chain run below.)
    cfg1 = None
    if rank == 0 and world == 1 and extras:
        n1 = 1 << 27
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps, warm):
            for _ in range(warm):
                fn()
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        def k2c_fields(t):
            return {"avg_launch_ms": round(t * 1e3, 5), "Msamples_per_s": round(n1 / t / 1e6, 1),
                    "achieved": round(9.0 * n1 / t / 1e9, 1), "frac": round(9.0 * n1 / t / 1e9 / HBM_PEAK_GBS, 4),
                    "read_only_frac": round(8.0 * n1 / t / 1e9 / HBM_PEAK_GBS, 4)}

        x_uni = torch.rand(2 * n1, device="cuda") * 2 - 1
        t_uni = k2c_time(x_uni, reps=1000, warm=400)
        dbg("cfg1 kernel done")
        sout = torch.empty(n1 // 4 + 64, device="cuda")
        t_plain = timed(lambda: L.check(L.lib.sdrhip_bench_stream_8to1(sptr, x_uni.data_ptr(), sout.data_ptr(), 8 * n1, 0)), 300, 50)
        t_nt = timed(lambda: L.check(L.lib.sdrhip_bench_stream_8to1(sptr, x_uni.data_ptr(), sout.data_ptr(), 8 * n1, 1)), 300, 50)
        half = (9 * n1 // 2) // 16 * 16
        src_c = x_uni.view(torch.uint8)[:half]
        dst_c = torch.empty(half, dtype=torch.uint8, device="cuda")
        t_copy = timed(lambda: L.check(L.lib.sdrhip_bench_copy2(sptr, src_c.data_ptr(), dst_c.data_ptr(), half, 0)), 300, 50)
        t_copy_nt = timed(lambda: L.check(L.lib.sdrhip_bench_copy2(sptr, src_c.data_ptr(), dst_c.data_ptr(), half, 1)), 300, 50)
        del dst_c, src_c, sout
        dbg("cfg1 ceilings done")
        # the data the FM pipeline actually feeds this stage: convert(u8 IQ), i.e. cfloat values k/128
        x_u8 = (torch.randint(0, 256, (2 * n1,), device="cuda", dtype=torch.uint8).to(torch.float32) - 128.0) * (1.0 / 128.0)
        t_u8d = k2c_time(x_u8, reps=600, warm=100)
        # the same launches alternating between two HIP streams (separate outputs): a stream of independent buffers keeps two
        # launches in flight, and the seam fix-up, the launch gaps and the ramps of one hide behind the tile kernel of the other
        k2c_streams = [torch.cuda.Stream(), torch.cuda.Stream()]

        def k2c_two_streams(x, reps, rounds=3):
            """(one-stream, two-stream) wall seconds per launch, alternating rounds (a row's place in a run moves it by +-5 %)"""
            k1 = (n1 - 128) // 8 + 1
            dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
            sts = k2c_streams              # the same two streams for both inputs: HIP deals streams onto a few hardware queues in
                                           # creation order, and a later pair may share one
            os_ = [torch.empty(2 * k1 + 64, device="cuda") for _ in range(2)]
            def go(ns, nr):
                for i in range(nr):
                    j = i % ns
                    dec.run(x.data_ptr(), 0, os_[j].data_ptr(), 0, k1, BLOCK, stream=sts[j].cuda_stream)
            go(1, 100)
            torch.cuda.synchronize()
            acc = [0.0, 0.0]
            for _ in range(rounds):
                for ns in (1, 2):
                    go(ns, 30)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    go(ns, reps)
                    torch.cuda.synchronize()
                    acc[ns - 1] += (time.perf_counter() - t0) / reps
            return acc[0] / rounds, acc[1] / rounds
        try:
            u1, u2 = k2c_two_streams(x_uni, 200)
            c1, c2 = k2c_two_streams(x_u8, 200)
            two_flight = {"input_uniform": {"one_stream_wall": k2c_fields(u1), "two_streams": k2c_fields(u2)},
                          "input_convert_u8": {"one_stream_wall": k2c_fields(c1), "two_streams": k2c_fields(c2)},
                          "what": "launches alternating between two HIP streams (separate outputs) against the same launches on one stream, "
                                  "wall clock per launch, three alternating rounds of 200 launches each: what a caller with a queue of "
                                  "independent buffers gets per buffer -- the seam fix-up, the launch gaps and the ramps of one launch run "
                                  "beside the tile kernel of the other (tools/k2c_two_streams.py)"}
        except Exception as e:                          # noqa: BLE001
            two_flight = f"failed: {e!r}"
        del x_u8
        cfg1 = {"kernel": "k_decimate_systolic<cfloat> + seam fix-up", "samples_per_launch": n1, "bound": "hbm", "peak": HBM_PEAK_GBS,
                "unit": "GB/s", **k2c_fields(t_uni), "input": "uniform [-1,1) f32",
                "when": "sustained: 1000 back-to-back launches after 400 warm-up launches of the same kernel, before anything else ran on the GPU",
                "input_convert_u8": {**k2c_fields(t_u8d), "input": "convert(u8 IQ): the values the FM pipeline feeds this stage (600 launches)"},
                "two_launches_in_flight": two_flight,
                "ceilings_same_process": {
                    "stream_8to1_plain_loads": {"ms": round(t_plain * 1e3, 5), "read_only_frac": round(8.0 * n1 / t_plain / 1e9 / HBM_PEAK_GBS, 4)},
                    "stream_8to1_nontemporal_loads": {"ms": round(t_nt * 1e3, 5), "read_only_frac": round(8.0 * n1 / t_nt / 1e9 / HBM_PEAK_GBS, 4)},
                    "copy_same_total_bytes": {"ms": round(t_copy * 1e3, 5), "total_GBps": round(2.0 * half / t_copy / 1e9, 1)},
                    "copy_same_total_bytes_nontemporal": {"ms": round(t_copy_nt * 1e3, 5), "total_GBps": round(2.0 * half / t_copy_nt / 1e9, 1)},
                    "kernel_over_nt_stream": round(t_nt / t_uni, 4),
                    "kernel_over_plain_stream": round(t_plain / t_uni, 4), "kernel_over_copy": round(t_copy / t_uni, 4),
                    "kernel_over_nt_copy": round(t_copy_nt / t_uni, 4),
                    "what": "measured right after the kernel (300 launches each), same process and power state: kernels that read the same "
                            "1 GiB with 16-byte loads and write 128 MiB with no arithmetic (8 loads in flight per thread, plain and "
                            "non-temporal), and one-shot copies moving the same 1.125 GiB in total (4 x 16 B in flight per thread, plain "
                            "and non-temporal).  kernel_over_nt_stream is the headline ratio: the decimator against the BEST the memory "
                            "system gives this traffic shape.  The decimator is power-limited (DESIGN.md 5): its MAC phase alone takes "
                            "~0.16-0.18 ms at a shader clock of ~1.7-1.9 GHz, and a row's place in a run moves it by +-5 % "
                            "(profiles/k2lab/run5.txt, ABAB rows)"}}
        dbg("cfg1 done")
    # per-stage HIP events inside the timed region (ten event records per pass) when a pass is long enough not to notice them;
    # a launch-bound shard (e.g. --blocks 128, BASELINE configs[4]) is timed without them and its stage times come from a
    # short separate run of the same passes
    # N = 1: `value` is measured with two passes in flight inside the library (sdrhip_fm_chain_set_overlap; what a streaming caller
    # gets for one flag) and stage_ms / roofline come from a one-pass-at-a-time run of the same passes right after it, so that
    # per-kernel durations are not blurred by co-resident kernels (BENCH_NO_LIB_OVERLAP=1: one pass at a time throughout).
    want_overlap = world == 1 and os.environ.get("BENCH_NO_LIB_OVERLAP") != "1" and args.blocks >= 2048
    events_in_region = args.blocks >= 2048 and not want_overlap
    main_run = measure(args.blocks, args.steps, args.warmup, args.passes_per_step, 0.3, events_in_region, lib_overlap=want_overlap, ppe=max(1, args.passes_per_exchange))
    single_pass = None
    if want_overlap:
        single_pass = measure(args.blocks, args.steps, 1, main_run["passes"], 0.05, True)
        main_run["stage_ms"] = single_pass["stage_ms"]
    elif not events_in_region:
        main_run["stage_ms"] = measure(args.blocks, max(2, args.steps // 4), 1, main_run["passes"], 0.05, True)["stage_ms"]
    dbg("main measurement done")
    # the same run on a frequency-modulated carrier instead of uniform random bytes (the chip is power-limited and the power
    # of a multiply depends on its operands: random bytes are the most expensive input there is)
    fm_input = None
    if extras:
        stf = max(2, args.steps // 4)
        rf = measure(args.blocks, stf, 1, main_run["passes"], 0.1, False, data="fm")
        fm_input = {"value": round(world * rf["S_len"] * rf["passes"] * stf / rf["elapsed"] / 1e6, 1), "unit": "Msamples/s",
                    "input": "u8 IQ of an FM carrier (1 kHz tone, 25 kHz deviation at 1.28 MS/s, amplitude 0.8 of full scale) + noise: "
                             "what the receiver is fed; the headline `value` is measured on uniform random bytes"}
        dbg("fm input done")

    # The reference example's OWN filters (examples/fm/Coeffs.hs as data: tests/golden/example_taps.npz -- 51 / 31 / 64 taps, SURVEY 8(d)'s
    # second tap set): the receiver examples/fm/fm.hs actually runs, on the same batch.  Fewer multiply-adds per sample than the
    # benchmark's 127 / 191 / 128 taps, other kernels (the 52-tap tile decimator, the 16-float-group resampler, the stand-alone fmDemod).
    example_taps = None
    if extras and rank == 0 and world == 1:
        try:
            chx = L.FmChain(8, S.taps_example_rf_decim(), 3, 10, S.taps_example_audio_resampler(), S.taps_example_audio_filter_half(), gain=0.2, block=BLOCK)
            S_x = args.blocks * BLOCK
            plx = sharding.ShardPlan(chx, 0, 1, S_x)
            bufx = torch.randint(0, 256, (2 * (S_x + plx.halo_cap),), dtype=torch.uint8, device="cuda")
            audx = torch.empty(plx.q1 - plx.q0, dtype=torch.float32, device="cuda")
            wsbx = chx.workspace_bytes(S_x + plx.halo_cap)
            wsx = torch.empty(wsbx, dtype=torch.uint8, device="cuda")
            runx = lambda: chx.run(bufx.data_ptr(), plx.s0, plx.n_in, audx.data_ptr(), plx.q0, plx.q1, wsx.data_ptr(), wsbx, stream=sptr)
            for _ in range(20):
                runx()
            torch.cuda.synchronize()
            nx = max(10, main_run["passes"] * max(2, args.steps // 4))
            t0 = time.perf_counter()
            for _ in range(nx):
                runx()
            torch.cuda.synchronize()
            tx = (time.perf_counter() - t0) / nx
            chx.enable_timing(True)
            for _ in range(10):
                runx()
            torch.cuda.synchronize()
            msx, _ = chx.read_timing()
            chx.enable_timing(False)
            example_taps = {"value": round(S_x / tx / 1e6, 1), "unit": "Msamples/s", "ms_per_pass": round(tx * 1e3, 4),
                            "stage_ms": {k: round(v, 5) for k, v in msx.items() if v},
                            "hbm_read_GBps_of_the_u8_input": round(2.0 * S_x / tx / 1e9, 1),
                            "what": "the same pass with the reference FM example's own tap tables (examples/fm/Coeffs.hs:11-154 as data: 51-tap RF decimator, "
                                    "31-tap 3/10 resampler, 64-tap symmetric audio filter), one pass at a time; bit-exact vs the restated Pipes "
                                    "(tests/test_gpu_fullsize.py::test_example_real_taps_chain)"}
            del bufx, audx, wsx
        except Exception as e:                          # noqa: BLE001
            example_taps = f"failed: {e!r}"
        dbg("example taps done")

    # The main workload with two passes in flight (two chain objects, workspaces and audio buffers on two HIP streams): what a
    # host that keeps the GPU fed with independent batches gets.  Reported beside `value`, which stays the one-stream figure
    # whose per-kernel durations (roofline, stage_ms, the rocprofv3 summaries) are not blurred by co-resident kernels.
    main_two = None
    if extras:
        try:
            stm = max(2, args.steps // 4)
            elm, pm2, slm, samem = measure_in_flight(args.blocks, stm, 1, 2)
            main_two = {"value": round(world * slm * pm2 * stm / elm / 1e6, 1), "unit": "Msamples/s", "ms_per_pass": round(elm / (pm2 * stm) * 1e3, 4),
                        "what": "the same passes queued on two HIP streams in turn (separate chain objects, input, workspace and audio "
                                "buffers): the memory-heavy tail kernels of one pass run beside the VALU-bound decimator of the other"
                                + ("" if samem is None else f"; both streams' audio identical: {samem}")}
        except Exception as e:                          # noqa: BLE001
            main_two = f"failed: {e!r}"
        dbg("two passes in flight done")

    # BASELINE configs[4]'s shard size: 2^20 samples (128 blocks) per GPU per pass -- launch/latency-bound, the case where the
    # halo exchange matters; reported next to the main line, same run
    shard_1m = None
    if extras and args.blocks != 128:
        st1 = max(2, args.steps // 4)
        r1 = measure(128, st1, 1, 0, 0.05, False, ppe=max(1, args.passes_per_exchange))
        shard_1m = {"samples_per_gpu_per_pass": r1["S_len"], "passes_per_step": r1["passes"],
                    "value": round(world * r1["S_len"] * r1["passes"] * st1 / r1["elapsed"] / 1e6, 1), "unit": "Msamples/s",
                    "us_per_pass": round(r1["elapsed"] / (r1["passes"] * st1) * 1e6, 2),
                    "passes_per_exchange": r1["ppe"], "exchange_us_per_pass": None if r1["exchange_us"] is None else round(r1["exchange_us"] / r1["ppe"], 2),
                    "note": "BASELINE configs[4] shard size (1M-sample block per GPU per pass): launch/latency-bound"}
        if world > 1:
            # configs[4] is quoted on THIS shard: answer it at the K the design recommends (16 super-blocks per exchange) as
            # well as at K = 1, whatever --passes-per-exchange the run was started with, so one 8-GPU run of the driver's command suffices
            for kk in (1, 16):
                if kk == r1["ppe"]:
                    rk = r1
                else:
                    rk = measure(128, st1, 1, 0, 0.05, False, ppe=kk)
                shard_1m[f"passes_per_exchange_{kk}"] = {
                    "value": round(world * rk["S_len"] * rk["passes"] * st1 / rk["elapsed"] / 1e6, 1),
                    "us_per_pass": round(rk["elapsed"] / (rk["passes"] * st1) * 1e6, 2),
                    "exchange_us_per_pass": None if rk["exchange_us"] is None else round(rk["exchange_us"] / rk["ppe"], 2)}
        try:
            el2, p2, sl2, same2 = measure_in_flight(128, st1, 1, 2)
            shard_1m["two_passes_in_flight"] = {
                "value": round(world * sl2 * p2 * st1 / el2 / 1e6, 1), "us_per_pass": round(el2 / (p2 * st1) * 1e6, 2),
                "what": "the same passes queued on two HIP streams in turn (separate input / audio buffers): consecutive passes of a "
                        "sharded stream depend only on raw input, so the ~4 us between two kernels of one stream need not be idle"
                        + ("" if same2 is None else f"; both streams' audio identical: {same2}")}
        except Exception as e:                          # noqa: BLE001
            shard_1m["two_passes_in_flight"] = f"failed: {e!r}"
        try:
            r2 = measure(128, st1, 1, 0, 0.05, False, graph=True)
            shard_1m["hipgraph"] = {"value": round(world * r2["S_len"] * r2["passes"] * st1 / r2["elapsed"] / 1e6, 1),
                                    "us_per_pass": round(r2["elapsed"] / (r2["passes"] * st1) * 1e6, 2),
                                    "what": "the same pass with the chain's kernels replayed from a hipGraph captured once "
                                            "(sdrhip_fm_chain_graph_*): one graph launch instead of one launch per kernel" + ("; the halo exchange stays outside the graph" if world > 1 else "")}
        except Exception as e:                          # noqa: BLE001
            shard_1m["hipgraph"] = f"failed: {e!r}"

    # N > 1: what the exchange costs -- the same passes with the halo exchange left out ("replicas only" as the
    # trivially-parallel upper bound)
    replicas = None
    if extras and world > 1:
        st1 = max(2, args.steps // 4)
        rm = measure(args.blocks, st1, 1, main_run["passes"], 0.05, False, do_exchange=False)
        replicas = {"what": "the same shards and passes without the halo exchange (independent replicas): the upper bound of the scaling",
                    "value": round(world * rm["S_len"] * rm["passes"] * st1 / rm["elapsed"] / 1e6, 1), "unit": "Msamples/s"}
        if args.blocks != 128:
            rs = measure(128, st1, 1, 0, 0.05, False, do_exchange=False)
            replicas["shard_1M_samples_per_gpu"] = {"value": round(world * rs["S_len"] * rs["passes"] * st1 / rs["elapsed"] / 1e6, 1),
                                                    "us_per_pass": round(rs["elapsed"] / (rs["passes"] * st1) * 1e6, 2)}
            if shard_1m is not None and replicas["shard_1M_samples_per_gpu"]["value"]:
                rv = replicas["shard_1M_samples_per_gpu"]["value"]
                shard_1m["scaling_efficiency"] = round(shard_1m["value"] / rv, 4)
                for kk in (1, 16):
                    if f"passes_per_exchange_{kk}" in shard_1m:
                        shard_1m[f"passes_per_exchange_{kk}"]["scaling_efficiency"] = round(shard_1m[f"passes_per_exchange_{kk}"]["value"] / rv, 4)
    dbg("shard_1m done")
    # configs[1] kernel again, right after the sustained chain run (the u8-fused chain is the hotter workload: the clock the
    # chip grants afterwards is lower)
    if cfg1 is not None:
        t_after = k2c_time(x_uni, reps=20, warm=5)
        del x_uni
        cfg1["after_chain_run"] = {"avg_launch_ms": round(t_after * 1e3, 5), "read_only_frac": round(8.0 * (1 << 27) / t_after / 1e9 / HBM_PEAK_GBS, 4),
                                   "frac": round(9.0 * (1 << 27) / t_after / 1e9 / HBM_PEAK_GBS, 4),
                                   "when": "same kernel and input, 20 launches right after the chain measurement of this process"}
        dbg("cfg1 after-chain done")

    # Socket power, power cap and shader clock behind the "power-limited" reading of the kernels (VERDICT r03 "next" #2): each
    # row keeps the GPU busy with ONE kind of launch for ~2.5 s while a sampler thread reads the amdgpu hwmon files of THIS
    # device (tools/power_probe.py: power1_input, freq1_input, power1_cap); the reported power is filtered by the SMU with a
    # time constant of a few hundred ms, hence the long rows and the mean over their last 60 %.
    power = None
    if rank == 0 and world == 1 and extras and os.environ.get("BENCH_NO_POWER") != "1":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import power_probe as PP
            bdf = PP.device_bdf(dev)
            smp = PP.HwmonSampler(0.05, bdf)
            if not smp.available:
                power = {"available": False, "device_bdf": bdf, "note": "no hwmon directory for this device under /sys/bus/pci/devices"}
            else:
                def power_row(fn, per_launch_s, samples_per_launch, seconds=2.5):
                    torch.cuda.synchronize()
                    n_l = max(8, int(seconds / max(per_launch_s, 1e-6)))
                    smp.start()
                    t0 = time.perf_counter()
                    for _ in range(n_l):
                        fn()
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    st = PP.HwmonSampler.summarize(smp.stop(), t_lo=0.4 * dt)
                    row = {"seconds": round(dt, 2), "launches": n_l, "us_per_launch": round(dt / n_l * 1e6, 1)}
                    if samples_per_launch:
                        row["Gsamples_per_s"] = round(samples_per_launch * n_l / dt / 1e9, 1)
                    if st:
                        row.update({"mean_w": round(st["mean_w"], 1), "min_w": round(st["min_w"], 1), "max_w": round(st["max_w"], 1),
                                    "mean_sclk_mhz": round(st["mean_sclk_mhz"], 0) if st["mean_sclk_mhz"] else None, "telemetry_samples": st["samples"]})
                    return row
                smp.start()
                time.sleep(1.0)
                idle = PP.HwmonSampler.summarize(smp.stop())
                npw = 1 << 27
                kpw = (npw - 128) // 8 + 1
                decp = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
                xo = torch.empty(2 * kpw + 64, device="cuda")
                xu = torch.randint(0, 256, (2 * npw,), dtype=torch.uint8, device="cuda")
                rows = {}
                rows["k2_u8_decimator_2^27_samples"] = power_row(lambda: decp.run_u8(xu.data_ptr(), 0, xo.data_ptr(), 0, kpw, BLOCK, stream=sptr), 1.8e-4, npw)
                del xu
                xf = torch.rand(2 * npw, device="cuda") * 2 - 1
                rows["k2_cfloat_decimator_2^27_samples"] = power_row(lambda: decp.run(xf.data_ptr(), 0, xo.data_ptr(), 0, kpw, BLOCK, stream=sptr), 2.4e-4, npw)
                so = torch.empty(npw // 4 + 64, device="cuda")
                rows["stream_8to1_nontemporal_loads_2^27_samples"] = power_row(
                    lambda: L.check(L.lib.sdrhip_bench_stream_8to1(sptr, xf.data_ptr(), so.data_ptr(), 8 * npw, 1)), 2.0e-4, npw)
                del xf, so, xo
                # the whole chain, the main workload's pass
                S_p = args.blocks * BLOCK
                plan_p = sharding.ShardPlan(chain, 0, 1, S_p)
                buf_p = torch.randint(0, 256, (2 * (S_p + plan_p.halo_cap),), dtype=torch.uint8, device="cuda")
                aud_p = torch.empty(plan_p.q1 - plan_p.q0, dtype=torch.float32, device="cuda")
                wsb = chain.workspace_bytes(S_p + plan_p.halo_cap)
                ws_p = torch.empty(wsb, dtype=torch.uint8, device="cuda")
                rows["fm_chain_pass"] = power_row(lambda: chain.run(buf_p.data_ptr(), plan_p.s0, plan_p.n_in, aud_p.data_ptr(), plan_p.q0, plan_p.q1,
                                                                     ws_p.data_ptr(), wsb, stream=sptr), 1.05e-3 * args.blocks / 65536, S_p)
                del buf_p, aud_p, ws_p
                power = {"available": True, "device_bdf": bdf, "cap_w": smp.cap_watts(),
                         "gpu_idle_for_1s_after_the_earlier_measurements": {"mean_w": round(idle["mean_w"], 1), "mean_sclk_mhz": round(idle["mean_sclk_mhz"], 0),
                                                                            "note": "clocks still raised; a cold idle GPU reads 250-255 W at 95-160 MHz (profiles/k2lab/r04_power_series.txt)"} if idle else None,
                         "rows": rows,
                         "how": "tools/power_probe.py HwmonSampler (amdgpu hwmon power1_input / freq1_input / power1_cap of the device's own PCI "
                                "address, 50 ms period), mean over the last 60 % of each ~2.5 s row of back-to-back launches"}
        except Exception as e:                          # noqa: BLE001
            power = {"available": False, "error": repr(e)}
        dbg("power rows done")

    # Host-streamed operation (PCIe inclusive, never `value`): the C-ABI host-block operators at the reference's own block
    # sizes, timed by the library's C loops (a compiled caller's cost per push; tools/host_stream_native.py)
    dbg("cfg1 done")
    host = None
    if rank == 0 and world == 1 and extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import host_stream_native as H
        host = {"unit": "Msamples/s of input (PCIe inclusive)", "cpu_single_thread_chain": cpu["single_thread_value"] if cpu else None,
                "note": "zero-copy lines: the source writes the library's pinned staging buffer itself (a radio's DMA target), so they carry "
                        "no source-side copy; *_memcpy: the caller's block is copied in by the push, which is what the reference's host path "
                        "(and cpu_single_thread_chain) includes"}
        host["note_adaptive"] = ("pushes are submitted at once while the GPU keeps up and pile up in the pinned staging buffer while it is busy "
                                 "(sdrhip_fm_stream_set_adaptive, on by default): *_every_push_its_own_launch is the same run with that off")
        # The link ceiling of THIS process (SURVEY 8(d) "Host-link ceiling"): pinned hipMemcpyAsync both ways and a kernel reading
        # pinned host memory in place (what the zero-copy pushes do), each over 256 MiB, best of three
        link = None
        try:
            nb = 256 << 20
            hp = torch.empty(nb, dtype=torch.uint8).pin_memory()
            dv = torch.empty(nb, dtype=torch.uint8, device="cuda")
            so = torch.empty(nb // 8 // 4 + 64, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def gbps(fn, reps=4):
                best = 0.0
                for _ in range(3):
                    fn()
                    torch.cuda.synchronize()
                    e0.record(stream)
                    for _ in range(reps):
                        fn()
                    e1.record(stream)
                    torch.cuda.synchronize()
                    best = max(best, reps * nb / (e0.elapsed_time(e1) * 1e-3) / 1e9)
                return round(best, 1)
            link = {"pinned_h2d_GBps": gbps(lambda: dv.copy_(hp, non_blocking=True)),
                    "pinned_d2h_GBps": gbps(lambda: hp.copy_(dv, non_blocking=True)),
                    "kernel_reads_pinned_host_GBps": gbps(lambda: L.check(L.lib.sdrhip_bench_stream_8to1(sptr, hp.data_ptr(), so.data_ptr(), nb, 1))),
                    "what": "256 MiB each, best of three rounds of four, HIP events on the stream: hipMemcpyAsync from / to pinned memory, and a "
                            "kernel streaming pinned host memory through 16-byte loads (sdrhip_bench_stream_8to1) -- the ceiling of the zero-copy pushes"}
            del hp, dv, so
        except Exception as e:                          # noqa: BLE001
            link = {"error": repr(e)}
        host["link"] = link
        # bytes on the link per input sample: 2 B of u8 IQ up, 4 * 3/80 B of audio down
        LINK_B = 2.0 + 4.0 * 3.0 / 80.0
        dev_rate = None                                  # device-resident rate of the same chain: what `compute` is in the overlap figure

        def annotate(sps, zero_copy):
            d = {"Msamples_per_s": round(sps / 1e6, 1), "link_GBps": round(sps * LINK_B / 1e9, 2)}
            if isinstance(link, dict) and "pinned_h2d_GBps" in link:
                ceil_ = link["kernel_reads_pinned_host_GBps"] if zero_copy else link["pinned_h2d_GBps"]
                d["link_ceiling_GBps"] = ceil_
                d["frac"] = round(d["link_GBps"] / ceil_, 3) if ceil_ else None
            return d
        for name, bpp, pushes, zc, co in (("fm_stream_1_block_per_push", 1, 20000, True, 0), ("fm_stream_1_block_per_push_memcpy", 1, 20000, False, 0),
                                          ("fm_stream_1_block_per_push_memcpy_every_push_its_own_launch", 1, 4000, False, 1),
                                          ("fm_stream_16_blocks_per_push", 16, 1000, True, 0),
                                          ("fm_stream_4096_blocks_per_push_zero_copy", 4096, 12, True, 0),
                                          ("fm_stream_4096_blocks_per_push_memcpy", 4096, 12, False, 0)):
            try:
                sps, _ = H.fm_stream_rate(L, chain, bpp * BLOCK, pushes, zc, co)
                host[name] = round(sps / 1e6, 1)
                host.setdefault("link_roofline", {})[name] = annotate(sps, bpp * BLOCK <= 199 * BLOCK)
                dbg(f"host {name} done")
            except Exception as e:                      # noqa: BLE001
                host[name] = f"failed: {e!r}"
        host["link_roofline_what"] = ("per host-streamed line: the bytes it moves over the link per second (2 B of u8 IQ up + 0.15 B of audio down per "
                                      "input sample) against the ceiling measured above -- the in-place read rate for pushes the kernels read in place "
                                      "(up to 200 source blocks), the pinned hipMemcpyAsync rate for the ones that go through the copy engines.  Small "
                                      "pushes are bound by launch latency, not by the link: their frac says how far")
        # the double-buffered path (4096-block pushes: upload of push i over compute of i-1 over download of i-2): what overlaps
        try:
            big = host["link_roofline"].get("fm_stream_4096_blocks_per_push_memcpy")
            if big and isinstance(link, dict) and link.get("pinned_h2d_GBps"):
                n_push = 4096 * BLOCK
                wall = n_push / (big["Msamples_per_s"] * 1e6)
                copy_t = n_push * 2.0 / (link["pinned_h2d_GBps"] * 1e9)
                planb = sharding.ShardPlan(chain, 0, 1, n_push)
                bufb = torch.randint(0, 256, (2 * (n_push + planb.halo_cap),), dtype=torch.uint8, device="cuda")
                audb = torch.empty(planb.q1 - planb.q0, dtype=torch.float32, device="cuda")
                wsbb = chain.workspace_bytes(n_push + planb.halo_cap)
                wsb_ = torch.empty(wsbb, dtype=torch.uint8, device="cuda")
                runb = lambda: chain.run(bufb.data_ptr(), planb.s0, planb.n_in, audb.data_ptr(), planb.q0, planb.q1, wsb_.data_ptr(), wsbb, stream=sptr)
                for _ in range(5):
                    runb()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    runb()
                torch.cuda.synchronize()
                comp_t = (time.perf_counter() - t0) / 20
                del bufb, audb, wsb_
                host["overlap_efficiency_4096_block_pushes"] = {
                    "value": round(max(copy_t, comp_t) / wall, 3), "wall_ms_per_push": round(wall * 1e3, 3), "copy_ms": round(copy_t * 1e3, 3),
                    "compute_ms": round(comp_t * 1e3, 3),
                    "what": "max(copy, compute) / wall for the copying push of 4096 source blocks: copy = its 64 MiB at the measured pinned H2D rate, "
                            "compute = the same samples device-resident, wall = per push as measured (the caller's copy into the pinned staging "
                            "buffer -- split over SDRHIP_COPY_THREADS helper threads since round 5 -- is the third stage of that pipeline)"}
        except Exception as e:                          # noqa: BLE001
            host["overlap_efficiency_4096_block_pushes"] = f"failed: {e!r}"
        # latency, which is what a real-time 1.28 MS/s source cares about (examples/fm/fm.hs:24; the reference plays the audio through
        # pulse, Pulse.hs:28): one 8192-sample block per push, the time from the push call to the pop of the audio it completed
        try:
            host["push_to_audio_us"] = {
                "paced_1.28_MS_per_s": H.fm_stream_latency(L, chain, BLOCK, 300, pace_us=6400.0),
                "paced_1.28_MS_per_s_every_push_its_own_launch": H.fm_stream_latency(L, chain, BLOCK, 300, pace_us=6400.0, adaptive_off=True),
                "unpaced_adaptive": H.fm_stream_latency(L, chain, BLOCK, 4000),
                "unpaced_every_push_its_own_launch": H.fm_stream_latency(L, chain, BLOCK, 2000, adaptive_off=True),
                "what": "sdrhip_bench_fm_stream_latency: audio leaves in 256-sample blocks, each charged to the push that made it computable; "
                        "paced = one push every 6.4 ms with the consumer polling in between (the GPU is idle when a push arrives: the figure is "
                        "launch + run + the result's way back); unpaced = pushes back to back (results lag behind by the submissions in flight, "
                        "and adaptive submission trades latency for throughput: fm_stream_1_block_per_push above is THAT run's throughput)"}
        except Exception as e:                          # noqa: BLE001
            host["push_to_audio_us"] = f"failed: {e!r}"
        try:
            # BASELINE configs[3]: firResampler 3/10, 191 taps, 65 536-float blocks, streamed (async double-buffered): 4 B up + 1.2 B down per element
            res = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
            for zc in (True, False):
                pp = L.Pipe("resampler", res, BLOCK)
                eps = H.pipe_rate(L, pp.h, 65536, 1, BLOCK, 4000, zc)
                row = {"Melements_per_s": round(eps / 1e6, 1), "us_per_push": round(65536 / eps * 1e6, 2), "link_GBps": round(eps * 5.2 / 1e9, 2)}
                if isinstance(link, dict) and "pinned_h2d_GBps" in link:
                    # 65 536 floats = 256 KiB per push (<= pipes.cpp kDirectBytes): the kernels read the pinned staging buffer in place
                    # over PCIe either way; memcpy = the caller's block is copied into that buffer by the push, zero_copy = the source wrote it there
                    ceil_ = link["kernel_reads_pinned_host_GBps"]
                    row["link_ceiling_GBps"] = ceil_
                    row["frac"] = round(row["link_GBps"] / ceil_, 3) if ceil_ else None
                host.setdefault("link_roofline", {})["config3_firResampler_65536_float_blocks_" + ("zero_copy" if zc else "memcpy")] = row
                if zc:
                    host["config3_firResampler_pipe_65536_float_blocks_Melements_per_s"] = row["Melements_per_s"]
            dec8 = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
            pd = L.Pipe("decimator", dec8, BLOCK)
            host["config1_firDecimator_pipe_8192_cfloat_blocks_Melements_per_s"] = round(H.pipe_rate(L, pd.h, BLOCK, 2, BLOCK, 20000, True) / 1e6, 1)
        except Exception as e:                          # noqa: BLE001
            host["pipes"] = f"failed: {e!r}"

    launch_sweep = None
    if rank == 0 and world == 1 and extras and os.environ.get("BENCH_NO_SWEEP") != "1":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import launch_sweep as LS
            launch_sweep = LS.sweep(L, S, 0.08)
        except Exception as e:                          # noqa: BLE001
            launch_sweep = f"failed: {e!r}"
        dbg("launch sweep done")

    if rank == 0:
        plan, S_len, passes, stage_ms = main_run["plan"], main_run["S_len"], main_run["passes"], main_run["stage_ms"]
        elapsed = main_run["elapsed"]
        total_samples = world * S_len * passes * args.steps
        k2_s = stage_ms["decimate"] * 1e-3
        # samples the TIMED decimate launch consumed: with the overlapped halo exchange only the chain object that computes
        # [q0, q_mid) carries timing events, and it reads the rank's own shard only
        k2_n = plan.shard_len if main_run["overlap"] else plan.k2_samples
        k2_alg_bytes = 3.0 * k2_n                                 # u8-fused K2 = 2 B read + 1 B written per input sample
        k2_flops = 64.0 * k2_n                                    # 2*2*P/D unfused flop per input sample
        hbm_achieved = k2_alg_bytes / k2_s / 1e9 if k2_s > 0 else 0.0
        valu_achieved = k2_flops / k2_s / 1e12 if k2_s > 0 else 0.0
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                now = k2_source_sha256()
                if tj.get("kernels_fast_sha256") != now:
                    traffic_note = "profiles/k2_traffic.json was measured on another version of decimate_tile.hpp / kernels_fast.hip: re-run tools/profile_bench.sh"
                elif tj.get("samples_per_launch") != plan.k2_samples:
                    traffic_note = "profiles/k2_traffic.json was measured at another launch size"
                else:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        tail_ms = sum(stage_ms.get(k, 0.0) for k in ("fm_demod", "resample", "filter", "fused_tail"))
        result = {
            "metric": "Msamples/s through FM pipeline (decim8->demod->resamp3/10->filt)",
            "value": round(total_samples / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            **({"audio_crc32_per_rank": main_run["crc"]} if main_run["crc"] is not None else {}),
            "config": {
                "workload": "full FM chain (u8 IQ -> decim8 127 taps -> fmDemod -> resamp 3/10 191 taps -> 128-tap sym filter -> *0.2), 8192-sample block seams",
                "blocks_per_gpu_per_pass": args.blocks,
                "samples_per_gpu_per_pass": S_len,
                "passes_per_step": passes,
                "samples_per_gpu_per_step": S_len * passes,
                "ms_per_pass": round(elapsed / (args.steps * passes) * 1e3, 4),
                "sharding": "none" if world == 1 else f"contiguous shards x{world}, halo exchange of {plan.halo_cap} samples per pass"
                            + (", overlapped with the outputs that need no halo" if main_run["overlap"] else ""),
                "ranks_seen_by_rccl": None if comm is None else comm.size,
                "passes_per_exchange": main_run["ppe"],
                "exchange_us_per_pass": None if main_run["exchange_us"] is None else round(main_run["exchange_us"] / main_run["ppe"], 2),
                "exchange_us_what": "one halo exchange alone on the compute stream (50 back to back, maximum over ranks) divided by the passes it serves",
                "devices_visible": ndev,
                **({"ranks_share_devices": True} if world > ndev else {}),
                "halo_transport": None if world == 1 else ("rccl: ncclSend/ncclRecv inside libsdr_hip.so (sdrhip_fm_chain_halo_exchange) on the compute stream"
                                                           if comm is not None else "host memory through gloo (fallback / plumbing check)"),
                "order": "AVX (bit-exact vs reference AVX path)",
                "overlap": ("two passes in flight inside the library (sdrhip_fm_chain_set_overlap): consecutive runs alternate between two internal "
                            "streams and workspace halves; both audio buffers identical: " + str(main_run["same_audio"])) if main_run["lib_overlap"]
                           else "none: one pass at a time on one stream",
            },
            "roofline": {
                "kernel": "k_decimate_systolic<u8> (u8->cfloat convert fused + 128-tap complex decimate-by-8, register-resident systolic walk) + seam fix-up",
                "bound": "valu",
                "achieved": round(valu_achieved, 2),
                "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(valu_achieved / VALU_PEAK_TFLOPS, 4),
                "peak_note": "f32 VALU with UNFUSED multiply and add (parity forbids FMA; MFMA is an fma chain): 256 CU x 4 SIMD x 32 lanes x 2.4 GHz",
                "traffic": traffic,
                **({"traffic_note": traffic_note} if traffic_note else {}),
                "avg_launch_ms": round(stage_ms["decimate"], 5),
                "algorithmic_flops_per_launch": k2_flops,
                "algorithmic_bytes_per_launch": k2_alg_bytes,
                "hbm": {"achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 4),
                        "note": "secondary roof: fusing the convert removed 85 % of this kernel's bytes"},
            },
            "roofline_config1_cfloat_decimate": cfg1,
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
            **({"stage_ms_note": "fm_demod 0: fmDemod runs inside the resampler's tile loader (the default since round 4) and is booked under `resample`"}
               if stage_ms.get("fm_demod", 1.0) == 0.0 and stage_ms.get("resample", 0.0) > 0.0 else {}),
            **({"one_pass_at_a_time": {"value": round(world * single_pass["S_len"] * single_pass["passes"] * args.steps / single_pass["elapsed"] / 1e6, 1),
                                       "ms_per_pass": round(single_pass["elapsed"] / (args.steps * single_pass["passes"]) * 1e3, 4),
                                       "what": "the same passes one at a time on one stream, with the per-stage HIP events in the timed region: the run "
                                               "stage_ms and roofline are taken from"}} if single_pass is not None else {}),
            **({} if events_in_region else {"stage_ms_from": "a separate short run of the same passes (the timed region of a launch-bound shard carries no event records)"}),
            "tail_ms": round(tail_ms, 5),
            "fm_carrier_input": fm_input,
            "example_taps_chain": example_taps,
            "two_passes_in_flight": main_two,
            "shard_1M_samples_per_gpu": shard_1m,
            "without_halo_exchange": replicas,
            # N > 1: how to read the line on its own -- each rank's own clock from the start of the timed region to the end of ITS work
            # (ms per pass; `value` uses the barrier-to-barrier maximum) and the share of the trivially-parallel bound (the same passes without the exchange) that survives it
            **({"per_rank_ms_per_pass": [round(e / (args.steps * passes) * 1e3, 4) for e in main_run["rank_elapsed"]],
                "scaling_efficiency": round((total_samples / elapsed / 1e6) / replicas["value"], 4) if replicas and replicas.get("value") else None,
                "scaling_efficiency_what": "value / without_halo_exchange.value: 1.0 = the halo exchange costs nothing; the driver's own "
                                           "efficiency (value at N over N x value at 1) needs the N = 1 run"} if world > 1 else {}),
            "host_streamed": host,
            "launch_size_sweep": launch_sweep,
            "power": power,
            "cpu_baseline": cpu,
        }
        emit(result, args)

    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if rccl_hung:
        sys.stdout.flush()
        os._exit(0)          # a thread is still inside ncclCommInitRank: do not wait for it at interpreter exit


if __name__ == "__main__":
    main()
