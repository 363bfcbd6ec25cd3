#!/usr/bin/env python3
"""bench.py -- Msamples/s of input IQ through the FM pipeline on MI355X.

Workload (BASELINE.json configs[2]; configs[4] for N > 1): the full FM chain of
examples/fm/fm.hs:34-41 -- u8 IQ -> cfloat (fused) -> 127(->128)-tap complex FIR
decimate-by-8 -> fmDemod -> polyphase resample 3/10 (191 taps) -> 128-tap (64
half-tap) symmetric FIR -> *0.2 -- with the reference Pipes' 8192-sample block
seams reproduced bit-exactly.  One "step" = one pass of that chain over one batch
of `--blocks` 8192-sample blocks per GPU (default 65536 blocks = 2^29 samples = 1 GiB of u8 IQ),
inputs already resident in HBM.

N > 1: one process per GPU (torch.distributed, backend "nccl" = RCCL).  The sample
stream is sharded contiguously, rank r owning samples [r*S, (r+1)*S) of each
super-block; every step each rank receives the head of its right neighbour's shard
(the composed ntaps-1 overlap of all four stages, ~4.4k samples = 8.7 KB of u8)
over RCCL send/recv and processes shard+halo.  Per-GPU work is fixed: weak scaling.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline`
(dominant kernel = the fused convert+decimate kernel, timed with HIP events on its
own stream inside the timed region) and `cpu_baseline` (the reference's own C
kernels, oracle/_ref, timed on this host's cores; N=1 only).
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
def cpu_chain_worker(seconds, nblk=64):
    """One thread: the FM chain over 8192-sample blocks exactly as the reference runs it
    (one C call per stage per block, Pipes re-blocking), looping for `seconds`.
    Returns (samples_per_second, kind)."""
    import numpy as np
    from oracle.oracle import Oracle, Ref, have_ref, duplicate
    import signals as S
    orc = Oracle()
    ref = Ref() if have_ref() else None
    kind = "reference" if ref is not None else "port"
    taps_d = np.concatenate([S.taps_decim127(), np.zeros(1, np.float32)])
    taps_dd = duplicate(taps_d)
    half = S.taps_audio_half64()
    prep = orc.prepare_coeffs(8, 3, 10, S.taps_resamp191())
    u8 = S.iq_u8(nblk * BLOCK)
    blocks = [np.ascontiguousarray(u8[2 * i * BLOCK:2 * (i + 1) * BLOCK]) for i in range(nblk)]
    n_dec = (BLOCK - 128) // 8 + 1

    def one_pass():
        dec_acc = []
        for b in blocks:
            if ref is not None:
                x = ref.convert("convertCAVX", b)
                d = ref.decim("decimateAVXRC", n_dec, 8, taps_dd, x, True)
            else:
                x = orc.convert_u8(b)
                d = orc.decimate_rc(4, n_dec, 8, taps_dd, x)
            dec_acc.append(d)
            if len(dec_acc) == 8:
                dd = np.concatenate(dec_acc)
                dec_acc = []
                y = orc.fm_demod(dd)                       # the reference's fmDemod is Haskell; restated
                m = (y.size * 3 - 192) // 10 + 1
                if ref is not None:
                    z, _ = ref.resample("resampleAVXRR", m, prep, 0, y)
                    ref.filt("filterAVXSymmetricRR", z.size - 127, half, z)
                else:
                    z, _ = orc.resample_rr(8, m, prep, 0, y)
                    orc.filter_sym_rr(8, z.size - 127, half, z)

    one_pass()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        one_pass()
        n += nblk * BLOCK
    return n / (time.perf_counter() - t0), kind


def cpu_baseline(seconds_single=6.0, seconds_all=8.0):
    """Single-thread (how the reference actually runs) and all-cores (one independent
    stream per core) numbers, each worker a separate process."""
    cores = len(os.sched_getaffinity(0))

    def spawn(secs):
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(secs)],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)

    def collect(ps):
        tot, kind = 0.0, "port"
        for p in ps:
            out, _ = p.communicate()
            try:
                r = json.loads(out.strip().splitlines()[-1])
                tot += r["sps"]
                kind = r["kind"]
            except Exception:
                pass
        return tot, kind

    single, kind = collect([spawn(seconds_single)])
    allc, _ = collect([spawn(seconds_all) for _ in range(cores)])
    return {
        "value": round(allc / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": kind,
        "single_thread_value": round(single / 1e6, 2),
        "sample": (f"full FM chain on 64 x 8192-sample u8 IQ blocks looped for {seconds_all:.0f} s per core "
                   f"({cores} independent streams, one process per core) and {seconds_single:.0f} s single-thread; "
                   "convertCAVX/decimateAVXRC/resampleAVXRR/filterAVXSymmetricRR from the reference's own C "
                   "(oracle/_ref, -O2 -mavx2 -msse4) when kind=reference, fmDemod from the restatement "
                   "(the reference's is Haskell); ctypes call overhead included"),
    }


# --------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=65536, help="8192-sample blocks per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:
        sps, kind = cpu_chain_worker(args.cpu_worker)
        print(json.dumps({"sps": sps, "kind": kind}))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

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

    # BENCH_BACKEND=gloo is a plumbing check only (several ranks may then share one GPU and the
    # halo travels through host memory); the measured configuration is "nccl" = RCCL over xGMI.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev)
    L.check(L.lib.sdrhip_set_device(dev), "sdrhip_set_device")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)               # creates the communicator now: an RCCL problem shows up here, not mid-run
                torch.cuda.synchronize()
            except Exception as e:                   # noqa: BLE001 -- SURVEY 8(e) fallback: same halos through host memory
                sys.stderr.write(f"bench: RCCL initialisation failed ({e!r}); falling back to gloo (halo through host memory)\n")
                try:
                    dist.destroy_process_group()
                except Exception:                    # noqa: BLE001
                    pass
                backend = "gloo"
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
    S_len = args.blocks * BLOCK
    plan = sharding.ShardPlan(chain, rank, world, S_len)          # owned outputs + halo for this rank
    gen = torch.Generator(device="cuda").manual_seed(S.SEED_IQ + rank)
    buf = torch.randint(0, 256, (2 * (S_len + plan.halo_cap),), dtype=torch.uint8, device="cuda", generator=gen)
    audio = torch.empty(plan.q1 - plan.q0, dtype=torch.float32, device="cuda")
    ws_bytes = chain.workspace_bytes(S_len + plan.halo_cap)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    # N > 1: the halo (the right neighbour's first ~4k samples) is exchanged over RCCL while this rank already computes
    # the outputs that need only its own samples, [q0, q_mid); the few that reach into the halo, [q_mid, q1), run on a
    # second stream as soon as the halo has landed.  Two chain objects: each keeps its own timing events.
    overlap = world > 1 and plan.q_mid > plan.q0 and os.environ.get("BENCH_NO_OVERLAP") != "1"
    if overlap:
        chain_b = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
        aux = torch.cuda.Stream()
        ws_b = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")

    def step():
        if not overlap:
            if world > 1:
                sharding.halo_exchange(buf, plan, dist, via_host=(backend != "nccl"))   # send/recv of the ntaps-1 overlap
            chain.run(buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q1, ws.data_ptr(), ws_bytes,
                      stream=sptr)
            return
        aux.wait_stream(stream)                      # the previous step's readers of the halo region are done
        with torch.cuda.stream(aux):
            if backend == "nccl":
                for req in sharding.halo_exchange_start(buf, plan, dist):
                    req.wait()                       # aux waits for RCCL's stream; the host does not block
            else:
                sharding.halo_exchange(buf, plan, dist, via_host=True)   # plumbing check (gloo): through host memory
            if plan.q1 > plan.q_mid:
                chain_b.run(buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr() + 4 * (plan.q_mid - plan.q0), plan.q_mid, plan.q1,
                            ws_b.data_ptr(), ws_bytes, stream=aux.cuda_stream)
        chain.run(buf.data_ptr(), plan.s0, plan.n_in, audio.data_ptr(), plan.q0, plan.q_mid, ws.data_ptr(), ws_bytes, stream=sptr)
        stream.wait_stream(aux)

    if overlap:
        try:                                         # never lose an N > 1 measurement to the scheduling refinement
            step()
            torch.cuda.synchronize()
        except Exception as e:                       # noqa: BLE001
            sys.stderr.write(f"bench: overlapped halo exchange failed ({e!r}); falling back to exchange-then-compute\n")
            overlap = False

    # Clock / power-state ramp: the first ~15 ms of sustained work on a fresh process run 10 % slow
    # (interleaved A/B in tools/pipeline_ab.py); spin the same step for ~0.3 s before the W warmup steps.
    # With several ranks every step is a send/recv with the neighbours, so all ranks must run the SAME number of steps: the
    # decision to go on is taken collectively (a time-based loop per rank can differ by one step and then deadlocks).
    t_ramp = time.perf_counter()
    while True:
        go = time.perf_counter() - t_ramp < 0.3
        if world > 1:
            flag = torch.tensor([1 if go else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            go = bool(flag.item())
        if not go:
            break
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    chain.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stage_ms, runs = chain.read_timing()
    chain.enable_timing(False)

    # BASELINE configs[1] (the north_star's roofline kernel): the same decimate-by-8 kernel fed
    # cfloat IQ (8 B read + 1 B written per input sample), device-resident, 8192-sample seams.
    cfg1 = None
    if rank == 0 and world == 1:
        n1 = 1 << 27
        k1 = (n1 - 128) // 8 + 1
        dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
        x1 = torch.rand(2 * n1, device="cuda") * 2 - 1
        o1 = torch.empty(2 * k1, device="cuda")
        for _ in range(3):
            dec.run(x1.data_ptr(), 0, o1.data_ptr(), 0, k1, BLOCK, stream=sptr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record(stream)
        for _ in range(reps):
            dec.run(x1.data_ptr(), 0, o1.data_ptr(), 0, k1, BLOCK, stream=sptr)
        e1.record(stream)
        torch.cuda.synchronize()
        t1 = e0.elapsed_time(e1) * 1e-3 / reps
        cfg1 = {"kernel": "k_decimate_c4 (cfloat in) + seam fix-up", "samples_per_launch": n1,
                "avg_launch_ms": round(t1 * 1e3, 5), "Msamples_per_s": round(n1 / t1 / 1e6, 1),
                "bound": "hbm", "achieved": round(9.0 * n1 / t1 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(9.0 * n1 / t1 / 1e9 / HBM_PEAK_GBS, 4),
                "read_only_frac": round(8.0 * n1 / t1 / 1e9 / HBM_PEAK_GBS, 4)}
        del x1, o1

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    audio_crc = None
    if os.environ.get("BENCH_CHECKSUM") == "1":      # plumbing checks: the same audio whichever way the step is scheduled
        import zlib
        audio_crc = [zlib.crc32(audio.cpu().numpy().tobytes())]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, audio_crc[0])
            audio_crc = gathered

    if rank == 0:
        total_samples = world * S_len * args.steps
        k2_s = stage_ms["decimate"] * 1e-3
        k2_alg_bytes = 3.0 * plan.k2_samples                      # u8-fused K2 = 2 B read + 1 B written per input sample
        k2_flops = 64.0 * plan.k2_samples                         # 2*2*P/D unfused flop per input sample
        achieved = k2_alg_bytes / k2_s / 1e9 if k2_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k2_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("samples_per_launch") == plan.k2_samples:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
            **({"audio_crc32_per_rank": audio_crc} if audio_crc is not None else {}),
            "config": {
                "workload": "full FM chain (u8 IQ -> decim8 127 taps -> fmDemod -> resamp 3/10 191 taps -> 128-tap sym filter -> *0.2), 8192-sample block seams",
                "blocks_per_gpu_per_step": args.blocks,
                "samples_per_gpu_per_step": S_len,
                "sharding": "none" if world == 1 else f"contiguous shards x{world}, {backend} halo exchange of {plan.halo_cap} samples/step"
                            + (", overlapped with the outputs that need no halo" if overlap else ""),
                "order": "AVX (bit-exact vs reference AVX path)",
            },
            "roofline": {
                "kernel": "k_decimate_c4 (u8->cfloat convert fused + 128-tap complex decimate-by-8) + seam fix-up",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "avg_launch_ms": round(stage_ms["decimate"], 5),
                "algorithmic_bytes_per_launch": k2_alg_bytes,
                "note": "this kernel is VALU-bound, not HBM-bound: see valu",
                "valu": {"achieved": round(k2_flops / k2_s / 1e12, 2) if k2_s > 0 else 0.0, "peak": VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s (unfused f32 mul+add)",
                         "frac": round(k2_flops / k2_s / 1e12 / VALU_PEAK_TFLOPS, 4) if k2_s > 0 else 0.0},
            },
            "roofline_config1_cfloat_decimate": cfg1,
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
            "cpu_baseline": cpu,
        }
        print(json.dumps(result))

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
