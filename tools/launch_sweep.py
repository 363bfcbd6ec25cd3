"""Launch-size sweep (VERDICT r04 "next" 6; SURVEY 8(d) C2 / C3: one call on one device-resident buffer of B 8192-sample blocks,
the shape of benchmarks/Benchmarks.hs:79-134): the full FM chain and BASELINE configs[1]'s cfloat decimator at
B in {1, 8, 64, 128, 512, 2048, 8192, 65536} blocks per launch.  For every size: the route the library picks on its own
(`auto`: one-kernel chain / decimator + fused tail / stage kernels; systolic or tile decimator) and every other route forced, so
that the thresholds that decide the routing (chain.cpp kSmallChainAutoOutputs, tail_shape_ok's 768 outputs, abi_device.cpp
kSmallSeamedLaunch, the systolic kernel's minimum) can be seen to sit at the crossovers.

    python tools/launch_sweep.py            (a table + one JSON line; bench.py embeds sweep() as `launch_size_sweep`)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCK = 8192
SIZES = (1, 8, 64, 128, 256, 512, 1024, 2048, 8192, 65536)      # VERDICT r04's eight sizes + 256 and 1024 (the one-kernel chain's crossover)


def _time(fn, seconds=0.12, min_reps=5):
    import torch
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-6)
    reps = max(min_reps, min(20000, int(seconds / one)))
    for _ in range(max(2, reps // 5)):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def chain_sweep(L, S, sizes=SIZES, seconds=0.12):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, BLOCK)
    nmax = max(sizes) * BLOCK
    q0m, q1m, halo = chain.plan(0, nmax, -1)
    u8 = torch.randint(0, 256, (2 * (nmax + halo + 64),), dtype=torch.uint8, device="cuda")
    ws_bytes = chain.workspace_bytes(nmax + halo + 64)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(q1m - q0m + 64, device="cuda")
    # (set_small_chain mode, set_fused_tail mode): 0 = never, 1 = always, 2 = auto
    routes = {"auto": (2, 2), "one_kernel_chain": (1, 2), "decimator_plus_fused_tail": (0, 1), "stage_kernels": (0, 0)}
    rows = []
    for b in sizes:
        n = b * BLOCK
        q0, q1, h = chain.plan(0, n, -1)
        row = {"blocks_per_launch": b, "samples_per_launch": n}
        for name, (small, tail) in routes.items():
            if name == "one_kernel_chain" and b > 2048:
                continue                      # a kernel that recomputes the first stage ~1.9 times over: pointless beyond launch-bound sizes
            if name == "decimator_plus_fused_tail" and b > 8192:
                continue
            chain.set_small_chain(small, 0, 0)
            chain.set_fused_tail(tail)
            run = lambda: chain.run(u8.data_ptr(), 0, n + h, out.data_ptr(), q0, q1, ws.data_ptr(), ws_bytes, stream=st)
            try:
                sys0, small0 = L.lib.sdrhip_debug_systolic_launches(), L.lib.sdrhip_debug_small_chain_launches()
                chain.enable_timing(True)
                run()
                torch.cuda.synchronize()
                ms, _ = chain.read_timing()
                chain.enable_timing(False)
                took = ("one-kernel chain" if L.lib.sdrhip_debug_small_chain_launches() > small0 or ms.get("fused_chain", 0) > 0 else
                        ("systolic" if L.lib.sdrhip_debug_systolic_launches() > sys0 else "tile") + " decimator + " +
                        ("fused tail" if ms.get("fused_tail", 0) > 0 else "fmDemod in the resampler's loader + filter" if ms.get("fm_demod", 1) == 0 else "three stage kernels"))
                t = _time(run, seconds)
            except Exception as e:            # noqa: BLE001 -- a forced route the shape does not admit
                row[name] = f"n/a ({e})"
                continue
            row[name] = {"us_per_launch": round(t * 1e6, 2), "Gsamples_per_s": round(n / t / 1e9, 2), "route": took}
        best = min(v["us_per_launch"] for k, v in row.items() if isinstance(v, dict))
        row["auto_over_best"] = round(row["auto"]["us_per_launch"] / best, 3)
        rows.append(row)
    chain.set_small_chain(2, 0, 0)
    chain.set_fused_tail(2)
    return rows


def k2c_sweep(L, S, sizes=SIZES, seconds=0.12):
    """BASELINE configs[1]: cfloat IQ, 127 -> 128 taps, decimate by 8, 8192-sample seams; the library's own choice (by launch size),
    the systolic kernel wherever its shape fits, the tile kernel everywhere."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    nmax = max(sizes) * BLOCK
    nmax = min(nmax, 1 << 27)                                   # 1 GiB of cfloat input
    x = torch.rand(2 * nmax, device="cuda") * 2 - 1
    out = torch.empty(2 * (nmax // 8) + 64, device="cuda")
    rows = []
    for b in sizes:
        n = min(b * BLOCK, nmax)
        K = (n - 128) // 8 + 1
        row = {"blocks_per_launch": n // BLOCK, "samples_per_launch": n}
        # two rounds over the routes, the second in reverse order, the better time of each: which kernel ran just before moves a
        # 10-us launch by several per cent (round 6: `auto` and the forced systolic kernel are the SAME launch at 256 blocks and
        # differed by 0.7 us, always in favour of whichever was not measured right after the tile kernel)
        routes = (("auto", 2), ("systolic_kernel", 1), ("tile_kernel", 0))
        best = {}
        for order in (routes, routes[::-1]):
            for name, on in order:
                L.lib.sdrhip_debug_set_systolic(on)
                s0 = L.lib.sdrhip_debug_systolic_launches()
                run = lambda: dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, BLOCK, stream=st)
                t = _time(run, seconds / 2)
                took = "systolic kernel + seam fix-up" if L.lib.sdrhip_debug_systolic_launches() > s0 else "tile kernel (seams in the kernel for short launches)"
                if name not in best or t < best[name][0]:
                    best[name] = (t, took)
        for name, _ in routes:
            t, took = best[name]
            row[name] = {"us_per_launch": round(t * 1e6, 2), "Gsamples_per_s": round(n / t / 1e9, 2), "read_only_frac_of_8TBps": round(8.0 * n / t / 8e12, 4), "route": took}
        row["auto_over_best"] = round(row["auto"]["us_per_launch"] / min(v["us_per_launch"] for v in row.values() if isinstance(v, dict)), 3)
        rows.append(row)
    L.lib.sdrhip_debug_set_systolic(int(os.environ.get("SDRHIP_SYSTOLIC", "2")))
    return rows


def sweep(L, S, seconds=0.12):
    return {"what": "one call on one device-resident buffer of B 8192-sample blocks (benchmarks/Benchmarks.hs:79-134's shape), wall clock over "
                    "back-to-back launches; `auto` = the route the library takes on its own, the other columns force a route; auto_over_best "
                    "= auto's time over the best column's (1.0 = the thresholds sit on the right side of the crossover)",
            "full_chain_u8": chain_sweep(L, S, seconds=seconds), "config1_cfloat_decimator": k2c_sweep(L, S, seconds=seconds)}


if __name__ == "__main__":
    import sdr_amd.lib as L
    import signals as S
    r = sweep(L, S, 0.25)
    for key in ("full_chain_u8", "config1_cfloat_decimator"):
        print("==", key)
        for row in r[key]:
            cols = "  ".join(f"{k}: {v['us_per_launch']:10.2f} us [{v['route']}]" if isinstance(v, dict) else f"{k}: {v}" for k, v in row.items()
                             if k not in ("blocks_per_launch", "samples_per_launch"))
            print(f"B = {row['blocks_per_launch']:6d}  {cols}")
    print(json.dumps(r))
