"""Fine launch-size sweep of the first stage alone (VERDICT r05 "weak" 7): the decimate-by-8, 128-tap decimator on B 8192-sample blocks,
cfloat in (BASELINE configs[1]) and u8 in (the chain's first stage), systolic kernel against tile kernel at every size, to see where
the crossover sits as a function of the number of workgroups (rounds of 1024 resident workgroups on 256 CUs).

    python tools/route_sweep_fine.py [seconds per point]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
BLOCK = 8192


def main():
    import torch
    import sdr_amd.lib as L
    import signals as S
    from launch_sweep import _time
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    st = torch.cuda.current_stream().cuda_stream
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    sizes = sorted(set(list(range(128, 1025, 64)) + list(range(1024, 4097, 128)) + list(range(4096, 8193, 512)) + [12288, 16384]))
    nmax = max(sizes) * BLOCK
    xf = torch.rand(2 * nmax, device="cuda") * 2 - 1
    xu = torch.randint(0, 256, (2 * nmax,), dtype=torch.uint8, device="cuda")
    out = torch.empty(2 * (nmax // 8) + 64, device="cuda")
    for kind in (("cfloat", "u8") if len(sys.argv) < 3 else sys.argv[2:]):
        print(f"== {kind}: B  nwg(systolic)  rounds  systolic(mode 1: non-temporal loads)_us  tile_us  auto_us  auto_route  auto/best")
        for b in sizes:
            n = b * BLOCK
            K = (n - 128) // 8 + 1
            nwg = ((K + 239) // 240 + 3) // 4
            t = {}
            for name, mode in (("systolic", 1), ("tile", 0), ("auto", 2)):
                L.lib.sdrhip_debug_set_systolic(mode)
                s0 = L.lib.sdrhip_debug_systolic_launches()
                if kind == "cfloat":
                    run = lambda: dec.run(xf.data_ptr(), 0, out.data_ptr(), 0, K, BLOCK, stream=st)
                else:
                    run = lambda: dec.run_u8(xu.data_ptr(), 0, out.data_ptr(), 0, K, BLOCK, stream=st)
                t[name] = _time(run, secs) * 1e6
                if name == "auto":
                    route = "systolic" if L.lib.sdrhip_debug_systolic_launches() > s0 else "tile"
            print(f"{b:6d} {nwg:6d} {nwg / 1024:6.2f} {t['systolic']:9.2f} {t['tile']:9.2f} {t['auto']:9.2f} {route:9s} {t['auto'] / min(t['systolic'], t['tile']):6.3f}", flush=True)
    L.lib.sdrhip_debug_set_systolic(2)


if __name__ == "__main__":
    main()
