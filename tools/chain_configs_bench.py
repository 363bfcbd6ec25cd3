"""Device-resident FM chain throughput for first stages other than the headline one (decimate by 8, 127 taps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def main():
    n = 1 << 28
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    print(L.device_name())
    for D, ntaps, I, Dr in ((8, 127, 3, 10), (8, 63, 3, 10), (8, 99, 3, 10), (8, 31, 3, 10), (4, 63, 3, 20), (16, 127, 3, 5), (10, 99, 1, 2), (8, 200, 3, 10)):
        chain = L.FmChain(D, S.gauss_taps(ntaps, ntaps), I, Dr, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
        q0, q1, _ = chain.plan(0, n, -1)
        q1 -= 4000
        wsb = chain.workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.empty(q1, device="cuda")
        run = lambda: chain.run(u8.data_ptr(), 0, n - 16384, out.data_ptr(), 0, q1, ws.data_ptr(), wsb, stream=st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"decimate /{D} {ntaps:3d} taps, resample {I}/{Dr}: {n / t / 1e9:7.1f} Gsamples/s")
        del ws, out


if __name__ == "__main__":
    main()
