"""Quick kernel timing on the GPU box (development aid; bench.py is the contract)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import sdr_amd.lib as L
import signals as S


def timeit(fn, iters=200, warm=100):
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    print(L.device_name())
    st = torch.cuda.current_stream().cuda_stream
    taps = S.taps_decim127()
    dec = L.Decimator(8, taps, L.ORDER_AVX, complex_=True)
    for log2n in (20, 24, 26, 27):
        n = 1 << log2n
        K = (n - 128) // 8 + 1
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
        out = torch.empty(2 * K, device="cuda")
        for seam in (0, 8192):
            t = timeit(lambda: dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            print(f"decimate cfloat n=2^{log2n} seam={seam}: {t*1e6:9.1f} us  {n/t/1e9:8.2f} Gsamp/s  read {8*n/t/1e12:6.3f} TB/s")
            t = timeit(lambda: dec.run_u8(u8.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st))
            print(f"decimate u8     n=2^{log2n} seam={seam}: {t*1e6:9.1f} us  {n/t/1e9:8.2f} Gsamp/s  read {2*n/t/1e12:6.3f} TB/s")
    # chain
    chain = L.FmChain(8, taps, 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    for log2n in (20, 24, 26):
        n = 1 << log2n
        u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
        q0, q1, halo = chain.plan(0, n, n)
        wsb = chain.workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        out = torch.empty(q1 - q0, device="cuda")
        t = timeit(lambda: chain.run(u8.data_ptr(), 0, n, out.data_ptr(), q0, q1, ws.data_ptr(), wsb, stream=st))
        print(f"fm chain n=2^{log2n}: {t*1e6:9.1f} us  {n/t/1e9:8.2f} Gsamp/s")
    # copy roofline reference
    a = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"torch copy 256MiB: {2*(1<<28)/t/1e12:.3f} TB/s (read+write)")


if __name__ == "__main__":
    main()
