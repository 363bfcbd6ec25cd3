"""Run the K2 decimate kernel a few times (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 26)
    mode = sys.argv[2] if len(sys.argv) > 2 else "both"
    K = (n - 128) // 8 + 1
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    out = torch.empty(2 * K, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if mode in ("both", "f32"):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        for _ in range(5):
            dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, 0, stream=st)
    if mode in ("both", "u8"):
        u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
        for _ in range(5):
            dec.run_u8(u8.data_ptr(), 0, out.data_ptr(), 0, K, 0, stream=st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
