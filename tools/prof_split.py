"""Run one lane-split kernel case a few times (for rocprofv3): real decimate /8, 128 taps, AVX order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import sdr_amd.lib as L
import signals as S


def main():
    n = 1 << 26
    K = (n - 128) // 8 + 1
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX)
    x = torch.rand(n, device="cuda") * 2 - 1
    out = torch.empty(K, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, 0, stream=st)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
