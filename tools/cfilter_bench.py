"""Complex filter (decimation 1) of 128 / 64 taps, AVX order, 2^24 samples with 8192-sample seams: G elements/s and T real MAC/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
from family_bench import timeit

st = torch.cuda.current_stream().cuda_stream
m = 1 << 24
xc = torch.rand(2 * m + 1024, device="cuda") * 2 - 1
out = torch.empty(2 * m, device="cuda")
for nt in (127, 63):
    f = L.Filter(S.gauss_taps(nt, nt), L.ORDER_AVX, complex_=True)
    K = m - f.num_coeffs + 1
    for seam in (8192, 0):
        t = timeit(lambda: f.run(xc.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st), iters=50, warm=20)
        print(f"complex filter {f.num_coeffs} taps seam {seam}: {t*1e6:8.1f} us  {m/t/1e9:7.1f} G elements/s  {K*f.num_coeffs*2/t/1e12:6.2f} T MAC/s")
