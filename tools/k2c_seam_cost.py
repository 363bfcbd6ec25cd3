import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
n = 1 << 27
K = (n - 128) // 8 + 1
dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
out = torch.empty(2 * K + 64, device="cuda")
stream = torch.cuda.current_stream(); st = stream.cuda_stream
x = torch.rand(2 * n, device="cuda") * 2 - 1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(seam, reps=20):
    for _ in range(10): dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
    e0.record(stream)
    for _ in range(reps): dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, seam, stream=st)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(3):
    a = t(0); b = t(8192)
    print(f"cfloat /8 128 taps 2^27 samples: no seams {a:.4f} ms ({8*n/a/1e9/8000*1e3:.3f} of read roof), 8192-sample seams {b:.4f} ms ({8*n/b/1e9/8000*1e3:.3f}); fix-up costs {1e3*(b-a):.1f} us")
# sustained: launch time over 3000 back-to-back launches (~0.7 s), per window of 100
print("sustained, 8192-sample seams, ms per launch in windows of 100 launches:")
row = []
for w in range(30):
    e0.record(stream)
    for _ in range(100): dec.run(x.data_ptr(), 0, out.data_ptr(), 0, K, 8192, stream=st)
    e1.record(stream); torch.cuda.synchronize()
    row.append(e0.elapsed_time(e1) / 100)
print(" ".join(f"{v:.4f}" for v in row))
