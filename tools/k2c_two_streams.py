"""The cfloat-in decimate-by-8 kernel (BASELINE configs[1]: 127 taps, 8192-sample seams) on 2^27 samples: launches back to back on
one stream against the same launches alternating between two streams (separate outputs) -- microseconds per launch by wall
clock over 400 launches, alternating rounds; uniform f32 data and u8-derived data."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
REPS = 400 if n <= (1 << 27) else 120
SEAM = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = (n - 128) // 8 + 1
dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [torch.empty(2 * k + 64, device="cuda") for _ in range(2)]
for name, x in (("uniform f32", torch.rand(2 * n, device="cuda") * 2 - 1),
                ("u8-derived ", (torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8).float() - 128.0) / 128.0)):
    def run(nstream, reps):
        for i in range(reps):
            j = i % nstream
            dec.run(x.data_ptr(), 0, outs[j].data_ptr(), 0, k, SEAM, stream=streams[j].cuda_stream)
    run(1, 300)
    torch.cuda.synchronize()
    rows = []
    for rnd in range(3):
        r = []
        for ns in (1, 2):
            run(ns, 50)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(ns, REPS)
            torch.cuda.synchronize()
            r.append((time.perf_counter() - t0) / REPS * 1e6)
        rows.append(r)
    same = bool(torch.equal(outs[0].view(torch.int32)[: 2 * k], outs[1].view(torch.int32)[: 2 * k]))
    print(f"2^{n.bit_length() - 1} samples, seam {SEAM}, {name}: us per launch, one stream / two streams: " + "  ".join(f"{a:6.1f}/{b:6.1f}" for a, b in rows) +
          f"   read-roof fraction {8.0 * n / (rows[-1][0] * 1e-6) / 8e12:.3f} / {8.0 * n / (rows[-1][1] * 1e-6) / 8e12:.3f}   same bits: {same}")
    del x
