"""Round 6: the seam fix-up inside the systolic decimator's launch (mode 2) against the fix-up as a second launch (mode 1: round 5's form),
BASELINE configs[1] (cfloat in, 2^27 samples) and the chain's first stage (u8 in, 2^29 samples), 8192-sample seams: alternating rows
of back-to-back launches in one process, HIP events on the launch stream.

    python tools/k2_fix_inside_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

B = 8192
A = 1      # round 5's form: non-temporal loads, fix-up as a second launch.  (profiles/r06/k2_fix_inside_ab_sizes.txt was taken with a temporary
           # mode 3 = mode 2's loads + the second launch at EVERY size, and no size bound on the in-launch fix-up: it is what set that bound)
st = torch.cuda.current_stream()
sp = st.cuda_stream
dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)


def row(fn, reps, warm):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for kind, log2n, reps in [(k, l, max(100, 400 >> max(0, l - 27))) for l in (22, 23, 24, 25, 26, 27, 28) for k in ("cfloat", "u8")] + [("u8", 29, 150)]:
    n = 1 << log2n
    K = (n - 128) // 8 + 1
    x = (torch.rand(2 * n, device="cuda") * 2 - 1) if kind == "cfloat" else torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    outs = {}
    for mode in (A, 2):
        L.lib.sdrhip_debug_set_systolic(mode)
        o = torch.empty(2 * K + 64, device="cuda")
        (dec.run if kind == "cfloat" else dec.run_u8)(x.data_ptr(), 0, o.data_ptr(), 0, K, B, stream=sp)
        torch.cuda.synchronize()
        outs[mode] = o
    same = torch.equal(outs[A][:2 * K].view(torch.int32), outs[2][:2 * K].view(torch.int32))
    o = outs[2]
    res = {A: [], 2: []}
    for rnd in range(4):
        for mode in ((A, 2) if rnd % 2 == 0 else (2, A)):
            L.lib.sdrhip_debug_set_systolic(mode)
            res[mode].append(row(lambda: (dec.run if kind == "cfloat" else dec.run_u8)(x.data_ptr(), 0, o.data_ptr(), 0, K, B, stream=sp), reps, reps // 4))
    a, b = sum(res[A]) / 4, sum(res[2]) / 4
    print(f"{kind:6s} 2^{log2n} samples, 8192-sample seams: fix-up as a second launch {a:8.2f} us  ({' '.join(f'{v:.1f}' for v in res[A])});  inside the launch {b:8.2f} us "
          f"({' '.join(f'{v:.1f}' for v in res[2])});  ratio {b / a:.4f};  same bits: {same}"
          + (f";  read-only fraction of 8 TB/s: {8.0 * n / (a * 1e-6) / 8e12:.4f} -> {8.0 * n / (b * 1e-6) / 8e12:.4f}" if kind == "cfloat" else ""), flush=True)
    del x, outs, o
L.lib.sdrhip_debug_set_systolic(2)
