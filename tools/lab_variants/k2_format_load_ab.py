"""Round 6 experiment: the u8 systolic decimator with the u8 -> f32 conversion done by the memory pipeline (buffer_load_format_xyzw, 8_8_8_8
USCALED; temporary mode 4) against the production kernel (mode 2): alternating rows, HIP events, bit comparison."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
B = 8192
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


for log2n, reps in ((24, 1500), (27, 500), (29, 150)):
    n = 1 << log2n
    K = (n - 128) // 8 + 1
    x = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    outs = {}
    for mode in (2, 4):
        L.lib.sdrhip_debug_set_systolic(mode)
        o = torch.full((2 * K + 64,), float("nan"), device="cuda")
        dec.run_u8(x.data_ptr(), 0, o.data_ptr(), 0, K, B, stream=sp)
        torch.cuda.synchronize()
        outs[mode] = o
    same = torch.equal(outs[2][:2 * K].view(torch.int32), outs[4][:2 * K].view(torch.int32))
    o = outs[2]
    res = {2: [], 4: []}
    for rnd in range(4):
        for mode in ((2, 4) if rnd % 2 == 0 else (4, 2)):
            L.lib.sdrhip_debug_set_systolic(mode)
            res[mode].append(row(lambda: dec.run_u8(x.data_ptr(), 0, o.data_ptr(), 0, K, B, stream=sp), reps, reps // 4))
    a, b = sum(res[2]) / 4, sum(res[4]) / 4
    print(f"u8 2^{log2n} samples: production {a:8.2f} us ({' '.join(f'{v:.1f}' for v in res[2])});  format loads {b:8.2f} us ({' '.join(f'{v:.1f}' for v in res[4])});  ratio {b / a:.4f};  same bits: {same}", flush=True)
L.lib.sdrhip_debug_set_systolic(2)
