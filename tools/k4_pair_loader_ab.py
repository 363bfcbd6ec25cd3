"""Round 6 (VERDICT r05 "next" 4): the fused fmDemod + resampler kernel with the pair loader (one 16-byte load per two inputs, predecessor by
DPP / v_readlane, 249 cycles per workgroup) against the production loader (two 8-byte loads per input, 256 cycles), both at six waves per
SIMD, inside the full chain at 2^29 samples: alternating rounds in one process, per-stage HIP-event times, audio compared bit for bit."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

B = 8192
n = 1 << 29
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
q0, q1, _ = chain.plan(0, n, n)
wsb = chain.workspace_bytes(n)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
auds = {}
for on in (0, 1):
    L.lib.sdrhip_experiment_set_pair_loader(on)
    a = torch.zeros(q1 - q0, device="cuda")
    chain.run(u8.data_ptr(), 0, n, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
    torch.cuda.synchronize()
    auds[on] = a
print("audio identical:", bool(torch.equal(auds[0].view(torch.int32), auds[1].view(torch.int32))), f"({q1 - q0} samples)")
a = auds[0]
for rnd in range(5):
    row = []
    for on in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        L.lib.sdrhip_experiment_set_pair_loader(on)
        for _ in range(10):
            chain.run(u8.data_ptr(), 0, n, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
        chain.enable_timing(True)
        for _ in range(40):
            chain.run(u8.data_ptr(), 0, n, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
        torch.cuda.synchronize()
        ms, runs = chain.read_timing()
        chain.enable_timing(False)
        row.append(f"{'pair loader' if on else 'production '}: resample {ms['resample']:.4f} decimate {ms['decimate']:.4f} filter {ms['filter']:.4f} ms")
    print(f"round {rnd}: " + "   |   ".join(row), flush=True)
L.lib.sdrhip_experiment_set_pair_loader(0)
