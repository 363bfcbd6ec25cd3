import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, time
import sdr_amd.lib as L
import signals as S
B = 8192
n = 1 << 29
ch = L.FmChain(8, S.taps_example_rf_decim(), 3, 10, S.taps_example_audio_resampler(), S.taps_example_audio_filter_half(), 0.2, B)
u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
q0, q1, _ = ch.plan(0, n, n)
wsb = ch.workspace_bytes(n); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
a = torch.empty(q1 - q0, device="cuda")
for fused in (False, True, False, True):
    ch.set_demod_fusion(fused)
    for _ in range(10): ch.run(u8.data_ptr(), 0, n, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
    torch.cuda.synchronize()
    ch.enable_timing(True)
    for _ in range(30): ch.run(u8.data_ptr(), 0, n, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
    torch.cuda.synchronize()
    ms, r = ch.read_timing(); ch.enable_timing(False)
    print("fused" if fused else "stage", {k: round(v, 4) for k, v in ms.items() if v}, "sum", round(sum(ms.values()), 4))
