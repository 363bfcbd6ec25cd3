"""A/B in ONE process, alternating: the tail of the chain (fmDemod, resampler, filter) as stage kernels (mode 0), as one
kernel (mode 1) and as fmDemod + fused resampler/filter (mode 3); per-stage HIP-event times per 2^29-sample pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

n = 1 << 29
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
q0, q1, _ = chain.plan(0, n, n)
wsb = chain.workspace_bytes(n)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
out = torch.empty(q1 - q0, device="cuda")
st = torch.cuda.current_stream().cuda_stream
run = lambda: chain.run(u8.data_ptr(), 0, n, out.data_ptr(), q0, q1, ws.data_ptr(), wsb, stream=st)
for _ in range(100):
    run()
torch.cuda.synchronize()
res = {}
for rnd in range(6):
    for mode in (0, 3, 1):
        chain.set_fused_tail(mode)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        chain.enable_timing(True)
        e0.record()
        for _ in range(40):
            run()
        e1.record()
        ms, _ = chain.read_timing()
        chain.enable_timing(False)
        torch.cuda.synchronize()
        tail = ms["fm_demod"] + ms["resample"] + ms["filter"] + ms["fused_tail"]
        res.setdefault(mode, []).append((tail, e0.elapsed_time(e1) / 40))
for mode in (0, 3, 1):
    v = res[mode]
    print(f"fused_tail mode {mode}: tail ms " + " ".join(f"{a:.4f}" for a, _ in v) + f" | mean tail {sum(a for a,_ in v)/len(v):.4f}  mean pass {sum(b for _,b in v)/len(v):.4f} ms")
