"""A/B in ONE process, alternating: the chain's convert+decimate stage with the FULL-tile instantiation on and off
(sdrhip_debug_set_full_tiles); per-stage HIP-event times.  Order effects (power state) are averaged out by alternation."""
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
res = {0: [], 1: []}
for rnd in range(8):
    for on in (1, 0):
        L.lib.sdrhip_debug_set_full_tiles(on)
        for _ in range(10):
            run()
        chain.enable_timing(True)
        for _ in range(40):
            run()
        ms, _ = chain.read_timing()
        chain.enable_timing(False)
        res[on].append(ms["decimate"])
for on in (1, 0):
    v = res[on]
    print(f"full tiles {on}: decimate stage ms per run: " + " ".join(f"{x:.4f}" for x in v) + f"  mean {sum(v)/len(v):.4f}")
