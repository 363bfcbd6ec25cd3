"""Full-size passes (2^29 samples) on one stream and alternating between two (separate chains, workspaces and outputs): µs per
pass and the per-stage HIP-event times of the chains (does the decimator's own duration change when the other pass's tail
kernels run beside it?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

n = 1 << 29
mk = lambda: L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
for nstream in (1, 2, 1, 2):
    chains = [mk() for _ in range(nstream)]
    q0, q1, _ = chains[0].plan(0, n, n)
    wsb = chains[0].workspace_bytes(n)
    wss = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(nstream)]
    outs = [torch.empty(q1 - q0, device="cuda") for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    def one(i):
        j = i % nstream
        chains[j].run(u8.data_ptr(), 0, n, outs[j].data_ptr(), q0, q1, wss[j].data_ptr(), wsb, stream=streams[j].cuda_stream)
    for i in range(150):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for i in range(reps):
        one(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    for c in chains:
        c.enable_timing(True)
    for i in range(40):
        one(i)
    torch.cuda.synchronize()
    ms = [c.read_timing()[0] for c in chains]
    for c in chains:
        c.enable_timing(False)
    print(f"{nstream} stream(s): {dt * 1e6:8.2f} us/pass = {n / dt / 1e9:6.1f} Gsample/s; stage ms (chain 0): " +
          " ".join(f"{k} {v:.4f}" for k, v in ms[0].items() if v > 0), flush=True)
    del chains, wss, outs
