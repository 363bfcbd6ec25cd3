"""Two passes in flight three ways, alternating in one process: one stream; the library's own overlap (sdrhip_fm_chain_set_overlap:
two internal streams); two chain objects on two torch streams (what bench.py's `two_passes_in_flight` does).  Full-size passes.
Usage: python tools/overlap_probe.py [rounds]   (run it with GPU_MAX_HW_QUEUES=8 as well: HIP maps streams onto a few hardware queues)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S

n = 1 << 29
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mk = lambda: L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
u8 = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
chains = [mk(), mk()]
q0, q1, _ = chains[0].plan(0, n, n)
wsb = chains[0].workspace_bytes(n)
chains[0].set_overlap(True)
wsb2 = chains[0].workspace_bytes(n)
chains[0].set_overlap(False)
wss = [torch.empty(wsb2, dtype=torch.uint8, device="cuda"), torch.empty(wsb, dtype=torch.uint8, device="cuda")]
outs = [torch.empty(q1 - q0, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def timed(fn, fin=lambda: None, warm=100, reps=200):
    for i in range(warm):
        fn(i)
    fin()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    fin()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for r in range(rounds):
    a = timed(lambda i: chains[0].run(u8.data_ptr(), 0, n, outs[0].data_ptr(), q0, q1, wss[0].data_ptr(), wsb, stream=streams[0].cuda_stream))
    chains[0].set_overlap(True)
    b = timed(lambda i: chains[0].run(u8.data_ptr(), 0, n, outs[i & 1].data_ptr(), q0, q1, wss[0].data_ptr(), wsb2, stream=streams[0].cuda_stream),
              fin=lambda: chains[0].join(streams[0].cuda_stream))
    chains[0].set_overlap(False)
    c = timed(lambda i: chains[i & 1].run(u8.data_ptr(), 0, n, outs[i & 1].data_ptr(), q0, q1, wss[i & 1].data_ptr(), wsb, stream=streams[i & 1].cuda_stream))
    print(f"round {r}: one stream {a * 1e6:7.1f} us/pass   library overlap {b * 1e6:7.1f}   two chains on two streams {c * 1e6:7.1f}", flush=True)
