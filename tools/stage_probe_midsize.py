import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, time
import sdr_amd.lib as L
import signals as S
B = 8192
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
for nb in (1024, 2048, 4096, 8192, 16384):
    n = nb * B
    u8 = torch.randint(0, 256, (2 * n + 16384,), dtype=torch.uint8, device="cuda")
    q0, q1, h = chain.plan(0, n, -1)
    wsb = chain.workspace_bytes(n + h + 64)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    a = torch.empty(q1 - q0 + 64, device="cuda")
    chain.set_small_chain(0)
    run = lambda: chain.run(u8.data_ptr(), 0, n + h, a.data_ptr(), q0, q1, ws.data_ptr(), wsb)
    for _ in range(50): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300 * 1e6
    chain.enable_timing(True)
    for _ in range(100): run()
    torch.cuda.synchronize()
    ms, runs = chain.read_timing()
    chain.enable_timing(False)
    print(nb, f"wall {wall:.1f} us", {k: round(v * 1e3, 1) for k, v in ms.items() if v}, "sum", round(sum(ms.values()) * 1e3, 1), "ideal from 65536:", round(968.9 * nb / 65536, 1))
