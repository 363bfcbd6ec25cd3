"""dcBlocker throughput on device-resident data: speculative chunks (several run-in lengths) vs the sequential walk."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sdr_amd.lib as L


def run(n, run_in, use_ws=True, reps=5):
    rng = np.random.default_rng(3)
    t = np.arange(n, dtype=np.float64)
    x = torch.from_numpy((0.3 * np.sin(2 * np.pi * t / 480.0) + 0.05 * rng.standard_normal(n) + 0.4).astype(np.float32)).cuda()
    out = torch.empty_like(x)
    fin = torch.zeros(2, dtype=torch.float32, device="cuda")
    wsb = L.lib.sdrhip_dc_blocker_workspace_bytes(n)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    def go():
        L.check(L.lib.sdrhip_dc_blocker_run(None, x.data_ptr(), out.data_ptr(), n, 0.0, 0.0, fin.data_ptr(),
                                            ws.data_ptr() if use_ws else None, wsb if use_ws else 0, run_in))
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    L.lib.sdrhip_stream_synchronize(None) if hasattr(L.lib, "sdrhip_stream_synchronize") else None
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = ws[:12].cpu().numpy().view(np.uint32)
    return dt, int(st[0]), f"{int(st[1])}/{int(st[2])}"


if __name__ == "__main__":
    print(L.device_name())
    for lg in (14, 16, 18, 20, 22, 24, 26):
        n = 1 << lg
        row = [f"n=2^{lg}"]
        if lg <= 20:
            dt, _, _ = run(n, 0, use_ws=False, reps=2)
            row.append(f"sequential {dt * 1e3:9.3f} ms ({n / dt / 1e6:8.1f} Ms/s)")
        for W in (0, 8192, 4096):
            dt, bad, rew = run(n, W)
            row.append(f"W={W or 12288}: {dt * 1e3:8.3f} ms ({n / dt / 1e6:9.1f} Ms/s, settle {bad}/{rew})")
        print(" | ".join(row))
