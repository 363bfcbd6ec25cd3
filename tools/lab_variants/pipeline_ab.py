import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, sdr_amd.lib as L, signals as S

def main():
    n = 1 << 29
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    u8 = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda")
    q0, q1, _ = chain.plan(0, n, -1)
    wsb = chain.workspace_bytes(n); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda"); out = torch.empty(q1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run(nsub, timing, reps=10):
        chain.set_pipelining(nsub); chain.enable_timing(timing)
        for _ in range(2): chain.run(u8.data_ptr(), 0, n - 8192, out.data_ptr(), 0, q1 - 4000, ws.data_ptr(), wsb, stream=st)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): chain.run(u8.data_ptr(), 0, n - 8192, out.data_ptr(), 0, q1 - 4000, ws.data_ptr(), wsb, stream=st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        chain.read_timing()
        return dt * 1e3
    res = {}
    for rnd in range(5):
        for key in ((1, False), (8, False), (4, False), (1, True), (8, True), (4, True)):
            res.setdefault(key, []).append(run(*key))
    for k, v in res.items():
        print(k, " ".join(f"{x:.3f}" for x in v), " median %.3f" % sorted(v)[len(v)//2])


if __name__ == "__main__":
    main()
