"""The chain at BASELINE configs[4]'s shard size (2^20 samples per pass, or argv[1] blocks): microseconds per pass on one
stream, from a hipGraph, and with 2 / 4 passes in flight on separate streams.  Under rocprofv3 --kernel-trace --stats it
gives the per-kernel durations of the small launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdr_amd.lib as L
import signals as S
from sdr_amd import sharding

BLOCK = 8192


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    S_len = blocks * BLOCK
    mk = lambda: L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), gain=0.2, block=BLOCK)
    chain = mk()
    plan = sharding.ShardPlan(chain, 0, 1, S_len)
    buf = torch.randint(0, 256, (2 * (S_len + plan.halo_cap),), dtype=torch.uint8, device="cuda")
    nq = plan.q1 - plan.q0
    ws_bytes = chain.workspace_bytes(S_len + plan.halo_cap)

    def bench(nstream, graph):
        chains = [mk() for _ in range(nstream)]
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        auds = [torch.empty(nq, dtype=torch.float32, device="cuda") for _ in range(nstream)]
        wss = [torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") for _ in range(nstream)]
        gs = [L.FmGraph(c, buf.data_ptr(), plan.s0, plan.n_in, a.data_ptr(), plan.q0, plan.q1, w.data_ptr(), ws_bytes)
              for c, a, w in zip(chains, auds, wss)] if graph else None

        def one(i):
            j = i % nstream
            if graph:
                gs[j].launch(streams[j].cuda_stream)
            else:
                chains[j].run(buf.data_ptr(), plan.s0, plan.n_in, auds[j].data_ptr(), plan.q0, plan.q1, wss[j].data_ptr(), ws_bytes,
                              stream=streams[j].cuda_stream)
        for i in range(200):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            one(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ref = auds[0].clone()
        same = all(torch.equal(a, ref) for a in auds)
        return dt * 1e6, same

    for nstream in (1, 2, 4):
        for graph in (False, True):
            us, same = bench(nstream, graph)
            print(f"blocks {blocks}: {nstream} stream(s){' graph' if graph else '      '}: {us:7.2f} us/pass = {S_len / us / 1e3:7.1f} Gsample/s  same_audio={same}", flush=True)


if __name__ == "__main__":
    main()
