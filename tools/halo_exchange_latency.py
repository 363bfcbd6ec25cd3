"""Cost of one halo exchange call (8 KB of u8 IQ) through the library's transports, as far as a 1-GPU box can show it:
RCCL send/recv to self (a ring of one: the launch / proxy overhead of ncclGroupStart..End, not the link) and the
single-process peer copy between two ranks that share the device.  Run on a GPU box: python tools/halo_exchange_latency.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import sdr_amd.lib as L
    import signals as S
    ch = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    halo = ch.halo_samples()
    shard = 1 << 20
    buf = torch.zeros(2 * (shard + halo), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()

    def timed(fn, n=2000):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / n * 1e6
        return host, total

    comm = L.Comm(1, 0, L.comm_unique_id())
    h, t = timed(lambda: comm.chain_halo_exchange(ch, buf.data_ptr(), shard, stream=st.cuda_stream))
    print(f"RCCL send/recv to self, {2 * halo} bytes: {h:.1f} us of host time per call, {t:.1f} us per call drained")
    comm.close()
    comms = L.Comm.local([0, 0], L.TRANSPORT_PEER_COPY)
    b2 = torch.zeros_like(buf)
    s2 = torch.cuda.Stream()
    h, t = timed(lambda: L.halo_exchange_all(comms, [st.cuda_stream, s2.cuda_stream], [buf.data_ptr(), b2.data_ptr()],
                                             [buf.data_ptr() + 2 * shard, b2.data_ptr() + 2 * shard], 2 * halo))
    print(f"peer copy, two ranks on one device, {2 * halo} bytes each way: {h:.1f} us of host time per call, {t:.1f} us per call drained")
    for c in comms:
        c.close()


if __name__ == "__main__":
    main()
