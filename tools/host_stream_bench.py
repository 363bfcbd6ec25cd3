"""PCIe-inclusive rates (never bench.py's `value`): (a) the whole FM chain with pinned, double-buffered
H2D/D2H around the device-resident chain, (b) the Pipe operators on 8192-sample host blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sdr_amd.lib as L
import signals as S

B = 8192


def chain_streamed(blocks_per_batch=8192, batches=12):
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    n = blocks_per_batch * B
    halo = (chain.max_halo() + 7) // 8 * 8
    q0, q1, _ = chain.plan(0, n, -1)
    host_in = [torch.randint(0, 256, (2 * (n + halo),), dtype=torch.uint8).pin_memory() for _ in range(2)]
    host_out = [torch.empty(q1 - q0, dtype=torch.float32).pin_memory() for _ in range(2)]
    dev_in = [torch.empty(2 * (n + halo), dtype=torch.uint8, device="cuda") for _ in range(2)]
    dev_out = [torch.empty(q1 - q0, dtype=torch.float32, device="cuda") for _ in range(2)]
    wsb = chain.workspace_bytes(n + halo)
    ws = [torch.empty(wsb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]

    def submit(i):
        s = streams[i & 1]
        with torch.cuda.stream(s):
            dev_in[i & 1].copy_(host_in[i & 1], non_blocking=True)
            chain.run(dev_in[i & 1].data_ptr(), 0, n + halo, dev_out[i & 1].data_ptr(), q0, q1, ws[i & 1].data_ptr(), wsb,
                      stream=s.cuda_stream)
            host_out[i & 1].copy_(dev_out[i & 1], non_blocking=True)

    for i in range(2):
        submit(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(batches):
        streams[i & 1].synchronize()
        submit(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"chain host-streamed (pinned, 2 streams, {n} samples/batch): {batches * n / dt / 1e6:10.1f} Msamples/s "
          f"({batches * 2 * n / dt / 1e9:.1f} GB/s H2D)")


def fm_stream(blocks_per_push, pushes):
    """The C-ABI streaming operator (sdrhip_fm_stream_*): pinned staging + 3 HIP streams inside the library."""
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    st = L.FmStream(chain, blocks_per_push * B, B)
    x = np.random.default_rng(2).integers(0, 256, 2 * blocks_per_push * B, dtype=np.uint8)
    for _ in range(4):
        st.push(x)
    t0 = time.perf_counter()
    nout = 0
    for _ in range(pushes):
        nout += len(st.push(x))
    nout += len(st.flush())
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(pushes):
        nout += len(st.push_inplace(st.input_buffer(blocks_per_push * B)))   # source writes straight into pinned memory
    nout += len(st.flush())
    dz = time.perf_counter() - t0
    print(f"sdrhip_fm_stream, {blocks_per_push:5d} source blocks/push: {pushes * blocks_per_push * B / dt / 1e6:10.1f} Msamples/s "
          f"({dt / pushes * 1e6:.0f} us/push); zero-copy input {pushes * blocks_per_push * B / dz / 1e6:10.1f} Msamples/s")


def fm_stream_coalesced(coalesce_blocks, pushes=4000):
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    st = L.FmStream(chain, B, B)
    st.set_coalesce(coalesce_blocks * B)
    for _ in range(4 * coalesce_blocks):
        st.push_inplace(st.input_buffer(B))
    t0 = time.perf_counter()
    for _ in range(pushes):
        st.push_inplace(st.input_buffer(B))
    st.flush()
    dt = time.perf_counter() - t0
    print(f"sdrhip_fm_stream, 8192-sample pushes coalesced x{coalesce_blocks:3d}: {pushes * B / dt / 1e6:10.1f} Msamples/s "
          f"({dt / pushes * 1e6:.1f} us/push incl. the ctypes call)")


def pipe_blocks(nblocks=512):
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    pipe = L.firDecimator(dec, B)
    x = np.random.default_rng(0).uniform(-1, 1, 2 * B).astype(np.float32)
    for _ in range(8):
        pipe.push(x)
    t0 = time.perf_counter()
    for _ in range(nblocks):
        pipe.push(x)
    pipe.flush()
    dt = time.perf_counter() - t0
    print(f"firDecimator Pipe on 8192-sample host blocks: {nblocks * B / dt / 1e6:10.1f} Msamples/s ({dt / nblocks * 1e6:.1f} us/block)")
    big = np.random.default_rng(0).uniform(-1, 1, 2 * B * 1024).astype(np.float32)
    pipe2 = L.firDecimator(dec, B)
    for _ in range(2):
        pipe2.push(big)
    t0 = time.perf_counter()
    for _ in range(8):
        pipe2.push(big)
    pipe2.flush()
    dt = time.perf_counter() - t0
    print(f"firDecimator Pipe on 8Mi-sample host blocks:  {8 * B * 1024 / dt / 1e6:10.1f} Msamples/s")


def resampler_pipe_cfg4(nblocks=256):
    """BASELINE configs[3]: fastResamplerR 3/10, 191 taps, 65536-sample blocks streamed (async double-buffered)."""
    r = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
    pipe = L.firResampler(r, 8192)
    x = np.random.default_rng(1).uniform(-1, 1, 65536).astype(np.float32)
    for _ in range(8):
        pipe.push(x)
    t0 = time.perf_counter()
    for _ in range(nblocks):
        pipe.push(x)
    pipe.flush()
    dt = time.perf_counter() - t0
    print(f"firResampler Pipe 3/10 on 65536-sample host blocks: {nblocks * 65536 / dt / 1e6:10.1f} Msamples/s ({dt / nblocks * 1e6:.1f} us/block)")


def pipes_coalesced():
    """Level-1 operators fed the reference's block sizes, coalesced inside the operator, zero-copy staging."""
    cases = [("firDecimator /8 128 taps, 8192-sample cfloat blocks", lambda: L.firDecimator(L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True), B), 8192, 2),
             ("firResampler 3/10 191 taps, 65536-float blocks (configs[3])", lambda: L.firResampler(L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX), 8192), 65536, 1),
             ("firFilter 64 half-taps symmetric, 8192-float blocks", lambda: L.firFilter(L.Filter(S.taps_audio_half64(), L.ORDER_AVX, sym=True), B), 8192, 1)]
    for name, mk, n, w in cases:
        for co in (1, 16, 128):
            pipe = mk()
            if co > 1:
                pipe.set_coalesce(co)
            src = np.random.default_rng(5).uniform(-1, 1, n * w).astype(np.float32)
            for _ in range(2 * co + 4):
                pipe.push(src)
            pushes = max(256, 8 * co)
            t0 = time.perf_counter()
            for _ in range(pushes):
                v = pipe.input_buffer(n)
                v[:] = src                       # the source writes into pinned memory (here: a host memcpy)
                pipe.push(v)
            pipe.flush()
            dt = time.perf_counter() - t0
            print(f"{name}, coalesce {co:3d}: {pushes * n / dt / 1e6:9.1f} M elements/s ({dt / pushes * 1e6:.1f} us/push)")


if __name__ == "__main__":
    print(L.device_name())
    chain_streamed()
    for bpp, pushes in ((1, 2000), (16, 1000), (256, 200), (4096, 24)):
        fm_stream(bpp, pushes)
    for cb in (4, 16, 64):
        fm_stream_coalesced(cb)
    pipe_blocks()
    pipes_coalesced()
    resampler_pipe_cfg4()
