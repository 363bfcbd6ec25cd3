"""Host-streamed throughput measured by the library's own C timing loops (sdrhip_bench_fm_stream / sdrhip_bench_pipe):
what a compiled caller pays per push, without the Python interpreter in the loop."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def fm_stream_rate(L, chain, n_samples, pushes, zero_copy=True, coalesce=0):
    sps, blocks = C.c_double(), C.c_longlong()
    L.check(L.lib.sdrhip_bench_fm_stream(chain.h, n_samples, pushes, int(zero_copy), coalesce, C.byref(sps), C.byref(blocks)),
            "sdrhip_bench_fm_stream")
    return sps.value, blocks.value


def fm_stream_latency(L, chain, n_samples, pushes, pace_us=0.0, adaptive_off=False):
    """p50 / p99 / max / mean push-to-audio latency and the mean push call, microseconds (sdrhip_bench_fm_stream_latency)."""
    out = (C.c_double * 5)()
    L.check(L.lib.sdrhip_bench_fm_stream_latency(chain.h, n_samples, pushes, float(pace_us), int(adaptive_off), out), "sdrhip_bench_fm_stream_latency")
    return {"p50_us": round(out[0], 1), "p99_us": round(out[1], 1), "max_us": round(out[2], 1), "mean_us": round(out[3], 1),
            "push_call_us": round(out[4], 2)}


def pipe_rate(L, pipe_handle, n, floats_per_element, block_out, pushes, zero_copy=True):
    eps = C.c_double()
    L.check(L.lib.sdrhip_bench_pipe(pipe_handle, n, floats_per_element, block_out, pushes, int(zero_copy), C.byref(eps)), "sdrhip_bench_pipe")
    return eps.value


def main():
    import sdr_amd.lib as L
    import signals as S
    B = 8192
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    # default = adaptive submission (sdrhip_fm_stream_set_adaptive): pushes pile up in the staging buffer while the GPU is busy
    for bpp, pushes in ((1, 20000), (2, 10000), (4, 5000), (16, 2000), (256, 100), (4096, 12)):
        for zc in (False, True):
            sps, blocks = fm_stream_rate(L, chain, bpp * B, pushes, zc)
            print(f"sdrhip_fm_stream {bpp:5d} block(s)/push, {'zero-copy' if zc else 'memcpy   '}: {sps / 1e6:9.1f} Msamples/s "
                  f"({bpp * B / sps * 1e6:7.2f} us/push, {blocks} audio blocks)")
    for bpp, pushes in ((1, 4000), (2, 3000), (4, 2000), (16, 1000)):
        for zc in (False, True):
            sps, blocks = fm_stream_rate(L, chain, bpp * B, pushes, zc, coalesce=1)
            print(f"sdrhip_fm_stream {bpp:5d} block(s)/push, {'zero-copy' if zc else 'memcpy   '}, every push its own launch: {sps / 1e6:9.1f} Msamples/s "
                  f"({bpp * B / sps * 1e6:7.2f} us/push, {blocks} audio blocks)")
    for zc in (False, True):
        sps, blocks = fm_stream_rate(L, chain, B, 20000, zc, coalesce=-8 * B)
        print(f"sdrhip_fm_stream     1 block(s)/push, {'zero-copy' if zc else 'memcpy   '}, adaptive <=  8 blocks: {sps / 1e6:9.1f} Msamples/s "
              f"({B / sps * 1e6:7.2f} us/push, {blocks} audio blocks)")
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    res = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
    fil = L.Filter(S.taps_audio_half64(), L.ORDER_AVX, sym=True)
    for name, mk, n, fpe in (("firDecimator /8 127 taps, 8192-sample cfloat blocks", lambda: L.Pipe("decimator", dec, B), B, 2),
                             ("firResampler 3/10 191 taps, 65536-float blocks (configs[3])", lambda: L.Pipe("resampler", res, B), 65536, 1),
                             ("firFilter 64 half-taps sym, 8192-float blocks", lambda: L.Pipe("filter", fil, B), B, 1)):
        for zc in (False, True):
            p = mk()
            r = pipe_rate(L, p.h, n, fpe, B, 20000 if n <= B else 4000, zc)
            print(f"{name}, {'zero-copy' if zc else 'memcpy   '}: {r / 1e6:8.1f} M elements/s ({n / r * 1e6:6.2f} us/push)")
        for zc in (False, True):
            p = mk()
            p.set_adaptive(0)
            r = pipe_rate(L, p.h, n, fpe, B, 2000, zc)
            print(f"{name}, {'zero-copy' if zc else 'memcpy   '}, every push its own launch: {r / 1e6:8.1f} M elements/s ({n / r * 1e6:6.2f} us/push)")
    # the receiver composed of four Level-1 Pipes, as fm.hs composes it
    sps, blocks = C.c_double(), C.c_longlong()
    L.check(L.lib.sdrhip_bench_fm_pipes(dec.h, res.h, fil.h, B, 20000, C.byref(sps), C.byref(blocks)), "sdrhip_bench_fm_pipes")
    print(f"firDecimator -> fmDemod -> firResampler -> firFilter as four Pipes, 8192-sample cfloat source blocks: {sps.value / 1e6:8.1f} Msamples/s "
          f"({B / sps.value * 1e6:6.2f} us per source block, {blocks.value} audio blocks)")
    # the map pipes (one output block per input block): fmDemod on 8192-sample cfloat blocks, dcBlockingFilter on 8192 floats
    for name, kind, n, fpe in (("fmDemod Pipe, 8192-sample cfloat blocks", "fm_demod", B, 2), ("dcBlockingFilter Pipe, 8192-float blocks", "dc_blocker", B, 1)):
        p = L.Pipe(kind)
        r = pipe_rate(L, p.h, n, fpe, B, 4000, False)
        print(f"{name}: {r / 1e6:8.1f} M elements/s ({n / r * 1e6:6.2f} us/push)")


if __name__ == "__main__":
    main()
