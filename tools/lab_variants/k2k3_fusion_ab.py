"""A/B of fmDemod inside the systolic decimator (sdrhip_fm_chain_set_decim_demod_fusion) against the two kernels: alternating rounds in one
process, 2^29 samples per pass, per-stage HIP-event times.  Run on a GPU box."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, sdr_amd.lib as L, signals as S
B = 8192
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
n = 1 << 29
u8 = torch.randint(0, 256, (2 * (n + 8192),), dtype=torch.uint8, device="cuda")
q0, q1, halo = chain.plan(0, n, -1)
ws_bytes = chain.workspace_bytes(n + 8192); ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
out = torch.empty(q1 - q0, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(k):
    for _ in range(k): chain.run(u8.data_ptr(), 0, n + halo, out.data_ptr(), q0, q1, ws.data_ptr(), ws_bytes, stream=st)
run(50); torch.cuda.synchronize()
knob = sys.argv[1] if len(sys.argv) > 1 else "decim_demod"      # decim_demod | resamp_demod | resamp_stream (fmDemod in the resampler's loader) | overlap-free knobs only
setter = {"decim_demod": chain.set_decim_demod_fusion, "resamp_demod": chain.set_demod_fusion,
          "resamp_stream": lambda on: L.lib.sdrhip_debug_set_resample_demod_stream(1 if on else 0),
          "resamp_stream_dma": lambda on: L.lib.sdrhip_debug_set_resample_demod_stream(1001 if on else 0),   # the LDS-DMA variant
          "resamp_stream_dma3": lambda on: L.lib.sdrhip_debug_set_resample_demod_stream(2001 if on else 0),   # round 5: streaming fmDemod + resampler vs the tile kernel
          "fused_tail1": lambda on: chain.set_fused_tail(1 if on else 2), "fused_tail3": lambda on: chain.set_fused_tail(3 if on else 2),
          "nsub2": lambda on: chain.set_pipelining(2 if on else 1), "nsub4": lambda on: chain.set_pipelining(4 if on else 1),
          "nsub8": lambda on: chain.set_pipelining(8 if on else 1)}[knob]
print("knob:", knob)
for rnd in range(3):
    for on in (1, 0):
        setter(bool(on))
        run(20); torch.cuda.synchronize()
        chain.enable_timing(True)
        t0 = time.perf_counter(); run(300); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        ms, _ = chain.read_timing(); chain.enable_timing(False)
        if rnd == 0 and on == 0: print("systolic launches so far: decimator", L.lib.sdrhip_debug_systolic_launches(), "resampler", L.lib.sdrhip_debug_resample_systolic_launches())
        print(f"fusion {on}: {dt*1e3:.4f} ms/pass  {n/dt/1e9:.1f} Gsamples/s  stages {ms['decimate']:.4f} {ms['fm_demod']:.4f} {ms['resample']:.4f} {ms['filter']:.4f} tail {ms.get('fused_tail', 0.0):.4f}")
