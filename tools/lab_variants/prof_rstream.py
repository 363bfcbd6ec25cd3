"""A few full-size passes of the FM chain with the streaming fmDemod + resampler (mode 1) and with the tile kernel (mode 0), for
rocprofv3 runs of those two kernels:  rocprofv3 --kernel-trace --stats -- python tools/prof_rstream.py [passes]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, sdr_amd.lib as L, signals as S
B = 8192
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
n = 1 << 29
u8 = torch.randint(0, 256, (2 * (n + 8192),), dtype=torch.uint8, device="cuda")
q0, q1, halo = chain.plan(0, n, -1)
ws_bytes = chain.workspace_bytes(n + 8192); ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
out = torch.empty(q1 - q0, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for mode in (1, 1001, 0):          # streaming kernel with the prefetch in registers / in LDS (global_load_lds) / the tile kernel
    L.lib.sdrhip_debug_set_resample_demod_stream(mode)
    for _ in range(passes):
        chain.run(u8.data_ptr(), 0, n + halo, out.data_ptr(), q0, q1, ws.data_ptr(), ws_bytes, stream=st)
    torch.cuda.synchronize()
print("stream launches", L.lib.sdrhip_debug_resample_demod_stream_launches())
