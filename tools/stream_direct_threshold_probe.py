"""Host-block operator: in-place pushes (kernels read pinned host memory over PCIe) against the copy-engine path, by push size.
SDRHIP_DIRECT_SAMPLES sets the in-place bound (default 33 blocks); run with a large and a small value to see both routes at every size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sdr_amd.lib as L
import signals as S
import host_stream_native as H
B = 8192
for bpp in [int(x) for x in os.environ.get("PROBE_BPP", "8,16,24,32,48,64,96,128,256").split(",")]:
    for zc in (True, False):
        chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
        sps, _ = H.fm_stream_rate(L, chain, bpp * B, max(60, 24000 // bpp), zc)
        print(f"{bpp:4d} blocks/push {'zero-copy' if zc else 'memcpy   '}: {sps / 1e6:9.1f} Msamples/s ({bpp * B / sps * 1e6:8.2f} us/push)", flush=True)
