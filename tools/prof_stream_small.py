"""sdrhip_fm_stream with one 8192-sample source block per push (for rocprofv3 --hip-trace --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sdr_amd.lib as L
import signals as S

def main():
    B = 8192
    bpp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
    st = L.FmStream(chain, bpp * B, B)
    x = np.random.default_rng(2).integers(0, 256, 2 * bpp * B, dtype=np.uint8)
    for _ in range(50):
        st.push(x)
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        st.push_inplace(st.input_buffer(bpp * B))
    st.flush()
    dt = time.perf_counter() - t0
    print(f"{bpp} blocks/push: {dt / n * 1e6:.1f} us/push, {n * bpp * B / dt / 1e6:.1f} Msamples/s")


if __name__ == "__main__":
    main()
