import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tools')
import sdr_amd.lib as L
import signals as S
import host_stream_native as H
B = 8192
for bpp, pushes in ((1, 20000), (4, 5000), (16, 1500), (32, 800)):
    for tile in (0, 159, 96):
        chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
        chain.set_small_chain(2, 0, tile)
        sps, blocks = H.fm_stream_rate(L, chain, bpp * B, pushes, True)
        print(f"{bpp:3d} blocks/push tile {tile:3d}: {sps / 1e6:9.1f} Msamples/s ({bpp * B / sps * 1e6:7.2f} us/push)", flush=True)
