import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools')
import sdr_amd.lib as L, signals as S, host_stream_native as H
B=8192
chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, B)
for bpp,pushes in ((1,20000),(2,10000),(16,1500)):
    for zc in (True, False):
        sps,_=H.fm_stream_rate(L, chain, bpp*B, pushes, zc); print(bpp, "zero-copy" if zc else "memcpy", round(sps/1e6,1))
print("unpaced adaptive", H.fm_stream_latency(L, chain, B, 4000))
print("paced", H.fm_stream_latency(L, chain, B, 300, pace_us=6400.0))
