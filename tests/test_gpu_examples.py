"""The plain-C replay example (examples/fm_replay.c) end to end: a u8 IQ capture file in, an audio
file out, through nothing but the C ABI -- compared bit for bit with the restated reference pipeline."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pipes_model as PM
import signals as S
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "bin", "fm_replay")
B = 8192


def _ensure_exe():
    if not os.path.exists(EXE):
        from sdr_amd import build as B
        B.build()                      # builds the library if needed and the plain-C examples (gcc)
    assert os.path.exists(EXE), "examples/bin/fm_replay missing: run `python -m sdr_amd.build`"


@pytest.mark.parametrize("blocks_per_push", [1, 7, 64])
def test_fm_replay_matches_reference_pipeline(tmp_path, oracle, blocks_per_push):
    _ensure_exe()
    nblk = 100
    u8 = S.iq_u8_fm(nblk * B)
    cap = tmp_path / "capture.u8"
    u8.tofile(cap)
    S.taps_decim127().tofile(str(cap) + ".decim.f32")
    S.taps_resamp191().tofile(str(cap) + ".resamp.f32")
    S.taps_audio_half64().tofile(str(cap) + ".audio_half.f32")
    out = tmp_path / "audio.f32"
    r = subprocess.run([EXE, str(cap), str(out), str(blocks_per_push)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32)
    blocks = [u8[2 * i * B: 2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10,
                                        S.taps_audio_half64(), 0.2, B, PM.ORDER_AVX))
    assert exp.size >= 2 * B
    assert got.size >= exp.size and got.size % B == 0      # the stream may be one block ahead of the four chained Pipes
    assert_bit_equal(got[: exp.size], exp, "fm_replay audio")


def test_library_before_torch_in_one_process():
    """`build()` imports the package (and with it libsdr_hip.so) before `smoke()` imports torch: both must see the GPU
    (one shared HIP runtime, sdr_amd/lib.py:_share_torch_hip_runtime)."""
    import sys
    code = ("import sdr_amd.lib as L\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "x = torch.arange(8, dtype=torch.float32, device='cuda')\n"
            "import __graft_entry__ as g\n"
            "g.smoke()\n"
            "maps = open('/proc/self/maps').read()\n"
            "libs = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}\n"
            "assert len(libs) == 1, libs\n"
            "print('ok', L.device_name())\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def _udp_attempt(tmp_path, u8, prefix, attempt):
    import socket
    import time
    _ensure_exe()
    out = tmp_path / f"audio_udp_{attempt}.f32"
    probe = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    proc = subprocess.Popen([EXE, f"udp:{port}", str(out), "4", prefix], stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 120
        line = ""
        while "listening" not in line:
            line = proc.stderr.readline()
            assert line or proc.poll() is None, "fm_replay exited before listening"
            assert time.time() < deadline
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        data = u8.tobytes()
        for i in range(0, len(data), 4096):                 # 4 datagrams per source block, paced
            tx.sendto(data[i:i + 4096], ("127.0.0.1", port))
            if (i // 4096) % 4 == 3:
                time.sleep(0.002 * (attempt + 1))
        time.sleep(0.05)
        tx.sendto(b"", ("127.0.0.1", port))                 # end of stream
        tx.close()
        rc = proc.wait(timeout=120)
        err = proc.stderr.read()
    finally:
        if proc.poll() is None:
            proc.kill()
    assert rc == 0, err
    return np.fromfile(out, np.float32)


def test_fm_replay_udp_source(tmp_path, oracle):
    """The same receiver fed from UDP datagrams on the loopback interface (the reference's udpSource,
    NetworkStream.hs:28-35): reassembled into source blocks, same audio.  UDP may drop a datagram under load (then every
    later sample is shifted): up to three attempts, slower each time."""
    nblk = 100
    u8 = S.iq_u8_fm(nblk * B)
    prefix = str(tmp_path / "taps")
    S.taps_decim127().tofile(prefix + ".decim.f32")
    S.taps_resamp191().tofile(prefix + ".resamp.f32")
    S.taps_audio_half64().tofile(prefix + ".audio_half.f32")
    blocks = [u8[2 * i * B: 2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10,
                                        S.taps_audio_half64(), 0.2, B, PM.ORDER_AVX))
    for attempt in range(3):
        got = _udp_attempt(tmp_path, u8, prefix, attempt)
        if got.size >= exp.size and np.array_equal(got[: exp.size].view(np.uint32), exp.view(np.uint32)):
            return
    assert got.size >= exp.size
    assert_bit_equal(got[: exp.size], exp, "fm_replay audio from UDP (third attempt)")
