"""Round 5 (VERDICT r04 "missing" 5; SURVEY 5: the reference itself over-reads at c_sources/convert.c:27,42): the host code of the
library -- pinned rings, leased streams, lent buffers, checkpoints, the copy helper threads -- under AddressSanitizer and
UndefinedBehaviorSanitizer.  `python -m sdr_amd.build --asan` builds libsdr_hip_asan.so (sanitizers on the host side only) and the
plain-C programs of examples/ against it; here they run on the GPU: the replay example from a file (1 / 7 / 64 source blocks per
push) and the soak driver of the host-block operators (ragged blocks, lent buffers, coalesced / adaptive submission, save / restore,
destroy with results pending).  Any sanitizer report is a failure; the audio of the replay runs is still compared bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import signals as S
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "bin")
B = 8192
# leak detection off: the HIP runtime keeps its own allocations for the life of the process; everything else stays on
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _ensure():
    if not os.path.exists(os.path.join(BIN, "pipes_soak_asan")):
        from sdr_amd import build as Bd
        Bd.build_asan()
    assert os.path.exists(os.path.join(BIN, "pipes_soak_asan"))


def _clean(r, what):
    log = r.stdout + r.stderr
    assert "AddressSanitizer" not in log and "runtime error:" not in log and "UndefinedBehaviorSanitizer" not in log, f"{what}:\n{log[-4000:]}"
    assert r.returncode == 0, f"{what}: exit {r.returncode}\n{log[-4000:]}"


def test_soak_of_the_host_block_operators_is_clean():
    _ensure()
    r = subprocess.run([os.path.join(BIN, "pipes_soak_asan")], capture_output=True, text=True, timeout=1200, env=ENV)
    _clean(r, "pipes_soak_asan")
    assert "pipes_soak: ok" in r.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "asan_pipes_soak.log"), "w") as f:
        f.write("$ ASAN_OPTIONS=" + ENV["ASAN_OPTIONS"] + " examples/bin/pipes_soak_asan\n" + r.stdout + r.stderr + f"exit {r.returncode}\n")


@pytest.mark.parametrize("blocks_per_push", [1, 7, 64])
def test_replay_from_a_file_is_clean_and_still_right(tmp_path, oracle, blocks_per_push):
    from oracle import pipes_model as PM
    _ensure()
    nblk = 100
    u8 = S.iq_u8_fm(nblk * B)
    cap = tmp_path / "capture.u8"
    u8.tofile(cap)
    S.taps_decim127().tofile(str(cap) + ".decim.f32")
    S.taps_resamp191().tofile(str(cap) + ".resamp.f32")
    S.taps_audio_half64().tofile(str(cap) + ".audio_half.f32")
    out = tmp_path / "audio.f32"
    r = subprocess.run([os.path.join(BIN, "fm_replay_asan"), str(cap), str(out), str(blocks_per_push)], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r, f"fm_replay_asan, {blocks_per_push} blocks per push")
    got = np.fromfile(out, np.float32)
    blocks = [u8[2 * i * B: 2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10, S.taps_audio_half64(), 0.2, B, PM.ORDER_AVX))
    assert got.size >= exp.size
    assert_bit_equal(got[: exp.size], exp, "fm_replay_asan audio")
    with open(os.path.join(ROOT, "gpurun_out", f"asan_fm_replay_{blocks_per_push}.log"), "w") as f:
        f.write(f"$ examples/bin/fm_replay_asan capture.u8 audio.f32 {blocks_per_push}\n" + r.stderr + f"exit {r.returncode}\n")


def test_replay_over_udp_is_clean(tmp_path):
    """The UDP source (NetworkStream.hs:28-35's counterpart) under the sanitizers: datagrams of 4096 bytes, a zero-length one ends it."""
    import socket
    import time
    _ensure()
    nblk = 40
    u8 = S.iq_u8_fm(nblk * B)
    prefix = str(tmp_path / "taps")
    S.taps_decim127().tofile(prefix + ".decim.f32")
    S.taps_resamp191().tofile(prefix + ".resamp.f32")
    S.taps_audio_half64().tofile(prefix + ".audio_half.f32")
    out = tmp_path / "audio_udp.f32"
    probe = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    probe.bind(("127.0.0.1", 0))
    port = probe.getsockname()[1]
    probe.close()
    proc = subprocess.Popen([os.path.join(BIN, "fm_replay_asan"), f"udp:{port}", str(out), "4", prefix], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=ENV)
    try:
        deadline = time.time() + 300
        line = ""
        while "listening" not in line:
            line = proc.stderr.readline()
            assert line or proc.poll() is None, "fm_replay_asan exited before listening"
            assert time.time() < deadline
        tx = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        raw = u8.tobytes()
        for i in range(0, len(raw), 4096):
            tx.sendto(raw[i:i + 4096], ("127.0.0.1", port))
            if (i // 4096) % 8 == 7:
                time.sleep(0.002)
        for _ in range(3):
            tx.sendto(b"", ("127.0.0.1", port))
        so, se = proc.communicate(timeout=300)
    finally:
        if proc.poll() is None:
            proc.kill()

    class R:
        stdout, stderr, returncode = so, line + se, proc.returncode
    _clean(R, "fm_replay_asan over UDP")
    assert np.fromfile(out, np.float32).size > 0
