"""oracle/cpu_chain_bench (bench.py's cpu_baseline as a compiled caller): the audio it produces block by block equals the
restated Pipes' (oracle/pipes_model.py) bit for bit -- so what it times IS the reference's receiver loop -- and its timing
mode reports a plausible rate."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "oracle", "cpu_chain_bench")

from conftest import assert_bit_equal  # noqa: E402
from oracle import pipes_model as PM  # noqa: E402
import signals as S  # noqa: E402


def write_taps(path):
    d, r, h = S.taps_decim127(), S.taps_resamp191(), S.taps_audio_half64()
    with open(path, "wb") as f:
        np.array([d.size, r.size, h.size], np.int32).tofile(f)
        for a in (d, r, h):
            np.ascontiguousarray(a, np.float32).tofile(f)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return EXE


@pytest.mark.parametrize("fm", [False, True])
def test_dump_equals_the_restated_pipes(exe, oracle, tmp_path, fm):
    nblk, B = 200, 8192
    u8 = (S.iq_u8_fm if fm else S.iq_u8)(nblk * B)
    taps, fin, fout = tmp_path / "taps.bin", tmp_path / "in.u8", tmp_path / "out.f32"
    write_taps(taps)
    u8.tofile(fin)
    out = subprocess.run([exe, str(taps), "--dump", str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    info = json.loads(out.stdout.strip().splitlines()[-1])
    got = np.fromfile(fout, np.float32)
    blocks = [u8[2 * i * B:2 * (i + 1) * B] for i in range(nblk)]
    exp = np.concatenate(PM.fm_receiver(oracle, blocks, S.taps_decim127(), 8, S.taps_resamp191(), 3, 10, S.taps_audio_half64(), 0.2, B))
    assert info["audio_blocks"] * B == got.size == exp.size and exp.size >= 5 * B
    assert_bit_equal(got, exp, "compiled receiver loop vs restated Pipes")


def test_timing_mode(exe, tmp_path):
    taps = tmp_path / "taps.bin"
    write_taps(taps)
    out = subprocess.run([exe, str(taps), "0.3", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["threads"] == 2 and r["kind"] in ("reference", "port")
    assert r["sps_total"] > 2e6 and r["sps_slowest_thread"] > 1e6
