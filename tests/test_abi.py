"""The C-ABI boundary without a GPU: libsdr_hip.so loads, exports every symbol that
include/sdr_hip.h declares (and every one of the reference's native symbols the hot
path's FFI binds), descriptors and planning work on the host, and anything that
needs the device fails LOUDLY instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import signals as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sdr_hip.h")

# the reference's exported native surface (cpuid.c excluded: x86 only)
REFERENCE_SYMBOLS = """filterRR filterSSERR filterAVXRR filterSSESymmetricRR filterAVXSymmetricRR filterRC filterSSERC
filterSSERC2 filterAVXRC filterAVXRC2 filterSSESymmetricRC filterAVXSymmetricRC dcBlocker decimateRR decimateSSERR
decimateAVXRR decimateSSESymmetricRR decimateAVXSymmetricRR decimateRC decimateSSERC decimateSSERC2 decimateAVXRC
decimateAVXRC2 decimateSSESymmetricRC decimateAVXSymmetricRC resampleRR resample2RR resampleSSERR resampleAVXRR
resample2RC resampleSSERC resampleAVXRC convertC convertCSSE convertCAVX convertCBladeRF convertCSSEBladeRF
convertCAVXBladeRF convertBladeRFTransmit scale scaleSSE scaleAVX""".split()


BENCH_HEADER = os.path.join(ROOT, "include", "sdr_hip_bench.h")


def declared_functions(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    skip = {"defined", "sizeof", "void"}      # `void (*handler)(...)`: a function-pointer parameter, not a function
    return sorted({n for n in names if n not in skip and not n.isupper()})


@pytest.fixture(scope="module")
def L():
    from sdr_amd import build as B
    if not os.path.exists(B.LIB):
        B.build()
    import sdr_amd.lib as L
    return L


def test_library_exports_every_declared_symbol(L):
    names = declared_functions()
    assert len(names) > 80
    product = C.CDLL(L.LIB_PATH)                       # a handle of its own: sdr_amd.lib attaches the bench library's names to `lib`
    missing = [n for n in names if not hasattr(product, n)]
    assert not missing, f"declared in sdr_hip.h but not exported: {missing}"


def test_measurement_utilities_live_in_their_own_library(L):
    """Round 6: libsdr_hip.so is the product; bench.py's ceilings and C timing loops (include/sdr_hip_bench.h) are
    libsdr_hip_bench.so, linked against it."""
    names = declared_functions(BENCH_HEADER)
    assert len(names) >= 7 and all(n.startswith("sdrhip_bench_") for n in names), names
    bench = C.CDLL(L.BENCH_LIB_PATH)
    assert not [n for n in names if not hasattr(bench, n)]
    product = C.CDLL(L.LIB_PATH)
    assert not [n for n in names if hasattr(product, n)], "the product library must not export measurement utilities"
    assert all(hasattr(L.lib, n) for n in names)       # ... and the Python driver reaches them through sdr_amd.lib all the same


def test_library_exports_the_reference_native_surface(L):
    missing = [n for n in REFERENCE_SYMBOLS if not hasattr(L.lib, n)]
    assert not missing
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsdr_ref.so")):
        import subprocess
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "oracle", "_ref", "libsdr_ref.so")],
                             capture_output=True, text=True).stdout
        ref_syms = {l.split()[-1] for l in out.splitlines() if " T " in l}
        assert not [s for s in ref_syms if s not in ("cpuid", "cpuid_extended") and not hasattr(L.lib, s)]


def test_no_cpu_fallback_in_the_product():
    """The product never imports / links the oracle."""
    import subprocess
    so = os.path.join(ROOT, "sdr_amd", "lib", "libsdr_hip.so")
    out = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "orc_" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sdr_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libsdr_oracle" not in src, f


def test_descriptors_and_padding_rules(L):
    """numCoeffs as the reference's constructors compute them (Filter.hs:169,244,284,324,422)."""
    t = S.taps_decim127()
    assert L.Decimator(8, t, L.ORDER_AVX, complex_=True).num_coeffs == 128
    assert L.Decimator(8, t, L.ORDER_SSE, complex_=True).num_coeffs == 128
    assert L.Decimator(8, t, L.ORDER_SCALAR, complex_=True).num_coeffs == 127
    assert L.Decimator(8, t, L.ORDER_AVX).num_coeffs == 128
    assert L.Filter(t[:51], L.ORDER_AVX).num_coeffs == 56
    assert L.Filter(t[:51], L.ORDER_SSE).num_coeffs == 52
    assert L.Filter(S.taps_audio_half64(), L.ORDER_AVX, sym=True).num_coeffs == 128
    r = L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_AVX)
    assert r.num_coeffs == 192 and r.num_groups == 3
    assert L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_SSE).num_coeffs == 192
    assert L.Resampler(3, 10, S.taps_resamp191(), L.ORDER_SCALAR).num_coeffs == 192   # roundUp 191 (3*1)
    assert [r.in_offset(m) for m in range(7)] == [0, 4, 7, 10, 14, 17, 20]
    assert [r.filter_offset(m) for m in range(6)] == [0, 2, 1, 0, 2, 1]
    assert [r.group(m) for m in range(6)] == [0, 1, 2, 0, 1, 2]
    r2 = L.Resampler(2, 4, S.gauss_taps(40, 1), L.ORDER_AVX)     # gcd != 1: a single group
    assert r2.num_groups == 1 and r2.in_offset(5) == 10
    with pytest.raises(L.SdrHipError):
        L.Filter(t[:30], L.ORDER_AVX, sym=True)
    with pytest.raises(L.SdrHipError):
        L.Filter(t[:32], L.ORDER_SCALAR, sym=True)           # "At least SSE4.2 required" (Filter.hs:261)
    with pytest.raises(L.SdrHipError):
        L.Resampler(10, 3, t)


def test_chain_planning_on_the_host(L):
    chain = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    total = 100 * 8192
    Q0, Q1, halo = chain.plan(0, total, total)
    assert Q0 == 0 and halo == 0
    K = (total - 128) // 8 + 1
    M = (K * 3 - 192) // 10 + 1
    assert Q1 == M - 127
    # shards partition the outputs; each needs only a right halo bounded by max_halo
    mh = chain.max_halo()
    assert 3000 < mh < 5000
    for n in (2, 3, 8):
        Slen = total // n // 8 * 8
        prev = 0
        for r in range(n):
            s0, s1 = r * Slen, (total if r == n - 1 else (r + 1) * Slen)
            q0, q1, h = chain.plan(s0, s1, total)
            assert q0 == prev and q1 >= q0 and 0 <= h <= mh
            prev = q1
        assert prev == Q1
    with pytest.raises(L.SdrHipError):
        L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 100)  # block < filter


def test_device_calls_fail_loudly_without_a_gpu(L):
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    buf = np.zeros(4096, np.float32)
    with pytest.raises(L.SdrHipError):
        dec.run(buf.ctypes.data, 0, buf.ctypes.data, 0, 4, 0)
    p = C.c_void_p()
    assert L.lib.sdrhip_malloc(C.byref(p), 1024) < 0
    assert b"hipMalloc" in L.lib.sdrhip_last_error()


def test_round2_entry_points_fail_loudly_without_a_gpu(L):
    """The multi-GPU, record-seam and spectrum entry points have no CPU fallback either: without a device they return an
    error and say why (run-time bound RCCL / hipFFT included)."""
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    # the rendezvous id is host-side bootstrap: some RCCL builds hand one out without a device (PyTorch's copy does, ROCm's
    # does not); a communicator, though, needs a GPU whichever copy the loader bound
    try:
        uid = L.comm_unique_id()
    except L.SdrHipError:
        uid = None
    if uid is not None:
        with pytest.raises(L.SdrHipError):
            L.Comm(1, 0, uid)
    with pytest.raises(L.SdrHipError):
        L.Comm.local([0], L.TRANSPORT_PEER_COPY)
    with pytest.raises(L.SdrHipError):
        L.Fft(1024)
    dec = L.Decimator(8, S.taps_decim127(), L.ORDER_AVX, complex_=True)
    x = np.zeros(2 * 8192, np.float32)
    out = np.zeros(2 * 1009, np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert L.lib.sdrhip_decimator_one(dec.h, 1009, fp(x), fp(out)) < 0
    assert L.lib.sdrhip_decimator_cross(dec.h, 15, fp(x), 120, fp(x), 8192, fp(out)) < 0
    # planning arithmetic needs no device: the halo of the FM chain is the composed ntaps-1 overlap, whole 16-byte vectors
    ch = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    h = ch.halo_samples()
    assert h % 8 == 0 and ch.max_halo() <= h < ch.max_halo() + 8


def test_drop_in_failures_reach_the_error_handler_instead_of_abort(L):
    """sdrhip_set_error_handler (round 4): a failure inside a void drop-in symbol calls the handler and returns to the caller
    (default without a handler: print + abort(), which this very test would not survive).  resampleRR's argument check needs
    no GPU: interpolation 0 is outside what the reference's own recurrence can index (resample.c:16-32)."""
    seen = []
    HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_char_p)
    cb = HANDLER(lambda code, msg: seen.append((code, msg.decode())))
    L.lib.sdrhip_set_error_handler.argtypes = [HANDLER]
    L.lib.sdrhip_set_error_handler.restype = None
    L.lib.sdrhip_set_error_handler(cb)
    try:
        x = np.zeros(64, np.float32)
        y = np.full(8, 7.0, np.float32)
        fp = C.POINTER(C.c_float)
        L.lib.resampleRR.argtypes = [C.c_int] * 5 + [fp] * 3
        L.lib.resampleRR.restype = None
        L.lib.resampleRR(8, 4, 0, 3, 0, x.ctypes.data_as(fp), x.ctypes.data_as(fp), y.ctypes.data_as(fp))
        assert len(seen) == 1 and seen[0][0] == -1 and "resampleRR" in seen[0][1], seen       # SDRHIP_ERR_ARG
        assert "resampleRR" in L.lib.sdrhip_last_error().decode()
        assert (y == 7.0).all()                       # returned at once: nothing computed, nothing written
        # without a device every other drop-in symbol fails at its first HIP call: same route
        if L.device_count() == 0:
            L.lib.scale.argtypes = [C.c_int, C.c_float, fp, fp]
            L.lib.scale.restype = None
            L.lib.scale(8, 2.0, x.ctypes.data_as(fp), y.ctypes.data_as(fp))
            assert len(seen) == 2 and seen[1][0] != 0
    finally:
        L.lib.sdrhip_set_error_handler(HANDLER())     # NULL: back to print + abort()


def test_systolic_strip_plan_covers_every_launch_exactly(L):
    """kernels_systolic.hip cuts a launch into wave-strips on the host (240 outputs per strip): whole strips (unguarded loads and
    stores) must lie entirely inside the launch's outputs AND samples, the strips together must cover every output, and the first
    strip that is not whole must really be ragged."""
    import random
    lib = L.lib
    rng = random.Random(4)
    counts = list(range(1, 1500)) + [61440 + d for d in range(-3, 500)] + [rng.randrange(1, 1 << 27) for _ in range(3000)] + [(1 << 26) + d for d in (-1, 0, 1, 14)]
    for count in counts:
        ns, nw = C.c_int(), C.c_int()
        lib.sdrhip_debug_systolic_plan(count, C.byref(ns), C.byref(nw))
        ns, nw = ns.value, nw.value
        avail = (count - 1) * 8 + 128                       # samples of the launch
        assert 0 <= nw <= ns and ns >= 1
        assert 240 * (ns - 1) + 239 >= count - 1, (count, ns)                           # the last strip reaches the last output
        if ns > 1:
            assert 240 * (ns - 2) + 239 < count - 1, (count, ns)                        # no strip is superfluous
        if nw > 0:
            t = nw - 1
            assert 240 * t + 239 <= count - 1, (count, nw)                              # all 240 outputs of a whole strip exist
            assert 1920 * t + 2048 <= avail, (count, nw)                                # and all 2048 samples it loads
        if nw < ns:
            t = nw
            assert 240 * t + 239 > count - 1 or 1920 * t + 2048 > avail, (count, nw)    # the first guarded strip is really ragged


def test_halo_staging_size():
    """The staging size of the batched halo exchange (sdrhip_fm_chain_halo_exchange_batch)."""
    import sdr_amd.lib as L
    import signals as S
    ch = L.FmChain(8, S.taps_decim127(), 3, 10, S.taps_resamp191(), S.taps_audio_half64(), 0.2, 8192)
    h = ch.halo_samples()
    assert ch.halo_staging_bytes(1) == 4 * h and ch.halo_staging_bytes(16) == 64 * h and ch.halo_staging_bytes(0) == 0
