"""ctypes loader for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under sdr_amd/ does.

Two libraries:
  * ``Oracle``  -> oracle/libsdr_oracle.so, our restatement (oracle/sdr_oracle.c)
  * ``Ref``     -> oracle/_ref/libsdr_ref.so, the reference's own C sources built
                   unmodified by oracle/Makefile (present only if it was built
                   in a container that has /root/reference; it travels as a .so).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libsdr_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libsdr_ref.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_u8p = C.POINTER(C.c_uint8)


def build():
    """Compile the restatement (and the reference build when its sources exist)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True, capture_output=True)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def have_ref():
    return os.path.exists(REF_SO)


def round_up(n, d):
    return ((n + d - 1) // d) * d


def duplicate(c):
    """Filter.hs:146-148"""
    return np.repeat(_f32(c), 2)


class Oracle:
    """Restatement.  Lane counts: real L in {1,4,8}; complex CL in {1,2,4}."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        L.orc_resample_rr.restype = C.c_int
        L.orc_resample_rc.restype = C.c_int
        L.orc_resample_cross_r.restype = C.c_int
        L.orc_resample_cross_c.restype = C.c_int
        L.orc_libm_atanf.restype = C.c_float
        L.orc_libm_atanf.argtypes = [C.c_float]
        L.orc_atanf_model.restype = C.c_float
        L.orc_atanf_model.argtypes = [C.c_float]
        L.orc_ghc_atan2f.restype = C.c_float
        L.orc_ghc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_atanf_sweep.restype = C.c_uint64
        L.orc_atanf_sweep.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_libc_version.restype = C.c_char_p
        L.orc_libc_version.argtypes = []
        L.orc_demod_sweep_fill.restype = None
        L.orc_demod_sweep_fill.argtypes = [C.c_int, C.c_uint64, C.c_int64, _f32p]
        L.orc_demod_sweep_check.restype = C.c_uint64
        L.orc_demod_sweep_check.argtypes = [C.c_int, C.c_uint64, C.c_int64, C.c_int, _f32p, C.POINTER(C.c_uint64)]
        L.orc_fm_demod.argtypes = [C.c_int, C.c_float, C.c_float, _f32p, _f32p]
        L.orc_scale.argtypes = [C.c_int, C.c_float, _f32p, _f32p]

    # A1
    def convert_u8(self, u8):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        out = np.empty(u8.size, np.float32)
        self.lib.orc_convert_u8(C.c_int(u8.size), u8.ctypes.data_as(_u8p), _fp(out))
        return out

    def convert_i16(self, i16):
        i16 = np.ascontiguousarray(i16, dtype=np.int16)
        out = np.empty(i16.size, np.float32)
        self.lib.orc_convert_i16(C.c_int(i16.size), i16.ctypes.data_as(C.POINTER(C.c_int16)), _fp(out))
        return out

    def scale(self, factor, x):
        x = _f32(x)
        out = np.empty_like(x)
        self.lib.orc_scale(x.size, C.c_float(factor), _fp(x), _fp(out))
        return out

    # real FIR family ------------------------------------------------------
    def filter_rr(self, L, num, coeffs, x):
        coeffs, x = _f32(coeffs), _f32(x)
        out = np.empty(num, np.float32)
        self.lib.orc_filter_rr(L, num, coeffs.size, _fp(coeffs), _fp(x), _fp(out))
        return out

    def filter_sym_rr(self, L, num, half, x):
        half, x = _f32(half), _f32(x)
        out = np.empty(num, np.float32)
        self.lib.orc_filter_sym_rr(L, num, half.size, _fp(half), _fp(x), _fp(out))
        return out

    def decimate_rr(self, L, num, factor, coeffs, x):
        coeffs, x = _f32(coeffs), _f32(x)
        out = np.empty(num, np.float32)
        self.lib.orc_decimate_rr(L, num, factor, coeffs.size, _fp(coeffs), _fp(x), _fp(out))
        return out

    def decimate_sym_rr(self, L, num, factor, half, x):
        half, x = _f32(half), _f32(x)
        out = np.empty(num, np.float32)
        self.lib.orc_decimate_sym_rr(L, num, factor, half.size, _fp(half), _fp(x), _fp(out))
        return out

    # complex data, real taps ----------------------------------------------
    def decimate_rc(self, CL, num, factor, coeffs_as_passed, x_iq):
        """coeffs_as_passed: plain taps for CL=1, DUPLICATED taps for CL=2/4.
        x_iq: interleaved float32 (2 per sample).  Returns interleaved."""
        c, x = _f32(coeffs_as_passed), _f32(x_iq)
        out = np.empty(2 * num, np.float32)
        self.lib.orc_decimate_rc(CL, num, factor, c.size, _fp(c), _fp(x), _fp(out))
        return out

    def filter_rc(self, CL, num, coeffs_as_passed, x_iq):
        return self.decimate_rc(CL, num, 1, coeffs_as_passed, x_iq)

    def decimate_rc2(self, CL, num, factor, coeffs, x_iq):
        c, x = _f32(coeffs), _f32(x_iq)
        out = np.empty(2 * num, np.float32)
        self.lib.orc_decimate_rc2(CL, num, factor, c.size, _fp(c), _fp(x), _fp(out))
        return out

    def decimate_sym_rc(self, CL, num, factor, half, x_iq):
        c, x = _f32(half), _f32(x_iq)
        out = np.empty(2 * num, np.float32)
        self.lib.orc_decimate_sym_rc(CL, num, factor, c.size, _fp(c), _fp(x), _fp(out))
        return out

    # resampler --------------------------------------------------------------
    def prepare_coeffs(self, n, interp, decim, coeffs):
        """FilterInternal.hs:297-319 -> dict(num_coeffs, num_groups, padded_len,
        increments, offsets, groups[num_groups, padded_len])."""
        coeffs = _f32(coeffs)
        cap = round_up((coeffs.size + interp - 1) // interp + 1, n) + n
        groups = np.zeros((interp, cap), np.float32)
        inc = np.zeros(max(interp, 1), np.int32)
        off = np.zeros(max(interp, 1), np.int32)
        nc, ng, pl = C.c_int(), C.c_int(), C.c_int()
        self.lib.orc_prepare_coeffs(n, interp, decim, _fp(coeffs), coeffs.size,
                                    C.byref(nc), C.byref(ng), C.byref(pl),
                                    inc.ctypes.data_as(_i32p), off.ctypes.data_as(_i32p), _fp(groups))
        flat = groups.reshape(-1)[: ng.value * pl.value].reshape(ng.value, pl.value).copy()
        return dict(num_coeffs=nc.value, num_groups=ng.value, padded_len=pl.value,
                    increments=inc[: ng.value].copy(), offsets=off[: ng.value].copy(), groups=flat)

    @staticmethod
    def _group_ptrs(groups):
        rows = [np.ascontiguousarray(g, dtype=np.float32) for g in groups]
        arr = (_f32p * len(rows))(*[_fp(r) for r in rows])
        return rows, arr

    def resample_rr(self, L, buf_size, prep, starting_group, x):
        x = _f32(x)
        out = np.empty(buf_size, np.float32)
        rows, arr = self._group_ptrs(prep["groups"])
        inc = np.ascontiguousarray(prep["increments"], np.int32)
        g = self.lib.orc_resample_rr(L, buf_size, prep["num_coeffs"], starting_group, prep["num_groups"],
                                     inc.ctypes.data_as(_i32p), arr, _fp(x), _fp(out))
        return out, g

    def resample_rc(self, CL, buf_size, prep, starting_group, x_iq):
        x = _f32(x_iq)
        out = np.empty(2 * buf_size, np.float32)
        rows, arr = self._group_ptrs(prep["groups"])
        inc = np.ascontiguousarray(prep["increments"], np.int32)
        g = self.lib.orc_resample_rc(CL, buf_size, prep["num_coeffs"], starting_group, prep["num_groups"],
                                     inc.ctypes.data_as(_i32p), arr, _fp(x), _fp(out))
        return out, g

    def resample_legacy_rr(self, buf_size, interp, decim, filter_offset, coeffs, x):
        coeffs, x = _f32(coeffs), _f32(x)
        out = np.empty(buf_size, np.float32)
        self.lib.orc_resample_legacy_rr(buf_size, coeffs.size, interp, decim, filter_offset,
                                        _fp(coeffs), _fp(x), _fp(out))
        return out

    # cross-buffer (Haskell) kernels ------------------------------------------
    def decimate_cross_r(self, factor, coeffs, num, last, nxt):
        coeffs, last, nxt = _f32(coeffs), _f32(last), _f32(nxt)
        out = np.empty(num, np.float32)
        self.lib.orc_decimate_cross_r(factor, coeffs.size, _fp(coeffs), num, _fp(last), last.size, _fp(nxt), _fp(out))
        return out

    def decimate_cross_c(self, factor, coeffs, num, last_iq, nxt_iq):
        coeffs, last, nxt = _f32(coeffs), _f32(last_iq), _f32(nxt_iq)
        out = np.empty(2 * num, np.float32)
        self.lib.orc_decimate_cross_c(factor, coeffs.size, _fp(coeffs), num, _fp(last), last.size // 2, _fp(nxt), _fp(out))
        return out

    def resample_cross_r(self, interp, decim, coeffs, filter_offset, count, last, nxt):
        coeffs, last, nxt = _f32(coeffs), _f32(last), _f32(nxt)
        out = np.empty(count, np.float32)
        off = self.lib.orc_resample_cross_r(interp, decim, coeffs.size, _fp(coeffs), filter_offset, count,
                                            _fp(last), last.size, _fp(nxt), _fp(out))
        return out, off

    def resample_cross_c(self, interp, decim, coeffs, filter_offset, count, last_iq, nxt_iq):
        coeffs, last, nxt = _f32(coeffs), _f32(last_iq), _f32(nxt_iq)
        out = np.empty(2 * count, np.float32)
        off = self.lib.orc_resample_cross_c(interp, decim, coeffs.size, _fp(coeffs), filter_offset, count,
                                            _fp(last), last.size // 2, _fp(nxt), _fp(out))
        return out, off

    # A5
    def fm_demod(self, x_iq, last=(0.0, 0.0)):
        x = _f32(x_iq)
        n = x.size // 2
        out = np.empty(n, np.float32)
        self.lib.orc_fm_demod(n, C.c_float(last[0]), C.c_float(last[1]), _fp(x), _fp(out))
        return out

    # N2: dcBlocker, filter.c:152-161
    def dc_blocker(self, x, last_sample=0.0, last_output=0.0):
        x = _f32(x)
        out = np.empty(x.size, np.float32)
        fs, fo = C.c_float(), C.c_float()
        self.lib.orc_dc_blocker.argtypes = [C.c_int64, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p]
        self.lib.orc_dc_blocker(x.size, last_sample, last_output, C.byref(fs), C.byref(fo), _fp(x), _fp(out))
        return out, fs.value, fo.value

    def atanf_sweep(self, lo, hi, step=1):
        """The fdlibm f32 model (the SPEC of fmDemod's atan) against this machine's libm atanf over bit patterns lo..hi:
        (arguments that differ, the first of them, the largest difference in ULP)."""
        bad, worst = C.c_uint32(0), C.c_uint32(0)
        n = self.lib.orc_atanf_sweep(lo, hi, step, C.byref(bad), C.byref(worst))
        return int(n), int(bad.value), int(worst.value)

    def libc_version(self):
        return self.lib.orc_libc_version().decode()

    def demod_sweep_fill(self, kind, idx0, n):
        """The synthetic stream of the exhaustive fmDemod checks (sdr_oracle.c: orc_demod_sweep_*): 2n complex samples."""
        iq = np.empty(4 * n, np.float32)
        self.lib.orc_demod_sweep_fill(kind, idx0, n, _fp(iq))
        return iq

    def demod_sweep_check(self, kind, idx0, n, at_stream_start, got):
        got = _f32(got)
        assert got.size == 2 * n
        first = C.c_uint64(0)
        bad = self.lib.orc_demod_sweep_check(kind, idx0, n, 1 if at_stream_start else 0, _fp(got), C.byref(first))
        return int(bad), int(first.value)


class Ref:
    """The reference's own compiled C (oracle/_ref).  Symbol names and
    signatures are the reference's."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        for n in ("resample2RR", "resampleSSERR", "resampleAVXRR", "resample2RC", "resampleSSERC", "resampleAVXRC"):
            getattr(self.lib, n).restype = C.c_int
        self.lib.scale.argtypes = [C.c_int, C.c_float, _f32p, _f32p]
        self.lib.scaleSSE.argtypes = [C.c_int, C.c_float, _f32p, _f32p]
        self.lib.scaleAVX.argtypes = [C.c_int, C.c_float, _f32p, _f32p]

    def dc_blocker(self, x, last_sample=0.0, last_output=0.0):
        x = _f32(x)
        out = np.empty(x.size, np.float32)
        fs, fo = C.c_float(), C.c_float()
        self.lib.dcBlocker.argtypes = [C.c_int, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p]
        self.lib.dcBlocker(x.size, last_sample, last_output, C.byref(fs), C.byref(fo), _fp(x), _fp(out))
        return out, fs.value, fo.value

    def convert(self, sym, u8, pad=16):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        # the SSE/AVX variants over-read up to 12 bytes past the input tail
        # (convert.c:27,42): give them slack.
        buf = np.zeros(u8.size + pad, np.uint8)
        buf[: u8.size] = u8
        out = np.empty(round_up(u8.size, 8), np.float32)
        getattr(self.lib, sym)(C.c_int(u8.size), buf.ctypes.data_as(_u8p), _fp(out))
        return out[: u8.size].copy()

    def filt(self, sym, num, coeffs_as_passed, x, complex_=False):
        c, x = _f32(coeffs_as_passed), _f32(x)
        out = np.empty(num * (2 if complex_ else 1), np.float32)
        getattr(self.lib, sym)(C.c_int(num), C.c_int(c.size), _fp(c), _fp(x), _fp(out))
        return out

    def decim(self, sym, num, factor, coeffs_as_passed, x, complex_=False):
        c, x = _f32(coeffs_as_passed), _f32(x)
        out = np.empty(num * (2 if complex_ else 1), np.float32)
        getattr(self.lib, sym)(C.c_int(num), C.c_int(factor), C.c_int(c.size), _fp(c), _fp(x), _fp(out))
        return out

    def resample(self, sym, buf_size, prep, starting_group, x, complex_=False):
        x = _f32(x)
        out = np.empty(buf_size * (2 if complex_ else 1), np.float32)
        rows, arr = Oracle._group_ptrs(prep["groups"])
        inc = np.ascontiguousarray(prep["increments"], np.int32)
        g = getattr(self.lib, sym)(C.c_int(buf_size), C.c_int(prep["num_coeffs"]), C.c_int(starting_group),
                                   C.c_int(prep["num_groups"]), inc.ctypes.data_as(_i32p), arr, _fp(x), _fp(out))
        return out, g

    def resample_legacy(self, buf_size, interp, decim, filter_offset, coeffs, x):
        coeffs, x = _f32(coeffs), _f32(x)
        out = np.empty(buf_size, np.float32)
        self.lib.resampleRR(C.c_int(buf_size), C.c_int(coeffs.size), C.c_int(interp), C.c_int(decim),
                            C.c_int(filter_offset), _fp(coeffs), _fp(x), _fp(out))
        return out

    def scale(self, sym, factor, x):
        x = _f32(x)
        out = np.empty_like(x)
        getattr(self.lib, sym)(x.size, C.c_float(factor), _fp(x), _fp(out))
        return out
