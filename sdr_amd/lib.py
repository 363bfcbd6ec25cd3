"""ctypes binding of libsdr_hip.so -- the ONLY way Python reaches the product.

There is no CPU fallback: if the HIP library is missing or fails to load, import
of this module raises.  (The CPU oracle lives under oracle/ and is never imported
from here.)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libsdr_hip.so")

ORDER_SCALAR, ORDER_SSE, ORDER_AVX = 0, 1, 2

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int)
_vp = C.c_void_p
_i64 = C.c_int64


class SdrHipError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own copy of libamdhip64 / libhsa-runtime64 and looks it up by
    file name, libsdr_hip.so asks for the system copy by SONAME; whichever loads second then brings a second runtime into
    the process and loses the GPU ("no ROCm-capable device").  If torch is installed, map ITS copies first (by path, without
    importing torch): our library's SONAME lookup then resolves to them, and a later `import torch` finds the very same
    files already mapped.  Without torch (a plain C host program, examples/fm_replay.c) the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def _load():
    if not os.path.exists(LIB_PATH):
        raise SdrHipError(
            f"{LIB_PATH} not found: build it with `python -m sdr_amd.build` "
            "(there is deliberately no CPU fallback)")
    _share_torch_hip_runtime()
    return C.CDLL(LIB_PATH)


lib = _load()

# ---- signatures -----------------------------------------------------------------
lib.sdrhip_version.restype = C.c_char_p
lib.sdrhip_set_small_launch_outputs.argtypes = [C.c_int]
lib.sdrhip_set_small_launch_outputs.restype = C.c_int
lib.sdrhip_last_error.restype = C.c_char_p
lib.sdrhip_device_name.argtypes = [C.c_char_p, C.c_int]
lib.sdrhip_malloc.argtypes = [C.POINTER(_vp), C.c_size_t]
lib.sdrhip_free.argtypes = [_vp]
lib.sdrhip_malloc_host.argtypes = [C.POINTER(_vp), C.c_size_t]
lib.sdrhip_free_host.argtypes = [_vp]
for _n in ("sdrhip_memcpy_h2d", "sdrhip_memcpy_d2h", "sdrhip_memcpy_d2d"):
    getattr(lib, _n).argtypes = [_vp, _vp, C.c_size_t, _vp]
lib.sdrhip_stream_create.argtypes = [C.POINTER(_vp)]
lib.sdrhip_stream_destroy.argtypes = [_vp]
lib.sdrhip_stream_sync.argtypes = [_vp]

lib.sdrhip_filter_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, _f32p, C.c_int]
lib.sdrhip_filter_sym_create.argtypes = [C.POINTER(_vp), C.c_int, _f32p, C.c_int]
lib.sdrhip_filter_num_coeffs.argtypes = [_vp]
lib.sdrhip_filter_destroy.argtypes = [_vp]
lib.sdrhip_filter_destroy.restype = None
lib.sdrhip_filter_run.argtypes = [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64]

lib.sdrhip_decimator_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _f32p, C.c_int]
lib.sdrhip_decimator_sym_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, _f32p, C.c_int]
lib.sdrhip_decimator_num_coeffs.argtypes = [_vp]
lib.sdrhip_decimator_factor.argtypes = [_vp]
lib.sdrhip_decimator_destroy.argtypes = [_vp]
lib.sdrhip_decimator_destroy.restype = None
lib.sdrhip_decimator_run.argtypes = [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64]
lib.sdrhip_decimator_run_u8.argtypes = [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64]

lib.sdrhip_resampler_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]
lib.sdrhip_resampler_num_coeffs.argtypes = [_vp]
lib.sdrhip_resampler_num_groups.argtypes = [_vp]
lib.sdrhip_resampler_destroy.argtypes = [_vp]
lib.sdrhip_resampler_destroy.restype = None
lib.sdrhip_resampler_in_offset.argtypes = [_vp, _i64]
lib.sdrhip_resampler_in_offset.restype = _i64
lib.sdrhip_resampler_filter_offset.argtypes = [_vp, _i64]
lib.sdrhip_resampler_group.argtypes = [_vp, _i64]
lib.sdrhip_resampler_run.argtypes = [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i64]

lib.sdrhip_convert_u8_run.argtypes = [_vp, _vp, _vp, _i64]
lib.sdrhip_convert_i16_run.argtypes = [_vp, _vp, _vp, _i64]
lib.sdrhip_scale_run.argtypes = [_vp, C.c_float, _vp, _vp, _i64]
lib.sdrhip_fm_demod_run.argtypes = [_vp, _vp, _i64, _vp, _i64, _i64, C.c_float, C.c_float]

lib.sdrhip_fm_chain_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, _f32p,
                                       C.c_int, _f32p, C.c_int, C.c_float, _i64]
lib.sdrhip_fm_chain_destroy.argtypes = [_vp]
lib.sdrhip_fm_chain_destroy.restype = None
lib.sdrhip_fm_chain_plan.argtypes = [_vp, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]
lib.sdrhip_fm_chain_ready.argtypes = [_vp, _i64]
lib.sdrhip_fm_chain_ready.restype = _i64
lib.sdrhip_fm_chain_max_halo.argtypes = [_vp]
lib.sdrhip_fm_chain_max_halo.restype = _i64
lib.sdrhip_fm_chain_workspace_bytes.argtypes = [_vp, _i64]
lib.sdrhip_fm_chain_workspace_bytes.restype = C.c_size_t
lib.sdrhip_fm_chain_run.argtypes = [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, C.c_size_t]

lib.sdrhip_fm_chain_graph_create.argtypes = [C.POINTER(_vp), _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, C.c_size_t]
lib.sdrhip_fm_chain_graph_launch.argtypes = [_vp, _vp]
lib.sdrhip_fm_chain_graph_destroy.argtypes = [_vp]
lib.sdrhip_fm_chain_graph_destroy.restype = None
lib.sdrhip_fm_chain_set_overlap.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_chain_join.argtypes = [_vp, _vp]
lib.sdrhip_fm_chain_set_fused_tail.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_chain_set_small_chain.argtypes = [_vp, C.c_int, _i64, C.c_int]
lib.sdrhip_debug_small_chain_launches.restype = C.c_longlong
lib.sdrhip_debug_resample_cycle_launches.restype = C.c_longlong
lib.sdrhip_debug_decimate_real16_launches.restype = C.c_longlong
lib.sdrhip_debug_systolic_launches.restype = C.c_longlong
lib.sdrhip_debug_set_systolic.argtypes = [C.c_int]
lib.sdrhip_debug_set_systolic.restype = None
lib.sdrhip_debug_systolic_plan.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.sdrhip_debug_systolic_plan.restype = None
lib.sdrhip_fm_chain_enable_timing.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_chain_read_timing.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]

lib.sdrhip_fm_chain_halo_samples.argtypes = [_vp]
lib.sdrhip_fm_chain_halo_samples.restype = _i64
lib.sdrhip_fm_chain_halo_exchange.argtypes = [_vp, _vp, _vp, _vp, _i64]
lib.sdrhip_fm_chain_halo_staging_bytes.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_chain_halo_staging_bytes.restype = C.c_size_t
lib.sdrhip_fm_chain_halo_exchange_batch.argtypes = [_vp, _vp, _vp, _vp, _i64, C.c_size_t, C.c_int, _vp]
lib.sdrhip_comm_get_unique_id.argtypes = [C.c_char_p]
lib.sdrhip_comm_init_rank.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_char_p]
lib.sdrhip_comm_init_local.argtypes = [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int), C.c_int]
lib.sdrhip_comm_destroy.argtypes = [_vp]
lib.sdrhip_comm_destroy.restype = None
lib.sdrhip_comm_rank.argtypes = [_vp]
lib.sdrhip_comm_size.argtypes = [_vp]
lib.sdrhip_comm_transport.argtypes = [_vp]
lib.sdrhip_comm_transport.restype = C.c_char_p
lib.sdrhip_halo_exchange.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t]
lib.sdrhip_halo_exchange_all.argtypes = [C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_size_t]
_f64p = C.POINTER(C.c_double)
lib.sdrhip_fft_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]
lib.sdrhip_fft_destroy.argtypes = [_vp]
lib.sdrhip_fft_destroy.restype = None
lib.sdrhip_fft_size.argtypes = [_vp]
lib.sdrhip_fft_bins.argtypes = [_vp]
lib.sdrhip_fft_run.argtypes = [_vp, _f64p, _f64p]
lib.sdrhip_fft_run_device.argtypes = [_vp, _vp, _vp, _vp]
lib.sdrhip_filter_one.argtypes = [_vp, C.c_int, _f32p, _f32p]
lib.sdrhip_filter_cross.argtypes = [_vp, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p]
lib.sdrhip_decimator_one.argtypes = [_vp, C.c_int, _f32p, _f32p]
lib.sdrhip_decimator_cross.argtypes = [_vp, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p]
lib.sdrhip_resampler_one.argtypes = [_vp, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
lib.sdrhip_resampler_cross.argtypes = [_vp, C.c_int, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p]
# ---- measurement utilities (include/sdr_hip_bench.h): libsdr_hip_bench.so, a library of its own beside the product (round 6); bench.py and
# tools/ reach them as lib.sdrhip_bench_* like everything else, so the names are attached to `lib` here
BENCH_LIB_PATH = os.path.join(HERE, "lib", "libsdr_hip_bench.so")
if not os.path.exists(BENCH_LIB_PATH):
    raise SdrHipError(f"{BENCH_LIB_PATH} not found: build it with `python -m sdr_amd.build`")
benchlib = C.CDLL(BENCH_LIB_PATH)
for _n in ("sdrhip_bench_stream_8to1", "sdrhip_bench_copy", "sdrhip_bench_copy2", "sdrhip_bench_fm_stream", "sdrhip_bench_fm_stream_latency",
           "sdrhip_bench_pipe", "sdrhip_bench_fm_pipes"):
    setattr(lib, _n, getattr(benchlib, _n))
lib.sdrhip_bench_stream_8to1.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int]
lib.sdrhip_bench_copy.argtypes = [_vp, _vp, _vp, C.c_size_t]
lib.sdrhip_bench_copy2.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int]
lib.sdrhip_bench_fm_stream.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
lib.sdrhip_bench_fm_stream_latency.argtypes = [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double)]
lib.sdrhip_bench_pipe.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
lib.sdrhip_bench_fm_pipes.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
lib.sdrhip_debug_tiled_launches.argtypes = []
lib.sdrhip_debug_tiled_launches.restype = C.c_longlong
lib.sdrhip_dc_blocker_workspace_bytes.argtypes = [C.c_int64]
lib.sdrhip_dc_blocker_workspace_bytes.restype = C.c_size_t
lib.sdrhip_dc_blocker_run.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_float, C.c_float, _vp, _vp, C.c_size_t, C.c_int]

lib.sdrhip_fm_stream_create.argtypes = [C.POINTER(_vp), _vp, C.c_int, C.c_int]
lib.sdrhip_fm_stream_destroy.argtypes = [_vp]
lib.sdrhip_fm_stream_destroy.restype = None
lib.sdrhip_fm_stream_push.argtypes = [_vp, _u8p, C.c_int]
lib.sdrhip_fm_stream_flush.argtypes = [_vp]
lib.sdrhip_fm_stream_poll.argtypes = [_vp]
lib.sdrhip_fm_stream_set_coalesce.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_stream_set_adaptive.argtypes = [_vp, C.c_int]
lib.sdrhip_fm_stream_input_buffer.argtypes = [_vp]
lib.sdrhip_fm_stream_input_buffer.restype = _vp
lib.sdrhip_fm_stream_pop.argtypes = [_vp, _f32p, C.c_int]

lib.sdrhip_pipe_fir_filter.argtypes = [C.POINTER(_vp), _vp, C.c_int]
lib.sdrhip_pipe_fir_decimator.argtypes = [C.POINTER(_vp), _vp, C.c_int]
lib.sdrhip_pipe_fir_resampler.argtypes = [C.POINTER(_vp), _vp, C.c_int]
lib.sdrhip_pipe_fm_demod.argtypes = [C.POINTER(_vp)]
lib.sdrhip_pipe_dc_blocker.argtypes = [C.POINTER(_vp)]
lib.sdrhip_pipe_set_coalesce.argtypes = [_vp, C.c_int]
lib.sdrhip_pipe_set_adaptive.argtypes = [_vp, C.c_int]
lib.sdrhip_pipe_input_buffer.argtypes = [_vp, C.c_int]
lib.sdrhip_pipe_input_buffer.restype = _vp
lib.sdrhip_pipe_push.argtypes = [_vp, _f32p, C.c_int]
lib.sdrhip_pipe_flush.argtypes = [_vp]
lib.sdrhip_pipe_poll.argtypes = [_vp]
lib.sdrhip_pipe_pop.argtypes = [_vp, _f32p, C.c_int]
lib.sdrhip_pipe_destroy.argtypes = [_vp]
lib.sdrhip_pipe_destroy.restype = None

lib.scale.argtypes = [C.c_int, C.c_float, _f32p, _f32p]
lib.scaleSSE.argtypes = [C.c_int, C.c_float, _f32p, _f32p]
lib.scaleAVX.argtypes = [C.c_int, C.c_float, _f32p, _f32p]
lib.fmDemodF.argtypes = [C.c_int, C.c_float, C.c_float, _f32p, _f32p]
lib.dcBlocker.argtypes = [C.c_int, C.c_float, C.c_float, _f32p, _f32p, _f32p, _f32p]
for _n in ("resample2RR", "resampleSSERR", "resampleAVXRR", "resample2RC", "resampleSSERC", "resampleAVXRC"):
    getattr(lib, _n).restype = C.c_int


def check(rc, what="sdrhip call"):
    if rc < 0:
        raise SdrHipError(f"{what} failed ({rc}): {lib.sdrhip_last_error().decode()}")
    return rc


def set_small_launch_outputs(outputs):
    """sdrhip_set_small_launch_outputs: returns the previous threshold."""
    return lib.sdrhip_set_small_launch_outputs(int(outputs))


def version():
    return lib.sdrhip_version().decode()


def device_count():
    return lib.sdrhip_device_count()


def device_name():
    buf = C.create_string_buffer(256)
    check(lib.sdrhip_device_name(buf, 256), "sdrhip_device_name")
    return buf.value.decode()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_f32p)


# ---- drop-in symbols on numpy host arrays (what the Haskell FFI would call) ----------
class DropIn:
    """Calls the reference-named symbols exactly as FilterInternal.hs's wrappers do."""

    @staticmethod
    def convert(sym, u8):
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        out = np.empty(u8.size, np.float32)
        getattr(lib, sym)(C.c_int(u8.size), u8.ctypes.data_as(_u8p), _fp(out))
        return out

    @staticmethod
    def convert_i16(sym, i16):
        i16 = np.ascontiguousarray(i16, dtype=np.int16)
        out = np.empty(i16.size, np.float32)
        getattr(lib, sym)(C.c_int(i16.size), i16.ctypes.data_as(_i16p), _fp(out))
        return out

    @staticmethod
    def convert_tx(x):
        x = _f32(x)
        out = np.empty(x.size, np.int16)
        lib.convertBladeRFTransmit(C.c_int(x.size), _fp(x), out.ctypes.data_as(_i16p))
        return out

    @staticmethod
    def scale(sym, factor, x):
        x = _f32(x)
        out = np.empty_like(x)
        getattr(lib, sym)(x.size, C.c_float(factor), _fp(x), _fp(out))
        return out

    @staticmethod
    def filt(sym, num, coeffs_as_passed, x, complex_=False):
        c, x = _f32(coeffs_as_passed), _f32(x)
        out = np.empty(num * (2 if complex_ else 1), np.float32)
        getattr(lib, sym)(C.c_int(num), C.c_int(c.size), _fp(c), _fp(x), _fp(out))
        return out

    @staticmethod
    def decim(sym, num, factor, coeffs_as_passed, x, complex_=False):
        c, x = _f32(coeffs_as_passed), _f32(x)
        out = np.empty(num * (2 if complex_ else 1), np.float32)
        getattr(lib, sym)(C.c_int(num), C.c_int(factor), C.c_int(c.size), _fp(c), _fp(x), _fp(out))
        return out

    @staticmethod
    def resample(sym, buf_size, num_coeffs, starting_group, increments, groups, x, complex_=False):
        x = _f32(x)
        out = np.empty(buf_size * (2 if complex_ else 1), np.float32)
        rows = [np.ascontiguousarray(g, dtype=np.float32) for g in groups]
        arr = (_f32p * len(rows))(*[_fp(r) for r in rows])
        inc = np.ascontiguousarray(increments, np.int32)
        g = getattr(lib, sym)(C.c_int(buf_size), C.c_int(num_coeffs), C.c_int(starting_group), C.c_int(len(rows)),
                              inc.ctypes.data_as(_i32p), arr, _fp(x), _fp(out))
        return out, g

    @staticmethod
    def resample_legacy(buf_size, interp, decim, filter_offset, coeffs, x):
        coeffs, x = _f32(coeffs), _f32(x)
        out = np.empty(buf_size, np.float32)
        lib.resampleRR(C.c_int(buf_size), C.c_int(coeffs.size), C.c_int(interp), C.c_int(decim),
                       C.c_int(filter_offset), _fp(coeffs), _fp(x), _fp(out))
        return out

    @staticmethod
    def fm_demod(x_iq, last=(0.0, 0.0)):
        x = _f32(x_iq)
        n = x.size // 2
        out = np.empty(n, np.float32)
        lib.fmDemodF(n, C.c_float(last[0]), C.c_float(last[1]), _fp(x), _fp(out))
        return out

    @staticmethod
    def dc_blocker(x, last_sample=0.0, last_output=0.0):
        x = _f32(x)
        out = np.empty_like(x)
        fs, fo = C.c_float(), C.c_float()
        lib.dcBlocker(x.size, C.c_float(last_sample), C.c_float(last_output), C.byref(fs), C.byref(fo), _fp(x), _fp(out))
        return out, fs.value, fo.value


# ---- descriptors ----------------------------------------------------------------------
class _Handle:
    _destroy = None

    def __init__(self):
        self.h = _vp()

    def close(self):
        if self.h:
            type(self)._destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Filter(_Handle):
    """fastFilter{C,SSE,AVX}{R,C} / fastFilterSym{SSE,AVX}R (Filter.hs:163-261)."""
    _destroy = lib.sdrhip_filter_destroy

    def __init__(self, coeffs, order=ORDER_AVX, complex_=False, sym=False):
        super().__init__()
        c = _f32(coeffs)
        if sym:
            check(lib.sdrhip_filter_sym_create(C.byref(self.h), order, _fp(c), c.size), "sdrhip_filter_sym_create")
        else:
            check(lib.sdrhip_filter_create(C.byref(self.h), order, int(complex_), _fp(c), c.size), "sdrhip_filter_create")
        self.complex = complex_
        self.num_coeffs = lib.sdrhip_filter_num_coeffs(self.h)

    def run(self, d_in, in_base, d_out, k_begin, k_end, seam_block=0, stream=None):
        check(lib.sdrhip_filter_run(self.h, stream, d_in, in_base, d_out, k_begin, k_end, seam_block), "sdrhip_filter_run")


class Decimator(_Handle):
    """fastDecimator{C,SSE,AVX}{R,C} / fastDecimatorSym{SSE,AVX}R (Filter.hs:277-387)."""
    _destroy = lib.sdrhip_decimator_destroy

    def __init__(self, factor, coeffs, order=ORDER_AVX, complex_=False, sym=False):
        super().__init__()
        c = _f32(coeffs)
        if sym:
            check(lib.sdrhip_decimator_sym_create(C.byref(self.h), order, factor, _fp(c), c.size), "sdrhip_decimator_sym_create")
        else:
            check(lib.sdrhip_decimator_create(C.byref(self.h), order, int(complex_), factor, _fp(c), c.size), "sdrhip_decimator_create")
        self.complex = complex_
        self.factor = factor
        self.num_coeffs = lib.sdrhip_decimator_num_coeffs(self.h)

    def run(self, d_in, in_base, d_out, k_begin, k_end, seam_block=0, stream=None):
        check(lib.sdrhip_decimator_run(self.h, stream, d_in, in_base, d_out, k_begin, k_end, seam_block), "sdrhip_decimator_run")

    def run_u8(self, d_in, in_base, d_out, k_begin, k_end, seam_block=0, stream=None):
        check(lib.sdrhip_decimator_run_u8(self.h, stream, d_in, in_base, d_out, k_begin, k_end, seam_block), "sdrhip_decimator_run_u8")


class Resampler(_Handle):
    """fastResampler{C,SSE,AVX}{R,C} (Filter.hs:408-502)."""
    _destroy = lib.sdrhip_resampler_destroy

    def __init__(self, interpolation, decimation, coeffs, order=ORDER_AVX, complex_=False):
        super().__init__()
        c = _f32(coeffs)
        check(lib.sdrhip_resampler_create(C.byref(self.h), order, int(complex_), interpolation, decimation, _fp(c), c.size),
              "sdrhip_resampler_create")
        self.complex = complex_
        self.I, self.D = interpolation, decimation
        self.num_coeffs = lib.sdrhip_resampler_num_coeffs(self.h)
        self.num_groups = lib.sdrhip_resampler_num_groups(self.h)

    def in_offset(self, m):
        return lib.sdrhip_resampler_in_offset(self.h, m)

    def filter_offset(self, m):
        return lib.sdrhip_resampler_filter_offset(self.h, m)

    def group(self, m):
        return lib.sdrhip_resampler_group(self.h, m)

    def run(self, d_in, in_base, d_out, k_begin, k_end, seam_block=0, stream=None, out_block=0):
        check(lib.sdrhip_resampler_run(self.h, stream, d_in, in_base, d_out, k_begin, k_end, seam_block, out_block),
              "sdrhip_resampler_run")


class FmChain(_Handle):
    """The FM receiver of examples/fm/fm.hs:34-41 as one device-resident object."""
    _destroy = lib.sdrhip_fm_chain_destroy

    def __init__(self, decim_factor, decim_taps, interpolation, decimation, resamp_taps, audio_half_taps,
                 gain=1.0, block=8192, order=ORDER_AVX):
        super().__init__()
        a, b, c = _f32(decim_taps), _f32(resamp_taps), _f32(audio_half_taps)
        check(lib.sdrhip_fm_chain_create(C.byref(self.h), order, decim_factor, _fp(a), a.size, interpolation,
                                         decimation, _fp(b), b.size, _fp(c), c.size, C.c_float(gain), block),
              "sdrhip_fm_chain_create")

    def plan(self, s0, s1, total_in=-1):
        q0, q1, halo = _i64(), _i64(), _i64()
        check(lib.sdrhip_fm_chain_plan(self.h, s0, s1, total_in, C.byref(q0), C.byref(q1), C.byref(halo)), "sdrhip_fm_chain_plan")
        return q0.value, q1.value, halo.value

    def ready(self, n_samples):
        """Audio outputs computable from the first n_samples samples of the stream."""
        return int(lib.sdrhip_fm_chain_ready(self.h, n_samples))

    def max_halo(self):
        return lib.sdrhip_fm_chain_max_halo(self.h)

    def halo_staging_bytes(self, count):
        return int(lib.sdrhip_fm_chain_halo_staging_bytes(self.h, count))

    def halo_samples(self):
        """max_halo rounded up to whole 16-byte vectors: the size of the halo message, the same on every rank."""
        return int(lib.sdrhip_fm_chain_halo_samples(self.h))

    def workspace_bytes(self, n_in):
        return lib.sdrhip_fm_chain_workspace_bytes(self.h, n_in)

    STAGES = ("decimate", "fm_demod", "resample", "filter", "fused_tail", "fused_chain")

    def set_overlap(self, on):
        """Two runs in flight: consecutive runs alternate between two internal streams and workspace halves (sdr_hip.h)."""
        check(lib.sdrhip_fm_chain_set_overlap(self.h, 1 if on else 0), "sdrhip_fm_chain_set_overlap")

    def join(self, stream=0):
        check(lib.sdrhip_fm_chain_join(self.h, stream), "sdrhip_fm_chain_join")

    def set_fused_tail(self, mode=2):
        """0 = stage kernels, 1 = the fused tail kernel wherever the chain's shape allows, 2 = auto (short runs only)."""
        check(lib.sdrhip_fm_chain_set_fused_tail(self.h, int(mode)), "sdrhip_fm_chain_set_fused_tail")

    def set_small_chain(self, mode=2, max_outputs=0, tile_outputs=0):
        """0 = never, 1 = always, 2 = auto: the whole chain as one kernel for launch-bound runs (kernels_small.hip)."""
        check(lib.sdrhip_fm_chain_set_small_chain(self.h, int(mode), int(max_outputs), int(tile_outputs)), "sdrhip_fm_chain_set_small_chain")

    def set_demod_fusion(self, on=True):
        """fmDemod inside the resampler's tile loader on large batches (sdrhip_fm_chain_set_demod_fusion)."""
        check(lib.sdrhip_fm_chain_set_demod_fusion(self.h, int(on)), "sdrhip_fm_chain_set_demod_fusion")

    def enable_timing(self, on=True):
        check(lib.sdrhip_fm_chain_enable_timing(self.h, int(on)), "sdrhip_fm_chain_enable_timing")

    def read_timing(self):
        """-> (dict stage -> mean ms per run, runs)"""
        ms = (C.c_double * 6)()
        runs = C.c_int()
        check(lib.sdrhip_fm_chain_read_timing(self.h, ms, C.byref(runs)), "sdrhip_fm_chain_read_timing")
        n = max(runs.value, 1)
        return {k: ms[i] / n for i, k in enumerate(self.STAGES)}, runs.value

    def run(self, d_in_u8, s0, n_in, d_audio, q0, q1, d_ws, ws_bytes, stream=None):
        check(lib.sdrhip_fm_chain_run(self.h, stream, d_in_u8, s0, n_in, d_audio, q0, q1, d_ws, ws_bytes), "sdrhip_fm_chain_run")


class Fft(_Handle):
    """fftw' / fftwReal' of SDR.FFT (FFT.hs:44-108) on hipFFT: Complex Double in FFTW's order, `batch` transforms per call."""
    _destroy = lib.sdrhip_fft_destroy

    def __init__(self, n, real_input=False, batch=1):
        super().__init__()
        check(lib.sdrhip_fft_create(C.byref(self.h), n, int(real_input), batch), "sdrhip_fft_create")
        self.n, self.real_input, self.batch = n, bool(real_input), batch
        self.bins = lib.sdrhip_fft_bins(self.h)

    def run(self, x):
        import numpy as np
        if self.real_input:
            a = np.ascontiguousarray(x, dtype=np.float64).reshape(self.batch, self.n)
        else:
            a = np.ascontiguousarray(x, dtype=np.complex128).reshape(self.batch, self.n)
        out = np.empty((self.batch, self.bins), np.complex128)
        check(lib.sdrhip_fft_run(self.h, a.ctypes.data_as(_f64p), out.ctypes.data_as(_f64p)), "sdrhip_fft_run")
        return out if self.batch > 1 else out[0]


TRANSPORT_RCCL, TRANSPORT_PEER_COPY = 1, 2
COMM_ID_BYTES = 128


def comm_unique_id():
    """The 128-byte rendezvous id rank 0 creates (ncclGetUniqueId) and hands to the other ranks out of band."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(lib.sdrhip_comm_get_unique_id(buf), "sdrhip_comm_get_unique_id")
    return buf.raw


class Comm(_Handle):
    """One rank's communicator for the halo exchange (RCCL point-to-point inside libsdr_hip.so)."""
    _destroy = lib.sdrhip_comm_destroy

    def __init__(self, nranks=None, rank=None, uid=None, _adopt=None):
        super().__init__()
        if _adopt is not None:
            self.h = _adopt
            return
        check(lib.sdrhip_comm_init_rank(C.byref(self.h), nranks, rank, uid), "sdrhip_comm_init_rank")

    @staticmethod
    def local(devices, transport=TRANSPORT_RCCL):
        """All `devices` from this process: one Comm per device, rank i on devices[i]."""
        n = len(devices)
        hs = (_vp * n)()
        devs = (C.c_int * n)(*devices)
        check(lib.sdrhip_comm_init_local(hs, n, devs, transport), "sdrhip_comm_init_local")
        return [Comm(_adopt=_vp(hs[i])) for i in range(n)]

    @property
    def rank(self):
        return lib.sdrhip_comm_rank(self.h)

    @property
    def size(self):
        return lib.sdrhip_comm_size(self.h)

    @property
    def transport(self):
        return lib.sdrhip_comm_transport(self.h).decode()

    def halo_exchange(self, d_send, d_recv, nbytes, stream=None):
        check(lib.sdrhip_halo_exchange(self.h, stream, d_send, d_recv, nbytes), "sdrhip_halo_exchange")

    def chain_halo_exchange(self, chain, d_buf, shard_samples, stream=None):
        check(lib.sdrhip_fm_chain_halo_exchange(chain.h, self.h, stream, d_buf, shard_samples), "sdrhip_fm_chain_halo_exchange")

    def chain_halo_exchange_batch(self, chain, d_buf, shard_samples, row_bytes, count, d_staging, stream=None):
        """The halos of `count` consecutive super-blocks (rows of row_bytes at d_buf) in one message pair."""
        check(lib.sdrhip_fm_chain_halo_exchange_batch(chain.h, self.h, stream, d_buf, shard_samples, row_bytes, count, d_staging),
              "sdrhip_fm_chain_halo_exchange_batch")


def halo_exchange_all(comms, streams, d_send, d_recv, nbytes):
    n = len(comms)
    hs = (_vp * n)(*[c.h for c in comms])
    ss = (_vp * n)(*[_vp(s) for s in streams])
    sd = (_vp * n)(*[_vp(p) for p in d_send])
    rv = (_vp * n)(*[_vp(p) for p in d_recv])
    check(lib.sdrhip_halo_exchange_all(hs, n, ss, sd, rv, nbytes), "sdrhip_halo_exchange_all")


class FmGraph(_Handle):
    """One FmChain.run with fixed arguments as a hipGraph (sdrhip_fm_chain_graph_*): one launch per pass."""
    _destroy = lib.sdrhip_fm_chain_graph_destroy

    def __init__(self, chain, d_in_u8, s0, n_in, d_audio, q0, q1, d_ws, ws_bytes):
        super().__init__()
        self.chain = chain
        check(lib.sdrhip_fm_chain_graph_create(C.byref(self.h), chain.h, d_in_u8, s0, n_in, d_audio, q0, q1, d_ws, ws_bytes),
              "sdrhip_fm_chain_graph_create")

    def launch(self, stream=None):
        check(lib.sdrhip_fm_chain_graph_launch(self.h, stream), "sdrhip_fm_chain_graph_launch")


class FmStream(_Handle):
    """u8 IQ host blocks in, audio host blocks out: the FM receiver of fm.hs:34-41 as one operator."""
    _destroy = lib.sdrhip_fm_stream_destroy

    def __init__(self, chain, max_block_samples, block_size_out=8192):
        super().__init__()
        self.chain = chain  # keep alive
        self.block_size_out = block_size_out
        check(lib.sdrhip_fm_stream_create(C.byref(self.h), chain.h, max_block_samples, block_size_out), "sdrhip_fm_stream_create")

    def set_coalesce(self, samples):
        check(lib.sdrhip_fm_stream_set_coalesce(self.h, samples), "sdrhip_fm_stream_set_coalesce")

    def set_adaptive(self, max_samples):
        check(lib.sdrhip_fm_stream_set_adaptive(self.h, max_samples), "sdrhip_fm_stream_set_adaptive")

    def _pop(self, ready):
        outs = []
        for _ in range(ready):
            o = np.empty(self.block_size_out, np.float32)
            check(lib.sdrhip_fm_stream_pop(self.h, _fp(o), self.block_size_out), "sdrhip_fm_stream_pop")
            outs.append(o)
        return outs

    def push(self, u8_iq):
        b = np.ascontiguousarray(u8_iq, dtype=np.uint8)
        return self._pop(check(lib.sdrhip_fm_stream_push(self.h, b.ctypes.data_as(_u8p), b.size // 2), "sdrhip_fm_stream_push"))

    def input_buffer(self, n_samples):
        """numpy view of the pinned staging buffer of the next push (fill it, then push_inplace)."""
        p = lib.sdrhip_fm_stream_input_buffer(self.h)
        if not p:
            raise SdrHipError(lib.sdrhip_last_error().decode())
        return np.ctypeslib.as_array(C.cast(p, _u8p), shape=(2 * n_samples,))

    def push_inplace(self, view):
        return self._pop(check(lib.sdrhip_fm_stream_push(self.h, view.ctypes.data_as(_u8p), view.size // 2), "sdrhip_fm_stream_push"))

    def flush(self):
        return self._pop(check(lib.sdrhip_fm_stream_flush(self.h), "sdrhip_fm_stream_flush"))

    def poll(self):
        """blocks the GPU has finished meanwhile (never waits)"""
        return self._pop(check(lib.sdrhip_fm_stream_poll(self.h), "sdrhip_fm_stream_poll"))

    def save(self):
        """sdrhip_fm_stream_save: drains the operator and returns its state as bytes (audio blocks that became ready stay
        inside the state / the stream: pop them from either)."""
        lib.sdrhip_fm_stream_state_bytes.restype = C.c_size_t
        lib.sdrhip_fm_stream_state_bytes.argtypes = [C.c_void_p]
        cap = lib.sdrhip_fm_stream_state_bytes(self.h)                   # includes room for the audio the drain adds
        buf = (C.c_ubyte * cap)()
        used = C.c_size_t()
        lib.sdrhip_fm_stream_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        check(lib.sdrhip_fm_stream_save(self.h, buf, cap, C.byref(used)), "sdrhip_fm_stream_save")
        return bytes(buf[: used.value])

    def restore(self, state):
        lib.sdrhip_fm_stream_restore.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        return self._pop(check(lib.sdrhip_fm_stream_restore(self.h, state, len(state)), "sdrhip_fm_stream_restore"))


class Pipe(_Handle):
    """firFilter / firDecimator / firResampler / fmDemod on host blocks (Filter.hs:532-727, Demod.hs:40-46)."""
    _destroy = lib.sdrhip_pipe_destroy

    def __init__(self, kind, desc=None, block_size_out=8192):
        super().__init__()
        self.desc = desc  # keep the descriptor alive
        self.block_size_out = block_size_out
        self.complex_in = bool(getattr(desc, "complex", False)) or kind == "fm_demod"
        self.complex_out = bool(getattr(desc, "complex", False))
        if kind == "filter":
            check(lib.sdrhip_pipe_fir_filter(C.byref(self.h), desc.h, block_size_out), "sdrhip_pipe_fir_filter")
        elif kind == "decimator":
            check(lib.sdrhip_pipe_fir_decimator(C.byref(self.h), desc.h, block_size_out), "sdrhip_pipe_fir_decimator")
        elif kind == "resampler":
            check(lib.sdrhip_pipe_fir_resampler(C.byref(self.h), desc.h, block_size_out), "sdrhip_pipe_fir_resampler")
        elif kind == "fm_demod":
            check(lib.sdrhip_pipe_fm_demod(C.byref(self.h)), "sdrhip_pipe_fm_demod")
        elif kind == "dc_blocker":
            check(lib.sdrhip_pipe_dc_blocker(C.byref(self.h)), "sdrhip_pipe_dc_blocker")
        else:
            raise ValueError(kind)
        self.kind = kind

    def push(self, block):
        """block: float32 array (interleaved for complex stages).  Returns list of output blocks."""
        b = _f32(block)
        n = b.size // (2 if self.complex_in else 1)
        self._cap = max(getattr(self, "_cap", 0), self.block_size_out, n)
        ready = check(lib.sdrhip_pipe_push(self.h, _fp(b), n), "sdrhip_pipe_push")
        return self._pop(ready)

    def set_coalesce(self, blocks):
        check(lib.sdrhip_pipe_set_coalesce(self.h, blocks), "sdrhip_pipe_set_coalesce")

    def set_adaptive(self, max_blocks):
        check(lib.sdrhip_pipe_set_adaptive(self.h, max_blocks), "sdrhip_pipe_set_adaptive")

    def input_buffer(self, n):
        """numpy view (n elements; interleaved pairs for complex stages) of the pinned staging memory of the next push."""
        ptr = lib.sdrhip_pipe_input_buffer(self.h, n)
        if not ptr:
            raise SdrHipError(lib.sdrhip_last_error().decode())
        return np.ctypeslib.as_array(C.cast(ptr, _f32p), shape=(n * (2 if self.complex_in else 1),))

    def flush(self):
        ready = check(lib.sdrhip_pipe_flush(self.h), "sdrhip_pipe_flush")
        return self._pop(ready)

    def poll(self):
        """blocks the GPU has finished meanwhile (never waits)"""
        return self._pop(check(lib.sdrhip_pipe_poll(self.h), "sdrhip_pipe_poll"))

    def save(self):
        """sdrhip_pipe_save: drains the pipe and returns its state as bytes (blocks that became ready stay poppable)."""
        lib.sdrhip_pipe_state_bytes.restype = C.c_size_t
        lib.sdrhip_pipe_state_bytes.argtypes = [C.c_void_p]
        cap = lib.sdrhip_pipe_state_bytes(self.h)
        buf = (C.c_ubyte * cap)()
        used = C.c_size_t()
        lib.sdrhip_pipe_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        check(lib.sdrhip_pipe_save(self.h, buf, cap, C.byref(used)), "sdrhip_pipe_save")
        return bytes(buf[: used.value])

    def restore(self, state, max_block=0):
        """max_block: fmDemod / dcBlockingFilter pipes hand blocks back at the length they came in; the longest one still
        pending in the state (this wrapper sizes its pop buffer by the longest block it has seen pushed)."""
        lib.sdrhip_pipe_restore.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._cap = max(getattr(self, "_cap", 0), int(max_block))
        return self._pop(check(lib.sdrhip_pipe_restore(self.h, state, len(state)), "sdrhip_pipe_restore"))

    def _pop(self, ready):
        outs = []
        cap = max(getattr(self, "_cap", 0), self.block_size_out) * (2 if self.complex_out else 1)
        for _ in range(ready):
            o = np.empty(cap, np.float32)
            got = check(lib.sdrhip_pipe_pop(self.h, _fp(o), cap // (2 if self.complex_out else 1)), "sdrhip_pipe_pop")
            outs.append(o[: got * (2 if self.complex_out else 1)].copy())
        return outs


# Operator-surface aliases with the reference's names (SDR.Filter / SDR.Demod / SDR.Util)
def firFilter(filt, block_size_out):
    return Pipe("filter", filt, block_size_out)


def firDecimator(decimator, block_size_out):
    return Pipe("decimator", decimator, block_size_out)


def firResampler(resampler, block_size_out):
    return Pipe("resampler", resampler, block_size_out)


def fmDemod():
    return Pipe("fm_demod")


def dcBlockingFilter():
    """Filter.hs:730-739."""
    return Pipe("dc_blocker")


def interleavedIQUnsignedByteToFloat(u8):
    """Util.hs:104-110: u8 IQ -> complex64 (length len/2)."""
    return DropIn.convert("convertC", u8).view(np.complex64)


def interleavedIQUnsignedByteToFloatFast(u8):
    """Util.hs:137-138 (featureSelect -> AVX2 variant)."""
    return DropIn.convert("convertCAVX", u8).view(np.complex64)
