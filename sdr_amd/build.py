"""Build libsdr_hip.so (gfx950) in-tree with hipcc.

    python -m sdr_amd.build [--force] [--save-temps]

Every translation unit is compiled with -ffp-contract=off: the library's parity
with the reference depends on unfused multiply/add.
hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU container.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsdr_hip.so")
# the measurement utilities of bench.py / tools (include/sdr_hip_bench.h): a library of their own beside the product, linked against it
CSRC_BENCH = os.path.join(HERE, "csrc_bench")
LIB_BENCH = os.path.join(LIBDIR, "libsdr_hip_bench.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REPO = os.path.dirname(HERE)
# -ffile-prefix-map: no absolute path of the checkout reaches the objects (__FILE__, debug / assert strings), so that a clean build of
# the same tree in ANY directory gives the same libsdr_hip.so, byte for byte -- the sha256 build() and smoke() print then says
# "a build of this tree", not just "the file that was pushed" (README.md "Reproducible build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", f"-ffile-prefix-map={REPO}=.",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC}"]


# Per-file extras.  kernels_chain.hip: the SLP vectoriser packs unrelated scalar f32 ops into
# v_pk_* (no faster than scalar on gfx950: 4 cycles vs 2) and pays for it in v_mov shuffles.
FILE_FLAGS = {"kernels_chain.hip": ["-fno-slp-vectorize"], "kernels_split.hip": ["-fno-slp-vectorize"],
              "kernels_tail.hip": ["-fno-slp-vectorize"], "kernels_cplx.hip": ["-fno-slp-vectorize"],
              "kernels_resample_cycle.hip": ["-fno-slp-vectorize"], "kernels_decimate_real.hip": ["-fno-slp-vectorize"],
              # packed operations written out as 2-vectors; the vectoriser would undo the DPP-fused additions
              "kernels_systolic.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def bench_sources():
    return sorted(os.path.join(CSRC_BENCH, f) for f in os.listdir(CSRC_BENCH) if f.endswith((".hip", ".cpp")))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(INCLUDE, "sdr_hip.h"))
    hs.append(os.path.join(INCLUDE, "sdr_hip_bench.h"))
    return hs


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _cuid(src):
    """hipcc's compilation-unit id (the `__hip_cuid_<id>` symbol) is by default a hash of the ABSOLUTE source path + options; pin it
    to the file's name so that it does not depend on where the checkout lies"""
    import hashlib
    return hashlib.sha1(("sdr_hip:" + os.path.basename(src)).encode()).hexdigest()[:16]


def _compile(src, obj, extra):
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra + [f"-cuid={_cuid(src)}", "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return r.stderr


def build(force=False, save_temps=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = headers()
    jobs = []
    objs = []
    extra = ["-save-temps=obj"] if save_temps else []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + hdrs):
            jobs.append((src, obj))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for warn in ex.map(lambda j: _compile(j[0], j[1], extra), jobs):
                if verbose and warn:
                    sys.stderr.write(warn)
    if jobs or force or _stale(LIB, objs):
        # objects in sorted order (sources() sorts); -pthread: chain.cpp's staging-copy helpers are std::thread
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    build_bench_lib(force, hdrs, extra)
    build_examples()
    return LIB


def build_bench_lib(force=False, hdrs=None, extra=()):
    """libsdr_hip_bench.so: csrc_bench/*.{hip,cpp} against the product library (rpath $ORIGIN: the two travel together)."""
    hdrs = hdrs if hdrs is not None else headers()
    objs, jobs = [], []
    for src in bench_sources():
        obj = os.path.join(OBJ, "bench_" + os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + hdrs):
            jobs.append((src, obj))
    for src, obj in jobs:
        _compile(src, obj, list(extra))
    if jobs or force or _stale(LIB_BENCH, objs + [LIB]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_BENCH] + objs + [f"-L{LIBDIR}", "-lsdr_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_BENCH


EXAMPLES = os.path.join(os.path.dirname(HERE), "examples")


def build_examples():
    """Plain-C programs over the C ABI (gcc, no HIP headers): examples/bin/<name>."""
    bindir = os.path.join(EXAMPLES, "bin")
    os.makedirs(bindir, exist_ok=True)
    for f in sorted(os.listdir(EXAMPLES)):
        if not f.endswith(".c"):
            continue
        src, exe = os.path.join(EXAMPLES, f), os.path.join(bindir, f[:-2])
        if not _stale(exe, [src, LIB, os.path.join(INCLUDE, "sdr_hip.h")]):
            continue
        cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-O2", f"-I{INCLUDE}", src, "-o", exe, f"-L{LIBDIR}", "-lsdr_hip", "-lm",
               "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../../sdr_amd/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed on {src}:\n{r.stdout}\n{r.stderr}")


# ---- sanitizer variant (round 5) -------------------------------------------------------------------------------------------
# `SDRHIP_ASAN=1 python -m sdr_amd.build` (or --asan): the same sources with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST
# code only (-Xarch_host: the device code is compiled as always) -> sdr_amd/lib/libsdr_hip_asan.so, and the plain-C programs of
# examples/ linked against it with clang's shared sanitizer runtime -> examples/bin/<name>_asan.  tests/test_gpu_sanitizers.py runs
# them on the GPU box; a clean log is kept under profiles/.  Not part of the product: nothing loads the _asan library by default.
OBJ_ASAN = os.path.join(HERE, "_obj_asan")
LIB_ASAN = os.path.join(LIBDIR, "libsdr_hip_asan.so")
ASAN_FLAGS = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined", "-g"]


def _clang_rt_dir():
    import glob
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return os.path.dirname(hits[0]) if hits else None


def build_asan(force=False):
    os.makedirs(OBJ_ASAN, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = headers()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_ASAN, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + hdrs):
            jobs.append((src, obj))
    if jobs:
        # -O1 instead of -O3 for the host side would change the device code too (one flag set per translation unit): keep -O3, the
        # sanitizers instrument what is left
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], ASAN_FLAGS), jobs))
    if jobs or force or _stale(LIB_ASAN, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", LIB_ASAN] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    rt = _clang_rt_dir()
    bindir = os.path.join(EXAMPLES, "bin")
    os.makedirs(bindir, exist_ok=True)
    clang = os.path.join(os.path.dirname(os.path.dirname(rt)), "..", "..", "..", "bin", "clang") if rt else "clang"
    clang = os.path.normpath(clang) if os.path.exists(os.path.normpath(clang)) else "/opt/rocm/lib/llvm/bin/clang"
    for f in sorted(os.listdir(EXAMPLES)):
        if not f.endswith(".c"):
            continue
        src, exe = os.path.join(EXAMPLES, f), os.path.join(bindir, f[:-2] + "_asan")
        if not (force or _stale(exe, [src, LIB_ASAN, os.path.join(INCLUDE, "sdr_hip.h")])):
            continue
        cmd = [clang, "-std=c99", "-Wall", "-Wextra", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libsan", "-DSDRHIP_FAST_EXIT",
               f"-I{INCLUDE}", src, "-o", exe, f"-L{LIBDIR}", "-lsdr_hip_asan", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../../sdr_amd/lib"]
        if rt:
            cmd.append("-Wl,-rpath," + rt)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"clang failed on {src}:\n{r.stdout}\n{r.stderr}")
    return LIB_ASAN


def sha256_of(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


if __name__ == "__main__":
    if "--asan" in sys.argv or os.environ.get("SDRHIP_ASAN") == "1":
        print(build_asan(force="--force" in sys.argv))
    else:
        path = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True)
        print(path, "sha256", sha256_of(path))
