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
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC}"]


# Per-file extras.  kernels_chain.hip: the SLP vectoriser packs unrelated scalar f32 ops into
# v_pk_* (no faster than scalar on gfx950: 4 cycles vs 2) and pays for it in v_mov shuffles.
FILE_FLAGS = {"kernels_chain.hip": ["-fno-slp-vectorize"], "kernels_split.hip": ["-fno-slp-vectorize"],
              "kernels_tail.hip": ["-fno-slp-vectorize"], "kernels_cplx.hip": ["-fno-slp-vectorize"],
              "kernels_resample_cycle.hip": ["-fno-slp-vectorize"], "kernels_decimate_real.hip": ["-fno-slp-vectorize"],
              # packed operations written out as 2-vectors; the vectoriser would undo the DPP-fused additions
              "kernels_systolic.hip": ["-fno-slp-vectorize"], "kernels_resample_systolic.hip": ["-fno-slp-vectorize"],
              "kernels_resample_stream.hip": ["-fno-slp-vectorize"]}
if os.environ.get("SDRHIP_SPLIT_DEFS"):   # tuning experiments, e.g. "-DSPLIT_CB=8 -DSPLIT_U=4"
    FILE_FLAGS["kernels_split.hip"] = FILE_FLAGS["kernels_split.hip"] + os.environ["SDRHIP_SPLIT_DEFS"].split()
if os.environ.get("SDRHIP_NO_SLP_FAST"):
    FILE_FLAGS["kernels_fast.hip"] = ["-fno-slp-vectorize"]
if os.environ.get("SDRHIP_CHAIN_DEFS"):   # tuning experiments on the tail kernels
    FILE_FLAGS["kernels_chain.hip"] = FILE_FLAGS["kernels_chain.hip"] + os.environ["SDRHIP_CHAIN_DEFS"].split()
if os.environ.get("SDRHIP_RSTREAM_DEFS"):  # tuning experiments on the streaming fmDemod + resampler, e.g. "-DSDRHIP_RSTREAM_MINW=5"
    FILE_FLAGS["kernels_resample_stream.hip"] = FILE_FLAGS["kernels_resample_stream.hip"] + os.environ["SDRHIP_RSTREAM_DEFS"].split()
if os.environ.get("SDRHIP_FAST_DEFS"):    # tuning experiments on the tiled decimator, e.g. "-DSDRHIP_INL_STEP=8"
    FILE_FLAGS["kernels_fast.hip"] = FILE_FLAGS.get("kernels_fast.hip", []) + os.environ["SDRHIP_FAST_DEFS"].split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(INCLUDE, "sdr_hip.h"))
    return hs


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra):
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + extra + ["-x", "hip", "-c", src, "-o", obj]
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
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    build_examples()
    return LIB


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
        cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-O2", f"-I{INCLUDE}", src, "-o", exe, f"-L{LIBDIR}", "-lsdr_hip",
               "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../../sdr_amd/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed on {src}:\n{r.stdout}\n{r.stderr}")


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True)
    print(path)
