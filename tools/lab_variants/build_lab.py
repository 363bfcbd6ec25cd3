"""Build tools/lab_variants/liblab.so: the kernel variants that were measured, found bit-equal and slower, and moved out of the
product library in round 6 (LABNOTES.md has the numbers; profiles/k4lab, profiles/k2lab the transcripts).

    python tools/lab_variants/build_lab.py [extra -D switches for the variants, e.g. -DSDRHIP_RSTREAM_MINW=5]

Compiled with -Dsdrhip=sdrlab_ns so that the product's headers are reused under another namespace and both libraries can be loaded
into one process (check_lab.py compares every variant with libsdr_hip.so bit for bit).  Not part of `python -m sdr_amd.build`, not
loaded by anything but check_lab.py."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sdr_amd import build as B  # noqa: E402

SOURCES = ["resample_demod_stream.hip", "resample_systolic.hip", "decimate_demod_systolic.hip", "demod_forms.hip", "lab_abi.cpp"]
LIB = os.path.join(HERE, "liblab.so")


def build(extra=()):
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for f in SOURCES:
        obj = os.path.join(objdir, f + ".o")
        cmd = [B.HIPCC] + B.FLAGS + ["-fno-slp-vectorize", "-Dsdrhip=sdrlab_ns", f"-I{HERE}"] + list(extra) + ["-x", "hip", "-c", os.path.join(HERE, f), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {f}:\n{r.stderr[-4000:]}")
        objs.append(obj)
    r = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(sys.argv[1:]))
