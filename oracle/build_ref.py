"""Builds the pieces of the reference that compile from its own sources with the image's toolchain:
  * c_utils/c_utils.pyx (Cython -> C -> gcc), the CPU helper on the Stage-III loss path;
  * the header-only glm the rasterizer vendors (DGR/third_party/glm), through oracle/glm_probe.cpp: a probe of
    glm's mat3 operator* / transpose, the operand order every bit-exact key of the rasterizer depends on.

Runs ONLY in the authoring container (needs /root/reference).  Output goes OUTSIDE the repository tree
($HGS_REF_OUT, default /tmp/hgs_ref): nothing compiled from the reference is ever part of the snapshot that
travels to the GPU box -- only the numeric fixtures generated with it (tests/golden/*.npz) and the timing
recorded in BASELINE.md do.  No reference source is copied: the .pyx is cythonized where it lies, the generated
C lands in a temp dir.

The CUDA rasterizer / simple-knn sources are NOT buildable here (need nvcc, cuda_runtime,
cooperative_groups, CUB, thrust) and no stand-ins are written for them -- see DESIGN.md.
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

REF = "/root/reference/c_utils/c_utils.pyx"
OUT = os.environ.get("HGS_REF_OUT", "/tmp/hgs_ref")
_HERE = os.path.dirname(os.path.abspath(__file__))
assert not os.path.abspath(OUT).startswith(os.path.dirname(_HERE) + os.sep), "HGS_REF_OUT must lie outside the repository"


def build():
    if not os.path.exists(REF):
        return None
    os.makedirs(OUT, exist_ok=True)
    existing = glob.glob(os.path.join(OUT, "c_utils*.so"))
    if existing and os.path.getmtime(existing[0]) >= os.path.getmtime(REF):
        return existing[0]
    import numpy as np
    tmp = tempfile.mkdtemp(prefix="hgs_ref_")
    try:
        c_file = os.path.join(tmp, "c_utils.c")
        subprocess.check_call([sys.executable, "-m", "cython", "-3", REF, "-o", c_file], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        ext = sysconfig.get_config_var("EXT_SUFFIX")
        so = os.path.join(OUT, "c_utils" + ext)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-I" + sysconfig.get_paths()["include"],
                               "-I" + np.get_include(), c_file, "-o", so])
        return so
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


GLM_INC = "/root/reference/submodules/diff-gaussian-rasterization/third_party/glm"
GLM_SO = os.path.join(OUT, "libglm_probe.so")


def build_glm_probe():
    """g++ on oracle/glm_probe.cpp with the reference's vendored glm on the include path (plain g++, no stand-ins)."""
    if not os.path.exists(os.path.join(GLM_INC, "glm", "glm.hpp")):
        return None
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glm_probe.cpp")
    if os.path.exists(GLM_SO) and os.path.getmtime(GLM_SO) >= os.path.getmtime(src):
        return GLM_SO
    subprocess.check_call(["g++", "-O0", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-I" + GLM_INC, src, "-o", GLM_SO])
    return GLM_SO


def load():
    """Import the built reference module (None if it was never built).  Fixture generators and tools/ref_cython_timing.py
    only: nothing that runs on the GPU box may call this."""
    so = glob.glob(os.path.join(OUT, "c_utils*.so"))
    if not so:
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("c_utils", so[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build())
    print(build_glm_probe())
