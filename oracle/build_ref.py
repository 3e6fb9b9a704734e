"""Builds the one piece of the reference that compiles from its own sources with the image's
toolchain: c_utils/c_utils.pyx (Cython -> C -> gcc), the CPU helper on the Stage-III loss path.

Runs ONLY in the authoring container (needs /root/reference).  Output goes to oracle/_ref/
(git-ignored, travels to the GPU box as a built artefact).  No reference source is copied: the
.pyx is cythonized where it lies, the generated C lands in a temp dir.

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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


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


def load():
    """Import the built reference module (None if it was never built)."""
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
