"""Counterpart of the reference's Cython module `c_utils` (c_utils/c_utils.pyx).  Only the function with a
caller is provided: filter_strand_list_segments (used by the smoothness loss, loss/losses.py:195).

  filter_strand_list_segments(strands_list)   the reference's contract (object array of [n_j, 2] int64 arrays ->
                                              [pairs, 2, 2] int64, .pyx:83-127): native code (_c_utils.c, CPython +
                                              numpy C API, built in-tree by hgs_runtime.build()) like the reference's
                                              compiled Cython -- no Python-level loop over the strands.
  filter_strand_segments_flat(offsets, rows)  the same pairs from the flat (offsets, rows) form the strand bookkeeping of
                                              this package keeps (scene.hair_topology.StrandsInfo.flat), what
                                              HairGaussianModel uses; native as well.
"""
import glob
import os
import subprocess
import sys
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_native = None


def build(force=False):
    """gcc -shared of _c_utils.c against this interpreter's and numpy's headers (in-tree; travels with the source)."""
    src = os.path.join(_HERE, "_c_utils.c")
    so = os.path.join(_HERE, "_c_utils" + sysconfig.get_config_var("EXT_SUFFIX"))
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # built under a private name and renamed into place: the ranks of a torchrun job (or pytest-xdist workers) may all
        # find the module missing at once, and none of them may import a half-written file
        tmp = f"{so}.{os.getpid()}.tmp"
        try:
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I" + sysconfig.get_paths()["include"],
                                   "-I" + np.get_include(), src, "-o", tmp])
            os.replace(tmp, so)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return so


class _NumpyFallback:
    """The numpy forms below behind the native module's two names -- only where the extension can neither be found nor
    built (no gcc): a HOST-side helper of the strand bookkeeping, not part of the GPU path (which has no fallback)."""

    @staticmethod
    def filter_strand_segments_flat(offsets, rows):
        return filter_strand_segments_flat_numpy(np.asarray(offsets, np.int64), np.asarray(rows, np.int64).reshape(-1, 2))

    @staticmethod
    def filter_strand_list_segments(strands_list):
        return filter_strand_segments_flat_numpy(*strands_to_flat(strands_list))


def _load():
    """The native module; built on first use when the .so is missing or older than its source (it is git-ignored: a fresh
    checkout of a CPU-only environment has none until something builds it)."""
    global _native
    if _native is None:
        src = os.path.join(_HERE, "_c_utils.c")
        so = glob.glob(os.path.join(_HERE, "_c_utils*.so"))
        try:
            stale = not so or os.path.getmtime(so[0]) < os.path.getmtime(src)
        except OSError:          # (the file another process is just replacing)
            stale = True
        if stale:
            try:
                so = [build()]
            except (OSError, subprocess.CalledProcessError) as e:
                if not so:
                    import warnings
                    warnings.warn(f"c_utils: native module not built ({e}); using the numpy form")
                    _native = _NumpyFallback
                    return _native
        import importlib.util
        spec = importlib.util.spec_from_file_location("c_utils._c_utils", so[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.modules["c_utils._c_utils"] = mod
        _native = mod
    return _native


def strands_to_flat(strands_list):
    lens = np.fromiter((s.shape[0] for s in strands_list), dtype=np.int64, count=len(strands_list))
    offsets = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    rows = (np.concatenate([np.asarray(s, np.int64).reshape(-1, 2) for s in strands_list], 0)
            if len(lens) else np.zeros((0, 2), np.int64))
    return offsets, rows


def filter_strand_segments_flat(offsets, rows):
    """All pairs of consecutive rows inside each strand: [pairs, 2, 2] int64 (native; 20x the numpy gather below)."""
    return _load().filter_strand_segments_flat(offsets, rows)


def filter_strand_segments_flat_numpy(offsets, rows):
    """The same result as one vectorised numpy gather (kept as the check of the native function)."""
    total = rows.shape[0]
    if total == 0:
        return np.empty((0, 2, 2), np.int64)
    keep = np.ones(total, bool)
    ends = offsets[1:][offsets[1:] > offsets[:-1]] - 1  # last row of every non-empty strand has no successor
    keep[ends] = False
    first = np.nonzero(keep)[0]
    return np.stack([rows[first], rows[first + 1]], axis=1)


def filter_strand_list_segments(strands_list):
    """strands_list: 1-D object array of [n_j, 2] int64 arrays -> [sum(max(n_j-1,0)), 2, 2] int64."""
    return _load().filter_strand_list_segments(strands_list)
