#!/usr/bin/env python3
"""BASELINE.md B2, reference side: the reference's own c_utils.pyx (built by oracle/build_ref.py OUTSIDE the repository)
against this package's native module on the same cores.  Runs in the authoring container only (needs /root/reference);
its output is recorded in BASELINE.md -- nothing compiled from the reference travels to the GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np


def med(fn, n=9):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))


def main():
    from oracle import build_ref
    import c_utils
    c_utils.build()
    build_ref.build()
    ref = build_ref.load()
    out = {"cores": os.cpu_count(), "ref_out": build_ref.OUT}
    rng = np.random.default_rng(0)
    for S in (2000, 5000, 10000):
        strands = np.empty(S, dtype=object)
        for j in range(S):
            strands[j] = rng.integers(0, 10**6, size=(100, 2)).astype(np.int64)
        a, b = ref.filter_strand_list_segments(strands), c_utils.filter_strand_list_segments(strands)
        assert np.array_equal(np.asarray(a), np.asarray(b))
        out[f"S{S}_reference_cython_ms"] = med(lambda: ref.filter_strand_list_segments(strands)) * 1e3
        out[f"S{S}_this_package_ms"] = med(lambda: c_utils.filter_strand_list_segments(strands)) * 1e3
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
