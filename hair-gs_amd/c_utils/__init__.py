"""Counterpart of the reference's Cython module `c_utils` (c_utils/c_utils.pyx).  Only the function with a
caller is provided: filter_strand_list_segments (used by the smoothness loss, loss/losses.py:195).
Vectorised numpy over a flat (offsets, rows) form; same input/output contract as the .pyx (:83-127)."""
import numpy as np


def strands_to_flat(strands_list):
    lens = np.fromiter((s.shape[0] for s in strands_list), dtype=np.int64, count=len(strands_list))
    offsets = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    rows = (np.concatenate([np.asarray(s, np.int64).reshape(-1, 2) for s in strands_list], 0)
            if len(lens) else np.zeros((0, 2), np.int64))
    return offsets, rows


def filter_strand_segments_flat(offsets, rows):
    """All pairs of consecutive rows inside each strand: [pairs, 2, 2] int64."""
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
    if strands_list is None:
        raise TypeError("Argument 'strands_list' must not be None")
    return filter_strand_segments_flat(*strands_to_flat(strands_list))
