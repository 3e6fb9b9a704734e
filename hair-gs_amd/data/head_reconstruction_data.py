"""Head reconstruction side file of a capture (reference data/head_reconstruction_data.py): head and scalp vertices; the scalp
vertices are the reference strand roots the strands are oriented by."""
from typing import NamedTuple

import numpy as np


class HeadReconstruction(NamedTuple):
    head_verts: np.ndarray
    scalp_verts: np.ndarray


def save_head_reconstruction_data_npz(file_path, head_verts, scalp_verts):
    """(The reference takes its HairData / HeadData containers, :20-35; the arrays they contribute are passed directly here.)"""
    np.savez(file_path, head_verts=np.asarray(head_verts), scalp_verts=np.asarray(scalp_verts))


def load_head_reconstruction_data_npz(path):
    d = np.load(path)
    return HeadReconstruction(head_verts=d["head_verts"], scalp_verts=d["scalp_verts"])
