"""Evaluation data in the common oriented-point format (reference data/eval_data.py): the ground truth of a
capture (`hair_eval_data.npz`, written by the reference's dataset scripts) and what a model under training is compared with.
The other reconstruction methods' loaders of the reference (Strand Integration, Neural Haircut) are outside this scope."""
import numpy as np

from loss.metrics import HairEvalData, compute_eval_data_from_gs, compute_eval_data_from_hair_gs  # noqa: F401


def load_hair_eval_data_npz(path):
    """points [N,3], directions [N,3] (normalised here, reference :23-37), points_id_to_strand_id [N]; `edges` of the file are
    returned as a fourth attribute."""
    d = np.load(path)
    dirs = np.asarray(d["directions"], dtype=np.float64)
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    out = HairEvalData(points=d["points"], directions=dirs, points_id_to_strand_id=d["points_id_to_strand_id"])
    out.edges = d["edges"] if "edges" in d.files else None
    return out


eval_data_loading_callbacks = {"gt": load_hair_eval_data_npz}
