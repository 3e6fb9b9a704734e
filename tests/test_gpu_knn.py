"""distCUDA2 parity (GPU): exact 3-NN, bit-exact against the oracle (and therefore the brute force)."""
import numpy as np
import pytest

from oracle import hgs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 3, 4, 5, 100, 1024, 1025, 4096, 4097, 20000, 70001])
def test_dist2_bit_exact(P):
    import torch
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(P)
    pts = (rng.normal(size=(P, 3)) * 0.2 + np.array([0.3, -0.1, 0.8])).astype(np.float32)
    if P >= 100:
        pts[: P // 10] = pts[P // 10: 2 * (P // 10)]  # duplicates -> zero distances, Morton ties
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    ref = O.dist2(pts)
    if P < 4:  # fewer than 3 neighbours: FLT_MAX sums overflow to inf in both
        assert np.array_equal(np.isinf(got), np.isinf(ref))
        return
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_dist2_empty_and_errors():
    import torch
    from simple_knn._C import distCUDA2
    assert distCUDA2(torch.empty((0, 3), device="cuda")).shape == (0,)
    with pytest.raises(Exception):
        distCUDA2(torch.zeros((4, 3)))  # CPU tensor: no CPU path
