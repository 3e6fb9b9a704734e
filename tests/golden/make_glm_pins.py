"""Generates tests/golden/glm_pins.npz: random mat3 inputs and what the reference's vendored glm
(DGR/third_party/glm, through libglm_probe.so built by oracle/build_ref.py outside the repository, $HGS_REF_OUT) returns for
A*B, transpose(A), transpose(M)*M and transpose(T)*transpose(V)*T.  Runs only where /root/reference exists;
the fixture (inputs + outputs, data only) is committed.  Inputs span magnitudes the rasterizer sees
(rotation-like entries, scales 1e-4..1, Jacobians of ~1e3) so that rounding differences between
operand orders would show."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402


def main():
    so = build_ref.build_glm_probe()
    assert so, "needs /root/reference (vendored glm)"
    L = C.CDLL(so)
    rng = np.random.default_rng(20260101)
    n = 1024
    mag = 10.0 ** rng.uniform(-4, 3, (n, 1))
    a = (rng.normal(size=(n, 9)) * mag).astype(np.float32)
    b = (rng.normal(size=(n, 9)) * 10.0 ** rng.uniform(-4, 3, (n, 1))).astype(np.float32)
    out = {k: np.zeros((n, 9), np.float32) for k in ("mul", "tr", "gram", "sandwich")}
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    for i in range(n):
        L.glm_probe_mul(p(a[i]), p(b[i]), p(out["mul"][i]))
        L.glm_probe_transpose(p(a[i]), p(out["tr"][i]))
        L.glm_probe_gram(p(a[i]), p(out["gram"][i]))
        L.glm_probe_sandwich(p(a[i]), p(b[i]), p(out["sandwich"][i]))
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "glm_pins.npz"), a=a, b=b, **out)
    print("wrote glm_pins.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
