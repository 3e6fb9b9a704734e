import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
from tests import gpu_util as G
from tests.test_gpu_raster import _scene
from oracle import hgs_oracle as O
s = _scene(sys.argv[1] if len(sys.argv) > 1 else "opaque_early_stop")
ref = O.forward(s); fw = G.run_forward(s); got = G.intermediates(s, fw)
d = got["n_contrib"] != ref["n_contrib"]
ys, xs = np.nonzero(d.reshape(s["H"], s["W"]))
print("flips", d.sum())
L = ref["ranges"][:, 1] - ref["ranges"][:, 0]
print("tile lens", L.tolist())
for y, x in list(zip(ys, xs))[:20]:
    t = (y // 16) * ((s["W"] + 15) // 16) + x // 16
    print(y, x, "tile", t, "L", L[t], "got", got["n_contrib"].reshape(s["H"], s["W"])[y, x], "ref", ref["n_contrib"].reshape(s["H"], s["W"])[y, x],
          "T got/ref", got["final_T"].reshape(s["H"], s["W"])[y, x], ref["final_T"].reshape(s["H"], s["W"])[y, x])
