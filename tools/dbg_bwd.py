import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import numpy as np
from oracle import hgs_oracle as O
from tests import scenes, gpu_util as G
s = scenes.random_scene(P=300, W=64, H=48, seed=1, sh_degree=0)
dpix = np.random.default_rng(123).normal(size=(3, s["H"], s["W"])).astype(np.float32)
f = O.forward(s); fw = G.run_forward(s); got = G.intermediates(s, fw)
f["n_contrib"], f["final_T"] = got["n_contrib"], got["final_T"]
gr = O.backward(s, f, dpix); g = G.run_backward(s, fw, dpix)
acc = gr["acc"]
mine = np.concatenate([g["dL_dmeans2D"][:, :2], g["dL_dconic"][:, [0, 1, 3]], g["dL_dopacity"], g["dL_dcolors"]], 1)
vis = np.nonzero(f["radii"] > 0)[0][:6]
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for i in vis:
    print(i, "tiles", f["tiles_touched"][i]); print("  ref", acc[i]); print("  gpu", mine[i])
