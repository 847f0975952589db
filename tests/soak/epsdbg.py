import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, os
from roibasedimagecompression_amd.ops import Rhccq, pack_rgb
from oracle import rhccq_oracle as O
rh = Rhccq(0)
g = np.load("tests/golden/g3_dbscan.npz")
eps = [float(e) for e in g["eps"]]
bad = 0
for name in g["names"]:
    P = g[f"pal_{name}"]
    for ei, e in enumerate(eps):
        labs, nc = rh.eps_components([pack_rgb(P)], [e])
        w = g[f"lab_{name}_{ei}"]
        if not np.array_equal(labs[0], w):
            bad += 1
            print("MISMATCH", name, len(P), "eps", e, "ncomp", nc[0], "want", w.max() + 1, "agree", (labs[0] == w).mean(), O.eps_threshold(e))
print("bad", bad)
