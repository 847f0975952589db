#!/usr/bin/env python3
"""G13: cluster_palette_colors_parallel with min_samples > 1 (clustering.py:160-437: DBSCAN leaves noise points, :262-271 keeps each as a
palette entry of its own) from the REFERENCE itself.  Build container only (imports /root/reference):

    python tests/golden/make_golden_dbscan_ms.py

The pipeline never calls the function this way (compute_clustering_params returns min_samples = 1), but 2 is the function's own
default.  Per case: the input crop, quality, min_samples, sklearn's DBSCAN labels on the crop's non-black palette (the call of
clustering.py:233-235), and the function's palette / indices.  Only data is written."""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402
from sklearn.cluster import DBSCAN  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from encoder.compression import clustering as R_clu  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def png(name):
    return np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"), dtype=np.uint8)


def main():
    lenna, kod = png("Lenna.png"), png("kodak_23.png")
    rng = np.random.default_rng(13)
    sparse = (rng.integers(0, 12, (40, 40, 3)) * 21).astype(np.uint8)           # a coarse lattice: isolated colours -> noise
    sparse[:3] = 0
    imgs = [lenna[230:262, 230:262].copy(), (lenna[300:340, 300:340] // 8 * 8).astype(np.uint8), kod[200:232, 200:232].copy(), sparse]
    out, k = {}, 0
    for im in imgs:
        for q, ms in ((60, 2), (80, 2), (80, 4), (90, 3), (95, 2), (100, 2)):
            d = quiet(R_clu.get_all_unique_colors, im, (0, 0))
            eps, _, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
            o = quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=ms, max_colors_per_cluster=mc)
            pal = np.array(d["palette"], np.int64).reshape(-1, 3)
            nb = pal[~np.all(pal == 0, axis=1)]
            lab = DBSCAN(eps=eps / 255.0, min_samples=ms, metric="euclidean").fit_predict(nb / 255.0)
            out[f"img{k}"] = im
            out[f"qm{k}"] = np.array([q, ms])
            out[f"lab{k}"] = lab.astype(np.int32)
            out[f"pal{k}"] = np.array(o["palette"], np.int64).reshape(-1, 3).astype(np.uint8)
            out[f"idx{k}"] = np.array(o["indices"], np.int64).reshape(-1).astype(np.int32)
            print(k, im.shape, q, ms, "colours", d["actual_colors"], "->", len(o["palette"]), "noise", int((lab == -1).sum()), "clusters", int(lab.max()) + 1)
            k += 1
    out["n"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "g13_dbscan_min_samples.npz"), **out)
    print("wrote g13_dbscan_min_samples.npz")


if __name__ == "__main__":
    main()
