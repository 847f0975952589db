#!/usr/bin/env python3
"""G11: MiniBatchKMeans golden vectors from scikit-learn ITSELF, at the reference's call site
(encoder/compression/clustering.py:207-218: MiniBatchKMeans(n_clusters=ceil(N*(q/100)/10), batch_size=1000,
random_state=42, n_init='auto').fit_predict(colours as float64)).  Build container only.

    python tests/golden/make_golden_mbk.py

Per case two fits are recorded:
  * "stable": the same call with the ONE np.argsort inside sklearn.cluster._kmeans._mini_batch_step forced to
    kind='stable' (its default unstable sort over massively tied counts depends on numpy's SIMD sort kernel for the
    host CPU; everything else is deterministic).  Recorded: the k-means++ picks (captured from _kmeans_plusplus),
    the init sample rows, n_steps_, cluster_centers_ (float64, bit for bit), sha256 of labels_, cluster sizes.
    The oracle and the HIP path must reproduce all of it exactly (Tier A).  Run with 1 OpenMP thread and with all
    cores: the generator asserts both give the same arrays.
  * "default": the untouched call (what the reference really runs, on this CPU): n_steps_, number of non-empty clusters,
    inertia_, PSNR of colour -> centre, and whether it happens to equal the stable fit.  Tier B reference numbers.

Inputs: images the reference ships (images/png/Lenna.png, 1.png, 13.png, 23.png -- copied as DATA into tests/golden/) and
the bench generator's synthetic photo (regenerated from its seed; sha256 of the pixels recorded).  No reference source
text is copied.  Versions in g11_mbk_sklearn.json."""
import hashlib
import json
import math
import os
import shutil
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np  # noqa: E402
import sklearn  # noqa: E402
import sklearn.cluster._kmeans as km  # noqa: E402
from PIL import Image  # noqa: E402
from sklearn.cluster import MiniBatchKMeans  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

from roibasedimagecompression_amd import synth  # noqa: E402  (pure numpy generator, no GPU)


class _StableNp:
    """numpy with argsort defaulting to kind='stable' (stands in for `np` inside sklearn.cluster._kmeans only)"""
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, *args, **kw):
        kw.setdefault("kind", "stable")
        return np.argsort(a, *args, **kw)


def fit(X, k, stable, threads=None):
    cap = {}
    real_pp = km._kmeans_plusplus

    def spy(Xs, n_clusters, **kw):
        centers, indices = real_pp(Xs, n_clusters, **kw)
        cap["picks"] = np.asarray(indices).copy()
        cap["sample"] = np.asarray(Xs).copy()
        return centers, indices
    old_np = km.np
    km._kmeans_plusplus = spy
    if stable:
        km.np = _StableNp()
    try:
        m = MiniBatchKMeans(n_clusters=k, batch_size=1000, random_state=42, n_init="auto")
        if threads:
            with threadpool_limits(limits=threads, user_api="openmp"):
                lab = m.fit_predict(X)
        else:
            lab = m.fit_predict(X)
    finally:
        km.np = old_np
        km._kmeans_plusplus = real_pp
    return lab.astype(np.int32), m, cap


def psnr_of(P, centres, lab):
    err = P.astype(np.float64) - centres[lab]
    mse = float(np.mean(err * err))
    return 10 * math.log10(255.0 ** 2 / mse)


def palette_of(img):
    u = np.unique(img.reshape(-1, 3), axis=0)
    return u[~np.all(u == 0, axis=1)]                     # clustering.py:185-199: black rows are set aside


def main():
    for name in ("Lenna.png", "1.png", "13.png", "23.png"):
        dst = os.path.join(HERE, "kodak_" + name if name != "Lenna.png" else "Lenna.png")
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF, "images/png", name), dst)
            os.chmod(dst, 0o644)

    def png(name):
        return np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"), dtype=np.uint8)

    cases = [
        ("lenna192_q20", png("Lenna.png")[128:320, 128:320], 20),
        ("lenna192_q10", png("Lenna.png")[128:320, 128:320], 10),
        ("lenna_full_q20", png("Lenna.png"), 20),
        ("kodak1_q20", png("kodak_1.png"), 20),
        ("kodak13_q10", png("kodak_13.png"), 10),
        ("kodak23_q20", png("kodak_23.png"), 20),
        ("synth_photo_1024_q20", synth.photo(1024, 1024, 1234), 20),          # bench generator, >= 1 Mpx
        ("synth_photo_640_q40", synth.photo(640, 640, 1235, sigma=3.0), 40),
    ]
    meta = {"numpy": np.__version__, "scikit-learn": sklearn.__version__, "python": sys.version.split()[0],
            "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
            "cases": {}}
    arrs = {}
    for name, img, q in cases:
        P = palette_of(img)
        n = len(P)
        assert n >= 10000, (name, n)
        k = math.ceil(n * (q / 100) / 10)
        X = P.astype(np.float64)
        t = time.time()
        lab1, m1, cap1 = fit(X, k, True, threads=1)
        labN, mN, capN = fit(X, k, True, threads=None)
        same_threads = (np.array_equal(lab1, labN) and np.array_equal(m1.cluster_centers_, mN.cluster_centers_)
                        and m1.n_steps_ == mN.n_steps_)
        assert same_threads, name
        t_stable = time.time() - t
        t = time.time()
        labD, mD, capD = fit(X, k, False, threads=None)
        t_default = time.time() - t
        assert np.array_equal(capD["picks"], cap1["picks"]), name               # the init does not depend on the argsort
        # init sample rows: recover the row indices from the colours (duplicates resolve to the same colour anyway)
        sample = cap1["sample"].astype(np.uint8)
        arrs[f"{name}_picks"] = cap1["picks"].astype(np.int32)
        arrs[f"{name}_sample_sha"] = np.frombuffer(hashlib.sha256(sample.tobytes()).digest(), np.uint8)
        arrs[f"{name}_centres"] = m1.cluster_centers_.astype(np.float64)
        arrs[f"{name}_sizes"] = np.bincount(lab1, minlength=k).astype(np.int32)
        c = {"q": q, "n": n, "k": k, "img_sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
             "img_shape": list(img.shape),
             "stable": {"n_steps": int(m1.n_steps_), "labels_sha256": hashlib.sha256(lab1.tobytes()).hexdigest(),
                        "n_nonempty": int((np.bincount(lab1, minlength=k) > 0).sum()), "inertia": float(m1.inertia_),
                        "psnr": psnr_of(P, m1.cluster_centers_, lab1), "seconds_1thread_plus_allcores": round(t_stable, 2)},
             "default": {"n_steps": int(mD.n_steps_), "n_nonempty": int((np.bincount(labD, minlength=k) > 0).sum()),
                         "inertia": float(mD.inertia_), "psnr": psnr_of(P, mD.cluster_centers_, labD),
                         "equals_stable": bool(np.array_equal(labD, lab1) and np.array_equal(mD.cluster_centers_, m1.cluster_centers_)),
                         "seconds_allcores": round(t_default, 2)}}
        meta["cases"][name] = c
        print(name, json.dumps(c))
    np.savez_compressed(os.path.join(HERE, "g11_mbk_sklearn.npz"), **arrs)
    json.dump(meta, open(os.path.join(HERE, "g11_mbk_sklearn.json"), "w"), indent=1)
    print("wrote g11_mbk_sklearn.npz / .json")


if __name__ == "__main__":
    main()
