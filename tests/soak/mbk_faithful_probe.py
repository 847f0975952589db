#!/usr/bin/env python3
"""diagnostic (build container only): oracle.minibatch_kmeans_labels against scikit-learn's MiniBatchKMeans itself, with
np.argsort inside sklearn.cluster._kmeans forced stable (the one step whose tie order is unportable)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import sklearn.cluster._kmeans as km
from sklearn.cluster import MiniBatchKMeans
from threadpoolctl import threadpool_limits

from oracle import rhccq_oracle as O


class _StableNp:
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, *args, **kw):
        kw.setdefault("kind", "stable")
        return np.argsort(a, *args, **kw)


def sk_fit(P, k, stable=True, threads=None):
    old = km.np
    if stable:
        km.np = _StableNp()
    try:
        m = MiniBatchKMeans(n_clusters=k, batch_size=1000, random_state=42, n_init="auto")
        if threads:
            with threadpool_limits(limits=threads, user_api="openmp"):
                lab = m.fit_predict(P.astype(np.float64))
        else:
            lab = m.fit_predict(P.astype(np.float64))
    finally:
        km.np = old
    return lab, m


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    q = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    base = rng.integers(0, 256, (64, 3))
    P = np.unique(np.clip(base[rng.integers(0, 64, n * 2)] + rng.normal(0, 14, (n * 2, 3)), 0, 255).astype(np.uint8), axis=0)[:n]
    P = P[~np.all(P == 0, axis=1)]
    k = int(np.ceil(len(P) * (q / 100) / 10))
    print("n", len(P), "k", k)
    for threads in (1, None):
        t = time.time()
        lab, m = sk_fit(P, k, True, threads)
        print("sklearn threads", threads, "steps", m.n_steps_, "time %.2f" % (time.time() - t))
        t = time.time()
        ol, info = O.minibatch_kmeans_labels(P, k, return_info=True)
        print("oracle steps", info["n_steps"], "time %.2f" % (time.time() - t))
        print("  centres equal", np.array_equal(info["centers"], m.cluster_centers_), "labels equal", np.array_equal(ol, lab),
              "max centre diff", np.abs(info["centers"] - m.cluster_centers_).max())
