"""diagnostic only: k-means++ chain of level-2-sized problems -- register chain (kpp_flat.h) against the third generation, us per
pick in one process; and KMeans on a level-3-sized palette, wall ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch
from roibasedimagecompression_amd.ops import Rhccq

rh = Rhccq(0)
rng = np.random.default_rng(5)
for n, k in ((40964, 1639), (59273, 1186), (200000, 2000), (12000, 120)):
    P = np.unique((rng.normal(128, 45, (n * 2, 3))).clip(0, 255).astype(np.uint8), axis=0)[:n]
    keys = (P[:, 0].astype(np.uint32) << 16) | (P[:, 1].astype(np.uint32) << 8) | P[:, 2]
    for opt, name in ((0, "auto"), (3, "third generation"), (0, "auto"), (3, "third generation")):
        rh.set_option(rh.OPT_INIT_KERNEL, opt)
        t = {}
        rh.minibatch_kmeans([keys], [k], timing=t)
        print(f"n {len(keys)} k {k} {name:18s} init {t['init_ms']:.3f} ms = {1e3 * t['init_ms'] / k:.2f} us per pick", flush=True)
rh.set_option(rh.OPT_INIT_KERNEL, 0)
P = np.unique((rng.normal(128, 45, (9000, 3))).clip(0, 255).astype(np.uint8), axis=0)[:4004]
keys = (P[:, 0].astype(np.uint32) << 16) | (P[:, 1].astype(np.uint32) << 8) | P[:, 2]
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, info = rh.kmeans_split([keys], [364], return_info=True)
    torch.cuda.synchronize()
    print(f"KMeans n {len(keys)} k 364: {1e3 * (time.perf_counter() - t0):.3f} ms, iterations {info[0][0]}", flush=True)
