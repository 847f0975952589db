"""diagnostic only: k-means++ chain time per pick by number of shards (workgroups per problem), one problem shaped like
a 4K segment (1.5 M colours, k = 30 000, 90 000 init samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roibasedimagecompression_amd.ops import Rhccq, pack_rgb

rh = Rhccq(0)
rng = np.random.default_rng(7)
base = rng.integers(0, 256, (40, 3))
P = np.unique(np.clip(base[rng.integers(0, 40, 2600000)] + rng.normal(0, 22, (2600000, 3)), 0, 255).astype(np.uint8), axis=0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
keys = pack_rgb(P)
print("colours", len(P), "k", k, "init samples", 3 * k)
ref = None
for shards in (1, 2, 4, 8, 1):
    rh.set_option(rh.OPT_INIT_SHARDS, shards)
    t = {}
    labs, info = rh.minibatch_kmeans([keys], [k], return_info=True, timing=t)
    ch = info["chosen"][:k]
    if ref is None:
        ref = ch
    print(f"shards {shards}: init {t['init_ms']:8.2f} ms = {1e3 * t['init_ms'] / k:6.3f} us per pick; picks equal: {np.array_equal(ch, ref)}")
