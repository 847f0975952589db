"""diagnostic only: ONE MiniBatchKMeans problem shaped like a 4K segment of the bench frame (k-means++ chain + steps);
run under rocprofv3 --pmc to read the instruction mix / stall counters of mbk_init2_kernel, or alone for us per pick."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd import synth
rh = Rhccq(0)
img = synth.photo(2160, 3840, 1234)
keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
pal = np.unique(keys[:, 1920:])
pal = pal[pal != 0]
k = math.ceil(len(pal) * 0.2 / 10)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(reps):
    t = {}
    labs, info = rh.minibatch_kmeans([pal], [k], return_info=True, timing=t)
    print("N", len(pal), "k", k, "init ms", round(t["init_ms"], 2), "us/pick", round(t["init_ms"] * 1e3 / k, 3), "picks sha", hash(info["chosen"].tobytes()) & 0xffffffff, flush=True)
