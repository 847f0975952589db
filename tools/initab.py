"""diagnostic only: k-means++ chain, generations 2 and 3 side by side in ONE process (box-to-box clocks differ by 10-20 %):
us per pick on the bench frame's longest chain (k = 30 128, 90 384 init samples) and on its shortest (k = 20 556)."""
import os, sys, math, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd import synth
rh = Rhccq(0)
img = synth.photo(2160, 3840, 1234)
keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
allp = np.unique(keys[:, 1920:])
allp = allp[allp != 0]
for n in (1506367, 1027775):
    pal = allp[:n]
    k = math.ceil(len(pal) * 0.2 / 10)
    for rep in range(2):
        for gen, cw in ((2, 1), (0, 1), (0, 2), (0, 3)):
            rh.set_option(rh.OPT_INIT_KERNEL, gen)
            rh.set_option(rh.OPT_INIT_CANDS_PER_WAVE, cw)
            t = {}
            labs, info = rh.minibatch_kmeans([pal], [k], return_info=True, timing=t)
            print("N", len(pal), "k", k, "generation", gen or 3, "candidates per wave", cw, "init ms", round(t["init_ms"], 2), "us/pick",
                  round(t["init_ms"] * 1e3 / k, 3), "picks crc", zlib.crc32(info["chosen"].tobytes()), flush=True)
rh.set_option(rh.OPT_INIT_KERNEL, 0)
rh.set_option(rh.OPT_INIT_CANDS_PER_WAVE, 1)
