import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, cProfile, pstats
from roibasedimagecompression_amd.ops import Rhccq, INT_MAX
from roibasedimagecompression_amd import frame as F, palette
import bench
rh = Rhccq(0)
img, rgb, specs, roi_mask, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
enc = F.FrameEncoder(rh)
out = enc.encode(rgb, specs)
S = enc.prepare(rgb, specs)
# run level1 once with the clustering result cached so that only the glue is profiled
real = palette.cluster_palettes
cache = {}
def cached(rh_, jobs):
    if "r" not in cache:
        cache["r"] = real(rh_, jobs)
    return cache["r"]
F.cluster_palettes = cached
enc.level1(S)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); enc.level1(S); torch.cuda.synchronize(); print("level1 glue ms", 1e3 * (time.perf_counter() - t0))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
