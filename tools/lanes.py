"""diagnostic only: level-1 MiniBatchKMeans of the 4K bench frame, one batched launch sequence vs per-problem pipelines"""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
rh = Rhccq(0)
enc = FrameEncoder(rh)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
S = enc.prepare(rgb, specs)
jobs, _ = enc.level1_jobs(S)
parts = [jb["keys_dev"][1:] if jb["has_black"] else jb["keys_dev"] for jb in jobs]
ks = [math.ceil(int(p.numel()) * 0.2 / 10) for p in parts]
for lanes in (1, 4, 1, 4, 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    labs, info = rh.minibatch_kmeans(parts, ks, return_info=True, return_device=True, lanes=lanes)
    torch.cuda.synchronize()
    print("lanes", lanes, "ms", round(1e3 * (time.perf_counter() - t0), 1), "steps", [int(s[5]) for s in info["state"]])
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = {}
    rh.minibatch_kmeans([parts[i]], [ks[i]], return_device=True, timing=t)
    torch.cuda.synchronize()
    print("alone", i, "k", ks[i], "ms", round(1e3 * (time.perf_counter() - t0), 1), "init ms", round(t["init_ms"], 1))
