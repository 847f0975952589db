"""diagnostic only: mini-batch steps each level-1 problem of the 4K bench frame needs (the slowest decides the phase)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
rh = Rhccq(0)
enc = FrameEncoder(rh)
for seed in (1234, 1235, 1236):
    _, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, seed, (2, 1), 20, 20, 2.0)
    S = enc.prepare(rgb, specs)
    jobs, _ = enc.level1_jobs(S)
    parts = [jb["keys_dev"][1:] if jb["has_black"] else jb["keys_dev"] for jb in jobs]
    ks = [math.ceil(int(p.numel()) * 0.2 / 10) for p in parts]
    labs, info = rh.minibatch_kmeans(parts, ks, return_info=True, return_device=True)
    print(seed, "k", ks, "steps", [int(s[5]) for s in info["state"]], "done flag", [int(s[4]) for s in info["state"]])
