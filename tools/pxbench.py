"""diagnostic only: the pixel-space neighbour pass (extension) at 4K, by radius"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
rh = Rhccq(0)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
print(json.dumps(bench.pixel_probe(rh, rgb)))
H, W = 2160, 3840
parent = rh.empty((H, W), torch.int32)
for radius in (0, 1, 2, 3, 4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    tot = 0.0
    for it in range(6):
        ev[0].record()
        rh._check(rh.lib.rhccq_px_neighbours(rh.ctx, rh._p(rgb), H, W, radius, 6.0, 1.0, 5, rh._p(rh._px_lut), rh._p(parent), C.c_void_p(0)), "px")
        ev[1].record()
        torch.cuda.synchronize()
        if it:
            tot += ev[0].elapsed_time(ev[1])
    print("radius", radius, "us", round(tot / 5 * 1e3, 1))
