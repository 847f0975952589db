"""diagnostic only: the pixel-space neighbour pass (extension) at 4K"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
rh = Rhccq(0)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
print(json.dumps(bench.pixel_probe(rh, rgb)))
