"""diagnostic only: three 4K frames through FrameEncoder (for a kernel trace of the last one)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
rh = Rhccq(0)
enc = FrameEncoder(rh)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
for _ in range(3):
    enc.encode_native(rgb, specs)          # the host bench.py times (rhccq_encode_frame)
torch.cuda.synchronize()
