"""diagnostic only: two 16-frame batches of 4K frames through FrameEncoder.encode_batch (for a kernel trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rh = Rhccq(0)
enc = FrameEncoder(rh)
frames = []
for i in range(B):
    _, rgb, specs, _, _ = bench.build_inputs(rh, 2160, 3840, 1234 + i, (2, 1), 20, 10, 2.0)
    frames.append((rgb, specs))
for _ in range(2):
    t0 = time.perf_counter()
    enc.encode_batch(frames)
    torch.cuda.synchronize()
    print("batch wall ms", round((time.perf_counter() - t0) * 1e3, 1), {k: round(v * 1e3, 1) for k, v in enc.timings.items()})
    enc.timings = {}
