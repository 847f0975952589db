"""diagnostic only: wall time of consecutive bench steps vs the sum of the encoder's stage timers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder

rh = Rhccq(0)
enc = FrameEncoder(rh)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
for i in range(6):
    enc.timings = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.one_step(rh, enc, rgb, specs, roi, 8)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: host {1e3*(t1-t0):.1f} ms, +sync {1e3*(t2-t1):.1f} ms, stage sum {1e3*sum(enc.timings.values()):.1f} ms")
