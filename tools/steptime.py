"""diagnostic only: wall time of consecutive bench steps and their stage timers (warm-up effects)."""
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
for i in range(5):
    enc.timings = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.one_step(rh, enc, rgb, specs, roi, 8)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"step {i}: {1e3*(t1-t0):.1f} ms", {k: round(v * 1e3, 1) for k, v in enc.timings.items()},
          "reserved MB", torch.cuda.memory_reserved() >> 20, flush=True)
