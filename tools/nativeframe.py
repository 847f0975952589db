"""diagnostic: the bench frame through rhccq_encode_frame, stage clocks (RHCCQ_TRACE=1 adds the per-fit phases on stderr)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
rh = Rhccq(0)
img, rgb, specs, roi_mask, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
enc = FrameEncoder(rh)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = enc.encode_native(rgb, specs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt * 1e3:.1f} ms", {k: round(v * 1e3, 2) for k, v in enc.timings.items()},
          {ci: {k: round(v * 1e3, 2) for k, v in tm.items()} for ci, tm in enc.class_timings.items()}, flush=True)
