"""diagnostic only: host-side (Python) profile of FrameEncoder.encode on one 4K frame -- where the CPU time
between kernel launches goes.  python tools/hostprof.py [frames_per_batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats, time
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rh = Rhccq(0)
enc = FrameEncoder(rh)
frames = []
for i in range(B):
    _, rgb, specs, _, _ = bench.build_inputs(rh, 2160, 3840, 1234 + i, (2, 1), 20, 20, 2.0)
    frames.append((rgb, specs))
run = (lambda: enc.encode(*frames[0])) if B == 1 else (lambda: enc.encode_batch(frames))
run()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
run()
torch.cuda.synchronize()
pr.disable()
print("wall ms", (time.perf_counter() - t0) * 1e3, {k: round(v * 1e3, 1) for k, v in enc.timings.items()})
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(40)
