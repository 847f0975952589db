"""one-off soak: many randomised frames, fused HIP encoder vs the oracle (same generator as tests/test_gpu_frame.py's
fuzz test, other seeds).  python tests/soak/soak.py FIRST COUNT"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_frame as T
from roibasedimagecompression_amd.ops import Rhccq

rh = Rhccq(0)
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_frame_fuzz_vs_oracle.__wrapped__(rh, seed) if hasattr(T.test_frame_fuzz_vs_oracle, "__wrapped__") else T.test_frame_fuzz_vs_oracle(rh, seed)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:200], flush=True)
    if (seed - first) % 25 == 24:
        print("done", seed - first + 1, "bad", bad, "elapsed", round(time.time() - t0), flush=True)
print("total", count, "bad", bad)
