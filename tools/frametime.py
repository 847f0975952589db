"""diagnostic only: one 4K frame, wall time per C-ABI call (synchronising around each) and the host time between
calls, per stage of FrameEncoder.encode."""
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
for _ in range(3):
    enc.encode(rgb, specs)
torch.cuda.synchronize()
lib = rh.lib
marks = []


class L:
    def __init__(self, lib):
        self._lib, self._w = lib, {}

    def __getattr__(self, n):
        f = getattr(self._lib, n)
        if not n.startswith("rhccq_") or n in ("rhccq_last_error", "rhccq_mbk_work_bytes", "rhccq_mbk_order_bytes", "rhccq_params", "rhccq_eps_threshold"):
            return f
        if n not in self._w:
            def g(*a, _f=f, _n=n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = _f(*a)
                torch.cuda.synchronize()
                marks.append((_n, t0, time.perf_counter()))
                return r
            self._w[n] = g
        return self._w[n]


rh.lib = L(lib)
t0 = time.perf_counter()
enc.timings = {}
enc.encode(rgb, specs)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("frame wall (with syncs) ms", round(1e3 * (t1 - t0), 1), {k: round(v * 1e3, 1) for k, v in enc.timings.items()})
agg, cnt = {}, {}
last, gaps = t0, []
for n, a, b in marks:
    agg[n] = agg.get(n, 0.0) + (b - a)
    cnt[n] = cnt.get(n, 0) + 1
    gaps.append((a - last, n))
    last = b
for n, v in sorted(agg.items(), key=lambda x: -x[1])[:14]:
    print(f"{n:28s} {1e3*v:8.2f} ms  calls {cnt[n]}")
print("host time between C calls ms", round(1e3 * (sum(g for g, _ in gaps) + t1 - last), 1))
print("largest host gaps (ms, before call):", [(round(1e3 * g, 2), n) for g, n in sorted(gaps, reverse=True)[:12]])
