"""diagnostic only: what follows level 1's clustering on a single 4K frame (merge per region, level 2, level 3, final remap):
wall time per stage, time inside the C calls (synchronised around each) and host time between them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
from roibasedimagecompression_amd.palette import cluster_palettes

rh = Rhccq(0)
enc = FrameEncoder(rh)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
for _ in range(2):
    enc.encode(rgb, specs)
torch.cuda.synchronize()
lib = rh.lib
marks = []


def wrap(name):
    f = getattr(lib, name)

    def g(*a):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a)
        torch.cuda.synchronize()
        marks.append((name, t0, time.perf_counter()))
        return r
    return g


class L:
    def __init__(self, lib):
        self._lib, self._w = lib, {}

    def __getattr__(self, n):
        if n.startswith("rhccq_") and n not in ("rhccq_last_error",):
            if n not in self._w:
                self._w[n] = wrap(n)
            return self._w[n]
        return getattr(self._lib, n)


S = enc.prepare(rgb, specs)
jobs, job_ids = enc.level1_jobs(S)
res = cluster_palettes(rh, jobs)
torch.cuda.synchronize()
rh.lib = L(lib)
for ln in getattr(rh, "_lanes", {}).values():
    ln[1].lib = L(ln[1].lib)


def stage(name, fn):
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    agg, last, gaps = {}, t0, 0.0
    for n, a, b in sorted(marks, key=lambda m: m[1]):
        agg[n] = agg.get(n, [0.0, 0])
        agg[n][0] += b - a
        agg[n][1] += 1
        gaps += max(0.0, a - last)
        last = max(last, b)
    gaps += t1 - last
    print(f"== {name}: wall {1e3*(t1-t0):.2f} ms, host between C calls {1e3*gaps:.2f} ms")
    for n, (v, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
        print(f"     {n:30s} {1e3*v:8.3f} ms  calls {c}")
    return out


per_class = stage("level1_finish (lut1, first positions, merge per region)", lambda: enc.level1_finish(S, job_ids, res))
lvl2, q2s, jobs2 = stage("level2_jobs (merge per class)", lambda: enc.level2_jobs(S, per_class))
print("level-2 palettes:", [len(j["keys"]) for j in jobs2])
res2 = stage("level2 cluster", lambda: cluster_palettes(rh, jobs2))
comps3 = stage("level2_finish", lambda: enc.level2_finish(lvl2, res2))
m3c, q3, job3 = stage("level3_job (merge)", lambda: enc.level3_job(S, comps3, q2s))
print("level-3 palette:", len(job3["keys"]), "q3", q3)
(res3,) = stage("level3 cluster", lambda: cluster_palettes(rh, [job3]))
stage("finish (compose + remap)", lambda: enc.finish(S, comps3, m3c, q3, res3))
