import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""diagnostic only: per-phase cycle shares of mbk_init_kernel (build with -DRHCCQ_STAMPS into dbg_build/)."""
import ctypes, sys, time, math
import numpy as np
import torch
from roibasedimagecompression_amd import _lib
_lib.LIB_PATH = "dbg_build/librhccq_dbg.so"
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd import synth
rh = Rhccq(0)
img = synth.photo(2160, 3840, 1234)
keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
pal = np.unique(keys[:, 1920:])          # ~1.5M colours like a non-ROI segment
pal = pal[pal != 0]
k = math.ceil(len(pal) * 0.2 / 10)
print("N", len(pal), "k", k)
t0 = time.perf_counter()
labs, info = rh.minibatch_kmeans([pal], [k], return_info=True)
torch.cuda.synchronize()
print("total s", time.perf_counter() - t0, "steps", info["state"][0][5])
out = (ctypes.c_ulonglong * 16)()
rh.lib.rhccq_debug_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh.lib.rhccq_debug_stamps(out))
v = np.array(list(out), dtype=np.float64)
names = ["search", "enumerate", "barrier1", "evaluate+b2", "commit+b3"]
tot = v[:5].sum()
print("items/pick", v[5] / (k - 1), "winner blocks/pick", v[6] / (k - 1), "picks with > 128 items", v[7] / (k - 1))
for n, x in zip(names, v[:5]):
    print(f"{n:18s} {x/ (k-1):10.0f} cycles/step  {100*x/tot:5.1f}%")
print("total cycles/step", tot / (k - 1))
print("sub-stamps (wave 0):", {i: round(v[i] / (k - 1)) for i in range(8, 16)})

out2 = (ctypes.c_ulonglong * 16)()
rh.lib.rhccq_debug_upd_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh.lib.rhccq_debug_upd_stamps(out2))
w = np.array(list(out2), dtype=np.float64)
calls = max(w[15], 1)
print("update kernel (problem 0, thread 0): calls", int(w[15]), "with reassignment", int(w[14]))
for n, x in zip(["tile arg-min + labels", "inertia tree", "hash member sums", "apply centres", "reassignment", "EWA / state"], w[:6]):
    print(f"{n:24s} {x / calls:10.0f} cycles/step")
print("total", w[:6].sum() / calls)
