"""diagnostic only: cycles per mini-batch step of the three update roles on the straggler problem of the bench frame
(build with -DRHCCQ_STAMPS into dbg_build/)"""
import os, sys, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from roibasedimagecompression_amd import _lib
_lib.LIB_PATH = "dbg_build/librhccq_dbg.so"
import bench
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder
rh = Rhccq(0)
enc = FrameEncoder(rh)
_, rgb, specs, roi, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
S = enc.prepare(rgb, specs)
jobs, _ = enc.level1_jobs(S)
parts = [jb["keys_dev"][1:] if jb["has_black"] else jb["keys_dev"] for jb in jobs]
ks = [math.ceil(int(p.numel()) * 0.2 / 10) for p in parts]
labs, info = rh.minibatch_kmeans([parts[1]], [ks[1]], return_info=True, return_device=True)
print("steps", int(info["state"][0][5]))
out2 = (ctypes.c_ulonglong * 16)()
rh._raw.rhccq_debug_upd_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_upd_stamps(out2))
u = np.array(list(out2), dtype=np.float64)
calls = max(u[15], 1)
print("update kernel: steps", int(calls), "reassign steps", int(u[14]), "cycles per step: role0", round(u[8] / calls), "role1", round(u[9] / calls),
      "role2", round(u[10] / calls), "| role 0 phases:", [round(x / calls) for x in u[:6]])
