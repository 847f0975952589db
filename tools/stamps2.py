import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""diagnostic only: per-phase cycle shares of mbk_init2_kernel (build with -DRHCCQ_STAMPS into dbg_build/)."""
import ctypes, time, math
import numpy as np
import torch
from roibasedimagecompression_amd import _lib
_lib.LIB_PATH = "dbg_build/librhccq_dbg.so"
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd import synth
rh = Rhccq(0)
img = synth.photo(2160, 3840, 1234)
keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
pal = np.unique(keys[:, 1920:])
pal = pal[pal != 0]
if len(sys.argv) > 1 and sys.argv[1] == "bench":     # the longest chain of the bench frame: k = 30 128, 90 384 init samples
    pal = pal[:1506367]
if len(sys.argv) > 1 and sys.argv[1] == "roi":       # shaped like the bench frame's straggler: k = 20 556
    pal = pal[:1027790]
k = math.ceil(len(pal) * 0.2 / 10)
print("N", len(pal), "k", k)
t = {}
labs, info = rh.minibatch_kmeans([pal], [k], return_info=True, timing=t)
print("init ms", t["init_ms"], "us/pick", t["init_ms"] * 1e3 / k, "steps", int(info["state"][0][5]), "overlapped launches", info["overlapped_launches"])
out = (ctypes.c_ulonglong * 16)()
rh._raw.rhccq_debug_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_stamps(out))
v = np.array(list(out), dtype=np.float64) / (k - 1)
names = ["search(wave0)", "barrier A", "enumerate", "barrier B", "evaluate", "barrier C", "argmax+commit", "barrier D"]
tot = v[:8].sum()
for n, x in zip(names, v[:8]):
    print(f"{n:18s} {x:10.0f} cycles/pick  {100*x/tot:5.1f}%")
print("total cycles/pick (thread 0)", tot)
print("items/pick", v[10], "picks not kept", v[11], "picks overflow", v[12], "gen 3, candidate 0: hit supers/pick", v[13], "leaf batches/pick", v[14])
out = (ctypes.c_ulonglong * 128)()
rh._raw.rhccq_debug_wave_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_wave_stamps(out))
w = np.array(list(out), dtype=np.float64).reshape(8, 16) / (k - 1)
for ph, name in enumerate(["search+E1", "E2", "evaluate", "commit", "g3:search", "g3:supers", "g3:list", "-"]):
    print(f"{name:10s} per wave:", " ".join(f"{x:5.0f}" for x in w[ph]))
out2 = (ctypes.c_ulonglong * 16)()
rh._raw.rhccq_debug_upd_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_upd_stamps(out2))
u = np.array(list(out2), dtype=np.float64)
calls = max(u[15], 1)
print("update kernel, problem 0: steps", int(calls), "cycles per step: role0", round(u[8] / calls), "role1", round(u[9] / calls), "role2", round(u[10] / calls),
      "| role 0 phases:", [round(x / calls) for x in u[:6]])
re = max(u[14], 1)
print("reassigning steps", int(u[14]), "cycles per such step: weights->LDS", round(u[6] / re), "max/hist/select sweeps", round(u[7] / re), "sweep 2", round(u[11] / re),
      "choice() replay", round(u[12] / re), "sweep 3", round(u[13] / re), "| whole reassignment phase (stamp 4)", round(u[4] / re), "draw behind it (stamp 5)", round(u[5] / re))
out3 = (ctypes.c_ulonglong * 8)()
rh._raw.rhccq_debug_pipe_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_pipe_stamps(out3))
q = np.array(list(out3), dtype=np.float64)
n = max(q[3], 1)
print("overlapped launches", int(q[3]), "cycles per launch: update role", round(q[0] / n), "draw role", round(q[1] / n), "inertia role", round(q[2] / n))
out4 = (ctypes.c_ulonglong * 8)()
rh._raw.rhccq_debug_spec_stamps.argtypes = [ctypes.c_void_p]
print("rc", rh._raw.rhccq_debug_spec_stamps(out4))
q = np.array(list(out4), dtype=np.float64)
n = max(q[7], 1)
print("speculative E-step workgroups", int(q[7]), "cycles per workgroup: tile load", round(q[0] / n), "exclude", round(q[1] / n), "loop", round(q[2] / n), "merge + store", round(q[3] / n))
