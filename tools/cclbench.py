"""diagnostic only: connected components (csrc/ccl.hip) and the buffer-zone split at 4K, HIP-event time per call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from roibasedimagecompression_amd import synth
from roibasedimagecompression_amd.ops import Rhccq

rh = Rhccq(0)
H, W = 2160, 3840
img = synth.photo(H, W, 77)
g = img[..., 1].astype(np.int32)
edge = (np.abs(np.diff(g, axis=1, prepend=g[:, :1])) + np.abs(np.diff(g, axis=0, prepend=g[:1])) > 9)
yy, xx = np.mgrid[0:H, 0:W]
blob = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 <= 0.45
rgb = torch.from_numpy(img).to(rh.device)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, m in (("edge map", edge), ("one blob", blob), ("all set", np.ones((H, W), bool)), ("noise 50%", np.random.default_rng(1).random((H, W)) < 0.5)):
    t = torch.from_numpy(np.ascontiguousarray(m)).to(rh.device)
    n, lab, st = rh.ccl(t, 8, cap=1 << 19)
    ms = timed(lambda: rh.ccl(t, 8, cap=1 << 19))
    print(f"ccl 8-conn {name:10s}: {n:8d} components, {ms:7.3f} ms per call (incl. count + statistics read-back), {H*W/ms/1e3:8.1f} Mpx/s")
rm = torch.from_numpy(blob.astype(np.uint8)).to(rh.device)
ms = timed(lambda: rh.roi_buffer(rm, rgb, 3))
print(f"roi_buffer R=3: {ms:.3f} ms, {12*H*W/ms/1e6:.1f} GB/s over 12 B/px")
