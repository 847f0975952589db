"""one-off: get_regions + extract_regions on the device vs the numpy / scipy restatement on the reference's own PNGs (tests/golden/*.png)"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
from oracle import rhccq_oracle as O
from roibasedimagecompression_amd.api import roi_chain as C, roi as R
bad = 0
for png in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "*.png"))):
    img = np.asarray(Image.open(png).convert("RGB"), dtype=np.uint8)
    t0 = time.time()
    got = C.get_regions(img)
    t1 = time.time()
    want = O.get_regions(img)
    ok = all(np.array_equal(g, w) for g, w in zip(got, want))
    a, b = R.extract_regions(img, got[4], got[5]), O.extract_regions(img, want[4], want[5])
    ok = ok and [(r["area"], tuple(r["bbox"])) for r in a[0] + a[1]] == [(r["area"], tuple(r["bbox"])) for r in b[0] + b[1]]
    bad += not ok
    print(os.path.basename(png), img.shape, "identical" if ok else "MISMATCH", "ROI %.3f" % got[1].mean(), "regions", len(a[0]), len(a[1]),
          "device %.0f ms, restatement %.1f s" % ((t1 - t0) * 1e3, time.time() - t1), flush=True)
print("bad", bad)
