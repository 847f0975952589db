"""one-off soak: get_regions + extract_regions on the device vs the numpy / scipy restatement over random synthetic images:
python tests/soak/roisoak.py FIRST COUNT [MIN_EDGE MAX_EDGE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import rhccq_oracle as O
from roibasedimagecompression_amd import synth
from roibasedimagecompression_amd.api import roi_chain as C, roi as R

first, count = int(sys.argv[1]), int(sys.argv[2])
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (60, 300)            # image edge range
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(lo, hi)), int(rng.integers(lo, hi + 60))
    kind = seed % 3
    img = synth.photo(h, w, seed, sigma=float(rng.choice([0.5, 1.0, 2.0, 3.0]))) if kind else synth.poster(h, w, seed)
    if kind == 2:                                             # a darker flat surround: ROI and non-ROI both present
        yy, xx = np.mgrid[0:h, 0:w]
        img[((yy - h / 2) / (h / 2)) ** 2 + ((xx - w / 2) / (w / 2)) ** 2 > 0.5] //= 4
    try:
        got, want = C.get_regions(img), O.get_regions(img)
        ok = all(np.array_equal(g, x) for g, x in zip(got, want))
        if ok:
            a, b = R.extract_regions(img, got[4], got[5]), O.extract_regions(img, want[4], want[5])
            ok = [r["area"] for r in a[0] + a[1]] == [r["area"] for r in b[0] + b[1]] and [tuple(r["bbox"]) for r in a[0] + a[1]] == [tuple(r["bbox"]) for r in b[0] + b[1]]
    except Exception as e:                                    # both sides must fail alike
        ok = False
        print("EXC seed", seed, type(e).__name__, str(e)[:120], flush=True)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, (h, w), kind, flush=True)
    if (seed - first) % 20 == 19:
        print("done", seed - first + 1, "bad", bad, "elapsed", round(time.time() - t0), flush=True)
print("total", count, "bad", bad)
