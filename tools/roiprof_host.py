"""diagnostic: host-side profile (cProfile) of the resident ROI stage on a 4K frame"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roibasedimagecompression_amd import synth
from roibasedimagecompression_amd.api import roi_chain as C, roi as R
img = synth.photo(2160, 3840, 1234)
for _ in range(2):
    out = C.get_regions(img)
    R.extract_regions(img, out[4], out[5])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = C.get_regions(img)
t1 = time.perf_counter()
roi, non = R.extract_regions(img, out[4], out[5])
pr.disable()
t2 = time.perf_counter()
print("get_regions", round((t1 - t0) * 1e3, 1), "extract", round((t2 - t1) * 1e3, 1))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
