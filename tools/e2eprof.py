"""diagnostic only: cProfile of the 4K script flow (tools/e2e4k.py), second run"""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import notebook_flow
from roibasedimagecompression_amd import synth
H, W = 2160, 3840
img = synth.photo(H, W, 1234)
yy, xx = np.mgrid[0:H, 0:W]
img[((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 > 0.5] //= 3
notebook_flow.notebook_flow(img, out_path="gpurun_out/e2e4k.rhccq")
pr = cProfile.Profile(); pr.enable()
notebook_flow.notebook_flow(img, out_path="gpurun_out/e2e4k.rhccq")
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
