"""diagnostic only: the ROI stage (get_regions + extract_regions) on a 4K synthetic photo, wall time per part (second run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from roibasedimagecompression_amd import synth
from roibasedimagecompression_amd.api import edges as E, roi_chain as C, roi as R

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
img = synth.photo(H, W, 1234)
for rep in range(2):
    t = {}
    t0 = time.perf_counter()
    edge_map = E.get_edge_map(img)
    torch.cuda.synchronize(); t["get_edge_map (21 Canny passes)"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    density = E.compute_local_density(edge_map, 3)
    thr = E.suggest_automatic_threshold(density, edge_map) / 100
    t["density + threshold (host numpy on 4K float maps)"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = C.process_and_unify_borders(edge_map, density, img, density_threshold=thr)
    torch.cuda.synchronize(); t["clean-up chain + buffer zone"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    roi, non = R.extract_regions(img, out[4], out[5])
    t["extract_regions"] = time.perf_counter() - t0
print({k: round(v * 1e3, 1) for k, v in t.items()}, "ms; ROI fraction", round(float(out[1].mean()), 3), "regions", len(roi), len(non),
      "edge pixels", round(float((edge_map > 0).mean()), 3), "total ms", round(sum(t.values()) * 1e3, 1), "Mpx/s", round(H * W / sum(t.values()) / 1e6, 1))
for rep in range(2):
    t0 = time.perf_counter()
    out = C.get_regions(img)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    roi, non = R.extract_regions(img, out[4], out[5])
    t2 = time.perf_counter()
print("get_regions (resident chain, six numpy arrays back):", round((t1 - t0) * 1e3, 1), "ms; extract_regions:", round((t2 - t1) * 1e3, 1), "ms;",
      round(H * W / (t2 - t0) / 1e6, 1), "Mpx/s for the whole ROI stage")
