"""Drop-in for the reference's encoder/ROI/edges.py: the whole module -- get_edge_map (20 adaptive threshold pairs scored on their
Canny edge maps + the final Canny on the colour image), compute_local_density, suggest_automatic_threshold and the helpers -- runs
on the MI355X (roibasedimagecompression_amd.api.edges, csrc/edges.hip + csrc/ccl.hip).  PARITY UNPINNED: OpenCV's cvtColor / Sobel /
Otsu / Canny are restated from their published integer implementations (no OpenCV in the build container)."""
from roibasedimagecompression_amd.api.edges import (compute_adaptive_canny_thresholds, compute_fast_canny_thresholds,  # noqa: F401
                                                    compute_local_density, evaluate_edge_quality, find_best_edges_by_quality,
                                                    get_edge_map, get_edge_map_fast, suggest_automatic_threshold)
