"""Drop-in for the reference's encoder/ROI/roi.py.  The region extraction -- extract_roi_nonroi, extract_regions,
extract_connected_regions(_fast), fuse_adjacent_regions_optimized, process_regions_with_reassignment -- runs on the MI355X
(roibasedimagecompression_amd.api.roi; connected components PARITY UNPINNED in their numbering, see there).  The heuristics
that produce the region map (get_regions, process_and_unify_borders, ...: OpenCV filters and morphology) come from the
reference's own file when its checkout sits behind this repository on sys.path and OpenCV is installed, otherwise they are
placeholders that raise."""
from roibasedimagecompression_amd._shim import downstream_getattr
from roibasedimagecompression_amd.api.roi import (extract_connected_regions, extract_connected_regions_fast,  # noqa: F401
                                                  extract_regions, extract_roi_nonroi, fuse_adjacent_regions_optimized,
                                                  process_regions_with_reassignment)

__getattr__ = downstream_getattr(__name__, __file__, (
    "get_regions", "remove_small_noise_regions", "detect_meaningful_borders", "protect_border_regions", "fill_closed_regions",
    "visualize_roi_nonroi_comparison", "process_and_unify_borders", "directional_region_unification", "plot_regions"))
