"""Drop-in for the reference's encoder/ROI/roi.py: the ROI stage on the MI355X -- get_regions and the chain behind it
(process_and_unify_borders, directional_region_unification, detect_meaningful_borders, protect_border_regions,
fill_closed_regions, remove_small_noise_regions: roibasedimagecompression_amd.api.roi_chain) and the region extraction
(extract_roi_nonroi, extract_regions, extract_connected_regions(_fast), fuse_adjacent_regions_optimized,
process_regions_with_reassignment: roibasedimagecompression_amd.api.roi).  PARITY UNPINNED wherever OpenCV is involved (restated
from its published algorithms; no OpenCV in the build container).  The matplotlib helpers and unused variants come from the
reference's own file when it is importable, otherwise they are placeholders that raise."""
from roibasedimagecompression_amd._shim import downstream_getattr
from roibasedimagecompression_amd.api.roi import (extract_connected_regions, extract_connected_regions_fast,  # noqa: F401
                                                  extract_regions, extract_roi_nonroi, fuse_adjacent_regions_optimized,
                                                  process_regions_with_reassignment)
from roibasedimagecompression_amd.api.roi_chain import (detect_meaningful_borders, directional_region_unification,  # noqa: F401
                                                        fill_closed_regions, get_regions, process_and_unify_borders,
                                                        protect_border_regions, remove_small_components_density_aware,
                                                        remove_small_components_density_aware_fast, remove_small_noise_regions)

__getattr__ = downstream_getattr(__name__, __file__, (
    "visualize_roi_nonroi_comparison", "plot_regions", "extract_connected_regions_with_tight_bbox", "extract_roi_nonroi_alt"))
