"""Drop-in for the reference's encoder/ROI/small_gaps.py: bridge_small_gaps_fast (the one the pipeline calls) runs on the MI355X
(roibasedimagecompression_amd.api.roi_chain; parity unpinned); the per-pixel Python variants (bridge_small_gaps and its helpers)
come from the reference's own file when it is importable, otherwise they are placeholders that raise."""
from roibasedimagecompression_amd._shim import downstream_getattr
from roibasedimagecompression_amd.api.roi_chain import bridge_small_gaps_fast  # noqa: F401

__getattr__ = downstream_getattr(__name__, __file__, (
    "bridge_small_gaps", "find_internal_gaps_strict_2d", "find_internal_gaps_relaxed_2d", "find_internal_gaps_density_aware_2d",
    "is_locally_surrounded_2d", "has_white_in_opposite_directions_2d", "is_internal_gap_2d", "is_pixel_on_edge",
    "create_gap_detection_kernels"))
