"""Placeholder for a stage UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).  When the reference
checkout sits behind this repository on sys.path its own module of this name is executed instead (roibasedimagecompression_amd/_shim.py);
otherwise the names exist so that the reference's import lines resolve and raise when called."""
from roibasedimagecompression_amd._shim import defer_to_downstream, upstream

if defer_to_downstream(__name__, __file__) is None:
    get_regions = upstream("get_regions")
    extract_regions = upstream("extract_regions")
    remove_small_noise_regions = upstream("remove_small_noise_regions")
    detect_meaningful_borders = upstream("detect_meaningful_borders")
    protect_border_regions = upstream("protect_border_regions")
    fill_closed_regions = upstream("fill_closed_regions")
    extract_roi_nonroi = upstream("extract_roi_nonroi")
    visualize_roi_nonroi_comparison = upstream("visualize_roi_nonroi_comparison")
    process_and_unify_borders = upstream("process_and_unify_borders")
    directional_region_unification = upstream("directional_region_unification")
    extract_connected_regions_fast = upstream("extract_connected_regions_fast")
