"""Drop-in for the reference's encoder/ROI/small_regions.py: remove_small_regions and connect_by_closing_fast (the two the pipeline
calls) run on the MI355X (roibasedimagecompression_amd.api.roi_chain; parity unpinned, OpenCV restated); the unused variants
(connect_nearby_pixels, the Voronoi / skeleton / region-growing connectors) come from the reference's own file when its checkout
sits behind this repository on sys.path and OpenCV is installed, otherwise they are placeholders that raise."""
from roibasedimagecompression_amd._shim import downstream_getattr
from roibasedimagecompression_amd.api.roi_chain import connect_by_closing_fast, remove_small_regions  # noqa: F401

__getattr__ = downstream_getattr(__name__, __file__, (
    "connect_nearby_pixels", "connect_by_dilation", "connect_by_voronoi", "is_polygon_connecting", "connect_by_skeleton",
    "connect_by_skeleton_fast", "connect_by_region_growing", "grow_region"))
