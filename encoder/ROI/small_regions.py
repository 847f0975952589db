"""Placeholder for a stage UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).  When the reference
checkout sits behind this repository on sys.path its own module of this name is executed instead (roibasedimagecompression_amd/_shim.py);
otherwise the names exist so that the reference's import lines resolve and raise when called."""
from roibasedimagecompression_amd._shim import defer_to_downstream, upstream

if defer_to_downstream(__name__, __file__) is None:
    remove_small_regions = upstream("remove_small_regions")
    connect_nearby_pixels = upstream("connect_nearby_pixels")
    connect_by_closing_fast = upstream("connect_by_closing_fast")
