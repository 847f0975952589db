"""Placeholder for a stage UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).  When the reference
checkout sits behind this repository on sys.path its own module of this name is executed instead (roibasedimagecompression_amd/_shim.py);
otherwise the names exist so that the reference's import lines resolve and raise when called."""
from roibasedimagecompression_amd._shim import defer_to_downstream, upstream

if defer_to_downstream(__name__, __file__) is None:
    compute_local_density = upstream("compute_local_density")
    suggest_automatic_threshold = upstream("suggest_automatic_threshold")
    get_edge_map = upstream("get_edge_map")
    get_edge_map_fast = upstream("get_edge_map_fast")
    find_best_edges_by_quality = upstream("find_best_edges_by_quality")
