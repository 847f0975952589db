"""Drop-in for the reference's encoder/compression/clustering.py: same names, MI355X implementation in
roibasedimagecompression_amd.api.clustering."""
from roibasedimagecompression_amd.api.clustering import *  # noqa: F401,F403
from roibasedimagecompression_amd.api import clustering as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

# names of the reference's module that the pipeline never calls (unused variants, debug helpers): taken from the reference's own file
# when its checkout sits behind this repository on sys.path, placeholders that raise otherwise
from roibasedimagecompression_amd._shim import downstream_getattr  # noqa: E402

__getattr__ = downstream_getattr(__name__, __file__, ('process_clusters_parallel', 'split_cluster_worker', 'cluster_palette_colors', 'split_by_luminance', 'hierarchical_color_clustering', 'create_clustered_result'))
