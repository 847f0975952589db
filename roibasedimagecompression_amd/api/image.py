"""Mirror of encoder/compression/image.py: quantize_image (image.py:243-286), level 3."""
from .clustering import cluster_palette_colors_parallel, compute_clustering_params
from .compression import optimize_compressed_dtype
from .merging import merge_region_components_simple


def quantize_image(image_components, original_image_height, original_image_width, quality=100):
    merged = merge_region_components_simple(list(image_components), roi_bbox=(0, 0, original_image_height, original_image_width))
    merged_segment = merged[0]
    eps, _, mc = compute_clustering_params(merged_segment["actual_colors"], quality, color_space="lab")
    seg = cluster_palette_colors_parallel(quality, merged_segment, eps=eps, min_samples=1, max_colors_per_cluster=mc)
    return optimize_compressed_dtype(seg)
