"""Mirror of encoder/compression/regions.py: region_quantization (regions.py:9-70), level 2."""
from .clustering import cluster_palette_colors_parallel, compute_clustering_params
from .merging import merge_region_components_simple


def region_quantization(regions_components, original_image_height, original_image_width, quality=50):
    flat = []
    for regions in regions_components:                   # regions.py:18-31: dicts or lists of dicts
        if isinstance(regions, dict):
            flat.append(regions)
        elif isinstance(regions, list):
            flat.extend(item for item in regions if isinstance(item, dict))
    merged = merge_region_components_simple(flat, roi_bbox=(0, 0, original_image_height, original_image_width))
    merged_segment = merged[0]                           # IndexError when empty, like regions.py:43
    eps, _, mc = compute_clustering_params(merged_segment["actual_colors"], quality, color_space="lab")
    return [cluster_palette_colors_parallel(quality, merged_segment, eps=eps, min_samples=1, max_colors_per_cluster=mc)]
