"""Mirror of encoder/compression/clustering.py (reference) over the HIP kernels.

  get_all_unique_colors            clustering.py:4-103    -> K1 (bitmap unique + rank)
  compute_clustering_params        clustering.py:108-135  -> rhccq_params (host)
  cluster_palette_colors_parallel  clustering.py:160-437  -> K3/K4 | K8, K7, K2, K6
  split_large_cluster              clustering.py:720-775  -> K7
Unlike the reference nothing is printed; `logging` (logger "rhccq") carries the messages."""
import logging

import numpy as np

from ..ops import clustering_params as _params, default_context, pack_rgb, unpack_rgb
from ..palette import cluster_palette as _cluster_palette
from ..segment import IndexList, as_index_array

log = logging.getLogger("rhccq")


def get_all_unique_colors(region_image, top_left_coords):
    if region_image is None or region_image.size == 0:
        return None                                               # clustering.py:9-10
    import torch
    rh = default_context()
    h, w, _ = region_image.shape
    img = np.ascontiguousarray(region_image, dtype=np.uint8)
    keys, idx = rh.unique_colors(torch.from_numpy(img).to(rh.device))
    palette = unpack_rgb(keys.cpu().numpy()).tolist()
    actual = len(palette)
    total = h * w
    index_dtype = np.uint8 if actual <= 256 else np.uint16
    compressed_size = actual * 3 + total * (1 if actual <= 256 else 2) + 50
    return {
        "method": "exact_colors", "top_left": top_left_coords, "shape": (h, w), "palette": palette,
        "indices": IndexList(dev=idx), "max_colors": actual, "actual_colors": actual, "index_dtype": str(index_dtype),
        "original_size": total * 3, "compressed_size": compressed_size,
        "compression_ratio": total * 3 / compressed_size if compressed_size > 0 else 0,
        "mse": 0.0, "psnr": float("inf"), "encoding": "exact",
    }


def compute_clustering_params(n_colors, quality, color_space="rgb"):
    """color_space is accepted and ignored, as in the reference (clustering.py:108-135)."""
    return _params(n_colors, quality)


def cluster_palette_colors_parallel(quality, compressed_data, eps=10.0, min_samples=2, max_colors_per_cluster=5, num_workers=None):
    if min_samples < 1:
        raise ValueError("min_samples must be >= 1")              # (sklearn's own parameter check)
    rh = default_context()
    palette = np.array(compressed_data["palette"], dtype=np.uint8).reshape(-1, 3)
    h, w = compressed_data["shape"]
    keys = pack_rgb(palette)
    if not np.any(keys != 0):
        return compressed_data                                    # "only black": returned unchanged (clustering.py:197-199)
    new_keys, mapping, info = _cluster_palette(rh, quality, keys, eps, max_colors_per_cluster, min_samples)
    import torch
    ind = compressed_data["indices"]
    d_idx = ind.device_tensor(rh) if isinstance(ind, IndexList) else torch.from_numpy(
        np.ascontiguousarray(as_index_array(ind).astype(np.int32))).to(rh.device)
    new_idx = rh.remap(d_idx, torch.from_numpy(mapping.astype(np.int32)).to(rh.device))
    new_palette = unpack_rgb(new_keys)
    n_new = len(new_palette)
    total = h * w
    original_size = compressed_data.get("original_size", total * 3)
    size = n_new * 3 + total * (1 if n_new <= 256 else 2) + 100
    log.info("clustered %d -> %d colours (%s)", len(palette), n_new, info.get("branch"))
    return {
        "method": "clustered_colors", "top_left": compressed_data["top_left"], "shape": (h, w),
        "palette": new_palette.tolist(), "indices": IndexList(dev=new_idx), "original_unique_colors": len(palette),
        "compressed_colors": n_new, "index_dtype": "uint8" if n_new <= 256 else "uint16", "original_size": original_size,
        "compressed_size": size, "compression_ratio": original_size / size if size > 0 else 0, "mse": 0.0, "psnr": float("inf"),
        "clustering_params": {"eps": eps, "min_samples": min_samples, "max_colors_per_cluster": max_colors_per_cluster},
        "encoding": "dbscan_clustered", "black_preserved": True, "parallel_processed": True,
    }


def split_large_cluster(cluster_colors, max_colors_per_cluster):
    """list of sub-cluster colour arrays, depth-first in child-label order (clustering.py:720-775)."""
    from ..palette import _n_splits
    rh = default_context()
    cols = np.asarray(cluster_colors).reshape(-1, 3)
    k = _n_splits(len(cols), max_colors_per_cluster)
    if k == 0:
        return [cols]
    lab = rh.kmeans_split([pack_rgb(cols)], [k])[0]
    out = []
    for i in range(k):
        sub = cols[lab == i]
        if len(sub) == 0:
            continue
        out.extend(split_large_cluster(sub, max_colors_per_cluster) if len(sub) > max_colors_per_cluster else [sub])
    return out


def find_color_index(palette, color):
    m = np.all(np.asarray(palette) == np.asarray(color), axis=1)
    return int(np.where(m)[0][0]) if m.any() else None
