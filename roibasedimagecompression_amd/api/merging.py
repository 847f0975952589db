"""Mirror of encoder/compression/merging.py: merge_region_components_simple (merging.py:8-120) on the
device (K5: first-position + paint kernels)."""
import numpy as np

from ..ops import default_context
from ..palette import merge_components
from ..segment import IndexList


def merge_region_components_simple(region_components, roi_bbox):
    if not region_components:
        return []
    if len(region_components) == 1:
        single = region_components[0].copy()
        if "actual_colors" not in single:
            single["actual_colors"] = len(single.get("palette", []))
        return [single]
    rh = default_context()
    comps = []
    for seg in region_components:
        c = {"top_left": seg["top_left"], "shape": seg["shape"], "palette": seg["palette"], "indices": seg["indices"]}
        if isinstance(seg["indices"], IndexList):
            c["indices_dev"] = seg["indices"].device_tensor(rh)
        comps.append(c)
    m = merge_components(rh, comps, roi_bbox)
    pal = [tuple(int(v) for v in row) for row in np.asarray(m["palette"]).reshape(-1, 3)]
    n = len(pal)
    dt = np.uint8 if n <= 256 else (np.uint16 if n <= 65536 else np.uint32)
    return [{
        "top_left": m["top_left"], "shape": m["shape"], "palette": pal, "indices": IndexList(dev=m["indices_dev"]),
        "indices_dtype": str(dt), "method": "merged", "actual_colors": n, "encoding": "roi_merged",
    }]


def visualize_merged_result(*args, **kwargs):
    raise NotImplementedError("plotting helper of the reference (merging.py:124-333); not part of the MI355X hot path")
