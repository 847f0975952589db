"""Mirror of encoder/compression/subregions.py: subregion_quantization (subregions.py:90-683), level 1 of
the hierarchy.  The per-region split score and SLIC segmentation are UPSTREAM of the replaced path
(encoder/subregions/*, scikit-image); they are imported by the same names as in the reference
(subregions.py:4-5), so a deployment keeps the reference's own upstream modules.  Everything after
the label map (crop, black fix, unique colours, clustering, merge per region) runs on the MI355X."""
import logging
import math

import numpy as np

from ..frame import ClassSpec, FrameEncoder
from ..ops import default_context, unpack_rgb
from ..segment import IndexList
from .slic import segment_dropped_by_find_contours

log = logging.getLogger("rhccq")


def _reference_segmenter(bbox_region, bbox_mask):
    """split score -> number of segments -> masked SLIC, exactly as subregions.py:125-161 chains them."""
    from encoder.subregions.split_score import calculate_split_score, normalize_result
    from encoder.subregions.slic import enhanced_slic_with_texture
    overall, _, _ = calculate_split_score(bbox_region, bbox_mask)
    window = math.ceil(math.ceil(math.log(bbox_region.size, 10)) * math.log(bbox_region.size))
    optimal = math.ceil(normalize_result(overall, window))
    if optimal <= 0:
        optimal = 1
    segments, _ = enhanced_slic_with_texture(bbox_region, bbox_mask, n_segments=optimal)
    return segments


def _comp_to_dict(enc, S, comp, quality):
    idx = enc.render_component(S, comp).reshape(-1)
    pal = unpack_rgb(comp.keys)
    n = len(pal)
    h, w = comp.shape
    if comp.merged:
        dt = np.uint8 if n <= 256 else (np.uint16 if n <= 65536 else np.uint32)
        return {"top_left": tuple(comp.top_left), "shape": (h, w), "palette": [tuple(int(v) for v in r) for r in pal],
                "indices": IndexList(dev=idx), "indices_dtype": str(dt), "method": "merged", "actual_colors": n,
                "encoding": "roi_merged"}
    size = n * 3 + h * w * (1 if n <= 256 else 2) + 100
    return {"method": "clustered_colors", "top_left": tuple(comp.top_left), "shape": (h, w), "palette": pal.tolist(),
            "indices": IndexList(dev=idx), "compressed_colors": n, "index_dtype": "uint8" if n <= 256 else "uint16",
            "original_size": h * w * 3, "compressed_size": size, "compression_ratio": h * w * 3 / size, "mse": 0.0,
            "psnr": float("inf"), "encoding": "dbscan_clustered", "black_preserved": True, "parallel_processed": True,
            "clustering_params": {"quality": quality, "min_samples": 1}}


last_stats = {}        # diagnostics of the most recent call: regions, SLIC segments encoded / dropped (tools/notebook_flow.py reads it)


def subregion_quantization(image_rgb, subregions, quality=10, subregion_type=None, debug=False, segmenter=None):
    """Returns, per region, the list the reference appends (subregions.py:634-679): [merged dict] when the
    region has several segments, [component dict] for one, [] for none.
    `segmenter(bbox_region, bbox_mask) -> int32 label map (0 = background)` is an extension hook; the
    default calls the reference's upstream split-score + SLIC modules by their reference names."""
    import torch
    rh = default_context()
    image_rgb = np.ascontiguousarray(image_rgb, dtype=np.uint8)
    H, W = image_rgb.shape[:2]
    seg_fn = segmenter or _reference_segmenter
    # one label map per LAYER: regions of one call may overlap (extract_regions appends the small ROI components, buffer zone
    # included, to the non-ROI list, roi.py:76-84); the reference treats every region on its own, so an overlapping region
    # goes to the first layer in which its pixels are still free
    layers = []
    nxt = dropped = 0
    for ri, region in enumerate(subregions):
        minr, minc, maxr, maxc = (int(v) for v in region["bbox"])
        mask = np.asarray(region["bbox_mask"], dtype=bool)
        seg = np.asarray(seg_fn(image_rgb[minr:maxr, minc:maxc], mask))
        layer = next((l for l in layers if not l["labels"][minr:maxr, minc:maxc][mask].any()), None)
        if layer is None:
            layer = {"labels": np.zeros((H, W), np.int32), "seg_region": [], "bboxes": [], "regions": [], "n": 0}
            layers.append(layer)
        local = len(layer["bboxes"])
        layer["bboxes"].append((minr, minc, maxr, maxc))
        layer["regions"].append(ri)
        view = layer["labels"][minr:maxr, minc:maxc]
        for sid in np.unique(seg):                       # ascending ids, 0 = background (slic.py:158-160)
            if sid == 0:
                continue
            m = (seg == sid) & mask
            if not m.any():
                continue
            if segment_dropped_by_find_contours(m):                     # (slic.py:188-193: a segment that fills its box yields no contour)
                dropped += 1
                continue
            nxt += 1
            layer["n"] += 1
            view[m] = layer["n"]
            layer["seg_region"].append(local)
    last_stats.update(regions=len(subregions), segments=nxt, segments_dropped=dropped, layers=len([l for l in layers if l["n"]]))
    if nxt == 0:
        return [[] for _ in subregions]
    layers = [l for l in layers if l["n"]]
    specs = [ClassSpec(torch.from_numpy(l["labels"]).to(rh.device), l["seg_region"], l["bboxes"], quality) for l in layers]
    enc = FrameEncoder(rh)
    S = enc.prepare(torch.from_numpy(np.array(image_rgb, dtype=np.uint8, order="C")).to(rh.device), specs)
    out = [[] for _ in subregions]
    for l, regs in zip(layers, enc.level1(S)):
        for ri, comp in zip(l["regions"], regs):
            out[ri] = [] if comp is None else [_comp_to_dict(enc, S, comp, quality)]
    if debug:
        log.info("%s: %d regions, %d segments", subregion_type or "subregions", len(subregions), nxt)
    return out
