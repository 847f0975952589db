"""Region extraction of the ROI stage on the MI355X (reference encoder/ROI/roi.py):

  extract_roi_nonroi               roi.py:685-718   region map -> ROI / non-ROI masks with the 3-px buffer zone (csrc/ccl.hip roi_buffer_kernel)
  extract_connected_regions_fast   roi.py:285-360   cv2.connectedComponentsWithStats(connectivity=8) -> region dicts
  extract_connected_regions        roi.py:262-283   skimage.measure.label + regionprops -> region dicts
  extract_regions                  roi.py:45-103    both masks -> region lists, small ROI regions moved to the non-ROI list
  fuse_adjacent_regions_optimized  roi.py:214-259
  process_regions_with_reassignment roi.py:109-212

The labelling runs on the device (union-find over run starts, csrc/ccl.hip; `Rhccq.ccl`); what comes back to the host is
what the reference's callers read: numpy region dicts.  PARITY: extract_roi_nonroi is scipy + numpy in the reference, checked
against the same calls.  The component PARTITION is unambiguous; OpenCV's label NUMBERING (= order of the region lists) is
restated from its published algorithms, PARITY UNPINNED (OpenCV is absent from the build container).
The edge / density / morphology heuristics that produce the region map (get_regions, process_and_unify_borders ...) are not
part of this module: encoder/ROI/roi.py takes them from the reference's own file when that is importable."""
import logging
import math

import numpy as np

from ..ops import default_context

log = logging.getLogger("rhccq")

_LAZY = ("mask", "full_image", "coords")


class RegionDict(dict):
    """The region dict of roi.py:349-358.  'bbox', 'bbox_mask', 'bbox_image', 'area', 'label' are stored; the three
    full-frame members ('mask', 'full_image', 'coords') are built from the label map on first access (the reference
    materialises two full frames per region up front: 33 MB per region at 4K)."""

    def __init__(self, base, labels, image):
        super().__init__(base)
        self._labels, self._image = labels, image

    def __missing__(self, key):
        if key not in _LAZY:
            raise KeyError(key)
        single = self._labels == self["label"]
        if key == "mask":
            val = single
        elif key == "coords":
            val = np.column_stack(np.nonzero(single))               # raster order
        else:
            val = np.zeros_like(self._image)
            val[single] = self._image[single]
        self[key] = val
        return val

    def __contains__(self, key):
        return super().__contains__(key) or key in _LAZY

    def get(self, key, default=None):
        return self[key] if key in self else default


def connected_components_with_stats(mask, connectivity=8, numbering="opencv", rh=None):
    """cv2.connectedComponentsWithStats(mask, connectivity) -> (num_labels, labels int32[H,W], stats int32[num_labels,5]) as numpy
    (the centroids, which no caller in the reference reads, are left out)."""
    import torch
    rh = rh or default_context()
    m = np.ascontiguousarray(np.asarray(mask) != 0).view(np.uint8)
    n, labels, stats = rh.ccl(torch.from_numpy(m).to(rh.device), connectivity, numbering=numbering)
    return n + 1, rh.to_host(labels), stats


def _region_dicts(mask, original_image, numbering):
    num, labels, stats = connected_components_with_stats(mask, 8, numbering)
    out = []
    for lab in range(1, num):
        x, y, w, h, area = (int(v) for v in stats[lab])
        out.append(RegionDict({"bbox_image": original_image[y:y + h, x:x + w], "bbox_mask": labels[y:y + h, x:x + w] == lab,
                               "bbox": (y, x, y + h, x + w), "area": area, "label": lab}, labels, original_image))
    return out


def extract_connected_regions_fast(mask, original_image):
    """roi.py:285-360"""
    return _region_dicts(mask, np.asarray(original_image), "opencv")


def extract_connected_regions(mask, original_image):
    """roi.py:262-283 (skimage.measure.label + regionprops: 8-connectivity, numbered by first pixel in raster order)"""
    return _region_dicts(mask, np.asarray(original_image), "raster")


def min_region_size(image_rgb):
    """roi.py:47-49: from the SIZE of the array (H * W * 3)"""
    return math.ceil(image_rgb.size / math.pow(10, math.ceil(math.log(image_rgb.size, 10)) - 3))


def extract_regions(image_rgb, roi_mask, nonroi_mask):
    """roi.py:45-103"""
    image_rgb = np.asarray(image_rgb)
    mn = min_region_size(image_rgb)
    roi_regions = extract_connected_regions_fast(roi_mask, image_rgb)
    nonroi_regions = extract_connected_regions_fast(nonroi_mask, image_rgb)
    log.info("initial - ROI: %d regions, non-ROI: %d regions (minimum region size %d)", len(roi_regions), len(nonroi_regions), mn)
    small = [r for r in roi_regions if r["area"] < mn]
    if small:
        for r in small:
            r["type"] = "nonroi"
        nonroi_regions.extend(small)
        roi_regions = [r for r in roi_regions if r["area"] >= mn]
        log.info("after reassignment - ROI: %d regions, non-ROI: %d regions", len(roi_regions), len(nonroi_regions))
    return roi_regions, nonroi_regions


def fuse_adjacent_regions_optimized(regions, image_shape, region_type="roi"):
    """roi.py:214-259.  As in the reference the regions' FULL-FRAME masks are painted through their bounding boxes, so the
    masks must have the bounding boxes' shape where the two differ (the reference raises there too)."""
    if len(regions) <= 1:
        return regions
    combined = np.zeros(image_shape[:2], np.uint8)
    for i, region in enumerate(regions):
        y1, x1, y2, x2 = region["bbox"]
        combined[y1:y2, x1:x2] = np.where(region["mask"], i + 1, combined[y1:y2, x1:x2])
    num, labels, _ = connected_components_with_stats(combined, 8)
    if num - 1 == len(regions):
        return regions
    fused = []
    for lab in range(1, num):
        m = labels == lab
        rows, cols = np.any(m, axis=1), np.any(m, axis=0)
        if not rows.any() or not cols.any():
            continue
        y1, y2 = np.where(rows)[0][[0, -1]]
        x1, x2 = np.where(cols)[0][[0, -1]]
        crop = m[y1:y2 + 1, x1:x2 + 1]
        fused.append({"bbox": (y1, x1, y2, x2), "mask": crop, "area": np.sum(crop), "type": region_type, "is_fused": True})
    return fused


def process_regions_with_reassignment(image_rgb, roi_mask, nonroi_mask):
    """roi.py:109-212: small regions swap class (both ways), then adjacent regions of a class are fused"""
    image_rgb = np.asarray(image_rgb)
    size = image_rgb.shape[0] * image_rgb.shape[1]
    mn = math.ceil(size / math.pow(10, math.ceil(math.log(size, 10)) - 3))
    roi_regions = extract_connected_regions(roi_mask, image_rgb)
    nonroi_regions = extract_connected_regions(nonroi_mask, image_rgb)
    new_roi, new_non = [], []
    for r in roi_regions:
        big = r["area"] >= mn
        r["type"] = "roi" if big else "nonroi"
        (new_roi if big else new_non).append(r)
    for r in nonroi_regions:
        big = r["area"] >= mn
        r["type"] = "nonroi" if big else "roi"
        (new_non if big else new_roi).append(r)
    if len(new_roi) > 1:
        new_roi = fuse_adjacent_regions_optimized(new_roi, image_rgb.shape, "roi")
    if len(new_non) > 1:
        new_non = fuse_adjacent_regions_optimized(new_non, image_rgb.shape, "nonroi")
    return new_roi, new_non


def extract_roi_nonroi(original_image, region_map, buffer_size=3, rh=None, rgb_dev=None, region_dev=None):
    """roi.py:685-718 -> (roi_image, nonroi_image, roi_mask, nonroi_mask) as numpy arrays.  rgb_dev / region_dev: the image / a 0-1 region
    map already on the device (the resident chain hands them over instead of uploading the frame a second time)"""
    import torch
    rh = rh or default_context()
    original_image = np.ascontiguousarray(original_image, dtype=np.uint8)
    region_map = np.asarray(region_map)
    if original_image.ndim != 3 or original_image.shape[2] != 3:
        raise ValueError("extract_roi_nonroi: an H x W x 3 image is expected")
    if buffer_size < 1:
        # scipy.ndimage.binary_dilation(iterations < 1) repeats until nothing changes: every pixel, if the set is not empty
        roi_core, non_core = region_map == 1, region_map == 0
        buf = np.full(region_map.shape, bool(roi_core.any() and non_core.any()))
        roi_mask, non_mask = roi_core | buf, non_core | buf
        roi_image, non_image = original_image.copy(), original_image.copy()
        roi_image[~roi_mask] = 0
        non_image[~non_mask] = 0
        return roi_image, non_image, roi_mask, non_mask
    if region_dev is None:
        rm = np.where(region_map == 1, 1, np.where(region_map == 0, 0, 2)).astype(np.uint8)
        region_dev = torch.from_numpy(rm).to(rh.device)
    if rgb_dev is None:
        rgb_dev = torch.from_numpy(np.array(original_image, dtype=np.uint8, order="C")).to(rh.device)
    ri, ni, m1, m0 = rh.roi_buffer(region_dev, rgb_dev, buffer_size)
    return tuple(rh.to_host(ri, ni, m1, m0))
