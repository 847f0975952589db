"""Clean-up chain of the ROI stage on the MI355X: the binary-mask heuristics that turn the edge map into the region map.

  remove_thin_structures_optimized, identify_thin_regions_ultrafast / _fast   thin_regions2.py:14-242
  remove_small_noise_regions, remove_small_components_density_aware(_fast)   roi.py:925-1093
  connect_by_closing_fast, remove_small_regions                              small_regions.py:4-21,175-194
  bridge_small_gaps_fast                                                     small_gaps.py:221-319
  detect_meaningful_borders, protect_border_regions, fill_closed_regions     roi.py:784-918
  directional_region_unification, process_and_unify_borders, get_regions     roi.py:14-40,527-607,720-782

PARITY UNPINNED (OpenCV is absent from the build container): connected components with statistics, rectangular / elliptical
morphology, the one-directional filter2D kernels, the 3x3 chamfer distance transform and the box densities are restated from
their published definitions (csrc/ccl.hip, morph.hip, edges.hip; oracle/rhccq_oracle.py restates them again with numpy / scipy:
device == restatement bit for bit, tests/test_gpu_roi.py).  Known deviations from the reference, on purpose:
  * `remove_small_components_density_aware_fast` (roi.py:1025-1093) falls off its end without returning the cleaned image, so the
    reference's `remove_small_noise_regions` raises on `255 - None` whenever a component is removed; here the function returns it;
  * filter2D kernels larger than 11 x 11 run through OpenCV's DFT path, whose float32 rounding noise (~1e-7) is not restated:
    box densities are float32(count) * float32(1 / k^2), and a directional response is positive exactly when a set pixel lies on
    the ray.
Functions take and return numpy arrays as the reference's do; the chain itself (`process_and_unify_borders`, `get_regions`) keeps
its intermediates on the device."""
import math

import numpy as np

from ..ops import default_context
from . import edges as _edges
from . import roi as _roi

RECT3 = [1, 1, 1]


def ellipse_half_widths(ksize):
    """cv2.getStructuringElement(MORPH_ELLIPSE, (ksize, ksize)) as one half-width per row"""
    r = c = ksize // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    return [min(int(np.rint(c * math.sqrt((r * r - (i - r) * (i - r)) * inv_r2))), c) for i in range(ksize)]


def _density_table(kernel_size):
    """density value of every window count (api.edges.compute_local_density's table for a 0 / 255 or 0 / 1 map)"""
    k2 = kernel_size * kernel_size
    term = (np.ones((kernel_size, kernel_size), np.float32) / np.float32(k2))[0, 0]
    if k2 < 130:
        table = np.zeros(k2 + 1, np.float32)
        for m in range(1, k2 + 1):
            table[m] = np.float32(table[m - 1] + term)
        return table
    return (np.arange(k2 + 1, dtype=np.float32) * term).astype(np.float32)


def _count_threshold(kernel_size, threshold, dtype=np.float32):
    """smallest window count whose density exceeds `threshold` (compared in `dtype`, as numpy compares a float32 map with a Python float)"""
    table = _density_table(kernel_size)
    above = np.flatnonzero(table.astype(dtype) > dtype(threshold))
    return int(above[0]) if len(above) else kernel_size * kernel_size + 1


class _Dev:
    """device-side forms of the chain's steps: uint8[H,W] planes (0 / 255) in, planes out"""

    def __init__(self, rh=None):
        self.rh = rh or default_context()

    def up(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(np.where(np.asarray(a) != 0, 255, 0).astype(np.uint8))).to(self.rh.device)

    def components(self, mask, connectivity=8):
        return self.rh.ccl(mask, connectivity, cap=1 << 14)

    def mean_density(self, labels, n, counts, kernel_size, areas):
        """per label: mean of the box density over its pixels; for the DFT-path kernel sizes (k^2 >= 130: density = count / k^2 in
        float32) it is taken from the exact integer sum of the window counts: (sum of counts) * float32(1 / k^2) / area"""
        if kernel_size * kernel_size < 130:
            # direct-path kernel sizes (k <= 11): a pixel's density is the sequential float32 sum table[count]; every table value is a
            # multiple of 2^-30 (1 / 121 >= 2^-7, 24-bit mantissa), so the per-label sum of table[count] * 2^30 is an exact integer
            table = (_density_table(kernel_size).astype(np.float64) * float(1 << 30)).astype(np.uint32)
            sums = self.rh.label_sum(labels, n, self.rh.lut_u16(counts, table)).astype(np.float64) / float(1 << 30)
            with np.errstate(all="ignore"):
                return np.where(areas > 0, sums / np.maximum(areas, 1), 0.0)
        sums = self.rh.label_sum(labels, n, counts).astype(np.float64)
        term = float(_density_table(kernel_size)[1])
        with np.errstate(all="ignore"):
            return np.where(areas > 0, sums * term / np.maximum(areas, 1), 0.0)

    def thin_ids(self, mask, n, labels, stats, min_region_size=10, thinness_threshold=0.3):
        if n == 0:
            return np.zeros(0, np.int64)
        dist = self.rh.dist_chamfer(mask)
        sums = self.rh.label_sum(labels, n, dist).astype(np.float64) / 65536.0
        areas = stats[:, 4].astype(np.float64)
        avg = np.where(areas > 0, sums / np.maximum(areas, 1), 0.0)[1:]
        max_dims = np.maximum(stats[1:, 2], stats[1:, 3]).astype(np.float64)
        norm = np.where(max_dims > 0, avg * 2 / np.maximum(max_dims, 1), 0.0)
        is_thin = ((1.0 - norm) > thinness_threshold) & (stats[1:, 4] >= min_region_size)
        return np.flatnonzero(is_thin) + 1

    def remove_thin(self, mask, density_threshold, window_size, connectivity=8):
        n, labels, stats = self.components(mask, connectivity)
        if n == 0:
            return mask
        if connectivity == 8:
            thin = self.thin_ids(mask, n, labels, stats)
        else:
            # identify_thin_regions_ultrafast labels 8-connected whatever the caller's connectivity; a component of the caller's
            # labelling is thin when it holds a pixel of a thin 8-connected region (thin_regions2.py:38-46)
            n8, labels8, stats8 = self.components(mask, 8)
            lut8 = np.zeros(n8 + 1, np.uint8)
            lut8[self.thin_ids(mask, n8, labels8, stats8)] = 255
            thin_px = self.rh.ccl_select(labels8, lut8)
            hits = self.rh.label_sum(labels, n, self.rh.box_count(thin_px, 1))
            thin = np.flatnonzero(hits[1:] > 0) + 1
        counts = self.rh.box_count(mask, window_size)
        areas = stats[:, 4].astype(np.float64)
        areas[0] = mask.numel() - areas[1:].sum()
        dens = self.mean_density(labels, n, counts, window_size, areas).astype(np.float32)
        keep = np.full(n + 1, 255, np.uint8)
        keep[0] = 0
        keep[thin[dens[thin] < np.float32(density_threshold)]] = 0
        return self.rh.ccl_select(labels, keep)

    def remove_small_density_aware(self, mask, min_size, counts, window_size, density_threshold):
        n, labels, stats = self.components(mask, 8)
        if n == 0:
            return mask
        areas = stats[:, 4].astype(np.float64)
        areas[0] = mask.numel() - areas[1:].sum()
        dens = self.mean_density(labels, n, counts, window_size, areas)
        keep = np.full(n + 1, 255, np.uint8)
        keep[0] = 0
        keep[1:][(stats[1:, 4] < min_size) & (dens[1:] < density_threshold)] = 0
        return self.rh.ccl_select(labels, keep)

    def remove_small_noise(self, mask, min_size=5, density_threshold=0.2, window_size=15):
        counts = self.rh.box_count(mask, window_size)
        white = self.remove_small_density_aware(mask, min_size, counts, window_size, density_threshold)
        black = self.remove_small_density_aware(self.rh.mask_op(white, None, "not"), min_size, counts, window_size, density_threshold)
        return self.rh.mask_op(black, None, "not")

    def bridge(self, mask, max_gap, density_threshold, local_window, regional_window):
        counts = self.rh.box_count(mask, regional_window)
        return self.rh.gap_bridge(mask, counts, _count_threshold(regional_window, density_threshold), min(max_gap, local_window))

    def borders(self, mask, sensitivity):
        m2, mx = self.rh.binary_sobel(mask)
        lut = np.zeros(256, np.uint8)
        if mx > 0:
            mags = np.sqrt(np.arange(33, dtype=np.float32))
            lut[:33] = np.where((mags / mags[mx]) > np.float32(sensitivity * 0.5), 255, 0)
        else:
            lut[:33] = np.where(np.zeros(33, np.float32) > np.float32(sensitivity * 0.5), 255, 0)
        strong = self.rh.lut_u8(m2, lut)
        return self.rh.morph(self.rh.morph_close(strong, RECT3), [2] * 5)

    def protect(self, mask, border, kernel_size):
        closed = self.rh.morph_close_rect(mask, kernel_size)         # (even sizes too: the reference's own default is 18)
        internal = self.rh.mask_op(self.rh.mask_op(closed, mask, "andnot"), border, "andnot")
        return self.rh.mask_op(mask, internal, "or")

    def fill_holes(self, mask, min_hole, max_hole, connectivity):
        inv = self.rh.mask_op(mask, None, "not")
        n, labels, stats = self.components(inv, connectivity)
        lut = np.zeros(n + 1, np.uint8)
        lut[1:] = np.where((stats[1:, 4] >= min_hole) & (stats[1:, 4] <= max_hole), 255, 0)
        return self.rh.mask_op(mask, self.rh.ccl_select(labels, lut), "or")

    def remove_small(self, mask, min_size):
        closed = self.rh.morph_close(mask, RECT3)
        n, labels, stats = self.components(closed, 8)
        lut = np.zeros(n + 1, np.uint8)
        lut[1:] = np.where(stats[1:, 4] >= min_size, 255, 0)
        return self.rh.ccl_select(labels, lut)

    def unify(self, mask):
        border = self.borders(mask, 0.5)
        protected = self.protect(mask, border, 15)
        bridged = self.bridge(protected, 25, 0.2, 15, 25)
        closed = self.fill_holes(bridged, 10, 10000, 4)
        return self.remove_small(closed, 5)


def _u8(t):
    return t.cpu().numpy()


# ---- the reference's functions (numpy in, numpy out) ------------------------------------------------------------------------
def identify_thin_regions_ultrafast(binary_image, min_region_size=10, thinness_threshold=0.3):
    d = _Dev()
    m = d.up(binary_image)
    n, labels, stats = d.components(m, 8)
    if n == 0:
        return np.zeros(np.shape(binary_image), bool)
    lut = np.zeros(n + 1, np.uint8)
    lut[d.thin_ids(m, n, labels, stats, min_region_size, thinness_threshold)] = 1
    return _u8(d.rh.ccl_select(labels, lut)).astype(bool)


identify_thin_regions_fast = identify_thin_regions_ultrafast


def remove_thin_structures_optimized(binary_image, density_threshold=0.2, thinness_threshold=0.3, window_size=25, min_region_size=10, connectivity=8):
    binary_image = np.asarray(binary_image)
    if np.sum(binary_image > 0) == 0:
        return binary_image
    d = _Dev()
    kept = _u8(d.remove_thin(d.up(binary_image), density_threshold, window_size, connectivity))
    return np.where(kept != 0, binary_image, 0).astype(binary_image.dtype)


def remove_small_components_density_aware_fast(binary_image, min_size, foreground=255, density_map=None, density_threshold=0.2, window_size=15):
    """roi.py:1025-1093 (with the `return cleaned_image` the reference's function lacks).  A `density_map` must be the
    compute_local_density map of `window_size`: the device works on its integer window counts, count = rint(density * k^2)."""
    import torch
    binary_image = np.asarray(binary_image)
    d = _Dev()
    fg = d.up(binary_image == foreground)
    if density_map is None:
        counts = d.rh.box_count(fg, window_size)
    else:
        cnt = np.rint(np.asarray(density_map, np.float64) * (window_size * window_size)).astype(np.uint16)
        counts = torch.from_numpy(cnt.view(np.int16)).to(d.rh.device)
    kept = _u8(d.remove_small_density_aware(fg, min_size, counts, window_size, density_threshold)) != 0
    out = binary_image.copy()
    out[(binary_image == foreground) & ~kept] = 0 if foreground == 255 else 255
    return out


remove_small_components_density_aware = remove_small_components_density_aware_fast


def remove_small_noise_regions(binary_image, min_size=5, density_threshold=0.2, window_size=15):
    d = _Dev()
    return _u8(d.remove_small_noise(d.up(np.asarray(binary_image) == 255), min_size, density_threshold, window_size))


def connect_by_closing_fast(binary_image, connection_distance, min_region_size=None):
    d = _Dev()
    return _u8(d.rh.morph_close(d.up(binary_image), ellipse_half_widths(connection_distance * 2 + 1)))


def remove_small_regions(binary_image, min_size=10, remove_thin_lines=False, kernel_size=3):
    d = _Dev()
    return _u8(d.remove_small(d.up(binary_image), min_size))


def bridge_small_gaps_fast(binary_image, max_gap=3, density_threshold=0.3, local_window=5, regional_window=25):
    """small_gaps.py:221-271.  A plane with more than one non-zero value (the notebook's cell 6 hands over 0 / 1 / 255 planes) weighs
    its pixels by their values in the regional density, as the reference's `binary_map / 255.0` does."""
    import torch
    binary_image = np.asarray(binary_image)
    d = _Dev()
    nz = np.unique(binary_image[binary_image != 0])
    if len(nz) <= 1:
        bridged = _u8(d.bridge(d.up(binary_image), max_gap, density_threshold, local_window, regional_window))
    else:
        plane = torch.from_numpy(np.ascontiguousarray(binary_image, dtype=np.uint8)).to(d.rh.device)
        if regional_window * regional_window < 130:
            # direct-path window on a multi-valued plane: the float32 density itself (sequential accumulation), thresholded on the device
            dens = d.rh.box_filter_seq(plane, regional_window, binary_image.max() > 1)
            cand = (dens > np.float32(density_threshold)).to(torch.int16)
            bridged = _u8(d.rh.gap_bridge(plane, cand, 1, min(max_gap, local_window)))
            out = binary_image.copy()
            out[(binary_image == 0) & (bridged != 0)] = 255
            return out
        sums = d.rh.box_sum(plane, regional_window)
        term = _density_table(regional_window)[1]
        scale = 255.0 if binary_image.max() > 1 else 1.0

        def dense(v):
            return np.float32(v / scale) * term > np.float32(density_threshold)
        lo, hi = 0, 255 * regional_window * regional_window + 1          # smallest window sum whose density passes (monotone)
        while lo < hi:
            mid = (lo + hi) // 2
            if dense(mid):
                hi = mid
            else:
                lo = mid + 1
        bridged = _u8(d.rh.gap_bridge(plane, sums, lo, min(max_gap, local_window)))
    out = binary_image.copy()
    out[(binary_image == 0) & (bridged != 0)] = 255
    return out


def detect_meaningful_borders(binary_image, sensitivity=0.7):
    d = _Dev()
    return _u8(d.borders(d.up(binary_image), sensitivity)) != 0


def protect_border_regions(binary_image, border_mask, kernel_size=18):
    binary_image = np.asarray(binary_image)
    d = _Dev()
    prot = _u8(d.protect(d.up(binary_image), d.up(border_mask), kernel_size))
    out = binary_image.copy()
    out[(binary_image == 0) & (prot != 0)] = 255
    return out


def fill_closed_regions(binary_image, min_hole_size=10, max_hole_size=1000, connectivity=4):
    binary_image = np.asarray(binary_image)
    if binary_image.max() <= 1:
        binary_image = (binary_image * 255).astype(np.uint8)
    d = _Dev()
    filled = _u8(d.fill_holes(d.up(binary_image == 255), min_hole_size, max_hole_size, connectivity))
    return np.where((filled != 0) & (binary_image != 255), 255, 0).astype(np.uint8) | binary_image


def directional_region_unification(binary_image, border_sensitivity=0.3, min_region_size=30, max_gap_to_bridge=50,
                                   noise_aggressiveness=2, unification_strength=0.4):
    binary_image = np.asarray(binary_image)
    if binary_image.max() <= 1:
        binary_image = (binary_image * 255).astype(np.uint8)
    d = _Dev()
    cleaned = _u8(d.unify(d.up(binary_image)))
    return cleaned, (cleaned > 0).astype(np.uint8)


def _unify_borders(d, borders_dev, original_image, rgb_dev=None):
    import torch
    m = d.remove_thin(borders_dev, 0.10, 25)
    m = d.remove_small_noise(m, 75)
    m = d.rh.morph_close(m, ellipse_half_widths(11))
    m = d.bridge(m, 100, 0.2, 15, 25)
    u_dev = d.unify(m)
    if u_dev.dtype == torch.bool:
        u_dev = u_dev.view(torch.uint8)
    # the region map goes on to the buffer-zone kernel on the device; the six arrays of the reference's tuple come back together
    # through page-locked buffers
    region_dev = (u_dev != 0).to(torch.uint8)
    if rgb_dev is None:
        rgb_dev = torch.from_numpy(np.array(original_image, dtype=np.uint8, order="C")).to(d.rh.device)
    ri, ni, m1, m0 = d.rh.roi_buffer(region_dev, rgb_dev, 3)
    unified, region_map, roi_image, non_image, roi_mask, non_mask = d.rh.to_host(u_dev, region_dev, ri, ni, m1, m0)
    return unified, region_map, roi_image, non_image, roi_mask, non_mask


def process_and_unify_borders(edge_map, edge_density, original_image, density_threshold=0.3, border_sensitivity=0.3, min_region_size=30,
                              max_gap_to_bridge=10, noise_aggressiveness=2, unification_strength=0.4):
    """roi.py:527-607 -> (unified_borders, region_map, roi_image, nonroi_image, roi_mask, nonroi_mask)"""
    edge_map, edge_density = np.asarray(edge_map), np.asarray(edge_density)
    borders = edge_map.copy()
    borders[~(edge_density > density_threshold)] = 0
    d = _Dev()
    return _unify_borders(d, d.up(borders > 0), original_image)


def get_regions(image_rgb):
    """roi.py:14-40.  Edge map, 3x3 densities, the automatic threshold and the chain stay on the device: the threshold
    mean(density at the edge pixels) / 100 comes from the histogram of the window counts at the edge pixels (the densities take ten
    values, the smallest non-zero one is 1/9, the threshold a few thousandths: its last bits cannot move the mask)."""
    image_rgb = np.asarray(image_rgb)
    d = _Dev()
    a, edge = _edges.edge_map_resident(image_rgb, d.rh)
    counts = d.rh.box_count(edge, 3)
    hist = d.rh.masked_hist(edge, counts, 10)
    table = _density_table(3)
    n_edge = int(hist.sum())
    threshold = (float(np.dot(hist, table.astype(np.float64))) / n_edge if n_edge else 0.1) / 100
    above = np.flatnonzero(table > np.float32(threshold))
    borders = d.rh.value_mask(counts, int(above[0]) if len(above) else 10, edge)
    return _unify_borders(d, borders, image_rgb, rgb_dev=a.rgb if image_rgb.ndim == 3 else None)
