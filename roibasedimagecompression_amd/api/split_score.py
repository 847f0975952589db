"""Mirror of encoder/subregions/split_score.py: calculate_split_score, normalize_result,
calculate_optimal_segments with the reference's names, arguments and return values.

The per-pixel work (gray / Lab conversion, Sobel magnitudes, uniform LBP codes, masked sums and histograms) runs in
one HIP pass (csrc/split_score.hip); the dozen scalar formulas that turn those statistics into the three scores are the
reference's own (split_score.py:33-131) on the host, in float64.

PARITY UNPINNED: the reference calls scikit-image, which is absent from the build container and unpinned in
requirements.txt; rgb2lab, rgb2gray, filters.sobel and feature.local_binary_pattern(8, 1, 'uniform') are restated from
their published definitions, the oracle states the same in numpy, and the two agree to float64 rounding."""
import math

import numpy as np
import torch

from ..ops import default_context


def calculate_optimal_segments(split_score, region_area, min_segments=5, max_segments=50):
    """split_score.py:8-13"""
    base_segments = min_segments + (max_segments - min_segments) * split_score
    area_factor = np.sqrt(region_area / 1000)
    optimal_segments = int(base_segments * area_factor)
    return np.clip(optimal_segments, min_segments, max_segments)


def scores_from_stats(sums, lbp_hist, gray_hist):
    """the scalar part of calculate_split_score (split_score.py:33-131) from the masked statistics"""
    n = float(sums[0])

    def std(s1, s2):
        m = s1 / n
        return math.sqrt(max(s2 / n - m * m, 0.0))
    l_std, a_std, b_std = std(sums[1], sums[2]), std(sums[3], sums[4]), std(sums[5], sums[6])
    color_variance = (l_std / 100 + a_std / 128 + b_std / 128) / 3
    gradient_score = (sums[7] / n) / 3
    color_score = float(np.clip(0.7 * color_variance + 0.3 * gradient_score, 0, 1))
    h = lbp_hist.astype(np.float64) / n                               # density over bins of width 1
    lbp_score = np.clip(-np.sum(h * np.log2(h + 1e-8)) / 3.0, 0, 1) if n > 10 else 0
    gm = sums[8] / n
    grad_score = np.clip(max(sums[9] / n - gm * gm, 0.0) * 50, 0, 1) if n > 10 else 0
    d = gray_hist.astype(np.float64) / n * 32.0                       # density over 32 bins of width 1/32
    entropy_score = np.clip(-np.sum(d * np.log2(d + 1e-8)) / 5.0, 0, 1) if n > 10 else 0
    std_score = np.clip(std(sums[10], sums[11]) * 2, 0, 1)
    texture_score = float(np.clip((lbp_score + grad_score + entropy_score + std_score) / 4, 0, 1))
    return 0.4 * color_score + 0.6 * texture_score, color_score, texture_score


def calculate_split_score(region_image, mask=None):
    """Returns (overall_score, color_score, texture_score); (0.0, 0.0, 0.0) for regions of fewer than 100 masked pixels
    (split_score.py:25-27)."""
    rh = default_context()
    img = np.ascontiguousarray(region_image)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise TypeError("calculate_split_score: uint8 RGB region image expected")
    d_mask = None
    if mask is not None:
        m = np.ascontiguousarray(np.asarray(mask, dtype=bool))
        if m.shape != img.shape[:2]:
            raise ValueError("mask shape does not match the region image")
        if m.sum() < 100:
            return 0.0, 0.0, 0.0
        d_mask = torch.from_numpy(m.view(np.uint8)).to(rh.device)
    sums, lbp_hist, gray_hist = rh.split_stats(torch.from_numpy(np.array(img, dtype=np.uint8, order="C")).to(rh.device), d_mask)
    if sums[0] < 100:
        return 0.0, 0.0, 0.0
    return scores_from_stats(sums, lbp_hist, gray_hist)


def normalize_result(score, window_size):
    """split_score.py:143-144"""
    return window_size / (1 + math.exp(-12 * (score - 0.5)))
