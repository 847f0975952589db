"""CPU checks of the oracle's region-extraction restatements (encoder/ROI/roi.py:45-103,285-360,685-718).
PARITY UNPINNED for OpenCV's label numbering (cv2 is absent from the build container): what is checked here is the
restated rule against a literal flood fill in block-raster order, the partition against scipy, and extract_roi_nonroi
(scipy + numpy in the reference too) against the L1-ball definition of its dilations."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rhccq_oracle as O  # noqa: E402


def masks():
    rng = np.random.default_rng(5)
    out = [("empty", np.zeros((5, 7), bool)), ("full", np.ones((6, 9), bool)), ("one", np.ones((1, 1), bool)),
           ("row", rng.random((1, 70)) < 0.5), ("col", rng.random((70, 1)) < 0.5)]
    for i, (h, w, d) in enumerate([(17, 23, 0.3), (40, 130, 0.45), (64, 64, 0.6), (33, 200, 0.2), (90, 67, 0.52)]):
        out.append((f"rand{i}", rng.random((h, w)) < d))
    yy, xx = np.mgrid[0:48, 0:80]
    out.append(("checker", (yy + xx) % 2 == 0))
    out.append(("dots", (yy % 2 == 0) & (xx % 2 == 0)))
    out.append(("stripes", yy % 3 == 0))
    return out


def flood_numbering(mask, connectivity):
    """literal restatement of the numbering rule: visit 2x2 blocks in raster order (pixels for 4-connectivity); an unlabelled
    foreground pixel starts the next component, flood-filled at once"""
    H, W = mask.shape
    lab = np.zeros((H, W), np.int32)
    nb = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([(-1, -1), (-1, 1), (1, -1), (1, 1)] if connectivity == 8 else [])
    order = ([(y, x) for by in range(0, H, 2) for bx in range(0, W, 2) for y in (by, by + 1) for x in (bx, bx + 1) if y < H and x < W]
             if connectivity == 8 else [(y, x) for y in range(H) for x in range(W)])
    n = 0
    for y, x in order:
        if not mask[y, x] or lab[y, x]:
            continue
        n += 1
        stack = [(y, x)]
        lab[y, x] = n
        while stack:
            cy, cx = stack.pop()
            for dy, dx in nb:
                qy, qx = cy + dy, cx + dx
                if 0 <= qy < H and 0 <= qx < W and mask[qy, qx] and not lab[qy, qx]:
                    lab[qy, qx] = n
                    stack.append((qy, qx))
    return n, lab


@pytest.mark.parametrize("connectivity", [4, 8])
def test_component_numbering_rule(connectivity):
    for name, m in masks():
        num, lab, stats = O.cv_connected_components_with_stats(m, connectivity)
        n, ref = flood_numbering(m, connectivity)
        assert num == n + 1 and np.array_equal(lab, ref), name
        for k in range(1, num):
            ys, xs = np.nonzero(lab == k)
            assert tuple(stats[k]) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)), name
        if (~m).any():
            assert stats[0, 4] == (~m).sum()


def test_extract_regions_and_buffer_zone():
    rng = np.random.default_rng(11)
    H, W = 60, 90
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    region_map = (((yy - 30) ** 2 + (xx - 40) ** 2 < 400) | ((yy < 6) & (xx > 80))).astype(np.uint8)
    ri, ni, rm, nm = O.extract_roi_nonroi(img, region_map, 3)
    # L1 ball of radius 3, border value 0
    def ball(core):
        out = np.zeros_like(core)
        for dy in range(-3, 4):
            for dx in range(-(3 - abs(dy)), 3 - abs(dy) + 1):
                sh = np.zeros_like(core)
                ys, yd = slice(max(0, -dy), H - max(0, dy)), slice(max(0, dy), H - max(0, -dy))
                xs, xd = slice(max(0, -dx), W - max(0, dx)), slice(max(0, dx), W - max(0, -dx))
                sh[yd, xd] = core[ys, xs]
                out |= sh
        return out
    buf = ball(region_map == 1) & ball(region_map == 0)
    assert np.array_equal(rm, (region_map == 1) | buf) and np.array_equal(nm, (region_map == 0) | buf)
    assert np.array_equal(ri, img * rm[..., None]) and np.array_equal(ni, img * nm[..., None])
    roi, non = O.extract_regions(img, rm, nm)
    mn = O.roi_min_region_size(img)
    assert mn == 162 and all(r["area"] >= mn for r in roi)            # 16 200 values -> ceil(16200 / 10^(5-3)), roi.py:47-49
    assert [r.get("type") for r in non].count("nonroi") == 1 and non[-1]["area"] < mn      # the small corner component moved over
    for r in roi + non:
        y0, x0, y1, x1 = r["bbox"]
        assert r["bbox_mask"].shape == (y1 - y0, x1 - x0) and r["bbox_mask"].sum() == r["area"] == len(r["coords"])
        assert np.array_equal(r["full_image"], img * r["mask"][..., None])


def test_edge_front_end_restatements():
    """known answers of the OpenCV restatements that need no OpenCV to verify"""
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 0, 0], [12, 200, 77]]], np.uint8)
    assert O.cv_rgb2gray(px).tolist() == [[255, 76, 150, 29, 0, 130]]                    # cvtColor's documented 8-bit results
    step = np.zeros((12, 16), np.uint8)
    step[:, 8:] = 200
    gx, gy = O.cv_sobel3(step, "reflect")
    assert (gy == 0).all() and (gx[:, 7] == 800).all() and (gx[:, 8] == 800).all() and (gx[:, :7] == 0).all()
    nm = O.cv_canny_nms(step)
    assert (nm[:, 7] == 800).all() and (nm[:, 8] == 0).all() and nm.sum() == 800 * 12     # "> left and >= right": the left pixel of the pair wins
    assert np.array_equal(O.cv_canny(step, 100, 200) > 0, nm > 0) and not O.cv_canny(step, 100, 900).any()
    # hysteresis: a weak run touching a strong pixel survives as a whole, an isolated weak run does not
    nm2 = np.zeros((5, 12), np.uint16)
    nm2[1, 1:6] = 60
    nm2[1, 6] = 300
    nm2[3, 8:11] = 60
    out = O.cv_canny(np.zeros((5, 12), np.uint8), 50, 200, nm2)
    assert out[1, 1:7].all() and not out[3].any()
    bim = np.r_[np.full(500, 40), np.full(300, 200)].astype(np.uint8).reshape(20, 40)
    assert 40 <= O.cv_otsu(bim) < 200
    d = O.local_density(np.pad(np.full((1, 1), 255, np.uint8), 3), 3)
    assert d.dtype == np.float32 and d[3, 3] == np.float32(1) / np.float32(9) and d[0, 0] == 0 and (d > 0).sum() == 9


def test_host_arithmetic_of_the_edge_front_end():
    """the host half of api/edges.py and api/roi_chain.py needs no GPU: percentiles and gradient statistics from histograms vs numpy on
    the expanded data, Otsu vs the oracle, OpenCV's ellipse, the density tables and count thresholds"""
    from roibasedimagecompression_amd.api import edges as E
    from roibasedimagecompression_amd.api import roi_chain as C
    rng = np.random.default_rng(8)
    for n in (1, 2, 7, 1000, 4321):
        x = np.sort(rng.integers(0, 60, n))
        v, c = np.unique(x, return_counts=True)
        for q in (0, 10, 25, 50, 70, 75, 90, 100):
            assert E._percentile(v.astype(np.float64), c, q) == np.percentile(x.astype(np.float64), q), (n, q)
        assert E._percentile(v, c, 25) == np.percentile(x.astype(np.uint8), 25)
    gray = rng.integers(0, 256, (60, 80)).astype(np.uint8)
    gray[:, :40] //= 3
    assert E._otsu(np.bincount(gray.ravel(), minlength=256)) == O.cv_otsu(gray)
    gx, gy = O.cv_sobel3(gray, "reflect")
    m2 = (gx * gx + gy * gy).ravel()
    v, c = np.unique(m2, return_counts=True)
    g = E._Gradient(v, c, m2.size)
    mag = np.sqrt(gx.astype(np.float64) ** 2 + gy.astype(np.float64) ** 2)
    assert abs(g.mean() - np.mean(mag)) < 1e-9 and abs(g.std() - np.std(mag)) < 1e-9
    assert g.percentile_nonzero(70) == np.percentile(mag[mag > 0], 70) and g.percentile_nonzero(90) == np.percentile(mag[mag > 0], 90)
    for k in (1, 3, 5, 11, 15, 21):
        assert C.ellipse_half_widths(k) == O.cv_ellipse_half_widths(k)
    assert C.ellipse_half_widths(5) == [0, 2, 2, 2, 0]                      # OpenCV's 5x5 ellipse: 00100 / 11111 / 11111 / 11111 / 00100
    one = np.zeros((9, 9), np.uint8)
    one[4, 4] = 255
    for k in (3, 5, 15):
        t = C._density_table(k)
        d = O.local_density(one, k)
        assert t[1] == d[4, 4] and t[0] == 0 and len(t) == k * k + 1
    assert C._count_threshold(25, 0.2) == 126 and C._count_threshold(3, 0.0041) == 1 and C._count_threshold(15, 0.2, np.float64) == 45
