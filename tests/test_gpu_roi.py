"""Region extraction of the ROI stage (encoder/ROI/roi.py:45-103,285-360,685-718) on the device vs the oracle.
GPU only.  Connected components: partition pinned by scipy.ndimage.label, OpenCV's numbering PARITY UNPINNED (restated rule,
tests/test_roi_cpu.py); extract_roi_nonroi: scipy + numpy in the reference too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _masks():
    from test_roi_cpu import masks
    rng = np.random.default_rng(9)
    out = masks()
    for i, (h, w, d) in enumerate([(130, 257, 0.5), (3, 64, 0.7), (65, 128, 0.35), (200, 193, 0.58)]):
        out.append((f"big{i}", rng.random((h, w)) < d))
    yy, xx = np.mgrid[0:96, 0:150]
    out.append(("spiral", ((np.hypot(yy - 48, xx - 75) + 6 * np.arctan2(yy - 48, xx - 75)) % 12) < 5))
    out.append(("comb", (xx % 4 == 0) | (yy == 95)))                       # long vertical runs joined at the bottom: deep trees
    out.append(("zigzag", ((yy + xx) % 7 == 0) | ((yy - xx) % 11 == 0)))
    return out


@pytest.mark.parametrize("connectivity", [4, 8])
def test_connected_components_vs_oracle(connectivity):
    import torch
    from oracle import rhccq_oracle as O
    from roibasedimagecompression_amd.ops import Rhccq
    rh = Rhccq(0)
    for name, m in _masks():
        t = torch.from_numpy(np.ascontiguousarray(m)).to(rh.device)
        for numbering in ("opencv", "raster"):
            n, lab, stats = rh.ccl(t, connectivity, cap=8, numbering=numbering)        # cap 8: the capacity retry is exercised
            num, want, wstats = O.cv_connected_components_with_stats(m, connectivity, numbering)
            assert n + 1 == num, (name, numbering)
            assert np.array_equal(lab.cpu().numpy(), want), (name, numbering)
            assert np.array_equal(stats[1:], wstats[1:]), (name, numbering)
            if (~m).any():
                assert np.array_equal(stats[0], wstats[0]), (name, numbering)
        keep = np.zeros(n + 1, np.uint8)
        keep[1::2] = 255
        assert np.array_equal(rh.ccl_select(lab, keep).cpu().numpy(), keep[want])


def test_connected_components_full_size_partition():
    """4K (too slow for the literal rule): the partition equals scipy.ndimage.label's, numbering keys strictly increase, areas add up"""
    import torch
    from scipy import ndimage
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.ops import Rhccq
    rh = Rhccq(0)
    H, W = 2160, 3840
    g = synth.photo(H, W, 77)[..., 1].astype(np.int32)
    m = (np.abs(np.diff(g, axis=1, prepend=g[:, :1])) + np.abs(np.diff(g, axis=0, prepend=g[:1])) > 9)   # an edge-map-like mask
    n, lab, stats = rh.ccl(torch.from_numpy(m).to(rh.device), 8)
    lab = lab.cpu().numpy()
    want, nw = ndimage.label(m, structure=np.ones((3, 3)))
    assert n == nw and np.array_equal(lab != 0, m)
    pairs = np.unique(np.stack([lab[m], want[m]], 1), axis=0)
    assert len(pairs) == n                                                   # a bijection between the two numberings
    assert stats[1:, 4].sum() == m.sum() and np.array_equal(stats[1:, 4], np.bincount(lab[m], minlength=n + 1)[1:])
    first = ndimage.minimum(np.arange(H * W).reshape(H, W), lab, np.arange(1, n + 1))
    key = ((first // W) >> 1) * ((W + 1) >> 1)
    assert np.all(np.diff(key) >= 0)                                         # block rows never decrease along the numbering


def test_extract_roi_nonroi_and_regions_vs_oracle():
    from oracle import rhccq_oracle as O
    from encoder.ROI.roi import extract_regions, extract_roi_nonroi, extract_connected_regions
    from roibasedimagecompression_amd import synth
    rng = np.random.default_rng(3)
    for H, W, bs in ((60, 90, 3), (131, 77, 3), (64, 200, 1), (100, 100, 5)):
        img = synth.photo(H, W, H)
        yy, xx = np.mgrid[0:H, 0:W]
        region_map = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2 < (min(H, W) / 3) ** 2) | ((yy < 5) & (xx > W - 9)) | (rng.random((H, W)) < 0.01)).astype(np.uint8)
        got = extract_roi_nonroi(img, region_map, bs)
        want = O.extract_roi_nonroi(img, region_map, bs)
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        roi, non = extract_regions(img, got[2], got[3])
        wroi, wnon = O.extract_regions(img, want[2], want[3])
        assert len(roi) == len(wroi) and len(non) == len(wnon)
        for r, w in zip(roi + non, wroi + wnon):
            assert tuple(r["bbox"]) == tuple(w["bbox"]) and r["area"] == w["area"] and r["label"] == w["label"] and r.get("type") == w.get("type")
            assert np.array_equal(r["bbox_mask"], w["bbox_mask"]) and np.array_equal(r["bbox_image"], w["bbox_image"])
            assert np.array_equal(r["mask"], w["mask"]) and np.array_equal(r["full_image"], w["full_image"]) and np.array_equal(r["coords"], w["coords"])
        assert "mask" in roi[0] and roi[0].get("nope") is None
    # skimage numbering (extract_connected_regions): first pixel in raster order
    regs = extract_connected_regions(got[3], img)
    firsts = [int(np.flatnonzero(r["mask"].ravel())[0]) for r in regs]
    assert firsts == sorted(firsts)
    # buffer_size < 1: scipy dilates until nothing changes
    a = extract_roi_nonroi(img, region_map, 0)
    b = O.extract_roi_nonroi(img, region_map, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_region_lists_through_level_one_with_overlapping_regions():
    """extract_roi_nonroi -> extract_regions -> subregion_quantization: the small ROI components that extract_regions appends
    to the non-ROI list overlap non-ROI regions in the buffer zone; every region must come out as if encoded on its own
    (the reference loops over regions), = the oracle's level 1 per region."""
    from oracle import rhccq_oracle as O
    from encoder.ROI.roi import extract_regions, extract_roi_nonroi
    from encoder.compression.subregions import subregion_quantization
    from roibasedimagecompression_amd import synth
    H, W = 96, 128
    img = synth.photo(H, W, 21, sigma=3.0)
    yy, xx = np.mgrid[0:H, 0:W]
    region_map = (((yy - 48) ** 2 + (xx - 60) ** 2 < 30 ** 2) | ((yy > 80) & (yy < 86) & (xx > 100) & (xx < 110))).astype(np.uint8)
    _, _, rm, nm = extract_roi_nonroi(img, region_map)
    roi, non = extract_regions(img, rm, nm)
    assert len(roi) == 1 and len(non) == 2 and non[-1].get("type") == "nonroi"
    assert (non[0]["mask"] & non[1]["mask"]).any()                           # the overlap this test is about

    def one_segment(bbox_region, bbox_mask):
        return bbox_mask.astype(np.int32)
    for regions, q in ((roi, 20), (non, 10)):
        got = subregion_quantization(img, regions, quality=q, segmenter=one_segment)
        assert len(got) == len(regions)
        for r, g in zip(regions, got):
            ref = O.level1_region(img, tuple(r["bbox"]), r["bbox_mask"], r["bbox_mask"].astype(np.int32), q)
            assert len(g) == len(ref) == 1
            assert tuple(g[0]["top_left"]) == tuple(ref[0]["top_left"]) and tuple(g[0]["shape"]) == tuple(ref[0]["shape"])
            assert np.array_equal(np.asarray(g[0]["palette"]).reshape(-1, 3), np.asarray(ref[0]["palette"]).reshape(-1, 3))
            assert np.array_equal(np.asarray(g[0]["indices"]).reshape(-1), np.asarray(ref[0]["indices"]).reshape(-1))
