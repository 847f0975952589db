"""The reference's end-to-end script flow (encoder/compression/test.py:77-151 = rhccq.ipynb cells 6-16) driven through the mirrored
modules only: image -> ROI detection -> regions -> split score + masked SLIC -> three clustering levels (-> .rhccq file).  Used by
bench.py (`value_with_upstream`), tests/test_gpu_notebook.py and tools/notebook_flow.py.  The ROI and SLIC stages are parity-unpinned
restatements of OpenCV / scikit-image."""
import time


def script_flow(image_rgb, roi_quality=20, nonroi_quality=10, out_path=None, container=True):
    """encoder/compression/test.py:77-151 (the script twin of the notebook: the flow that wrote images/rhccq_20_10/*.rhccq) with the
    reference's import lines.  (The notebook's own cell 6 inlines the ROI chain and, through a uint8 overflow in
    `(connected * 255).astype(np.uint8)`, feeds 0 / 1 / 255 images to the later steps; the script calls get_regions.)"""
    from encoder.ROI.roi import get_regions, extract_regions
    from encoder.compression.subregions import subregion_quantization
    from encoder.compression.regions import region_quantization
    from encoder.compression.image import quantize_image
    from encoder.compression.compression import lossless_compress_optimized, save_compressed
    t = {}
    t0 = time.perf_counter()
    unified, region_map, roi_image, nonroi_image, roi_mask, nonroi_mask = get_regions(image_rgb)
    roi_regions, nonroi_regions = extract_regions(image_rgb, roi_mask, nonroi_mask)
    t["roi_stage"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    from roibasedimagecompression_amd.api import subregions as _sub
    roi_components = subregion_quantization(image_rgb, roi_regions, quality=roi_quality, subregion_type="ROI", debug=False)
    seg_roi = dict(_sub.last_stats)
    nonroi_components = subregion_quantization(image_rgb, nonroi_regions, quality=nonroi_quality, subregion_type="nonROI", debug=False)
    seg_non = dict(_sub.last_stats)
    t["level1"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    H, W = image_rgb.shape[:2]
    q2r, q2n = min(roi_quality * 2, 100), min(nonroi_quality * 2, 100)
    try:
        roi2 = region_quantization(roi_components, quality=q2r, original_image_height=H, original_image_width=W)
    except Exception:                                       # the notebook's bare `except: roi_components = []`
        roi2 = []
    try:
        non2 = region_quantization(nonroi_components, quality=q2n, original_image_height=H, original_image_width=W)
    except Exception:
        non2 = []
    final = quantize_image(roi2 + non2, quality=min(q2r + q2n, 100), original_image_height=H, original_image_width=W)
    t["levels23"] = time.perf_counter() - t0
    pkg = None
    if container or out_path:                               # zlib level 9 on one host core: 1.3 s of a 1.6 s 4K run
        t0 = time.perf_counter()
        pkg = lossless_compress_optimized(final["palette"], final["indices"], final["shape"])
        if out_path:
            save_compressed(pkg, out_path)
        t["container"] = time.perf_counter() - t0
    return final, pkg, {"region_map_roi_fraction": float(region_map.mean()), "roi_regions": len(roi_regions), "nonroi_regions": len(nonroi_regions),
                        "roi_segments": seg_roi.get("segments", 0), "nonroi_segments": seg_non.get("segments", 0),
                        "segments_dropped": seg_roi.get("segments_dropped", 0) + seg_non.get("segments_dropped", 0),
                        "edge_fraction": float((unified > 0).mean()), "seconds": {k: round(v, 3) for k, v in t.items()}}
