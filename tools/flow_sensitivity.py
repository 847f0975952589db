"""diagnostic: WHICH unpinned upstream stage explains the distance to the reference's artefacts?  The script flow of
tools/notebook_flow.py on the outlier images with one stage's output replaced at a time:
  roi=all / roi=none   the ROI map (edge detection + clean-up chain, OpenCV restated) replaced by everything / nothing
  seg x2 / seg x0.5    the number of SLIC segments per region (split score, scikit-image restated) doubled / halved
and PSNR / palette size / bytes next to the artefact's.  python tools/flow_sensitivity.py [image numbers]"""
import os, sys, math, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def flow(img, roi_override=None, seg_scale=1.0, qr=20, qn=10):
    from encoder.ROI.roi import get_regions, extract_regions
    from encoder.compression.subregions import subregion_quantization
    from encoder.compression.regions import region_quantization
    from encoder.compression.image import quantize_image
    from encoder.compression.compression import lossless_compress_optimized, save_compressed
    from encoder.subregions.split_score import calculate_split_score, normalize_result
    from encoder.subregions.slic import enhanced_slic_with_texture
    from decoder.uncompression.uncompression import load_compressed, lossless_decompress, decompress_color_quantization
    H, W = img.shape[:2]
    unified, region_map, roi_image, nonroi_image, roi_mask, nonroi_mask = get_regions(img)
    if roi_override == "all":
        roi_mask, nonroi_mask = np.ones((H, W), bool), np.zeros((H, W), bool)
    elif roi_override == "none":
        roi_mask, nonroi_mask = np.zeros((H, W), bool), np.ones((H, W), bool)
    roi_regions, nonroi_regions = extract_regions(img, roi_mask, nonroi_mask)

    def segmenter(region, mask):
        overall, _, _ = calculate_split_score(region, mask)
        window = math.ceil(math.ceil(math.log(region.size, 10)) * math.log(region.size))
        optimal = max(1, math.ceil(normalize_result(overall, window) * seg_scale))
        return enhanced_slic_with_texture(region, mask, n_segments=optimal)[0]
    rc = subregion_quantization(img, roi_regions, quality=qr, subregion_type="ROI", segmenter=segmenter) if roi_regions else []
    nc = subregion_quantization(img, nonroi_regions, quality=qn, subregion_type="nonROI", segmenter=segmenter) if nonroi_regions else []
    q2r, q2n = min(qr * 2, 100), min(qn * 2, 100)
    try:
        r2 = region_quantization(rc, quality=q2r, original_image_height=H, original_image_width=W)
    except Exception:
        r2 = []
    try:
        n2 = region_quantization(nc, quality=q2n, original_image_height=H, original_image_width=W)
    except Exception:
        n2 = []
    final = quantize_image(r2 + n2, quality=min(q2r + q2n, 100), original_image_height=H, original_image_width=W)
    pkg = lossless_compress_optimized(final["palette"], final["indices"], final["shape"])
    path = "gpurun_out/sens.rhccq"
    save_compressed(pkg, path)
    rec = np.asarray(decompress_color_quantization(lossless_decompress(load_compressed(path)))["image"])
    mse = float(np.mean((rec.astype(np.float64) - img.astype(np.float64)) ** 2))
    return {"colours": int(len(np.unique(rec.reshape(-1, 3), axis=0))), "psnr": round(10 * math.log10(255.0 ** 2 / mse), 2), "bytes": os.path.getsize(path),
            "roi_fraction": round(float(np.mean(roi_mask)), 3)}


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--scale":
    # every Kodak image with the SLIC segment count scaled: how far from the artefacts with e.g. 1 / 0.7^2 = 2.04 x the segments,
    # the count before the area adjustment of slic.py:82?
    os.makedirs("gpurun_out", exist_ok=True)
    ref = json.load(open(os.path.join(G, "g14_artefact_stats.json")))
    sc = float(sys.argv[2])
    out = {}
    for n in range(1, 25):
        img = np.asarray(Image.open(os.path.join(G, f"kodak_{n}.png")).convert("RGB"), np.uint8)
        out[f"kodak_{n}"] = {"artefact": ref[f"kodak_{n}"], f"seg x{sc}": flow(img, seg_scale=sc)}
        print(f"kodak_{n}", json.dumps(out[f"kodak_{n}"]), flush=True)
    d = [v[f"seg x{sc}"]["psnr"] - v["artefact"]["psnr"] for v in out.values()]
    b = [100.0 * (v[f"seg x{sc}"]["bytes"] - v["artefact"]["bytes"]) / v["artefact"]["bytes"] for v in out.values()]
    c = [v[f"seg x{sc}"]["colours"] - v["artefact"]["colours"] for v in out.values()]
    out["summary"] = {"scale": sc, "dpsnr_mean": float(np.mean(d)), "dpsnr_min": float(np.min(d)), "dpsnr_max": float(np.max(d)),
                      "abs_dbytes_pct_mean": float(np.mean(np.abs(b))), "abs_dbytes_pct_max": float(np.max(np.abs(b))), "dcolours_mean": float(np.mean(c)),
                      "abs_dcolours_mean": float(np.mean(np.abs(c)))}
    print("summary", json.dumps(out["summary"]))
    json.dump(out, open(f"gpurun_out/flow_scale_{sc}.json", "w"), indent=1)
    sys.exit(0)

if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    ref = json.load(open(os.path.join(G, "g14_artefact_stats.json")))
    out = {}
    for n in [int(a) for a in sys.argv[1:]] or [3, 17, 19, 20, 6, 24]:
        img = np.asarray(Image.open(os.path.join(G, f"kodak_{n}.png")).convert("RGB"), np.uint8)
        row = {"artefact": ref[f"kodak_{n}"]}
        for name, kw in (("as built", {}), ("roi=all", {"roi_override": "all"}), ("roi=none", {"roi_override": "none"}), ("seg x2", {"seg_scale": 2.0}),
                         ("seg x0.5", {"seg_scale": 0.5})):
            row[name] = flow(img, **kw)
        out[f"kodak_{n}"] = row
        print(f"kodak_{n}", json.dumps(row), flush=True)
    json.dump(out, open("gpurun_out/flow_sensitivity.json", "w"), indent=1)
