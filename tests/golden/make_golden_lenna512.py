#!/usr/bin/env python3
"""G12: BASELINE.json configs[0] -- the whole 512x512 Lenna PNG through the REFERENCE's three-level chain
(rhccq.ipynb:978-1039: subregion_quantization -> region_quantization x2 -> quantize_image), quality tiers (20, 10).
Build container only (imports /root/reference; ~1 minute).

    python tests/golden/make_golden_lenna512.py

encoder.compression.subregions itself is not importable here (it pulls scikit-image through encoder.subregions.*), so its
own logic (subregions.py:315-449,634-679: crop +-2 px, mask, black-in-segment fix, get_all_unique_colors,
cluster_palette_colors_parallel, merge per region) is driven as the same call sequence of the importable functions on a
fixed label map: an 8x8 grid of 64-px tiles (64 segments, about what the reference's SLIC produces on Lenna) split into the
ROI class (centred ellipse, 35 % of the frame, dilated by 3 px) and the non-ROI class.  Everything downstream of the
label map is the reference's code.  Only data is written: per level the palette, sha256 + a sample of the index map,
the final index map in full, and the container bytes' hash."""
import contextlib
import hashlib
import io
import json
import os
import sys
import time

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402
import sklearn  # noqa: E402
from PIL import Image  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from encoder.compression import clustering as R_clu  # noqa: E402
    from encoder.compression import merging as R_mrg  # noqa: E402
    from encoder.compression import compression as R_cmp  # noqa: E402
    from encoder.compression import regions as R_reg  # noqa: E402
    from encoder.compression import image as R_img  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def seg_arrays(seg):
    return (np.array(seg["palette"], dtype=np.int64).reshape(-1, 3).astype(np.uint8),
            np.array(seg["indices"], dtype=np.int64).reshape(-1))


def labelmaps(H, W, tiles, roi_frac=0.35, overlap=3):
    """segment ids (>= 0) of the ROI class and of the non-ROI class, -1 elsewhere (same construction as G6)"""
    yy, xx = np.mgrid[0:H, 0:W]
    ell = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 <= roi_frac * 4 / np.pi
    dil = ell.copy()
    for _ in range(overlap):
        d2 = dil.copy()
        d2[1:] |= dil[:-1]; d2[:-1] |= dil[1:]; d2[:, 1:] |= dil[:, :-1]; d2[:, :-1] |= dil[:, 1:]
        dil = d2
    tid = (yy // (H // tiles)) * tiles + (xx // (W // tiles))
    return np.where(dil, tid, -1), np.where(~ell, tid, -1)


def level1(img, seglab, q):
    mask = seglab >= 0
    rows, cols = np.where(mask)
    minr, maxr, minc, maxc = rows.min(), rows.max() + 1, cols.min(), cols.max() + 1
    region_image = img[minr:maxr, minc:maxc]
    bbox_mask = mask[minr:maxr, minc:maxc]
    lab = seglab[minr:maxr, minc:maxc]
    comps = []
    for sid in np.unique(lab[lab >= 0]):
        segment_mask = (lab == sid) & bbox_mask
        r, c = np.where(segment_mask)
        h, w = region_image.shape[:2]
        r0, r1 = max(0, r.min() - 2), min(h - 1, r.max() + 2)
        c0, c1 = max(0, c.min() - 2), min(w - 1, c.max() + 2)
        crop = region_image[r0:r1 + 1, c0:c1 + 1]
        mcrop = segment_mask[r0:r1 + 1, c0:c1 + 1]
        segimg = np.zeros_like(crop)
        px = crop[mcrop].copy()
        isb = np.all(px == 0, axis=1)
        if isb.any() and (~isb).any():                      # black-in-segment fix (subregions.py:393-421)
            nb = px[~isb]
            for i in np.where(isb)[0]:
                px[i] = nb[np.argmin(np.linalg.norm(nb - px[i], axis=1))]
        segimg[mcrop] = px
        d = quiet(R_clu.get_all_unique_colors, segimg, (int(r0 + minr), int(c0 + minc)))
        eps, ms, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
        comps.append(quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=1, max_colors_per_cluster=mc))
    if len(comps) > 1:
        return [quiet(R_mrg.merge_region_components_simple, comps, (int(minr), int(minc), int(maxr), int(maxc)))]
    return [comps]


def main():
    img = np.asarray(Image.open(os.path.join(HERE, "Lenna.png")).convert("RGB"), dtype=np.uint8)
    H, W = img.shape[:2]
    lr, ln = labelmaps(H, W, 8)
    q_roi, q_non = 20, 10
    t0 = time.time()
    roi1 = level1(img, lr, q_roi)
    non1 = level1(img, ln, q_non)
    t1 = time.time()
    q2r, q2n = min(q_roi * 2, 100), min(q_non * 2, 100)
    roi2 = quiet(R_reg.region_quantization, roi1, H, W, q2r)
    non2 = quiet(R_reg.region_quantization, non1, H, W, q2n)
    q3 = min(q2r + q2n, 100)
    fin = quiet(R_img.quantize_image, roi2 + non2, H, W, q3)
    t2 = time.time()
    pk = quiet(R_cmp.lossless_compress_optimized, fin["palette"], fin["indices"], fin["shape"])
    tmp = os.path.join(HERE, "_g12_tmp.rhccq")
    quiet(R_cmp.save_compressed, pk, tmp)
    raw = open(tmp, "rb").read()
    os.remove(tmp)
    arrs = {"lab_roi": lr.astype(np.int8), "lab_non": ln.astype(np.int8), "q": np.array([q_roi, q_non])}
    meta = {"numpy": np.__version__, "scikit-learn": sklearn.__version__, "seconds_level1": round(t1 - t0, 1),
            "seconds_levels23": round(t2 - t1, 2), "container_sha256": hashlib.sha256(raw).hexdigest(), "container_bytes": len(raw),
            "fin_dtype": str(fin.get("indices_dtype", "none")), "levels": {}}
    for nm, s in (("roi1", roi1[0][0]), ("non1", non1[0][0]), ("roi2", roi2[0]), ("non2", non2[0]), ("fin", fin)):
        pal, idx = seg_arrays(s)
        arrs[f"{nm}_pal"] = pal
        meta["levels"][nm] = {"colours": int(len(pal)), "top_left": [int(v) for v in s["top_left"]], "shape": [int(v) for v in s["shape"]],
                              "indices_sha256": hashlib.sha256(idx.astype(np.int32).tobytes()).hexdigest()}
        rec = pal[idx].astype(np.float64)
        h, w = s["shape"]
        r0, c0 = s["top_left"]
        ref = img[r0:r0 + h, c0:c0 + w].reshape(-1, 3).astype(np.float64)
        cov = idx != 0 if nm != "fin" else np.ones(len(idx), bool)
        mse = float(np.mean((rec[cov] - ref[cov]) ** 2))
        meta["levels"][nm]["psnr_covered"] = 10 * np.log10(255.0 ** 2 / mse)
    arrs["fin_idx"] = seg_arrays(fin)[1].astype(np.uint8 if len(fin["palette"]) <= 256 else np.uint16)
    np.savez_compressed(os.path.join(HERE, "g12_lenna512.npz"), **arrs)
    json.dump(meta, open(os.path.join(HERE, "g12_lenna512.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
