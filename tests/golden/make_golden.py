#!/usr/bin/env python3
"""Generate golden vectors by running the *reference* RHCCQ encoder in this container.

Usage (build container only; /root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference is imported from /root/reference (read-only) with bytecode writing
disabled; only *data* (inputs and the reference's outputs) is written to
tests/golden/.  No reference source text is copied.

Fixture ids follow SURVEY.md Appendix B (G1..G8) plus G9/G10 for the k-means
branches.  Library versions are recorded in versions.json because the reference
does not pin scikit-learn / numpy (requirements.txt:2,6).
"""
import contextlib
import hashlib
import io
import json
import math
import os
import pickle
import struct
import sys
import zlib

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
    from decoder.uncompression import uncompression as R_dec  # noqa: E402
from sklearn.cluster import DBSCAN  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def load_png(name):
    return np.asarray(Image.open(os.path.join(REF, "images/png", name)).convert("RGB"), dtype=np.uint8)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def seg_arrays(seg):
    """palette -> uint8[K,3], indices -> int64[h*w] (what downstream code reads)."""
    return (np.array(seg["palette"], dtype=np.int64).reshape(-1, 3).astype(np.uint8),
            np.array(seg["indices"], dtype=np.int64).reshape(-1))


def poster(h, w, seed, ncol=12):
    rng = np.random.default_rng(seed)
    cols = []
    while len(cols) < ncol:
        c = rng.integers(8, 248, 3)
        if all(np.abs(c - o).sum() >= 60 and np.linalg.norm(c - o) >= 60 for o in cols):
            cols.append(c)
    cols = np.array(cols, dtype=np.float64)
    lab = rng.integers(0, ncol, (max(2, h // 8), max(2, w // 8)))
    lab = np.kron(lab, np.ones((8, 8), dtype=int))[:h, :w]
    img = cols[lab]
    # 1-px blended edges
    blur = img.copy()
    blur[:, 1:] = 0.5 * (img[:, 1:] + img[:, :-1])
    edge = np.zeros((h, w), bool)
    edge[:, 1:] = lab[:, 1:] != lab[:, :-1]
    img[edge] = blur[edge]
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    versions = {"numpy": np.__version__, "scikit-learn": sklearn.__version__,
                "python": sys.version.split()[0],
                "reference_snapshot": "2026-01-16"}
    json.dump(versions, open(os.path.join(HERE, "versions.json"), "w"), indent=1)

    lenna = load_png("Lenna.png")
    kod1 = load_png("1.png")
    kod13 = load_png("13.png")
    kod23 = load_png("23.png")
    rng = np.random.default_rng(20260116)

    # ---------------- G1 get_all_unique_colors (clustering.py:4-103) ----------------
    crops = []
    c = lenna[100:132, 200:232].copy(); crops.append(c)                   # dense, no black
    c = kod23[300:324, 400:432].copy(); c[:5] = 0; c[:, :3] = 0; crops.append(c)  # black background
    c = rng.integers(0, 4, (16, 16, 3)).astype(np.uint8); crops.append(c)  # many duplicates incl. (0,0,0)
    c = np.zeros((7, 5, 3), np.uint8); crops.append(c)                     # all black, ragged shape
    c = lenna[0:1, 0:37].copy(); crops.append(c)                           # single row
    g1 = {}
    for i, c in enumerate(crops):
        d = quiet(R_clu.get_all_unique_colors, c, (3 + i, 7 * i))
        pal, idx = seg_arrays(d)
        g1[f"img{i}"] = c; g1[f"pal{i}"] = pal; g1[f"idx{i}"] = idx.astype(np.int32)
        g1[f"meta{i}"] = np.array([d["actual_colors"], d["shape"][0], d["shape"][1],
                                   d["top_left"][0], d["top_left"][1]])
        assert d["index_dtype"] in ("<class 'numpy.uint8'>", "<class 'numpy.uint16'>")
    g1["n"] = np.array(len(crops))
    save("g1_unique.npz", **g1)
    assert quiet(R_clu.get_all_unique_colors, np.zeros((0, 0, 3), np.uint8), (0, 0)) is None

    # ---------------- G2 compute_clustering_params (clustering.py:108-135) ----------------
    tab = []
    for n in (1, 2, 10, 255, 256, 257, 2534, 9999, 10000, 65536, 148279):
        for q in range(1, 101):
            eps, ms, mc = R_clu.compute_clustering_params(n, q, color_space="lab")
            tab.append([n, q, float(eps).hex(), int(ms), int(mc)])
    json.dump(tab, open(os.path.join(HERE, "g2_params.json"), "w"))
    print("wrote g2_params.json", len(tab))

    # ---------------- G3 DBSCAN labels at the reference call site (clustering.py:233-235) -------------
    # The reference evaluates DBSCAN(eps=eps/255.0, min_samples=1, metric='euclidean')
    #   .fit_predict(palette.astype(float)/255.0); labels are not returned by the reference function,
    # so the same expression is evaluated here on the same kind of input (sorted unique palettes).
    def uniq(a):
        a = np.unique(a.reshape(-1, 3), axis=0)
        return a[~np.all(a == 0, axis=1)]
    pals = {
        "rand50": uniq(rng.integers(0, 256, (50, 3)).astype(np.uint8)),
        "rand500": uniq(rng.integers(0, 256, (500, 3)).astype(np.uint8)),
        "rand3000": uniq(rng.integers(0, 256, (3000, 3)).astype(np.uint8)),
        "lattice": uniq((rng.integers(0, 9, (3000, 3)) * 32).clip(0, 255).astype(np.uint8)),  # exact-boundary pairs at eps=32,64,96
        "gapped": uniq(poster(64, 64, 5)),
        "lenna64": uniq(lenna[200:264, 200:264]),
        "dark": uniq(rng.integers(0, 6, (400, 3)).astype(np.uint8)),
    }
    # sparse already-quantised palettes (what levels 2/3 see): decode committed artefacts safely
    class Safe(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in (("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
                                  ("numpy.core.multiarray", "scalar")):
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"forbidden {module}.{name}")

    def safe_load(path):
        raw = open(path, "rb").read()
        assert raw[:5] == b"RHCCQ"
        n = struct.unpack("<I", raw[5:9])[0]
        return Safe(io.BytesIO(zlib.decompress(raw[9:9 + n]))).load()

    def decode(path):
        d = safe_load(path)
        h, w = int(d["s"][0]), int(d["s"][1])
        pal = np.frombuffer(zlib.decompress(d["p"]), np.uint8).reshape(-1, 3)
        idx = np.frombuffer(zlib.decompress(d["i"]), np.dtype(d.get("d", "uint16")))
        return pal, idx, (h, w), d
    lpal, _, _, _ = decode(os.path.join(REF, "images/rhccq_20_10/Lenna_compressed.rhccq"))
    pals["lenna_final"] = lpal[~np.all(lpal == 0, axis=1)]
    apal, _, _, _ = decode(os.path.join(REF, "images/rhccq/adidas_compressed.rhccq"))
    pals["adidas_final"] = apal[~np.all(apal == 0, axis=1)]
    eps_list = [1.0, 3.0, 12.8, 25.6, 32.0, 51.2, 64.0, 76.8, 96.0, 102.4, 115.2]
    g3 = {"eps": np.array(eps_list), "names": np.array(list(pals))}
    for name, P in pals.items():
        g3[f"pal_{name}"] = P
        for ei, eps in enumerate(eps_list):
            lab = DBSCAN(eps=eps / 255.0, min_samples=1, metric="euclidean").fit_predict(P.astype(float) / 255.0)
            assert lab.min() >= 0
            g3[f"lab_{name}_{ei}"] = lab.astype(np.int32)
    save("g3_dbscan.npz", **g3)

    # ---------------- G9 KMeans splitting at the reference call site (clustering.py:720-775) ----------
    g9 = {}
    k9 = 0
    for name, P, mcs in (("lenna64", pals["lenna64"], (150, 400, 1200)),
                         ("rand500", pals["rand500"], (20, 45, 200)),
                         ("lenna_final", pals["lenna_final"], (1, 3, 30)),
                         ("rand3000", pals["rand3000"], (120, 1000)),
                         ("dark", pals["dark"], (7, 40))):
        for mc in mcs:
            subs = quiet(R_clu.split_large_cluster, P, mc)
            # encode as label per input row (rows are unique so membership is recoverable)
            keyof = {tuple(r): i for i, r in enumerate(P.tolist())}
            lab = np.full(len(P), -1, np.int32)
            for si, s in enumerate(subs):
                for r in s.tolist():
                    lab[keyof[tuple(r)]] = si
            assert lab.min() >= 0
            g9[f"pal{k9}"] = P; g9[f"mc{k9}"] = np.array(mc); g9[f"lab{k9}"] = lab
            g9[f"name{k9}"] = np.array(name)
            k9 += 1
    g9["n"] = np.array(k9)
    save("g9_kmeans_split.npz", **g9)

    # ---------------- G4 cluster_palette_colors_parallel (clustering.py:160-437) ----------------
    g4 = {}
    k4 = 0
    imgs4 = [lenna[230:278, 230:278].copy(), kod1[100:148, 300:348].copy(), kod13[40:88, 40:88].copy(),
             poster(48, 48, 11), kod23[200:232, 200:248].copy(), lenna[300:316, 300:364].copy()]
    imgs4[1][:6] = 0           # black background rows
    imgs4[4][:, :9] = 0
    for ii, im in enumerate(imgs4):
        for q in (10, 20, 40, 60, 90, 100):
            d = quiet(R_clu.get_all_unique_colors, im, (0, 0))
            eps, ms, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
            o = quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=1, max_colors_per_cluster=mc)
            pal, idx = seg_arrays(o)
            g4[f"img{k4}"] = im; g4[f"q{k4}"] = np.array(q)
            g4[f"pal{k4}"] = pal; g4[f"idx{k4}"] = idx.astype(np.int32)
            k4 += 1
    g4["n"] = np.array(k4)
    save("g4_cluster.npz", **g4)

    # ---------------- G10 MiniBatchKMeans branch (clustering.py:207-230), N >= 10000 ----------------
    im10 = lenna[128:320, 128:320].copy()
    d = quiet(R_clu.get_all_unique_colors, im10, (0, 0))
    assert d["actual_colors"] >= 10000, d["actual_colors"]
    g10 = {"img": im10}
    for q in (10, 20):
        eps, ms, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
        o = quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=1, max_colors_per_cluster=mc)
        pal, idx = seg_arrays(o)
        g10[f"pal_q{q}"] = pal; g10[f"idx_q{q}"] = idx.astype(np.int32)
    save("g10_minibatch.npz", **g10)

    # ---------------- G5 merge_region_components_simple (merging.py:8-120) ----------------
    def mkseg(img, tl):
        return quiet(R_clu.get_all_unique_colors, img, tl)
    a = poster(12, 10, 1, 5); a[:3] = 0
    b = poster(9, 14, 2, 5); b[:, :4] = 0
    cimg = poster(8, 8, 3, 4)
    cases = {
        "overlap": ([mkseg(a, (2, 3)), mkseg(b, (5, 1)), mkseg(cimg, (0, 0))], (0, 0, 16, 16)),
        "out_of_canvas": ([mkseg(a, (-2, 4)), mkseg(b, (10, 8)), mkseg(cimg, (4, -3))], (0, 0, 14, 14)),
        "offset_bbox": ([mkseg(a, (12, 13)), mkseg(b, (15, 11))], (10, 10, 30, 30)),
        "single": ([mkseg(a, (2, 3))], (0, 0, 16, 16)),
    }
    g5 = {"names": np.array(list(cases))}
    for name, (comps, bbox) in cases.items():
        out = quiet(R_mrg.merge_region_components_simple, comps, bbox)
        g5[f"{name}_bbox"] = np.array(bbox); g5[f"{name}_n"] = np.array(len(comps))
        for ci, s in enumerate(comps):
            pal, idx = seg_arrays(s)
            g5[f"{name}_c{ci}_pal"] = pal; g5[f"{name}_c{ci}_idx"] = idx.astype(np.int32)
            g5[f"{name}_c{ci}_meta"] = np.array([s["top_left"][0], s["top_left"][1], s["shape"][0], s["shape"][1]])
        assert len(out) == 1
        pal, idx = seg_arrays(out[0])
        g5[f"{name}_out_pal"] = pal; g5[f"{name}_out_idx"] = idx.astype(np.int32)
        g5[f"{name}_out_meta"] = np.array([out[0]["top_left"][0], out[0]["top_left"][1],
                                           out[0]["shape"][0], out[0]["shape"][1], out[0]["actual_colors"]])
    assert quiet(R_mrg.merge_region_components_simple, [], (0, 0, 4, 4)) == []
    save("g5_merge.npz", **g5)

    # ---------------- G6 three-level chain (rhccq.ipynb:978-1039 flow; subregions.py:315-449 restated
    # as the same call sequence of importable functions on a fixed label map) ----------------
    def chain(img, seglab_roi, seglab_non, q_roi, q_non, tag, store):
        H, W = img.shape[:2]
        def level1(seglab, q):
            """one region per class with bbox = tight bbox of the class mask; segments = label ids."""
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
                pad = 2
                h, w = region_image.shape[:2]
                r0, r1 = max(0, r.min() - pad), min(h - 1, r.max() + pad)
                c0, c1 = max(0, c.min() - pad), min(w - 1, c.max() + pad)
                crop = region_image[r0:r1 + 1, c0:c1 + 1]
                mcrop = segment_mask[r0:r1 + 1, c0:c1 + 1]
                segimg = np.zeros_like(crop)
                segimg[mcrop] = crop[mcrop]
                # black-in-segment fix (subregions.py:393-421), same arithmetic
                px = crop[mcrop]
                isb = np.all(px == 0, axis=1)
                if isb.any() and (~isb).any():
                    nb = px[~isb]
                    for i in np.where(isb)[0]:
                        dist = np.linalg.norm(nb - px[i], axis=1)
                        px[i] = nb[np.argmin(dist)]
                    segimg[mcrop] = px
                d = quiet(R_clu.get_all_unique_colors, segimg, (int(r0 + minr), int(c0 + minc)))
                eps, ms, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
                comps.append(quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=1,
                                   max_colors_per_cluster=mc))
            if len(comps) > 1:
                return [quiet(R_mrg.merge_region_components_simple, comps, (int(minr), int(minc), int(maxr), int(maxc)))]
            return [comps]
        roi1 = level1(seglab_roi, q_roi)
        non1 = level1(seglab_non, q_non)
        q2r, q2n = min(q_roi * 2, 100), min(q_non * 2, 100)
        roi2 = quiet(R_reg.region_quantization, roi1, H, W, q2r)
        non2 = quiet(R_reg.region_quantization, non1, H, W, q2n)
        q3 = min(q2r + q2n, 100)
        fin = quiet(R_img.quantize_image, roi2 + non2, H, W, q3)
        for nm, s in (("roi1", roi1[0][0]), ("non1", non1[0][0]), ("roi2", roi2[0]), ("non2", non2[0]), ("fin", fin)):
            pal, idx = seg_arrays(s)
            store[f"{tag}_{nm}_pal"] = pal; store[f"{tag}_{nm}_idx"] = idx.astype(np.int32)
            store[f"{tag}_{nm}_meta"] = np.array([s["top_left"][0], s["top_left"][1], s["shape"][0], s["shape"][1]])
        store[f"{tag}_img"] = img; store[f"{tag}_lab_roi"] = seglab_roi.astype(np.int32)
        store[f"{tag}_lab_non"] = seglab_non.astype(np.int32)
        store[f"{tag}_q"] = np.array([q_roi, q_non])
        store[f"{tag}_fin_dtype"] = np.array(fin.get("indices_dtype", "none"))
        pk = quiet(R_cmp.lossless_compress_optimized, fin["palette"], fin["indices"], fin["shape"])
        return fin, pk

    g6 = {}
    def labelmaps(H, W, tiles, roi_frac=0.35, overlap=3):
        yy, xx = np.mgrid[0:H, 0:W]
        ell = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 <= roi_frac * 4 / np.pi
        # dilate roi by `overlap` px (square) to mimic extract_roi_nonroi's buffer (roi.py:685-718)
        dil = ell.copy()
        for _ in range(overlap):
            d2 = dil.copy()
            d2[1:] |= dil[:-1]; d2[:-1] |= dil[1:]; d2[:, 1:] |= dil[:, :-1]; d2[:, :-1] |= dil[:, 1:]
            dil = d2
        tid = (yy // (H // tiles)) * tiles + (xx // (W // tiles))
        return np.where(dil, tid, -1), np.where(~ell, tid, -1)
    im6 = lenna[220:284, 220:284].copy()
    lr, ln = labelmaps(64, 64, 2)
    fin, pk = chain(im6, lr, ln, 20, 10, "lenna64", g6)
    im6b = poster(64, 64, 21); im6b[10:14, 10:14] = 0      # black pixels inside segments -> K0b
    fin_b, pk_b = chain(im6b, lr, ln, 20, 10, "poster64", g6)
    im6c = kod23[256:352, 384:480].copy()
    lr3, ln3 = labelmaps(96, 96, 3)
    fin_c, pk_c = chain(im6c, lr3, ln3, 25, 25, "kodak96", g6)   # q=25 -> eps 96/64 integer boundaries
    g6["tags"] = np.array(["lenna64", "poster64", "kodak96"])
    save("g6_chain.npz", **g6)

    # ---------------- G7 container bytes (compression.py:119-220) ----------------
    tmp = os.path.join("/tmp", "g7_tmp.rhccq")
    quiet(R_cmp.save_compressed, pk, tmp)
    raw = open(tmp, "rb").read()
    os.remove(tmp)
    open(os.path.join(HERE, "g7_lenna64.rhccq"), "wb").write(raw)
    pal, idx = seg_arrays(fin)
    save("g7_container.npz", pal=pal, idx=idx.astype(np.int32), shape=np.array(fin["shape"]),
         p=np.frombuffer(pk["p"], np.uint8), i=np.frombuffer(pk["i"], np.uint8), d=np.array(pk["d"]), l=np.array(pk["l"]))
    p2, i2, s2 = quiet(R_dec.lossless_decompress, pk)
    assert np.array_equal(np.array(p2), pal) and i2 == idx.tolist()

    # ---------------- G8 decoder known-answer table over the 36 committed artefacts ----------------
    g8 = {}
    for sub in ("rhccq_20_10", "rhccq"):
        for fn in sorted(os.listdir(os.path.join(REF, "images", sub))):
            pal, idx, shape, d = decode(os.path.join(REF, "images", sub, fn))
            g8[f"{sub}/{fn}"] = {
                "shape": list(shape), "l": int(d["l"]), "d": d.get("d", "uint16"),
                "palette_sha256": hashlib.sha256(pal.tobytes()).hexdigest(),
                "indices_sha256": hashlib.sha256(idx.tobytes()).hexdigest(),
                "file_sha256": hashlib.sha256(open(os.path.join(REF, "images", sub, fn), "rb").read()).hexdigest(),
                "max_index": int(idx.max()),
            }
    json.dump(g8, open(os.path.join(HERE, "g8_rhccq_kat.json"), "w"), indent=1)
    print("wrote g8_rhccq_kat.json", len(g8))
    # small artefacts the reference ships as its de-facto regression data (data files, not source)
    for sub, fn in (("rhccq", "shell-logo_compressed.rhccq"), ("rhccq", "adidas_compressed.rhccq"),
                    ("rhccq_20_10", "compressed_23.rhccq")):
        open(os.path.join(HERE, fn), "wb").write(open(os.path.join(REF, "images", sub, fn), "rb").read())


if __name__ == "__main__":
    main()
