#!/usr/bin/env python3
"""Which golden cases does the REFERENCE itself reproduce on another host?  Build container only.

    python tests/golden/make_stability.py            # writes tests/golden/g_stability.json (+ g11_untouched.npz)

The k-means pieces of the path run inside scikit-learn (KMeans: clustering.py:751-752; MiniBatchKMeans: :211-218), whose
results depend on the host in two places: the summation order of OpenBLAS' kernels for the CPU type (k-means++ potentials
that tie exactly in integer arithmetic are told apart by rounding noise) and the tie order of numpy's unstable SIMD
argsort (the low-count reassignment of MiniBatchKMeans).  This script re-runs the reference on the INPUTS of the committed
fixtures (G4, G6, G9, G10, G13 through the reference's own functions; G11 through scikit-learn at the reference's call
site) in child processes under five host settings

    default | OPENBLAS_CORETYPE=Sandybridge | NPY_DISABLE_CPU_FEATURES=<AVX512 family> | both |
    NPY_DISABLE_CPU_FEATURES=<AVX512 family> AVX2 FMA3   (numpy's scalar sort kernels: the setting of record since round 4,
                                                          restated in oracle/npy_argsort.c -- make_golden_npysort.py)

and records per case a hash of the output under each setting, whether all of them agree (`reference_stable`) and whether the
default run equals the committed fixture (`default_equals_fixture`: the fixtures are genuine).  The tests then demand
bit-exactness (Tier A, or A' = same pixels under a permuted palette, the reference's own thread-completion order) on every
reference-stable case and report Tier-B deltas only where the reference itself is not reproducible.
For G11 the UNTOUCHED MiniBatchKMeans fit (no forced stable argsort) is recorded too: its labels / step count per setting,
and -- where all settings agree -- the labels themselves (g11_untouched.npz), to compare the HIP path with directly.
Only data is written; no reference source text is copied."""
import contextlib
import hashlib
import io
import json
import math
import os
import subprocess
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"

AVX512 = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
SETTINGS = {
    "default": {},
    "openblas_sandybridge": {"OPENBLAS_CORETYPE": "Sandybridge"},
    "numpy_no_avx512": {"NPY_DISABLE_CPU_FEATURES": AVX512},
    "both": {"OPENBLAS_CORETYPE": "Sandybridge", "NPY_DISABLE_CPU_FEATURES": AVX512},
    "numpy_scalar": {"NPY_DISABLE_CPU_FEATURES": AVX512 + " AVX2 FMA3"},
}


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(a.tobytes())
    return h.hexdigest()[:16]


def worker():
    sys.path.insert(0, REF)          # (the repository's own `encoder` package must NOT be importable here: a regular package
    import numpy as np               #  would win over the reference's namespace package; ROOT is appended further down)
    from PIL import Image
    with contextlib.redirect_stdout(io.StringIO()):
        from encoder.compression import clustering as R_clu
        from encoder.compression import merging as R_mrg
        from encoder.compression import regions as R_reg
        from encoder.compression import image as R_img

    def quiet(fn, *a, **k):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*a, **k)

    def seg_arrays(seg):
        return (np.array(seg["palette"], dtype=np.int64).reshape(-1, 3).astype(np.uint8),
                np.array(seg["indices"], dtype=np.int64).reshape(-1))

    def out_hashes(pal, idx):
        """exact = palette + indices; pixels = the decoded colours only (equal under a permuted palette: tier A')"""
        pal = np.asarray(pal).reshape(-1, 3)
        idx = np.asarray(idx).reshape(-1)
        px = pal[np.minimum(idx, len(pal) - 1)] if len(pal) else np.zeros((0, 3), np.uint8)
        return {"exact": sha(pal.astype(np.uint8), idx.astype(np.int64)), "pixels": sha(px.astype(np.uint8)), "colours": int(len(pal))}

    res = {}
    # ---- G4: cluster_palette_colors_parallel (clustering.py:160-437)
    g4 = np.load(os.path.join(HERE, "g4_cluster.npz"))
    for k in range(int(g4["n"])):
        im, q = g4[f"img{k}"], int(g4[f"q{k}"])
        d = quiet(R_clu.get_all_unique_colors, im, (0, 0))
        eps, ms, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
        o = quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=1, max_colors_per_cluster=mc)
        h = out_hashes(*seg_arrays(o))
        h["fixture"] = out_hashes(g4[f"pal{k}"], g4[f"idx{k}"])
        res[f"g4/{k}"] = h
    # ---- G9: split_large_cluster (clustering.py:720-775)
    g9 = np.load(os.path.join(HERE, "g9_kmeans_split.npz"))
    for k in range(int(g9["n"])):
        P, mc = g9[f"pal{k}"], int(g9[f"mc{k}"])
        subs = quiet(R_clu.split_large_cluster, P, mc)
        keyof = {tuple(r): i for i, r in enumerate(P.tolist())}
        lab = np.full(len(P), -1, np.int32)
        for si, s in enumerate(subs):
            for r in s.tolist():
                lab[keyof[tuple(r)]] = si
        res[f"g9/{k}"] = {"exact": sha(lab), "pixels": sha(lab), "colours": int(lab.max()) + 1,
                          "fixture": {"exact": sha(g9[f"lab{k}"].astype(np.int32)), "pixels": sha(g9[f"lab{k}"].astype(np.int32))}}
    # ---- G13: the same function with min_samples > 1
    g13 = np.load(os.path.join(HERE, "g13_dbscan_min_samples.npz"))
    for k in range(int(g13["n"])):
        im = g13[f"img{k}"]
        q, ms = (int(v) for v in g13[f"qm{k}"])
        d = quiet(R_clu.get_all_unique_colors, im, (0, 0))
        eps, _, mc = R_clu.compute_clustering_params(d["actual_colors"], q, color_space="lab")
        o = quiet(R_clu.cluster_palette_colors_parallel, q, d, eps=eps, min_samples=ms, max_colors_per_cluster=mc)
        h = out_hashes(*seg_arrays(o))
        h["fixture"] = out_hashes(g13[f"pal{k}"], g13[f"idx{k}"])
        res[f"g13/{k}"] = h
    # ---- G10: MiniBatch branch of the reference's function
    g10 = np.load(os.path.join(HERE, "g10_minibatch.npz"))
    d10 = quiet(R_clu.get_all_unique_colors, g10["img"], (0, 0))
    for q in (10, 20):
        eps, ms, mc = R_clu.compute_clustering_params(d10["actual_colors"], q, color_space="lab")
        o = quiet(R_clu.cluster_palette_colors_parallel, q, d10, eps=eps, min_samples=1, max_colors_per_cluster=mc)
        h = out_hashes(*seg_arrays(o))
        h["fixture"] = out_hashes(g10[f"pal_q{q}"], g10[f"idx_q{q}"])
        res[f"g10/q{q}"] = h
    # ---- G6: the three-level chain, as make_golden.py runs it (same call sequence on the fixture's label maps)
    g6 = np.load(os.path.join(HERE, "g6_chain.npz"))
    for tag in (str(t) for t in g6["tags"]):
        img, lr, ln = g6[f"{tag}_img"], g6[f"{tag}_lab_roi"], g6[f"{tag}_lab_non"]
        q_roi, q_non = (int(v) for v in g6[f"{tag}_q"])
        H, W = img.shape[:2]

        def level1(seglab, q):
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
                segimg[mcrop] = crop[mcrop]
                px = crop[mcrop]
                isb = np.all(px == 0, axis=1)
                if isb.any() and (~isb).any():
                    nb = px[~isb]
                    for i in np.where(isb)[0]:
                        px[i] = nb[np.argmin(np.linalg.norm(nb - px[i], axis=1))]
                    segimg[mcrop] = px
                dd = quiet(R_clu.get_all_unique_colors, segimg, (int(r0 + minr), int(c0 + minc)))
                eps, ms, mc = R_clu.compute_clustering_params(dd["actual_colors"], q, color_space="lab")
                comps.append(quiet(R_clu.cluster_palette_colors_parallel, q, dd, eps=eps, min_samples=1, max_colors_per_cluster=mc))
            if len(comps) > 1:
                return [quiet(R_mrg.merge_region_components_simple, comps, (int(minr), int(minc), int(maxr), int(maxc)))]
            return [comps]
        roi1, non1 = level1(lr, q_roi), level1(ln, q_non)
        q2r, q2n = min(q_roi * 2, 100), min(q_non * 2, 100)
        roi2 = quiet(R_reg.region_quantization, roi1, H, W, q2r)
        non2 = quiet(R_reg.region_quantization, non1, H, W, q2n)
        fin = quiet(R_img.quantize_image, roi2 + non2, H, W, min(q2r + q2n, 100))
        for nm, s in (("roi1", roi1[0][0]), ("non1", non1[0][0]), ("roi2", roi2[0]), ("non2", non2[0]), ("fin", fin)):
            h = out_hashes(*seg_arrays(s))
            h["fixture"] = out_hashes(g6[f"{tag}_{nm}_pal"], g6[f"{tag}_{nm}_idx"])
            res[f"g6/{tag}/{nm}"] = h
    # ---- G11: scikit-learn's MiniBatchKMeans at the reference's call site, UNTOUCHED (no forced stable argsort)
    from sklearn.cluster import MiniBatchKMeans
    sys.path.append(ROOT)            # behind the reference, whose `encoder` / `decoder` modules are already loaded
    from roibasedimagecompression_amd import synth

    def png(name):
        return np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"), dtype=np.uint8)
    cases = [
        ("lenna192_q20", lambda: png("Lenna.png")[128:320, 128:320], 20),
        ("lenna192_q10", lambda: png("Lenna.png")[128:320, 128:320], 10),
        ("lenna_full_q20", lambda: png("Lenna.png"), 20),
        ("kodak1_q20", lambda: png("kodak_1.png"), 20),
        ("kodak13_q10", lambda: png("kodak_13.png"), 10),
        ("kodak23_q20", lambda: png("kodak_23.png"), 20),
        ("synth_photo_1024_q20", lambda: synth.photo(1024, 1024, 1234), 20),
        ("synth_photo_640_q40", lambda: synth.photo(640, 640, 1235, sigma=3.0), 40),
    ]
    want_labels = os.environ.get("RHCCQ_STABILITY_LABELS")
    keep = {}
    for name, mk, q in cases:
        u = np.unique(mk().reshape(-1, 3), axis=0)
        P = u[~np.all(u == 0, axis=1)]
        k = math.ceil(len(P) * (q / 100) / 10)
        m = MiniBatchKMeans(n_clusters=k, batch_size=1000, random_state=42, n_init="auto")
        lab = m.fit_predict(P.astype(np.float64)).astype(np.int32)
        err = P.astype(np.float64) - m.cluster_centers_[lab]
        res[f"g11/{name}"] = {"exact": sha(lab, m.cluster_centers_), "pixels": sha(lab), "colours": int((np.bincount(lab, minlength=k) > 0).sum()),
                              "n_steps": int(m.n_steps_), "k": int(k), "psnr": 10 * math.log10(255.0 ** 2 / float(np.mean(err * err)))}
        keep[name] = lab
    if want_labels:
        np.savez_compressed(want_labels, **keep)
    json.dump(res, sys.stdout)


def main():
    runs = {}
    tmp_labels = "/tmp/g11_untouched_default.npz"
    for name, env in SETTINGS.items():
        e = dict(os.environ)
        e.update(env)
        e["PYTHONDONTWRITEBYTECODE"] = "1"
        if name == "default":
            e["RHCCQ_STABILITY_LABELS"] = tmp_labels
        print("running the reference under", name, env, flush=True)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=e, check=True, stdout=subprocess.PIPE).stdout
        runs[name] = json.loads(out)
    import numpy as np
    import sklearn
    cases = {}
    for key in runs["default"]:
        per = {s: runs[s][key] for s in SETTINGS}
        ex = {s: per[s]["exact"] for s in SETTINGS}
        pxs = {s: per[s]["pixels"] for s in SETTINGS}
        d = per["default"]
        c = {"reference_stable": len(set(ex.values())) == 1,
             "reference_stable_pixels": len(set(pxs.values())) == 1,       # ... up to the palette order
             "exact": ex, "colours": {s: per[s]["colours"] for s in SETTINGS}}
        if "fixture" in d:
            c["default_equals_fixture"] = d["exact"] == d["fixture"]["exact"]
            c["default_equals_fixture_pixels"] = d["pixels"] == d["fixture"]["pixels"]
        if "n_steps" in d:
            c["n_steps"] = {s: per[s]["n_steps"] for s in SETTINGS}
            c["psnr"] = {s: per[s]["psnr"] for s in SETTINGS}
            c["k"] = d["k"]
        cases[key] = c
    # G11: where the untouched fit is the same under every setting, keep its labels (sha + step count in the json, arrays in the npz)
    lab = np.load(tmp_labels)
    stable11 = {k.split("/", 1)[1]: v for k, v in cases.items() if k.startswith("g11/") and v["reference_stable"]}
    np.savez_compressed(os.path.join(HERE, "g11_untouched.npz"), **{n: lab[n] for n in stable11})
    meta = {"numpy": np.__version__, "scikit-learn": sklearn.__version__, "python": sys.version.split()[0],
            "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
            "settings": SETTINGS, "cases": cases}
    json.dump(meta, open(os.path.join(HERE, "g_stability.json"), "w"), indent=1)
    for fam in ("g4", "g6", "g9", "g10", "g11", "g13"):
        ks = [k for k in cases if k.startswith(fam + "/")]
        unstable = [k.split("/", 1)[1] for k in ks if not cases[k]["reference_stable"]]
        print(fam, len(ks), "cases; not reproducible across settings:", unstable)
    print("wrote g_stability.json, g11_untouched.npz")


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker()
    else:
        main()
