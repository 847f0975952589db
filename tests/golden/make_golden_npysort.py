#!/usr/bin/env python3
"""G15 / G11-scalar: the ONE unportable step of the path, pinned to numpy itself under a STATED host setting.
Build container only.

    python tests/golden/make_golden_npysort.py      # writes g15_npysort.npz, g11_scalar.npz, g11_scalar.json

scikit-learn's MiniBatchKMeans (reference call site encoder/compression/clustering.py:207-218) calls np.argsort on the
massively tied per-centre counts (_mini_batch_step); numpy's default argsort is unstable and has three kernels (x86-simd-sort
AVX-512 / AVX2, scalar introsort), so the fit depends on the host from k >= 500 on.  This script re-runs numpy and
scikit-learn UNTOUCHED (nothing patched) in a child process under

    NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"

i.e. with numpy's scalar sort kernels (what numpy runs on any host without AVX2) and records
  * G15: np.argsort(w) for tied count vectors like the ones of the path and for random float64 vectors (sizes 2 .. 30 128);
  * G10-scalar: the reference's own cluster_palette_colors_parallel on the G10 crop (q = 10 / 20) under that setting;
  * G11-scalar: for the 8 G11 inputs the k-means++ picks, n_steps_, cluster_centers_ (float64, bit for bit), the labels'
    sha256, cluster sizes, non-empty clusters and PSNR of sklearn's own fit_predict under that setting.
The oracle (oracle/npy_argsort.c inside oracle/mbk_oracle.c) and the HIP path must reproduce all of it exactly.
Only data is written; no reference or library source text is copied."""
import hashlib
import json
import math
import os
import subprocess
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SCALAR = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"


def worker(out_prefix):
    import contextlib
    import io
    sys.path.insert(0, "/root/reference")     # (the repository's own `encoder` package must not shadow the reference's here)
    import numpy as np
    with contextlib.redirect_stdout(io.StringIO()):
        from encoder.compression import clustering as R_clu
    sys.path.append(ROOT)
    import sklearn
    import sklearn.cluster._kmeans as km
    from PIL import Image
    from sklearn.cluster import MiniBatchKMeans
    from numpy._core._multiarray_umath import __cpu_features__ as feats
    from roibasedimagecompression_amd import synth
    assert not feats["AVX2"] and not feats["AVX512F"] and not feats["AVX512_SKX"], "scalar setting did not take"

    # ---- G15: argsort vectors
    rng = np.random.default_rng(2026)
    arrs = {}
    i = 0
    for n in list(range(2, 40)) + [64, 100, 257, 1000, 1536, 4097, 15809, 20771, 30128]:
        for trial in range(3 if n < 4097 else 2):
            if trial == 2:
                w = rng.random(n)                                            # no ties at all
            else:
                w = np.zeros(n)
                np.add.at(w, rng.integers(0, n, max(1, min(n, 1000 * (trial + 1)) // (2 - trial))), 1.0)   # counts of 1-2 batches
            arrs[f"w{i}"] = w.astype(np.float32) if trial < 2 else w         # the counts are small integers: exact in float32
            arrs[f"o{i}"] = np.argsort(w.astype(np.float64)).astype(np.int32)
            i += 1
    # adversarial for the depth limit: organ-pipe and sawtooth inputs
    for w in (np.concatenate([np.arange(3000), np.arange(3000)[::-1]]).astype(np.float64), (np.arange(9000) % 7).astype(np.float64),
              np.zeros(5000)):
        arrs[f"w{i}"] = w.astype(np.float32)
        arrs[f"o{i}"] = np.argsort(w).astype(np.int32)
        i += 1
    arrs["n"] = np.int64(i)
    np.savez_compressed(out_prefix + "g15_npysort.npz", **arrs)

    # ---- G11 under the scalar setting: sklearn untouched, only spied on
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
    meta = {"numpy": np.__version__, "scikit-learn": sklearn.__version__, "python": sys.version.split()[0],
            "NPY_DISABLE_CPU_FEATURES": os.environ.get("NPY_DISABLE_CPU_FEATURES"),
            "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
            "cases": {}}
    out = {}
    for name, mk, q in cases:
        img = mk()
        u = np.unique(img.reshape(-1, 3), axis=0)
        P = u[~np.all(u == 0, axis=1)]                                       # clustering.py:185-199
        n = len(P)
        k = math.ceil(n * (q / 100) / 10)
        cap = {}
        real_pp = km._kmeans_plusplus

        def spy(Xs, n_clusters, **kw):
            centers, indices = real_pp(Xs, n_clusters, **kw)
            cap["picks"] = np.asarray(indices).copy()
            return centers, indices
        km._kmeans_plusplus = spy
        try:
            m = MiniBatchKMeans(n_clusters=k, batch_size=1000, random_state=42, n_init="auto")
            lab = m.fit_predict(P.astype(np.float64)).astype(np.int32)
        finally:
            km._kmeans_plusplus = real_pp
        sizes = np.bincount(lab, minlength=k).astype(np.int32)
        err = P.astype(np.float64) - m.cluster_centers_[lab]
        psnr = 10 * math.log10(255.0 ** 2 / float(np.mean(err * err)))
        out[f"{name}_picks"] = cap["picks"].astype(np.int32)
        out[f"{name}_centres"] = m.cluster_centers_.astype(np.float64)
        out[f"{name}_sizes"] = sizes
        meta["cases"][name] = {"q": q, "n": int(n), "k": int(k), "n_steps": int(m.n_steps_),
                               "labels_sha256": hashlib.sha256(lab.tobytes()).hexdigest(),
                               "n_nonempty": int((sizes > 0).sum()), "inertia": float(m.inertia_), "psnr": psnr,
                               "img_sha256": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()}
        print(name, json.dumps(meta["cases"][name]), flush=True)
    # ---- G10 under the scalar setting: the reference's OWN function (cluster_palette_colors_parallel, MiniBatch branch) on the G10
    # crop -- at q = 20 (k = 659 >= 500) its capped reassignment runs np.argsort, so the fixture of record is the scalar one
    g10 = np.load(os.path.join(HERE, "g10_minibatch.npz"))
    with contextlib.redirect_stdout(io.StringIO()):
        d10 = R_clu.get_all_unique_colors(g10["img"], (0, 0))
        for q in (10, 20):
            eps, ms, mc = R_clu.compute_clustering_params(d10["actual_colors"], q, color_space="lab")
            o = R_clu.cluster_palette_colors_parallel(q, d10, eps=eps, min_samples=1, max_colors_per_cluster=mc)
            out[f"g10_pal_q{q}"] = np.array(o["palette"], dtype=np.int64).reshape(-1, 3).astype(np.uint8)
            out[f"g10_idx_q{q}"] = np.array(o["indices"], dtype=np.int64).reshape(-1).astype(np.int32)
    meta["g10"] = {f"q{q}": {"colours": int(len(out[f"g10_pal_q{q}"])), "equals_default_fixture": bool(
        np.array_equal(out[f"g10_pal_q{q}"], g10[f"pal_q{q}"]) and np.array_equal(out[f"g10_idx_q{q}"], g10[f"idx_q{q}"]))} for q in (10, 20)}
    print("g10", meta["g10"], flush=True)
    np.savez_compressed(out_prefix + "g11_scalar.npz", **out)
    json.dump(meta, open(out_prefix + "g11_scalar.json", "w"), indent=1)


def main():
    e = dict(os.environ)
    e["NPY_DISABLE_CPU_FEATURES"] = SCALAR
    e["PYTHONDONTWRITEBYTECODE"] = "1"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", HERE + os.sep], env=e, check=True)
    print("wrote g15_npysort.npz, g11_scalar.npz, g11_scalar.json")


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker(sys.argv[sys.argv.index("--worker") + 1])
    else:
        main()
