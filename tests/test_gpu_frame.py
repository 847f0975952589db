"""Fused frame encoder (HIP path) vs the oracle's three-level chain and vs the reference's golden
outputs.  GPU only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def rh():
    from roibasedimagecompression_amd.ops import Rhccq
    return Rhccq(0)


def specs_from_labels(rh, lab_list, qualities):
    """one region per class, bbox = tight bbox of the class mask; lab: 0 = none, ids 1.."""
    import torch
    from roibasedimagecompression_amd.frame import ClassSpec
    specs, oracle_classes = [], []
    for lab, q in zip(lab_list, qualities):
        mask = lab > 0
        rows, cols = np.where(mask)
        bbox = (int(rows.min()), int(cols.min()), int(rows.max()) + 1, int(cols.max()) + 1)
        n_seg = int(lab.max())
        specs.append(ClassSpec(torch.from_numpy(lab.astype(np.int32)).to(rh.device), np.zeros(n_seg, np.int64), [bbox], q))
        sl = (slice(bbox[0], bbox[2]), slice(bbox[1], bbox[3]))
        oracle_classes.append([{"bbox": bbox, "bbox_mask": mask[sl], "seglabels": lab[sl].astype(np.int32)}])
    return specs, oracle_classes


def same_result(a, b):
    import torch
    return (np.array_equal(a["palette"], b["palette"]) and a["indices_dtype"] == b["indices_dtype"] and torch.equal(a["indices"], b["indices"])
            and tuple(a["shape"]) == tuple(b["shape"]) and tuple(a["top_left"]) == tuple(b["top_left"]) and np.array_equal(a["n_unique"], b["n_unique"]))


def run_both(rh, img, lab_list, qualities):
    """the Python host (FrameEncoder.encode) against the oracle's chain; the NATIVE host (rhccq_encode_frame, csrc/encode_frame.hip) must
    give the Python host's result bit for bit on every frame that comes through here (all 32 fuzz frames among them)"""
    import torch
    from oracle import rhccq_oracle as O
    from roibasedimagecompression_amd.frame import FrameEncoder
    specs, oc = specs_from_labels(rh, lab_list, qualities)
    enc = FrameEncoder(rh)
    rgb = torch.from_numpy(img.copy()).to(rh.device)
    out = enc.encode(rgb, specs)
    nat = enc.encode_native(rgb, specs)
    assert same_result(out, nat), "rhccq_encode_frame differs from FrameEncoder.encode"
    ref = O.encode_frame(img, oc, qualities)
    return out, ref


def indices_np(out):
    idx = out["indices"].cpu().numpy()
    if out["indices_dtype"] == "uint16":
        idx = idx.view(np.uint16)
    return idx.astype(np.int64)


@pytest.mark.parametrize("tag", ["lenna64", "poster64", "kodak96"])
def test_frame_vs_oracle_and_golden(rh, tag):
    g = np.load(os.path.join(G, "g6_chain.npz"))
    img = g[f"{tag}_img"]
    labs = [g[f"{tag}_lab_roi"] + 1, g[f"{tag}_lab_non"] + 1]
    qs = [int(v) for v in g[f"{tag}_q"]]
    out, ref = run_both(rh, img, labs, qs)
    fin = ref["final"]
    # HIP path == oracle, bit for bit (palette, indices, dtype)
    assert np.array_equal(out["palette"], np.asarray(fin["palette"]).reshape(-1, 3))
    assert np.array_equal(indices_np(out).reshape(-1), np.asarray(fin["indices"]).reshape(-1))
    assert out["indices_dtype"] == fin["indices_dtype"]
    # vs the reference's own output: Tier A when the k-means partitions reproduce, else Tier B
    gp, gi = g[f"{tag}_fin_pal"], g[f"{tag}_fin_idx"]
    exact = np.array_equal(out["palette"], gp) and np.array_equal(indices_np(out).reshape(-1), gi)
    rec = out["palette"][indices_np(out)].reshape(img.shape).astype(np.float64)
    gref = gp[gi].reshape(img.shape).astype(np.float64)
    p_mine = 10 * np.log10(255 ** 2 / np.mean((rec - img) ** 2))
    p_ref = 10 * np.log10(255 ** 2 / np.mean((gref - img) ** 2))
    print(tag, "Tier A" if exact else "Tier B", p_mine, p_ref)
    assert abs(p_mine - p_ref) < 0.5
    if tag == "poster64":
        assert exact


def test_frame_many_segments_photo_and_poster(rh):
    """cfg1-like: 256x256, 8x8 grid of segments per class, (20,10) preset; black pixels inside
    segments; HIP path == oracle bit for bit."""
    from roibasedimagecompression_amd import synth
    H = W = 256
    for name, img in (("photo", synth.photo(H, W, 1234)), ("poster", synth.poster(H, W, 1235))):
        img = img.copy()
        img[40:44, 100:140] = 0                                  # in-segment black -> K0b
        (lr, nr, _), (ln, nn, _) = synth.frame_classes(H, W, (8, 8))
        out, ref = run_both(rh, img, [lr, ln], [20, 10])
        fin = ref["final"]
        assert np.array_equal(out["palette"], np.asarray(fin["palette"]).reshape(-1, 3)), name
        assert np.array_equal(indices_np(out).reshape(-1), np.asarray(fin["indices"]).reshape(-1)), name


def test_frame_minibatch_branch_and_single_class(rh):
    """A segment with >= 10 000 colours takes the MiniBatchKMeans branch; one class only ->
    level-3 passthrough of a single component."""
    from roibasedimagecompression_amd import synth
    H, W = 192, 256
    img = synth.photo(H, W, 77, sigma=6.0)
    lab = np.ones((H, W), np.int32)
    lab[:, W // 2:] = 2
    out, ref = run_both(rh, img, [lab], [20])
    fin = ref["final"]
    assert (out["n_unique"] >= 10000).any()
    assert np.array_equal(out["palette"], np.asarray(fin["palette"]).reshape(-1, 3))
    assert np.array_equal(indices_np(out).reshape(-1), np.asarray(fin["indices"]).reshape(-1))


def test_frame_several_regions_per_class(rh):
    """Two connected regions per class (extract_regions yields one dict per connected region), one of them a
    single-segment region (kept as a component, subregions.py:675-679), with overlapping region boxes."""
    import torch
    from oracle import rhccq_oracle as O
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    H, W = 120, 160
    img = synth.photo(H, W, 99)
    img[30:33, 20:40] = 0
    yy, xx = np.mgrid[0:H, 0:W]
    roi_a = ((yy - 35) ** 2 + (xx - 45) ** 2) <= 28 ** 2            # region 0 of the ROI class (2 segments)
    roi_b = ((yy - 85) ** 2 / 0.6 + (xx - 120) ** 2) <= 26 ** 2      # region 1 (1 segment)
    non = ~(roi_a | roi_b)
    non_a = non & (xx < 90)                                         # two non-ROI regions split by a vertical cut
    non_b = non & (xx >= 90)
    specs, oracle_classes = [], []
    for masks, q, seg_rule in (((roi_a, roi_b), 20, (2, 1)), ((non_a, non_b), 10, (3, 2))):
        lab = np.zeros((H, W), np.int32)
        seg_region, boxes, oc = [], [], []
        nxt = 0
        for ri, (m, nseg) in enumerate(zip(masks, seg_rule)):
            rows, cols = np.where(m)
            bbox = (int(rows.min()), int(cols.min()), int(rows.max()) + 1, int(cols.max()) + 1)
            boxes.append(bbox)
            band = ((yy - bbox[0]) * nseg) // (bbox[2] - bbox[0])      # nseg horizontal bands inside the region
            local = np.where(m, band + 1, 0)
            for s_id in range(1, nseg + 1):
                if (local == s_id).any():
                    nxt += 1
                    lab[local == s_id] = nxt
                    seg_region.append(ri)
            sl = (slice(bbox[0], bbox[2]), slice(bbox[1], bbox[3]))
            oc.append({"bbox": bbox, "bbox_mask": m[sl], "seglabels": local[sl].astype(np.int32)})
        specs.append(ClassSpec(torch.from_numpy(lab).to(rh.device), seg_region, boxes, q))
        oracle_classes.append(oc)
    out = FrameEncoder(rh).encode(torch.from_numpy(img).to(rh.device), specs)
    ref = O.encode_frame(img, oracle_classes, [20, 10])["final"]
    assert np.array_equal(out["palette"], np.asarray(ref["palette"]).reshape(-1, 3))
    assert np.array_equal(indices_np(out).reshape(-1), np.asarray(ref["indices"]).reshape(-1))


def test_encode_batch_equals_frame_by_frame(rh):
    """several frames through one batched clustering call per level == each frame on its own"""
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    frames = []
    for seed, (H, W), tiles in ((1, (96, 128), (2, 2)), (2, (128, 96), (3, 2)), (3, (192, 256), (1, 2))):
        img = synth.photo(H, W, seed, sigma=5.0 if seed == 3 else 2.0)
        (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
        specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
                 ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
        frames.append((torch.from_numpy(img).to(rh.device), specs))
    enc = FrameEncoder(rh)
    single = [enc.encode(rgb, specs) for rgb, specs in frames]
    batch = enc.encode_batch(frames)
    for a, b in zip(single, batch):
        assert np.array_equal(a["palette"], b["palette"]) and a["indices_dtype"] == b["indices_dtype"]
        assert torch.equal(a["indices"], b["indices"])


def test_stream_encoder_equals_frame_by_frame(rh):
    """batches in flight on two lanes (host threads with their own HIP stream and context) == each frame alone"""
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    from roibasedimagecompression_amd.stream import StreamEncoder
    frames = []
    for seed in range(7):
        H, W = (96, 128) if seed % 2 else (160, 192)
        img = synth.photo(H, W, 40 + seed, sigma=5.0 if seed == 3 else 2.0)
        (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (2, 2))
        specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
                 ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
        frames.append((torch.from_numpy(img).to(rh.device), specs))
    enc = FrameEncoder(rh)
    single = [enc.encode(rgb, specs) for rgb, specs in frames]
    se = StreamEncoder(0, batch=2, lanes=2)
    got = se.run(frames)
    again = se.run(frames[:3])                                   # lanes (streams, contexts) are reused across runs
    se.close()
    assert len(got) == len(single) and len(again) == 3
    for a, b in zip(single, again):
        assert np.array_equal(a["palette"], b["palette"]) and torch.equal(a["indices"], b["indices"])
    for a, b in zip(single, got):
        assert np.array_equal(a["palette"], b["palette"]) and a["indices_dtype"] == b["indices_dtype"]
        assert torch.equal(a["indices"], b["indices"])


@pytest.mark.parametrize("seed", range(32))
def test_frame_fuzz_vs_oracle(rh, seed):
    """Randomised frames (size, photo / poster mix, noise, black patches, segment grid, one or two quality tiers,
    ROI shape): the fused HIP encoder equals the oracle's chain bit for bit."""
    from roibasedimagecompression_amd import synth
    rng = np.random.default_rng(1000 + seed)
    big = seed >= 24                                              # a few larger frames: MiniBatch-branch segments
    H, W = int(rng.integers(40, 110) * (2 if big else 1)), int(rng.integers(40, 130) * (2 if big else 1))
    sigma = float(rng.choice([0.0, 1.0, 2.0, 6.0]))
    img = synth.photo(H, W, 500 + seed, sigma=sigma) if rng.random() < 0.7 else synth.poster(H, W, 500 + seed)
    img = img.copy()
    if rng.random() < 0.5:                                        # poster island inside a photo: gapped palette
        y0, x0 = int(rng.integers(0, H - 20)), int(rng.integers(0, W - 20))
        img[y0:y0 + 20, x0:x0 + 20] = synth.poster(20, 20, seed)
    for _ in range(int(rng.integers(0, 3))):                      # black patches (in-segment black fix, all-black segments)
        y0, x0 = int(rng.integers(0, H - 6)), int(rng.integers(0, W - 6))
        img[y0:y0 + int(rng.integers(1, 6)), x0:x0 + int(rng.integers(1, 6))] = 0
    tiles = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
    (lr, nr, _), (ln, nn, _) = synth.frame_classes(H, W, tiles)
    q = [int(rng.choice([5, 10, 20, 35, 50])), int(rng.choice([5, 10, 20, 35, 50]))]
    labs = [lr, ln]
    if rng.random() < 0.2:                                        # a single class: level-3 passthrough
        labs, q = [np.where((lr > 0) | (ln > 0), np.maximum(lr, ln), 0).astype(np.int32)], q[:1]
    out, ref = run_both(rh, img, labs, q)
    fin = ref["final"]
    assert np.array_equal(out["palette"], np.asarray(fin["palette"]).reshape(-1, 3)), (seed, H, W, tiles, q)
    assert np.array_equal(indices_np(out).reshape(-1), np.asarray(fin["indices"]).reshape(-1)), (seed, H, W, tiles, q)
    assert out["indices_dtype"] == fin["indices_dtype"]


def test_table_overflow_falls_back_to_the_serial_path(rh, monkeypatch):
    """The pipelined encoder reserves a slice of the frame-wide tables per class from a heuristic bound of the clustered palette
    sizes; a class that outgrows it raises _TableOverflow while the other class keeps writing the shared tables, and encode()
    redoes the levels serially.  Forced here by shrinking the bound: the result must equal the pipelined result,
    the serial result (PIPELINE_CLASSES = False) and the oracle, and the half-written per-pixel tables must be gone."""
    import torch
    from oracle import rhccq_oracle as O
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import FrameEncoder
    H, W = 200, 240
    img = synth.photo(H, W, 4242, sigma=6.0).copy()
    img[50:53, 60:90] = 0
    (lr, nr, _), (ln, nn, _) = synth.frame_classes(H, W, (1, 1))
    specs, oc = specs_from_labels(rh, [lr, ln], [20, 10])
    rgb = torch.from_numpy(img).to(rh.device)
    ref = O.encode_frame(img, oc, [20, 10])["final"]
    outs = {}
    enc = FrameEncoder(rh)
    outs["pipelined"] = enc.encode(rgb, specs)
    assert (outs["pipelined"]["n_unique"] >= 10000).any()          # MiniBatch-branch jobs: the only ones with a bound below the palette
    assert getattr(enc, "table_overflows", 0) == 0
    monkeypatch.setattr(FrameEncoder, "TABLE_BOUND", (0, 8))
    enc2 = FrameEncoder(rh)
    outs["overflow"] = enc2.encode(rgb, specs)
    assert enc2.table_overflows == 1
    monkeypatch.undo()
    monkeypatch.setattr(FrameEncoder, "PIPELINE_CLASSES", False)
    outs["serial"] = FrameEncoder(rh).encode(rgb, specs)
    for name, out in outs.items():
        assert np.array_equal(out["palette"], np.asarray(ref["palette"]).reshape(-1, 3)), name
        assert np.array_equal(indices_np(out).reshape(-1), np.asarray(ref["indices"]).reshape(-1)), name


@pytest.mark.parametrize("seed", [3, 11, 25, 28])
def test_serial_classes_equal_pipelined_classes(rh, seed, monkeypatch):
    """PIPELINE_CLASSES = False (what the tiled encoder and single-class frames run) against the class pipelines on fuzz frames"""
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import FrameEncoder
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(60, 160)), int(rng.integers(60, 200))
    img = synth.photo(H, W, 500 + seed, sigma=float(rng.choice([0.0, 2.0, 6.0]))).copy()
    img[10:13, 20:40] = 0
    (lr, nr, _), (ln, nn, _) = synth.frame_classes(H, W, (int(rng.integers(1, 4)), int(rng.integers(1, 4))))
    specs, _ = specs_from_labels(rh, [lr, ln], [20, 10])
    rgb = torch.from_numpy(img).to(rh.device)
    a = FrameEncoder(rh).encode(rgb, specs)
    monkeypatch.setattr(FrameEncoder, "PIPELINE_CLASSES", False)
    b = FrameEncoder(rh).encode(rgb, specs)
    assert np.array_equal(a["palette"], b["palette"]) and a["indices_dtype"] == b["indices_dtype"]
    assert torch.equal(a["indices"], b["indices"])


@pytest.mark.parametrize("seed", range(12))
def test_sort_based_unique_path_equals_bitmap_path_and_oracle(rh, seed):
    """Frames cut into very many segments take their unique colours from ONE device sort of (job, colour) keys and keep the rank of
    every pixel's colour (rhccq_job_sort_unique, rhccq_job_index_ranked, rhccq_frame_remap_ranked) instead of 6 MiB of bitmap tables
    per job.  Forced here on the fuzz generator's frames (black patches -> recoloured in-mask black and all-black segments, crops
    that show background, one or two tiers, single class): palette and every index equal the bitmap path's and the oracle's."""
    import torch
    from oracle import rhccq_oracle as O
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import FrameEncoder
    rng = np.random.default_rng(7000 + seed)
    H, W = int(rng.integers(40, 150)), int(rng.integers(40, 170))
    img = (synth.photo(H, W, 900 + seed, sigma=float(rng.choice([0.0, 2.0, 6.0]))) if rng.random() < 0.7 else synth.poster(H, W, 900 + seed)).copy()
    for _ in range(int(rng.integers(1, 4))):
        y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
        img[y0:y0 + int(rng.integers(1, 8)), x0:x0 + int(rng.integers(1, 8))] = 0
    tiles = (int(rng.integers(1, 7)), int(rng.integers(1, 7)))
    (lr, nr, _), (ln, nn, _) = synth.frame_classes(H, W, tiles)
    q = [int(rng.choice([10, 20, 50])), int(rng.choice([5, 10, 20]))]
    labs = [lr, ln]
    if seed % 5 == 4:
        labs, q = [np.where((lr > 0) | (ln > 0), np.maximum(lr, ln), 0).astype(np.int32)], q[:1]
    specs, oc = specs_from_labels(rh, labs, q)
    rgb = torch.from_numpy(img).to(rh.device)
    a = FrameEncoder(rh).encode(rgb, specs)
    enc = FrameEncoder(rh)
    enc.SORT_UNIQUE_MIN_JOBS = 0                             # every frame through the sort
    b = enc.encode(rgb, specs)
    assert np.array_equal(a["palette"], b["palette"]) and torch.equal(a["indices"], b["indices"]) and a["indices_dtype"] == b["indices_dtype"], (seed, tiles)
    assert np.array_equal(a["n_unique"], b["n_unique"])
    fin = O.encode_frame(img, oc, q)["final"]
    assert np.array_equal(b["palette"], np.asarray(fin["palette"]).reshape(-1, 3)) and np.array_equal(indices_np(b).reshape(-1), np.asarray(fin["indices"]).reshape(-1))


def test_fine_grid_4k_frame_through_the_sort_path(rh):
    """A 3840x2160 frame cut into a 60 x 64 grid (thousands of jobs at 6 MiB of bitmap tables each; one 0.4 GB sort instead): the default
    encoder switches to the sort path by itself; size-independent properties (determinism, indices inside the palette, every entry
    used, PSNR) and, on a 1080p frame with 1 536 jobs where both paths fit comfortably, equality with the bitmap path."""
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder

    def frame(H, W, seed, tiles):
        img = synth.photo(H, W, seed)
        (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
        specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
                 ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
        return img, torch.from_numpy(img).to(rh.device), specs
    img, rgb, specs = frame(1080, 1920, 31, (24, 32))
    assert sum(sp.n_seg for sp in specs) <= FrameEncoder.SORT_UNIQUE_MIN_JOBS
    a = FrameEncoder(rh).encode(rgb, specs)
    enc = FrameEncoder(rh)
    enc.SORT_UNIQUE_MIN_JOBS = 0
    b = enc.encode(rgb, specs)
    assert np.array_equal(a["palette"], b["palette"]) and torch.equal(a["indices"], b["indices"])
    img, rgb, specs = frame(2160, 3840, 32, (60, 64))
    assert sum(sp.n_seg for sp in specs) > FrameEncoder.SORT_UNIQUE_MIN_JOBS
    enc = FrameEncoder(rh)
    o1 = enc.encode(rgb, specs)
    o2 = enc.encode(rgb, specs)
    assert np.array_equal(o1["palette"], o2["palette"]) and torch.equal(o1["indices"], o2["indices"])
    pal, idx = np.asarray(o1["palette"]), indices_np(o1).reshape(2160, 3840)
    assert idx.min() >= 0 and idx.max() < len(pal) and (np.bincount(idx.ravel(), minlength=len(pal))[1:] > 0).all()
    err = pal[idx].astype(np.float64) - img
    assert 10 * np.log10(255.0 ** 2 / np.mean(err ** 2)) > 26.0


def test_native_entry_many_jobs_regions_and_errors(rh):
    """rhccq_encode_frame beyond what the fuzz frames reach: more than 64 jobs (the scan sets colour bits with atomics instead of byte flags),
    several regions per class with overlapping boxes, a class without any pixel, and the error paths of the C entry (palette buffer too small ->
    RHCCQ_E_LIMIT with the needed size, more than 2 048 segments -> RHCCQ_E_LIMIT, a segment naming a region that does not exist -> RHCCQ_E_ARG)."""
    import ctypes as C
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd._lib import ClassDesc, FrameResult
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    H, W = 180, 240
    img = synth.photo(H, W, 321, sigma=2.0).copy()
    img[20:24, 30:90] = 0
    rgb = torch.from_numpy(img).to(rh.device)
    enc = FrameEncoder(rh)
    # (a) 10 x 10 segments per class: 200 jobs
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (10, 10))
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
    assert nr + nn > 64
    assert same_result(enc.encode(rgb, specs), enc.encode_native(rgb, specs))
    # (b) two regions per class with overlapping boxes, and a third class whose label map is empty
    yy, xx = np.mgrid[0:H, 0:W]
    a = ((yy - 50) ** 2 + (xx - 60) ** 2) <= 40 ** 2
    b = ((yy - 120) ** 2 / 0.5 + (xx - 170) ** 2) <= 45 ** 2
    lab_roi = np.zeros((H, W), np.int32)
    lab_roi[a & (xx < 60)] = 1
    lab_roi[a & (xx >= 60)] = 2
    lab_roi[b & ~a] = 3
    lab_non = np.where(~(a | b), 1 + (xx >= W // 2), 0).astype(np.int32)

    def box(m):
        r, c = np.where(m)
        return (int(r.min()), int(c.min()), int(r.max()) + 1, int(c.max()) + 1)
    specs2 = [ClassSpec(torch.from_numpy(lab_roi).to(rh.device), [0, 0, 1], [box(a), box(b & ~a)], 20),
              ClassSpec(torch.from_numpy(lab_non).to(rh.device), [0, 0], [box(lab_non > 0)], 10)]
    ref = enc.encode(rgb, specs2)
    assert same_result(ref, enc.encode_native(rgb, specs2))
    empty = ClassSpec(torch.zeros((H, W), dtype=torch.int32, device=rh.device), [0], [(0, 0, H, W)], 35)
    got = enc.encode_native(rgb, specs2 + [empty])
    # (the empty class adds nothing but its share of the level-3 quality: q3 = min(40 + 20 + 70, 100) = 100 instead of 60 -- compare with the
    # Python host, which takes the serial path for a frame with an empty class)
    assert same_result(enc.encode(rgb, specs2 + [empty]), got)
    # (c) the C entry's error paths
    descs = (ClassDesc * 2)()
    keep = []
    for d, c in zip(descs, specs2):
        sr, rb = np.ascontiguousarray(c.seg_region, np.int32), np.ascontiguousarray(c.region_bbox, np.int32)
        keep += [sr, rb]
        d.labels, d.n_seg, d.n_region, d.seg_region, d.region_bbox, d.quality = c.labels.data_ptr(), c.n_seg, len(rb), sr.ctypes.data, rb.ctypes.data, c.quality
    out = torch.empty((H * W,), dtype=torch.int32, device=rh.device)
    res = FrameResult()
    pal = np.empty((4, 3), np.uint8)
    rc = rh.lib.rhccq_encode_frame(rh.ctx, rh._p(rgb), H, W, descs, 2, pal.ctypes.data, 4, rh._p(out), None, C.byref(res))
    assert rc == -3 and res.n_colours == len(ref["palette"]) > 4                     # RHCCQ_E_LIMIT + the size to come back with
    bad = np.array([0, 0, 7], np.int32)
    descs[0].seg_region = bad.ctypes.data
    pal = np.empty((1 << 16, 3), np.uint8)
    rc = rh.lib.rhccq_encode_frame(rh.ctx, rh._p(rgb), H, W, descs, 2, pal.ctypes.data, 1 << 16, rh._p(out), None, C.byref(res))
    assert rc == -1 and b"region" in rh._raw.rhccq_last_error(rh.ctx)               # RHCCQ_E_ARG
    descs[0].seg_region = keep[0].ctypes.data
    descs[0].n_seg = 3000
    rc = rh.lib.rhccq_encode_frame(rh.ctx, rh._p(rgb), H, W, descs, 2, pal.ctypes.data, 1 << 16, rh._p(out), None, C.byref(res))
    assert rc == -3 and b"2048" in rh._raw.rhccq_last_error(rh.ctx)
    # ... and the context still works afterwards
    assert same_result(ref, enc.encode_native(rgb, specs2))


def test_native_entry_from_several_host_threads(rh):
    """rhccq_encode_frame is thread-compatible per context: three host threads, each with a context of its own, encode different frames at the
    same time (the MT19937 word table and its device copy are shared process-wide behind locks; lanes, streams and arenas belong to the context).
    Every result must equal the one the same frame gives alone."""
    import threading
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    from roibasedimagecompression_amd.ops import Rhccq
    frames = []
    for i, (H, W) in enumerate(((200, 260), (160, 300), (240, 200))):
        img = synth.photo(H, W, 7000 + i, sigma=6.0).copy()
        (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (1 + i % 2, 1))
        frames.append((img, lr, nr, br, ln, nn, bn))

    def specs_on(ctx, f):
        img, lr, nr, br, ln, nn, bn = f
        return (torch.from_numpy(img).to(ctx.device), [ClassSpec(torch.from_numpy(lr).to(ctx.device), np.zeros(nr, np.int64), [br], 20),
                                                        ClassSpec(torch.from_numpy(ln).to(ctx.device), np.zeros(nn, np.int64), [bn], 10)])
    alone = []
    for f in frames:
        rgb, specs = specs_on(rh, f)
        alone.append(FrameEncoder(rh).encode_native(rgb, specs))
    assert any((a["n_unique"] >= 10000).any() for a in alone)          # MiniBatch-branch problems: the shared word table is in use
    got, errors = [None] * len(frames), []

    def run(i):
        try:
            torch.cuda.set_device(rh.device)
            stream = torch.cuda.Stream(rh.device)
            with torch.cuda.stream(stream):
                ctx = Rhccq(rh.device.index)
                rgb, specs = specs_on(ctx, frames[i])
                enc = FrameEncoder(ctx)
                for _ in range(3):
                    got[i] = enc.encode_native(rgb, specs)
                stream.synchronize()
        except BaseException as e:
            errors.append(e)
    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(frames))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, g in zip(alone, got):
        assert np.array_equal(a["palette"], g["palette"]) and a["indices_dtype"] == g["indices_dtype"]
        assert np.array_equal(a["indices"].cpu().numpy(), g["indices"].cpu().numpy())
