"""Full-size cases of BASELINE.json (configs[1] single 4K frame, configs[2] 1080p two-tier batch) through
size-independent properties -- the numpy oracle would need hours at these sizes.  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh():
    from roibasedimagecompression_amd.ops import Rhccq
    return Rhccq(0)


def _frame(rh, H, W, seed, q_roi, q_non, tiles=(2, 1)):
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec
    img = synth.photo(H, W, seed)
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], q_roi),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], q_non)]
    return img, torch.from_numpy(img).to(rh.device), specs, lr, ln


def _idx(out):
    idx = out["indices"].cpu().numpy()
    return (idx.view(np.uint16) if out["indices_dtype"] == "uint16" else idx).astype(np.int64)


def test_4k_frame_properties(rh):
    import torch
    from roibasedimagecompression_amd.frame import FrameEncoder
    H, W = 2160, 3840
    img, rgb, specs, lr, ln = _frame(rh, H, W, 1234, 20, 20)
    enc = FrameEncoder(rh)
    a = enc.encode(rgb, specs)
    b = enc.encode(rgb, specs)
    assert np.array_equal(a["palette"], b["palette"]) and torch.equal(a["indices"], b["indices"])      # deterministic
    # the native host (rhccq_encode_frame) == the Python host, configs[1] at full size, twice (warm lanes, reused arenas)
    for _ in range(2):
        n = enc.encode_native(rgb, specs)
        assert np.array_equal(a["palette"], n["palette"]) and a["indices_dtype"] == n["indices_dtype"] and torch.equal(a["indices"], n["indices"])
        assert np.array_equal(a["n_unique"], n["n_unique"]) and tuple(a["shape"]) == tuple(n["shape"])
    pal, idx = np.asarray(a["palette"]), _idx(a).reshape(H, W)
    assert idx.min() >= 0 and idx.max() < len(pal)
    used = np.bincount(idx.ravel(), minlength=len(pal)) > 0
    assert used[1:].all()                                    # index 0 is the reserved transparent black (merging.py:60)
    keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
    # one (class, segment, colour) -> one final index: the whole hierarchy is a function of the palette entry
    rng = np.random.default_rng(0)
    # (pixels of the 3 px ROI / non-ROI overlap are painted by whichever class comes last: left out)
    for lab, other in ((lr, ln), (ln, lr)):
        for s in range(1, int(lab.max()) + 1):
            pos = np.flatnonzero((lab.ravel() == s) & (other.ravel() == 0))
            pos = pos[rng.integers(0, len(pos), 400000)]
            k, i = keys.ravel()[pos], idx.ravel()[pos]
            order = np.argsort(k, kind="stable")
            k, i = k[order], i[order]
            same = k[1:] == k[:-1]
            assert (i[1:][same] == i[:-1][same]).all()
    # a pixel covered by both classes (the 3 px overlap) takes the colour of the class painted last; every pixel
    # decodes to a colour near its own: the three levels average at most eps-connected / k-means neighbours
    dec = pal[idx]
    err = dec.astype(np.float64) - img
    psnr = 10 * np.log10(255.0 ** 2 / np.mean(err ** 2))
    assert psnr > 28.0, psnr
    assert np.abs(err).max() < 128


def test_4k_dct_extension_parseval(rh):
    """EXTENSION (no reference counterpart): orthonormal 8x8 DCT-II keeps each block's energy (tolerance 1e-4
    relative: float32 coefficients of float64 dot products) and its DC term is 8 x the block mean."""
    import torch
    H, W = 2160, 3840
    img, rgb, specs, lr, ln = _frame(rh, H, W, 1234, 20, 20)
    roi = torch.from_numpy((lr > 0).astype(np.uint8)).to(rh.device)
    luma, qstep = rh.luma_qstep(rgb, roi, 8, 4.0, 16.0)
    coef, q = rh.dct_quant(luma, 8, qstep)
    L = luma.cpu().numpy().astype(np.float64).reshape(H // 8, 8, W // 8, 8)
    C = coef.cpu().numpy().astype(np.float64).reshape(H // 8, 8, W // 8, 8)
    e_l, e_c = (L ** 2).sum(axis=(1, 3)), (C ** 2).sum(axis=(1, 3))
    assert np.abs(e_c - e_l).max() <= 1e-4 * e_l.max()
    assert np.abs(C[:, 0, :, 0] - 8.0 * L.mean(axis=(1, 3))).max() <= 1e-3
    qs = qstep.cpu().numpy().astype(np.float64)
    want = np.rint(C / qs[:, None, :, None])
    got = q.cpu().numpy().reshape(H // 8, 8, W // 8, 8)
    assert (np.abs(got - want) <= 1).all() and (got == want).mean() > 0.999     # ties of rint on f32-rounded coef


def test_1080p_two_tier_batch_equals_frame_by_frame(rh):
    """configs[2] shape (two-tier 20/10, 1080p), a batch of 4 instead of 64 to bound the test time"""
    import torch
    from roibasedimagecompression_amd.frame import FrameEncoder
    frames = []
    for i in range(4):
        _, rgb, specs, _, _ = _frame(rh, 1080, 1920, 1234 + i, 20, 10, tiles=(3, 4))
        frames.append((rgb, specs))
    enc = FrameEncoder(rh)
    batch = enc.encode_batch(frames)
    for i in (0, 3):
        single = enc.encode(*frames[i])
        assert np.array_equal(single["palette"], batch[i]["palette"])
        assert torch.equal(single["indices"], batch[i]["indices"])


def test_config4_stream_of_4k_frames_16x16_dct(rh):
    """BASELINE.json configs[4] on one GPU: 8 DISTINCT 4K frames, two quality tiers (20, 10), through stream.StreamEncoder
    (batches of 4, 2 batches in flight on host threads with their own HIP streams) plus the 16x16 DCT extension.
    Size-independent properties: the stream result equals each frame encoded alone, bit for bit (palette and every index);
    running the stream again gives the same bytes; every palette entry is used; Parseval for the 16x16 orthonormal DCT
    (sum of squared coefficients == sum of squared samples, per block, to float32 accuracy); quantised integers ==
    rint(coefficient / step)."""
    import torch
    from roibasedimagecompression_amd.frame import FrameEncoder
    from roibasedimagecompression_amd.stream import StreamEncoder
    H, W = 2160, 3840
    frames, masks = [], []
    for i in range(8):
        img, rgb, specs, lr, ln = _frame(rh, H, W, 4321 + i, 20, 10)
        frames.append((rgb, specs))
        masks.append(torch.from_numpy((lr > 0).astype(np.uint8)).to(rh.device))
    se = StreamEncoder(0, batch=4, lanes=2)
    got = se.run(frames)
    again = se.run(frames)
    se.close()
    enc = FrameEncoder(rh)
    for i in (0, 5):                                         # two of them alone (a 4K frame alone costs ~0.25 s)
        alone = enc.encode(*frames[i])
        assert np.array_equal(alone["palette"], got[i]["palette"]) and torch.equal(alone["indices"], got[i]["indices"])
    sizes = set()
    for a, b in zip(got, again):
        assert np.array_equal(a["palette"], b["palette"]) and torch.equal(a["indices"], b["indices"])
        idx = _idx(a)
        assert idx.max() < len(a["palette"]) and (np.bincount(idx.ravel(), minlength=len(a["palette"]))[1:] > 0).all()
        sizes.add(len(a["palette"]))
    assert len(sizes) > 1                                    # distinct frames, distinct palettes
    # 16x16 DCT extension on one of the frames
    rgb, m = frames[3][0], masks[3]
    luma, qstep = rh.luma_qstep(rgb, m, 16, 4.0, 16.0)
    coef, q = rh.dct_quant(luma, 16, qstep)
    lb = luma.double().reshape(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3)
    cb = coef.double().reshape(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3)
    e_in, e_out = (lb * lb).sum(dim=(2, 3)), (cb * cb).sum(dim=(2, 3))
    assert float(((e_in - e_out).abs() / e_in.clamp_min(1.0)).max()) < 1e-5
    want_q = torch.round(cb / qstep.double()[:, :, None, None]).to(torch.int16)
    got_q = q.reshape(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3)
    # rint of a float32 coefficient that sits within 1e-5 of a half-integer multiple of the step may land either way
    diff = (want_q != got_q)
    near = ((cb / qstep.double()[:, :, None, None]) % 1.0 - 0.5).abs() < 1e-4
    assert not bool((diff & ~near).any())


def test_config2_batch_of_64_1080p_frames(rh):
    """BASELINE.json configs[2] AT ITS STATED SIZE: a batch of 64 DISTINCT 1080p frames, two quality tiers (20, 10), 12 segments per
    class = 1 536 (frame, segment) jobs in ONE FrameEncoder.encode_batch call -- beyond 64 jobs the colour flags are set by the atomic
    scan path (job_scan_kernel<false>), which no smaller test reaches at size.  Size-independent properties: two sampled frames equal
    their frame-by-frame encode bit for bit (palette and every index); every frame's indices stay inside its palette and every palette
    entry but the transparent one is used; distinct frames give distinct palettes; PSNR of the decoded frame stays above 26 dB."""
    import torch
    from roibasedimagecompression_amd.frame import FrameEncoder
    H, W, B = 1080, 1920, 64
    frames, imgs = [], {}
    for i in range(B):
        img, rgb, specs, _, _ = _frame(rh, H, W, 5000 + i, 20, 10, tiles=(3, 4))
        frames.append((rgb, specs))
        if i in (0, 17, 63):
            imgs[i] = img
    assert sum(sp.n_seg for _, specs in frames for sp in specs) > 64 * 20
    enc = FrameEncoder(rh)
    batch = enc.encode_batch(frames)
    assert len(batch) == B
    for i in (17, 63):
        single = enc.encode(*frames[i])
        assert np.array_equal(single["palette"], batch[i]["palette"]), i
        assert torch.equal(single["indices"], batch[i]["indices"]), i
    sizes = set()
    for i, out in enumerate(batch):
        pal, idx = np.asarray(out["palette"]), _idx(out)
        assert idx.shape == (H, W) and idx.min() >= 0 and idx.max() < len(pal)
        assert (np.bincount(idx.ravel(), minlength=len(pal))[1:] > 0).all()
        sizes.add((len(pal), int(idx.sum() % 1000003)))
        if i in imgs:
            err = pal[idx].astype(np.float64) - imgs[i]
            assert 10 * np.log10(255.0 ** 2 / np.mean(err ** 2)) > 26.0
    assert len(sizes) > B // 2


def test_config4_stream_of_256_4k_frames(rh):
    """BASELINE.json configs[4] AT ITS STATED LENGTH on one GPU: a stream of 256 4K frames (8 distinct frames, 32 times each: the host
    generator needs a second per distinct frame) through stream.StreamEncoder with the bench's shape (batches of 24, 5 in flight), tiers
    (20, 10).  Every occurrence of a frame gives the same palette and the same index map as its first one and as the frame encoded
    alone -- whatever batch and lane it travelled in."""
    import torch
    from roibasedimagecompression_amd.frame import FrameEncoder
    from roibasedimagecompression_amd.stream import StreamEncoder
    H, W = 2160, 3840
    distinct = []
    for i in range(8):
        _, rgb, specs, _, _ = _frame(rh, H, W, 8800 + i, 20, 10)
        distinct.append((rgb, specs))
    order = [(7 * j + j // 8) % 8 for j in range(256)]       # neighbours differ, every batch mixes the frames differently
    se = StreamEncoder(0, batch=24, lanes=5)
    got = se.run([distinct[i] for i in order])
    se.close()
    assert len(got) == 256
    first = {}
    for j, i in enumerate(order):
        if i not in first:
            first[i] = got[j]
        else:
            assert np.array_equal(got[j]["palette"], first[i]["palette"]) and torch.equal(got[j]["indices"], first[i]["indices"]), (j, i)
    enc = FrameEncoder(rh)
    for i in (2, 6):
        alone = enc.encode(*distinct[i])
        assert np.array_equal(alone["palette"], first[i]["palette"]) and torch.equal(alone["indices"], first[i]["indices"])
