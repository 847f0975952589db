"""Synthetic inputs of SURVEY.md 8d (identical on every box: numpy default_rng(seed)).

photo   per channel: sum of three bilinearly-upsampled uniform grids (H/64, H/16, H/4 cells; weights
        0.6/0.3/0.1) * 255 + N(0, sigma), clipped to uint8 (Kodak-like, k-means heavy);
poster  12 flat colours >= 60 apart with 1-px blended vertical edges (gapped palette: several
        eps-components);
labels  ROI mask = centred ellipse covering 35 % of the frame, dilated by 3 px (the ROI / non-ROI
        overlap band of extract_roi_nonroi, roi.py:685-718); segments = a tiles x tiles grid inside each
        class, ids ascending in raster order of the tiles (1-based; 0 = not in the class).
"""
import numpy as np


def _upsample(g, H, W):
    gh, gw = g.shape
    ys = np.linspace(0, gh - 1, H)
    xs = np.linspace(0, gw - 1, W)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, gh - 1)
    x1 = np.minimum(x0 + 1, gw - 1)
    fy = (ys - y0)[:, None].astype(np.float32)
    fx = (xs - x0)[None, :].astype(np.float32)
    g = g.astype(np.float32)
    top = g[y0][:, x0] * (1 - fx) + g[y0][:, x1] * fx
    bot = g[y1][:, x0] * (1 - fx) + g[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def photo(H, W, seed, sigma=2.0):
    rng = np.random.default_rng(seed)
    out = np.empty((H, W, 3), np.uint8)
    for c in range(3):
        acc = np.zeros((H, W), np.float32)
        for cells, wt in ((64, 0.6), (16, 0.3), (4, 0.1)):
            gh, gw = max(2, H // cells), max(2, W // cells)
            acc += np.float32(wt) * _upsample(rng.random((gh, gw)), H, W)
        acc = acc * np.float32(255.0) + rng.normal(0, sigma, (H, W)).astype(np.float32)
        out[..., c] = np.clip(acc, 0, 255).astype(np.uint8)
    return out


def poster(H, W, seed, ncol=12, cell=8):
    rng = np.random.default_rng(seed)
    cols = []
    while len(cols) < ncol:
        c = rng.integers(8, 248, 3)
        if all(np.linalg.norm(c - o) >= 60 for o in cols):
            cols.append(c)
    cols = np.array(cols, dtype=np.float64)
    lab = rng.integers(0, ncol, (max(2, -(-H // cell)), max(2, -(-W // cell))))
    lab = np.kron(lab, np.ones((cell, cell), dtype=int))[:H, :W]
    img = cols[lab]
    blur = img.copy()
    blur[:, 1:] = 0.5 * (img[:, 1:] + img[:, :-1])
    edge = np.zeros((H, W), bool)
    edge[:, 1:] = lab[:, 1:] != lab[:, :-1]
    img[edge] = blur[edge]
    return np.clip(img, 0, 255).astype(np.uint8)


def roi_masks(H, W, roi_frac=0.35, overlap=3):
    yy, xx = np.mgrid[0:H, 0:W]
    ell = ((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 <= roi_frac * 4 / np.pi
    dil = ell.copy()
    for _ in range(overlap):
        d2 = dil.copy()
        d2[1:] |= dil[:-1]
        d2[:-1] |= dil[1:]
        d2[:, 1:] |= dil[:, :-1]
        d2[:, :-1] |= dil[:, 1:]
        dil = d2
    return dil, ~ell


def grid_labels(mask, tiles_y, tiles_x):
    """1-based segment ids (dense, ascending in tile raster order) for the pixels of `mask`."""
    H, W = mask.shape
    yy, xx = np.mgrid[0:H, 0:W]
    th, tw = -(-H // tiles_y), -(-W // tiles_x)
    tid = (yy // th) * tiles_x + (xx // tw)
    lab = np.where(mask, tid + 1, 0)
    ids = np.unique(lab[lab > 0])
    remap = np.zeros(int(lab.max()) + 1, np.int32)
    remap[ids] = np.arange(1, len(ids) + 1)
    return remap[lab].astype(np.int32), len(ids)


def frame_classes(H, W, tiles, device=None):
    """(roi_labels, n_roi_seg, roi_bbox), (non_labels, n_non_seg, non_bbox): one region per class whose
    bbox is the tight bbox of the class mask (what extract_regions yields for one connected region)."""
    roi, non = roi_masks(H, W)
    out = []
    for m in (roi, non):
        lab, n = grid_labels(m, tiles[0], tiles[1])
        rows, cols = np.where(m)
        bbox = (int(rows.min()), int(cols.min()), int(rows.max()) + 1, int(cols.max()) + 1)
        out.append((lab, n, bbox))
    return out
