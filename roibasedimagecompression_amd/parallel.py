"""Multi-GPU execution of the RHCCQ hot path: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on MI355X; "gloo" in the CPU tests).

  frame-parallel  frames are independent: frames are dealt round-robin to the ranks, no data-path
                  collective (cfg3 / cfg5 and bench.py).
  tile-parallel   one frame cut into tiles, one tile per rank (cfg4: 8K as 2x4 tiles).  Pixels stay
                  tile-local, palettes are exchanged:
                    1. every rank runs the per-pixel scan of its tile (stats + colour bitmaps per segment);
                    2. ONE all-gather moves {segment bitmaps (2 MiB each), segment stats} -- fixed size, one
                       hop over the xGMI mesh; every rank ORs the bitmaps, so all ranks hold the same sorted
                       per-segment palettes (== np.unique over the whole segment);
                    3. palette-space clustering runs redundantly and bit-identically on every rank;
                       the first-seen positions that only fix palette ORDER at the merges are reduced with
                       one small MIN all-reduce over palette-sized arrays (KBs);
                    4. every rank remaps its own tile with the composed LUT.
                  (A segment with black pixels inside needs one more 8-byte-per-segment MIN reduction.)
"""
import numpy as np
import torch
import torch.distributed as dist

from .frame import FrameEncoder, _Comp, _merge, _scatter_min          # noqa: F401
from .ops import INT_MAX

__all__ = ["shard_frames", "all_gather_stack", "all_reduce_min_", "TiledFrameEncoder", "tile_grid", "tiled_ccl", "stitch_tiles",
           "exchange_segment_tables", "reduce_black_fix", "reduce_first_positions"]


def shard_frames(n_frames, rank, world):
    """frame indices owned by `rank` (round-robin: consecutive frames go to different GPUs)."""
    return list(range(rank, n_frames, world))


def tile_grid(H, W, rows, cols):
    """[(r0, c0, h, w)] of a rows x cols tiling in rank order (row-major), edges absorb the remainder."""
    rs = [H * i // rows for i in range(rows + 1)]
    cs = [W * j // cols for j in range(cols + 1)]
    return [(rs[i], cs[j], rs[i + 1] - rs[i], cs[j + 1] - cs[j]) for i in range(rows) for j in range(cols)]


def _via_cpu(group):
    return dist.get_backend(group) == "gloo"


def all_gather_stack(t, group=None):
    """all-gather equally-shaped tensors -> [world, *t.shape] (one collective)."""
    world = dist.get_world_size(group)
    if world == 1:
        return t.unsqueeze(0)
    src = t.cpu() if (_via_cpu(group) and t.is_cuda) else t
    flat = src.contiguous().reshape(-1)
    out = torch.empty((world * flat.numel(),), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.reshape((world,) + tuple(src.shape)).to(t.device)


def all_reduce_min_(t, group=None):
    if dist.get_world_size(group) == 1:
        return t
    if _via_cpu(group) and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MIN, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return t


# ---- the three exchanges of the tile-parallel frame encoder, as functions of tensors (CPU tensors under gloo in the tests, device tensors
# under RCCL): what crosses the ranks, in which order, and how it is combined -- tests/test_distributed_cpu.py runs them at world 8 ------
def exchange_segment_tables(bitmaps, stats, origins, group=None):
    """THE data-path collective: ONE all-gather of every tile's {segment colour bitmaps (int32[n_jobs][words]), segment statistics
    (int32[n_jobs][6], tile coordinates)}.  -> (OR of the bitmaps over the tiles: the segments' colour sets of the whole frame, a
    tensor like `bitmaps`; st int64[n_jobs][6] = (min row, max row, min col, max col, pixels, black pixels) in FRAME coordinates).
    origins: int64[world][2] top-left corner of every rank's tile."""
    n_jobs = int(bitmaps.shape[0])
    payload = torch.cat([bitmaps.reshape(-1), stats.reshape(-1)])
    allp = all_gather_stack(payload, group)
    nb = bitmaps.numel()
    allb = allp[:, :nb].reshape(allp.shape[0], n_jobs, -1)
    merged = allb[0].clone()
    for i in range(1, allb.shape[0]):
        merged |= allb[i]
    alls = allp[:, nb:].reshape(allp.shape[0], n_jobs, 6).cpu().numpy().astype(np.int64)
    st = np.zeros((n_jobs, 6), np.int64)
    cnt = alls[:, :, 4]
    big = np.int64(INT_MAX)
    st[:, 0] = np.where(cnt > 0, alls[:, :, 0] + origins[:, None, 0], big).min(0)
    st[:, 1] = np.where(cnt > 0, alls[:, :, 1] + origins[:, None, 0], -1).max(0)
    st[:, 2] = np.where(cnt > 0, alls[:, :, 2] + origins[:, None, 1], big).min(0)
    st[:, 3] = np.where(cnt > 0, alls[:, :, 3] + origins[:, None, 1], -1).max(0)
    st[:, 4], st[:, 5] = cnt.sum(0), alls[:, :, 5].sum(0)
    return merged.reshape(bitmaps.shape), st


def reduce_black_fix(has, norm, pos_global, key, needs_fix, device, group=None):
    """The black-in-segment fix across tiles (subregions.py:393-421: the in-mask non-black pixel with the smallest R^2 + G^2 + B^2, first in
    raster order): lexicographic MIN over the ranks of (norm2, frame raster position) -- one 8-byte-per-segment MIN all-reduce --, then the
    winner's colour rides in a second one.  has / norm / pos_global / key: this tile's candidate per job.  -> uint32[n_jobs] fix keys."""
    packed = np.where(has, (norm.astype(np.int64) << 40) | pos_global.astype(np.int64), np.int64(2 ** 62))
    red = all_reduce_min_(torch.from_numpy(packed.copy()).to(device), group).cpu().numpy()
    mine = has & (packed == red)
    keyred = all_reduce_min_(torch.from_numpy(np.where(mine, key.astype(np.int64), np.int64(2 ** 62))).to(device), group).cpu().numpy()
    return np.where(needs_fix, keyred, 0).astype(np.uint32)


def reduce_first_positions(p_local, tile, frame_w, group=None):
    """tile-local first raster positions (int64 tensor, INT_MAX = the entry has no pixel in this tile) -> frame raster positions -> MIN over
    the ranks (one palette-sized all-reduce, reduced in place on the device under RCCL).  -> numpy int64"""
    r0, c0, h, w = tile
    g = torch.where(p_local >= INT_MAX, torch.full_like(p_local, INT_MAX),
                    (torch.div(p_local, w, rounding_mode="floor") + r0) * frame_w + (p_local % w + c0))
    return all_reduce_min_(g, group).cpu().numpy().astype(np.int64)


class TiledFrameEncoder(FrameEncoder):
    """FrameEncoder whose per-pixel passes see one tile of the frame; `prepare` is the only stage that
    differs (exchange after the scan, global positions); levels 1-3 are inherited unchanged and run
    redundantly on every rank."""

    PIPELINE_CLASSES = False     # the collectives of the two classes must come in one order on every rank: classes one after the other

    def __init__(self, rh, frame_shape, tile, group=None):
        super().__init__(rh)
        self.frame_shape = tuple(frame_shape)
        self.tile = tuple(tile)               # (r0, c0, h, w) of this rank's tile
        self.group = group

    def _global_pos(self, p_local):
        r0, c0, h, w = self.tile
        W = self.frame_shape[1]
        p = np.asarray(p_local, dtype=np.int64)
        return np.where(p >= INT_MAX, np.int64(INT_MAX), (p // w + r0) * W + (p % w + c0))

    def prepare(self, rgb, classes):
        """rgb: the tile (uint8[h,w,3]); classes[i].labels: the tile of the class label map (global segment
        ids); region boxes are in frame coordinates."""
        rh = self.rh
        r0t, c0t, h, w = self.tile
        H, W = self.frame_shape
        assert tuple(rgb.shape[:2]) == (h, w)
        labels = [c.labels for c in classes]
        job_base = np.concatenate([[0], np.cumsum([c.n_seg for c in classes])]).astype(np.int64)
        n_jobs = int(job_base[-1])
        bitmaps, stats = rh.new_job_state(n_jobs)
        rh.job_scan(rgb, labels, job_base[:-1], bitmaps, stats, black_is_colour=False)
        # ---- the one data-path collective: bitmaps + stats of every tile -----------------------------
        bitmaps, st = exchange_segment_tables(bitmaps, stats, self.tile_origins, self.group)
        count, n_black = st[:, 4], st[:, 5]
        present = count > 0
        job_class = np.repeat(np.arange(len(classes)), [c.n_seg for c in classes])
        job_region = np.concatenate([c.seg_region for c in classes])
        rb = np.concatenate([c.region_bbox[c.seg_region] for c in classes])
        r0 = np.maximum(rb[:, 0], st[:, 0] - 2)
        r1 = np.minimum(rb[:, 2] - 1, st[:, 1] + 2)
        c0 = np.maximum(rb[:, 1], st[:, 2] - 2)
        c1 = np.minimum(rb[:, 3] - 1, st[:, 3] + 2)
        crop_area = (r1 - r0 + 1) * (c1 - c0 + 1)
        has_bg = present & (crop_area > count)
        needs_fix = present & (n_black > 0) & (count > n_black)
        all_black = present & (n_black > 0) & (count == n_black)
        fix_key = None
        if needs_fix.any():
            best = torch.full((n_jobs,), -1, dtype=torch.int64, device=rh.device)
            rh.job_blackfix(rgb, labels, job_base[:-1], rh.dev(needs_fix.astype(np.uint8)), best)
            b = best.cpu().numpy().view(np.uint64)
            norm = (b >> np.uint64(40)).astype(np.int64)
            pos = self._global_pos(np.where(b == np.uint64(2 ** 64 - 1), INT_MAX, (b & np.uint64((1 << 40) - 1)).astype(np.int64)))
            # colour at the local best position (valid where this tile has a candidate)
            lp = (b & np.uint64((1 << 40) - 1)).astype(np.int64)
            has = b != np.uint64(2 ** 64 - 1)
            px = rgb.reshape(-1, 3)[torch.from_numpy(np.where(has, lp, 0)).to(rh.device)].cpu().numpy().astype(np.int64)
            key = (px[:, 0] << 16) | (px[:, 1] << 8) | px[:, 2]
            fk = reduce_black_fix(has, norm, pos, key, needs_fix, rh.device, self.group)
            fix_key = rh.dev(fk.view(np.int32))
        rh.job_set_black(bitmaps, np.nonzero(has_bg | all_black)[0])
        chunk, counts = rh.bitmap_count(bitmaps)
        P = counts.cpu().numpy().astype(np.int64)
        pal_off = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
        d_pal_off = rh.dev(pal_off[:-1].copy())
        total = int(pal_off[-1])
        prefix, keys_dev = rh.bitmap_emit(bitmaps, chunk, d_pal_off, total)
        return {"H": H, "W": W, "rgb": rgb, "classes": classes, "labels": labels, "job_base": job_base, "n_jobs": n_jobs,
                "bitmaps": bitmaps, "prefix": prefix, "pal_off": pal_off, "d_pal_off": d_pal_off, "fix_key": fix_key,
                "keys_dev": keys_dev, "has_black": has_bg | all_black, "P": P, "present": present, "job_class": job_class,
                "job_region": job_region, "crop": (r0, r1, c0, c1), "total": total}

    def first_positions(self, S, fp_lut, n_entries):
        """tile-local first positions -> frame raster positions -> MIN over ranks: the only other exchange of
        the tiled path, the size of the clustered palettes (a few 10^4 int64 per segment at 4K)."""
        p = self.first_positions_dev(S, fp_lut, n_entries).to(torch.int64)      # stays on the device up to the collective (RCCL reduces it in place)
        return reduce_first_positions(p, self.tile, self.frame_shape[1], self.group)

    @property
    def tile_origins(self):
        return self._origins

    def set_tiles(self, tiles):
        """tiles: [(r0, c0, h, w)] of every rank, rank order."""
        self._origins = np.array([[t[0], t[1]] for t in tiles], dtype=np.int64)
        return self


# ---- connected components of a tiled mask: the seam stitch (last bullet; BASELINE.json north_star) ------------------
# Region extraction of the ROI stage labels a frame-sized mask (cv2.connectedComponentsWithStats in the reference,
# encoder/ROI/roi.py:285-360).  Tile-parallel: every rank labels ITS tile (csrc/ccl.hip), then ONE all-gather moves, per tile, the
# labels along its four borders and its component table (ordering key, bounding box, area per component; the payload length goes
# round in a 4-byte all-gather first).  Every rank then runs the same union-find over the label pairs that face each other across
# a seam (8-connectivity: also diagonally, across tile corners too), numbers the merged components exactly as the single-GPU kernel
# does -- by the smallest ordering key of their parts -- and rewrites its own tile through a look-up table.  Pixels never leave
# their tile; labels, numbering and statistics equal rhccq_ccl on the whole mask (tests/test_gpu_tiled.py).
def stitch_tiles(tiles, frame_shape, parts, connectivity=8):
    """tiles: [(r0, c0, h, w)] rank order; parts[r] = dict(n, top, bottom, left, right (int arrays: the tile's border labels),
    stats int[n + 1][5] (cv2 column order, tile coordinates), keys uint[n + 1] (frame coordinates)).  Pure numpy, identical on every
    rank.  -> (n_global, [lut_r: int32[n_r + 1] local -> global label], stats int32[n_global + 1][5])"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    H, W = frame_shape
    ns = np.array([int(p["n"]) for p in parts], np.int64)
    base = np.concatenate([[0], np.cumsum(ns)])
    N = int(base[-1])

    def gid(r, lab):                                            # node of (rank, local label >= 1); -1 for background
        lab = np.asarray(lab, np.int64)
        return np.where(lab > 0, base[r] + lab - 1, -1)
    ea, eb = [], []
    steps = (-1, 0, 1) if connectivity == 8 else (0,)

    def seam(a, b):                                             # two facing lines of node ids (frame-long, -1 = nothing there)
        for d in steps:
            if d == 0:
                x, y = a, b
            elif d == 1:
                x, y = a[:-1], b[1:]
            else:
                x, y = a[1:], b[:-1]
            m = (x >= 0) & (y >= 0)
            ea.append(x[m])
            eb.append(y[m])
    for y in sorted({t[0] for t in tiles if t[0] > 0}):         # horizontal seams: the frame-wide rows y - 1 (above) and y (below)
        above, below = np.full(W, -1, np.int64), np.full(W, -1, np.int64)
        for r, (r0, c0, h, w) in enumerate(tiles):
            if r0 + h == y:
                above[c0:c0 + w] = gid(r, parts[r]["bottom"])
            if r0 == y:
                below[c0:c0 + w] = gid(r, parts[r]["top"])
        seam(above, below)
    for x in sorted({t[1] for t in tiles if t[1] > 0}):         # vertical seams: the frame-high columns x - 1 and x
        lcol, rcol = np.full(H, -1, np.int64), np.full(H, -1, np.int64)
        for r, (r0, c0, h, w) in enumerate(tiles):
            if c0 + w == x:
                lcol[r0:r0 + h] = gid(r, parts[r]["right"])
            if c0 == x:
                rcol[r0:r0 + h] = gid(r, parts[r]["left"])
        seam(lcol, rcol)
    if N == 0:
        comp = np.zeros(0, np.int64)
        n_comp = 0
    else:
        ea = np.concatenate(ea) if ea else np.zeros(0, np.int64)
        eb = np.concatenate(eb) if eb else np.zeros(0, np.int64)
        g = coo_matrix((np.ones(len(ea), np.int8), (ea, eb)), shape=(N, N))
        n_comp, comp = connected_components(g, directed=False)
    # component tables in frame coordinates
    key = np.concatenate([np.asarray(p["keys"], np.int64)[1:int(p["n"]) + 1] for p in parts]) if N else np.zeros(0, np.int64)
    st = [np.asarray(p["stats"], np.int64).reshape(-1, 5) for p in parts]
    L = np.concatenate([s_[1:n + 1, 0] + t[1] for s_, n, t in zip(st, ns, tiles)]) if N else np.zeros(0, np.int64)
    T = np.concatenate([s_[1:n + 1, 1] + t[0] for s_, n, t in zip(st, ns, tiles)]) if N else np.zeros(0, np.int64)
    R = L + (np.concatenate([s_[1:n + 1, 2] for s_, n in zip(st, ns)]) if N else 0)
    B = T + (np.concatenate([s_[1:n + 1, 3] for s_, n in zip(st, ns)]) if N else 0)
    A = np.concatenate([s_[1:n + 1, 4] for s_, n in zip(st, ns)]) if N else np.zeros(0, np.int64)
    big = np.iinfo(np.int64).max
    ckey, cL, cT = np.full(n_comp, big), np.full(n_comp, big), np.full(n_comp, big)
    cR, cB, cA = np.zeros(n_comp, np.int64), np.zeros(n_comp, np.int64), np.zeros(n_comp, np.int64)
    np.minimum.at(ckey, comp, key)
    np.minimum.at(cL, comp, L)
    np.minimum.at(cT, comp, T)
    np.maximum.at(cR, comp, R)
    np.maximum.at(cB, comp, B)
    np.add.at(cA, comp, A)
    order = np.argsort(ckey, kind="stable")                     # keys are unique: a component's first block / pixel is its own
    label_of = np.empty(n_comp, np.int64)
    label_of[order] = np.arange(1, n_comp + 1)
    luts = []
    for r in range(len(parts)):
        lut = np.zeros(int(ns[r]) + 1, np.int32)
        lut[1:] = label_of[comp[base[r]:base[r + 1]]]
        luts.append(lut)
    stats = np.zeros((n_comp + 1, 5), np.int32)
    stats[1:, 0], stats[1:, 1] = cL[order], cT[order]
    stats[1:, 2], stats[1:, 3], stats[1:, 4] = (cR - cL)[order], (cB - cT)[order], cA[order]
    # the background row: the union of the tiles' background boxes
    bg = [(s_[0], t) for s_, t in zip(st, tiles) if s_[0, 4] > 0]
    if bg:
        l0 = min(s0[0] + t[1] for s0, t in bg)
        t0 = min(s0[1] + t[0] for s0, t in bg)
        r1 = max(s0[0] + s0[2] + t[1] for s0, t in bg)
        b1 = max(s0[1] + s0[3] + t[0] for s0, t in bg)
        stats[0] = (l0, t0, r1 - l0, b1 - t0, sum(int(s0[4]) for s0, _ in bg))
    return n_comp, luts, stats


def tiled_ccl(rh, mask_tile, tile, frame_shape, tiles, group=None, connectivity=8, numbering="opencv"):
    """Connected components of a frame-sized mask, one tile per rank.  mask_tile: this rank's tile (device u8 / bool [h, w]);
    tile = (r0, c0, h, w) of this rank, tiles = every rank's, rank order (parallel.tile_grid; even origins for OpenCV's 8-connectivity
    numbering).  -> (n, labels of this tile (device int32[h, w], FRAME-wide numbering), stats np.int32[n + 1][5] as rhccq_ccl gives
    for the whole mask): identical to labelling the whole mask on one GPU."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    r0, c0, h, w = tile
    H, W = frame_shape
    n, lab, st = rh.ccl(mask_tile, connectivity, numbering=numbering)
    keys = rh.ccl_keys(lab, n, r0, c0, W, connectivity, numbering)
    strips = torch.cat([lab[0, :], lab[-1, :], lab[:, 0], lab[:, -1]]).to(torch.int64)
    payload = torch.cat([torch.tensor([n, h, w], dtype=torch.int64, device=rh.device), strips,
                         torch.from_numpy(st.astype(np.int64).reshape(-1)).to(rh.device), torch.from_numpy(keys.astype(np.int64)).to(rh.device)])
    if world > 1:
        lens = all_gather_stack(torch.tensor([payload.numel()], dtype=torch.int64, device=rh.device), group).reshape(-1)
        m = int(lens.max().item())
        padded = torch.zeros((m,), dtype=torch.int64, device=rh.device)
        padded[:payload.numel()] = payload
        allp = all_gather_stack(padded, group).cpu().numpy()       # THE seam all-gather
    else:
        allp = payload.cpu().numpy()[None]
    parts = []
    for r in range(world):
        p = allp[r]
        nr, hr, wr = int(p[0]), int(p[1]), int(p[2])
        o = 3
        top, bottom = p[o:o + wr], p[o + wr:o + 2 * wr]
        left, right = p[o + 2 * wr:o + 2 * wr + hr], p[o + 2 * wr + hr:o + 2 * wr + 2 * hr]
        o += 2 * wr + 2 * hr
        stats_r = p[o:o + 5 * (nr + 1)].reshape(nr + 1, 5)
        o += 5 * (nr + 1)
        parts.append({"n": nr, "top": top, "bottom": bottom, "left": left, "right": right, "stats": stats_r, "keys": p[o:o + nr + 1]})
    n_glob, luts, stats = stitch_tiles(tiles, (H, W), parts, connectivity)
    out = rh.remap(lab.reshape(-1), rh.dev(luts[rank])).reshape(h, w)
    return n_glob, out, stats
