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

__all__ = ["shard_frames", "all_gather_stack", "all_reduce_min_", "TiledFrameEncoder", "tile_grid"]


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
        payload = torch.cat([bitmaps.reshape(-1), stats.reshape(-1)])
        allp = all_gather_stack(payload, self.group)
        nb = bitmaps.numel()
        allb = allp[:, :nb].reshape(allp.shape[0], n_jobs, -1)
        bitmaps = allb[0].clone()
        for i in range(1, allb.shape[0]):
            bitmaps |= allb[i]
        alls = allp[:, nb:].reshape(allp.shape[0], n_jobs, 6).cpu().numpy().astype(np.int64)
        origins = self.tile_origins
        st = np.zeros((n_jobs, 6), np.int64)
        cnt = alls[:, :, 4]
        big = np.int64(INT_MAX)
        rmin = np.where(cnt > 0, alls[:, :, 0] + origins[:, None, 0], big).min(0)
        rmax = np.where(cnt > 0, alls[:, :, 1] + origins[:, None, 0], -1).max(0)
        cmin = np.where(cnt > 0, alls[:, :, 2] + origins[:, None, 1], big).min(0)
        cmax = np.where(cnt > 0, alls[:, :, 3] + origins[:, None, 1], -1).max(0)
        st[:, 0], st[:, 1], st[:, 2], st[:, 3] = rmin, rmax, cmin, cmax
        st[:, 4], st[:, 5] = cnt.sum(0), alls[:, :, 5].sum(0)
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
            # lexicographic MIN over ranks of (norm2, global position); the colour rides in a second reduction
            packed = np.where(has, (norm << 40) | pos, np.int64(2 ** 62))
            red = torch.from_numpy(packed.copy())
            red = all_reduce_min_(red.to(rh.device), self.group).cpu().numpy()
            mine = has & (packed == red)
            keyred = torch.from_numpy(np.where(mine, key, np.int64(2 ** 62)))
            keyred = all_reduce_min_(keyred.to(rh.device), self.group).cpu().numpy()
            fk = np.where(needs_fix, keyred, 0).astype(np.uint32)
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
        local = super().first_positions(S, fp_lut, n_entries)
        t = torch.from_numpy(self._global_pos(local)).to(self.rh.device)
        return all_reduce_min_(t, self.group).cpu().numpy().astype(np.int64)

    @property
    def tile_origins(self):
        return self._origins

    def set_tiles(self, tiles):
        """tiles: [(r0, c0, h, w)] of every rank, rank order."""
        self._origins = np.array([[t[0], t[1]] for t in tiles], dtype=np.int64)
        return self
