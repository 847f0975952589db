"""Fused frame encoder: the three-level RHCCQ palette hierarchy of rhccq.ipynb:978-1039
(subregion_quantization -> region_quantization x2 -> quantize_image) for one RGB frame whose
ROI / non-ROI segment label maps are given (ROI detection and SLIC are upstream).

Per-pixel work runs in four streaming HIP passes (scan, [black-fix], first-position, final remap);
everything between them is palette-space work: HIP kernels for the clustering (K3/K4/K7/K8/K2) and
small numpy bookkeeping for the reference's ordering rules (first-seen merge order, black first,
passthrough of single components).  A pixel's index is never materialised between the levels: the
three levels are composed into one LUT per (segment, level-1 palette index) and applied once.

Reference semantics reproduced (paths relative to the reference root):
  encoder/compression/subregions.py:315-449,634-679   per-segment crop(+2 px), black-in-segment fix,
                                                     unique colours, cluster(q), merge per region
  encoder/compression/regions.py:9-70                 merge per class on the frame canvas, cluster(2q)
  encoder/compression/image.py:243-286                merge ROI + non-ROI, cluster(q3), index dtype
  encoder/compression/merging.py:16-21,52-82          single-component passthrough, reversed painting,
                                                     first-seen global palette, black = 0 = transparent
"""
import time

import numpy as np
import torch

from .ops import INT_MAX, clustering_params, unpack_rgb
from .hostsort import _stable_order, _unique_first_inverse
from .palette import cluster_palettes

__all__ = ["ClassSpec", "FrameEncoder"]

_FP_NONE = np.int64(INT_MAX)


class ClassSpec:
    """One region class (ROI or non-ROI) of a frame.

    labels      int32[H,W] device tensor: 0 = pixel not in the class, s >= 1 = SLIC segment id
                (ids are global within the class, ascending inside each region as the reference
                iterates them, slic.py:158-160);
    seg_region  int array[n_seg]: region index of segment id s (entry s-1);
    region_bbox int array[R,4]: (minr, minc, maxr, maxc) of each connected region (roi.py:349-358);
    quality     level-1 quality of the class (rhccq.ipynb:584-585)."""

    def __init__(self, labels, seg_region, region_bbox, quality):
        self.labels = labels
        self.seg_region = np.asarray(seg_region, dtype=np.int64)
        self.region_bbox = np.asarray(region_bbox, dtype=np.int64).reshape(-1, 4)
        self.quality = quality
        self.n_seg = len(self.seg_region)


class _TableOverflow(Exception):
    pass


class _Comp:
    """A component of the hierarchy in palette space."""
    __slots__ = ("keys", "fp", "top_left", "shape", "maps", "merged")

    def __init__(self, keys, fp, top_left, shape, maps, merged):
        self.keys = keys          # uint32[K] palette keys (palette order)
        self.fp = fp              # int64[K] first absolute raster position showing the entry
        self.top_left = top_left
        self.shape = shape
        self.maps = maps          # {job: int32[K1_job]} index in the job's CLUSTERED level-1 palette -> index into keys
        self.merged = merged      # True: canvas semantics (index 0 = black = uncovered)


def _scatter_min(n, idx, val, out=None):
    """out[idx[i]] = min(out[idx[i]], val[i]): one native loop (numpy's ufunc.at is slow, sort + reduceat costs a sort)."""
    if out is None:
        out = np.full(n, _FP_NONE, np.int64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    if len(idx) == 0:
        return out
    val = np.ascontiguousarray(val, dtype=np.int64)
    from . import _lib
    if _lib.load().rhccq_scatter_min_host(len(out), idx.ctypes.data, val.ctypes.data, len(idx), out.ctypes.data) != 0:
        raise IndexError("scatter_min: index outside the table")
    return out


def _merge(comps, bbox):
    """merge_region_components_simple in palette space (merging.py:8-120): one native hash pass (rhccq_merge_palettes_host)."""
    if not comps:
        return None
    if len(comps) == 1:
        return comps[0]
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    n = len(comps)
    keys = [np.ascontiguousarray(c.keys, dtype=np.uint32) for c in comps]
    fps = [np.ascontiguousarray(c.fp, dtype=np.int64) for c in comps]
    luts = [np.empty(len(k), np.int32) for k in keys]
    counts = np.array([len(k) for k in keys], np.int32)
    total = int(counts.sum())
    gkeys = np.empty(total + 1, np.uint32)
    gfp = np.empty(total + 1, np.int64)
    n_out = C.c_int64(0)
    pk = (C.c_void_p * n)(*[k.ctypes.data for k in keys])
    pf = (C.c_void_p * n)(*[f.ctypes.data for f in fps])
    pl = (C.c_void_p * n)(*[l.ctypes.data for l in luts])
    rc = lib.rhccq_merge_palettes_host(n, pk, pf, counts.ctypes.data, int(_FP_NONE), gkeys.ctypes.data, gfp.ctypes.data, pl, C.byref(n_out))
    if rc != 0:
        raise RuntimeError(f"rhccq_merge_palettes_host failed ({rc})")
    m = int(n_out.value)
    maps = {}
    for c, lut in zip(comps, luts):
        for job, mp in c.maps.items():
            maps[job] = lut[mp]
    minr, minc, maxr, maxc = bbox
    return _Comp(gkeys[:m].copy(), gfp[:m].copy(), (minr, minc), (maxr - minr, maxc - minc), maps, True)


class FrameEncoder:
    def __init__(self, rh):
        self.rh = rh
        self.timings = {}

    def _t(self, name, t0, sync=False):
        if sync:
            torch.cuda.synchronize()
        self.timings[name] = self.timings.get(name, 0.0) + (time.perf_counter() - t0)

    def first_positions(self, S, fp_lut, n_entries):
        """int64[n_entries]: first frame-raster position of every entry fp_lut maps (job, rank) to
        (INT_MAX where an entry has no pixel).  The tiled (multi-GPU) encoder overrides this."""
        return self.first_positions_dev(S, fp_lut, n_entries).cpu().numpy().astype(np.int64)

    SORT_UNIQUE_MIN_JOBS = 2048   # beyond: unique colours by a device sort (rhccq_job_sort_unique) instead of 6 MiB of bitmap tables per job

    def _first_pos_pass(self, rh, S, fp, fp_lut, ci=None):
        """one pass over the pixels (of class `ci`, or of all classes): atomicMin of the raster position into fp[fp_lut[(job, rank)]]"""
        labels = S["labels"] if ci is None else S["labels"][ci:ci + 1]
        job_base = S["job_base"][:-1] if ci is None else S["job_base"][ci:ci + 1]
        if S.get("rankmap") is not None:
            rm = S["rankmap"] if ci is None else S["rankmap"][ci:ci + 1]
            rh.job_index_ranked(S["H"], S["W"], labels, job_base, rm, S["d_pal_off"], fp, fp_lut)
        elif ci is not None and S.get("e1map") is not None:
            # (pipelined classes: keep every pixel's level-1 entry for the final remap -- it then needs no colour, rank or lut1 gather)
            rh.job_index_entries(S["rgb"], labels, job_base, S["bitmaps"], S["prefix"], S["d_pal_off"], S["fix_key"], fp, fp_lut, S["e1map"][ci])
        else:
            rh.job_index(S["rgb"], labels, job_base, S["bitmaps"], S["prefix"], S["d_pal_off"], S["fix_key"], want_idx=False, first_pos=fp, fp_lut=fp_lut)

    def _remap_pass(self, S, lut1, default_index, out_dtype, d_lut2):
        rh = self.rh
        if S.get("e1map") is not None and S.get("e1map_complete"):
            return rh.frame_remap_entries(S["H"], S["W"], S["labels"], S["job_base"][:-1], S["e1map"], default_index, out_dtype, lut2=d_lut2)
        if S.get("rankmap") is not None:
            return rh.frame_remap_ranked(S["H"], S["W"], S["labels"], S["job_base"][:-1], S["rankmap"], S["d_pal_off"], lut1, default_index, out_dtype, lut2=d_lut2)
        return rh.frame_remap(S["rgb"], S["labels"], S["job_base"][:-1], S["bitmaps"], S["prefix"], S["d_pal_off"], S["fix_key"], lut1, default_index,
                              out_dtype, lut2=d_lut2)

    def first_positions_dev(self, S, fp_lut, n_entries):
        """the same as a device tensor (int32[n_entries], positions in THIS encoder's pixel raster)"""
        rh = self.rh
        fp = torch.full((max(n_entries, 1),), INT_MAX, dtype=torch.int32, device=rh.device)
        self._first_pos_pass(rh, S, fp, fp_lut)
        return fp[:n_entries]

    # ------------------------------------------------------------------------------------------
    def prepare(self, rgb, classes):
        """Per-pixel passes 1-3 (K0, K0b, K1): per-segment stats, colour bitmaps, sorted palettes and
        the first raster position of every palette entry.  Returns the frame state dict."""
        rh = self.rh
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous()
        labels = [c.labels for c in classes]
        job_base = np.concatenate([[0], np.cumsum([c.n_seg for c in classes])]).astype(np.int64)
        n_jobs = int(job_base[-1])
        if n_jobs == 0:
            raise ValueError("no segments")
        t0 = time.perf_counter()
        # frames cut into very many segments: 6 MiB of bitmap tables per job stops fitting; one device sort of (job, colour) keys
        # bounds the memory by the pixel count instead (the rank of every pixel's colour is then STORED, not recomputed)
        sort_path = n_jobs > self.SORT_UNIQUE_MIN_JOBS
        if sort_path:
            bitmaps, stats = None, rh.job_stats(rgb, labels, job_base[:-1], n_jobs)
        else:
            bitmaps, stats = rh.new_job_state(n_jobs)
            rh.job_scan(rgb, labels, job_base[:-1], bitmaps, stats, black_is_colour=False)
        st = stats.cpu().numpy().astype(np.int64)
        self._t("scan", t0)
        t0 = time.perf_counter()
        count, n_black = st[:, 4], st[:, 5]
        present = count > 0
        # crop = tight bbox +-2 px clamped to the region (subregions.py:346-352)
        job_class = np.repeat(np.arange(len(classes)), [c.n_seg for c in classes])
        job_region = np.concatenate([c.seg_region for c in classes])
        rb = np.concatenate([c.region_bbox[c.seg_region] for c in classes])
        r0 = np.maximum(rb[:, 0], st[:, 0] - 2)
        r1 = np.minimum(rb[:, 2] - 1, st[:, 1] + 2)
        c0 = np.maximum(rb[:, 1], st[:, 2] - 2)
        c1 = np.minimum(rb[:, 3] - 1, st[:, 3] + 2)
        crop_area = (r1 - r0 + 1) * (c1 - c0 + 1)
        has_bg = present & (crop_area > count)
        needs_fix = present & (n_black > 0) & (count > n_black)           # subregions.py:393-421
        all_black = present & (n_black > 0) & (count == n_black)
        fix_key = None
        if needs_fix.any():
            best = torch.full((n_jobs,), -1, dtype=torch.int64, device=rh.device)      # ~0ull
            rh.job_blackfix(rgb, labels, job_base[:-1], rh.dev(needs_fix.astype(np.uint8)), best)
            pos = (best.cpu().numpy().view(np.uint64) & np.uint64((1 << 40) - 1)).astype(np.int64)
            px = rgb.reshape(-1, 3)[torch.from_numpy(np.where(needs_fix, pos, 0)).to(rh.device)].cpu().numpy().astype(np.uint32)
            fk = ((px[:, 0] << 16) | (px[:, 1] << 8) | px[:, 2]).astype(np.uint32)
            fk[~needs_fix] = 0
            fix_key = rh.dev(fk.view(np.int32))
        rankmap = None
        if sort_path:
            P, keys_dev, rankmap = rh.job_sort_unique(rgb, labels, job_base[:-1], n_jobs, fix_key, np.nonzero(has_bg | all_black)[0])
            pal_off = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
            d_pal_off = rh.dev(pal_off[:-1].copy())
            total = int(pal_off[-1])
            prefix = None
        else:
            rh.job_set_black(bitmaps, np.nonzero(has_bg | all_black)[0])
            chunk, counts = rh.bitmap_count(bitmaps)
            P = counts.cpu().numpy().astype(np.int64)
            pal_off = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
            d_pal_off = rh.dev(pal_off[:-1].copy())
            total = int(pal_off[-1])
            prefix, keys_dev = rh.bitmap_emit(bitmaps, chunk, d_pal_off, total)
        self._t("unique", t0)
        return {"H": H, "W": W, "rgb": rgb, "classes": classes, "labels": labels, "job_base": job_base, "n_jobs": n_jobs, "rankmap": rankmap,
                "bitmaps": bitmaps, "prefix": prefix, "pal_off": pal_off, "d_pal_off": d_pal_off, "fix_key": fix_key,
                "keys_dev": keys_dev, "has_black": has_bg | all_black, "P": P, "present": present, "job_class": job_class,
                "job_region": job_region, "crop": (r0, r1, c0, c1), "total": total}

    # ---- the three levels as (collect jobs) / (apply results) pairs, so that the palettes of several frames
    # can share one batched clustering launch per level (encode_batch) ---------------------------------------
    def level1_jobs(self, S, only_class=None):
        """Clustering jobs of every segment palette of the frame (subregions.py:426-449); `only_class`: of one class."""
        classes, pal_off = S["classes"], S["pal_off"]
        jobs, job_ids = [], []
        # palettes of >= 10 000 colours stay in HBM (MiniBatch branch); the small ones (DBSCAN branch) are
        # brought to the host in ONE copy
        nblk = S["has_black"].astype(np.int64)
        present = S["present"] if only_class is None else (S["present"] & (S["job_class"] == only_class))
        big = present & (S["P"] - nblk >= 10000)
        small_ids = np.nonzero(present & ~big)[0]
        host_keys = {}
        if len(small_ids):
            lo, hi = int(pal_off[small_ids[0]]), int(pal_off[small_ids[-1] + 1])
            chunk = S["keys_dev"][lo:hi].cpu().numpy().view(np.uint32)
            for j in small_ids:
                host_keys[int(j)] = chunk[pal_off[j] - lo:pal_off[j + 1] - lo]
        for j in np.nonzero(present)[0]:
            q = classes[S["job_class"][j]].quality
            eps, _, mc = clustering_params(int(S["P"][j]), q)
            jb = {"quality": q, "eps": eps, "mc": mc}
            if big[j]:
                jb["keys_dev"] = S["keys_dev"][pal_off[j]:pal_off[j + 1]]
                jb["has_black"] = bool(S["has_black"][j])
            else:
                jb["keys"] = host_keys[int(j)]
            jobs.append(jb)
            job_ids.append(j)
        return jobs, job_ids

    def level1_finish(self, S, job_ids, res):
        """lut1, first positions of the clustered entries, merge per region (subregions.py:634-679).
        Returns per class the list of region components."""
        rh = self.rh
        classes, pal_off, job_base = S["classes"], S["pal_off"], S["job_base"]
        r0, r1, c0, c1 = S["crop"]
        # first raster position of every CLUSTERED palette entry (fixes the first-seen order of the merges,
        # merging.py:77-79): one streaming pass whose atomicMin table is only sum(K_j) entries
        seg_comp = {}
        new_off = np.concatenate([[0], np.cumsum([len(nk) for nk, _, _ in res])]).astype(np.int64)
        any_resident = any(mp is None for _, mp, _ in res)
        if not any_resident:                                                # all mappings on the host: one upload
            lut1_host = np.zeros(max(S["total"], 1), np.int32)
            for i, (j, (nk, mp, info)) in enumerate(zip(job_ids, res)):
                lut1_host[pal_off[j]:pal_off[j + 1]] = new_off[i] + mp
            lut1 = rh.dev(lut1_host)
        else:
            lut1 = torch.zeros((max(S["total"], 1),), dtype=torch.int32, device=rh.device)
            for i, (j, (nk, mp, info)) in enumerate(zip(job_ids, res)):
                if mp is None:                                              # mapping resident on the device
                    lut1[pal_off[j]:pal_off[j + 1]] = info["mapping_dev"] + int(new_off[i])
                else:
                    lut1[pal_off[j]:pal_off[j + 1]] = rh.dev((new_off[i] + mp).astype(np.int32))
        S["lut1"] = lut1                                                   # (job, rank) -> global clustered-palette entry
        S["k1_off"] = {int(j): (int(new_off[i]), int(new_off[i + 1])) for i, j in enumerate(job_ids)}
        S["k1_total"] = int(new_off[-1])
        fp_new_all = self.first_positions(S, S["lut1"], int(new_off[-1]))
        for i, (j, (nk, mp, info)) in enumerate(zip(job_ids, res)):
            seg_comp[j] = _Comp(nk, fp_new_all[new_off[i]:new_off[i + 1]], (int(r0[j]), int(c0[j])),
                                (int(r1[j] - r0[j] + 1), int(c1[j] - c0[j] + 1)), {int(j): np.arange(len(nk), dtype=np.int32)}, False)
        per_class = []
        for ci, cls in enumerate(classes):
            regs = []
            for r in range(len(cls.region_bbox)):
                segs = [seg_comp[j] for j in range(job_base[ci], job_base[ci + 1]) if S["job_region"][j] == r and j in seg_comp]
                # subregions.py:637-679: > 1 components are merged on the region canvas, one is kept as is
                regs.append(_merge(segs, tuple(int(v) for v in cls.region_bbox[r])) if segs else None)
            per_class.append(regs)
        return per_class

    def level1(self, S):
        t0 = time.perf_counter()
        jobs, job_ids = self.level1_jobs(S)
        res = cluster_palettes(self.rh, jobs)
        self._t("level1_cluster", t0)
        t0 = time.perf_counter()
        per_class = self.level1_finish(S, job_ids, res)
        self._t("merge1", t0)
        return per_class

    def level2_jobs(self, S, per_class):
        """merge per class on the frame canvas (regions.py:34-46) -> one clustering job per class."""
        H, W = S["H"], S["W"]
        lvl2, q2s = [], []
        for ci, cls in enumerate(S["classes"]):
            regs = [r for r in per_class[ci] if r is not None]
            q2 = min(cls.quality * 2, 100)
            q2s.append(q2)
            if not regs:
                continue                                                                  # rhccq.ipynb:1009-1013
            lvl2.append((ci, _merge(regs, (0, 0, H, W)), q2))
        jobs2 = []
        for ci, comp, q2 in lvl2:
            eps, _, mc = clustering_params(len(comp.keys), q2)
            jobs2.append({"keys": comp.keys, "quality": q2, "eps": eps, "mc": mc})
        return lvl2, q2s, jobs2

    @staticmethod
    def level2_finish(lvl2, res2):
        comps3 = []
        for (ci, comp, q2), (nk, mp, info) in zip(lvl2, res2):
            fp_new = _scatter_min(len(nk), mp, comp.fp)
            comps3.append(_Comp(nk, fp_new, comp.top_left, comp.shape, {job: mp[m] for job, m in comp.maps.items()}, comp.merged))
        return comps3

    def level3_job(self, S, comps3, q2s):
        """merge ROI + non-ROI (image.py:246-256) -> the level-3 clustering job."""
        if not comps3:
            raise IndexError("no components")
        q3 = min(sum(q2s), 100)
        m3c = _merge(comps3, (0, 0, S["H"], S["W"]))
        eps, _, mc = clustering_params(len(m3c.keys), q3)
        return m3c, q3, {"keys": m3c.keys, "quality": q3, "eps": eps, "mc": mc}

    def finish(self, S, comps3, m3c, q3, res3, want_levels=False, levels=None, profile=False):
        """compose levels 2-3 into lut2 and run the final per-pixel remap."""
        rh = self.rh
        H, W = S["H"], S["W"]
        fk3, mp3, info3 = res3
        multi = len(comps3) > 1
        t0 = time.perf_counter()
        lut2 = np.full(max(S["k1_total"], 1), -1, np.int32)
        for c2 in comps3:
            for job, m in c2.maps.items():
                if multi:
                    painted = c2.keys[m] != 0                           # black = transparent (merging.py:75)
                    v = np.where(painted, mp3[m3c.maps[job]], -1)
                else:
                    v = mp3[m]
                a, b = S["k1_off"][job]
                lut2[a:b] = v
        if multi:
            default_index = int(mp3[0])
        else:
            blk = np.nonzero(fk3 == 0)[0]
            default_index = int(blk[0]) if len(blk) else 0
        max_index = int(max(lut2.max(initial=0), default_index))
        out_dtype = torch.uint8 if max_index < 256 else (torch.int16 if max_index < 65536 else torch.int32)
        dtype_name = "uint8" if max_index < 256 else ("uint16" if max_index < 65536 else "uint32")
        d_lut2 = rh.dev(lut2)
        self._t("compose", t0)
        t0 = time.perf_counter()
        out = self._remap_pass(S, S["lut1"], default_index, out_dtype, d_lut2)
        self._t("remap", t0, sync=profile)
        result = {"palette": unpack_rgb(fk3), "indices": out, "indices_dtype": dtype_name,
                  "shape": (H, W) if multi else tuple(m3c.shape), "top_left": (0, 0) if multi else tuple(m3c.top_left),
                  "quality3": q3, "n_unique": S["P"], "info3": info3}
        if want_levels:
            levels["state"] = S
            result["levels"] = levels
        return result

    # ---- levels 1 and 2 of the region classes side by side ------------------------------------------------------
    # Nothing of a class's level 1 -> merge -> level 2 chain depends on the other class (regions.py:9-70 is called once per
    # class, rhccq.ipynb:1001-1013); only quantize_image (level 3) needs both.  On a 4K frame one class is usually done long
    # before the other (its k-means++ chains are shorter, or the other class holds a straggling mini-batch problem), so each
    # class runs as a pipeline of its own -- host thread, HIP stream, sibling context -- and its merge bookkeeping and level 2
    # disappear behind the slower class's level 1.
    PIPELINE_CLASSES = True       # (the tiled multi-GPU encoder turns it off: its collectives must stay in one order on every rank)
    TABLE_BOUND = (2, 64)         # room reserved per MiniBatch-branch job in the frame-wide tables: a * (clusters + black) + b entries

    def _class_pipeline(self, S, ci, rhc, table_base, table_room, lut1, fp_all):
        """level 1 -> merge per region -> merge per class -> level 2 of class `ci` on context `rhc` (current stream: its own)"""
        classes, pal_off, job_base = S["classes"], S["pal_off"], S["job_base"]
        r0, r1, c0, c1 = S["crop"]
        tm, t_prev = {}, time.perf_counter()

        def mark(name):
            nonlocal t_prev
            t = time.perf_counter()
            tm[name] = tm.get(name, 0.0) + (t - t_prev)
            t_prev = t
        jobs, job_ids = self.level1_jobs(S, only_class=ci)
        res = cluster_palettes(rhc, jobs)
        mark("level1_cluster")
        # entries of this class's clustered level-1 palettes live at table_base[ci] + ...: a slice of the frame-wide tables
        # (lut1 values, first positions, lut2) fixed BEFORE the clustering from an upper bound of the palette sizes
        new_off = table_base + np.concatenate([[0], np.cumsum([len(nk) for nk, _, _ in res])]).astype(np.int64)
        if int(new_off[-1]) - table_base > table_room:
            raise _TableOverflow()                               # (k-means splits blew a palette past its bound: the caller goes serial)
        for i, (j, (nk, mp, info)) in enumerate(zip(job_ids, res)):
            if mp is None:
                lut1[pal_off[j]:pal_off[j + 1]] = info["mapping_dev"] + int(new_off[i])
            else:
                lut1[pal_off[j]:pal_off[j + 1]] = rhc.dev((new_off[i] + mp).astype(np.int32))
        k1_off = {int(j): (int(new_off[i]), int(new_off[i + 1])) for i, j in enumerate(job_ids)}
        # first raster position of every clustered entry of THIS class: one pass over the class's label map
        self._first_pos_pass(rhc, S, fp_all, lut1, ci)
        lo, hi = int(new_off[0]), int(new_off[-1])
        fp_new = fp_all[lo:hi].cpu().numpy().astype(np.int64) if hi > lo else np.zeros(0, np.int64)
        mark("first_positions")
        seg_comp = {}
        for i, (j, (nk, mp, info)) in enumerate(zip(job_ids, res)):
            seg_comp[j] = _Comp(nk, fp_new[new_off[i] - lo:new_off[i + 1] - lo], (int(r0[j]), int(c0[j])),
                                (int(r1[j] - r0[j] + 1), int(c1[j] - c0[j] + 1)), {int(j): np.arange(len(nk), dtype=np.int32)}, False)
        cls = classes[ci]
        regs = []
        for r in range(len(cls.region_bbox)):
            segs = [seg_comp[j] for j in range(job_base[ci], job_base[ci + 1]) if S["job_region"][j] == r and j in seg_comp]
            regs.append(_merge(segs, tuple(int(v) for v in cls.region_bbox[r])) if segs else None)
        live = [r for r in regs if r is not None]
        q2 = min(cls.quality * 2, 100)
        comp3 = None
        if live:                                                                              # rhccq.ipynb:1009-1013
            comp = _merge(live, (0, 0, S["H"], S["W"]))
            mark("merge")
            eps, _, mc = clustering_params(len(comp.keys), q2)
            (res2,) = cluster_palettes(rhc, [{"keys": comp.keys, "quality": q2, "eps": eps, "mc": mc}])
            mark("level2_cluster")
            (comp3,) = self.level2_finish([(ci, comp, q2)], [res2])
            mark("level2_finish")
        self.class_timings[ci] = tm
        return k1_off, regs, comp3, q2

    def _encode_pipelined(self, S):
        rh = self.rh
        classes = S["classes"]
        nblk = S["has_black"].astype(np.int64)
        # upper bound of every job's clustered palette: ceil(N q / 1000) clusters (+ black) in the MiniBatch branch, the palette
        # itself otherwise
        bound = np.zeros(S["n_jobs"], np.int64)
        for j in np.nonzero(S["present"])[0]:
            n_col = int(S["P"][j] - nblk[j])
            q = classes[S["job_class"][j]].quality
            bound[j] = (min(int(S["P"][j]), self.TABLE_BOUND[0] * (-(-n_col * q // 1000) + int(nblk[j])) + self.TABLE_BOUND[1])
                        if n_col >= 10000 else int(S["P"][j]))
        base = [int(bound[:S["job_base"][ci]].sum()) for ci in range(len(classes))]
        room = [int(bound[S["job_base"][ci]:S["job_base"][ci + 1]].sum()) for ci in range(len(classes))]
        total = int(bound.sum())
        lut1 = torch.zeros((max(S["total"], 1),), dtype=torch.int32, device=rh.device)
        fp_all = torch.full((max(total, 1),), INT_MAX, dtype=torch.int32, device=rh.device)
        if S.get("rankmap") is None:
            S["e1map"] = torch.empty((len(classes), S["H"] * S["W"]), dtype=torch.int32, device=rh.device)
        here = torch.cuda.current_stream(rh.device)
        ready = torch.cuda.Event()
        ready.record(here)
        out, errors = [None] * len(classes), []
        self.class_timings = {}

        def run(ci):
            try:
                torch.cuda.set_device(rh.device)
                stream, rhc = rh._lane(("class", ci))
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)
                    out[ci] = self._class_pipeline(S, ci, rhc, base[ci], room[ci], lut1, fp_all)
                    done = torch.cuda.Event()
                    done.record(stream)
                    out[ci] += (done,)
            except BaseException as e:                        # surfaced to the caller below
                errors.append(e)

        from .ops import lane_pool
        futs = [lane_pool("class").submit(run, ci) for ci in range(len(classes))]
        for f in futs:
            f.result()
        if errors:
            torch.cuda.synchronize(rh.device)                     # (the other class may still be writing the shared tables)
            raise errors[0]
        S["lut1"], S["k1_off"], S["k1_total"] = lut1, {}, total
        S["e1map_complete"] = S.get("e1map") is not None       # every class has stored its pixels' level-1 entries
        per_class, comps3, q2s = [], [], []
        for k1_off, regs, comp3, q2, done in out:
            here.wait_event(done)
            S["k1_off"].update(k1_off)
            per_class.append(regs)
            q2s.append(q2)
            if comp3 is not None:
                comps3.append(comp3)
        # (lut1, fp_all and e1map were allocated on `here` and written on the class streams: every later use or free on `here`
        # is ordered behind those writes by the wait_event(done) above -- no record_stream needed)
        return per_class, comps3, q2s

    def encode(self, rgb, classes, want_levels=False, profile=False):
        """rgb: uint8[H,W,3] device tensor; classes: [ClassSpec] in precedence order (ROI first).
        Returns dict(palette uint8[K,3], indices (device tensor [H,W], dtype by max index),
        indices_dtype, shape, top_left, levels (optional))."""
        rh = self.rh
        self.timings = {}
        S = self.prepare(rgb, classes)
        if self.PIPELINE_CLASSES and not profile and len(classes) > 1 and all(S["present"][S["job_class"] == ci].any() for ci in range(len(classes))):
            t0 = time.perf_counter()
            try:
                per_class, comps3, q2s = self._encode_pipelined(S)
            except _TableOverflow:
                # a class's k-means splits outgrew its slice of the tables (the bound is a heuristic): both pipelines have
                # finished (or been drained), their half-written tables are dropped and the serial path redoes the levels
                per_class = None
                for key in ("e1map", "e1map_complete", "lut1", "k1_off", "k1_total"):
                    S.pop(key, None)
                self.table_overflows = getattr(self, "table_overflows", 0) + 1
            if per_class is not None:
                self._t("levels_1_2_per_class", t0)
                levels = {"level1": per_class, "level2": comps3} if want_levels else None
                t0 = time.perf_counter()
                m3c, q3, job3 = self.level3_job(S, comps3, q2s)
                (res3,) = cluster_palettes(rh, [job3])
                self._t("level3", t0)
                return self.finish(S, comps3, m3c, q3, res3, want_levels, levels, profile)
        per_class = self.level1(S)
        t0 = time.perf_counter()
        lvl2, q2s, jobs2 = self.level2_jobs(S, per_class)
        self._t("merge2", t0)
        t0 = time.perf_counter()
        comps3 = self.level2_finish(lvl2, cluster_palettes(rh, jobs2))
        levels = {"level1": per_class, "level2": comps3} if want_levels else None
        self._t("level2", t0)
        t0 = time.perf_counter()
        m3c, q3, job3 = self.level3_job(S, comps3, q2s)
        (res3,) = cluster_palettes(rh, [job3])
        self._t("level3", t0)
        return self.finish(S, comps3, m3c, q3, res3, want_levels, levels, profile)

    def encode_native(self, rgb, classes):
        """The same frame through rhccq_encode_frame: frame.py + palette.py + the MiniBatchKMeans driver as native host code behind ONE C
        entry (csrc/encode_frame.hip) -- region classes and MiniBatchKMeans problems on std::threads with HIP streams of
        their own, no interpreter on the path.  Same result dict as encode() (bit-identical: tests/test_gpu_frame.py); `timings` /
        `class_timings` receive the entry's own host clocks."""
        import ctypes as C
        from ._lib import ClassDesc, FrameResult
        rh = self.rh
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous()
        descs = (ClassDesc * len(classes))()
        keep = []
        for d, c in zip(descs, classes):
            assert c.labels.dtype == torch.int32 and c.labels.is_contiguous() and c.labels.numel() == H * W
            sr = np.ascontiguousarray(c.seg_region, dtype=np.int32)
            rb = np.ascontiguousarray(c.region_bbox, dtype=np.int32).reshape(-1, 4)
            keep += [sr, rb]
            d.labels, d.n_seg, d.n_region = c.labels.data_ptr(), int(c.n_seg), int(len(rb))
            d.seg_region, d.region_bbox, d.quality = sr.ctypes.data, rb.ctypes.data, int(c.quality)
        n_jobs = sum(int(c.n_seg) for c in classes)
        out = torch.empty((H * W,), dtype=torch.int32, device=rh.device)
        n_unique = np.zeros(max(n_jobs, 1), np.int64)
        res = FrameResult()
        pal_cap = 1 << 16
        while True:
            pal = np.empty((pal_cap, 3), np.uint8)
            rc = rh.lib.rhccq_encode_frame(rh.ctx, rh._p(rgb), H, W, descs, len(classes), pal.ctypes.data, pal_cap, rh._p(out), n_unique.ctypes.data,
                                           C.byref(res))
            if rc == -3 and res.n_colours > pal_cap:             # RHCCQ_E_LIMIT: the palette is larger than the buffer
                pal_cap = int(res.n_colours)
                continue
            rh._check(rc, "encode_frame")
            break
        eb = int(res.index_bytes)
        dt = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[eb]
        idx = out.view(torch.uint8)[:H * W * eb].view(dt).reshape(H, W)
        names = ("scan", "unique", "levels_1_2_per_class", "level3", "compose", "remap", "total")
        self.timings = {n: res.ms[i] * 1e-3 for i, n in enumerate(names)}
        cn = ("level1_cluster", "first_positions_merge", "level2_cluster", "level2_finish")
        self.class_timings = {ci: {n: res.class_ms[ci][i] * 1e-3 for i, n in enumerate(cn)} for ci in range(min(len(classes), 4))}
        return {"palette": pal[:int(res.n_colours)].copy(), "indices": idx, "indices_dtype": {1: "uint8", 2: "uint16", 4: "uint32"}[eb],
                "shape": (int(res.shape[0]), int(res.shape[1])), "top_left": (int(res.top_left[0]), int(res.top_left[1])),
                "quality3": int(res.quality3), "n_unique": n_unique[:n_jobs]}

    def encode_batch(self, frames):
        """frames: [(rgb, classes)].  Same results as encode() frame by frame, but the palettes of all frames
        go through ONE batched clustering call per level: the sequential k-means++ chains of all segments of
        all frames run side by side (one workgroup each) instead of leaving 250 CUs idle."""
        rh = self.rh
        self.timings = {}
        t0 = time.perf_counter()
        states = [self.prepare(rgb, classes) for rgb, classes in frames]
        l1 = [self.level1_jobs(S) for S in states]
        flat = [jb for jobs, _ in l1 for jb in jobs]
        res = cluster_palettes(rh, flat)
        self._t("level1_cluster", t0)
        t0 = time.perf_counter()
        per_class, o = [], 0
        for S, (jobs, job_ids) in zip(states, l1):
            per_class.append(self.level1_finish(S, job_ids, res[o:o + len(jobs)]))
            o += len(jobs)
        l2 = [self.level2_jobs(S, pc) for S, pc in zip(states, per_class)]
        res2 = cluster_palettes(rh, [jb for _, _, jobs2 in l2 for jb in jobs2])
        comps3, o = [], 0
        for lvl2, q2s, jobs2 in l2:
            comps3.append(self.level2_finish(lvl2, res2[o:o + len(jobs2)]))
            o += len(jobs2)
        self._t("level2", t0)
        t0 = time.perf_counter()
        l3 = [self.level3_job(S, c3, q2s) for S, c3, (_, q2s, _) in zip(states, comps3, l2)]
        res3 = cluster_palettes(rh, [job for _, _, job in l3])
        self._t("level3", t0)
        return [self.finish(S, c3, m3c, q3, r3) for S, c3, (m3c, q3, _), r3 in zip(states, comps3, l3, res3)]

    def render_component(self, S, comp):
        """Index map (device int32 [h,w]) of one level-1/level-2 component as the reference would
        return it: canvas semantics for merged components (uncovered = 0 = black), crop semantics for a
        single segment (background = the palette's black entry)."""
        rh = self.rh
        lut2 = np.full(max(S["k1_total"], 1), -1, np.int32)
        for job, m in comp.maps.items():
            a, b = S["k1_off"][job]
            lut2[a:b] = m
        default = 0
        if not comp.merged:
            blk = np.nonzero(comp.keys == 0)[0]
            default = int(blk[0]) if len(blk) else 0
        full = self._remap_pass(S, S["lut1"], default, torch.int32, rh.dev(lut2))
        r, c = comp.top_left
        h, w = comp.shape
        return full[r:r + h, c:c + w].contiguous()
