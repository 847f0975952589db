"""Palette-space host logic of the RHCCQ hierarchy (Python, as in the reference) over the HIP kernels.

  cluster_palettes   encoder/compression/clustering.py:160-437 (cluster_palette_colors_parallel), batched
  merge_components   encoder/compression/merging.py:8-120      (merge_region_components_simple)

Ordering rules reproduced: black rows first; clusters with <= mc colours in
ascending label order; oversize clusters in ascending label order (the reference uses thread
completion order), each contributing its k-means children depth-first; old->new mapping of split
children through the first equal palette row (find_color_index, clustering.py:803-808); mapping
table stored as uint16 (clustering.py:373).
"""
import logging
import math

import numpy as np
import torch

from .hostsort import _stable_order
from .ops import MINIBATCH_THRESHOLD, clustering_params, pack_rgb, unpack_rgb

__all__ = ["cluster_palettes", "cluster_palette", "merge_components", "clustering_params"]

log = logging.getLogger("rhccq")
MAPPING_ENTRIES = 1 << 16     # the reference's mapping_array is uint16 (clustering.py:373): new indices beyond 65 535 wrap


def _flag_wrap(info, n_new):
    """a clustered palette of more than 65 536 entries does not fit the reference's uint16 mapping_array --
    its indices wrap modulo 65 536 there, and here (bit parity); the deviation a caller may want to know about is flagged"""
    if n_new > MAPPING_ENTRIES:
        info["mapping_wrapped"] = True
        log.warning("clustered palette has %d entries: indices above 65535 wrap as in the reference's uint16 mapping_array "
                    "(encoder/compression/clustering.py:373)", n_new)
    return info


class _Node:
    __slots__ = ("seg", "members", "children", "leaf_id")

    def __init__(self, seg, members):
        self.seg, self.members, self.children, self.leaf_id = seg, members, None, -1


def _n_splits(n, mc):
    """split_large_cluster's n_splits (clustering.py:739-747); 0 = do not split."""
    if n <= mc:
        return 0
    ns = max(2, (n + mc - 1) // mc)
    ns = min(ns, n)
    if n <= 2 or ns < 2:
        return 0
    return ns


def _cluster_resident(rh, jobs, idxs, results):
    """MiniBatch-branch jobs whose sorted palette is resident on the device (`keys_dev`, black only at
    index 0): labels, member sums and the palette -> new-palette mapping stay in HBM; only k-sized
    tables cross PCIe.  Jobs that turn out to have an oversize cluster fall back to the host path."""
    fallback = []
    ks, parts = [], []
    for s in idxs:
        jb = jobs[s]
        nbk = jb["keys_dev"][1:] if jb["has_black"] else jb["keys_dev"]
        parts.append(nbk)
        ks.append(math.ceil(int(nbk.numel()) * (jb["quality"] / 100) / 10))
    labs = rh.minibatch_kmeans(parts, ks, return_device=True)
    # member sums of every cluster of every job in ONE launch and ONE copy (labels offset by the jobs' cluster offsets)
    koff = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
    if len(parts) > 1:
        all_keys = torch.cat(parts)
        all_lab = torch.cat([lab + int(koff[i]) for i, lab in enumerate(labs)])
    else:
        all_keys, all_lab = parts[0], labs[0]
    _, sums_dev = rh.cluster_means(all_keys, all_lab, int(koff[-1]))
    sums_all = np.ascontiguousarray(sums_dev.cpu().numpy())
    luts, plans = [], []
    for i, (s, nbk, k, lab) in enumerate(zip(idxs, parts, ks, labs)):
        jb = jobs[s]
        sums = sums_all[koff[i]:koff[i + 1]]
        nblack = 1 if jb["has_black"] else 0
        new_keys = np.empty(nblack + k, np.uint32)
        lut = np.empty(k, np.int32)
        # floor means of the non-empty clusters in label order + the uint16 mapping_array (clustering.py:373): one native pass
        n_present = int(rh.lib.rhccq_cluster_plan_host(sums.ctypes.data, k, int(jb["mc"]), nblack, new_keys.ctypes.data, lut.ctypes.data))
        if n_present == -1:                                # a cluster needs k-means splitting: host path
            jb["keys"] = jb["keys_dev"].cpu().numpy().view(np.uint32)
            jb["_labels"] = lab.cpu().numpy()
            fallback.append(s)
            luts.append(np.zeros(k, np.int32))
            plans.append(None)
            continue
        if n_present < 0:
            raise RuntimeError("rhccq_cluster_plan_host: bad argument")
        luts.append(lut)
        plans.append((new_keys[:nblack + n_present].copy(), nblack, n_present))
    # label -> new palette index of every job in one gather
    mapped_all = rh.remap(all_lab, rh.dev(np.concatenate(luts)))
    offs = np.concatenate([[0], np.cumsum([int(p.numel()) for p in parts])]).astype(np.int64)
    for i, s in enumerate(idxs):
        if plans[i] is None:
            continue
        new_keys, nblack, n_present = plans[i]
        mapped = mapped_all[offs[i]:offs[i + 1]]
        mapping_dev = torch.cat([torch.zeros(1, dtype=torch.int32, device=rh.device), mapped]) if nblack else mapped
        results[s] = (new_keys, None, _flag_wrap({"branch": "minibatch", "n_clusters": n_present, "n_large": 0, "mapping_dev": mapping_dev},
                                                 nblack + n_present))
    return fallback


def cluster_palettes(rh, jobs):
    """jobs: list of dicts {keys: uint32[P] palette keys in palette order, quality, eps, mc}; a job may
    instead carry {keys_dev: device int32[P] sorted keys, has_black} (then its mapping comes back as a
    device tensor in info["mapping_dev"] and the host mapping is None).
    Returns per job (new_keys uint32[K], mapping uint16-valued int32[P] or None, info dict)."""
    S = len(jobs)
    pre = {}
    resident = [s for s, jb in enumerate(jobs) if "keys_dev" in jb and
                int(jb["keys_dev"].numel()) - (1 if jb["has_black"] else 0) >= MINIBATCH_THRESHOLD]
    if resident:
        _cluster_resident(rh, jobs, resident, pre)
    for s, jb in enumerate(jobs):
        if "keys" not in jb and s not in pre:
            jb["keys"] = jb["keys_dev"].cpu().numpy().view(np.uint32)
    nb_idx, labels = [None] * S, [None] * S
    db_jobs, mb_jobs = [], []
    for s, jb in enumerate(jobs):
        if s in pre:
            continue
        keys = np.asarray(jb["keys"]).astype(np.uint32)
        jb["keys"] = keys
        nb_idx[s] = np.nonzero(keys != 0)[0]
        n = len(nb_idx[s])
        if n == 0:
            continue
        if "_labels" in jb:                                # resident job that fell back: labels already known
            labels[s] = jb.pop("_labels")
            continue
        (mb_jobs if n >= MINIBATCH_THRESHOLD else db_jobs).append(s)
    ms_jobs = [s for s in db_jobs if jobs[s].get("min_samples", 1) > 1]      # noise points exist (clustering.py:233-271)
    db_jobs = [s for s in db_jobs if s not in ms_jobs]
    for s in ms_jobs:
        labels[s] = rh.dbscan_labels(jobs[s]["keys"][nb_idx[s]], jobs[s]["eps"], jobs[s]["min_samples"])
    if db_jobs:
        labs, _ = rh.eps_components([jobs[s]["keys"][nb_idx[s]] for s in db_jobs], [jobs[s]["eps"] for s in db_jobs])
        for s, l in zip(db_jobs, labs):
            labels[s] = l
    if mb_jobs:
        ks = [math.ceil(len(nb_idx[s]) * (jobs[s]["quality"] / 100) / 10) for s in mb_jobs]
        labs = rh.minibatch_kmeans([jobs[s]["keys"][nb_idx[s]] for s in mb_jobs], ks)
        for s, l in zip(mb_jobs, labs):
            labels[s] = l
    # ---- classify clusters (vectorised: a 4K segment has ~30 000 of them), build split trees for the
    # oversize ones (breadth first on the device, depth-first output order)
    small_leaf = [None] * S        # per job: leaf id (within the job's small clusters) per label, -1 = oversize
    n_small = [0] * S
    noise_of = [None] * S          # per job: indices of the noise points (label -1): each keeps its own colour, listed right
    #                                behind the black rows in index order (clustering.py:262-271) = the first "small" leaves
    larges = [[] for _ in range(S)]
    frontier = []
    for s in range(S):
        if labels[s] is None:
            continue
        lab = labels[s]
        noise = np.nonzero(lab < 0)[0]
        noise_of[s] = noise
        if len(noise):
            lab = np.where(lab < 0, int(lab.max()) + 1, lab)   # parked behind the last real label, never "big", never "small"
        cnt = np.bincount(lab)
        if len(noise):
            cnt[-1] = 0
        present = cnt > 0
        big = cnt > jobs[s]["mc"]
        is_small = present & ~big
        sl = np.full(len(cnt), -1, np.int64)
        sl[is_small] = len(noise) + np.arange(int(is_small.sum()))          # ascending label order, behind the noise leaves
        small_leaf[s] = sl
        n_small[s] = len(noise) + int(is_small.sum())
        if big.any():
            order = _stable_order(lab)
            starts = np.concatenate([[0], np.cumsum(np.bincount(lab))])
            for l in np.nonzero(big)[0]:                       # ascending label; ascending index inside
                node = _Node(s, order[starts[l]:starts[l + 1]])
                larges[s].append(node)
                frontier.append(node)
    while frontier:
        todo = [(nd, _n_splits(len(nd.members), jobs[nd.seg]["mc"])) for nd in frontier]
        run = [(nd, k) for nd, k in todo if k > 0]
        frontier = []
        if not run:
            break
        labs = rh.kmeans_split([jobs[nd.seg]["keys"][nb_idx[nd.seg][nd.members]] for nd, _ in run], [k for _, k in run])
        # children of every node of this level at once: one stable sort of (node, label) instead of an argsort
        # per node (a 4K frame has ~10^2 split nodes per level, a batch of frames ~10^3)
        sizes = np.array([len(l) for l in labs], np.int64)
        cat = np.concatenate(labs).astype(np.int64)
        node_of = np.repeat(np.arange(len(run), dtype=np.int64), sizes)
        members = np.concatenate([nd.members for nd, _ in run])
        key = node_of * (int(cat.max()) + 1) + cat
        order = _stable_order(key) if int(key.max()) < (1 << 31) else np.argsort(key, kind="stable")   # ascending label; ascending index inside
        skey = key[order]
        starts = np.concatenate([[0], np.flatnonzero(skey[1:] != skey[:-1]) + 1, [len(skey)]])
        child_node = node_of[order[starts[:-1]]]
        smem = members[order]
        for nd, _ in run:
            nd.children = []
        for ci in range(len(starts) - 1):
            nd = run[child_node[ci]][0]
            ch = _Node(nd.seg, smem[starts[ci]:starts[ci + 1]])
            nd.children.append(ch)
            mc = jobs[nd.seg]["mc"]
            if len(ch.members) > mc and _n_splits(len(ch.members), mc) > 0:
                frontier.append(ch)
    # ---- leaves in reference order, floor means on the device (K2)
    results = []
    all_keys, all_leaf = [], []
    leaf_base = 0
    plan = []
    for s in range(S):
        if s in pre:
            plan.append("resident")
            continue
        keys = jobs[s]["keys"]
        black = np.nonzero(keys == 0)[0]
        if labels[s] is None:
            plan.append(None)
            continue
        split_leaves = []

        def walk(nd):
            if nd.children is None:
                split_leaves.append(nd.members)
            else:
                for ch in nd.children:
                    walk(ch)
        for nd in larges[s]:
            walk(nd)
        leaf_of = small_leaf[s][np.where(labels[s] < 0, 0, labels[s])]      # -1 where the point sits in an oversize cluster
        leaf_of[noise_of[s]] = np.arange(len(noise_of[s]))
        for li, rel in enumerate(split_leaves):
            leaf_of[rel] = n_small[s] + li
        n_leaves = n_small[s] + len(split_leaves)
        all_keys.append(keys[nb_idx[s]])
        all_leaf.append((leaf_of + leaf_base).astype(np.int32))
        plan.append((black, leaf_of, split_leaves, n_leaves, leaf_base))
        leaf_base += n_leaves
    means = None
    if leaf_base:
        dk = torch.from_numpy(np.concatenate(all_keys).view(np.int32)).to(rh.device)
        dl = torch.from_numpy(np.concatenate(all_leaf)).to(rh.device)
        means = rh.cluster_means(dk, dl, leaf_base)[0].cpu().numpy().view(np.uint32)
    for s in range(S):
        if s in pre:
            results.append(pre[s])
            continue
        keys = jobs[s]["keys"]
        P = len(keys)
        if plan[s] is None:                                # only black: returned unchanged (clustering.py:197-199)
            results.append((keys.copy(), np.arange(P, dtype=np.int32), {"branch": "none"}))
            continue
        black, leaf_of, split_leaves, n_leaves, base = plan[s]
        nb = nb_idx[s]
        mapping = np.zeros(P, np.int32)
        mapping[black] = np.arange(len(black))
        new_keys = np.concatenate([np.zeros(len(black), np.uint32), means[base:base + n_leaves]])
        mapping[nb] = (len(black) + leaf_of) & 0xFFFF       # uint16 mapping_array (clustering.py:373)
        if split_leaves and len(np.unique(keys)) != P:
            # find_color_index (clustering.py:803-808): a split child maps only the FIRST palette row equal
            # to each of its colours; later duplicate rows stay unmapped (0)
            order = _stable_order(keys)
            skeys = keys[order]
            for li, rel in enumerate(split_leaves):
                tgt = nb[rel]
                first = order[np.searchsorted(skeys, keys[tgt], side="left")]
                mapping[tgt[first != tgt]] = 0
        info = {"branch": "minibatch" if len(nb) >= MINIBATCH_THRESHOLD else "dbscan",
                "n_clusters": n_small[s] - len(noise_of[s]) + len(larges[s]), "n_large": len(larges[s]), "n_noise": int(len(noise_of[s]))}
        results.append((new_keys, mapping, _flag_wrap(info, len(new_keys))))
    return results


def cluster_palette(rh, quality, keys, eps, mc, min_samples=1):
    return cluster_palettes(rh, [{"keys": keys, "quality": quality, "eps": eps, "mc": mc, "min_samples": int(min_samples)}])[0]


def merge_components(rh, comps, bbox):
    """merge_region_components_simple on the device (K5).  comps: dicts with top_left, shape,
    palette (uint8[K,3] or list), indices (flat int array / device int32 tensor).  Returns None for
    no components, the component itself for one (merging.py:16-21), else a dict with numpy
    palette uint8[K,3] and a device int32 canvas under 'indices_dev' (+ lazily 'indices')."""
    if not comps:
        return None
    if len(comps) == 1:
        c = dict(comps[0])
        c["single"] = True
        return c
    minr, minc, maxr, maxc = (int(v) for v in bbox)
    H, W = maxr - minr, maxc - minc
    canvas = torch.zeros((H, W), dtype=torch.int32, device=rh.device)
    colors = [0]
    lut_of = {0: 0}
    for seg in reversed(comps):
        h, w = (int(v) for v in seg["shape"])
        pal = np.asarray(seg["palette"], dtype=np.uint8).reshape(-1, 3)
        idx = seg.get("indices_dev")
        if idx is None:
            idx = torch.from_numpy(np.ascontiguousarray(np.asarray(seg["indices"]).reshape(-1).astype(np.int32))).to(rh.device)
        top, left = int(seg["top_left"][0]) - minr, int(seg["top_left"][1]) - minc
        if len(pal) == 0 or h * w == 0:
            continue
        fp = rh.merge_firstpos(idx, h, w, top, left, H, W, len(pal)).cpu().numpy()
        pk = pack_rgb(pal)
        lut = np.full(len(pal), -1, np.int32)
        seen = np.nonzero((fp != 2 ** 31 - 1) & (pk != 0))[0]
        for j in seen[np.argsort(fp[seen], kind="stable")]:   # first-seen order in the component's raster
            key = int(pk[j])
            if key not in lut_of:
                lut_of[key] = len(colors)
                colors.append(key)
        for j in seen:
            lut[j] = lut_of[int(pk[j])]
        rh.merge_paint(idx, h, w, top, left, canvas, torch.from_numpy(lut).to(rh.device), len(pal))
    out = {"top_left": (minr, minc), "shape": (H, W), "palette": unpack_rgb(np.array(colors, dtype=np.uint32)),
           "indices_dev": canvas.reshape(-1), "single": False}
    out["indices"] = rh.to_host(out["indices_dev"])        # (33 MB at 4K: through a page-locked buffer)
    return out
