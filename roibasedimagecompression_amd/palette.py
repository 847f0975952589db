"""Palette-space host logic of the RHCCQ hierarchy (Python, as in the reference) over the HIP kernels.

  cluster_palettes   encoder/compression/clustering.py:160-437 (cluster_palette_colors_parallel), batched
  merge_components   encoder/compression/merging.py:8-120      (merge_region_components_simple)

Ordering rules reproduced: black rows first; clusters with <= mc colours in
ascending label order; oversize clusters in ascending label order (the reference uses thread
completion order), each contributing its k-means children depth-first; old->new mapping of split
children through the first equal palette row (find_color_index, clustering.py:803-808); mapping
table stored as uint16 (clustering.py:373).
"""
import math

import numpy as np
import torch

from .ops import MINIBATCH_THRESHOLD, clustering_params, pack_rgb, unpack_rgb

__all__ = ["cluster_palettes", "cluster_palette", "merge_components", "clustering_params"]


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


def cluster_palettes(rh, jobs):
    """jobs: list of dicts {keys: uint32[P] palette keys in palette order, quality, eps, mc}.
    Returns per job (new_keys uint32[K], mapping uint16-valued int32[P], info dict)."""
    S = len(jobs)
    nb_idx, labels = [None] * S, [None] * S
    db_jobs, mb_jobs = [], []
    for s, jb in enumerate(jobs):
        keys = np.asarray(jb["keys"]).astype(np.uint32)
        jb["keys"] = keys
        nb_idx[s] = np.nonzero(keys != 0)[0]
        n = len(nb_idx[s])
        if n == 0:
            continue
        (mb_jobs if n >= MINIBATCH_THRESHOLD else db_jobs).append(s)
    if db_jobs:
        labs, _ = rh.eps_components([jobs[s]["keys"][nb_idx[s]] for s in db_jobs], [jobs[s]["eps"] for s in db_jobs])
        for s, l in zip(db_jobs, labs):
            labels[s] = l
    if mb_jobs:
        ks = [math.ceil(len(nb_idx[s]) * (jobs[s]["quality"] / 100) / 10) for s in mb_jobs]
        labs = rh.minibatch_kmeans([jobs[s]["keys"][nb_idx[s]] for s in mb_jobs], ks)
        for s, l in zip(mb_jobs, labs):
            labels[s] = l
    # ---- classify clusters, build split trees (breadth first on the device, depth-first output order)
    smalls, larges = [[] for _ in range(S)], [[] for _ in range(S)]
    frontier = []
    for s in range(S):
        if labels[s] is None:
            continue
        lab = labels[s]
        order = np.argsort(lab, kind="stable")
        bounds = np.flatnonzero(np.diff(lab[order])) + 1
        for rel in np.split(order, bounds):               # ascending label; ascending index inside
            if len(rel) > jobs[s]["mc"]:
                node = _Node(s, rel)
                larges[s].append(node)
                frontier.append(node)
            else:
                smalls[s].append(rel)
    while frontier:
        todo = [(nd, _n_splits(len(nd.members), jobs[nd.seg]["mc"])) for nd in frontier]
        run = [(nd, k) for nd, k in todo if k > 0]
        frontier = []
        if not run:
            break
        labs = rh.kmeans_split([jobs[nd.seg]["keys"][nb_idx[nd.seg][nd.members]] for nd, _ in run], [k for _, k in run])
        for (nd, k), lab in zip(run, labs):
            nd.children = []
            order = np.argsort(lab, kind="stable")
            bounds = np.flatnonzero(np.diff(lab[order])) + 1
            for sub in np.split(order, bounds):
                ch = _Node(nd.seg, nd.members[sub])
                nd.children.append(ch)
                if len(ch.members) > jobs[nd.seg]["mc"] and _n_splits(len(ch.members), jobs[nd.seg]["mc"]) > 0:
                    frontier.append(ch)
    # ---- leaves in reference order, floor means on the device (K2)
    results = []
    all_keys, all_leaf = [], []
    leaf_base = 0
    plan = []
    for s in range(S):
        keys = jobs[s]["keys"]
        P = len(keys)
        black = np.nonzero(keys == 0)[0]
        if labels[s] is None:
            plan.append(None)
            continue
        leaves = [(rel, False) for rel in smalls[s]]

        def walk(nd):
            if nd.children is None:
                leaves.append((nd.members, True))
            else:
                for ch in nd.children:
                    walk(ch)
        for nd in larges[s]:
            walk(nd)
        leaf_of = np.full(len(nb_idx[s]), -1, np.int32)
        for li, (rel, _) in enumerate(leaves):
            leaf_of[rel] = leaf_base + li
        all_keys.append(keys[nb_idx[s]])
        all_leaf.append(leaf_of)
        plan.append((black, leaves, leaf_base))
        leaf_base += len(leaves)
    means = None
    if leaf_base:
        dk = torch.from_numpy(np.concatenate(all_keys).astype(np.int64).astype(np.int32)).to(rh.device)
        dl = torch.from_numpy(np.concatenate(all_leaf)).to(rh.device)
        means = rh.cluster_means(dk, dl, leaf_base)[0].cpu().numpy().astype(np.uint32)
    for s in range(S):
        keys = jobs[s]["keys"]
        P = len(keys)
        if plan[s] is None:                                # only black: returned unchanged (clustering.py:197-199)
            results.append((keys.copy(), np.arange(P, dtype=np.int32), {"branch": "none"}))
            continue
        black, leaves, base = plan[s]
        nb = nb_idx[s]
        mapping = np.zeros(P, np.int32)
        mapping[black] = np.arange(len(black))
        new_keys = np.concatenate([np.zeros(len(black), np.uint32), means[base:base + len(leaves)]])
        has_dups = len(np.unique(keys)) != P
        if has_dups:
            order = np.argsort(keys, kind="stable")
            skeys = keys[order]
        for li, (rel, from_split) in enumerate(leaves):
            tgt = nb[rel]
            if from_split and has_dups:                    # find_color_index: first equal row only
                tgt = np.unique(order[np.searchsorted(skeys, keys[tgt], side="left")])
            mapping[tgt] = (len(black) + li) & 0xFFFF       # uint16 mapping_array (clustering.py:373)
        info = {"branch": "minibatch" if len(nb) >= MINIBATCH_THRESHOLD else "dbscan",
                "n_clusters": len(smalls[s]) + len(larges[s]), "n_large": len(larges[s])}
        results.append((new_keys, mapping, info))
    return results


def cluster_palette(rh, quality, keys, eps, mc):
    return cluster_palettes(rh, [{"keys": keys, "quality": quality, "eps": eps, "mc": mc}])[0]


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
    out["indices"] = out["indices_dev"].cpu().numpy()
    return out
