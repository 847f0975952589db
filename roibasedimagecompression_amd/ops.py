"""Thin Python wrappers over the C ABI: torch tensors own device memory, kernels run on torch's
current HIP stream.  Names follow the reference's domain: jobs (segments / crops), palettes, keys
(uint32 R<<16|G<<8|B held in int32 tensors), labels, components.

Reference functions served (paths relative to the reference root):
  unique_colors            encoder/compression/clustering.py:4-103   (get_all_unique_colors)
  clustering_params        encoder/compression/clustering.py:108-135
  eps_components           encoder/compression/clustering.py:233-235 (DBSCAN.fit_predict)
  kmeans_split             encoder/compression/clustering.py:720-775 (KMeans.fit_predict)
  minibatch_kmeans         encoder/compression/clustering.py:207-230 (MiniBatchKMeans.fit_predict)
  cluster_means            encoder/compression/clustering.py:304-310,346-355
  remap / decode           encoder/compression/clustering.py:373-377, decoder/.../uncompression.py:209
  merge_firstpos / paint   encoder/compression/merging.py:52-82
"""
import ctypes as C
import math
import os
import threading

import numpy as np
import torch

from . import _lib
from ._lib import MbkProblem, RhccqError
from .mt import MtWords

INT_MAX = 2 ** 31 - 1
BITMAP_WORDS = 524288
MINIBATCH_THRESHOLD = 10000          # clustering.py:205
SEED = 42                            # random_state=42 at every sklearn call site of the reference


_DRAW_POOL = None


def host_thread_budget():
    """host threads ONE rank may keep busy: the cores this process may use, shared out over the ranks of the node (LOCAL_WORLD_SIZE under
    torch.distributed.run: 8 ranks on one host must not start 8 x (8 draw + 16 lane + 16 class) workers on each other's cores)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(2, cores // ranks)


def _draw_pool():
    global _DRAW_POOL
    if _DRAW_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _DRAW_POOL = ThreadPoolExecutor(max_workers=min(8, host_thread_budget()), thread_name_prefix="rhccq-draw")
    return _DRAW_POOL


_LANE_POOLS = {}


def lane_pool(kind):
    """Persistent worker threads for the per-problem / per-class pipelines of a frame (one pool per nesting level, so that a
    class pipeline waiting for its problems never occupies the workers those need): handing a task to a parked worker costs
    tens of microseconds, starting a thread per problem and frame cost 0.3-0.8 ms each in front of the k-means++ chains."""
    pool = _LANE_POOLS.get(kind)
    if pool is None:
        with _words_lock:
            pool = _LANE_POOLS.get(kind)
            if pool is None:
                from concurrent.futures import ThreadPoolExecutor
                # (the workers mostly wait for the GPU: more of them than cores is fine, but not an unbounded multiple per rank)
                pool = _LANE_POOLS[kind] = ThreadPoolExecutor(max_workers=min(16, 2 * host_thread_budget()), thread_name_prefix=f"rhccq-{kind}")
    return pool


def pack_rgb(rgb):
    rgb = np.asarray(rgb, dtype=np.uint8).reshape(-1, 3).astype(np.uint32)
    return (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]


def unpack_rgb(keys):
    keys = np.asarray(keys).astype(np.uint32)
    return np.stack([(keys >> 16) & 255, (keys >> 8) & 255, keys & 255], axis=1).astype(np.uint8)


def morton3(keys):
    """24-bit Z-order code of packed colours: bit i of R, G, B -> bits 3i+2, 3i+1, 3i."""
    def spread(v):
        v = v.astype(np.uint32) & np.uint32(0xFF)
        v = (v | (v << np.uint32(16))) & np.uint32(0xFF0000FF)
        v = (v | (v << np.uint32(8))) & np.uint32(0x0F00F00F)
        v = (v | (v << np.uint32(4))) & np.uint32(0xC30C30C3)
        v = (v | (v << np.uint32(2))) & np.uint32(0x49249249)
        return v
    keys = np.asarray(keys).astype(np.uint32)
    return (spread(keys >> np.uint32(16)) << np.uint32(2)) | (spread(keys >> np.uint32(8)) << np.uint32(1)) | spread(keys)


def clustering_params(n_colors, quality):
    """compute_clustering_params through the C ABI (host-only entry point)."""
    lib = _lib.load()
    if quality == 0:
        raise ZeroDivisionError("float division by zero")     # clustering.py:129 behaviour
    eps, mc = C.c_double(), C.c_int64()
    rc = lib.rhccq_params(int(n_colors), float(quality), C.byref(eps), C.byref(mc))
    if rc:
        raise RhccqError(f"rhccq_params failed ({rc})")
    return eps.value, 1, mc.value


def eps_threshold(eps):
    lib = _lib.load()
    thr, bnd, r2 = C.c_int32(), C.c_int32(), C.c_double()
    rc = lib.rhccq_eps_threshold(float(eps), C.byref(thr), C.byref(bnd), C.byref(r2))
    if rc:
        raise RhccqError(f"rhccq_eps_threshold({eps}) failed ({rc})")
    return thr.value, bnd.value, r2.value


class _MtStream:
    """numpy legacy MT19937 RandomState(42): u0 = the draw consumed by choice() for the first centre,
    then the uniform(size=T) draws of k-means++ are consecutive doubles.  Every KMeans fit of the
    reference restarts from seed 42, so one cached prefix serves all split problems."""

    def __init__(self):
        self.host = np.zeros(0)
        self.dev = None

    def ensure(self, n, device):
        if len(self.host) < n + 1:
            m = max(n + 1, 2 * len(self.host), 1 << 16)
            self.host = np.random.RandomState(SEED).random_sample(m)
            self.dev = None
        if self.dev is None or self.dev.device != device:
            self.dev = torch.from_numpy(self.host[1:].copy()).to(device)
        return self.dev

    def u0(self):
        if len(self.host) == 0:
            self.host = np.random.RandomState(SEED).random_sample(1 << 16)
            self.dev = None
        return self.host[0]


_first_cache = {}
_first_lock = threading.Lock()
_words_lock = threading.Lock()


def first_centre_index(n, u0):
    """RandomState.choice(n, p=ones/n): searchsorted(cumsum(p)/cumsum(p)[-1], u0, 'right').  Called from the lanes of
    a StreamEncoder and from the draw pool: the value is computed locally, the cache only ever hands out finished ones."""
    key = (n, u0)
    v = _first_cache.get(key)
    if v is None:
        p = np.full(n, 1.0) / np.float64(n)
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        v = min(int(np.searchsorted(cdf, u0, side="right")), n - 1)
        with _first_lock:
            if len(_first_cache) > 4096:                     # a stream of frames brings ever new (n, u0): keep it bounded
                _first_cache.clear()
            _first_cache[key] = v
    return v


class _StreamBoundLib:
    """The C ABI as seen by one Rhccq: before every entry point that takes the context, the context is re-bound to
    torch's CURRENT stream when that changed (rhccq_ctx_set_stream) -- every buffer of this module is allocated, zeroed
    and freed by torch on the current stream, so the kernels must run there too (a `with torch.cuda.stream(s):` around
    the mirrored API would otherwise race the memsets against kernels on the stream captured at construction)."""

    def __init__(self, lib, owner):
        self._lib, self._owner = lib, owner

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        args = _lib.PROTOTYPES.get(name, (None, []))[1]
        if not args or args[0] is not _lib.c_void_p or name in ("rhccq_ctx_destroy", "rhccq_last_error", "rhccq_ctx_set_stream", "rhccq_stream"):
            setattr(self, name, fn)
            return fn
        owner = self._owner

        def bound(*a):
            owner._bind_stream()
            return fn(*a)
        bound.__name__ = name
        setattr(self, name, bound)
        return bound


class Rhccq:
    """One context per (process, device).  All methods take / return torch CUDA tensors unless a
    name says numpy."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RhccqError("no HIP device visible: the RHCCQ product path needs an MI355X (there is no CPU fallback)")
        self._raw = _lib.load()
        self.lib = _StreamBoundLib(self._raw, self)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._bound = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        rc = self._raw.rhccq_ctx_create(device, C.c_void_p(self._bound), C.byref(h))
        if rc:
            raise RhccqError(f"rhccq_ctx_create failed ({rc})")
        self.ctx = h
        self.mt = _MtStream()
        self.mtw = MtWords()
        self._mtw_dev = None

    OPT_INIT_LDS_BLOCKS, OPT_INIT_MAX_ITEMS, OPT_INIT_KERNEL, OPT_INIT_SHARDS, OPT_INIT_CANDS_PER_WAVE, OPT_REASSIGN_LDS = 1, 2, 3, 4, 5, 6
    OPT_REASSIGN_ORDER = 7           # 1 (default): numpy's scalar-quicksort tie order of the capped reassignment; 0: stable (rounds 1-3)

    def _bind_stream(self):
        """kernels follow torch's current stream (see _StreamBoundLib)"""
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound:
            rc = self._raw.rhccq_ctx_set_stream(self.ctx, C.c_void_p(s))
            if rc:
                raise RhccqError(f"rhccq_ctx_set_stream failed ({rc})")
            self._bound = s

    def set_option(self, option, value):
        """rhccq_ctx_set_int: thresholds between equivalent kernel paths (include/rhccq.h)"""
        self._check(self.lib.rhccq_ctx_set_int(self.ctx, int(option), int(value)), "ctx_set_int")
        # the sibling contexts of pipelined calls (one per lane / per class) follow: an option set on the context a caller holds
        # must reach the kernels wherever they are launched from
        self.__dict__.setdefault("_options", {})[int(option)] = int(value)
        for _, ln in self.__dict__.get("_lanes", {}).values():
            ln.set_option(option, value)

    def _mt_words_dev(self, n):
        """at least the first n raw MT19937 words on the device (uploaded once, regrown geometrically; the lanes of a
        pipelined call share their parent's table)"""
        owner = getattr(self, "_parent", None) or self
        t = owner._mtw_dev
        if t is None or t.numel() < n:
            with _words_lock:
                t = owner._mtw_dev
                if t is None or t.numel() < n:
                    have = 0 if t is None else t.numel()
                    w = owner.mtw.ensure(max(n, 2 * have, 1 << 22))
                    t = torch.from_numpy(w.view(np.int32)).to(self.device)      # blocking copy: complete on return
                    torch.cuda.current_stream(self.device).synchronize()
                    # a superseded table may still be read by kernels another lane has queued on ITS stream: keep it alive until
                    # every stream of the family has passed an event recorded now, then hand its block back to the allocator
                    retired = owner.__dict__.setdefault("_mtw_retired", [])
                    retired[:] = [(tab, evs) for tab, evs in retired if not all(e.query() for e in evs)]
                    if owner._mtw_dev is not None:
                        streams = [torch.cuda.current_stream(self.device), torch.cuda.default_stream(self.device)]
                        streams += [ls for ls, _ in owner.__dict__.get("_lanes", {}).values()]
                        evs = []
                        for ls in streams:
                            e = torch.cuda.Event()
                            e.record(ls)
                            evs.append(e)
                        retired.append((owner._mtw_dev, evs))
                    owner._mtw_dev = t
        t.record_stream(torch.cuda.current_stream(self.device))                 # (allocated on whichever lane grew it, used on this one)
        return t

    def close(self):
        if getattr(self, "ctx", None):
            self._raw.rhccq_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers --------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc:
            msg = self._raw.rhccq_last_error(self.ctx)
            raise RhccqError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def dev(self, arr, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device, non_blocking=False)

    def to_host(self, *tensors):
        """device tensors -> numpy arrays through page-locked landing buffers (torch's caching host allocator), all copies in flight
        together and ONE synchronisation: a 25 MB frame comes back in ~1 ms instead of ~4 through pageable memory.  The arrays own
        their buffers (freed with them)."""
        outs = []
        for t in tensors:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            outs.append(h)
        torch.cuda.current_stream(self.device).synchronize()
        res = [h.numpy() for h in outs]
        return res[0] if len(res) == 1 else res

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _class_args(self, labels, job_base):
        n = len(job_base)
        ptrs = (C.c_void_p * n)(*[(l.data_ptr() if l is not None else 0) for l in labels])
        bases = (C.c_int32 * n)(*[int(b) for b in job_base])
        return n, ptrs, bases

    # -- K0 / K1 ----------------------------------------------------------------------------------
    def new_job_state(self, n_jobs):
        bitmaps = self.zeros((n_jobs, BITMAP_WORDS), torch.int32)
        stats = torch.tensor([INT_MAX, -1, INT_MAX, -1, 0, 0], dtype=torch.int32, device=self.device).repeat(n_jobs, 1).contiguous()
        return bitmaps, stats

    BYTEMAP_MAX_JOBS = 64        # 16 MiB of byte flags per job: up to 1 GiB of the 288 GB
    scan_events = None           # a list here makes job_scan() bracket its kernel with HIP events (bench.py)

    def job_scan(self, rgb, labels, job_base, bitmaps, stats, black_is_colour, bytemaps=None):
        """K0 + K1a.  With few jobs the colour flags go through byte maps (plain stores) and are packed into
        the bitmaps; with many jobs bits are set directly with atomics."""
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        n_jobs = bitmaps.shape[0]
        if bytemaps is None and n_jobs <= self.BYTEMAP_MAX_JOBS:
            bytemaps = self.zeros((n_jobs, 1 << 24), torch.uint8)
        if bytemaps is not None:
            ev = None
            if self.scan_events is not None:                  # measurement hook (bench.py): HIP events on the launch stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self._check(self.lib.rhccq_job_scan_bytes(self.ctx, self._p(rgb), H, W, n, ptrs, bases, int(black_is_colour),
                                                      self._p(bytemaps), self._p(stats)), "job_scan_bytes")
            if ev is not None:
                ev[1].record()
                self.scan_events.append(ev)
            self._check(self.lib.rhccq_bytemap_pack(self.ctx, self._p(bytemaps), n_jobs, self._p(bitmaps)), "bytemap_pack")
        else:
            self._check(self.lib.rhccq_job_scan(self.ctx, self._p(rgb), H, W, n, ptrs, bases, int(black_is_colour),
                                                self._p(bitmaps), self._p(stats)), "job_scan")

    def job_set_black(self, bitmaps, jobs):
        if len(jobs) == 0:
            return
        j = self.dev(np.asarray(jobs, dtype=np.int32))
        self._check(self.lib.rhccq_job_set_black(self.ctx, self._p(bitmaps), self._p(j), len(jobs)), "job_set_black")

    def bitmap_count(self, bitmaps):
        n_jobs = bitmaps.shape[0]
        chunk = self.empty((n_jobs, 512), torch.int32)
        counts = self.empty((n_jobs,), torch.int32)
        self._check(self.lib.rhccq_bitmap_count(self.ctx, self._p(bitmaps), n_jobs, self._p(chunk), self._p(counts)), "bitmap_count")
        return chunk, counts

    def bitmap_emit(self, bitmaps, chunk, pal_off, total, want_keys=True):
        n_jobs = bitmaps.shape[0]
        prefix = self.empty((n_jobs, BITMAP_WORDS, 2), torch.int32)            # (bitmap word, exclusive prefix) pairs
        keys = self.empty((max(int(total), 1),), torch.int32) if want_keys else None
        self._check(self.lib.rhccq_bitmap_emit(self.ctx, self._p(bitmaps), n_jobs, self._p(chunk), self._p(pal_off),
                                               self._p(prefix), self._p(keys)), "bitmap_emit")
        return prefix, keys

    def job_blackfix(self, rgb, labels, job_base, needs_fix, best):
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        self._check(self.lib.rhccq_job_blackfix(self.ctx, self._p(rgb), H, W, n, ptrs, bases, self._p(needs_fix), self._p(best)),
                    "job_blackfix")

    # -- many-segment frames: unique colours by one device sort (rhccq_job_sort_unique) ------------------------------
    def job_stats(self, rgb, labels, job_base, n_jobs):
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        stats = torch.tensor([INT_MAX, -1, INT_MAX, -1, 0, 0], dtype=torch.int32, device=self.device).repeat(n_jobs, 1).contiguous()
        self._check(self.lib.rhccq_job_stats(self.ctx, self._p(rgb), H, W, n, ptrs, bases, self._p(stats)), "job_stats")
        return stats

    def job_sort_unique(self, rgb, labels, job_base, n_jobs, fix_key, black_jobs):
        """-> (P np.int64[n_jobs] palette sizes, keys int32[sum P] device (sorted per job, job order), rankmap int32[n_class, H*W] device)"""
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        n, ptrs, bases = self._class_args(labels, job_base)
        bj = self.dev(np.asarray(black_jobs, dtype=np.int32)) if len(black_jobs) else None
        total = n * H * W + len(black_jobs)
        tb = int(self.lib.rhccq_job_sort_unique_bytes(total))
        tmp = self.empty((tb,), torch.uint8)
        rankmap = self.empty((n, H * W), torch.int32)
        keys = self.empty((total,), torch.int32)
        start = self.empty((n_jobs,), torch.int32)
        nu = self.empty((1,), torch.int32)
        self._check(self.lib.rhccq_job_sort_unique(self.ctx, self._p(rgb), H, W, n, ptrs, bases, int(n_jobs), self._p(fix_key), self._p(bj), len(black_jobs),
                                                   self._p(tmp), tb, self._p(rankmap), self._p(keys), self._p(start), self._p(nu)), "job_sort_unique")
        st = start.cpu().numpy().astype(np.int64)
        n_unique = int(nu.cpu()[0])
        # palette sizes: the distance to the next job that has colours
        nxt = np.full(n_jobs + 1, n_unique, np.int64)
        for j in range(n_jobs - 1, -1, -1):
            nxt[j] = st[j] if st[j] >= 0 else nxt[j + 1]
        P = np.where(st >= 0, nxt[1:] - st, 0).astype(np.int64)
        return P, keys[:max(n_unique, 1)], rankmap

    def job_index_ranked(self, H, W, labels, job_base, rankmap, pal_off, first_pos, fp_lut):
        n, ptrs, bases = self._class_args(labels, job_base)
        self._check(self.lib.rhccq_job_index_ranked(self.ctx, int(H), int(W), n, ptrs, bases, self._p(rankmap), self._p(pal_off), self._p(first_pos),
                                                    self._p(fp_lut)), "job_index_ranked")

    def frame_remap_ranked(self, H, W, labels, job_base, rankmap, pal_off, lut, default_index, out_dtype, lut2=None):
        n, ptrs, bases = self._class_args(labels, job_base)
        out = self.empty((int(H), int(W)), out_dtype)
        self._check(self.lib.rhccq_frame_remap_ranked(self.ctx, int(H), int(W), n, ptrs, bases, self._p(rankmap), self._p(pal_off), self._p(lut), self._p(lut2),
                                                      int(default_index), self._p(out), out.element_size()), "frame_remap_ranked")
        return out

    def job_index(self, rgb, labels, job_base, bitmaps, prefix, pal_off, fix_key=None, want_idx=True, first_pos=None, fp_lut=None):
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        idx = self.empty((n, H * W), torch.int32) if want_idx else None
        self._check(self.lib.rhccq_job_index(self.ctx, self._p(rgb), H, W, n, ptrs, bases, self._p(bitmaps), self._p(prefix),
                                             self._p(pal_off), self._p(fix_key), self._p(idx), self._p(first_pos), self._p(fp_lut)),
                    "job_index")
        return idx

    def job_index_entries(self, rgb, labels, job_base, bitmaps, prefix, pal_off, fix_key, first_pos, fp_lut, entries_out):
        """first positions as job_index + the table entry every pixel shows, into entries_out (int32[len(labels)][H*W] view)"""
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        assert entries_out.dtype == torch.int32 and entries_out.is_contiguous() and entries_out.numel() == n * H * W
        self._check(self.lib.rhccq_job_index_entries(self.ctx, self._p(rgb), H, W, n, ptrs, bases, self._p(bitmaps), self._p(prefix), self._p(pal_off),
                                                     self._p(fix_key), self._p(first_pos), self._p(fp_lut), self._p(entries_out)), "job_index_entries")

    def frame_remap_entries(self, H, W, labels, job_base, entries, default_index, out_dtype, lut2=None):
        n, ptrs, bases = self._class_args(labels, job_base)
        out = self.empty((int(H), int(W)), out_dtype)
        self._check(self.lib.rhccq_frame_remap_entries(self.ctx, int(H), int(W), n, ptrs, bases, self._p(entries), self._p(lut2), int(default_index),
                                                       self._p(out), out.element_size()), "frame_remap_entries")
        return out

    def frame_remap(self, rgb, labels, job_base, bitmaps, prefix, pal_off, fix_key, lut, default_index, out_dtype, lut2=None):
        H, W = rgb.shape[0], rgb.shape[1]
        n, ptrs, bases = self._class_args(labels, job_base)
        out = self.empty((H, W), out_dtype)
        self._check(self.lib.rhccq_frame_remap(self.ctx, self._p(rgb), H, W, n, ptrs, bases, self._p(bitmaps), self._p(prefix),
                                               self._p(pal_off), self._p(fix_key), self._p(lut), self._p(lut2), int(default_index),
                                               self._p(out), out.element_size()), "frame_remap")
        return out

    def unique_colors(self, rgb):
        """get_all_unique_colors core: rgb uint8[H,W,3] (device) -> (keys int32[P] sorted, idx int32[H*W])."""
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.shape[-1] == 3
        bitmaps, stats = self.new_job_state(1)
        self.job_scan(rgb, [None], [0], bitmaps, stats, black_is_colour=True)
        chunk, counts = self.bitmap_count(bitmaps)
        total = int(counts.cpu()[0])
        pal_off = self.zeros((1,), torch.int64)
        prefix, keys = self.bitmap_emit(bitmaps, chunk, pal_off, total)
        idx = self.job_index(rgb, [None], [0], bitmaps, prefix, pal_off)
        return keys[:total], idx[0]

    # -- K3 / K4 ----------------------------------------------------------------------------------
    def eps_components(self, key_list, eps_list):
        """Batched DBSCAN(min_samples=1) labels.  key_list: numpy uint32/int arrays; returns
        (list of int32 numpy label arrays, list of component counts)."""
        n_prob = len(key_list)
        if n_prob == 0:
            return [], []
        sizes = [len(k) for k in key_list]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        if offs[-1] == 0:
            return [np.zeros(0, np.int32) for _ in key_list], [0] * n_prob
        desc = np.zeros((n_prob, 4), np.int32)
        r2 = np.zeros(n_prob, np.float64)
        for i, e in enumerate(eps_list):
            thr, bnd, rr = eps_threshold(e)
            desc[i] = (offs[i], sizes[i], thr, bnd)
            r2[i] = rr
        keys = self.dev(np.concatenate(key_list).astype(np.uint32).view(np.int32))
        d_desc, d_r2 = self.dev(desc), self.dev(r2)
        labels = self.empty((int(offs[-1]),), torch.int32)
        ncomp = self.empty((n_prob,), torch.int32)
        self._check(self.lib.rhccq_eps_components(self.ctx, self._p(keys), self._p(d_desc), self._p(d_r2), n_prob, int(max(sizes)),
                                                  self._p(labels), self._p(ncomp)), "eps_components")
        lab = labels.cpu().numpy()
        nc = ncomp.cpu().numpy()
        return [lab[offs[i]:offs[i + 1]] for i in range(n_prob)], [int(v) for v in nc]

    def dbscan_labels(self, keys, eps, min_samples):
        """sklearn DBSCAN(eps / 255, min_samples).fit_predict on one palette (uint32 keys, < 10 000 colours): core points = at least
        min_samples colours within eps (itself included); clusters = eps-components of the core points, numbered by their lowest core
        index; a non-core point takes the smallest label among its core neighbours, -1 (noise) without one.  -> int32 numpy labels"""
        keys = np.ascontiguousarray(keys).astype(np.uint32)
        n = len(keys)
        if n == 0:
            return np.zeros(0, np.int32)
        if min_samples <= 1:
            return self.eps_components([keys], [eps])[0][0]
        thr, bnd, rr = eps_threshold(eps)
        d_keys = self.dev(keys.view(np.int32))
        counts = self.empty((n,), torch.int32)
        self._check(self.lib.rhccq_eps_counts(self.ctx, self._p(d_keys), n, thr, bnd, rr, self._p(counts)), "eps_counts")
        core = counts.cpu().numpy() >= int(min_samples)
        core_label = np.full(n, -1, np.int32)
        if core.any():
            core_label[core] = self.eps_components([keys[core]], [eps])[0][0]
        d_core = self.dev(core_label)
        out = self.empty((n,), torch.int32)
        self._check(self.lib.rhccq_eps_border(self.ctx, self._p(d_keys), n, thr, bnd, rr, self._p(d_core), self._p(out)), "eps_border")
        return out.cpu().numpy()

    # -- K7 -----------------------------------------------------------------------------------------
    def kmeans_split(self, key_list, k_list, return_info=False):
        """Batched KMeans(k, random_state=42).fit_predict labels (KM64).  numpy in / numpy out."""
        n_prob = len(key_list)
        if n_prob == 0:
            return ([], []) if return_info else []
        sizes = [len(k) for k in key_list]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        koff = np.concatenate([[0], np.cumsum(k_list)]).astype(np.int64)
        desc = np.zeros((n_prob, 6), np.int32)
        need = 1
        u0 = self.mt.u0()
        for i, (n, k) in enumerate(zip(sizes, k_list)):
            assert 1 <= k <= n
            T = 2 + int(math.log(k))
            desc[i] = (offs[i], n, k, 0, first_centre_index(n, u0), T)
            need = max(need, (k - 1) * T)
        rand = self.mt.ensure(need, self.device)
        keys = self.dev(np.concatenate(key_list).astype(np.uint32).view(np.int32))
        d_desc, d_koff = self.dev(desc), self.dev(koff[:-1].copy())
        work = self.empty((8 * int(koff[-1]) + 8,), torch.float64)
        labels = self.empty((int(offs[-1]),), torch.int32)
        info = self.zeros((n_prob, 4), torch.int32)
        self._check(self.lib.rhccq_kmeans(self.ctx, self._p(keys), self._p(d_desc), self._p(d_koff), self._p(rand), n_prob,
                                          int(max(sizes)), self._p(work), self._p(labels), self._p(info)), "kmeans")
        lab = labels.cpu().numpy()
        out = [lab[offs[i]:offs[i + 1]] for i in range(n_prob)]
        if return_info:
            return out, info.cpu().numpy()
        return out

    # -- K2 -----------------------------------------------------------------------------------------
    def cluster_means(self, keys, labels, k):
        """floor-mean colour per label: keys int32[n] device, labels int32[n] device -> int32[k] keys."""
        sums = self.zeros((max(k, 1), 4), torch.int64)
        self._check(self.lib.rhccq_cluster_sums(self.ctx, self._p(keys), self._p(labels), keys.numel(), max(k, 1), self._p(sums)), "cluster_sums")
        out = self.empty((max(k, 1),), torch.int32)
        self._check(self.lib.rhccq_cluster_means(self.ctx, self._p(sums), k, self._p(out)), "cluster_means")
        return out[:k], sums[:k]

    # -- K8 -----------------------------------------------------------------------------------------
    MBK_LANES = 4        # problems of a small batch run as independent pipelines on this many HIP streams
    MBK_OVERLAP = True   # a lone running problem: E-step of step t + 1 beside the update of step t (bit-identical; k8_overlap.h)
    MBK_OVERLAP_MIN_K = 1024
    MBK_OVERLAP_POLL = 64    # steps per overlapped call (the state is looked at one call behind: no drain between the calls)
    MBK_FIRST_POLL = 16

    def _lane(self, i):
        """a sibling context on its own HIP stream (host thread `i` of a pipelined call); shares the MT19937 word table"""
        lanes = self.__dict__.setdefault("_lanes", {})
        if i not in lanes:
            stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(stream):
                rh = Rhccq(self.device.index)
            rh.mtw, rh._parent = self.mtw, getattr(self, "_parent", None) or self      # (the root context owns the MT19937 word table)
            for opt, val in self.__dict__.get("_options", {}).items():
                rh.set_option(opt, val)
            lanes[i] = (stream, rh)
        return lanes[i]

    def _minibatch_lanes(self, key_list, k_list, return_info, poll_steps, return_device, estep, estep_split, n_lanes):
        """Independent problems as independent pipelines: init -> steps -> assign of each lane's problems on its own HIP
        stream, driven by its own host thread (the C calls and the state polls release the GIL).  The k-means++ chain of
        one problem and the mini-batch steps of another then overlap instead of queueing behind each other on one
        stream: a single 4K frame's level 1 costs max(init_p + steps_p) rather than max(init) + max(steps)."""
        n_prob = len(key_list)
        # longest chains first, each to the lane with the least work so far
        order = sorted(range(n_prob), key=lambda i: -k_list[i])
        groups, load = [[] for _ in range(n_lanes)], [0] * n_lanes
        for i in order:
            g = load.index(min(load))
            groups[g].append(i)
            load[g] += k_list[i]
        parts = [k if torch.is_tensor(k) else self.dev(np.ascontiguousarray(np.asarray(k)).astype(np.uint32).view(np.int32)) for k in key_list]
        here = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(here)
        results, errors = [None] * n_lanes, []

        def run(g):
            try:
                torch.cuda.set_device(self.device)
                stream, rh = self._lane(g)
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)
                    idx = groups[g]
                    out = rh.minibatch_kmeans([parts[i] for i in idx], [k_list[i] for i in idx], return_info=True, poll_steps=poll_steps,
                                              return_device=True, estep=estep, estep_split=estep_split, lanes=1)
                    done = torch.cuda.Event()
                    done.record(stream)
                    results[g] = (out, done)
            except BaseException as e:                        # surfaced to the caller below
                errors.append(e)

        futs = [lane_pool("mbk").submit(run, g) for g in range(n_lanes) if groups[g]]
        for f in futs:
            f.result()
        if errors:
            raise errors[0]
        labs = [None] * n_prob
        infos = [None] * n_prob
        n_overlapped = 0
        for g in range(n_lanes):
            if not groups[g]:
                continue
            (lg, ig), done = results[g]
            here.wait_event(done)
            n_overlapped += ig["overlapped_launches"]
            for j, i in enumerate(groups[g]):
                lg[j].record_stream(here)
                labs[i] = lg[j]
                a, b = int(ig["koff"][j]), int(ig["koff"][j + 1])
                infos[i] = (ig["state"][j], ig["centres"][a:b], ig["chosen"][a:b], ig["weights"][a:b])
        if not return_device:
            labs = [l.cpu().numpy() for l in labs]
        if return_info:
            koff = np.concatenate([[0], np.cumsum(k_list)]).astype(np.int64)
            return labs, {"state": np.stack([x[0] for x in infos]), "centres": np.concatenate([x[1] for x in infos]),
                          "chosen": np.concatenate([x[2] for x in infos]), "koff": koff, "weights": np.concatenate([x[3] for x in infos]),
                          "overlapped_launches": n_overlapped}
        return labs

    def _overlapped_steps(self, keys, probs, k, n, limit, step, st, cur_max, centres, weights, state, work, wbytes, estep_split, chunk,
                          words_per_step):
        """The remaining steps of ONE problem through rhccq_mbk_steps_overlapped, chunk after chunk WITHOUT draining the stream in
        between: chunk i + 1 is queued before chunk i's state is looked at (an asynchronous copy into pinned memory behind each
        chunk), so the kernels of a ~2000-step problem run back to back.  Everything the host needs for the next chunk it can
        derive itself: which steps reassign follows from "samples since the last reassignment" (+ batch per step, reset at
        10 k), the carry flags come back from the call, and the MT19937 table is sized for all remaining steps up front.  A
        problem that stops inside chunk i leaves chunk i + 1 as launches that return at once.  Returns (steps launched, state,
        launches that went through the overlapped entry)."""
        bs = min(1000, n)
        par = step & 1
        since = int(st[0, 12 if par else 3])
        split = estep_split or next((sp for sp in (1, 2, 4, 8) if ((k + 511) // 512) * 2 * sp >= 1536), 8)
        # MT19937 words for a BOUNDED horizon -- the chunks in flight plus the one being queued, counted from the last cursor the host
        # has seen -- regrown between chunks (a superseded table stays alive for the kernels already queued on it, _mt_words_dev):
        # a draw of 1000 rows consumes < 2000 words on average even at the worst acceptance (1/2), a reassigning step (at most one in
        # 10 k / 1000 + the first steps) another ~1400 for its shuffle; twice that plus the kernels' own end-of-table margins.  (Sizing
        # for all 100 n / 1000 possible steps asked for ~2.5 GB per 1.5 M-colour problem that then stopped after tens of steps.)
        cur_known, steps_known = int(cur_max), int(step)
        carry = C.c_int32(0)
        stream = torch.cuda.current_stream(self.device)
        pending = []                                         # (pinned copy of the state, event) per chunk in flight
        n_ov = 0
        # three pinned landing buffers, kept with the context (allocating pinned memory costs ~0.1 ms a time; two chunks are in flight)
        ring = self.__dict__.setdefault("_pinned_state", [torch.empty((1, 16), dtype=torch.float64, pin_memory=True) for _ in range(3)])
        n_chunk = 0
        while True:
            if step < limit:
                ns = int(min(chunk, limit - step))
                words = self._mt_words_dev(cur_known + (step - steps_known + ns + 4) * 4200 + 8 * words_per_step)
                self._check(self.lib.rhccq_mbk_steps_overlapped(self.ctx, self._p(keys), probs, 1, step, ns, self._p(words), words.numel(),
                                                                self._p(centres), self._p(weights), self._p(state), self._p(work), wbytes,
                                                                split, since, C.byref(carry)), "mbk_steps_overlapped")
                for _ in range(ns):                          # the schedule's own arithmetic (sklearn _random_reassign)
                    since += bs
                    if since >= 10 * k:
                        since = 0
                step += ns
                n_ov += ns
                host = ring[n_chunk % 3]
                n_chunk += 1
                host.copy_(state, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
                pending.append((host, ev))
            if len(pending) >= 2 or step >= limit:
                host, ev = pending.pop(0)
                ev.synchronize()
                st = host.numpy().copy()
                cur_known, steps_known = int(max(st[0, 9], st[0, 14])), int(st[0, 5])
                if st[0, 4] >= 3 or st[0, 11] != 0 or st[0, 5] >= limit:
                    break
                if not pending and step >= limit:
                    break
        if pending:                                          # launches queued behind the stop: they return at once
            pending[-1][1].synchronize()
            st = pending[-1][0].numpy().copy()
        return step, st, n_ov

    def npysort_head(self, w, cap, depth0=-1, use_lds=True):
        """the SET np.argsort(w)[:cap] under numpy's scalar sort kernel (rhccq_npysort_head; w: non-negative integer counts as a
        float64 device tensor or numpy array): bool[k] numpy mask.  The selection inside a capped mini-batch reassignment."""
        w = w if torch.is_tensor(w) else self.dev(np.ascontiguousarray(w, dtype=np.float64))
        k = int(w.numel())
        scratch = self.empty((16 * k,), torch.uint8)
        mask = self.empty(((k + 31) // 32,), torch.int32)
        self._check(self.lib.rhccq_npysort_head(self.ctx, self._p(w), k, int(cap), int(depth0), int(bool(use_lds)), self._p(scratch), self._p(mask)),
                    "npysort_head")
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:k]
        return bits.astype(bool)

    def minibatch_kmeans(self, key_list, k_list, return_info=False, poll_steps=64, return_device=False, timing=None, estep="auto", estep_split=0,
                         lanes=None):
        """Batched MiniBatchKMeans(k, batch_size=1000, random_state=42).fit_predict labels: sklearn's fit operation
        for operation (RandomState(42) replayed from its raw MT19937 words, k-means++ in draw order, batch-ordered
        centre updates, the reassignment's np.argsort in the tie order of numpy's scalar quicksort -- csrc/k8_npysort.h;
        OPT_REASSIGN_ORDER = 0 gives the stable order of rounds 1-3).  key_list items are numpy arrays or
        device int32 tensors (kept resident); labels come back as numpy arrays, or as device tensors
        with return_device=True.  `timing` (a dict) receives the HIP-event duration of the k-means++ launch
        (events on the launch stream).  `estep`: "auto" | "tiles" | "grid" -- how the batch E-step searches the
        centres (include/rhccq.h, identical results); "auto" looks at the centres of the problems still running."""
        n_prob = len(key_list)
        if n_prob == 0:
            return ([], []) if return_info else []
        # few problems (a single frame): every problem its own pipeline; many (a batch of frames): one batched launch per
        # phase keeps all CUs busy anyway and costs the host far less
        if lanes is None:
            lanes = min(n_prob, self.MBK_LANES) if (timing is None and 1 < n_prob <= 2 * self.MBK_LANES) else 1
        if lanes > 1 and n_prob > 1:
            return self._minibatch_lanes(key_list, k_list, return_info, poll_steps, return_device, estep, estep_split, min(lanes, n_prob))
        probs = (MbkProblem * n_prob)()
        sizes = [int(k.numel()) if torch.is_tensor(k) else len(k) for k in key_list]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        koff = np.concatenate([[0], np.cumsum(k_list)]).astype(np.int64)
        parts = [k if torch.is_tensor(k) else self.dev(np.ascontiguousarray(np.asarray(k)).astype(np.uint32).view(np.int32)) for k in key_list]
        keys = parts[0] if n_prob == 1 else torch.cat(parts)
        init_list = []
        ioff = roff = 0
        # numpy's RandomState(42) stream of every problem, replayed from the cached raw MT19937 words (mt.py): the
        # sample indices on the host (vectorised rejection sampling), the k-means++ uniforms on the device
        mtw = self.mtw
        upos = []

        def randint(pos, n, size, want):
            # (the native loop of rhccq_mt_randint_host: the problems' draws run on several threads, numpy's passes queue on the GIL)
            win = int(size / (n / (1 << int(n - 1).bit_length())) * 1.05) + 256 if n > 1 else 0
            out = np.empty(size, np.int32) if want else None
            while True:
                w = mtw.ensure(pos + win)
                used = int(self.lib.rhccq_mt_randint_host(w.ctypes.data, len(w), pos, n, size, out.ctypes.data if want else None))
                if used >= 0:
                    return out, used
                if used != -1:
                    raise RhccqError("rhccq_mt_randint_host: bad argument")
                win = 2 * win + 1024

        def draw(args):
            n, k = args
            bs = min(1000, n)
            init_size = 3 * bs
            if init_size < k:
                init_size = 3 * k
            init_size = min(init_size, n)
            pos = 0
            _, used = randint(pos, n, init_size, False)          # validation_indices: stream position only
            pos += used
            if init_size < n:
                init_idx, used = randint(pos, n, init_size, True)
                pos += used
            else:
                init_idx = np.arange(n, dtype=np.int32)
            first = first_centre_index(init_size, mtw.double(pos))
            return init_idx, first, pos + 2

        todo = list(zip(sizes, k_list))
        # the numpy passes of a replay release the GIL: the problems' draws run side by side (the GPU waits for them)
        drawn = list(_draw_pool().map(draw, todo)) if n_prob > 1 else [draw(todo[0])]
        cursor0 = []
        for i, ((n, k), (init_idx, first, pos)) in enumerate(zip(todo, drawn)):
            init_size = len(init_idx)
            T = 2 + int(math.log(k))
            nu = max((k - 1) * T, 1)
            cursor0.append(pos + 2 * (k - 1) * T)             # stream position behind the k-means++ uniforms
            p = probs[i]
            p.off, p.n, p.k, p.koff = int(offs[i]), n, k, int(koff[i])
            p.init_off, p.init_n, p.rand_off, p.first, p.T = ioff, init_size, roff, first, T
            init_list.append(init_idx)
            upos.append((pos, nu, roff))
            ioff += init_size
            roff += nu
        words = self._mt_words_dev(max(pos + 2 * nu for pos, nu, _ in upos))
        d_rand = self.empty((roff,), torch.float64)
        for pos, nu, ro in upos:
            self._check(self.lib.rhccq_mt_uniforms(self.ctx, self._p(words), pos, nu, C.c_void_p(d_rand.data_ptr() + 8 * ro)), "mt_uniforms")
        d_init = self.dev(np.concatenate(init_list))           # sklearn's draw order
        # internal pruning index: the sample positions in Morton order of their colours, on the device: 64 consecutive
        # entries form a compact box, which is what the exact block pruning of mbk_init_kernel relies on
        obytes = int(self.lib.rhccq_mbk_order_bytes(ioff))
        otmp = self.empty((obytes,), torch.uint8)
        d_perm = self.empty((ioff,), torch.int32)
        self._check(self.lib.rhccq_mbk_order(self.ctx, self._p(keys), probs, n_prob, self._p(d_init), self._p(d_perm), self._p(otmp), obytes),
                    "mbk_order")
        K = int(koff[-1])
        centres = self.zeros((K, 4), torch.float64)
        chosen = self.zeros((K,), torch.int32)
        if timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._check(self.lib.rhccq_mbk_init(self.ctx, self._p(keys), probs, n_prob, self._p(d_init), self._p(d_perm), self._p(d_rand),
                                            self._p(centres), self._p(chosen)), "mbk_init")
        if timing is not None:
            ev[1].record()
            ev[1].synchronize()
            timing["init_ms"] = ev[0].elapsed_time(ev[1])
        weights = self.zeros((K,), torch.float64)
        st0 = np.zeros((n_prob, 16))
        st0[:, 8] = k_list                                   # every centre starts with zero weight
        st0[:, 9] = cursor0                                  # MT19937 words consumed so far
        state = self.dev(st0)
        cur_max = max(cursor0)
        WORDS_PER_STEP = 16384                               # kWordsMargin of mbk_update_kernel
        wbytes = int(self.lib.rhccq_mbk_work_bytes(probs, n_prob))
        work = self.empty((max(wbytes, 8),), torch.uint8)
        k_arr = np.asarray(k_list, np.int64)
        limit = np.array([(100 * n) // min(1000, n) for n in sizes], np.int64)
        running = np.ones(n_prob, bool)
        step = 0                                             # launch index = step index of every problem still running
        carry = C.c_int32(0)
        st = st0
        n_overlapped = 0

        def check(st):
            if (st[:, 4] == 3).any():
                raise RhccqError("mini-batch steps ran past the end of the MT19937 word table (internal sizing error)")
            if (st[:, 4] == 4).any():
                raise RhccqError("the sharded k-means++ chain gave up waiting for a partner workgroup (RHCCQ_OPT_INIT_SHARDS = 1 avoids the hand-offs)")
            if (st[:, 4] == 5).any():
                raise RhccqError("the overlapped mini-batch schedule and the device state disagree about a reassignment (internal error)")

        while running.any():
            par = step & 1
            # a lone problem whose centres all carry weight: the next E-step starts beside the update (rhccq_mbk_steps_overlapped)
            tiles_mode = {"tiles": 1, "grid": 2}.get(estep) or (2 if int(k_arr[running].sum()) >= 200000 else 1)
            if (self.MBK_OVERLAP and n_prob == 1 and step > 0 and tiles_mode == 1 and k_list[0] >= self.MBK_OVERLAP_MIN_K
                    and st[0, 13 if par else 8] == 0):
                step, st, n_ov = self._overlapped_steps(keys, probs, k_list[0], sizes[0], int(limit[0]), step, st, cur_max, centres, weights,
                                                        state, work, wbytes, estep_split, max(poll_steps, self.MBK_OVERLAP_POLL), WORDS_PER_STEP)
                n_overlapped += n_ov
                check(st)
                break                                        # (the problem has stopped or used up its steps)
            # most problems converge within a dozen steps: look early once, so that a finished problem does not sit through the
            # launches of a whole poll interval
            ns = int(min(poll_steps if step else min(poll_steps, self.MBK_FIRST_POLL), max(1, limit[running].max() - step)))
            # the grid E-step pays when many centres are in flight; once only stragglers are left the tiled
            # brute force has fewer and shorter launches per step
            mode = tiles_mode
            # few workgroups left (a straggler problem): several threads share a batch point in the tiled E-step
            wgs = int(((k_arr[running] + 511) // 512).sum()) * 2
            split = estep_split or next((sp for sp in (1, 2, 4, 8) if wgs * sp >= 1536), 8)
            words = self._mt_words_dev(cur_max + (ns + 3) * WORDS_PER_STEP)     # a step consumes at most WORDS_PER_STEP
            if step > 0:
                # no running problem reassigns during these steps (no centre without weight, fewer than 10 k samples since the last
                # reassignment throughout): the second launch of a reassigning step is left out (RHCCQ_STEPS_NO_REASSIGN)
                bs_arr = np.minimum(1000, np.asarray(sizes, np.int64))
                quiet = (st[:, 13 if par else 8] == 0) & (st[:, 12 if par else 3] + ns * bs_arr < 10 * k_arr)
                if quiet[running].all():
                    mode |= 0x100
            self._check(self.lib.rhccq_mbk_steps(self.ctx, self._p(keys), probs, n_prob, step, ns, self._p(words), words.numel(),
                                                 self._p(centres), self._p(weights), self._p(state), self._p(work), wbytes, mode, split),
                        "mbk_steps")
            step += ns
            st = state.cpu().numpy()
            cur_max = int(max(st[:, 9].max(), st[:, 14].max()))
            check(st)
            running = (st[:, 11] == 0) & (st[:, 5] < limit)
        labels = self.empty((int(offs[-1]),), torch.int32)
        self._check(self.lib.rhccq_mbk_assign(self.ctx, self._p(keys), probs, n_prob, self._p(centres), self._p(work), wbytes,
                                              self._p(labels)), "mbk_assign")
        if return_device:
            out = [labels[offs[i]:offs[i + 1]] for i in range(n_prob)]
        else:
            lab = labels.cpu().numpy()
            out = [lab[offs[i]:offs[i + 1]] for i in range(n_prob)]
        if return_info:
            return out, {"state": state.cpu().numpy(), "centres": centres.cpu().numpy(), "chosen": chosen.cpu().numpy(),
                         "koff": koff, "weights": weights.cpu().numpy(), "overlapped_launches": n_overlapped}
        return out

    # -- EXTENSION: pixel-space DBSCAN on (x, y, L, a, b) (no reference counterpart) ------------------------
    @staticmethod
    def px_tables():
        """float32[256 + 2048]: 8-bit sRGB -> linear, then 1024 pairs (f(i/1024), f((i+1)/1024) - f(i/1024)) of the CIE-Lab
        transfer function (cube root above 0.008856, linear toe below); values evaluated in float64 and rounded once,
        differences taken in float32; shared with oracle.px_dbscan"""
        v = np.arange(256, dtype=np.float64) / 255.0
        lin = np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)
        t = np.arange(1025, dtype=np.float64) / 1024.0
        f = np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0).astype(np.float32)
        fd = np.stack([f[:-1], f[1:] - f[:-1]], axis=1).reshape(-1)
        return np.concatenate([lin, fd]).astype(np.float32)

    def px_dbscan(self, rgb, radius, eps, spatial_weight, min_pts, want_count=False):
        """rgb uint8[H,W,3] device -> labels int32[H,W] (0 = noise, else 1 + smallest pixel index of the cluster),
        core mask, optional neighbour counts."""
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous()
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        if getattr(self, "_px_lut", None) is None:
            self._px_lut = self.dev(self.px_tables())
        parent = self.empty((H, W), torch.int32)
        count = self.empty((H, W), torch.uint8) if want_count else None
        self._check(self.lib.rhccq_px_neighbours(self.ctx, self._p(rgb), H, W, int(radius), float(eps), float(spatial_weight), int(min_pts),
                                                 self._p(self._px_lut), self._p(parent), self._p(count)), "px_neighbours")
        core = parent >= 0
        labels = self.empty((H, W), torch.int32)
        self._check(self.lib.rhccq_px_expand(self.ctx, self._p(rgb), H, W, int(radius), float(eps), float(spatial_weight), self._p(self._px_lut),
                                             self._p(parent), self._p(labels)), "px_expand")
        return (labels, core, count) if want_count else (labels, core)

    # -- quality metrics (comparison.py:30-80) ------------------------------------------------------
    def error_sums(self, a, b):
        """a, b: uint8[H,W,3] device -> int64[5]: per-channel sum of squared differences, sum |d|, max |d|."""
        assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        sums = self.empty((5,), torch.int64)
        self._check(self.lib.rhccq_error_sums(self.ctx, self._p(a), self._p(b), a.numel() // 3, self._p(sums)), "error_sums")
        return sums.cpu().numpy()

    def error_tables(self, a, b, want_maxerr=False):
        """a, b: uint8[H,W,3] device -> int64[256][3] rows by worst-channel error (+ uint8[H,W] per-pixel worst error)"""
        assert a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
        tab = self.empty((256, 3), torch.int64)
        me = self.empty(tuple(a.shape[:2]), torch.uint8) if want_maxerr else None
        self._check(self.lib.rhccq_error_tables(self.ctx, self._p(a), self._p(b), a.numel() // 3, self._p(tab), self._p(me)), "error_tables")
        return tab.cpu().numpy(), me

    def ssim7(self, a, b):
        """mean SSIM per channel (float64[3]) with skimage's defaults for win_size=7, data_range=255."""
        H, W = int(a.shape[0]), int(a.shape[1])
        nb = int(self.lib.rhccq_ssim7_blocks(H, W))
        if nb == 0:
            raise ValueError("win_size exceeds image extent")          # skimage's message
        part = self.empty((nb, 3), torch.float64)
        self._check(self.lib.rhccq_ssim7_sums(self.ctx, self._p(a), self._p(b), H, W, self._p(part), nb), "ssim7")
        return part.cpu().numpy().sum(axis=0) / float((H - 6) * (W - 6))

    # -- split score (split_score.py:15-142) ----------------------------------------------------------
    def split_stats(self, rgb, mask=None):
        """rgb uint8[H,W,3] device, mask uint8[H,W] device or None -> (sums float64[12], lbp_hist int64[10], gray_hist int64[32])"""
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.shape[-1] == 3
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        nb = int(self.lib.rhccq_split_stats_blocks(H, W))
        part = self.empty((nb, 12), torch.float64)
        hist = self.empty((42,), torch.int32)
        self._check(self.lib.rhccq_split_stats(self.ctx, self._p(rgb), H, W, self._p(mask), self._p(part), nb, self._p(hist)), "split_stats")
        h = hist.cpu().numpy().astype(np.int64)
        return part.cpu().numpy().sum(axis=0), h[:10], h[10:]

    # -- ROI stage: connected components, buffer zone ---------------------------------------------------
    def ccl(self, mask, connectivity=8, cap=4096, numbering="opencv", host_stats=True):
        """mask uint8 / bool [H,W] device -> (n, labels int32[H,W] device, stats np.int32[n + 1, 5]) with cv2's conventions:
        label 0 = background, components numbered as cv2.connectedComponentsWithStats numbers them (csrc/ccl.hip), stats
        columns CC_STAT_LEFT, TOP, WIDTH, HEIGHT, AREA.  numbering="raster": components numbered by their first pixel in
        raster order (scipy.ndimage.label, skimage.measure.label); numbering="ids": unordered ids 1..n, no statistics (None)."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        assert mask.dtype == torch.uint8 and mask.dim() == 2 and mask.is_contiguous()
        H, W = int(mask.shape[0]), int(mask.shape[1])
        labels = self.empty((H, W), torch.int32)
        count = self.empty((1,), torch.int32)
        while True:
            wb = int(self.lib.rhccq_ccl_work_bytes(H, W, cap))
            work = self.empty((wb,), torch.uint8)
            stats = self.empty((cap + 1, 5), torch.int32)
            self._check(self.lib.rhccq_ccl(self.ctx, self._p(mask), H, W, int(connectivity), {"opencv": 0, "raster": 1, "ids": 2}[numbering], self._p(work),
                                           wb, cap, self._p(labels), self._p(None if numbering == "ids" else stats), self._p(count)), "ccl")
            n = int(count.cpu()[0])
            if n <= cap or numbering == "ids":
                break
            cap = n
        if numbering == "ids":
            return n, labels, None
        return n, labels, (stats[:n + 1].cpu().numpy() if host_stats else stats)   # host_stats=False: the device tensor [cap + 1][5]

    def ccl_keys(self, labels, n, y0, x0, frame_w, connectivity=8, numbering="opencv"):
        """-> np.uint32[n + 1]: the ordering key (frame coordinates) of every component of a tile labelled by ccl(); see rhccq_ccl_keys"""
        H, W = int(labels.shape[0]), int(labels.shape[1])
        keys = self.empty((n + 1,), torch.int32)
        self._check(self.lib.rhccq_ccl_keys(self.ctx, self._p(labels), H, W, int(y0), int(x0), int(frame_w), {"opencv": 0, "raster": 1}[numbering],
                                            int(connectivity), int(n), self._p(keys)), "ccl_keys")
        return keys.cpu().numpy().view(np.uint32)

    def ccl_select(self, labels, lut):
        """labels int32[H,W] device, lut uint8[n + 1] (numpy, or a device tensor) -> uint8[H,W] device = lut[labels]"""
        out = self.empty(tuple(labels.shape), torch.uint8)
        d_lut = lut if torch.is_tensor(lut) else self.dev(np.asarray(lut, np.uint8))   # (named: a temporary would be freed before the launch)
        self._check(self.lib.rhccq_ccl_select(self.ctx, self._p(labels), self._p(d_lut), labels.numel(), self._p(out)), "ccl_select")
        return out

    def roi_buffer(self, region_map, rgb, buffer_size=3):
        """extract_roi_nonroi on the device: -> (roi_image, nonroi_image uint8[H,W,3], roi_mask, nonroi_mask bool[H,W])"""
        assert region_map.dtype == torch.uint8 and rgb.dtype == torch.uint8 and region_map.is_contiguous() and rgb.is_contiguous()
        H, W = int(rgb.shape[0]), int(rgb.shape[1])
        assert tuple(region_map.shape) == (H, W) and rgb.shape[-1] == 3
        rm, nm = self.empty((H, W), torch.uint8), self.empty((H, W), torch.uint8)
        ri, ni = torch.empty_like(rgb), torch.empty_like(rgb)
        self._check(self.lib.rhccq_roi_buffer(self.ctx, self._p(region_map), self._p(rgb), H, W, int(buffer_size), self._p(rm), self._p(nm),
                                              self._p(ri), self._p(ni)), "roi_buffer")
        return ri, ni, rm.view(torch.bool), nm.view(torch.bool)

    # -- ROI stage: edge front end (csrc/edges.hip) -------------------------------------------------------
    def edges_gray(self, rgb):
        """rgb uint8[H,W,3] device -> (gray uint8[H,W] device, histogram np.int64[256])"""
        assert rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.shape[-1] == 3
        gray = self.empty(tuple(rgb.shape[:2]), torch.uint8)
        hist = self.empty((256,), torch.int32)
        self._check(self.lib.rhccq_edges_gray(self.ctx, self._p(rgb), gray.numel(), self._p(gray), self._p(hist)), "edges_gray")
        return gray, hist.cpu().numpy().astype(np.int64)

    def edges_grad_hist(self, gray):
        """gray uint8[H,W] device -> (values, counts): the distinct gx^2 + gy^2 of the 3x3 Sobel (BORDER_REFLECT_101), ascending"""
        H, W = int(gray.shape[0]), int(gray.shape[1])
        hist = self.empty((int(self.lib.rhccq_edges_m2_bins()),), torch.int32)
        self._check(self.lib.rhccq_edges_grad_hist(self.ctx, self._p(gray), H, W, self._p(hist)), "edges_grad_hist")
        h = self.to_host(hist)                               # (8 MB of bins: through a page-locked buffer)
        v = np.flatnonzero(h)
        return v.astype(np.int64), h[v].astype(np.int64)

    def canny_nms(self, img):
        """img uint8[H,W] or [H,W,3] device -> uint16[H,W] device (stored as int16): Canny's magnitude at the local maxima, 0 elsewhere"""
        assert img.dtype == torch.uint8 and img.is_contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        cn = 1 if img.dim() == 2 else int(img.shape[2])
        mag, nm = self.empty((H, W), torch.int16), self.empty((H, W), torch.int16)
        dxy = self.empty((H, W), torch.int32)
        self._check(self.lib.rhccq_canny_nms(self.ctx, self._p(img), H, W, cn, self._p(mag), self._p(dxy), self._p(nm)), "canny_nms")
        return nm

    def canny_label(self, nm, low, gray=None):
        """first half of Canny's hysteresis on the device: components of {nm > low} (unordered ids: their numbering does not matter here) and
        per label the max of nm, the sums of gray and gray^2, the pixel count.  -> (n, labels int32[H,W] device, red device int64[n + 1][4]);
        only the component count crosses to the host"""
        H, W = int(nm.shape[0]), int(nm.shape[1])
        mask = self.empty((H, W), torch.uint8)
        self._check(self.lib.rhccq_edges_above(self.ctx, self._p(nm), H * W, int(low), self._p(mask)), "edges_above")
        n, labels, _ = self.ccl(mask, 8, cap=0, numbering="ids")
        red = self.empty((n + 1, 4), torch.int64)
        self._check(self.lib.rhccq_label_reduce(self.ctx, self._p(labels), self._p(nm), self._p(gray), labels.numel(), n, self._p(red)), "label_reduce")
        return n, labels, red

    def canny_verdict(self, n, red, high, want_lut=False):
        """second half: a component is an edge when its max exceeds `high`.  -> (lut uint8[n + 1] device or None, (edge components, edge
        pixels, sum gray, sum gray^2) as Python ints): 32 bytes cross to the host"""
        out4 = self.empty((4,), torch.int64)
        lut = self.empty((n + 1,), torch.uint8) if want_lut else None
        self._check(self.lib.rhccq_edge_score(self.ctx, self._p(red), n, int(high), self._p(out4), self._p(lut)), "edge_score")
        return lut, tuple(int(v) for v in out4.cpu().numpy())

    CANNY_SCORES_CAP = 1 << 20      # labels the fused scoring reduces per labelling (32 MB of per-label sums); beyond: the two-step path
    CANNY_SCORES_NESTED = True      # one union-find grown over the descending thresholds (rhccq_canny_scores_nested) instead of a labelling per `low`

    def canny_scores(self, nm, gray, pairs, nested=None):
        """[(edge components, edge pixels, sum gray, sum gray^2)] of several (low, high) pairs (normalised, low <= high) in ONE call and ONE
        read-back: rhccq_canny_scores_nested (every pixel linked once over the whole search), or with nested=False rhccq_canny_scores (pairs
        that share `low` share a labelling of {nm > low} from scratch) -- the same numbers"""
        H, W = int(nm.shape[0]), int(nm.shape[1])
        if self.CANNY_SCORES_NESTED if nested is None else nested:
            lows = (C.c_int32 * len(pairs))(*[int(p[0]) for p in pairs])
            highs = (C.c_int32 * len(pairs))(*[int(p[1]) for p in pairs])
            wb = int(self.lib.rhccq_canny_scores_nested_bytes(H, W))
            work = self.empty((wb,), torch.uint8)
            out = self.empty((len(pairs), 5), torch.int64)
            self._check(self.lib.rhccq_canny_scores_nested(self.ctx, self._p(nm), self._p(gray), H, W, lows, highs, len(pairs), self._p(work), wb, self._p(out)),
                        "canny_scores_nested")
            return [tuple(int(v) for v in row[:4]) for row in out.cpu().numpy()]
        order = sorted(range(len(pairs)), key=lambda i: pairs[i])
        lows = (C.c_int32 * len(pairs))(*[int(pairs[i][0]) for i in order])
        highs = (C.c_int32 * len(pairs))(*[int(pairs[i][1]) for i in order])
        cap = self.CANNY_SCORES_CAP
        wb = int(self.lib.rhccq_canny_scores_bytes(H, W, cap))
        work = self.empty((wb,), torch.uint8)
        out = self.empty((len(pairs), 5), torch.int64)
        self._check(self.lib.rhccq_canny_scores(self.ctx, self._p(nm), self._p(gray), H, W, lows, highs, len(pairs), cap, self._p(work), wb, self._p(out)),
                    "canny_scores")
        res = out.cpu().numpy()
        fours = [None] * len(pairs)
        for j, i in enumerate(order):
            if int(res[j, 4]) > cap:                              # more components than the fused reduction holds: score this pair the long way
                fours[i] = self.canny_components(nm, int(pairs[i][0]), int(pairs[i][1]), gray)[2]
            else:
                fours[i] = tuple(int(v) for v in res[j, :4])
        return fours

    def canny_components(self, nm, low, high, gray=None, want_lut=False):
        """both halves -> (labels, lut or None, the four numbers)"""
        n, labels, red = self.canny_label(nm, low, gray)
        lut, four = self.canny_verdict(n, red, high, want_lut)
        return labels, lut, four

    def label_reduce(self, labels, n, val16=None, val8=None):
        """-> np.uint64[n + 1, 4]: per label {max of val16, sum of val8, sum of val8^2, pixel count}"""
        red = self.empty((n + 1, 4), torch.int64)
        self._check(self.lib.rhccq_label_reduce(self.ctx, self._p(labels), self._p(val16), self._p(val8), labels.numel(), n, self._p(red)), "label_reduce")
        return red.cpu().numpy().view(np.uint64)

    def box_count(self, mask, kernel_size):
        """mask uint8 / bool [H,W] device -> uint16[H,W] device (int16 storage): non-zero pixels per k x k window, BORDER_REFLECT_101"""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        H, W = int(mask.shape[0]), int(mask.shape[1])
        out = self.empty((H, W), torch.int16)
        self._check(self.lib.rhccq_box_count(self.ctx, self._p(mask), H, W, int(kernel_size), self._p(out)), "box_count")
        return out

    # -- ROI stage: clean-up chain (csrc/morph.hip); masks are uint8[H,W] device planes, set = non-zero, results 0 / 255 ----
    def morph(self, mask, half_widths, erode=False):
        """cv2.dilate / cv2.erode by the symmetric structuring element given as one half-width per row (-1 = empty row)"""
        H, W = int(mask.shape[0]), int(mask.shape[1])
        hw = (C.c_int32 * len(half_widths))(*[int(v) for v in half_widths])
        out = self.empty((H, W), torch.uint8)
        self._check(self.lib.rhccq_morph_dilate(self.ctx, self._p(mask), H, W, len(half_widths) // 2, hw, int(erode), int(erode), self._p(out)), "morph_dilate")
        return out

    def morph_close(self, mask, half_widths):
        return self.morph(self.morph(mask, half_widths), half_widths, erode=True)

    def morph_rect(self, mask, ksize, erode=False):
        """cv2.dilate / cv2.erode by a ksize x ksize rectangle, OpenCV's anchor ksize // 2 (an even element reaches one pixel
        further up / left than down / right)"""
        H, W = int(mask.shape[0]), int(mask.shape[1])
        a, b = int(ksize) // 2, int(ksize) - 1 - int(ksize) // 2
        left = (C.c_int32 * (a + b + 1))(*([a] * (a + b + 1)))
        right = (C.c_int32 * (a + b + 1))(*([b] * (a + b + 1)))
        out = self.empty((H, W), torch.uint8)
        self._check(self.lib.rhccq_morph_dilate_spans(self.ctx, self._p(mask), H, W, a, b, left, right, int(erode), int(erode), self._p(out)), "morph_dilate_spans")
        return out

    def morph_close_rect(self, mask, ksize):
        return self.morph_rect(self.morph_rect(mask, ksize), ksize, erode=True)

    def box_filter_seq(self, plane, kernel_size, scale255):
        """plane uint8[H,W] device -> float32[H,W] device: compute_local_density through OpenCV's direct filter2D path (odd kernels <= 11)"""
        H, W = int(plane.shape[0]), int(plane.shape[1])
        out = self.empty((H, W), torch.float32)
        self._check(self.lib.rhccq_box_filter_seq(self.ctx, self._p(plane), H, W, int(kernel_size), int(bool(scale255)), self._p(out)), "box_filter_seq")
        return out

    def lut_u16(self, values, table_u32):
        """values uint16 (int16 storage) device plane, table np.uint32[n] -> int32 device plane of table[value] (bit copies through rhccq_lut_u16_f32)"""
        t = self.dev(np.ascontiguousarray(table_u32, np.uint32).view(np.float32))
        out = self.empty(tuple(values.shape), torch.int32)
        self._check(self.lib.rhccq_lut_u16_f32(self.ctx, self._p(values), self._p(t), int(t.numel()), values.numel(), self._p(out)), "lut_u16_f32")
        return out

    def mask_op(self, a, b, op):
        """op: 'or', 'and', 'andnot' (a & ~b), 'not' (~a)"""
        out = torch.empty_like(a)
        self._check(self.lib.rhccq_mask_op(self.ctx, self._p(a), self._p(b), a.numel(), ("or", "and", "andnot", "not").index(op), self._p(out)), "mask_op")
        return out

    def gap_bridge(self, mask, counts, min_count, reach):
        out = torch.empty_like(mask)
        self._check(self.lib.rhccq_gap_bridge(self.ctx, self._p(mask), self._p(counts), counts.element_size(), int(mask.shape[0]), int(mask.shape[1]),
                                              int(min_count), int(reach), self._p(out)), "gap_bridge")
        return out

    def dist_chamfer(self, mask):
        """cv2.distanceTransform(mask, DIST_L2, 3) in fixed point (int32[H,W] device, 16 fractional bits)"""
        H, W = int(mask.shape[0]), int(mask.shape[1])
        hz = self.empty((H, W), torch.int16)
        dist = self.empty((H, W), torch.int32)
        self._check(self.lib.rhccq_dist_chamfer(self.ctx, self._p(mask), H, W, self._p(hz), self._p(dist)), "dist_chamfer")
        return dist

    def binary_sobel(self, mask):
        """-> (uint8[H,W] device: gx^2 + gy^2 of the 0/1 image's 3x3 Sobel, its maximum)"""
        H, W = int(mask.shape[0]), int(mask.shape[1])
        m2 = self.empty((H, W), torch.uint8)
        mx = self.empty((1,), torch.int32)
        self._check(self.lib.rhccq_binary_sobel(self.ctx, self._p(mask), H, W, self._p(m2), self._p(mx)), "binary_sobel")
        return m2, int(mx.cpu()[0])

    def lut_u8(self, plane, lut256):
        out = torch.empty_like(plane)
        d_lut = self.dev(np.asarray(lut256, np.uint8))
        self._check(self.lib.rhccq_lut_u8(self.ctx, self._p(plane), self._p(d_lut), plane.numel(), self._p(out)), "lut_u8")
        return out

    def box_sum(self, plane, kernel_size):
        """plane uint8[H,W] device -> int32[H,W] device: sum of the pixel values per k x k window, BORDER_REFLECT_101"""
        H, W = int(plane.shape[0]), int(plane.shape[1])
        out = self.empty((H, W), torch.int32)
        self._check(self.lib.rhccq_box_sum(self.ctx, self._p(plane), H, W, int(kernel_size), self._p(out)), "box_sum")
        return out

    def masked_hist(self, mask, values, n_bins):
        """-> np.int64[n_bins]: histogram of the uint16 plane (int16 storage) over the pixels where mask is set"""
        hist = self.empty((n_bins,), torch.int64)
        self._check(self.lib.rhccq_masked_hist(self.ctx, self._p(mask), self._p(values), values.numel(), int(n_bins), self._p(hist)), "masked_hist")
        return hist.cpu().numpy()

    def value_mask(self, values, min_value, mask=None):
        """-> uint8 plane: 255 where values >= min_value (and mask set, when given)"""
        out = self.empty(tuple(values.shape), torch.uint8)
        self._check(self.lib.rhccq_value_mask(self.ctx, self._p(mask), self._p(values), values.numel(), int(min_value), self._p(out)), "value_mask")
        return out

    def label_sum(self, labels, n, values):
        """-> np.uint64[n + 1]: per label (0 = background included) the sum of a uint16 (int16 storage) or non-negative int32 plane"""
        sums = self.empty((n + 1,), torch.int64)
        self._check(self.lib.rhccq_label_sum(self.ctx, self._p(labels), self._p(values), values.element_size(), labels.numel(), n, self._p(sums)), "label_sum")
        return sums.cpu().numpy().view(np.uint64)

    # -- K6 / decode ------------------------------------------------------------------------------
    def remap(self, idx, lut):
        out = torch.empty_like(idx)
        self._check(self.lib.rhccq_remap(self.ctx, self._p(idx), idx.numel(), self._p(lut), lut.numel(), self._p(out)), "remap")
        return out

    def decode(self, idx, palette):
        """palette uint8[K,3] device, idx uint8/uint16(int16 storage)/int32 device flat -> uint8[n,3]."""
        out = self.empty((idx.numel(), 3), torch.uint8)
        self._check(self.lib.rhccq_decode(self.ctx, self._p(idx), idx.element_size(), idx.numel(), self._p(palette),
                                          palette.shape[0], self._p(out)), "decode")
        return out

    # -- K5 ---------------------------------------------------------------------------------------
    def merge_firstpos(self, idx, h, w, top, left, canvas_h, canvas_w, pal_n):
        fp = torch.full((max(pal_n, 1),), INT_MAX, dtype=torch.int32, device=self.device)
        self._check(self.lib.rhccq_merge_firstpos(self.ctx, self._p(idx), h, w, top, left, canvas_h, canvas_w, pal_n, self._p(fp)),
                    "merge_firstpos")
        return fp[:pal_n]

    def merge_paint(self, idx, h, w, top, left, canvas, lut, pal_n):
        self._check(self.lib.rhccq_merge_paint(self.ctx, self._p(idx), h, w, top, left, canvas.shape[0], canvas.shape[1],
                                               self._p(lut), pal_n, self._p(canvas)), "merge_paint")

    # -- extension ----------------------------------------------------------------------------------
    def luma_qstep(self, rgb, roi_mask, block, q_roi, q_bg):
        H, W = rgb.shape[0], rgb.shape[1]
        luma = self.empty((H, W), torch.float32)
        qstep = self.empty((H // block, W // block), torch.float32)
        self._check(self.lib.rhccq_luma_qstep(self.ctx, self._p(rgb), self._p(roi_mask), H, W, block, float(q_roi), float(q_bg),
                                              self._p(luma), self._p(qstep)), "luma_qstep")
        return luma, qstep

    def dct_quant(self, plane, block, qstep, want_coef=True):
        H, W = plane.shape
        coef = self.empty((H, W), torch.float32) if want_coef else None
        q = self.empty((H, W), torch.int16)
        self._check(self.lib.rhccq_dct_quant(self.ctx, self._p(plane), H, W, block, self._p(qstep), self._p(coef), self._p(q)), "dct_quant")
        return coef, q

    def sync(self):
        self._check(self.lib.rhccq_sync(self.ctx), "sync")


_default = {}


def default_context(device=None):
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    if device not in _default:
        _default[device] = Rhccq(device)
    return _default[device]
