"""Tile-parallel encode (2 ranks, gloo, both on the one GPU of the test box) == single-GPU encode,
bit for bit.  GPU only; at most 2 processes touch the card."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(size):
    from roibasedimagecompression_amd import synth
    if size == "8k":                                         # BASELINE.json configs[3]: one 7680x4320 frame
        H, W = 4320, 7680
        img = synth.photo(H, W, 1234)
        (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (2, 1))
        return img, (lr, nr, br), (ln, nn, bn)
    H, W = 128, 192
    img = synth.photo(H, W, 31)
    img[60:64, 90:100] = 0                                   # in-segment black across the tile seam
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (2, 2))
    return img, (lr, nr, br), (ln, nn, bn)


def _worker(rank, world, port, ret, size):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from roibasedimagecompression_amd.frame import ClassSpec
        from roibasedimagecompression_amd.ops import Rhccq
        from roibasedimagecompression_amd.parallel import TiledFrameEncoder, tile_grid
        img, (lr, nr, br), (ln, nn, bn) = _inputs(size)
        H, W = img.shape[:2]
        rh = Rhccq(0)
        tiles = tile_grid(H, W, 1, world)
        r0, c0, h, w = tiles[rank]
        sl = (slice(r0, r0 + h), slice(c0, c0 + w))
        specs = [ClassSpec(torch.from_numpy(np.ascontiguousarray(lr[sl])).to(rh.device), np.zeros(nr, np.int64), [br], 20),
                 ClassSpec(torch.from_numpy(np.ascontiguousarray(ln[sl])).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
        enc = TiledFrameEncoder(rh, (H, W), tiles[rank]).set_tiles(tiles)
        out = enc.encode(torch.from_numpy(np.ascontiguousarray(img[sl])).to(rh.device), specs)
        idx = out["indices"].cpu().numpy()
        ret[rank] = (out["palette"], idx.view(np.uint16) if out["indices_dtype"] == "uint16" else idx, tiles[rank])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", ["small", "8k"])
def test_tiled_two_ranks_equals_single_gpu(size):
    """`8k`: the full-size frame of BASELINE.json configs[3] (too large for the oracle) -- the size-independent
    property is that the tile-parallel result equals the single-GPU one bit for bit."""
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    from roibasedimagecompression_amd.ops import Rhccq
    img, (lr, nr, br), (ln, nn, bn) = _inputs(size)
    H, W = img.shape[:2]
    rh = Rhccq(0)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
    single = FrameEncoder(rh).encode(torch.from_numpy(img).to(rh.device), specs)
    sidx = single["indices"].cpu().numpy()
    if single["indices_dtype"] == "uint16":
        sidx = sidx.view(np.uint16)
    mgr = mp.Manager()
    ret = mgr.dict()
    del specs
    torch.cuda.empty_cache()
    mp.spawn(_worker, args=(2, _free_port(), ret, size), nprocs=2, join=True)
    full = np.zeros((H, W), np.int64)
    for rank in (0, 1):
        pal, idx, (r0, c0, h, w) = ret[rank]
        assert np.array_equal(pal, single["palette"])
        full[r0:r0 + h, c0:c0 + w] = idx
    assert np.array_equal(full, sidx.astype(np.int64))


def _nccl_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from roibasedimagecompression_amd.frame import ClassSpec
        from roibasedimagecompression_amd.ops import Rhccq
        from roibasedimagecompression_amd.parallel import TiledFrameEncoder, tile_grid
        img, (lr, nr, br), (ln, nn, bn) = _inputs("small")
        H, W = img.shape[:2]
        rh = Rhccq(rank)
        tiles = tile_grid(H, W, 1, world)
        r0, c0, h, w = tiles[rank]
        sl = (slice(r0, r0 + h), slice(c0, c0 + w))
        specs = [ClassSpec(torch.from_numpy(np.ascontiguousarray(lr[sl])).to(rh.device), np.zeros(nr, np.int64), [br], 20),
                 ClassSpec(torch.from_numpy(np.ascontiguousarray(ln[sl])).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
        enc = TiledFrameEncoder(rh, (H, W), tiles[rank]).set_tiles(tiles)
        out = enc.encode(torch.from_numpy(np.ascontiguousarray(img[sl])).to(rh.device), specs)
        idx = out["indices"].cpu().numpy()
        ret[rank] = (out["palette"], idx.view(np.uint16) if out["indices_dtype"] == "uint16" else idx, tiles[rank])
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL path (the one-GPU box runs the same code over gloo)")
def test_tiled_two_ranks_over_rccl():
    """the tile-parallel exchange over backend `nccl` (= RCCL over xGMI), payloads staying on the device, two GPUs"""
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    from roibasedimagecompression_amd.ops import Rhccq
    img, (lr, nr, br), (ln, nn, bn) = _inputs("small")
    H, W = img.shape[:2]
    rh = Rhccq(0)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], 20),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], 10)]
    single = FrameEncoder(rh).encode(torch.from_numpy(img).to(rh.device), specs)
    sidx = single["indices"].cpu().numpy()
    if single["indices_dtype"] == "uint16":
        sidx = sidx.view(np.uint16)
    ret = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    full = np.zeros((H, W), np.int64)
    for rank in (0, 1):
        pal, idx, (r0, c0, h, w) = ret[rank]
        assert np.array_equal(pal, single["palette"])
        full[r0:r0 + h, c0:c0 + w] = idx
    assert np.array_equal(full, sidx.astype(np.int64))


def _ccl_mask(size):
    """an edge-map-like mask: thresholded gradient of a synthetic photo plus sparse noise (tens of thousands of components at 8K,
    many of them crossing the tile seams)"""
    from roibasedimagecompression_amd import synth
    H, W = (4320, 7680) if size == "8k" else (96, 160)
    img = synth.photo(H, W, 77).astype(np.int32).sum(2)
    g = np.abs(np.diff(img, axis=0, prepend=img[:1])) + np.abs(np.diff(img, axis=1, prepend=img[:, :1]))
    m = g > 14
    m ^= np.random.default_rng(4).random((H, W)) < 0.002
    m[H // 2 - 1:H // 2 + 1, ::7] = True                     # runs that straddle the horizontal seam of a 2-row tiling
    return m


def _ccl_tile_worker(rank, world, port, ret, size, grid):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hashlib
        from roibasedimagecompression_amd.ops import Rhccq
        from roibasedimagecompression_amd.parallel import tile_grid, tiled_ccl
        m = _ccl_mask(size)
        H, W = m.shape
        rh = Rhccq(0)
        tiles = tile_grid(H, W, *grid)
        r0, c0, h, w = tiles[rank]
        t = torch.from_numpy(np.ascontiguousarray(m[r0:r0 + h, c0:c0 + w])).to(rh.device)
        res = {}
        for conn, numbering in ((8, "opencv"), (4, "opencv"), (8, "raster")):
            n, lab, stats = tiled_ccl(rh, t, tiles[rank], (H, W), tiles, None, conn, numbering)
            res[(conn, numbering)] = (n, hashlib.sha256(lab.cpu().numpy().tobytes()).hexdigest(), stats)
        ret[rank] = (res, tiles[rank])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size,grid", [("small", (1, 2)), ("small", (2, 1)), ("8k", (1, 2)), ("8k", (2, 1))])
def test_tiled_connected_components_equal_single_gpu(size, grid):
    """Seam-stitched connected components (parallel.tiled_ccl: per-tile rhccq_ccl + rhccq_ccl_keys, ONE all-gather of border labels
    and component tables, union-find across the seams, relabel) on 2 ranks sharing the GPU (gloo) == rhccq_ccl on the whole mask:
    labels, numbering (OpenCV's block order / raster order) and statistics, 4- and 8-connectivity; `8k` = the 7680x4320 frame of
    BASELINE.json configs[3] cut side by side and one above the other."""
    import hashlib
    from roibasedimagecompression_amd.ops import Rhccq
    from roibasedimagecompression_amd.parallel import tile_grid
    m = _ccl_mask(size)
    H, W = m.shape
    rh = Rhccq(0)
    t = torch.from_numpy(m).to(rh.device)
    want = {}
    for conn, numbering in ((8, "opencv"), (4, "opencv"), (8, "raster")):
        n, lab, stats = rh.ccl(t, conn, numbering=numbering)
        want[(conn, numbering)] = (n, lab.cpu().numpy(), stats)
    del t
    torch.cuda.empty_cache()
    ret = mp.Manager().dict()
    mp.spawn(_ccl_tile_worker, args=(2, _free_port(), ret, size, grid), nprocs=2, join=True)
    tiles = tile_grid(H, W, *grid)
    for key, (n, lab, stats) in want.items():
        for rank in (0, 1):
            res, (r0, c0, h, w) = ret[rank]
            gn, gsha, gstats = res[key]
            assert gn == n, (key, gn, n)
            assert np.array_equal(gstats[1:], stats[1:]) and np.array_equal(gstats[0], stats[0]), key
            assert gsha == hashlib.sha256(np.ascontiguousarray(lab[r0:r0 + h, c0:c0 + w]).tobytes()).hexdigest(), (key, rank)
    assert want[(8, "opencv")][0] > (10000 if size == "8k" else 10)
