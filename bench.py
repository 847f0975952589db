#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the RHCCQ encoder hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input resident in HBM.  Four workloads (--mode):

  frame   (default; BASELINE.json configs[1]) one 3840x2160 RGB frame per GPU: per-segment unique colours ->
          three-level palette clustering (rhccq.ipynb:978-1039) -> final palette + index map, followed by the
          DCT/quantisation EXTENSION tile (8x8, two-tier ROI/background steps) that BASELINE.json's metric names
          but the reference does not contain; the extension is inside the timed region so that no
          named work is skipped, its share is reported in `stages_ms`.  Frames are independent: every rank encodes its
          own frame, no data-path collective; value = pixels of all ranks / max-over-ranks time ("weak").
  batch   (configs[2]) a batch of 64 distinct 1920x1080 frames per GPU and step, as two sub-batches of 32 in flight through
          stream.StreamEncoder (each one FrameEncoder.encode_batch: one batched clustering call per level for its frames),
          two quality tiers (20, 10); value = pixels of all ranks / time ("weak").
  tiled   (configs[3]) ONE 7680x4320 frame cut into one tile per GPU (2x4 at 8 GPUs), pixels stay tile-local, palettes
          are exchanged: 1 all-gather (segment bitmaps + stats) + up to 3 small MIN all-reduces over RCCL
          (parallel.TiledFrameEncoder); value = frame pixels / max-over-ranks time ("strong").
  stream  (configs[4]) a stream of 4K frames per GPU through stream.StreamEncoder (batches in flight on host threads
          with their own HIP streams), 16x16 DCT extension, two quality tiers; value = pixels of all ranks / time ("weak").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode frame|batch|tiled|stream]

With --gpus N > 1 and no torch.distributed environment the script starts the N ranks itself (torch.distributed.run on
127.0.0.1) BEFORE anything touches a GPU, and exits with their code; the driver's own launch line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
is taken as is.  A --gpus that does not match WORLD_SIZE, or more ranks than devices, is an error (exit 2), never a silent
single-GPU run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TOPS = 157.3           # fp32 / int32 vector peak of the chip (MI355X_MICROARCH.md), for the brute-force equivalent
N_CUS = 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["frame", "batch", "tiled", "stream"], default="frame")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--quality", type=int, default=20, help="ROI quality tier (configs[1]: the only tier)")
    ap.add_argument("--quality-bg", type=int, default=None, help="background tier (stream / tiled modes default to 10: rhccq_20_10)")
    ap.add_argument("--sigma", type=float, default=2.0, help="sensor-noise sigma of the synthetic photo")
    ap.add_argument("--block", type=int, default=None, help="DCT extension block (8; 16 in stream mode)")
    ap.add_argument("--cpu-sample", type=int, default=1280, help="edge of the CPU-baseline crop (0 = skip)")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--host", choices=["native", "python"], default="native",
                    help="frame mode: host side of the encoder -- rhccq_encode_frame (C++, csrc/encode_frame.hip) or the Python FrameEncoder")
    ap.add_argument("--frames-per-step", type=int, default=None,
                    help="frames encoded together per step and GPU (frame mode: 1 = configs[1]; stream mode: 12)")
    ap.add_argument("--lanes", type=int, default=None, help="stream mode: batches in flight on this many host threads (10)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process would see, counted WITHOUT touching HIP (the parent must stay GPU-free: it only starts the ranks): KFD
    topology nodes with SIMDs, cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None when the topology cannot be read (then the
    ranks themselves report a missing device)."""
    import glob
    n = 0
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    for path in paths:
        try:
            props = dict(ln.split(None, 1) for ln in open(path).read().splitlines() if " " in ln)
            n += int(props.get("simd_count", "0")) > 0
        except Exception:
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """--gpus N without a torch.distributed environment: start the N ranks (before any GPU call) and relay their exit code"""
    have = visible_gpus()
    if have is not None and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} device(s) are visible\n")
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def build_inputs(rh, H, W, seed, tiles, q_roi, q_non, sigma, tile=None):
    """synthetic frame + class label maps on the device; `tile` = (r0, c0, h, w) keeps only that window (tiled mode)"""
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec
    img = synth.photo(H, W, seed, sigma=sigma)
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
    if tile is not None:
        r0, c0, h, w = tile
        sl = (slice(r0, r0 + h), slice(c0, c0 + w))
        img_d, lr_d, ln_d = (np.ascontiguousarray(a[sl]) for a in (img, lr, ln))
    else:
        img_d, lr_d, ln_d = img, lr, ln
    rgb = torch.from_numpy(img_d).to(rh.device)
    specs = [ClassSpec(torch.from_numpy(lr_d).to(rh.device), np.zeros(nr, np.int64), [br], q_roi),
             ClassSpec(torch.from_numpy(ln_d).to(rh.device), np.zeros(nn, np.int64), [bn], q_non)]
    roi_mask = torch.from_numpy((lr_d > 0).astype(np.uint8)).to(rh.device)
    return img, rgb, specs, roi_mask, (lr, ln, br, bn)


def dct_ext(rh, rgb, roi_mask, block):
    luma, qstep = rh.luma_qstep(rgb, roi_mask, block, 4.0, 16.0)
    return rh.dct_quant(luma, block, qstep, want_coef=False)


def one_step(rh, enc, rgb, specs, roi_mask, block, stage_acc=None, extra=(), native=False):
    """one pass of the hot path over one batch: the frame (plus `extra` further frames when the step is a
    batch, --frames-per-step) and the DCT/quantisation extension of every frame"""
    import torch
    if extra:
        out = enc.encode_batch([(rgb, specs)] + [(r, sp) for r, sp, _ in extra])[0]
    elif native:
        out = enc.encode_native(rgb, specs)
    else:
        out = enc.encode(rgb, specs)
    t0 = time.perf_counter()
    dct_ext(rh, rgb, roi_mask, block)
    for r, sp, m in extra:
        dct_ext(rh, r, m, block)
    if stage_acc is not None:
        torch.cuda.synchronize()
        for k, v in enc.timings.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        stage_acc["dct_ext"] = stage_acc.get("dct_ext", 0.0) + (time.perf_counter() - t0)
    return out


# ---- probes -------------------------------------------------------------------------------------------------------
def hbm_probe(rh, rgb, specs, iters=5):
    """HIP-event timing (events recorded on the stream the kernel is launched on) of the heaviest HBM-streaming
    kernel of the path, job_scan_kernel<true> (K0 + K1a: one read of RGB + every class label map, one byte flag
    per pixel): `iters` launches on cleared flags, as in a real frame."""
    import torch
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    labels = [c.labels for c in specs]
    job_base = np.concatenate([[0], np.cumsum([c.n_seg for c in specs])])[:-1]
    n_jobs = sum(c.n_seg for c in specs)
    n_before = len(rh.scan_events)
    for it in range(iters):
        bitmaps, stats = rh.new_job_state(n_jobs)
        rh.job_scan(rgb, labels, job_base, bitmaps, stats, black_is_colour=False)
    torch.cuda.synchronize()
    times = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in rh.scan_events]
    t = float(np.mean(times[n_before:]))            # back-to-back launches: bracket == kernel time (rocprofv3 agrees to ~1 %)
    algo_bytes = H * W * (3 + 4 * len(specs))       # RGB + one int32 label per class, read once
    # HBM traffic per launch from the PMC passes kept under profiles/ (separate FETCH_SIZE / WRITE_SIZE runs of
    # this command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)
    traffic, source = None, None
    for name in ("r04_pmc_fetch_write_kb.json", "r03_pmc_fetch_write_kb.json", "r02_pmc_fetch_write_kb.json", "r01_pmc_fetch_write_kb.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc) and H * W == 3840 * 2160:
            rec = json.load(open(pmc)).get("void rhccq::job_scan_kernel<true>")
            if rec:
                traffic = (2 * rec["FETCH_SIZE"]["mean"] + rec["WRITE_SIZE"]["mean"]) * 1024
                source = f"profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (NOT measured in this run)"
                break
    return {"bound": "hbm", "kernel": "job_scan_kernel<true>", "achieved": algo_bytes / t / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": algo_bytes / t / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_measured_in_this_run": False,
            "traffic_source": source, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_s": t, "launches": iters,
            "share_of_step_gpu_time": "< 0.1 %: the per-pixel passes (scan, index, remap) are ~0.9 ms of the step"}


def chain_probe(rh, enc, rgb, specs, ms_per_step):
    """The kernel that dominates the step: the sequential k-means++ chain of the level-1 MiniBatchKMeans problems (one
    workgroup per segment palette), HIP events around its launch.  Neither HBM nor MFMA bounds it: each pick depends on the
    previous one, and one pick is a chain of LDS / L2 round trips and ~10^3 instructions per wave on ONE CU.  Reported as the
    brute-force-equivalent rate: sum over picks of T candidates x init_size samples pair evaluations (8 integer ops each,
    SURVEY.md 8d) against the chip's vector peak -- the kernel prunes, so what it really issues is lower still."""
    import math
    S = enc.prepare(rgb, specs)
    jobs, _ = enc.level1_jobs(S)
    parts, ks = [], []
    for jb in jobs:
        if "keys_dev" not in jb:
            continue
        nbk = jb["keys_dev"][1:] if jb["has_black"] else jb["keys_dev"]
        if int(nbk.numel()) >= 10000:
            parts.append(nbk)
            ks.append(math.ceil(int(nbk.numel()) * (jb["quality"] / 100) / 10))
    if not parts:
        return None
    t = {}
    rh.minibatch_kmeans(parts, ks, return_device=True, timing=t)
    pairs = sum((k - 1) * (2 + int(math.log(k))) * max(3 * k, 3000) for k in ks)
    ops = 8.0 * pairs
    sec = t["init_ms"] * 1e-3
    gen3 = max(max(3 * k, 3000) for k in ks) <= 98304
    return {"bound": "valu-issue on one CU per problem (neither hbm nor mfma: a sequential chain of dependent picks)",
            "kernel": "mbk_init3_kernel" if gen3 else "mbk_init2_kernel", "achieved": ops / sec / 1e12, "peak": VALU_PEAK_TOPS,
            "unit": "Tops/s (brute-force equivalent)", "frac": ops / sec / 1e12 / VALU_PEAK_TOPS, "traffic": None,
            "launch_ms": t["init_ms"], "problems": len(ks), "picks_longest_chain": max(ks),
            "us_per_pick": t["init_ms"] * 1e3 / max(ks), "share_of_step": t["init_ms"] / ms_per_step,
            "brute_force_pair_evaluations": pairs, "cus_occupied": len(ks), "cus": N_CUS,
            "note": "one workgroup per problem; the chain cannot leave its CU: the per-pick time (candidate search + box descent -> evaluate -> "
                    "commit, 3 barriers) sets the step; a pick is bound by the instructions the four waves of a SIMD issue together "
                    "(~8 500 wave-instructions over 4 SIMDs x 4 cycles), DESIGN.md section 8"}


def pixel_probe(rh, rgb, iters=10):
    """EXTENSION (no reference counterpart; BASELINE.json's north_star names it): the pixel-space fixed-radius
    neighbour pass over (x, y, L, a, b) -- 3 B read + 4 B written per pixel -- and the union-find expansion,
    HIP events on the launch stream."""
    import ctypes as C
    import torch
    H, W = int(rgb.shape[0]), int(rgb.shape[1])
    radius, eps, ws, min_pts = 2, 6.0, 1.0, 5
    rh.px_dbscan(rgb, radius, eps, ws, min_pts)                      # warm-up (+ the linearisation table upload)
    parent = rh.empty((H, W), torch.int32)
    labels = rh.empty((H, W), torch.int32)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_n = t_e = 0.0
    for _ in range(iters):
        ev[0].record()
        rh._check(rh.lib.rhccq_px_neighbours(rh.ctx, rh._p(rgb), H, W, radius, eps, ws, min_pts, rh._p(rh._px_lut), rh._p(parent), C.c_void_p(0)),
                  "px_neighbours")
        ev[1].record()
        rh._check(rh.lib.rhccq_px_expand(rh.ctx, rh._p(rgb), H, W, radius, eps, ws, rh._p(rh._px_lut), rh._p(parent), rh._p(labels)), "px_expand")
        ev[2].record()
        torch.cuda.synchronize()
        t_n += ev[0].elapsed_time(ev[1]) * 1e-3
        t_e += ev[1].elapsed_time(ev[2]) * 1e-3
    t_n, t_e = t_n / iters, t_e / iters
    algo = H * W * 7
    return {"extension": True, "kernel": "px_neighbours_kernel", "params": {"radius": radius, "eps": eps, "spatial_weight": ws, "min_pts": min_pts},
            "algorithmic_bytes_per_launch": algo, "avg_launch_s": t_n, "achieved_GB_s": algo / t_n / 1e9, "frac_of_hbm_peak": algo / t_n / 1e9 / HBM_PEAK_GBS,
            "expansion_s": t_e, "clusters": int(torch.unique(labels).numel()) - 1, "Mpixels_per_s_neighbours_plus_expansion": H * W / (t_n + t_e) / 1e6}


def roi_stage_probe(rh, img):
    """UPSTREAM of the timed path (parity unpinned): the reference's ROI stage on the same frame -- get_regions (21 Canny
    passes, the morphological clean-up chain) + extract_regions -- through the mirrored API: numpy frame in host memory in, numpy masks
    and region dicts out; connected-component labelling alone by HIP events (25 B/px algorithmic)."""
    import torch
    from roibasedimagecompression_amd.api import roi as R, roi_chain as C
    H, W = img.shape[:2]
    C.get_regions(img)                                                 # warm-up
    t0 = time.perf_counter()
    out = C.get_regions(img)
    t1 = time.perf_counter()
    roi, non = R.extract_regions(img, out[4], out[5])
    t2 = time.perf_counter()
    mask = torch.from_numpy(np.ascontiguousarray(out[0] != 0)).to(rh.device)
    rh.ccl(mask, 8)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        n, _, _ = rh.ccl(mask, 8, host_stats=False)
    ev[1].record()
    torch.cuda.synchronize()
    t_ccl = ev[0].elapsed_time(ev[1]) * 1e-3 / 5
    return {"upstream_stage": True, "parity": "unpinned (OpenCV restated)", "get_regions_s": t1 - t0, "extract_regions_s": t2 - t1,
            "Mpixels_per_s": H * W / (t2 - t0) / 1e6, "roi_fraction": float(out[1].mean()), "regions": [len(roi), len(non)],
            "ccl": {"kernel": "ccl_* (csrc/ccl.hip), region map of the frame", "components": n, "avg_call_s": t_ccl,
                    "algorithmic_bytes_per_call": 25 * H * W, "achieved_GB_s": 25 * H * W / t_ccl / 1e9,
                    "frac_of_hbm_peak": 25 * H * W / t_ccl / 1e9 / HBM_PEAK_GBS}}


def upstream_probe(H, W, q_roi, q_bg, sigma, reps=2):
    """The whole encoder of the reference's script (encoder/compression/test.py:77-151) on a 4K frame, upstream stages INSIDE the
    clock: get_regions (21 Canny passes + clean-up chain) -> extract_regions -> per region split score + masked SLIC -> the three
    clustering levels -> final palette + index map; no container (its zlib level 9 is 1.3 s of host time on one core and has no
    GPU counterpart).  Through the mirrored Python API: the frame starts as a numpy array in host memory (the upload, ~1 ms, is
    inside).  The ROI / SLIC stages restate OpenCV / scikit-image: parity unpinned.  The frame is the bench generator's photo with a
    darker, flatter surround (the plain generator frame is noise everywhere: one ROI region, no background class)."""
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.flow import script_flow
    img = synth.photo(H, W, 1234, sigma=sigma)
    yy, xx = np.mgrid[0:H, 0:W]
    img[((yy - H / 2) / (H / 2)) ** 2 + ((xx - W / 2) / (W / 2)) ** 2 > 0.5] //= 3
    script_flow(img, q_roi, q_bg, container=False)                    # warm-up
    best, best_info, colours = None, None, 0
    for _ in range(reps):
        t0 = time.perf_counter()
        final, _, info = script_flow(img, q_roi, q_bg, container=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, best_info, colours = dt, info, len(final["palette"])
    return {"value_with_upstream": H * W / best / 1e6, "unit": "Mpixels/s", "seconds": best, "parity": "unpinned upstream (OpenCV / scikit-image restated); "
            "the three clustering levels inside are the parity path", "frame": f"{W}x{H} synthetic photo (seed 1234, sigma={sigma}) with a darker surround outside an ellipse",
            "tiers": [q_roi, q_bg], "container": "not included (host zlib level 9)", "final_colours": int(colours),
            "stages_ms": {k: round(v * 1e3, 1) for k, v in best_info["seconds"].items()},
            "roi_fraction": best_info["region_map_roi_fraction"], "regions": [best_info["roi_regions"], best_info["nonroi_regions"]],
            "segments": [best_info["roi_segments"], best_info["nonroi_segments"]]}


def neighbour_probe(rh):
    """K3/K4 eps-components microbench of SURVEY.md 8d (`palette-only`): 256 palettes x 4000 colours."""
    rng = np.random.default_rng(99)
    keys = [np.unique(rng.integers(0, 1 << 24, 4000).astype(np.uint32)) for _ in range(256)]
    out = {}
    for eps in (12.8, 51.2, 102.4):
        t0 = time.perf_counter()
        rh.eps_components(keys, [eps] * len(keys))
        t = time.perf_counter() - t0
        n = sum(len(k) for k in keys)
        out[str(eps)] = {"palettes": len(keys), "points": n, "wall_s_incl_h2d": t, "GB_s_7B_per_point": 7 * n / t / 1e9,
                         "pair_tests_per_s": sum(len(k) * (len(k) - 1) / 2 for k in keys) / t}
    return out


def host_cores():
    """cores this process may really use: the affinity mask, cut by the cgroup CPU quota (a GPU box hands a 1-GPU job a share
    of a large host: OpenMP threads beyond the quota only spin against each other)"""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return max(1, min(n, 32))


def cpu_baseline(img, lab_roi, lab_non, size, qs):
    """The CPU restatement (kind 'port': oracle/, numpy + the native C pieces oracle/mbk_oracle.c, km64_estep.c) on a bounded
    centre crop of the same frame and label maps, same stage boundaries (unique -> 3 levels -> final palette + indices), timed
    with one thread and with all host cores (OpenMP over the MiniBatchKMeans / KMeans E-steps and the k-means++ candidates;
    the numpy bookkeeping between them stays single-threaded)."""
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # before libgomp loads: idle threads sleep instead of spinning
    from oracle import rhccq_oracle as O
    H, W = img.shape[:2]
    r0, c0 = (H - size) // 2, (W - size) // 2
    crop = img[r0:r0 + size, c0:c0 + size]
    classes, q_used = [], []
    for lab, q in zip((lab_roi, lab_non), qs):
        l = lab[r0:r0 + size, c0:c0 + size]
        m = l > 0
        if not m.any():
            continue
        _, dense = np.unique(l[m], return_inverse=True)
        l2 = np.zeros_like(l)
        l2[m] = dense + 1
        rows, cols = np.where(m)
        bbox = (int(rows.min()), int(cols.min()), int(rows.max()) + 1, int(cols.max()) + 1)
        sl = (slice(bbox[0], bbox[2]), slice(bbox[1], bbox[3]))
        classes.append([{"bbox": bbox, "bbox_mask": m[sl], "seglabels": l2[sl].astype(np.int32)}])
        q_used.append(q)
    nproc = host_cores()
    model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
    legs = {}
    for threads in (nproc, 1):
        t0 = time.perf_counter()
        O.encode_frame(crop, classes, q_used, minibatch=lambda pts, k, th=threads: O.minibatch_kmeans_native(pts, k, threads=th)[0])
        legs[threads] = time.perf_counter() - t0
        if nproc == 1:
            break
    dt = legs[nproc]
    # a sample whose k matches the frame's: level 1 of the frame's LARGEST segment at full size (unique colours -> MiniBatchKMeans with
    # the frame's own k ~ 20-30 k -> member means); the crop above has k ~ 5 k per segment and the chain cost grows ~k^2, so its
    # per-pixel rate flatters the CPU
    labs = (lab_roi, lab_non)
    sizes = [(int((l == s).sum()), ci, int(s)) for ci, l in enumerate(labs) for s in np.unique(l[l > 0])]
    npx, ci, sid = max(sizes)
    m = labs[ci] == sid
    rows, cols = np.where(m)
    sl = (slice(int(rows.min()), int(rows.max()) + 1), slice(int(cols.min()), int(cols.max()) + 1))
    seg = img[sl].copy()
    seg[~m[sl]] = 0
    t0 = time.perf_counter()
    pal, idx = O.unique_colors(seg)
    eps, _, mc = O.clustering_params(len(pal), qs[ci])
    k_seg = int(np.ceil((len(pal) - int((pal == 0).all(axis=1).any())) * (qs[ci] / 100) / 10))
    O.cluster_palette(qs[ci], pal, idx, eps, mc, minibatch=lambda pts, k, th=nproc: O.minibatch_kmeans_native(pts, k, threads=th)[0])
    dt_seg = time.perf_counter() - t0
    return {"value": npx / dt_seg / 1e6, "unit": "Mpixels/s", "cores": nproc, "kind": "port", "cpu_model": model, "nproc": nproc,
            "sample": f"level 1 (unique colours -> MiniBatchKMeans, k = {k_seg} -> member means) of the frame's largest segment at FULL size: {npx} px, "
                      f"{len(pal)} colours, native C (OpenMP) + numpy restatement, {dt_seg:.1f} s; levels 2-3 (k of a few 10^3) are not in this leg",
            "k": k_seg, "seconds": round(dt_seg, 1),
            "crop_three_levels": {"value": size * size / dt / 1e6, "unit": "Mpixels/s", "cores": nproc, "seconds": round(dt, 1),
                                  "sample": f"centre {size}x{size} crop of the same frame and label maps through all three levels (k ~ 5 k per segment: flatters the CPU per pixel)"},
            "one_thread": {"value": size * size / legs[1] / 1e6, "unit": "Mpixels/s", "cores": 1, "seconds": round(legs[1], 1), "sample": "the crop leg"},
            "reference_cpu_of_record": {"value": 0.046, "unit": "Mpixels/s", "cores": 8,
                                        "what": "the reference itself (sklearn 1.7.2) in the build container, whole Lenna 512x512 as one segment, "
                                                "N = 148 279 colours, k = 2 966 (BASELINE.md section 3); 0.0060 Mpixels/s for its 64-segment notebook path"}}


def carried_cpu_baseline():
    for name in ("r04_bench4k.json", "r03_bench4k.json", "r02_bench4k.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                cb = json.load(open(path)).get("cpu_baseline")
            except Exception:
                cb = None
            if cb:
                cb = dict(cb)
                cb["carried_from"] = f"profiles/{name}: the N = 1 run of configs[1] on an MI355X box (the CPU leg is not re-timed at N > 1)"
                return cb
    return None


# ---- main -----------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)                                  # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("RHCCQ_BENCH_REHEARSAL") == "1"     # several ranks on ONE GPU over gloo: exercises the
    if world != args.gpus and not rehearsal:                       # launch path on a one-GPU box, never a bench line
        sys.stderr.write(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}\n")
        sys.exit(2)
    import roibasedimagecompression_amd  # noqa: F401  (sets GPU_MAX_HW_QUEUES before the HIP runtime starts)
    import torch
    import torch.distributed as dist
    if rehearsal:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not rehearsal and torch.cuda.device_count() < world:
            sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} device(s)\n")
            sys.exit(2)
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from roibasedimagecompression_amd.frame import FrameEncoder
    from roibasedimagecompression_amd.ops import Rhccq
    Rhccq.scan_events = []          # class-wide: the lanes of the stream regime own their own contexts
    rh = Rhccq(local)
    mode = args.mode
    batch_mode = mode == "batch"                          # configs[2]: 64 1080p frames per step, tiers (20, 10): two sub-batches of 32 in flight
    if batch_mode:                                        # (383 Mpx/s against 338-352 for one encode_batch call over all 64: same box, round 4)
        mode = "stream"
    H = args.height or (4320 if mode == "tiled" else (1080 if batch_mode else 2160))
    W = args.width or (7680 if mode == "tiled" else (1920 if batch_mode else 3840))
    block = args.block or (16 if (mode == "stream" and not batch_mode) else 8)
    q_roi = args.quality
    q_bg = args.quality_bg if args.quality_bg is not None else (args.quality if (mode == "frame" and not batch_mode) else 10)
    # stream regime, round 4 (no straggling problems any more: smaller batches, more of them in flight): 910 Mpx/s at 5 lanes x 24 frames,
    # 754 at 4 x 32, 806 at 8 x 16, 883 at 4 x 48, 1 011 at 12 x 8, 1 015 at 8 x 12, 1 031 at 10 x 10, 1 014-1 113 at 10 x 12 (same box pairs)
    B = args.frames_per_step or (32 if batch_mode else (12 if mode == "stream" else 1))
    L = args.lanes or (2 if batch_mode else (10 if mode == "stream" else 1))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stages, out, enc = {}, None, None
    if mode == "tiled":
        # configs[3]: one frame, one tile per rank; rows x cols as square as the rank count allows (2 x 4 at 8 ranks)
        from roibasedimagecompression_amd.parallel import TiledFrameEncoder, tile_grid
        rows = max(r for r in (1, 2, 3, 4) if world % r == 0 and r * r <= world)
        tiles = tile_grid(H, W, rows, world // rows)
        img, rgb, specs, roi_mask, (lr, ln, br, bn) = build_inputs(rh, H, W, 1234, (2, 1), q_roi, q_bg, args.sigma, tile=tiles[rank])
        enc = TiledFrameEncoder(rh, (H, W), tiles[rank]).set_tiles(tiles) if world > 1 else FrameEncoder(rh)

        def step():
            o = enc.encode(rgb, specs)
            dct_ext(rh, rgb, roi_mask, block)
            return o
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        px = H * W * args.steps
        workload = (f"configs[3]: single {W}x{H} RGB synthetic 'photo' frame (seed 1234, sigma={args.sigma}) cut into {rows}x{world // rows} tiles, one per GPU; "
                    f"quality tiers ({q_roi},{q_bg}), 2 segments per class; exchange = 1 all-gather (segment bitmaps + stats) + MIN all-reduces of "
                    f"palette-sized first-position tables over {'RCCL' if not rehearsal else 'gloo'}; {block}x{block} DCT extension per tile")
        scaling, par = "strong", f"tile-parallel {rows}x{world // rows}"
    elif mode == "stream":
        # configs[4]: a stream of 4K frames per GPU, batches of B frames, L batches in flight
        from roibasedimagecompression_amd.stream import StreamEncoder
        frames, masks = [], []
        D = B * L if batch_mode else B                      # distinct frames: configs[2] = ONE batch of B * L different frames per step
        for i in range(D):
            img, r_i, sp_i, m_i, (lr, ln, br, bn) = build_inputs(rh, H, W, 1234 + rank * D + i, (2, 1), q_roi, q_bg, args.sigma)
            frames.append((r_i, sp_i))
            masks.append(m_i)
        rgb, specs = frames[0]
        se = StreamEncoder(local, batch=B, lanes=L)
        reps = B * L // D                                   # passes over the distinct frames per step

        def run(n_steps):
            outs = se.run(frames * (reps * n_steps))
            for _ in range(reps * n_steps):
                for (r, _), m in zip(frames, masks):
                    dct_ext(rh, r, m, block)
            return outs
        run(max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        out = run(args.steps)[0]
        barrier()
        dt = time.perf_counter() - t0
        px = H * W * args.steps * world * B * L
        workload = ((f"configs[2]: batch of {B * L} {W}x{H} RGB synthetic 'photo' frames per GPU per step ({D} distinct frames, seeds 1234+rank*{D}+i, " if batch_mode else
                     f"configs[4]: stream of {B * L * args.steps} {W}x{H} RGB synthetic 'photo' frames per GPU ({B} distinct frames, seeds 1234+rank*{B}+i, ")
                    + f"sigma={args.sigma}) through StreamEncoder: batches of {B}, {L} in flight; quality tiers ({q_roi},{q_bg}), three-level palette "
                    f"hierarchy, {block}x{block} DCT extension")
        scaling, par = "weak", f"frame-parallel x{world}, {L} lanes x batch {B}"
    else:
        enc = FrameEncoder(rh)
        img, rgb, specs, roi_mask, (lr, ln, br, bn) = build_inputs(rh, H, W, 1234 + rank * B, (2, 1), q_roi, q_bg, args.sigma)
        extra = []
        for i in range(1, B):
            _, r_i, sp_i, m_i, _ = build_inputs(rh, H, W, 1234 + rank * B + i, (2, 1), q_roi, q_bg, args.sigma)
            extra.append((r_i, sp_i, m_i))
        native = args.host == "native" and B == 1
        for _ in range(args.warmup):
            one_step(rh, enc, rgb, specs, roi_mask, block, extra=extra, native=native)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_step(rh, enc, rgb, specs, roi_mask, block, extra=extra, native=native)
        barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            one_step(rh, enc, rgb, specs, roi_mask, block, stage_acc=stages, extra=extra, native=native)
        other_host = None
        if rank == 0 and B == 1 and world == 1 and not args.no_probes:
            # the other host side on the same frame (same kernels, same result): how much of the step is the host language
            for _ in range(2):
                one_step(rh, enc, rgb, specs, roi_mask, block, native=not native)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step(rh, enc, rgb, specs, roi_mask, block, native=not native)
            torch.cuda.synchronize()
            other_host = {"host": "python (FrameEncoder.encode)" if native else "native (rhccq_encode_frame)",
                          "ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3}
        px = H * W * args.steps * world * B
        workload = ((f"configs[1]: single {W}x{H} RGB synthetic 'photo' frame per GPU (seed 1234+rank, sigma={args.sigma}), " if B == 1 else
                     f"batch of {B} {W}x{H} RGB synthetic 'photo' frames per GPU per step (seeds 1234+rank*{B}+i, sigma={args.sigma}), ")
                    + (f"one quality tier q={q_roi} (levels {q_roi}/{min(2 * q_roi, 100)}/{min(4 * q_roi, 100)}), " if q_roi == q_bg else f"quality tiers ({q_roi},{q_bg}), ")
                    + f"2 segments per class, ROI ellipse 35 % + 3 px overlap; {block}x{block} DCT + two-tier quantisation extension in the timed region")
        scaling, par = "weak", f"frame-parallel x{world}"
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=rh.device)
        if rehearsal:
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    line = {
        "metric": ("Mpixels/s encoded (ROI cluster + DCT/quant) at 4K RGB" if not batch_mode else "Mpixels/s encoded (ROI cluster + DCT/quant), batches of 1080p RGB frames") if mode != "tiled" else "Mpixels/s encoded (ROI cluster + DCT/quant), one 8K RGB frame tiled over the GPUs",
        "value": px / dt / 1e6, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8 keys / int32 exact k-means++ / f64 Lloyd+mini-batch", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearsal else ""),
        "config": {"workload": workload, "mode": args.mode, "host": ("native C++ (rhccq_encode_frame)" if (mode == "frame" and B == 1 and args.host == "native") else "python"), "frames_per_step_per_gpu": B * L if mode == "stream" else B, "lanes": L, "parallelism": par},
    }
    if rank == 0:
        if stages:
            line["stages_ms"] = {k: round(v * 1e3, 3) for k, v in stages.items()}
            if getattr(enc, "class_timings", None):        # the two class pipelines run side by side: their own stage clocks
                line["stages_ms"]["per_class"] = {str(ci): {k: round(v * 1e3, 3) for k, v in tm.items()} for ci, tm in sorted(enc.class_timings.items())}
        if mode == "frame" and B == 1 and other_host is not None:
            line["other_host"] = other_host
        line["final_colours"] = int(len(out["palette"]))
        line["unique_colours_per_segment"] = [int(v) for v in out["n_unique"]]
        probes = not args.no_probes and world == 1 and mode == "frame" and B == 1
        roof = hbm_probe(rh, rgb, specs)                  # per rank: this rank's frame (its tile in tiled mode)
        dom = chain_probe(rh, enc, rgb, specs, dt / args.steps * 1e3) if probes else None
        # `roofline` = the kernel the step is made of (flat scalars: the driver's parse keeps scalars only); the heaviest
        # HBM-streaming pass of the path -- < 0.1 % of the step -- sits beside it as `roofline_hbm`
        line["roofline"] = dom if dom is not None else roof
        line["roofline_hbm"] = roof
        if probes:
            up = upstream_probe(H, W, q_roi, 10 if q_bg == q_roi else q_bg, args.sigma)
            line["value_with_upstream"] = up["value_with_upstream"]
            line["with_upstream"] = up
            line["neighbour_pass"] = neighbour_probe(rh)
            line["pixel_neighbour_pass_extension"] = pixel_probe(rh, rgb)
            line["roi_stage_upstream"] = roi_stage_probe(rh, img)
        if args.cpu_sample and world == 1:
            line["cpu_baseline"] = cpu_baseline(img, lr, ln, min(args.cpu_sample, H, W), (q_roi, q_bg))
        else:
            # N > 1: the CPU leg is timed at N = 1 only (one bounded sample per box): nothing measured in THIS run goes under the
            # key; the N = 1 figure of record rides along under a key of its own
            line["cpu_baseline"] = None
            line["cpu_baseline_carried"] = carried_cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
