#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the RHCCQ encoder hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one synthetic 3840x2160 RGB frame resident in HBM
(BASELINE.json configs[1]): per-segment unique colours -> three-level palette clustering (level 1 /
level 2 / level 3 of rhccq.ipynb:978-1039) -> final palette + index map, followed by the
DCT/quantisation EXTENSION tile (8x8 blocks, two-tier ROI/background steps) that BASELINE.json's
metric names but the reference does not contain; the extension is inside the timed
region so that no named work is skipped, and its share is reported in `stages_ms`.

Multi-GPU: frames are independent (SURVEY.md 8e "frame-parallel"): every rank encodes its own frame,
no data-path collective; value = pixels of all ranks / max-over-ranks time ("weak" scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_inputs(rh, H, W, seed, tiles, q_roi, q_non, sigma):
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec
    img = synth.photo(H, W, seed, sigma=sigma)
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
    rgb = torch.from_numpy(img).to(rh.device)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], q_roi),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], q_non)]
    roi_mask = torch.from_numpy((lr > 0).astype(np.uint8)).to(rh.device)
    return img, rgb, specs, roi_mask, (lr, ln, br, bn)


def one_step(rh, enc, rgb, specs, roi_mask, block, stage_acc=None, extra=()):
    """one pass of the hot path over one batch: the frame (plus `extra` further frames when the step is a
    batch, --frames-per-step) and the DCT/quantisation extension of every frame"""
    import torch
    if extra:
        outs = enc.encode_batch([(rgb, specs)] + [(r, sp) for r, sp, _ in extra])
        out = outs[0]
    else:
        out = enc.encode(rgb, specs)
    t0 = time.perf_counter()
    luma, qstep = rh.luma_qstep(rgb, roi_mask, block, 4.0, 16.0)
    coef, q = rh.dct_quant(luma, block, qstep, want_coef=False)
    for r, sp, m in extra:
        l2, q2 = rh.luma_qstep(r, m, block, 4.0, 16.0)
        rh.dct_quant(l2, block, q2, want_coef=False)
    if stage_acc is not None:
        torch.cuda.synchronize()
        for k, v in enc.timings.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        stage_acc["dct_ext"] = stage_acc.get("dct_ext", 0.0) + (time.perf_counter() - t0)
    return out, q


def roofline_probe(rh, rgb, specs, iters=5):
    """HIP-event timing (events recorded on the stream the kernel is launched on) of the heaviest HBM-streaming
    kernel of the path, job_scan_kernel<true> (K0 + K1a: one read of RGB + every class label map, one byte flag
    per pixel): `iters` launches on cleared flags, as in a real frame.  `rh.scan_events` also brackets every
    in-frame launch of the process; those brackets are reported separately because an event recorded right
    behind the 64 MiB flag memset adds its own packet latency (~100 us) to the kernel time.  `rocprofv3
    --kernel-trace --stats` of the default command (profiles/) averages all launches: 143.7 us over 12."""
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
    t_all = float(np.mean(times))                   # brackets of in-frame launches also hold the event packets' own latency
    t = float(np.mean(times[n_before:]))            # back-to-back launches: bracket == kernel time (rocprofv3 agrees to ~1 %)
    px = H * W
    algo_bytes = px * (3 + 4 * len(specs))          # RGB + one int32 label per class, read once
    # HBM traffic per launch from the PMC passes kept under profiles/ (separate FETCH_SIZE / WRITE_SIZE runs of
    # this command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_kb.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc)).get("void rhccq::job_scan_kernel<true>")
        if rec and H * W == 3840 * 2160:
            traffic = (2 * rec["FETCH_SIZE"]["mean"] + rec["WRITE_SIZE"]["mean"]) * 1024
    return {"bound": "hbm", "kernel": "job_scan_kernel<true>", "achieved": algo_bytes / t / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": algo_bytes / t / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_s": t, "launches": iters,
            "avg_event_bracket_s_all_launches": t_all, "all_launches": len(times),
            "note": "per-pixel passes are < 0.3 % of the step; the step is bound by the sequential k-means++ chain "
                    "(dominant_kernel); HIP-event brackets include the dispatch latency onto an idle stream (~15-25 us "
                    "above the kernel time rocprofv3 reports)"}


def chain_probe(rh, enc, rgb, specs, ms_per_step):
    """The kernel that dominates the step: mbk_init_kernel, the sequential k-means++ chain of the level-1
    MiniBatchKMeans problems (one workgroup per segment palette).  HIP events around its launch."""
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
    return {"kernel": "mbk_init_kernel", "launch_ms": t["init_ms"], "problems": len(ks), "picks_longest_chain": max(ks),
            "us_per_pick": t["init_ms"] * 1e3 / max(ks), "share_of_step": t["init_ms"] / ms_per_step,
            "bound": "one workgroup per problem; each pick depends on the previous one; the per-pick time is the "
                     "instruction issue of one CU over ~1000 instructions per wave (DESIGN.md section 3)"}


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
        out[str(eps)] = {"palettes": len(keys), "points": n, "wall_s_incl_h2d": t, "GB_s_7B_per_point": 7 * n / t / 1e9}
    return out


def cpu_baseline(img, lab_roi, lab_non, size, q):
    """The oracle (kind 'port', numpy, 1 thread of the host) on a bounded crop of the same frame and
    label maps: same stage boundaries (unique -> 3 levels -> final palette + indices)."""
    from oracle import rhccq_oracle as O
    H, W = img.shape[:2]
    r0, c0 = (H - size) // 2, (W - size) // 2
    crop = img[r0:r0 + size, c0:c0 + size]
    classes = []
    for lab in (lab_roi, lab_non):
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
    t0 = time.perf_counter()
    O.encode_frame(crop, classes, [q] * len(classes))
    dt = time.perf_counter() - t0
    return {"value": size * size / dt / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"centre {size}x{size} crop of the same synthetic frame and label maps, numpy oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--quality", type=int, default=20, help="one ROI quality tier (configs[1])")
    ap.add_argument("--sigma", type=float, default=2.0, help="sensor-noise sigma of the synthetic photo")
    ap.add_argument("--block", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=576, help="edge of the CPU-baseline crop (0 = skip)")
    ap.add_argument("--no-probes", action="store_true")
    ap.add_argument("--frames-per-step", type=int, default=1,
                    help="frames encoded together per step (default 1 = BASELINE configs[1], a single frame; > 1 is the "
                         "stream / batch regime of configs[2] and [4]: one batched clustering launch per level)")
    ap.add_argument("--stream-probe", action="store_true",
                    help="also time 48 frames in the stream regime (3 lanes x batches of 8) and report it as `stream_regime`; "
                         "off by default so that every job_scan launch of the default command belongs to the single-frame workload "
                         "(the population the roofline object and the committed rocprof summary average)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="stream regime: batches of --frames-per-step frames in flight on this many host threads, each with "
                         "its own HIP stream (stream.StreamEncoder); a step is then lanes x frames-per-step frames")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("RHCCQ_BENCH_REHEARSAL") == "1"     # several ranks on ONE GPU over gloo: exercises the
    if rehearsal:                                                   # launch path on a one-GPU box, never a bench line
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from roibasedimagecompression_amd.ops import Rhccq
    from roibasedimagecompression_amd.frame import FrameEncoder
    Rhccq.scan_events = []          # class-wide: the lanes of the stream regime own their own contexts
    rh = Rhccq(local)
    enc = FrameEncoder(rh)
    H, W = args.height, args.width
    # at >= 4K the reference's SLIC scaling yields <= 2 segments per region (SURVEY.md 8a preface)
    B = max(1, args.frames_per_step)
    img, rgb, specs, roi_mask, (lr, ln, br, bn) = build_inputs(rh, H, W, 1234 + rank * B, (2, 1), args.quality, args.quality, args.sigma)
    extra = []
    for i in range(1, B):
        _, r_i, sp_i, m_i, _ = build_inputs(rh, H, W, 1234 + rank * B + i, (2, 1), args.quality, args.quality, args.sigma)
        extra.append((r_i, sp_i, m_i))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    L = max(1, args.lanes)
    if L > 1:
        # stream regime with several batches in flight: the timed region is ONE run over steps x lanes batches
        from roibasedimagecompression_amd.stream import StreamEncoder
        from roibasedimagecompression_amd.frame import ClassSpec  # noqa: F401
        batch_frames = [(rgb, specs)] + [(r, sp) for r, sp, _ in extra]
        masks = [roi_mask] + [m for _, _, m in extra]
        se = StreamEncoder(local, batch=B, lanes=L)

        def run(n_steps):
            outs = se.run(batch_frames * (L * n_steps))
            for _ in range(L * n_steps):                        # the DCT extension of every frame, as in one_step
                for (r, _), m in zip(batch_frames, masks):
                    l2, q2 = rh.luma_qstep(r, m, args.block, 4.0, 16.0)
                    rh.dct_quant(l2, args.block, q2, want_coef=False)
            return outs
        run(max(args.warmup, 1))
        barrier()
        t0 = time.perf_counter()
        out = run(args.steps)[0]
        barrier()
        dt = time.perf_counter() - t0
        B = B * L
    else:
        for _ in range(args.warmup):
            one_step(rh, enc, rgb, specs, roi_mask, args.block, extra=extra)
        barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(args.steps):
            out, q = one_step(rh, enc, rgb, specs, roi_mask, args.block, extra=extra)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=rh.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stages = {}
    if rank == 0:
        one_step(rh, enc, rgb, specs, roi_mask, args.block, stage_acc=stages, extra=extra)
    px = H * W * args.steps * world * B
    line = {
        "metric": "Mpixels/s encoded (ROI cluster + DCT/quant) at 4K RGB",
        "value": px / dt / 1e6, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8 keys / int32 exact k-means++ / f64 Lloyd+mini-batch", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearsal else ""),
        "config": {"workload": (f"configs[1]: single {W}x{H} RGB synthetic 'photo' frame per GPU (seed 1234+rank, sigma={args.sigma}), " if B == 1 else
                                f"batch of {B} {W}x{H} RGB synthetic 'photo' frames per GPU per step (seeds 1234+rank*{B}+i, sigma={args.sigma}), ")
                               + f"one quality tier q={args.quality} (levels {args.quality}/{min(2*args.quality,100)}/{min(4*args.quality,100)}), "
                               "2 segments per class, ROI ellipse 35 % + 3 px overlap; "
                               f"{args.block}x{args.block} DCT + two-tier quantisation extension in the timed region",
                   "frames_per_step_per_gpu": B, "lanes": L, "parallelism": f"frame-parallel x{world}"},
    }
    if rank == 0:
        line["stages_ms"] = {k: round(v * 1e3, 3) for k, v in stages.items()}
        line["final_colours"] = int(len(out["palette"]))
        line["unique_colours_per_segment"] = [int(v) for v in out["n_unique"]]
        if not args.no_probes and world == 1:
            line["dominant_kernel"] = chain_probe(rh, enc, rgb, specs, dt / args.steps * 1e3 / B)
            line["neighbour_pass"] = neighbour_probe(rh)
            line["pixel_neighbour_pass_extension"] = pixel_probe(rh, rgb)
        if args.stream_probe and world == 1 and B == 1 and L == 1:
            # the same path in the stream regime (configs[4]: many 4K frames on one GPU): batches of 8 frames share one
            # batched clustering launch per level (the sequential k-means++ chains run side by side) and 3 batches are
            # in flight on 3 host threads / HIP streams, so one batch's host work hides behind another's GPU work
            from roibasedimagecompression_amd.stream import StreamEncoder
            nb, lanes, reps = 8, 3, 6
            more = [(rgb, specs)]
            for i in range(1, 2 * nb):
                _, r_i, sp_i, _, _ = build_inputs(rh, H, W, 1234 + i, (2, 1), args.quality, args.quality, args.sigma)
                more.append((r_i, sp_i))
            se = StreamEncoder(local, batch=nb, lanes=lanes)
            se.run(more + more[:nb])                                            # warm-up: one batch per lane
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            se.run(more * (reps // 2))
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            line["stream_regime"] = {"frames": reps * nb, "frames_per_batch": nb, "lanes": lanes, "value": reps * nb * H * W / d1 / 1e6,
                                     "unit": "Mpixels/s", "wall_s": d1,
                                     "note": "palette hierarchy only (no DCT extension); python bench.py --frames-per-step 16 --lanes 4 "
                                             "times this regime as the main value"}
            del more
        line["roofline"] = roofline_probe(rh, rgb, specs)
        if args.cpu_sample and world == 1:
            line["cpu_baseline"] = cpu_baseline(img, lr, ln, args.cpu_sample, args.quality)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
