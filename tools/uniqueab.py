"""diagnostic: unique colours of the bench frame's 4 jobs by the bitmap path (scan + pack + count + emit) against the
sort path (rhccq_job_sort_unique), HIP-event times on the launch stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from roibasedimagecompression_amd.ops import Rhccq
rh = Rhccq(0)
_, rgb, specs, _, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
labels = [c.labels for c in specs]
job_base = np.concatenate([[0], np.cumsum([c.n_seg for c in specs])])[:-1]
n_jobs = sum(c.n_seg for c in specs)


def bitmap_path():
    bitmaps, stats = rh.new_job_state(n_jobs)
    rh.job_scan(rgb, labels, job_base, bitmaps, stats, black_is_colour=False)
    chunk, counts = rh.bitmap_count(bitmaps)
    P = counts.cpu().numpy().astype(np.int64)
    pal_off = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
    rh.bitmap_emit(bitmaps, chunk, rh.dev(pal_off[:-1].copy()), int(pal_off[-1]))
    return P


def sort_path():
    return rh.job_sort_unique(rgb, labels, job_base, n_jobs, None, np.zeros(0, np.int64))[0]


for name, fn in (("bitmap path (scan + pack + count + emit)", bitmap_path), ("sort path (rhccq_job_sort_unique)", sort_path)):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        P = fn()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{name}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms per frame (4 jobs, 8.3 Mpx x 2 classes); palette sizes {P.tolist()}")
