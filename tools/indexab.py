"""diagnostic: job_index_entries (the first-position pass of one class, the last per-pixel pass that gathers) on the bench frame under two builds of
the library, HIP events on the launch stream:  python tools/indexab.py libA.so libB.so"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np, torch
    from roibasedimagecompression_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    import bench
    from roibasedimagecompression_amd.ops import Rhccq, INT_MAX
    from roibasedimagecompression_amd.frame import FrameEncoder
    rh = Rhccq(0)
    _, rgb, specs, _, _ = bench.build_inputs(rh, 2160, 3840, 1234, (2, 1), 20, 20, 2.0)
    enc = FrameEncoder(rh)
    S = enc.prepare(rgb, specs)
    total = int(S["total"])
    lut = torch.arange(total, dtype=torch.int32, device=rh.device)
    e1 = torch.empty((1, S["H"] * S["W"]), dtype=torch.int32, device=rh.device)
    times = []
    for ci in (0, 1):
        for rep in range(6):
            fp = torch.full((total,), INT_MAX, dtype=torch.int32, device=rh.device)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            rh.job_index_entries(S["rgb"], S["labels"][ci:ci + 1], S["job_base"][ci:ci + 1], S["bitmaps"], S["prefix"], S["d_pal_off"], S["fix_key"], fp, lut, e1)
            ev[1].record()
            torch.cuda.synchronize()
            if rep:
                times.append(ev[0].elapsed_time(ev[1]) * 1e3)
    import hashlib
    print(f"{np.mean(times):.1f} {hashlib.sha256(fp.cpu().numpy().tobytes()).hexdigest()[:10]}")
    sys.exit(0)
for l in sys.argv[1:3] * 2:
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
    print(l, out.stdout.strip() or out.stderr[-600:], "(us per class pass, sha of the first positions)", flush=True)
