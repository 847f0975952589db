"""diagnostic: the k-means++ chain of the bench frame's longest problem (k = 30 128) under two builds of the library on ONE box, alternating
child processes (box-to-box clocks differ by ~10 %, so only a same-box A/B says anything):  python tools/chain_ab.py libA.so libB.so [reps]
(`opt:N` instead of a path: the library in the tree with RHCCQ_OPT_INIT_KERNEL = N)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(HERE))
    import math
    import numpy as np
    from roibasedimagecompression_amd import _lib
    opt = None
    if sys.argv[2].startswith("opt:"):                   # the library in the tree with RHCCQ_OPT_INIT_KERNEL set (3 = round 3's shared work list, 0 = default)
        opt = int(sys.argv[2][4:])
    else:
        _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    from roibasedimagecompression_amd.ops import Rhccq
    from roibasedimagecompression_amd import synth
    rh = Rhccq(0)
    if opt is not None:
        rh.set_option(rh.OPT_INIT_KERNEL, opt)
    img = synth.photo(2160, 3840, 1234)
    keys = (img[..., 0].astype(np.uint32) << 16) | (img[..., 1].astype(np.uint32) << 8) | img[..., 2]
    pal = np.unique(keys[:, 1920:])
    pal = pal[pal != 0][:1506367]
    k = math.ceil(len(pal) * 0.2 / 10)
    best = None
    for _ in range(3):
        t = {}
        labs, info = rh.minibatch_kmeans([pal], [k], return_info=True, timing=t)
        best = t["init_ms"] if best is None else min(best, t["init_ms"])
    import hashlib
    print(f"{best * 1e3 / k:.4f} {hashlib.sha256(info['chosen'].tobytes()).hexdigest()[:12]} {int(info['state'][0][5])}")
    sys.exit(0)
libs = [a for a in sys.argv[1:] if not a.isdigit()]          # two or more builds
reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
res = {l: [] for l in libs}
for r in range(reps):
    for l in libs:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
        if out.returncode:
            print(out.stderr[-2000:])
            sys.exit(1)
        us, sha, steps = out.stdout.split()
        res[l].append(float(us))
        print(l, us, "us/pick, picks", sha, "steps", steps, flush=True)
for l in libs:
    print(l, "best", min(res[l]), "mean", sum(res[l]) / len(res[l]))
