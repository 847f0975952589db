import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from roibasedimagecompression_amd.ops import Rhccq
from roibasedimagecompression_amd.frame import FrameEncoder, ClassSpec
from roibasedimagecompression_amd import synth
rh = Rhccq(0)
for (H, W, tiles, qs) in ((512, 512, (8, 8), (20, 10)), (1080, 1920, (3, 4), (20, 10)), (768, 512, (7, 7), (20, 10))):
    img = synth.photo(H, W, 1234)
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, tiles)
    rgb = torch.from_numpy(img).to(rh.device)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], qs[0]),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], qs[1])]
    enc = FrameEncoder(rh)
    enc.encode(rgb, specs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = enc.encode(rgb, specs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{W}x{H} segs {nr}+{nn}: {dt*1e3:.1f} ms  {H*W/dt/1e6:.2f} Mpx/s  colours {len(out['palette'])}  uniq/seg max {int(out['n_unique'].max())}", {k: round(v*1e3,1) for k,v in enc.timings.items()})
    if "--profile" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        enc.encode(rgb, specs)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
