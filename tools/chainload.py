"""diagnostic only: a level-2-sized k-means++ chain (one workgroup on one CU, samples in LDS) alone on the chip and beside a stream that
keeps the other CUs busy (float64 matrix products): does the rest of the chip slow a kernel that shares neither CU nor memory traffic?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import threading, time
import numpy as np
import torch
from roibasedimagecompression_amd.ops import Rhccq

rh = Rhccq(0)
rng = np.random.default_rng(5)
n, k = 59273, 1186
P = np.unique((rng.normal(128, 45, (n * 2, 3))).clip(0, 255).astype(np.uint8), axis=0)[:n]
keys = (P[:, 0].astype(np.uint32) << 16) | (P[:, 1].astype(np.uint32) << 8) | P[:, 2]


def chain():
    t = {}
    rh.minibatch_kmeans([keys], [k], timing=t)
    return 1e3 * t["init_ms"] / k


print("alone:", [round(chain(), 2) for _ in range(3)], "us per pick", flush=True)
for name, make in (("float64 matmul 4096^3 on another stream", lambda: (torch.randn(4096, 4096, dtype=torch.float64, device="cuda"),) * 2),
                   ("float32 elementwise over 1 GB on another stream", lambda: (torch.randn(256 << 20, device="cuda"), None))):
    a, b = make()
    stop = [False]
    side = torch.cuda.Stream()

    def load():
        with torch.cuda.stream(side):
            while not stop[0]:
                for _ in range(4):
                    if b is not None:
                        torch.mm(a, b)
                    else:
                        a.mul_(1.0000001)
                side.synchronize()

    th = threading.Thread(target=load)
    th.start()
    time.sleep(0.3)
    print(f"beside {name}:", [round(chain(), 2) for _ in range(3)], "us per pick", flush=True)
    stop[0] = True
    th.join()
    del a, b
torch.cuda.synchronize()
print("alone again:", [round(chain(), 2) for _ in range(2)], "us per pick")
