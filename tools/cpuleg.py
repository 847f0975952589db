"""diagnostic only: the cpu_baseline leg of bench.py alone (no GPU), with the core counts the box reports"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roibasedimagecompression_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "host_cores", bench.host_cores())
for p in ("/sys/fs/cgroup/cpu.max",):
    try:
        print(p, open(p).read().strip())
    except Exception as e:
        print(p, e)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H, W = 2160, 3840
img = synth.photo(H, W, 1234)
(lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (2, 1))
t = time.time()
print(bench.cpu_baseline(img, lr, ln, size, (20, 20)), "total s", time.time() - t)
