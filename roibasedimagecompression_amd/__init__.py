"""MI355X-native RHCCQ encoder hot path (palette hierarchy: unique colours -> eps-components /
k-means -> floor-mean palettes -> merge -> remap).  Host side in Python, kernels in csrc/ behind the C
ABI of include/rhccq.h.  The CPU oracle (oracle/) is NOT part of this package."""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); two streams that share a queue run their
# kernels one after the other.  The pipelined mini-batch lanes (ops.Rhccq._minibatch_lanes) and the StreamEncoder lanes
# each put a long single-workgroup k-means++ chain on their stream, so a collision costs a whole chain: ask for more
# queues.  Read by the HIP runtime when it initialises (first device call), so this must run before that -- import this
# package before touching the GPU; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments in device memory: the mini-batch step loop is three dependent launches per step, and fetching each kernel's
# arguments from host memory costs 7 % on a 4K frame (measured with the variable forced to 0; unset behaves like 1 on ROCm 7.2).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from ._lib import RhccqError, load as load_library  # noqa: E402,F401

__all__ = ["RhccqError", "load_library"]
