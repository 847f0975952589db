"""MI355X-native RHCCQ encoder hot path (palette hierarchy: unique colours -> eps-components /
k-means -> floor-mean palettes -> merge -> remap).  Host side in Python, kernels in csrc/ behind the C
ABI of include/rhccq.h.  The CPU oracle (oracle/) is NOT part of this package."""
from ._lib import RhccqError, load as load_library  # noqa: F401

__all__ = ["RhccqError", "load_library"]
