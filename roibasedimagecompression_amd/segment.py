"""Segment dicts of the reference (clustering.py:80-95,407-429; merging.py:109-118) with a lazy index
list: the reference hands `indices` around as a flat Python list (8.3 M ints at 4K); here it is a
list-like view over a numpy array and/or a device tensor that only materialises when touched
(`len`, iteration, indexing, `.count`, `np.array(...)`, `.tolist()` all work as on a list)."""
from collections.abc import Sequence

import numpy as np


class IndexList(Sequence):
    __slots__ = ("_np", "_dev")

    def __init__(self, array=None, dev=None):
        self._np = None if array is None else np.asarray(array).reshape(-1)
        self._dev = dev                                  # flat device tensor (torch) or None

    # -- materialisation ---------------------------------------------------------------------------
    def numpy(self):
        if self._np is None:
            from .ops import default_context
            a = default_context(self._dev.device.index).to_host(self._dev).reshape(-1) if self._dev.numel() >= (1 << 16) else self._dev.cpu().numpy().reshape(-1)
            if a.dtype == np.int16:                      # uint16 storage of the device kernels
                a = a.view(np.uint16)
            self._np = a
        return self._np

    def device_tensor(self, rh):
        import torch
        if self._dev is None or self._dev.dtype != torch.int32:
            self._dev = torch.from_numpy(np.ascontiguousarray(self.numpy().astype(np.int32))).to(rh.device)
        return self._dev

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def tolist(self):
        return self.numpy().tolist()

    # -- list protocol -----------------------------------------------------------------------------
    def __len__(self):
        return int(self._np.size if self._np is not None else self._dev.numel())

    def __getitem__(self, i):
        v = self.numpy()[i]
        return IndexList(v) if isinstance(i, slice) else int(v)

    def __iter__(self):
        return iter(self.numpy().tolist())

    def count(self, value):
        return int(np.count_nonzero(self.numpy() == value))

    def index(self, value, *args):
        return self.numpy().tolist().index(value, *args)

    def __eq__(self, other):
        try:
            return len(other) == len(self) and bool(np.array_equal(self.numpy(), np.asarray(other).reshape(-1)))
        except Exception:
            return NotImplemented

    def __repr__(self):
        return f"IndexList(n={len(self)})"

    def max(self):
        return int(self.numpy().max()) if len(self) else 0


def as_index_array(indices):
    if isinstance(indices, IndexList):
        return indices.numpy()
    return np.asarray(indices).reshape(-1)
