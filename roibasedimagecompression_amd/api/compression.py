"""Mirror of encoder/compression/compression.py: the lossless back-end and the .rhccq container
(compression.py:10-22,119-142,151-220,326-413).  Host-side zlib/pickle exactly as the reference (this is
the NEXT-3 row of SURVEY.md 8f: it sits after the replaced path)."""
import pickle
import struct
import zlib

import numpy as np

from ..segment import IndexList, as_index_array


def compress_palette(palette):
    return zlib.compress(np.array(palette, dtype=np.uint8).tobytes(), level=9)


def compress_indices_simple(indices_list):
    return zlib.compress(np.array(as_index_array(indices_list), dtype=np.uint16).tobytes(), level=9)


def compress_indices_simple_optimized(indices_data, dtype=np.uint8):
    return zlib.compress(as_index_array(indices_data).astype(dtype, copy=False).tobytes(), level=9)


def _dtype_for(max_index):
    if max_index < 256:
        return np.uint8, "uint8"
    if max_index < 65536:
        return np.uint16, "uint16"
    return np.uint32, "uint32"


def lossless_compress_optimized(palette, indices_list, shape, use_manual_rle=False):
    if not isinstance(indices_list, (list, np.ndarray, IndexList)):
        raise TypeError(f"indices_list must be list or numpy array, got {type(indices_list)}")
    arr = as_index_array(indices_list)
    dtype, name = _dtype_for(int(arr.max()) if arr.size else 0)
    return {"s": shape, "l": len(palette), "p": compress_palette(palette),
            "i": compress_indices_simple_optimized(arr, dtype), "d": name}


def lossless_compress(palette, indices_list, shape, use_manual_rle=False):
    if use_manual_rle:
        raise NotImplementedError("legacy RLE+zlib variant (compression.py:25-66) is unused by the pipeline")
    return {"s": shape, "ps": len(palette), "p": compress_palette(palette), "i": compress_indices_simple(indices_list)}


def save_compressed(compressed_data, filename):
    body = zlib.compress(pickle.dumps(compressed_data, protocol=5), level=9)
    with open(filename, "wb") as f:
        f.write(b"RHCCQ")
        f.write(struct.pack("<I", len(body)))
        f.write(body)
    return len(body) + 8


def optimize_compressed_dtype(compressed_data):
    """dtype tag by max index (compression.py:326-413); indices stay numerically identical."""
    if "indices" not in compressed_data:
        return compressed_data
    arr = as_index_array(compressed_data["indices"])
    _, name = _dtype_for(int(arr.max()) if arr.size else 0)
    out = compressed_data.copy()
    out["indices_dtype"] = name
    out["indices_optimized"] = True
    if "palette" in out:
        out["actual_colors"] = len(out["palette"])
    return out
