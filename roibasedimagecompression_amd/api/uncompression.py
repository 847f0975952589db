"""Mirror of decoder/uncompression/uncompression.py (uncompression.py:58-291).  The container is a
pickle from an untrusted source: it is read through an allow-list unpickler (the reference uses bare
pickle.loads, uncompression.py:150).  The palette[index] gather runs on the device (rhccq_decode)."""
import io
import pickle
import struct
import zlib

import numpy as np

from ..segment import as_index_array


class _SafeUnpickler(pickle.Unpickler):
    _OK = {("numpy._core.multiarray", "scalar"), ("numpy.core.multiarray", "scalar"), ("numpy", "dtype")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"forbidden global {module}.{name} in .rhccq container")


def load_compressed(filename):
    with open(filename, "rb") as f:
        if f.read(5) != b"RHCCQ":
            raise ValueError("Invalid file format")
        size = struct.unpack("<I", f.read(4))[0]
        body = f.read(size)
    return _SafeUnpickler(io.BytesIO(zlib.decompress(body))).load()


def decompress_palette(palette_data, palette_size):
    pal = np.frombuffer(zlib.decompress(palette_data), dtype=np.uint8).reshape(-1, 3)
    return [tuple(int(v) for v in row) for row in pal[:palette_size]]


def decompress_indices_simple(indices_data, total_pixels, dtype_str="uint16"):
    raw = zlib.decompress(indices_data)
    if dtype_str in ("uint8", "uint16", "uint32"):
        dt = np.dtype(dtype_str)
    else:
        bpp = len(raw) / total_pixels if total_pixels > 0 else 2
        dt = np.dtype(np.uint8 if bpp <= 1 else (np.uint16 if bpp <= 2 else np.uint32))
    return np.frombuffer(raw, dtype=dt).tolist()


def lossless_decompress(compressed_data):
    shape = compressed_data["s"]
    h, w = shape
    palette = decompress_palette(compressed_data["p"], compressed_data["l"])
    return palette, decompress_indices_simple(compressed_data["i"], h * w, compressed_data.get("d", "uint16")), shape


def _gather(palette, indices, h, w):
    import torch
    from ..ops import default_context
    rh = default_context()
    pal = np.array(palette, dtype=np.uint8).reshape(-1, 3)
    idx = as_index_array(indices).astype(np.int32)
    rec = rh.decode(torch.from_numpy(np.ascontiguousarray(idx)).to(rh.device), torch.from_numpy(pal).to(rh.device))
    return rec.cpu().numpy().reshape(h, w, 3)


def decompress_color_quantization(compressed_data):
    if isinstance(compressed_data, tuple):
        palette, indices, shape = compressed_data
        top_left, quality = (0, 0), 50
        h, w = shape
    else:
        top_left = compressed_data.get("top_left", (0, 0))
        quality = compressed_data.get("quality", 50)
        if "shape" in compressed_data:
            h, w = compressed_data["shape"]
        else:
            h, w = compressed_data["s"]
        palette, indices, _ = lossless_decompress(compressed_data)
    return {"image": _gather(palette, indices, h, w), "top_left": top_left, "shape": (h, w),
            "method": "color_quantization", "quality": quality}


def partial_decompress_color_quantization(compressed_data):
    if isinstance(compressed_data, tuple):
        palette, indices, shape = compressed_data
        top_left, quality = (0, 0), 50
    else:
        top_left = compressed_data.get("top_left", (0, 0))
        quality = compressed_data.get("quality", 50)
        palette, indices, shape = compressed_data["palette"], compressed_data["indices"], compressed_data["shape"]
    h, w = shape
    return {"image": _gather(palette, indices, h, w), "top_left": top_left, "shape": (h, w),
            "method": "color_quantization", "quality": quality}
