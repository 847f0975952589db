"""Drop-in for the reference's decoder/uncompression/uncompression.py."""
from roibasedimagecompression_amd.api import uncompression as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

# names of the reference's module that the pipeline never calls (unused variants, debug helpers): taken from the reference's own file
# when its checkout sits behind this repository on sys.path, placeholders that raise otherwise
from roibasedimagecompression_amd._shim import downstream_getattr  # noqa: E402

__getattr__ = downstream_getattr(__name__, __file__, ('decompress_indices_rle_huffman',))
