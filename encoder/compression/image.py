"""Drop-in for the reference's encoder/compression/image.py: same names, MI355X implementation in
roibasedimagecompression_amd.api.image."""
from roibasedimagecompression_amd.api.image import *  # noqa: F401,F403
from roibasedimagecompression_amd.api import image as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})

# names of the reference's module that the pipeline never calls (unused variants, debug helpers): taken from the reference's own file
# when its checkout sits behind this repository on sys.path, placeholders that raise otherwise
from roibasedimagecompression_amd._shim import downstream_getattr  # noqa: E402

__getattr__ = downstream_getattr(__name__, __file__, ('fill_black_holes_in_segment', 'fill_black_holes_vectorized'))
