"""Drop-in for the reference's encoder/compression/image.py: same names, MI355X implementation in
roibasedimagecompression_amd.api.image."""
from roibasedimagecompression_amd.api.image import *  # noqa: F401,F403
from roibasedimagecompression_amd.api import image as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
