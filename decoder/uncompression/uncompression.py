"""Drop-in for the reference's decoder/uncompression/uncompression.py."""
from roibasedimagecompression_amd.api import uncompression as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
