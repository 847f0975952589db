"""Drop-in for the reference's encoder/compression/clustering.py: same names, MI355X implementation in
roibasedimagecompression_amd.api.clustering."""
from roibasedimagecompression_amd.api.clustering import *  # noqa: F401,F403
from roibasedimagecompression_amd.api import clustering as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
