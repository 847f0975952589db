"""decoder/uncompression/comparison.py of the reference: `calculate_quality_metrics` and `calculate_adaptive_quality_metrics` run on the MI355X
(roibasedimagecompression_amd/api/comparison.py).  The OpenCV / matplotlib helpers of that module come from the reference's own
file when its checkout sits behind this repository on sys.path (they are looked up there lazily); otherwise they are
placeholders that raise."""
from roibasedimagecompression_amd._shim import downstream_getattr
from roibasedimagecompression_amd.api.comparison import calculate_adaptive_quality_metrics, calculate_quality_metrics  # noqa: F401

_HELPERS = ("create_difference_visualization", "print_quality_report", "plot_comparison", "print_adaptive_metrics")

# the same loader as the encoder mirrors: the reference's file is executed once, an ImportError inside it (it needs OpenCV,
# matplotlib, scikit-image) is caught and the documented placeholders are served instead
__getattr__ = downstream_getattr(__name__, __file__, _HELPERS)
