"""decoder/uncompression/comparison.py of the reference: `calculate_quality_metrics` and `calculate_adaptive_quality_metrics` run on the MI355X
(roibasedimagecompression_amd/api/comparison.py).  The OpenCV / matplotlib helpers of that module come from the reference's own
file when its checkout sits behind this repository on sys.path (they are looked up there lazily); otherwise they are
placeholders that raise."""
import importlib.util
import os
import sys

from roibasedimagecompression_amd._shim import upstream
from roibasedimagecompression_amd.api.comparison import calculate_adaptive_quality_metrics, calculate_quality_metrics  # noqa: F401

_HELPERS = ("create_difference_visualization", "print_quality_report", "plot_comparison", "print_adaptive_metrics")


def _downstream_module():
    pkg = sys.modules[__name__.rpartition(".")[0]]
    here = os.path.dirname(os.path.abspath(__file__))
    for d in pkg.__path__:
        cand = os.path.join(d, "comparison.py")
        if os.path.abspath(d) != here and os.path.isfile(cand):
            spec = importlib.util.spec_from_file_location(__name__ + "._reference", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
    return None


def __getattr__(name):
    if name in _HELPERS:
        mod = _downstream_module()
        fn = getattr(mod, name) if mod is not None else upstream(name)
        globals()[name] = fn
        return fn
    raise AttributeError(name)
