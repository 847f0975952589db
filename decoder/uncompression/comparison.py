"""decoder/uncompression/comparison.py of the reference: `calculate_quality_metrics` runs on the MI355X
(roibasedimagecompression_amd/api/comparison.py); the OpenCV / matplotlib helpers of that module are placeholders
-- a deployment keeps the reference's own functions for them."""
from roibasedimagecompression_amd.api.comparison import calculate_quality_metrics  # noqa: F401


def _upstream(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name}: stage outside the MI355X hot path -- keep the reference's module for it "
                                  "(see INTEGRATION.md)")
    fn.__name__ = name
    return fn


create_difference_visualization = _upstream("create_difference_visualization")
print_quality_report = _upstream("print_quality_report")
plot_comparison = _upstream("plot_comparison")
calculate_adaptive_quality_metrics = _upstream("calculate_adaptive_quality_metrics")
print_adaptive_metrics = _upstream("print_adaptive_metrics")
