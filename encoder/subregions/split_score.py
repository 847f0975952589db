"""Drop-in for the reference's encoder/subregions/split_score.py: same names, MI355X implementation in
roibasedimagecompression_amd.api.split_score (parity unpinned: scikit-image semantics restated)."""
from roibasedimagecompression_amd.api.split_score import (calculate_optimal_segments, calculate_split_score,  # noqa: F401
                                                          normalize_result)
