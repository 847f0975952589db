"""Drop-in for the reference's encoder/ROI/thin_regions2.py: the whole module runs on the MI355X
(roibasedimagecompression_amd.api.roi_chain: connected components, chamfer distance transform, box densities; parity unpinned)."""
from roibasedimagecompression_amd.api.roi_chain import (identify_thin_regions_fast, identify_thin_regions_ultrafast,  # noqa: F401
                                                        remove_thin_structures_optimized)
