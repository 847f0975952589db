"""Placeholder for a stage that is UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).
The names exist so that the reference's import lines resolve; a deployment keeps the reference's own
module here (it needs OpenCV / scikit-image, which this build does not re-implement yet)."""


def _upstream(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name}: stage outside the MI355X hot path -- keep the reference's module for it "
                                  "(see INTEGRATION.md)")
    fn.__name__ = name
    return fn


enhanced_slic_with_texture = _upstream("enhanced_slic_with_texture")
extract_slic_segment_boundaries = _upstream("extract_slic_segment_boundaries")
visualize_split_analysis = _upstream("visualize_split_analysis")
watershed_segmentation_with_mask = _upstream("watershed_segmentation_with_mask")
