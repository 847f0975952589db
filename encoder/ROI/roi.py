"""Placeholder for a stage that is UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).
The names exist so that the reference's import lines resolve; a deployment keeps the reference's own
module here (it needs OpenCV / scikit-image, which this build does not re-implement yet)."""


def _upstream(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name}: stage outside the MI355X hot path -- keep the reference's module for it "
                                  "(see INTEGRATION.md)")
    fn.__name__ = name
    return fn


get_regions = _upstream("get_regions")
extract_regions = _upstream("extract_regions")
remove_small_noise_regions = _upstream("remove_small_noise_regions")
detect_meaningful_borders = _upstream("detect_meaningful_borders")
protect_border_regions = _upstream("protect_border_regions")
fill_closed_regions = _upstream("fill_closed_regions")
extract_roi_nonroi = _upstream("extract_roi_nonroi")
visualize_roi_nonroi_comparison = _upstream("visualize_roi_nonroi_comparison")
process_and_unify_borders = _upstream("process_and_unify_borders")
directional_region_unification = _upstream("directional_region_unification")
extract_connected_regions_fast = _upstream("extract_connected_regions_fast")
