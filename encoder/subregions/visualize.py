"""Placeholder for a stage that is UPSTREAM/DOWNSTREAM of the MI355X hot path (SURVEY.md 8f "next" rows).
The names exist so that the reference's import lines resolve; a deployment keeps the reference's own
module here (it needs OpenCV / scikit-image, which this build does not re-implement yet)."""


def _upstream(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"{name}: stage outside the MI355X hot path -- keep the reference's module for it "
                                  "(see INTEGRATION.md)")
    fn.__name__ = name
    return fn


plot_regions = _upstream("plot_regions")
