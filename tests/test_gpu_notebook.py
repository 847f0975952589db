"""The reference's end-to-end flow (encoder/compression/test.py:77-151, the script twin of rhccq.ipynb: the flow that wrote the
images/rhccq_20_10/*.rhccq artefacts) driven through the mirrored modules ONLY -- ROI detection, region extraction, split score +
masked SLIC, the three clustering levels, the container -- from the PNG to the .rhccq file and back.  GPU only.

What can be compared with the reference here is Tier B: the stages upstream of the hot path are parity-unpinned restatements of
OpenCV / scikit-image (neither library exists in the build container), so EVERY artefact the reference ships together with its
source image is the yardstick (copied as data into tests/golden/): the 25 files of images/rhccq_20_10 (tiers 20 / 10) and the 8 Kodak
files of images/rhccq, whose unrecorded settings turn out to be (100, 100) (profiles/r03_script_flow.json: the only preset that
reproduces their palette and file sizes).  Per image: PSNR against the original, palette size, file size, next to the artefact's
(tests/golden/g14_artefact_stats.json, decoded on the CPU by the oracle).  Observed over the 25 (20, 10) pairs: dPSNR mean -0.42 dB
(-2.49 ... +1.46), |dbytes| mean 3.3 % (max 10.5 %); which stage is responsible is examined in profiles/r03_script_flow.json
(tools/flow_sensitivity.py): both the ROI map and the SLIC segment count move the outliers by 1 - 3 dB, neither explains all."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
ARTEFACT = json.load(open(os.path.join(G, "g14_artefact_stats.json")))

# dPSNR (dB) and dbytes (fraction) allowed against the artefact: the observed value + a margin; the rest of the set stays within
# 0.9 dB / 7 %
OUTLIERS = {"kodak_3": (2.7, 0.12), "kodak_17": (1.6, 0.07), "kodak_20": (1.7, 0.07), "kodak_19": (1.0, 0.11)}
NAMES = ["Lenna"] + [f"kodak_{i}" for i in range(1, 25)]


def _flow():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import notebook_flow
    finally:
        sys.path.pop(0)
    return notebook_flow


@pytest.mark.parametrize("name", NAMES)
def test_script_flow_vs_the_reference_artefact(name, tmp_path):
    nf = _flow()
    png = os.path.join(G, "Lenna.png" if name == "Lenna" else name + ".png")
    art = os.path.join(G, "Lenna_compressed_20_10.rhccq" if name == "Lenna" else "compressed_" + name.split("_")[1] + ".rhccq")
    out = str(tmp_path / "out.rhccq")
    rep = nf.report(png, art, out)
    ref, mine = rep["reference_artefact"], rep["this_build"]
    want = ARTEFACT[name]
    assert ref["colours"] == want["colours"] and abs(ref["psnr"] - want["psnr"]) < 0.006 and ref["bytes"] == want["bytes"]   # the artefact decodes to what it always did
    assert rep["roi_regions"] + rep["nonroi_regions"] >= 1 and 0.05 < rep["region_map_roi_fraction"] <= 1.0
    assert rep["roi_segments"] + rep["nonroi_segments"] >= 2
    dpsnr, dbytes = OUTLIERS.get(name, (0.9, 0.07))
    assert abs(mine["psnr"] - ref["psnr"]) <= dpsnr, rep
    assert abs(mine["bytes"] - ref["bytes"]) <= dbytes * ref["bytes"], rep
    assert 0.68 * ref["colours"] <= mine["colours"] <= 1.4 * ref["colours"], rep       # final palettes: 100 - 150 colours on both sides
    if name in ("Lenna", "kodak_23", "kodak_8"):                                        # deterministic: the same file again
        rep2 = nf.report(png, art, str(tmp_path / "again.rhccq"))
        assert open(out, "rb").read() == open(str(tmp_path / "again.rhccq"), "rb").read() and rep2["this_build"] == mine


def test_script_flow_set_statistics(tmp_path):
    """over the whole (20, 10) set the build sits 0.4 dB below the artefacts at 3 % smaller files: bounds on the MEANS, so that a
    systematic drift of the unpinned stages shows even while every image stays inside its own tolerance"""
    nf = _flow()
    d, b = [], []
    for name, png, art in nf.all_pairs():
        rep = nf.report(png, art, str(tmp_path / "o.rhccq"))
        d.append(rep["this_build"]["psnr"] - rep["reference_artefact"]["psnr"])
        b.append((rep["this_build"]["bytes"] - rep["reference_artefact"]["bytes"]) / rep["reference_artefact"]["bytes"])
    assert -0.6 <= sum(d) / len(d) <= 0.0, d
    assert sum(abs(x) for x in b) / len(b) <= 0.04 and -0.04 <= sum(b) / len(b) <= 0.0, b


@pytest.mark.parametrize("n", range(1, 9))
def test_script_flow_vs_the_other_settings_artefacts(n, tmp_path):
    """images/rhccq/compressed_{1..8}.rhccq: settings (100, 100) -- inferred, the reference does not record them.  Near-lossless
    palettes of 13 000 - 63 000 colours: palette size within 2.5 %, file size within 1.5 %, PSNR above 39 dB and within 9 dB of the artefact's."""
    nf = _flow()
    rep = nf.report(os.path.join(G, f"kodak_{n}.png"), os.path.join(G, f"other_settings_compressed_{n}.rhccq"), str(tmp_path / "o.rhccq"), 100, 100)
    ref, mine = rep["reference_artefact"], rep["this_build"]
    want = ARTEFACT[f"other_{n}"]
    assert ref["colours"] == want["colours"] and ref["bytes"] == want["bytes"]
    assert abs(mine["colours"] - ref["colours"]) <= 0.025 * ref["colours"], rep
    assert abs(mine["bytes"] - ref["bytes"]) <= 0.015 * ref["bytes"], rep
    # near-lossless on both sides (39.6 - 60.7 dB in the artefacts; observed here 39.6 - 52.3, up to 8.5 dB below where a handful of
    # k = n KMeans splits resolve differently -- at this quality every merged colour costs decibels)
    assert mine["psnr"] >= max(39.0, ref["psnr"] - 9.0), rep
