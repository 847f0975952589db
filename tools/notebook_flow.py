"""diagnostic: the reference's end-to-end flow (encoder/compression/test.py = rhccq.ipynb cells 6-16) driven through the mirrored modules only, image file -> .rhccq,
compared with the artefact the reference ships for the same image and settings:  python tools/notebook_flow.py [png] [artefact]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image


from roibasedimagecompression_amd.flow import script_flow as notebook_flow  # noqa: E402,F401


def report(png, artefact, out_path, roi_quality=20, nonroi_quality=10):
    from decoder.uncompression.uncompression import load_compressed, lossless_decompress, decompress_color_quantization
    img = np.asarray(Image.open(png).convert("RGB"), dtype=np.uint8)
    final, pkg, info = notebook_flow(img, roi_quality, nonroi_quality, out_path=out_path)
    rec = np.asarray(decompress_color_quantization(lossless_decompress(load_compressed(out_path)))["image"])

    def stats(a):
        mse = float(np.mean((a.astype(np.float64) - img.astype(np.float64)) ** 2))
        return {"colours": int(len(np.unique(a.reshape(-1, 3), axis=0))), "psnr": round(10 * np.log10(255.0 ** 2 / mse), 2)}
    ref = np.asarray(decompress_color_quantization(lossless_decompress(load_compressed(artefact)))["image"])
    return {"image": os.path.basename(png), "this_build": dict(stats(rec), bytes=os.path.getsize(out_path)),
            "reference_artefact": dict(stats(ref), bytes=os.path.getsize(artefact)), **info}


def all_pairs():
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    return [("Lenna", os.path.join(g, "Lenna.png"), os.path.join(g, "Lenna_compressed_20_10.rhccq"))] + \
           [(f"kodak_{i}", os.path.join(g, f"kodak_{i}.png"), os.path.join(g, f"compressed_{i}.rhccq")) for i in range(1, 25)]


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--all":
    # every (image, artefact) pair the reference ships for the (20, 10) preset -> one JSON table (profiles/r03_script_flow.json)
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    table = {}
    for name, png, art in all_pairs():
        table[name] = report(png, art, os.path.join("gpurun_out", "flow_" + name + ".rhccq"))
        print(name, table[name], flush=True)
    json.dump(table, open(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/script_flow_all.json", "w"), indent=1)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--infer":
    # the artefacts under images/rhccq/ were written with OTHER settings the reference does not record: try presets
    import json
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs("gpurun_out", exist_ok=True)
    table = {}
    for i in (1, 5):
        for qr, qn in ((100, 100), (95, 95), (90, 90), (100, 50), (80, 80)):
            try:
                r = report(os.path.join(g, f"kodak_{i}.png"), os.path.join(g, f"other_settings_compressed_{i}.rhccq"), os.path.join("gpurun_out", "infer.rhccq"), qr, qn)
                table[f"kodak_{i}@({qr},{qn})"] = {"this_build": r["this_build"], "artefact": r["reference_artefact"]}
            except Exception as e:                          # (q = 100 divides by zero nowhere, but be safe)
                table[f"kodak_{i}@({qr},{qn})"] = {"error": repr(e)}
            print(f"kodak_{i}@({qr},{qn})", table[f"kodak_{i}@({qr},{qn})"], flush=True)
    json.dump(table, open("gpurun_out/script_flow_infer.json", "w"), indent=1)
    sys.exit(0)

if __name__ == "__main__":
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    pairs = [(os.path.join(g, "Lenna.png"), os.path.join(g, "Lenna_compressed_20_10.rhccq")), (os.path.join(g, "kodak_23.png"), os.path.join(g, "compressed_23.rhccq")),
             (os.path.join(g, "kodak_1.png"), os.path.join(g, "compressed_1.rhccq")), (os.path.join(g, "kodak_13.png"), os.path.join(g, "compressed_13.rhccq")),
             (os.path.join(g, "kodak_5.png"), os.path.join(g, "compressed_5.rhccq")), (os.path.join(g, "kodak_15.png"), os.path.join(g, "compressed_15.rhccq"))]
    if len(sys.argv) > 2:
        pairs = [(sys.argv[1], sys.argv[2])]
    os.makedirs("gpurun_out", exist_ok=True)
    for png, art in pairs:
        print(report(png, art, os.path.join("gpurun_out", "flow_" + os.path.basename(png) + ".rhccq")), flush=True)
