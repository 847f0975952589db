"""A host program with no Python in it (tests/native/encode_frame_host.cpp: C++, the HIP runtime API and include/rhccq.h only) encodes a
frame through rhccq_encode_frame in a child process; its palette and index map must equal FrameEncoder.encode's.  GPU only."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_host_without_python_encodes_the_same_frame(tmp_path):
    import torch
    from roibasedimagecompression_amd import synth
    from roibasedimagecompression_amd.frame import ClassSpec, FrameEncoder
    from roibasedimagecompression_amd.ops import Rhccq
    exe = str(tmp_path / "encode_frame_host")
    lib_dir = os.path.join(ROOT, "roibasedimagecompression_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "encode_frame_host.cpp"),
                           "-o", exe, "-L", lib_dir, "-l:librhccq_hip.so", f"-Wl,-rpath,{lib_dir}"])
    H, W = 320, 416
    img = synth.photo(H, W, 909, sigma=6.0).copy()              # (MiniBatch-branch segments among them)
    img[100:104, 50:120] = 0
    (lr, nr, br), (ln, nn, bn) = synth.frame_classes(H, W, (2, 1))
    qs = (20, 10)
    frame = tmp_path / "frame.bin"
    with open(frame, "wb") as f:
        f.write(np.array([H, W, 2], np.int32).tobytes())
        for n, b, q in ((nr, br, qs[0]), (nn, bn, qs[1])):
            f.write(np.array([n, 1, q], np.int32).tobytes())
            f.write(np.zeros(n, np.int32).tobytes())
            f.write(np.array(b, np.int32).tobytes())
        f.write(np.ascontiguousarray(img).tobytes())
        f.write(np.ascontiguousarray(lr, np.int32).tobytes())
        f.write(np.ascontiguousarray(ln, np.int32).tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(frame), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    n_col, eb, sh0, sh1, tl0, tl1 = np.frombuffer(raw[:24], np.int32)
    pal = np.frombuffer(raw[24:24 + 3 * n_col], np.uint8).reshape(-1, 3)
    idx = np.frombuffer(raw[24 + 3 * n_col:], {1: np.uint8, 2: np.uint16, 4: np.uint32}[int(eb)]).reshape(H, W)
    rh = Rhccq(0)
    specs = [ClassSpec(torch.from_numpy(lr).to(rh.device), np.zeros(nr, np.int64), [br], qs[0]),
             ClassSpec(torch.from_numpy(ln).to(rh.device), np.zeros(nn, np.int64), [bn], qs[1])]
    ref = FrameEncoder(rh).encode(torch.from_numpy(img).to(rh.device), specs)
    assert (ref["n_unique"] >= 10000).any()
    ridx = ref["indices"].cpu().numpy()
    if ref["indices_dtype"] == "uint16":
        ridx = ridx.view(np.uint16)
    assert np.array_equal(pal, ref["palette"]) and np.array_equal(idx.astype(np.int64), ridx.astype(np.int64))
    assert (int(sh0), int(sh1)) == tuple(ref["shape"]) and (int(tl0), int(tl1)) == tuple(ref["top_left"])
