"""decoder/uncompression/comparison.py:30-80 on the device: `calculate_quality_metrics` with the reference's keys
and numpy scalar types.  The other names of that module (image loading through OpenCV, matplotlib reports) stay
with the reference.

Arithmetic: the error statistics come from exact integer sums (the reference averages float32 arrays with
numpy's pairwise float32 summation: its `mse`, `mae`, `mse_r/g/b` agree to float32 rounding, ~1e-7 relative);
`psnr` is scikit-image's float64 `10*log10(255^2 / mse)`; `ssim` is scikit-image's
structural_similarity(data_range=255, channel_axis=2, win_size=7) from exact integer window sums."""
import numpy as np
import torch

from .. import ops

_RH = None


def _rh():
    global _RH
    if _RH is None:
        _RH = ops.Rhccq(0)
    return _RH


def calculate_quality_metrics(original, reconstructed):
    """Same keys as the reference: psnr, ssim, mse, rmse, mae, max_error, mse_r, mse_g, mse_b."""
    rh = _rh()
    original = np.ascontiguousarray(original)
    reconstructed = np.ascontiguousarray(reconstructed)
    if original.shape != reconstructed.shape or original.ndim != 3 or original.shape[2] != 3:
        raise ValueError("Input images must have the same dimensions.")      # skimage's check_shape_equality
    if original.dtype != np.uint8 or reconstructed.dtype != np.uint8:
        raise TypeError("calculate_quality_metrics: uint8 RGB images expected (what cv2.imread yields)")
    a, b = torch.from_numpy(original).to(rh.device), torch.from_numpy(reconstructed).to(rh.device)
    return metrics_from_device(rh, a, b)


def metrics_from_device(rh, a, b):
    """a, b: uint8[H,W,3] device tensors."""
    H, W = int(a.shape[0]), int(a.shape[1])
    s = rh.error_sums(a, b).astype(np.int64)
    n_ch = H * W
    sq = int(s[0]) + int(s[1]) + int(s[2])
    mse64 = sq / (3.0 * n_ch)
    m = {}
    with np.errstate(divide="ignore"):
        m["psnr"] = np.float64(10.0) * np.log10(np.float64(255.0 ** 2) / np.float64(mse64))
    try:
        m["ssim"] = np.float64(rh.ssim7(a, b).mean())
    except ValueError:
        m["ssim"] = np.float64(0.0)                      # the reference's fallback for images a 7x7 window does not fit
    m["mse"] = np.float32(mse64)
    m["rmse"] = np.sqrt(m["mse"])
    m["mae"] = np.float32(int(s[3]) / (3.0 * n_ch))
    m["max_error"] = np.float32(int(s[4]))
    for i, ch in enumerate("rgb"):
        m[f"mse_{ch}"] = np.float32(int(s[i]) / float(n_ch))
    return m
