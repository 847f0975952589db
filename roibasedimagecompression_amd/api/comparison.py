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


def calculate_adaptive_quality_metrics(original, reconstructed):
    """comparison.py:345-536: quality metrics with adaptive outlier exclusion, same keys and nesting as the reference.

    Every statistic of the reference is a function of the per-pixel worst-channel error (an integer 0..255) and of sums over
    pixel subsets defined by thresholds on it, so one device pass builds a 256-row table (pixels, squared differences,
    absolute differences per error value) and the rest is the reference's own arithmetic on that table: the percentiles
    and the float32 mean / std come from numpy on the (sorted) multiset of error values -- the float32 mean may differ
    from the reference's raster-order pairwise sum in its last bits --, the subset MSE / MAE are exact integer sums
    (the reference averages float32 arrays: agreement to ~1e-7 relative), SSIM is the device kernel of
    calculate_quality_metrics (scikit-image's algorithm, parity unpinned), the masked SSIM greys the outliers on the device."""
    rh = _rh()
    original = np.ascontiguousarray(original)
    reconstructed = np.ascontiguousarray(reconstructed)
    if original.shape != reconstructed.shape or original.ndim != 3 or original.shape[2] != 3:
        raise ValueError("Input images must have the same dimensions.")
    if original.dtype != np.uint8 or reconstructed.dtype != np.uint8:
        raise TypeError("calculate_adaptive_quality_metrics: uint8 RGB images expected")
    a, b = torch.from_numpy(original).to(rh.device), torch.from_numpy(reconstructed).to(rh.device)
    tab, d_maxerr = rh.error_tables(a, b, want_maxerr=True)
    cnt, sq, ab = tab[:, 0], tab[:, 1], tab[:, 2]
    n = int(cnt.sum())
    err = np.repeat(np.arange(256, dtype=np.float32), cnt)             # max_error_per_pixel, sorted
    stats = {"min": float(np.min(err)), "max": float(np.max(err)), "mean": float(np.mean(err)), "median": float(np.median(err)),
             "std": float(np.std(err)), "q75": float(np.percentile(err, 75)), "q90": float(np.percentile(err, 90)),
             "q95": float(np.percentile(err, 95)), "q99": float(np.percentile(err, 99))}
    q1, q3 = np.percentile(err, 25), np.percentile(err, 75)
    iqr_threshold = q3 + 2.5 * (q3 - q1)
    z_threshold = 3.0
    percentile_threshold = np.percentile(err, 99)
    if stats["mean"] > stats["median"] * 1.5:
        adaptive_threshold = stats["median"] + 3 * stats["std"]
    else:
        adaptive_threshold = stats["mean"] + 2.5 * stats["std"]
    values = np.arange(256, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (values - np.float32(stats["mean"])) / np.float32(stats["std"])
    masks = {"iqr": values > iqr_threshold, "zscore": np.abs(z) > z_threshold, "percentile": values > percentile_threshold,
             "adaptive": values > adaptive_threshold}                    # per error VALUE: which values are outliers
    best_method, best = None, None
    for name, mk in masks.items():
        pct = int(cnt[mk].sum()) / n * 100
        if 0.1 <= pct <= 10.0:
            best_method, best = name, mk
            break
    if best_method is None:
        best_method, best = "adaptive", masks["adaptive"]
    outlier_count = int(cnt[best].sum())
    outlier_pct = outlier_count / n * 100
    metrics = {"error_distribution": stats,
               "outlier_detection": {"method": best_method,
                                     "threshold": float({"iqr": iqr_threshold, "zscore": stats["mean"] + z_threshold * stats["std"],
                                                         "percentile": percentile_threshold, "adaptive": adaptive_threshold}[best_method]),
                                     "outlier_count": outlier_count, "outlier_percentage": float(outlier_pct),
                                     "inlier_count": int(n - outlier_count), "inlier_percentage": float(100 - outlier_pct)}}

    def subset(mk):
        c = int(cnt[mk].sum())
        mse = np.float32(int(sq[mk].sum()) / (3.0 * c))
        return c, mse, np.float32(int(ab[mk].sum()) / (3.0 * c))

    def psnr(mse):
        return 10 * np.log10(255 * 255 / mse) if mse > 0 else float("inf")
    c_all, mse_all, mae_all = subset(np.ones(256, bool))
    metrics["all_pixels"] = {"psnr": psnr(mse_all), "mse": float(mse_all), "rmse": float(np.sqrt(mse_all)), "mae": float(mae_all),
                             "max_error": stats["max"], "pixel_count": n}
    if 0 < outlier_count < n:
        c_in, mse_in, mae_in = subset(~best)
        metrics["without_outliers"] = {"psnr": psnr(mse_in), "mse": float(mse_in), "rmse": float(np.sqrt(mse_in)), "mae": float(mae_in),
                                       "max_error": float(np.max(values[~best & (cnt > 0)])), "pixel_count": c_in}
    for percentile in (99, 95, 90, 75):
        threshold = np.percentile(err, percentile)
        mk = values <= threshold
        c_p, mse_p, _ = subset(mk)
        if c_p > 0:
            metrics[f"percentile_{percentile}"] = {"psnr": psnr(mse_p), "mse": float(mse_p), "max_error_included": float(threshold),
                                                   "pixel_count": c_p, "percentage": float(percentile)}
    try:
        metrics["ssim"] = {"full": float(rh.ssim7(a, b).mean())}
        if 0 < outlier_count < n:
            out2d = torch.from_numpy(best).to(rh.device)[d_maxerr.long()]          # outlier mask per pixel
            grey = torch.full_like(a, 128)
            am, bm = torch.where(out2d[..., None], grey, a).contiguous(), torch.where(out2d[..., None], grey, b).contiguous()
            metrics["ssim"]["without_outliers"] = float(rh.ssim7(am, bm).mean())
    except ValueError:
        metrics["ssim"] = {"full": 0}
    present = cnt > 0
    hist, edges = np.histogram(values[present], bins=50, weights=cnt[present])    # same bin edges as over the per-pixel array
    metrics["error_histogram"] = {"bins": [int(v) for v in hist], "bin_edges": edges.tolist()}
    return metrics
