"""Edge front end of the ROI stage on the MI355X (reference encoder/ROI/edges.py):

  get_edge_map / find_best_edges_by_quality   edges.py:35-71   20 adaptive threshold pairs, each scored on its Canny edge map, then
                                                               the winner applied to the COLOUR image
  compute_adaptive_canny_thresholds           edges.py:88-169  Otsu / gradient percentiles / gradient statistics / hybrid
  evaluate_edge_quality                       edges.py:73-86   mean 8-connected component size x std of the gray values on the edges
  compute_local_density                       edges.py:173-195 normalised k x k box filter (BORDER_REFLECT_101)
  suggest_automatic_threshold                 edges.py:4-32
  get_edge_map_fast / compute_fast_canny_thresholds   edges.py:200-298

PARITY UNPINNED (OpenCV is absent from the build container): cvtColor, Sobel, Otsu and Canny are restated from OpenCV's
published integer implementations (csrc/edges.hip, oracle/rhccq_oracle.py cv_*); device == restatement bit for bit
(tests/test_gpu_roi.py).  Design: the gray image, its histogram, the histogram of the squared Sobel magnitude and Canny's
non-maximum suppression are computed ONCE (none depends on the thresholds); each threshold pair then costs one connected-component
labelling of {nm > low} plus a per-label reduction, and the score needs only per-label numbers -- no edge map is materialised
until the winner is known.  Arrays come back as numpy, as the reference's callers expect."""
import math

import numpy as np

from ..ops import default_context


# ---- host arithmetic on histograms -----------------------------------------------------------------------------------------
def _otsu(hist):
    """getThreshVal_Otsu_8u on a 256-bin histogram"""
    h = hist.astype(np.float64)
    total = float(hist.sum())
    scale = 1.0 / total
    mu = float((np.arange(256) * h).sum()) * scale
    mu1 = q1 = 0.0
    max_sigma = max_val = 0.0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p_i = h[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, float(i)
    return max_val


def _percentile(values, counts, q):
    """np.percentile(x, q) (method 'linear') of the multiset {values[i] repeated counts[i] times}, values ascending"""
    n = int(counts.sum())
    cum = np.cumsum(counts)
    vi = (n - 1) * (q / 100.0)
    lo = math.floor(vi)
    t = vi - lo
    a = float(values[np.searchsorted(cum, lo, side="right")])
    b = float(values[np.searchsorted(cum, min(lo + 1, n - 1), side="right")])
    d = b - a
    return b - d * (1 - t) if t >= 0.5 else a + d * t


class _Gradient:
    """statistics of the float64 gradient magnitude sqrt(gx^2 + gy^2) from the histogram of gx^2 + gy^2"""

    def __init__(self, m2, counts, n_pixels):
        self.mag = np.sqrt(m2.astype(np.float64))
        self.counts = counts
        self.n = n_pixels
        self._mean = None
        self._std = None
        self._pct = {}            # the threshold search asks for the same percentiles / moments at five sensitivities: computed once

    def mean(self):
        if self._mean is None:
            self._mean = float(np.dot(self.mag, self.counts)) / self.n
        return self._mean

    def std(self):
        if self._std is None:
            d = self.mag - self.mean()
            self._std = math.sqrt(float(np.dot(d * d, self.counts)) / self.n)
        return self._std

    def percentile_nonzero(self, q):
        if q not in self._pct:
            nz = self.mag > 0
            self._pct[q] = _percentile(self.mag[nz], self.counts[nz], q) if nz.any() else None
        return self._pct[q]


class EdgeAnalysis:
    """everything about one image that the threshold search reuses"""

    def __init__(self, image, rh=None):
        import torch
        self.rh = rh or default_context()
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self.image = image
        dev = torch.from_numpy(np.array(image, dtype=np.uint8, order="C")).to(self.rh.device)
        if image.ndim == 3:
            self.rgb = dev
            self.gray, self.hist = self.rh.edges_gray(dev)
        else:
            self.rgb = None
            self.gray = dev
            self.hist = np.bincount(image.ravel(), minlength=256).astype(np.int64)
        self.H, self.W = image.shape[:2]
        self._grad = None
        self._nm = {}
        self._otsu_val = None

    def gradient(self):
        if self._grad is None:
            m2, c = self.rh.edges_grad_hist(self.gray)
            self._grad = _Gradient(m2, c, self.H * self.W)
        return self._grad

    def otsu(self):
        if self._otsu_val is None:
            self._otsu_val = _otsu(self.hist)
        return self._otsu_val

    def nm(self, colour=False):
        if colour not in self._nm:
            self._nm[colour] = self.rh.canny_nms(self.rgb if colour else self.gray)
        return self._nm[colour]

    def thresholds(self, method="otsu", sensitivity=1.0):
        """edges.py:88-169"""
        if method == "otsu":
            o = self.otsu()
            lo, hi = max(10, int(o * 0.5 * sensitivity)), min(255, int(o * 1.5 * sensitivity))
        elif method == "percentile":
            g = self.gradient()
            p70 = g.percentile_nonzero(70)
            if p70 is not None:
                lp, hp = p70 * sensitivity, g.percentile_nonzero(90) * sensitivity
            else:
                lp, hp = 50 * sensitivity, 150 * sensitivity
            lo, hi = max(10, int(lp)), min(255, int(hp))
        elif method == "gradient":
            g = self.gradient()
            mean, std = g.mean(), g.std()
            lo, hi = max(10, int((mean - 0.5 * std) * sensitivity)), min(255, int((mean + 0.5 * std) * sensitivity))
        elif method == "hybrid":
            o = self.otsu()
            mean = self.gradient().mean()
            lo = max(10, int((o * 0.5 + mean * 0.5) * sensitivity))
            hi = min(255, int((o * 1.5 + mean * 1.0) * sensitivity))
        else:
            lo, hi = 50, 150
        lo = max(10, min(200, lo))
        hi = max(lo + 10, min(255, hi))
        return lo, hi

    @staticmethod
    def _norm(low, high):
        low, high = math.floor(low), math.floor(high)
        return (high, low) if low > high else (low, high)

    def _score_of(self, four):
        n_comp, n_edge, s1, s2 = four
        if n_comp == 0:
            return float("nan"), 0.0
        contrast = math.sqrt(n_edge * s2 - s1 * s1) / n_edge       # population standard deviation from exact integer sums
        return (n_edge / n_comp) * contrast, n_edge / (self.H * self.W)

    def score(self, low, high):
        """evaluate_edge_quality(cv2.Canny(gray, low, high), gray) without the edge map: (score, fraction of edge pixels)"""
        low, high = self._norm(low, high)
        return self._score_of(self.rh.canny_components(self.nm(False), low, max(high, 0), self.gray)[2])

    def scores(self, pairs):
        """{(low, high): score(low, high)} for several threshold pairs: pairs that share `low` share the labelling of {nm > low} and the
        per-label reduction; only the verdict against `high` (32 bytes back) is repeated"""
        pairs = list(pairs)
        norm = [(self._norm(lo, hi)[0], max(self._norm(lo, hi)[1], 0)) for lo, hi in pairs]
        uniq = sorted(set(norm))
        # all pairs in ONE call and ONE read-back (rhccq_canny_scores: the component counts stay on the device); round 3 paid two small
        # synchronous read-backs per pair, ~16 ms of a 29 ms stage at 4K
        fours = dict(zip(uniq, self.rh.canny_scores(self.nm(False), self.gray, uniq)))
        return {p: self._score_of(fours[n]) for p, n in zip(pairs, norm)}

    def canny(self, low, high, colour=False):
        """cv2.Canny(gray or colour image, low, high) -> uint8[H,W] device (0 / 255)"""
        low, high = math.floor(low), math.floor(high)
        if low > high:
            low, high = high, low
        labels, lut, _ = self.rh.canny_components(self.nm(colour), low, max(high, 0), None, want_lut=True)
        return self.rh.ccl_select(labels, lut)


# ---- the reference's functions ----------------------------------------------------------------------------------------------
def compute_adaptive_canny_thresholds(image, method="otsu", sensitivity=1.0):
    return EdgeAnalysis(image).thresholds(method, sensitivity)


def _best_thresholds(a, debug=False):
    """the threshold search of edges.py:40-71 on one EdgeAnalysis: -> (low, high, method)"""
    best_score, best = -1, None
    grid = [(method, sensitivity) + a.thresholds(method, sensitivity) for method in ("otsu", "percentile", "gradient", "hybrid")
            for sensitivity in (0.5, 0.7, 1.0, 1.3, 1.5)]
    cache = a.scores({(lo, hi) for _, _, lo, hi in grid})
    for method, sensitivity, lo, hi in grid:                  # the reference's iteration order: the first best score wins
        score, density = cache[(lo, hi)]
        if debug:
            print(f"{method} (sens: {sensitivity}): ({lo}, {hi}) -> density: {density:.4f}, score: {score:.4f}")
        if score > best_score:
            best_score, best = score, (lo, hi, method)
    if best is None:
        raise UnboundLocalError("find_best_edges_by_quality: no threshold pair produced an edge")
    return best


def find_best_edges_by_quality(image_rgb, debug=False):
    a = EdgeAnalysis(image_rgb)
    lo, hi, method = _best_thresholds(a, debug)
    return a.canny(lo, hi).cpu().numpy(), lo, hi, method


def edge_map_resident(image_rgb, rh=None):
    """get_edge_map with the result left on the device: -> (EdgeAnalysis, uint8[H,W] device plane, 0 / 255)"""
    a = EdgeAnalysis(image_rgb, rh)
    lo, hi, _ = _best_thresholds(a)
    return a, a.canny(lo, hi, colour=True)


def get_edge_map(image_rgb):
    return edge_map_resident(image_rgb)[1].cpu().numpy()


def evaluate_edge_quality(edges, gray):
    import torch
    rh = default_context()
    e = torch.from_numpy(np.ascontiguousarray(np.asarray(edges) > 0).view(np.uint8)).to(rh.device)
    g = torch.from_numpy(np.ascontiguousarray(gray, dtype=np.uint8)).to(rh.device)
    n, labels, stats = rh.ccl(e, 8, cap=1 << 16)
    if n == 0:
        return float("nan")
    red = rh.label_reduce(labels, n, None, g)
    n_edge = int(stats[1:, 4].sum())
    s1, s2 = int(red[1:, 1].sum()), int(red[1:, 2].sum())
    return (n_edge / n) * (math.sqrt(n_edge * s2 - s1 * s1) / n_edge)


def compute_local_density(binary_map, kernel_size=15):
    """-> float32[H,W].  The integer window counts come from the device; kernels up to 11 x 11 take OpenCV's direct filter path, a
    float32 accumulator over the taps (= the count-fold sequential float32 sum of 1 / k^2); larger ones its DFT path, restated as
    float32(count) * float32(1 / k^2) (the DFT's own rounding noise, ~1e-7, has no closed form).  Maps with several non-zero values:
    window SUMS of the values on the device for the DFT-path sizes, the literal tap loop on the host for the small ones."""
    import torch
    rh = default_context()
    bm = np.asarray(binary_map)
    nz = bm[bm != 0]
    kernel = np.ones((kernel_size, kernel_size), np.float32)
    kernel /= kernel.sum()
    k2 = kernel_size * kernel_size
    if nz.size and (nz != nz.flat[0]).any():
        if k2 >= 130 and bm.dtype == np.uint8:
            sums = rh.box_sum(torch.from_numpy(np.ascontiguousarray(bm)).to(rh.device), kernel_size).cpu().numpy().astype(np.float64)
            return (sums / 255.0 if bm.max() > 1 else sums).astype(np.float32) * kernel[0, 0]
        m = bm / 255.0 if bm.max() > 1 else bm
        src = np.pad(m.astype(np.float32), kernel_size // 2, mode="reflect")
        H, W = bm.shape
        if k2 >= 130:
            acc = np.zeros((H, W), np.float64)
            for dy in range(kernel_size):
                for dx in range(kernel_size):
                    acc += src[dy:dy + H, dx:dx + W]
            return acc.astype(np.float32) * kernel[0, 0]
        acc = np.zeros((H, W), np.float32)
        for dy in range(kernel_size):
            for dx in range(kernel_size):
                acc = (acc + kernel[dy, dx] * src[dy:dy + H, dx:dx + W]).astype(np.float32)
        return acc
    value = 1.0
    if nz.size:
        value = float(nz.flat[0]) / 255.0 if bm.max() > 1 else float(nz.flat[0])
    term = np.float32(kernel[0, 0] * np.float32(value))
    cnt = rh.box_count(torch.from_numpy(np.ascontiguousarray(bm != 0).view(np.uint8)).to(rh.device), kernel_size)
    if k2 < 130:
        table = np.zeros(k2 + 1, np.float32)
        for m in range(1, k2 + 1):
            table[m] = np.float32(table[m - 1] + term)
    else:
        table = (np.arange(k2 + 1, dtype=np.float32) * term).astype(np.float32)
    d_table = rh.dev(table)
    out = rh.empty(tuple(cnt.shape), torch.float32)
    rh._check(rh.lib.rhccq_lut_u16_f32(rh.ctx, rh._p(cnt), rh._p(d_table), len(table), cnt.numel(), rh._p(out)), "lut_u16_f32")
    return out.cpu().numpy()


def suggest_automatic_threshold(edge_density_map, edge_map, method="mean"):
    vals = edge_density_map[edge_map > 0]
    if len(vals) == 0:
        return 0.1
    if method == "median":
        return np.median(vals)
    if method == "percentile":
        return np.percentile(vals, 70)
    return np.mean(vals)


def compute_fast_canny_thresholds(gray, method="percentile_fast", _analysis=None):
    """edges.py:235-298"""
    a = _analysis or EdgeAnalysis(gray)
    if method == "percentile_fast":
        v = np.flatnonzero(a.hist)
        low_val, high_val = _percentile(v, a.hist[v], 25), _percentile(v, a.hist[v], 75)
        low = max(10, min(100, int(low_val * 0.7)))
        high = max(50, min(200, int(high_val * 1.3)))
        if high < low * 2:
            high = low * 2
        if high > 255:
            high = 255
    elif method == "gradient_fast":
        raise NotImplementedError("compute_fast_canny_thresholds('gradient_fast'): no caller in the reference")
    else:
        raise NotImplementedError("compute_fast_canny_thresholds('hybrid_fast'): no caller in the reference")
    low = max(10, min(100, low))
    high = max(low * 2, min(200, high))
    return low, high


def get_edge_map_fast(image_rgb):
    """edges.py:200-233"""
    a = EdgeAnalysis(image_rgb)
    low, high = compute_fast_canny_thresholds(None, "percentile_fast", _analysis=a)
    g = a.gradient()
    p10 = g.percentile_nonzero(10)
    if p10 is not None:
        low = int((low + p10) / 2)
        high = int((high + g.percentile_nonzero(90)) / 2)
    return a.canny(low, high).cpu().numpy()
