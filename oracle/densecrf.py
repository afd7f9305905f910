"""CPU restatement of the per-line DenseCRF of the reference's mask refinement.  TEST INFRASTRUCTURE ONLY: imported by
tests/, never by the product path.

The reference calls pydensecrf (``refine_mask``, /root/reference/manga_translator/mask_refinement/text_mask_utils.py:68-94):

    unary = unary_from_softmax([1 - m, m])                     # -log(clip(p, 1e-5, 1))
    d = DenseCRF2D(w, h, 2); d.setUnaryEnergy(unary)
    d.addPairwiseGaussian(sxy=1, compat=3, DIAG_KERNEL, NO_NORMALIZATION)
    d.addPairwiseBilateral(sxy=23, srgb=7, rgbim=img, compat=20, DIAG_KERNEL, NO_NORMALIZATION)
    Q = d.inference(5); mask = argmax(Q) * 255

pydensecrf is a third-party dependency whose source is NOT under /root/reference (requirements.txt: ``pydensecrf`` from
git, lucasb-eyer/pydensecrf, which wraps Philipp Kraehenbuehl's densecrf 2013 release) and it is installed nowhere this
can run — **parity unpinned**.  This file restates the published algorithm of that library: mean-field inference
(Kraehenbuehl & Koltun, NIPS 2011) with Potts compatibilities, the message passing done by a permutohedral-lattice
filter (Adams, Baek & Davis, Eurographics 2010) exactly as densecrf's ``permutohedral.cpp`` does it — elevate, round to the
nearest remainder-0 point, rank, barycentric weights, splat to the d+1 simplex vertices, one [1/2, 1, 1/2] blur along each
of the d+1 lattice axes, slice with the 1 / (1 + 2^-d) factor — in fp32 with the library's evaluation order.  What pins
it: properties in tests/test_densecrf.py (the lattice filter against a brute-force Gaussian, symmetry, a hand-checked
single-point lattice), and the HIP implementation is compared with THIS file."""
from __future__ import annotations

import numpy as np

f32 = np.float32
KEY_BITS = 12  # lattice coordinates are packed 12 bits each into one int64 (asserted)


def unary_from_mask(mask_u8: np.ndarray) -> np.ndarray:
    """text_mask_utils.py:74-79 + pydensecrf.utils.unary_from_softmax(sm, clip=1e-5) -> [2, N] float32."""
    m = mask_u8.reshape(-1)
    sm = np.stack([(255 - m), m]).astype(f32) / f32(255.0)  # cv2.bitwise_not(rawmask), rawmask
    return (-np.log(np.clip(sm, f32(1e-5), f32(1.0)))).astype(f32)


def features_gaussian(h: int, w: int, sxy: float) -> np.ndarray:
    """DenseCRF2D::addPairwiseGaussian: feature (x / sx, y / sy) per pixel, pixel index y * W + x -> [2, N] float32."""
    ys, xs = np.mgrid[0:h, 0:w]
    return np.stack([xs.reshape(-1).astype(f32) / f32(sxy), ys.reshape(-1).astype(f32) / f32(sxy)])


def features_bilateral(img: np.ndarray, sxy: float, srgb: float) -> np.ndarray:
    """DenseCRF2D::addPairwiseBilateral: (x / sx, y / sy, r / sr, g / sg, b / sb) -> [5, N] float32."""
    h, w, _ = img.shape
    g = features_gaussian(h, w, sxy)
    c = img.reshape(-1, 3).astype(f32).T / f32(srgb)
    return np.concatenate([g, c]).astype(f32)


class Permutohedral:
    """densecrf's Permutohedral::init + seqCompute for a [d, N] feature matrix."""

    def __init__(self, feature: np.ndarray):
        feature = np.ascontiguousarray(feature, dtype=f32)
        d, N = feature.shape
        self.d, self.N = d, N
        inv_std_dev = f32(np.sqrt(2.0 / 3.0) * (d + 1))
        scale = np.array([f32(1.0 / np.sqrt(float((i + 2) * (i + 1))) * float(inv_std_dev)) for i in range(d)], dtype=f32)
        # elevate: y = E p
        elevated = np.empty((d + 1, N), dtype=f32)
        sm = np.zeros(N, dtype=f32)
        for j in range(d, 0, -1):
            cf = feature[j - 1] * scale[j - 1]
            elevated[j] = sm - f32(j) * cf
            sm = sm + cf
        elevated[0] = sm
        # closest remainder-0 point
        down, up = f32(1.0) / f32(d + 1), f32(d + 1)
        v = down * elevated
        upv, dnv = np.ceil(v) * up, np.floor(v) * up
        rem0 = np.where(upv - elevated < elevated - dnv, upv, dnv).astype(f32)
        ssum = np.rint(rem0 * down).astype(np.int64).sum(0)  # integer-valued terms: the library's int accumulation is exact
        # rank of each coordinate's residual
        diff = elevated - rem0
        rank = np.zeros((d + 1, N), dtype=np.int64)
        for i in range(d):
            for j in range(i + 1, d + 1):
                lt = diff[i] < diff[j]
                rank[i] += lt
                rank[j] += ~lt
        rank += ssum[None, :]
        lo, hi = rank < 0, rank > d
        rank = np.where(lo, rank + d + 1, np.where(hi, rank - (d + 1), rank))
        rem0 = np.where(lo, rem0 + up, np.where(hi, rem0 - up, rem0)).astype(f32)
        # barycentric weights
        bary = np.zeros((d + 2, N), dtype=f32)
        cols = np.arange(N)
        for i in range(d + 1):
            vv = (elevated[i] - rem0[i]) * down
            bary[d - rank[i], cols] += vv
            bary[d - rank[i] + 1, cols] -= vv
        bary[0] = (1.0 + bary[d + 1].astype(np.float64) + bary[0].astype(np.float64)).astype(f32)
        self.barycentric = np.ascontiguousarray(bary[:d + 1].T)  # [N, d+1]
        # simplex vertices -> lattice points (hash table == unique keys)
        canonical = np.empty((d + 1, d + 1), dtype=np.int64)
        for i in range(d + 1):
            canonical[i, :d - i + 1] = i
            canonical[i, d - i + 1:] = i - (d + 1)
        rem_i = rem0.astype(np.int64)
        keys = np.empty((N, d + 1, d), dtype=np.int64)
        for r in range(d + 1):
            for i in range(d):
                keys[:, r, i] = rem_i[i] + canonical[r, rank[i]]
        packed = self._pack(keys.reshape(-1, d))
        self.keys, inv = np.unique(packed, return_inverse=True)
        self.offset = inv.reshape(N, d + 1)
        self.M = len(self.keys)
        # blur neighbours along each of the d+1 axes (-1 = absent)
        kd = self._unpack(self.keys)
        self.n1 = np.empty((d + 1, self.M), dtype=np.int64)
        self.n2 = np.empty((d + 1, self.M), dtype=np.int64)
        for j in range(d + 1):
            a, b = kd - 1, kd + 1
            if j < d:
                a[:, j] = kd[:, j] + d
                b[:, j] = kd[:, j] - d
            self.n1[j] = self._find(self._pack(a))
            self.n2[j] = self._find(self._pack(b))

    def _bits(self) -> int:
        return 24 if self.d <= 2 else KEY_BITS   # the position-only kernel of a page-tall crop needs more than 12 bits per coordinate

    def _pack(self, k: np.ndarray) -> np.ndarray:
        bits = self._bits()
        half = 1 << (bits - 1)
        assert k.min() >= -half and k.max() < half, "lattice coordinate outside the packed-key range"
        out = np.zeros(len(k), dtype=np.int64)
        for i in range(k.shape[1]):
            out = (out << bits) | (k[:, i] + half)
        return out

    def _unpack(self, p: np.ndarray) -> np.ndarray:
        bits = self._bits()
        half = 1 << (bits - 1)
        out = np.empty((len(p), self.d), dtype=np.int64)
        for i in range(self.d - 1, -1, -1):
            out[:, i] = (p & ((1 << bits) - 1)) - half
            p = p >> bits
        return out

    def _find(self, p: np.ndarray) -> np.ndarray:
        pos = np.clip(np.searchsorted(self.keys, p), 0, self.M - 1)
        return np.where(self.keys[pos] == p, pos, -1)

    def compute(self, inp: np.ndarray) -> np.ndarray:
        """seqCompute(out, in, value_size): inp [N, V] float32 -> [N, V] float32 (no normalisation)."""
        d, N, M = self.d, self.N, self.M
        V = inp.shape[1]
        values = np.zeros((M + 2, V), dtype=f32)
        # splat in the library's order (point-major, vertex-minor): np.add.at applies the updates sequentially in fp32
        idx = (self.offset + 1).reshape(-1)
        contrib = (self.barycentric[:, :, None] * inp[:, None, :]).astype(f32).reshape(-1, V)
        np.add.at(values, idx, contrib)
        for j in range(d + 1):
            n1, n2 = self.n1[j] + 1, self.n2[j] + 1
            new = np.zeros_like(values)
            nb = (values[n1] + values[n2]).astype(f32)
            new[1:M + 1] = (values[1:M + 1].astype(np.float64) + 0.5 * nb.astype(np.float64)).astype(f32)
            values = new
        alpha = f32(1.0) / (f32(1.0) + f32(np.power(f32(2.0), f32(-d))))
        out = np.zeros((N, V), dtype=f32)
        for j in range(d + 1):
            out += (self.barycentric[:, j, None] * values[self.offset[:, j] + 1]) * alpha
        return out


def _exp_and_normalize(x: np.ndarray) -> np.ndarray:
    """DenseCRF::expAndNormalize on [M, N]: column-wise softmax with the max subtracted."""
    e = np.exp((x - x.max(0, keepdims=True)).astype(f32)).astype(f32)
    return (e / e.sum(0, keepdims=True, dtype=f32)).astype(f32)


def inference(unary: np.ndarray, kernels, n_iterations: int = 5, trace=None) -> np.ndarray:
    """DenseCRF::inference: ``kernels`` = [(Permutohedral, potts_weight)]; returns Q [M, N] float32."""
    q = _exp_and_normalize(-unary)
    for it in range(n_iterations):
        tmp1 = -unary
        for lat, wgt in kernels:
            tmp2 = (-f32(wgt)) * lat.compute(np.ascontiguousarray(q.T)).T  # PottsCompatibility::apply
            tmp1 = (tmp1 - tmp2).astype(f32)
        q = _exp_and_normalize(tmp1)
        if trace is not None:
            trace.append(q.copy())
    return q


def refine_mask(rgbimg: np.ndarray, rawmask: np.ndarray, n_iterations: int = 5, return_q: bool = False):
    """text_mask_utils.refine_mask (:71-94): rgb crop [h, w, 3] u8 + mask crop [h, w] u8 -> u8 mask in {0, 255}."""
    if rawmask.ndim == 3:
        rawmask = rawmask[:, :, 0]
    h, w = rgbimg.shape[:2]
    unary = unary_from_mask(rawmask)
    kernels = [(Permutohedral(features_gaussian(h, w, 1)), 3.0), (Permutohedral(features_bilateral(rgbimg, 23, 7)), 20.0)]
    q = inference(unary, kernels, n_iterations)
    res = (np.argmax(q, axis=0).reshape(h, w) * 255).astype(np.uint8)
    return (res, q) if return_q else res


def gaussian_filter_bruteforce(feature: np.ndarray, inp: np.ndarray) -> np.ndarray:
    """What the lattice approximates: out_i = sum_j exp(-|f_i - f_j|^2 / 2) in_j  (float64, O(N^2), small N only)."""
    f = feature.astype(np.float64).T
    d2 = ((f[:, None, :] - f[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.5 * d2) @ inp.astype(np.float64)
