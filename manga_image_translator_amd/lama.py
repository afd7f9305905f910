"""LaMa-MPE / LaMa-large inpainting generator on the gfx950 engine.

Same network as ``FFCResNetGenerator`` + ``MPE`` of the reference
(/root/reference/manga_translator/inpainting/inpainting_lama_mpe.py:545-632,713-726), laid out for
MI355X:

* activations fp32 NHWC; the bottleneck state is ONE [B,h,w,512] tensor whose first 128 channels
  are the local branch and last 384 the global branch, so ``convl2l(x_l) + convg2l(x_g)`` (:365) is a
  single 3x3 conv over 512 input channels and torch.cat / ConcatTupleLayer (:535-542) cost nothing;
* every BatchNorm is folded into the producing conv's epilogue, ReLU / sigmoid / residual adds too;
* FourierUnit's rfftn / irfftn (:228,252) run as LDS butterflies: a mixed-radix real FFT along W
  (mit_rfft_rows / mit_irfft_rows, 182 = 2*7*13 for the BASELINE page) and a radix-2 complex FFT along H
  (mit_fft_cols); sizes those kernels do not cover fall back to dense DFT GEMMs on the MFMA kernel.  The
  re/im planes stay planar so the 384->384 spectral 1x1 conv reads them as two "taps" and no
  permute/stack/complex copy exists;
* uint8 pages in, uint8 pages out: only bytes cross PCIe.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, PAD_REFLECT, PAD_ZERO, conv_gemm_desc, launch_conv_gemm, tensor_map

LOCAL_C, GLOBAL_C, SPEC_C = 128, 384, 192
MPE_S = 256


def _bn(sd, prefix, eps=1e-5):
    return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"], sd[prefix + ".running_var"], eps)


def _cat_bn(sd, p1, p2):
    a, b = _bn(sd, p1), _bn(sd, p2)
    return tuple(torch.cat([x, y]) for x, y in zip(a[:4], b[:4])) + (a[4],)


def _round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------
# DFT matrices (float64 on the host, rounded once to fp32)
# ------------------------------------------------------------------------------------------

def dft_matrices(h: int, w: int):
    """Dense real matrices of the ortho-normalised rfft2 / irfft2 over an [h, w] grid.

    F1   [2*wk, wp]  : row (t,kw), col w     real->complex DFT along W (t=0 re, t=1 im), wp = w padded to 16
    G2   [2h, 2h->pad16] : row (t',kh), col (t,h) complex DFT along H on planar re/im
    G2i  [2h, 2h->pad16] : row (t,h), col (t',kh) inverse complex DFT along H
    Fi   [w, 2*wk->pad16] : row w, col (t,kw)  complex->real inverse along W (Hermitian weights; the
                       imaginary parts of the DC / Nyquist bins are ignored exactly like pocketfft's c2r)
    """
    wk = w // 2 + 1
    wp = _round_up(w, 16)  # K of the GEMM: a multiple of the K-tile keeps it on conv_gemm's fast path
    kw = np.arange(wk)[:, None]
    xs = np.arange(w)[None, :]
    ang = 2.0 * np.pi * ((kw * xs) % w) / w
    sw = 1.0 / math.sqrt(w)
    F1 = np.zeros((2 * wk, wp), dtype=np.float64)
    F1[:wk, :w] = np.cos(ang) * sw
    F1[wk:, :w] = -np.sin(ang) * sw
    kh = np.arange(h)[:, None]
    ys = np.arange(h)[None, :]
    angh = 2.0 * np.pi * ((kh * ys) % h) / h
    sh = 1.0 / math.sqrt(h)
    Gr, Gi = np.cos(angh) * sh, -np.sin(angh) * sh
    k2p = _round_up(2 * h, 16)
    G2 = np.zeros((2 * h, k2p), dtype=np.float64)
    G2[:, :2 * h] = np.block([[Gr, -Gi], [Gi, Gr]])
    Gri, Gii = np.cos(angh) * sh, np.sin(angh) * sh
    G2i = np.zeros((2 * h, k2p), dtype=np.float64)
    G2i[:, :2 * h] = np.block([[Gri, -Gii], [Gii, Gri]])
    kp = _round_up(2 * wk, 16)
    a = np.full(wk, 2.0)
    a[0] = 1.0
    if w % 2 == 0:
        a[wk - 1] = 1.0
    Fi = np.zeros((w, kp), dtype=np.float64)
    Fi[:, :wk] = (np.cos(ang) * sw * a[:, None]).T
    sin_t = np.sin(ang)
    sin_t[0, :] = 0.0
    if w % 2 == 0:
        sin_t[wk - 1, :] = 0.0
    Fi[:, wk:2 * wk] = (-sin_t * sw * a[:, None]).T
    f32 = lambda m: torch.from_numpy(np.ascontiguousarray(m.astype(np.float32)))
    return f32(F1), f32(G2), f32(G2i), f32(Fi)


def rfft_row_tables(w: int) -> torch.Tensor:
    """Twiddle tables mit_rfft_rows / mit_irfft_rows consume for an even row length w: (cos, sin)(2 pi j / (w/2)), j < w/2,
    then (cos, sin)(2 pi k / w), k <= w/2 — float64 on the host, rounded once to fp32."""
    n = w // 2
    a = np.concatenate([2.0 * np.pi * np.arange(n) / n, 2.0 * np.pi * np.arange(n + 1) / w])
    return torch.from_numpy(np.stack([np.cos(a), np.sin(a)], 1).astype(np.float32)).contiguous()


# ------------------------------------------------------------------------------------------
# host tables for the MPE index kernels (same formulas as OpenCV's resize)
# ------------------------------------------------------------------------------------------

def _area_taps(n_src: int, n_dst: int):
    """cv2.INTER_AREA taps per destination index (area mean when shrinking, the INTER_AREA flavour of
    bilinear otherwise) -> (start[int32], count[int32], weights[float64, n_dst x maxtaps])."""
    scale = n_src / n_dst
    rows = []
    for d in range(n_dst):
        if n_dst <= n_src:
            lo, hi = d * scale, (d + 1) * scale
            s0, s1 = int(math.floor(lo)), min(int(math.ceil(hi)), n_src)
            ws = [max(0.0, min(hi, s + 1) - max(lo, s)) / scale for s in range(s0, s1)]
        else:
            inv = n_dst / n_src
            s0 = int(math.floor(d * scale))
            f = (d + 1) - (s0 + 1) * inv
            f = 0.0 if f <= 0 else f - math.floor(f)
            if s0 >= n_src - 1:
                s0, f = n_src - 1, 0.0
            ws = [1.0 - f] + ([f] if f > 0 else [])
        rows.append((s0, ws))
    maxt = max(len(ws) for _, ws in rows)
    start = np.array([s for s, _ in rows], dtype=np.int32)
    cnt = np.array([len(ws) for _, ws in rows], dtype=np.int32)
    wts = np.zeros((n_dst, maxt), dtype=np.float64)
    for d, (_, ws) in enumerate(rows):
        wts[d, :len(ws)] = ws
    return start, cnt, wts, maxt


def _nearest_map(n_dst: int, n_src: int) -> np.ndarray:
    """cv2.INTER_NEAREST source index per destination index."""
    return np.minimum(np.floor(np.arange(n_dst) * (n_src / n_dst)).astype(np.int64), n_src - 1).astype(np.int32)


class _RowPackedStem:
    """The 7x7 4->64 stem (ReflectionPad2d(3) + Conv2d + BN + ReLU, :560-563) on the 32-bit-offset fast GEMM kernel.

    With Cin = 4 the implicit GEMM's K-tiles straddle taps, which sends the layer to the generic kernel (84 TFLOP/s).  On a
    reflect-padded input [B, H+6, W+8, 4] a kernel ROW is one contiguous read of 8 pixels x 4 channels = 32 floats, so the layer
    becomes 7 taps of "Cin = 32" with pixel stride 4: K = 224 instead of 196 (the 8th pixel meets zero weights), whole 16-float
    K-tiles, no padding logic in the gather.  The accumulation order (ky, kx, c) is the generic kernel's, and the extra terms are
    exact zeros, so the result is bit-identical to ``ops.Conv2d`` with reflect padding."""

    K, PAD, WIN = 7, 3, 8

    def __init__(self, conv: "ops.Conv2d", weight: torch.Tensor, device):
        Cout, Cin, kh, kw = weight.shape
        if (kh, kw) != (self.K, self.K) or Cin > 4:
            raise ValueError("row-packed stem: 7x7 kernels with <= 4 input channels only")
        w = torch.zeros(self.K, self.WIN, 4, Cout, dtype=torch.float32)   # [ky][pixel][c][n]
        w[:, :self.K, :Cin] = weight.detach().to(torch.float32).permute(2, 3, 1, 0)
        self.w, self.Kp, self.Np = ops.pack_weight_kn(w.reshape(self.K * self.WIN * 4, Cout), device)
        self.Cout, self.scale, self.bias, self.act = Cout, conv.scale, conv.bias, conv.act

    def padded_shape(self, B, H, W):
        return (B, H + 2 * self.PAD, W + self.WIN, 4)

    def __call__(self, xp: torch.Tensor, out: torch.Tensor, lut: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
        """``lut`` = (rows int32 [B*H*W], table1 [r1, Cout], table2 [r2, Cout]): the epilogue's two-table row lookup
        (MitConvGemm.lut_rows) — out = act(bn(conv)) + table1[rows & 0xffff] + table2[rows >> 16], added in that order."""
        B, Hp, Wp, _ = xp.shape
        H, W = Hp - 2 * self.PAD, Wp - self.WIN
        d = conv_gemm_desc(
            a=xp, NB=B, Hi=Hp, Wi=W, Cin=self.WIN * 4, a_strides=(Hp * Wp * 4, Wp * 4, 4), Ho=H, Wo=W, sy=1, sx=1,
            taps=[(ky, 0, 0) for ky in range(self.K)], pad_mode=PAD_ZERO, w=self.w, ldw=self.Np, Kw=self.Kp, Nw=self.Np,
            N=self.Cout, c=tensor_map(out), scale=self.scale, bias=self.bias, act=self.act)
        if lut is not None:
            rows, t1, t2 = lut
            if rows.dtype != torch.int32 or rows.numel() != B * H * W or t1.shape[1] != self.Cout or t2.shape[1] != self.Cout:
                raise ValueError("row-packed stem: bad lookup tables")
            d.lut_rows, d.lut1, d.lut2, d.lut_ld = rows.data_ptr(), t1.data_ptr(), t2.data_ptr(), self.Cout
        launch_conv_gemm(d)


class _FFC:
    """One FFC_BN_ACT of a res-block (:372-399): packed layers."""

    def __init__(self, sd, p, device, winograd=True):
        w_l = torch.cat([sd[p + ".ffc.convl2l.weight"], sd[p + ".ffc.convg2l.weight"]], dim=1)  # [128, 512, 3, 3]
        self.winograd = winograd
        if winograd:  # F(4x4, 3x3): both convs read the same transformed input (convl2g its first 128 channels)
            self.to_l = ops.WinogradConv3x3(w_l, None, pad_mode=PAD_REFLECT, bn=_bn(sd, p + ".bn_l"), act=ACT_RELU, device=device)
            self.l2g = ops.WinogradConv3x3(sd[p + ".ffc.convl2g.weight"], None, pad_mode=PAD_REFLECT, device=device)
        else:
            self.to_l = ops.Conv2d(w_l, None, padding=1, pad_mode=PAD_REFLECT, bn=_bn(sd, p + ".bn_l"), act=ACT_RELU,
                                   device=device)
            self.l2g = ops.Conv2d(sd[p + ".ffc.convl2g.weight"], None, padding=1, pad_mode=PAD_REFLECT, device=device)
        st = p + ".ffc.convg2g"
        self.st_in = ops.Conv2d(sd[st + ".conv1.0.weight"], None, bn=_bn(sd, st + ".conv1.1"), act=ACT_RELU, device=device)
        # spectral 1x1 conv: reference channel index is c*2 + t (:229-231,245-246); ours is planar t*C + c
        wf = sd[st + ".fu.conv_layer.weight"].reshape(SPEC_C, 2, SPEC_C, 2)  # [c_out, t_out, c_in, t_in]
        wf = wf.permute(3, 2, 1, 0).reshape(2 * SPEC_C, 2 * SPEC_C)  # [(t_in, c_in), (t_out, c_out)]
        self.fu_w, self.fu_Kp, self.fu_Np = ops.pack_weight_kn(wf, device)
        g, b, m, v, eps = _bn(sd, st + ".fu.bn")
        perm = lambda t: t.reshape(SPEC_C, 2).t().reshape(-1)
        sc, bi = ops.fold_bn(perm(g), perm(b), perm(m), perm(v), eps)
        self.fu_scale, self.fu_bias = sc.to(device), bi.to(device)
        # conv2 (192->384) carries the global branch's BN + ReLU (+ pre = convl2g(x_l), + residual)
        self.st_out = ops.Conv2d(sd[st + ".conv2.weight"], None, bn=_bn(sd, p + ".bn_g"), act=ACT_RELU, device=device)


class LamaEngine:
    """Batched LaMa generator. ``forward(img_u8[B,H,W,3], mask_u8[B,H,W]) -> u8 [B,H,W,3]`` (device tensors)."""

    def __init__(self, gen_sd: Dict[str, torch.Tensor], mpe_sd: Optional[Dict[str, torch.Tensor]] = None,
                 n_blocks: int = 9, device="cuda", fft_h: bool = True, winograd: bool = True, fft_w: bool = True,
                 row_packed_stem: bool = True):
        self.device = torch.device(device)
        self.winograd = winograd  # False: the FFC blocks' 3x3 convolutions in direct (9-tap) form, for A/B comparison
        self.fft_h = fft_h  # False: keep the H-axis transform on the dense DFT GEMM (for A/B comparison)
        self.fft_w = fft_w  # False: keep the W-axis transform on the dense DFT GEMM (for A/B comparison)
        self.n_blocks = n_blocks
        sd, dev = gen_sd, self.device
        self.stem = ops.Conv2d(sd["model.1.ffc.convl2l.weight"], None, padding=3, pad_mode=PAD_REFLECT,
                               bn=_bn(sd, "model.1.bn_l"), act=ACT_RELU, device=dev)
        self.stem_packed = _RowPackedStem(self.stem, sd["model.1.ffc.convl2l.weight"], dev) if row_packed_stem else None
        self.down1 = ops.Conv2d(sd["model.2.ffc.convl2l.weight"], None, stride=2, padding=1, pad_mode=PAD_REFLECT,
                                bn=_bn(sd, "model.2.bn_l"), act=ACT_RELU, device=dev)
        self.down2 = ops.Conv2d(sd["model.3.ffc.convl2l.weight"], None, stride=2, padding=1, pad_mode=PAD_REFLECT,
                                bn=_bn(sd, "model.3.bn_l"), act=ACT_RELU, device=dev)
        w3 = torch.cat([sd["model.4.ffc.convl2l.weight"], sd["model.4.ffc.convl2g.weight"]], dim=0)  # 256 -> 128+384
        self.down3 = ops.Conv2d(w3, None, stride=2, padding=1, pad_mode=PAD_REFLECT,
                                bn=_cat_bn(sd, "model.4.bn_l", "model.4.bn_g"), act=ACT_RELU, device=dev)
        self.blocks = [(_FFC(sd, f"model.{5 + i}.conv1", dev, winograd), _FFC(sd, f"model.{5 + i}.conv2", dev, winograd))
                       for i in range(n_blocks)]
        base = 5 + n_blocks + 1
        self.ups = []
        for i in range(3):
            p = base + 3 * i
            self.ups.append(ops.ConvTranspose2d(sd[f"model.{p}.weight"], sd[f"model.{p}.bias"], stride=2, padding=1,
                                                output_padding=1, bn=_bn(sd, f"model.{p + 1}"), act=ACT_RELU, device=dev))
        p = base + 10
        self.out_conv = ops.ConvSmallCout(sd[f"model.{p}.weight"], sd[f"model.{p}.bias"], pad_mode=PAD_REFLECT,
                                          act=ACT_SIGMOID, device=dev)
        self.mpe = None
        if mpe_sd is not None:
            self.mpe = dict(emb=mpe_sd["rel_pos_emb.weight"].to(torch.float32).to(dev).contiguous(),
                            dirw=mpe_sd["direct_emb.weight"].to(torch.float32).to(dev).contiguous(),
                            alpha5=float(mpe_sd["alpha5"]), alpha6=float(mpe_sd["alpha6"]))
            # The two adds of FFCResNetGenerator.forward (:611-612) as lookup tables for the stem's epilogue, built with the arithmetic of
            # mpe_add_kernel (fp32, same order): t1[rel] = emb[rel] * alpha5; t2[bits] = (0 + w[k0] + w[k1] + …, set bits ascending) * alpha6
            emb, dirw = self.mpe["emb"].cpu(), self.mpe["dirw"].cpu()
            a5, a6 = torch.tensor(self.mpe["alpha5"], dtype=torch.float32), torch.tensor(self.mpe["alpha6"], dtype=torch.float32)
            t2 = torch.zeros(16, dirw.shape[1], dtype=torch.float32)
            for bits in range(16):
                acc = torch.zeros(dirw.shape[1], dtype=torch.float32)
                for k in range(4):
                    if bits & (1 << k):
                        acc = acc + dirw[k]
                t2[bits] = acc * a6
            self.mpe["lut1"], self.mpe["lut2"] = (emb * a5).contiguous().to(dev), t2.contiguous().to(dev)
        # layout between the last up-convolution and the 7x7 output convolution: 4 = sixteen 4-channel planes (the output convolution's
        # LDS-DMA kernel, round 6), 16 (or 1) = four 16-channel planes (its register-staged kernel, round 5), 0 = NHWC — same values
        pt = os.environ.get("MIT_LAMA_PLANAR_TAIL", "4")
        self.planar_tail = {"": 0, "0": 0, "1": 16, "16": 16, "4": 4}.get(pt, 4)
        self.mpe_in_stem = os.environ.get("MIT_LAMA_MPE_SEPARATE", "") in ("", "0")   # False: the separate mit_lama_mpe_add pass (A/B, tests)
        self._ws = ops.Workspace(self.device)
        self._tw: Dict[int, torch.Tensor] = {}
        self._tw_rows: Dict[int, torch.Tensor] = {}
        self._dft = ops.ShapeCache(4)       # per-(h, w) DFT matrices: a few page shapes stay resident, older ones are dropped
        self._mpe_tabs = ops.ShapeCache(4)  # per-(H, W) resize tap tables of the MPE index kernels

    # -- workspace -------------------------------------------------------------------------
    def _buf(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        """Named workspace slab, grown to the largest request (ops.Workspace): memory is bounded by the largest page seen."""
        return self._ws.buf(name, *shape, dtype=dtype)

    def release_workspace(self):
        self._ws.release()
        self._dft.clear()
        self._mpe_tabs.clear()

    def _dft_mats(self, h, w):
        return self._dft.get((h, w), lambda: tuple(m.to(self.device) for m in dft_matrices(h, w)))

    def _twiddles(self, h):
        """(cos, sin)(2 pi k / h), k < h/2, rounded once from float64 — the table mit_fft_cols consumes."""
        if h not in self._tw:
            ang = 2.0 * np.pi * np.arange(h // 2) / h
            self._tw[h] = torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)).to(self.device).contiguous()
        return self._tw[h]

    def _row_tables(self, w):
        if w not in self._tw_rows:
            self._tw_rows[w] = rfft_row_tables(w).to(self.device)
        return self._tw_rows[w]

    def _fft_h(self, src, dst, dst_strides, B, h, ncols, plane, inverse):
        """H-axis complex FFT on planar re/im [B,2,h,ncols] (src) -> dst with (batch, plane, row) strides ``dst_strides``."""
        _lib.check(_lib.load().mit_fft_cols(src.data_ptr(), 2 * plane, plane, ncols, dst.data_ptr(), *dst_strides,
                                            self._twiddles(h).data_ptr(), B, h, ncols, int(inverse), 1.0 / math.sqrt(h),
                                            C.c_void_p(ops.current_stream())), "mit_fft_cols")

    def _mpe_tables(self, H, W):
        def make():
            ys, yc, yw, ymax = _area_taps(H, MPE_S)
            xs, xc, xw, xmax = _area_taps(W, MPE_S)
            dev = self.device
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            return dict(ys=t(ys), yc=t(yc), yw=t(yw), ymax=ymax, xs=t(xs), xc=t(xc), xw=t(xw), xmax=xmax,
                        ymap=t(_nearest_map(H, MPE_S)), xmap=t(_nearest_map(W, MPE_S)))

        return self._mpe_tabs.get((H, W), make)

    # -- FourierUnit (:214-257): LDS-butterfly FFTs (dense DFT GEMMs for the sizes they do not cover) ---------------------------------------------------
    _dbg: Optional[dict] = None  # diagnostics only (scripts/diag_ffc_load.py): intermediate tensors of the FIRST FFC layer of a forward

    def _dbg_put(self, name: str, t: torch.Tensor):
        if self._dbg is not None and name not in self._dbg:
            self._dbg[name] = t.clone()

    def _fourier_unit(self, ffc: _FFC, t1: torch.Tensor, t2: torch.Tensor):
        """t2 = t1 + irfft2(relu(bn(conv1x1(rfft2(t1)))))   (x + fu(x), :305)."""
        B, h, w, Cc = t1.shape
        wk = w // 2 + 1
        plane = h * wk * Cc
        Y = self._buf("fu_Y", B, 2, h, wk, Cc)
        Zf = self._buf("fu_Z", B, 2, h, wk, Cc)
        Z2 = self._buf("fu_Z2", B, 2, h, wk, Cc)
        U = self._buf("fu_U", B, h, 2, wk, Cc)
        one = [(0, 0, 0)]
        # algorithmic cost of one real 2-D FFT of [h, w] per channel: 2.5 * h*w * log2(h*w) (half of a complex 5 N log2 N);
        # each of its two GEMM launches is credited half of it in the roofline probe instead of its dense-DFT FLOPs
        fft_half = 0.5 * 2.5 * h * w * math.log2(h * w) * Cc * B
        tag = lambda: _lib.load().mit_prof_tag_next(fft_half)
        # S1: real DFT along W: mixed-radix LDS butterflies when w is an even product of small primes (W/8 = 182 = 2*7*13 for
        # the BASELINE page), else the dense DFT GEMM  rows (t,kw) = F1 @ t1[b,h] ([w] x [C]);  z = (b, h)
        lib = _lib.load()
        use_fft_w = self.fft_w and Cc % 4 == 0 and bool(lib.mit_rfft_rows_supported(w))
        use_fft = h >= 2 and (h & (h - 1)) == 0 and h <= 512 and self.fft_h
        if not (use_fft_w and use_fft):
            F1, G2, G2i, Fi = self._dft_mats(h, w)
        st = C.c_void_p(ops.current_stream())
        if use_fft_w:
            _lib.check(lib.mit_rfft_rows(t1.data_ptr(), h * w * Cc, w * Cc, Cc, Y.data_ptr(), 2 * plane, plane, wk * Cc, Cc,
                                         self._row_tables(w).data_ptr(), B, h, w, Cc, 1.0 / math.sqrt(w), st), "mit_rfft_rows")
        else:
            cm = ops.MitTensorMap()
            cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = Y.data_ptr(), 2 * plane, wk * Cc, 0, plane, Cc
            tag()
            launch_conv_gemm(conv_gemm_desc(
                a=F1, NB=1, Hi=2, Wi=wk, Cin=F1.shape[1], a_strides=(0, wk * F1.shape[1], F1.shape[1]), Ho=2, Wo=wk, sy=1,
                sx=1, taps=one, pad_mode=PAD_ZERO, w=t1, ldw=Cc, Kw=w, Nw=Cc, N=Cc, c=cm, Z=B * h, zdiv=h,
                w_zs=(h * w * Cc, w * Cc)))
        self._dbg_put("fu1_rfft_rows", Y)
        # S2: complex DFT along H on planar re/im: LDS-butterfly FFT when h is a power of two (H/8 = 256 for the BASELINE
        # page), else the dense [2h x 2h] DFT GEMM  Z[b] = G2 @ Y[b]
        if use_fft:
            self._fft_h(Y, Zf, (2 * plane, plane, wk * Cc), B, h, wk * Cc, plane, False)
        else:
            cm = ops.MitTensorMap()
            cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = Zf.data_ptr(), 0, 2 * plane, 0, 0, wk * Cc
            tag()
            launch_conv_gemm(conv_gemm_desc(
                a=G2, NB=1, Hi=1, Wi=2 * h, Cin=G2.shape[1], a_strides=(0, 0, G2.shape[1]), Ho=1, Wo=2 * h, sy=1, sx=1, taps=one,
                pad_mode=PAD_ZERO, w=Y, ldw=wk * Cc, Kw=2 * h, Nw=wk * Cc, N=wk * Cc, c=cm, Z=B, zdiv=1 << 30,
                w_zs=(0, 2 * plane)))
        self._dbg_put("fu2_fft_cols", Zf)
        # spectral 1x1 conv + BN + ReLU: two taps = re plane, im plane; planar output via the column split
        cm = ops.MitTensorMap()
        cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = Z2.data_ptr(), 0, 0, 2 * plane, wk * Cc, Cc
        cm.nsplit, cm.nhi = Cc, plane
        launch_conv_gemm(conv_gemm_desc(
            a=Zf, NB=B, Hi=h, Wi=wk, Cin=Cc, a_strides=(2 * plane, wk * Cc, Cc), Ho=h, Wo=wk, sy=1, sx=1,
            taps=[(0, 0, 0), (0, 0, plane)], pad_mode=PAD_ZERO, w=ffc.fu_w, ldw=ffc.fu_Np, Kw=ffc.fu_Kp, Nw=ffc.fu_Np,
            N=2 * Cc, c=cm, scale=ffc.fu_scale, bias=ffc.fu_bias, act=ACT_RELU))
        self._dbg_put("fu3_spectral_conv", Z2)
        # S3: inverse complex DFT along H.  U[b,h,t] rows (t,h) = G2i @ Z2[b]
        if use_fft:
            self._fft_h(Z2, U, (2 * plane, wk * Cc, 2 * wk * Cc), B, h, wk * Cc, plane, True)
        else:
            cm = ops.MitTensorMap()
            cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = U.data_ptr(), 0, 2 * plane, 0, wk * Cc, 2 * wk * Cc
            tag()
            launch_conv_gemm(conv_gemm_desc(
                a=G2i, NB=1, Hi=2, Wi=h, Cin=G2i.shape[1], a_strides=(0, h * G2i.shape[1], G2i.shape[1]), Ho=2, Wo=h, sy=1, sx=1, taps=one,
                pad_mode=PAD_ZERO, w=Z2, ldw=wk * Cc, Kw=2 * h, Nw=wk * Cc, N=wk * Cc, c=cm, Z=B, zdiv=1 << 30,
                w_zs=(0, 2 * plane)))
        self._dbg_put("fu4_ifft_cols", U)
        # S4: complex->real inverse DFT along W, + t1.  t2[b,h] = Fi @ U[b,h] ([2wk] x [C]) + t1[b,h]
        if use_fft_w:
            _lib.check(lib.mit_irfft_rows(U.data_ptr(), 2 * plane, wk * Cc, 2 * wk * Cc, Cc, t2.data_ptr(), h * w * Cc, w * Cc,
                                          Cc, t1.data_ptr(), h * w * Cc, w * Cc, Cc, self._row_tables(w).data_ptr(), B, h, w, Cc,
                                          1.0 / math.sqrt(w), st), "mit_irfft_rows")
            return
        cm = ops.MitTensorMap()
        cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = t2.data_ptr(), 0, w * Cc, 0, 0, Cc
        pm = ops.MitTensorMap()
        pm.base, pm.zs1, pm.zs0, pm.bs, pm.ys, pm.xs = t1.data_ptr(), 0, w * Cc, 0, 0, Cc
        tag()
        launch_conv_gemm(conv_gemm_desc(
            a=Fi, NB=1, Hi=1, Wi=w, Cin=Fi.shape[1], a_strides=(0, 0, Fi.shape[1]), Ho=1, Wo=w, sy=1, sx=1, taps=one,
            pad_mode=PAD_ZERO, w=U, ldw=Cc, Kw=2 * wk, Nw=Cc, N=Cc, c=cm, post=pm, Z=B * h, zdiv=1 << 30,
            w_zs=(0, 2 * wk * Cc)))

    def _ffc(self, ffc: _FFC, x: torch.Tensor, out: torch.Tensor, residual: Optional[torch.Tensor]):
        """FFC_BN_ACT.forward (:395-399 over :349-369) on the fused [B,h,w,512] state."""
        B, h, w, _ = x.shape
        x_l, x_g = x[..., :LOCAL_C], x[..., LOCAL_C:]
        res_l = None if residual is None else residual[..., :LOCAL_C]
        res_g = None if residual is None else residual[..., LOCAL_C:]
        P = self._buf("ffc_P", B, h, w, GLOBAL_C)
        if ffc.winograd:
            T = ops.WinogradConv3x3.tiles(B, h, w)
            V = self._buf("wino_v", 36, T, LOCAL_C + GLOBAL_C)
            ffc.to_l.transform_input(x, V)
            self._dbg_put("w1_wino_input", V)
            ffc.to_l.gemm_output(V, self._buf("wino_ml", 36, T, LOCAL_C), out[..., :LOCAL_C], post=res_l)
            self._dbg_put("w2_wino_products_l", self._buf("wino_ml", 36, T, LOCAL_C))
            self._dbg_put("w3_out_local", out[..., :LOCAL_C])
            ffc.l2g.gemm_output(V, self._buf("wino_mg", 36, T, GLOBAL_C), P)
            self._dbg_put("w4_wino_products_g", self._buf("wino_mg", 36, T, GLOBAL_C))
            self._dbg_put("w5_P", P)
        else:
            ffc.to_l(x, out=out[..., :LOCAL_C], post=res_l)  # convl2l(x_l) + convg2l(x_g) -> bn_l -> relu (+ id_l)
            ffc.l2g(x_l, out=P)  # convl2g(x_l), raw
        t1 = self._buf("ffc_t1", B, h, w, SPEC_C)
        t2 = self._buf("ffc_t2", B, h, w, SPEC_C)
        ffc.st_in(x_g, out=t1)  # SpectralTransform.conv1 (:272-277)
        self._dbg_put("s1_st_in", t1)
        self._fourier_unit(ffc, t1, t2)
        self._dbg_put("fu5_irfft_rows_plus_t1", t2)
        ffc.st_out(t2, out=out[..., LOCAL_C:], pre=P, post=res_g)
        self._dbg_put("s2_out_global", out[..., LOCAL_C:])  # conv2(x + fu(x)) + convl2g -> bn_g -> relu (+ id_g)

    # -- full generator ------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, mask_u8: torch.Tensor, taps: Optional[dict] = None, composite: bool = True) -> torch.Tensor:
        """LamaFourier.__call__ (:713-726) + the tensor pre/post of _infer (:82-117) for pages whose
        H, W are multiples of 8.  img_u8 [B,H,W,3] u8, mask_u8 [B,H,W] u8 (device) -> u8 [B,H,W,3].
        ``composite=False`` returns ``img_inpainted`` of :111 (the network's bytes everywhere) instead of the final composite
        of :117 — the plugin resizes that back to the page size first when the page was resized (:112-117)."""
        if img_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8:
            raise TypeError("LamaEngine.forward expects uint8 page and mask tensors")
        if img_u8.dim() != 4 or img_u8.shape[-1] != 3 or tuple(mask_u8.shape) != tuple(img_u8.shape[:3]):
            raise ValueError(f"bad shapes: page {tuple(img_u8.shape)}, mask {tuple(mask_u8.shape)}")
        B, H, W, _ = img_u8.shape
        if H % 8 or W % 8:
            raise ValueError("LamaEngine.forward: H and W must be multiples of 8 (the plugin resizes first, :67-79)")
        img_u8, mask_u8 = img_u8.contiguous(), mask_u8.contiguous()
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        s64 = self._buf("full64", B, H, W, 64)
        packed = self.stem_packed is not None and H > 3 and W > 3
        lut = None
        if self.mpe is not None:   # the masked position encoding's index maps first: the stem's epilogue consumes them
            tb = self._mpe_tables(H, W)
            hole = self._buf("mpe_hole", B, MPE_S, MPE_S, dtype=torch.uint8)
            rel = self._buf("mpe_rel", B, MPE_S, MPE_S, dtype=torch.uint8)
            dr = self._buf("mpe_dir", B, MPE_S, MPE_S, dtype=torch.uint8)
            _lib.check(lib.mit_lama_mpe_index(mask_u8.data_ptr(), B, H, W, tb["ys"].data_ptr(), tb["yc"].data_ptr(),
                                              tb["yw"].data_ptr(), tb["ymax"], tb["xs"].data_ptr(), tb["xc"].data_ptr(),
                                              tb["xw"].data_ptr(), tb["xmax"], hole.data_ptr(), rel.data_ptr(),
                                              dr.data_ptr(), st), "mit_lama_mpe_index")
            if packed and self.mpe_in_stem:
                rows = self._buf("mpe_rows", B * H * W, dtype=torch.int32)
                _lib.check(lib.mit_lama_mpe_rows(mask_u8.data_ptr(), rel.data_ptr(), dr.data_ptr(), tb["ymap"].data_ptr(),
                                                 tb["xmap"].data_ptr(), rows.data_ptr(), B, H, W, st), "mit_lama_mpe_rows")
                lut = (rows, self.mpe["lut1"], self.mpe["lut2"])
        if packed:
            xp = self._buf("in4p", *self.stem_packed.padded_shape(B, H, W))
            _lib.check(lib.mit_lama_prep_padded(img_u8.data_ptr(), mask_u8.data_ptr(), xp.data_ptr(), B, H, W, 3, xp.shape[2], st),
                       "mit_lama_prep_padded")
            self.stem_packed(xp, s64, lut=lut)   # x_l += rel_pos; x_l += direct (:611-612) inside the stem's epilogue
        else:  # row_packed_stem=False: the plain reflect-padded Conv2d on the generic kernel (A/B comparison)
            x4 = self._buf("in4", B, H, W, 4)
            _lib.check(lib.mit_lama_prep(img_u8.data_ptr(), mask_u8.data_ptr(), x4.data_ptr(), B, H, W, st), "mit_lama_prep")
            self.stem(x4, out=s64)
        if self.mpe is not None:
            if lut is None:   # the separate pass (MIT_LAMA_MPE_SEPARATE=1, or the unpacked stem): bit-identical to the epilogue form
                _lib.check(lib.mit_lama_mpe_add(s64.data_ptr(), mask_u8.data_ptr(), rel.data_ptr(), dr.data_ptr(),
                                                tb["ymap"].data_ptr(), tb["xmap"].data_ptr(), self.mpe["emb"].data_ptr(),
                                                self.mpe["dirw"].data_ptr(), self.mpe["alpha5"], self.mpe["alpha6"], B, H, W,
                                                st), "mit_lama_mpe_add")
            if taps is not None:
                taps["mpe_rel"], taps["mpe_dir"], taps["mpe_hole"] = rel.clone(), dr.clone(), hole.clone()
        if taps is not None:
            taps["stem"] = s64.clone()
        d1 = self._buf("d1", B, H // 2, W // 2, 128)
        self.down1(s64, out=d1)
        d2 = self._buf("d2", B, H // 4, W // 4, 256)
        self.down2(d1, out=d2)
        h, w = H // 8, W // 8
        X = self._buf("X", B, h, w, 512)
        T = self._buf("Xtmp", B, h, w, 512)
        self.down3(d2, out=X)
        if taps is not None:
            taps["down"] = X.clone()
        for i, (c1, c2) in enumerate(self.blocks):  # FFCResnetBlock.forward :421-436
            self._ffc(c1, X, T, None)
            self._ffc(c2, T, X, X)  # in place: each element reads its own residual before it is overwritten
            if taps is not None:
                taps[f"block{i}"] = X.clone()
        u1 = self._buf("d2", B, H // 4, W // 4, 256)
        self.ups[0](X, out=u1)
        u2 = self._buf("d1", B, H // 2, W // 2, 128)
        self.ups[1](u1, out=u2)
        pred = self._buf("pred", B, H, W, 3)
        if self.planar_tail:   # the last up-convolution writes 64 / P planes of P channels, the 7x7 output convolution reads them slice by slice
            P = self.planar_tail
            pm = P == 4 and H % 2 == 0 and W % 2 == 0   # 4-channel planes as four dense parity sub-images: both sides move consecutive pixels
            u3 = self._buf("full64", 64 // P, B, H, W, P)
            self.ups[2](u2, out=u3, planes=64 // P, parity_major=pm)
            self.out_conv(u3, out=pred, parity_major=pm)
        else:                  # MIT_LAMA_PLANAR_TAIL=0: NHWC between the two (A/B, tests) — same values, other addresses
            u3 = self._buf("full64", B, H, W, 64)
            self.ups[2](u2, out=u3)
            self.out_conv(u3, out=pred)
        if taps is not None:
            taps["pred"] = pred.clone()
        out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=self.device)
        _lib.check(lib.mit_lama_post(pred.data_ptr(), 3, img_u8.data_ptr(), mask_u8.data_ptr(), out.data_ptr(), B, H, W,
                                     int(composite), st), "mit_lama_post")
        return out

    # algorithmic FLOPs of one page (0.9706 MFLOP per input pixel for 9 blocks)
    def flops_per_page(self, H: int, W: int) -> float:
        px = H * W
        conv = 2.0 * px * (49 * 4 * 64 + 49 * 64 * 3)  # stem + out
        conv += 2.0 * (px / 4) * 9 * 64 * 128 + 2.0 * (px / 16) * 9 * 128 * 256 + 2.0 * (px / 64) * 9 * 256 * 512
        per_ffc = 2.0 * (px / 64) * (9 * (128 * 128 + 384 * 128 + 128 * 384) + 384 * 192 + 192 * 384)
        per_ffc += 2.0 * (px / 64) * ((W // 8 // 2 + 1) / (W // 8)) * 384 * 384
        conv += per_ffc * 2 * self.n_blocks
        conv += 2.0 * (px / 64) * 9 * 512 * 256 + 2.0 * (px / 16) * 9 * 256 * 128 + 2.0 * (px / 4) * 9 * 128 * 64
        return conv
