"""Host-side operators over the C-ABI: weight packing, descriptors, launches.

PyTorch is used only for device memory and streams (``torch.Tensor.data_ptr()``,
``torch.cuda.current_stream()``); every FLOP runs in ``libmit_hip.so``.

Activations are fp32 NHWC tensors ``[B, H, W, C]`` (any strides as long as the channel
stride is 1 — channel slices and spatially strided views are passed straight to the kernel,
which is how concat and the sub-pixel form of ConvTranspose2d are expressed without copies).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as _lib
from .lib import (ACT_GELU, ACT_LEAKY, ACT_NONE, ACT_POST_FIRST, ACT_RELU, ACT_SIGMOID, ACT_SILU, MIT_MAX_TAPS, PAD_REFLECT,
                  PAD_ZERO, MitConvGemm, MitPGemm, MitTensorMap)

__all__ = [
    "ACT_NONE", "ACT_RELU", "ACT_LEAKY", "ACT_SILU", "ACT_SIGMOID", "ACT_GELU", "ACT_POST_FIRST", "PAD_ZERO", "PAD_REFLECT",
    "Conv2d", "ConvSmallCout", "split_mode", "set_split_mode", "gemm_mode", "ConvTranspose2d", "UpsampleConv2d", "fold_bn", "conv_gemm_desc", "launch_conv_gemm", "current_stream",
]


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Workspace:
    """Named grow-only slabs: ``buf(name, *shape)`` returns a view of the slab ``(name, dtype)``, which is reallocated only
    when a larger request arrives.  Device memory therefore stays bounded by the LARGEST page seen, not by the number of
    distinct page shapes (all users of an engine are ordered on one stream, so a slab can be re-viewed at a new shape by the
    next call; the caching allocator keeps a replaced slab alive until the work queued on it has run)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._slabs = {}

    def buf(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        key = (name, dtype)
        t = self._slabs.get(key)
        if t is None or t.numel() < max(n, 1):
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._slabs[key] = t
        return t[:n].view(*shape)

    def release(self) -> None:
        self._slabs.clear()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._slabs.values())


class ShapeCache:
    """Tiny LRU for per-(H, W) device tables (DFT matrices, resize taps): at most ``capacity`` page shapes stay resident."""

    def __init__(self, capacity: int = 4):
        self.capacity, self._d = capacity, {}

    def get(self, key, make):
        if key in self._d:
            self._d[key] = self._d.pop(key)  # most recently used last
            return self._d[key]
        v = make()
        self._d[key] = v
        while len(self._d) > self.capacity:
            self._d.pop(next(iter(self._d)))
        return v

    def clear(self):
        self._d.clear()

    def __len__(self):
        return len(self._d)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check_nhwc(t: torch.Tensor, what: str) -> None:
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")
    if t.dim() != 4:
        raise ValueError(f"{what}: expected [B,H,W,C], got shape {tuple(t.shape)}")
    if t.shape[-1] > 1 and t.stride(-1) != 1:
        raise ValueError(f"{what}: channel stride must be 1, got strides {t.stride()}")


def tensor_map(t: Optional[torch.Tensor], zs1: int = 0, zs0: int = 0, nsplit: int = 0, nhi: int = 0) -> MitTensorMap:
    """MitTensorMap of a ``[NB, Ho, Wo, N]`` view (or a disabled map for ``None``)."""
    m = MitTensorMap()
    if t is None:
        m.base = None
        return m
    _check_nhwc(t, "tensor_map")
    m.base = t.data_ptr()
    m.zs1, m.zs0 = zs1, zs0
    m.bs, m.ys, m.xs = t.stride(0), t.stride(1), t.stride(2)
    m.nsplit, m.nhi = nsplit, nhi
    return m


def conv_gemm_desc(*, a: torch.Tensor, NB: int, Hi: int, Wi: int, Cin: int, a_strides: Tuple[int, int, int],
                   Ho: int, Wo: int, sy: int, sx: int, taps: Sequence[Tuple[int, int, int]], pad_mode: int,
                   w: torch.Tensor, ldw: int, Kw: int, Nw: int, N: int, c: MitTensorMap,
                   pre: Optional[MitTensorMap] = None, post: Optional[MitTensorMap] = None,
                   scale: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                   alpha: float = 0.0, Z: int = 1, zdiv: int = 1, a_zs: Tuple[int, int] = (0, 0),
                   w_zs: Tuple[int, int] = (0, 0)) -> MitConvGemm:
    """Fill a ``MitConvGemm`` descriptor. Pure host logic (usable without a GPU)."""
    if len(taps) == 0 or len(taps) > MIT_MAX_TAPS:
        raise ValueError(f"ntaps {len(taps)} out of range")
    d = MitConvGemm()
    d.a = a.data_ptr()
    d.a_zs1, d.a_zs0 = a_zs
    d.a_bs, d.a_ys, d.a_xs = a_strides
    d.NB, d.Hi, d.Wi, d.Cin = NB, Hi, Wi, Cin
    d.Ho, d.Wo, d.sy, d.sx = Ho, Wo, sy, sx
    d.ntaps, d.pad_mode = len(taps), pad_mode
    for i, (dy, dx, off) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i], d.tap_off[i] = dy, dx, off
    d.w = w.data_ptr()
    d.w_zs1, d.w_zs0 = w_zs
    d.ldw, d.Kw, d.Nw = ldw, Kw, Nw
    if _SPLITS:
        planes, zs = _split_for(w, ldw, Kw, w_zs)
        if planes is not None:
            d.w_split, d.ws_zs0 = planes.data_ptr(), zs
    d.N, d.Z, d.zdiv = N, Z, zdiv
    d.c = c
    d.pre = pre if pre is not None else tensor_map(None)
    d.post = post if post is not None else tensor_map(None)
    d.scale = _ptr(scale)
    d.bias = _ptr(bias)
    d.act, d.act_alpha = act, alpha
    return d


def launch_conv_gemm(desc: MitConvGemm, cfg: int = -1, stream: Optional[int] = None) -> None:
    lib = _lib.load()
    s = current_stream() if stream is None else stream
    _lib.check(lib.mit_conv_gemm_cfg(C.byref(desc), cfg, C.c_void_p(s)), "mit_conv_gemm")


def fold_bn(gamma: torch.Tensor, beta: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, eps: float,
            conv_bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode BatchNorm2d after a conv as per-channel (scale, bias): y = conv*scale + bias.

    Same algebra as the reference's own fuse step (ctd_utils/utils/yolov5_utils.py:22-42), but
    the scale stays in the epilogue instead of being multiplied into the weights.
    """
    g, b, m, v = (t.detach().to(torch.float64) for t in (gamma, beta, mean, var))
    scale = g / torch.sqrt(v + eps)
    bias = b - m * scale
    if conv_bias is not None:
        bias = bias + conv_bias.detach().to(torch.float64) * scale
    return scale.to(torch.float32), bias.to(torch.float32)


def split_mode() -> int:
    """The GEMM mode of ``mit_conv_gemm`` (include/mit_hip.h, ``mit_gemm_mode_get``): 6 (default) | 9 = packers also build the
    three-bf16-plane form of their weights and the automatic tile choice takes the split-bf16 tiles for them; 0 = fp32 MFMA tiles
    only.  The initial value comes from ``MIT_GEMM_SPLIT`` in the environment when set."""
    return int(_lib.load().mit_gemm_mode_get())  # the 3-pair tiles are a test ladder (explicit tile index only), never a mode


def set_split_mode(mode: int) -> int:
    """Switch the GEMM mode at run time; returns the previous one.  Weights packed while the mode was 0 carry no planes and stay on the
    fp32 tiles; weights packed in mode 6 | 9 carry them and follow the mode of the moment (so one engine can be timed in both)."""
    lib = _lib.load()
    prev = int(lib.mit_gemm_mode_get())
    _lib.check(lib.mit_gemm_mode_set(int(mode)), "mit_gemm_mode_set")
    return prev


class gemm_mode:
    """``with ops.gemm_mode(0): ...`` — the GEMM mode (and, optionally, the smallest launch the split tiles take, in 128 x 64 tiles)
    inside the block, the previous values restored after it."""

    def __init__(self, mode: int, min_tiles: Optional[int] = None):
        self.mode, self.min_tiles, self.prev, self.prev_min = int(mode), min_tiles, None, None

    def __enter__(self):
        self.prev = set_split_mode(self.mode)
        if self.min_tiles is not None:
            self.prev_min = int(_lib.load().mit_gemm_split_min_tiles(int(self.min_tiles)))
        return self

    def __exit__(self, *exc):
        set_split_mode(self.prev)
        if self.prev_min is not None:
            _lib.load().mit_gemm_split_min_tiles(self.prev_min)
        return False


_SPLITS: Dict[int, Tuple["weakref.ref", torch.Tensor, int, int, int]] = {}   # data_ptr -> (weight, planes, nz, Kp, Np)


def split_weight(w: torch.Tensor) -> torch.Tensor:
    """fp32 ``[Kp, Np]`` or ``[nz, Kp, Np]`` (contiguous, on the GPU) -> int16 planes ``[nz, 3, Kp / 8, Np, 8]`` with
    ``w == hi + mid + lo`` exactly (bf16 bit patterns), laid out as conv_gemm_split_kernel stages them."""
    if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() not in (2, 3):
        raise ValueError("split_weight: contiguous fp32 [Kp, Np] or [nz, Kp, Np] expected")
    nz = 1 if w.dim() == 2 else w.shape[0]
    Kp, Np = w.shape[-2], w.shape[-1]
    if Kp % 8 or Np % 4:
        raise ValueError(f"split_weight: Kp % 8 == 0 and Np % 4 == 0 needed (got {Kp} x {Np})")
    out = torch.empty(nz, 3, Kp // 8, Np, 8, dtype=torch.int16, device=w.device)
    _lib.check(_lib.load().mit_gemm_split_pack(w.data_ptr(), Kp * Np, nz, Kp, Np, out.data_ptr(), C.c_void_p(current_stream())),
               "mit_gemm_split_pack")
    return out


def register_split(w: torch.Tensor, force: bool = False) -> Optional[torch.Tensor]:
    """Attach split planes to a packed weight (no-op unless ``split_mode()`` or ``force``): ``conv_gemm_desc`` finds them by the
    tensor's identity, so every layer built on ``pack_weight_kn`` / ``WinogradConv3x3`` gets the split tiles without further plumbing."""
    if w.device.type != "cuda" or not (force or split_mode()):
        return None
    planes = split_weight(w)
    nz = 1 if w.dim() == 2 else w.shape[0]
    key = w.data_ptr()
    _SPLITS[key] = (weakref.ref(w, lambda _r, k=key: _SPLITS.pop(k, None)), planes, nz, w.shape[-2], w.shape[-1])
    return planes


def _split_for(w: torch.Tensor, ldw: int, Kw: int, w_zs: Tuple[int, int]):
    e = _SPLITS.get(w.data_ptr())
    if e is None or e[0]() is not w or (e[3], e[4]) != (Kw, ldw) or w_zs[0] != 0 or (e[2] > 1 and w_zs[1] != Kw * ldw):
        return None, 0
    return e[1], (3 * Kw * ldw if e[2] > 1 else 0)


# ---- planar operands (include/mit_hip.h "planar operands", csrc/pgemm.hip) ---------------------------------------------------------
def split_planes(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 ``[R, K]`` (row stride free, K % 8 == 0) -> int16 planes ``[3, K / 8, R, 8]`` with x == hi + mid + lo exactly."""
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1 or x.shape[1] % 8:
        raise ValueError("split_planes: fp32 [R, K] with unit column stride and K % 8 == 0 expected")
    R, K = x.shape
    if out is None:
        out = torch.empty(3, K // 8, R, 8, dtype=torch.int16, device=x.device)
    elif out.dtype != torch.int16 or tuple(out.shape) != (3, K // 8, R, 8) or not out.is_contiguous():
        raise ValueError("split_planes: out must be a contiguous int16 [3, K / 8, R, 8]")
    _lib.check(_lib.load().mit_split_planes(x.data_ptr(), x.stride(0), R, K, out.data_ptr(), R, C.c_void_p(current_stream())), "mit_split_planes")
    return out


def join_planes(planes: torch.Tensor) -> torch.Tensor:
    """int16 planes ``[3, K / 8, R, 8]`` -> fp32 ``[R, K]`` (the exact sum of the three bf16 planes)."""
    if planes.dtype != torch.int16 or planes.dim() != 4 or planes.shape[0] != 3 or planes.shape[3] != 8 or not planes.is_contiguous():
        raise ValueError("join_planes: contiguous int16 [3, K / 8, R, 8] expected")
    _, K8, R, _ = planes.shape
    out = torch.empty(R, K8 * 8, dtype=torch.float32, device=planes.device)
    _lib.check(_lib.load().mit_join_planes(planes.data_ptr(), R, R, K8 * 8, out.data_ptr(), K8 * 8, C.c_void_p(current_stream())), "mit_join_planes")
    return out


def pgemm_tile(name: str) -> int:
    """Index of a ``mit_pgemm`` tile by name (``mit_pgemm_tile_name``)."""
    L, i = _lib.load(), 0
    while True:
        n = L.mit_pgemm_tile_name(i)
        if n is None:
            raise KeyError(name)
        if n.decode() == name:
            return i
        i += 1


def pgemm(a_planes: torch.Tensor, w: torch.Tensor, N: int, *, out: Optional[torch.Tensor] = None, out_planes: Optional[torch.Tensor] = None,
          pre: Optional[torch.Tensor] = None, post: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
          bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, alpha: float = 0.0, nprod: int = 0, tile: int = -1):
    """``mit_pgemm``: C = epilogue(A @ W) with A given as planes ``[3, K / 8, M, 8]`` (split_planes, or a planar producer) and W a packed
    weight carrying split planes (pack_weight_kn / register_split).  Output fp32 ``out [M, N]`` (row stride free) or ``out_planes
    [3, N / 8, M, 8]``.  Bit-identical to the split-bf16 tiles of mit_conv_gemm on the same operands."""
    if a_planes.dtype != torch.int16 or a_planes.dim() != 4 or a_planes.shape[0] != 3 or a_planes.shape[3] != 8 or not a_planes.is_contiguous():
        raise ValueError("pgemm: a_planes must be a contiguous int16 [3, K / 8, M, 8]")
    _, K8, M, _ = a_planes.shape
    e = _SPLITS.get(w.data_ptr())
    if e is None or e[0]() is not w or w.dim() != 2:
        raise ValueError("pgemm: w must be a 2-D packed weight with registered split planes")
    Kp, Np = w.shape
    if Kp != K8 * 8 or N > Np:
        raise ValueError(f"pgemm: operand shapes do not match (A has K = {K8 * 8}, W is {Kp} x {Np}, N = {N})")
    d = MitPGemm()
    d.a_planes, d.a_zs, d.lda = a_planes.data_ptr(), 0, M
    d.w_planes, d.w_zs, d.ldw = e[1].data_ptr(), 0, Np
    d.M, d.N, d.K, d.Z = M, N, Kp, 1
    if (out is None) == (out_planes is None):
        raise ValueError("pgemm: exactly one of out / out_planes")
    keep = [a_planes, e[1]]
    if out is not None:
        if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or out.stride(1) != 1:
            raise ValueError("pgemm: out must be fp32 [M, N] with unit column stride")
        d.c, d.ldc = out.data_ptr(), out.stride(0)
        for t, nm in ((pre, "pre"), (post, "post")):
            if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (M, N) or t.stride(1) != 1):
                raise ValueError(f"pgemm: {nm} must be fp32 [M, N] with unit column stride")
        if pre is not None:
            d.pre, d.ld_pre = pre.data_ptr(), pre.stride(0)
        if post is not None:
            d.post, d.ld_post = post.data_ptr(), post.stride(0)
    else:
        if out_planes.dtype != torch.int16 or tuple(out_planes.shape) != (3, N // 8, M, 8) or N % 8 or not out_planes.is_contiguous():
            raise ValueError("pgemm: out_planes must be a contiguous int16 [3, N / 8, M, 8] (N % 8 == 0)")
        if pre is not None or post is not None:
            raise ValueError("pgemm: pre / post are not available with planar output")
        d.c_planes, d.ld_cp = out_planes.data_ptr(), M
    d.scale, d.bias = _ptr(scale), _ptr(bias)
    d.act, d.act_alpha, d.nprod, d.tile = act, alpha, nprod, tile
    _lib.check(_lib.load().mit_pgemm(C.byref(d), C.c_void_p(current_stream())), "mit_pgemm")
    return out if out is not None else out_planes


def pack_weight_kn(w_kn: torch.Tensor, device) -> Tuple[torch.Tensor, int, int]:
    """Zero-pad a [K, N] matrix to [Kp(16), Np(4)] fp32 on ``device``; returns (w, Kp, Np)."""
    K, N = w_kn.shape
    Kp, Np = _round_up(K, 16), _round_up(N, 4)
    out = torch.zeros(Kp, Np, dtype=torch.float32)
    out[:K, :N] = w_kn.detach().to(torch.float32)
    out = out.to(device).contiguous()
    register_split(out)
    return out, Kp, Np


@dataclass
class _Packed:
    w: torch.Tensor
    Kp: int
    Np: int
    taps: List[Tuple[int, int, int]]


class Conv2d:
    """nn.Conv2d (+ folded eval BatchNorm2d + activation) as one ``mit_conv_gemm`` launch.

    weight: [Cout, Cin, kh, kw] (torch layout).  Input channels are zero-padded to a multiple
    of 4 (``cin_pad``); the input tensor must then carry that many channels.
    """

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride=1, padding=0,
                 pad_mode: int = PAD_ZERO, bn=None, act: int = ACT_NONE, alpha: float = 0.0, device="cuda",
                 out_scale: Optional[torch.Tensor] = None):
        Cout, Cin, kh, kw = weight.shape
        self.Cout, self.Cin_raw = Cout, Cin
        self.Cin = _round_up(Cin, 4)
        self.kh, self.kw = kh, kw
        self.sy, self.sx = (stride, stride) if isinstance(stride, int) else stride
        self.py, self.px = (padding, padding) if isinstance(padding, int) else padding
        self.pad_mode = pad_mode
        self.act, self.alpha = act, alpha
        w = weight.detach().to(torch.float32)
        if self.Cin != Cin:
            w = torch.cat([w, torch.zeros(Cout, self.Cin - Cin, kh, kw)], dim=1)
        w_kn = w.permute(2, 3, 1, 0).reshape(kh * kw * self.Cin, Cout)  # K = (ky, kx, ci)
        self.w, self.Kp, self.Np = pack_weight_kn(w_kn, device)
        self.taps = [(ky - self.py, kx - self.px, 0) for ky in range(kh) for kx in range(kw)]
        scale = bias_t = None
        if bn is not None:
            scale, bias_t = fold_bn(*bn, conv_bias=bias)
        elif bias is not None:
            bias_t = bias.detach().to(torch.float32)
        if out_scale is not None:  # e.g. ConvNeXt layer-scale gamma: gamma * (conv + b)
            os_ = out_scale.detach().to(torch.float32).reshape(-1)
            scale = os_ if scale is None else scale * os_
            if bias_t is not None:
                bias_t = bias_t * os_
        self.scale = None if scale is None else scale.to(device).contiguous()
        self.bias = None if bias_t is None else bias_t.to(device).contiguous()

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        return ((H + 2 * self.py - self.kh) // self.sy + 1, (W + 2 * self.px - self.kw) // self.sx + 1)

    def desc(self, x: torch.Tensor, out: torch.Tensor, pre: Optional[torch.Tensor] = None,
             post: Optional[torch.Tensor] = None) -> MitConvGemm:
        _check_nhwc(x, "Conv2d input")
        _check_nhwc(out, "Conv2d output")
        B, H, W, Cx = x.shape
        if Cx != self.Cin:
            raise ValueError(f"Conv2d: input has {Cx} channels, layer expects {self.Cin}")
        Ho, Wo = self.out_hw(H, W)
        if tuple(out.shape) != (B, Ho, Wo, self.Cout):
            raise ValueError(f"Conv2d: output shape {tuple(out.shape)} != {(B, Ho, Wo, self.Cout)}")
        for t, nm in ((pre, "pre"), (post, "post")):
            if t is not None and tuple(t.shape) != tuple(out.shape):
                raise ValueError(f"Conv2d: {nm} shape {tuple(t.shape)} != output {tuple(out.shape)}")
        return conv_gemm_desc(
            a=x, NB=B, Hi=H, Wi=W, Cin=self.Cin, a_strides=(x.stride(0), x.stride(1), x.stride(2)), Ho=Ho, Wo=Wo,
            sy=self.sy, sx=self.sx, taps=self.taps, pad_mode=self.pad_mode, w=self.w, ldw=self.Np, Kw=self.Kp,
            Nw=self.Np, N=self.Cout, c=tensor_map(out), pre=tensor_map(pre), post=tensor_map(post),
            scale=self.scale, bias=self.bias, act=self.act, alpha=self.alpha)

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, pre: Optional[torch.Tensor] = None,
                 post: Optional[torch.Tensor] = None, cfg: int = -1) -> torch.Tensor:
        if out is None:
            Ho, Wo = self.out_hw(x.shape[1], x.shape[2])
            out = torch.empty(x.shape[0], Ho, Wo, self.Cout, dtype=torch.float32, device=x.device)
        launch_conv_gemm(self.desc(x, out, pre, post), cfg)
        return out


class WinogradConv3x3:
    """Stride-1, pad-1 3x3 conv (+ folded eval BatchNorm2d + ReLU / LeakyReLU + residual) as Winograd F(4x4, 3x3):
    ``mit_wino43_input`` (B^T d B per 4x4 output tile) -> 36 independent [T x Cin] @ [Cin x Cout] products on ``mit_conv_gemm``
    (Z = 36) -> ``mit_wino43_output`` (A^T m A, epilogue).  2.25 multiplies per output instead of 9.

    weight [Cout, Cin, 3, 3] is transformed once (U = G g G^T in float64, rounded to fp32).  The transformed input V
    [36, T, C] can be shared by several layers that read (a channel prefix of) the same tensor: call ``transform_input``
    once and ``gemm_output`` per layer (``v_channels`` = row length of V)."""

    G = ((1 / 4, 0, 0), (-1 / 6, -1 / 6, -1 / 6), (-1 / 6, 1 / 6, -1 / 6), (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6), (0, 0, 1))

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, pad_mode: int = PAD_ZERO, bn=None,
                 act: int = ACT_NONE, alpha: float = 0.0, device="cuda"):
        Cout, Cin, kh, kw = weight.shape
        if (kh, kw) != (3, 3) or Cin % 4 or Cout % 2:
            raise ValueError(f"WinogradConv3x3: 3x3 kernels with Cin % 4 == 0 and even Cout only (got {tuple(weight.shape)})")
        if act not in (ACT_NONE, ACT_RELU, ACT_LEAKY):
            raise ValueError("WinogradConv3x3: activation must be none / relu / leaky")
        self.Cin, self.Cout, self.pad_mode, self.act, self.alpha = Cin, Cout, pad_mode, act, alpha
        G = torch.tensor(self.G, dtype=torch.float64)
        U = torch.einsum("ia,ocab,jb->ijco", G, weight.detach().to(torch.float64), G).reshape(36, Cin, Cout)
        self.Kp, self.Np = _round_up(Cin, 16), _round_up(Cout, 4)
        u = torch.zeros(36, self.Kp, self.Np, dtype=torch.float32)
        u[:, :Cin, :Cout] = U.to(torch.float32)
        self.u = u.to(device).contiguous()
        register_split(self.u)
        scale = bias_t = None
        if bn is not None:
            scale, bias_t = fold_bn(*bn, conv_bias=bias)
        elif bias is not None:
            bias_t = bias.detach().to(torch.float32)
        self.scale = None if scale is None else scale.to(device).contiguous()
        self.bias = None if bias_t is None else bias_t.to(device).contiguous()

    @staticmethod
    def tiles(B: int, H: int, W: int) -> int:
        return B * ((H + 3) // 4) * ((W + 3) // 4)

    def transform_input(self, x: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """x NHWC [B,H,W,C] (channel-slice views allowed) -> v [36, T, C]."""
        _check_nhwc(x, "WinogradConv3x3 input")
        B, H, W, Cx = x.shape
        T = self.tiles(B, H, W)
        if tuple(v.shape) != (36, T, Cx) or not v.is_contiguous():
            raise ValueError(f"WinogradConv3x3: V must be contiguous [36, {T}, {Cx}] (got {tuple(v.shape)})")
        _lib.check(_lib.load().mit_wino43_input(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), v.data_ptr(), B, H, W, Cx,
                                                self.pad_mode, C.c_void_p(current_stream())), "mit_wino43_input")
        return v

    def gemm_desc(self, v: torch.Tensor, m: torch.Tensor) -> MitConvGemm:
        _, T, Cv = v.shape
        if Cv < self.Cin or tuple(m.shape) != (36, T, self.Cout) or not m.is_contiguous():
            raise ValueError(f"WinogradConv3x3: bad V / M shapes {tuple(v.shape)} / {tuple(m.shape)}")
        cm = MitTensorMap()
        cm.base, cm.zs1, cm.zs0, cm.bs, cm.ys, cm.xs = m.data_ptr(), 0, T * self.Cout, 0, 0, self.Cout
        return conv_gemm_desc(a=v, NB=1, Hi=1, Wi=T, Cin=self.Cin, a_strides=(0, 0, Cv), Ho=1, Wo=T, sy=1, sx=1, taps=[(0, 0, 0)],
                              pad_mode=PAD_ZERO, w=self.u, ldw=self.Np, Kw=self.Kp, Nw=self.Np, N=self.Cout, c=cm, Z=36,
                              zdiv=1 << 30, a_zs=(0, T * Cv), w_zs=(0, self.Kp * self.Np))

    def gemm_output(self, v: torch.Tensor, m: torch.Tensor, out: torch.Tensor, post: Optional[torch.Tensor] = None, cfg: int = -1):
        """36 products from a ready V (its first ``Cin`` channels) and the output transform into ``out`` [B,H,W,Cout]."""
        _check_nhwc(out, "WinogradConv3x3 output")
        B, H, W, Co = out.shape
        if Co != self.Cout or self.tiles(B, H, W) != v.shape[1]:
            raise ValueError(f"WinogradConv3x3: output {tuple(out.shape)} does not match V {tuple(v.shape)}")
        if post is not None and tuple(post.shape) != tuple(out.shape):
            raise ValueError("WinogradConv3x3: residual shape differs from the output")
        launch_conv_gemm(self.gemm_desc(v, m), cfg)
        ps = (0, 0, 0) if post is None else (post.stride(0), post.stride(1), post.stride(2))
        _lib.check(_lib.load().mit_wino43_output(m.data_ptr(), out.data_ptr(), out.stride(0), out.stride(1), out.stride(2), _ptr(post), *ps,
                                                 _ptr(self.scale), _ptr(self.bias), B, H, W, self.Cout, self.act, self.alpha,
                                                 C.c_void_p(current_stream())), "mit_wino43_output")
        return out

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, post: Optional[torch.Tensor] = None,
                 v: Optional[torch.Tensor] = None, m: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, H, W, Cx = x.shape
        T = self.tiles(B, H, W)
        if out is None:
            out = torch.empty(B, H, W, self.Cout, dtype=torch.float32, device=x.device)
        v = torch.empty(36, T, Cx, dtype=torch.float32, device=x.device) if v is None else v
        m = torch.empty(36, T, self.Cout, dtype=torch.float32, device=x.device) if m is None else m
        self.transform_input(x, v)
        return self.gemm_output(v, m, out, post)


class ConvSmallCout:
    """k x k stride-1 "same" conv with <= 4 output channels (+ bias + activation) on ``mit_conv_small_cout``.

    weight [Cout, Cin, k, k]; used for LaMa's 7x7 64->3 output conv (inpainting_lama_mpe.py:597-600)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, pad_mode: int = PAD_ZERO,
                 act: int = ACT_NONE, alpha: float = 0.0, device="cuda"):
        Cout, Cin, kh, kw = weight.shape
        if kh != kw or kh not in (3, 5, 7) or Cout > 4 or Cin % 16:
            raise ValueError(f"ConvSmallCout: unsupported shape {tuple(weight.shape)}")
        self.Cout, self.Cin, self.k, self.pad_mode, self.act, self.alpha = Cout, Cin, kh, pad_mode, act, alpha
        w4 = torch.zeros(kh * kw, Cin, 4, dtype=torch.float32)
        w4[:, :, :Cout] = weight.detach().to(torch.float32).permute(2, 3, 1, 0).reshape(kh * kw, Cin, Cout)
        self.w4 = w4.to(device).contiguous()
        # Cout <= 3: the same weights channel-fastest, [taps][Cin / 4][4 outputs][4 channels] — operand pairs (channels c, c + 1) of the
        # packed-FMA kernel, naturally aligned (no op_sel broadcast in the ISA)
        self.w_pairs = (w4.reshape(kh * kw, Cin // 4, 4, 4).permute(0, 1, 3, 2).to(device).contiguous() if Cout <= 3 else None)
        self.bias = None if bias is None else bias.detach().to(torch.float32).to(device).contiguous()

    def __call__(self, x: torch.Tensor, out: torch.Tensor, parity_major: bool = False) -> torch.Tensor:
        """``parity_major``: the P = 4 planes hold each image as four dense sub-images [2][2][H / 2][W / 2][4] (what
        ``ConvTranspose2d(..., planes=, parity_major=True)`` writes).
        ``x``: [B,H,W,C] NHWC, or the same activations as planes [C / P, B, H, W, P], P = 16 or 4 (contiguous; what
        ``ConvTranspose2d(..., planes=)`` writes) — the packed kernel then loads whole lines per channel group (P = 16) or takes each
        4-channel slice's tile straight into LDS by DMA (P = 4)."""
        plane = 0
        if x.dim() == 5:   # planes of 16 channels (register-staged kernel) or of 4 (LDS-DMA kernel: a slice's tile rows are contiguous runs)
            pc = x.shape[4]
            if self.w_pairs is None or pc not in (4, 16) or x.shape[0] * pc < self.Cin or not x.is_contiguous():
                raise ValueError(f"ConvSmallCout: planar input must be contiguous [Cin / P, B, H, W, P] with P = 4 or 16 for Cout <= 3 (got {tuple(x.shape)})")
            if pc == 4 and self.pad_mode != PAD_REFLECT:
                raise ValueError("ConvSmallCout: 4-channel planes are read by the LDS-DMA kernel, which implements reflect padding only")
            if parity_major and (pc != 4 or x.shape[2] % 2 or x.shape[3] % 2):
                raise ValueError("ConvSmallCout: the parity-major layout needs 4-channel planes of even height and width")
            plane = x.stride(0)
            # (an NHWC-shaped handle on plane 0 for the shape checks below; the kernel addresses the planes itself)
            x = x[0].as_strided((x.shape[1], x.shape[2], x.shape[3], self.Cin), (x.stride(1), x.stride(2), x.stride(3), 1))
        _check_nhwc(x, "ConvSmallCout input")
        _check_nhwc(out, "ConvSmallCout output")
        B, H, W, Cx = x.shape
        if Cx < self.Cin or tuple(out.shape[:3]) != (B, H, W) or out.shape[3] < self.Cout:
            raise ValueError(f"ConvSmallCout: bad shapes {tuple(x.shape)} -> {tuple(out.shape)}")
        if not (x.stride(2) * W == x.stride(1) and x.stride(1) * H == x.stride(0) and out.stride(2) * W == out.stride(1)
                and out.stride(1) * H == out.stride(0)):
            raise ValueError("ConvSmallCout: pixel-dense tensors required")
        lib = _lib.load()
        if parity_major and not plane:
            raise ValueError("ConvSmallCout: parity_major without planar input")
        _lib.check(lib.mit_conv_small_cout(x.data_ptr(), x.stride(2), -plane if parity_major else plane, self.w4.data_ptr(), _ptr(self.w_pairs), _ptr(self.bias), out.data_ptr(),
                                           out.stride(2), B, H, W, self.Cin, self.Cout, self.k, self.pad_mode, self.act,
                                           self.alpha, C.c_void_p(current_stream())), "mit_conv_small_cout")
        return out


class UpsampleConv2d:
    """``Upsample(scale_factor=2, mode='nearest')`` followed by a 3x3 zero-padded conv (+ bias + activation) WITHOUT the
    upsampled tensor: output pixels of parity (a, b) are a 2x2 convolution of the low-resolution input whose weights are
    the sums of the 3x3 taps that land on the same source pixel (ESRGAN's upconv_block, upscaling/esrgan_pytorch.py:317-324).
    2.25x fewer MACs and no 4x-sized intermediate; each parity class is one launch writing ``out[:, a::2, b::2]``.
    The tap merge adds weights before multiplying, so results differ from conv(upsample(x)) by fp32 round-off only."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = ACT_NONE, alpha: float = 0.0,
                 device="cuda"):
        Cout, Cin, kh, kw = weight.shape
        if (kh, kw) != (3, 3) or Cin % 4:
            raise ValueError("UpsampleConv2d: 3x3 kernels with Cin % 4 == 0 only")
        self.Cin, self.Cout, self.act, self.alpha = Cin, Cout, act, alpha
        w = weight.detach().to(torch.float64)
        # source offset of tap k for output parity a: floor((a + k - 1) / 2)  ->  a=0: (-1, 0, 0), a=1: (0, 0, 1)
        groups = {0: {-1: [0], 0: [1, 2]}, 1: {0: [0, 1], 1: [2]}}
        self.sub: List[Tuple[int, int, _Packed]] = []
        for a in (0, 1):
            for b in (0, 1):
                taps, blocks = [], []
                for dy, kys in groups[a].items():
                    for dx, kxs in groups[b].items():
                        wsum = sum(w[:, :, ky, kx] for ky in kys for kx in kxs)  # [Cout, Cin]
                        taps.append((dy, dx, 0))
                        blocks.append(wsum.t().to(torch.float32))  # [Cin, Cout]
                wk, Kp, Np = pack_weight_kn(torch.cat(blocks, dim=0), device)
                self.sub.append((a, b, _Packed(wk, Kp, Np, taps)))
        self.bias = None if bias is None else bias.detach().to(torch.float32).to(device).contiguous()

    def descs(self, x: torch.Tensor, out: torch.Tensor) -> List[MitConvGemm]:
        _check_nhwc(x, "UpsampleConv2d input")
        _check_nhwc(out, "UpsampleConv2d output")
        B, H, W, Cx = x.shape
        if Cx != self.Cin or tuple(out.shape) != (B, 2 * H, 2 * W, self.Cout):
            raise ValueError(f"UpsampleConv2d: bad shapes {tuple(x.shape)} -> {tuple(out.shape)}")
        ds = []
        for a, b, pk in self.sub:
            ov = out[:, a::2, b::2]
            ds.append(conv_gemm_desc(
                a=x, NB=B, Hi=H, Wi=W, Cin=self.Cin, a_strides=(x.stride(0), x.stride(1), x.stride(2)), Ho=H, Wo=W, sy=1, sx=1,
                taps=pk.taps, pad_mode=PAD_ZERO, w=pk.w, ldw=pk.Np, Kw=pk.Kp, Nw=pk.Np, N=self.Cout, c=tensor_map(ov),
                bias=self.bias, act=self.act, alpha=self.alpha))
        return ds

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, cfg: int = -1) -> torch.Tensor:
        if out is None:
            out = torch.empty(x.shape[0], 2 * x.shape[1], 2 * x.shape[2], self.Cout, dtype=torch.float32, device=x.device)
        for d in self.descs(x, out):
            launch_conv_gemm(d, cfg)
        return out


class ConvTranspose2d:
    """nn.ConvTranspose2d (+ folded BN + activation) as ``stride**2`` sub-pixel convolutions.

    Output pixels of parity (py, px) are an ordinary stride-1 convolution of the input with the
    kernel taps of matching parity; each class is one launch writing the strided output view
    ``out[:, py::s, px::s]`` (no zero-stuffing, no scatter).  weight: [Cin, Cout, kh, kw].
    Covers k3 s2 p1 op1 (inpainting_lama_mpe.py:587-589), k4 s2 p1 (ctd_utils/basemodel.py:20)
    and k2 s2 (basemodel.py:93,96).
    """

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, stride=2, padding=0,
                 output_padding=0, bn=None, act: int = ACT_NONE, alpha: float = 0.0, device="cuda"):
        Cin, Cout, kh, kw = weight.shape
        if Cin % 4:
            raise ValueError("ConvTranspose2d: Cin must be a multiple of 4")
        self.Cin, self.Cout, self.k, self.s, self.p, self.op = Cin, Cout, (kh, kw), stride, padding, output_padding
        self.act, self.alpha = act, alpha
        w = weight.detach().to(torch.float32)
        self.sub: List[Tuple[int, int, _Packed]] = []
        s, p = stride, padding
        for py in range(s):
            kys = [ky for ky in range(kh) if (py + p - ky) % s == 0]
            for px in range(s):
                kxs = [kx for kx in range(kw) if (px + p - kx) % s == 0]
                taps, blocks = [], []
                for ky in kys:
                    for kx in kxs:
                        taps.append(((py + p - ky) // s, (px + p - kx) // s, 0))
                        blocks.append(w[:, :, ky, kx])  # [Cin, Cout]
                if not taps:
                    raise ValueError("ConvTranspose2d: empty parity class (kernel smaller than stride)")
                wk, Kp, Np = pack_weight_kn(torch.cat(blocks, dim=0), device)
                self.sub.append((py, px, _Packed(wk, Kp, Np, taps)))
        scale = bias_t = None
        if bn is not None:
            scale, bias_t = fold_bn(*bn, conv_bias=bias)
        elif bias is not None:
            bias_t = bias.detach().to(torch.float32)
        self.scale = None if scale is None else scale.to(device).contiguous()
        self.bias = None if bias_t is None else bias_t.to(device).contiguous()

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        return ((H - 1) * self.s - 2 * self.p + self.k[0] + self.op, (W - 1) * self.s - 2 * self.p + self.k[1] + self.op)

    def descs(self, x: torch.Tensor, out: torch.Tensor, planes: int = 0, parity_major: bool = False) -> List[MitConvGemm]:
        """``planes`` = P > 0: ``out`` is [P, B, Ho, Wo, Cout / P] (contiguous) and every launch writes through a column-split map
        (MitTensorMap.nsplit = Cout / P): channel group g of a pixel goes to plane g.  The consumer that reads channel groups (the 7x7
        output convolution) then finds each group in whole lines."""
        split = nhi = 0
        if planes:
            if out.dim() != 5 or out.shape[0] != planes or out.shape[4] * planes != self.Cout or not out.is_contiguous():
                raise ValueError(f"ConvTranspose2d: planar output must be contiguous [{planes}, B, Ho, Wo, {self.Cout // planes}]")
            split, nhi = out.shape[4], out.stride(0)
            out = out[0].as_strided((out.shape[1], out.shape[2], out.shape[3], self.Cout), (out.stride(1), out.stride(2), out.stride(3), 1))
        _check_nhwc(x, "ConvTranspose2d input")
        _check_nhwc(out, "ConvTranspose2d output")
        B, H, W, Cx = x.shape
        if Cx != self.Cin:
            raise ValueError(f"ConvTranspose2d: input has {Cx} channels, layer expects {self.Cin}")
        Ho, Wo = self.out_hw(H, W)
        if tuple(out.shape) != (B, Ho, Wo, self.Cout):
            raise ValueError(f"ConvTranspose2d: output shape {tuple(out.shape)} != {(B, Ho, Wo, self.Cout)}")
        if parity_major and not (planes and self.s == 2 and Ho % 2 == 0 and Wo % 2 == 0):
            raise ValueError("ConvTranspose2d: the parity-major layout is for planar output of a stride-2 layer with even output size")
        ds = []
        for py, px, pk in self.sub:
            ov = out[:, py::self.s, px::self.s]
            if ov.shape[1] == 0 or ov.shape[2] == 0:
                continue
            if parity_major:   # every plane holds the image as four dense sub-images, one per output parity class: [2 (py)][2 (px)][Ho / 2][Wo / 2][P]
                pc, h2, w2 = self.Cout // planes, Ho // 2, Wo // 2   # — a class's launch then writes consecutive pixels (an interleaved
                ov = out.as_strided((B, h2, w2, self.Cout), (Ho * Wo * pc, w2 * pc, pc, 1),      # image would put them 2 P floats apart)
                                    out.storage_offset() + (py * 2 + px) * h2 * w2 * pc)
            ds.append(conv_gemm_desc(
                a=x, NB=B, Hi=H, Wi=W, Cin=self.Cin, a_strides=(x.stride(0), x.stride(1), x.stride(2)),
                Ho=ov.shape[1], Wo=ov.shape[2], sy=1, sx=1, taps=pk.taps, pad_mode=PAD_ZERO, w=pk.w, ldw=pk.Np,
                Kw=pk.Kp, Nw=pk.Np, N=self.Cout, c=tensor_map(ov, nsplit=split, nhi=nhi), scale=self.scale, bias=self.bias, act=self.act,
                alpha=self.alpha))
        return ds

    def __call__(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, cfg: int = -1, planes: int = 0, parity_major: bool = False) -> torch.Tensor:
        if out is None:
            Ho, Wo = self.out_hw(x.shape[1], x.shape[2])
            out = (torch.empty(planes, x.shape[0], Ho, Wo, self.Cout // planes, dtype=torch.float32, device=x.device) if planes else
                   torch.empty(x.shape[0], Ho, Wo, self.Cout, dtype=torch.float32, device=x.device))
        for d in self.descs(x, out, planes, parity_major):
            launch_conv_gemm(d, cfg)
        return out
