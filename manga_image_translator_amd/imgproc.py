"""8-bit image resizes of the plugin glue, on the GPU (``mit_resize_u8``) with host-built tap tables, plus a numpy twin.

What the reference does with OpenCV around the inpainter and the detectors
(/root/reference/manga_translator/inpainting/inpainting_lama_mpe.py:63-79,112-113 — ``resize_keep_aspect`` =
``cv2.INTER_LINEAR_EXACT`` (utils/generic.py:251-255), then ``cv2.INTER_LINEAR`` to a multiple of 8 and back;
detection/common.py:79-84):

  mode 0  INTER_LINEAR        coefficients round(f * 2048) as int16, 32-bit row sums, the 8-bit vertical formula of resize.cpp
  mode 1  exact 2x shrink     2x2 box mean — OpenCV substitutes INTER_AREA for both linear flavours at scale 1/2
  mode 2  INTER_LINEAR_EXACT  resize_bitExact: coefficients round(f * 256) (ufixedpoint16), (r0*h0 + r1*h1 + 32768) >> 16

The tables are the same for the device kernel and for ``resize_u8_host`` (numpy, used by CPU tests and small host-side calls), so
the two are bit-identical by construction; the oracle keeps an independent per-pixel restatement (oracle/imgproc.py).  Parity
with the real OpenCV is unpinned (it is installed nowhere this runs).
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

INTER_LINEAR, BOX_2X, INTER_LINEAR_EXACT = 0, 1, 2


@lru_cache(maxsize=64)
def linear_taps(n_src: int, n_dst: int, exact: bool) -> Tuple[np.ndarray, np.ndarray]:
    """Per destination index: first source index (int32) and the two uint16 weights of (idx, idx + 1).

    INTER_LINEAR: f = float32((d + 0.5) * (n_src / n_dst) - 0.5), weights (round((1 - fr) * 2048), round(fr * 2048)) in float32
    arithmetic (resize.cpp: saturate_cast<short>(v * INTER_RESIZE_COEF_SCALE)); INTER_LINEAR_EXACT: f in double with
    scale = 1 / (n_dst / n_src), weights (256 - round(fr * 256), round(fr * 256)).  Out-of-range source positions clamp to the first /
    last sample with full weight, like OpenCV's dmin / dmax handling."""
    d = np.arange(n_dst, dtype=np.float64)
    if exact:
        scale = 1.0 / (n_dst / n_src)
        f = scale * (d + 0.5) - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0
        one = 256
        c1 = np.rint(fr * 256.0).astype(np.int64)
    else:
        f = ((d + 0.5) * (n_src / n_dst) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        one = 2048
        c1 = None
    lo = i0 < 0
    hi = (i0 >= n_src - 1) | (n_src <= 1)
    idx = np.where(lo, 0, np.where(hi, n_src - 1, i0)).astype(np.int32)
    if exact:
        c1 = np.where(lo | hi, 0, c1)
        c0 = one - c1
    else:
        fr = np.where(lo | hi, np.float32(0), fr).astype(np.float32)
        c1 = np.rint(fr * np.float32(2048)).astype(np.int64)
        c0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int64)
    coef = np.stack([c0, c1], 1).astype(np.uint16)
    idx.setflags(write=False)
    coef.setflags(write=False)
    return idx, coef


def pick_mode(sh: int, sw: int, dh: int, dw: int, exact: bool) -> int:
    if sh == 2 * dh and sw == 2 * dw:
        return BOX_2X
    return INTER_LINEAR_EXACT if exact else INTER_LINEAR


def resize_u8_host(src: np.ndarray, dsize: Tuple[int, int], exact: bool = False) -> np.ndarray:
    """cv2.resize(src, (w, h), INTER_LINEAR / INTER_LINEAR_EXACT) for uint8 [H,W] or [H,W,C] in numpy, from the same tables the
    device kernel uses."""
    dw, dh = int(dsize[0]), int(dsize[1])
    squeeze = src.ndim == 2
    s = src[..., None] if squeeze else src
    sh, sw = s.shape[:2]
    mode = pick_mode(sh, sw, dh, dw, exact)
    if (sh, sw) == (dh, dw):
        out = s.copy()
    elif mode == BOX_2X:
        t = s.astype(np.int32)
        out = ((t[0::2, 0::2] + t[0::2, 1::2] + t[1::2, 0::2] + t[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        yi, yc = linear_taps(sh, dh, exact)
        xi, xc = linear_taps(sw, dw, exact)
        y1, x1 = np.minimum(yi + 1, sh - 1), np.minimum(xi + 1, sw - 1)
        t = s.astype(np.int64)
        rows = t[:, xi] * xc[None, :, 0, None].astype(np.int64) + t[:, x1] * xc[None, :, 1, None].astype(np.int64)
        b0, b1 = yc[:, 0, None, None].astype(np.int64), yc[:, 1, None, None].astype(np.int64)
        if mode == INTER_LINEAR:
            out = (((b0 * (rows[yi] >> 4)) >> 16) + ((b1 * (rows[y1] >> 4)) >> 16) + 2) >> 2
        else:
            out = (b0 * rows[yi] + b1 * rows[y1] + 32768) >> 16
        out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out


def keep_aspect_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """(new_w, new_h) of resize_keep_aspect (utils/generic.py:251-255): python round() of the scaled sides."""
    ratio = float(size) / max(h, w)
    return round(w * ratio), round(h * ratio)


def resize_keep_aspect_host(img: np.ndarray, size: int) -> np.ndarray:
    return resize_u8_host(img, keep_aspect_size(img.shape[0], img.shape[1], size), exact=True)


def resize_u8(src: torch.Tensor, dsize: Tuple[int, int], exact: bool = False) -> torch.Tensor:
    """Device resize of a uint8 tensor [B,H,W,C] (or [B,H,W]) to (w, h) = ``dsize`` through ``mit_resize_u8``."""
    from . import lib as _lib
    from . import ops

    if src.dtype != torch.uint8 or src.dim() not in (3, 4):
        raise ValueError(f"resize_u8 expects uint8 [B,H,W] or [B,H,W,C], got {src.dtype} {tuple(src.shape)}")
    if not src.is_cuda:
        raise ValueError("resize_u8 runs on device tensors (resize_u8_host is the numpy twin for host arrays)")
    squeeze = src.dim() == 3
    s = (src[..., None] if squeeze else src).contiguous()
    B, sh, sw, Cc = s.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (sh, sw) == (dh, dw):
        return src.clone()
    mode = pick_mode(sh, sw, dh, dw, exact)
    out = torch.empty(B, dh, dw, Cc, dtype=torch.uint8, device=s.device)
    tabs = [None] * 4
    if mode != BOX_2X:
        yi, yc = linear_taps(sh, dh, exact)
        xi, xc = linear_taps(sw, dw, exact)
        blob = np.concatenate([yi.view(np.uint8), xi.view(np.uint8), yc.reshape(-1).view(np.uint8), xc.reshape(-1).view(np.uint8)])
        dev = torch.from_numpy(blob.copy()).to(s.device)   # one small upload (a few KB) per call
        o1, o2, o3 = 4 * dh, 4 * dh + 4 * dw, 4 * dh + 4 * dw + 4 * dh
        tabs = [dev.data_ptr(), dev.data_ptr() + o2, dev.data_ptr() + o1, dev.data_ptr() + o3]
        keep = dev  # noqa: F841 - alive until the launch is queued (same stream as the free)
    _lib.check(_lib.load().mit_resize_u8(s.data_ptr(), B, sh, sw, Cc, out.data_ptr(), dh, dw, mode, tabs[0], tabs[1], tabs[2], tabs[3],
                                         C.c_void_p(ops.current_stream())), "mit_resize_u8")
    return out[..., 0] if squeeze else out


def select_u8(mask: torch.Tensor, thr: int, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """out = mask >= thr ? a : b per pixel; mask u8 [B,H,W], a / b u8 [B,H,W,C] on the device."""
    from . import lib as _lib
    from . import ops

    if a.shape != b.shape or tuple(mask.shape) != tuple(a.shape[:3]) or a.dtype != torch.uint8 or mask.dtype != torch.uint8:
        raise ValueError(f"select_u8: bad operands {tuple(mask.shape)} {tuple(a.shape)} {tuple(b.shape)}")
    a, b, mask = a.contiguous(), b.contiguous(), mask.contiguous()
    out = torch.empty_like(a)
    _lib.check(_lib.load().mit_select_u8(mask.data_ptr(), int(thr), a.data_ptr(), b.data_ptr(), out.data_ptr(), mask.numel(), a.shape[3],
                                         C.c_void_p(ops.current_stream())), "mit_select_u8")
    return out


# ------------------------------------------------------------------------------------------------------------------------
# cv2.bilateralFilter (8-bit, 3 channels) — mask_refinement/text_mask_utils.py:159, detection/default.py:64
# ------------------------------------------------------------------------------------------------------------------------

@lru_cache(maxsize=8)
def bilateral_tables(d: int, sigma_color: float, sigma_space: float):
    """The tables of OpenCV's bilateralFilter_8u: (radius, tap offsets (dy << 16 | dx & 0xffff) int32, spatial weights float32,
    colour weights float32[256 * 3]).  radius = d / 2 (or round(1.5 sigma_space) for d <= 0), circular support in row-major
    order, weights (float)exp(double)."""
    sigma_color = sigma_color if sigma_color > 0 else 1.0
    sigma_space = sigma_space if sigma_space > 0 else 1.0
    gc, gs = -0.5 / (sigma_color * sigma_color), -0.5 / (sigma_space * sigma_space)
    radius = max(int(np.rint(sigma_space * 1.5)) if d <= 0 else d // 2, 1)
    color_w = np.exp((np.arange(256 * 3, dtype=np.float64) ** 2) * gc).astype(np.float32)
    ofs, wts = [], []
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = np.sqrt(float(i) * i + float(j) * j)
            if r > radius:
                continue
            ofs.append(np.int32(np.uint32((i & 0xffff) << 16 | (j & 0xffff)).astype(np.int32)))
            wts.append(np.float32(np.exp(r * r * gs)))
    return radius, np.asarray(ofs, dtype=np.int32), np.asarray(wts, dtype=np.float32), color_w


_BILATERAL_DEV = {}


def bilateral_filter_u8(img: torch.Tensor, d: int = 17, sigma_color: float = 80.0, sigma_space: float = 80.0) -> torch.Tensor:
    """cv2.bilateralFilter(img, d, sigma_color, sigma_space) for uint8 RGB pages [B,H,W,3] (or [H,W,3]) on the device
    (``mit_bilateral_u8c3``)."""
    from . import lib as _lib
    from . import ops

    if img.dtype != torch.uint8 or img.shape[-1] != 3 or img.dim() not in (3, 4) or not img.is_cuda:
        raise ValueError(f"bilateral_filter_u8 expects a uint8 device tensor [B,H,W,3] or [H,W,3], got {img.dtype} {tuple(img.shape)}")
    squeeze = img.dim() == 3
    s = (img[None] if squeeze else img).contiguous()
    key = (int(d), float(sigma_color), float(sigma_space), s.device)
    if key not in _BILATERAL_DEV:
        radius, ofs, wts, cw = bilateral_tables(int(d), float(sigma_color), float(sigma_space))
        _BILATERAL_DEV[key] = (radius, torch.from_numpy(ofs).to(s.device), torch.from_numpy(wts).to(s.device), torch.from_numpy(cw).to(s.device))
    radius, ofs, wts, cw = _BILATERAL_DEV[key]
    out = torch.empty_like(s)
    B, H, W, _ = s.shape
    _lib.check(_lib.load().mit_bilateral_u8c3(s.data_ptr(), out.data_ptr(), B, H, W, radius, ofs.numel(), ofs.data_ptr(), wts.data_ptr(),
                                              cw.data_ptr(), C.c_void_p(ops.current_stream())), "mit_bilateral_u8c3")
    return out[0] if squeeze else out


# ---- page edges: PIL <-> the uint8 RGB arrays the stage plugins exchange (utils/generic.py:223-249) ----------------------------

def load_image(img):
    """PIL image -> (uint8 RGB array [H, W, 3], alpha channel or None): RGBA and palette pages are flattened onto white and their alpha
    is kept for ``dump_image`` (utils/generic.py:223-239); everything else is ``convert('RGB')``."""
    from PIL import Image

    if img.mode in ("RGBA", "P"):
        if img.mode == "P":
            img = img.convert("RGBA")
        img.load()  # split() needs the pixel data
        background = Image.new("RGB", img.size, (255, 255, 255))
        alpha = img.split()[3]
        background.paste(img, mask=alpha)
        return np.array(background), alpha
    return np.array(img.convert("RGB")), None


def dump_image(img_pil, img: np.ndarray, alpha_ch=None):
    """Result array -> RGBA PIL image of the result's size, the page's alpha re-attached when it had one (utils/generic.py:241-249)."""
    from PIL import Image

    if alpha_ch is not None:
        if img.shape[2] != 4:
            img = np.concatenate([img.astype(np.uint8), np.array(alpha_ch).astype(np.uint8)[..., None]], axis=2)
    else:
        img = img.astype(np.uint8)
    result = img_pil.convert("RGBA").resize((img.shape[1], img.shape[0]))
    result.paste(Image.fromarray(img), mask=alpha_ch)
    return result
