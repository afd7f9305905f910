"""DenseCRF refinement of the text-line mask crops of a page, batched on the GPU (``mit_densecrf_refine``).

Replaces ``refine_mask`` of the reference (/root/reference/manga_translator/mask_refinement/text_mask_utils.py:68-94), which
runs pydensecrf's DenseCRF2D once per text line on the CPU: Gaussian pairwise term (sxy=1, compat=3), bilateral term (sxy=23,
srgb=7, compat=20), 5 mean-field iterations, argmax.  Here every line of the page goes through ONE call: the (already
bilateral-filtered) page stays on the device, only the small mask crops travel.  pydensecrf is not importable anywhere this runs,
so parity with it is unpinned; the kernels are checked against a CPU restatement of the library's algorithm (oracle/densecrf.py).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops

# refine_mask's constants (text_mask_utils.py:83-91)
GAUSS_SXY, GAUSS_COMPAT = 1.0, 3.0
BILATERAL_SXY, BILATERAL_SRGB, BILATERAL_COMPAT = 23.0, 7.0, 20.0
ITERATIONS = 5


def unary_lut() -> np.ndarray:
    """-log(clip([1 - m/255, m/255], 1e-5, 1)) for m = 0..255 (text_mask_utils.py:74-79 + pydensecrf.utils.unary_from_softmax),
    [256, 2] float32 — evaluated once on the host so the device works from the same floats as numpy."""
    m = np.arange(256, dtype=np.uint8)
    sm = np.stack([255 - m, m], 1).astype(np.float32) / np.float32(255.0)
    return (-np.log(np.clip(sm, np.float32(1e-5), np.float32(1.0)))).astype(np.float32)


class DenseCrfRefiner:
    """Batched ``refine_mask`` over the crops of one page.  The workspace grows to the largest batch seen and is reused."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self._lut = torch.from_numpy(unary_lut()).to(self.device).contiguous()
        self._ws = None

    def release_workspace(self):
        self._ws = None

    def refine(self, page_dev: torch.Tensor, rects: Sequence[Tuple[int, int, int, int]], masks: Sequence[np.ndarray],
               return_q: bool = False, iterations: int = ITERATIONS, packed: bool = False):
        """page_dev u8 [H,W,3] (device), rects (x, y, w, h) per crop, masks u8 [h, w] per crop (host) -> list of u8 [h, w] masks in
        {0, 255} (and the final marginals [h, w, 2] per crop when ``return_q``).  ``packed``: the refined crops stay on the device —
        returns (u8 device tensor with the crops back to back, row-major, [offset of each crop])."""
        if page_dev.dtype != torch.uint8 or page_dev.dim() != 3 or page_dev.shape[2] != 3 or not page_dev.is_cuda:
            raise ValueError(f"DenseCrfRefiner.refine expects a uint8 device page [H,W,3], got {page_dev.dtype} {tuple(page_dev.shape)}")
        if len(rects) != len(masks):
            raise ValueError("one mask per crop rectangle")
        if not rects:
            if packed:
                return torch.empty(0, dtype=torch.uint8, device=self.device), []
            return ([], []) if return_q else []
        page_dev = page_dev.contiguous()
        H, W, _ = page_dev.shape
        crops = (_lib.MitCrfCrop * len(rects))()
        flat = []
        for i, ((x, y, w, h), m) in enumerate(zip(rects, masks)):
            m = np.ascontiguousarray(m, dtype=np.uint8)
            if m.shape != (h, w):
                raise ValueError(f"crop {i}: mask {m.shape} does not match its rectangle {(h, w)}")
            crops[i].x, crops[i].y, crops[i].w, crops[i].h = int(x), int(y), int(w), int(h)
            flat.append(m.reshape(-1))
        L = _lib.load()
        need = L.mit_densecrf_workspace_bytes(C.byref(crops), len(rects))
        if need < 0:
            raise RuntimeError("DenseCrfRefiner: bad crop list or batch too large")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        mask_dev = torch.from_numpy(np.concatenate(flat)).to(self.device)
        out_dev = torch.empty_like(mask_dev)
        q_dev = torch.empty(mask_dev.numel(), 2, dtype=torch.float32, device=self.device) if return_q else None
        _lib.check(L.mit_densecrf_refine(page_dev.data_ptr(), H, W, C.byref(crops), len(rects), mask_dev.data_ptr(), out_dev.data_ptr(),
                                         q_dev.data_ptr() if return_q else None, GAUSS_SXY, GAUSS_COMPAT, BILATERAL_SXY, BILATERAL_SRGB,
                                         BILATERAL_COMPAT, int(iterations), self._lut.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                         C.c_void_p(ops.current_stream())), "mit_densecrf_refine")
        if packed:
            offs, o = [], 0
            for (x, y, w, h) in rects:
                offs.append(o)
                o += w * h
            return out_dev, offs
        out = out_dev.cpu().numpy()
        res, qs, o = [], [], 0
        qh = q_dev.cpu().numpy() if return_q else None
        for (x, y, w, h) in rects:
            res.append(out[o:o + w * h].reshape(h, w).copy())
            if return_q:
                qs.append(qh[o:o + w * h].reshape(h, w, 2).copy())
            o += w * h
        return (res, qs) if return_q else res
