"""ComicTextDetector network (``--detector ctd``) on the gfx950 engine.

Same graph as ``TextDetBase.forward`` of the reference
(/root/reference/manga_translator/detection/ctd_utils/basemodel.py:234-238): fused YOLOv5s
backbone layers 0-9 -> ``UnetHead`` (:56-72) -> ``DBHead`` (:100-119).  YOLO layers 10-24 and
Detect only produce ``blks``, which ``ComicTextDetector._infer`` discards (ctd.py:142,150-151); they
are not executed (and not counted in the FLOP figure).

Layout: fp32 NHWC; every torch.cat of the reference is a pre-allocated buffer whose channel slices
the producers write directly (C3's cat, SPPF's cat, the U-Net skips); BatchNorm / LeakyReLU /
SiLU / ReLU / sigmoid / residual adds live in the conv epilogues; ConvTranspose2d runs as four
sub-pixel convolutions writing strided views.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SILU

INPUT_SIZE = 1024  # ctd.py:84


def _bn(sd, p, eps=1e-5):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], eps)


def _fused_yolo_conv(sd, p, k, s, device):
    """yolov5 Conv after Model.fuse() (yolo.py:185-192): BN folded INTO the fp32 weights exactly like
    fuse_conv_and_bn (yolov5_utils.py:22-42; BN eps 1e-3 from initialize_weights :52-56), then SiLU."""
    w = sd[p + ".conv.weight"].to(torch.float32)
    bw, bb, rm, rv = (sd[p + ".bn." + n].to(torch.float32) for n in ("weight", "bias", "running_mean", "running_var"))
    eps = 1e-3
    scale = bw.div(torch.sqrt(eps + rv))
    fw = (w.reshape(w.shape[0], -1) * scale[:, None]).reshape(w.shape)
    fb = bb - bw.mul(rm).div(torch.sqrt(rv + eps))
    return ops.Conv2d(fw, fb, stride=s, padding=(2 if k == 6 else k // 2), act=ACT_SILU, device=device)


def _head_conv(sd, p, k, device):
    """heads' Conv (common.py:30-46): conv (no bias) -> BN -> LeakyReLU(0.1)."""
    return ops.Conv2d(sd[p + ".conv.weight"], None, padding=k // 2, bn=_bn(sd, p + ".bn"), act=ACT_LEAKY, alpha=0.1,
                      device=device)


class _C3:
    """C3 (common.py:126-136) with n Bottlenecks (e=1.0, shortcut as given)."""

    def __init__(self, sd, p, c1, c2, n, mk, shortcut=True):
        self.c_ = c2 // 2
        self.c2 = c2
        self.cv1, self.cv2, self.cv3 = mk(p + ".cv1", 1), mk(p + ".cv2", 1), mk(p + ".cv3", 1)
        self.m = [(mk(f"{p}.m.{j}.cv1", 1), mk(f"{p}.m.{j}.cv2", 3)) for j in range(n)]
        self.shortcut = shortcut

    def __call__(self, eng, x, out, tag):
        B, H, W, _ = x.shape
        cat = eng._buf(f"{tag}.cat", B, H, W, 2 * self.c_)
        y = cat[..., :self.c_]
        self.cv2(x, out=cat[..., self.c_:])
        self.cv1(x, out=y)
        t = eng._buf(f"{tag}.t", B, H, W, self.c_)
        for a, b in self.m:
            a(y, out=t)
            b(t, out=y, post=y if self.shortcut else None)  # x + cv2(cv1(x)), in place (element-wise safe)
        return self.cv3(cat, out=out)


class CtdEngine:
    """Batched text-detection network: u8 pages -> (mask u8, lines fp32) on the device."""

    def __init__(self, yolo_sd: Dict[str, torch.Tensor], seg_sd: Dict[str, torch.Tensor], det_sd: Dict[str, torch.Tensor],
                 device="cuda"):
        self.device = dev = torch.device(device)
        ymk = lambda k_s: (lambda p, k: _fused_yolo_conv(yolo_sd, p, k, k_s, dev))
        y1 = lambda p, k: _fused_yolo_conv(yolo_sd, p, k, 1, dev)
        self.y0 = _fused_yolo_conv(yolo_sd, "model.0", 6, 2, dev)
        self.y1 = _fused_yolo_conv(yolo_sd, "model.1", 3, 2, dev)
        self.y2 = _C3(yolo_sd, "model.2", 64, 64, 1, y1)
        self.y3 = _fused_yolo_conv(yolo_sd, "model.3", 3, 2, dev)
        self.y4 = _C3(yolo_sd, "model.4", 128, 128, 2, y1)
        self.y5 = _fused_yolo_conv(yolo_sd, "model.5", 3, 2, dev)
        self.y6 = _C3(yolo_sd, "model.6", 256, 256, 3, y1)
        self.y7 = _fused_yolo_conv(yolo_sd, "model.7", 3, 2, dev)
        self.y8 = _C3(yolo_sd, "model.8", 512, 512, 1, y1)
        self.y9a = y1("model.9.cv1", 1)
        self.y9b = y1("model.9.cv2", 1)

        def up_c3(sd, p, cin, mid, cout):
            hk = lambda q, k: _head_conv(sd, q, k, dev)
            c3 = _C3(sd, p + ".conv.0", cin, mid, 1, hk)
            up = ops.ConvTranspose2d(sd[p + ".conv.1.weight"], None, stride=2, padding=1, bn=_bn(sd, p + ".conv.2"),
                                     act=ACT_RELU, device=dev)
            return c3, up

        s = seg_sd
        self.down1 = _C3(s, "down_conv1.conv", 512, 512, 1, lambda q, k: _head_conv(s, q, k, dev))
        self.up0 = up_c3(s, "upconv0", 512, 512, 256)
        self.up2 = up_c3(s, "upconv2", 768, 512, 256)
        self.up3 = up_c3(s, "upconv3", 512, 512, 256)
        self.up4 = up_c3(s, "upconv4", 384, 256, 128)
        self.up5 = up_c3(s, "upconv5", 192, 128, 64)
        self.up6 = ops.ConvTranspose2d(s["upconv6.0.weight"], None, stride=2, padding=1, act=ACT_SIGMOID, device=dev)
        d = det_sd
        self.d_up3 = up_c3(d, "upconv3", 512, 512, 256)
        self.d_up4 = up_c3(d, "upconv4", 384, 256, 128)
        self.d_conv = ops.Conv2d(d["conv.0.weight"], d["conv.0.bias"], bn=_bn(d, "conv.1"), act=ACT_RELU, device=dev)

        def branch(p, first_bias):
            c0 = ops.Conv2d(d[p + ".0.weight"], d[p + ".0.bias"] if first_bias else None, padding=1, bn=_bn(d, p + ".1"),
                            act=ACT_RELU, device=dev)
            t1 = ops.ConvTranspose2d(d[p + ".3.weight"], d[p + ".3.bias"], stride=2, bn=_bn(d, p + ".4"), act=ACT_RELU,
                                     device=dev)
            t2 = ops.ConvTranspose2d(d[p + ".6.weight"], d[p + ".6.bias"], stride=2, act=ACT_SIGMOID, device=dev)
            return c0, t1, t2

        self.br_binarize = branch("binarize", True)
        self.br_thresh = branch("thresh", False)
        self._ws = ops.Workspace(self.device)
        self._prep_tabs: Dict[Tuple, dict] = {}

    def _buf(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        """Named workspace slab, grown to the largest request (ops.Workspace): memory is bounded by the largest page seen."""
        return self._ws.buf(name, *shape, dtype=dtype)

    def release_workspace(self):
        self._ws.release()

    # -- letterbox geometry (imgproc_utils.py:69-100) ------------------------------------------
    @staticmethod
    def letterbox_geometry(H: int, W: int, S: int = INPUT_SIZE):
        r = min(S / H, S / W)
        nw, nh = int(round(W * r)), int(round(H * r))
        return nh, nw, S - nw, S - nh  # nh, nw, dw, dh

    def _resize_tables(self, H, W, nh, nw):
        key = (H, W, nh, nw)
        if key not in self._prep_tabs:
            while len(self._prep_tabs) >= 8:  # a few KB per page shape: keep the most recent shapes only
                self._prep_tabs.pop(next(iter(self._prep_tabs)))

            def taps(n_src, n_dst):  # OpenCV resize.cpp linear coefficients, 11-bit fixed point
                idx = np.zeros(n_dst, dtype=np.int32)
                co = np.zeros((n_dst, 2), dtype=np.int16)
                scale = n_src / n_dst
                for dd in range(n_dst):
                    f = np.float32((dd + 0.5) * scale - 0.5)
                    s0 = int(np.floor(f))
                    f = np.float32(f - s0)
                    if s0 < 0:
                        s0, f = 0, np.float32(0)
                    if s0 >= n_src - 1:
                        s0, f = n_src - 1, np.float32(0)
                    idx[dd] = s0
                    co[dd, 0] = int(np.rint(np.float32((np.float32(1.0) - f) * np.float32(2048))))
                    co[dd, 1] = int(np.rint(np.float32(f * np.float32(2048))))
                return idx, co
            yi, yc = taps(H, nh)
            xi, xc = taps(W, nw)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._prep_tabs[key] = dict(yi=t(yi), yc=t(yc), xi=t(xi), xc=t(xc))
        return self._prep_tabs[key]

    def _up(self, pair, x, out, tag):
        c3, up = pair
        B, H, W, _ = x.shape
        mid = self._buf(tag + ".mid", B, H, W, c3.c2)
        c3(self, x, mid, tag)
        return up(mid, out=out)

    @torch.no_grad()
    def forward(self, pages_u8: torch.Tensor, taps: Optional[dict] = None):
        """pages_u8 [B,H,W,3] u8 (device) -> (mask_u8 [B,S-dh,S-dw] u8, lines [B,2,S-dh,S-dw] fp32 views, (dw, dh)).

        Covers ctd.py:137-153 + postprocess_mask (:30-44); the contour / refine-mask post-processing
        (:155-177, OpenCV/pyclipper) stays with the caller."""
        if pages_u8.dtype != torch.uint8 or pages_u8.dim() != 4 or pages_u8.shape[-1] != 3:
            raise ValueError(f"CtdEngine.forward expects u8 [B,H,W,3], got {pages_u8.dtype} {tuple(pages_u8.shape)}")
        pages_u8 = pages_u8.contiguous()
        B, H, W, _ = pages_u8.shape
        S = INPUT_SIZE
        nh, nw, dw, dh = self.letterbox_geometry(H, W, S)
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        x = self._buf("in4", B, S, S, 4)
        if (nh, nw) == (H, W):
            mode, tb = 0, None
        elif H == 2 * nh and W == 2 * nw:
            mode, tb = 1, None
        else:
            mode, tb = 2, self._resize_tables(H, W, nh, nw)
        _lib.check(lib.mit_ctd_prep(pages_u8.data_ptr(), B, H, W, nh, nw, S, mode,
                                    tb["yi"].data_ptr() if tb else None, tb["yc"].data_ptr() if tb else None,
                                    tb["xi"].data_ptr() if tb else None, tb["xc"].data_ptr() if tb else None,
                                    x.data_ptr(), st), "mit_ctd_prep")
        if taps is not None:
            taps["input"] = x.clone()
        # ---- YOLOv5s backbone (yolo.py:115-134), features written straight into the U-Net concat buffers ----
        a0 = self._buf("y0", B, 512, 512, 32)
        self.y0(x, out=a0)
        cat160 = self._buf("cat160", B, 256, 256, 192)  # [f160 | u160]
        f160 = cat160[..., :64]
        self.y1(a0, out=f160)
        a2 = self._buf("y2", B, 256, 256, 64)
        self.y2(self, f160, a2, "y2")
        cat80 = self._buf("cat80", B, 128, 128, 384)  # [f80 | u80]  (UnetHead)
        f80 = cat80[..., :128]
        self.y3(a2, out=f80)
        a4 = self._buf("y4", B, 128, 128, 128)
        self.y4(self, f80, a4, "y4")
        cat40 = self._buf("cat40", B, 64, 64, 512)  # [f40 | u40]  (shared by both heads)
        f40 = cat40[..., :256]
        self.y5(a4, out=f40)
        a6 = self._buf("y6", B, 64, 64, 256)
        self.y6(self, f40, a6, "y6")
        cat20 = self._buf("cat20", B, 32, 32, 768)  # [f20 | u20]
        f20 = cat20[..., :512]
        self.y7(a6, out=f20)
        a8 = self._buf("y8", B, 32, 32, 512)
        self.y8(self, f20, a8, "y8")
        sp = self._buf("sppf", B, 32, 32, 1024)  # SPPF cat [x | y1 | y2 | y3] (common.py:190-197)
        self.y9a(a8, out=sp[..., :256])
        for j in range(3):
            src, dst = sp[..., 256 * j:256 * (j + 1)], sp[..., 256 * (j + 1):256 * (j + 2)]
            _lib.check(lib.mit_maxpool_nhwc(src.data_ptr(), 1024, dst.data_ptr(), 1024, B, 32, 32, 256, 5, st), "mit_maxpool_nhwc")
        f3 = self._buf("f3", B, 32, 32, 512)
        self.y9b(sp, out=f3)
        if taps is not None:
            taps.update(f160=f160.clone(), f80=f80.clone(), f40=f40.clone(), f20=f20.clone(), f3=f3.clone())
        # ---- UnetHead (basemodel.py:56-72) ----
        p16 = self._buf("p16", B, 16, 16, 512)
        _lib.check(lib.mit_avgpool2_nhwc(f3.data_ptr(), 512, p16.data_ptr(), 512, B, 16, 16, 512, st), "mit_avgpool2_nhwc")
        d10 = self._buf("d10", B, 16, 16, 512)
        self.down1(self, p16, d10, "down1")
        self._up(self.up0, d10, cat20[..., 512:], "up0")  # u20 -> cat20
        self._up(self.up2, cat20, cat40[..., 256:], "up2")  # u40 -> cat40
        self._up(self.up3, cat40, cat80[..., 128:], "up3")  # u80 -> cat80
        self._up(self.up4, cat80, cat160[..., 64:], "up4")  # u160 -> cat160
        u320 = self._buf("u320", B, 512, 512, 64)
        self._up(self.up5, cat160, u320, "up5")
        mask = self._buf("mask", B, S, S, 1)
        self.up6(u320, out=mask)
        # ---- DBHead (basemodel.py:100-119) ----
        dcat80 = self._buf("dcat80", B, 128, 128, 384)  # [f80 | u80'] with DBHead's own upconv3
        _lib.check(lib.mit_copy_channels(f80.data_ptr(), 384, dcat80.data_ptr(), 384, B * 128 * 128, 128, st), "mit_copy_channels")
        self._up(self.d_up3, cat40, dcat80[..., 128:], "dup3")
        dx = self._buf("dx", B, 256, 256, 128)
        self._up(self.d_up4, dcat80, dx, "dup4")
        dc = self._buf("dc", B, 256, 256, 64)
        self.d_conv(dx, out=dc)
        lines = self._buf("lines", B, 2, S, S)
        for plane, (c0, t1, t2) in ((0, self.br_binarize), (1, self.br_thresh)):  # cat((shrink, threshold)) :119
            b0 = self._buf("db0", B, 256, 256, 16)
            c0(dc, out=b0)
            b1 = self._buf("db1", B, 512, 512, 16)
            t1(b0, out=b1)
            t2(b1, out=lines[:, plane].unsqueeze(-1))
        mask_u8 = self._buf("mask_u8", B, S, S, dtype=torch.uint8)
        _lib.check(lib.mit_map_to_u8(mask.data_ptr(), mask_u8.data_ptr(), B * S * S, 0, 0.0, st), "mit_map_to_u8")
        if taps is not None:
            taps["mask_f32"] = mask.clone()
        self.last_mask_f32 = mask[..., 0]  # [B,S,S] view of the workspace, valid until the next forward (the tiled path averages floats)
        return mask_u8[:, :S - dh, :S - dw], lines[:, :, :S - dh, :S - dw], (dw, dh)

    def shrink_bitmap(self, lines: torch.Tensor, thr: float = 0.3) -> torch.Tensor:
        """SegDetectorRepresenter.binarize (db_utils.py:75): pred[:, 0] > thresh, as u8 on the device."""
        shrink = lines[:, 0].contiguous()
        out = torch.empty(shrink.shape, dtype=torch.uint8, device=self.device)
        _lib.check(_lib.load().mit_map_to_u8(shrink.data_ptr(), out.data_ptr(), shrink.numel(), 1, thr,
                                             C.c_void_p(ops.current_stream())), "mit_map_to_u8")
        return out

    @staticmethod
    def flops_per_page() -> float:
        """Executed FLOPs of one 1024x1024 forward (YOLO 0-9 + heads), 2*MAC."""
        def conv(cin, cout, k, hw):
            return 2.0 * cin * cout * k * k * hw

        def c3(c1, c2, n, hw):
            c_ = c2 // 2
            return conv(c1, c_, 1, hw) * 2 + conv(2 * c_, c2, 1, hw) + n * (conv(c_, c_, 1, hw) + conv(c_, c_, 3, hw))

        f = conv(3, 32, 6, 512 ** 2) + conv(32, 64, 3, 256 ** 2) + c3(64, 64, 1, 256 ** 2) + conv(64, 128, 3, 128 ** 2)
        f += c3(128, 128, 2, 128 ** 2) + conv(128, 256, 3, 64 ** 2) + c3(256, 256, 3, 64 ** 2) + conv(256, 512, 3, 32 ** 2)
        f += c3(512, 512, 1, 32 ** 2) + conv(512, 256, 1, 32 ** 2) + conv(1024, 512, 1, 32 ** 2)

        def up(cin, mid, cout, hw):  # C3 + ConvT k4 s2 (flop counter convention: 2*in_px*k*k*Cin*Cout)
            return c3(cin, mid, 1, hw) + conv(mid, cout, 4, hw)

        f += c3(512, 512, 1, 16 ** 2) + up(512, 512, 256, 16 ** 2) + up(768, 512, 256, 32 ** 2) + up(512, 512, 256, 64 ** 2)
        f += up(384, 256, 128, 128 ** 2) + up(192, 128, 64, 256 ** 2) + conv(64, 1, 4, 512 ** 2)
        f += up(512, 512, 256, 64 ** 2) + up(384, 256, 128, 128 ** 2) + conv(128, 64, 1, 256 ** 2)
        f += 2 * (conv(64, 16, 3, 256 ** 2) + conv(16, 16, 2, 256 ** 2) + conv(16, 1, 2, 512 ** 2))
        return f
