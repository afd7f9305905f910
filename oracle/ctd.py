"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference ComicTextDetector network.

Functional fp32 torch-CPU restatement of ``TextDetBase.forward``
(/root/reference/manga_translator/detection/ctd_utils/basemodel.py:234-238): fused YOLOv5s
backbone (yolov5/yolo.py:115-134, common.py:30-49,94-135,181-197), ``UnetHead`` (:56-72) and
``DBHead`` (:100-119), plus the tensor pre/post of ``ComicTextDetector._infer`` (ctd.py:129-179) that
does not need OpenCV.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

YOLO layers 10-24 + Detect only feed ``blks``, which ``_infer`` discards (ctd.py:142,150-151): the
restatement (and the engine) stop at layer 9; tests/test_oracle_vs_reference.py checks ``mask`` and
``lines`` against the full reference forward.  Parity status: pinned against the reference module
imported in the build container; no golden vectors exist in the reference's own tests.  The
cv2.resize restatement used by letterbox is "parity unpinned" (cv2 is not installed).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def fuse_conv_bn(sd: SD, p: str, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """fuse_conv_and_bn (ctd_utils/utils/yolov5_utils.py:22-42), fp32 like the reference."""
    w = sd[p + ".conv.weight"]
    bw, bb, rm, rv = sd[p + ".bn.weight"], sd[p + ".bn.bias"], sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"]
    w_bn = torch.diag(bw.div(torch.sqrt(eps + rv)))
    fw = torch.mm(w_bn, w.view(w.shape[0], -1)).view(w.shape)
    b_conv = torch.zeros(w.shape[0])
    fb = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + (bb - bw.mul(rm).div(torch.sqrt(rv + eps)))
    return fw, fb


class _Yolo:
    """Fused (Conv+BN folded, SiLU) yolov5 blocks; BN eps = 1e-3 (initialize_weights, yolov5_utils.py:52-56)."""

    def __init__(self, sd: SD):
        self.sd = sd
        self.cache = {}

    def conv(self, x, p, k, s):
        if p not in self.cache:
            self.cache[p] = fuse_conv_bn(self.sd, p, 1e-3)
        w, b = self.cache[p]
        return F.silu(F.conv2d(x, w, b, stride=s, padding=k // 2 if k != 6 else 2))  # Conv.forward_fuse common.py:48-49

    def c3(self, x, p, n, shortcut=True):
        y = self.conv(x, p + ".cv1", 1, 1)
        for j in range(n):  # Bottleneck common.py:94-105
            t = self.conv(self.conv(y, f"{p}.m.{j}.cv1", 1, 1), f"{p}.m.{j}.cv2", 3, 1)
            y = y + t if shortcut else t
        return self.conv(torch.cat((y, self.conv(x, p + ".cv2", 1, 1)), dim=1), p + ".cv3", 1, 1)  # common.py:135-136

    def sppf(self, x, p):
        x = self.conv(x, p + ".cv1", 1, 1)
        y1 = F.max_pool2d(x, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        return self.conv(torch.cat([x, y1, y2, F.max_pool2d(y2, 5, 1, 2)], 1), p + ".cv2", 1, 1)  # common.py:190-197


def yolo_features(sd: SD, x: torch.Tensor) -> List[torch.Tensor]:
    """Model._forward_once (yolo.py:115-134) up to layer 9 with out_indices [1,3,5,7,9]."""
    y = _Yolo(sd)
    x = y.conv(x, "model.0", 6, 2)
    f160 = x = y.conv(x, "model.1", 3, 2)
    x = y.c3(x, "model.2", 1)
    f80 = x = y.conv(x, "model.3", 3, 2)
    x = y.c3(x, "model.4", 2)
    f40 = x = y.conv(x, "model.5", 3, 2)
    x = y.c3(x, "model.6", 3)
    f20 = x = y.conv(x, "model.7", 3, 2)
    x = y.c3(x, "model.8", 1)
    f3 = y.sppf(x, "model.9")
    return [f160, f80, f40, f20, f3]


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _hconv(x, sd, p, k):
    """heads' ``Conv`` (common.py:30-46) unfused: conv -> BN(eps 1e-5) -> LeakyReLU(0.1)."""
    return F.leaky_relu(_bn(F.conv2d(x, sd[p + ".conv.weight"], None, padding=k // 2), sd, p + ".bn"), 0.1)


def _hc3(x, sd, p):
    y = _hconv(x, sd, p + ".cv1", 1)
    y = y + _hconv(_hconv(y, sd, p + ".m.0.cv1", 1), sd, p + ".m.0.cv2", 3)
    return _hconv(torch.cat((y, _hconv(x, sd, p + ".cv2", 1)), dim=1), sd, p + ".cv3", 1)


def _up_c3(x, sd, p):
    """double_conv_up_c3.forward (basemodel.py:15-26)."""
    x = _hc3(x, sd, p + ".conv.0")
    x = F.conv_transpose2d(x, sd[p + ".conv.1.weight"], None, stride=2, padding=1)
    return torch.relu(_bn(x, sd, p + ".conv.2"))


def unet_head(sd: SD, f160, f80, f40, f20, f3):
    """UnetHead.forward in TEXTDET_INFERENCE mode (basemodel.py:56-72)."""
    d10 = _hc3(F.avg_pool2d(f3, 2, 2), sd, "down_conv1.conv")  # double_conv_c3 :28-39
    u20 = _up_c3(d10, sd, "upconv0")
    u40 = _up_c3(torch.cat([f20, u20], dim=1), sd, "upconv2")
    u80 = _up_c3(torch.cat([f40, u40], dim=1), sd, "upconv3")
    u160 = _up_c3(torch.cat([f80, u80], dim=1), sd, "upconv4")
    u320 = _up_c3(torch.cat([f160, u160], dim=1), sd, "upconv5")
    mask = torch.sigmoid(F.conv_transpose2d(u320, sd["upconv6.0.weight"], None, stride=2, padding=1))
    return mask, [f80, f40, u40]


def db_head(sd: SD, f80, f40, u40):
    """DBHead.forward, eval, step_eval=False (basemodel.py:100-119)."""
    u80 = _up_c3(torch.cat([f40, u40], dim=1), sd, "upconv3")
    x = _up_c3(torch.cat([f80, u80], dim=1), sd, "upconv4")
    x = torch.relu(_bn(F.conv2d(x, sd["conv.0.weight"], sd["conv.0.bias"]), sd, "conv.1"))

    def branch(p, first_bias):
        t = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"] if first_bias else None, padding=1)
        t = torch.relu(_bn(t, sd, p + ".1"))
        t = F.conv_transpose2d(t, sd[p + ".3.weight"], sd[p + ".3.bias"], stride=2)
        t = torch.relu(_bn(t, sd, p + ".4"))
        return torch.sigmoid(F.conv_transpose2d(t, sd[p + ".6.weight"], sd[p + ".6.bias"], stride=2))

    threshold_maps = branch("thresh", False)  # :106, _init_thresh :125-137
    shrink_maps = branch("binarize", True)  # :107-108
    return torch.cat((shrink_maps, threshold_maps), dim=1)  # :119


def textdet_forward(yolo_sd: SD, seg_sd: SD, det_sd: SD, x: torch.Tensor, taps=None):
    """TextDetBase.forward (basemodel.py:234-238) -> (mask [B,1,S,S], lines [B,2,S,S])."""
    feats = yolo_features(yolo_sd, x)
    if taps is not None:
        for n, f in zip(("f160", "f80", "f40", "f20", "f3"), feats):
            taps[n] = f
    mask, feats2 = unet_head(seg_sd, *feats)
    lines = db_head(det_sd, *feats2)
    return mask, lines


# ---- tensor pre/post of ComicTextDetector._infer that needs no contour code ----

def resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC.

    Restated from OpenCV's resize.cpp: an exact 2x shrink is routed to the 2x2 box average
    ((a+b+c+d+2)>>2); otherwise fixed-point bilinear with 11-bit coefficients, pixel centres
    aligned ((d+0.5)*scale-0.5), edge clamped, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2.
    """
    dw, dh = dsize
    sh, sw = src.shape[:2]
    if sh == 2 * dh and sw == 2 * dw:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(n_src, n_dst):
        scale = n_src / n_dst
        idx = np.zeros(n_dst, dtype=np.int64)
        co = np.zeros((n_dst, 2), dtype=np.int64)
        for d in range(n_dst):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= n_src - 1:
                s, f = n_src - 1, np.float32(0)
            idx[d] = s
            co[d, 0] = int(np.rint(np.float32((np.float32(1.0) - f) * np.float32(2048))))
            co[d, 1] = int(np.rint(np.float32(f * np.float32(2048))))
        return idx, co

    yi, yc = taps(sh, dh)
    xi, xc = taps(sw, dw)
    s = src.astype(np.int64)
    x1 = np.minimum(xi + 1, sw - 1)
    rows = s[:, xi] * xc[:, 0][None, :, None] + s[:, x1] * xc[:, 1][None, :, None]  # [sh, dw, C]
    y1 = np.minimum(yi + 1, sh - 1)
    out = (((yc[:, 0][:, None, None] * (rows[yi] >> 4)) >> 16) + ((yc[:, 1][:, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(im: np.ndarray, new_shape=(1024, 1024)):
    """letterbox(auto=False) (ctd_utils/utils/imgproc_utils.py:69-100): resize, pad bottom/right with 0."""
    shape = im.shape[:2]
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad)
    out = np.zeros((new_shape[0], new_shape[1], im.shape[2]), dtype=np.uint8)
    out[:im.shape[0], :im.shape[1]] = im
    return out, (r, r), (int(dw), int(dh))


def preprocess_img(img: np.ndarray, input_size=(1024, 1024)):
    """preprocess_img (ctd.py:17-28): the BGR2RGB + [::-1] pair cancels, so channels stay as given."""
    img_in, ratio, (dw, dh) = letterbox(img, new_shape=input_size)
    t = img_in.transpose((2, 0, 1))
    t = np.array([np.ascontiguousarray(t)]).astype(np.float32) / 255
    return torch.from_numpy(t), ratio, int(dw), int(dh)


def infer_maps(yolo_sd: SD, seg_sd: SD, det_sd: SD, image: np.ndarray, taps=None):
    """ComicTextDetector._infer :137-155 up to the OpenCV post-processing: returns
    (mask u8 [1024-dh, 1024-dw], lines f32 [1,2,1024-dh,1024-dw])."""
    img_in, ratio, dw, dh = preprocess_img(image)
    with torch.no_grad():
        mask, lines = textdet_forward(yolo_sd, seg_sd, det_sd, img_in, taps)
    mask = mask.squeeze()
    mask = mask[..., :mask.shape[0] - dh, :mask.shape[1] - dw]  # :152
    lines = lines[..., :lines.shape[2] - dh, :lines.shape[3] - dw]  # :153
    mask_u8 = (mask.numpy() * 255).astype(np.uint8)  # postprocess_mask :30-44
    return mask_u8, lines.numpy()
