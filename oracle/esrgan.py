"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference ESRGAN upscaler (RRDBNet, 4x).

Functional fp32 torch-CPU restatement of ``RRDBNet.forward`` and the tensor part of ``ESRGANUpscalerPytorch._infer``
(/root/reference/manga_translator/upscaling/esrgan_pytorch.py:28-167,537-549), driven by a state_dict with the
reference's key names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: pinned against the reference module imported in the build container (tests/golden/esrgan.npz, made by
oracle/make_golden.py).  The final PIL ``Image.resize(BILINEAR)`` by ratio/4 is host glue outside the dense path and is
not restated.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _conv(x, sd: SD, p: str, act: bool):
    y = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)  # conv_block: zero pad, bias (:345-380)
    return F.leaky_relu(y, 0.2) if act else y


def rdb(x, sd: SD, p: str):
    """ResidualDenseBlock_5C.forward (:152-166), plus=False."""
    x1 = _conv(x, sd, p + ".conv1.0", True)
    x2 = _conv(torch.cat((x, x1), 1), sd, p + ".conv2.0", True)
    x3 = _conv(torch.cat((x, x1, x2), 1), sd, p + ".conv3.0", True)
    x4 = _conv(torch.cat((x, x1, x2, x3), 1), sd, p + ".conv4.0", True)
    x5 = _conv(torch.cat((x, x1, x2, x3, x4), 1), sd, p + ".conv5.0", False)
    return x5 * 0.2 + x


def rrdbnet_forward(sd: SD, x: torch.Tensor, nb: int) -> torch.Tensor:
    """RRDBNet.forward (:67-75) for in_nc = 3 (no pixel-unshuffle), upscale 4, upconv blocks."""
    fea = _conv(x, sd, "model.0", False)
    t = fea
    for i in range(nb):  # RRDB.forward (:103-112)
        o = t
        for r in (1, 2, 3):
            o = rdb(o, sd, f"model.1.sub.{i}.RDB{r}")
        t = o * 0.2 + t
    t = fea + _conv(t, sd, f"model.1.sub.{nb}", False)  # ShortcutBlock
    for idx in (3, 6):  # upconv_block (:317-324): nearest x2 + conv + LeakyReLU
        t = _conv(F.interpolate(t, scale_factor=2, mode="nearest"), sd, f"model.{idx}", True)
    t = _conv(t, sd, "model.8", True)
    return _conv(t, sd, "model.10", False)


def infer(sd: SD, image_rgb: np.ndarray, nb: int) -> np.ndarray:
    """_infer (:537-546) up to the PIL resize: RGB u8 [H,W,3] -> 4x RGB u8 [4H,4W,3] (BGR inside the network)."""
    x = torch.from_numpy(image_rgb[:, :, ::-1].copy()).float().div(255.0).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        y = rrdbnet_forward(sd, x, nb)[0]
    out = (y.clip(0, 1).permute(1, 2, 0).numpy()[:, :, ::-1].copy() * 255.0).astype(np.uint8)
    return out
