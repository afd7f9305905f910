"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference ``default`` detector network.

Functional fp32 torch-CPU restatement of ``TextDetection.forward``
(/root/reference/manga_translator/detection/default_utils/DBNet_resnet34.py:98-125), ``DBHead.forward`` (eval,
default_utils/DBHead.py:25-33) and the tensor part of ``det_batch_forward_default`` (detection/default.py:15-25), driven by a
state_dict with the reference's key names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Parity status: everything but the ResNet-34 definition is pinned against the reference module imported in the build
container (tests/golden/dbnet.npz): torchvision is installed nowhere we can run, so ``torchvision.models.resnet34`` is
restated here (``ResNet34`` below, the standard BasicBlock [3, 4, 6, 3] network) and injected into the reference import —
that restatement itself is **unpinned**.  The OpenCV pre/post-processing (bilateralFilter, resize, SegDetectorRepresenter)
is host glue outside the dense path.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def resnet_features(sd: SD, x: torch.Tensor, p: str = "backbone"):
    """conv1/bn1/relu/maxpool + layer1..4 as used by TextDetection.forward :99-107 -> (h4, h8, h16, h32)."""
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=3), sd, p + ".bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (planes, n, stride) in enumerate(LAYERS, start=1):
        for b in range(n):  # torchvision BasicBlock.forward
            q = f"{p}.layer{li}.{b}"
            s = stride if b == 0 else 1
            out = F.relu(_bn(F.conv2d(x, sd[q + ".conv1.weight"], stride=s, padding=1), sd, q + ".bn1"))
            out = _bn(F.conv2d(out, sd[q + ".conv2.weight"], padding=1), sd, q + ".bn2")
            idt = x
            if (q + ".downsample.0.weight") in sd:
                idt = _bn(F.conv2d(x, sd[q + ".downsample.0.weight"], stride=s), sd, q + ".downsample.1")
            x = F.relu(out + idt)
        feats.append(x)
    return feats


def _cbr(x, sd, p, i):
    return F.relu(_bn(F.conv2d(x, sd[f"{p}.conv.{i}.weight"], padding=1), sd, f"{p}.conv.{i + 1}"))


def double_conv(x, sd, p):  # :22-52 (stride 2: AvgPool2d(2, 2) first)
    x = F.avg_pool2d(x, 2, 2)
    return _cbr(_cbr(_cbr(x, sd, p, 0), sd, p, 3), sd, p, 6)


def double_conv_up(x, sd, p):  # :54-75
    x = _cbr(_cbr(x, sd, p, 0), sd, p, 3)
    x = F.conv_transpose2d(x, sd[p + ".conv.6.weight"], None, stride=2, padding=1)
    return F.relu(_bn(x, sd, p + ".conv.7"))


def db_head(x, sd, p="conv_db"):
    """DBHead.forward in eval mode (DBHead.py:25-33): cat(shrink logits, sigmoid threshold map)."""
    def branch(q, first_bias, last_sigmoid):
        y = F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"] if first_bias else None, padding=1)
        y = F.relu(_bn(y, sd, q + ".1"))
        y = F.relu(_bn(F.conv_transpose2d(y, sd[q + ".3.weight"], sd[q + ".3.bias"], stride=2, padding=1), sd, q + ".4"))
        y = F.conv_transpose2d(y, sd[q + ".6.weight"], sd[q + ".6.bias"], stride=2, padding=1)
        return torch.sigmoid(y) if last_sigmoid else y
    return torch.cat((branch(p + ".binarize", True, False), branch(p + ".thresh", False, True)), dim=1)


def text_detection_forward(sd: SD, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """TextDetection.forward :98-125 -> (db [B,2,H,W] (shrink logits, threshold), mask [B,1,H/2,W/2])."""
    h4, h8, h16, h32 = resnet_features(sd, x)
    h64 = double_conv(h32, sd, "down_conv1")
    h128 = double_conv(h64, sd, "down_conv2")
    h256 = double_conv(h128, sd, "down_conv3")
    up256 = double_conv_up(h256, sd, "upconv1")
    up128 = double_conv_up(torch.cat([up256, h128], 1), sd, "upconv2")
    up64 = double_conv_up(torch.cat([up128, h64], 1), sd, "upconv3")
    up32 = double_conv_up(torch.cat([up64, h32], 1), sd, "upconv4")
    up16 = double_conv_up(torch.cat([up32, h16], 1), sd, "upconv5")
    up8 = double_conv_up(torch.cat([up16, h8], 1), sd, "upconv6")
    up4 = double_conv_up(torch.cat([up8, h4], 1), sd, "upconv7")
    m = up4
    for i in (0, 2, 4):
        m = F.relu(F.conv2d(m, sd[f"conv_mask.{i}.weight"], sd[f"conv_mask.{i}.bias"], padding=1))
    m = torch.sigmoid(F.conv2d(m, sd["conv_mask.6.weight"], sd["conv_mask.6.bias"]))
    return db_head(up8, sd), m


def det_batch_forward(sd: SD, batch_u8: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """det_batch_forward_default (default.py:15-25): u8 [N,H,W,3] -> (db.sigmoid() [N,2,H,W], mask [N,1,H/2,W/2])."""
    x = torch.from_numpy(batch_u8.astype(np.float32) / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        db, mask = text_detection_forward(sd, x)
    return db.sigmoid().numpy(), mask.numpy()


# ---- nn.Module form of torchvision's resnet34, only for injection into the reference import (ref_import.dbnet) ----
class _BasicBlock(nn.Module):
    def __init__(self, inpl, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inpl != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + idt)


class ResNet34(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for li, (planes, n, stride) in enumerate(LAYERS, start=1):
            blocks = []
            for b in range(n):
                blocks.append(_BasicBlock(inpl, planes, stride if b == 0 else 1))
                inpl = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def resnet34(pretrained=False, **_):
    return ResNet34()
