"""State-dict layout of the reference ``default`` detector (``detect-20241225.ckpt`` = ``TextDetection``).

/root/reference/manga_translator/detection/default_utils/DBNet_resnet34.py:77-125 (torchvision ResNet-34 backbone, three
average-pooled ``double_conv`` downs, seven ``double_conv_up`` blocks, ``conv_mask``) and default_utils/DBHead.py:8-70.
torchvision is not installed anywhere we can run, so the ResNet-34 key names are restated from its well-known layout
(conv1 / bn1 / layer{1..4}.{i}.{conv1,bn1,conv2,bn2,downsample.{0,1}} / fc); everything else is pinned against the
reference module (tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

from .synth import Schema, bn_entries

RESNET34_LAYERS = [(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]  # (planes, blocks, stride of the first block)


def resnet34_schema(prefix: str = "backbone") -> Schema:
    p = prefix
    s: Schema = [(f"{p}.conv1.weight", (64, 3, 7, 7), "conv")] + bn_entries(f"{p}.bn1", 64)
    inpl = 64
    for li, (planes, n, stride) in enumerate(RESNET34_LAYERS, start=1):
        for b in range(n):
            q = f"{p}.layer{li}.{b}"
            s += [(q + ".conv1.weight", (planes, inpl, 3, 3), "conv")] + bn_entries(q + ".bn1", planes)
            s += [(q + ".conv2.weight", (planes, planes, 3, 3), "conv")] + bn_entries(q + ".bn2", planes, "*0.4")
            if b == 0 and (stride != 1 or inpl != planes):
                s += [(q + ".downsample.0.weight", (planes, inpl, 1, 1), "conv")] + bn_entries(q + ".downsample.1", planes)
            inpl = planes
    s += [(f"{p}.fc.weight", (1000, 512), "linear"), (f"{p}.fc.bias", (1000,), "bias")]  # unused by TextDetection.forward
    return s


def _double_conv(p: str, cin: int, mid: int, out: int) -> Schema:
    return ([(f"{p}.conv.0.weight", (mid, cin, 3, 3), "conv")] + bn_entries(f"{p}.conv.1", mid)
            + [(f"{p}.conv.3.weight", (mid, mid, 3, 3), "conv")] + bn_entries(f"{p}.conv.4", mid)
            + [(f"{p}.conv.6.weight", (out, mid, 3, 3), "conv")] + bn_entries(f"{p}.conv.7", out))


def _double_conv_up(p: str, cin: int, mid: int, out: int) -> Schema:
    return ([(f"{p}.conv.0.weight", (mid, cin, 3, 3), "conv")] + bn_entries(f"{p}.conv.1", mid)
            + [(f"{p}.conv.3.weight", (mid, mid, 3, 3), "conv")] + bn_entries(f"{p}.conv.4", mid)
            + [(f"{p}.conv.6.weight", (mid, out, 4, 4), "convT")] + bn_entries(f"{p}.conv.7", out))


def text_detection_schema() -> Schema:
    s = resnet34_schema()
    # DBHead(64, 0) (DBHead.py:8-70): binarize has conv biases, thresh has none
    s += [("conv_db.binarize.0.weight", (16, 64, 3, 3), "conv"), ("conv_db.binarize.0.bias", (16,), "bias")] + bn_entries("conv_db.binarize.1", 16)
    s += [("conv_db.binarize.3.weight", (16, 16, 4, 4), "convT"), ("conv_db.binarize.3.bias", (16,), "bias")] + bn_entries("conv_db.binarize.4", 16)
    s += [("conv_db.binarize.6.weight", (16, 1, 4, 4), "convT*3.0"), ("conv_db.binarize.6.bias", (1,), "bias")]
    s += [("conv_db.thresh.0.weight", (16, 64, 3, 3), "conv")] + bn_entries("conv_db.thresh.1", 16)
    s += [("conv_db.thresh.3.weight", (16, 16, 4, 4), "convT"), ("conv_db.thresh.3.bias", (16,), "bias")] + bn_entries("conv_db.thresh.4", 16)
    s += [("conv_db.thresh.6.weight", (16, 1, 4, 4), "convT*3.0"), ("conv_db.thresh.6.bias", (1,), "bias")]
    for i, (cin, cout) in enumerate(((64, 64), (64, 64), (64, 32))):
        s += [(f"conv_mask.{2 * i}.weight", (cout, cin, 3, 3), "conv"), (f"conv_mask.{2 * i}.bias", (cout,), "bias")]
    s += [("conv_mask.6.weight", (1, 32, 1, 1), "conv*4.0"), ("conv_mask.6.bias", (1,), "bias")]
    for j in (1, 2, 3):
        s += _double_conv(f"down_conv{j}", 512, 512, 512)
    for name, cin, mid, out in (("upconv1", 512, 512, 256), ("upconv2", 768, 512, 256), ("upconv3", 768, 512, 256),
                                ("upconv4", 768, 512, 256), ("upconv5", 512, 256, 128), ("upconv6", 256, 128, 64),
                                ("upconv7", 128, 64, 64)):
        s += _double_conv_up(name, cin, mid, out)
    return s
