"""State-dict layout of the reference ComicTextDetector checkpoint (``comictextdetector.pt``).

The checkpoint is ``{'blk_det': {'cfg': ..., 'weights': sd}, 'text_seg': sd, 'text_det': sd}``
(/root/reference/manga_translator/detection/ctd_utils/basemodel.py:205-214, yolov5/yolo.py:286-310).
The YOLOv5 cfg lives inside the checkpoint; offline we carry the yolov5s-v6 layout the survey
identified (nc=2, depth .33, width .50; output [1,64512,7] for 1024^2 matches DEFAULT_LANG_LIST).
tests/test_schema.py pins every name/shape against the reference modules built from this cfg.
"""
from __future__ import annotations

from .synth import Schema, bn_entries

YOLOV5S_CFG = dict(
    nc=2, depth_multiple=0.33, width_multiple=0.50,
    anchors=[[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]],
    backbone=[[-1, 1, 'Conv', [64, 6, 2, 2]], [-1, 1, 'Conv', [128, 3, 2]], [-1, 3, 'C3', [128]],
              [-1, 1, 'Conv', [256, 3, 2]], [-1, 6, 'C3', [256]], [-1, 1, 'Conv', [512, 3, 2]], [-1, 9, 'C3', [512]],
              [-1, 1, 'Conv', [1024, 3, 2]], [-1, 3, 'C3', [1024]], [-1, 1, 'SPPF', [1024, 5]]],
    head=[[-1, 1, 'Conv', [512, 1, 1]], [-1, 1, 'nn.Upsample', [None, 2, 'nearest']], [[-1, 6], 1, 'Concat', [1]],
          [-1, 3, 'C3', [512, False]], [-1, 1, 'Conv', [256, 1, 1]], [-1, 1, 'nn.Upsample', [None, 2, 'nearest']],
          [[-1, 4], 1, 'Concat', [1]], [-1, 3, 'C3', [256, False]], [-1, 1, 'Conv', [256, 3, 2]],
          [[-1, 14], 1, 'Concat', [1]], [-1, 3, 'C3', [512, False]], [-1, 1, 'Conv', [512, 3, 2]],
          [[-1, 10], 1, 'Concat', [1]], [-1, 3, 'C3', [1024, False]], [[17, 20, 23], 1, 'Detect', ['nc', 'anchors']]])


def _conv(prefix: str, c1: int, c2: int, k: int) -> Schema:
    """yolov5 ``Conv`` (common.py:30-49): conv (no bias) + BatchNorm2d."""
    return [(prefix + ".conv.weight", (c2, c1, k, k), "conv")] + bn_entries(prefix + ".bn", c2)


def _c3(prefix: str, c1: int, c2: int, n: int) -> Schema:
    """yolov5 ``C3`` (common.py:126-136)."""
    c_ = c2 // 2
    s = _conv(prefix + ".cv1", c1, c_, 1) + _conv(prefix + ".cv2", c1, c_, 1) + _conv(prefix + ".cv3", 2 * c_, c2, 1)
    for j in range(n):
        s += _conv(f"{prefix}.m.{j}.cv1", c_, c_, 1) + _conv(f"{prefix}.m.{j}.cv2", c_, c_, 3)
    return s


# (layer index, kind, c1, c2, k / n) of the yolov5s-v6 graph at width 0.5 / depth 0.33
YOLO_LAYERS = [
    (0, "conv", 3, 32, 6), (1, "conv", 32, 64, 3), (2, "c3", 64, 64, 1), (3, "conv", 64, 128, 3), (4, "c3", 128, 128, 2),
    (5, "conv", 128, 256, 3), (6, "c3", 256, 256, 3), (7, "conv", 256, 512, 3), (8, "c3", 512, 512, 1),
    (9, "sppf", 512, 512, 5),
    (10, "conv", 512, 256, 1), (13, "c3", 512, 256, 1), (14, "conv", 256, 128, 1), (17, "c3", 256, 128, 1),
    (18, "conv", 128, 128, 3), (20, "c3", 256, 256, 1), (21, "conv", 256, 256, 3), (23, "c3", 512, 512, 1),
]


def layers_from_cfg(cfg: dict):
    """The (index, kind, c1, c2, k | n, stride) table of a yolov5 model cfg as the reference builds it from the checkpoint
    (``Model(ckpt['cfg'])``, ctd_utils/yolov5/yolo.py:286-292 -> ``parse_model`` :208-259): output channels are
    ``ceil(c2 * width_multiple / 8) * 8``, repeat counts above 1 become ``max(round(n * depth_multiple), 1)``, Concat sums its
    sources, Upsample / anything else keeps the channel count.  Only Conv / C3 / SPPF rows are returned (they own the weights)."""
    import math

    gd, gw = float(cfg["depth_multiple"]), float(cfg["width_multiple"])
    anchors = cfg["anchors"]
    na = len(anchors[0]) // 2 if isinstance(anchors, (list, tuple)) else int(anchors)
    no = na * (int(cfg["nc"]) + 5)
    ch = [int(cfg.get("ch", 3))]
    rows = []
    for i, (f, n, m, args) in enumerate(list(cfg["backbone"]) + list(cfg["head"])):
        name = m if isinstance(m, str) else getattr(m, "__name__", str(m))
        args = [no if a == "no" else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if name in ("Conv", "C3", "SPPF"):
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = int(math.ceil(c2 * gw / 8) * 8)
            if name == "Conv":
                rows.append((i, "conv", c1, c2, args[1] if len(args) > 1 else 1, args[2] if len(args) > 2 else 1))
            elif name == "C3":
                rows.append((i, "c3", c1, c2, n, 1))
            else:
                rows.append((i, "sppf", c1, c2, args[1] if len(args) > 1 else 5, 1))
        elif name == "Concat":
            c2 = sum(ch[x] for x in f)
        elif name == "Detect":
            c2 = ch[f[0]] if isinstance(f, (list, tuple)) else ch[f]
        else:
            c2 = ch[f]
        if i == 0:
            ch = []
        ch.append(c2)
    return rows


# stride of the Conv rows of YOLO_LAYERS (the yolov5s-v6 graph): every 3x3 / 6x6 Conv of the backbone and the two PAN down-convolutions
_YOLO_STRIDES = {0: 2, 1: 2, 3: 2, 5: 2, 7: 2, 10: 1, 14: 1, 18: 2, 21: 2}


def check_yolo_cfg(cfg) -> None:
    """Raise unless the checkpoint's own model cfg (``ckpt['blk_det']['cfg']``) describes the graph ``CtdEngine`` executes: the engine —
    like the reference's ``UnetHead`` / ``DBHead``, which hard-code the channel counts of the five tapped features (basemodel.py:41-72) —
    is built for the yolov5s-v6 layout; a checkpoint with another layout must fail here, loudly, not late or silently."""
    if not isinstance(cfg, dict):
        raise ValueError(f"comictextdetector.pt: blk_det.cfg is {type(cfg).__name__}, expected the yolov5 model dict")
    want = [(i, k, c1, c2, kn, _YOLO_STRIDES.get(i, 1)) for i, k, c1, c2, kn in YOLO_LAYERS]
    got = layers_from_cfg(cfg)
    if got != want:
        diff = [f"layer {g[0]}: checkpoint {g[1:]} != engine {w[1:]}" for g, w in zip(got, want) if g != w]
        if len(got) != len(want):
            diff.append(f"{len(got)} weighted layers in the checkpoint, {len(want)} in the engine")
        raise ValueError("comictextdetector.pt: the checkpoint's yolov5 cfg is not the yolov5s-v6 graph this engine implements: " + "; ".join(diff[:6]))


def yolo_schema() -> Schema:
    s: Schema = []
    for i, kind, c1, c2, kn in YOLO_LAYERS:
        p = f"model.{i}"
        if kind == "conv":
            s += _conv(p, c1, c2, kn)
        elif kind == "c3":
            s += _c3(p, c1, c2, kn)
        else:  # SPPF (common.py:181-197)
            s += _conv(p + ".cv1", c1, c1 // 2, 1) + _conv(p + ".cv2", c1 * 2, c2, 1)
    s.append(("model.24.anchors", (3, 3, 2), "normal"))
    for j, c in enumerate((128, 256, 512)):
        s.append((f"model.24.m.{j}.weight", (21, c, 1, 1), "conv"))
        s.append((f"model.24.m.{j}.bias", (21,), "bias"))
    return s


def _up_c3(prefix: str, in_ch: int, mid_ch: int, out_ch: int) -> Schema:
    """``double_conv_up_c3`` (basemodel.py:15-26): C3 -> ConvTranspose2d(k4 s2 p1, no bias) -> BN -> ReLU."""
    return (_c3(prefix + ".conv.0", in_ch + mid_ch, mid_ch, 1)
            + [(prefix + ".conv.1.weight", (mid_ch, out_ch, 4, 4), "convT")] + bn_entries(prefix + ".conv.2", out_ch))


def unet_head_schema() -> Schema:
    """``UnetHead`` (basemodel.py:41-72)."""
    s = _c3("down_conv1.conv", 512, 512, 1)
    s += _up_c3("upconv0", 0, 512, 256) + _up_c3("upconv2", 256, 512, 256) + _up_c3("upconv3", 0, 512, 256)
    s += _up_c3("upconv4", 128, 256, 128) + _up_c3("upconv5", 64, 128, 64)
    s.append(("upconv6.0.weight", (64, 1, 4, 4), "convT*6.0"))
    return s


def db_head_schema() -> Schema:
    """``DBHead(64)`` (basemodel.py:77-154)."""
    s = _up_c3("upconv3", 0, 512, 256) + _up_c3("upconv4", 128, 256, 128)
    s += [("conv.0.weight", (64, 128, 1, 1), "conv"), ("conv.0.bias", (64,), "bias")] + bn_entries("conv.1", 64)
    s += [("binarize.0.weight", (16, 64, 3, 3), "conv"), ("binarize.0.bias", (16,), "bias")] + bn_entries("binarize.1", 16)
    s += [("binarize.3.weight", (16, 16, 2, 2), "convT*0.5"), ("binarize.3.bias", (16,), "bias")] + bn_entries("binarize.4", 16)
    s += [("binarize.6.weight", (16, 1, 2, 2), "convT*3.0"), ("binarize.6.bias", (1,), "bias*-12.0")]
    s += [("thresh.0.weight", (16, 64, 3, 3), "conv")] + bn_entries("thresh.1", 16)
    s += [("thresh.3.weight", (16, 16, 2, 2), "convT*0.5"), ("thresh.3.bias", (16,), "bias")] + bn_entries("thresh.4", 16)
    s += [("thresh.6.weight", (16, 1, 2, 2), "convT*3.0"), ("thresh.6.bias", (1,), "bias")]
    return s


CTD_GAIN = 1.3  # conv gain for the synthetic ctd weights: keeps the SiLU/LeakyReLU stack from shrinking to a constant
