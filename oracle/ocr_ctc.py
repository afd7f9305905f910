"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference ``48px_ctc`` OCR model.

Functional fp32 torch-CPU restatement of ``OCR.decode`` / ``decode_ctc_top1``
(/root/reference/manga_translator/ocr/model_48px_ctc.py:463-494), the FAN ResNet backbone (:277-403) and the custom
encoder layer (:180-275), driven by a state_dict with the reference's key names.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

Parity status: pinned against the reference module imported in the build container (tests/golden/ocr_ctc.npz, made by
oracle/make_golden.py).  Quirk kept: the encoder runs WITHOUT a key-padding mask on chunks zero-padded to max_w+7+128
(:84, :450-451), so a line's logits depend on its chunk's width.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
EMBD, HEADS = 320, 8
CHANNELS = [80, 160, 320, 320]
LAYERS = [4, 6, 8, 6]


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _block(x, sd: SD, p: str):
    """BasicBlock.forward :389-403 (pre-activation)."""
    out = F.conv2d(F.relu(_bn(x, sd, p + ".bn1")), sd[p + ".conv1.weight"], padding=1)
    out = F.conv2d(F.relu(_bn(out, sd, p + ".bn2")), sd[p + ".conv2.weight"], padding=1)
    res = x
    if (p + ".downsample.1.weight") in sd:
        res = F.conv2d(_bn(x, sd, p + ".downsample.0"), sd[p + ".downsample.1.weight"])
    return out + res


def backbone(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """ResNet.forward :335-370."""
    p = "backbone.ConvNet"
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv0_1.weight"], padding=1), sd, p + ".bn0_1"))
    x = F.conv2d(x, sd[p + ".conv0_2.weight"], padding=1)
    pools = [lambda t: F.avg_pool2d(t, 2, 2), lambda t: F.avg_pool2d(t, 2, 2),
             lambda t: F.avg_pool2d(t, 2, stride=(2, 1), padding=(0, 1)), None]
    for li in range(1, 5):
        if pools[li - 1] is not None:
            x = pools[li - 1](x)
        for b in range(LAYERS[li - 1]):
            x = _block(x, sd, f"{p}.layer{li}.{b}")
        if li < 4:
            x = F.conv2d(F.relu(_bn(x, sd, f"{p}.bn{li}")), sd[f"{p}.conv{li}.weight"], padding=1)
    x = F.conv2d(F.relu(_bn(x, sd, p + ".bn4_1")), sd[p + ".conv4_1.weight"], stride=(2, 1), padding=(1, 1))
    x = F.conv2d(F.relu(_bn(x, sd, p + ".bn4_2")), sd[p + ".conv4_2.weight"])
    return _bn(x, sd, p + ".bn4_3")


def encoder_layer(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """CustomTransformerEncoderLayer.forward :237-257 (norm_first, no masks, eval: dropout off)."""
    N, T, _ = x.shape
    h = F.layer_norm(x, (EMBD,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    hp = h + sd[p + ".pe.pe"][:, :T]  # PE on query and key only (:263-265)
    w, b = sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"]
    q = F.linear(hp, w[:EMBD], b[:EMBD]).view(N, T, HEADS, -1).transpose(1, 2)
    k = F.linear(hp, w[EMBD:2 * EMBD], b[EMBD:2 * EMBD]).view(N, T, HEADS, -1).transpose(1, 2)
    v = F.linear(h, w[2 * EMBD:], b[2 * EMBD:]).view(N, T, HEADS, -1).transpose(1, 2)
    att = torch.softmax((q * (q.shape[-1] ** -0.5)) @ k.transpose(-1, -2), dim=-1) @ v
    att = att.transpose(1, 2).reshape(N, T, EMBD)
    x = x + F.linear(att, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    h = F.layer_norm(x, (EMBD,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return x + h


def forward(sd: SD, img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """OCR.forward :463-471 -> (char logits [N,T,dict], colour values [N,T,6])."""
    feats = backbone(sd, img).squeeze(2).permute(0, 2, 1)
    for i in range(3):
        feats = encoder_layer(feats, sd, f"encoders.layers.{i}")
    h = F.gelu(F.layer_norm(feats, (EMBD,), sd["char_pred_norm.0.weight"], sd["char_pred_norm.0.bias"], 1e-5))
    return F.linear(h, sd["char_pred.weight"], sd["char_pred.bias"]), F.linear(feats, sd["color_pred1.0.weight"], sd["color_pred1.0.bias"])


def decode_ctc_top1(logits: torch.Tensor, colors: torch.Tensor, blank: int = 0) -> List[List[tuple]]:
    """decode_ctc_top1 :473-494: greedy argmax, collapse repeats, drop blanks; (char id, log-prob, 6 clamped colours)."""
    logprobs = logits.log_softmax(2)
    _, idx = logprobs.max(2)
    colors = colors.clamp(0, 1)
    out: List[List[tuple]] = []
    for b in range(logits.shape[0]):
        line, last = [], blank
        for t in range(logits.shape[1]):
            ch = int(idx[b, t])
            if ch != last and ch != blank:
                line.append((ch, float(logprobs[b, t, ch]), *[float(c) for c in colors[b, t]]))
            last = ch
        out.append(line)
    return out


def make_chunks(region_imgs: List[np.ndarray], max_chunk_size: int = 16):
    """Host batching of Model48pxCTCOCR._infer :77-104: as the 48px model but padded to max_w + 7 + 128 (:84)."""
    perm = sorted(range(len(region_imgs)), key=lambda i: region_imgs[i].shape[1])
    for c in range(0, len(perm), max_chunk_size):
        indices = perm[c:c + max_chunk_size]
        widths = [region_imgs[i].shape[1] for i in indices]
        max_width = (4 * (max(widths) + 7) // 4) + 128
        region = np.zeros((len(indices), 48, max_width, 3), dtype=np.uint8)
        for j, i in enumerate(indices):
            region[j, :, :widths[j], :] = region_imgs[i]
        t = (torch.from_numpy(region).float() - 127.5) / 127.5
        yield indices, widths, t.permute(0, 3, 1, 2).contiguous()
