"""State-dict layout of the reference ``48px_ctc`` OCR model (``ocr-ctc.ckpt``).

``OCR(dictionary, max_len)`` of /root/reference/manga_translator/ocr/model_48px_ctc.py:447-461: FAN-style pre-activation
ResNet backbone (:277-370, BasicBlock :372-403, layers [4, 6, 8, 6]) -> 3 x CustomTransformerEncoderLayer(320, 8 heads,
FFN 1280, GELU, norm_first, sinusoidal PE on q/k :180-275) -> LayerNorm + GELU + Linear(dict) and Linear(6) colour head.
tests/test_oracle_vs_reference.py pins every name and shape against the reference module's own state_dict.
"""
from __future__ import annotations

import math

from .synth import Schema, bn_entries

EMBD, FFN, HEADS, N_LAYERS = 320, 1280, 8, 3
CTC_GAIN = 1.5  # conv / linear gain of the synthetic weights: with 1.0 the greedy path collapses to one class per line
CHANNELS = [80, 160, 320, 320]
LAYERS = [4, 6, 8, 6]


def resnet_schema(prefix: str = "backbone.ConvNet") -> Schema:
    p = prefix
    s: Schema = [(f"{p}.conv0_1.weight", (40, 3, 3, 3), "conv")] + bn_entries(f"{p}.bn0_1", 40)
    s += [(f"{p}.conv0_2.weight", (40, 40, 3, 3), "conv")]
    inpl = 40
    for li, (planes, n) in enumerate(zip(CHANNELS, LAYERS), start=1):
        for b in range(n):
            q = f"{p}.layer{li}.{b}"
            # residual branches would grow ~1.4x per block with unit BN gains: damp the second BN so 24 blocks stay O(1)
            s += bn_entries(q + ".bn1", inpl) + [(q + ".conv1.weight", (planes, inpl, 3, 3), "conv")]
            s += bn_entries(q + ".bn2", planes, "*0.3") + [(q + ".conv2.weight", (planes, planes, 3, 3), "conv")]
            if b == 0 and inpl != planes:
                s += bn_entries(q + ".downsample.0", inpl) + [(q + ".downsample.1.weight", (planes, inpl, 1, 1), "conv")]
            inpl = planes
        if li < 4:
            s += bn_entries(f"{p}.bn{li}", planes) + [(f"{p}.conv{li}.weight", (planes, planes, 3, 3), "conv")]
    s += bn_entries(f"{p}.bn4_1", 320) + [(f"{p}.conv4_1.weight", (320, 320, 3, 3), "conv")]
    s += bn_entries(f"{p}.bn4_2", 320) + [(f"{p}.conv4_2.weight", (320, 320, 3, 3), "conv")]
    s += bn_entries(f"{p}.bn4_3", 320)
    return s


def ocr_ctc_schema(dict_size: int) -> Schema:
    s = resnet_schema()
    for i in range(N_LAYERS):
        p = f"encoders.layers.{i}"
        s += [(p + ".self_attn.in_proj_weight", (3 * EMBD, EMBD), "linear"), (p + ".self_attn.in_proj_bias", (3 * EMBD,), "bias"),
              (p + ".self_attn.out_proj.weight", (EMBD, EMBD), "linear"), (p + ".self_attn.out_proj.bias", (EMBD,), "bias"),
              (p + ".linear1.weight", (FFN, EMBD), "linear"), (p + ".linear1.bias", (FFN,), "bias"),
              (p + ".linear2.weight", (EMBD, FFN), "linear"), (p + ".linear2.bias", (EMBD,), "bias"),
              (p + ".norm1.weight", (EMBD,), "ln_w"), (p + ".norm1.bias", (EMBD,), "bn_b"),
              (p + ".norm2.weight", (EMBD,), "ln_w"), (p + ".norm2.bias", (EMBD,), "bn_b"),
              (p + ".pe.pe", (1, 2048, EMBD), "sinus_pe")]
    s += [("char_pred_norm.0.weight", (EMBD,), "ln_w"), ("char_pred_norm.0.bias", (EMBD,), "bn_b"),
          ("char_pred.weight", (dict_size, EMBD), "linear*3.0"), ("char_pred.bias", (dict_size,), "bias"),
          ("color_pred1.0.weight", (6, EMBD), "linear*0.02"), ("color_pred1.0.bias", (6,), "bias*8.0")]
    return s
