"""State-dict layout of the reference 48px OCR (``ocr_ar_48px.ckpt``): ConvNeXt feature extractor,
4 XPOS encoder layers, 5 XPOS decoder layers, tied embedding/prediction head, colour heads
(/root/reference/manga_translator/ocr/model_48px.py:216-276,496-541).
tests/test_schema.py pins every name/shape against the reference module's own state_dict."""
from __future__ import annotations

from .synth import Schema, bn_entries

EMBD = 320
BASE = EMBD // 8  # 40


def _block(p: str, dim: int, ks: int) -> Schema:
    """ConvNeXtBlock (:184-214); note ``gamma`` registers first."""
    return ([(p + ".gamma", (1, dim, 1, 1), "gamma*0.3"), (p + ".dwconv.weight", (dim, 1, ks, ks), "conv*1.2"),
             (p + ".dwconv.bias", (dim,), "bias")] + bn_entries(p + ".norm", dim)
            + [(p + ".pwconv1.weight", (4 * dim, dim, 1, 1), "conv*1.2"), (p + ".pwconv1.bias", (4 * dim,), "bias"),
               (p + ".pwconv2.weight", (dim, 4 * dim, 1, 1), "conv*1.2"), (p + ".pwconv2.bias", (dim,), "bias")])


def _cbr(p: str, i: int, cin: int, cout: int, kh: int, kw: int) -> Schema:
    return [(f"{p}.{i}.weight", (cout, cin, kh, kw), "conv*1.2"), (f"{p}.{i}.bias", (cout,), "bias")] + bn_entries(f"{p}.{i + 1}", cout)


def _attn(p: str) -> Schema:
    s: Schema = []
    for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
        s += [(f"{p}.{n}.weight", (EMBD, EMBD), "linear"), (f"{p}.{n}.bias", (EMBD,), "bias")]
    s.append((f"{p}.xpos.scale", (EMBD // 4 // 2,), "xpos_scale"))
    return s


def _ffn_norms(p: str, n_norm: int) -> Schema:
    s: Schema = [(p + ".linear1.weight", (2048, EMBD), "linear*1.2"), (p + ".linear1.bias", (2048,), "bias"),
                 (p + ".linear2.weight", (EMBD, 2048), "linear"), (p + ".linear2.bias", (EMBD,), "bias")]
    for i in range(1, n_norm + 1):
        s += [(f"{p}.norm{i}.weight", (EMBD,), "ln_w"), (f"{p}.norm{i}.bias", (EMBD,), "bn_b")]
    return s


def ocr48_schema(dict_size: int) -> Schema:
    s: Schema = []
    s += _cbr("backbone.stem", 0, 3, BASE, 7, 7) + _cbr("backbone.stem", 3, BASE, 2 * BASE, 2, 2) + _cbr("backbone.stem", 6, 2 * BASE, 2 * BASE, 3, 3)
    for i in range(4):
        s += _block(f"backbone.block1.{i}", 2 * BASE, 7)
    s += _cbr("backbone.down1", 0, 2 * BASE, 4 * BASE, 2, 2)
    for i in range(12):
        s += _block(f"backbone.block2.{i}", 4 * BASE, 7)
    s += _cbr("backbone.down2", 0, 4 * BASE, 8 * BASE, 2, 1)
    for i in range(10):
        s += _block(f"backbone.block3.{i}", 8 * BASE, 5)
    s += _cbr("backbone.down3", 0, 8 * BASE, 8 * BASE, 2, 1)
    for i in range(8):
        s += _block(f"backbone.block4.{i}", 8 * BASE, 3)
    s += _cbr("backbone.down4", 0, 8 * BASE, 8 * BASE, 3, 1)
    for i in range(4):
        s += _attn(f"encoders.{i}.self_attn") + _ffn_norms(f"encoders.{i}", 2)
    for i in range(5):
        s += _attn(f"decoders.{i}.self_attn") + _attn(f"decoders.{i}.multihead_attn") + _ffn_norms(f"decoders.{i}", 3)
    s.append(("embd.weight", (dict_size, EMBD), "embed*2.0"))
    s += [("pred1.0.weight", (EMBD, EMBD), "linear*1.2"), ("pred1.0.bias", (EMBD,), "bias")]
    s += [("pred.weight", (dict_size, EMBD), "tie:embd.weight"), ("pred.bias", (dict_size,), "bias")]
    s += [("color_pred1.0.weight", (64, EMBD), "linear"), ("color_pred1.0.bias", (64,), "bias")]
    for n, c in (("color_pred_fg", 3), ("color_pred_bg", 3), ("color_pred_fg_ind", 2), ("color_pred_bg_ind", 2)):
        s += [(f"{n}.weight", (c, 64), "linear"), (f"{n}.bias", (c,), "bias")]
    return s


def synth_dictionary(size: int = 6004):
    """Synthetic alphabet of the real file's size class: '<PAD>'?? no — index 0 blank/pad, 1 <S>, 2 </S>, 3 <SP>
    (model_48px.py:131-137,678)."""
    d = ["<PAD>", "<S>", "</S>", "<SP>"]
    d += [chr(0x4E00 + i) for i in range(size - len(d))]
    return d
