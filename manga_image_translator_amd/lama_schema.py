"""State-dict layout of the reference LaMa generators (key names and shapes).

Mirrors ``FFCResNetGenerator`` as configured by ``LamaFourier``
(/root/reference/manga_translator/inpainting/inpainting_lama_mpe.py:545-601,644-659):
stem 7x7 4->64, three stride-2 downs (the last one splitting 128 local / 384 global), 9 (lama_mpe)
or 18 (lama_large) FFC res-blocks, three ConvTranspose ups, 7x7 64->3 + sigmoid.
tests/test_schema.py pins these names/shapes against the reference module's own state_dict.
"""
from __future__ import annotations

from .synth import Schema, bn_entries

NGF = 64
LOCAL_C, GLOBAL_C = 128, 384  # ratio_gin = ratio_gout = 0.75 of 512


def lama_generator_schema(n_blocks: int) -> Schema:
    s: Schema = []
    s.append(("model.1.ffc.convl2l.weight", (64, 4, 7, 7), "conv"))
    s += bn_entries("model.1.bn_l", 64)
    s.append(("model.2.ffc.convl2l.weight", (128, 64, 3, 3), "conv"))
    s += bn_entries("model.2.bn_l", 128)
    s.append(("model.3.ffc.convl2l.weight", (256, 128, 3, 3), "conv"))
    s += bn_entries("model.3.bn_l", 256)
    s.append(("model.4.ffc.convl2l.weight", (LOCAL_C, 256, 3, 3), "conv"))
    s.append(("model.4.ffc.convl2g.weight", (GLOBAL_C, 256, 3, 3), "conv"))
    s += bn_entries("model.4.bn_l", LOCAL_C)
    s += bn_entries("model.4.bn_g", GLOBAL_C)
    for i in range(n_blocks):
        for cv in ("conv1", "conv2"):
            p = f"model.{5 + i}.{cv}"
            s.append((p + ".ffc.convl2l.weight", (LOCAL_C, LOCAL_C, 3, 3), "conv"))
            s.append((p + ".ffc.convl2g.weight", (GLOBAL_C, LOCAL_C, 3, 3), "conv"))
            s.append((p + ".ffc.convg2l.weight", (LOCAL_C, GLOBAL_C, 3, 3), "conv"))
            s.append((p + ".ffc.convg2g.conv1.0.weight", (GLOBAL_C // 2, GLOBAL_C, 1, 1), "conv"))
            s += bn_entries(p + ".ffc.convg2g.conv1.1", GLOBAL_C // 2)
            s.append((p + ".ffc.convg2g.fu.conv_layer.weight", (GLOBAL_C, GLOBAL_C, 1, 1), "conv"))
            s += bn_entries(p + ".ffc.convg2g.fu.bn", GLOBAL_C)
            s.append((p + ".ffc.convg2g.conv2.weight", (GLOBAL_C, GLOBAL_C // 2, 1, 1), "conv"))
            # random residual branches would grow ~1.8x per block; damp the block's last BN so 18 blocks stay O(1)
            damp = "*0.12" if cv == "conv2" else ""
            s += bn_entries(p + ".bn_l", LOCAL_C, damp)
            s += bn_entries(p + ".bn_g", GLOBAL_C, damp)
    base = 5 + n_blocks + 1  # ConcatTupleLayer sits at 5 + n_blocks
    c = 512
    for i in range(3):
        p = base + 3 * i
        s.append((f"model.{p}.weight", (c, c // 2, 3, 3), "convT"))
        s.append((f"model.{p}.bias", (c // 2,), "bias"))
        s += bn_entries(f"model.{p + 1}", c // 2)
        c //= 2
    p = base + 9 + 1  # ReflectionPad2d at base + 9
    s.append((f"model.{p}.weight", (3, 64, 7, 7), "conv*2.0"))
    s.append((f"model.{p}.bias", (3,), "bias"))
    return s


def lama_mpe_schema() -> Schema:
    """``str_state_dict``: MPE parameters (:616-623). alpha5/alpha6 are 0 at init; non-zero here so the
    positional path is exercised."""
    return [("alpha5", (), "scalar:0.35"), ("alpha6", (), "scalar:-0.25"),
            ("rel_pos_emb.weight", (128, 64), "normal"), ("direct_emb.weight", (4, 64), "normal")]
