"""Synthetic, seeded inputs: model weights in the reference's state_dict layouts and manga-like pages.

No checkpoints exist offline, so parity and benchmarks run on random-init
weights of the reference architectures.  Every tensor is drawn from its own generator seeded by
(seed, tensor name), so the state_dict does not depend on iteration order and is identical in
this container and on the GPU box.  Real checkpoints (same key names) can be loaded instead.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

# schema entry: (name, shape, kind)
#   kind: conv (OIHW), convT (IOHW, stride 2), linear (out,in), bias, bn_w, bn_b, bn_rm, bn_rv, nbt,
#         embed, gamma (layer scale), scalar:<value>, buffer:<tag>
Schema = List[Tuple[str, Tuple[int, ...], str]]


def _gen(seed: int, name: str) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    g = torch.Generator()
    g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
    return g


def check_state_dict(sd: Dict[str, torch.Tensor], schema: Schema, what: str) -> Dict[str, torch.Tensor]:
    """Raise ValueError unless ``sd`` holds every tensor of ``schema`` at its shape (extra keys — ``num_batches_tracked``, positional
    tables a reference loader deletes — are ignored).  Run on every checkpoint before an engine is built from it, so a file with another
    layout fails at load time with the list of differences instead of a KeyError or a silent mis-shape deep inside the packers."""
    missing = [n for n, _, k in schema if n not in sd and not k.startswith("nbt")]
    wrong = [f"{n}: {tuple(sd[n].shape)} != {tuple(shape)}" for n, shape, _ in schema if n in sd and tuple(sd[n].shape) != tuple(shape)]
    if missing or wrong:
        raise ValueError(f"{what}: not the expected state_dict layout — {len(missing)} missing tensors (e.g. {missing[:4]}), "
                         f"{len(wrong)} with another shape (e.g. {wrong[:4]})")
    return sd


def synth_state_dict(schema: Schema, seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for name, shape, kind in schema:
        g = _gen(seed, name)
        shape = tuple(shape)
        kind, _, mul = kind.partition("*")  # optional "*<factor>" suffix scales the draw
        if kind == "conv":
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        elif kind == "convT":
            fan_in = shape[0] * int(np.prod(shape[2:])) / 4.0
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        elif kind == "linear":
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(shape[-1]))
        elif kind == "embed":
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(shape[-1]))
        elif kind == "bias":
            t = torch.randn(shape, generator=g) * 0.05
        elif kind == "bn_w":
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif kind == "bn_b":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_rm":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_rv":
            t = torch.rand(shape, generator=g) * 1.0 + 0.5
        elif kind == "ln_w":
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif kind == "gamma":
            t = torch.rand(shape, generator=g) * 0.4 + 0.1
        elif kind == "nbt":
            t = torch.zeros(shape, dtype=torch.int64)
        elif kind.startswith("scalar:"):
            t = torch.full(shape, float(kind.split(":", 1)[1]))
        elif kind == "normal":
            t = torch.randn(shape, generator=g)
        elif kind == "xpos_scale":  # XPOS.scale buffer (ocr/xpos_relative_position.py:50-52)
            hd = shape[0] * 2
            t = (torch.arange(0, hd, 2) + 0.4 * hd) / (1.4 * hd)
        elif kind == "sinus_pe":  # PositionalEncoding.pe buffer (ocr/model_48px_ctc.py:163-174), [1, max_len, d_model]
            _, max_len, d_model = shape
            pe = torch.zeros(max_len, d_model)
            position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
            div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
            pe[:, 0::2] = torch.sin(position * div_term)
            pe[:, 1::2] = torch.cos(position * div_term)
            t = pe.unsqueeze(0)
        elif kind.startswith("tie:"):  # shares storage with an earlier entry (pred.weight = embd.weight, model_48px.py:536)
            t = sd[kind.split(":", 1)[1]]
        else:
            raise ValueError(f"unknown schema kind {kind!r} for {name}")
        if mul:
            t = t * float(mul)
        sd[name] = t
    return sd


def bn_entries(prefix: str, c: int, wmul: str = "") -> Schema:
    return [(f"{prefix}.weight", (c,), "bn_w" + wmul), (f"{prefix}.bias", (c,), "bn_b"),
            (f"{prefix}.running_mean", (c,), "bn_rm"), (f"{prefix}.running_var", (c,), "bn_rv"),
            (f"{prefix}.num_batches_tracked", (), "nbt")]


# ---------------------------------------------------------------------------------------
# Synthetic pages (BASELINE.md §3 / SURVEY.md §8d): white background, dark panel borders,
# 32 text boxes per page of glyph-like blobs; the generator's own quads feed the OCR stage
# and the dilated boxes are the inpainting mask.
# ---------------------------------------------------------------------------------------

def synth_page(index: int, height: int = 2048, width: int = 1456, n_boxes: int = 32, seed: int = 1234, disjoint: bool = False):
    """Returns (page u8 [H,W,3] RGB, quads int64 [n_boxes,4,2] (x,y), mask u8 [H,W] in {0,255}).
    ``disjoint``: text boxes keep 24 px (at 2048 x 1456) apart from each other, like speech bubbles on a real page — what a detector
    needs to tell them apart (the coupled benchmark); the default places them independently (the stages are then fed the generator's
    own quads)."""
    rng = np.random.default_rng(seed + index)
    page = np.clip(rng.normal(245.0, 5.0, size=(height, width, 1)), 0, 255).astype(np.uint8).repeat(3, axis=2)
    # panel borders
    for _ in range(24):
        if rng.random() < 0.5:
            y = int(rng.integers(0, height - 6))
            x0 = int(rng.integers(0, width // 2))
            x1 = int(rng.integers(x0 + 1, width))
            page[y:y + 6, x0:x1] = 20
        else:
            x = int(rng.integers(0, width - 6))
            y0 = int(rng.integers(0, height // 2))
            y1 = int(rng.integers(y0 + 1, height))
            page[y0:y1, x:x + 6] = 20
    quads = np.zeros((n_boxes, 4, 2), dtype=np.int64)
    mask = np.zeros((height, width), dtype=np.uint8)
    sh, sw = height / 2048.0, width / 1456.0
    for b in range(n_boxes):
        vertical = b < n_boxes // 2
        if vertical:
            bw = int(rng.integers(48, 65) * sw)
            bh = int(rng.integers(300, 601) * sh)
        else:
            bw = int(rng.integers(200, 501) * sw)
            bh = int(rng.integers(40, 57) * sh)
        bw, bh = max(bw, 8), max(bh, 8)
        x0 = int(rng.integers(8, max(9, width - bw - 8)))
        y0 = int(rng.integers(8, max(9, height - bh - 8)))
        if disjoint:
            gap = max(int(24 * min(sh, sw)), 2)
            for _ in range(400):   # rejection sampling against the boxes placed so far
                prev = quads[:b]
                if not np.any((prev[:, 0, 0] - gap < x0 + bw) & (prev[:, 2, 0] + gap > x0) & (prev[:, 0, 1] - gap < y0 + bh) & (prev[:, 2, 1] + gap > y0)):
                    break
                x0 = int(rng.integers(8, max(9, width - bw - 8)))
                y0 = int(rng.integers(8, max(9, height - bh - 8)))
        page[y0:y0 + bh, x0:x0 + bw] = 250
        # glyph-like blobs on a 1.2x pitch, 70 % density
        g = int(rng.integers(3, 8))
        pitch = max(int(round(g * 1.2)), g + 1)
        for yy in range(y0 + 2, y0 + bh - g - 1, pitch):
            for xx in range(x0 + 2, x0 + bw - g - 1, pitch):
                if rng.random() < 0.7:
                    page[yy:yy + g, xx:xx + g] = int(rng.integers(0, 40))
        quads[b] = [[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]]
        d = 5
        mask[max(0, y0 - d):min(height, y0 + bh + d), max(0, x0 - d):min(width, x0 + bw + d)] = 255
    return page, quads, mask
