"""State-dict layout of the reference ESRGAN upscaler (``4xESRGAN.pth`` = RRDBNet(3, 3, nf=64, nb=23, upscale=4)).

Key names of ``RRDBNet`` as built by ``ESRGANUpscalerPytorch._load``
(/root/reference/manga_translator/upscaling/esrgan_pytorch.py:28-64,512-526): ``model.0`` feature conv,
``model.1.sub.{i}.RDB{1..3}.conv{1..5}.0`` dense blocks, ``model.1.sub.{nb}`` trunk conv, ``model.3`` / ``model.6``
up-convs (after nearest x2), ``model.8`` / ``model.10`` HR convs.  tests/test_oracle_vs_reference.py pins the names and
shapes against the reference module's own state_dict.
"""
from __future__ import annotations

from .synth import Schema

NF, GC = 64, 32


def rrdbnet_schema(nb: int = 23) -> Schema:
    s: Schema = [("model.0.weight", (NF, 3, 3, 3), "conv"), ("model.0.bias", (NF,), "bias")]
    for i in range(nb):
        for r in (1, 2, 3):
            p = f"model.1.sub.{i}.RDB{r}"
            for k in range(1, 5):
                s += [(f"{p}.conv{k}.0.weight", (GC, NF + (k - 1) * GC, 3, 3), "conv"), (f"{p}.conv{k}.0.bias", (GC,), "bias")]
            s += [(f"{p}.conv5.0.weight", (NF, NF + 4 * GC, 3, 3), "conv"), (f"{p}.conv5.0.bias", (NF,), "bias")]
    s += [(f"model.1.sub.{nb}.weight", (NF, NF, 3, 3), "conv"), (f"model.1.sub.{nb}.bias", (NF,), "bias")]
    for idx in (3, 6, 8):
        s += [(f"model.{idx}.weight", (NF, NF, 3, 3), "conv"), (f"model.{idx}.bias", (NF,), "bias")]
    s += [("model.10.weight", (3, NF, 3, 3), "conv*0.3"), ("model.10.bias", (3,), "bias*8.0")]
    return s
