"""ESRGAN 4x upscaler (``--upscaler 4xultrasharp``: RRDBNet with the ``4x-UltraSharp`` weights) on the gfx950 engine.

Same network as ``RRDBNet(3, 3, nf=64, nb=23, upscale=4)`` of the reference
(/root/reference/manga_translator/upscaling/esrgan_pytorch.py:28-167) and the tensor part of
``ESRGANUpscalerPytorch._infer`` (:537-546).  MI355X layout:

* fp32 NHWC; every dense block works in ONE ``[B,H,W,192]`` buffer ``[x | x1 | x2 | x3 | x4]``: conv_k reads the first
  64+32(k-1) channels and writes its 32 outputs into the next slice, so the four torch.cat of
  ``ResidualDenseBlock_5C.forward`` (:152-166) cost nothing; conv5's epilogue does ``* 0.2 + x`` and writes the next block's x;
* nearest-x2 + 3x3 conv (upconv_block :317-324) as four 2x2 parity convolutions on the low-resolution tensor
  (ops.UpsampleConv2d): no 4x / 16x sized upsampled intermediates, 2.25x fewer MACs;
* the BGR flip of :541/:545 lives in the first conv's input-channel order and the last conv's output-channel order; the
  3-channel output conv runs on the VALU small-Cout kernel; clip + x255 truncation to uint8 on the GPU.

The final PIL bilinear resize by ratio/4 (:546) is host glue outside the dense path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import lib as _lib
from . import ops
from .ops import ACT_LEAKY, ACT_NONE, PAD_ZERO

NF, GC = 64, 32


class _RDB:
    def __init__(self, sd, p, device):
        self.convs = [ops.Conv2d(sd[f"{p}.conv{k}.0.weight"], sd[f"{p}.conv{k}.0.bias"], padding=1, act=ACT_LEAKY, alpha=0.2,
                                 device=device) for k in range(1, 5)]
        # x5 * 0.2 + x: the 0.2 rides in the epilogue scale (bias pre-multiplied), x comes in as ``post``
        self.conv5 = ops.Conv2d(sd[f"{p}.conv5.0.weight"], sd[f"{p}.conv5.0.bias"], padding=1, device=device,
                                out_scale=torch.full((NF,), 0.2))


class EsrganEngine:
    """Batched 4x upscaler: u8 RGB pages [B,H,W,3] -> u8 RGB [B,4H,4W,3] (device tensors)."""

    def __init__(self, sd: Dict[str, torch.Tensor], nb: int = 23, device="cuda"):
        self.device = dev = torch.device(device)
        self.nb = nb
        w0 = sd["model.0.weight"].detach().float().flip(1)  # network input is BGR (:541): flip the input channels instead
        self.fea = ops.Conv2d(w0, sd["model.0.bias"], padding=1, device=dev)
        self.blocks = [[_RDB(sd, f"model.1.sub.{i}.RDB{r}", dev) for r in (1, 2, 3)] for i in range(nb)]
        self.trunk = ops.Conv2d(sd[f"model.1.sub.{nb}.weight"], sd[f"model.1.sub.{nb}.bias"], padding=1, device=dev)
        self.up1 = ops.UpsampleConv2d(sd["model.3.weight"], sd["model.3.bias"], act=ACT_LEAKY, alpha=0.2, device=dev)
        self.up2 = ops.UpsampleConv2d(sd["model.6.weight"], sd["model.6.bias"], act=ACT_LEAKY, alpha=0.2, device=dev)
        self.hr0 = ops.Conv2d(sd["model.8.weight"], sd["model.8.bias"], padding=1, act=ACT_LEAKY, alpha=0.2, device=dev)
        # output is BGR (:545 flips back): emit RGB directly by flipping the output channels
        self.hr1 = ops.ConvSmallCout(sd["model.10.weight"].detach().float().flip(0), sd["model.10.bias"].detach().float().flip(0),
                                     pad_mode=PAD_ZERO, device=dev)
        self._ws = ops.Workspace(self.device)

    def _buf(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        """Named workspace slab, grown to the largest request (ops.Workspace): memory is bounded by the largest page seen."""
        return self._ws.buf(name, *shape, dtype=dtype)

    def release_workspace(self):
        self._ws.release()

    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, taps=None) -> torch.Tensor:
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
            raise ValueError(f"EsrganEngine.forward expects u8 [B,H,W,3], got {img_u8.dtype} {tuple(img_u8.shape)}")
        img_u8 = img_u8.contiguous()
        B, H, W, _ = img_u8.shape
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        x4 = self._buf("in4", B, H, W, 4)
        zero_mask = self._buf("zmask", B, H, W, dtype=torch.uint8)
        zero_mask.zero_()
        # (rgb / 255 * (1 - 0), 0): the LaMa prep kernel with an all-zero mask is exactly ``float() / 255`` (:541)
        _lib.check(lib.mit_lama_prep(img_u8.data_ptr(), zero_mask.data_ptr(), x4.data_ptr(), B, H, W, st), "mit_lama_prep")
        fea = self._buf("fea", B, H, W, NF)
        self.fea(x4, out=fea)
        cat = [self._buf(f"cat{j}", B, H, W, NF + 4 * GC) for j in range(2)]  # ping-pong dense-block buffers
        t = self._buf("trunk_in", B, H, W, NF)   # RRDB input (kept for the outer residual)
        n = B * H * W * NF
        cur = 0
        cat[0][..., :NF].copy_(fea)
        src = fea
        for i, rrdb in enumerate(self.blocks):
            for r, rdb in enumerate(rrdb):
                buf = cat[cur]
                for k, conv in enumerate(rdb.convs):
                    conv(buf[..., :NF + k * GC], out=buf[..., NF + k * GC:NF + (k + 1) * GC])
                if r < 2:  # next dense block's x goes straight into the other buffer's first slice
                    rdb.conv5(buf, out=cat[cur ^ 1][..., :NF], post=buf[..., :NF])
                    cur ^= 1
                else:     # RRDB output: (x5 * 0.2 + x) * 0.2 + rrdb_in
                    o3 = self._buf("rdb3_out", B, H, W, NF)
                    rdb.conv5(buf, out=o3, post=buf[..., :NF])
                    dst = t if i + 1 == self.nb else self._buf(f"rrdb_out{i & 1}", B, H, W, NF)
                    _lib.check(lib.mit_axpy(dst.data_ptr(), 0.2, o3.data_ptr(), src.data_ptr(), n, st), "mit_axpy")
                    src = dst
                    if i + 1 < self.nb:
                        cat[cur ^ 1][..., :NF].copy_(dst)
                        cur ^= 1
        lr = self._buf("lr", B, H, W, NF)
        self.trunk(src, out=lr, post=fea)  # ShortcutBlock: fea + LR_conv(trunk)
        if taps is not None:
            taps["lr"] = lr.clone()
        u1 = self._buf("u1", B, 2 * H, 2 * W, NF)
        self.up1(lr, out=u1)
        u2 = self._buf("u2", B, 4 * H, 4 * W, NF)
        self.up2(u1, out=u2)
        h0 = self._buf("h0", B, 4 * H, 4 * W, NF)
        self.hr0(u2, out=h0)
        y = self._buf("y", B, 4 * H, 4 * W, 3)
        self.hr1(h0, out=y)
        if taps is not None:
            taps["out_float"] = y.clone()
        out = torch.empty(B, 4 * H, 4 * W, 3, dtype=torch.uint8, device=self.device)
        _lib.check(lib.mit_map_to_u8(y.data_ptr(), out.data_ptr(), y.numel(), 2, 0.0, st), "mit_map_to_u8")
        return out

    def flops_per_input_pixel(self) -> float:
        """Executed MACs x 2 per low-resolution pixel (the up-convs counted at their merged 2x2 form)."""
        rdb = 9 * (sum((NF + k * GC) * GC for k in range(4)) + (NF + 4 * GC) * NF)
        lr = 9 * 3 * NF + self.nb * 3 * rdb + 9 * NF * NF
        hr = 4 * 4 * NF * NF + 16 * 4 * NF * NF + 16 * 9 * NF * NF + 16 * 9 * NF * 3
        return 2.0 * (lr + hr)
