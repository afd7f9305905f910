"""``--detector default`` network (DBNet on ResNet-34) on the gfx950 engine.

Same graph as ``TextDetection.forward`` (/root/reference/manga_translator/detection/default_utils/DBNet_resnet34.py:98-125)
+ ``DBHead`` (default_utils/DBHead.py:25-33) + the tensor part of ``det_batch_forward_default`` (detection/default.py:15-25).

Layout: fp32 NHWC; each U-Net concat ``cat([up, skip])`` is a pre-allocated buffer whose two channel slices are written by
the producing layers (the backbone's skip features land there directly and are read back from the slice by the next
backbone layer / average pool); every BatchNorm rides in a conv epilogue; the BasicBlock's ``relu(bn(conv2) + identity)``
uses the epilogue's post-before-activation form; ConvTranspose2d k4 s2 p1 runs as four sub-pixel convolutions.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import lib as _lib
from . import ops
from .dbnet_schema import RESNET34_LAYERS
from .ops import ACT_NONE, ACT_POST_FIRST, ACT_RELU, ACT_SIGMOID


def _bn(sd, p, eps=1e-5):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], eps)


class _Basic:
    def __init__(self, sd, q, stride, dev):
        self.conv1 = ops.Conv2d(sd[q + ".conv1.weight"], None, stride=stride, padding=1, bn=_bn(sd, q + ".bn1"), act=ACT_RELU, device=dev)
        self.conv2 = ops.Conv2d(sd[q + ".conv2.weight"], None, padding=1, bn=_bn(sd, q + ".bn2"), act=ACT_RELU | ACT_POST_FIRST, device=dev)
        self.down = None
        if (q + ".downsample.0.weight") in sd:
            self.down = ops.Conv2d(sd[q + ".downsample.0.weight"], None, stride=stride, bn=_bn(sd, q + ".downsample.1"), device=dev)


class _Triple:
    """conv-bn-relu x 2 then (conv | convT)-bn-relu: double_conv (:22-52) / double_conv_up (:54-75)."""

    def __init__(self, sd, p, up, dev):
        cbr = lambda i: ops.Conv2d(sd[f"{p}.conv.{i}.weight"], None, padding=1, bn=_bn(sd, f"{p}.conv.{i + 1}"), act=ACT_RELU, device=dev)
        self.a, self.b = cbr(0), cbr(3)
        if up:
            self.c = ops.ConvTranspose2d(sd[f"{p}.conv.6.weight"], None, stride=2, padding=1, bn=_bn(sd, f"{p}.conv.7"), act=ACT_RELU, device=dev)
        else:
            self.c = cbr(6)


class DbnetEngine:
    """Batched default-detector network: u8 pages (H, W multiples of 64) -> (db [B,2,H,W] after sigmoid, mask [B,H/2,W/2])."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda"):
        self.device = dev = torch.device(device)
        p = "backbone"
        self.stem = ops.Conv2d(sd[p + ".conv1.weight"], None, stride=2, padding=3, bn=_bn(sd, p + ".bn1"), act=ACT_RELU, device=dev)
        self.layers = []
        for li, (planes, n, stride) in enumerate(RESNET34_LAYERS, start=1):
            self.layers.append([_Basic(sd, f"{p}.layer{li}.{b}", stride if b == 0 else 1, dev) for b in range(n)])
        self.downs = [_Triple(sd, f"down_conv{j}", False, dev) for j in (1, 2, 3)]
        self.ups = [_Triple(sd, f"upconv{j}", True, dev) for j in range(1, 8)]
        d = "conv_db"
        def branch(q, first_bias, last_act):
            c0 = ops.Conv2d(sd[q + ".0.weight"], sd[q + ".0.bias"] if first_bias else None, padding=1, bn=_bn(sd, q + ".1"), act=ACT_RELU, device=dev)
            t1 = ops.ConvTranspose2d(sd[q + ".3.weight"], sd[q + ".3.bias"], stride=2, padding=1, bn=_bn(sd, q + ".4"), act=ACT_RELU, device=dev)
            t2 = ops.ConvTranspose2d(sd[q + ".6.weight"], sd[q + ".6.bias"], stride=2, padding=1, act=last_act, device=dev)
            return c0, t1, t2
        # det_batch_forward_default applies sigmoid to BOTH planes (default.py:23): logits -> sigmoid; the threshold map, which
        # DBHead already passed through a sigmoid, gets a second one (done in place after its own)
        self.binarize = branch(d + ".binarize", True, ACT_SIGMOID)
        self.thresh = branch(d + ".thresh", False, ACT_SIGMOID)
        self.mask_convs = [ops.Conv2d(sd[f"conv_mask.{i}.weight"], sd[f"conv_mask.{i}.bias"], padding=1, act=ACT_RELU, device=dev) for i in (0, 2, 4)]
        self.mask_out = ops.Conv2d(sd["conv_mask.6.weight"], sd["conv_mask.6.bias"], act=ACT_SIGMOID, device=dev)
        self._ws = ops.Workspace(self.device)

    def _buf(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        """Named workspace slab, grown to the largest request (ops.Workspace): memory is bounded by the largest page seen."""
        return self._ws.buf(name, *shape, dtype=dtype)

    def release_workspace(self):
        self._ws.release()

    def _triple(self, tr: _Triple, x, out, tag):
        B, H, W, _ = x.shape
        a = self._buf(tag + ".a", B, H, W, tr.a.Cout)
        tr.a(x, out=a)
        b = self._buf(tag + ".b", B, H, W, tr.b.Cout)
        tr.b(a, out=b)
        return tr.c(b, out=out)

    @torch.no_grad()
    def forward(self, img_u8: torch.Tensor, taps: Optional[dict] = None):
        if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
            raise ValueError(f"DbnetEngine.forward expects u8 [B,H,W,3], got {img_u8.dtype} {tuple(img_u8.shape)}")
        B, H, W, _ = img_u8.shape
        if H % 256 or W % 256:
            raise ValueError("DbnetEngine.forward: H and W must be multiples of 256 (resize_aspect_ratio pads to 256, imgproc.py:54-65)")
        img_u8 = img_u8.contiguous()
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        x = self._buf("in4", B, H, W, 4)
        _lib.check(lib.mit_u8_to_f32_nhwc4(img_u8.data_ptr(), x.data_ptr(), B * H * W, 1, st), "mit_u8_to_f32_nhwc4")
        s2 = self._buf("stem", B, H // 2, W // 2, 64)
        self.stem(x, out=s2)
        h, w = H // 4, W // 4
        pooled = self._buf("pool", B, h, w, 64)
        _lib.check(lib.mit_maxpool2d_nhwc(s2.data_ptr(), pooled.data_ptr(), B, H // 2, W // 2, 64, 3, 2, 1, st), "mit_maxpool2d_nhwc")
        # concat buffers [up | skip] of the decoder; the backbone writes its skips into them
        cat4 = self._buf("cat4", B, h, w, 128)                # [up8 64 | h4 64]
        cat8 = self._buf("cat8", B, h // 2, w // 2, 256)      # [up16 128 | h8 128]
        cat16 = self._buf("cat16", B, h // 4, w // 4, 512)    # [up32 256 | h16 256]
        cat32 = self._buf("cat32", B, h // 8, w // 8, 768)    # [up64 256 | h32 512]
        cat64 = self._buf("cat64", B, h // 16, w // 16, 768)  # [up128 256 | h64 512]
        cat128 = self._buf("cat128", B, h // 32, w // 32, 768)  # [up256 256 | h128 512]
        skips = [cat4[..., 64:], cat8[..., 128:], cat16[..., 256:], cat32[..., 256:]]
        cur = pooled
        for li, blocks in enumerate(self.layers):
            for bi, blk in enumerate(blocks):
                Bc, Hc, Wc, _ = cur.shape
                Ho, Wo = blk.conv1.out_hw(Hc, Wc)
                planes = blk.conv1.Cout
                mid = self._buf(f"l{li}.mid", B, Ho, Wo, planes)
                blk.conv1(cur, out=mid)
                idt = cur
                if blk.down is not None:
                    idt = self._buf(f"l{li}.idt", B, Ho, Wo, planes)
                    blk.down(cur, out=idt)
                last = bi == len(blocks) - 1
                out = skips[li] if last else self._buf(f"l{li}.x{bi & 1}", B, Ho, Wo, planes)
                blk.conv2(mid, out=out, post=idt)  # relu(bn2(conv2) + identity)
                cur = out
        h4, h8, h16, h32 = skips
        # three average-pooled downs (:109-111); h64 / h128 land in their concat slices
        def down(tr, src, out, tag):
            Bc, Hc, Wc, Cc = src.shape
            pl = self._buf(tag + ".pool", B, Hc // 2, Wc // 2, Cc)
            _lib.check(lib.mit_avgpool2_nhwc(src.data_ptr(), src.stride(2), pl.data_ptr(), Cc, B, Hc // 2, Wc // 2, Cc, st), "mit_avgpool2_nhwc")
            return self._triple(tr, pl, out, tag)
        h64 = down(self.downs[0], h32, cat64[..., 256:], "d1")
        h128 = down(self.downs[1], h64, cat128[..., 256:], "d2")
        h256 = self._buf("h256", B, h // 64, w // 64, 512)
        down(self.downs[2], h128, h256, "d3")
        # seven ups (:113-119)
        self._triple(self.ups[0], h256, cat128[..., :256], "u1")
        self._triple(self.ups[1], cat128, cat64[..., :256], "u2")
        self._triple(self.ups[2], cat64, cat32[..., :256], "u3")
        self._triple(self.ups[3], cat32, cat16[..., :256], "u4")
        self._triple(self.ups[4], cat16, cat8[..., :128], "u5")
        self._triple(self.ups[5], cat8, cat4[..., :64], "u6")
        up4 = self._buf("up4", B, 2 * h, 2 * w, 64)
        self._triple(self.ups[6], cat4, up4, "u7")
        up8 = cat4[..., :64]
        # DBHead on up8 (1/4 resolution) -> full-resolution planes (:121, DBHead.py:25-33) + db.sigmoid() (default.py:23)
        db = self._buf("db", B, 2, H, W)
        for plane, (c0, t1, t2) in ((0, self.binarize), (1, self.thresh)):
            b0 = self._buf("db.b0", B, h, w, 16)
            c0(up8, out=b0)
            b1 = self._buf("db.b1", B, 2 * h, 2 * w, 16)
            t1(b0, out=b1)
            t2(b1, out=db[:, plane].unsqueeze(-1))
        for b in range(B):  # the second sigmoid on the (already sigmoided) threshold plane
            _lib.check(lib.mit_sigmoid_inplace(db[b, 1].data_ptr(), H * W, st), "mit_sigmoid_inplace")
        # conv_mask on up4 (1/2 resolution) (:88-94)
        m = up4
        for i, conv in enumerate(self.mask_convs):
            o = self._buf(f"mask{i}", B, 2 * h, 2 * w, conv.Cout)
            conv(m, out=o)
            m = o
        mask = self._buf("mask", B, 2 * h, 2 * w, 1)
        self.mask_out(m, out=mask)
        if taps is not None:
            taps.update(h4=h4.clone(), h32=h32.clone(), up8=up8.clone(), up4=up4.clone())
        return db, mask[..., 0]
