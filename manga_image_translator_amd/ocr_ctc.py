"""48px CTC OCR (``--ocr 48px_ctc``) on the gfx950 engine: FAN ResNet backbone, 3-layer encoder, greedy CTC decode.

Same model as ``OCR`` of /root/reference/manga_translator/ocr/model_48px_ctc.py:447-494:
  backbone (:277-403) -> 3 x CustomTransformerEncoderLayer (:180-275) -> LayerNorm+GELU+Linear(dict), Linear(6) -> decode_ctc_top1.

MI355X layout: fp32 NHWC; every BatchNorm that sits between two convs is folded into the producing conv's epilogue
(with ReLU); the pre-activation BN+ReLU whose input also feeds the residual is one elementwise pass
(mit_affine_act_nhwc); the layer-closing ``bn -> relu`` rides in the last block's conv2 epilogue with the residual as the
``pre`` operand; the down-sample branch's BN is folded into its 1x1 conv's weights; the positional encoding on q/k is a
precomputed ``PE @ W_qk`` table added in the q|k|v projection's epilogue (``pre`` map with batch stride 0).

The reference feeds the encoder WITHOUT a padding mask on chunks padded to max_w+7+128 (:84,:450-451), so results depend
on the chunk; chunks are therefore formed exactly as the reference does and run one at a time.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .ocr48 import Linear
from .ops import ACT_GELU, ACT_NONE, ACT_RELU

EMBD, HEADS, HEAD_DIM, FFN = 320, 8, 40, 1280
CHANNELS = [80, 160, 320, 320]
LAYERS = [4, 6, 8, 6]


def _bn(sd, p, eps=1e-5):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], eps)


def _affine(sd, p, device):
    sc, bi = ops.fold_bn(*_bn(sd, p))
    return sc.to(device).contiguous(), bi.to(device).contiguous()


class _Block:
    """Pre-activation BasicBlock (:372-403)."""

    def __init__(self, sd, p, inpl, planes, device, closing_bn=None):
        self.pre = _affine(sd, p + ".bn1", device)                       # relu(bn1(x)): elementwise (x also feeds the residual)
        self.conv1 = ops.Conv2d(sd[p + ".conv1.weight"], None, padding=1, bn=_bn(sd, p + ".bn2"), act=ACT_RELU, device=device)
        # conv2 (+ residual); the layer's closing bn + relu (:345-347 etc.) rides here for the last block: act((acc + res) * s + b)
        self.conv2 = ops.Conv2d(sd[p + ".conv2.weight"], None, padding=1, bn=closing_bn, act=ACT_RELU if closing_bn else ACT_NONE,
                                device=device)
        self.closing = closing_bn is not None
        self.down = None
        if (p + ".downsample.1.weight") in sd:  # Sequential(BatchNorm2d, conv1x1) on the residual (:321-326): fold BN into the conv
            g, b, m, v, eps = _bn(sd, p + ".downsample.0")
            s = (g.double() / torch.sqrt(v.double() + eps))
            t = b.double() - m.double() * s
            w = sd[p + ".downsample.1.weight"].double()                   # [planes, inpl, 1, 1]
            self.down = ops.Conv2d((w * s[None, :, None, None]).float(), (w[:, :, 0, 0] @ t).float(), device=device)


class OcrCtcEngine:
    """forward(): u8 line crops of one reference chunk -> (logits [N,T,dict], colours [N,T,6]); decode(): greedy CTC."""

    def __init__(self, sd: Dict[str, torch.Tensor], dict_size: int, device="cuda"):
        self.device = dev = torch.device(device)
        self.dict_size = dict_size
        p = "backbone.ConvNet"
        self.conv0_1 = ops.Conv2d(sd[p + ".conv0_1.weight"], None, padding=1, bn=_bn(sd, p + ".bn0_1"), act=ACT_RELU, device=dev)
        self.conv0_2 = ops.Conv2d(sd[p + ".conv0_2.weight"], None, padding=1, device=dev)
        self.layers: List[List[_Block]] = []
        self.tails = []
        inpl = 40
        for li, (planes, n) in enumerate(zip(CHANNELS, LAYERS), start=1):
            closing = _bn(sd, f"{p}.bn{li}") if li < 4 else _bn(sd, p + ".bn4_1")
            blocks = []
            for b in range(n):
                blocks.append(_Block(sd, f"{p}.layer{li}.{b}", inpl, planes, dev, closing_bn=closing if b == n - 1 else None))
                inpl = planes
            self.layers.append(blocks)
            if li < 4:
                self.tails.append(ops.Conv2d(sd[f"{p}.conv{li}.weight"], None, padding=1, device=dev))
        # conv4_1 (s(2,1), p1) carries bn4_2 + relu, conv4_2 (p0) carries bn4_3 (:362-368)
        self.conv4_1 = ops.Conv2d(sd[p + ".conv4_1.weight"], None, stride=(2, 1), padding=(1, 1), bn=_bn(sd, p + ".bn4_2"),
                                  act=ACT_RELU, device=dev)
        self.conv4_2 = ops.Conv2d(sd[p + ".conv4_2.weight"], None, padding=0, bn=_bn(sd, p + ".bn4_3"), device=dev)
        self.enc = []
        s = HEAD_DIM ** -0.5
        for i in range(3):
            q = f"encoders.layers.{i}"
            w, b = sd[q + ".self_attn.in_proj_weight"].float(), sd[q + ".self_attn.in_proj_bias"].float()
            col_scale = torch.cat([torch.full((EMBD,), s), torch.ones(2 * EMBD)])  # F.multi_head_attention scales q
            qkv = Linear(w, b, dev, col_scale)
            # (x + pe) @ W_qk = x @ W_qk + pe @ W_qk: the second term is a [T, 960] table (zero for the v columns)
            pe = _sinus_pe(2048, EMBD).double()  # recomputed like the reference, which drops the checkpoint's pe.pe (:45-48)
            pew = torch.zeros(pe.shape[0], 3 * EMBD, dtype=torch.float64)
            pew[:, :2 * EMBD] = pe @ w[:2 * EMBD].double().t()
            self.enc.append(dict(
                qkv=qkv, pew=pew.float().to(dev).contiguous(),
                out=Linear(sd[q + ".self_attn.out_proj.weight"], sd[q + ".self_attn.out_proj.bias"], dev),
                ff1=Linear(sd[q + ".linear1.weight"], sd[q + ".linear1.bias"], dev),
                ff2=Linear(sd[q + ".linear2.weight"], sd[q + ".linear2.bias"], dev),
                ln=[(sd[f"{q}.norm{j}.weight"].float().to(dev), sd[f"{q}.norm{j}.bias"].float().to(dev)) for j in (1, 2)]))
        self.pred_ln = (sd["char_pred_norm.0.weight"].float().to(dev), sd["char_pred_norm.0.bias"].float().to(dev))
        self.char_pred = Linear(sd["char_pred.weight"], sd["char_pred.bias"], dev)
        self.color_pred = Linear(sd["color_pred1.0.weight"], sd["color_pred1.0.bias"], dev)
        self._ws: Dict[Tuple, torch.Tensor] = {}

    def _buf(self, name, *shape, dtype=torch.float32):
        n = max(int(np.prod(shape)), 1)
        key = (name, dtype)
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t[:n].view(*shape)

    def release_workspace(self):
        self._ws.clear()

    def _affine_relu(self, x, sc_bi, out):
        B, H, W, Cc = x.shape
        _lib.check(_lib.load().mit_affine_act_nhwc(x.data_ptr(), x.stride(2), sc_bi[0].data_ptr(), sc_bi[1].data_ptr(), out.data_ptr(),
                                                   out.stride(2), B * H * W, Cc, 1, C.c_void_p(ops.current_stream())), "mit_affine_act_nhwc")

    def _pool(self, x, name, kh, kw, sh, sw, ph, pw):
        B, H, W, Cc = x.shape
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        out = self._buf(name, B, Ho, Wo, Cc)
        _lib.check(_lib.load().mit_avgpool_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cc, kh, kw, sh, sw, ph, pw,
                                                C.c_void_p(ops.current_stream())), "mit_avgpool_nhwc")
        return out

    def _backbone(self, x: torch.Tensor) -> torch.Tensor:
        """ResNet.forward (:335-370). x [N,48,Wp,4] -> [N,1,T,320]."""
        B = x.shape[0]
        a = self._buf("c01", B, x.shape[1], x.shape[2], 40)
        self.conv0_1(x, out=a)
        cur = self._buf("c02", B, x.shape[1], x.shape[2], 40)
        self.conv0_2(a, out=cur)
        pools = [(2, 2, 2, 2, 0, 0), (2, 2, 2, 2, 0, 0), (2, 2, 2, 1, 0, 1), None]
        for li, blocks in enumerate(self.layers):
            if pools[li] is not None:
                cur = self._pool(cur, f"pool{li}", *pools[li])
            _, H, W, _ = cur.shape
            for bi, blk in enumerate(blocks):
                cin, planes = cur.shape[3], CHANNELS[li]
                pre = self._buf(f"pre{li}", B, H, W, cin)
                self._affine_relu(cur, blk.pre, pre)
                mid = self._buf(f"mid{li}", B, H, W, planes)
                blk.conv1(pre, out=mid)
                res = cur
                if blk.down is not None:
                    res = self._buf(f"res{li}", B, H, W, planes)
                    blk.down(cur, out=res)
                nxt = self._buf(f"x{li}_{bi & 1}", B, H, W, planes)
                if blk.closing:
                    blk.conv2(mid, out=nxt, pre=res)    # relu(bn(conv2 + residual)): input of the layer's trailing conv
                else:
                    blk.conv2(mid, out=nxt, post=res)   # conv2 + residual
                cur = nxt
            if li < 3:
                t = self._buf(f"tail{li}", B, H, W, CHANNELS[li])
                self.tails[li](cur, out=t)
                cur = t
        a = self._buf("c41", B, *self.conv4_1.out_hw(cur.shape[1], cur.shape[2]), 320)
        self.conv4_1(cur, out=a)
        f = self._buf("c42", B, *self.conv4_2.out_hw(a.shape[1], a.shape[2]), 320)
        self.conv4_2(a, out=f)
        return f

    @torch.no_grad()
    def forward(self, region_u8: torch.Tensor, taps: Optional[dict] = None):
        """One reference chunk: region_u8 [N,48,Wp,3] u8 (device) -> (logits [N,T,dict], colours [N,T,6]) fp32."""
        if region_u8.dtype != torch.uint8 or region_u8.dim() != 4 or region_u8.shape[1] != 48 or region_u8.shape[3] != 3:
            raise ValueError(f"OcrCtcEngine.forward expects u8 [N,48,Wp,3], got {region_u8.dtype} {tuple(region_u8.shape)}")
        region_u8 = region_u8.contiguous()
        N, _, Wp, _ = region_u8.shape
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        x = self._buf("in", N, 48, Wp, 4)
        _lib.check(lib.mit_ocr_prep(region_u8.data_ptr(), x.data_ptr(), N, 48, Wp, st), "mit_ocr_prep")
        feat = self._backbone(x)
        if feat.shape[1] != 1:
            raise RuntimeError(f"backbone height {feat.shape[1]} != 1")
        T = feat.shape[2]
        M = N * T
        mem = feat.reshape(M, EMBD)
        if taps is not None:
            taps["backbone"] = mem.reshape(N, T, EMBD).clone()
        nrm = self._buf("nrm", M, EMBD)
        qkv = self._buf("qkv", M, 3 * EMBD)
        att = self._buf("att", M, EMBD)
        ffh = self._buf("ffh", M, FFN)
        ln = lambda src, wb, dst: _lib.check(lib.mit_layernorm(src.data_ptr(), src.stride(0), wb[0].data_ptr(), wb[1].data_ptr(),
                                                               dst.data_ptr(), dst.stride(0), M, EMBD, 1e-5, st), "mit_layernorm")
        for ly in self.enc:  # CustomTransformerEncoderLayer.forward (:237-257), norm_first
            ln(mem, ly["ln"][0], nrm)
            # q|k|v = norm(x) @ W + b, + (PE @ W_qk)[t] on the q|k columns, q scaled by head_dim**-0.5 (epilogue column scale)
            cm = ops.MitTensorMap()
            cm.base, cm.bs, cm.xs = qkv.data_ptr(), T * 3 * EMBD, 3 * EMBD
            pm = ops.MitTensorMap()
            pm.base, pm.bs, pm.xs = ly["pew"].data_ptr(), 0, 3 * EMBD
            lin = ly["qkv"]
            ops.launch_conv_gemm(ops.conv_gemm_desc(
                a=nrm, NB=N, Hi=1, Wi=T, Cin=EMBD, a_strides=(T * EMBD, 0, EMBD), Ho=1, Wo=T, sy=1, sx=1, taps=[(0, 0, 0)],
                pad_mode=ops.PAD_ZERO, w=lin.w, ldw=lin.Np, Kw=lin.Kp, Nw=lin.Np, N=lin.N, c=cm, pre=pm, scale=lin.scale, bias=lin.bias))
            rs, ts = T * 3 * EMBD, 3 * EMBD
            _lib.check(lib.mit_attention_heads(qkv.data_ptr(), rs, ts, qkv.data_ptr() + 4 * EMBD, rs, ts, qkv.data_ptr() + 8 * EMBD, rs, ts,
                                               att.data_ptr(), T * EMBD, EMBD, None, N, T, T, 1, HEADS, HEAD_DIM, st), "mit_attention_heads")
            ly["out"](att, mem, post=mem)
            ln(mem, ly["ln"][1], nrm)
            ly["ff1"](nrm, ffh, act=ACT_GELU)
            ly["ff2"](ffh, mem, post=mem)
        if taps is not None:
            taps["encoded"] = mem.reshape(N, T, EMBD).clone()
        ln(mem, self.pred_ln, nrm)                                   # char_pred_norm: LayerNorm -> GELU (:435)
        _gelu_inplace(nrm)
        Dp = (self.dict_size + 3) // 4 * 4
        logits = torch.empty(M, Dp, device=self.device)
        self.char_pred(nrm, logits)
        colors = torch.empty(M, 8, device=self.device)
        self.color_pred(mem, colors)
        return logits[:, :self.dict_size].reshape(N, T, self.dict_size), colors[:, :6].reshape(N, T, 6)

    @torch.no_grad()
    def decode(self, logits: torch.Tensor, colors: torch.Tensor, blank: int = 0) -> List[List[tuple]]:
        """decode_ctc_top1 (:473-494): log-softmax + argmax on the GPU, repeat-collapse / blank-drop on the host.

        Returns per line [(char id, log-prob, fr, fg, fb, br, bg, bb)] with colours clamped to [0, 1]."""
        N, T, D = logits.shape
        lib = _lib.load()
        vals = torch.empty(N * T, 5, device=self.device)
        idx = torch.empty(N * T, 5, dtype=torch.int32, device=self.device)
        flat = logits.reshape(N * T, D)
        if flat.stride(1) != 1:
            flat = flat.contiguous()
        _lib.check(lib.mit_logsoftmax_top5(flat.data_ptr(), flat.stride(0), N * T, D, -1, vals.data_ptr(), idx.data_ptr(),
                                           C.c_void_p(ops.current_stream())), "mit_logsoftmax_top5")
        best = idx[:, 0].reshape(N, T).cpu().numpy()
        lp = vals[:, 0].reshape(N, T).cpu().numpy()
        col = colors.clamp(0, 1).cpu().numpy()
        out: List[List[tuple]] = []
        for b in range(N):
            line, last = [], blank
            for t in range(T):
                ch = int(best[b, t])
                if ch != last and ch != blank:
                    line.append((ch, float(lp[b, t]), *[float(c) for c in col[b, t]]))
                last = ch
            out.append(line)
        return out

    @staticmethod
    def make_chunks(region_imgs: List[np.ndarray], max_chunk_size: int = 16):
        """Model48pxCTCOCR._infer's batching (:77-88): sorted by width, groups of 16, padded to max_w + 7 + 128."""
        perm = sorted(range(len(region_imgs)), key=lambda i: region_imgs[i].shape[1])
        for c in range(0, len(perm), max_chunk_size):
            indices = perm[c:c + max_chunk_size]
            widths = [region_imgs[i].shape[1] for i in indices]
            max_width = (4 * (max(widths) + 7) // 4) + 128
            region = np.zeros((len(indices), 48, max_width, 3), dtype=np.uint8)
            for j, i in enumerate(indices):
                region[j, :, :widths[j], :] = region_imgs[i]
            yield indices, widths, region


def _sinus_pe(max_len: int, d_model: int) -> torch.Tensor:
    """PositionalEncoding.pe (model_48px_ctc.py:163-174), fp32 like the reference."""
    import math

    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _gelu_inplace(x2d: torch.Tensor) -> None:
    """x <- gelu(x), erf form (the nn.GELU of char_pred_norm, model_48px_ctc.py:435), one elementwise launch."""
    _lib.check(_lib.load().mit_gelu_inplace(x2d.data_ptr(), x2d.numel(), C.c_void_p(ops.current_stream())), "mit_gelu_inplace")
