"""48px OCR (``--ocr 48px``) on the gfx950 engine: ConvNeXt backbone, XPOS encoder, native beam decoder.

Same model as ``OCR`` of the reference (/root/reference/manga_translator/ocr/model_48px.py:496-801):
  backbone (:216-276) -> 4 encoder layers (:543-546) -> beam search over 5 decoder layers (:678-801).

MI355X layout
* the backbone/encoder run per chunk exactly as the reference batches them (sorted by width, 16 lines,
  zero-padded uint8 crops of width max(w)+7, :79-91) because the conv stack sees that padding;
* the decoder does NOT: it only sees the encoder memory through a key mask, so the lines of ALL chunks
  (and pages) handed to ``decode`` are decoded together in one native loop (mit_ocr48_decode) —
  hundreds of beam rows per GEMM instead of 80;
* BatchNorm folded into conv epilogues; dwconv + BN in one kernel; layer-scale gamma, GELU, ReLU and
  residual adds in GEMM epilogues; q|k|v projections fused into one GEMM writing straight into the KV cache.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .lib import MitLinear, MitOcr48DecodeArgs, MitOcr48Decoder, MitXposTables
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, launch_conv_gemm, conv_gemm_desc, tensor_map

EMBD, HEADS, HEAD_DIM, FF = 320, 4, 80, 2048
XPOS_IMAX, XPOS_PMAX = 2048, 1024


def _bn(sd, p, eps=1e-5):
    return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"], eps)


class Linear:
    """nn.Linear as a packed [K, N] matrix on the device (+ optional per-column scale)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], device, col_scale: Optional[torch.Tensor] = None):
        N, K = weight.shape
        self.K, self.N = K, N
        self.w, self.Kp, self.Np = ops.pack_weight_kn(weight.detach().to(torch.float32).t(), device)
        b = None if bias is None else bias.detach().to(torch.float32)
        sc = None
        if col_scale is not None:
            sc = col_scale.to(torch.float32)
            b = None if b is None else b * sc
        self.scale = None if sc is None else sc.to(device).contiguous()
        self.bias = None if b is None else b.to(device).contiguous()

    def c_struct(self) -> MitLinear:
        m = MitLinear()
        m.w, m.ldw, m.K, m.N, m.Kp, m.Np = self.w.data_ptr(), self.Np, self.K, self.N, self.Kp, self.Np
        m.scale = None if self.scale is None else self.scale.data_ptr()
        m.bias = None if self.bias is None else self.bias.data_ptr()
        planes, _ = ops._split_for(self.w, self.Np, self.Kp, (0, 0))   # packed in a split GEMM mode: the decoder's launches use them too
        m.w_split = None if planes is None else planes.data_ptr()
        return m

    def __call__(self, x: torch.Tensor, out: torch.Tensor, act: int = ACT_NONE, post: Optional[torch.Tensor] = None,
                 nsplit: int = 0, nhi: int = 0):
        """x [M, K] (row stride free), out [M, N] (or split columns), post [M, N]."""
        M = x.shape[0]
        cm = ops.MitTensorMap()
        cm.base, cm.xs, cm.nsplit, cm.nhi = out.data_ptr(), out.stride(0), nsplit, nhi
        pm = None
        if post is not None:
            pm = ops.MitTensorMap()
            pm.base, pm.xs = post.data_ptr(), post.stride(0)
        launch_conv_gemm(conv_gemm_desc(
            a=x, NB=1, Hi=1, Wi=M, Cin=self.K, a_strides=(0, 0, x.stride(0)), Ho=1, Wo=M, sy=1, sx=1, taps=[(0, 0, 0)],
            pad_mode=ops.PAD_ZERO, w=self.w, ldw=self.Np, Kw=self.Kp, Nw=self.Np, N=self.N, c=cm, post=pm, scale=self.scale,
            bias=self.bias, act=act))
        return out


def xpos_tables(scale_vec: torch.Tensor, device):
    """Tables for MitXposTables, computed with the reference's own fp32 expressions
    (xpos_relative_position.py:9-16 fixed_pos_embedding, :54-57 scale ** (pos / scale_base), :66-67 1/scale)."""
    sv = scale_vec.detach().to(torch.float32).cpu()
    dim = sv.shape[0]
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim) / dim))
    sinus = torch.einsum("i , j -> i j", torch.arange(0, XPOS_IMAX, dtype=torch.float), inv_freq).to(sv)
    pos = torch.arange(-XPOS_PMAX, XPOS_PMAX, 1).to(sv).div(EMBD)[:, None]  # scale_base = embed_dim (model_48px.py:316)
    scale = sv ** pos
    t = dict(cos=torch.cos(sinus), sin=torch.sin(sinus), scale=scale, iscale=1 / scale)
    return {k: v.contiguous().to(device) for k, v in t.items()}


class _EncLayer:
    def __init__(self, sd, p, device):
        a = p + ".self_attn"
        s = HEAD_DIM ** -0.5
        wq, wk, wv = (sd[f"{a}.{n}_proj.weight"] for n in "qkv")
        bq, bk, bv = (sd[f"{a}.{n}_proj.bias"] for n in "qkv")
        col_scale = torch.cat([torch.full((EMBD,), s), torch.ones(2 * EMBD)])
        self.qkv = Linear(torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0), device, col_scale)
        self.out = Linear(sd[a + ".out_proj.weight"], sd[a + ".out_proj.bias"], device)
        self.ff1 = Linear(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], device)
        self.ff2 = Linear(sd[p + ".linear2.weight"], sd[p + ".linear2.bias"], device)
        self.ln = [(sd[f"{p}.norm{i}.weight"].float().to(device), sd[f"{p}.norm{i}.bias"].float().to(device)) for i in (1, 2)]


def fused_mlp_row_permutation(hidden: int) -> torch.Tensor:
    """Row order of pwconv2's weight for ``mit_convnext_mlp``: row k' of the permuted matrix is row ``perm[k']`` of the original.

    The first contraction is computed transposed, so its 32x32 accumulator holds, for a lane's pixel, the hidden indices
    ``32 hb + (r & 3) + 8 (r >> 2) + 4 lh`` in registers r = 0..15 (lh = lane >> 5).  Registers r = 8 s .. 8 s + 7 are handed to MFMA
    step s of the second contraction as the lane's "8 consecutive k"; the B operand therefore has to list the hidden rows in that
    order: k' = 32 hb + 16 s + 8 lh + j  <-  32 hb + (j & 3) + 8 (2 s + (j >> 2)) + 4 lh.  Within every group of 16 rows (one MFMA
    step) the permutation only reorders: each step still contracts the hidden values 16 (2 hb + s) .. + 15."""
    if hidden % 32:
        raise ValueError("hidden width must be a multiple of 32")
    k = torch.arange(hidden)
    hb, rem = k // 32, k % 32
    s_, lh, j = rem // 16, (rem % 16) // 8, rem % 8
    return 32 * hb + (j & 3) + 8 * (2 * s_ + (j >> 2)) + 4 * lh


class _Block:
    """ConvNeXtBlock (:184-214)."""

    def __init__(self, sd, p, dim, ks, device):
        self.ks = ks
        w = sd[p + ".dwconv.weight"].detach().float()  # [dim, 1, ks, ks]
        self.dw_w = w.reshape(dim, ks * ks).t().contiguous().to(device)  # [ks*ks][dim]
        sc, bi = ops.fold_bn(*_bn(sd, p + ".norm", 1e-6), conv_bias=sd[p + ".dwconv.bias"])
        self.dw_scale, self.dw_bias = sc.to(device), bi.to(device)
        self.pw1 = ops.Conv2d(sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"], act=ACT_GELU, device=device)
        self.pw2 = ops.Conv2d(sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"], out_scale=sd[p + ".gamma"], device=device)
        self.dim = dim
        self._fused = None   # (w1 planes, w2 row-permuted planes, ldn2): built on first use in a split GEMM mode

    def _fused_planes(self):
        """Plane tables of the one-launch form (mit_convnext_mlp): pwconv1's planes as the tiles use them, and pwconv2's weight with
        its rows in the order in which the first contraction's accumulator registers hold the hidden index."""
        if self._fused is None:
            w1, w2 = self.pw1, self.pw2
            if w1.Kp != self.dim or w1.Np != 4 * self.dim or w2.Kp != 4 * self.dim:
                raise RuntimeError(f"ConvNeXt block: unexpected packed shapes {w1.Kp}x{w1.Np}, {w2.Kp}x{w2.Np}")
            p1, _ = ops._split_for(w1.w, w1.Np, w1.Kp, (0, 0))
            if p1 is None:   # packed while GEMM mode 0 was on, then switched to 6 (mit_gemm_mode_set): no planes registered — the
                return None  # two-launch form handles that (its tiles fall back to the fp32 MFMA), as it did before the fused kernel
            perm = fused_mlp_row_permutation(4 * self.dim)
            w2p = w2.w.view(w2.Kp, w2.Np)[perm.to(w2.w.device)].contiguous()
            self._fused = (p1, ops.split_weight(w2p), w2.Np)
        return self._fused

    def mlp(self, t: torch.Tensor, h: torch.Tensor, x: torch.Tensor, rows: int):
        """x += gamma * pwconv2(gelu(pwconv1(t))) (:207-213) on flat [rows, C] views: ONE launch with the hidden activations in
        registers where the library has that width and the split-bf16 p6 mode is on (mit_convnext_mlp), else the two GEMM launches
        through the [rows, 4C] buffer ``h``.  Which form runs depends only on the layer width and the GEMM mode — never on the batch."""
        lib = _lib.load()
        planes = self._fused_planes() if fused_mlp_enabled() and ops.split_mode() == 6 and lib.mit_convnext_mlp_supported(self.dim) else None
        if planes is not None:
            p1, p2, ldn2 = planes
            _lib.check(lib.mit_convnext_mlp(t.data_ptr(), self.dim, rows, self.dim, p1.data_ptr(), self.pw1.bias.data_ptr(), p2.data_ptr(), ldn2,
                                            ops._ptr(self.pw2.scale), ops._ptr(self.pw2.bias), x.data_ptr(), self.dim, x.data_ptr(), self.dim,
                                            C.c_void_p(ops.current_stream())), "mit_convnext_mlp")
            return
        t4, h4, x4 = t.view(1, 1, rows, self.dim), h.view(1, 1, rows, 4 * self.dim), x.view(1, 1, rows, self.dim)
        self.pw1(t4, out=h4)
        self.pw2(h4, out=x4, post=x4)


_FUSED_MLP = [os.environ.get("MIT_OCR_FUSED_MLP", "1") not in ("", "0")]


def fused_mlp_enabled() -> bool:
    return _FUSED_MLP[0]


def set_fused_mlp(on: bool) -> bool:
    """Switch the one-launch ConvNeXt pointwise pair (tests, A/B); returns the previous setting.  MIT_OCR_FUSED_MLP=0 in the environment
    starts with it off."""
    prev, _FUSED_MLP[0] = _FUSED_MLP[0], bool(on)
    return prev


class Ocr48Engine:
    """encode(): u8 line crops of one chunk -> encoder memory; decode(): beam search over any number of lines."""

    def __init__(self, sd: Dict[str, torch.Tensor], dict_size: int, device="cuda"):
        self.device = dev = torch.device(device)
        self.dict_size = dict_size
        cbr = lambda p, i, s, pad: ops.Conv2d(sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"], stride=s, padding=pad,
                                              bn=_bn(sd, f"{p}.{i + 1}"), act=ACT_RELU, device=dev)
        b = "backbone."
        self.stem = [cbr(b + "stem", 0, 1, 3), cbr(b + "stem", 3, 2, 0), cbr(b + "stem", 6, 1, 1)]
        self.stages = []
        for name, dim, n, ks in (("block1", 80, 4, 7), ("block2", 160, 12, 7), ("block3", 320, 10, 5), ("block4", 320, 8, 3)):
            self.stages.append([_Block(sd, f"{b}{name}.{i}", dim, ks, dev) for i in range(n)])
        self.downs = [cbr(b + "down1", 0, 2, 0), cbr(b + "down2", 0, (2, 1), 0), cbr(b + "down3", 0, (2, 1), 0),
                      cbr(b + "down4", 0, 1, 0)]
        self.enc = [_EncLayer(sd, f"encoders.{i}", dev) for i in range(4)]
        self.tables = xpos_tables(sd["encoders.0.self_attn.xpos.scale"], dev)
        self.xpos = MitXposTables()
        self.xpos.cos_t, self.xpos.sin_t = self.tables["cos"].data_ptr(), self.tables["sin"].data_ptr()
        self.xpos.scale_t, self.xpos.iscale_t = self.tables["scale"].data_ptr(), self.tables["iscale"].data_ptr()
        self.xpos.imax, self.xpos.pmax = XPOS_IMAX, XPOS_PMAX
        # ---- decoder weights (kept alive in self._keep) ----
        self._keep: List = []
        d = MitOcr48Decoder()
        s = HEAD_DIM ** -0.5
        self.mem_kv = []
        for l in range(5):
            p = f"decoders.{l}"
            a, m = p + ".self_attn", p + ".multihead_attn"
            col_scale = torch.cat([torch.full((EMBD,), s), torch.ones(2 * EMBD)])
            lin = dict(
                qkv=Linear(torch.cat([sd[f"{a}.{n}_proj.weight"] for n in "qkv"], 0),
                           torch.cat([sd[f"{a}.{n}_proj.bias"] for n in "qkv"], 0), dev, col_scale),
                out=Linear(sd[a + ".out_proj.weight"], sd[a + ".out_proj.bias"], dev),
                q2=Linear(sd[m + ".q_proj.weight"], sd[m + ".q_proj.bias"], dev, torch.full((EMBD,), s)),
                out2=Linear(sd[m + ".out_proj.weight"], sd[m + ".out_proj.bias"], dev),
                ff1=Linear(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], dev),
                ff2=Linear(sd[p + ".linear2.weight"], sd[p + ".linear2.bias"], dev))
            self._keep.append(lin)
            ly = d.layers[l]
            for k, v in lin.items():
                setattr(ly, k, v.c_struct())
            for i in (1, 2, 3):
                wt, bt = sd[f"{p}.norm{i}.weight"].float().to(dev), sd[f"{p}.norm{i}.bias"].float().to(dev)
                self._keep += [wt, bt]
                setattr(ly, f"ln{i}_w", wt.data_ptr())
                setattr(ly, f"ln{i}_b", bt.data_ptr())
            # cross-attention K|V projection of the encoder memory (done once per line at encode time)
            self.mem_kv.append(Linear(torch.cat([sd[m + ".k_proj.weight"], sd[m + ".v_proj.weight"]], 0),
                                      torch.cat([sd[m + ".k_proj.bias"], sd[m + ".v_proj.bias"]], 0), dev))
        self.embd = sd["embd.weight"].detach().float().to(dev).contiguous()
        self.pred1 = Linear(sd["pred1.0.weight"], sd["pred1.0.bias"], dev)
        self.pred = Linear(sd["pred.weight"], sd["pred.bias"], dev)
        self.color1 = Linear(sd["color_pred1.0.weight"], sd["color_pred1.0.bias"], dev)
        heads = ("color_pred_fg", "color_pred_bg", "color_pred_fg_ind", "color_pred_bg_ind")
        self.color_heads = Linear(torch.cat([sd[h + ".weight"] for h in heads], 0), torch.cat([sd[h + ".bias"] for h in heads], 0), dev)
        d.embd = self.embd.data_ptr()
        d.pred1, d.pred = self.pred1.c_struct(), self.pred.c_struct()
        d.color1, d.color_heads = self.color1.c_struct(), self.color_heads.c_struct()
        d.xpos = self.xpos
        d.dict_size = dict_size
        self.dec = d
        self._ws: Dict[Tuple, torch.Tensor] = {}
        self.lines_attention_max_len = int(_lib.load().mit_attention_lines_xpos_max_len(HEAD_DIM))   # longest line of the one-launch form
        self.per_chunk_attention = False   # True: the encoder's attention chunk by chunk (rotate q, rotate k, attention): the reference form for tests

    def _buf(self, name, *shape, dtype=torch.float32):
        """Named workspace slab, grown to the largest request (chunk widths vary from call to call; all users are
        ordered on one stream, so a slab can be re-viewed at a new shape by the next chunk)."""
        n = max(int(math.prod(shape)), 1)
        key = (name, dtype)
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t[:n].view(*shape)

    def release_workspace(self):
        self._ws.clear()

    # -- ConvNext_FeatureExtractor.forward (:262-276) -----------------------------------------
    def _backbone(self, x: torch.Tensor, tag: str) -> torch.Tensor:
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        for ci, conv in enumerate(self.stem):
            x = conv(x, out=self._buf(f"{tag}.s{ci}", x.shape[0], *conv.out_hw(x.shape[1], x.shape[2]), conv.Cout))
        for si, (blocks, down) in enumerate(zip(self.stages, self.downs)):
            B, H, W, Cc = x.shape
            t = self._buf(f"{tag}.dw{si}", B, H, W, Cc)
            h4 = self._buf(f"{tag}.h{si}", B, H, W, 4 * Cc)
            for blk in blocks:
                _lib.check(lib.mit_dwconv_nhwc(x.data_ptr(), blk.dw_w.data_ptr(), blk.dw_scale.data_ptr(), blk.dw_bias.data_ptr(),
                                               t.data_ptr(), B, H, W, Cc, blk.ks, st), "mit_dwconv_nhwc")
                blk.mlp(t, h4, x, B * H * W)  # input + gamma * pwconv2(gelu(pwconv1(t))), in place (:207-213)
            x = down(x, out=self._buf(f"{tag}.d{si}", B, *down.out_hw(H, W), down.Cout))
        return x

    # -- page-group path: all chunks of a group share every row-wise op ------------------------------------------
    def _cat_views(self, name, shapes, C_):
        """One slab holding [N,H,W,C_] images back to back; returns (flat [rows, C_], per-chunk 4-D views, row starts)."""
        rows = [n * h * w for n, h, w in shapes]
        starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
        flat = self._buf(name, int(starts[-1]), C_)
        views = [flat[starts[i]:starts[i + 1]].view(n, h, w, C_) for i, (n, h, w) in enumerate(shapes)]
        return flat, views, starts

    def _ragged_table(self, shapes, starts, stage_tabs):
        from .lib import MitRaggedSeg

        segs = (MitRaggedSeg * len(shapes))()
        g = 0
        for i, (n, h, w) in enumerate(shapes):
            segs[i].pixel_start, segs[i].group_start, segs[i].B, segs[i].H, segs[i].W = int(starts[i]), g, n, h, w
            g += n * h * ((w + 3) // 4)
        stage_tabs.append((bytes(segs), g))

    @torch.no_grad()
    def encode_group(self, regions: Sequence[torch.Tensor], klen_all: torch.Tensor, Lmax: int, mem_k: Optional[torch.Tensor] = None,
                     mem_v: Optional[torch.Tensor] = None, first_lines: Optional[Sequence[int]] = None):
        """Backbone + encoder + cross-attention K/V for several chunks at once.

        regions[c] u8 [N_c,48,Wp_c,3] (device); klen_all int32 (device), indexed by pooled line number: chunk c owns
        lines [first_lines[c], first_lines[c] + N_c) (default: back to back from 0) of the pooled outputs
        mem_k / mem_v [5, n_lines, Lmax, 320] (allocated here when not given).  Every row-wise operator (the
        pointwise convs = 96 % of the backbone FLOPs, LayerNorm, all Linear layers) runs ONCE over the rows of all
        chunks; only the spatial ops (stem / downsampling convs, XPOS rotation, attention) are launched per chunk and
        the depthwise conv through its ragged form.  Row results do not depend on which other rows share a launch, so
        this is bitwise identical to encode() chunk by chunk.  Returns (mem_k, mem_v, keep) when it allocated the pooled
        tensors, else the staging buffers to keep alive."""
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        nc = len(regions)
        Ns = [int(r.shape[0]) for r in regions]
        Wps = [int(r.shape[2]) for r in regions]
        # ---- stem, per chunk (7x7 s1, 2x2 s2, 3x3 s1) ----
        sh0 = [(n, 48, wp) for n, wp in zip(Ns, Wps)]
        _, x_in, _ = self._cat_views("g.in", sh0, 4)
        for r, xi in zip(regions, x_in):
            _lib.check(lib.mit_ocr_prep(r.contiguous().data_ptr(), xi.data_ptr(), xi.shape[0], 48, xi.shape[2], st), "mit_ocr_prep")
        cur, cur_sh = x_in, sh0
        for ci, conv in enumerate(self.stem):
            sh = [(n, *conv.out_hw(h, w)) for n, h, w in cur_sh]
            flat, views, starts = self._cat_views(f"g.s{ci}", sh, conv.Cout)
            for xi, oi in zip(cur, views):
                conv(xi, out=oi)
            cur, cur_sh = views, sh
        # ---- ragged tables for the four stages (uploaded in one pinned copy) ----
        stage_shapes, tabs = [], []
        sh = cur_sh
        for down in self.downs:
            stage_shapes.append(sh)
            rows = [n * h * w for n, h, w in sh]
            self._ragged_table(sh, np.concatenate([[0], np.cumsum(rows)]), tabs)
            sh = [(n, *down.out_hw(h, w)) for n, h, w in sh]
        blob = b"".join(t for t, _ in tabs)
        stage = torch.empty(len(blob), dtype=torch.uint8, pin_memory=True)
        stage.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        tab_dev = stage.to(self.device, non_blocking=True)
        tab_off = np.concatenate([[0], np.cumsum([len(t) for t, _ in tabs])])
        # ---- ConvNeXt stages ----
        for si, (blocks, down) in enumerate(zip(self.stages, self.downs)):
            sh = stage_shapes[si]
            Cc = cur[0].shape[3]
            rows = sum(n * h * w for n, h, w in sh)
            x_flat = flat  # slab holding `cur`
            t_flat = self._buf(f"g.dw{si}", rows, Cc)
            h_flat = self._buf(f"g.h{si}", rows, 4 * Cc)
            heights = {h for _, h, _ in sh}
            common_h = heights.pop() if len(heights) == 1 else 0  # every chunk of a 48 px recogniser has the same height per stage
            for blk in blocks:
                _lib.check(lib.mit_dwconv_nhwc_ragged_rows(x_flat.data_ptr(), blk.dw_w.data_ptr(), blk.dw_scale.data_ptr(),
                                                           blk.dw_bias.data_ptr(), t_flat.data_ptr(),
                                                           tab_dev.data_ptr() + int(tab_off[si]), nc, tabs[si][1], Cc, blk.ks,
                                                           common_h, st), "mit_dwconv_nhwc_ragged_rows")
                blk.mlp(t_flat, h_flat, x_flat, rows)
            nsh = [(n, *down.out_hw(h, w)) for n, h, w in sh]
            flat, views, starts = self._cat_views(f"g.d{si}", nsh, down.Cout)
            for xi, oi in zip(cur, views):
                down(xi, out=oi)
            cur, cur_sh = views, nsh
        # ---- transformer encoder over the concatenated memory [sum N_c * L_c, 320] ----
        Ls = [w for _, _, w in cur_sh]
        for wp, L in zip(Wps, Ls):
            if L != self.memory_len(wp):
                raise RuntimeError(f"backbone length {L} != memory_len({wp})")
        mem = flat
        M = mem.shape[0]
        nrm = self._buf("g.nrm", M, EMBD)
        qkv = self._buf("g.qkv", 3, M, EMBD)
        qr = self._buf("g.qr", M, EMBD)
        kr = self._buf("g.kr", M, EMBD)
        att = self._buf("g.att", M, EMBD)
        ffh = self._buf("g.ffh", M, FF)
        row0 = starts  # first memory row of each chunk
        own = mem_k is None
        if first_lines is None:
            first_lines = np.concatenate([[0], np.cumsum(Ns)])[:-1].tolist()
        line0 = [int(f) for f in first_lines]
        # one launch per layer for all lines of the group (XPOS rotation folded into the attention kernel) when the group's lines are
        # consecutive in the pooled memory; else chunk by chunk (rotate q, rotate k, attention) — bitwise the same either way
        # (lines longer than the kernel's LDS form holds — mit_attention_lines_xpos_max_len, 308 positions = crops wider than ~1230 px —
        # take the chunk-by-chunk path, which has no length limit)
        ragged = (all(line0[c + 1] == line0[c] + Ns[c] for c in range(nc - 1)) and max(Ls) <= self.lines_attention_max_len
                  and not self.per_chunk_attention)
        if ragged:
            tab = np.empty((sum(Ns), 2), dtype=np.int32)
            i = 0
            for c in range(nc):
                tab[i:i + Ns[c], 0] = int(row0[c]) + np.arange(Ns[c], dtype=np.int64) * Ls[c]
                tab[i:i + Ns[c], 1] = Ls[c]
                i += Ns[c]
            tab_pin = torch.from_numpy(tab).pin_memory()
            lines_dev = tab_pin.to(self.device, non_blocking=True)
            klen_g = klen_all[line0[0]:line0[0] + sum(Ns)]
            Lg = int(max(Ls))
        for ly in self.enc:
            self._layernorm(mem, *ly.ln[0], nrm)
            ly.qkv(nrm, qkv[0], nsplit=EMBD, nhi=M * EMBD)
            if ragged:
                _lib.check(lib.mit_attention_lines_xpos(qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), att.data_ptr(), EMBD,
                                                        lines_dev.data_ptr(), klen_g.data_ptr(), sum(Ns), Lg, HEADS, HEAD_DIM,
                                                        C.byref(self.xpos), st), "mit_attention_lines_xpos")
            else:
                for c in range(nc):
                    N, L = Ns[c], Ls[c]
                    a, b = int(row0[c]), int(row0[c + 1])
                    minpos, LE = -((L + 1) // 2), L * EMBD
                    self._rotate(qkv[0][a:b], qr[a:b], N, L, 0, minpos, False, LE, EMBD, LE, EMBD)
                    self._rotate(qkv[1][a:b], kr[a:b], N, L, 0, minpos, True, LE, EMBD, LE, EMBD)
                    self._attention(qr[a:b], kr[a:b], qkv[2][a:b], att[a:b], klen_all[line0[c]:line0[c] + N], N, L, L, 1,
                                    ((LE, EMBD),) * 4)
            ly.out(att, mem, post=mem)
            self._layernorm(mem, *ly.ln[1], nrm)
            ly.ff1(nrm, ffh, act=ACT_RELU)
            ly.ff2(ffh, mem, post=mem)
        # ---- cross-attention keys / values of the five decoder layers, pooled and padded to Lmax ----
        if own:
            mem_k, mem_v = self.alloc_memory(sum(Ns), Lmax)
        ktmp = self._buf("g.ktmp", M, EMBD)
        vtmp = self._buf("g.vtmp", M, EMBD)
        for l in range(5):
            self.mem_kv[l](mem, ktmp, nsplit=EMBD, nhi=(vtmp.data_ptr() - ktmp.data_ptr()) // 4)
            if ragged:
                _lib.check(lib.mit_memory_kv_lines(ktmp.data_ptr(), vtmp.data_ptr(), EMBD, mem_k[l].data_ptr(), mem_v[l].data_ptr(),
                                                   mem_k.shape[2] * EMBD, lines_dev.data_ptr(), sum(Ns), line0[0], M, Lg, HEAD_DIM,
                                                   C.byref(self.xpos), st), "mit_memory_kv_lines")
                continue
            for c in range(nc):
                N, L = Ns[c], Ls[c]
                a, b, l0 = int(row0[c]), int(row0[c + 1]), int(line0[c])
                self._rotate(ktmp[a:b], mem_k[l, l0:l0 + N], N, L, 0, -((L + 1) // 2), True, L * EMBD, EMBD, Lmax * EMBD, EMBD)
                mem_v[l, l0:l0 + N, :L].copy_(vtmp[a:b].view(N, L, EMBD))
        if ragged:
            stage = (stage, tab_pin, lines_dev)
        return (mem_k, mem_v, (stage, tab_dev)) if own else (stage, tab_dev)

    def _rotate(self, src, dst, R, T, i0, p0, downscale, src_rs, src_ts, dst_rs, dst_ts):
        _lib.check(_lib.load().mit_xpos_rotate(src.data_ptr(), src_rs, src_ts, dst.data_ptr(), dst_rs, dst_ts, R, T, i0, p0,
                                               int(downscale), C.byref(self.xpos), C.c_void_p(ops.current_stream())), "mit_xpos_rotate")

    def _layernorm(self, x2d, w, b, out2d):
        _lib.check(_lib.load().mit_layernorm(x2d.data_ptr(), x2d.stride(0), w.data_ptr(), b.data_ptr(), out2d.data_ptr(),
                                             out2d.stride(0), x2d.shape[0], EMBD, 1e-5, C.c_void_p(ops.current_stream())), "mit_layernorm")

    def _attention(self, q, k, v, out, klen, R, Tq, Tk, kv_div, strides):
        (q_rs, q_ts), (k_rs, k_ts), (v_rs, v_ts), (o_rs, o_ts) = strides
        _lib.check(_lib.load().mit_attention(q.data_ptr(), q_rs, q_ts, k.data_ptr(), k_rs, k_ts, v.data_ptr(), v_rs, v_ts,
                                             out.data_ptr(), o_rs, o_ts, None if klen is None else klen.data_ptr(), R, Tq, Tk,
                                             kv_div, C.c_void_p(ops.current_stream())), "mit_attention")

    @staticmethod
    def memory_len(Wp: int) -> int:
        """Backbone output length for a padded width Wp: the two k2 s2 convs floor it twice (:222-224,:237-239)."""
        return (Wp // 2) // 2

    @staticmethod
    def valid_len(width: int, L: int) -> int:
        """Unmasked memory positions of a line (:684-688)."""
        return min((width + 3) // 4 + 2, L)

    @torch.no_grad()
    def encode(self, region_u8: torch.Tensor, widths: Sequence[int], taps: Optional[dict] = None,
               klen: Optional[torch.Tensor] = None):
        """One reference chunk (:83-120 + :682-689): region_u8 [N,48,Wp,3] u8 (device), widths of the unpadded crops.

        Returns (mem_k [5,N,L,320], mem_v [5,N,L,320], mem_len [N] int32, L): the per-decoder-layer cross-attention
        keys (projected + XPOS-rotated for this chunk's length L) and values of the encoder memory."""
        if region_u8.dtype != torch.uint8 or region_u8.dim() != 4 or region_u8.shape[1] != 48 or region_u8.shape[3] != 3:
            raise ValueError(f"Ocr48Engine.encode expects u8 [N,48,Wp,3], got {region_u8.dtype} {tuple(region_u8.shape)}")
        region_u8 = region_u8.contiguous()
        N, _, Wp, _ = region_u8.shape
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        tag = "enc"
        x = self._buf(tag + ".in", N, 48, Wp, 4)
        _lib.check(lib.mit_ocr_prep(region_u8.data_ptr(), x.data_ptr(), N, 48, Wp, st), "mit_ocr_prep")
        feat = self._backbone(x, tag)  # [N,1,L,320]
        L = feat.shape[2]
        mem = feat.reshape(N * L, EMBD)  # 'N C 1 W -> N W C' is free in NHWC
        if taps is not None:
            taps["backbone"] = mem.reshape(N, L, EMBD).clone()
        if L != self.memory_len(Wp):
            raise RuntimeError(f"backbone length {L} != memory_len({Wp})")
        if klen is None:  # (a synchronous upload; batch callers pass a device tensor prepared up front)
            klen = torch.tensor([self.valid_len(w, L) for w in widths], dtype=torch.int32).to(self.device)
        M = N * L
        nrm = self._buf(tag + ".nrm", M, EMBD)
        qkv = self._buf(tag + ".qkv", 3, M, EMBD)
        qr = self._buf(tag + ".qr", M, EMBD)
        kr = self._buf(tag + ".kr", M, EMBD)
        att = self._buf(tag + ".att", M, EMBD)
        ffh = self._buf(tag + ".ffh", M, FF)
        minpos = -((L + 1) // 2)  # python: -(L) // 2
        LE = L * EMBD
        for ly in self.enc:  # transformer_encoder_forward (:278-292), norm_first
            self._layernorm(mem, *ly.ln[0], nrm)
            ly.qkv(nrm, qkv[0], nsplit=EMBD, nhi=M * EMBD)
            self._rotate(qkv[0], qr, N, L, 0, minpos, False, LE, EMBD, LE, EMBD)
            self._rotate(qkv[1], kr, N, L, 0, minpos, True, LE, EMBD, LE, EMBD)
            self._attention(qr, kr, qkv[2], att, klen, N, L, L, 1, ((LE, EMBD),) * 4)
            ly.out(att, mem, post=mem)
            self._layernorm(mem, *ly.ln[1], nrm)
            ly.ff1(nrm, ffh, act=ACT_RELU)
            ly.ff2(ffh, mem, post=mem)
        if taps is not None:
            taps["memory"] = mem.reshape(N, L, EMBD).clone()
        mem_k = torch.empty(5, N, L, EMBD, device=self.device)
        mem_v = torch.empty(5, N, L, EMBD, device=self.device)
        ktmp = self._buf(tag + ".ktmp", M, EMBD)
        for l in range(5):
            # one GEMM: K columns land in ktmp, V columns straight in mem_v[l] (column split with an arbitrary plane offset)
            self.mem_kv[l](mem, ktmp, nsplit=EMBD, nhi=(mem_v[l].data_ptr() - ktmp.data_ptr()) // 4)
            self._rotate(ktmp, mem_k[l], N, L, 0, minpos, True, LE, EMBD, LE, EMBD)
        return mem_k, mem_v, klen, L

    @torch.no_grad()
    def decode(self, mem_k: torch.Tensor, mem_v: torch.Tensor, mem_len: torch.Tensor, max_seq_length: int = 255,
               suppress_eos: bool = False, trace: bool = False, graph: Optional[bool] = None):
        """Beam search (:691-801) over N lines at once. mem_k/mem_v [5,N,L,320], mem_len [N] int32.

        Returns a dict of device tensors: tokens [N,T+1] int32, length [N], prob [N], colors [N,T,10]
        (+ trace_logits [T,N*5,dict], trace_hist when ``trace``)."""
        _, N, L, _ = mem_k.shape
        T = max_seq_length
        lib = _lib.load()
        nbytes = lib.mit_ocr48_decode_workspace_bytes(N, T, self.dict_size)
        ws = self._buf("decode.ws", nbytes, dtype=torch.uint8)
        dev = self.device
        res_tok = torch.zeros(N, T + 1, dtype=torch.int32, device=dev)
        res_len = torch.zeros(N, dtype=torch.int32, device=dev)
        res_prob = torch.zeros(N, dtype=torch.float32, device=dev)
        res_row = torch.zeros(N, dtype=torch.int32, device=dev)
        colors = torch.empty(N * 5, T, 12, dtype=torch.float32, device=dev)
        a = MitOcr48DecodeArgs()
        a.N, a.L = N, L
        a.mem_k, a.mem_v, a.mem_len = mem_k.contiguous().data_ptr(), mem_v.contiguous().data_ptr(), mem_len.data_ptr()
        a.max_seq_length, a.start_tok, a.end_tok, a.max_finished, a.suppress_eos = T, 1, 2, 2, int(suppress_eos)
        a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
        a.graph_mode = 0 if graph is None else (1 if graph else 2)   # None: off unless MIT_OCR_DECODE_GRAPH=1 (hipGraph replay of the steps: no gain measured)
        a.res_tok, a.res_len, a.res_prob, a.res_row, a.colors = (t.data_ptr() for t in (res_tok, res_len, res_prob, res_row, colors))
        out = {}
        if trace:
            out["trace_logits"] = torch.zeros(T, N * 5, self.dict_size, device=dev)
            out["trace_hist"] = torch.zeros(T, N * 5, T + 1, dtype=torch.int32, device=dev)
            a.trace_logits, a.trace_hist = out["trace_logits"].data_ptr(), out["trace_hist"].data_ptr()
        _lib.check(lib.mit_ocr48_decode(C.byref(self.dec), C.byref(a), C.c_void_p(ops.current_stream())), "mit_ocr48_decode")
        sel = colors.reshape(N, 5, T, 12)[torch.arange(N, device=dev), (res_row.long() % 5)]
        out.update(tokens=res_tok, length=res_len, prob=res_prob, colors=sel[..., :10], steps_run=a.steps_run)
        return out

    # -- host batching of Model48pxOCR._infer (:79-91) -----------------------------------------
    @staticmethod
    def make_chunks(region_imgs: List[np.ndarray], max_chunk_size: int = 16):
        perm = sorted(range(len(region_imgs)), key=lambda i: region_imgs[i].shape[1])
        for c in range(0, len(perm), max_chunk_size):
            indices = perm[c:c + max_chunk_size]
            widths = [region_imgs[i].shape[1] for i in indices]
            max_width = 4 * (max(widths) + 7) // 4  # == max + 7, the reference's precedence quirk (:86)
            region = np.zeros((len(indices), 48, max_width, 3), dtype=np.uint8)
            for j, i in enumerate(indices):
                region[j, :, :widths[j], :] = region_imgs[i]
            yield indices, widths, region

    @torch.no_grad()
    def recognize(self, region_imgs: List[np.ndarray], max_seq_length: int = 255, suppress_eos: bool = False):
        """All lines of one or more pages: per-chunk encode, one pooled decode. Results in the reference's chunk order."""
        order, mks, mvs, lens = [], [], [], []
        for indices, widths, region in self.make_chunks(region_imgs):
            mk, mv, kl, L = self.encode(torch.from_numpy(region).to(self.device), widths)
            order += indices
            mks.append(mk)
            mvs.append(mv)
            lens.append(kl)
        Lmax = max(m.shape[2] for m in mks)
        pad = lambda m: m if m.shape[2] == Lmax else torch.cat([m, m.new_zeros(5, m.shape[1], Lmax - m.shape[2], EMBD)], 2)
        mem_k, mem_v = torch.cat([pad(m) for m in mks], 1), torch.cat([pad(m) for m in mvs], 1)
        out = self.decode(mem_k, mem_v, torch.cat(lens), max_seq_length, suppress_eos)
        out["order"] = order
        return out

    # -- Model48pxOCR._infer (:67-120) for a batch of device-resident pages ----------------------------------------
    def plan_pages(self, quads_per_page, H: int, W: int, directions=None):
        """Host planning for ``recognize_pages``: per page, rectification geometry of every line (textline.warp_plans) and
        the reference's chunking (sorted by crop width, groups of 16, padded to max+7, model_48px.py:79-86).

        Returns a dict: ``records`` (WARP_LINE_DTYPE, one per line, ordered chunk by chunk), ``chunks``
        [(first_record, n_lines, widths, padded_width, page)], ``order`` [(page, line)] per record, ``klens`` int32, ``Lmax``."""
        from . import textline as TL

        recs, chunks, order, klens = [], [], [], []
        n = 0
        for p, quads in enumerate(quads_per_page):
            dirs = [q.direction for q in quads] if directions is None else list(directions[p])
            rec = TL.warp_plans(quads, dirs, H, W, 48)
            rec["page"] = p
            widths = np.where(rec["vertical"] != 0, rec["dh"], rec["dw"])
            for idx, ws, wp in TL.chunk_plan(widths.tolist()):
                r = rec[idx].copy()
                r["out_row"] = np.arange(len(idx))
                recs.append(r)
                order += [(p, i) for i in idx]
                L = self.memory_len(wp)
                klens += [self.valid_len(w, L) for w in ws]
                chunks.append((n, len(idx), ws, wp, p))
                n += len(idx)
        records = np.concatenate(recs) if recs else np.zeros(0, dtype=TL.WARP_LINE_DTYPE)
        Lmax = max((self.memory_len(c[3]) for c in chunks), default=0)
        return dict(records=records, chunks=chunks, order=order, klens=np.asarray(klens, dtype=np.int32), Lmax=Lmax, H=H, W=W)

    def upload_plan(self, plan):
        """One pinned staging buffer -> one asynchronous upload of the line records and key lengths (a pageable copy would
        stall behind queued GPU work).  Adds ``lines_dev`` / ``klen_dev`` / ``_keep`` to the plan."""
        raw = plan["records"].tobytes()
        kb = plan["klens"].tobytes()
        stage = torch.empty(len(raw) + len(kb), dtype=torch.uint8, pin_memory=True)
        stage.copy_(torch.frombuffer(bytearray(raw + kb), dtype=torch.uint8))
        table = stage.to(self.device, non_blocking=True)
        plan["lines_dev"], plan["klen_dev"] = table[:len(raw)], table[len(raw):].view(torch.int32)
        plan["_keep"] = [stage, table]
        return plan

    @torch.no_grad()
    def encode_planned(self, pages_u8: torch.Tensor, plan, chunk_ids: Sequence[int], mem_k: torch.Tensor, mem_v: torch.Tensor):
        """Rectify (mit_ocr_warp_lines) and encode the given chunks of an uploaded plan; their cross-attention K/V land in
        rows [first_record, first_record + n) of the pooled ``mem_k`` / ``mem_v`` [5, n_lines, Lmax, 320]."""
        if not chunk_ids:
            return
        lib = _lib.load()
        st = C.c_void_p(ops.current_stream())
        H, W = plan["H"], plan["W"]
        rec_bytes = plan["records"].dtype.itemsize
        regions = []
        for ci in chunk_ids:
            first, n, ws, wp, _ = plan["chunks"][ci]
            region = torch.empty(n, 48, wp, 3, dtype=torch.uint8, device=self.device)
            _lib.check(lib.mit_ocr_warp_lines(pages_u8.data_ptr(), H, W, plan["lines_dev"].data_ptr() + first * rec_bytes, n,
                                              region.data_ptr(), 48, wp, st), "mit_ocr_warp_lines")
            regions.append(region)
        firsts = [plan["chunks"][ci][0] for ci in chunk_ids]
        keep = self.encode_group(regions, plan["klen_dev"], mem_k.shape[2], mem_k, mem_v, firsts)
        plan["_keep"].append(keep)

    @torch.no_grad()
    def recognize_pages(self, pages_u8: torch.Tensor, quads_per_page, max_seq_length: int = 255, suppress_eos: bool = False,
                        directions=None, group_pages: int = 8):
        """pages_u8 [P,H,W,3] u8 (device); quads_per_page[p] = list of textline.Quadrilateral.  Each line is rectified on
        the GPU straight into its chunk tensor, chunks are encoded ``group_pages`` pages at a time (row-wise operators over
        all their chunks at once), and the lines of ALL pages are decoded in one pooled beam search.  ``directions[p][i]``
        overrides a line's own direction (the reference takes a majority vote over merge-graph components,
        ocr/common.py:12-39).  Returns decode()'s dict plus ``order`` = [(page, line)] in result-row order."""
        if pages_u8.dtype != torch.uint8 or pages_u8.dim() != 4 or pages_u8.shape[-1] != 3:
            raise ValueError(f"recognize_pages expects u8 [P,H,W,3], got {pages_u8.dtype} {tuple(pages_u8.shape)}")
        pages_u8 = pages_u8.contiguous()
        P, H, W, _ = pages_u8.shape
        if len(quads_per_page) != P:
            raise ValueError("one quad list per page expected")
        plan = self.plan_pages(quads_per_page, H, W, directions)
        n_lines = len(plan["order"])
        if n_lines == 0:   # pages without text lines: nothing to upload or decode
            return dict(order=[], tokens=None)
        plan = self.upload_plan(plan)
        mem_k, mem_v = self.alloc_memory(n_lines, plan["Lmax"])
        for p0 in range(0, P, group_pages):
            ids = [i for i, c in enumerate(plan["chunks"]) if p0 <= c[4] < p0 + group_pages]
            self.encode_planned(pages_u8, plan, ids, mem_k, mem_v)
        out = self.decode(mem_k, mem_v, plan["klen_dev"], max_seq_length, suppress_eos)
        out["order"], out["_stage"] = plan["order"], plan["_keep"]
        return out

    def alloc_memory(self, n_lines: int, Lmax: int):
        """Zeroed pooled cross-attention K / V [5, n_lines, Lmax, 320] (positions beyond a line's memory stay zero and masked)."""
        return (torch.zeros(5, n_lines, Lmax, EMBD, device=self.device), torch.zeros(5, n_lines, Lmax, EMBD, device=self.device))

    @staticmethod
    def backbone_flops(N: int, Wp: int) -> float:
        """58.6 MFLOP per line x padded pixel column."""
        return 58.6e6 * N * Wp
