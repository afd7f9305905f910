"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference 48px OCR model.

Functional fp32 torch-CPU restatement of
  /root/reference/manga_translator/ocr/model_48px.py  (ConvNext_FeatureExtractor :216-276,
  XposMultiheadAttention :294-394, OCR.encoder_forward :543-546, OCR.decoder_forward :548-572,
  OCR.infer_beam_batch_tensor :678-801) and ocr/xpos_relative_position.py (:9-71),
driven by a state_dict with the reference's key names.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product never does.

The restatement keeps the reference's behaviour including its quirks: K/V are recomputed from the
cached layer inputs at every step (:561-566), XPOS positions are centred on the current length
(xpos_relative_position.py:54-59), and the beam step re-orders ``out_idx`` / ``log_probs`` but NOT
``cached_activations`` (:730-735) — row r keeps its own history whatever hypothesis lands on it.
Parity status: pinned against the reference module imported in the build container
(tests/test_oracle_vs_reference.py); the reference's own tests hold no vectors for this path.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
EMBD, HEADS, HEAD_DIM = 320, 4, 80


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _cbr(x, sd, p, i, stride, padding):
    return torch.relu(_bn(F.conv2d(x, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"], stride=stride, padding=padding), sd, f"{p}.{i + 1}"))


def _cnblock(x, sd, p, ks):
    """ConvNeXtBlock.forward :203-214 (BatchNorm eps 1e-6 :196)."""
    dim = x.shape[1]
    y = F.conv2d(x, sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"], padding=ks // 2, groups=dim)
    y = _bn(y, sd, p + ".norm", 1e-6)
    y = F.gelu(F.conv2d(y, sd[p + ".pwconv1.weight"], sd[p + ".pwconv1.bias"]))
    y = F.conv2d(y, sd[p + ".pwconv2.weight"], sd[p + ".pwconv2.bias"])
    return x + sd[p + ".gamma"] * y


def backbone(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """ConvNext_FeatureExtractor.forward :262-276. [N,3,48,W] -> [N,320,1,W']."""
    p = "backbone."
    x = _cbr(x, sd, p + "stem", 0, 1, 3)
    x = _cbr(x, sd, p + "stem", 3, 2, 0)
    x = _cbr(x, sd, p + "stem", 6, 1, 1)
    for i in range(4):
        x = _cnblock(x, sd, f"{p}block1.{i}", 7)
    x = _cbr(x, sd, p + "down1", 0, 2, 0)
    for i in range(12):
        x = _cnblock(x, sd, f"{p}block2.{i}", 7)
    x = _cbr(x, sd, p + "down2", 0, (2, 1), 0)
    for i in range(10):
        x = _cnblock(x, sd, f"{p}block3.{i}", 5)
    x = _cbr(x, sd, p + "down3", 0, (2, 1), 0)
    for i in range(8):
        x = _cnblock(x, sd, f"{p}block4.{i}", 3)
    return _cbr(x, sd, p + "down4", 0, 1, 0)


# ---- XPOS (ocr/xpos_relative_position.py) ----

def xpos_tables(scale_vec: torch.Tensor, length: int, offset: int, downscale: bool):
    """The (cos*scale, sin*scale) rows XPOS.forward (:54-71) applies to a [*, length, 80] tensor."""
    min_pos = -(length + offset) // 2
    max_pos = length + offset + min_pos
    scale = scale_vec ** torch.arange(min_pos, max_pos, 1).to(scale_vec).div(EMBD)[:, None]  # scale_base = 320 (:316)
    seq_len, dim = scale.shape
    inv_freq = 1.0 / (10000 ** (torch.arange(0, dim) / dim))
    sinusoid = torch.einsum("i , j -> i j", torch.arange(0, seq_len, dtype=torch.float), inv_freq).to(scale)
    sin, cos = torch.sin(sinusoid), torch.cos(sinusoid)
    if scale.shape[0] > length:
        scale, sin, cos = scale[-length:], sin[-length:], cos[-length:]
    if downscale:
        scale = 1 / scale
    dup = lambda m: m.view(-1, 1).repeat(1, 2).view(m.shape[0], -1)
    return dup(sin * scale), dup(cos * scale)


def xpos_apply(x: torch.Tensor, scale_vec: torch.Tensor, offset: int, downscale: bool) -> torch.Tensor:
    sin, cos = xpos_tables(scale_vec, x.shape[1], offset, downscale)
    x1, x2 = x[:, :, ::2], x[:, :, 1::2]
    rot = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return x * cos + rot * sin


def xpos_mha(sd: SD, p: str, query, key, value, key_padding_mask=None, q_offset=0):
    """XposMultiheadAttention.forward :327-394 (k_offset is always 0 at the call sites)."""
    bsz, tgt_len, _ = query.shape
    src_len = key.shape[1]
    q = F.linear(query, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]) * (HEAD_DIM ** -0.5)
    k = F.linear(key, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(value, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])
    sh = lambda t, n: t.view(bsz, n, HEADS, HEAD_DIM).transpose(1, 2).reshape(bsz * HEADS, n, HEAD_DIM)
    q, k, v = sh(q, tgt_len), sh(k, src_len), sh(v, src_len)
    k = xpos_apply(k, sd[p + ".xpos.scale"], 0, True)
    q = xpos_apply(q, sd[p + ".xpos.scale"], q_offset, False)
    w = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        w = w.view(bsz, HEADS, tgt_len, src_len).masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        w = w.view(bsz * HEADS, tgt_len, src_len)
    w = F.softmax(w, dim=-1, dtype=torch.float32)
    a = torch.bmm(w, v).transpose(0, 1).reshape(tgt_len, bsz, EMBD).transpose(0, 1)
    return F.linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (EMBD,), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _ffn(x, sd, p):
    return F.linear(torch.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])


def encoder(sd: SD, memory: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """OCR.encoder_forward :543-546 / transformer_encoder_forward :278-292 (norm_first)."""
    for i in range(4):
        p = f"encoders.{i}"
        n = _ln(memory, sd, p + ".norm1")
        memory = memory + xpos_mha(sd, p + ".self_attn", n, n, n, key_padding_mask=mask)
        memory = memory + _ffn(_ln(memory, sd, p + ".norm2"), sd, p)
    return memory


def decoder_step(sd: SD, embd, cached, memory, memory_mask, step: int):
    """OCR.decoder_forward :548-572. cached [R, 6, T, 320] is updated in place."""
    tgt = embd
    for l in range(5):
        p = f"decoders.{l}"
        comb = torch.cat([cached[:, l, :step, :], tgt], dim=1)
        cached[:, l, step, :] = tgt.squeeze(1)
        n1 = _ln(comb, sd, p + ".norm1")
        tgt = tgt + xpos_mha(sd, p + ".self_attn", _ln(tgt, sd, p + ".norm1"), n1, n1, q_offset=step)
        tgt = tgt + xpos_mha(sd, p + ".multihead_attn", _ln(tgt, sd, p + ".norm2"), memory, memory, key_padding_mask=memory_mask, q_offset=step)
        tgt = tgt + _ffn(_ln(tgt, sd, p + ".norm3"), sd, p)
    cached[:, 5, step, :] = tgt.squeeze(1)
    return tgt.squeeze(1)


def _logprobs(sd: SD, decoded):
    h = F.gelu(F.linear(decoded, sd["pred1.0.weight"], sd["pred1.0.bias"]))
    return F.linear(h, sd["pred.weight"], sd["pred.bias"]).log_softmax(-1)


def encode_lines(sd: SD, img: torch.Tensor, widths: List[int]):
    """infer_beam_batch_tensor :682-689: backbone + masked encoder. Returns (memory [N,L,320], mask [N,L])."""
    mem = backbone(sd, img)
    mem = mem.squeeze(2).permute(0, 2, 1)  # 'N C 1 W -> N W C'
    mask = torch.zeros(img.shape[0], mem.shape[1], dtype=torch.bool)
    for i, w in enumerate(widths):
        mask[i, (w + 3) // 4 + 2:] = True
    return encoder(sd, mem, mask), mask


def infer_beam_batch_tensor(sd: SD, img: torch.Tensor, widths: List[int], beams_k: int = 5, start_tok=1, end_tok=2,
                            max_finished_hypos: int = 2, max_seq_length: int = 255, trace: Optional[list] = None,
                            suppress_eos: bool = False):
    """OCR.infer_beam_batch_tensor :678-801.  ``trace`` (optional list) receives, per step, the raw logits
    ("OCR logits" of BASELINE.md = pred(pred1(decoded)), via their log-softmax) and the beam state.
    ``suppress_eos`` sets the end-token log-prob to -inf (the fixed-length timing configuration)."""
    N = img.shape[0]
    memory, input_mask = encode_lines(sd, img, widths)

    def lp(decoded):
        x = _logprobs(sd, decoded)
        if suppress_eos:
            x = x.clone()
            x[:, end_tok] = float("-inf")
        return x

    out_idx = torch.full((N, 1), start_tok, dtype=torch.long)
    cached = torch.zeros(N, 6, max_seq_length, EMBD)
    decoded = decoder_step(sd, F.embedding(out_idx[:, -1:], sd["embd.weight"]), cached, memory, input_mask, 0)
    logp = lp(decoded)
    vals, idx = torch.topk(logp, beams_k, dim=1)
    if trace is not None:
        trace.append(dict(step=0, logp=logp.clone(), out_idx=out_idx.clone()))
    out_idx = torch.cat([out_idx.unsqueeze(1).expand(-1, beams_k, -1), idx.unsqueeze(-1)], dim=-1).reshape(-1, 2)
    log_probs = vals.reshape(-1, 1)
    memory = memory.repeat_interleave(beams_k, dim=0)
    input_mask = input_mask.repeat_interleave(beams_k, dim=0)
    cached = cached.repeat_interleave(beams_k, dim=0)
    batch_index = torch.arange(N).repeat_interleave(beams_k, dim=0)
    finished = {}
    n_rem = N
    for step in range(1, max_seq_length):
        decoded = decoder_step(sd, F.embedding(out_idx[:, -1:], sd["embd.weight"]), cached, memory, input_mask, step)
        logp = lp(decoded)
        vals, idx = torch.topk(logp, beams_k, dim=1)
        if trace is not None:
            trace.append(dict(step=step, logp=logp.clone(), out_idx=out_idx.clone(), batch_index=batch_index.clone()))
        fin = out_idx[:, -1] == end_tok
        vals[fin] = 0  # :716-718
        idx[fin] = end_tok
        new_idx = torch.cat([out_idx.unsqueeze(1).expand(-1, beams_k, -1), idx.unsqueeze(-1)], dim=-1).view(n_rem, -1, step + 2)
        new_lp = (log_probs.unsqueeze(1).expand(-1, beams_k, -1) + vals.unsqueeze(-1)).view(n_rem, -1)
        top_lp, top_i = new_lp.topk(beams_k, dim=1)  # :730
        out_idx = torch.gather(new_idx, 1, top_i.unsqueeze(-1).expand(-1, -1, step + 2)).reshape(-1, step + 2)
        log_probs = top_lp.reshape(-1, 1)
        fcount = (out_idx[:, -1] == end_tok).view(n_rem, beams_k).sum(dim=1)
        done = (fcount >= max_finished_hypos).nonzero(as_tuple=False).flatten().tolist()
        if not done:
            continue
        for i in done:  # :748-754
            best = int(top_lp[i].argmax())
            finished[int(batch_index[beams_k * i])] = (out_idx[i * beams_k + best], float(torch.exp(top_lp[i][best])),
                                                       cached[i * beams_k + best])
        keep = [i * beams_k + j for i in range(n_rem) if i not in done for j in range(beams_k)]
        if not keep:
            break
        n_rem = len(keep) // beams_k
        sel = torch.tensor(keep)
        out_idx, log_probs, memory = out_idx[sel], log_probs[sel], memory[sel]
        cached, input_mask, batch_index = cached[sel], input_mask[sel], batch_index[sel]
    for i in range(N):  # fallback :774-784
        if i not in finished:
            rows = (batch_index == i).nonzero(as_tuple=True)[0]
            r = rows[0]
            finished[i] = (out_idx[r], float(torch.exp(log_probs[r])), cached[r])
    result = []
    for i in range(N):  # :789-799
        final_idx, prob, cache = finished[i]
        feats = torch.relu(F.linear(cache[-1].unsqueeze(0), sd["color_pred1.0.weight"], sd["color_pred1.0.bias"]))
        heads = [F.linear(feats, sd[n + ".weight"], sd[n + ".bias"])[0]
                 for n in ("color_pred_fg", "color_pred_bg", "color_pred_fg_ind", "color_pred_bg_ind")]
        result.append((final_idx[1:], prob, *heads))
    return result


def make_chunks(region_imgs: List[np.ndarray], max_chunk_size: int = 16):
    """Host batching of Model48pxOCR._infer :79-91,115-116: sort by width, chunks of 16, zero-pad the uint8
    crops to max(w)+7 (the ``4 * (max + 7) // 4`` quirk), normalise to [-1, 1]. Yields (indices, widths, tensor)."""
    perm = sorted(range(len(region_imgs)), key=lambda i: region_imgs[i].shape[1])
    for c in range(0, len(perm), max_chunk_size):
        indices = perm[c:c + max_chunk_size]
        widths = [region_imgs[i].shape[1] for i in indices]
        max_width = 4 * (max(widths) + 7) // 4
        region = np.zeros((len(indices), 48, max_width, 3), dtype=np.uint8)
        for j, i in enumerate(indices):
            region[j, :, :widths[j], :] = region_imgs[i]
        t = (torch.from_numpy(region).float() - 127.5) / 127.5
        yield indices, widths, t.permute(0, 3, 1, 2).contiguous()
