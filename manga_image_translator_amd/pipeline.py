"""The dense hot path of one page batch: detect (ctd) -> OCR (48px) -> inpaint (LaMa-MPE), all on one GPU.

This is the batch-mode counterpart of the three ``_infer`` bodies the reference runs one page at a time
(/root/reference/manga_translator/detection/ctd.py:129-179, ocr/model_48px.py:67-180,
inpainting/inpainting_lama_mpe.py:56-118, driven by manga_translator.py:1491-1519).  Pages, quads and masks arrive as
device-resident uint8 tensors and host-side ``Quadrilateral`` lists; only uint8 maps, token ids and a few floats per line
leave the device.

Nothing synchronises with the host; OCR lines of the whole batch are decoded in one pooled beam search.  Optionally
(``overlap=True``) LaMa runs on the caller's stream and detector + OCR on a side stream that joins at the end.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import os
import numpy as np
import torch

from . import ctd, ctd_schema, lama, lama_schema, ocr48, ocr_schema, synth
from .textline import Quadrilateral

DICT_SIZE = 6004  # synthetic dictionary size; real runs take it from alphabet-all-v7.txt


def synthetic_weights(seed: int = 0, lama_blocks: int = 9, dict_size: int = DICT_SIZE) -> Dict[str, Dict[str, torch.Tensor]]:
    """Seeded random weights in the reference architectures' own state_dict layouts (no checkpoints exist offline)."""
    g = ctd_schema.CTD_GAIN
    w = {
        "ctd.yolo": synth.synth_state_dict(ctd_schema.yolo_schema(), seed=seed, gain=g),
        "ctd.seg": synth.synth_state_dict(ctd_schema.unet_head_schema(), seed=seed, gain=g),
        "ctd.det": synth.synth_state_dict(ctd_schema.db_head_schema(), seed=seed, gain=g),
        "ocr48": synth.synth_state_dict(ocr_schema.ocr48_schema(dict_size), seed=seed),
        "lama.gen": synth.synth_state_dict(lama_schema.lama_generator_schema(lama_blocks), seed=seed),
    }
    if lama_blocks == 9:  # lama_mpe; lama_large (18 blocks) runs without MPE (inpainting_lama_mpe.py:132)
        w["lama.mpe"] = synth.synth_state_dict(lama_schema.lama_mpe_schema(), seed=seed)
    return w


@dataclass
class PageBatchResult:
    """Device tensors produced by one ``PageEngine.run``."""
    det_mask: torch.Tensor          # u8 [B, h, w]  postprocess_mask output before the resize to page size (ctd.py:30-44,152)
    det_shrink: torch.Tensor        # u8 [B, h, w]  lines[:, 0] > 0.3 (db_utils.py:75), the representer's input bitmap
    ocr_tokens: Optional[torch.Tensor]   # i32 [n_lines, T+1]
    ocr_length: Optional[torch.Tensor]   # i32 [n_lines]
    ocr_prob: Optional[torch.Tensor]     # f32 [n_lines]
    ocr_colors: Optional[torch.Tensor]   # f32 [n_lines, T, 10]
    ocr_order: List                      # [(page, line)] for every result row
    inpainted: torch.Tensor         # u8 [B, H, W, 3]
    keep: list = field(default_factory=list)  # staging buffers that must outlive the queued copies

    def packed(self) -> torch.Tensor:
        """All per-page results as one contiguous uint8 tensor (what the multi-GPU gather moves)."""
        parts = [self.det_mask.reshape(-1), self.det_shrink.reshape(-1), self.inpainted.reshape(-1)]
        if self.ocr_tokens is not None:
            parts += [self.ocr_tokens.reshape(-1).view(torch.uint8), self.ocr_length.view(torch.uint8),
                      self.ocr_prob.view(torch.uint8), self.ocr_colors.reshape(-1).view(torch.uint8)]
        return torch.cat(parts)

    def packed_pages(self, lines_per_page: int) -> torch.Tensor:
        """One fixed-size uint8 record per page, [B, record_bytes]: detector maps | inpainted page | ``lines_per_page`` OCR slots
        (tokens, length, prob, colours) placed by the line's index on its page — page i's record does not depend on which
        other pages shared its batch or on the order the pooled beam search returned the lines, which is what makes the
        multi-GPU gather order-independent (rank r's rows are the records of its contiguous page block)."""
        B = self.det_mask.shape[0]
        parts = [self.det_mask.reshape(B, -1), self.det_shrink.reshape(B, -1), self.inpainted.reshape(B, -1)]
        if self.ocr_tokens is not None:
            rows = torch.tensor([p * lines_per_page + l for p, l in self.ocr_order], dtype=torch.long)
            if rows.numel() and (int(rows.max()) >= B * lines_per_page or len(set(rows.tolist())) != rows.numel()):
                raise ValueError("packed_pages: a page has more lines than lines_per_page")
            rows = rows.to(self.ocr_tokens.device)
            for t in (self.ocr_tokens, self.ocr_length.reshape(-1, 1), self.ocr_prob.reshape(-1, 1), self.ocr_colors):
                flat = t.reshape(t.shape[0], -1).contiguous().view(torch.uint8)
                slot = torch.zeros(B * lines_per_page, flat.shape[1], dtype=torch.uint8, device=flat.device)
                slot.index_copy_(0, rows, flat)
                parts.append(slot.reshape(B, -1))
        return torch.cat(parts, dim=1)


class PageEngine:
    """Owns the three stage engines of one GPU."""

    def __init__(self, weights: Dict[str, Dict[str, torch.Tensor]], device="cuda", lama_blocks: int = 9,
                 dict_size: int = DICT_SIZE, ctd_mb: int = 16, lama_mb: int = 16, group: int = 16, overlap: bool = False):
        self.device = torch.device(device)
        self.ctd = ctd.CtdEngine(weights["ctd.yolo"], weights["ctd.seg"], weights["ctd.det"], device=self.device)
        self.ocr = ocr48.Ocr48Engine(weights["ocr48"], dict_size, device=self.device)
        self.lama = lama.LamaEngine(weights["lama.gen"], weights.get("lama.mpe"), n_blocks=lama_blocks, device=self.device)
        self.ctd_mb, self.lama_mb, self.group = ctd_mb, lama_mb, group
        self.overlap, self._side = overlap, None

    @torch.no_grad()
    def run(self, pages_u8: torch.Tensor, quads_per_page: Sequence[Sequence[Quadrilateral]], masks_u8: torch.Tensor,
            max_seq_length: int = 255, suppress_eos: bool = False, stages: Sequence[str] = ("detect", "ocr", "inpaint")
            ) -> PageBatchResult:
        """pages_u8 [B,H,W,3] u8, masks_u8 [B,H,W] u8 (0/255), both on the device; quads_per_page[b] = the page's text lines.

        The stages are fed independently (detector: pages; OCR: pages + quads; inpainter: pages + masks) because the
        reference's own stage coupling — contours, unclip, mask refinement — is host glue outside this path."""
        B, H, W, _ = pages_u8.shape
        nh, nw, dw, dh = self.ctd.letterbox_geometry(H, W)
        S = ctd.INPUT_SIZE
        dev = self.device
        det_mask = torch.zeros(B, S - dh, S - dw, dtype=torch.uint8, device=dev)
        det_shrink = torch.zeros(B, S - dh, S - dw, dtype=torch.uint8, device=dev)
        inpainted = torch.empty(B, H, W, 3, dtype=torch.uint8, device=dev) if "inpaint" in stages else pages_u8
        # overlap=True: two HIP streams — LaMa (long MFMA kernels with quantised tails) on the caller's stream, detector + OCR
        # (many short kernels) on a side stream, so the short kernels fill the CUs the big launches leave idle (+8 % pages/s
        # measured).  The stages share no buffers (separate engines / workspaces; pages and masks are read-only), so no
        # ordering is needed until the join.  Off by default: concurrent kernels stretch each other's durations, which makes
        # per-kernel roofline numbers meaningless (bench.py --overlap turns it on).
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap and "inpaint" in stages and len(stages) > 1 else main
        side.wait_stream(main)
        ocr_plan = None
        with torch.cuda.stream(side):
            if "ocr" in stages:  # host planning for every page up front (vectorised, ~1.5 ms per page), one table upload
                ocr_plan = self.ocr.upload_plan(self.ocr.plan_pages(quads_per_page, H, W))
                if len(ocr_plan["order"]) == 0:
                    ocr_plan = None
                else:
                    mem_k, mem_v = self.ocr.alloc_memory(len(ocr_plan["order"]), ocr_plan["Lmax"])
        r = None
        for g0 in range(0, B, self.group):
            g1 = min(B, g0 + self.group)
            if "inpaint" in stages:
                for i in range(g0, g1, self.lama_mb):
                    j = min(g1, i + self.lama_mb)
                    inpainted[i:j].copy_(self.lama.forward(pages_u8[i:j], masks_u8[i:j]))
            with torch.cuda.stream(side):
                if "detect" in stages:
                    for i in range(g0, g1, self.ctd_mb):
                        j = min(g1, i + self.ctd_mb)
                        m, lines, _ = self.ctd.forward(pages_u8[i:j])
                        det_mask[i:j].copy_(m)
                        det_shrink[i:j].copy_(self.ctd.shrink_bitmap(lines))
                if ocr_plan is not None:  # backbone + encoder of this group's chunks; their K/V land in the pooled memory
                    ids = [i for i, c in enumerate(ocr_plan["chunks"]) if g0 <= c[4] < g1]
                    self.ocr.encode_planned(pages_u8, ocr_plan, ids, mem_k, mem_v)
        if ocr_plan is not None:
            with torch.cuda.stream(side):
                # one beam search over the lines of the whole batch: decode GEMMs see 5 x n_lines rows instead of 5 x 16
                r = self.ocr.decode(mem_k, mem_v, ocr_plan["klen_dev"], max_seq_length, suppress_eos)
        main.wait_stream(side)
        if r is None:
            return PageBatchResult(det_mask, det_shrink, None, None, None, None, [], inpainted, [])
        for t in (r["tokens"], r["length"], r["prob"], r["colors"]):
            t.record_stream(main)  # allocated on the side stream, consumed by the caller on the main one
        return PageBatchResult(det_mask, det_shrink, r["tokens"], r["length"], r["prob"], r["colors"], ocr_plan["order"], inpainted,
                               ocr_plan["_keep"] + [mem_k, mem_v])

    def _side_stream(self):
        if self._side is None:
            # (MIT_SIDE_STREAM_PRIORITY=-1 gives the detector + OCR stream precedence over LaMa's queue: measured +2.8 % vs +4.2 % for the default 0)
            self._side = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("MIT_SIDE_STREAM_PRIORITY", "0")))
        return self._side

    def flops_per_page(self, H: int, W: int, line_widths: Sequence[int], steps: int) -> Dict[str, float]:
        """Algorithmic FLOPs of one page (SURVEY.md §8d conventions), per stage."""
        ocr_bb = sum(58.6e6 * len(ws) * wp for ws, wp in line_widths)
        return {"detect": self.ctd.flops_per_page(), "ocr_backbone": ocr_bb, "inpaint": self.lama.flops_per_page(H, W)}


def quads_from_array(quads: np.ndarray) -> List[Quadrilateral]:
    """int [K,4,2] corner arrays -> Quadrilateral objects."""
    return [Quadrilateral(q) for q in np.asarray(quads)]
