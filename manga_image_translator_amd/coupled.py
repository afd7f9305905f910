"""The COUPLED page path: detector -> boxes -> refine_mask -> OCR -> text-line merge -> mask refinement -> inpainting, for a batch
of pages, in the order and with the arguments the reference's orchestrator uses for one page
(/root/reference/manga_translator/manga_translator.py:432-622: ``_run_detection`` :1208, ``_run_ocr`` :745,
``_run_textline_merge`` :772, ``_run_mask_refinement`` :1356-1358 with ``mask_dilation_offset`` = 20 and ``kernel_size`` = 3
(config.py:342-344), ``_run_inpainting`` :1360-1364).

``pipeline.PageEngine`` times the three dense stages fed independently; here each stage consumes what the previous one
produced, so the glue between them — box extraction (a4), the detector's ``refine_mask`` (a5), the merge graph (f3), the mask
refinement with its DenseCRF (f1) — is inside the measured path.  Dense work runs batched on the GPU; the per-page host steps
(contours -> boxes in native C++, direction vote, merge graph, component labelling) run on a thread pool (ctypes / numpy / scipy
release the GIL) while the stream keeps executing.  A batch larger than one group of 16 pages flows through three stage threads
(detector + boxes + refine_mask | OCR | merge + mask refinement + LaMa) that work on different groups at the same time (``run(group=)``).

Random-init networks do not detect text (their sigmoid maps hover around 0.5 everywhere), so a benchmark passes ``inject``: per-page
maps a trained head would have produced for the synthetic page (``synthetic_head_outputs``); they replace the network's own outputs
AFTER the network has run (its cost is paid in full) — a stand-in for trained weights, never part of the product path (the plugins do
not know about it).
"""
from __future__ import annotations

import concurrent.futures as cf
import math
import os
import threading
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ctd, hostglue, imgproc, lama, mask_refinement as MR, ocr48, textline_merge as TM
from .textline import Quadrilateral

BOX_THRESH = 0.6          # ctd.py:157-159
GPU_BOXES = os.environ.get("MIT_BOXES_HOST", "0") in ("", "0")   # MIT_BOXES_HOST=1: the per-page host extraction of rounds 3-5 (A/B)
MASK_DILATION_OFFSET = 20  # config.py:344
KERNEL_SIZE = 3            # config.py:342


@dataclass
class CoupledResult:
    textlines: List[List[Quadrilateral]]      # per page: the detector's lines with OCR text / prob / colours set
    regions: List[list]                        # per page: TextBlocks of the merge
    mask: torch.Tensor                         # u8 [B,H,W] final inpainting mask (device)
    inpainted: torch.Tensor                    # u8 [B,H,W,3] (device)
    seconds: Dict[str, float] = field(default_factory=dict)   # host wall time per phase (the GPU runs asynchronously underneath)


def synthetic_head_outputs(page: np.ndarray, quads: np.ndarray, map_hw, shrink_ratio: float = 0.4, unclip_ratio: float = 1.5):
    """What a TRAINED ctd head would emit for a synthetic page (benchmark stand-in for weights, see the module docstring):
    * the DB shrink map [h, w] float32: 0.9 inside every text box shrunk so that SegDetectorRepresenter's unclip
      (distance = area * 1.5 / perimeter, db_utils.py:150-155) grows it back to the box — d solves d * 2(w + h - 4d) = 1.5 (w - 2d)(h - 2d);
    * the text mask [h, w] uint8: 255 on the dark (glyph) pixels inside the boxes.
    ``map_hw`` = the network's un-padded output size (page / 2 for a 2048 x 1456 page)."""
    H, W = page.shape[:2]
    h, w = map_hw
    sy, sx = h / H, w / W
    prob = np.zeros((h, w), np.float32)
    box = np.zeros((H, W), bool)
    for q in np.asarray(quads):
        x0, y0, x1, y1 = q[:, 0].min() * sx, q[:, 1].min() * sy, q[:, 0].max() * sx, q[:, 1].max() * sy
        bw, bh = x1 - x0, y1 - y0
        S = bw + bh
        disc = 25 * S * S - 84 * bw * bh
        d = (5 * S - math.sqrt(max(disc, 0.0))) / 28 if bw > 0 and bh > 0 else 0.0
        d = min(d, 0.45 * min(bw, bh))
        ya, yb, xa, xb = int(round(y0 + d)), int(round(y1 - d)), int(round(x0 + d)), int(round(x1 - d))
        if yb > ya and xb > xa:
            prob[ya:yb, xa:xb] = 0.9
        box[int(q[:, 1].min()):int(q[:, 1].max()), int(q[:, 0].min()):int(q[:, 0].max())] = True
    glyph = ((page.min(-1) < 128) & box).astype(np.uint8) * 255
    mask = imgproc.resize_u8_host(glyph, (w, h))
    mask[mask > 0] = 255
    return prob, mask


_STAGE_OF = {"detect+boxes+refine_mask": "det", "ocr": "ocr", "textline_merge+mask_refinement": "tail", "inpaint (enqueue)": "tail"}


class CoupledPageEngine:
    """Owns the stage engines of one GPU and a host thread pool."""

    def __init__(self, weights: Dict[str, Dict[str, torch.Tensor]], dictionary: Sequence[str], device="cuda", lama_blocks: int = 9,
                 ctd_mb: int = 16, lama_mb: int = 16, host_workers: int = 16, mask_workers: int = 4, side_stream: Optional[bool] = None):
        self.device = torch.device(device)
        self.dictionary = list(dictionary)
        self.ctd = ctd.CtdEngine(weights["ctd.yolo"], weights["ctd.seg"], weights["ctd.det"], device=self.device)
        self.ocr = ocr48.Ocr48Engine(weights["ocr48"], len(self.dictionary), device=self.device)
        self.lama = lama.LamaEngine(weights["lama.gen"], weights.get("lama.mpe"), n_blocks=lama_blocks, device=self.device)
        self.ctd_mb, self.lama_mb = ctd_mb, lama_mb
        self.pool = cf.ThreadPoolExecutor(max_workers=host_workers, thread_name_prefix="mit-host")
        # mask refinement: a few pages in flight at once — each worker thread owns a backend (its DenseCRF workspace); their launches
        # go to the same stream, a page's own launches stay in order on it, and the host parts (labelling, assignment) of one page
        # run while another page's kernels execute
        self.mask_pool = cf.ThreadPoolExecutor(max_workers=mask_workers, thread_name_prefix="mit-mask")
        self._tls = threading.local()
        self.side_stream = bool(int(os.environ.get("MIT_COUPLED_SIDE_STREAM", "1"))) if side_stream is None else bool(side_stream)
        self._side = None
        self.stage_streams = bool(int(os.environ.get("MIT_COUPLED_STAGE_STREAMS", "0")))
        self._stage = None
        self.ocr_slots = int(os.environ.get("MIT_COUPLED_OCR_SLOTS", "1"))   # pipeline slots recognised together (the OCR stage's own granularity)

    def _mask_backend(self):
        be = getattr(self._tls, "backend", None)
        if be is None:
            be = self._tls.backend = MR.GpuMaskBackend(self.device)
        return be

    def close(self):
        self.pool.shutdown(wait=True)
        self.mask_pool.shutdown(wait=True)

    # ---- stage 1: detector network + boxes (host pool) + mask resize + refine_mask ------------------------------------------
    @torch.no_grad()
    def detect(self, pages_u8: torch.Tensor, inject=None):
        """-> (textlines per page, refined mask u8 [B,H,W] on the device): ComicTextDetector._infer (ctd.py:129-179) for a batch."""
        B, H, W, _ = pages_u8.shape
        futures: List[Optional[cf.Future]] = [None] * B
        mask_full = torch.empty(B, H, W, dtype=torch.uint8, device=self.device)
        keep = []
        for i in range(0, B, self.ctd_mb):
            j = min(B, i + self.ctd_mb)
            mask_u8, lines, _ = self.ctd.forward(pages_u8[i:j])
            if inject is not None:   # benchmark stand-in for trained weights: the maps a trained head would emit REPLACE the random-init
                lines[:, 0] = inject["prob"][i:j]     # network's (its sigmoid output hovers around 0.5 everywhere: one page-sized blob);
                mask_u8 = inject["mask"][i:j]         # the network has run in full by now, its cost is in the measurement
            if GPU_BOXES:   # SegDetectorRepresenter (db_utils.py:40-216) where the map is (csrc/ctd_boxes.hip): enqueued behind the network,
                keep.append((i, j, hostglue.boxes_from_bitmap_gpu_launch(lines[:, 0], 0.3, W, H, unclip_ratio=1.5, min_sside=2.0)))   # collected below
                mask_full[i:j] = imgproc.resize_u8(mask_u8.contiguous(), (W, H))      # cv2.resize(mask, (w, h), INTER_LINEAR) (ctd.py:162)
                continue
            host = torch.empty(lines.shape, dtype=torch.float32, pin_memory=True)   # [b,2,h,w]: box_score_fast needs the float map
            host.copy_(lines, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            mask_full[i:j] = imgproc.resize_u8(mask_u8.contiguous(), (W, H))          # cv2.resize(mask, (w, h), INTER_LINEAR) (ctd.py:162)
            keep.append(host)
            for b in range(i, j):
                futures[b] = self.pool.submit(self._boxes_of_page, host, b - i, ev, H, W)
        if GPU_BOXES:
            textlines = [None] * B
            for i, j, h in keep:
                for b, (boxes, scores) in zip(range(i, j), hostglue.boxes_from_bitmap_gpu_collect(h)):
                    k = scores > BOX_THRESH
                    textlines[b] = [Quadrilateral(pts.astype(np.int64), "", float(s)) for pts, s in zip(boxes[k], scores[k])]
        else:
            textlines = [f.result() for f in futures]
        refined = torch.empty_like(mask_full)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        caller = torch.cuda.current_stream()

        def refine(b):  # refine_mask(img, mask, textlines) (ctd.py:177): batched over the page's lines on the device; its three phases
            torch.cuda.set_device(dev_index)   # alternate with host arithmetic (histograms -> candidates), so a few pages run interleaved
            with torch.no_grad(), torch.cuda.stream(caller):   # a pool thread's own current stream is the default one: launch into the caller's
                return hostglue.refine_mask_gpu(pages_u8[b], mask_full[b], textlines[b], None)

        for b, m in enumerate(self.mask_pool.map(refine, range(B))):
            refined[b] = m
        # the event recorded behind the raw masks' last writer TRAVELS WITH THE TENSOR (merge_and_refine's side stream waits for this, not
        # for the whole queue): no table keyed by address that a failed stage could leave behind for an unrelated tensor to inherit
        refined.mit_ready_event = torch.cuda.current_stream().record_event()
        return textlines, refined

    @staticmethod
    def _boxes_of_page(host_lines: torch.Tensor, k: int, ev, H: int, W: int) -> List[Quadrilateral]:
        ev.synchronize()
        boxes, scores = hostglue.ctd_boxes(host_lines[k:k + 1].numpy(), H, W)    # SegDetectorRepresenter (db_utils.py:40-216), native C++
        keep = scores > BOX_THRESH
        return [Quadrilateral(pts.astype(np.int64), "", float(s)) for pts, s in zip(boxes[keep], scores[keep])]

    # ---- stage 2: OCR ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def recognize(self, pages_u8: torch.Tensor, textlines: List[List[Quadrilateral]], max_seq_length: int, suppress_eos: bool,
                  prob_threshold: float):
        """Model48pxOCR._infer (model_48px.py:67-180) for a batch: direction vote per page (host pool), rectification + recognition
        + one pooled beam search on the GPU, text / probability / colours written back into the lines; lines under the threshold dropped."""
        from . import plugins as P, textline as TL

        def vote(lines):
            return list(TL.generate_text_direction(lines))

        pairs = list(self.pool.map(vote, textlines))
        quads = [[q for q, _ in pr] for pr in pairs]
        dirs = [[d for _, d in pr] for pr in pairs]
        r = self.ocr.recognize_pages(pages_u8, quads, max_seq_length=max_seq_length, suppress_eos=suppress_eos, directions=dirs)
        out: List[List[Quadrilateral]] = [[] for _ in textlines]
        if not r["order"]:
            return out
        toks, lens = r["tokens"].cpu().numpy(), r["length"].cpu().numpy()
        probs, cols = r["prob"].cpu().numpy(), r["colors"].cpu().numpy()
        decoded = P.decode_lines(toks, lens, cols, self.dictionary, rows=[row for row in range(len(r["order"])) if probs[row] >= prob_threshold])
        for row, (p, i) in enumerate(r["order"]):
            q, prob = quads[p][i], float(probs[row])
            q.assigned_direction = dirs[p][i]
            if prob < prob_threshold:
                continue
            q.text, (q.fg_r, q.fg_g, q.fg_b), (q.bg_r, q.bg_g, q.bg_b) = decoded[row]
            q.prob = prob
            if q.text.strip():                   # manga_translator.py:762-770: lines without text are dropped after OCR
                out[p].append(q)
        return out

    # ---- stages 3 + 4: text-line merge, mask refinement -------------------------------------------------------------------
    def merge_and_refine(self, pages_u8: torch.Tensor, textlines: List[List[Quadrilateral]], mask_raw: torch.Tensor):
        B, H, W, _ = pages_u8.shape
        main = torch.cuda.current_stream()
        # Mask refinement alternates short kernels with host arithmetic (labelling -> assignment -> CRF -> dilation): on the caller's
        # stream each of its read-backs waits for whatever the other stages queued ahead (a LaMa group: a quarter of a second), and the
        # GPU idles through its host phases once that has drained.  With ``side_stream`` (the default; MIT_COUPLED_SIDE_STREAM=0 turns it
        # off) the page workers launch into a high-priority stream of their own instead, beside the queued work: it waits for the event
        # the detector stage recorded behind the raw masks (not for the queue), and the inpainter's launches (caller's stream) wait for
        # its end.  23.4 -> 26.2 pages/s; same bytes as the one-stream path (bench: coupled.batch.pipeline.side_stream_results_equal_one_stream)
        # — since the library is built without the SLP vectoriser's packed-fp32 instructions (with them 3-7 of 64 pages
        # came back with a few dozen mask bytes different whenever two streams' kernels were resident at once).
        ready = getattr(mask_raw, "mit_ready_event", None)   # set by detect(); a mask from anywhere else has none: one-stream path
        side = self._side_stream() if self.side_stream and ready is not None else main
        with torch.cuda.stream(side):
            if side is not main:
                side.wait_event(ready)
                mask_raw.record_stream(side)
            final = torch.zeros(B, H, W, dtype=torch.uint8, device=self.device)   # (from the side stream's own pool of blocks)
        regions = list(self.pool.map(lambda ls: TM.dispatch_sync(ls, W, H) if ls else [], textlines))
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def refine(b):
            if not regions[b]:
                return None      # no text: the orchestrator returns the page as it is (manga_translator.py:500-504)
            torch.cuda.set_device(dev_index)
            with torch.no_grad(), torch.cuda.stream(side):
                m = MR.dispatch_device(regions[b], pages_u8[b], mask_raw[b], dilation_offset=MASK_DILATION_OFFSET, kernel_size=KERNEL_SIZE,
                                       backend=self._mask_backend())
                final[b] = m
                return True

        list(self.mask_pool.map(refine, range(B)))
        if side is not main:
            main.wait_stream(side)
            final.record_stream(main)
        return regions, final

    def _stage_stream_set(self):
        if self._stage is None:
            self._stage = {k: torch.cuda.Stream(device=self.device) for k in ("det", "ocr", "tail")}
        return self._stage

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device, priority=-1)
        return self._side

    # ---- the whole path -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, pages_u8: torch.Tensor, max_seq_length: int = 255, suppress_eos: bool = False, prob_threshold: float = 0.2,
            inject=None, group: Optional[int] = 16) -> CoupledResult:
        """``group``: pages per pipeline slot.  A batch larger than one group flows through three stage threads (detector + boxes +
        refine_mask | OCR | merge + mask refinement + LaMa), each working on a different group at a time, so the host phases of one
        group run while the kernels of its neighbours execute; every stage sees the groups in order and owns its engines, and each
        page's result is what the unpipelined call (``group=None``) gives."""
        if pages_u8.dtype != torch.uint8 or pages_u8.dim() != 4 or pages_u8.shape[-1] != 3 or not pages_u8.is_cuda:
            raise ValueError(f"CoupledPageEngine.run expects a uint8 device tensor [B,H,W,3], got {pages_u8.dtype} {tuple(pages_u8.shape)}")
        B = pages_u8.shape[0]
        if group is not None and B > group:
            return self._run_pipelined(pages_u8, max_seq_length, suppress_eos, prob_threshold, inject, int(group))
        sec = {}
        t = time.perf_counter()
        textlines, mask_raw = self.detect(pages_u8, inject)
        sec["detect+boxes+refine_mask"] = time.perf_counter() - t
        t = time.perf_counter()
        textlines = self.recognize(pages_u8, textlines, max_seq_length, suppress_eos, prob_threshold)
        sec["ocr"] = time.perf_counter() - t
        t = time.perf_counter()
        regions, mask = self.merge_and_refine(pages_u8, textlines, mask_raw)
        sec["textline_merge+mask_refinement"] = time.perf_counter() - t
        t = time.perf_counter()
        inpainted = torch.empty_like(pages_u8)
        for i in range(0, B, self.lama_mb):
            j = min(B, i + self.lama_mb)
            inpainted[i:j].copy_(self.lama.forward(pages_u8[i:j], mask[i:j]))
        sec["inpaint (enqueue)"] = time.perf_counter() - t
        return CoupledResult(textlines, regions, mask, inpainted, sec)

    def _inpaint(self, pages_u8: torch.Tensor, mask: torch.Tensor, out: torch.Tensor):
        for i in range(0, pages_u8.shape[0], self.lama_mb):
            j = min(pages_u8.shape[0], i + self.lama_mb)
            out[i:j].copy_(self.lama.forward(pages_u8[i:j], mask[i:j]))

    def _run_pipelined(self, pages_u8, max_seq_length, suppress_eos, prob_threshold, inject, group) -> CoupledResult:
        B, H, W, _ = pages_u8.shape
        spans = [(i, min(B, i + group)) for i in range(0, B, group)]
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        sec = {"detect+boxes+refine_mask": 0.0, "ocr": 0.0, "textline_merge+mask_refinement": 0.0, "inpaint (enqueue)": 0.0}
        mask = torch.empty(B, H, W, dtype=torch.uint8, device=self.device)
        inpainted = torch.empty_like(pages_u8)
        caller = torch.cuda.current_stream()
        # One stream per stage thread (``stage_streams``, MIT_COUPLED_STAGE_STREAMS=1): a stage's read-backs wait for ITS kernels only, and
        # the kernels of the stages overlap on the GPU.  The streams start behind everything the caller queued, and the caller's stream
        # continues behind them at the end.  Same results; measured 25.4 vs 26.1 pages/s for the default (every stage thread launches
        # into the caller's stream, only the mask stage has its side stream): with 35 ms of kernels per page the GPU is the bound, and
        # the latency-bound OCR decode is the stage that loses when it has to share the CUs with LaMa.
        if self.stage_streams:
            st = self._stage_stream_set()
            start = caller.record_event()
            for x in st.values():
                x.wait_event(start)
        else:
            st = {"det": caller, "ocr": caller, "tail": caller}

        def timed(key, fn, *a):
            torch.cuda.set_device(dev_index)
            with torch.no_grad(), torch.cuda.stream(st[_STAGE_OF[key]]):
                t = time.perf_counter()
                r = fn(*a)
                sec[key] += time.perf_counter() - t           # one thread per key: no race
                return r

        def st_detect(a, b):
            inj = None if inject is None else {k: v[a:b] for k, v in inject.items()}
            return timed("detect+boxes+refine_mask", self.detect, pages_u8[a:b], inj)

        def st_ocr(first, futs):   # ``ocr_slots`` consecutive slots in one recognition: its beam search sees that many more rows per launch
            dets = [f.result() for f in futs]
            a, b = spans[first][0], spans[first + len(futs) - 1][1]
            tl = timed("ocr", self.recognize, pages_u8[a:b], [t for d in dets for t in d[0]], max_seq_length, suppress_eos, prob_threshold)
            out, k = [], 0
            for (sa, sb), d in zip(spans[first:first + len(futs)], dets):
                out.append((tl[k:k + sb - sa], d[1]))
                k += sb - sa
            return out

        def st_tail(a, b, f_ocr, k):
            tl, mraw = f_ocr.result()[k]
            regions, m = timed("textline_merge+mask_refinement", self.merge_and_refine, pages_u8[a:b], tl, mraw)
            with torch.cuda.stream(st["tail"]):
                mask[a:b] = m
            timed("inpaint (enqueue)", self._inpaint, pages_u8[a:b], mask[a:b], inpainted[a:b])
            return tl, regions

        with cf.ThreadPoolExecutor(1, "mit-st-det") as e1, cf.ThreadPoolExecutor(1, "mit-st-ocr") as e2, cf.ThreadPoolExecutor(1, "mit-st-tail") as e3:
            f1 = [e1.submit(st_detect, a, b) for a, b in spans]
            ns = max(1, int(self.ocr_slots))
            f2 = [e2.submit(st_ocr, i, f1[i:i + ns]) for i in range(0, len(spans), ns)]
            f3 = [e3.submit(st_tail, a, b, f2[i // ns], i % ns) for i, (a, b) in enumerate(spans)]
            done = [f.result() for f in f3]
        for x in st.values():
            if x is not caller:
                caller.wait_stream(x)
        textlines = [t for tl, _ in done for t in tl]
        regions = [r for _, rg in done for r in rg]
        return CoupledResult(textlines, regions, mask, inpainted, sec)
