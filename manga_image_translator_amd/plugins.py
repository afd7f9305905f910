"""Drop-in detector / OCR / inpainter plugins: the reference's plugin interface over the gfx950 engine.

Each class honours the contract of its reference counterpart:

  HipComicTextDetector   <- ComicTextDetector   (/root/reference/manga_translator/detection/ctd.py:60-179)
  HipDefaultDetector     <- DefaultDetector     (detection/default.py:27-103)
  HipModel48pxOCR        <- Model48pxOCR        (ocr/model_48px.py:25-180)
  HipModel48pxCTCOCR     <- Model48pxCTCOCR     (ocr/model_48px_ctc.py:30-160)
  HipLamaMPEInpainter    <- LamaMPEInpainter    (inpainting/inpainting_lama_mpe.py:26-118)
  HipLamaLargeInpainter  <- LamaLargeInpainter  (inpainting/inpainting_lama_mpe.py:121-136)
  HipESRGANUpscaler      <- ESRGANUpscalerPytorch (upscaling/esrgan_pytorch.py:512-549)

Same lifecycle (``__init__`` touches no GPU; ``await load(device)`` / ``unload()`` / ``infer(...)``; infer before load
raises), same ``_infer`` signatures, argument meaning and return types, errors as Python exceptions.  When the
reference package is importable the classes derive from its ``OfflineDetector`` / ``OfflineOCR`` /
``OfflineInpainter`` and ``register()`` adds them to ``DETECTORS`` / ``OCRS`` / ``INPAINTERS``;
otherwise they derive from a minimal mirror of ``ModelWrapper`` (utils/inference.py:330-350) so the contract can be
exercised stand-alone.  The detectors' box extraction (contours -> min-area boxes -> unclip) runs on the native host routines of the C-ABI library
(hostglue.py; the reference's OpenCV/pyclipper version can be injected instead); mask refinement and image resizing are
NOT part of the dense path: they are taken from the reference package when present, or injected by the caller.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import pipeline
from .textline import Quadrilateral

try:  # inside the reference's environment: be a real plugin
    from manga_translator.detection.common import OfflineDetector as _DetBase  # type: ignore
    from manga_translator.inpainting.common import OfflineInpainter as _InpBase  # type: ignore
    from manga_translator.ocr.common import OfflineOCR as _OcrBase  # type: ignore
    from manga_translator.upscaling.common import OfflineUpscaler as _UpBase  # type: ignore
    from manga_translator.utils import Quadrilateral as _RefQuadrilateral  # type: ignore

    HAVE_REFERENCE = True
except Exception:  # stand-alone: mirror the ModelWrapper lifecycle
    HAVE_REFERENCE = False
    _RefQuadrilateral = Quadrilateral

    class _Wrapper:
        """Mirror of ModelWrapper's load/unload/infer protocol (utils/inference.py:318-350)."""
        _key = "hip"

        _MODEL_SUB_DIR = ""
        _MODEL_DIR = os.environ.get("MIT_MODEL_DIR", "models")   # ModelWrapper._MODEL_DIR (utils/inference.py:94): BASE_PATH/models there

        def __init__(self, *args, **kwargs):
            self._loaded = False
            self._downloaded = self._check_downloaded()

        def is_loaded(self) -> bool:
            return self._loaded

        def is_downloaded(self) -> bool:
            return self._downloaded

        @property
        def model_dir(self) -> str:
            return os.path.join(self._MODEL_DIR, self._MODEL_SUB_DIR)

        def _get_file_path(self, *args) -> str:
            return os.path.join(self.model_dir, *args)

        def _check_downloaded(self) -> bool:
            """The files named by ``_MODEL_MAPPING`` exist under model_dir (utils/inference.py:262-293)."""
            for key, m in self._MODEL_MAPPING.items():
                names = list(m["archive"]) if "archive" in m else [m["file"] if m.get("file", ".") != "." else m["url"].rsplit("/", 1)[-1]]
                if not all(os.path.exists(self._get_file_path(n)) for n in names):
                    return False
            return True

        async def download(self, force: bool = False):
            """Stand-alone mode has no downloader (that is the reference's ModelWrapper.download): the checkpoints must already be
            in place, or the weights handed to the constructor."""
            if not self.is_downloaded():
                raise FileNotFoundError(f"{self._key}: checkpoint files {sorted(self._MODEL_MAPPING)} are not under {self.model_dir!r} "
                                        "and there is no downloader outside the reference package; pass weights= or copy the files")

        async def load(self, device: str, *args, **kwargs):
            if not self.is_downloaded():
                await self.download()
            if not self.is_loaded():
                await self._load(*args, **kwargs, device=device)
                self._loaded = True

        async def unload(self):
            if self.is_loaded():
                await self._unload()
                self._loaded = False

        async def reload(self, device: str, *args, **kwargs):
            await self.unload()
            await self.load(*args, **kwargs, device=device)

        async def infer(self, *args, **kwargs):
            if not self.is_loaded():
                raise Exception(f"{self._key}: Tried to forward pass without having loaded the model.")
            return await self._infer(*args, **kwargs)

    # the sub-directories the reference's OfflineDetector / OfflineOCR / OfflineInpainter / OfflineUpscaler keep their files in
    # (detection/common.py:138, ocr/common.py:54, inpainting/common.py:17, upscaling/common.py): a checkpoint tree laid out for the
    # reference is found by the stand-alone plugins as it is
    _DetBase = type("_DetBase", (_Wrapper,), {"_MODEL_SUB_DIR": "detection"})
    _OcrBase = type("_OcrBase", (_Wrapper,), {"_MODEL_SUB_DIR": "ocr"})
    _InpBase = type("_InpBase", (_Wrapper,), {"_MODEL_SUB_DIR": "inpainting"})
    _UpBase = type("_UpBase", (_Wrapper,), {"_MODEL_SUB_DIR": "upscaling"})


_RELEASE = "https://github.com/zyddnys/manga-image-translator/releases/download/beta-0.3/"


def set_model_dir(path: str) -> None:
    """Root directory of the checkpoint files for every plugin constructed afterwards — ``ModelWrapper._MODEL_DIR`` (utils/inference.py:94),
    of the reference's class when the plugins subclass it, of the stand-alone mirror otherwise.  ``<path>/<_MODEL_SUB_DIR>/<file>`` as
    the reference lays them out."""
    if HAVE_REFERENCE:
        from manga_translator.utils.inference import ModelWrapper as base  # type: ignore
    else:
        base = _Wrapper
    base._MODEL_DIR = str(path)


def _own_quad(q) -> Quadrilateral:
    """A text line in this package's geometry type: the object itself when it already is one (the detector plugins of this package return
    them; building it again would re-sort its corners with five numpy calls, 65 us a line), else built from the reference object's points."""
    return q if type(q) is Quadrilateral else Quadrilateral(np.asarray(q.pts))


def _weights_handed_over(plugin, *given) -> None:
    """State dicts injected through the constructor stand for the checkpoint files: nothing is left to download, so
    ModelWrapper.load() (utils/inference.py:330-338) must not try to fetch ``_MODEL_MAPPING`` (there is no network offline)."""
    if all(g is not None for g in given):
        plugin._downloaded = True


def _gpu_device(device: str) -> torch.device:
    """The reference passes 'cpu' | 'cuda' | 'mps' | 'xpu' (ROCm = 'cuda').  This backend has no CPU path."""
    if not str(device).startswith("cuda"):
        raise RuntimeError(f"the HIP backend runs on an MI355X only: device={device!r} (use --use-gpu)")
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the HIP backend has no CPU fallback")
    return torch.device(device)


class HipComicTextDetector(_DetBase):
    """``--detector ctd`` on the HIP engine."""
    _KEY = _key = "ctd_hip"
    # the torch checkpoint of the reference's own mapping (detection/ctd.py:63-74; its ONNX twin is the reference's CPU path)
    _MODEL_MAPPING: Dict = {
        "model-cuda": {
            "url": _RELEASE + "comictextdetector.pt",
            "hash": "1f90fa60aeeb1eb82e2ac1167a66bf139a8a61b8780acd351ead55268540cccb",
            "file": ".",
        },
    }

    def __init__(self, *args, weights: Optional[Dict[str, Dict[str, torch.Tensor]]] = None,
                 boxes_from_maps: Optional[Callable] = None, refine: Optional[Callable] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self._weights, self._boxes, self._refine = weights, boxes_from_maps, refine
        self.engine = None
        self.input_size = (1024, 1024)  # ctd.py:84: fixed, whatever detect_size the caller passes
        _weights_handed_over(self, weights)

    async def _load(self, device: str, input_size=1024, **_):
        from . import ctd

        dev = _gpu_device(device)
        w = self._weights or _load_ctd_checkpoint(self)
        self.engine = ctd.CtdEngine(w["ctd.yolo"], w["ctd.seg"], w["ctd.det"], device=dev)
        self.device, self.input_size = device, (input_size, input_size)

    async def _unload(self):
        if self.engine is not None:
            self.engine.release_workspace()  # device slabs go back to the allocator before the weights do
        self.engine = None

    @torch.no_grad()
    async def _infer(self, image: np.ndarray, detect_size: int, text_threshold: float, box_threshold: float,
                     unclip_ratio: float, verbose: bool = False):
        """-> (textlines, mask_refined u8 [H,W], None).  Like the reference, ignores detect_size / thresholds /
        unclip_ratio (fixed 1024, 0.3, 0.6, 1.5: ctd.py:84,102,157)."""
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError(f"expected uint8 RGB [H,W,3], got {image.dtype} {image.shape}")
        im_h, im_w = image.shape[:2]
        page = torch.from_numpy(np.ascontiguousarray(image)).to(self.engine.device)[None]
        from . import hostglue, imgproc, rearrange

        S = self.input_size[0]
        if rearrange.plan(im_h, im_w, S) is not None:    # webtoon strip: det_rearrange_forward (ctd.py:137, generic.py:876-997)
            lines_map, mask_f = rearrange.forward(image, self._tiles_forward, S)
            mask_u8 = torch.from_numpy((mask_f.squeeze() * 255).astype(np.uint8)).to(self.engine.device)[None]  # postprocess_mask (:155)
        else:
            mask_u8, lines, _ = self.engine.forward(page)    # postprocess_mask already applied on the GPU (ctd.py:30-44)
            lines_map = None                          # [1,2,h,w] on the device, cropped to the un-padded area (:152-153)
        if self._refine is None:                         # cv2.resize(mask, (w, h), INTER_LINEAR) (:162) on the GPU as well
            mask_full = imgproc.resize_u8(mask_u8[:1].contiguous(), (im_w, im_h))[0]
        # SegDetectorRepresenter(thresh=0.3) (:102,156): on the GPU where the map already is (csrc/ctd_boxes.hip: only the boxes cross
        # PCIe); an injected extractor, or a rearranged strip whose stitched map was assembled on the host, takes the numpy map
        if self._boxes is None and lines_map is None and lines.is_cuda and not os.environ.get("MIT_BOXES_HOST"):   # (MIT_BOXES_HOST=1: the host routine, A/B;
            # a map that lives on the host — an injected stand-in engine, tests/boundary_checks.py — goes to the host routine as well)
            boxes, scores = hostglue.ctd_boxes_gpu(lines, im_h, im_w)[0]
        else:
            boxes_fn = self._boxes or _native_ctd_boxes
            boxes, scores = boxes_fn(lines_map if lines_map is not None else lines.cpu().numpy(), im_h, im_w)
        keep = np.where(scores > 0.6)                        # box_thresh (:157-159)
        boxes, scores = boxes[keep], scores[keep]
        textlines = [_RefQuadrilateral(pts.astype(int), "", float(s)) for pts, s in zip(boxes, scores)]
        if self._refine is not None:                         # injected resize + refine_mask (e.g. the reference's OpenCV one)
            return textlines, self._refine(image, mask_u8[0].cpu().numpy(), textlines, im_h, im_w), None
        # refine_mask(image, mask, textlines, refine_mode=None) (:177): page and mask are already on the device (csrc/ctd_refine.hip)
        return textlines, hostglue.refine_mask_gpu(page[0], mask_full, textlines, None).cpu().numpy(), None


    def _tiles_forward(self, squares: np.ndarray):
        """det_batch_forward_ctd (ctd.py:106-127) for <= 4 rearranged squares: u8 [n,s,s,3] -> (lines [n,2,S,S], mask [n,1,S,S])
        float32.  Squares larger than the input size are shrunk on the GPU (square_pad_resize's INTER_LINEAR, generic.py:870-872);
        at exactly S x S the engine's letterbox is the identity, i.e. the reference's plain ``/ 255``."""
        from . import imgproc

        S = self.input_size[0]
        t = torch.from_numpy(np.ascontiguousarray(squares)).to(self.engine.device)
        if t.shape[1] != S:
            t = imgproc.resize_u8(t, (S, S))
        _, lines, _ = self.engine.forward(t)
        return lines.cpu().numpy(), self.engine.last_mask_f32.cpu().numpy()[:, None]


class HipDefaultDetector(_DetBase):
    """``--detector default`` (DBNet on ResNet-34) on the HIP engine."""
    _KEY = _key = "default_hip"
    _MODEL_MAPPING: Dict = {  # detection/default.py:28-34
        "model": {
            "url": _RELEASE + "detect-20241225.ckpt",
            "hash": "67ce1c4ed4793860f038c71189ba9630a7756f7683b1ee5afb69ca0687dc502e",
            "file": ".",
        },
    }

    def __init__(self, *args, weights: Optional[Dict[str, torch.Tensor]] = None, preprocess: Optional[Callable] = None,
                 boxes_from_maps: Optional[Callable] = None, resize2x: Optional[Callable] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self._weights, self._pre, self._boxes, self._resize2x = weights, preprocess, boxes_from_maps, resize2x
        self.engine = None
        _weights_handed_over(self, weights)

    async def _load(self, device: str):
        from . import dbnet

        dev = _gpu_device(device)
        sd = self._weights
        if sd is None:
            ck = torch.load(_ckpt_path(self, "detect-20241225.ckpt"), map_location="cpu")
            sd = ck["model"] if "model" in ck else ck
        self.engine = dbnet.DbnetEngine(sd, device=dev)
        self.device = device

    async def _unload(self):
        if self.engine is not None:
            self.engine.release_workspace()  # device slabs go back to the allocator before the weights do
        self.engine = None

    @torch.no_grad()
    async def _infer(self, image: np.ndarray, detect_size: int, text_threshold: float, box_threshold: float,
                     unclip_ratio: float, verbose: bool = False):
        """-> (textlines, raw_mask u8 [H,W], None) (default.py:56-103).  bilateralFilter + resize_aspect_ratio (:62) and the
        network run on the GPU; SegDetectorRepresenter (:73-77) is the native host extraction (csrc/hostglue.hip), the x2 mask
        resize (:89) a numpy bilinear; every step can still be injected (preprocess= / boxes_from_maps= / resize2x=)."""
        from . import imgproc, rearrange

        boxes_fn = self._boxes or _native_dbnet_boxes
        resize2x = self._resize2x or (lambda m: _resize2x_f32(m))
        if rearrange.plan(image.shape[0], image.shape[1], detect_size) is not None:  # webtoon strip (default.py:60, generic.py:876-997)
            def tiles(squares):  # det_batch_forward_default (:15-25): x / 127.5 - 1 happens inside the engine
                t = torch.from_numpy(np.ascontiguousarray(squares)).to(self.engine.device)
                if t.shape[1] != detect_size:
                    t = imgproc.resize_u8(t, (detect_size, detect_size))
                d, m = self.engine.forward(t)
                return d.cpu().numpy(), m.cpu().numpy()[:, None]

            db, mask4 = rearrange.forward(image, tiles, detect_size)
            mask = mask4[0, 0]
            h, w = image.shape[:2]
            ratio, pad_w, pad_h = 1.0, 0, 0
        else:
            if self._pre is not None:
                img_resized, target_ratio, pad_w, pad_h = self._pre(image, detect_size)
                page = torch.from_numpy(np.ascontiguousarray(img_resized)).to(self.engine.device)[None]
            else:  # cv2.bilateralFilter + resize_aspect_ratio (:62) on the device: the page crosses PCIe once, as bytes
                page, target_ratio, pad_w, pad_h = default_preprocess_gpu(
                    torch.from_numpy(np.ascontiguousarray(image)).to(self.engine.device), detect_size)
            ratio = 1 / target_ratio
            h, w = int(page.shape[1]), int(page.shape[2])
            db, mask = self.engine.forward(page)
            if self._boxes is None and db.is_cuda and not os.environ.get("MIT_BOXES_HOST"):   # SegDetectorRepresenter (:73-77) where the map is: csrc/ctd_boxes.hip
                from . import hostglue

                boxes_fn = lambda d, hh, ww, tt, bt, ur: hostglue.dbnet_boxes_gpu(d, hh, ww, tt, bt, ur)[0]   # noqa: E731
            else:
                db = db.cpu().numpy()
            mask = mask[0].cpu().numpy()
        boxes, scores = boxes_fn(db, h, w, text_threshold, box_threshold, unclip_ratio)
        if boxes.size == 0:
            polys, scores = [], []
        else:
            idx = boxes.reshape(boxes.shape[0], -1).sum(axis=1) > 0
            polys = (boxes[idx].astype(np.float64) * ratio).astype(np.int64)  # adjustResultCoordinates with ratio_net = 1 (:83)
        textlines = [_RefQuadrilateral(pts.astype(int), "", float(s)) for pts, s in zip(polys, scores)]
        textlines = [q for q in textlines if q.area > 16]
        mask_resized = resize2x(mask)
        if pad_h > 0:
            mask_resized = mask_resized[:-pad_h, :]
        elif pad_w > 0:
            mask_resized = mask_resized[:, :-pad_w]
        return textlines, np.clip(mask_resized * 255, 0, 255).astype(np.uint8), None


class HipModel48pxOCR(_OcrBase):
    """``--ocr 48px`` on the HIP engine."""
    _KEY = _key = "48px_hip"
    _MODEL_MAPPING: Dict = {  # ocr/model_48px.py:28-37
        "model": {
            "url": _RELEASE + "ocr_ar_48px.ckpt",
            "hash": "29daa46d080818bb4ab239a518a88338cbccff8f901bef8c9db191a7cb97671d",
        },
        "dict": {
            "url": _RELEASE + "alphabet-all-v7.txt",
            "hash": "f5722368146aa0fbcc9f4726866e4efc3203318ebb66c811d8cbbe915576538a",
        },
    }

    def __init__(self, *args, weights: Optional[Dict[str, torch.Tensor]] = None, dictionary: Optional[Sequence[str]] = None,
                 **kwargs):
        super().__init__(*args, **kwargs)
        self._weights, self.dictionary = weights, dictionary
        self.engine = None
        _weights_handed_over(self, weights, dictionary)

    async def _load(self, device: str):
        from . import ocr48

        dev = _gpu_device(device)
        if self._weights is None or self.dictionary is None:
            self._weights, self.dictionary = _load_ocr_checkpoint(self)
        self.engine = ocr48.Ocr48Engine(self._weights, len(self.dictionary), device=dev)
        self.device = device

    async def _unload(self):
        if self.engine is not None:
            self.engine.release_workspace()  # device slabs go back to the allocator before the weights do
        self.engine = None

    def _directions(self, textlines):
        """(line, direction) in processing order: the merge-graph majority vote of ocr/common.py:12-39 (the reference's own
        method when the package is present, textline.generate_text_direction otherwise)."""
        if HAVE_REFERENCE:
            return list(self._generate_text_direction(textlines))
        from . import textline as TL

        own = [_own_quad(q) for q in textlines]
        back = {id(o): q for o, q in zip(own, textlines)}
        return [(back[id(o)], d) for o, d in TL.generate_text_direction(own)]

    @torch.no_grad()
    async def _infer(self, image: np.ndarray, textlines: List, config=None, verbose: bool = False, ignore_bubble: int = 0,
                     max_seq_length: int = 255, suppress_eos: bool = False):
        """Sets text / prob / fg_* / bg_* on the same Quadrilateral objects and returns those above the threshold,
        in the reference's sorted-by-width chunk order (model_48px.py:67-180)."""
        threshold = 0.2 if config is None or getattr(config, "prob", None) is None else config.prob
        pairs = self._directions(textlines)
        if not pairs:
            return []
        quads = [q for q, _ in pairs]
        dirs = [[d for _, d in pairs]]
        page = torch.from_numpy(np.ascontiguousarray(image)).to(self.engine.device)[None]
        own = [_own_quad(q) for q in quads]  # geometry in this package's type
        r = self.engine.recognize_pages(page, [own], max_seq_length=max_seq_length, suppress_eos=suppress_eos, directions=dirs)
        toks, lens = r["tokens"].cpu().numpy(), r["length"].cpu().numpy()
        probs, cols = r["prob"].cpu().numpy(), r["colors"].cpu().numpy()
        out = []
        decoded = decode_lines(toks, lens, cols, self.dictionary, rows=[row for row in range(len(r["order"])) if probs[row] >= threshold])
        for row, (_, i) in enumerate(r["order"]):
            q, prob = quads[i], float(probs[row])
            q.assigned_direction = dirs[0][i]
            if prob < threshold:
                continue
            txt, fgc, bgc = decoded[row]            # (decode_line of the tokens after the start symbol)
            q.text, q.prob = txt, prob
            q.fg_r, q.fg_g, q.fg_b = fgc
            q.bg_r, q.bg_g, q.bg_b = bgc
            out.append(q)
        return out


class HipModel48pxCTCOCR(HipModel48pxOCR):
    """``--ocr 48px_ctc`` on the HIP engine (ocr/model_48px_ctc.py:62-160)."""
    _KEY = _key = "48px_ctc_hip"
    _MODEL_MAPPING: Dict = {  # ocr/model_48px_ctc.py:19-28
        "model": {
            "url": _RELEASE + "ocr-ctc.zip",
            "hash": "fc61c52f7a811bc72c54f6be85df814c6b60f63585175db27cb94a08e0c30101",
            "archive": {
                "ocr-ctc.ckpt": ".",
                "alphabet-all-v5.txt": ".",
            },
        },
    }

    async def _load(self, device: str):
        from . import ocr_ctc

        dev = _gpu_device(device)
        if self._weights is None or self.dictionary is None:
            self._weights, self.dictionary = _load_ocr_ctc_checkpoint(self)
        self.engine = ocr_ctc.OcrCtcEngine(self._weights, len(self.dictionary), device=dev)
        self.device = device

    def _rectify(self, page: torch.Tensor, quads, dirs, idx, records: np.ndarray, wp: int) -> torch.Tensor:
        """Lines ``idx`` of the page rectified into one zero-padded chunk tensor u8 [n, 48, wp, 3] on the device (mit_ocr_warp_lines:
        get_transformed_region + the chunk packing of model_48px_ctc.py:83-91).  ``quads`` / ``dirs`` are not needed by the kernel — the
        records carry the geometry — they are there for host stand-ins in tests."""
        import ctypes as C

        from . import lib as _lib, ops

        dev = page.device
        records["out_row"] = np.arange(len(idx))
        lines_dev = torch.frombuffer(bytearray(records.tobytes()), dtype=torch.uint8).to(dev)
        region = torch.empty(len(idx), 48, wp, 3, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().mit_ocr_warp_lines(page.data_ptr(), page.shape[1], page.shape[2], lines_dev.data_ptr(), len(idx), region.data_ptr(), 48, wp,
                                                  C.c_void_p(ops.current_stream())), "mit_ocr_warp_lines")
        return region

    @torch.no_grad()
    async def _infer(self, image: np.ndarray, textlines: List, config=None, verbose: bool = False):
        """Same contract as the 48px plugin; chunks are padded to max_w + 7 + 128 (:84), the line probability is
        exp(mean log-prob) against a 0.5 default threshold (:66,:124), colours average over non-space characters (:116-123).
        ``config.ignore_bubble`` in 1..50 applies the reference's frame / colour heuristic to every rectified crop (:91-93, utils/bubble.py):
        a rejected line's row of the chunk stays zero and is recognised as such, exactly as the reference's ``continue`` leaves it."""
        from . import textline as TL

        threshold = 0.5 if config is None or getattr(config, "prob", None) is None else config.prob
        ignore_bubble = int(getattr(config, "ignore_bubble", 0) or 0) if config is not None else 0
        pairs = self._directions(textlines)
        if not pairs:
            return []
        quads = [q for q, _ in pairs]
        dirs = [d for _, d in pairs]
        H, W = image.shape[:2]
        dev = self.engine.device
        page = torch.from_numpy(np.ascontiguousarray(image)).to(dev)[None]
        own = [_own_quad(q) for q in quads]
        rec = TL.warp_plans(own, dirs, H, W, 48)
        widths = np.where(rec["vertical"] != 0, rec["dh"], rec["dw"]).tolist()
        out = []
        for idx, ws, wp in TL.chunk_plan(widths):
            wp += 128
            region = self._rectify(page, own, dirs, idx, rec[idx].copy(), wp)
            if 1 <= ignore_bubble <= 50:
                host = region.cpu().numpy()
                for j, w_line in enumerate(ws):
                    if TL.is_ignore(host[j, :, :w_line], ignore_bubble):
                        region[j] = 0
            logits, colors = self.engine.forward(region)
            for j, line in enumerate(self.engine.decode(logits, colors, 0)):
                q = quads[idx[j]]
                q.assigned_direction = dirs[idx[j]]
                res = decode_ctc_line(line, self.dictionary)
                if res is None or res[1] < threshold:
                    continue
                q.text, q.prob = res[0], res[1]
                q.fg_r, q.fg_g, q.fg_b = res[2]
                q.bg_r, q.bg_g, q.bg_b = res[3]
                out.append(q)
        return out


def decode_ctc_line(line, dictionary: Sequence[str]):
    """[(char id, log-prob, fr, fg, fb, br, bg, bb)] -> (text, prob, fg rgb, bg rgb) or None for an empty line:
    model_48px_ctc.py:105-134 (AvgMeter means; colours only over non-space characters; prob = exp(mean log-prob))."""
    if not line:
        return None
    chars, lp_sum = [], 0.0
    acc = [0] * 6
    n_col = 0
    for chid, logprob, *cols in line:
        ch = dictionary[int(chid)]
        if ch == "<SP>":
            ch = " "
        chars.append(ch)
        lp_sum += logprob
        if ch != " ":
            for k in range(6):
                acc[k] += int(cols[k] * 255)
            n_col += 1
    prob = float(np.exp(lp_sum / len(line)))
    mean = [int(a / n_col) if n_col else 0 for a in acc]
    return "".join(chars), prob, tuple(mean[:3]), tuple(mean[3:])


def decode_line(token_ids: np.ndarray, colors: np.ndarray, dictionary: Sequence[str]) -> Tuple[str, Tuple[int, int, int], Tuple[int, int, int]]:
    """Token ids + colour-head rows -> (text, fg rgb, bg rgb): model_48px.py:124-158 (AvgMeter means of int(c*255))."""
    has_fg = colors[:, 7] > colors[:, 6]
    has_bg = colors[:, 9] > colors[:, 8]
    seq: List[str] = []
    acc = [[0, 0] for _ in range(6)]  # sum, count for fr fg fb br bg bb

    def add(k, v):
        acc[k][0] += int(v * 255)
        acc[k][1] += 1

    for t, chid in enumerate(token_ids):
        ch = dictionary[int(chid)]
        if ch == "<S>":
            continue
        if ch == "</S>":
            break
        seq.append(" " if ch == "<SP>" else ch)
        if has_fg[t]:
            for k in range(3):
                add(k, colors[t, k])
        src = colors[t, 3:6] if has_bg[t] else colors[t, 0:3]
        for k in range(3):
            add(3 + k, src[k])
    mean = [min(max(int(s / c) if c else 0, 0), 255) for s, c in acc]
    return "".join(seq), tuple(mean[:3]), tuple(mean[3:])


def decode_lines(tokens: np.ndarray, lengths: np.ndarray, colors: np.ndarray, dictionary: Sequence[str], rows=None
                 ) -> List[Tuple[str, Tuple[int, int, int], Tuple[int, int, int]]]:
    """``decode_line`` for all result rows of a decode at once (tokens [n, T + 1] with the start symbol in column 0, lengths [n],
    colours [n, T, 10]): the same integer arithmetic, vectorised over lines and positions — a page group's 512 lines cost one pass of
    numpy instead of half a millisecond of interpreter each.  ``rows``: only these rows (others give None)."""
    tokens, lengths, colors = np.asarray(tokens), np.asarray(lengths), np.asarray(colors, dtype=np.float32)
    n, T = tokens.shape[0], colors.shape[1]
    if n == 0:
        return []
    ids = tokens[:, 1:1 + T].astype(np.int64)
    pos = np.arange(ids.shape[1])[None, :]
    inside = pos < (lengths.astype(np.int64)[:, None] - 1)
    s_id, e_id = dictionary.index("<S>"), dictionary.index("</S>")
    is_end = inside & (ids == e_id)
    cut = np.where(is_end.any(1), is_end.argmax(1), ids.shape[1])          # the first </S> ends the line
    keep = inside & (pos < cut[:, None]) & (ids != s_id)                   # <S> is skipped, not counted
    c = colors[:, :ids.shape[1]]
    q = (c[..., :6] * np.float32(255)).astype(np.int64)                    # int(v * 255) on float32 values: truncation
    has_fg = keep & (c[..., 7] > c[..., 6])
    has_bg = c[..., 9] > c[..., 8]
    fg_sum = (q[..., 0:3] * has_fg[..., None]).sum(1)
    fg_cnt = has_fg.sum(1)
    bsrc = np.where(has_bg[..., None], q[..., 3:6], q[..., 0:3])
    bg_sum = (bsrc * keep[..., None]).sum(1)
    bg_cnt = keep.sum(1)

    def mean(sm, cnt):   # int(sum / count) clamped to a byte; 0 without samples
        m = np.where(cnt[:, None] > 0, np.trunc(sm / np.maximum(cnt, 1)[:, None]), 0).astype(np.int64)
        return np.clip(m, 0, 255)

    fg, bg = mean(fg_sum, fg_cnt), mean(bg_sum, bg_cnt)
    chars = np.asarray([" " if ch == "<SP>" else ch for ch in dictionary], dtype=object)
    out: List = [None] * n
    for r in (range(n) if rows is None else rows):
        out[r] = ("".join(chars[ids[r, keep[r]]]), tuple(int(v) for v in fg[r]), tuple(int(v) for v in bg[r]))
    return out


class HipLamaMPEInpainter(_InpBase):
    """``--inpainter lama_mpe`` on the HIP engine."""
    _KEY = _key = "lama_mpe_hip"
    _MODEL_MAPPING: Dict = {  # inpainting/inpainting_lama_mpe.py:32-38
        "model": {
            "url": _RELEASE + "inpainting_lama_mpe.ckpt",
            "hash": "d625aa1b3e0d0408acfd6928aa84f005867aa8dbb9162480346a4e20660786cc",
            "file": ".",
        },
    }
    N_BLOCKS, USE_MPE, CKPT = 9, True, "inpainting_lama_mpe.ckpt"

    def __init__(self, *args, weights: Optional[Dict[str, Dict[str, torch.Tensor]]] = None,
                 resize: Optional[Callable] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self._weights, self._resize = weights, resize
        self.engine = None
        _weights_handed_over(self, weights)

    async def _load(self, device: str):
        from . import lama

        dev = _gpu_device(device)
        w = self._weights or _load_lama_checkpoint(self)
        self.engine = lama.LamaEngine(w["lama.gen"], w.get("lama.mpe") if self.USE_MPE else None, n_blocks=self.N_BLOCKS,
                                      device=dev)
        self.device = device

    async def _unload(self):
        if self.engine is not None:
            self.engine.release_workspace()  # device slabs go back to the allocator before the weights do
        self.engine = None

    @torch.no_grad()
    async def _infer(self, image: np.ndarray, mask: np.ndarray, config=None, inpainting_size: int = 1024,
                     verbose: bool = False) -> np.ndarray:
        """image u8 [H,W,3], mask u8 [H,W] -> inpainted [H,W,3] (inpainting_lama_mpe.py:56-118), any page size: the
        resize_keep_aspect / multiple-of-8 / back-to-page resizes and the final composite run on the GPU (imgproc.py).  Always
        fp32: the reference's CPU path never autocasts (:93-95) and that is the parity target."""
        if image.ndim != 3 or image.shape[2] != 3 or mask.shape != image.shape[:2]:
            raise ValueError(f"bad shapes: image {image.shape}, mask {mask.shape}")
        if image.dtype != np.uint8 or mask.dtype != np.uint8:
            raise ValueError(f"expected uint8 page and mask, got {image.dtype} / {mask.dtype}")
        from . import imgproc

        height, width = image.shape[:2]
        dev = self.engine.device
        img0 = torch.from_numpy(np.ascontiguousarray(image)).to(dev)[None]     # the page crosses PCIe once, as bytes
        msk0 = torch.from_numpy(np.ascontiguousarray(mask)).to(dev)[None]
        img, msk = img0, msk0
        rs = self._resize  # optional injected callable (img, (w, h), "keep_aspect" | "linear") -> ndarray: e.g. the real OpenCV
        if max(height, width) > inpainting_size:                                # resize_keep_aspect = INTER_LINEAR_EXACT (:64-66)
            dsize = imgproc.keep_aspect_size(height, width, inpainting_size)
            img, msk = self._resized(img, dsize, "keep_aspect", rs), self._resized(msk, dsize, "keep_aspect", rs)
        h, w = img.shape[1:3]
        new_h, new_w = (h + 7) // 8 * 8, (w + 7) // 8 * 8                      # pad_size 8, by RESIZING (INTER_LINEAR, :67-79)
        if (new_h, new_w) != (h, w):
            img, msk = self._resized(img, (new_w, new_h), "linear", rs), self._resized(msk, (new_w, new_h), "linear", rs)
        resized = (new_h, new_w) != (height, width)
        out = self.engine.forward(img, msk, composite=not resized)  # resized: img_inpainted of :111, every pixel from the network
        if resized:                                                             # back to the page size (:112-113)
            out = self._resized(out, (width, height), "linear", rs)
        # img_inpainted * mask_original + img_original * (1 - mask_original), mask_original = mask >= 127 (:57-61,116)
        return imgproc.select_u8(msk0, 127, out, img0)[0].cpu().numpy()

    @staticmethod
    def _resized(t: torch.Tensor, dsize, mode: str, injected: Optional[Callable]) -> torch.Tensor:
        """[1,H,W(,C)] u8 device tensor -> (w, h) = dsize: ``mit_resize_u8`` on the GPU, or the injected host callable."""
        from . import imgproc

        if injected is None:
            return imgproc.resize_u8(t, dsize, exact=(mode == "keep_aspect"))
        return torch.from_numpy(np.ascontiguousarray(injected(t[0].cpu().numpy(), dsize, mode))).to(t.device)[None]


class HipLamaLargeInpainter(HipLamaMPEInpainter):
    """``--inpainter lama_large``: 18 blocks, no MPE (inpainting_lama_mpe.py:121-136)."""
    _KEY = _key = "lama_large_hip"
    _MODEL_MAPPING: Dict = {  # inpainting/inpainting_lama_mpe.py:123-129
        "model": {
            "url": "https://huggingface.co/dreMaz/AnimeMangaInpainting/resolve/main/lama_large_512px.ckpt",
            "hash": "11d30fbb3000fb2eceae318b75d9ced9229d99ae990a7f8b3ac35c8d31f2c935",
            "file": ".",
        },
    }
    N_BLOCKS, USE_MPE, CKPT = 18, False, "lama_large_512px.ckpt"


class HipESRGANUpscaler(_UpBase):
    """``--upscaler 4xultrasharp`` (RRDBNet 4x) on the HIP engine."""
    _KEY = _key = "4xultrasharp_hip"
    _MODEL_MAPPING: Dict = {  # upscaling/esrgan_pytorch.py:513-518
        "4x-UltraSharp": {
            "url": _RELEASE + "4xESRGAN.pth",
            "hash": "545805ce2d861ee90972b5fa50b851f19ee4bb35dedd2eb090be1f7c935b6b00",
        },
    }
    _VALID_UPSCALE_RATIOS = [2, 3, 4]

    def __init__(self, *args, weights: Optional[Dict[str, torch.Tensor]] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self._weights = weights
        self.engine = None
        _weights_handed_over(self, weights)

    async def _load(self, device: str):
        from . import esrgan

        dev = _gpu_device(device)
        sd = self._weights or _load_esrgan_checkpoint(self)
        nb = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.1.sub.") and ".RDB" in k)  # infer_params (:470-509)
        self.engine = esrgan.EsrganEngine(sd, nb=nb, device=dev)
        self.device = device

    async def _unload(self):
        if self.engine is not None:
            self.engine.release_workspace()  # device slabs go back to the allocator before the weights do
        self.engine = None

    @torch.no_grad()
    async def _infer(self, image_batch: List, upscale_ratio: float) -> List:
        """List[PIL.Image] -> List[PIL.Image]: 4x on the GPU, then PIL's bilinear resize by ratio/4 (esrgan_pytorch.py:537-549)."""
        from PIL import Image

        assert upscale_ratio <= 4
        ratio = upscale_ratio / 4
        out = []
        for img in image_batch:  # pages of a batch may differ in size: one launch sequence per page
            rgb = np.ascontiguousarray(np.array(img.convert("RGB")))
            up = self.engine.forward(torch.from_numpy(rgb).to(self.engine.device)[None])[0].cpu().numpy()
            im = Image.fromarray(up)
            out.append(im.resize(size=(int(round(im.size[0] * ratio)), int(round(im.size[1] * ratio))), resample=Image.Resampling.BILINEAR))
        return out


# ---- pieces taken from the reference package when it is importable ---------------------------------------------

def _native_ctd_boxes(lines_map, im_h, im_w):
    """SegDetectorRepresenter(thresh=0.3)(None, lines_map, height, width) on the native host routines (hostglue.py)."""
    from . import hostglue

    return hostglue.ctd_boxes(lines_map, im_h, im_w)


def _native_refine(image, mask, textlines, im_h, im_w):
    """cv2.resize(mask, (w, h), INTER_LINEAR) + refine_mask(image, mask, textlines, refine_mode=None) (ctd.py:162,177)."""
    from . import hostglue

    return hostglue.refine_mask(image, hostglue.resize_linear_u8(mask, (im_w, im_h)), textlines, None)


def _native_dbnet_boxes(db, h, w, text_threshold, box_threshold, unclip_ratio):
    from . import hostglue

    return hostglue.dbnet_boxes(db, h, w, text_threshold, box_threshold, unclip_ratio)


def _reference_boxes():
    if not HAVE_REFERENCE:
        raise RuntimeError("box extraction needs the reference's SegDetectorRepresenter (OpenCV/pyclipper); pass boxes_from_maps=")
    from manga_translator.detection.ctd_utils.utils.db_utils import SegDetectorRepresenter  # type: ignore

    rep = SegDetectorRepresenter(thresh=0.3)

    def fn(lines_map, im_h, im_w):
        lines, scores = rep(None, lines_map, height=im_h, width=im_w)
        return lines[0], scores[0]

    return fn


def _reference_refine():
    if not HAVE_REFERENCE:
        raise RuntimeError("mask refinement needs the reference's refine_mask (OpenCV); pass refine=")
    import cv2  # type: ignore
    from manga_translator.detection.ctd_utils.textmask import refine_mask  # type: ignore

    def fn(image, mask, textlines, im_h, im_w):
        mask = cv2.resize(mask, (im_w, im_h), interpolation=cv2.INTER_LINEAR)
        return refine_mask(image, mask, textlines, refine_mode=None)

    return fn


def default_preprocess_gpu(image: torch.Tensor, detect_size: int):
    """``imgproc.resize_aspect_ratio(cv2.bilateralFilter(image, 17, 80, 80), detect_size, cv2.INTER_LINEAR, mag_ratio=1)``
    (detection/default.py:62, default_utils/imgproc.py:37-70) on the device: bilateral filter (mit_bilateral_u8c3), 8-bit
    INTER_LINEAR resize of the long side to ``detect_size`` (mit_resize_u8), zero canvas padded right / bottom to a multiple of 256.
    image u8 [H,W,3] (device) -> (page u8 [1,H',W',3], ratio, pad_w, pad_h)."""
    from . import imgproc

    height, width = int(image.shape[0]), int(image.shape[1])
    ratio = detect_size / max(height, width)
    target_h, target_w = int(round(height * ratio)), int(round(width * ratio))
    proc = imgproc.resize_u8(imgproc.bilateral_filter_u8(image, 17, 80.0, 80.0)[None], (target_w, target_h))
    pad_h = (256 - target_h % 256) % 256
    pad_w = (256 - target_w % 256) % 256
    if pad_h or pad_w:
        canvas = torch.zeros(1, target_h + pad_h, target_w + pad_w, 3, dtype=torch.uint8, device=image.device)
        canvas[:, :target_h, :target_w] = proc
        proc = canvas
    return proc, ratio, pad_w, pad_h


def _reference_default_preprocess():
    if not HAVE_REFERENCE:
        raise RuntimeError("the default detector's bilateral filter + resize need OpenCV; pass preprocess=")
    import cv2  # type: ignore
    from manga_translator.detection.default_utils import imgproc  # type: ignore

    def fn(image, detect_size):
        img, ratio, _, pad_w, pad_h = imgproc.resize_aspect_ratio(cv2.bilateralFilter(image, 17, 80, 80), detect_size,
                                                                  cv2.INTER_LINEAR, mag_ratio=1)
        return img, ratio, pad_w, pad_h

    return fn


def _reference_default_boxes():
    if not HAVE_REFERENCE:
        raise RuntimeError("box extraction needs the reference's dbnet_utils.SegDetectorRepresenter; pass boxes_from_maps=")
    from manga_translator.detection.default_utils import dbnet_utils  # type: ignore

    def fn(db, h, w, text_threshold, box_threshold, unclip_ratio):
        det = dbnet_utils.SegDetectorRepresenter(text_threshold, box_threshold, unclip_ratio=unclip_ratio)
        boxes, scores = det({"shape": [(h, w)]}, db)
        return boxes[0], scores[0]

    return fn


def _resize2x_f32(m: np.ndarray) -> np.ndarray:
    """cv2.resize(mask, (2w, 2h), INTER_LINEAR) on a float32 map (default.py:89): plain bilinear at pixel centres with edge
    replication — float data takes OpenCV's unquantised path (coefficients 1 - f, f in float32)."""
    h, w = m.shape

    def taps(n):
        f = ((np.arange(2 * n, dtype=np.float64) + 0.5) * 0.5 - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        fr = (f - i0.astype(np.float32)).astype(np.float32)
        lo, hi = i0 < 0, i0 >= n - 1
        i0 = np.where(lo, 0, np.where(hi, n - 1, i0))
        fr = np.where(lo | hi, np.float32(0), fr).astype(np.float32)
        return i0, np.minimum(i0 + 1, n - 1), (np.float32(1) - fr).astype(np.float32), fr

    y0, y1, wy0, wy1 = taps(h)
    x0, x1, wx0, wx1 = taps(w)
    m = m.astype(np.float32)
    rows = m[:, x0] * wx0[None, :] + m[:, x1] * wx1[None, :]
    return (rows[y0] * wy0[:, None] + rows[y1] * wy1[:, None]).astype(np.float32)


def _reference_resize2x():
    if not HAVE_REFERENCE:
        raise RuntimeError("the x2 mask resize needs OpenCV; pass resize2x=")
    import cv2  # type: ignore

    return lambda m: cv2.resize(m, (m.shape[1] * 2, m.shape[0] * 2), interpolation=cv2.INTER_LINEAR)


def _reference_resize():
    if not HAVE_REFERENCE:
        raise RuntimeError("pages that need resizing (max side > inpainting_size or not a multiple of 8) need OpenCV; pass resize=")
    import cv2  # type: ignore
    from manga_translator.utils import resize_keep_aspect  # type: ignore

    def fn(img, dsize, mode):
        if mode == "keep_aspect":
            return resize_keep_aspect(img, max(dsize))
        return cv2.resize(img, dsize, interpolation=cv2.INTER_LINEAR)

    return fn


def _ckpt_path(plugin, name: str) -> str:
    if hasattr(plugin, "_get_file_path"):
        return plugin._get_file_path(name)
    return os.path.join("models", name)


def _load_ctd_checkpoint(plugin):
    """comictextdetector.pt = {'blk_det': {'cfg','weights'}, 'text_seg', 'text_det'} (ctd_utils/basemodel.py:205-214)."""
    from . import ctd_schema as S, synth

    ck = torch.load(_ckpt_path(plugin, "comictextdetector.pt"), map_location="cpu")
    S.check_yolo_cfg(ck["blk_det"]["cfg"])  # the reference builds the backbone from it (yolov5/yolo.py:286-292)
    return {"ctd.yolo": synth.check_state_dict(ck["blk_det"]["weights"], S.yolo_schema(), "comictextdetector.pt blk_det.weights"),
            "ctd.seg": synth.check_state_dict(ck["text_seg"], S.unet_head_schema(), "comictextdetector.pt text_seg"),
            "ctd.det": synth.check_state_dict(ck["text_det"], S.db_head_schema(), "comictextdetector.pt text_det")}


def _read_dictionary(path: str):
    with open(path, "r", encoding="utf-8") as fp:
        return [s[:-1] for s in fp.readlines()]  # model_48px.py:47-48: every line loses its last character (the newline)


def _load_ocr_checkpoint(plugin):
    """ocr_ar_48px.ckpt (a bare state_dict) + alphabet-all-v7.txt (model_48px.py:46-52)."""
    from . import ocr_schema, synth

    dictionary = _read_dictionary(_ckpt_path(plugin, "alphabet-all-v7.txt"))
    sd = torch.load(_ckpt_path(plugin, "ocr_ar_48px.ckpt"), map_location="cpu")
    return synth.check_state_dict(sd, ocr_schema.ocr48_schema(len(dictionary)), "ocr_ar_48px.ckpt"), dictionary


def _load_ocr_ctc_checkpoint(plugin):
    """ocr-ctc.ckpt ({'model': state_dict} or bare; its three ``encoders.layers.N.pe.pe`` tables are dropped by the reference and
    unused here) + alphabet-all-v5.txt (model_48px_ctc.py:19-28,38-48)."""
    from . import ocr_ctc_schema, synth

    dictionary = _read_dictionary(_ckpt_path(plugin, "alphabet-all-v5.txt"))
    sd = torch.load(_ckpt_path(plugin, "ocr-ctc.ckpt"), map_location="cpu")
    sd = sd["model"] if "model" in sd else sd
    return synth.check_state_dict(sd, ocr_ctc_schema.ocr_ctc_schema(len(dictionary)), "ocr-ctc.ckpt"), dictionary


def _load_lama_checkpoint(plugin):
    """{'gen_state_dict', 'str_state_dict'?} (inpainting_lama_mpe.py:818-825)."""
    from . import lama_schema, synth

    ck = torch.load(_ckpt_path(plugin, plugin.CKPT), map_location="cpu")
    out = {"lama.gen": synth.check_state_dict(ck["gen_state_dict"], lama_schema.lama_generator_schema(plugin.N_BLOCKS), plugin.CKPT)}
    if "str_state_dict" in ck:
        out["lama.mpe"] = ck["str_state_dict"]
        if plugin.USE_MPE:
            synth.check_state_dict(out["lama.mpe"], lama_schema.lama_mpe_schema(), plugin.CKPT + " str_state_dict")
    return out


def _load_esrgan_checkpoint(plugin):
    """4xESRGAN.pth: a bare RRDBNet state_dict; the block count comes from its keys (esrgan_pytorch.py:526-528, infer_params :476-510)."""
    from . import esrgan_schema, synth

    sd = torch.load(_ckpt_path(plugin, "4xESRGAN.pth"), map_location="cpu")
    nb = 1 + max((int(k.split(".")[3]) for k in sd if k.startswith("model.1.sub.") and ".RDB" in k), default=-1)
    if nb <= 0:
        raise ValueError("4xESRGAN.pth: no RRDB trunk (model.1.sub.<n>.…) in the state_dict")
    return synth.check_state_dict(sd, esrgan_schema.rrdbnet_schema(nb), "4xESRGAN.pth")


def register() -> None:
    """Add the HIP backends to the reference's registries (needs the reference package; INTEGRATION.md shows the
    matching enum members)."""
    if not HAVE_REFERENCE:
        raise RuntimeError("manga_translator is not importable: nothing to register into")
    from manga_translator.detection import DETECTORS  # type: ignore
    from manga_translator.inpainting import INPAINTERS  # type: ignore
    from manga_translator.ocr import OCRS  # type: ignore

    from manga_translator.upscaling import UPSCALERS  # type: ignore
    from manga_translator import config as _cfg  # type: ignore

    def key(enum_name: str, value: str):
        """The enum member when the maintainer has added it to config.py, else the plain string: the
        registries are ordinary dicts and ``get_detector`` & co only look the key up (detection/__init__.py:22-28)."""
        enum = getattr(_cfg, enum_name, None)
        try:
            return enum(value)
        except Exception:
            return value

    for reg, enum_name, cls in ((DETECTORS, "Detector", HipComicTextDetector), (DETECTORS, "Detector", HipDefaultDetector),
                                (OCRS, "Ocr", HipModel48pxOCR), (OCRS, "Ocr", HipModel48pxCTCOCR),
                                (INPAINTERS, "Inpainter", HipLamaMPEInpainter), (INPAINTERS, "Inpainter", HipLamaLargeInpainter),
                                (UPSCALERS, "Upscaler", HipESRGANUpscaler)):
        reg[key(enum_name, cls._KEY)] = cls
