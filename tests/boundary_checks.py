"""Run in a FRESH interpreter by tests/test_reference_boundary.py: the plugin contract against the reference's REAL base classes.

oracle.ref_boundary.install() makes ``manga_translator.{utils,config,detection,ocr,inpainting,upscaling}`` importable from
/root/reference, so manga_image_translator_amd.plugins takes its HAVE_REFERENCE branch: every plugin here derives from the
reference's own OfflineDetector / OfflineOCR / OfflineInpainter / OfflineUpscaler (utils/inference.py ModelWrapper) and is driven
through the reference's own callers (CommonDetector.detect, CommonOCR.recognize, CommonInpainter.inpaint, get_detector ...).
No GPU: where a call would reach the dense engine, a stand-in engine returns fixed tensors — what is under test is the boundary
(types, lifecycle, registry, checkpoint lookup, download flow, error propagation), not the kernels.  Six checks go further and
run the reference's REAL ``_infer`` of the ctd detector, the default detector, both OCRs, the LaMa inpainter and the ESRGAN
upscaler with a stubbed network beside
the plugin's ``_infer`` with the same stub: everything either side does around the network must produce identical bytes.
"""
import asyncio
import hashlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np
import torch

from oracle import ref_boundary as RB

model_dir = tempfile.mkdtemp(prefix="mit_models_")
RB.install(model_dir=model_dir)

import manga_translator.detection as RD  # noqa: E402
import manga_translator.inpainting as RI  # noqa: E402
import manga_translator.ocr as RO  # noqa: E402
import manga_translator.upscaling as RU  # noqa: E402
import manga_translator.utils as U  # noqa: E402
from manga_translator.utils import inference as INF  # noqa: E402

from manga_image_translator_amd import plugins as P  # noqa: E402

run = asyncio.new_event_loop().run_until_complete
done = []


def check(name):
    def deco(fn):
        fn()
        done.append(name)
        print("ok", name, flush=True)
        return fn
    return deco


PLUGINS = [(P.HipComicTextDetector, RD.OfflineDetector, "detection"), (P.HipDefaultDetector, RD.OfflineDetector, "detection"),
           (P.HipModel48pxOCR, RO.OfflineOCR, "ocr"), (P.HipModel48pxCTCOCR, RO.OfflineOCR, "ocr"),
           (P.HipLamaMPEInpainter, RI.OfflineInpainter, "inpainting"), (P.HipLamaLargeInpainter, RI.OfflineInpainter, "inpainting"),
           (P.HipESRGANUpscaler, RU.OfflineUpscaler, "upscaling")]


@check("subclassing + construction")
def _():
    assert P.HAVE_REFERENCE and P._RefQuadrilateral is U.Quadrilateral
    for cls, base, sub in PLUGINS:
        assert issubclass(cls, base) and issubclass(cls, INF.ModelWrapper) and issubclass(cls, INF.InfererModule)
        p = cls()                                         # no arguments, like get_detector() (detection/__init__.py:22-28)
        assert p._key == cls._KEY and p.model_dir == os.path.join(model_dir, sub) and os.path.isdir(p.model_dir)
        assert not p.is_loaded() and not p.is_downloaded()  # checkpoints absent -> ModelWrapper wants to download
        assert p._get_file_path("x.ckpt") == os.path.join(model_dir, sub, "x.ckpt") == P._ckpt_path(p, "x.ckpt")
        try:
            run(p.infer(np.zeros((8, 8, 3), np.uint8), 2))
            raise SystemExit("infer before load must raise")
        except Exception as e:
            assert "Tried to forward pass without having loaded the model" in str(e) and cls._KEY in str(e)
        run(p.unload())                                   # no-op


@check("_MODEL_MAPPING equals the reference's entries")
def _():
    from manga_translator.detection.ctd import ComicTextDetector
    from manga_translator.detection.default import DefaultDetector
    from manga_translator.inpainting.inpainting_lama_mpe import LamaLargeInpainter, LamaMPEInpainter
    from manga_translator.ocr.model_48px import Model48pxOCR
    from manga_translator.ocr.model_48px_ctc import Model48pxCTCOCR
    from manga_translator.upscaling.esrgan_pytorch import ESRGANUpscalerPytorch

    def norm(m):  # _check_for_malformed_model_mapping adds file='.' in place when it is missing
        return {k: {kk: vv for kk, vv in v.items() if not (kk == "file" and vv == ".")} for k, v in m.items()}

    assert norm(P.HipComicTextDetector._MODEL_MAPPING)["model-cuda"] == norm(ComicTextDetector._MODEL_MAPPING)["model-cuda"]
    for ours, ref in ((P.HipDefaultDetector, DefaultDetector), (P.HipModel48pxOCR, Model48pxOCR), (P.HipModel48pxCTCOCR, Model48pxCTCOCR),
                      (P.HipLamaMPEInpainter, LamaMPEInpainter), (P.HipLamaLargeInpainter, LamaLargeInpainter),
                      (P.HipESRGANUpscaler, ESRGANUpscalerPytorch)):
        assert norm(ours._MODEL_MAPPING) == norm(ref._MODEL_MAPPING), ours.__name__


@check("device handling through ModelWrapper.load; injected weights skip the download")
def _():
    for cls, _, _ in PLUGINS:
        kw = dict(weights={}, dictionary=[]) if issubclass(cls, P.HipModel48pxOCR) else dict(weights={})
        p = cls(**kw)
        assert p.is_downloaded()
        try:
            run(p.load("cpu"))
            raise SystemExit("load('cpu') must raise")
        except RuntimeError as e:
            assert "MI355X only" in str(e)
        assert not p.is_loaded()


@check("ModelWrapper.download() drives our _MODEL_MAPPING (offline: the fetch is redirected to a local file)")
def _():
    payload = os.urandom(4096)
    sha = hashlib.sha256(payload).hexdigest()

    class Probe(P.HipLamaMPEInpainter):
        _KEY = "lama_probe_hip"
        _MODEL_MAPPING = {"model": {"url": "https://example.invalid/releases/probe_lama.ckpt", "hash": sha, "file": "."}}

    fetched = []

    def fake_fetch(url, path):
        fetched.append(url)
        with open(path, "wb") as f:
            f.write(payload)

    INF.download_url_with_progressbar = fake_fetch
    p = Probe()
    assert not p.is_downloaded()
    run(p.download())
    assert fetched == ["https://example.invalid/releases/probe_lama.ckpt"] and p.is_downloaded()
    assert open(p._get_file_path("probe_lama.ckpt"), "rb").read() == payload
    assert Probe().is_downloaded()                       # a second instance finds the file
    # a wrong hash is reported through the reference's own exception
    class Bad(Probe):
        _KEY = "lama_bad_hip"
        _MODEL_MAPPING = {"model": {"url": "https://example.invalid/releases/bad_lama.ckpt", "hash": "0" * 64, "file": "."}}

    INF.prompt_yes_no = lambda *a, **k: False
    try:
        run(Bad().download())
        raise SystemExit("hash mismatch must abort")
    except KeyboardInterrupt:
        pass


@check("checkpoint lookup: _load reads the files ModelWrapper placed under model_dir")
def _():
    from manga_image_translator_amd import lama_schema, synth

    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(9))
    mpe = synth.synth_state_dict(lama_schema.lama_mpe_schema())
    p = P.HipLamaMPEInpainter()
    torch.save({"gen_state_dict": sd, "str_state_dict": mpe}, p._get_file_path("inpainting_lama_mpe.ckpt"))
    assert P.HipLamaMPEInpainter().is_downloaded()
    w = P._load_lama_checkpoint(p)
    assert set(w) == {"lama.gen", "lama.mpe"} and torch.equal(w["lama.gen"]["model.1.ffc.convl2l.weight"], sd["model.1.ffc.convl2l.weight"])


@check("checkpoint files in the reference's own layouts load through every plugin's loader; other layouts fail at load time")
def _():
    import copy

    from manga_image_translator_amd import ctd_schema as CS, esrgan_schema, ocr_ctc_schema, ocr_schema, synth

    # comictextdetector.pt: {'blk_det': {'cfg', 'weights'}, 'text_seg', 'text_det'} (ctd_utils/basemodel.py:205-214, yolov5/yolo.py:286-292)
    det = P.HipComicTextDetector()
    ck = {"blk_det": {"cfg": copy.deepcopy(CS.YOLOV5S_CFG), "weights": synth.synth_state_dict(CS.yolo_schema())},
          "text_seg": synth.synth_state_dict(CS.unet_head_schema()), "text_det": synth.synth_state_dict(CS.db_head_schema())}
    torch.save(ck, det._get_file_path("comictextdetector.pt"))
    w = P._load_ctd_checkpoint(det)
    assert set(w) == {"ctd.yolo", "ctd.seg", "ctd.det"} and torch.equal(w["ctd.yolo"]["model.0.conv.weight"], ck["blk_det"]["weights"]["model.0.conv.weight"])
    # the reference's own parse_model accepts the cfg and yields the channel table the engine was built for
    ref_rows = CS.layers_from_cfg(ck["blk_det"]["cfg"])
    assert [r[:5] for r in ref_rows] == CS.YOLO_LAYERS
    bad = copy.deepcopy(ck)
    bad["blk_det"]["cfg"]["width_multiple"] = 0.75            # yolov5m-like widths: the heads' hard-coded channels no longer match
    torch.save(bad, det._get_file_path("comictextdetector.pt"))
    try:
        P._load_ctd_checkpoint(det)
        raise SystemExit("a checkpoint with another yolov5 cfg must be refused")
    except ValueError as e:
        assert "yolov5s-v6" in str(e)
    bad = copy.deepcopy(ck)
    del bad["text_seg"]["down_conv1.conv.cv1.conv.weight"]
    torch.save(bad, det._get_file_path("comictextdetector.pt"))
    try:
        P._load_ctd_checkpoint(det)
        raise SystemExit("a checkpoint with a missing tensor must be refused")
    except ValueError as e:
        assert "text_seg" in str(e) and "missing" in str(e)

    # ocr_ar_48px.ckpt (bare state_dict) + alphabet-all-v7.txt, one entry per line (model_48px.py:46-52)
    ocr = P.HipModel48pxOCR()
    dictionary = ocr_schema.synth_dictionary(97)
    with open(ocr._get_file_path("alphabet-all-v7.txt"), "w", encoding="utf-8") as fp:
        fp.write("".join(c + "\n" for c in dictionary))
    sd = synth.synth_state_dict(ocr_schema.ocr48_schema(len(dictionary)))
    torch.save(sd, ocr._get_file_path("ocr_ar_48px.ckpt"))
    got_sd, got_dict = P._load_ocr_checkpoint(ocr)
    assert got_dict == dictionary and torch.equal(got_sd["embd.weight"], sd["embd.weight"])
    with open(ocr._get_file_path("alphabet-all-v7.txt"), "a", encoding="utf-8") as fp:
        fp.write("extra\n")                                    # dictionary and embedding rows out of step
    try:
        P._load_ocr_checkpoint(ocr)
        raise SystemExit("dictionary / checkpoint size mismatch must be refused")
    except ValueError as e:
        assert "another shape" in str(e)

    # ocr-ctc.ckpt = {'model': state_dict incl. the pe.pe tables the reference deletes} + alphabet-all-v5.txt (model_48px_ctc.py:38-48)
    ctc = P.HipModel48pxCTCOCR()
    with open(ctc._get_file_path("alphabet-all-v5.txt"), "w", encoding="utf-8") as fp:
        fp.write("".join(c + "\n" for c in dictionary))
    sd = synth.synth_state_dict(ocr_ctc_schema.ocr_ctc_schema(len(dictionary)))
    assert all(f"encoders.layers.{i}.pe.pe" in sd for i in range(3))   # the tables the reference's loader deletes are part of the file
    torch.save({"model": sd}, ctc._get_file_path("ocr-ctc.ckpt"))
    got_sd, got_dict = P._load_ocr_ctc_checkpoint(ctc)
    assert got_dict == dictionary and all(torch.equal(got_sd[k], v) for k, v in list(sd.items())[:8])

    # 4xESRGAN.pth: bare old-arch RRDBNet state_dict, block count inferred from the keys (esrgan_pytorch.py:476-528)
    up = P.HipESRGANUpscaler()
    sd = synth.synth_state_dict(esrgan_schema.rrdbnet_schema(3))
    torch.save(sd, up._get_file_path("4xESRGAN.pth"))
    got = P._load_esrgan_checkpoint(up)
    assert set(got) == set(sd) and 1 + max(int(k.split(".")[3]) for k in got if ".RDB" in k) == 3


@check("register() writes into the reference's registries; get_* constructs and caches our classes")
def _():
    P.register()
    for reg, get, key, cls in ((RD.DETECTORS, RD.get_detector, "ctd_hip", P.HipComicTextDetector),
                               (RD.DETECTORS, RD.get_detector, "default_hip", P.HipDefaultDetector),
                               (RO.OCRS, RO.get_ocr, "48px_hip", P.HipModel48pxOCR), (RO.OCRS, RO.get_ocr, "48px_ctc_hip", P.HipModel48pxCTCOCR),
                               (RI.INPAINTERS, RI.get_inpainter, "lama_mpe_hip", P.HipLamaMPEInpainter),
                               (RI.INPAINTERS, RI.get_inpainter, "lama_large_hip", P.HipLamaLargeInpainter),
                               (RU.UPSCALERS, RU.get_upscaler, "4xultrasharp_hip", P.HipESRGANUpscaler)):
        assert reg[key] is cls
        inst = get(key)
        assert isinstance(inst, cls) and get(key) is inst


class FakeCtdEngine:
    """Stand-in for CtdEngine.forward: fixed maps with two text-line blobs (the dense network is not under test here)."""
    device = torch.device("cpu")

    def forward(self, page):
        H, W = page.shape[1:3]
        r = min(1024 / H, 1024 / W)
        nh, nw = int(round(H * r)), int(round(W * r))
        lines = torch.zeros(1, 2, nh, nw)
        mask = torch.zeros(1, nh, nw, dtype=torch.uint8)
        for (y0, y1, x0, x1) in ((40, 90, 60, 420), (200, 520, 500, 560)):
            lines[0, :, y0:y1, x0:x1] = 0.9
            mask[0, y0:y1, x0:x1] = 255
        return mask, lines, None

    def release_workspace(self):
        pass


@check("CommonDetector.detect (the reference's caller) -> our _infer -> real Quadrilaterals, native host glue")
def _():
    from manga_image_translator_amd import imgproc

    # no GPU here: the stand-in engine hands out CPU tensors, so the device resize is replaced by its numpy twin (same tables)
    imgproc.resize_u8 = lambda t, dsize, exact=False: torch.from_numpy(np.stack([imgproc.resize_u8_host(x, dsize, exact) for x in t.numpy()]))
    # likewise the device refine_mask is replaced by the host routine it is bit-identical to (tests/test_ctd_refine_gpu.py)
    from manga_image_translator_amd import hostglue
    hostglue.refine_mask_gpu = lambda pg, pm, quads, mode=None: torch.from_numpy(hostglue.refine_mask(pg.numpy(), pm.numpy(), quads, mode))
    det = P.HipComicTextDetector(weights={})
    det.engine, det._loaded = FakeCtdEngine(), True
    page = np.full((1200, 840, 3), 245, np.uint8)
    page[60:100, 90:500] = 20
    before = page.copy()
    tls, raw_mask, mask = run(det.detect(page, 1024, 0.5, 0.7, 2.3, False, False, False, False))
    assert np.array_equal(page, before)                  # the caller keeps using the page (detection/common.py:20)
    assert len(tls) == 2 and all(type(q) is U.Quadrilateral for q in tls) and mask is None
    assert raw_mask.dtype == np.uint8 and raw_mask.shape == (1200, 840)
    assert all(q.area > 1 and q.prob > 0.6 and q.pts.shape == (4, 2) and q.pts.dtype.kind == "i" for q in tls)
    assert sorted(q.direction for q in tls) == ["h", "v"]
    # rotate=True: the reference rotates the page, calls us, and maps the boxes back (detection/common.py:24-26,59-60)
    tls_r, raw_r, _ = run(det.detect(np.ascontiguousarray(np.rot90(page, k=1)), 1024, 0.5, 0.7, 2.3, False, False, True, False))
    assert len(tls_r) == 2 and raw_r.shape == (840, 1200)
    run(det.unload())
    assert det.engine is None and not det.is_loaded()


@check("the reference's REAL ComicTextDetector._infer (network stubbed) == HipComicTextDetector._infer (engine stubbed identically)")
def _():
    """Both sides see the same network outputs; everything after the network is each side's own code: the reference's letterbox
    crop, postprocess_mask, SegDetectorRepresenter, 0.6 filter, cv2.resize and textmask.refine_mask (its own Python over the
    cv2 / pyclipper / shapely stand-ins) against the plugin's native box extraction, table-driven resize and refine_mask."""
    from PIL import Image, ImageDraw

    import manga_translator.detection.ctd as RC
    import manga_translator.detection.ctd_utils.utils.db_utils as DBU
    from manga_image_translator_amd import hostglue, imgproc
    from oracle import ref_import as R

    DBU.cv2, DBU.pyclipper, DBU.Polygon = R.box_shims()
    imgproc.resize_u8 = lambda t, dsize, exact=False: torch.from_numpy(np.stack([imgproc.resize_u8_host(x, dsize, exact) for x in t.numpy()]))
    hostglue.refine_mask_gpu = lambda pg, pm, quads, mode=None: torch.from_numpy(hostglue.refine_mask(pg.numpy(), pm.numpy(), quads, mode))

    H, W = 1200, 840                                   # letterbox: r = 1024 / 1200, un-padded area 1024 x 717
    r = min(1024 / H, 1024 / W)
    nh, nw = int(round(H * r)), int(round(W * r))
    rng = np.random.default_rng(11)
    page = np.full((H, W, 3), 238, np.uint8) - rng.integers(0, 20, (H, W, 3)).astype(np.uint8)
    lines_im, mask_im = Image.new("F", (1024, 1024), 0.03), Image.new("F", (1024, 1024), 0.0)
    dl, dm = ImageDraw.Draw(lines_im), ImageDraw.Draw(mask_im)
    boxes = [(60, 40, 420, 90), (500, 200, 560, 520), (100, 600, 600, 660), (80, 800, 300, 830), (640, 700, 700, 1000)]
    for k, (x0, y0, x1, y1) in enumerate(boxes):
        dl.rectangle([x0 + 6, y0 + 6, x1 - 6, y1 - 6], fill=[0.95, 0.9, 0.8, 0.55, 0.7][k])   # one below the 0.6 score filter
        dm.rectangle([x0, y0, x1, y1], fill=0.85)
        X0, Y0, X1, Y1 = [int(v / r) for v in (x0, y0, x1, y1)]
        for _ in range(40):                            # dark strokes on the page under the line
            sx, sy = int(rng.integers(X0, max(X1 - 10, X0 + 1))), int(rng.integers(Y0, max(Y1 - 12, Y0 + 1)))
            page[sy:sy + int(rng.integers(4, 12)), sx:sx + int(rng.integers(3, 9))] = int(rng.integers(0, 70))
    lines_f = np.asarray(lines_im).copy()
    mask_f = np.asarray(mask_im).copy()
    lines_f[:, nw:] = 0.0
    mask_f[:, nw:] = 0.0
    lines_map = np.stack([lines_f, lines_f * 0.5])[None].astype(np.float32)       # [1, 2, 1024, 1024]
    mask_map = mask_f[None, None].astype(np.float32)                              # [1, 1, 1024, 1024]

    ref = RC.ComicTextDetector.__new__(RC.ComicTextDetector)                       # the reference's own class, no checkpoint
    ref.model = lambda img_in: (None, torch.from_numpy(mask_map.copy()), torch.from_numpy(lines_map.copy()))
    ref.backend, ref.half, ref.device, ref.input_size = "torch", False, "cpu", (1024, 1024)
    ref.seg_rep = DBU.SegDetectorRepresenter(thresh=0.3)
    tls_ref, mask_ref, extra = run(RC.ComicTextDetector._infer(ref, page.copy(), 1024, 0.5, 0.7, 2.3))
    assert extra is None and len(tls_ref) == 4 and mask_ref.shape == (H, W) and mask_ref.any()

    class SameMapsEngine:
        device = torch.device("cpu")

        def forward(self, pg):                         # what CtdEngine returns for these network outputs (ctd.py:30-44,152-153)
            m = torch.from_numpy((mask_map[0, 0, :nh, :nw] * 255).astype(np.uint8))[None]
            return m, torch.from_numpy(lines_map[..., :nh, :nw].copy()), None

        def release_workspace(self):
            pass

    det = P.HipComicTextDetector(weights={})
    det.engine, det._loaded = SameMapsEngine(), True
    tls, mask, extra = run(det.infer(page.copy(), 1024, 0.5, 0.7, 2.3))
    assert extra is None and len(tls) == len(tls_ref)
    for a, b in zip(tls, tls_ref):
        assert type(a) is type(b) is U.Quadrilateral and np.array_equal(a.pts, b.pts) and abs(a.prob - b.prob) < 1e-6
    assert mask.dtype == mask_ref.dtype and np.array_equal(mask, mask_ref)
    run(det.unload())


@check("the reference's REAL LamaMPEInpainter._infer (network stubbed) == HipLamaMPEInpainter._infer (engine stubbed identically)")
def _():
    """Pages that are aligned, not a multiple of 8, and larger than inpainting_size: the reference's resize_keep_aspect /
    cv2.resize / tensor prep / uint8 truncation / resize back / composite (inpainting_lama_mpe.py:56-118, its own code over the cv2
    stand-in) against the plugin's table-driven resizes, engine contract and select composite — same stub network on both sides."""
    import logging

    import manga_translator.inpainting.inpainting_lama_mpe as RL
    from manga_image_translator_amd import imgproc

    def net(x, m):  # any deterministic map [1,3,H,W] x [1,1,H,W] -> [1,3,H,W] in [0, 1]; the same function object on both sides
        p = torch.nn.functional.avg_pool2d(x, 3, 1, 1)
        return (0.55 * x + 0.35 * p + 0.1 * m).clamp(0, 1)

    class StubLama(RL.LamaFourier):
        def __init__(self):      # not the reference's constructor: no generator is built
            pass

        def __call__(self, img, mask):
            return net(img, mask)

    ref = RL.LamaMPEInpainter.__new__(RL.LamaMPEInpainter)
    ref.model, ref.device, ref.logger = StubLama(), "cpu", logging.getLogger("ref-lama")

    class SameNetEngine:
        device = torch.device("cpu")

        def forward(self, img, msk, composite=True):   # LamaEngine's contract: u8 [B,H,W,3] + u8 [B,H,W] -> u8 [B,H,W,3]
            x = img.permute(0, 3, 1, 2).float() / 255.0
            m = msk[:, None].float() / 255.0
            m[m < 0.5] = 0
            m[m >= 0.5] = 1
            x = x * (1 - m)
            out = torch.from_numpy((net(x, m).permute(0, 2, 3, 1).numpy() * 255.0).astype(np.uint8))
            if composite:                               # img_inpainted * mask_original + img_original * (1 - mask_original) (:57-61,117)
                out = torch.where((msk >= 127)[..., None], out, img)
            return out

        def release_workspace(self):
            pass

    imgproc.resize_u8 = lambda t, dsize, exact=False: torch.from_numpy(np.stack([imgproc.resize_u8_host(x, dsize, exact) for x in t.numpy()]))
    imgproc.select_u8 = lambda mask, thr, a, b: torch.where((mask >= thr)[..., None], a, b)
    inp = P.HipLamaMPEInpainter(weights={})
    inp.engine, inp._loaded = SameNetEngine(), True
    rng = np.random.default_rng(21)
    for (H, W, size) in ((64, 72, 1024), (250, 187, 1024), (300, 232, 256), (260, 520, 256)):
        page = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        mask = np.zeros((H, W), np.uint8)
        mask[H // 4:H // 2, W // 5:3 * W // 4] = 255
        mask[rng.random((H, W)) < 0.02] = int(rng.integers(100, 160))   # values around the 127 threshold
        want = run(RL.LamaMPEInpainter._infer(ref, page.copy(), mask.copy(), None, size))
        got = run(inp.infer(page.copy(), mask.copy(), None, size))
        assert got.shape == want.shape == (H, W, 3) and got.dtype == np.uint8
        assert np.array_equal(got, want.astype(np.uint8)), (H, W, size, int((got != want).sum()))
    run(inp.unload())


@check("the reference's REAL DefaultDetector._infer (network stubbed) == HipDefaultDetector._infer (engine stubbed identically)")
def _():
    """bilateralFilter + resize_aspect_ratio, SegDetectorRepresenter with the caller's thresholds, adjustResultCoordinates, the
    area > 16 filter (and the reference's pairing of the filtered boxes with the unfiltered score list), the x2 mask resize, the
    padding crop and the uint8 clip (default.py:56-103) — the reference's own code over the stand-ins against the plugin."""
    import logging

    import manga_translator.detection.default as RDf
    import manga_translator.detection.default_utils.dbnet_utils as DBN
    from manga_image_translator_amd import imgproc
    from oracle import imgproc as OI, make_golden as MG, ref_import as R

    DBN.cv2, DBN.pyclipper, DBN.Polygon = R.box_shims()
    cv = sys.modules["cv2"]
    base_resize = cv.resize

    def resize(src, dsize, interpolation=1, **kw):     # the float32 x2 map resize of default.py:89 (plain bilinear at pixel centres)
        if src.dtype != np.float32:
            return base_resize(src, dsize, interpolation=interpolation, **kw)
        h, w = src.shape
        out = np.empty((dsize[1], dsize[0]), np.float32)
        fy = ((np.arange(dsize[1]) + 0.5) * (h / dsize[1]) - 0.5).astype(np.float32)
        fx = ((np.arange(dsize[0]) + 0.5) * (w / dsize[0]) - 0.5).astype(np.float32)
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        wy, wx = (fy - y0).astype(np.float32), (fx - x0).astype(np.float32)
        wy[(y0 < 0) | (y0 >= h - 1)] = 0
        wx[(x0 < 0) | (x0 >= w - 1)] = 0
        y0, x0 = np.clip(y0, 0, h - 1), np.clip(x0, 0, w - 1)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        rows = src[:, x0] * (np.float32(1) - wx)[None] + src[:, x1] * wx[None]
        return (rows[y0] * (np.float32(1) - wy)[:, None] + rows[y1] * wy[:, None]).astype(np.float32)

    cv.resize = RDf.cv2.resize = resize
    cv.bilateralFilter = RDf.cv2.bilateralFilter = lambda img, d, sc, ss: OI.bilateral_filter_u8(img, d, float(sc), float(ss))
    try:
        H, W, size = 300, 210, 512                     # long side -> 512: 512 x 358, padded to 512 x 512 (pad_w 154)
        rng = np.random.default_rng(5)
        page = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        pred = MG.boxes_scene(6, 512, 512)
        pred[:, 358:] = 0.0
        db = np.stack([pred, pred * 0.5])[None].astype(np.float32)
        mask_small = np.ascontiguousarray(pred[::2, ::2][None, None] * 1.2)    # some values clip at 255
        seen = {}

        def fake_forward(batch, device):
            seen["batch"] = np.asarray(batch)
            return db.copy(), mask_small.copy()

        RDf.det_batch_forward_default = fake_forward
        ref = RDf.DefaultDetector.__new__(RDf.DefaultDetector)
        ref.device, ref.logger = "cpu", logging.getLogger("ref-default")
        tls_ref, raw_ref, extra = run(RDf.DefaultDetector._infer(ref, page.copy(), size, 0.4, 0.6, 2.0))
        assert extra is None and len(tls_ref) >= 3 and raw_ref.shape == (512, 358) and raw_ref.max() == 255

        def pre_twin(image, detect_size):              # default_preprocess_gpu's host twin (tests/test_dbnet_gpu.py pins the device form)
            img = image.numpy()
            ratio = detect_size / max(img.shape[:2])
            th, tw = int(round(img.shape[0] * ratio)), int(round(img.shape[1] * ratio))
            proc = imgproc.resize_u8_host(OI.bilateral_filter_u8(img, 17, 80.0, 80.0), (tw, th))
            ph, pw = (256 - th % 256) % 256, (256 - tw % 256) % 256
            canvas = np.zeros((th + ph, tw + pw, 3), np.uint8)
            canvas[:th, :tw] = proc
            return torch.from_numpy(canvas)[None], ratio, pw, ph

        P.default_preprocess_gpu = pre_twin

        class SameMapsEngine:
            device = torch.device("cpu")

            def forward(self, pg):
                seen["page"] = pg[0].numpy()
                return torch.from_numpy(db.copy()), torch.from_numpy(mask_small[0].copy())

            def release_workspace(self):
                pass

        det = P.HipDefaultDetector(weights={})
        det.engine, det._loaded = SameMapsEngine(), True
        tls, raw, extra = run(det.infer(page.copy(), size, 0.4, 0.6, 2.0))
        assert np.array_equal(seen["page"], seen["batch"][0])          # both networks were fed the same bytes
        assert extra is None and len(tls) == len(tls_ref)
        for a, b in zip(tls, tls_ref):
            assert np.array_equal(a.pts, b.pts) and abs(a.prob - b.prob) < 1e-6
        assert raw.dtype == raw_ref.dtype and np.array_equal(raw, raw_ref)
        run(det.unload())
    finally:
        cv.resize = base_resize


@check("CommonOCR.recognize: the direction vote is the reference's own _generate_text_direction and equals the native one")
def _():
    from manga_image_translator_amd import textline as TL

    pts = [np.array([[100, 100], [160, 100], [160, 400], [100, 400]]), np.array([[170, 100], [230, 100], [230, 390], [170, 390]]),
           np.array([[400, 600], [800, 600], [800, 650], [400, 650]]), np.array([[240, 105], [300, 105], [300, 180], [240, 180]])]
    ref_lines = [U.Quadrilateral(p, "", 1.0) for p in pts]
    ocr = P.HipModel48pxOCR(weights={}, dictionary=[])
    got = ocr._directions(ref_lines)
    own = TL.generate_text_direction([TL.Quadrilateral(p) for p in pts])
    assert [(tuple(map(tuple, q.pts)), d) for q, d in got] == [(tuple(map(tuple, q.pts)), d) for q, d in own]
    assert all(type(q) is U.Quadrilateral for q, _ in got)


@check("the reference's REAL Model48pxOCR._infer (recogniser stubbed) == HipModel48pxOCR._infer (engine stubbed identically)")
def _():
    """Direction vote, rectified crops, the sorted-by-width chunks of 16, the probability threshold, the token / colour decode
    (AvgMeter means, <S> </S> <SP>), attribute assignment and the order of the returned lines (model_48px.py:67-180, the
    reference's own code over the cv2 stand-in) against the plugin over Ocr48Engine's REAL plan_pages.  The recogniser on both
    sides is the same function of a line's rectified pixels."""
    import logging
    import zlib

    import manga_translator.ocr.model_48px as RM
    from manga_image_translator_amd import ocr48
    from oracle import textline as OT

    dictionary = ["<S>", "</S>", "<SP>"] + list("abcdefghijklmnopqrstuvwxyz0123456789")
    seen = {"ref": [], "hip": []}

    def read_line(crop):  # u8 [48, w, 3] -> (token ids, prob, colour rows [T,10]); deterministic in the pixels
        rng = np.random.default_rng(zlib.crc32(np.ascontiguousarray(crop).tobytes()))
        T = int(rng.integers(1, 14))
        toks = rng.integers(2, len(dictionary), T)
        if rng.random() < 0.3:
            toks[int(rng.integers(0, T))] = 1           # an end symbol in the middle cuts the line there
        if rng.random() < 0.2:
            toks[0] = 0                                  # a start symbol is skipped, its colour row too
        cols = rng.random((T, 10)).astype(np.float32)
        cols[:, :6] = cols[:, :6] * 1.3 - 0.15           # colour heads are unbounded: the clamp to [0, 255] must act
        return toks.astype(np.int64), float(np.float32(rng.random())), cols  # a probability both sides hold exactly

    class StubOCR:
        def __init__(self):
            self.dictionary = dictionary

        def infer_beam_batch_tensor(self, image_tensor, widths, beams_k=5, max_seq_length=255):
            assert beams_k == 5 and max_seq_length == 255 and image_tensor.shape[1:3] == (3, 48)
            u8 = (image_tensor * 127.5 + 127.5).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
            assert np.all(u8[0, :, max(widths):] == 0)   # the chunk is zero padded to 4 * (max + 7) // 4 (:84-85)
            ret = []
            for j, w in enumerate(widths):
                seen["ref"].append(u8[j, :, :w].copy())
                toks, prob, cols = read_line(u8[j, :, :w])
                c = torch.from_numpy(cols)
                ret.append((torch.from_numpy(toks), prob, c[:, 0:3], c[:, 3:6], c[:, 6:8], c[:, 8:10]))
            return ret

    ref = RM.Model48pxOCR.__new__(RM.Model48pxOCR)
    ref.model, ref.use_gpu, ref.device, ref.logger = StubOCR(), False, "cpu", logging.getLogger("ref-ocr")

    class SameReaderEngine(ocr48.Ocr48Engine):
        device = torch.device("cpu")

        def __init__(self):      # no weights: only the host planning of the real engine is used
            pass

        def recognize_pages(self, pages_u8, quads_per_page, max_seq_length=255, suppress_eos=False, directions=None, group_pages=8):
            assert max_seq_length == 255 and not suppress_eos and pages_u8.dtype == torch.uint8
            P_, H, W, _ = pages_u8.shape
            plan = self.plan_pages(quads_per_page, H, W, directions)          # Ocr48Engine's own chunking
            n = len(plan["order"])
            tokens, length = np.zeros((n, 16), np.int64), np.zeros(n, np.int64)
            prob, colors = np.zeros(n, np.float32), np.zeros((n, 16, 10), np.float32)
            for row, (p, i) in enumerate(plan["order"]):
                crop = OT.get_transformed_region(pages_u8[p].numpy(), np.asarray(quads_per_page[p][i].pts), directions[p][i], 48)
                seen["hip"].append(crop)
                toks, pr, cols = read_line(crop)
                tokens[row, 1:1 + len(toks)], length[row] = toks, len(toks) + 1   # decode()'s rows start with <S>
                prob[row], colors[row, :len(toks)] = pr, cols
            t = torch.from_numpy
            return dict(tokens=t(tokens), length=t(length), prob=t(prob), colors=t(colors), order=plan["order"])

        def release_workspace(self):
            pass

    rng = np.random.default_rng(33)
    page = rng.integers(0, 256, (900, 700, 3)).astype(np.uint8)
    boxes = []
    for k in range(21):                                    # more than one chunk of 16; vertical, horizontal and skewed lines
        x, y = int(rng.integers(5, 500)), int(rng.integers(5, 650))
        if k % 3 == 0:
            w, h = int(rng.integers(20, 40)), int(rng.integers(80, 230))
        else:
            w, h = int(rng.integers(60, 190)), int(rng.integers(18, 44))
        sk = int(rng.integers(-6, 7)) if k % 4 == 0 else 0
        boxes.append(np.array([[x, y + sk], [x + w, y], [x + w, y + h - sk], [x, y + h]]))
    boxes[7][:, 0] = boxes[3][:, 0]                        # equal widths: the stable sort's tie order matters
    boxes[7][:, 1] = boxes[3][:, 1] + 250
    for prob_cfg in (None, 0.55):
        cfg = type("Cfg", (), {"prob": prob_cfg})()
        want_lines = [U.Quadrilateral(b.copy(), "", 1.0) for b in boxes]
        got_lines = [U.Quadrilateral(b.copy(), "", 1.0) for b in boxes]
        seen["ref"].clear(), seen["hip"].clear()
        want = run(RM.Model48pxOCR._infer(ref, page.copy(), want_lines, cfg))
        ocr = P.HipModel48pxOCR(weights={}, dictionary=dictionary)
        ocr.engine, ocr._loaded = SameReaderEngine(), True
        got = run(ocr.infer(page.copy(), got_lines, cfg))
        assert len(seen["ref"]) == len(seen["hip"]) == len(boxes)
        assert all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(seen["ref"], seen["hip"])), "crop order / pixels"
        assert 0 < len(got) == len(want) < len(boxes), (len(got), len(want))
        key = lambda q: (tuple(map(tuple, np.asarray(q.pts).tolist())), q.text, float(q.prob), q.fg_r, q.fg_g, q.fg_b,
                         q.bg_r, q.bg_g, q.bg_b, q.direction)
        for a, b in zip(got, want):
            assert key(a) == key(b), (key(a), key(b))
        assert all(any(q is l for l in got_lines) for q in got)       # the caller's own objects come back, mutated
        assert any(q.text and " " not in q.text for q in got) and any(q.fg_r in (0, 255) or q.bg_r in (0, 255) for q in got)
        run(ocr.unload())


@check("the reference's REAL ESRGANUpscalerPytorch._infer (network stubbed) == HipESRGANUpscaler._infer (engine stubbed identically)")
def _():
    """PIL -> RGB array -> BGR / 255 tensor batch -> network -> clip -> x 255 truncation -> RGB -> PIL -> bilinear resize by ratio / 4
    (esrgan_pytorch.py:537-549, the reference's own code with the real PIL) against the plugin's u8-in / u8-out engine contract."""
    from PIL import Image

    import manga_translator.upscaling.esrgan_pytorch as RE

    def net(x):  # BGR [B,3,h,w] in [0,1] -> [B,3,4h,4w], deliberately leaving [0,1] so the clip acts
        up = torch.nn.functional.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)
        return up * 1.15 - 0.05 + 0.02 * up.flip(1)

    ref = RE.ESRGANUpscalerPytorch.__new__(RE.ESRGANUpscalerPytorch)
    ref.model, ref.device = net, "cpu"

    class SameNetEngine:
        device = torch.device("cpu")

        def forward(self, rgb_u8):                       # EsrganEngine's contract: u8 RGB [B,h,w,3] -> u8 RGB [B,4h,4w,3]
            x = rgb_u8.flip(-1).permute(0, 3, 1, 2).float() / 255.0
            y = net(x).clip(0, 1).permute(0, 2, 3, 1).flip(-1)
            return torch.from_numpy((y.numpy() * 255.0).astype(np.uint8))

        def release_workspace(self):
            pass

    up = P.HipESRGANUpscaler(weights={})
    up.engine, up._loaded = SameNetEngine(), True
    rng = np.random.default_rng(5)
    for ratio in (2, 3, 4):
        batch = [Image.fromarray(rng.integers(0, 256, (37, 52, 3)).astype(np.uint8)) for _ in range(2)]
        batch.append(Image.fromarray(rng.integers(0, 256, (37, 52)).astype(np.uint8)))          # a grey page: convert("RGB") on both sides
        want = run(RE.ESRGANUpscalerPytorch._infer(ref, batch, ratio))
        got = run(up.infer(batch, ratio))
        assert len(got) == len(want) == 3
        for g, w in zip(got, want):
            assert g.size == w.size == (int(round(52 * ratio)), int(round(37 * ratio))) and g.mode == w.mode == "RGB"
            assert np.array_equal(np.asarray(g), np.asarray(w)), ratio
    assert run(up.infer([], 2)) == [] and run(RE.ESRGANUpscalerPytorch._infer(ref, [], 2)) == []
    run(up.unload())


@check("the reference's REAL Model48pxCTCOCR._infer (recogniser stubbed) == HipModel48pxCTCOCR._infer (engine stubbed identically)")
def _():
    """Same idea for the CTC recogniser (model_48px_ctc.py:62-160): chunks padded to max + 7 + 128, exp(mean log-prob) against the 0.5
    default threshold, colours averaged over non-space characters only.  The recogniser on both sides is a function of the WHOLE padded
    chunk row, so a different padding width or zero fill would show; the plugin's device rectification is replaced by its host twin."""
    import logging
    import zlib

    import manga_translator.ocr.model_48px_ctc as RC
    from oracle import textline as OT

    dictionary = ["<blank>", "<SP>"] + list("abcdefghijklmnopqrstuvwxyz")

    def read_row(row):  # u8 [48, wp, 3] -> [(chid, logprob, fr, fg, fb, br, bg, bb)]
        rng = np.random.default_rng(zlib.crc32(np.ascontiguousarray(row).tobytes()))
        n = int(rng.integers(0, 9))                        # empty lines are dropped on both sides
        out = []
        for _ in range(n):
            cols = rng.random(6).astype(np.float32)
            chid = 1 if rng.random() < 0.2 else int(rng.integers(1, len(dictionary)))   # <SP> often enough: its colours do not count
            out.append((chid, float(np.float32(-rng.random() * 1.5)), *[float(c) for c in cols]))
        return out

    class StubCTC:
        def __init__(self):
            self.dictionary = dictionary

        def decode(self, images, widths, blank, verbose=False):
            assert blank == 0 and images.shape[1:3] == (3, 48) and images.shape[3] == 4 * (max(widths) + 7) // 4 + 128
            u8 = (images * 127.5 + 127.5).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()
            return [read_row(u8[j]) for j in range(len(widths))]

    ref = RC.Model48pxCTCOCR.__new__(RC.Model48pxCTCOCR)
    ref.model, ref.use_gpu, ref.device, ref.logger = StubCTC(), False, "cpu", logging.getLogger("ref-ctc")

    class SameReaderEngine:
        device = torch.device("cpu")

        def forward(self, region_u8):                      # OcrCtcEngine: chunk tensor -> (logits, colours); here the rows themselves
            return region_u8, None

        def decode(self, logits, colors, blank):
            assert blank == 0
            return [read_row(r.numpy()) for r in logits]

        def release_workspace(self):
            pass

    def host_rectify(page, quads, dirs, idx, records, wp):  # the device routine's host twin (tests/test_ocr_gpu.py pins the kernel to it)
        region = np.zeros((len(idx), 48, wp, 3), np.uint8)
        for j, i in enumerate(idx):
            crop = OT.get_transformed_region(page[0].numpy(), np.asarray(quads[i].pts), dirs[i], 48)
            region[j, :, :crop.shape[1]] = crop
        return torch.from_numpy(region)

    rng = np.random.default_rng(44)
    page = rng.integers(0, 256, (700, 640, 3)).astype(np.uint8)
    boxes = []
    for k in range(19):
        x, y = int(rng.integers(5, 430)), int(rng.integers(5, 450))
        w, h = (int(rng.integers(20, 40)), int(rng.integers(80, 230))) if k % 3 == 0 else (int(rng.integers(60, 190)), int(rng.integers(18, 44)))
        boxes.append(np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]]))
    for k, b in enumerate(boxes):                          # every other line sits in a clean white bubble: the bubble filter keeps those
        if k % 2 == 0:
            x0, y0, x1, y1 = b[0, 0], b[0, 1], b[2, 0], b[2, 1]
            page[max(y0 - 4, 0):y1 + 5, max(x0 - 4, 0):x1 + 5] = 250
            page[y0 + 6:y1 - 5:3, x0 + 6:x1 - 5] = 20
    kept = {}
    for prob_cfg, bubble in ((None, 0), (0.3, 0), (0.3, 10)):
        cfg = type("Cfg", (), {"prob": prob_cfg, "ignore_bubble": bubble})()
        want_lines = [U.Quadrilateral(b.copy(), "", 1.0) for b in boxes]
        got_lines = [U.Quadrilateral(b.copy(), "", 1.0) for b in boxes]
        want = run(RC.Model48pxCTCOCR._infer(ref, page.copy(), want_lines, cfg))
        ocr = P.HipModel48pxCTCOCR(weights={}, dictionary=dictionary)
        ocr.engine, ocr._loaded, ocr._rectify = SameReaderEngine(), True, host_rectify
        got = run(ocr.infer(page.copy(), got_lines, cfg))
        key = lambda q: (tuple(map(tuple, np.asarray(q.pts).tolist())), q.text, float(q.prob), q.fg_r, q.fg_g, q.fg_b, q.bg_r, q.bg_g, q.bg_b)
        assert 0 < len(got) == len(want) < len(boxes), (len(got), len(want))
        for a, b in zip(got, want):
            assert key(a) == key(b), (key(a), key(b))
        assert all(any(q is l for l in got_lines) for q in got), "the caller's own objects must come back"
        kept[(prob_cfg, bubble)] = [key(q) for q in got]
        run(ocr.unload())
    assert kept[(0.3, 10)] != kept[(0.3, 0)]              # the filter (the reference's own is_ignore on its side) changed what is read
    assert any(" " in k[1] for ks in kept.values() for k in ks)   # <SP> went through the text / colour rules somewhere


@check("exceptions propagate through the reference's infer()/inpaint() wrappers")
def _():
    inp = P.HipLamaMPEInpainter(weights={})
    inp._loaded = True
    try:
        run(inp.inpaint(np.zeros((16, 16, 3), np.uint8), np.zeros((8, 8), np.uint8), None, 1024))
        raise SystemExit("bad shapes must raise")
    except ValueError as e:
        assert "bad shapes" in str(e)
    up = P.HipESRGANUpscaler(weights={})
    assert run(up.upscale(["img"], 1)) == ["img"]          # ratio 1 short-circuits in CommonUpscaler.upscale


print(f"ALL {len(done)} BOUNDARY CHECKS PASSED")
