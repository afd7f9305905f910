"""Run in a FRESH interpreter by tests/test_reference_boundary.py: the plugin contract against the reference's REAL base classes.

oracle.ref_boundary.install() makes ``manga_translator.{utils,config,detection,ocr,inpainting,upscaling}`` importable from
/root/reference, so manga_image_translator_amd.plugins takes its HAVE_REFERENCE branch: every plugin here derives from the
reference's own OfflineDetector / OfflineOCR / OfflineInpainter / OfflineUpscaler (utils/inference.py ModelWrapper) and is driven
through the reference's own callers (CommonDetector.detect, CommonOCR.recognize, CommonInpainter.inpaint, get_detector ...).
No GPU: where a call would reach the dense engine, a stand-in engine returns fixed tensors — what is under test is the boundary
(types, lifecycle, registry, checkpoint lookup, download flow, error propagation), not the kernels.
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
