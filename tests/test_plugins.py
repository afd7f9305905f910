"""Plugin contract without a GPU: lifecycle, device handling, error behaviour, and the OCR result
decoding (model_48px.py:121-175) against a line-by-line restatement that uses the reference's AvgMeter arithmetic."""
import asyncio

import numpy as np
import pytest

from manga_image_translator_amd import plugins as P


def run(coro):
    return asyncio.new_event_loop().run_until_complete(coro)


@pytest.mark.parametrize("cls", [P.HipComicTextDetector, P.HipDefaultDetector, P.HipModel48pxOCR, P.HipModel48pxCTCOCR, P.HipLamaMPEInpainter, P.HipLamaLargeInpainter,
                                 P.HipESRGANUpscaler])
def test_lifecycle_and_device_errors(cls):
    p = cls()                                   # constructed with no arguments, touches no GPU
    assert not p.is_loaded()
    with pytest.raises(Exception, match="without having loaded"):
        run(p.infer(np.zeros((8, 8, 3), np.uint8), 2))
    if not P.HAVE_REFERENCE:
        # like ModelWrapper.load (utils/inference.py:330-338) the download step comes first; stand-alone there is no downloader
        assert not p.is_downloaded() and p._MODEL_MAPPING
        with pytest.raises(FileNotFoundError, match="pass weights="):
            run(p.load("cuda"))
    kw = dict(weights={}, dictionary=[]) if issubclass(cls, P.HipModel48pxOCR) else dict(weights={})
    p = cls(**kw)                               # state dicts handed over: nothing to download
    assert p.is_downloaded() and not p.is_loaded()
    with pytest.raises(RuntimeError, match="MI355X only"):
        run(p.load("cpu"))                      # the reference passes 'cpu' without --use-gpu: no CPU fallback here
    assert not p.is_loaded()
    run(p.unload())                             # unloading an unloaded plugin is a no-op, like ModelWrapper.unload


def test_model_mappings_are_well_formed():
    """The checks ModelWrapper._check_for_malformed_model_mapping applies (utils/inference.py:124-134), and the sha256 pins."""
    import re

    for cls in (P.HipComicTextDetector, P.HipDefaultDetector, P.HipModel48pxOCR, P.HipModel48pxCTCOCR, P.HipLamaMPEInpainter,
                P.HipLamaLargeInpainter, P.HipESRGANUpscaler):
        assert cls._MODEL_MAPPING and cls._KEY.endswith("_hip")
        for key, m in cls._MODEL_MAPPING.items():
            assert re.search(r"^https?://", m["url"]) and re.fullmatch(r"[0-9a-f]{64}", m["hash"]) and not ("file" in m and "archive" in m)


def test_variants():
    assert (P.HipLamaMPEInpainter.N_BLOCKS, P.HipLamaMPEInpainter.USE_MPE) == (9, True)
    assert (P.HipLamaLargeInpainter.N_BLOCKS, P.HipLamaLargeInpainter.USE_MPE) == (18, False)
    if not P.HAVE_REFERENCE:
        with pytest.raises(RuntimeError):
            P.register()


def _ref_decode(tokens, fg_pred, bg_pred, fg_ind, bg_ind, dictionary):
    """model_48px.py:124-158 verbatim in structure (AvgMeter = running sum / count)."""
    class Avg:
        def __init__(self):
            self.s, self.c = 0, 0

        def __call__(self, v=None):
            if v is not None:
                self.s += v
                self.c += 1
            return self.s / self.c if self.c > 0 else 0

    has_fg, has_bg = fg_ind[:, 1] > fg_ind[:, 0], bg_ind[:, 1] > bg_ind[:, 0]
    m = [Avg() for _ in range(6)]
    seq = []
    for chid, cf, cb, hf, hb in zip(tokens, fg_pred, bg_pred, has_fg, has_bg):
        ch = dictionary[chid]
        if ch == "<S>":
            continue
        if ch == "</S>":
            break
        if ch == "<SP>":
            ch = " "
        seq.append(ch)
        if hf:
            for k in range(3):
                m[k](int(cf[k] * 255))
        src = cb if hb else cf
        for k in range(3):
            m[3 + k](int(src[k] * 255))
    vals = [min(max(int(x()), 0), 255) for x in m]
    return "".join(seq), tuple(vals[:3]), tuple(vals[3:])


def test_decode_line_matches_reference_logic():
    rng = np.random.default_rng(0)
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x3041 + i) for i in range(60)]
    for trial in range(50):
        n = int(rng.integers(1, 12))
        toks = rng.integers(3, len(dictionary), size=n)
        if trial % 3 == 0:
            toks[rng.integers(0, n)] = 2          # an </S> in the middle stops the line
        if trial % 5 == 0:
            toks[0] = 1
        cols = rng.normal(0.5, 0.6, size=(n, 10)).astype(np.float32)  # out-of-range colours exercise the clamps
        got = P.decode_line(toks, cols, dictionary)
        ref = _ref_decode(toks, cols[:, 0:3], cols[:, 3:6], cols[:, 6:8], cols[:, 8:10], dictionary)
        assert got == ref


def test_decode_lines_equals_decode_line_row_by_row():
    """The vectorised decoder of a whole result tensor against the per-line definition: every row, including </S> in the middle, <S>
    inside a line, lengths shorter than the tensor and rows with no foreground sample."""
    rng = np.random.default_rng(1)
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x3041 + i) for i in range(60)]
    n, T = 200, 14
    toks = rng.integers(3, len(dictionary), size=(n, T + 1))
    toks[:, 0] = 1
    lens = rng.integers(1, T + 2, size=n)            # the start symbol counts
    for r in range(0, n, 3):
        toks[r, rng.integers(1, T + 1)] = 2
    for r in range(0, n, 7):
        toks[r, rng.integers(1, T + 1)] = 1
    cols = rng.normal(0.5, 0.6, size=(n, T, 10)).astype(np.float32)
    cols[5:9, :, 7] = -5.0                            # never a foreground colour: mean of nothing = 0
    got = P.decode_lines(toks, lens, cols, dictionary)
    for r in range(n):
        k = int(lens[r]) - 1
        assert got[r] == P.decode_line(toks[r, 1:1 + k], cols[r, :k], dictionary), r
    some = P.decode_lines(toks, lens, cols, dictionary, rows=[3, 8])
    assert some[3] == got[3] and some[8] == got[8] and some[0] is None
    assert P.decode_lines(toks[:0], lens[:0], cols[:0], dictionary) == []


def test_decode_ctc_line_matches_reference_logic():
    """model_48px_ctc.py:105-134: mean log-prob -> prob, colours averaged over non-space characters only."""
    dictionary = ["<PAD>", "<S>", "</S>", "<SP>", "a", "b", "c"]
    line = [(4, -0.1, 0.2, 0.4, 0.6, 0.9, 0.8, 0.7), (3, -0.5, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0), (6, -0.3, 0.4, 0.0, 1.0, 0.1, 0.2, 0.3)]
    txt, prob, fg, bg = P.decode_ctc_line(line, dictionary)
    assert txt == "a c" and prob == pytest.approx(np.exp((-0.1 - 0.5 - 0.3) / 3))
    assert fg == (int((int(0.2 * 255) + int(0.4 * 255)) / 2), int((int(0.4 * 255) + 0) / 2), int((int(0.6 * 255) + 255) / 2))
    assert bg == (int((int(0.9 * 255) + int(0.1 * 255)) / 2), int((int(0.8 * 255) + int(0.2 * 255)) / 2), int((int(0.7 * 255) + int(0.3 * 255)) / 2))
    assert P.decode_ctc_line([], dictionary) is None
