"""Kernels of this library next to ANOTHER queue's kernels on the same CUs (a second process on the device, a second stream).

Round 3 found mit_rfft_rows / mit_irfft_rows returning wrong workgroups (7-15 of 60 launches) while another process looped the 128 x 128
split-bf16 GEMM tile; round 4 found the cause in the victims' own instructions: the SLP vectoriser had packed their scalar fp32
arithmetic into v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers, and on gfx950 such an instruction now and then returns a wrong
16-lane pass while another kernel's MFMA waves share the CU (xpos_rotate_kernel: the second product of x.x * c + (-x.y) * s missing in 16
consecutive lanes).  The library is built with -fno-slp-vectorize since; these tests keep the hazard visible:
  * the FFT rows kernels must be bit-reproducible while a child process hammers the device with that GEMM tile, in the default launch
    form and in the co-tenant-safe one (MIT_COTENANT_SAFE: narrow LDS reads + a whole CU's LDS — the mitigation that shipped before
    the cause was known, kept as a switch);
  * the page engine with its stages on two streams must give the bytes of the one-stream run.
(The engine-level two-process form — two ranks running whole page engines on one GPU at the same time — is tests/test_dist_gpu.py.)"""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

AGGRESSOR = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import torch
    from manga_image_translator_amd import ops
    g = torch.Generator().manual_seed(1)
    with ops.gemm_mode(6):
        conv = ops.Conv2d(torch.randn(1280, 320, 1, 1, generator=g) * 0.05, torch.randn(1280, generator=g), act=ops.ACT_GELU, device="cuda")
        x = torch.randn(1, 256, 256, 320, generator=g).cuda()
        out = torch.empty(1, 256, 256, 1280, device="cuda")
        conv(x, out=out)            # M = 65536: the 128 x 128 split tile (split128x128x16p6o), 3 workgroups per CU
        torch.cuda.synchronize()
        print("ready", flush=True)
        t0 = time.time()
        while time.time() - t0 < float(sys.argv[1]):
            for _ in range(20):
                conv(x, out=out)
            torch.cuda.synchronize()
""") % ROOT


class _Aggressor:
    def __init__(self, seconds):
        self.p = subprocess.Popen([sys.executable, "-c", AGGRESSOR, str(seconds)], stdout=subprocess.PIPE, text=True)
        line = self.p.stdout.readline()
        if line.strip() != "ready":
            self.p.kill()
            raise RuntimeError(f"aggressor process did not start: {line!r}")

    def stop(self):
        self.p.terminate()      # this exact child, nothing matched by name
        try:
            self.p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            self.p.kill()
            self.p.wait()


def _fft_rows_roundtrip(x, tabs, B, h, w, C_):
    import ctypes as C
    import math
    from manga_image_translator_amd import lib, ops

    L = lib.load()
    wk = w // 2 + 1
    plane = h * wk * C_
    Y = torch.full((B, 2, h, wk, C_), float("nan"), device="cuda")
    out = torch.full((B, h, w, C_), float("nan"), device="cuda")
    st = C.c_void_p(ops.current_stream())
    lib.check(L.mit_rfft_rows(x.data_ptr(), h * w * C_, w * C_, C_, Y.data_ptr(), 2 * plane, plane, wk * C_, C_, tabs.data_ptr(), B, h, w, C_,
                              1.0 / math.sqrt(w), st), "mit_rfft_rows")
    U = Y.permute(0, 2, 1, 3, 4).contiguous()     # [B, h, 2, wk, C]: the layout mit_irfft_rows reads in the FourierUnit
    lib.check(L.mit_irfft_rows(U.data_ptr(), 2 * plane, wk * C_, 2 * wk * C_, C_, out.data_ptr(), h * w * C_, w * C_, C_, x.data_ptr(),
                               h * w * C_, w * C_, C_, tabs.data_ptr(), B, h, w, C_, 1.0 / math.sqrt(w), st), "mit_irfft_rows")
    torch.cuda.synchronize()
    return Y.clone(), out


def _count_disturbed(launches=60):
    from manga_image_translator_amd.lama import rfft_row_tables

    results = {}
    for w in (24, 182):     # the size of scripts/cotenant_check (radix 4, 3) and the BASELINE page's W / 8 (radix 7, 13)
        B, h, C_ = 2, 64, 192
        g = torch.Generator().manual_seed(w)
        x = torch.randn(B, h, w, C_, generator=g).cuda()
        tabs = rfft_row_tables(w).cuda()
        results[w] = (x, tabs, B, h, C_, _fft_rows_roundtrip(x, tabs, B, h, w, C_))
    agg = _Aggressor(seconds=40)
    try:
        bad = {w: 0 for w in results}
        for _ in range(launches):
            for w, (x, tabs, B, h, C_, (Y0, o0)) in results.items():
                Y, o = _fft_rows_roundtrip(x, tabs, B, h, w, C_)
                bad[w] += int(not (torch.equal(Y, Y0) and torch.equal(o, o0)))
    finally:
        agg.stop()
    return bad


def test_fft_rows_are_bit_stable_beside_a_looping_split_tile_process_in_safe_mode(cuda):
    from manga_image_translator_amd import lib

    prev = lib.load().mit_cotenant_safe_set(1)
    try:
        bad = _count_disturbed()
    finally:
        lib.load().mit_cotenant_safe_set(prev)
    assert bad == {24: 0, 182: 0}, f"launches that differ from the quiet run, per row length: {bad}"


def test_fft_rows_are_bit_stable_beside_a_looping_split_tile_process_default_mode(cuda):
    """The default launches (wide LDS reads, CUs shared with whatever else is resident): 55-58 of 60 launches were disturbed at both row
    lengths while the library was built with the SLP vectoriser; none may be now."""
    from manga_image_translator_amd import lib

    prev = lib.load().mit_cotenant_safe_set(0)
    try:
        bad = _count_disturbed()
    finally:
        lib.load().mit_cotenant_safe_set(prev)
    assert bad == {24: 0, 182: 0}, f"launches that differ from the quiet run, per row length: {bad}"


def test_two_stream_page_engine_gives_the_one_stream_bytes(cuda):
    """PageEngine(overlap=True) — LaMa on the caller's stream, detector + OCR on a second one, kernels of both resident at once —
    against the one-stream engine on the same pages: every result tensor identical, from the first call to the fourth (with the SLP
    build the OCR results differed from the second call on, and the inpainted pages too without the safe mode)."""
    import numpy as np
    from manga_image_translator_amd import pipeline, synth

    H, W, n = 1024, 728, 4
    weights = pipeline.synthetic_weights()
    gen = [synth.synth_page(40 + i, H, W, n_boxes=12) for i in range(n)]
    pages = torch.from_numpy(np.stack([g[0] for g in gen])).to(cuda)
    masks = torch.from_numpy(np.stack([g[2] for g in gen])).to(cuda)
    quads = [pipeline.quads_from_array(g[1]) for g in gen]
    one = pipeline.PageEngine(weights, device=cuda, ctd_mb=2, lama_mb=2, group=2, overlap=False)
    two = pipeline.PageEngine(weights, device=cuda, ctd_mb=2, lama_mb=2, group=2, overlap=True)
    kw = dict(max_seq_length=8, suppress_eos=True)

    def grab(r):
        torch.cuda.synchronize()
        return [t.clone() for t in (r.det_mask, r.det_shrink, r.ocr_tokens, r.ocr_prob, r.ocr_colors, r.inpainted)]

    ref = grab(one.run(pages, quads, masks, **kw))
    for call in range(4):
        got = grab(two.run(pages, quads, masks, **kw))
        for name, a, b in zip(("det_mask", "det_shrink", "ocr_tokens", "ocr_prob", "ocr_colors", "inpainted"), got, ref):
            assert torch.equal(a, b), f"two-stream call {call}: {name} differs in {int((a != b).sum())} elements"
