"""Kernels of this library next to ANOTHER queue's kernels on the same CUs (a second process on the device, a second stream).

Round 3 found mit_rfft_rows / mit_irfft_rows returning wrong workgroups (7-15 of 60 launches) while another process looped the 128 x 128
split-bf16 GEMM tile; round 4 traced it to the two-address 8-byte LDS reads (ds_read2_b64 / ds_read2st64_b64) hipcc had merged the
butterfly inputs into — a co-resident kernel that mixes MFMAs with LDS traffic disturbs exactly those — and the co-tenant-safe launches (MIT_COTENANT_SAFE /
mit_cotenant_safe_set) read the pairs with 4-byte loads AND take a whole CU's LDS (csrc/fft_rows.hip; scripts/cotenant_check,
profiles/r04b_cotenant_check.log).
This test keeps the hazard visible: the FFT rows kernels must be bit-reproducible while a child process hammers the device with that GEMM
tile (the engine-level form — two ranks running whole page engines on one GPU at the same time — is tests/test_dist_gpu.py)."""
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


def test_fft_rows_beside_a_looping_split_tile_process_default_mode(cuda):
    """Default launches (one queue per GPU is the supported configuration: wide LDS reads, the FFT rows kernels share their CUs): beside
    a co-resident MFMA + LDS kernel of another process their results are disturbed (round 4: 55-58 of 60 launches at both row lengths) —
    the open hazard, recorded as an expected failure so that a fix shows up as XPASS."""
    from manga_image_translator_amd import lib

    prev = lib.load().mit_cotenant_safe_set(0)
    try:
        bad = _count_disturbed()
    finally:
        lib.load().mit_cotenant_safe_set(prev)
    if any(bad.values()):
        pytest.xfail(f"mit_rfft_rows / mit_irfft_rows beside a co-resident MFMA + LDS kernel: launches of 60 that differ, per row length: {bad} (DESIGN §7)")
