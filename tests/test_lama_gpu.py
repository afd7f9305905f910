"""LaMa stage parity: HIP engine (through the C-ABI) vs the CPU oracle restatement of the reference.

Tolerances (fp32 parity mode, exact-fp32 MFMA): the network is ~40-80 fp32 layers deep, so the
sigmoid output is compared at 2e-4 absolute (observed ~1e-5); the uint8 page must be identical
except where the oracle's pre-truncation value sits within 0.05 of an integer (then +-1 level).
Integer maps (MPE ring / direction indices) must be bit-exact.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_blocks, mpe, cuda, seed=0):
    from manga_image_translator_amd import lama, lama_schema, synth

    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(n_blocks), seed=seed)
    mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema(), seed=seed) if mpe else None
    eng = lama.LamaEngine(sd, mpe_sd, n_blocks=n_blocks, device=cuda)
    return sd, mpe_sd, eng


def _nchw(t):
    return t.permute(0, 3, 1, 2)


def test_fourier_unit_parity(cuda):
    from oracle import lama as OL

    sd, _, eng = _setup(1, False, cuda)
    g = torch.Generator().manual_seed(3)
    for (B, h, w) in [(2, 8, 11), (1, 33, 34), (1, 16, 23)]:
        t1 = torch.randn(B, h, w, 192, generator=g)
        t2 = torch.empty(B, h, w, 192, device=cuda)
        eng._fourier_unit(eng.blocks[0][0], t1.to(cuda), t2)
        torch.cuda.synchronize()
        x = _nchw(t1)
        ref = x + OL.fourier_unit(x, sd, "model.5.conv1.ffc.convg2g.fu")
        err = (_nchw(t2.cpu()) - ref).abs().max().item()
        assert err < 5e-5, (B, h, w, err)


@pytest.mark.parametrize("n_blocks,mpe,B,H,W", [(18, False, 2, 64, 88), (9, True, 1, 264, 272), (9, True, 2, 256, 320),
                                                 (9, True, 1, 2048, 1456),    # the BASELINE page, shipped path (Winograd FFC blocks, LDS FFT)
                                                 (18, False, 1, 512, 512)],   # lama_large at its nominal 512 px
                         ids=["large-64x88", "mpe-264x272", "mpe-256x320x2", "mpe-BASELINE-2048x1456", "large-512x512"])
def test_lama_page_parity(cuda, gemm_mode, oracle_memo, n_blocks, mpe, B, H, W):
    from manga_image_translator_amd import synth
    from oracle import lama as OL

    sd, mpe_sd, eng = _setup(n_blocks, mpe, cuda)
    pages, masks = [], []
    for i in range(B):
        p, _, m = synth.synth_page(i, H, W, n_boxes=6)
        pages.append(p)
        masks.append(m)
    masks[0][3, 5] = 127  # the mask==127 corner case of _infer :59-60 vs :86-87
    img = torch.from_numpy(np.stack(pages)).to(cuda)
    msk = torch.from_numpy(np.stack(masks)).to(cuda)
    taps = {}
    out = eng.forward(img, msk, taps=taps)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(B):
        def run_oracle(i=i):
            ot = {}
            r = OL.infer(sd, mpe_sd, pages[i], masks[i], n_blocks, ot)
            keep = ("stem", f"block{n_blocks - 1}_l", f"block{n_blocks - 1}_g", "out_float")
            return r, {k: ot[k] for k in keep}

        ref, otaps = oracle_memo(("lama", n_blocks, mpe, B, H, W, i), run_oracle)
        if mpe:
            rel, _, direct = OL.load_masked_position_encoding((masks[i].astype(np.float32) / 255.0 >= 0.5).astype(np.float32))
            ymap = np.minimum(np.floor(np.arange(H) * (256 / H)).astype(np.int64), 255)
            xmap = np.minimum(np.floor(np.arange(W) * (256 / W)).astype(np.int64), 255)
            got_rel = taps["mpe_rel"][i].cpu().numpy()[ymap][:, xmap]
            got_dir = taps["mpe_dir"][i].cpu().numpy()[ymap][:, xmap]
            inm = masks[i].astype(np.float32) / 255.0 >= 0.5
            assert np.array_equal(np.where(inm, got_rel, 0), rel), "MPE ring index mismatch"
            bits = (direct * np.array([1, 2, 4, 8])).sum(-1)
            assert np.array_equal(np.where(inm, got_dir, 0), bits), "MPE direction bits mismatch"
        stem_err = (_nchw(taps["stem"][i:i + 1].cpu()) - otaps["stem"]).abs().max().item()
        assert stem_err < 2e-5, stem_err
        last = n_blocks - 1
        blk = torch.cat([otaps[f"block{last}_l"], otaps[f"block{last}_g"]], dim=1)
        blk_err = (_nchw(taps[f"block{last}"][i:i + 1].cpu()) - blk).abs().max().item()
        assert blk_err < 2e-4 * max(1.0, blk.abs().max().item()), blk_err
        # predicted float image (sigmoid output, before composite)
        m01 = torch.from_numpy((masks[i].astype(np.float32) / 255.0 >= 0.5).astype(np.float32))[None, None]
        pred = _nchw(taps["pred"][i:i + 1].cpu())
        comp = pred * m01 + (1 - m01) * otaps["out_float"]  # outside the mask the oracle float is the page itself
        ferr = ((comp - otaps["out_float"]) * m01).abs().max().item()
        assert ferr < 2e-4, ferr
        # uint8 page
        diff = out[i].astype(np.int32) - ref.astype(np.int32)
        bad = np.argwhere(diff != 0)
        if len(bad):
            assert np.abs(diff).max() <= 1
            of = (otaps["out_float"][0].permute(1, 2, 0).numpy() * 255.0)
            frac = np.abs(of - np.round(of))
            assert all(frac[tuple(b)] < 0.05 for b in bad), "uint8 mismatch away from a truncation boundary"
            assert len(bad) < 1e-3 * diff.size
        print(f"lama {n_blocks} blocks {H}x{W} page {i} gemm mode {gemm_mode}: stem {stem_err:.2e}, last block {blk_err:.2e} (range {blk.abs().max().item():.2f}), "
              f"sigmoid {ferr:.2e}, u8 diffs {len(bad)} of {diff.size}")


def test_lama_rejects_bad_input(cuda):
    _, _, eng = _setup(1, False, cuda)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 60, 64, 3, dtype=torch.uint8, device=cuda), torch.zeros(1, 60, 64, dtype=torch.uint8, device=cuda))
    with pytest.raises(TypeError):
        eng.forward(torch.zeros(1, 64, 64, 3, device=cuda), torch.zeros(1, 64, 64, dtype=torch.uint8, device=cuda))


def test_fft_h_matches_dft_gemm_at_page_size(cuda):
    """FourierUnit at the BASELINE page's spectral size (256 x 182, 192 ch): the LDS-butterfly FFTs (mixed-radix along W,
    radix-2 along H) and the dense DFT-GEMM paths agree to fp32 round-off in every combination, and all match torch.fft
    (the reference's rfftn/irfftn) through the oracle."""
    from manga_image_translator_amd import lama, lama_schema, synth
    from oracle import lama as OL

    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(1))
    g = torch.Generator().manual_seed(11)
    t1 = torch.randn(1, 256, 182, 192, generator=g)
    outs = []
    for fft_h, fft_w in ((True, True), (False, False), (True, False), (False, True)):  # shipped path first
        eng = lama.LamaEngine(sd, None, n_blocks=1, device=cuda, fft_h=fft_h, fft_w=fft_w)
        t2 = torch.empty(1, 256, 182, 192, device=cuda)
        eng._fourier_unit(eng.blocks[0][0], t1.to(cuda), t2)
        torch.cuda.synchronize()
        outs.append(t2.cpu())
    x = _nchw(t1)
    ref = _nchw((x + OL.fourier_unit(x, sd, "model.5.conv1.ffc.convg2g.fu")).permute(0, 2, 3, 1))
    scale = ref.abs().max().item()
    for o in outs[1:]:
        assert (outs[0] - o).abs().max().item() < 2e-5 * scale
    for o in outs:
        assert (_nchw(o) - ref).abs().max().item() < 5e-5 * scale


@pytest.mark.parametrize("h,ncols,inverse", [(32, 36, False), (64, 128, True), (128, 100, False), (256, 92 * 4, False), (256, 92 * 4, True),
                                             (512, 64, True)])
def test_fft_cols_against_torch_fft(cuda, h, ncols, inverse):
    """mit_fft_cols (radix-2 graph, up to 4 stages per LDS round trip) against torch.fft in float64: every power-of-two
    height the pass splitter can see (5..9 stages = 3+2, 3+3, 4+3, 4+4, 3+3+3), ragged column counts, both directions."""
    import ctypes as C
    import math

    from manga_image_translator_amd import lib as L, ops

    g = torch.Generator().manual_seed(h + ncols)
    B = 3
    x = torch.randn(B, 2, h, ncols, generator=g)
    k = torch.arange(h // 2, dtype=torch.float64) * (2 * math.pi / h)
    tw = torch.stack([torch.cos(k), torch.sin(k)], 1).to(torch.float32).contiguous().to(cuda)
    xg, out = x.to(cuda), torch.zeros(B, 2, h, ncols, device=cuda)
    plane = h * ncols
    L.check(L.load().mit_fft_cols(xg.data_ptr(), 2 * plane, plane, ncols, out.data_ptr(), 2 * plane, plane, ncols, tw.data_ptr(), B, h,
                                  ncols, int(inverse), 1.0 / math.sqrt(h), C.c_void_p(ops.current_stream())), "mit_fft_cols")
    torch.cuda.synchronize()
    z = torch.complex(x[:, 0].double(), x[:, 1].double())
    ref = (torch.fft.ifft if inverse else torch.fft.fft)(z, dim=1, norm="ortho")
    got = torch.complex(out[:, 0].double().cpu(), out[:, 1].double().cpu())
    assert (got - ref).abs().max() < 5e-6 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("w,C_,h", [(182, 192, 5), (8, 4, 3), (42, 36, 2), (64, 32, 4), (210, 68, 2), (26, 192, 3), (364, 64, 2),
                                    (512, 40, 2), (4, 8, 1), (198, 32, 2)])
def test_rfft_rows_against_torch_fft(cuda, w, C_, h):
    """mit_rfft_rows / mit_irfft_rows (packed real FFT, Stockham stages over the radices {2,3,4,5,7,11,13}) against
    torch.fft.rfft / irfft(norm='ortho') in float64: the BASELINE width 182 = 2*7*13, every radix, ragged channel counts,
    the residual add, and Im(DC) / Im(Nyquist) ignored by the inverse like pocketfft's c2r."""
    import ctypes as C
    import math

    from manga_image_translator_amd import lama, lib as L, ops

    Lh = L.load()
    assert Lh.mit_rfft_rows_supported(w)
    g = torch.Generator().manual_seed(w * 7 + C_)
    B, wk = 2, w // 2 + 1
    x = torch.randn(B, h, w, C_, generator=g)
    tabs = lama.rfft_row_tables(w).to(cuda)
    st = C.c_void_p(ops.current_stream())
    xg = x.to(cuda)
    Y = torch.full((B, 2, h, wk, C_), float("nan"), device=cuda)
    plane = h * wk * C_
    L.check(Lh.mit_rfft_rows(xg.data_ptr(), h * w * C_, w * C_, C_, Y.data_ptr(), 2 * plane, plane, wk * C_, C_, tabs.data_ptr(), B, h, w,
                             C_, 1.0 / math.sqrt(w), st), "mit_rfft_rows")
    torch.cuda.synchronize()
    ref = torch.fft.rfft(x.double(), dim=2, norm="ortho")  # [B, h, wk, C]
    got = torch.complex(Y[:, 0].double().cpu(), Y[:, 1].double().cpu())
    tol = 5e-6 * max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max() < tol
    # inverse on a spectrum with junk in Im(DC) / Im(Nyquist), laid out [B, h, 2, wk, C] like the engine's U buffer
    spec = torch.randn(B, h, 2, wk, C_, generator=g)
    res = torch.randn(B, h, w, C_, generator=g)
    out = torch.full((B, h, w, C_), float("nan"), device=cuda)
    sg, rg = spec.to(cuda), res.to(cuda)
    for use_res in (True, False):
        L.check(Lh.mit_irfft_rows(sg.data_ptr(), h * 2 * wk * C_, wk * C_, 2 * wk * C_, C_, out.data_ptr(), h * w * C_, w * C_, C_,
                                  rg.data_ptr() if use_res else None, h * w * C_, w * C_, C_, tabs.data_ptr(), B, h, w, C_,
                                  1.0 / math.sqrt(w), st), "mit_irfft_rows")
        torch.cuda.synchronize()
        z = torch.complex(spec[:, :, 0].double(), spec[:, :, 1].double())
        ref = torch.fft.irfft(z, n=w, dim=2, norm="ortho") + (res.double() if use_res else 0.0)
        assert (out.double().cpu() - ref).abs().max() < 5e-6 * max(1.0, ref.abs().max().item())


def test_rfft_rows_rejects_unsupported_widths(cuda):
    from manga_image_translator_amd import lib as L

    Lh = L.load()
    for w in (181, 362, 2 * 17, 514, 2, 0, -4):  # odd, 2 * prime > 13, too long, too short
        assert not Lh.mit_rfft_rows_supported(w)
    assert Lh.mit_rfft_rows(None, 0, 0, 0, None, 0, 0, 0, 0, None, 1, 1, 182, 192, 1.0, None) != 0


@pytest.mark.parametrize("H,W,B", [(64, 72, 2), (256, 184, 1), (16, 24, 1)])
def test_row_packed_stem_is_bit_identical(cuda, H, W, B):
    from manga_image_translator_amd import ops

    with ops.gemm_mode(0):  # a statement about the fp32 tiles: the row-packed form (fast kernel) against the plain one (generic kernel)
        _row_packed_stem_check(cuda, H, W, B)


def _row_packed_stem_check(cuda, H, W, B):
    """The 7x7 4->64 stem as 7 taps of one contiguous 32-float read on the reflect-padded input (fast kernel) == the plain
    reflect-padded Conv2d on the generic kernel: same (ky, kx, c) accumulation order, the surplus terms are exact zeros."""
    from manga_image_translator_amd import lama, lama_schema, synth

    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(1))
    rng = np.random.default_rng(H + W)
    img = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3)).astype(np.uint8)).to(cuda)
    msk = torch.from_numpy((rng.random((B, H, W)) < 0.3).astype(np.uint8) * 255).to(cuda)
    taps = []
    for packed in (True, False):
        eng = lama.LamaEngine(sd, None, n_blocks=1, device=cuda, row_packed_stem=packed)
        t = {}
        eng.forward(img, msk, taps=t)
        torch.cuda.synchronize()
        taps.append(t)
    assert torch.equal(taps[0]["stem"], taps[1]["stem"])
    assert torch.equal(taps[0]["pred"], taps[1]["pred"])


@pytest.mark.parametrize("H,W,B", [(264, 272, 2), (512, 384, 3)])
def test_mpe_in_the_stem_epilogue_is_bit_identical_to_the_separate_pass(cuda, gemm_mode, H, W, B):
    """The masked position encoding's two adds (FFCResNetGenerator.forward, inpainting_lama_mpe.py:611-612) as the stem launch's two-table
    row lookup (MitConvGemm.lut_rows; mit_lama_mpe_rows) against the separate read-modify-write pass (mit_lama_mpe_add) they replace:
    the stem output and the final page must be the same bytes, in both GEMM modes (fp32 tiles and split tiles share the epilogue)."""
    from manga_image_translator_amd import synth

    _, _, eng = _setup(9, True, cuda)
    gen = [synth.synth_page(i, H, W, n_boxes=5) for i in range(B)]
    img = torch.from_numpy(np.stack([g[0] for g in gen])).to(cuda)
    msk = torch.from_numpy(np.stack([g[2] for g in gen])).to(cuda)
    assert eng.mpe_in_stem and msk.any()
    t1, t2 = {}, {}
    out1 = eng.forward(img, msk, taps=t1).clone()
    eng.mpe_in_stem = False
    try:
        out2 = eng.forward(img, msk, taps=t2).clone()
    finally:
        eng.mpe_in_stem = True
    torch.cuda.synchronize()
    assert torch.equal(t1["mpe_rel"], t2["mpe_rel"]) and torch.equal(t1["mpe_dir"], t2["mpe_dir"]) and t1["mpe_rel"].any() and t1["mpe_dir"].any()
    assert torch.equal(t1["stem"], t2["stem"]), f"{(t1['stem'] != t2['stem']).sum().item()} stem values differ"
    assert torch.equal(out1, out2)


def test_row_lookup_epilogue_is_validated(cuda):
    """mit_conv_gemm refuses a lookup without its tables, with rows shorter than N, or on a batched (Z > 1) launch."""
    from manga_image_translator_amd import ops

    x = torch.randn(1, 8, 8, 16, device=cuda)
    conv = ops.Conv2d(torch.randn(32, 16, 1, 1), None, device=cuda)
    out = torch.empty(1, 8, 8, 32, device=cuda)
    rows = torch.zeros(64, dtype=torch.int32, device=cuda)
    t = torch.zeros(4, 32, device=cuda)
    d = conv.desc(x, out)
    d.lut_rows, d.lut1, d.lut2, d.lut_ld = rows.data_ptr(), t.data_ptr(), 0, 32
    with pytest.raises(RuntimeError, match="lut_rows needs"):
        ops.launch_conv_gemm(d)
    d.lut2, d.lut_ld = t.data_ptr(), 16
    with pytest.raises(RuntimeError, match="lut_rows needs"):
        ops.launch_conv_gemm(d)
    d.lut_ld = 32
    ops.launch_conv_gemm(d)   # valid: all rows 0 of zero tables
    torch.cuda.synchronize()
    d.post = ops.tensor_map(out)
    with pytest.raises(RuntimeError, match="post residual"):
        ops.launch_conv_gemm(d)
    # a layer that takes the generic kernel (Cin = 4): the lookup exists only as an instantiation of the fast / split tiles' epilogue
    x4 = torch.randn(1, 8, 8, 4, device=cuda)
    conv4 = ops.Conv2d(torch.randn(32, 4, 1, 1), None, device=cuda)
    d4 = conv4.desc(x4, out)
    d4.lut_rows, d4.lut1, d4.lut2, d4.lut_ld = rows.data_ptr(), t.data_ptr(), t.data_ptr(), 32
    with pytest.raises(RuntimeError, match="fast or split tile"):
        ops.launch_conv_gemm(d4)


def test_planar_tail_is_bit_identical_to_the_nhwc_tail(cuda, gemm_mode):
    """The last up-convolution writing four 16-channel planes (column-split output map on the float4 epilogue) and the 7x7 output
    convolution reading them group by group against the NHWC tensor between the two: same values at other addresses — the page must
    be the same bytes; and each half against its NHWC form in isolation."""
    from manga_image_translator_amd import ops, synth

    _, _, eng = _setup(9, True, cuda)
    gen = [synth.synth_page(i, 264, 272, n_boxes=5) for i in range(2)]
    img = torch.from_numpy(np.stack([g[0] for g in gen])).to(cuda)
    msk = torch.from_numpy(np.stack([g[2] for g in gen])).to(cuda)
    assert eng.planar_tail == 4        # the shipped tail: sixteen 4-channel planes, the output convolution's LDS-DMA kernel (round 6)
    t1, t2, t3 = {}, {}, {}
    out1 = eng.forward(img, msk, taps=t1).clone()
    try:
        eng.planar_tail = 0            # NHWC between the two
        out2 = eng.forward(img, msk, taps=t2).clone()
        eng.planar_tail = 16           # four 16-channel planes, the register-staged kernel (round 5)
        out3 = eng.forward(img, msk, taps=t3).clone()
    finally:
        eng.planar_tail = 4
    torch.cuda.synchronize()
    assert torch.equal(t1["pred"], t2["pred"]) and torch.equal(out1, out2)
    assert torch.equal(t1["pred"], t3["pred"]) and torch.equal(out1, out3)
    # the two halves alone
    g = torch.Generator().manual_seed(5)
    up = eng.ups[2]
    x = torch.randn(2, 20, 28, up.Cin, generator=g).to(cuda)
    nhwc = up(x)
    planes = up(x, planes=4)
    assert planes.shape == (4, 2, 40, 56, 16) and torch.equal(planes.permute(1, 2, 3, 0, 4).reshape(2, 40, 56, 64), nhwc)
    planes4 = up(x, planes=16)
    assert planes4.shape == (16, 2, 40, 56, 4) and torch.equal(planes4.permute(1, 2, 3, 0, 4).reshape(2, 40, 56, 64), nhwc)
    o1 = torch.empty(2, 40, 56, 3, device=cuda)
    o2 = torch.empty(2, 40, 56, 3, device=cuda)
    o3 = torch.empty(2, 40, 56, 3, device=cuda)
    eng.out_conv(nhwc, out=o1)
    eng.out_conv(planes, out=o2)
    eng.out_conv(planes4, out=o3)      # tile overhang on both axes (40 x 56 against 16 x 64 tiles), reflection on all four sides
    pm = up(x, planes=16, parity_major=True)   # every plane as four dense parity sub-images [2][2][20][28][4]
    want = planes4.view(16, 2, 20, 2, 28, 2, 4).permute(0, 1, 3, 5, 2, 4, 6).reshape(16, 2, 40, 56, 4)
    o4 = torch.empty(2, 40, 56, 3, device=cuda)
    eng.out_conv(pm, out=o4, parity_major=True)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(o1, o3) and torch.equal(pm, want) and torch.equal(o1, o4)
