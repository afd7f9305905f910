"""mit_pgemm (csrc/pgemm.hip): plain GEMMs on operands that arrive as three bf16 planes.

The arithmetic is that of the split-bf16 tiles of mit_conv_gemm (same plane pairs, same order, fp32 accumulation in the MFMA), so the
bar is bit-identity with ``split128x128x16p6o`` / ``split128x64x16p6o`` on the same operands — for fp32 output directly, for planar output
against ``split_planes`` of that result — plus the exactness of the plane split itself.  scripts/pgemm_check.cpp is the torch-free twin of
these tests (it also times every tile); profiles/r04*_pgemm_check.log hold its output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(name):
    from manga_image_translator_amd import lib
    L, i = lib.load(), 0
    while True:
        n = L.mit_conv_gemm_config_name(i)
        if n is None:
            raise KeyError(name)
        if n.decode() == name:
            return i
        i += 1


def _tiles():
    from manga_image_translator_amd import lib
    L, i, out = lib.load(), 0, []
    while L.mit_pgemm_tile_name(i) is not None:
        out.append(L.mit_pgemm_tile_name(i).decode())
        i += 1
    return out


def test_split_planes_is_exact_and_joins_back(cuda):
    from manga_image_translator_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 200, generator=g) * torch.logspace(-20, 20, 200)   # 40 decades of magnitude across the columns
    x[::13] = 0.0
    xd = x.cuda()
    pl = ops.split_planes(xd)
    assert tuple(pl.shape) == (3, 25, 1000, 8)
    assert torch.equal(ops.join_planes(pl).cpu(), x)
    # plane 0 is the round-to-nearest-even bf16 of x, plane 1 that of the (exact) residual
    hi = pl[0].cpu().view(torch.bfloat16).float().permute(1, 0, 2).reshape(1000, 200)
    assert torch.equal(hi, x.to(torch.bfloat16).float())
    mid = pl[1].cpu().view(torch.bfloat16).float().permute(1, 0, 2).reshape(1000, 200)
    assert torch.equal(mid, (x - hi).to(torch.bfloat16).float())
    # a row-strided view (a channel slice of a wider tensor) gives the planes of the slice
    wide = torch.randn(64, 96, generator=g).cuda()
    assert torch.equal(ops.join_planes(ops.split_planes(wide[:, 32:64])), wide[:, 32:64])


CASES = [
    # M, K, N, act, pre, post, post_first
    (1000, 48, 200, 1, False, True, False),     # ragged M and N, three K-tiles
    (130, 16, 8, 0, False, False, False),       # one K-tile, one narrow tile
    (4096, 320, 1280, 5, False, False, False),  # ConvNeXt pw1 (gelu)
    (4100, 1280, 320, 0, False, True, False),   # ConvNeXt pw2 (+ residual), M not a multiple of the tile
    (9000, 192, 384, 1, True, True, False),     # spectral conv2: pre + relu + post
    (3000, 256, 256, 1, False, True, True),     # residual joins before the activation
    (2048, 320, 6004, 0, False, False, False),  # N % 8 != 0 (logits)
    (160, 320, 960, 0, False, False, False),    # one page's decoder: q | k | v projection (20 k steps: the unrolled few-row kernel)
    (160, 2048, 320, 0, False, True, False),    # ... ff2 + residual (128 k steps)
    (150, 320, 2048, 1, False, False, False),   # ... ff1 (relu), rows not a multiple of 32
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}")
def test_pgemm_equals_the_split_tiles_bit_for_bit(cuda, case):
    from manga_image_translator_amd import ops
    from manga_image_translator_amd.ocr48 import Linear

    M, K, N, act, has_pre, has_post, post_first = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g)
    x[:, ::7] *= 4.0
    lin = Linear(torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g) * 0.1, "cuda", col_scale=1.0 + 0.1 * torch.randn(N, generator=g))
    assert ops.register_split(lin.w, force=True) is not None
    xd = x.cuda()
    pre = torch.randn(M, N, generator=g).cuda() if has_pre else None
    post = torch.randn(M, N, generator=g).cuda() if has_post else None
    flags = act | (ops.ACT_POST_FIRST if post_first else 0)
    r = N % 128
    narrow = N <= 64 or (r != 0 and r <= 64)
    ref = torch.empty(M, N, device="cuda")
    cm = ops.MitTensorMap()
    cm.base, cm.xs = ref.data_ptr(), N
    maps = {}
    for t, nm in ((pre, "pre"), (post, "post")):
        if t is not None:
            maps[nm] = ops.MitTensorMap()
            maps[nm].base, maps[nm].xs = t.data_ptr(), N
    d = ops.conv_gemm_desc(a=xd, NB=1, Hi=1, Wi=M, Cin=K, a_strides=(0, 0, K), Ho=1, Wo=M, sy=1, sx=1, taps=[(0, 0, 0)], pad_mode=ops.PAD_ZERO,
                           w=lin.w, ldw=lin.Np, Kw=lin.Kp, Nw=lin.Np, N=N, c=cm, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, **maps)
    ops.launch_conv_gemm(d, _cfg("split128x64x16p6o" if narrow else "split128x128x16p6o"))
    apl = ops.split_planes(xd)
    planar_ok = N % 8 == 0 and pre is None and post is None
    ref_planes = ops.split_planes(ref) if planar_ok else None
    ran = 0
    for name in _tiles():
        if "p9" in name or name.startswith("x"):
            continue
        tile = ops.pgemm_tile(name)
        if name[-1] in "PQ":
            if not planar_ok:
                continue
            got = torch.full((3, N // 8, M, 8), -1, dtype=torch.int16, device="cuda")
            ops.pgemm(apl, lin.w, N, out_planes=got, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, nprod=6, tile=tile)
            assert torch.equal(got, ref_planes), name
        else:
            got = torch.full((M, N), float("nan"), device="cuda")
            ops.pgemm(apl, lin.w, N, out=got, pre=pre, post=post, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, nprod=6, tile=tile)
            assert torch.equal(got, ref), name
        ran += 1
    assert ran >= 4
    # the automatic tile choice, and the nine-pair form against the nine-pair split tile
    got = torch.empty(M, N, device="cuda")
    ops.pgemm(apl, lin.w, N, out=got, pre=pre, post=post, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, nprod=6)
    assert torch.equal(got, ref)
    if not narrow:
        ops.launch_conv_gemm(d, _cfg("split128x128x16p9m"))
        ref9 = ref.clone()
        ops.pgemm(apl, lin.w, N, out=got, pre=pre, post=post, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, nprod=9)
        assert torch.equal(got, ref9)
        got.fill_(float("nan"))
        ops.pgemm(apl, lin.w, N, out=got, pre=pre, post=post, scale=lin.scale, bias=lin.bias, act=flags, alpha=0.1, nprod=9, tile=ops.pgemm_tile("pgrows32d6p9"))
        assert torch.equal(got, ref9)


def test_pgemm_in_place_residual_and_chained_planar_layers(cuda):
    """x <- x + W2 gelu(W1 x) with the hidden tensor only ever in planar form (the ConvNeXt block's two pointwise layers,
    ocr/model_48px.py:203-214), in place on x like the engine does, against the same chain on the split tiles of mit_conv_gemm."""
    from manga_image_translator_amd import ops
    from manga_image_translator_amd.ocr48 import Linear

    M, C = 5000, 160
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, C, generator=g).cuda()
    l1 = Linear(torch.randn(4 * C, C, generator=g) * 0.05, torch.randn(4 * C, generator=g) * 0.1, "cuda")
    l2 = Linear(torch.randn(C, 4 * C, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1, "cuda", col_scale=0.1 * torch.randn(C, generator=g))
    for l in (l1, l2):
        assert ops.register_split(l.w, force=True) is not None
    with ops.gemm_mode(6):
        h = torch.empty(M, 4 * C, device="cuda")
        want = x.clone()
        l1(x, h, act=ops.ACT_GELU)
        l2(h, want, post=want)
    hp = torch.empty(3, 4 * C // 8, M, 8, dtype=torch.int16, device="cuda")
    ops.pgemm(ops.split_planes(x), l1.w, 4 * C, out_planes=hp, bias=l1.bias, act=ops.ACT_GELU, nprod=6)
    got = x.clone()
    ops.pgemm(hp, l2.w, C, out=got, post=got, scale=l2.scale, bias=l2.bias, nprod=6)
    assert torch.equal(got, want)
