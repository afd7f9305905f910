"""Parity of mit_conv_gemm (through the C-ABI) against torch CPU references.

Tolerance: the MFMA path is an exact-fp32 k-ordered fmaf chain, so the only difference from
the float64 reference is fp32 round-off: |err| <= 2e-6 * sum_k |a_k*w_k| (+ a small epilogue
term).  That bound is asserted elementwise.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_mfma_tiles():
    """These are tests of the fp32 tiles (bit-equality between tile shapes, the generic and the fast kernel): the layers are packed and
    launched in GEMM mode 0.  The split-bf16 tiles have their own file (test_gemm_split_gpu.py)."""
    from manga_image_translator_amd import ops

    with ops.gemm_mode(0):
        yield



def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _pad(x, p, mode):
    if p == 0:
        return x
    return F.pad(x, (p, p, p, p), mode="reflect" if mode == "reflect" else "constant")


def _act(v, act, alpha):
    from manga_image_translator_amd import ops
    return {
        ops.ACT_NONE: lambda t: t,
        ops.ACT_RELU: torch.relu,
        ops.ACT_LEAKY: lambda t: F.leaky_relu(t, alpha),
        ops.ACT_SILU: F.silu,
        ops.ACT_SIGMOID: torch.sigmoid,
        ops.ACT_GELU: F.gelu,
    }[act](v)


def _tile(name):
    from manga_image_translator_amd import lib
    L, i = lib.load(), 0
    while L.mit_conv_gemm_config_name(i) is not None:
        if L.mit_conv_gemm_config_name(i).decode() == name:
            return i
        i += 1
    raise KeyError(name)


CASES = [
    # B, Cin, Cout, H, W, k, s, p, mode, act, bn, tile (name; None = automatic)
    (1, 16, 32, 17, 23, 3, 1, 1, "zero", 0, False, None),
    (2, 128, 128, 20, 26, 3, 1, 1, "reflect", 1, True, None),
    (1, 384, 128, 24, 18, 3, 1, 1, "reflect", 1, True, "128x128x16"),
    (1, 128, 384, 24, 18, 3, 1, 1, "reflect", 0, False, "128x128x16"),
    (1, 4, 64, 40, 36, 7, 1, 3, "reflect", 1, True, None),
    (1, 64, 3, 33, 29, 7, 1, 3, "reflect", 4, False, None),
    (2, 64, 128, 32, 28, 3, 2, 1, "reflect", 1, True, None),
    (1, 3, 32, 64, 48, 6, 2, 2, "zero", 3, False, None),
    (1, 256, 256, 16, 16, 1, 1, 0, "zero", 2, True, "128x128x16"),
    (3, 192, 384, 9, 7, 1, 1, 0, "zero", 0, False, "128x64x16"),
    (1, 64, 16, 31, 17, 3, 1, 1, "zero", 1, True, "128x32x16"),
    (1, 80, 320, 12, 40, 1, 1, 0, "zero", 5, False, "128x64x16"),
    (1, 32, 64, 8, 8, 3, 1, 1, "zero", 0, False, "128x64x16"),
    (1, 160, 160, 6, 33, (2, 1), (2, 1), 0, "zero", 1, True, None),
    (2, 64, 1, 20, 24, 3, 1, 1, "zero", 4, True, None),      # N = 1: conv_gemv_kernel, 16 lanes per row
    (1, 16, 1, 19, 21, 1, 1, 0, "zero", 4, False, None),     # N = 1, Cin = 16: 4 lanes per row
    (1, 16, 2, 15, 18, 3, 1, 1, "reflect", 1, True, "gemv4"),   # N = 2 forced onto gemv4
    (1, 128, 4, 9, 40, 3, 2, 1, "zero", 2, False, "gemv16"),     # N = 4, stride 2, forced onto gemv16
    (2, 320, 320, 7, 33, 1, 1, 0, "zero", 5, False, "fast64x64x16w8c"),   # the 64 x 64 fast tile (decoder-sized GEMM), GELU
    (1, 64, 96, 12, 10, 3, 1, 1, "reflect", 1, True, "fast64x64x16w8c"),  # 64 x 64 fast tile, 3x3 reflect, ragged N
    (1, 64, 32, 31, 17, 3, 1, 1, "zero", 1, True, "fast128x32x16w4c"),     # 128 x 32 fast tile (ESRGAN's growth-32 convolutions)
    (2, 160, 32, 12, 10, 3, 1, 1, "zero", 2, False, "fast128x32x16w4c"),
    (1, 64, 16, 9, 40, 3, 1, 1, "reflect", 0, False, "fast128x32x16w4c"),  # ragged N on it
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv2d_parity(cuda, case):
    from manga_image_translator_amd import ops

    B, Cin, Cout, H, W, k, s, p, mode, act, bn, cfg = case
    kh, kw = (k, k) if isinstance(k, int) else k
    g = torch.Generator().manual_seed(1234 + Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, kh, kw, generator=g) / (Cin * kh * kw) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    bn_t = None
    if bn:
        bn_t = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
                torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-5)
    alpha = 0.1
    layer = ops.Conv2d(w, b, stride=s, padding=p, pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO,
                       bn=bn_t, act=act, alpha=alpha, device=cuda)
    cin_p = layer.Cin
    xg = torch.zeros(B, H, W, cin_p)
    xg[..., :Cin] = _nhwc(x)
    xg = xg.to(cuda)
    post = torch.randn(B, Cout, *layer.out_hw(H, W), generator=g)
    out = layer(xg, post=_nhwc(post).to(cuda), cfg=-1 if cfg is None else _tile(cfg))
    torch.cuda.synchronize()

    xd, wd = x.double(), w.double()
    pp = p if isinstance(p, int) else p
    ref = F.conv2d(_pad(xd, pp, mode), wd, b.double(), stride=s)
    mag = F.conv2d(_pad(xd.abs(), pp, mode), wd.abs(), b.abs().double(), stride=s)
    if bn:
        gam, bet, mu, var, eps = [t.double() if torch.is_tensor(t) else t for t in bn_t]
        sc = (gam / torch.sqrt(var + eps)).view(1, -1, 1, 1)
        ref = (ref - mu.view(1, -1, 1, 1)) * sc + bet.view(1, -1, 1, 1)
        mag = (mag + mu.abs().view(1, -1, 1, 1)) * sc.abs() + bet.abs().view(1, -1, 1, 1)
    ref = _act(ref, act, alpha) + post.double()
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    bound = 2e-6 * (mag + post.abs().double()) + 1e-6
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"max err/bound = {worst:.3f}, max abs err {err.max().item():.3e}"


TCASES = [
    # B, Cin, Cout, H, W, k, s, p, op, act, bn
    (1, 64, 32, 9, 11, 3, 2, 1, 1, 1, True),
    (2, 128, 64, 8, 8, 4, 2, 1, 0, 1, True),
    (1, 16, 16, 13, 7, 2, 2, 0, 0, 1, True),
    (1, 16, 1, 10, 12, 2, 2, 0, 0, 4, False),
    (1, 64, 1, 12, 9, 4, 2, 1, 0, 4, False),
]


@pytest.mark.parametrize("case", TCASES, ids=[f"t{i}" for i in range(len(TCASES))])
def test_conv_transpose2d_parity(cuda, case):
    from manga_image_translator_amd import ops

    B, Cin, Cout, H, W, k, s, p, op, act, bn = case
    g = torch.Generator().manual_seed(99 + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) / (Cin * k * k / (s * s)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    bn_t = None
    if bn:
        bn_t = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1,
                torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5, 1e-5)
    layer = ops.ConvTranspose2d(w, b, stride=s, padding=p, output_padding=op, bn=bn_t, act=act, alpha=0.1, device=cuda)
    out = layer(_nhwc(x).to(cuda))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=s, padding=p, output_padding=op)
    mag = F.conv_transpose2d(x.double().abs(), w.double().abs(), b.double().abs(), stride=s, padding=p, output_padding=op)
    if bn:
        gam, bet, mu, var, eps = [t.double() if torch.is_tensor(t) else t for t in bn_t]
        sc = (gam / torch.sqrt(var + eps)).view(1, -1, 1, 1)
        ref = (ref - mu.view(1, -1, 1, 1)) * sc + bet.view(1, -1, 1, 1)
        mag = (mag + mu.abs().view(1, -1, 1, 1)) * sc.abs() + bet.abs().view(1, -1, 1, 1)
    ref = _act(ref, act, 0.1)
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    worst = (err / (2e-6 * mag + 1e-6)).max().item()
    assert worst <= 1.0, f"max err/bound = {worst:.3f}"


def test_conv_channel_slices_and_pre(cuda):
    """Channel-sliced input/output views and the pre-activation addend (FFC wiring)."""
    from manga_image_translator_amd import ops

    g = torch.Generator().manual_seed(7)
    B, H, W = 2, 14, 10
    x = torch.randn(B, H, W, 512, generator=g)
    w = torch.randn(384, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
    pre = torch.randn(B, H, W, 384, generator=g)
    layer = ops.Conv2d(w, None, padding=1, pad_mode=ops.PAD_REFLECT, act=ops.ACT_RELU, device=cuda)
    xg = x.to(cuda)
    outbuf = torch.full((B, H, W, 512), 7.0, device=cuda)
    layer(xg[..., :128], out=outbuf[..., 128:], pre=pre.to(cuda))
    torch.cuda.synchronize()
    xin = x[..., :128].permute(0, 3, 1, 2).double()
    ref = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w.double())
    ref = torch.relu(ref + pre.permute(0, 3, 1, 2).double())
    got = outbuf.cpu()
    assert torch.all(got[..., :128] == 7.0)
    err = (got[..., 128:].permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 2e-5, err


def test_bad_descriptor_raises(cuda):
    from manga_image_translator_amd import ops

    w = torch.randn(8, 8, 1, 1)
    layer = ops.Conv2d(w, device=cuda)
    x = torch.zeros(1, 4, 4, 8, device=cuda)
    d = layer.desc(x, torch.empty(1, 4, 4, 8, device=cuda))
    d.Cin = 6
    with pytest.raises(RuntimeError, match="Cin"):
        ops.launch_conv_gemm(d)


@pytest.mark.parametrize("cout,cin,k,mode,act", [(3, 64, 7, "reflect", "sigmoid"), (1, 16, 3, "zero", "relu"), (4, 32, 5, "reflect", "none")])
def test_conv_small_cout(cuda, cout, cin, k, mode, act):
    """mit_conv_small_cout (VALU direct conv for <= 4 output channels) vs torch conv2d on the CPU; sizes that are not
    multiples of the 32x8 tile exercise the overhang, reflect padding at every border."""
    import torch.nn.functional as F

    from manga_image_translator_amd import ops

    g = torch.Generator().manual_seed(7)
    B, H, W = 2, 19, 45
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, H, W, generator=g)
    layer = ops.ConvSmallCout(w, b, pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO,
                              act={"sigmoid": ops.ACT_SIGMOID, "relu": ops.ACT_RELU, "none": ops.ACT_NONE}[act], device=cuda)
    wide = torch.full((B, H, W, cout + 2), 7.0, device=cuda)  # output is a channel slice: neighbours must stay untouched
    out = wide[..., :cout]
    layer(x.permute(0, 2, 3, 1).contiguous().to(cuda), out=out)
    torch.cuda.synchronize()
    r = k // 2
    xp = F.pad(x, (r, r, r, r), mode="reflect") if mode == "reflect" else F.pad(x, (r, r, r, r))
    ref = F.conv2d(xp, w, b)
    ref = {"sigmoid": torch.sigmoid, "relu": torch.relu, "none": lambda v: v}[act](ref).permute(0, 2, 3, 1)
    assert (out.cpu() - ref).abs().max().item() < 2e-5
    assert (wide[..., cout:] == 7.0).all()
    with pytest.raises(ValueError):
        ops.ConvSmallCout(torch.zeros(5, 16, 3, 3), device=cuda)


def test_large_batch_is_split_for_the_fast_kernel(cuda):
    """A batch whose activations exceed 2^31 elements (LaMa's first stride-2 conv at 16 pages: 3.05 G floats) is cut into runs of
    whole images for the 32-bit-offset fast kernel instead of dropping to the generic one; the k-sequential accumulation makes
    the two bit-identical."""
    from manga_image_translator_amd import ops

    B, H, W, Cin, Cout = 12, 2048, 1456, 64, 128
    assert B * H * W * Cin > 2 ** 31
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.randn(B, H, W, Cin, device=cuda, generator=g)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    layer = ops.Conv2d(w, None, stride=2, padding=1, pad_mode=ops.PAD_REFLECT, act=ops.ACT_RELU, device=cuda)
    fast = layer(x)                 # auto: split + fast tile
    generic = layer(x, cfg=_tile("128x128x16"))       # the generic 128x128 kernel on the whole batch
    torch.cuda.synchronize()
    assert torch.equal(fast, generic)
    assert fast[-1].abs().sum().item() > 0   # the last run of images was written


def test_narrow_fast_tile_is_bit_identical_to_the_generic_kernel(cuda):
    """fast128x32x16w4c (the automatic choice for N <= 32) against the generic 128x32 tile it replaces: same k-sequential chain."""
    from manga_image_translator_amd import lib as L, ops

    lib = L.load()
    names = {lib.mit_conv_gemm_config_name(i).decode(): i for i in range(64) if lib.mit_conv_gemm_config_name(i)}
    g = torch.Generator().manual_seed(11)
    for (B, Cin, Cout, H, W, k) in ((2, 96, 32, 20, 24, 3), (1, 192, 32, 9, 33, 3), (1, 64, 24, 16, 16, 1)):
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        layer = ops.Conv2d(w, torch.randn(Cout, generator=g), padding=k // 2, act=ops.ACT_LEAKY, alpha=0.2, device=cuda)
        x = torch.randn(B, H, W, Cin, generator=g).to(cuda)
        auto = layer(x)
        fast = layer(x, cfg=names["fast128x32x16w4c"])
        generic = layer(x, cfg=names["128x32x16"])
        torch.cuda.synchronize()
        assert torch.equal(fast, generic) and torch.equal(auto, generic)
