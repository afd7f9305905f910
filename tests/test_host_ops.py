"""Host logic of the operators (no GPU): the descriptors built by ops.Conv2d / ops.ConvTranspose2d / fold_bn, evaluated by
the numpy descriptor emulator (tests/_desc_emulator.py, the documented semantics of include/mit_hip.h), reproduce
torch's conv2d / conv_transpose2d + eval BatchNorm + activation on the CPU.  The same descriptors are what the GPU
kernel receives, so this pins packing, tap tables, padding modes, strided views and epilogue folding on their own."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _desc_emulator as EMU
from manga_image_translator_amd import ops


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _bn_params(c, g):
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
            torch.rand(c, generator=g) + 0.5, 1e-5)


@pytest.mark.parametrize("cin,cout,k,s,p,mode,act", [
    (4, 8, 3, 1, 1, ops.PAD_REFLECT, ops.ACT_RELU), (8, 5, 3, 2, 1, ops.PAD_REFLECT, ops.ACT_NONE),
    (3, 6, 7, 1, 3, ops.PAD_REFLECT, ops.ACT_SIGMOID), (3, 8, 6, 2, 2, ops.PAD_ZERO, ops.ACT_SILU),
    (12, 7, 1, 1, 0, ops.PAD_ZERO, ops.ACT_LEAKY), (4, 4, (3, 1), (2, 1), 0, ops.PAD_ZERO, ops.ACT_GELU)])
def test_conv2d_descriptor(cin, cout, k, s, p, mode, act):
    g = torch.Generator().manual_seed(1)
    kh, kw = (k, k) if isinstance(k, int) else k
    w = torch.randn(cout, cin, kh, kw, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    bn = _bn_params(cout, g)
    x = torch.randn(2, cin, 9, 11, generator=g)
    layer = ops.Conv2d(w, b, stride=s, padding=p, pad_mode=mode, bn=bn, act=act, alpha=0.1, device="cpu")
    xin = _nhwc(x)
    if layer.Cin != cin:
        xin = torch.cat([xin, torch.zeros(*xin.shape[:3], layer.Cin - cin)], dim=-1).contiguous()
    Ho, Wo = layer.out_hw(9, 11)
    out = torch.full((2, Ho, Wo, cout), float("nan"))
    post = torch.randn(2, Ho, Wo, cout, generator=g)
    EMU.run(layer.desc(xin, out, post=post))
    pad = (p, p) if isinstance(p, int) else p
    xp = F.pad(x, (pad[1], pad[1], pad[0], pad[0]), mode="reflect") if (mode == ops.PAD_REFLECT and max(pad) > 0) else x
    ref = F.conv2d(xp, w, b, stride=s, padding=0 if mode == ops.PAD_REFLECT else p)
    ref = F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, bn[4])
    ref = {ops.ACT_RELU: torch.relu, ops.ACT_NONE: lambda v: v, ops.ACT_SIGMOID: torch.sigmoid, ops.ACT_SILU: F.silu,
           ops.ACT_LEAKY: lambda v: F.leaky_relu(v, 0.1), ops.ACT_GELU: F.gelu}[act](ref)
    ref = _nhwc(ref) + post
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5), (out - ref).abs().max()


@pytest.mark.parametrize("k,s,p,op", [(3, 2, 1, 1), (4, 2, 1, 0), (2, 2, 0, 0)])
def test_conv_transpose_subpixel_descriptors(k, s, p, op):
    g = torch.Generator().manual_seed(2)
    cin, cout = 8, 6
    w = torch.randn(cin, cout, k, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    bn = _bn_params(cout, g)
    x = torch.randn(2, cin, 5, 7, generator=g)
    layer = ops.ConvTranspose2d(w, b, stride=s, padding=p, output_padding=op, bn=bn, act=ops.ACT_RELU, device="cpu")
    Ho, Wo = layer.out_hw(5, 7)
    big = torch.full((2, Ho, Wo, cout + 3), float("nan"))  # output is a channel slice of a wider (concat) buffer
    out = big[..., 2:2 + cout]
    xin = _nhwc(x)  # keep alive: descriptors hold raw pointers
    for d in layer.descs(xin, out):
        EMU.run(d)
    ref = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op)
    ref = torch.relu(F.batch_norm(ref, bn[2], bn[3], bn[0], bn[1], False, 0.0, bn[4]))
    assert torch.allclose(out, _nhwc(ref), atol=2e-5, rtol=1e-5)
    assert torch.isnan(big[..., :2]).all() and torch.isnan(big[..., 2 + cout:]).all(), "neighbouring channels untouched"


def test_fold_bn_matches_batchnorm():
    g = torch.Generator().manual_seed(3)
    bn = _bn_params(16, g)
    x = torch.randn(4, 16, 3, 3, generator=g)
    cb = torch.randn(16, generator=g)
    sc, bi = ops.fold_bn(*bn, conv_bias=cb)
    ref = F.batch_norm(x + cb[None, :, None, None], bn[2], bn[3], bn[0], bn[1], False, 0.0, bn[4])
    assert torch.allclose(x * sc[None, :, None, None] + bi[None, :, None, None], ref, atol=1e-5)


def test_descriptor_validation_errors():
    with pytest.raises(TypeError):
        ops.tensor_map(torch.zeros(1, 2, 2, 4, dtype=torch.float64))
    with pytest.raises(ValueError):
        ops.tensor_map(torch.zeros(1, 4, 2, 2).permute(0, 2, 3, 1)[..., ::2])
    layer = ops.Conv2d(torch.zeros(4, 8, 1, 1), device="cpu")
    with pytest.raises(ValueError):
        layer.desc(torch.zeros(1, 2, 2, 4), torch.zeros(1, 2, 2, 4))
    with pytest.raises(ValueError):
        ops.ConvTranspose2d(torch.zeros(6, 4, 2, 2), device="cpu")


def test_lama_dft_matrices_are_the_ortho_rfft2():
    """The FourierUnit's rfftn/irfftn (inpainting_lama_mpe.py:228,252) as dense DFT matrices: F1/G2 reproduce
    torch.fft.rfftn(norm='ortho') and G2i/Fi its inverse, for an odd and an even, non-power-of-two width."""
    from manga_image_translator_amd.lama import dft_matrices

    for h, w in ((8, 11), (6, 14)):
        F1, G2, G2i, Fi = (m.double() for m in dft_matrices(h, w))
        wk = w // 2 + 1
        x = torch.randn(h, w, dtype=torch.float64, generator=torch.Generator().manual_seed(h * w))
        Y = F1[:, :w] @ x.t()                     # [(t,kw), h]
        Yre, Yim = Y[:wk].t(), Y[wk:].t()          # [h, wk]
        Z = G2[:, :2 * h] @ torch.cat([Yre, Yim], 0)  # [(t',kh), wk]
        ref = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")
        assert torch.allclose(Z[:h], ref.real, atol=1e-6) and torch.allclose(Z[h:], ref.imag, atol=1e-6)
        U = G2i[:, :2 * h] @ Z                     # [(t,h), wk]
        back = torch.cat([U[:h], U[h:]], 1) @ Fi[:, :2 * wk].t()  # [h, w]
        assert torch.allclose(back, x, atol=1e-6)


def test_upsample_conv_parity_descriptors():
    """ops.UpsampleConv2d (nearest x2 + 3x3 conv as four merged 2x2 parity convs on the low-res input) vs
    conv2d(interpolate(x)) — ESRGAN's upconv_block (upscaling/esrgan_pytorch.py:317-324)."""
    g = torch.Generator().manual_seed(4)
    cin, cout = 8, 6
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(2, cin, 5, 7, generator=g)
    layer = ops.UpsampleConv2d(w, b, act=ops.ACT_LEAKY, alpha=0.2, device="cpu")
    xin = _nhwc(x)
    out = torch.full((2, 10, 14, cout), float("nan"))
    for d in layer.descs(xin, out):
        EMU.run(d)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1), 0.2)
    assert torch.allclose(out, _nhwc(ref), atol=2e-5, rtol=1e-5), (out - _nhwc(ref)).abs().max()
    with pytest.raises(ValueError):
        ops.UpsampleConv2d(torch.zeros(4, 4, 5, 5), device="cpu")


def test_post_before_activation_epilogue():
    """MIT_ACT_POST_FIRST: relu(bn(conv) + identity) — torchvision BasicBlock's tail — in one descriptor."""
    g = torch.Generator().manual_seed(6)
    w = torch.randn(8, 8, 3, 3, generator=g) * 0.2
    bn = _bn_params(8, g)
    x = torch.randn(1, 8, 6, 5, generator=g)
    idt = torch.randn(1, 8, 6, 5, generator=g)
    layer = ops.Conv2d(w, None, padding=1, bn=bn, act=ops.ACT_RELU | ops.ACT_POST_FIRST, device="cpu")
    xin, pin = _nhwc(x), _nhwc(idt)
    out = torch.empty(1, 6, 5, 8)
    EMU.run(layer.desc(xin, out, post=pin))
    ref = torch.relu(F.batch_norm(F.conv2d(x, w, padding=1), bn[2], bn[3], bn[0], bn[1], False, 0.0, bn[4]) + idt)
    assert torch.allclose(out, _nhwc(ref), atol=2e-5)


# ---- Winograd F(4x4, 3x3): the host side (U = G g G^T, the Z = 36 GEMM descriptor, strides of a shared V) -----------------
_BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]], dtype=np.float64)
_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def _wino_input(x, reflect):
    """include/mit_hip.h mit_wino43_input in numpy: x [B,H,W,C] -> V [36, T, C] (float64)."""
    B, H, W, Cc = x.shape
    th, tw = (H + 3) // 4, (W + 3) // 4

    def src(i, n):
        if reflect:
            i = -i if i < 0 else i
            i = 2 * n - 2 - i if i >= n else i
            return max(i, 0)
        return i if 0 <= i < n else -1

    V = np.zeros((36, B * th * tw, Cc))
    for b in range(B):
        for ty in range(th):
            for tx in range(tw):
                d = np.zeros((6, 6, Cc))
                for r in range(6):
                    for s in range(6):
                        iy, ix = src(ty * 4 - 1 + r, H), src(tx * 4 - 1 + s, W)
                        if iy >= 0 and ix >= 0:
                            d[r, s] = x[b, iy, ix]
                V[:, (b * th + ty) * tw + tx] = np.einsum("ir,rsc,js->ijc", _BT, d, _BT).reshape(36, Cc)
    return V


def _wino_output(M, B, H, W):
    th, tw = (H + 3) // 4, (W + 3) // 4
    N = M.shape[2]
    y = np.zeros((B, th * 4, tw * 4, N))
    for t in range(B * th * tw):
        b, ty, tx = t // (th * tw), (t // tw) % th, t % tw
        y[b, ty * 4:ty * 4 + 4, tx * 4:tx * 4 + 4] = np.einsum("ar,rsn,es->aen", _AT, M[:, t].reshape(6, 6, N), _AT)
    return y[:, :H, :W]


@pytest.mark.parametrize("reflect", [True, False])
def test_winograd_host_side_against_conv2d(reflect):
    """WinogradConv3x3's transformed weights and its 36-way GEMM descriptor (run by the emulator), wrapped in numpy versions of
    the two transform kernels, reproduce F.conv2d — for the stand-alone layer and for a layer reading a channel prefix of a
    wider shared V."""
    g = torch.Generator().manual_seed(21)
    B, H, W, Cw, Cin, Cout = 2, 7, 10, 32, 16, 6
    x = torch.randn(B, H, W, Cw, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 12
    layer = ops.WinogradConv3x3(w, None, pad_mode=ops.PAD_REFLECT if reflect else ops.PAD_ZERO, device="cpu")
    T = layer.tiles(B, H, W)
    V = torch.from_numpy(_wino_input(x.numpy().astype(np.float64), reflect)).to(torch.float32).contiguous()  # all Cw channels
    M = torch.zeros(36, T, Cout)
    EMU.run(layer.gemm_desc(V, M))  # reads the first Cin channels of every V row
    got = _wino_output(M.numpy().astype(np.float64), B, H, W)
    xp = F.pad(x[..., :Cin].permute(0, 3, 1, 2).double(), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xp, w.double()).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_rfft_row_plan_and_tables():
    """Host side of the mixed-radix W-axis FFT: the widths the kernel accepts (C-ABI predicate, no GPU needed) and the
    twiddle tables, checked by running the kernel's algorithm (packed real FFT: Stockham stages + untangle) in numpy on
    those very tables against numpy.fft.rfft / irfft."""
    import numpy as np

    from manga_image_translator_amd import lama, lib

    L = lib.load()
    assert all(L.mit_rfft_rows_supported(w) for w in (182, 8, 42, 64, 210, 26, 364, 512, 4))
    assert not any(L.mit_rfft_rows_supported(w) for w in (181, 362, 34, 514, 2, 0))

    def plan(n):  # same greedy order as make_plan in csrc/fft_rows.hip
        out = []
        for r in (4, 2, 3, 5, 7, 11, 13):
            while n % r == 0:
                out.append(r)
                n //= r
        assert n == 1
        return out

    def stockham(z, tw, inverse):
        N = len(z)
        x, n, s = z.copy(), N, 1
        for P in plan(N):
            m, y = n // P, np.empty_like(z)
            for bf in range(N // P):
                q, p = bf % s, bf // s
                xi = np.array([x[q + s * (p + m * i)] for i in range(P)])
                for k in range(P):
                    wp = np.exp((2j if inverse else -2j) * np.pi * np.arange(P) * k / P)
                    t = tw[(p * k * (N // n)) % N]
                    y[q + s * (P * p + k)] = (xi * wp).sum() * (t if inverse else np.conj(t))
            x, n, s = y, m, s * P
        return x

    rng = np.random.default_rng(5)
    for w in (182, 42, 64):
        N = w // 2
        tabs = lama.rfft_row_tables(w).numpy().astype(np.float64)
        assert tabs.shape == (2 * N + 1, 2)
        tw, tw2 = tabs[:N, 0] + 1j * tabs[:N, 1], tabs[N:, 0] + 1j * tabs[N:, 1]
        x = rng.standard_normal(w)
        Z = stockham(x[0::2] + 1j * x[1::2], tw, False)
        k = np.arange(N + 1)
        zk, zn = Z[k % N], np.conj(Z[(N - k) % N])
        X = ((zk + zn) - 1j * np.conj(tw2) * (zk - zn)) * 0.5 / np.sqrt(w)
        assert np.abs(X - np.fft.rfft(x, norm="ortho")).max() < 1e-6
        A, Bc = X[:N].copy(), np.conj(X[N - np.arange(N)])
        A[0] = A[0].real
        Bc[0] = Bc[0].real
        zz = stockham((A + Bc) + 1j * tw2[:N] * (A - Bc), tw, True) / np.sqrt(w)
        back = np.empty(w)
        back[0::2], back[1::2] = zz.real, zz.imag
        assert np.abs(back - x).max() < 1e-6


def _pil_pages():
    from PIL import Image

    rng = np.random.default_rng(8)
    rgb = Image.fromarray(rng.integers(0, 256, (20, 28, 3)).astype(np.uint8))
    rgba = Image.fromarray(rng.integers(0, 256, (20, 28, 4)).astype(np.uint8), "RGBA")
    grey = Image.fromarray(rng.integers(0, 256, (20, 28)).astype(np.uint8))
    pal = rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=16)
    pal.info["transparency"] = 3
    return {"RGB": rgb, "RGBA": rgba, "L": grey, "P": pal}


def test_load_and_dump_image_round_trip():
    """imgproc.load_image / dump_image (utils/generic.py:223-249): modes, the white flattening of transparent pixels, the alpha that
    comes back on the result."""
    from manga_image_translator_amd import imgproc

    for mode, page in _pil_pages().items():
        arr, alpha = imgproc.load_image(page)
        assert arr.dtype == np.uint8 and arr.shape == (20, 28, 3) and (alpha is not None) == (mode in ("RGBA", "P"))
        out = imgproc.dump_image(page, arr, alpha)
        assert out.mode == "RGBA" and out.size == page.size
        if mode == "RGBA":
            a = np.array(page)[..., 3:4].astype(np.float64) / 255
            want = np.array(page)[..., :3] * a + 255 * (1 - a)                      # PIL's paste-through-mask blend, to within rounding
            assert np.abs(arr.astype(np.float64) - want).max() <= 1.0
            assert np.array_equal(np.array(out)[..., 3], np.array(page)[..., 3])
        if mode == "RGB":
            assert np.array_equal(arr, np.array(page)) and np.array_equal(np.array(out)[..., :3], arr)
    big = np.zeros((40, 56, 3), np.uint8)                                            # an upscaled result: the container follows its size
    assert imgproc.dump_image(_pil_pages()["RGB"], big).size == (56, 40)


def test_load_and_dump_image_match_the_reference_functions():
    from oracle import ref_import as R

    if not R.available():
        pytest.skip("/root/reference not present")
    from manga_image_translator_amd import imgproc

    G = R.generic()
    for mode, page in _pil_pages().items():
        a0, al0 = G.load_image(page.copy())
        a1, al1 = imgproc.load_image(page.copy())
        assert np.array_equal(a0, a1) and (al0 is None) == (al1 is None)
        if al0 is not None:
            assert np.array_equal(np.array(al0), np.array(al1))
        res = (a0.astype(np.int32) // 2).astype(np.uint8)
        assert np.array_equal(np.array(G.dump_image(page, res.copy(), al0)), np.array(imgproc.dump_image(page, res.copy(), al1)))
