"""Winograd F(4x4, 3x3) path (mit_wino43_input -> 36 GEMMs -> mit_wino43_output) against a float64 convolution.

Tolerance: 5e-5 of the output range per layer (observed ~1e-5; the direct form is at ~3e-7) — two orders inside the 2e-4
the LaMa stage states for its block outputs; tests/test_lama_gpu.py holds the whole network to that stage tolerance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,Cin,Cout,H,W,mode,act,bn,res", [
    (2, 16, 32, 8, 8, "zero", 0, False, False),
    (1, 128, 384, 24, 18, "reflect", 0, False, False),     # convl2g, ragged tile count in W
    (2, 512, 128, 13, 22, "reflect", 1, True, True),       # convl2l + convg2l fused, partial tiles both ways
    (1, 64, 32, 5, 7, "zero", 2, True, True),
    (1, 32, 64, 2, 3, "reflect", 1, False, False),         # smaller than one tile
])
def test_winograd_conv_parity(cuda, B, Cin, Cout, H, W, mode, act, bn, res):
    from manga_image_translator_amd import ops

    g = torch.Generator().manual_seed(Cin * 3 + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bn_t = None
    if bn:
        bn_t = (torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1,
                torch.rand(Cout, generator=g) + 0.5, 1e-5)
    post = torch.randn(B, Cout, H, W, generator=g) if res else None
    layer = ops.WinogradConv3x3(w, None, pad_mode=ops.PAD_REFLECT if mode == "reflect" else ops.PAD_ZERO, bn=bn_t, act=act, alpha=0.2,
                                device=cuda)
    out = layer(_nhwc(x).to(cuda), post=None if post is None else _nhwc(post).to(cuda))
    torch.cuda.synchronize()
    xd = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    ref = F.conv2d(xd, w.double())
    if bn:
        ga, be, mu, var, eps = (t.double() if torch.is_tensor(t) else t for t in bn_t)
        ref = (ref - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps) * ga[None, :, None, None] + be[None, :, None, None]
    ref = {0: lambda t: t, 1: torch.relu, 2: lambda t: F.leaky_relu(t, 0.2)}[act](ref)
    if res:
        ref = ref + post.double()
    got = out.cpu().permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())


def test_shared_input_transform_and_channel_slices(cuda):
    """The FFC use: one transformed input serves a 512-channel conv and a conv over its first 128 channels; outputs land in
    channel slices of a wider tensor.  Must equal the stand-alone calls bit for bit."""
    from manga_image_translator_amd import ops

    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 16, 12
    x = torch.randn(B, H, W, 512, generator=g).to(cuda)
    wa = torch.randn(128, 512, 3, 3, generator=g) / 68
    wb = torch.randn(384, 128, 3, 3, generator=g) / 34
    la = ops.WinogradConv3x3(wa, None, pad_mode=ops.PAD_REFLECT, act=ops.ACT_RELU, device=cuda)
    lb = ops.WinogradConv3x3(wb, None, pad_mode=ops.PAD_REFLECT, device=cuda)
    T = la.tiles(B, H, W)
    V = torch.empty(36, T, 512, device=cuda)
    wide = torch.zeros(B, H, W, 512, device=cuda)
    P = torch.empty(B, H, W, 384, device=cuda)
    la.transform_input(x, V)
    la.gemm_output(V, torch.empty(36, T, 128, device=cuda), wide[..., :128])
    lb.gemm_output(V, torch.empty(36, T, 384, device=cuda), P)
    ya, yb = la(x), lb(x[..., :128])
    torch.cuda.synchronize()
    assert torch.equal(wide[..., :128], ya) and torch.equal(P, yb) and float(wide[..., 128:].abs().max()) == 0.0
