"""ESRGAN (RRDBNet 4x) stage parity: HIP engine vs the CPU oracle restatement of the reference.

Tolerance: ~30-70 fp32 conv layers with 0.2-scaled residuals; the float output is compared at 1e-4 * max|ref|
(the merged-tap up-convs and the 0.2 folded into the epilogue change rounding, not values); the uint8 image must be
identical except at a x255 truncation boundary (+-1 level)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nb,B,H,W", [(2, 2, 24, 40), (4, 1, 37, 51), (23, 1, 128, 136)],  # nb = 23: the 4xESRGAN checkpoint's depth
                         ids=["nb2", "nb4", "nb23-128x136"])
def test_esrgan_parity(cuda, nb, B, H, W):
    from manga_image_translator_amd import esrgan, esrgan_schema, synth
    from oracle import esrgan as OE

    sd = synth.synth_state_dict(esrgan_schema.rrdbnet_schema(nb))
    eng = esrgan.EsrganEngine(sd, nb=nb, device=cuda)
    pages = [synth.synth_page(20 + i, H, W, n_boxes=2)[0] for i in range(B)]
    taps = {}
    out = eng.forward(torch.from_numpy(np.stack(pages)).to(cuda), taps=taps)
    torch.cuda.synchronize()
    assert out.shape == (B, 4 * H, 4 * W, 3) and out.dtype == torch.uint8
    for i in range(B):
        x = torch.from_numpy(pages[i][:, :, ::-1].copy()).float().div(255.0).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            y = OE.rrdbnet_forward(sd, x, nb)[0]                     # BGR planes
        got = taps["out_float"][i].cpu().permute(2, 0, 1).flip(0)    # engine emits RGB
        err = (got - y).abs().max().item()
        assert err < 1e-4 * max(1.0, y.abs().max().item()), err
        ref_u8 = OE.infer(sd, pages[i], nb)
        d = np.abs(out[i].cpu().numpy().astype(np.int32) - ref_u8.astype(np.int32))
        bad = np.argwhere(d != 0)
        if len(bad):
            yf = y.clip(0, 1).permute(1, 2, 0).numpy()[:, :, ::-1] * 255.0
            frac = np.abs(yf - np.round(yf))
            assert d.max() <= 1 and all(frac[tuple(b)] < 0.05 for b in bad) and len(bad) < 2e-3 * d.size


def test_esrgan_rejects_bad_input(cuda):
    from manga_image_translator_amd import esrgan, esrgan_schema, synth

    eng = esrgan.EsrganEngine(synth.synth_state_dict(esrgan_schema.rrdbnet_schema(1)), nb=1, device=cuda)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 8, 8, 3, device=cuda))


def test_esrgan_plugin(cuda):
    import asyncio

    from PIL import Image

    from manga_image_translator_amd import esrgan_schema, plugins as P, synth
    from oracle import esrgan as OE

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    sd = synth.synth_state_dict(esrgan_schema.rrdbnet_schema(2))
    up = P.HipESRGANUpscaler(weights=sd)
    run(up.load("cuda"))
    assert up.engine.nb == 2
    page = synth.synth_page(5, 32, 48, n_boxes=2)[0]
    outs = run(up.infer([Image.fromarray(page)], 2))
    assert len(outs) == 1 and outs[0].size == (96, 64)  # 4x then x0.5 through PIL (esrgan_pytorch.py:546)
    ref = Image.fromarray(OE.infer(sd, page, 2)).resize((96, 64), resample=Image.Resampling.BILINEAR)
    d = np.abs(np.asarray(outs[0]).astype(np.int32) - np.asarray(ref).astype(np.int32))
    assert d.max() <= 1
    run(up.unload())
