"""mit_resize_u8 / mit_select_u8 on the GPU against the numpy twin (bit-exact: same tap tables, integer arithmetic), and the LaMa
plugin on pages that need the resize legs against the reference's own _infer output (tests/golden/lama_resize.npz)."""
import asyncio
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("sh,sw,dh,dw,c", [(37, 53, 20, 31, 3), (64, 48, 32, 24, 3), (100, 70, 128, 90, 1), (250, 333, 256, 336, 3),
                                           (256, 336, 250, 333, 3), (1, 9, 4, 20, 3), (2048, 1456, 1024, 728, 3), (1441, 1025, 2048, 1456, 1),
                                           (4096, 2880, 2048, 1440, 3), (2048, 1440, 4096, 2880, 3)])
def test_resize_u8_bit_exact(cuda, sh, sw, dh, dw, c):
    from manga_image_translator_amd import imgproc as IP

    rng = np.random.default_rng(sh + dw)
    B = 2 if sh * sw < 1 << 20 else 1
    src = rng.integers(0, 256, size=(B, sh, sw, c), dtype=np.uint8)
    src = src[..., 0] if c == 1 else src
    for exact in (False, True):
        got = IP.resize_u8(torch.from_numpy(src).to(cuda), (dw, dh), exact=exact)
        torch.cuda.synchronize()
        assert got.dtype == torch.uint8 and tuple(got.shape[:3]) == (B, dh, dw)
        for b in range(B):
            assert np.array_equal(got[b].cpu().numpy(), IP.resize_u8_host(src[b], (dw, dh), exact=exact)), (exact, b)


def test_select_u8(cuda):
    from manga_image_translator_amd import imgproc as IP

    rng = np.random.default_rng(0)
    m = rng.integers(0, 256, size=(2, 33, 47), dtype=np.uint8)
    m[0, 0, :4] = [126, 127, 128, 0]
    a, b = (rng.integers(0, 256, size=(2, 33, 47, 3), dtype=np.uint8) for _ in range(2))
    got = IP.select_u8(torch.from_numpy(m).to(cuda), 127, torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)).cpu().numpy()
    assert np.array_equal(got, np.where((m >= 127)[..., None], a, b))
    with pytest.raises(ValueError):
        IP.select_u8(torch.from_numpy(m).to(cuda), 127, torch.from_numpy(a[:, :8]).to(cuda), torch.from_numpy(b).to(cuda))
    with pytest.raises(ValueError):
        IP.resize_u8(torch.zeros(1, 8, 8, 3, device=cuda), (4, 4))


def test_lama_plugin_on_pages_that_need_resizing(cuda):
    """HipLamaMPEInpainter._infer with no injected callables on a 250x333 page (not a multiple of 8) and a 300x200 page above
    inpainting_size = 160 (resize_keep_aspect first): the reference's own LamaMPEInpainter._infer produced the expected bytes
    (oracle/make_golden.py:golden_lama_resize, cv2 stand-in).  Bytes equal except +-1 where a truncation or a fixed-point
    interpolation of a truncation flip lands; nothing outside the original mask changes."""
    from manga_image_translator_amd import lama_schema, plugins as P, synth

    run = lambda c: asyncio.new_event_loop().run_until_complete(c)
    g = np.load(os.path.join(GOLDEN, "lama_resize.npz"))
    w = {"lama.gen": synth.synth_state_dict(lama_schema.lama_generator_schema(9)), "lama.mpe": synth.synth_state_dict(lama_schema.lama_mpe_schema())}
    inp = P.HipLamaMPEInpainter(weights=w)
    run(inp.load("cuda"))
    for tag in ("a", "b"):
        page, mask, size = g[f"page_{tag}"], g[f"mask_{tag}"], int(g[f"size_{tag}"])
        before = page.copy()
        out = run(inp.infer(page, mask, None, size))
        assert out.shape == page.shape and out.dtype == np.uint8 and np.array_equal(page, before)
        d = np.abs(out.astype(np.int32) - g[f"out_{tag}"].astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3, (tag, d.max(), (d != 0).mean())
        assert np.array_equal(out[mask < 127], page[mask < 127])
        print(f"lama plugin {page.shape[:2]} size {size}: {int((d != 0).sum())} of {d.size} bytes differ by 1")
    with pytest.raises(ValueError):
        run(inp.infer(g["page_a"], g["mask_a"][:10], None, 1024))
    run(inp.unload())
