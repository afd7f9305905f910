"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the reference LaMa-MPE / LaMa-large inpainter.

Functional fp32 torch-CPU restatement of
  /root/reference/manga_translator/inpainting/inpainting_lama_mpe.py
driven by a state_dict with the reference's own key names (``gen_state_dict`` /
``str_state_dict`` of the checkpoint, :818-825).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product never does.

Parity status: the reference's tests hold NO golden vectors for this path —
the restatement is pinned against the reference module itself, imported in the build
container (tests/test_oracle_vs_reference.py) and against fixtures generated from it
(tests/golden/lama_*.npz, made by oracle/make_golden.py).  The cv2-dependent glue
(INTER_AREA resize, filter2D, INTER_NEAREST) is restated from OpenCV's documented semantics
because cv2 is not installed anywhere we can run: that part is "parity unpinned".
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _bn(x: torch.Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> torch.Tensor:
    # nn.BatchNorm2d in eval mode (FFC_BN_ACT.bn_l/bn_g :387-388, SpectralTransform.conv1[1] :275, FourierUnit.bn :198)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, eps)


def _conv_reflect(x: torch.Tensor, w: torch.Tensor, stride: int, pad: int) -> torch.Tensor:
    # nn.Conv2d(..., padding_mode='reflect') (:333-340)
    if pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, None, stride=stride)


def fourier_unit(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """FourierUnit.forward :214-257 (no spatial scaling, no pos-encoding, no SE, fft_norm='ortho')."""
    b, c, h, w = x.shape
    ff = torch.fft.rfftn(x, dim=(-2, -1), norm="ortho")  # :228
    ff = torch.stack((ff.real, ff.imag), dim=-1).permute(0, 1, 4, 2, 3).contiguous()  # :229-230
    ff = ff.view(b, -1, h, w // 2 + 1)  # channel = c*2 + {re,im} :231
    ff = F.conv2d(ff, sd[prefix + ".conv_layer.weight"])  # :242
    ff = torch.relu(_bn(ff, sd, prefix + ".bn"))  # :243
    ff = ff.view(b, -1, 2, h, w // 2 + 1).permute(0, 1, 3, 4, 2).contiguous()  # :245-246
    ff = torch.complex(ff[..., 0], ff[..., 1])  # :249
    return torch.fft.irfftn(ff, s=(h, w), dim=(-2, -1), norm="ortho")  # :252


def spectral_transform(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """SpectralTransform.forward :286-307 with stride 1 and enable_lfu=False (:649-657)."""
    x = torch.relu(_bn(F.conv2d(x, sd[prefix + ".conv1.0.weight"]), sd, prefix + ".conv1.1"))  # :289
    out = fourier_unit(x, sd, prefix + ".fu")  # :290
    return F.conv2d(x + out, sd[prefix + ".conv2.weight"])  # :305 (xs = 0)


def ffc_bn_act(x_l, x_g, sd: SD, prefix: str, stride: int, pad: int):
    """FFC_BN_ACT.forward :395-399 around FFC.forward :349-369 (not gated), ReLU activations."""
    has = lambda n: (prefix + ".ffc." + n + ".weight") in sd or (prefix + ".ffc." + n + ".conv1.0.weight") in sd
    out_l = out_g = None
    if has("convl2l"):
        out_l = _conv_reflect(x_l, sd[prefix + ".ffc.convl2l.weight"], stride, pad)
        if has("convg2l"):
            out_l = out_l + _conv_reflect(x_g, sd[prefix + ".ffc.convg2l.weight"], stride, pad)  # :365
    if has("convl2g"):
        out_g = _conv_reflect(x_l, sd[prefix + ".ffc.convl2g.weight"], stride, pad)
        if has("convg2g"):
            out_g = out_g + spectral_transform(x_g, sd, prefix + ".ffc.convg2g")  # :367
    if out_l is not None:
        out_l = torch.relu(_bn(out_l, sd, prefix + ".bn_l"))
    if out_g is not None:
        out_g = torch.relu(_bn(out_g, sd, prefix + ".bn_g"))
    return out_l, out_g


def generator_forward(sd: SD, img: torch.Tensor, mask: torch.Tensor, n_blocks: int,
                      rel_pos: Optional[torch.Tensor] = None, direct: Optional[torch.Tensor] = None,
                      taps: Optional[dict] = None) -> torch.Tensor:
    """FFCResNetGenerator.forward :603-613 for the LamaFourier configuration (:644-659).

    ``taps`` (optional dict) receives intermediate activations for layer-level parity checks.
    """
    x = torch.cat([img * (1 - mask), mask], dim=1)  # :604
    x = F.pad(x, (3, 3, 3, 3), mode="reflect")  # model.0 ReflectionPad2d(3) :554
    x_l, _ = ffc_bn_act(x, None, sd, "model.1", 1, 0)  # :555
    if rel_pos is not None:
        x_l = x_l + rel_pos  # :611
        x_l = x_l + direct  # :612
    if taps is not None:
        taps["stem"] = x_l
    x_l, _ = ffc_bn_act(x_l, None, sd, "model.2", 2, 1)  # downsample :566-571
    x_l, _ = ffc_bn_act(x_l, None, sd, "model.3", 2, 1)
    x_l, x_g = ffc_bn_act(x_l, None, sd, "model.4", 2, 1)  # ratio_gout = 0.75 :561-563
    if taps is not None:
        taps["down_l"], taps["down_g"] = x_l, x_g
    for i in range(n_blocks):  # FFCResnetBlock.forward :421-436
        p = f"model.{5 + i}"
        id_l, id_g = x_l, x_g
        x_l, x_g = ffc_bn_act(x_l, x_g, sd, p + ".conv1", 1, 1)
        x_l, x_g = ffc_bn_act(x_l, x_g, sd, p + ".conv2", 1, 1)
        x_l, x_g = id_l + x_l, id_g + x_g
        if taps is not None:
            taps[f"block{i}_l"], taps[f"block{i}_g"] = x_l, x_g
    x = torch.cat([x_l, x_g], dim=1)  # ConcatTupleLayer :535-542
    base = 5 + n_blocks + 1
    for i in range(3):  # upsample :585-591
        p = base + 3 * i
        x = F.conv_transpose2d(x, sd[f"model.{p}.weight"], sd[f"model.{p}.bias"], stride=2, padding=1, output_padding=1)
        x = torch.relu(_bn(x, sd, f"model.{p + 1}"))
    p = base + 9
    x = F.pad(x, (3, 3, 3, 3), mode="reflect")  # :597
    x = F.conv2d(x, sd[f"model.{p + 1}.weight"], sd[f"model.{p + 1}.bias"])  # :598
    return torch.sigmoid(x)  # add_out_act='sigmoid' :599-600,644


def mpe_forward(mpe_sd: SD, rel_pos: torch.Tensor, direct: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MPE.forward :625-632. rel_pos int64 [B,H,W], direct int64 [B,H,W,4] -> two [B,64,H,W]."""
    b, h, w = rel_pos.shape
    rel = F.embedding(rel_pos.reshape(b, h * w), mpe_sd["rel_pos_emb.weight"])
    rel = rel.reshape(b, h, w, -1).permute(0, 3, 1, 2) * mpe_sd["alpha5"]
    d = torch.matmul(direct.reshape(b, h * w, 4).to(torch.float32), mpe_sd["direct_emb.weight"])
    d = d.reshape(b, h, w, -1).permute(0, 3, 1, 2) * mpe_sd["alpha6"]
    return rel, d


# ---- cv2 restatements (semantics from the OpenCV documentation; cv2 itself is unavailable) ----

def resize_area_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) for uint8 single-channel images.

    Shrinking in both axes (the real-page case, page >= 256 px): each destination pixel is the
    area-weighted mean of the source box [dy*sy, (dy+1)*sy) x [dx*sx, (dx+1)*sx), rounded to
    nearest (saturate_cast).  Otherwise OpenCV falls back to its bilinear code with the
    INTER_AREA coefficient rule  s = floor(d*scale), f = (d+1) - (s+1)/scale, f = f<=0 ? 0 : f-floor(f).
    """
    dw, dh = dsize
    sh, sw = src.shape

    def area_weights(n_src: int, n_dst: int) -> np.ndarray:
        scale = n_src / n_dst
        wmat = np.zeros((n_dst, n_src), dtype=np.float64)
        for d in range(n_dst):
            lo, hi = d * scale, (d + 1) * scale
            s0, s1 = int(np.floor(lo)), min(int(np.ceil(hi)), n_src)
            for s in range(s0, s1):
                wmat[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
            wmat[d] /= scale
        return wmat

    def linear_area_weights(n_src: int, n_dst: int) -> np.ndarray:
        scale = n_src / n_dst
        inv = n_dst / n_src
        wmat = np.zeros((n_dst, n_src), dtype=np.float64)
        for d in range(n_dst):
            s = int(np.floor(d * scale))
            f = (d + 1) - (s + 1) * inv
            f = 0.0 if f <= 0 else f - np.floor(f)
            if s < 0:
                s, f = 0, 0.0
            if s >= n_src - 1:
                s, f = n_src - 1, 0.0
            wmat[d, s] += 1.0 - f
            if f > 0:
                wmat[d, s + 1] += f
        return wmat

    if dh <= sh and dw <= sw:
        wy, wx = area_weights(sh, dh), area_weights(sw, dw)
    else:
        wy, wx = linear_area_weights(sh, dh), linear_area_weights(sw, dw)
    out = wy @ src.astype(np.float64) @ wx.T
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def resize_nearest(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src index = min(floor(dst * src/dst_size), src-1)."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    ys = np.minimum(np.floor(np.arange(dh) * (sh / dh)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(dw) * (sw / dw)).astype(np.int64), sw - 1)
    return src[ys][:, xs]


def _filter_any(mask3: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """(cv2.filter2D(mask3, -1, kernel) > 0) with BORDER_REFLECT_101 for a 0/1 map and a 0/1 3x3 kernel
    (correlation, anchor at the centre)."""
    p = np.pad(mask3, 1, mode="reflect")
    h, w = mask3.shape
    acc = np.zeros_like(mask3)
    for ky in range(3):
        for kx in range(3):
            if kernel[ky, kx]:
                acc = acc + p[ky:ky + h, kx:kx + w]
    return (acc > 0).astype(mask3.dtype)


def load_masked_position_encoding(mask01: np.ndarray):
    """LamaFourier.load_masked_position_encoding :751-815. mask01: float [H,W] in {0,1}.

    Returns (rel_pos int32 [H,W], abs_pos int32 [256,256], direct int32 [H,W,4]).
    """
    mask = (mask01 * 255).astype(np.uint8)  # :752
    ones_filter = np.ones((3, 3), dtype=np.float32)
    d_filters = [np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]], dtype=np.float32),
                 np.array([[0, 0, 0], [1, 1, 0], [1, 1, 0]], dtype=np.float32),
                 np.array([[0, 1, 1], [0, 1, 1], [0, 0, 0]], dtype=np.float32),
                 np.array([[0, 0, 0], [0, 1, 1], [0, 1, 1]], dtype=np.float32)]
    str_size, pos_num = 256, 128
    ori_mask = mask.copy() / 255  # :761-763
    ori_h, ori_w = mask.shape
    mask = resize_area_u8(mask, (str_size, str_size))  # :764
    mask[mask > 0] = 255  # :765
    h, w = mask.shape
    mask3 = 1.0 - (mask / 255.0)  # :767-768
    pos = np.zeros((h, w), dtype=np.int32)
    direct = np.zeros((h, w, 4), dtype=np.int32)
    i = 0
    if mask3.max() > 0:  # :773
        while np.sum(1 - mask3) > 0:  # :775
            i += 1
            mask3_ = _filter_any(mask3, ones_filter)  # :777-778
            sub_mask = mask3_ - mask3
            pos[sub_mask == 1] = i  # :780
            for d, df in enumerate(d_filters):  # :782-800
                m = _filter_any(mask3, df) - mask3
                direct[m == 1, d] = 1
            mask3 = mask3_
    abs_pos = pos.copy()
    rel_pos = pos / (str_size / 2)  # :805
    rel_pos = (rel_pos * pos_num).astype(np.int32)
    rel_pos = np.clip(rel_pos, 0, pos_num - 1)
    if ori_w != w or ori_h != h:  # :809-813
        rel_pos = resize_nearest(rel_pos, (ori_w, ori_h))
        rel_pos[ori_mask == 0] = 0
        direct = resize_nearest(direct, (ori_w, ori_h))
        direct[ori_mask == 0, :] = 0
    return rel_pos, abs_pos, direct


def lama_call(sd: SD, mpe_sd: Optional[SD], img: torch.Tensor, mask: torch.Tensor, n_blocks: int,
              taps: Optional[dict] = None) -> torch.Tensor:
    """LamaFourier.__call__ :713-726 in inpaint_only mode. img [1,3,H,W] (already masked), mask [1,1,H,W]."""
    rel = d = None
    if mpe_sd is not None:
        rel_pos, _, direct = load_masked_position_encoding(mask[0][0].cpu().numpy())  # :717
        rel_pos = torch.from_numpy(rel_pos.astype(np.int64)).unsqueeze(0)
        direct = torch.from_numpy(direct.astype(np.int64)).unsqueeze(0)
        rel, d = mpe_forward(mpe_sd, rel_pos, direct)  # :720
    pred = generator_forward(sd, img, mask, n_blocks, rel, d, taps)  # :723
    return pred * mask + (1 - mask) * img  # :726


def infer(sd: SD, mpe_sd: Optional[SD], image: np.ndarray, mask: np.ndarray, n_blocks: int,
          taps: Optional[dict] = None, inpainting_size: Optional[int] = None) -> np.ndarray:
    """LamaMPEInpainter._infer :56-118 on the CPU (fp32, no autocast).  ``inpainting_size=None`` asserts the no-resize case
    (max(H, W) <= inpainting_size and H, W multiples of 8: the BASELINE 2048x1456 page); with a size, the page is first resized
    like the reference does — resize_keep_aspect (INTER_LINEAR_EXACT, :64-66), then INTER_LINEAR to a multiple of 8 (:67-79) and
    back (:112-113) — through the oracle's restatements of those OpenCV resizes (oracle/imgproc.py, oracle/ctd.py)."""
    from . import ctd as OC, imgproc as OI

    def lin(a, dsize):
        return OC.resize_linear_u8(a[..., None], dsize)[..., 0] if a.ndim == 2 else OC.resize_linear_u8(a, dsize)

    img_original = np.copy(image)
    mask_original = np.copy(mask)
    mask_original[mask_original < 127] = 0  # :59-61
    mask_original[mask_original >= 127] = 1
    mask_original = mask_original[:, :, None]
    height, width, _ = image.shape
    if inpainting_size is None:
        assert height % 8 == 0 and width % 8 == 0, "pass inpainting_size for pages that need the resize path"
    elif max(height, width) > inpainting_size:  # :64-66
        image, mask = OI.resize_keep_aspect(image, inpainting_size), OI.resize_keep_aspect(mask, inpainting_size)
    h, w, _ = image.shape
    new_h, new_w = (h + 7) // 8 * 8, (w + 7) // 8 * 8  # :67-76
    if (new_h, new_w) != (h, w):  # :77-79
        image, mask = lin(image, (new_w, new_h)), lin(mask, (new_w, new_h))
    img_t = torch.from_numpy(image).permute(2, 0, 1).unsqueeze(0).float() / 255.0  # :82
    mask_t = torch.from_numpy(mask).unsqueeze(0).unsqueeze(0).float() / 255.0  # :85
    mask_t[mask_t < 0.5] = 0
    mask_t[mask_t >= 0.5] = 1
    with torch.no_grad():
        img_t *= (1 - mask_t)  # :92
        out = lama_call(sd, mpe_sd, img_t, mask_t, n_blocks, taps)  # :95
    if taps is not None:
        taps["out_float"] = out
    inpainted = (out.squeeze(0).permute(1, 2, 0).numpy() * 255.0).astype(np.uint8)  # :111
    if (new_h, new_w) != (height, width):  # :112-113
        inpainted = lin(inpainted, (width, height))
    return inpainted * mask_original + img_original * (1 - mask_original)  # :117
