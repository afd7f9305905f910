"""TEST INFRASTRUCTURE — generates the committed fixtures under tests/golden/ by running the REFERENCE's own code.

Run in the build container (where /root/reference exists):   python -m oracle.make_golden

The reference's tests hold no golden vectors for the dense path, so the pins are produced here by
importing the reference's nn.Module / helper files by path (oracle/ref_import.py) and running them, fp32 on the CPU,
on small seeded inputs with the seeded synthetic weights (manga_image_translator_amd/synth.py — the weights are
regenerated from their seed wherever the fixtures are consumed, only inputs and outputs are stored).  Wherever the
reference code calls OpenCV the call lands in ref_import.cv2_shim(), i.e. in the oracle's restatement of that
primitive: what is pinned is the reference's Python + ATen path, not OpenCV's pixels.

Fixtures (all small; each .npz also records the reference file whose code produced it):
  lama_mpe.npz     LamaFourier(use_mpe=True).__call__  (inpainting_lama_mpe.py:713-726, :751-815)   64x72
  lama_large.npz   LamaFourier(large_arch=True).__call__                                             48x40
  lama_resize.npz  LamaMPEInpainter._infer as a whole (inpainting_lama_mpe.py:56-118), pages 250x333 and 300x200 (inpainting_size 160)
  ctd.npz          preprocess_img + TextDetBase.forward (ctd.py:17-28, ctd_utils/basemodel.py:234-238) 120x90 page
  ocr48.npz        OCR.infer_beam_batch_tensor (ocr/model_48px.py:678-801) on 5 crops, dict 97, T = 9
  ocr_ctc.npz      OCR.forward + OCR.decode (ocr/model_48px_ctc.py:463-494) on 3 crops padded to max_w+7+128, dict 97
  dbnet.npz        TextDetection.forward + sigmoid (detection/default_utils/DBNet_resnet34.py:98-125, default.py:15-25) on a 256x256 page (fp16 maps + fp32 crops)
  esrgan.npz       RRDBNet.forward + the tensor part of ESRGANUpscalerPytorch._infer (upscaling/esrgan_pytorch.py:67-75,537-546), nb = 2, 40x56 page
  rearrange.npz    det_rearrange_forward + square_pad_resize (utils/generic.py:848-997) on tall / wide strips, stand-in network
  direction.npz    quadrilateral_can_merge_region + CommonOCR._generate_text_direction (utils/generic.py:653-698, ocr/common.py:12-39)
  refine_mask.npz  refine_mask / merge_mask_list / enlarge_window (detection/ctd_utils/textmask.py:16-174) on a 384x320 page
  textline_merge.json  the line sets + expected groupings of the reference's test/test_textline_merge.py and the reference code's own output
  mask_refinement.npz  the reference's mask_refinement.dispatch / complete_mask on a synthetic page (DenseCRF + bilateralFilter stubbed)
  textline.npz     sort_pnts / Quadrilateral / get_transformed_region (utils/generic.py:324-481) on 12 quads
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from manga_image_translator_amd import ctd_schema, lama_schema, ocr_schema, synth  # noqa: E402
from oracle import ref_import as R  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
OCR_DICT = 97


def build_ref_lama(n_blocks: int, mpe: bool):
    L = R.lama()
    L.cv2 = R.cv2_shim()
    m = L.LamaFourier(build_discriminator=False, use_mpe=mpe, large_arch=(n_blocks == 18))
    sd = synth.synth_state_dict(lama_schema.lama_generator_schema(n_blocks))
    m.generator.load_state_dict(sd, strict=True)
    mpe_sd = None
    if mpe:
        mpe_sd = synth.synth_state_dict(lama_schema.lama_mpe_schema())
        m.mpe.load_state_dict(mpe_sd, strict=True)
    return m.eval(), sd, mpe_sd


def ref_lama_infer(m, image: np.ndarray, mask: np.ndarray):
    """The tensor part of LamaMPEInpainter._infer (:82-117) around the reference model (no resize: H, W % 8 == 0)."""
    img_t = torch.from_numpy(image).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    mask_t = torch.from_numpy(mask).unsqueeze(0).unsqueeze(0).float() / 255.0
    mask_t[mask_t < 0.5] = 0
    mask_t[mask_t >= 0.5] = 1
    with torch.no_grad():
        img_t = img_t * (1 - mask_t)
        out = m(img_t, mask_t)
    return out


def golden_lama():
    for name, nb, mpe, (H, W), seed in (("lama_mpe", 9, True, (64, 72), 3), ("lama_large", 18, False, (48, 40), 4)):
        m, _, _ = build_ref_lama(nb, mpe)
        page, _, mask = synth.synth_page(seed, H, W, n_boxes=3)
        mask[5, 7] = 127
        out = ref_lama_infer(m, page, mask)
        extra = {}
        if mpe:
            mk = (mask.astype(np.float32) / 255.0 >= 0.5).astype(np.float32)
            rel, ab, direct = m.load_masked_position_encoding(mk)
            extra = dict(rel_pos=rel.astype(np.int32), direct=direct.astype(np.int8))
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), page=page, mask=mask, out_float=out.numpy(),
                            n_blocks=nb, source="manga_translator/inpainting/inpainting_lama_mpe.py", **extra)
        print(name, out.shape, float(out.mean()))


def golden_lama_resize():
    """The reference's own ``LamaMPEInpainter._infer`` (inpainting_lama_mpe.py:56-118) — the whole method, resize legs included —
    executed on the CPU with the cv2 stand-in: a page that is not a multiple of 8 (resize to 256x336 and back) and a page above
    ``inpainting_size`` (resize_keep_aspect = INTER_LINEAR_EXACT first)."""
    import asyncio
    from unittest import mock

    L = R.lama()
    shim = R.cv2_shim()
    G = R.generic()
    L.cv2 = G.cv2 = shim
    L.resize_keep_aspect = G.resize_keep_aspect            # the reference's own helper (utils/generic.py:251-255) over the stand-in
    m, _, _ = build_ref_lama(9, True)
    plug = L.LamaMPEInpainter.__new__(L.LamaMPEInpainter)
    plug.model, plug.device, plug.logger = m, "cpu", mock.MagicMock()
    out = {}
    for tag, (H, W), size, seed in (("a", (250, 333), 1024, 5), ("b", (300, 200), 160, 6)):
        page, _, mask = synth.synth_page(seed, H, W, n_boxes=3)
        mask[4, 9] = 127
        res = asyncio.new_event_loop().run_until_complete(plug._infer(page, mask, None, size))
        out.update({f"page_{tag}": page, f"mask_{tag}": mask, f"size_{tag}": size, f"out_{tag}": np.asarray(res).astype(np.uint8)})
        assert np.asarray(res).min() >= 0 and np.asarray(res).max() <= 255
        print("lama_resize", tag, res.shape, res.dtype, float(np.asarray(res).mean()))
    np.savez_compressed(os.path.join(GOLDEN, "lama_resize.npz"), source="manga_translator/inpainting/inpainting_lama_mpe.py:56-118", **out)


def build_ref_ctd():
    bm, yolo = R.ctd()
    shim = R.cv2_shim()
    g = ctd_schema.CTD_GAIN
    ysd = synth.synth_state_dict(ctd_schema.yolo_schema(), gain=g)
    ssd = synth.synth_state_dict(ctd_schema.unet_head_schema(), gain=g)
    dsd = synth.synth_state_dict(ctd_schema.db_head_schema(), gain=g)
    holder = torch.nn.Module()
    holder.blk_det = yolo.load_yolov5_ckpt({"cfg": ctd_schema.YOLOV5S_CFG, "weights": ysd})
    holder.text_seg = bm.UnetHead(act="leaky")
    holder.text_seg.load_state_dict(ssd, strict=True)
    holder.text_det = bm.DBHead(64, act="leaky")
    holder.text_det.load_state_dict(dsd, strict=True)
    holder.eval()
    fwd = lambda x: bm.TextDetBase.forward(holder, x)
    return fwd, shim, (ysd, ssd, dsd)


def golden_ctd():
    fwd, shim, _ = build_ref_ctd()
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "_ref_imgproc", os.path.join(R.PKG, "detection/ctd_utils/utils/imgproc_utils.py"))
    ip = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ip)
    ip.cv2 = shim
    page = synth.synth_page(7, 120, 90, n_boxes=3)[0]
    # preprocess_img (ctd.py:17-28) with the reference's own letterbox(); the BGR2RGB + [::-1] pair cancels
    img_in, ratio, (dw, dh) = ip.letterbox(page, new_shape=(128, 128), auto=False, stride=64)
    x = torch.from_numpy(np.ascontiguousarray(img_in.transpose(2, 0, 1))[None].astype(np.float32) / 255)
    with torch.no_grad():
        _, mask, lines = fwd(x)
    np.savez_compressed(os.path.join(GOLDEN, "ctd.npz"), page=page, net_in=x.numpy(), mask=mask.numpy(), lines=lines.numpy(),
                        dw=dw, dh=dh, source="manga_translator/detection/ctd_utils/basemodel.py + utils/imgproc_utils.py")
    print("ctd", mask.shape, lines.shape, float(mask.mean()), float(lines.mean()), (dw, dh))


def build_ref_ocr():
    M = R.ocr48()
    dictionary = [f"c{i}" for i in range(OCR_DICT)]
    model = M.OCR(dictionary, 768)
    sd = synth.synth_state_dict(ocr_schema.ocr48_schema(OCR_DICT))
    model.load_state_dict(sd, strict=True)
    return model.eval(), sd


def golden_ocr():
    model, _ = build_ref_ocr()
    rng = np.random.default_rng(21)
    widths = [41, 58, 90, 91, 133]
    Wp = max(widths) + 7
    region = np.zeros((len(widths), 48, Wp, 3), dtype=np.uint8)
    for i, w in enumerate(widths):
        region[i, :, :w] = rng.integers(0, 256, size=(48, w, 3), dtype=np.uint8)
    img = ((torch.from_numpy(region).float() - 127.5) / 127.5).permute(0, 3, 1, 2).contiguous()
    T = 9
    with torch.no_grad():
        res = model.infer_beam_batch_tensor(img, widths, beams_k=5, max_seq_length=T)
        mem = model.backbone(img).squeeze(2).permute(0, 2, 1)
    tok = np.zeros((len(widths), T + 1), dtype=np.int64)
    ln = np.zeros(len(widths), dtype=np.int64)
    prob = np.zeros(len(widths), dtype=np.float64)
    cols = np.zeros((len(widths), T + 1, 10), dtype=np.float32)
    for i, (idx, p, fg, bg, fgi, bgi) in enumerate(res):
        n = len(idx)
        tok[i, :n], ln[i], prob[i] = idx.numpy(), n, float(p)
        cols[i, :n] = torch.cat([fg, bg, fgi, bgi], dim=-1).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "ocr48.npz"), region=region, widths=np.array(widths), T=T, tokens=tok, length=ln,
                        prob=prob, colors=cols, backbone=mem.numpy(), dict_size=OCR_DICT,
                        source="manga_translator/ocr/model_48px.py")
    print("ocr48", tok[:, :6], ln, prob)


def golden_textline():
    G = R.generic()
    G.cv2 = R.cv2_shim()
    rng = np.random.default_rng(9)
    H, W = 200, 260
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    quads, sorted_pts, dirs, crops, ratios, fonts = [], [], [], [], [], []
    for k in range(12):
        vertical = k % 2 == 0
        bw, bh = (int(rng.integers(14, 30)), int(rng.integers(60, 150))) if vertical else (int(rng.integers(60, 150)), int(rng.integers(12, 30)))
        x0, y0 = int(rng.integers(0, W - bw)), int(rng.integers(0, H - bh))
        q = np.array([[x0, y0], [x0 + bw, y0], [x0 + bw, y0 + bh], [x0, y0 + bh]], dtype=np.int64)
        if k % 3 == 1:
            a = np.deg2rad(rng.uniform(-9, 9))
            c = q.mean(0)
            q = np.rint((q - c) @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T + c).astype(np.int64)
        elif k % 3 == 2:
            q = q + rng.integers(-3, 4, size=(4, 2))
        q = q[rng.permutation(4)]  # arbitrary corner order in, canonical order out
        quad = G.Quadrilateral(q.copy(), "", 0)
        region = quad.get_transformed_region(img, quad.direction, 48)
        quads.append(q)
        sorted_pts.append(np.asarray(quad.pts))
        dirs.append(quad.direction)
        ratios.append(quad.aspect_ratio)
        fonts.append(quad.font_size)
        crops.append(region)
    wmax = max(c.shape[1] for c in crops)
    packed = np.zeros((len(crops), 48, wmax, 3), dtype=np.uint8)
    for i, c in enumerate(crops):
        packed[i, :, :c.shape[1]] = c
    np.savez_compressed(os.path.join(GOLDEN, "textline.npz"), image=img, quads=np.array(quads), sorted_pts=np.array(sorted_pts),
                        direction=np.array(dirs), aspect_ratio=np.array(ratios), font_size=np.array(fonts),
                        crops=packed, crop_width=np.array([c.shape[1] for c in crops]),
                        source="manga_translator/utils/generic.py")
    print("textline", dirs, [c.shape for c in crops][:4])


def build_ref_ocr_ctc():
    from manga_image_translator_amd import ocr_ctc_schema

    M = R.ocr_ctc()
    model = M.OCR([f"c{i}" for i in range(OCR_DICT)], 768)
    sd = synth.synth_state_dict(ocr_ctc_schema.ocr_ctc_schema(OCR_DICT), gain=ocr_ctc_schema.CTC_GAIN)
    model.load_state_dict(sd, strict=True)
    return model.eval(), sd


def golden_ocr_ctc():
    model, _ = build_ref_ocr_ctc()
    rng = np.random.default_rng(33)
    widths = [37, 64, 101]
    Wp = max(widths) + 7 + 128  # model_48px_ctc.py:84
    region = np.zeros((len(widths), 48, Wp, 3), dtype=np.uint8)
    for i, w in enumerate(widths):
        region[i, :, :w] = rng.integers(0, 256, size=(48, w, 3), dtype=np.uint8)
    img = ((torch.from_numpy(region).float() - 127.5) / 127.5).permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        logits, colors = model(img)
        texts = model.decode(img, widths, 0)
    T = logits.shape[1]
    nmax = max(1, max(len(t) for t in texts))
    ids = np.full((len(widths), nmax), -1, dtype=np.int64)
    vals = np.zeros((len(widths), nmax, 7), dtype=np.float32)
    for i, line in enumerate(texts):
        for j, item in enumerate(line):
            ids[i, j] = int(item[0])
            vals[i, j] = [float(v) for v in item[1:]]
    np.savez_compressed(os.path.join(GOLDEN, "ocr_ctc.npz"), region=region, widths=np.array(widths), logits=logits.numpy(),
                        colors=colors.numpy(), ids=ids, vals=vals, dict_size=OCR_DICT, source="manga_translator/ocr/model_48px_ctc.py")
    print("ocr_ctc", logits.shape, T, [len(t) for t in texts], float(logits.std()))


def fake_detector(batch: np.ndarray, device=None):
    """Deterministic stand-in for the detector network inside det_rearrange_forward: per-pixel functions of the input squares
    (db at input resolution, mask at half resolution like the default detector's), so the golden pins the tiling / stitching
    and not a network."""
    b = np.asarray(batch).astype(np.float32) / 255.0
    n, s = b.shape[0], b.shape[1]
    ramp = (np.arange(s, dtype=np.float32) / s)[None, :, None]
    db = np.stack([b[..., 0], b[..., 1] * 0.5 + 0.25 * ramp + 0.0 * b[..., 1]], axis=1)
    m = b[..., 2]
    mask = ((m[:, 0::2, 0::2] + m[:, 0::2, 1::2] + m[:, 1::2, 0::2] + m[:, 1::2, 1::2]) * np.float32(0.25))[:, None]
    return db.astype(np.float32), mask.astype(np.float32)


def golden_rearrange():
    """The reference's own det_rearrange_forward (utils/generic.py:876-997, with square_pad_resize :848-874 over the cv2 stand-in)
    on a tall strip, a wide strip (transposed path), a strip whose squares need shrinking and a page that is not rearranged."""
    G = R.generic()
    G.cv2 = R.cv2_shim()
    import hashlib

    out = {}
    for tag, (H, W), tgt, seed in (("tall", (1400, 150), 320, 21), ("wide", (130, 1000), 192, 22), ("shrink", (2000, 350), 256, 23),
                                   ("none", (900, 600), 512, 24)):
        page = synth.synth_page(seed, H, W, n_boxes=6)[0]
        db, mask = G.det_rearrange_forward(page, fake_detector, tgt, 4, "cpu", False)
        out[f"shape_{tag}"], out[f"tgt_{tag}"], out[f"seed_{tag}"] = np.array([H, W]), tgt, seed  # the page is synth_page(seed, H, W, n_boxes=6)
        if db is None:
            assert tag == "none" and mask is None
            continue
        db, mask = np.ascontiguousarray(db, dtype=np.float32), np.ascontiguousarray(mask, dtype=np.float32)
        # full maps as sha256 of their float32 bytes (bit-exactness is the bar), plus shapes and a strided sample for diagnostics
        out[f"db_shape_{tag}"], out[f"mask_shape_{tag}"] = np.array(db.shape), np.array(mask.shape)
        out[f"db_sha_{tag}"], out[f"mask_sha_{tag}"] = hashlib.sha256(db.tobytes()).hexdigest(), hashlib.sha256(mask.tobytes()).hexdigest()
        out[f"db_sample_{tag}"], out[f"mask_sample_{tag}"] = db[..., ::7, ::5].copy(), mask[..., ::7, ::5].copy()
        print("rearrange", tag, db.shape, mask.shape, float(db.mean()), float(mask.mean()))
    np.savez_compressed(os.path.join(GOLDEN, "rearrange.npz"), source="manga_translator/utils/generic.py:848-997", **out)


def golden_direction():
    """The reference's own merge test + direction vote (utils/generic.py:653-698, ocr/common.py:12-39) on random line sets,
    executed with the shapely shim of ref_import (networkx is the real one)."""
    import itertools
    from collections import Counter

    import networkx as nx

    G = R.generic()
    shim = R.shapely_shim()
    G.Polygon, G.MultiPoint = shim.Polygon, shim.MultiPoint
    rng = np.random.default_rng(17)
    sets, merges, orders, dirs = [], [], [], []
    for trial in range(6):
        quads = []
        n_col = int(rng.integers(2, 5))
        x = 30
        for c in range(n_col):  # vertical text columns packed closely (they merge), plus a few stray horizontal lines
            w, h = int(rng.integers(18, 26)), int(rng.integers(120, 260))
            y = int(rng.integers(20, 60))
            q = np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]])
            if c % 2 == 1:
                q = q + rng.integers(-2, 3, size=(4, 2))
            quads.append(q)
            x += w + int(rng.integers(2, 14))
        for _ in range(int(rng.integers(1, 4))):
            w, h = int(rng.integers(90, 200)), int(rng.integers(16, 24))
            x0, y0 = int(rng.integers(10, 300)), int(rng.integers(300, 420))
            quads.append(np.array([[x0, y0], [x0 + w, y0], [x0 + w, y0 + h], [x0, y0 + h]]))
            if rng.random() < 0.5:  # a second line right below it: the pair merges
                quads.append(np.array([[x0 + 2, y0 + h + 3], [x0 + w, y0 + h + 3], [x0 + w, y0 + 2 * h + 3], [x0 + 2, y0 + 2 * h + 3]]))
        objs = [G.Quadrilateral(q.copy(), "", 0) for q in quads]
        m = np.zeros((len(objs), len(objs)), dtype=np.uint8)
        gr = nx.Graph()
        for i in range(len(objs)):
            gr.add_node(i)
        for (u, a), (v, b) in itertools.combinations(enumerate(objs), 2):
            if G.quadrilateral_can_merge_region(a, b, aspect_ratio_tol=1):
                m[u, v] = m[v, u] = 1
                gr.add_edge(u, v)
        order, dd = [], []
        for node_set in nx.algorithms.components.connected_components(gr):  # ocr/common.py:27-39
            nodes = list(node_set)
            majority = Counter([objs[i].direction for i in nodes]).most_common(1)[0][0]
            if majority == "h":
                nodes = sorted(nodes, key=lambda i: objs[i].aabb.y + objs[i].aabb.h // 2)
            else:
                nodes = sorted(nodes, key=lambda i: -(objs[i].aabb.x + objs[i].aabb.w))
            order += nodes
            dd += [majority] * len(nodes)
        sets.append(np.array(quads))
        merges.append(m)
        orders.append(np.array(order))
        dirs.append(np.array(dd))
    np.savez_compressed(os.path.join(GOLDEN, "direction.npz"), n_sets=len(sets), source="manga_translator/utils/generic.py + ocr/common.py",
                        **{f"quads{i}": s for i, s in enumerate(sets)}, **{f"merge{i}": m for i, m in enumerate(merges)},
                        **{f"order{i}": o for i, o in enumerate(orders)}, **{f"dir{i}": d for i, d in enumerate(dirs)})
    print("direction", [int(m.sum() // 2) for m in merges], [d.tolist() for d in dirs][:2])


def golden_refine_mask():
    """The reference's refine_mask (ctd_utils/textmask.py:158-174) and enlarge_window, executed with the cv2 stand-in, on
    a synthetic page whose predicted mask is a blurred version of the text boxes."""
    from scipy import ndimage

    from manga_image_translator_amd import textline as TL

    tm = R.textmask()
    page, quads, _ = synth.synth_page(21, 384, 320, n_boxes=6)
    rng = np.random.default_rng(3)
    page = np.clip(page.astype(np.int32) + rng.integers(-6, 7, size=page.shape), 0, 255).astype(np.uint8)  # three distinct channels
    pred = np.zeros((384, 320), np.float32)
    for q in quads:
        pred[q[0, 1]:q[2, 1], q[0, 0]:q[2, 0]] = 1.0
    pred = (ndimage.gaussian_filter(pred, 2.0) * 255).astype(np.uint8)
    lines = [TL.Quadrilateral(q) for q in quads]  # refine_mask only reads .xyxy
    out_none = tm.refine_mask(page, pred.copy(), lines, refine_mode=None)
    out_inpaint = tm.refine_mask(page, pred.copy(), lines, refine_mode=tm.REFINEMASK_INPAINT)
    np.savez_compressed(os.path.join(GOLDEN, "refine_mask.npz"), page=page, pred=pred, quads=np.array(quads), out_none=out_none,
                        out_inpaint=out_inpaint, source="manga_translator/detection/ctd_utils/textmask.py")
    print("refine_mask", float((out_none > 0).mean()), float((out_inpaint > 0).mean()), float((pred > 60).mean()))


def golden_textline_merge():
    """The reference's own golden tests for the text-line merge (test/test_textline_merge.py): the line sets and expected
    groupings are extracted from the test source, and the reference's merge_bboxes_text_region is run on them (shapely
    stand-in, real networkx) to record what the code itself returns (groups in yield order, colours)."""
    import ast
    import importlib.util
    import json

    src = open(os.path.join(R.REF_ROOT, "test", "test_textline_merge.py")).read()
    cases = []
    for node in ast.parse(src).body:
        if isinstance(node, ast.AsyncFunctionDef) and node.name.startswith("test_merge"):
            env = {}
            for st in node.body:
                if isinstance(st, ast.Assign):
                    exec(compile(ast.Module([st], []), "<case>", "exec"), {}, env)
            cases.append(dict(name=node.name, width=env["width"], height=env["height"], lines=env["lines"],
                              expected=env["expected_combinations"]))
    G = R.generic()
    shim = R.shapely_shim()
    G.Polygon, G.MultiPoint = shim.Polygon, shim.MultiPoint
    import sys
    import types

    utils = types.ModuleType("manga_translator.utils")
    utils.TextBlock = type("TextBlock", (), {})
    utils.Quadrilateral, utils.quadrilateral_can_merge_region = G.Quadrilateral, G.quadrilateral_can_merge_region
    saved = sys.modules.get("manga_translator.utils")
    sys.modules["manga_translator.utils"] = utils
    try:
        spec = importlib.util.spec_from_file_location("manga_translator.textline_merge", os.path.join(R.PKG, "textline_merge", "__init__.py"))
        TM = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(TM)
    finally:
        if saved is not None:
            sys.modules["manga_translator.utils"] = saved
    TM.Polygon = shim.Polygon
    rng = np.random.default_rng(5)
    for c in cases:
        cols = rng.integers(0, 256, size=(len(c["lines"]), 6))
        quads = [G.Quadrilateral(np.array(l), "", 1, *[int(v) for v in col]) for l, col in zip(c["lines"], cols)]
        groups, colors = [], []
        for txtlns, fg, bg in TM.merge_bboxes_text_region(quads, c["width"], c["height"]):
            groups.append([next(i for i, q in enumerate(quads) if q is t) for t in txtlns])
            colors.append([list(map(int, fg)), list(map(int, bg))])
        c["colors_in"] = cols.tolist()
        c["ref_groups"], c["ref_colors"] = groups, colors
        c["ref_passes_own_test"] = sorted(map(sorted, groups)) == sorted(map(sorted, c["expected"]))
    json.dump(dict(source="test/test_textline_merge.py + manga_translator/textline_merge/__init__.py", cases=cases),
              open(os.path.join(GOLDEN, "textline_merge.json"), "w"))
    print("textline_merge", len(cases), [c["ref_passes_own_test"] for c in cases])


def mask_refinement_scene(seed=3, H=360, W=300):
    """Synthetic page for the mask-refinement pin: text lines (two of them rotated), stroke-like blobs inside them, strays just
    outside (adopted through the distance rule), far strays (dropped), specks (<= 9 px) and one blob larger than its line."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), dtype=np.uint8)
    boxes = [(30, 40, 200, 36), (40, 100, 30, 200), (120, 120, 150, 30), (110, 200, 160, 44), (20, 320, 120, 24)]
    lines = []
    for k, (x, y, w, h) in enumerate(boxes):
        q = np.array([[x, y], [x + w, y], [x + w, y + h], [x, y + h]], dtype=np.float64)
        if k in (2, 3):  # a few degrees of rotation about the centre
            a = np.deg2rad(6 if k == 2 else -4)
            c = q.mean(0)
            q = (q - c) @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]).T + c
        lines.append(np.rint(q).astype(np.int32))
        for _ in range(14):  # strokes
            bw, bh = int(rng.integers(3, 12)), int(rng.integers(3, 12))
            bx, by = int(rng.integers(x + 2, max(x + 3, x + w - bw - 2))), int(rng.integers(y + 2, max(y + 3, y + h - bh - 2)))
            mask[by:by + bh, bx:bx + bw] = 255
    for (bx, by, bw, bh) in [(232, 60, 5, 8), (236, 50, 6, 8), (262, 44, 9, 9), (5, 250, 7, 7), (150, 300, 3, 3), (280, 340, 2, 4), (100, 160, 5, 9)]:
        mask[by:by + bh, bx:bx + bw] = 255
    mask[318:350, 18:150] = 255  # larger than line 4: never assigned
    mask[rng.random((H, W)) < 0.002] = 255
    return img, mask, np.stack(lines)


def mask_refinement_bubble_page(img):
    g = np.repeat(img[..., :1], 3, axis=2).copy()
    g[44:70, 60:140] = img[44:70, 60:140]          # colour under the first text line
    g[330:, :] = 240                                # bright band reaching the bottom frame, under the last line
    return g


def mask_refinement_stubs():
    refine = lambda rgb, m: np.where(rgb[..., 1] > 40, m, 0).astype(np.uint8)  # stands in for the DenseCRF (uses both crops)
    bilateral = lambda img, *a: (img // 4) * 4                                   # stands in for cv2.bilateralFilter
    return refine, bilateral


def golden_mask_refinement():
    """mask_refinement.dispatch / complete_mask of the reference (cv2 / shapely stand-ins; DenseCRF and bilateralFilter stubbed — see
    ref_import.mask_refinement) on the synthetic scene above."""
    import asyncio

    refine, bilateral = mask_refinement_stubs()
    mr, tmu, G = R.mask_refinement(refine_stub=refine, bilateral_stub=bilateral)
    img, mask, lines = mask_refinement_scene()
    out = {"img": img, "mask": mask, "lines": lines}
    for tag, off, ks in (("a", 0, 3), ("b", 6, 5)):
        quads = [G.Quadrilateral(l.astype(np.float64), "", 0) for l in lines]
        m = mask.copy()
        out[f"complete_{tag}"] = tmu.complete_mask(img.copy(), m, quads, dilation_offset=off, kernel_size=ks)
        out[f"complete_{tag}_mask_after"] = m
        region = type("Region", (), {"lines": lines})()
        out[f"dispatch_{tag}"] = asyncio.run(mr.dispatch([region], img.copy(), mask.copy(), "fit_text", off, 0, False, ks))
    out["dispatch_none"] = asyncio.run(mr.dispatch([type("Region", (), {"lines": np.zeros((0, 4, 2), np.int32)})()], img.copy(),
                                                   np.zeros_like(mask), "fit_text", 0, 0, False, 3))
    # the --ignore-bubble stage (dispatch :34-50) with the reference's own is_ignore: a grey page (no colour anywhere) with a coloured
    # patch under the first line and a bright band along the bottom frame under the last one
    mrb, _, _ = R.mask_refinement(refine_stub=refine, bilateral_stub=bilateral, with_bubble=True)
    img_b = mask_refinement_bubble_page(img)
    out["img_bubble"] = img_b
    region = type("Region", (), {"lines": lines})()
    for lv in (10, 40):
        out[f"dispatch_bubble{lv}"] = asyncio.run(mrb.dispatch([region], img_b.copy(), mask.copy(), "fit_text", 0, lv, False, 3))
    np.savez_compressed(os.path.join(GOLDEN, "mask_refinement.npz"), **out)
    print("mask_refinement", {k: (v.shape, int(v.sum() // 255) if v.dtype == np.uint8 and v.ndim == 2 else "") for k, v in out.items()})


def build_ref_dbnet():
    from manga_image_translator_amd import dbnet_schema

    M = R.dbnet()
    net = M.TextDetection()
    sd = synth.synth_state_dict(dbnet_schema.text_detection_schema(), gain=1.2)
    net.load_state_dict(sd, strict=True)
    return net.eval(), sd


def golden_dbnet():
    net, _ = build_ref_dbnet()
    page = synth.synth_page(13, 256, 256, n_boxes=3)[0]
    x = torch.from_numpy(page[None].astype(np.float32) / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()  # default.py:19
    with torch.no_grad():
        db, mask = net(x)
    np.savez_compressed(os.path.join(GOLDEN, "dbnet.npz"), page=page, db=db.sigmoid().numpy().astype(np.float16),
                        db_logit_stats=np.array([float(db[:, 0].mean()), float(db[:, 0].std())]),
                        mask=mask.numpy().astype(np.float16), db_f32_crop=db.sigmoid().numpy()[:, :, 96:160, 96:160],
                        mask_f32_crop=mask.numpy()[:, :, 32:96, 32:96], source="manga_translator/detection/default_utils/DBNet_resnet34.py")
    print("dbnet", db.shape, mask.shape, float(db.sigmoid().mean()), float(db.sigmoid().std()), float(mask.mean()), float(mask.std()))


def build_ref_esrgan(nb: int):
    from manga_image_translator_amd import esrgan_schema

    E = R.esrgan()
    net = E.RRDBNet(in_nc=3, out_nc=3, nf=64, nb=nb, upscale=4, plus=False)
    sd = synth.synth_state_dict(esrgan_schema.rrdbnet_schema(nb))
    net.load_state_dict(sd, strict=True)
    return net.eval(), sd


def golden_esrgan():
    nb = 2
    net, _ = build_ref_esrgan(nb)
    page = synth.synth_page(11, 40, 56, n_boxes=2)[0]
    x = torch.from_numpy(page[:, :, ::-1].copy()).float().div(255.0).permute(2, 0, 1).unsqueeze(0)  # _infer :541
    with torch.no_grad():
        y = net(x)
    out = (y[0].clip(0, 1).permute(1, 2, 0).numpy()[:, :, ::-1].copy() * 255.0).astype(np.uint8)  # :545
    np.savez_compressed(os.path.join(GOLDEN, "esrgan.npz"), page=page, out_float=y.numpy(), out_u8=out, nb=nb,
                        source="manga_translator/upscaling/esrgan_pytorch.py")
    print("esrgan", y.shape, float(y.mean()), float(y.std()))


def boxes_scene(seed: int, H: int = 128, W: int = 160) -> np.ndarray:
    """A probability map with rotated text-line blobs of different confidence, a speck, a blob with a hole and one touching the
    border: every branch of boxes_from_bitmap (size filters, score filter, hole contours, clipping to the destination size)."""
    from PIL import Image, ImageDraw

    rng = np.random.default_rng(seed)
    im = Image.new("F", (W, H), 0.02)
    d = ImageDraw.Draw(im)
    for _ in range(7):
        cx, cy = int(rng.integers(10, W - 10)), int(rng.integers(8, H - 8))
        w, h, a = int(rng.integers(8, 50)), int(rng.integers(3, 14)), float(rng.uniform(0, np.pi))
        c, s_ = np.cos(a), np.sin(a)
        d.polygon([(cx + c * x - s_ * y, cy + s_ * x + c * y) for x, y in ((-w, -h), (w, -h), (w, h), (-w, h))],
                  fill=float(rng.uniform(0.45, 0.97)))
    arr = np.asarray(im).copy()
    arr[5:7, 5:8] = 0.9                               # a speck (short side < 2 / 3)
    arr[H - 4:, W // 3:W // 3 + 40] = 0.88            # a blob on the bottom border
    ys, xs = np.nonzero(arr > 0.6)
    if len(ys):
        k = len(ys) // 2
        arr[max(ys[k] - 2, 0):ys[k] + 2, max(xs[k] - 2, 0):xs[k] + 2] = 0.05  # a hole inside a blob
    return arr.astype(np.float32)


def bubble_crops(n=40, seed=23):
    """Rectified-line-like crops for the ignore-bubble filter: white / black bubbles with glyph strokes inside, artwork-like noise, a
    clean frame with a coloured interior, and frames whose dark share sits exactly on the thresholds."""
    rng = np.random.default_rng(seed)
    crops = []
    for k in range(n):
        w = int(rng.integers(24, 120))
        kind = k % 5
        if kind == 0:    # white bubble, dark strokes away from the frame
            img = np.full((48, w, 3), int(rng.integers(200, 256)), np.uint8)
            img[8:40, 6:w - 6][rng.random((32, w - 12)) < 0.3] = int(rng.integers(0, 60))
        elif kind == 1:  # black bubble
            img = np.full((48, w, 3), int(rng.integers(0, 100)), np.uint8)
            img[8:40, 6:w - 6][rng.random((32, w - 12)) < 0.3] = int(rng.integers(180, 256))
        elif kind == 2:  # artwork: grey noise everywhere (frame share near 50 %)
            g = rng.integers(0, 256, (48, w, 1)).astype(np.uint8)
            img = np.repeat(g, 3, axis=2)
        elif kind == 3:  # white frame, coloured interior
            img = np.full((48, w, 3), 250, np.uint8)
            img[10:38, 8:w - 8] = rng.integers(0, 256, (28, w - 16, 3)).astype(np.uint8)
        else:            # a frame with a controlled share of dark values (thresholds 10 / 90 % and around)
            img = np.full((48, w, 3), 255, np.uint8)
            frame = np.zeros((48, w), bool)
            frame[:2] = frame[-2:] = True
            frame[:, :2] = frame[:, -2:] = True
            ys, xs = np.nonzero(frame)
            share = [0.05, 0.1, 0.100001, 0.5, 0.9, 0.95, 0.0999][(k // 5) % 7]
            pick = rng.permutation(len(ys))[:int(round(share * len(ys)))]
            img[ys[pick], xs[pick]] = 0
        crops.append(img)
    return crops


def golden_bubble():
    """utils/bubble.py is_ignore — the reference's own function (cv2.threshold from the stand-in) — on bubble_crops() for several
    --ignore-bubble values, including the out-of-range ones that switch the filter off."""
    B = R.bubble()
    crops = bubble_crops()
    levels = [0, 1, 5, 10, 25, 50, 51]
    want = np.array([[bool(B.is_ignore(c, lv)) for lv in levels] for c in crops])
    np.savez_compressed(os.path.join(GOLDEN, "bubble.npz"), levels=np.array(levels), want=want, widths=np.array([c.shape[1] for c in crops]))


def golden_boxes():
    """SegDetectorRepresenter of both detectors — the reference's own Python (db_utils.py:127-216, dbnet_utils.py:97-190) run with
    the cv2 / pyclipper / shapely stand-ins of ref_import.segdet — on seeded probability maps."""
    out = {}
    ctd = R.segdet("ctd").SegDetectorRepresenter(thresh=0.3)                      # ctd.py:102
    dbn = R.segdet("default")
    for i, (seed, dh, dw) in enumerate(((3, 256, 320), (4, 128, 160), (5, 300, 333))):
        pred = boxes_scene(seed)
        lm = np.stack([pred, pred])[None]
        b, s = ctd(None, lm, height=dh, width=dw)
        out[f"pred{i}"], out[f"dest{i}"] = pred, np.array([dh, dw])
        out[f"ctd_boxes{i}"], out[f"ctd_scores{i}"] = np.asarray(b[0], dtype=np.int64), np.asarray(s[0], dtype=np.float32)
        for j, (tt, bt, ur) in enumerate(((0.5, 0.7, 2.3), (0.3, 0.5, 1.5))):      # default.py:73-77 (config defaults) + a looser set
            det = dbn.SegDetectorRepresenter(tt, bt, unclip_ratio=ur)
            b2, s2 = det({"shape": [(dh, dw)]}, lm)
            out[f"dbnet_boxes{i}_{j}"], out[f"dbnet_scores{i}_{j}"] = np.asarray(b2[0], dtype=np.int64), np.asarray(s2[0], dtype=np.float32)
    out["dbnet_params"] = np.array([[0.5, 0.7, 2.3], [0.3, 0.5, 1.5]])
    np.savez_compressed(os.path.join(GOLDEN, "boxes.npz"), **out)


def share_stream_scene():
    """A worker's ``/execute`` byte stream (mode/share.py:63-66 framing) and a schedule of network chunk boundaries that cuts inside
    headers, inside payloads and between frames."""
    rng = np.random.default_rng(41)
    big = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    frames = [(1, "detection".encode()), (1, b""), (1, "文字認識 ocr".encode("utf-8")), (0, big), (1, b"x"), (2, "stage exploded".encode()), (0, b"")]
    stream = b"".join(bytes([st]) + len(pl).to_bytes(4, "big") + pl for st, pl in frames)
    cuts = sorted(set([0, 1, 3, 5, 6, 14, 15, 19, 20, 21, 40, 41 + 65536, len(stream) - 30, len(stream) - 5, len(stream) - 4, len(stream) - 1, len(stream)]))
    return stream, cuts


def golden_share_stream():
    """server/sent_data_internal.py:36-66 (process_stream's loop body + handle_buffer + extract_header), the reference's own functions, run
    over ``share_stream_scene``: what the front server's ``sender`` receives after every network chunk."""
    import hashlib
    import json

    C = R.server_client()
    stream, cuts = share_stream_scene()
    calls, buffer, per_chunk = [], b"", []
    for a, b in zip(cuts[:-1], cuts[1:]):
        n0 = len(calls)
        buffer += stream[a:b]                                            # process_stream: buffer += chunk
        buffer = C.handle_buffer(buffer, lambda st, data: calls.append((int(st), bytes(data))))
        per_chunk.append({"end": b, "delivered": len(calls) - n0, "left_in_buffer": len(buffer)})
    out = {"stream_sha256": hashlib.sha256(stream).hexdigest(), "stream_len": len(stream), "cuts": cuts, "per_chunk": per_chunk,
           "calls": [{"status": st, "len": len(d), "sha256": hashlib.sha256(d).hexdigest(), "head": d[:16].hex()} for st, d in calls],
           "header_of_first_frame": list(C.extract_header(stream[:5]))}
    with open(os.path.join(GOLDEN, "share_stream.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    if not R.available():
        raise SystemExit("/root/reference is not present: fixtures can only be regenerated in the build container")
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    golden_textline()
    golden_ocr()
    golden_ctd()
    golden_lama()
    golden_lama_resize()
    golden_esrgan()
    golden_ocr_ctc()
    golden_dbnet()
    golden_direction()
    golden_rearrange()
    golden_refine_mask()
    golden_textline_merge()
    golden_mask_refinement()
    golden_boxes()
    golden_bubble()
    golden_share_stream()


if __name__ == "__main__":
    main()
