"""One-off table (build container only: needs /root/reference): the CPU oracle that bench.py times as ``cpu_baseline`` (kind "port")
against the REFERENCE's own modules on the same inputs, same weights, same thread count — how representative is the port of the
reference's CPU speed?  Prints one JSON line (committed as profiles/r03_cpu_oracle_vs_reference.json)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from manga_image_translator_amd import synth
from oracle import ctd as OC, lama as OL, make_golden as MG, ocr48 as OO

torch.set_num_threads(int(os.environ.get("THREADS", "8")))
out = {"threads": torch.get_num_threads(), "host_cpus": os.cpu_count()}


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(min(ts), 3)


with torch.no_grad():
    # LaMa-MPE on a 768 x 512 page (the generator is fully convolutional)
    m, sd, mpe_sd = MG.build_ref_lama(9, True)
    page, _, mask = synth.synth_page(1, 768, 512, n_boxes=8)
    out["lama_mpe_768x512"] = {"reference_s": best(lambda: MG.ref_lama_infer(m, page, mask)), "oracle_s": best(lambda: OL.infer(sd, mpe_sd, page, mask, 9))}
    # ctd on one 2048 x 1456 page (letterboxed to 1024^2)
    fwd, shim, (ysd, ssd, dsd) = MG.build_ref_ctd()
    page = synth.synth_page(2, 2048, 1456, n_boxes=8)[0]
    x_in = OC.preprocess_img(page)[0]
    out["ctd_1024x1024"] = {"reference_s": best(lambda: fwd(x_in)), "oracle_s": best(lambda: OC.infer_maps(ysd, ssd, dsd, page))}
    # 48px OCR: one chunk of 16 lines, padded width 263, 12 decode steps (dictionary of make_golden: 97 entries)
    model, osd = MG.build_ref_ocr()
    rng = np.random.default_rng(3)
    widths = [120 + 9 * i for i in range(16)]
    Wp = max(widths) + 7
    region = np.zeros((16, 48, Wp, 3), np.uint8)
    for i, w in enumerate(widths):
        region[i, :, :w] = rng.integers(0, 256, (48, w, 3), dtype=np.uint8)
    img = ((torch.from_numpy(region).float() - 127.5) / 127.5).permute(0, 3, 1, 2)
    out["ocr48_16x263_12steps"] = {"reference_s": best(lambda: model.infer_beam_batch_tensor(img, widths, beams_k=5, max_seq_length=12), n=2),
                                   "oracle_s": best(lambda: OO.infer_beam_batch_tensor(osd, img, widths, max_seq_length=12), n=2)}
for k, v in out.items():
    if isinstance(v, dict):
        v["oracle_over_reference"] = round(v["oracle_s"] / v["reference_s"], 3)
print(json.dumps(out))
