"""Worker-pool serving on a GPU box: two shared-mode workers (manga_translator/mode/share.py:47-174 protocol), each a
process pinned with HIP_VISIBLE_DEVICES, each loading the three HIP plugins, answer the same request with byte-identical results — alone
and while the other worker is busy on the same GPU — and equal to the plugin chain run in this process."""
import asyncio
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, W, LINES, D = 512, 384, 5, 128


def _port(n=2):
    """First port of a run of ``n`` consecutive free ports (the pool numbers its workers base, base + 1, ...)."""
    for _ in range(50):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            base = s.getsockname()[1]
        ok = True
        for k in range(1, n):
            with socket.socket() as t:
                try:
                    t.bind(("127.0.0.1", base + k))
                except OSError:
                    ok = False
                    break
        if ok:
            return base
    raise RuntimeError("no run of free ports found")


def test_two_workers_on_one_box_return_identical_pages(cuda):
    from manga_image_translator_amd import serve, synth

    page, quads, gmask = synth.synth_page(3, H, W, n_boxes=LINES, disjoint=True)
    page2 = synth.synth_page(4, H, W, n_boxes=LINES, disjoint=True)[0]
    # (synthetic weights: the detector fires on nothing and the refined mask would be empty — the request brings the generator's text
    # lines and mask, as the benchmark does, so that every stage has work)
    cfg = {"textlines": np.asarray(quads).tolist(), "mask": gmask, "ocr": {"max_seq_length": 8, "suppress_eos": True, "prob": 0.0},
           "inpainter": {"inpainting_size": 512}}
    base = _port()
    pool = serve.WorkerPool(gpus=["0", "0"], base_port=base, worker_args=["--dict-size", str(D)])   # one GPU on the box: both pinned to it
    with pool:
        a, b = pool.executors.list
        ia, ib = asyncio.run(a.sent(None, None, method="device_info")), asyncio.run(b.sent(None, None, method="device_info"))
        assert ia["visible_devices"] == ib["visible_devices"] == "0" and ia["n_visible"] == ib["n_visible"] == 1 and ia["pid"] != ib["pid"]
        ra, rb = asyncio.run(a.sent(page, cfg)), asyncio.run(b.sent(page, cfg))
        # the streaming endpoint the reference's front server uses (server/instance.py:19-20 -> sent_data_internal.py:13-58): the frames
        # end with ONE result frame whose payload is the pickled result, identical to /simple_execute's.  The client here is
        # ExecutorInstance.sent_stream, whose parser is pinned to the reference's handle_buffer by tests/golden/share_stream.json (the
        # reference tree itself does not exist on the GPU box); the image is what that client sends: a PIL image.
        import pickle

        from PIL import Image

        frames = []
        asyncio.run(a.sent_stream(Image.fromarray(page), cfg, lambda st, payload: frames.append((st, payload))))
        assert [st for st, _ in frames][-1] == 0 and all(st == 1 for st, _ in frames[:-1])
        rs = pickle.loads(frames[-1][1])
        frames = []
        asyncio.run(a.sent_stream(page[:, :, 0], cfg, lambda st, payload: frames.append((st, payload))))     # not HxWx3: the engine's error, as a frame
        assert len(frames) == 1 and frames[0][0] == 2 and b"HxWx3" in frames[0][1]

        async def both():   # the two workers busy at the same time, on different pages, then swapped
            return await asyncio.gather(a.sent(page, cfg), b.sent(page2, cfg), )
        ca, cb2 = asyncio.run(both())
        rb2 = asyncio.run(a.sent(page2, cfg))
        outs = asyncio.run(pool.map([page, page2, page, page2], cfg))
    for x, y in ((ra, rb), (ra, ca), (cb2, rb2), (outs[0], ra), (outs[1], rb2), (outs[2], ra), (outs[3], rb2)):
        assert np.array_equal(x["inpainted"], y["inpainted"]) and np.array_equal(x["mask"], y["mask"]) and np.array_equal(x["mask_raw"], y["mask_raw"])
        assert x["textlines"] == y["textlines"]
    assert np.array_equal(rs["inpainted"], ra["inpainted"]) and np.array_equal(rs["mask"], ra["mask"]) and rs["textlines"] == ra["textlines"]
    assert ra["inpainted"].shape == (H, W, 3) and ra["inpainted"].dtype == np.uint8 and len(ra["textlines"]) >= 1
    assert not np.array_equal(ra["inpainted"], page)            # something was inpainted
    # the same chain in this process (the plugins called directly): what a worker returns is what the plugins compute
    eng = serve.DenseStages({"dict_size": D})
    here = asyncio.new_event_loop().run_until_complete(eng.translate(page, cfg))
    assert np.array_equal(here["inpainted"], ra["inpainted"]) and np.array_equal(here["mask"], ra["mask"]) and here["textlines"] == ra["textlines"]
