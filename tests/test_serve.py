"""Worker-pool serving without a GPU: the shared-mode worker protocol (manga_translator/mode/share.py:47-174), the
executor table (server/instance.py:35-65) and the one-process-per-GPU pinning, with a stand-in engine inside the workers."""
import asyncio
import os
import pickle
import socket
import urllib.error
import urllib.request

import numpy as np
import pytest

from manga_image_translator_amd import serve


def _free_ports(n):
    socks = [socket.socket() for _ in range(n + 2)]
    for s in socks:
        s.bind(("127.0.0.1", 0))
    ports = sorted(s.getsockname()[1] for s in socks)
    for s in socks:
        s.close()
    for i in range(len(ports) - n + 1):   # a run of n consecutive free ports is not guaranteed: fall back to the first
        if ports[i + n - 1] - ports[i] == n - 1:
            return ports[i]
    return ports[0]


def test_frames_round_trip_and_partial_buffers():
    chunks = serve.frame(1, b"detection") + serve.frame(1, b"") + serve.frame(0, b"x" * 70000)
    got, rest = serve.parse_frames(chunks[:-10])
    assert got == [(1, b"detection"), (1, b"")] and len(rest) == 5 + 70000 - 10          # the incomplete frame stays in the buffer
    got2, rest2 = serve.parse_frames(rest + chunks[-10:])
    assert got2 == [(0, b"x" * 70000)] and rest2 == b""
    assert serve.frame(2, b"e")[:5] == b"\x02\x00\x00\x00\x01"                            # status, 4-byte big-endian length


def test_restricted_unpickler_takes_arrays_and_refuses_code():
    a = {"image": np.arange(12, dtype=np.uint8).reshape(2, 2, 3), "config": {"ocr": {"prob": 0.1}}}
    b = serve.restricted_loads(pickle.dumps(a))
    assert np.array_equal(b["image"], a["image"]) and b["config"] == a["config"]
    with pytest.raises(pickle.UnpicklingError):
        serve.restricted_loads(pickle.dumps(os.getcwd))          # posix.getcwd: not on the allow-list (mode/share.py:14-33)


def test_executor_table_hands_out_free_workers_and_waits_when_all_are_busy():
    async def go():
        ex = serve.Executors()
        ex.register(serve.ExecutorInstance("127.0.0.1", 1))
        ex.register(serve.ExecutorInstance("127.0.0.1", 2))
        a, b = await ex.find_executor(), await ex.find_executor()
        assert {a.port, b.port} == {1, 2} and ex.free_executors() == 0
        waiter = asyncio.create_task(ex.find_executor())
        await asyncio.sleep(0.05)
        assert not waiter.done()
        await ex.free_executor(b)
        c = await asyncio.wait_for(waiter, 2)
        assert c is b and c.busy
    asyncio.run(go())


def test_pool_starts_one_pinned_worker_per_gpu_and_serves_requests():
    """Two workers for "GPUs" 3 and 5: each process must see exactly its own device id; the protocol rules of the reference worker hold
    (nonce -> 401, busy -> 429, unknown / private method -> 404 / 403, an engine error -> 500 or an error frame); requests spread over
    both workers."""
    img = np.arange(4 * 5 * 3, dtype=np.uint8).reshape(4, 5, 3)
    pool = serve.WorkerPool(gpus=["3", "5"], base_port=_free_ports(2), env={"MIT_SERVE_ENGINE": "tests._serve_stub:make"})
    with pool:
        a, b = pool.executors.list
        ra, rb = asyncio.run(a.sent(img, {})), asyncio.run(b.sent(img, {}))
        assert ra["visible"] == "3" and rb["visible"] == "5" and ra["pid"] != rb["pid"]
        assert ra["sum"] == rb["sum"] == int(img.sum()) and ra["shape"] == [4, 5, 3]
        # nonce
        bad = serve.ExecutorInstance(a.ip, a.port, nonce="wrong")
        with pytest.raises(RuntimeError, match="HTTP 401"):
            asyncio.run(bad.sent(img, {}))
        with pytest.raises(RuntimeError, match="HTTP 404"):
            asyncio.run(a.sent(img, {}, method="nothing_here"))
        with pytest.raises(RuntimeError, match="HTTP 403"):
            asyncio.run(a.sent(img, {}, method="_private"))
        with pytest.raises(RuntimeError, match="HTTP 500.*stage exploded"):
            asyncio.run(a.sent(img, {}, method="fail"))
        assert asyncio.run(a.sent(img, {}))["sum"] == int(img.sum())          # the lock was released after the failure

        async def busy():   # a second request to a worker that is executing one: 429, and /is_locked says so meanwhile
            t = asyncio.create_task(a.sent(img, {"sleep": 1.0}))
            await asyncio.sleep(0.3)
            with urllib.request.urlopen(f"{a.url}/is_locked") as r:
                locked = r.read()
            with pytest.raises(RuntimeError, match="HTTP 429"):
                await a.sent(img, {})
            await t
            return locked
        assert b"true" in asyncio.run(busy())

        frames = []
        asyncio.run(a.sent_stream(img, {}, lambda st, payload: frames.append((st, payload))))
        assert [f[0] for f in frames] == [1, 1, 0] and frames[0][1] == b"detection" and pickle.loads(frames[-1][1])["sum"] == int(img.sum())
        frames = []
        asyncio.run(a.sent_stream(img, {}, lambda st, payload: frames.append((st, payload)), method="fail"))
        assert frames == [(2, b"stage exploded")]

        batch = asyncio.run(b.sent_batch([img, img + 1, img + 2], {}, batch_size=2))      # server/instance.py:22-26
        assert [o["sum"] for o in batch] == [int((img + i).sum()) for i in range(3)] and {o["visible"] for o in batch} == {"5"}

        outs = asyncio.run(pool.map([img + i for i in range(6)], {"sleep": 0.2}))
        assert [o["sum"] for o in outs] == [int((img + i).sum()) for i in range(6)]        # results in request order
        assert {o["visible"] for o in outs} == {"3", "5"}                                  # both workers took requests
    assert not pool.procs


# ---- pinned to the reference's own client -----------------------------------------------------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_parse_frames_equals_the_reference_handle_buffer_on_the_golden_stream():
    """tests/golden/share_stream.json holds what the reference's ``handle_buffer`` (server/sent_data_internal.py:44-66) delivered, chunk
    by chunk, for a stream cut inside headers, inside payloads and between frames (oracle/make_golden.py:golden_share_stream, regenerated
    and compared by tests/test_oracle_vs_reference.py).  This package's client parser must deliver the same frames after the same
    chunks — which is what lets the GPU-box test (no reference tree there) drive the worker with ``ExecutorInstance.sent_stream``."""
    import hashlib
    import json

    from oracle import make_golden as MG

    g = json.load(open(os.path.join(GOLDEN, "share_stream.json")))
    stream, cuts = MG.share_stream_scene()
    assert hashlib.sha256(stream).hexdigest() == g["stream_sha256"] and cuts == g["cuts"] and len(stream) == g["stream_len"]
    buf, calls = b"", []
    for (a, b), want in zip(zip(cuts[:-1], cuts[1:]), g["per_chunk"]):
        frames, buf = serve.parse_frames(buf + stream[a:b])
        calls += frames
        assert (b, len(frames), len(buf)) == (want["end"], want["delivered"], want["left_in_buffer"])
    assert [(st, len(d), hashlib.sha256(d).hexdigest()) for st, d in calls] == [(c["status"], c["len"], c["sha256"]) for c in g["calls"]]
    assert [c["status"] for c in g["calls"]] == [1, 1, 1, 0, 1, 2, 0] and buf == b""
    # and the worker's framing is the inverse: frame() of the delivered calls reproduces the stream byte for byte
    assert b"".join(serve.frame(st, d) for st, d in calls) == stream


def _serve_in_thread(worker):
    """The stand-alone worker's FastAPI app on a uvicorn server in a thread of THIS process (so that the request body's
    ``manga_translator.Config`` unpickles against the stand-in oracle/ref_import.py installs)."""
    import threading
    import time

    import uvicorn

    server = uvicorn.Server(uvicorn.Config(worker.app(), host=worker.host, port=worker.port, log_level="warning"))
    th = threading.Thread(target=server.run, daemon=True)
    th.start()
    t_end = time.time() + 30
    while not server.started:
        if time.time() > t_end:
            raise TimeoutError("uvicorn did not start")
        time.sleep(0.05)
    return server, th


@pytest.mark.skipif(not os.path.isdir("/root/reference/server"), reason="/root/reference not present (build container only)")
def test_reference_client_drives_the_worker(monkeypatch):
    """The reference's front-server client — ``fetch_data_stream`` -> ``process_stream`` -> ``handle_buffer``
    (server/sent_data_internal.py:13-66, loaded by path, unmodified) — posts ``{"image": PIL image, "config": Config}`` to this
    package's worker ``/execute/translate`` and receives progress, progress, result (1, 1, 0) with a payload that unpickles; an engine
    failure arrives as ONE error frame (2); a wrong nonce is the HTTPException the reference's server would relay (401)."""
    from fastapi import HTTPException
    from PIL import Image

    from oracle import ref_import as R

    C = R.server_client()
    monkeypatch.setenv("MIT_SERVE_ENGINE", "tests._serve_stub:make")
    worker = serve.make_worker({"host": "127.0.0.1", "port": _free_ports(1), "nonce": "n0nce"})
    assert isinstance(worker, serve.HipShareWorker)
    server, th = _serve_in_thread(worker)
    try:
        url = f"http://127.0.0.1:{worker.port}/execute/"
        arr = np.arange(6 * 7 * 3, dtype=np.uint8).reshape(6, 7, 3)
        img = Image.fromarray(arr)
        cfg = R.RefConfig(detector={"detection_size": 1024}, ocr={"prob": None})
        got = []
        asyncio.run(C.fetch_data_stream(url + "translate", img, cfg, lambda st, data: got.append((st, data)), headers={"X-Nonce": "n0nce"}))
        assert [st for st, _ in got] == [1, 1, 0] and got[0][1] == b"detection" and got[1][1] == b"finished"
        res = pickle.loads(got[-1][1])
        assert res["sum"] == int(arr.astype(np.int64).sum()) and res["shape"] == [6, 7, 3]
        got = []
        asyncio.run(C.fetch_data_stream(url + "fail", img, cfg, lambda st, data: got.append((st, data)), headers={"X-Nonce": "n0nce"}))
        assert got == [(2, b"stage exploded")]
        with pytest.raises(HTTPException) as e:
            asyncio.run(C.fetch_data_stream(url + "translate", img, cfg, lambda st, data: None, headers={"X-Nonce": "wrong"}))
        assert e.value.status_code == 401
        with pytest.raises(HTTPException) as e:
            asyncio.run(C.fetch_data_stream(url + "nothing_here", img, cfg, lambda st, data: None, headers={"X-Nonce": "n0nce"}))
        assert e.value.status_code == 404
        # the worker is free again after each of these (the lock is released on every path)
        with urllib.request.urlopen(f"http://127.0.0.1:{worker.port}/is_locked") as r:
            assert b"false" in r.read()
    finally:
        server.should_exit = True
        th.join(timeout=20)


def test_make_worker_is_the_references_mangashare_when_it_imports(monkeypatch):
    """With ``manga_translator.mode.share`` importable the worker object is the reference's own class (plugins registered first, the
    checkpoint root handed to ModelWrapper) — the stand-alone mirror is only for images where the reference cannot be imported."""
    import sys
    import types

    from manga_image_translator_amd import plugins as P

    made, registered, dirs = [], [], []

    class MangaShare:
        def __init__(self, params):
            made.append(params)

    for name in ("manga_translator", "manga_translator.mode"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            monkeypatch.setitem(sys.modules, name, m)
    share = types.ModuleType("manga_translator.mode.share")
    share.MangaShare = MangaShare
    monkeypatch.setitem(sys.modules, "manga_translator.mode.share", share)
    monkeypatch.setattr(P, "register", lambda: registered.append(True))
    monkeypatch.setattr(P, "set_model_dir", lambda d: dirs.append(d))
    monkeypatch.delenv("MIT_SERVE_ENGINE", raising=False)
    w = serve.make_worker({"port": 1, "model_dir": "/data/ckpt"})
    assert isinstance(w, MangaShare) and registered == [True] and dirs == ["/data/ckpt"] and made[0]["port"] == 1


def test_pinning_environment_for_hip_and_rocr_lists(monkeypatch):
    """an entry of ROCR_VISIBLE_DEVICES is narrowed at the ROCr level (HIP ordinals would index INTO the filtered list)."""
    for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,GPU-abc")
    gpus = serve.visible_gpus()
    assert gpus == [("ROCR_VISIBLE_DEVICES", "4"), ("ROCR_VISIBLE_DEVICES", "GPU-abc")]
    e = serve.pin_env({"ROCR_VISIBLE_DEVICES": "4,GPU-abc", "HIP_VISIBLE_DEVICES": "1", "CUDA_VISIBLE_DEVICES": "0"}, gpus[1])
    assert e["ROCR_VISIBLE_DEVICES"] == "GPU-abc" and "HIP_VISIBLE_DEVICES" not in e and "CUDA_VISIBLE_DEVICES" not in e
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert serve.visible_gpus() == [("HIP_VISIBLE_DEVICES", "1"), ("HIP_VISIBLE_DEVICES", "0")]      # HIP ordinals win: they count within the ROCr list
    e = serve.pin_env({"ROCR_VISIBLE_DEVICES": "4,GPU-abc"}, ("HIP_VISIBLE_DEVICES", "1"))
    assert e["HIP_VISIBLE_DEVICES"] == "1" and e["ROCR_VISIBLE_DEVICES"] == "4,GPU-abc"
    assert serve.pin_env({}, "3")["HIP_VISIBLE_DEVICES"] == "3"


def test_unpickler_allow_list_is_by_name_not_by_module():
    """``builtins`` as a module admits eval / exec / getattr; the allow-list here names the types a request is made of."""
    import builtins

    for fn in (builtins.eval, builtins.exec, builtins.getattr, builtins.__import__, np.load):
        with pytest.raises(pickle.UnpicklingError):
            serve.restricted_loads(pickle.dumps(fn))
    body = {"image": np.zeros((2, 3, 3), np.uint8), "config": {"a": (1, 2.5, None, True), "b": {1, 2}, "c": b"x", "r": range(3), "f": np.float32(1.5)}}
    back = serve.restricted_loads(pickle.dumps(body))
    assert back["config"]["a"] == (1, 2.5, None, True) and back["config"]["f"] == np.float32(1.5) and back["image"].shape == (2, 3, 3)
    from PIL import Image

    im = serve.restricted_loads(pickle.dumps(Image.fromarray(np.zeros((4, 5, 3), np.uint8))))
    assert im.size == (5, 4)


def test_no_nonce_is_refused_off_loopback(monkeypatch):
    monkeypatch.setenv("MIT_SERVE_ENGINE", "tests._serve_stub:make")
    with pytest.raises(ValueError, match="refused"):
        serve.HipShareWorker({"host": "0.0.0.0", "port": 1, "nonce": "None"})
    assert serve.HipShareWorker({"host": "127.0.0.1", "port": 1, "nonce": "None"}).nonce is None
