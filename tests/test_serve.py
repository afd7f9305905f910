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
