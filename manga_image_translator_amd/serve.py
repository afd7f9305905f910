"""Worker-pool serving over the GPUs of a node: one reference ``shared``-mode worker per GPU.

What the reference has (nothing here replaces its server):

* ``manga_translator/mode/share.py:47-174`` ``MangaShare`` — a worker process that owns one ``MangaTranslator`` and serves
  ``POST /simple_execute/{method}`` (pickled attributes in, pickled result out) and ``POST /execute/{method}`` (a stream of frames
  ``status:1 | length:4 big-endian | payload`` — 0 result, 1 progress, 2 error), one request at a time (``429`` while busy), guarded by
  an ``X-Nonce`` header, with a restricted unpickler for the request body;
* ``server/instance.py:10-67`` ``ExecutorInstance`` / ``Executors`` — the front server's table of such workers (ip, port, busy) with
  ``find_executor`` / ``free_executor``; workers are entered through ``POST /register`` (``server/main.py:45-50``);
* ``server/main.py:244-276`` starts exactly ONE worker, on port + 1, and lets the framework pick the GPU.

What this module adds, and nothing more: the worker of that protocol with the HIP plugins inside, pinned to ONE GPU
(``HIP_VISIBLE_DEVICES=<i>`` is set by the pool before the process imports torch), and a pool that starts one per visible GPU and
registers each with a front server.  With ``manga_translator`` importable the worker IS the reference's ``MangaShare`` (``make_worker``: the
plugins are added to its registries first, ``plugins.register()``); without it (this image has no OpenCV, so the orchestrator cannot be imported) the
same endpoints serve ``DenseStages`` — the three dense stages chained the way the orchestrator chains them
(``manga_translator.py:432-622``: detect -> OCR -> text-line merge -> mask refinement -> inpaint) — so the pool, the wire format and the
GPU pinning are testable here.  The product path has no CPU fallback: a worker without a GPU fails at load.
"""
# (no `from __future__ import annotations` here: FastAPI resolves the endpoint parameters' annotations at decoration time)
import asyncio
import io
import os
import pickle
import secrets
import subprocess
import sys
import time
from threading import Lock
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

# ---- wire format (mode/share.py:63-66, server/sent_data_internal.py:44-66) -----------------------------------------------------------
STATUS_RESULT, STATUS_PROGRESS, STATUS_ERROR = 0, 1, 2


def frame(status: int, payload: bytes) -> bytes:
    """One chunk of the ``/execute`` stream: status (1 byte), payload length (4 bytes, big endian), payload."""
    if not 0 <= status <= 255:
        raise ValueError("status must fit one byte")
    return bytes([status]) + len(payload).to_bytes(4, "big") + payload


def parse_frames(buffer: bytes) -> Tuple[List[Tuple[int, bytes]], bytes]:
    """All complete frames at the head of ``buffer`` and the unconsumed rest (``handle_buffer``, sent_data_internal.py:44-58)."""
    out = []
    while len(buffer) >= 5:
        n = int.from_bytes(buffer[1:5], "big")
        if len(buffer) < 5 + n:
            break
        out.append((buffer[0], buffer[5:5 + n]))
        buffer = buffer[5 + n:]
    return out, buffer


# The reference's allow-list is by MODULE (mode/share.py:14-24) and admits all of ``builtins`` — eval, exec, getattr, __import__ … —
# so its "restricted" unpickler still executes whatever a request asks for.  Same modules here, but by (module, name): the container
# and scalar types a request body is made of, numpy's array reconstruction, PIL images, and the reference's config / geometry classes.
SAFE_PICKLE_MODULES = frozenset({   # modules whose classes are admitted wholesale: data-only packages of the reference and PIL (below)
    "manga_translator", "manga_translator.utils", "manga_translator.utils.generic", "manga_translator.config",
})
SAFE_PICKLE_NAMES = frozenset({
    ("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bytes", "bytearray", "str", "bool", "slice",
                              "range", "object")
} | {("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "deque")} | {
    (m, n) for m in ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric")
    for n in ("ndarray", "dtype", "_reconstruct", "scalar", "_frombuffer")
} | {("numpy.dtypes", n) for n in ("UInt8DType", "Int8DType", "Int16DType", "UInt16DType", "Int32DType", "UInt32DType", "Int64DType", "UInt64DType",
                                    "Float16DType", "Float32DType", "Float64DType", "BoolDType")})


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if (module, name) in SAFE_PICKLE_NAMES or module in SAFE_PICKLE_MODULES or module.startswith("PIL."):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"Deserialization of {module}.{name} is not allowed")


def restricted_loads(data: bytes):
    return RestrictedUnpickler(io.BytesIO(data)).load()


# ---- what a worker hosts when the orchestrator itself cannot be imported ---------------------------------------------------------------
class DenseStages:
    """The three HIP plugins of one GPU, chained as ``MangaTranslator._translate`` chains the stages it owns
    (manga_translator.py:432-622): detection -> OCR of the detected lines -> text-line merge -> mask refinement -> inpainting.
    ``translate(image, config)`` has the worker method's name and argument names (``server/instance.py:19-20`` sends
    ``{"image": …, "config": …}`` to ``…/translate``); it returns the dense stages' results as plain numpy / builtins."""

    def __init__(self, params: Optional[dict] = None):
        self.params = dict(params or {})
        self._loaded = False

    async def _load(self):
        if self._loaded:
            return
        import torch

        from . import pipeline, plugins as P

        if not torch.cuda.is_available():
            raise RuntimeError("a serving worker needs its GPU (HIP_VISIBLE_DEVICES selects it); there is no CPU path")
        ckpt = self.params.get("model_dir")
        if ckpt:      # real checkpoints in the reference's layouts (plugins._load_*_checkpoint), under <model_dir>/<detection|ocr|inpainting>/
            if not os.path.isdir(ckpt):
                raise FileNotFoundError(f"--model-dir {ckpt!r} is not a directory")
            P.set_model_dir(os.path.abspath(ckpt))
            self.det, self.ocr, self.inp = P.HipComicTextDetector(), P.HipModel48pxOCR(), P.HipLamaMPEInpainter()
        else:         # no checkpoint offline: seeded synthetic weights of the reference architectures (synth.py), identical in every worker
            d = int(self.params.get("dict_size", 512))
            w = pipeline.synthetic_weights(dict_size=d)
            dictionary = ["<PAD>", "<S>", "</S>", "<SP>"] + [chr(0x4E00 + i) for i in range(d - 4)]
            self.det = P.HipComicTextDetector(weights=w)
            self.ocr = P.HipModel48pxOCR(weights=w["ocr48"], dictionary=dictionary)
            self.inp = P.HipLamaMPEInpainter(weights=w)
        for p in (self.det, self.ocr, self.inp):
            await p.load("cuda")
        self.device_name = torch.cuda.get_device_name(0)
        self._loaded = True

    async def translate(self, image, config=None):
        from . import mask_refinement as MR, textline_merge as TM

        await self._load()
        cfg = _as_dict(config)
        page = np.ascontiguousarray(np.asarray(image.convert("RGB") if hasattr(image, "convert") else image, dtype=np.uint8))
        if page.ndim != 3 or page.shape[2] != 3:
            raise ValueError(f"image must be HxWx3 (got {page.shape})")
        H, W = page.shape[:2]
        det = cfg.get("detector", {})
        tls, mask_raw, _ = await self.det.infer(page, int(det.get("detection_size", 1024)), float(det.get("text_threshold", 0.5)),
                                                float(det.get("box_threshold", 0.7)), float(det.get("unclip_ratio", 2.3)))
        lines_in = cfg.get("textlines")     # a request may bring its own text lines (quads [n,4,2]) instead of the detector's
        if lines_in is not None:
            from .textline import Quadrilateral

            tls = [Quadrilateral(np.asarray(p, dtype=np.int64), "", 1.0) for p in lines_in]
        ocr = cfg.get("ocr", {})

        class _Cfg:   # OcrConfig.prob (config.py): None unless the request sets it — the plugin then keeps the reference's 0.2 (model_48px.py:104)
            prob = None if ocr.get("prob") is None else float(ocr["prob"])

        steps = int(ocr.get("max_seq_length", 255))
        lines = await self.ocr.infer(page, tls, _Cfg(), False, int(ocr.get("ignore_bubble", 0)), steps, bool(ocr.get("suppress_eos", False))) if tls else []
        lines = [l for l in lines if l.text.strip()]
        inp = cfg.get("inpainter", {})
        mask_in = cfg.get("mask")          # a request may bring the mask to inpaint ([H, W] uint8) instead of the refined detector mask
        if mask_in is not None:
            mask = np.ascontiguousarray(np.asarray(mask_in, dtype=np.uint8))
            if mask.shape != (H, W):
                raise ValueError(f"mask must be {H}x{W} (got {mask.shape})")
            out = await self.inp.infer(page, mask, None, int(inp.get("inpainting_size", 2048)))
        elif lines:
            regions = TM.dispatch_sync(lines, W, H)
            mask = MR.dispatch_sync(regions, page, mask_raw, "fit_text", int(cfg.get("mask_dilation_offset", 20)), 0, False, int(cfg.get("kernel_size", 3)))
            out = await self.inp.infer(page, mask, None, int(inp.get("inpainting_size", 2048)))
        else:   # no text: the orchestrator returns the page as it is (manga_translator.py:500-504)
            mask, out = np.zeros((H, W), np.uint8), page
        return {"textlines": [{"pts": np.asarray(l.pts).tolist(), "text": l.text, "prob": float(l.prob),
                               "fg": [int(l.fg_r), int(l.fg_g), int(l.fg_b)], "bg": [int(l.bg_r), int(l.bg_g), int(l.bg_b)]} for l in lines],
                "mask_raw": np.asarray(mask_raw), "mask": np.asarray(mask), "inpainted": np.asarray(out), "device": self.device_name,
                "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}

    async def translate_batch(self, images, config=None, batch_size: int = 1):
        """``/simple_execute/translate_batch`` (server/instance.py:22-26 sends ``{"images", "config", "batch_size"}``): the pages of a
        request one after the other through ``translate`` — the plugins are the reference's page-at-a-time interface; the batched engines
        (pipeline.PageEngine, coupled.CoupledPageEngine) are what a batch job calls directly."""
        return [await self.translate(im, config) for im in images]

    async def device_info(self, image=None, config=None):   # (the executor's send calls always carry both attributes)
        await self._load()
        import torch

        return {"device": self.device_name, "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"), "n_visible": torch.cuda.device_count(),
                "pid": os.getpid()}


def _as_dict(config) -> dict:
    if config is None:
        return {}
    if isinstance(config, dict):
        return config
    for m in ("model_dump", "dict"):
        if hasattr(config, m):
            return getattr(config, m)()
    return dict(vars(config))


def _make_engine(params: dict):
    """The object whose methods the worker exposes: the reference's MangaTranslator (HIP plugins registered) when it imports,
    else DenseStages; MIT_SERVE_ENGINE=module:callable substitutes another factory (protocol tests without a GPU)."""
    hook = os.environ.get("MIT_SERVE_ENGINE")
    if hook:
        import importlib

        mod, _, fn = hook.partition(":")
        return getattr(importlib.import_module(mod), fn)(params)
    try:
        from manga_translator import MangaTranslator  # type: ignore  # noqa: F401
    except Exception:
        return DenseStages(params)
    from . import plugins as P

    P.register()    # ctd_hip / 48px_hip / lama_mpe_hip … become detector / ocr / inpainter choices of Config
    return MangaTranslator(params)


def make_worker(params: dict):
    """The worker process's server object.  With the reference importable it IS the reference's ``MangaShare`` (mode/share.py:47-174) —
    its endpoints, its unpickler, its ``MangaTranslator`` — and this package only adds its plugins to the registries first
    (``plugins.register()``: ``ctd_hip`` / ``48px_hip`` / ``lama_mpe_hip`` … become choices of ``Config``).  Only where the reference cannot
    be imported (this image: no OpenCV) the stand-alone mirror below serves the same protocol around ``DenseStages``."""
    if not os.environ.get("MIT_SERVE_ENGINE"):
        try:
            from manga_translator.mode.share import MangaShare  # type: ignore
        except Exception:
            MangaShare = None
        if MangaShare is not None:
            from . import plugins as P

            P.register()
            if params.get("model_dir"):
                P.set_model_dir(os.path.abspath(params["model_dir"]))
            return MangaShare(params)
    return HipShareWorker(params)


class HipShareWorker:
    """Stand-alone mirror of ``MangaShare`` (mode/share.py:47-174) around ``_make_engine``: same endpoints, nonce rule, one-request lock
    and stream framing — pinned to the reference's own CLIENT (server/sent_data_internal.py) by tests/test_serve.py."""

    def __init__(self, params: Optional[dict] = None):
        params = dict(params or {})
        self.manga = _make_engine(params)
        self.host = params.get("host", "127.0.0.1")
        self.port = int(params.get("port", 5003))
        nonce = params.get("nonce", None)
        if not nonce:
            nonce = secrets.token_hex(16)
        if nonce == "None":   # the reference's way of switching the check off: only for a loopback listener (a request body is a pickle)
            if self.host not in ("127.0.0.1", "localhost", "::1"):
                raise ValueError(f"nonce 'None' (no authentication) is refused on host {self.host!r}: bind to 127.0.0.1 or set a nonce")
            nonce = None
        self.nonce = nonce
        self.lock = Lock()
        self.progress_queue: Optional[asyncio.Queue] = None
        if hasattr(self.manga, "add_progress_hook"):
            async def hook(state: str, finished: bool):
                await self.progress_queue.put(frame(STATUS_PROGRESS, state.encode("utf-8")))
                await asyncio.sleep(0)

            self.manga.add_progress_hook(hook)

    def app(self):
        from fastapi import FastAPI, HTTPException, Path, Request, Response
        from starlette.responses import StreamingResponse

        app = FastAPI()
        self.progress_queue = asyncio.Queue()

        def check_nonce(request: Request):
            if self.nonce and request.headers.get("X-Nonce") != self.nonce:
                raise HTTPException(401, detail="Nonce does not match")

        def check_lock():
            if not self.lock.acquire(blocking=False):
                raise HTTPException(status_code=429, detail="some Method is already being executed.")

        def get_fn(method_name: str):
            if method_name.startswith("_"):
                raise HTTPException(status_code=403, detail="These functions are not allowed to be executed remotely")
            method = getattr(self.manga, method_name, None)
            if not callable(method):
                raise HTTPException(status_code=404, detail="Method not found")
            return method

        async def call(method, attr):
            return await method(**attr) if asyncio.iscoroutinefunction(method) else method(**attr)

        @app.get("/is_locked")
        async def is_locked():
            return {"locked": self.lock.locked()}

        @app.post("/simple_execute/{method_name}")
        async def simple_execute(request: Request, method_name: str = Path(...)):
            check_nonce(request)
            method = get_fn(method_name)
            try:
                attr = restricted_loads(await request.body())
            except Exception as e:
                raise HTTPException(status_code=400, detail=f"bad request body: {e}")
            check_lock()
            try:
                return Response(content=pickle.dumps(await call(method, attr)), media_type="application/octet-stream")
            except Exception as e:  # noqa: BLE001 - reported to the caller as the reference does
                raise HTTPException(status_code=500, detail=str(e))
            finally:
                self.lock.release()

        @app.post("/execute/{method_name}")
        async def execute(request: Request, method_name: str = Path(...)):
            check_nonce(request)
            method = get_fn(method_name)
            try:
                attr = restricted_loads(await request.body())
            except Exception as e:
                raise HTTPException(status_code=400, detail=f"bad request body: {e}")
            check_lock()
            while not self.progress_queue.empty():   # progress frames of earlier /simple_execute calls (nobody read them) do not belong
                self.progress_queue.get_nowait()     # to this stream

            async def run():
                try:
                    await self.progress_queue.put(frame(STATUS_RESULT, pickle.dumps(await call(method, attr))))
                except Exception as e:  # noqa: BLE001
                    await self.progress_queue.put(frame(STATUS_ERROR, str(e).encode("utf-8")))
                finally:
                    self.lock.release()

            async def stream():
                while True:
                    chunk = await self.progress_queue.get()
                    yield chunk
                    if chunk[0] != STATUS_PROGRESS:
                        break

            asyncio.create_task(run())
            return StreamingResponse(stream(), media_type="application/octet-stream")

        return app

    async def listen(self):
        import uvicorn

        server = uvicorn.Server(uvicorn.Config(self.app(), host=self.host, port=self.port, log_level="warning"))
        await server.serve()


# ---- the pool: one worker per GPU, and the front server's view of it -----------------------------------------------------------------
class ExecutorInstance:
    """``server/instance.py:10-33``: a worker's address and its busy flag, with the two send calls."""

    def __init__(self, ip: str, port: int, nonce: Optional[str] = None, gpu: Optional[int] = None):
        self.ip, self.port, self.busy, self.nonce, self.gpu = ip, int(port), False, nonce, gpu

    @property
    def url(self) -> str:
        return f"http://{self.ip}:{self.port}"

    def _headers(self) -> Dict[str, str]:
        return {"X-Nonce": self.nonce} if self.nonce else {}

    async def sent(self, image, config, method: str = "translate"):
        import aiohttp

        data = pickle.dumps({"image": image, "config": config})
        async with aiohttp.ClientSession() as s:
            async with s.post(f"{self.url}/simple_execute/{method}", data=data, headers=self._headers()) as r:
                body = await r.read()
                if r.status != 200:
                    raise RuntimeError(f"worker {self.url}: HTTP {r.status}: {body[:300]!r}")
                return pickle.loads(body)   # our own worker's reply

    async def sent_batch(self, images, config, batch_size: int = 1):
        """``ExecutorInstance.sent_batch`` (server/instance.py:22-26)."""
        import aiohttp

        data = pickle.dumps({"images": list(images), "config": config, "batch_size": int(batch_size)})
        async with aiohttp.ClientSession() as s:
            async with s.post(f"{self.url}/simple_execute/translate_batch", data=data, headers=self._headers()) as r:
                body = await r.read()
                if r.status != 200:
                    raise RuntimeError(f"worker {self.url}: HTTP {r.status}: {body[:300]!r}")
                return pickle.loads(body)

    async def sent_stream(self, image, config, sender: Callable[[int, bytes], None], method: str = "translate"):
        import aiohttp

        data = pickle.dumps({"image": image, "config": config})
        async with aiohttp.ClientSession() as s:
            async with s.post(f"{self.url}/execute/{method}", data=data, headers=self._headers()) as r:
                if r.status != 200:
                    raise RuntimeError(f"worker {self.url}: HTTP {r.status}: {(await r.read())[:300]!r}")
                buf = b""
                async for chunk in r.content.iter_any():
                    frames, buf = parse_frames(buf + chunk)
                    for st, payload in frames:
                        sender(st, payload)


class Executors:
    """``server/instance.py:35-65``: the table of workers; ``find_executor`` hands out a free one (waiting for a release when all are
    busy), ``free_executor`` returns it."""

    def __init__(self):
        self.list: List[ExecutorInstance] = []
        self.lock = asyncio.Lock()
        self.event = asyncio.Event()

    def register(self, instance: ExecutorInstance):
        self.list.append(instance)

    def free_executors(self) -> int:
        return len([x for x in self.list if not x.busy])

    async def find_executor(self) -> ExecutorInstance:
        async with self.lock:
            while True:
                inst = next((x for x in self.list if not x.busy), None)
                if inst is not None:
                    inst.busy = True
                    return inst
                await self.event.wait()

    async def free_executor(self, instance: ExecutorInstance):
        instance.busy = False
        self.event.set()
        self.event.clear()


def visible_gpus() -> List[Tuple[str, str]]:
    """The devices a pool starts workers on, as (environment variable, value) pairs that pin ONE process to ONE of them.
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES entries are HIP ordinals (within whatever ROCR_VISIBLE_DEVICES leaves); an entry of
    ROCR_VISIBLE_DEVICES (index or GPU-UUID) is a ROCr-level name and can only be narrowed at that level — ``HIP_VISIBLE_DEVICES=<entry>``
    beside an inherited ROCR list would index INTO the filtered list and find nothing.  With none set: 0 … n-1 as rocm reports them
    (counted in a child process so that THIS process never initialises a GPU it would then keep memory on)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            return [("HIP_VISIBLE_DEVICES", x) for x in v.split(",") if x != ""]
    v = os.environ.get("ROCR_VISIBLE_DEVICES")
    if v:
        return [("ROCR_VISIBLE_DEVICES", x) for x in v.split(",") if x != ""]
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    n = int(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else 0
    return [("HIP_VISIBLE_DEVICES", str(i)) for i in range(n)]


def pin_env(env: Dict[str, str], gpu) -> Dict[str, str]:
    """``env`` narrowed to one device: ``gpu`` is a (variable, value) pair of ``visible_gpus()`` or a bare HIP ordinal."""
    var, val = gpu if isinstance(gpu, tuple) else ("HIP_VISIBLE_DEVICES", str(gpu))
    env = dict(env)
    env.pop("CUDA_VISIBLE_DEVICES", None)
    if var == "ROCR_VISIBLE_DEVICES":
        env.pop("HIP_VISIBLE_DEVICES", None)     # the one device ROCr leaves is HIP ordinal 0
        env["ROCR_VISIBLE_DEVICES"] = str(val)
    else:
        env["HIP_VISIBLE_DEVICES"] = str(val)    # an inherited ROCR_VISIBLE_DEVICES stays: the ordinal counts within it
    return env


class WorkerPool:
    """Starts one worker process per GPU — ``python -m manga_image_translator_amd.serve worker --port P`` with ``HIP_VISIBLE_DEVICES=<gpu>`` —
    on consecutive ports from ``base_port`` (the reference's single worker sits on port + 1, server/main.py:288), waits until each answers
    ``/is_locked``, and can enter them into a front server's table (``POST /register``, server/main.py:45-50)."""

    def __init__(self, gpus: Optional[Sequence[str]] = None, host: str = "127.0.0.1", base_port: int = 5003, nonce: Optional[str] = None,
                 worker_args: Sequence[str] = (), env: Optional[Dict[str, str]] = None):
        self.gpus = list(gpus) if gpus is not None else visible_gpus()
        if not self.gpus:
            raise RuntimeError("no GPU visible: a worker pool needs at least one (HIP_VISIBLE_DEVICES)")
        self.host, self.base_port = host, int(base_port)
        self.nonce = nonce or secrets.token_hex(16)
        self.worker_args, self.env = list(worker_args), dict(env or {})
        self.procs: List[subprocess.Popen] = []
        self.executors = Executors()

    def start(self, timeout: float = 900.0) -> "WorkerPool":
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for k, gpu in enumerate(self.gpus):
            env = dict(os.environ)
            env.update(self.env)
            env = pin_env(env, gpu)                          # the worker sees exactly one device, as cuda:0
            env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
            port = self.base_port + k
            cmd = [sys.executable, "-m", "manga_image_translator_amd.serve", "worker", "--host", self.host, "--port", str(port),
                   "--nonce", self.nonce, *self.worker_args]
            self.procs.append(subprocess.Popen(cmd, env=env, cwd=root))
            self.executors.register(ExecutorInstance(self.host, port, self.nonce, gpu=k))
        self._wait_ready(timeout)
        return self

    def _wait_ready(self, timeout: float):
        import urllib.request

        t_end = time.time() + timeout
        for inst, proc in zip(self.executors.list, self.procs):
            while True:
                if proc.poll() is not None:
                    self.stop()
                    raise RuntimeError(f"worker on port {inst.port} exited with code {proc.returncode} before it was ready")
                try:
                    with urllib.request.urlopen(f"{inst.url}/is_locked", timeout=2) as r:
                        if r.status == 200:
                            break
                except OSError:
                    pass
                if time.time() > t_end:
                    self.stop()
                    raise TimeoutError(f"worker on port {inst.port} did not come up within {timeout:.0f} s")
                time.sleep(0.25)

    def register_with(self, server_url: str, server_nonce: str):
        """Enter every worker into a running front server (``python server/main.py``): what ``start_translator_client_proc`` does for
        its one local worker, done for one per GPU.  The server takes the ip from the connection and the port from the body."""
        import json
        import urllib.request

        for inst in self.executors.list:
            req = urllib.request.Request(server_url.rstrip("/") + "/register", data=json.dumps({"ip": inst.ip, "port": inst.port}).encode(),
                                         headers={"Content-Type": "application/json", "X-Nonce": server_nonce}, method="POST")
            with urllib.request.urlopen(req, timeout=30) as r:
                if r.status != 200:
                    raise RuntimeError(f"register of {inst.url} failed: HTTP {r.status}")

    async def run(self, image, config=None, method: str = "translate"):
        """One request on whichever worker is free (the front server's ``find_executor`` -> ``sent`` -> ``free_executor`` cycle)."""
        inst = await self.executors.find_executor()
        try:
            return await inst.sent(image, config, method)
        finally:
            await self.executors.free_executor(inst)

    async def map(self, images: Sequence, config=None) -> List:
        """All images over the pool, as many in flight as there are workers; results in the order of ``images``."""
        return list(await asyncio.gather(*[self.run(im, config) for im in images]))

    def stop(self):
        for p in self.procs:
            if p.poll() is None:
                p.terminate()
        for p in self.procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        self.procs = []

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()


def _main(argv=None) -> int:
    import argparse

    ap = argparse.ArgumentParser(prog="python -m manga_image_translator_amd.serve")
    sub = ap.add_subparsers(dest="cmd", required=True)
    w = sub.add_parser("worker", help="one shared-mode worker on the GPU HIP_VISIBLE_DEVICES names (mode/share.py)")
    w.add_argument("--host", default="127.0.0.1")
    w.add_argument("--port", type=int, default=5003)
    w.add_argument("--nonce", default=os.getenv("MT_WEB_NONCE") or None)
    w.add_argument("--model-dir", default=None, help="directory with the reference's checkpoints; synthetic weights without it")
    w.add_argument("--dict-size", type=int, default=512)
    w.add_argument("--lazy", action="store_true", help="load the plugins at the first request instead of before listening (default: before, so "
                                                      "that a worker without its GPU or its checkpoints never reports ready)")
    w.add_argument("--preload", action="store_true", help=argparse.SUPPRESS)   # the default since round 6
    a = ap.parse_args(argv)
    params = {"host": a.host, "port": a.port, "nonce": a.nonce, "model_dir": a.model_dir, "dict_size": a.dict_size, "use_gpu": True}
    worker = make_worker(params)
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    if not a.lazy and hasattr(worker.manga, "_load"):    # DenseStages: GPU visible? checkpoints found? — fail before /is_locked answers
        loop.run_until_complete(worker.manga._load())
    loop.run_until_complete(worker.listen())
    return 0


if __name__ == "__main__":
    sys.exit(_main())
