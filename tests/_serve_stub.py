"""Stand-in engine for the serving-protocol tests (MIT_SERVE_ENGINE=tests._serve_stub:make): no GPU, no plugins — the worker protocol,
the pool and the GPU pinning are what is under test."""
import asyncio
import os

import numpy as np


class Stub:
    def __init__(self, params):
        self.params = params
        self.hooks = []

    def add_progress_hook(self, fn):
        self.hooks.append(fn)

    async def translate(self, image, config=None):
        a = np.asarray(image)
        for h in self.hooks:
            await h("detection", False)
        cfg = config if isinstance(config, dict) else dict(vars(config)) if config is not None else {}   # the reference's client sends a Config object
        await asyncio.sleep(float(cfg.get("sleep", 0.0)))
        for h in self.hooks:
            await h("finished", True)
        return {"sum": int(a.astype(np.int64).sum()), "shape": list(a.shape), "visible": os.environ.get("HIP_VISIBLE_DEVICES"), "pid": os.getpid()}

    async def translate_batch(self, images, config=None, batch_size=1):
        return [await self.translate(im, config) for im in images]

    async def fail(self, image, config=None):
        raise ValueError("stage exploded")

    def _private(self):
        return "no"


def make(params):
    return Stub(params)
