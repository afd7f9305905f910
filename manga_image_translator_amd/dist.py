"""Multi-GPU sharding of a page job: one process per GPU, pages split into contiguous blocks.

Pages are independent units (no cross-page state in detect / OCR / inpaint — the reference only carries page context
into its translators, /root/reference/manga_translator/manga_translator.py:417-428), so the data path has no
collective.  RCCL (``torch.distributed`` backend "nccl" on ROCm) is used for exactly two things:

* ``broadcast_weights`` — rank 0 loads / synthesises the state_dicts once; every other rank receives them as one
  flat byte arena (a single large broadcast instead of ~1500 small ones: xGMI links are point-to-point, so one big
  transfer per peer is the cheap shape);
* ``gather_pages`` — per-page results (uint8 maps, inpainted pages, token ids) are gathered to rank 0.

Both work unchanged on the "gloo" backend with CPU tensors (tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

StateDicts = Dict[str, Dict[str, torch.Tensor]]


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched by it."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group when WORLD_SIZE > 1 (env:// rendezvous). Returns (rank, world, local_rank)."""
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:  # MIT_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a 1-GPU box)
            backend = os.environ.get("MIT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of ``n_items`` for ``rank``; the first ``n_items % world`` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def _comm_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_weights(weights: Optional[StateDicts], src: int = 0) -> StateDicts:
    """Every rank returns the state_dicts held by ``src`` (other ranks may pass None).

    Metadata (names, shapes, dtypes, offsets) travels as a pickled object, the payload as one uint8 arena."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if weights is None:
            raise ValueError("single-process run needs the weights")
        return weights
    rank = dist.get_rank()
    meta: List = [None]
    arena = None
    if rank == src:
        if weights is None:
            raise ValueError("source rank must hold the weights")
        entries, chunks, off = [], [], 0
        for group in sorted(weights):
            for name in sorted(weights[group]):
                t = weights[group][name].detach().cpu().contiguous()
                nbytes = t.numel() * t.element_size()
                entries.append((group, name, tuple(t.shape), str(t.dtype).replace("torch.", ""), off, nbytes))
                chunks.append(t.reshape(-1).view(torch.uint8) if t.numel() else torch.empty(0, dtype=torch.uint8))
                off += (nbytes + 15) // 16 * 16
        arena = torch.zeros(max(off, 16), dtype=torch.uint8)
        for (_, _, _, _, o, nb), c in zip(entries, chunks):
            arena[o:o + nb] = c
        meta = [(entries, arena.numel())]
    dist.broadcast_object_list(meta, src=src)
    entries, total = meta[0]
    dev = _comm_device()
    buf = arena.to(dev) if rank == src else torch.empty(total, dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    if rank == src:
        return weights
    host = buf.cpu()
    out: StateDicts = {}
    for group, name, shape, dtype, o, nb in entries:
        dt = getattr(torch, dtype)
        t = host[o:o + nb].view(dt).reshape(shape) if nb else torch.empty(shape, dtype=dt)
        out.setdefault(group, {})[name] = t
    return out


def gather_pages(packed: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather equal-sized per-rank result tensors to ``dst``: returns [world, *packed.shape] there, None elsewhere.

    Point-to-point into the root (``dist.gather``): each peer uses its own xGMI link to rank 0, no ring."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed.unsqueeze(0)
    world, rank = dist.get_world_size(), dist.get_rank()
    packed = packed.contiguous()
    if dist.get_backend() != "nccl" and packed.is_cuda:  # gloo rehearsal: stage through host memory
        packed = packed.cpu()
    if rank == dst:
        out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
        dist.gather(packed, list(out.unbind(0)), dst=dst)
        return out
    dist.gather(packed, None, dst=dst)
    return None


class PageGather:
    """The per-step result gather with one step of slack: ``submit(packed)`` issues the gather asynchronously (RCCL runs it on its own
    stream, ordered after the work that produced ``packed``) and returns the PREVIOUS step's gathered tensor; the caller's stream never
    waits for peer traffic before launching the next step's kernels — rank 0 receives ``world - 1`` result blocks (0.8 GB each at 64
    pages) over its xGMI links while its next batch is already computing.  ``wait()`` drains the last one (call it before the end of a
    timed region).  At most one gather is in flight, so at most two result blocks per rank are alive.  Same semantics on gloo."""

    def __init__(self, dst: int = 0):
        self.dst = dst
        self._work = None
        self._out: Optional[torch.Tensor] = None
        self._keep = None
        self.last_bytes = 0

    def submit(self, packed: torch.Tensor) -> Optional[torch.Tensor]:
        prev = self.wait()
        if not dist.is_initialized() or dist.get_world_size() == 1:
            self._out = packed.unsqueeze(0)
            self.last_bytes = self._out.numel() * self._out.element_size()
            return prev
        world, rank = dist.get_world_size(), dist.get_rank()
        packed = packed.contiguous()
        if dist.get_backend() != "nccl" and packed.is_cuda:  # gloo rehearsal: stage through host memory
            packed = packed.cpu()
        if rank == self.dst:
            out = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
            self._work = dist.gather(packed, list(out.unbind(0)), dst=self.dst, async_op=True)
            self._out = out
            self.last_bytes = out.numel() * out.element_size()
        else:
            self._work = dist.gather(packed, None, dst=self.dst, async_op=True)
            self._out = None
        self._keep = packed  # alive until the gather has been waited for
        return prev

    def wait(self) -> Optional[torch.Tensor]:
        """Result of the gather in flight ([world, *shape] on ``dst``, None elsewhere or when nothing was submitted); a failure raises."""
        if self._work is not None:
            self._work.wait()  # nccl: the current stream waits for the collective (no host block); gloo: blocks
            self._work = None
        out, self._out, self._keep = self._out, None, None
        return out


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
