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


def page_checksum(t: torch.Tensor) -> torch.Tensor:
    """Two wrapping int64 sums over a uint8 tensor, computed where the tensor lives: the plain sum of its 8-byte words and the sum of
    its 32 KB chunk sums weighted by the chunk's position (so that chunks landing in the wrong place are seen too).  What a rank
    sends beside its result block so that rank 0 can verify what arrived."""
    flat = t.contiguous().reshape(-1)
    if flat.dtype != torch.uint8:
        flat = flat.view(torch.uint8)
    if flat.storage_offset() % 8:   # the word view needs an 8-byte aligned start (the value must not depend on where the bytes sit)
        flat = flat.clone()
    n8 = flat.numel() // 8 * 8
    words = flat[:n8].view(torch.int64)
    CH = 4096
    m = words.numel() // CH * CH
    rows = words[:m].view(-1, CH).sum(dim=1)
    s0 = rows.sum() + words[m:].sum() + flat[n8:].to(torch.int64).sum()
    s1 = (rows * torch.arange(1, rows.numel() + 1, dtype=torch.int64, device=flat.device)).sum()
    return torch.stack([s0, s1])


class PageGather:
    """The per-step result gather to rank ``dst`` (point-to-point over xGMI, no ring), verified end to end.

    ``submit(packed)`` issues the gather of the ranks' result blocks and of their checksums (``page_checksum``, computed on the source
    rank before the send) and returns the PREVIOUS step's gathered tensor; on ``dst`` every received block is checksummed again and
    compared — on the device, accumulated into a counter that ``check()`` reads (one host sync, outside the step loop) and turns into an
    error.  By default the caller's stream is ordered behind the gather before ``submit`` returns: no RCCL kernel runs beside the next
    step's compute kernels (two queues sharing CUs is the configuration DESIGN §7 restricts).  ``async_op=True`` (or MIT_GATHER_ASYNC=1)
    gives the gather one step of slack instead — rank 0 receives ``world - 1`` blocks (0.8 GB each at 64 pages) while its next batch is
    computing; ``wait()`` drains it (call it before the end of a timed region).  At most one gather is in flight, so at most two result
    blocks per rank are alive.  Same semantics on gloo.

    Layout of the result on ``dst``: ``[world, *shape]`` as a VIEW into a slab whose rows are padded to 8 bytes (every rank's block starts
    word-aligned for the checksum), so when a block's byte size is not a multiple of 8 the result is not contiguous across dim 0: index
    it per rank (``out[r]`` is dense) or call ``.contiguous()`` before ``.view(-1)`` / handing ``data_ptr()`` to native code."""

    def __init__(self, dst: int = 0, async_op: Optional[bool] = None, verify: bool = True):
        self.dst = dst
        self.async_op = (os.environ.get("MIT_GATHER_ASYNC", "0") not in ("", "0")) if async_op is None else bool(async_op)
        self.verify = verify
        self._work: List = []
        self._out: Optional[torch.Tensor] = None
        self._sums: Optional[torch.Tensor] = None
        self._keep = None
        self._bad: Optional[torch.Tensor] = None   # blocks whose checksum did not match, counted where the blocks live
        self.verified_blocks = 0
        self.last_bytes = 0
        self.submits = 0
        self._host_wait_s = 0.0
        self._events: List = []      # (start, stop) device events bracketing a gather on the caller's stream (nccl)
        self._event_ms = 0.0         # ... and the total of the pairs already folded (_fold_events)
        self._t_issue: Optional[float] = None
        self._on_device = False

    @property
    def wait_ms(self) -> float:
        """Time this rank spent inside its gathers so far (issue -> received and checksummed): device events on the caller's stream when
        the tensors live on the GPU (the stream is what waits for RCCL), the host clock otherwise.  Reading it synchronises the events."""
        self._fold_events(keep=0)
        return self._host_wait_s * 1e3 + self._event_ms

    def _fold_events(self, keep: int) -> None:
        """Finished (start, stop) pairs beyond the newest ``keep`` go into the running total: a long job holds a bounded number of events.
        A pair whose stop is not recorded yet (an async gather in flight) stays."""
        done = [p for p in self._events if p[1] is not None]
        for pair in done[:max(0, len(done) - keep)]:
            pair[1].synchronize()
            self._event_ms += pair[0].elapsed_time(pair[1])
            self._events.remove(pair)

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
        self.submits += 1
        self._on_device = bool(packed.is_cuda)
        if not self.async_op:   # synchronous form: the clock runs from the issue (async: from the wait, in _finish — the interval
            self._start_clock()  # issue -> _finish would span the next step's compute)
        cs = page_checksum(packed) if self.verify else None
        if rank == self.dst:
            n = packed.numel()
            pitch = (n * packed.element_size() + 7) // 8 * 8 // packed.element_size() if (8 % packed.element_size()) == 0 else n
            slab = torch.empty(world, pitch, dtype=packed.dtype, device=packed.device)   # every rank's block starts 8-byte aligned
            out = slab[:, :n].unflatten(1, tuple(packed.shape)) if packed.dim() != 1 else slab[:, :n]
            self._work = [dist.gather(packed, [slab[r, :n].view(packed.shape) for r in range(world)], dst=self.dst, async_op=True)]
            if self.verify:
                self._sums = torch.empty(world, 2, dtype=torch.int64, device=packed.device)
                self._work.append(dist.gather(cs, list(self._sums.unbind(0)), dst=self.dst, async_op=True))
            self._out = out
            self.last_bytes = out.numel() * out.element_size()
        else:
            self._work = [dist.gather(packed, None, dst=self.dst, async_op=True)]
            if self.verify:
                self._work.append(dist.gather(cs, None, dst=self.dst, async_op=True))
            self._out = None
        self._keep = (packed, cs)  # alive until the gather has been waited for
        if not self.async_op:
            self._finish()
        return prev

    def _start_clock(self) -> None:
        if self._on_device:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._events.append([e0, None])
            self._fold_events(keep=8)
        else:
            self._t_issue = _now()

    def _finish(self) -> None:
        if self._work and self.async_op:
            self._start_clock()
        for w in self._work:
            w.wait()  # nccl: the current stream waits for the collective (no host block); gloo: blocks
        self._work = []
        if self._out is not None and self._sums is not None:
            bad = torch.zeros((), dtype=torch.int64, device=self._out.device)
            for r in range(self._out.shape[0]):
                bad = bad + (page_checksum(self._out[r]) != self._sums[r]).any().to(torch.int64)
            self._bad = bad if self._bad is None else self._bad + bad
            self.verified_blocks += int(self._out.shape[0])
            self._sums = None
        if self._events and self._events[-1][1] is None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._events[-1][1] = e1
        if self._t_issue is not None:
            self._host_wait_s += _now() - self._t_issue
            self._t_issue = None

    def wait(self) -> Optional[torch.Tensor]:
        """Result of the gather in flight ([world, *shape] on ``dst``, None elsewhere or when nothing was submitted); a failure raises."""
        self._finish()
        out, self._out, self._keep = self._out, None, None
        return out

    def check(self) -> int:
        """Blocks verified so far; raises if any block's checksum on ``dst`` differed from the one its source rank computed."""
        self._finish()
        if self._bad is not None and int(self._bad.item()) != 0:
            raise RuntimeError(f"PageGather: {int(self._bad.item())} gathered result block(s) failed their checksum")
        return self.verified_blocks


def _now() -> float:
    import time

    return time.perf_counter()


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
