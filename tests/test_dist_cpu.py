"""The N > 1 path on CPU: two processes over gloo (127.0.0.1) run the same sharding / weight-broadcast / result-gather
code that bench.py runs over RCCL, with small CPU tensors."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from manga_image_translator_amd import dist as D


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 64, 1024, 1001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    try:
        r, w, _ = D.init(backend="gloo")
        assert (r, w) == (rank, world)
        # weights exist on rank 0 only; every rank must end up with identical tensors (incl. int64 / 0-d / empty ones)
        weights = None
        if rank == 0:
            g = torch.Generator().manual_seed(5)
            weights = {"a": {"w": torch.randn(7, 3, generator=g), "nbt": torch.tensor(3, dtype=torch.int64),
                             "e": torch.empty(0)}, "b": {"emb": torch.randn(5, 4, generator=g).double()}}
        got = D.broadcast_weights(weights)
        g = torch.Generator().manual_seed(5)
        assert torch.equal(got["a"]["w"], torch.randn(7, 3, generator=g)) and got["a"]["nbt"].item() == 3
        assert got["a"]["nbt"].dtype == torch.int64 and got["a"]["e"].numel() == 0
        assert torch.equal(got["b"]["emb"], torch.randn(5, 4, generator=g).double())
        # pages: every rank "processes" its own block; rank 0 gathers the packed results
        n_pages = 6
        lo, hi = D.shard_range(n_pages, rank, world)
        packed = torch.arange(lo, hi, dtype=torch.uint8).repeat_interleave(4)  # 4 result bytes per page
        out = D.gather_pages(packed)
        if rank == 0:
            assert out.shape == (world, (hi - lo) * 4)
            assert out.reshape(-1).tolist() == [p for p in range(n_pages) for _ in range(4)]
        else:
            assert out is None
        # the per-step form bench.py uses: submit() returns the previous step's block, wait() drains the last one; both the stream-ordered
        # default and the one-step-slack form, each block verified against the checksum its source rank sent
        for async_op in (False, True):
            pg = D.PageGather(async_op=async_op)
            seen = []
            for stepno in range(3):
                prev = pg.submit(packed + stepno)
                seen.append(prev)
            seen.append(pg.wait())
            assert pg.wait() is None                                    # nothing left in flight
            if rank == 0:
                assert seen[0] is None and pg.last_bytes == world * packed.numel()
                for stepno, blk in enumerate(seen[1:]):
                    assert blk.reshape(-1).tolist() == [p + stepno for p in range(n_pages) for _ in range(4)]
                assert pg.check() == 3 * world                          # every received block was checksummed and matched
            else:
                assert all(x is None for x in seen) and pg.check() == 0
        assert D.max_over_ranks(float(rank + 1)) == float(world)
        D.barrier()
        torch.distributed.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e)))
        raise


def test_two_rank_gloo_broadcast_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(p.exitcode == 0 for p in procs)


def test_single_process_paths():
    w = {"a": {"w": torch.ones(2)}}
    assert D.broadcast_weights(w) is w
    with pytest.raises(ValueError):
        D.broadcast_weights(None)
    t = torch.arange(4)
    assert torch.equal(D.gather_pages(t), t[None])
    pg = D.PageGather()
    assert pg.submit(t) is None and torch.equal(pg.submit(t + 1), t[None]) and torch.equal(pg.wait(), (t + 1)[None]) and pg.wait() is None
    assert D.max_over_ranks(2.5) == 2.5


def test_page_checksum_sees_flipped_bytes_and_misplaced_chunks():
    from manga_image_translator_amd import dist as D

    g = torch.Generator().manual_seed(0)
    t = torch.randint(0, 256, (3, 100_003), dtype=torch.uint8, generator=g)     # not a multiple of 8 bytes, several 32 KB chunks
    ref = D.page_checksum(t)
    assert ref.dtype == torch.int64 and tuple(ref.shape) == (2,) and torch.equal(ref, D.page_checksum(t.clone()))
    for pos in (0, 12345, t.numel() - 1):
        u = t.clone().reshape(-1)
        u[pos] ^= 0x40
        assert not torch.equal(D.page_checksum(u), ref), pos
    u = t.clone().reshape(-1)
    a, b = u[:32768].clone(), u[65536:65536 + 32768].clone()
    u[:32768], u[65536:65536 + 32768] = b, a                                      # two chunks swapped: the plain sum alone would not see it
    cs = D.page_checksum(u)
    assert cs[0] == ref[0] and cs[1] != ref[1]
    assert torch.equal(D.page_checksum(t.view(torch.int8)), ref)                  # any dtype: the bytes are what counts
    # a tampered block is reported by the gather object
    pg = D.PageGather()
    pg._out, pg._sums = t[None].clone(), ref[None].clone()
    pg._out[0, 1, 7] ^= 1
    import pytest
    with pytest.raises(RuntimeError, match="failed their checksum"):
        pg.check()
