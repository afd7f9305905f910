"""Split-bf16 contraction (csrc/conv_gemm_split.h conv_gemm_split_kernel, include/mit_hip.h mit_gemm_split_pack) — what can be pinned without a GPU:

* the arithmetic claim: an fp32 number is exactly hi + mid + lo of three round-to-nearest-even bf16 numbers, every plane product is
  exact in fp32, and dropping the three smallest of the nine plane pairs ("p6") costs less than fp32 rounding itself;
* the kernel's LDS staging: an index-for-index emulation of how a K-tile is written (A: split float4 chunks as 8-byte halves of
  16-byte cells, rows XOR-swizzled per k slab; W: whole cells) and how the MFMA operand fragments are read back, for every shipped
  tile shape — each lane must receive (its row | column, its k group) with A and W agreeing on k, the 8-byte writes must be
  bank-conflict free and every 16-lane read group must cover one aligned 256-byte run;
* the host plumbing that attaches split planes to a descriptor (identity-checked registry).
The kernel itself is compared with the fp32 tiles in tests/test_gemm_split_gpu.py."""
import gc
import os

import numpy as np
import pytest
import torch

from manga_image_translator_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(np.float32)


def planes(v):
    h = bf16_rne(v)
    r = v - h
    m = bf16_rne(r)
    return [h, m, bf16_rne(r - m)]


PAIRS = [(2, 2), (1, 2), (2, 1), (0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]   # kSplitPA / kSplitPB, smallest first


def test_three_bf16_planes_are_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1 << 18) * np.exp(rng.uniform(-30, 30, 1 << 18))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 3.0e38, 1.1754944e-38 * 4096, 1 + 2 ** -23, 255.99998]
    h, m, l = planes(x)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
    assert np.array_equal(((h + m) + l), x)                               # also when summed in fp32, largest first
    for p in (h, m, l):                                                      # each plane has at most 8 significant bits
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)
    a, b = planes(x[:4096]), planes(x[4096:8192])
    for p, q in PAIRS:                                                       # 8 x 8 bits: every plane product is exact in fp32
        prod64 = a[p].astype(np.float64) * b[q].astype(np.float64)
        ok = np.isfinite(prod64) & (np.abs(prod64) > 1e-30) & (np.abs(prod64) < 1e38)
        with np.errstate(over="ignore", under="ignore"):
            assert np.array_equal((a[p] * b[q])[ok].astype(np.float64), prod64[ok])


def test_pair_ladder_error_against_fp32_rounding():
    rng = np.random.default_rng(1)
    K = 1152
    a = rng.standard_normal((256, K)).astype(np.float32)
    b = (rng.standard_normal((K, 64)) * 0.05).astype(np.float32)
    A, B = planes(a), planes(b)
    exact = a.astype(np.float64) @ b.astype(np.float64)
    ymax = np.abs(exact).max()
    err = {}
    for name, n in (("p9", 9), ("p6", 6), ("p3", 3)):
        acc = np.zeros_like(exact)
        for p, q in PAIRS[9 - n:]:
            acc += A[p].astype(np.float64) @ B[q].astype(np.float64)
        err[name] = np.abs(acc - exact).max() / ymax
    fp32 = np.abs((a @ b) - exact).max() / ymax
    assert err["p9"] < 1e-14                  # all nine pairs: the decomposition itself loses nothing
    assert err["p6"] < 5e-8 < fp32 * 0.5      # six pairs: truncation well below fp32's own rounding of the same sum
    assert 1e-7 < err["p3"] < 2e-5            # three pairs is a 16-bit-significand product: not an fp32 substitute


TILES = [(128, 128, 16, 2, 2), (128, 64, 16, 2, 2), (128, 128, 32, 2, 2), (64, 64, 16, 2, 2)]    # conv_gemm_cfgs.inc groups 5 / 6


@pytest.mark.parametrize("BM,BN,BK,WAVES_M,WAVES_N", TILES)
def test_lds_staging_emulation(BM, BN, BK, WAVES_M, WAVES_N):
    KH, KS, KQ = BK // 8, BK // 16, BK // 4
    A_ITERS, A_MSTEP = BM * KQ // 256, 256 // KQ
    SA, SB = BM, BN                                     # no padding: the A rows are XOR-swizzled per k slab instead
    swz = lambda kh: kh * (64 // BK)                    # split_swz<BK>
    A_TILE, B_TILE = 3 * KH * SA, 3 * KH * SB
    B_CPP = KH * BN
    B_CELLS = 3 * B_CPP
    B_ITERS = (B_CELLS + 255) // 256
    WM, WN = BM // WAVES_M, BN // WAVES_N
    TM, TN = WM // 32, WN // 32
    assert A_ITERS >= 1 and (BM * KQ) % 256 == 0
    # LDS as 16-bit slots holding (plane, row-or-column, k) tags; -1 = never written
    a_lds = np.full((A_TILE * 8, 3), -1, np.int64)
    b_lds = np.full((B_TILE * 8, 3), -1, np.int64)
    for tid in range(256):
        aq, am = tid % KQ, tid // KQ
        kh, half = aq >> 1, aq & 1
        for i in range(A_ITERS):
            ml = (am + i * A_MSTEP) ^ swz(kh)
            for pl in range(3):
                u = ((pl * KH + kh) * SA + ml) * 2 + half           # u32x2 index (store_tile)
                for j in range(4):                                  # dword 0 = (k0, k1), dword 1 = (k2, k3)
                    assert a_lds[u * 4 + j, 0] == -1
                    a_lds[u * 4 + j] = (pl, am + i * A_MSTEP, aq * 4 + j)
        for i in range(B_ITERS):
            c = tid + i * 256
            if c >= B_CELLS:
                continue
            pl, rem = divmod(c, B_CPP)
            kh, n = divmod(rem, BN)
            dst = (pl * KH + kh) * SB + n
            for j in range(8):                                      # a packed cell: 8 consecutive k of column n (mit_gemm_split_pack)
                assert b_lds[dst * 8 + j, 0] == -1
                b_lds[dst * 8 + j] = (pl, n, kh * 8 + j)
    # bank conflicts of the A stores under the LDS model of MI355X_MICROARCH.md: ds_write_b64 is served in groups of 16 consecutive
    # lanes, bank of byte address a = (a / 4) mod 32 — a group's 16 eight-byte halves must cover the 32 banks exactly once
    for wave in range(4):
        for g in range(4):
            for i in range(A_ITERS):
                for pl in range(3):
                    banks = []
                    for lane in range(16 * g, 16 * g + 16):
                        tid = wave * 64 + lane
                        aq, am = tid % KQ, tid // KQ
                        kh, half = aq >> 1, aq & 1
                        byte = (((pl * KH + kh) * SA + ((am + i * A_MSTEP) ^ swz(kh))) * 2 + half) * 8
                        banks += [(byte // 4) % 32, (byte // 4 + 1) % 32]
                    assert sorted(banks) == list(range(32)), (wave, g, i, pl, sorted(banks))
    # ... and of the A fragment reads: ds_read_b128 lane groups {0-3,12-15,20-27} ... over 64 banks
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for lh in range(2):
            for ks in range(KS):
                banks = []
                for li in grp:
                    byte = ((2 * ks + lh) * SA + (li ^ swz(2 * ks + lh))) * 16
                    banks += [(byte // 4 + j) % 64 for j in range(4)]
                assert sorted(banks) == list(range(64))
    for wave in range(4):
        wm0, wn0 = (wave // WAVES_N) * WM, (wave % WAVES_N) * WN
        for lane in range(64):
            li, lh = lane & 31, lane >> 5
            for ks in range(KS):
                for pl in range(3):
                    for mi in range(TM):
                        cell = (pl * KH + 2 * ks + lh) * SA + ((wm0 + mi * 32 + li) ^ swz(2 * ks + lh))
                        got = a_lds[cell * 8:cell * 8 + 8]
                        assert np.array_equal(got[:, 0], [pl] * 8) and np.array_equal(got[:, 1], [wm0 + mi * 32 + li] * 8)
                        ka = got[:, 2]
                    for ni in range(TN):
                        cell = lh * SB + wn0 + li + (pl * KH + 2 * ks) * SB + ni * 32
                        got = b_lds[cell * 8:cell * 8 + 8]
                        assert np.array_equal(got[:, 0], [pl] * 8) and np.array_equal(got[:, 1], [wn0 + ni * 32 + li] * 8)
                        kb = got[:, 2]
                    assert np.array_equal(ka, kb) and np.array_equal(ka, (2 * ks + lh) * 8 + np.arange(8))
    # every k of the tile is consumed exactly once per (row, column): lanes halves x steps cover BK
    assert sorted(((2 * ks + lh) * 8 + j) for ks in range(KS) for lh in range(2) for j in range(8)) == list(range(BK))
    # (the A stores' bank check is above, in the hardware's own terms: groups of 16 consecutive lanes over 32 banks.  Until round 6 this
    # test — and the kernel — assumed half-waves over 64 banks; SQ_LDS_BANK_CONFLICT showed a 2-way conflict on every A store,
    # profiles/r10p_pmc_split_tile.json -> r10q: 20.5 M -> 0.37 M conflict cycles per launch)
    # ds_read_b128 of the A fragments: each group of 16 consecutive lanes reads 16 distinct cells of one aligned 16-cell run
    for lh in range(2):
        for ks in range(KS):
            for g in range(2):
                rows = [(g * 16 + l) ^ swz(2 * ks + lh) for l in range(16)]
                assert len(set(rows)) == 16 and len({r // 16 for r in rows}) == 1


def test_pack_layout_matches_kernel_addressing():
    """mit_gemm_split_pack's cell order [z][plane][k / 8][n][8] against the kernel's W addressing (b_src + ld_k8 * ldn)."""
    nz, Kp, Np, BN, BK, n0 = 2, 48, 72, 64, 16, 64
    K8, KH = Kp // 8, BK // 8
    cells = np.arange(nz * 3 * K8 * Np).reshape(nz, 3, K8, Np)              # cell ids in memory order
    ws_zs0 = 3 * Kp * Np                                                    # uint16 elements per slice
    B_CPP = KH * BN
    for z0 in range(nz):
        base = z0 * ws_zs0 // 8                                             # u32x4 pointer arithmetic: 8 elements per cell
        for kt in range(Kp // BK):
            for c in range(3 * B_CPP):
                pl, rem = divmod(c, B_CPP)
                kh, n = divmod(rem, BN)
                if n0 + n >= Np:
                    continue
                idx = base + (kt * KH) * Np + (pl * K8 + kh) * Np + n0 + n
                assert idx == cells[z0, pl, kt * KH + kh, n0 + n]


def test_descriptor_picks_up_registered_planes_by_identity():
    w = torch.zeros(32, 8)
    a = torch.zeros(1, 1, 4, 32)
    out = torch.zeros(1, 1, 4, 8)
    mk = lambda ww, **kw: ops.conv_gemm_desc(a=a, NB=1, Hi=1, Wi=4, Cin=32, a_strides=(128, 128, 32), Ho=1, Wo=4, sy=1, sx=1,
                                              taps=[(0, 0, 0)], pad_mode=ops.PAD_ZERO, w=ww, ldw=kw.get("ldw", 8), Kw=32, Nw=8, N=8,
                                              c=ops.tensor_map(out), w_zs=kw.get("w_zs", (0, 0)))
    assert not mk(w).w_split
    fake = torch.zeros(1, 3, 4, 8, 8, dtype=torch.int16)
    import weakref
    key = w.data_ptr()
    ops._SPLITS[key] = (weakref.ref(w, lambda _r, k=key: ops._SPLITS.pop(k, None)), fake, 1, 32, 8)
    try:
        d = mk(w)
        assert d.w_split == fake.data_ptr() and d.ws_zs0 == 0
        assert not mk(w, ldw=12).w_split                                    # another leading dimension: not the registered layout
        assert not mk(w, w_zs=(8, 0)).w_split                               # z1-strided weights are not supported by the split tiles
        other = torch.zeros(32, 8)
        assert not mk(other).w_split
        view = w[:]                                                         # same storage, different tensor object: identity decides
        assert view.data_ptr() == w.data_ptr() and not mk(view).w_split
        del w, d, view
        gc.collect()
        assert key not in ops._SPLITS                                       # the entry dies with the weight: no stale pointers
    finally:
        ops._SPLITS.pop(key, None)


def test_descriptor_strides_for_batched_weights():
    """[nz, Kp, Np] weights (the 36 Winograd slices): planes are found only for the plain slice stride, and ws_zs0 is one slice of planes."""
    import weakref

    u = torch.zeros(4, 16, 8)
    a = torch.zeros(1, 1, 4, 16)
    out = torch.zeros(1, 1, 4, 8)
    mk = lambda w_zs: ops.conv_gemm_desc(a=a, NB=1, Hi=1, Wi=4, Cin=16, a_strides=(64, 64, 16), Ho=1, Wo=4, sy=1, sx=1, taps=[(0, 0, 0)],
                                         pad_mode=ops.PAD_ZERO, w=u, ldw=8, Kw=16, Nw=8, N=8, c=ops.tensor_map(out), Z=4, zdiv=1 << 30, w_zs=w_zs)
    fake = torch.zeros(4, 3, 2, 8, 8, dtype=torch.int16)
    key = u.data_ptr()
    ops._SPLITS[key] = (weakref.ref(u), fake, 4, 16, 8)
    try:
        d = mk((0, 16 * 8))
        assert d.w_split == fake.data_ptr() and d.ws_zs0 == 3 * 16 * 8
        assert not mk((0, 2 * 16 * 8)).w_split          # every other slice: not the packed layout
        assert not mk((0, 0)).w_split                    # one slice broadcast over z: the planes hold four different ones
    finally:
        ops._SPLITS.pop(key, None)


def test_gemm_mode_switch_and_initial_value():
    """mit_gemm_mode_set / _get (include/mit_hip.h): 6 unless MIT_GEMM_SPLIT says otherwise, switchable at run time, anything but
    0 | 6 | 9 refused (the 3-pair tiles are a test ladder reachable by explicit tile index only)."""
    import subprocess
    import sys

    prev = ops.split_mode()
    try:
        for m in (0, 9, 6):
            ops.set_split_mode(m)
            assert ops.split_mode() == m
        for bad in (3, 1, -1, 12):
            with pytest.raises(RuntimeError, match="mode must be"):
                ops.set_split_mode(bad)
        assert ops.split_mode() == 6
        with ops.gemm_mode(0):
            assert ops.split_mode() == 0
        assert ops.split_mode() == 6
    finally:
        ops.set_split_mode(prev)
    assert ops.register_split(torch.zeros(16, 4)) is None                   # CPU tensors are never split (no GPU, no planes)
    code = "from manga_image_translator_amd import lib; print(lib.load(build_if_missing=False).mit_gemm_mode_get())"
    for env, want in ((None, 6), ("0", 0), ("9", 9), ("6", 6), ("3", 0), ("x", 0), ("", 6)):
        e = {k: v for k, v in os.environ.items() if k != "MIT_GEMM_SPLIT"}
        if env is not None:
            e["MIT_GEMM_SPLIT"] = env
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, cwd=ROOT, check=True).stdout
        assert int(out.strip().splitlines()[-1]) == want, (env, out)
