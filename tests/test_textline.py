"""Host-side text-line logic of the product (manga_image_translator_amd/textline.py) vs the golden vectors made by the
reference's own ``sort_pnts`` / ``Quadrilateral`` / ``get_transformed_region`` (tests/golden/textline.npz), and vs the
oracle.  Integer results (corner order, direction, crop geometry, chunk plan) must be identical."""
import os

import numpy as np
import pytest

from manga_image_translator_amd import textline as TL
from oracle import ocr48 as OO, textline as OT

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textline.npz"))


def test_quadrilateral_matches_reference_fixture():
    for k in range(len(G["quads"])):
        q = TL.Quadrilateral(G["quads"][k])
        assert np.array_equal(q.pts, G["sorted_pts"][k])
        assert q.direction == str(G["direction"][k])
        assert q.aspect_ratio == pytest.approx(float(G["aspect_ratio"][k]), rel=1e-6)
        assert q.font_size == pytest.approx(float(G["font_size"][k]), rel=1e-6)
        assert q.valid in (True, False)
        bb = q.aabb
        assert (bb.x, bb.y) == tuple(G["quads"][k].min(0)) and bb.w >= 0 and q.area > 0


def test_warp_plan_geometry_matches_reference_crops():
    H, W = G["image"].shape[:2]
    for k in range(len(G["quads"])):
        q = TL.Quadrilateral(G["quads"][k])
        pl = TL.warp_plan(q, q.direction, H, W)
        assert pl.width == int(G["crop_width"][k])
        assert (pl.dw, pl.dh) == ((48, pl.width) if q.direction == "v" else (pl.width, 48))
        assert q.assigned_direction == q.direction
        # the inverse map sends the destination corners back onto the (crop-relative) quad corners
        dst = np.array([[0, 0, 1], [pl.dw - 1, 0, 1], [pl.dw - 1, pl.dh - 1, 1], [0, pl.dh - 1, 1]], dtype=np.float64)
        back = (pl.minv @ dst.T).T
        back = back[:, :2] / back[:, 2:]
        assert np.allclose(back, q.pts - np.array([pl.x1, pl.y1]), atol=1e-6)
        assert 0 <= pl.x1 and pl.x1 + pl.cw <= W and 0 <= pl.y1 and pl.y1 + pl.ch <= H
    with pytest.raises(ValueError):
        TL.warp_plan(TL.Quadrilateral(G["quads"][0]), "x", H, W)
    with pytest.raises(ValueError):
        TL.sort_pnts(np.zeros((3, 2)))


def test_homography_solver_agreement_and_tie_sensitivity():
    """8x8 solve (product, oracle) == DLT/SVD solve to ~1e-9; the rectified crops they produce differ in at most a
    handful of pixels (exact 1/32 rounding ties), which bounds how much real OpenCV could differ for this step."""
    H, W = G["image"].shape[:2]
    flips = total = 0
    for k in range(len(G["quads"])):
        q = TL.Quadrilateral(G["quads"][k])
        pl = TL.warp_plan(q, q.direction, H, W)
        src = q.pts.astype(np.int64) - np.array([pl.x1, pl.y1])
        dst = np.array([[0, 0], [pl.dw - 1, 0], [pl.dw - 1, pl.dh - 1], [0, pl.dh - 1]], dtype=np.float32)
        a, b = OT.find_homography_4pt(src, dst), OT.find_homography_4pt_dlt(src, dst)
        assert np.array_equal(a, TL.homography_4pt(src, dst))
        assert np.allclose(a, b, rtol=1e-8, atol=1e-9)
        crop = G["image"][pl.y1:pl.y1 + pl.ch, pl.x1:pl.x1 + pl.cw]
        ra, rb = OT.warp_perspective_u8(crop, a, (pl.dw, pl.dh)), OT.warp_perspective_u8(crop, b, (pl.dw, pl.dh))
        flips += int((ra != rb).any(-1).sum())
        total += ra.shape[0] * ra.shape[1]
    assert flips <= 0.01 * total, (flips, total)


def test_chunk_plan_matches_reference_batching():
    rng = np.random.default_rng(0)
    for n in (1, 5, 16, 17, 40):
        widths = rng.integers(20, 600, size=n).tolist()
        if n == 40:
            widths[3] = widths[7] = widths[11]  # ties must keep input order (stable sort, model_48px.py:79)
        crops = [np.zeros((48, w, 3), dtype=np.uint8) for w in widths]
        ref = [(idx, ws, int(t.shape[3])) for idx, ws, t in OO.make_chunks(crops)]
        assert TL.chunk_plan(widths) == ref
    assert TL.chunk_plan([]) == []


def test_batched_plans_equal_per_line_plans():
    """textline.warp_plans (one batched 8x8 solve per page) gives bit-identical records to warp_plan line by line."""
    H, W = G["image"].shape[:2]
    quads = [TL.Quadrilateral(q) for q in G["quads"]]
    dirs = [q.direction for q in quads]
    dirs[0] = "h" if dirs[0] == "v" else "v"  # an overridden direction (majority vote of a merge-graph component)
    rec = TL.warp_plans(quads, dirs, H, W)
    assert rec.dtype.itemsize == 112
    for r, q, d in zip(rec, quads, dirs):
        pl = TL.warp_plan(q, d, H, W)
        assert (r["x1"], r["y1"], r["cw"], r["ch"], r["dw"], r["dh"], bool(r["vertical"])) == (pl.x1, pl.y1, pl.cw, pl.ch, pl.dw, pl.dh, pl.vertical)
        assert np.array_equal(r["minv"].reshape(3, 3), pl.minv)
        assert q.assigned_direction == d
    assert len(TL.warp_plans([], [], H, W)) == 0
    with pytest.raises(ValueError):
        TL.warp_plans(quads[:1], ["x"], H, W)


def test_warp_record_layout_matches_c_struct():
    import ctypes as C

    from manga_image_translator_amd import lib

    assert TL.WARP_LINE_DTYPE.itemsize == C.sizeof(lib.MitWarpLine)
    for name, *_ in lib.MitWarpLine._fields_:
        assert TL.WARP_LINE_DTYPE.fields[name][1] == getattr(lib.MitWarpLine, name).offset, name


def test_merge_graph_and_direction_vote_match_reference_fixture():
    """quadrilateral_can_merge_region + the OCR-side direction vote vs the reference's own code
    (utils/generic.py:653-698, ocr/common.py:12-39, run with a shapely stand-in; tests/golden/direction.npz)."""
    D = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "direction.npz"))
    edges = 0
    for s in range(int(D["n_sets"])):
        quads = [TL.Quadrilateral(q) for q in D[f"quads{s}"]]
        n = len(quads)
        m = np.zeros((n, n), dtype=np.uint8)
        for u in range(n):
            for v in range(u + 1, n):
                if TL.quadrilateral_can_merge_region(quads[u], quads[v], aspect_ratio_tol=1):
                    m[u, v] = m[v, u] = 1
        assert np.array_equal(m, D[f"merge{s}"]), s
        edges += int(m.sum() // 2)
        got = list(TL.generate_text_direction(quads))
        assert [quads.index(q) for q, _ in got] == D[f"order{s}"].tolist()
        assert [d for _, d in got] == D[f"dir{s}"].tolist()
    assert edges >= 10


def test_polygon_distance_closed_form():
    sq = lambda x, y, s: np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s]], dtype=float)
    assert TL.polygon_distance(sq(0, 0, 10), sq(13, 0, 10)) == pytest.approx(3.0)
    assert TL.polygon_distance(sq(0, 0, 10), sq(13, 14, 10)) == pytest.approx(5.0)       # corner to corner (3, 4, 5)
    assert TL.polygon_distance(sq(0, 0, 10), sq(5, 5, 10)) == 0.0                          # overlap
    assert TL.polygon_distance(sq(0, 0, 10), sq(2, 2, 3)) == 0.0                           # containment
    assert TL.polygon_distance(sq(0, 0, 10), sq(10, 0, 4)) == 0.0                          # touching edge
    a, b = TL.Quadrilateral(sq(0, 0, 10).astype(int)), TL.Quadrilateral(sq(20, 0, 10).astype(int))
    assert a.poly_distance(b) == pytest.approx(10.0) and a.is_axis_aligned and a.is_approximate_axis_aligned
    assert a.xyxy == (0, 0, 10, 10) and np.allclose(a.centroid, [5, 5])


def test_is_ignore_matches_the_reference_function():
    """textline.is_ignore against utils/bubble.py:28-84 itself (tests/golden/bubble.npz: the reference's function on seeded crops —
    bubbles, artwork, coloured interiors, frames exactly on the thresholds — for --ignore-bubble in {0, 1, 5, 10, 25, 50, 51})."""
    import os

    from manga_image_translator_amd import textline as TL
    from oracle import make_golden as MG

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bubble.npz"))
    crops = MG.bubble_crops()
    assert [c.shape[1] for c in crops] == g["widths"].tolist()
    got = np.array([[TL.is_ignore(c, int(lv)) for lv in g["levels"]] for c in crops])
    assert np.array_equal(got, g["want"])
    assert not got[:, 0].any() and not got[:, -1].any() and got[:, 3].any() and not got[:, 3].all()   # 0 and 51 switch it off


def test_native_quad_pair_distances_equal_polygon_distance_bit_for_bit():
    """mit_quad_pair_distances (host C++ behind the direction vote and the merge graph) against textline.polygon_distance on random
    quadrilaterals — apart, touching at a vertex, sharing an edge, crossing, one inside the other, degenerate (repeated vertices): the
    same double arithmetic in the same order, so every distance is the same double."""
    rng = np.random.default_rng(7)
    quads = []
    for k in range(120):
        c = rng.uniform(0, 300, size=2)
        w, h, a = rng.uniform(5, 120), rng.uniform(5, 60), rng.uniform(0, np.pi)
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        pts = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) / 2) @ R.T + c
        if k % 3 == 0:
            pts = np.rint(pts)                      # integer corners: exact ties / touching configurations occur
        quads.append(pts)
    quads.append(np.array([[0, 0], [10, 0], [10, 10], [0, 10]], dtype=np.float64))
    quads.append(np.array([[10, 10], [20, 10], [20, 20], [10, 20]], dtype=np.float64))      # touches the previous one in a corner
    quads.append(np.array([[10, 0], [20, 0], [20, 10], [10, 10]], dtype=np.float64))         # shares an edge with the first
    quads.append(np.array([[2, 2], [4, 2], [4, 4], [2, 4]], dtype=np.float64))               # inside the first
    quads.append(np.array([[5, 5], [5, 5], [5, 5], [5, 5]], dtype=np.float64))               # a point
    quads.append(np.array([[30, 0], [40, 0], [40, 0], [30, 0]], dtype=np.float64))           # a segment
    objs = [TL.Quadrilateral(q) for q in quads]
    for o, q in zip(objs, quads):
        o.pts = q                                   # (the constructor sorts / rounds: compare on the raw rings)
    pairs = [(u, v) for u in range(len(quads)) for v in range(u + 1, len(quads))]
    got = TL.quad_pair_distances(objs, pairs)
    assert len(got) == len(pairs)
    zero = 0
    for (u, v), d in zip(pairs, got):
        ref = TL.polygon_distance(quads[u], quads[v])
        assert d == ref and np.signbit(d) == np.signbit(ref), (u, v, d, ref)
        zero += d == 0.0
    assert zero > 50 and zero < len(pairs) - 1000       # both regimes are exercised
    assert TL.quad_pair_distances(objs, []) == []


def test_native_quad_pair_distances_property():
    """Property form of the comparison above (hypothesis): arbitrary finite corner coordinates, integer-valued ones included (exact ties,
    collinear and repeated corners) — the native distance is the Python routine's double, and it is symmetric in its arguments."""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies
    coord = st.one_of(st.integers(-50, 50).map(float), st.floats(-1e3, 1e3, allow_nan=False, allow_infinity=False, width=64))
    quad = st.lists(st.tuples(coord, coord), min_size=4, max_size=4)

    @hyp.settings(max_examples=300, deadline=None)
    @hyp.given(quad, quad)
    def check(a, b):
        qa, qb = np.array(a, dtype=np.float64), np.array(b, dtype=np.float64)
        oa, ob = TL.Quadrilateral(np.array([[0, 0], [4, 0], [4, 2], [0, 2]])), TL.Quadrilateral(np.array([[0, 0], [4, 0], [4, 2], [0, 2]]))
        oa.pts, ob.pts = qa, qb
        d_ab, d_ba = TL.quad_pair_distances([oa, ob], [(0, 1), (1, 0)])
        ref = TL.polygon_distance(qa, qb)
        assert d_ab == ref, (a, b, d_ab, ref)
        assert d_ba == TL.polygon_distance(qb, qa)

    check()


def test_scalar_sort_pnts_agrees_with_the_numpy_formulation_on_every_input():
    """textline.sort_pnts (the decision sequence on Python scalars, handing tie cases to the numpy formulation) against
    textline._sort_pnts_np (generic.py:324-354 operation for operation) on generic quadrilaterals, axis-aligned and rotated rectangles,
    jittered rectangles, exact squares and tiny-grid points (many equal keys, repeated points), int64 / float32 / float64."""
    from manga_image_translator_amd import textline as TL

    rng = np.random.default_rng(7)
    fallbacks = [0]
    orig = TL._sort_pnts_np

    def counting(p):
        fallbacks[0] += 1
        return orig(p)

    n_rect_fast = 0
    try:
        TL._sort_pnts_np = counting
        for it in range(16000):
            kind = it % 8
            if kind == 0:
                q = rng.integers(0, 2000, (4, 2))
            elif kind == 1:
                x, y, w, h = rng.integers(0, 1000, 4)
                q = np.array([[x, y], [x + w + 1, y], [x + w + 1, y + h + 1], [x, y + h + 1]])[rng.permutation(4)]
            elif kind in (2, 7):
                c = rng.uniform(100, 900, 2); w, h = rng.uniform(5, 400, 2); t = rng.uniform(0, np.pi)
                R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
                q = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) @ R.T + c)[rng.permutation(4)]
                if kind == 7:
                    q = np.rint(q).astype(np.int64)
            elif kind == 3:
                q = rng.uniform(0, 1000, (4, 2)).astype(np.float32)
            elif kind == 4:
                c = rng.integers(100, 900, 2); w, h = rng.integers(1, 300, 2)
                q = (np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) + c + rng.integers(-2, 3, (4, 2)))[rng.permutation(4)].astype(np.int64)
            elif kind == 5:
                s = rng.integers(1, 200); x, y = rng.integers(0, 500, 2)
                q = np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s]])[rng.permutation(4)]
            else:
                q = rng.integers(0, 6, (4, 2))
            before = fallbacks[0]
            got, gv = TL.sort_pnts(q)
            n_rect_fast += kind == 1 and fallbacks[0] == before
            want, wv = orig(q)
            assert gv == wv and got.dtype == want.dtype and np.array_equal(got, want), q.tolist()
    finally:
        TL._sort_pnts_np = orig
    assert n_rect_fast > 1900   # the detector's (and the bench's) axis-aligned boxes take the scalar path


@pytest.mark.parametrize("dtype", [np.int64, np.float64, np.float32])
def test_prefilled_geometry_equals_the_lazy_properties(dtype):
    """textline.prefill_geometry (one vectorised pass over a page's quads) against the per-quad cached properties it replaces —
    structure, font_size, aspect_ratio, aabb, extent, is_approximate_axis_aligned — bit for bit, and the direction vote on top of both."""
    from manga_image_translator_amd import textline as TL

    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(2, 40))
        raw = []
        for i in range(n):
            c = rng.uniform(100, 1900, 2); w, h = rng.uniform(4, 300, 2); t = rng.uniform(0, np.pi) if i % 3 else 0.0
            R = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
            q = np.array([[-w, -h], [w, -h], [w, h], [-w, h]]) @ R.T + c
            raw.append(np.rint(q).astype(dtype) if dtype is np.int64 else q.astype(dtype))
        lazy = [TL.Quadrilateral(q) for q in raw]
        fill = [TL.Quadrilateral(q) for q in raw]
        assert TL.prefill_geometry(fill)
        for a, b in zip(lazy, fill):
            assert all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(a.structure, b.structure))
            for name in ("font_size", "aspect_ratio", "aabb", "extent", "is_approximate_axis_aligned"):
                va, vb = getattr(a, name), b.__dict__[name]
                assert type(va) is type(vb) and va == vb, name
        lazy2 = [TL.Quadrilateral(q) for q in raw]
        orig, TL.prefill_geometry = TL.prefill_geometry, lambda quads: False
        try:
            want = [(id_, d) for id_, d in ((lazy2.index(q), d) for q, d in TL.generate_text_direction(lazy2))]
        finally:
            TL.prefill_geometry = orig
        fill2 = [TL.Quadrilateral(q) for q in raw]
        got = [(fill2.index(q), d) for q, d in TL.generate_text_direction(fill2)]
        assert got == want
    big = [TL.Quadrilateral(np.array([[0, 0], [9000, 0], [9000, 50], [0, 50]]) + i) for i in range(3)]
    assert not TL.prefill_geometry(big)   # vectors past 2048: the lazy path (norm's dot product is no longer exact)
