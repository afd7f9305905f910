"""Text-line merge: groups recognised lines into text regions — the step right after OCR.

Mirror of the reference's ``textline_merge`` (/root/reference/manga_translator/textline_merge/__init__.py:10-208):
merge graph (``quadrilateral_can_merge_region`` with the coarse tolerances of :140-141), recursive split of each connected
component along its most deviating minimum-spanning-tree edge (``split_text_region`` :10-88), per-region colour means,
direction vote and reading order (:150-184), and ``dispatch`` (:186-208) building the region objects.  Host-side logic
(no GPU work); shapely / networkx are replaced by ``textline.polygon_distance`` / a hull area and a small Kruskal.

Pinned by the reference's own golden tests: tests/golden/textline_merge.json holds the line sets and expected groupings of
test/test_textline_merge.py, extracted by oracle/make_golden.py, together with what the reference code itself returns.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import Iterable, List, Sequence, Set, Tuple

import numpy as np

from . import textline as TL
from .textline import Quadrilateral


def _hull_area(points) -> float:
    h = TL._convex_hull(np.asarray(points, dtype=np.float64))
    if len(h) < 3:
        return 0.0
    x, y = h[:, 0], h[:, 1]
    return float(abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))) / 2)


def _dist(x1, y1, x2, y2) -> float:  # utils/generic2.py:73-74
    return math.sqrt((x1 - x2) ** 2 + (y1 - y2) ** 2)


def line_distance(a: Quadrilateral, b: Quadrilateral, rho: float = 0.5) -> float:
    """Quadrilateral.distance_impl (utils/generic.py:546-596): distance between matching anchor points of two lines
    (left / right / middle for horizontal text, top / bottom for vertical), chosen by how well the lines are aligned."""
    fs = max(a.font_size, b.font_size)
    if a.assigned_direction == "h":
        d1 = _hull_area([a.pts[0], a.pts[3], b.pts[0], b.pts[3]]) / fs
        d2 = _hull_area([a.pts[2], a.pts[1], b.pts[2], b.pts[1]]) / fs
        d3 = _hull_area([a.structure[0], a.structure[1], b.structure[0], b.structure[1]]) / fs
        pattern = "h_left"
        if d1 < fs * rho:
            pattern = "h_left"
        if d2 < fs * rho and d2 < d1:
            pattern = "h_right"
        if d3 < fs * rho and d3 < d1 and d3 < d2:
            pattern = "h_middle"
        if pattern == "h_left":
            return _dist(a.pts[0][0], a.pts[0][1], b.pts[0][0], b.pts[0][1])
        if pattern == "h_right":
            return _dist(a.pts[1][0], a.pts[1][1], b.pts[1][0], b.pts[1][1])
        return _dist(a.structure[0][0], a.structure[0][1], b.structure[0][0], b.structure[0][1])
    d1 = _hull_area([a.pts[0], a.pts[1], b.pts[0], b.pts[1]]) / fs
    d2 = _hull_area([a.pts[2], a.pts[3], b.pts[2], b.pts[3]]) / fs
    pattern = "v_top"
    if d1 < fs * rho:
        pattern = "v_top"
    if d2 < fs * rho and d2 < d1:
        pattern = "v_bottom"
    if pattern == "v_top":
        return _dist(a.pts[0][0], a.pts[0][1], b.pts[0][0], b.pts[0][1])
    return _dist(a.pts[2][0], a.pts[2][1], b.pts[2][0], b.pts[2][1])


def _components(nodes: Sequence[int], edges: Iterable[Tuple[int, int]]) -> List[Set[int]]:
    """Connected components by breadth-first search, in first-node order.  Each component is a Python ``set`` filled in BFS
    order over insertion-ordered adjacency — the construction networkx's ``connected_components`` uses — so that iterating
    it (``list(component)``, which decides tie order downstream) gives the order the reference sees."""
    adj = {n: [] for n in nodes}
    for u, v in edges:
        if v not in adj[u]:
            adj[u].append(v)
            adj[v].append(u)
    done: Set[int] = set()
    out: List[Set[int]] = []
    for s in nodes:
        if s in done:
            continue
        seen = {s}
        level = [s]
        while level and len(seen) < len(nodes):
            nxt = []
            for v in level:
                for w in adj[v]:
                    if w not in seen:
                        seen.add(w)
                        nxt.append(w)
            level = nxt
        done.update(seen)
        out.append(seen)
    return out


def _kruskal(nodes: Sequence[int], weighted: List[Tuple[int, int, float]]) -> List[Tuple[int, int, float]]:
    """networkx minimum_spanning_edges(algorithm='kruskal'): edges sorted by weight (stable), union-find."""
    parent = {n: n for n in nodes}

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    out = []
    for u, v, w in sorted(weighted, key=lambda e: e[2]):
        ru, rv = find(u), find(v)
        if ru != rv:
            parent[rv] = ru
            out.append((u, v, w))
    return out


def split_text_region(bboxes: Sequence[Quadrilateral], indices: Iterable[int], width, height, gamma=0.5, sigma=2) -> List[Set[int]]:
    """textline_merge/__init__.py:10-88."""
    idx = list(indices)
    if len(idx) == 1:
        return [set(idx)]
    if len(idx) == 2:
        a, b = bboxes[idx[0]], bboxes[idx[1]]
        fs = max(a.font_size, b.font_size)
        if line_distance(a, b) < (1 + gamma) * fs and abs(a.angle - b.angle) < 0.2 * np.pi:
            return [set(idx)]
        return [{idx[0]}, {idx[1]}]
    weighted = [(u, v, line_distance(bboxes[u], bboxes[v])) for u, v in itertools.combinations(idx, 2)]
    edges = sorted(_kruskal(idx, weighted), key=lambda e: e[2], reverse=True)
    dists = [e[2] for e in edges]
    fontsize = np.mean([bboxes[i].font_size for i in idx])
    d_std, d_mean = np.std(dists), np.mean(dists)
    std_threshold = max(0.3 * fontsize + 5, 5)
    b1, b2 = bboxes[edges[0][0]], bboxes[edges[0][1]]
    max_poly_distance = TL.polygon_distance(b1.pts, b2.pts)
    max_centroid_alignment = min(abs(b1.centroid[0] - b2.centroid[0]), abs(b1.centroid[1] - b2.centroid[1]))
    if (dists[0] <= d_mean + d_std * sigma or dists[0] <= fontsize * (1 + gamma)) and \
            (d_std < std_threshold or max_poly_distance == 0 and max_centroid_alignment < 5):
        return [set(idx)]
    ans: List[Set[int]] = []
    for comp in _components(idx, [(u, v) for u, v, _ in edges[1:]]):  # drop the most deviating edge, recurse
        ans.extend(split_text_region(bboxes, comp, width, height))
    return ans


def merge_bboxes_text_region(bboxes: Sequence[Quadrilateral], width, height):
    """textline_merge/__init__.py:112-184: yields (lines in reading order, fg colour, bg colour) per text region."""
    n = len(bboxes)
    pairs = TL.near_pairs(bboxes)   # the pairs of itertools.combinations(range(n), 2) that can pass the predicate's first test
    edges = [(u, v) for (u, v), d in zip(pairs, TL.quad_pair_distances(bboxes, pairs))
             if TL.quadrilateral_can_merge_region(bboxes[u], bboxes[v], aspect_ratio_tol=1.3, font_size_ratio_tol=2,
                                                  char_gap_tolerance=1, char_gap_tolerance2=3, dist=d)]
    regions: List[Set[int]] = []
    for comp in _components(list(range(n)), edges):
        regions.extend(split_text_region(bboxes, comp, width, height))
    for node_set in regions:
        nodes = list(node_set)
        lines = [bboxes[i] for i in nodes]
        # round(np.mean(...)) of the integer colour components: the integer sum is exact in float64, so the true division below is the
        # same double (six numpy reductions per region cost a millisecond per page)
        mean = lambda attr: round(sum(int(getattr(b, attr)) for b in lines) / len(lines))
        fg = (mean("fg_r"), mean("fg_g"), mean("fg_b"))
        bg = (mean("bg_r"), mean("bg_g"), mean("bg_b"))
        dirs = [b.direction for b in lines]
        counts = {}
        for d in dirs:
            counts[d] = counts.get(d, 0) + 1
        top = sorted(counts.items(), key=lambda kv: (-kv[1], dirs.index(kv[0])))[:2]  # Counter.most_common(2)
        if len(top) == 1 or top[0][1] != top[1][1]:
            majority = top[0][0]
        else:  # tie: the line with the most extreme aspect ratio decides (:160-168)
            best = -100
            majority = top[0][0]
            for b in lines:
                if b.aspect_ratio > best:
                    best, majority = b.aspect_ratio, b.direction
                if 1.0 / b.aspect_ratio > best:
                    best, majority = 1.0 / b.aspect_ratio, b.direction
        if majority == "h":
            nodes = sorted(nodes, key=lambda i: bboxes[i].centroid[1])
        elif majority == "v":
            nodes = sorted(nodes, key=lambda i: -bboxes[i].centroid[0])
        yield [bboxes[i] for i in nodes], fg, bg


@dataclass
class TextBlock:
    """The fields of the reference's TextBlock (utils/textblock.py:39-110) that ``dispatch`` fills."""
    lines: np.ndarray                   # int32 [n, 4, 2]
    texts: List[str]
    font_size: int
    angle: float
    prob: float
    fg_colors: Tuple[int, int, int]
    bg_colors: Tuple[int, int, int]
    text: str = ""
    translation: str = ""
    language: str = "unknown"

    def __post_init__(self):
        self.lines = np.array(self.lines, dtype=np.int32)
        self.font_size = round(self.font_size)
        self.text = self.texts[0] if self.texts else ""
        if self.text and len(self.texts) > 1:  # CJK lines join without a space (:81-89)
            for txt in self.texts[1:]:
                first_cjk = "\u3000" <= self.text[-1] <= "\u9fff"
                second_cjk = bool(txt) and ("\u3000" <= txt[0] <= "\u9fff")
                self.text += txt if (first_cjk or second_cjk) else " " + txt


def dispatch_sync(textlines: Sequence[Quadrilateral], width: int, height: int, block_factory=None) -> List[TextBlock]:
    """textline_merge.dispatch (:186-208).  Keeps the reference's normalisation of the region probability by the area of ALL
    text lines (:197).  ``block_factory(lines, texts, font_size, angle, prob, fg, bg)`` builds the region object — the mirror
    ``TextBlock`` by default; inside the reference pass ``lambda l, t, fs, a, p, fg, bg: manga_translator.utils.TextBlock(l, t,
    font_size=fs, angle=a, prob=p, fg_color=fg, bg_color=bg)`` (its own call, :199-206)."""
    make = block_factory or TextBlock
    regions: List[TextBlock] = []
    total_area = sum(t.area for t in textlines)
    for lines, fg, bg in merge_bboxes_text_region(textlines, width, height):
        logp = sum(np.log(t.prob) * t.area for t in lines) / total_area
        angle = np.rad2deg(np.mean([t.angle for t in lines])) - 90
        if abs(angle) < 3:
            angle = 0
        regions.append(make([t.pts for t in lines], [t.text for t in lines], int(min(t.font_size for t in lines)), float(angle),
                            float(np.exp(logp)), fg, bg))
    return regions


async def dispatch(textlines: Sequence[Quadrilateral], width: int, height: int, verbose: bool = False, block_factory=None) -> List[TextBlock]:
    return dispatch_sync(textlines, width, height, block_factory)
