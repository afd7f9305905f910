"""Text-line merge against the reference's own golden tests.

tests/golden/textline_merge.json = the 11 cases of the reference's test/test_textline_merge.py (line sets + expected groupings,
extracted by oracle/make_golden.py) plus what the reference's merge_bboxes_text_region itself returned on them (group order,
reading order inside each group, colour means for seeded colours)."""
import asyncio
import json
import os

import numpy as np
import pytest

from manga_image_translator_amd import textline as TL
from manga_image_translator_amd import textline_merge as TM

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "textline_merge.json")))
CASES = GOLD["cases"]


def _quads(case):
    return [TL.Quadrilateral(np.array(l), "", 1, *col) for l, col in zip(case["lines"], case["colors_in"])]


def _run(case):
    quads = _quads(case)
    groups, colors = [], []
    for lines, fg, bg in TM.merge_bboxes_text_region(quads, case["width"], case["height"]):
        groups.append([next(i for i, q in enumerate(quads) if q is t) for t in lines])
        colors.append([list(fg), list(bg)])
    return groups, colors


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_golden_groupings(case):
    """The assertion of the reference's run_test (test/test_textline_merge.py:27-46): same groups, any order."""
    groups, _ = _run(case)
    assert sorted(map(sorted, groups)) == sorted(map(sorted, case["expected"]))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_matches_reference_output_exactly(case):
    """Stronger: region order, reading order within each region and colour means equal the reference code's output."""
    assert case["ref_passes_own_test"]
    groups, colors = _run(case)
    assert groups == case["ref_groups"]
    assert colors == case["ref_colors"]


def test_line_distance_patterns():
    a = TL.Quadrilateral(np.array([[0, 0], [100, 0], [100, 20], [0, 20]]))
    b = TL.Quadrilateral(np.array([[0, 30], [60, 30], [60, 50], [0, 50]]))
    a.assigned_direction = b.assigned_direction = "h"      # left-aligned lines: anchor = left corners
    assert TM.line_distance(a, b) == 30.0
    c = TL.Quadrilateral(np.array([[40, 30], [100, 30], [100, 50], [40, 50]]))
    c.assigned_direction = "h"                              # right-aligned
    assert TM.line_distance(a, c) == 30.0
    a.assigned_direction = None                             # default branch = vertical anchors (top corners)
    assert TM.line_distance(a, b) == 30.0
    assert abs(TM.line_distance(a, c) - 50.0) < 1e-9


def test_dispatch_blocks():
    case = CASES[0]
    quads = _quads(case)
    rng = np.random.default_rng(0)
    for i, q in enumerate(quads):
        q.text, q.prob = f"t{i}", float(rng.uniform(0.5, 1.0))
    blocks = asyncio.run(TM.dispatch(quads, case["width"], case["height"]))
    assert sorted(sorted(map(tuple, b.lines.reshape(len(b.lines), -1).tolist())) for b in blocks) == \
        sorted(sorted(tuple(np.array(TL.sort_pnts(np.array(case["lines"][i]))[0]).reshape(-1).tolist()) for i in g) for g in case["expected"])
    total = sum(q.area for q in quads)
    for b, g in zip(blocks, case["ref_groups"]):
        assert b.texts == [f"t{i}" for i in g] and b.text == " ".join(b.texts)
        want = np.exp(sum(np.log(quads[i].prob) * quads[i].area for i in g) / total)
        assert abs(b.prob - want) < 1e-12
        assert b.font_size == round(int(min(quads[i].font_size for i in g)))
        assert b.lines.dtype == np.int32 and b.lines.shape == (len(g), 4, 2)
        assert b.angle == 0 or abs(b.angle) >= 3


def test_cjk_join():
    b = TM.TextBlock(np.zeros((2, 4, 2)), ["こん", "にちは"], 10, 0, 1.0, (0, 0, 0), (0, 0, 0))
    assert b.text == "こんにちは"
    b = TM.TextBlock(np.zeros((2, 4, 2)), ["hello", "world"], 10, 0, 1.0, (0, 0, 0), (0, 0, 0))
    assert b.text == "hello world"


def test_block_factory_receives_the_reference_constructor_arguments():
    case = CASES[0]
    quads = _quads(case)
    for q in quads:
        q.text, q.prob = "x", 0.9
    got = TM.dispatch_sync(quads, case["width"], case["height"], block_factory=lambda *a: a)
    ref = TM.dispatch_sync(quads, case["width"], case["height"])
    assert len(got) == len(ref)
    for (lines, texts, fs, angle, prob, fg, bg), b in zip(got, ref):
        assert np.array_equal(np.array(lines, dtype=np.int32), b.lines) and texts == b.texts
        assert (fs, angle, fg, bg) == (b.font_size, b.angle, b.fg_colors, b.bg_colors) and abs(prob - b.prob) < 1e-15
