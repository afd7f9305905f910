"""Host-side glue between the detector networks and the text lines, on the native (C++) routines of libmit_hip.so.

``boxes_from_bitmap`` stands in for ``SegDetectorRepresenter.__call__`` of the reference
(/root/reference/manga_translator/detection/ctd_utils/utils/db_utils.py:40-171 for ``ctd``,
detection/default_utils/dbnet_utils.py:16-144 for ``default``), which runs on OpenCV + pyclipper + shapely.  The native
code restates those libraries' published algorithms (csrc/hostglue.hip); parity with the real libraries is unpinned
because none of them is installed where this repo is built or tested.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import lib as _lib


def boxes_from_bitmap(pred: np.ndarray, thresh: float, dest_width: int, dest_height: int, *, unclip_ratio: float,
                      min_sside: float, box_thresh: float = 0.0, min_sside_out: float = 0.0, roll_start: bool = False,
                      max_candidates: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    """pred f32 [H,W] -> (boxes int64 [n,4,2], scores f32 [n]), one slot per contour like the reference (zeros = skipped)."""
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    if pred.ndim != 2:
        raise ValueError(f"boxes_from_bitmap expects a 2-D map, got shape {pred.shape}")
    bitmap = np.ascontiguousarray(pred > thresh, dtype=np.uint8)  # binarize (db_utils.py:75)
    H, W = pred.shape
    boxes = np.zeros((max_candidates, 4, 2), dtype=np.int64)
    scores = np.zeros(max_candidates, dtype=np.float32)
    n = C.c_int(0)
    lib = _lib.load()
    _lib.check(lib.mit_boxes_from_bitmap(pred.ctypes.data, bitmap.ctypes.data, H, W, int(dest_width), int(dest_height), max_candidates,
                                         float(unclip_ratio), float(min_sside), float(box_thresh), float(min_sside_out), int(roll_start),
                                         boxes.ctypes.data, scores.ctypes.data, C.byref(n)), "mit_boxes_from_bitmap")
    return boxes[:n.value], scores[:n.value]


def ctd_boxes(lines_map: np.ndarray, im_h: int, im_w: int) -> Tuple[np.ndarray, np.ndarray]:
    """ComicTextDetector's call: SegDetectorRepresenter(thresh=0.3) on lines_map[:, 0] (ctd.py:102,156; unclip 1.5, sside >= 2)."""
    return boxes_from_bitmap(lines_map[0, 0], 0.3, im_w, im_h, unclip_ratio=1.5, min_sside=2.0)


def dbnet_boxes(db: np.ndarray, h: int, w: int, text_threshold: float, box_threshold: float, unclip_ratio: float):
    """DefaultDetector's call: dbnet_utils.SegDetectorRepresenter(text_threshold, box_threshold, unclip_ratio) on db[:, 0]
    (default.py:73-77; min_size 3, expanded short side >= 5, corners rolled to start at the smallest x + y)."""
    return boxes_from_bitmap(db[0, 0], text_threshold, w, h, unclip_ratio=unclip_ratio, min_sside=3.0, box_thresh=box_threshold,
                             min_sside_out=5.0, roll_start=True)


def contour_count(bitmap: np.ndarray) -> Tuple[int, int]:
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
    n, p = C.c_int(0), C.c_int64(0)
    _lib.check(_lib.load().mit_find_contours_count(bitmap.ctypes.data, bitmap.shape[0], bitmap.shape[1], C.byref(n), C.byref(p)))
    return n.value, p.value
