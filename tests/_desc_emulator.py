"""Test helper: evaluates a ``MitConvGemm`` descriptor on the CPU with numpy, straight from the documented semantics
in include/mit_hip.h.  It lets the ``not gpu`` suite check the HOST logic (weight packing, tap tables, sub-pixel
ConvTranspose decomposition, BN folding, strided/concat views) against torch without a GPU.  Test infrastructure only:
tiny problems, float64 accumulation, never imported by the package."""
import ctypes as C
import math

import numpy as np


def _view(ptr, n):
    return np.ctypeslib.as_array((C.c_float * int(n)).from_address(int(ptr)))


def _act(v, act, alpha):
    if act == 1:
        return np.maximum(v, 0)
    if act == 2:
        return np.where(v > 0, v, v * alpha)
    if act == 3:
        return v / (1 + np.exp(-v))
    if act == 4:
        return 1 / (1 + np.exp(-v))
    if act == 5:
        return 0.5 * v * (1 + np.vectorize(math.erf)(v * 0.7071067811865476))
    return v


def _map_offsets(tm, z1, z0, nb, oy, ox, n):
    col = n if tm.nsplit == 0 else (n // tm.nsplit) * tm.nhi + (n % tm.nsplit)
    return (z1 * tm.zs1 + z0 * tm.zs0 + nb[:, None] * tm.bs + oy[:, None] * tm.ys + ox[:, None] * tm.xs + col[None, :]).astype(np.int64)


def run(d):
    """Executes the descriptor, writing the C operand in place (host memory)."""
    M = d.NB * d.Ho * d.Wo
    m = np.arange(M)
    nb, rem = m // (d.Ho * d.Wo), m % (d.Ho * d.Wo)
    oy, ox = rem // d.Wo, rem % d.Wo
    n = np.arange(d.N)
    scale = _view(d.scale, d.N).astype(np.float64) if d.scale else np.ones(d.N)
    bias = _view(d.bias, d.N).astype(np.float64) if d.bias else np.zeros(d.N)
    for z in range(d.Z):
        z1, z0 = z // d.zdiv, z % d.zdiv
        acc = np.zeros((M, d.N), dtype=np.float64)
        for t in range(d.ntaps):
            iy = oy * d.sy + d.tap_dy[t]
            ix = ox * d.sx + d.tap_dx[t]
            if d.pad_mode == 1:
                iy = np.where(iy < 0, -iy, np.where(iy >= d.Hi, 2 * d.Hi - 2 - iy, iy))
                ix = np.where(ix < 0, -ix, np.where(ix >= d.Wi, 2 * d.Wi - 2 - ix, ix))
                ok = np.ones(M, dtype=bool)
            else:
                ok = (iy >= 0) & (iy < d.Hi) & (ix >= 0) & (ix < d.Wi)
            base = z1 * d.a_zs1 + z0 * d.a_zs0 + nb * d.a_bs + iy * d.a_ys + ix * d.a_xs + d.tap_off[t]
            base = np.where(ok, base, 0).astype(np.int64)
            ci = np.arange(d.Cin)
            a_off = base[:, None] + ci[None, :]
            A = _view(d.a, a_off.max() + 1)[a_off].astype(np.float64) * ok[:, None]
            k = t * d.Cin + ci
            kk = np.where(k < d.Kw, k, 0)
            w_off = (z1 * d.w_zs1 + z0 * d.w_zs0 + kk[:, None] * d.ldw + np.where(n < d.Nw, n, 0)[None, :]).astype(np.int64)
            Wm = _view(d.w, w_off.max() + 1)[w_off].astype(np.float64)
            Wm = Wm * (k < d.Kw)[:, None] * (n < d.Nw)[None, :]
            acc += A @ Wm
        if d.pre.base:
            off = _map_offsets(d.pre, z1, z0, nb, oy, ox, n)
            acc += _view(d.pre.base, off.max() + 1)[off]
        v = acc * scale[None, :] + bias[None, :]
        post = None
        if d.post.base:
            off = _map_offsets(d.post, z1, z0, nb, oy, ox, n)
            post = _view(d.post.base, off.max() + 1)[off]
        if post is not None and (d.act & 0x100):
            v = v + post
        v = _act(v, d.act & 0xff, d.act_alpha)
        if post is not None and not (d.act & 0x100):
            v = v + post
        off = _map_offsets(d.c, z1, z0, nb, oy, ox, n)
        _view(d.c.base, off.max() + 1)[off] = v.astype(np.float32)
