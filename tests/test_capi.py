"""The C-ABI library: loads, exports every function include/mit_hip.h declares, and the ctypes mirror of every struct
has the C compiler's size and field offsets (gcc compiles a probe against the real header).  No compute calls: those
are the ``-m gpu`` tests."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mit_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mit_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from manga_image_translator_amd import lib

    handle = lib.load(build_if_missing=True)
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), f"{n} declared in mit_hip.h but not exported by libmit_hip.so"
        assert n in lib.SYMBOLS, f"{n} has no ctypes prototype in lib.SYMBOLS"
    assert set(lib.SYMBOLS) <= set(names), sorted(set(lib.SYMBOLS) - set(names))
    assert handle.mit_abi_version() == lib.MIT_ABI_VERSION
    assert handle.mit_conv_gemm_config_name(0) == b"128x128x16" and handle.mit_conv_gemm_config_name(99) is None


def test_error_channel_without_gpu():
    """Argument validation happens before any HIP call, so it is observable on a CPU-only box."""
    from manga_image_translator_amd import lib

    handle = lib.load()
    assert handle.mit_conv_gemm(None, None) != 0
    assert b"null descriptor" in handle.mit_last_error()
    d = lib.MitConvGemm()
    assert handle.mit_conv_gemm(C.byref(d), None) != 0
    assert b"null operand" in handle.mit_last_error()
    with pytest.raises(RuntimeError, match="null operand"):
        lib.check(1, "probe")
    assert handle.mit_ocr_warp_lines(None, 1, 1, None, 0, None, 48, 8, None) != 0


def test_pgemm_argument_checks_without_gpu():
    """mit_pgemm validates its descriptor before any HIP call (sizes, alignment, output kind, activation)."""
    from manga_image_translator_amd import lib

    handle = lib.load()
    assert handle.mit_pgemm(None, None) != 0 and b"null descriptor" in handle.mit_last_error()
    d = lib.MitPGemm()
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"null operand" in handle.mit_last_error()
    d.a_planes, d.w_planes = 4096, 8192     # never dereferenced: every case below is refused first
    d.M, d.N, d.K, d.Z, d.lda, d.ldw = 128, 64, 24, 1, 128, 64
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"K % 16" in handle.mit_last_error()
    d.K = 32
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"exactly one of c / c_planes" in handle.mit_last_error()
    d.c, d.ldc = 16384, 62
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"ldc % 4" in handle.mit_last_error()
    d.ldc, d.act = 64, lib.ACT_SILU
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"none / relu / gelu" in handle.mit_last_error()
    d.act, d.lda = lib.ACT_RELU, 100
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"lda / ldw" in handle.mit_last_error()
    d.lda, d.c, d.c_planes, d.ld_cp, d.N = 128, None, 16384, 128, 60
    assert handle.mit_pgemm(C.byref(d), None) != 0 and b"N % 8" in handle.mit_last_error()
    assert handle.mit_pgemm_supported(C.byref(d)) == 0
    d.N = 64
    assert handle.mit_pgemm_supported(C.byref(d)) == 1
    names = []
    while handle.mit_pgemm_tile_name(len(names)) is not None:
        names.append(handle.mit_pgemm_tile_name(len(names)).decode())
    assert names[:4] == ["pg128x128s3p6", "pg128x64s3p6", "pg128x128s3p6P", "pg128x64s3p6P"] and len(set(names)) == len(names)
    assert handle.mit_split_planes(None, 0, 1, 8, None, 1, None) != 0 and b"null pointer" in handle.mit_last_error()
    assert handle.mit_split_planes(4096, 12, 4, 12, 8192, 4, None) != 0 and b"K % 8" in handle.mit_last_error()


def test_ctypes_structs_match_c_layout(tmp_path):
    from manga_image_translator_amd import lib

    structs = {"MitTensorMap": lib.MitTensorMap, "MitConvGemm": lib.MitConvGemm, "MitXposTables": lib.MitXposTables,
               "MitLinear": lib.MitLinear, "MitOcrDecoderLayer": lib.MitOcrDecoderLayer, "MitOcr48Decoder": lib.MitOcr48Decoder,
               "MitOcr48DecodeArgs": lib.MitOcr48DecodeArgs, "MitProfStat": lib.MitProfStat, "MitWarpLine": lib.MitWarpLine,
               "MitDilateJob": lib.MitDilateJob, "MitMaskRun": lib.MitMaskRun, "MitPGemm": lib.MitPGemm}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for name, st in structs.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for fname, *_ in st._fields_:
            lines.append(f'printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, st in structs.items():
        assert int(out[name]) == C.sizeof(st), name
        for fname, *_ in st._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(st, fname).offset, f"{name}.{fname}"


def test_stale_library_is_refused(tmp_path, monkeypatch):
    """lib.load() compares the digest embedded in the binary with the digest of the sources in the tree: a library built
    from other sources is an error when rebuilding is not allowed (never validated or measured silently)."""
    from manga_image_translator_amd import build, lib

    handle = lib.load(build_if_missing=True)
    assert handle.mit_source_digest().decode() == build.source_digest()
    assert not lib._stale(lib.lib_path(), build.source_digest())
    assert lib._stale(lib.lib_path(), "0" * 64) and lib._stale(tmp_path / "missing.so", build.source_digest())
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(build, "source_digest", lambda: "0" * 64)
    with pytest.raises(RuntimeError, match="different sources"):
        lib.load(build_if_missing=False)


def test_workspace_is_bounded_by_the_largest_request():
    """Engine workspaces are grow-only slabs keyed by name: heterogeneous page sizes re-view one slab instead of allocating a
    buffer set per shape."""
    import torch

    from manga_image_translator_amd import ops

    ws = ops.Workspace("cpu")
    a = ws.buf("x", 2, 8, 8, 4)
    assert a.shape == (2, 8, 8, 4) and a.is_contiguous()
    big = ws.buf("x", 1, 32, 16, 4)
    n = ws.nbytes()
    for shape in [(2, 8, 8, 4), (1, 16, 16, 4), (1, 31, 16, 4), (1, 32, 16, 4)]:
        t = ws.buf("x", *shape)
        assert t.data_ptr() == big.data_ptr() and tuple(t.shape) == shape
    assert ws.nbytes() == n == 1 * 32 * 16 * 4 * 4
    assert ws.buf("x", 4, dtype=torch.uint8).data_ptr() != big.data_ptr()  # other dtype, other slab
    ws.release()
    assert ws.nbytes() == 0
    sc = ops.ShapeCache(2)
    made = []
    for k in [(1, 1), (2, 2), (1, 1), (3, 3), (2, 2)]:
        sc.get(k, lambda k=k: made.append(k) or k)
    assert made == [(1, 1), (2, 2), (3, 3), (2, 2)] and len(sc) == 2
