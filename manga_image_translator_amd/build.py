"""Builds the gfx950 C-ABI library (``libmit_hip.so``) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build
container; the resulting ``.so`` is git-ignored but travels with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libmit_hip.so"
_STAMP = PKG_DIR / "csrc" / ".build_stamp"

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-gpu-rdc",
    "-ffp-contract=off",  # keep a*b+c unfused outside MFMA: epilogues match torch's two-rounding order
    # No SLP vectoriser: on gfx950 it turns pairs of scalar fp32 operations into v_pk_mul_f32 / v_pk_add_f32 with op_sel / neg modifiers,
    # and those were measured to return a wrong 16-lane pass now and then WHILE ANOTHER KERNEL'S MFMA WAVES SHARE THE CU (a second
    # stream or process): xpos_rotate_kernel lost the second product of `x.x * c + (-x.y) * s` in 16 consecutive lanes, the FFT rows and
    # DenseCRF kernels were hit the same way (profiles/r05q_noslp.log).  Without the pass every multi-stream /
    # multi-process result is identical to the one-stream result; explicit ext_vector arithmetic (plain v_pk_* without modifiers) stays.
    "-fno-slp-vectorize",
    "-Wall",
    "-Wno-unused-function",
]


# Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) is switched off in the code generator for every translation unit that does
# not ask for it: -fno-slp-vectorize alone still left the loop vectoriser and the vector combiner forming such instructions WITH op_sel /
# neg modifiers from scalar source (winograd.hip: 30, map_to_u8_kernel: 2, …) — the very form that misbehaved beside MFMA co-tenants.
# The units below use packed math on purpose (explicit ext_vector arithmetic whose operands are whole register pairs: the channel-pair
# FMAs of the 7x7 64->3 convolution, the float4 epilogues of the conv_gemm tiles — the generic tile runs 2.1x slower without them
# (profiles/r07h) —, the planar GEMM) and are checked by tests/test_build_flags.py to contain no packed-fp32 instruction with a
# modifier; every other unit must contain none at all.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PACKED_FP32_BY_DESIGN = {"conv_small_cout.hip", "pgemm.hip"} | {f"conv_gemm_inst{i}.hip" for i in range(8)}


# Kernel-argument preloading (gfx940+): the leading scalar arguments of a kernel arrive in user SGPRs with the wave instead of behind a scalar
# load from the kernarg segment.  Used where a launch is a few microseconds long and begins with that round trip: the few-row GEMMs and
# the attention / bookkeeping kernels of the B = 1 decoder (their signatures lead with the values the first requests need).  The code object keeps the
# load-based prologue for firmware without the feature.
KERNARG_PRELOAD = {"pgemm.hip": 12, "pgemm_rows_ln.hip": 12, "ocr_kernels.hip": 12}


def flags_for(src_name: str) -> list:
    """Compiler flags of one translation unit."""
    f = HIPCC_FLAGS + ([] if src_name in PACKED_FP32_BY_DESIGN else NO_PACKED_FP32)
    if src_name in KERNARG_PRELOAD:
        f = f + ["-mllvm", f"-amdgpu-kernarg-preload-count={KERNARG_PRELOAD[src_name]}"]
    return f


if os.environ.get("MIT_WITH_SLP"):  # A/B only: the build of rounds 1-4 (reproduces the co-tenancy failures)
    HIPCC_FLAGS.remove("-fno-slp-vectorize")
if os.environ.get("MIT_CONV_EXPERIMENTS"):  # rejected scheduling variants + timing ablations of the conv kernel (scripts/bench_conv.py)
    HIPCC_FLAGS.append("-DMIT_CONV_EXPERIMENTS")


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 kernels")
    return exe


def _sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def source_digest() -> str:
    """sha256 over every source the library is built from + the compiler flags (what mit_source_digest() must return)."""
    return _digest()


def _digest() -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [PKG_DIR.parent / "include" / "mit_hip.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(HIPCC_FLAGS + NO_PACKED_FP32 + sorted(PACKED_FP32_BY_DESIGN)).encode())
    return h.hexdigest()


def _obj_digest(src: Path) -> str:
    """Digest of one translation unit: its source, every header it can include, the flags."""
    h = hashlib.sha256()
    for p in [src] + sorted(list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc"))) + [PKG_DIR.parent / "include" / "mit_hip.h"]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(flags_for(src.name)).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every ``csrc/*.hip`` for gfx950 and link ``libmit_hip.so``.  Incremental per translation unit (an object is
    reused while its source, the headers and the flags are unchanged); the library as a whole carries the digest of all
    sources (``mit_source_digest()``), which is what ``lib.load()`` checks."""
    digest = _digest()
    if not force and LIB_PATH.exists() and _STAMP.exists() and _STAMP.read_text().strip() == digest:
        return LIB_PATH
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    procs = []
    objs = []
    for src in _sources():
        obj = objdir / (src.stem + ".o")
        stamp = objdir / (src.stem + ".digest")
        objs.append(str(obj))
        od = _obj_digest(src) + (digest if src.name == "capi.hip" else "")
        if obj.exists() and stamp.exists() and stamp.read_text() == od:
            continue
        cmd = [hipcc, *flags_for(src.name), "-c", str(src), "-o", str(obj)]
        if src.name == "capi.hip":  # mit_source_digest(): lets lib.load() detect a stale binary
            cmd.insert(1, f'-DMIT_SOURCE_DIGEST="{digest}"')
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, stamp, od, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, stamp, od, proc in procs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed for {src.name} ---\n{out}\n")
        else:
            stamp.write_text(od)
            # (the host half of each compile says "'-packed-fp32-ops' is not a recognized feature for this target": expected, the feature
            # belongs to the device half)
            noise = "is not a recognized feature for this target"
            rest = "\n".join(l for l in out.splitlines() if noise not in l)
            if verbose and rest.strip():
                sys.stderr.write(rest + "\n")
    if failed:
        raise RuntimeError("hipcc compilation failed")
    tmp = LIB_PATH.with_name(f"{LIB_PATH.name}.tmp{os.getpid()}")  # a concurrent dlopen never sees a half-written library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *objs]
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB_PATH)
    finally:
        if tmp.exists():
            tmp.unlink()
    _STAMP.write_text(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
