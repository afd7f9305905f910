#!/bin/bash
# Builds scripts/cotenant_check (see cotenant_check.cpp) against the in-tree libmit_hip.so.
set -e
cd "$(dirname "$0")/.."
python -m manga_image_translator_amd.build >/dev/null
# the FFT rows device code (the inside of fft_rows.hip's anonymous namespace up to and including rfft_rows_kernel), with two hooks, so that
# cotenant_check can instantiate VARIANTS of the victim kernel in its own namespaces — generated, not committed
python - <<'PY'
src = open("manga_image_translator_amd/csrc/fft_rows.hip").read().split("\n")
a = next(i for i, l in enumerate(src) if l.startswith("namespace {")) + 1
b = next(i for i, l in enumerate(src) if "void irfft_rows_kernel" in l)
while not src[b - 1].startswith("}"):   # back to the end of rfft_rows_kernel (skip the comment above the inverse kernel)
    b -= 1
core = "\n".join(src[a:b])
core = core.replace("run_stages<false, SAFE>(bufA, bufB, tw, plan, N, slot, c)", "MIT_RUN_STAGES(bufA, bufB, tw, plan, N, slot, c)")
# the kernels are templates on SAFE (narrow LDS reads) in the library; here they are plain functions and SAFE is a namespace constant
core = core.replace("template <bool SAFE>\n__global__", "__global__")
core = "constexpr bool SAFE = false;\n" + core
fixed = """template <bool INV>
__device__ __forceinline__ float2 *run_fixed12(float2 *a, float2 *b, const float2 *tw, int N, int slot, int c) {
    stage<4, INV, false>(a, b, tw, N, 12, 1, slot, c);   // the plan of N = 12 (w = 24), spelled out: no switch, no plan loads in the loop
    __syncthreads();
    stage<3, INV, false>(b, a, tw, N, 3, 4, slot, c);
    __syncthreads();
    return a;
}

"""
core = core.replace("constexpr int LD_UNROLL", fixed + "constexpr int LD_UNROLL", 1)
# two more hooks inside stage<P>(): how a butterfly input is read from LDS, where a twiddle comes from
n1 = core.count("x[i] = lds_pair<SAFE>(src + (q + s * (p + m * i)) * CC + c);")
core = core.replace("x[i] = lds_pair<SAFE>(src + (q + s * (p + m * i)) * CC + c);", "x[i] = MIT_LDS_READ(src, (q + s * (p + m * i)) * CC + c);")
n2 = core.count("float2 w = tw[p * k * tws];")
core = core.replace("float2 w = tw[p * k * tws];", "float2 w = MIT_TW(tw, p * k * tws, N);")
assert n1 == 1 and n2 == 1, (n1, n2)
# third hook: where the workgroup's coordinates come from (a 3-D grid, or one flat index decoded in the kernel)
for ax in "xyz":
    core = core.replace("blockIdx." + ax, "MIT_BID_" + ax.upper())
assert "MIT_RUN_STAGES" in core and "run_fixed12" in core
open("scripts/_fft_rows_core.inc", "w").write(core + "\n")
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc scripts/cotenant_check.cpp -o scripts/cotenant_check -Iinclude -Lmanga_image_translator_amd -lmit_hip \
      -Wl,-rpath,'$ORIGIN/../manga_image_translator_amd'
echo built scripts/cotenant_check
