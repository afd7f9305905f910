// pgemm_rows.h — the few-row planar GEMM of pgemm.hip (pgemm_rows_kernel) as the native decoder loop launches it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mit_hip.h"

// Decoder extras beside a MitPGemm (zero-initialised = none)
struct PgRowsExt {
    int nsplit;             // != 0 (multiple of 8): output column n lives at (n / nsplit) * nhi + n % nsplit (the q | k | v projection)
    int64_t nhi;
    const int *dyn;         // device step counter: the fp32 output starts *dyn * c_dyn floats further
    int64_t c_dyn;
    uint16_t *also_planes;  // planar copy of the result beside the fp32 output (N % 8 == 0)
    int64_t also_ld;
    int splitk;             // != 0 and K == 2048: four waves sum a quarter of K each (fixed order; not the k-sequential rounding)
};
// k steps a few-row GEMM wave requests ahead (4 / 6 / 8 / 10 measured within a few per cent of each other, in scripts/pgemm_check and in
// the decoder: a launch is bound by its accumulator chain and its start-up, not by round trips to the memory side)
constexpr int PG_ROWS_DEPTH = 6;
// C = epilogue(A @ W) for planar A, one wave per 32 x 32 block (Z == 1).  d.c and / or d.c_planes | x.also_planes; d.tile is ignored,
// d.nprod 6 | 9.  Returns 0, or 1 with mit_last_error() set.
int mit_pgemm_rows(const MitPGemm &d, const PgRowsExt &x, hipStream_t s);

// A = LayerNorm(x) instead of planes (pgemm_rows_ln.hip): x fp32 [M][ldx] (ldx % 4 == 0), K == 320; d.a_planes / d.lda are ignored.
// Bit for bit ocrk_layernorm(planes) followed by mit_pgemm_rows.
struct PgRowsLn {
    const float *x;
    int64_t ldx;
    const float *w, *b;   // LayerNorm weight / bias [K]
    float eps;
};
int mit_pgemm_rows_ln(const MitPGemm &d, const PgRowsExt &x, const PgRowsLn &ln, hipStream_t s);
