"""Host-side logic of the 48px OCR engine that needs no GPU."""
def test_fused_mlp_row_permutation_matches_the_accumulator_layout():
    """mit_convnext_mlp hands the first contraction's accumulator registers to the second contraction as A-operand fragments; the
    host permutes pwconv2's rows to the order in which those registers hold the hidden index (csrc/mlp_fused.hip).  Checked against an
    independent emulation of the 32x32 MFMA C layout: lane (li, lh), register r holds row (r & 3) + 8 (r >> 2) + 4 lh."""
    import torch

    from manga_image_translator_amd.ocr48 import fused_mlp_row_permutation

    for hidden in (320, 640):
        perm = fused_mlp_row_permutation(hidden)
        assert sorted(perm.tolist()) == list(range(hidden))                         # a permutation
        for hb in range(hidden // 32):
            for s in range(2):                                                      # MFMA step of the second contraction
                rows_of_step = set()
                for lh in range(2):
                    for j in range(8):                                              # the lane's 8 "consecutive k" of step s
                        r = 8 * s + j                                               # ... are its accumulator registers 8 s .. 8 s + 7
                        hidden_index = 32 * hb + (r & 3) + 8 * (r >> 2) + 4 * lh    # what register r of a lane with this lh holds
                        kprime = 32 * hb + 16 * s + 8 * lh + j                      # the B-operand row that meets it
                        assert int(perm[kprime]) == hidden_index
                        rows_of_step.add(hidden_index)
                # every step still contracts one aligned group of 16 hidden values (same sums as the two-launch form, other k slots)
                assert rows_of_step == set(range(32 * hb + 16 * s, 32 * hb + 16 * s + 16))
    with __import__("pytest").raises(ValueError):
        fused_mlp_row_permutation(100)
