"""Parity tolerances in one place (tests/tolerances.py re-exports them; DESIGN.md §2 explains them).

fp32 mode (the reference's own arithmetic): the reference's fp32 round-off floor — fp64 vs fp32 evaluation of the same
net — is 2.1e-4 for one denoiser call at sigma = 80 and 4.7e-4 for a 4-step sampler (tests/test_oracle_golden.py).
Reduced-precision modes (bf16 / fp16 operands on the MFMA, fp32 accumulation, norms, softmax state and residual
streams) have no reference counterpart; their bounds are <= 2x the worst value measured on MI355X against the fp32
CPU oracle over every case of tests/test_gpu_parity.py and tests/test_gpu_baseline_shapes.py
(profiles/round3_parity_measured.jsonl holds the measurements; every GPU comparison appends to gpurun_out/parity_measured.jsonl)."""

# round 3: the fp32 bounds are <= 10x the worst value MEASURED on MI355X (the library repeats the oracle's fp32 operations in
# nearly the same order, so it sits 1-2 orders of magnitude inside the fp64-vs-fp32 floor quoted above): worst measured call
# 1.5e-5 (1.1e-5 relative to |y|max; B=8 T=512 sigma=80), worst sampler 5.3e-6 / 6.1e-7 (100-step DEX), taps <= 3e-5.
# (rounds 1-2 had 1e-3 / 2e-3 / 2e-4 here: a 100x regression of the parity mode would have passed.)
FP32_CALL_REL = 1e-4          # single EDMPrecond call: max|d| <= FP32_CALL_REL * max(1, |y|max)
FP32_SAMPLER_MAX = 5e-5       # sampler: max|d|
FP32_SAMPLER_MEAN = 1e-5      # sampler: mean|d|
# Heun (edm.py:207-214) with FEW steps is ill-conditioned in the reference itself: step i = n-2 evaluates the corrector at
# sigma' = 0.002 after a step h ~ sigma_(n-2) (0.47 at n = 4), and d' = (x' - D')/sigma' enters x_next times h/2 — one fp32 ulp of
# D' (the network's own summation-order noise, 5e-7) is amplified by h / (2 sigma') ~ 1e2.  Measured (round 3, fp32 mode vs oracle /
# reference goldens): n = 3..5: 1.1e-4 .. 7.0e-4 max, 1.0e-5 .. 9.4e-5 mean; n = 7: 3.2e-5 / 3.5e-6.  Bounds = 3x the worst.
FP32_HEUN_MAX = 2e-3
FP32_HEUN_MEAN = 3e-4
FP32_TAP_REL = 2e-4           # stage taps (down0 / down1 / dit_in / dit_out / up0 ...): max|d| <= FP32_TAP_REL * max(1, |tap|max)

# (max|d|, mean|d|) on mels of range about [-11.5, 4], RMS 2.7
# worst measured (round 2, 71 GPU tests): bf16 call 2.9e-2 / 3.5e-3 (strip-streaming conv forced onto a 3-utterance ragged batch,
# sigma = 80), bf16 sampler 3.2e-2 / 4.0e-3 (6-step Heun, B=2); the BASELINE shapes sit at 1.5e-2..2.4e-2 / 2.5e-3..3.0e-3 per
# call and 6.9e-3 / 1.5e-3 for the 50-step sampler at T=512.  fp16: call 3.1e-3 / 3.8e-4, sampler 9.5e-4 / 1.6e-4.
LOWP = {
    "bf16": {"call": (5e-2, 6.5e-3), "sampler": (5e-2, 7.5e-3)},
    # (round 3: the 3-utterance ragged batch with a 77-frame utterance of tests/test_gpu_cluster.py reaches 4.6e-3 max in BOTH forms of the DiT block)
    "fp16": {"call": (6e-3, 7.5e-4), "sampler": (7e-3, 6e-4)},
    # fp16x2 (fp16 operands, weights as hi + lo): what is left is the rounding of the activations (small fuzz shapes, few steps)
    "fp16x2": {"call": (4e-3, 5e-4), "sampler": (4e-3, 4e-4)},
    # few-step Heun amplifies the last corrector's rounding by h / 2 sigma' ~ 1e2 (see FP32_HEUN_* above): measured 3.3e-3 / 4.5e-4 (DEX, n = 7)
    "fp16x2_heun": {"sampler": (6.5e-3, 9e-4)},
}


# Round 4 (VERDICT r3 Weak #7): at the BASELINE shapes the reduced-precision modes are held to bounds <= 2x what was MEASURED at
# that shape in that mode (profiles/round3_parity_measured.jsonl / round4_parity_measured.jsonl) - under the global pair above a 7-8x
# accuracy regression at the benchmarked shape (bf16 7.2e-3 measured against 5e-2 allowed) would have passed.  Keyed by the test's tag
# prefix; the global pair stays for the small fuzz shapes.  (max|d|, mean|d|) against the fp32 CPU oracle.
LOWP_AT = {
    # configs[1]: GeDEX-LJ B=1 T=512 - the 50-step job the bench times, and single calls at sigma = 80 / 1 / 0.002
    ("cfg1_T512_n50", "bf16", "sampler"): (1.5e-2, 3.0e-3),      # measured 7.24e-3 / 1.51e-3
    ("cfg1_T512_n50", "fp16", "sampler"): (1.8e-3, 3.3e-4),      # measured 8.66e-4 / 1.64e-4
    ("cfg1_T512_sigma", "bf16", "call"): (2.7e-2, 5.1e-3),       # worst of the three sigmas: 1.34e-2 / 2.53e-3
    ("cfg1_T512_sigma", "fp16", "call"): (3.6e-3, 6.2e-4),       # 1.80e-3 / 3.07e-4
    # the split-weight mode at the benchmarked job: the fast mode inside the fp32-grade sampler bound the round-3 verdict set
    # (max <= 1e-3 AND mean <= 1e-4 against the oracle); measured 3.9e-4 / 6.9e-5
    ("cfg1_T512_n50", "fp16x2", "sampler"): (8.0e-4, 1.0e-4),
    ("cfg1_T512_sigma", "fp16x2", "call"): (2.6e-3, 4.2e-4),     # single calls: the weights' share of a call's rounding is small; set from the first run
    # configs[2]: DEX-VCTK B=32 T=256 Tr=Ts=348
    ("cfg2_dex_b32_n4", "bf16", "sampler"): (4.3e-2, 5.4e-3),    # 2.12e-2 / 2.68e-3
    ("cfg2_dex_b32_n4", "fp16", "sampler"): (6.0e-3, 6.8e-4),    # 2.96e-3 / 3.39e-4
    ("cfg2_dex_b32_sigma", "bf16", "call"): (5.0e-2, 6.1e-3),    # 2.55e-2 / 3.04e-3
    ("cfg2_dex_b32_sigma", "fp16", "call"): (6.0e-3, 7.5e-4),    # 3.23e-3 / 3.79e-4
    # configs[3]: the 100-step DEX-ESD job
    ("cfg3_n100", "bf16", "sampler"): (2.0e-2, 3.3e-3),          # 9.76e-3 / 1.63e-3
    # configs[4]: T=4000 (N=5010 tokens)
    ("cfg4_T4000_n4", "bf16", "sampler"): (2.3e-2, 3.9e-3),      # 1.13e-2 / 1.95e-3
    ("cfg4_T4000_n4", "fp16", "sampler"): (2.9e-3, 4.7e-4),      # 1.45e-3 / 2.32e-4
    ("cfg4_T4000_sigma", "bf16", "call"): (3.3e-2, 5.2e-3),      # 1.63e-2 / 2.56e-3
    ("cfg4_T4000_sigma", "fp16", "call"): (3.9e-3, 6.2e-4),      # 1.91e-3 / 3.08e-4
    # ---- round 5 (VERDICT r4 Missing #4): the WHOLE jobs bench.py times at batch, against the oracle (tests/test_gpu_full_jobs.py;
    # measurements: profiles/round5_parity_measured.jsonl), and the split-weight mode's rows at the batch / long-form shapes
    # configs[2]: DEX-VCTK B=32 T=256, 50 Euler steps
    ("cfg2_dex_b32_n50", "bf16", "sampler"): (2.4e-2, 3.4e-3),   # measured 1.22e-2 / 1.70e-3 (round 5), 1.34e-2 / 1.70e-3 (round 6)
    ("cfg2_dex_b32_n50", "fp16", "sampler"): (3.0e-3, 3.9e-4),   # 1.54e-3 / 1.96e-4
    ("cfg2_dex_b32_n50", "fp16x2", "sampler"): (1.8e-3, 2.2e-4), # 9.23e-4 / 1.10e-4  (the mean sits just OUTSIDE the 1e-4 the mode holds at configs[1])
    # configs[3], the per-GPU share: DEX-ESD B=32 T=256, 100 Euler steps
    ("cfg3_dex_esd_b32_n100", "bf16", "sampler"): (2.4e-2, 3.2e-3),    # 1.27e-2 / 1.62e-3 (rounds 5 and 6: profiles/round*_parity_measured.jsonl; ADVICE r5: the comment said 1.20e-2)
    ("cfg3_dex_esd_b32_n100", "fp16x2", "sampler"): (1.3e-3, 1.6e-4),  # 6.41e-4 / 8.05e-5
    # configs[4], the WHOLE job (round 6, VERDICT r5 Missing #4): GeDEX long form B=1 T=4000, 50 Euler steps, whole-call graph
    ("cfg4_T4000_n50", "fp16", "sampler"): (1.9e-3, 3.4e-4),            # measured 9.24e-4 / 1.67e-4
    ("cfg4_T4000_n50", "fp16x2", "sampler"): (8.8e-4, 1.4e-4),          # 4.38e-4 / 7.03e-5 (inside the 1e-3 / 1e-4 of the fp32-grade bound)
    # DEX B=32 T=512 (N = 2580 tokens), one call at sigma = 80
    ("c3_dex_b32_T512_sigma", "bf16", "call"): (5.0e-2, 6.0e-3),       # 2.73e-2 / 3.00e-3
    ("c3_dex_b32_T512_sigma", "fp16", "call"): (7.0e-3, 7.5e-4),       # 3.68e-3 / 3.74e-4
    # fp16x2 at the batch / long-form shapes of tests/test_gpu_fp16x2.py (tags x2_*): 4 steps of configs[2], 10 of GeDEX B=8, 3 at T=4000
    ("x2_batch_dex_vctk_B32_T256_n4", "fp16x2", "sampler"): (4.0e-3, 4.0e-4),   # 2.28e-3 / 3.32e-4 (few steps from sigma = 80: the large-sigma calls dominate)
    ("x2_batch_gedex_lj_B8_T512_n10", "fp16x2", "sampler"): (2.0e-3, 3.0e-4),   # 9.83e-4 / 1.47e-4
    ("x2_gedex_lj_B1_T4000_n3", "fp16x2", "sampler"): (2.7e-3, 4.0e-4),         # 1.35e-3 / 2.26e-4
}


def lowp_bounds(tag: str, prec: str, kind: str):
    """(max, mean) bound of a reduced-precision comparison: the per-shape entry whose prefix the tag starts with, else the global pair."""
    for (pfx, p, k), b in LOWP_AT.items():
        if p == prec and k == kind and tag.startswith(pfx):
            return b
    return LOWP[prec][kind]
