"""Parity tolerances in one place (the tests state them; DESIGN.md §2 explains them).

fp32 mode (the reference's own arithmetic): the reference's fp32 round-off floor — fp64 vs fp32 evaluation of the same
net — is 2.1e-4 for one denoiser call at sigma = 80 and 4.7e-4 for a 4-step sampler (tests/test_oracle_golden.py).
Reduced-precision modes (bf16 / fp16 operands on the MFMA, fp32 accumulation, norms, softmax state and residual
streams) have no reference counterpart; their bounds are <= 2x the worst value measured on MI355X against the fp32
CPU oracle over every case of tests/test_gpu_parity.py and tests/test_gpu_baseline_shapes.py
(profiles/round2_parity_measured.jsonl holds the measurements)."""

FP32_CALL_REL = 1e-3          # single EDMPrecond call: max|d| <= FP32_CALL_REL * max(1, |y|max)
FP32_SAMPLER_MAX = 2e-3       # sampler: max|d|
FP32_SAMPLER_MEAN = 2e-4      # sampler: mean|d|

# (max|d|, mean|d|) on mels of range about [-11.5, 4], RMS 2.7
# worst measured (round 2, 71 GPU tests): bf16 call 2.9e-2 / 3.5e-3 (strip-streaming conv forced onto a 3-utterance ragged batch,
# sigma = 80), bf16 sampler 3.2e-2 / 4.0e-3 (6-step Heun, B=2); the BASELINE shapes sit at 1.5e-2..2.4e-2 / 2.5e-3..3.0e-3 per
# call and 6.9e-3 / 1.5e-3 for the 50-step sampler at T=512.  fp16: call 3.1e-3 / 3.8e-4, sampler 9.5e-4 / 1.6e-4.
LOWP = {
    "bf16": {"call": (5e-2, 6.5e-3), "sampler": (5e-2, 7.5e-3)},
    "fp16": {"call": (6e-3, 7.5e-4), "sampler": (4e-3, 6e-4)},
}
