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
LOWP = {
    "bf16": {"call": (5e-2, 8e-3), "sampler": (5e-2, 8e-3)},
    "fp16": {"call": (1e-2, 1.5e-3), "sampler": (1e-2, 1.5e-3)},
}
