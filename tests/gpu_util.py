"""Helpers shared by the GPU parity tests and tests/diag_gpu.py: build an engine on cuda:0 with the
portable synthetic weights, run the CPU oracle on the same inputs, compare outputs and stage taps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402
from oracle import dex_oracle as O  # noqa: E402

_ENG = {}
_LOG = os.path.join(ROOT, "gpurun_out")
# the CPU oracle's small-tensor torch ops run SLOWER on all 256 logical CPUs of the GPU box's host than on 16 threads (bench.py's
# cpu_baseline measured both: 16 wins); the whole-job tests (tests/test_gpu_full_jobs.py) run the oracle for minutes
torch.set_num_threads(min(16, torch.get_num_threads()))


def record(tag, **vals):
    """Measured errors go to gpurun_out/parity_measured.jsonl (scratch) so tolerances can be set from data."""
    import json
    try:
        os.makedirs(_LOG, exist_ok=True)
        with open(os.path.join(_LOG, "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps({"tag": tag, **{k: float(v) for k, v in vals.items()}}) + "\n")
    except OSError:
        pass


def fp32_call_ok(tag, got, ref):
    """fp32 mode, one EDMPrecond call: max|d| <= FP32_CALL_REL * max(1, |y|max) (tests/tolerances.py); records the measurement."""
    from tests.tolerances import FP32_CALL_REL
    e = np.abs(got - ref)
    record(f"{tag}:fp32:call", max=e.max(), mean=e.mean(), ref_absmax=np.abs(ref).max())
    assert np.isfinite(got).all() and e.max() <= FP32_CALL_REL * max(1.0, np.abs(ref).max()), (tag, float(e.max()), float(np.abs(ref).max()))


def fp32_sampler_ok(tag, got, ref, heun=False):
    """fp32 mode, a whole sampler call: max|d| <= FP32_SAMPLER_MAX and mean|d| <= FP32_SAMPLER_MEAN (Heun with few steps: the
    FP32_HEUN_* bounds, see tests/tolerances.py for why); records the measurement."""
    from tests import tolerances as TL
    mx, mn = (TL.FP32_HEUN_MAX, TL.FP32_HEUN_MEAN) if heun else (TL.FP32_SAMPLER_MAX, TL.FP32_SAMPLER_MEAN)
    e = np.abs(got - ref)
    record(f"{tag}:fp32:{'heun' if heun else 'sampler'}", max=e.max(), mean=e.mean(), ref_absmax=np.abs(ref).max())
    assert np.isfinite(got).all() and e.max() <= mx and e.mean() <= mn, (tag, float(e.max()), float(e.mean()))


def fp32_taps_ok(tag, terr):
    from tests.tolerances import FP32_TAP_REL
    for k, (err, mx) in terr.items():
        record(f"{tag}.{k}:fp32:tap", max=err, ref_absmax=mx)
        assert err <= FP32_TAP_REL * max(1.0, mx), (tag, k, err, mx)


_ORACLE = {}          # oracle outputs are mode-independent: computed once per (case, sigma / n) and reused across precisions


_case_key = synth.case_key


def engine_for(name):
    """One engine (HIP context + packed synthetic weights) per preset, cached for the session."""
    from dex_tts_amd.engine import ScoreNetEngine
    if name not in _ENG:
        cfg = C.PRESETS[name]()
        eng = ScoreNetEngine(cfg, torch.device("cuda", 0))
        w = synth.make_weights(C.param_shapes(cfg))
        eng.load_weights({k: torch.from_numpy(v) for k, v in w.items()})
        _ENG[name] = (cfg, eng, w)
    return _ENG[name]


make_case = synth.make_case


def oracle_kwargs(case, dtype=torch.float32):
    kw = {}
    if "ref" in case:
        kw["ref"] = [torch.from_numpy(r).to(dtype) for r in case["ref"]]
        kw["sty"] = torch.from_numpy(case["sty"]).to(dtype)
        kw["sty_lengths"] = torch.from_numpy(np.asarray(case["sty_lengths"]))
    if "spk" in case:
        kw["spk"] = torch.from_numpy(case["spk"]).to(dtype)
    return kw


def engine_kwargs(case):
    kw = {}
    if "ref" in case:
        kw["ref"] = [torch.from_numpy(r) for r in case["ref"]]
        kw["sty"] = torch.from_numpy(case["sty"])
        kw["sty_lengths"] = torch.from_numpy(np.asarray(case["sty_lengths"]))
    if "spk" in case:
        kw["spk"] = torch.from_numpy(case["spk"])
    return kw


def nhwc_rows(t: torch.Tensor) -> np.ndarray:
    if t.dim() == 4:
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).numpy()
    return t.reshape(-1, t.shape[-1]).numpy()


def run_precond(name, case, sigma, dtype=torch.float32, with_taps=True):
    """Returns (gpu_out, oracle_out, {tap: (max_abs_err, max_abs_ref)})."""
    cfg, eng, w = engine_for(name)
    W = O.as_torch(w, dtype)
    mu, mask, eps = (torch.from_numpy(case[k]) for k in ("mu", "mask", "eps"))
    x = mu + float(sigma) * eps
    got = eng.denoise_once(x, sigma, mask, mu, **engine_kwargs(case))
    gtaps = eng.taps() if with_taps else {}
    got = got.cpu().numpy()
    taps = {} if with_taps else None
    okey = (name, "precond", float(sigma), str(dtype), _case_key(case))
    if not with_taps and okey in _ORACLE:
        ref = _ORACLE[okey]
    else:
        ref = O.edm_precond(W, cfg, x.to(dtype), torch.tensor(float(sigma), dtype=dtype), mask.to(dtype), mu.to(dtype),
                            taps=taps, **oracle_kwargs(case, dtype)).to(torch.float32).numpy()
        _ORACLE[okey] = ref
    terr = {}
    if with_taps:
        for k, v in gtaps.items():
            if k in taps:
                r = nhwc_rows(taps[k].to(torch.float32))
                g = v.cpu().numpy()
                if r.shape != g.shape:
                    terr[k] = (float("inf"), float(np.abs(r).max()))
                else:
                    terr[k] = (float(np.abs(g - r).max()), float(np.abs(r).max()))
    return got, ref, terr


def run_sampler(name, case, n_steps, use_graph=False, solver="euler"):
    cfg, eng, w = engine_for(name)
    W = O.as_torch(w, torch.float32)
    mu, mask, z = (torch.from_numpy(case[k]) for k in ("mu", "mask", "z"))
    got = eng.sample(z, mask, mu, n_steps, use_graph=use_graph, solver=solver, **engine_kwargs(case)).cpu().numpy()
    okey = (name, "sampler", int(n_steps), solver, _case_key(case))
    if okey not in _ORACLE:
        _ORACLE[okey] = oracle_sampler_stored(name, case, n_steps, solver)
    return got, _ORACLE[okey]


ORACLE_JOBS = os.path.join(ROOT, "tests", "golden", "oracle_jobs")


def oracle_job_path(name, case, n_steps, solver="euler"):
    return synth.stored_job_path(ROOT, name, case, n_steps, solver)


def oracle_sampler_stored(name, case, n_steps, solver="euler"):
    """The CPU oracle's result of a whole sampler job.  The long batch jobs (B = 32, 50 / 100 Euler steps: 5 - 11 minutes of host time
    each) are committed under tests/golden/oracle_jobs/ - the oracle's OUTPUT on the portable synthetic inputs, keyed by a hash of the
    inputs, written by oracle/make_oracle_jobs.py on the CPU - so the GPU suite does not spend a quarter of an hour of box time on them;
    any job without a stored file is computed here."""
    path = oracle_job_path(name, case, n_steps, solver)
    if os.path.exists(path):
        return np.load(path)
    cfg = C.PRESETS[name]()
    w = synth.make_weights(C.param_shapes(cfg))
    W = O.as_torch(w, torch.float32)
    mu, mask, z = (torch.from_numpy(case[k]) for k in ("mu", "mask", "z"))
    return O.diffusion_infer(W, cfg, mask, mu, n_steps, z, solver=solver, **oracle_kwargs(case)).numpy()
