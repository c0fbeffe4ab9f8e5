"""the training-loss branch ``Diffusion.forward(..., infer=False)`` = ``EDMLoss.forward`` (edm.py:22-68).
Goldens (tests/golden/edm_loss.npz) come from the reference's own EDMLoss on fixed draws (oracle/make_golden_loss.py)."""
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import config as C, edm, synth
from oracle import dex_oracle as O
from oracle.make_golden_loss import CASES, LOSS_TYPES, case_inputs

GOLD = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "edm_loss.npz")))
t = torch.from_numpy


def _okw(kw):
    return {k: ([t(r) for r in v] if k == "ref" else t(np.asarray(v))) for k, v in kw.items() if k != "ref_lengths"}


@pytest.mark.parametrize("name,B,T,lengths,dex_dims,seed", CASES)
def test_oracle_loss_matches_reference_golden(name, B, T, lengths, dex_dims, seed):
    cfg, mu, mask, x0, kw = case_inputs(name, B, T, lengths, dex_dims)
    W = O.as_torch(synth.make_weights(C.param_shapes(cfg), seed=0))
    rnd, eps = t(GOLD[f"{name}_rnd_normal"]), t(GOLD[f"{name}_eps"])
    with torch.no_grad():
        for lt in LOSS_TYPES[:3] if B > 2 else LOSS_TYPES:          # the 3-utterance case: three types are enough CPU time
            got = float(O.edm_loss(W, cfg, t(x0), t(mask), t(mu), rnd, eps, loss_type=lt, **_okw(kw)))
            want = float(GOLD[f"{name}_{lt}"])
            assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (lt, got, want)      # batched vs per-utterance evaluation: summation order only


@pytest.mark.parametrize("lt", LOSS_TYPES)
def test_host_weight_equals_oracle_weight(lt):
    sigma = torch.exp(torch.linspace(-4.5, 3.5, 41)).reshape(-1, 1, 1)
    assert torch.equal(edm.loss_weight(sigma, lt), O.edm_loss_weight(sigma, lt))


def test_unknown_loss_type_and_grad_refused():
    with pytest.raises(ValueError):
        edm.loss_weight(torch.ones(1, 1, 1), "bogus")
    with pytest.raises(TypeError):
        edm.EDMLoss()(object(), torch.zeros(1, 80, 4), torch.ones(1, 1, 4), torch.zeros(1, 80, 4))


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,T,lengths,dex_dims,seed", CASES)
def test_gpu_loss_matches_reference_golden(name, B, T, lengths, dex_dims, seed):
    from dex_tts_amd.diffusion import from_config
    cfg, mu, mask, x0, kw = case_inputs(name, B, T, lengths, dex_dims)
    m = from_config(cfg)
    w = synth.make_weights(C.param_shapes(cfg), seed=0)
    sd = {}
    for k, v in w.items():
        sd[f"denoise_fn.{k}"] = t(v); sd[f"precond_model.model.{k}"] = t(v)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    c = lambda a: t(np.asarray(a)).cuda()
    rnd, eps = c(GOLD[f"{name}_rnd_normal"]), c(GOLD[f"{name}_eps"])
    dex = ([c(r) for r in kw["ref"]], c(kw["ref_lengths"]), c(kw["sty"]), c(kw["sty_lengths"])) if cfg.variant == "dex" else ()
    spk = c(kw["spk"]) if "spk" in kw else None
    for lt in LOSS_TYPES:
        m.loss_fn.loss_type = lt
        got = float(m.loss_fn(m.precond_model, c(x0), c(mask), c(mu), *dex, spk=spk, rnd_normal=rnd, eps=eps))
        want = float(GOLD[f"{name}_{lt}"])
        assert abs(got - want) <= 2e-5 * max(1.0, abs(want)), (lt, got, want)


@pytest.mark.gpu
def test_gpu_forward_infer_false_draws_like_the_reference():
    """``Diffusion.forward(x, mask, mu, infer=False)``: the module makes the reference's two draws from the device generator
    (randn([B,1,1]) then randn_like(x0)); replaying the same draws through the oracle gives the same loss."""
    from dex_tts_amd.diffusion import from_config
    name, B, T, lengths, dex_dims, seed = CASES[0]
    cfg, mu, mask, x0, kw = case_inputs(name, B, T, lengths, dex_dims)
    m = from_config(cfg)
    w = synth.make_weights(C.param_shapes(cfg), seed=0)
    sd = {}
    for k, v in w.items():
        sd[f"denoise_fn.{k}"] = t(v); sd[f"precond_model.model.{k}"] = t(v)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    torch.manual_seed(5)
    loss = float(m(t(x0).cuda(), t(mask).cuda(), t(mu).cuda(), infer=False))
    torch.manual_seed(5)
    rnd = torch.randn([B, 1, 1], device="cuda"); eps = torch.randn_like(t(x0).cuda())
    with torch.no_grad():
        ref = float(O.edm_loss(O.as_torch(w), cfg, t(x0), t(mask), t(mu), rnd.cpu(), eps.cpu()))
    assert abs(loss - ref) <= 2e-5 * max(1.0, abs(ref)), (loss, ref)


def test_diffusion_module_deepcopies_and_pickles_without_sharing_its_owner():
    """the loss hook used to be a lambda closing over the ORIGINAL module (a deep copy evaluated with the original's weights,
    torch.save failed).  It is resolved through a weakref to the owner at call time now."""
    import copy, io, pickle
    import torch
    from dex_tts_amd import config as C
    from dex_tts_amd.diffusion import from_config
    m = from_config(C.gedex_lj())
    c = copy.deepcopy(m)
    assert c.precond_model._owner() is c and m.precond_model._owner() is m
    assert c.precond_model.model is c.denoise_fn and c.denoise_fn is not m.denoise_fn
    k = next(iter(m.state_dict()))
    assert torch.equal(c.state_dict()[k], m.state_dict()[k])
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert m.precond_model._owner() is m                        # saving did not touch the live module
    r._bind_owner()                                              # (what forward(infer=False) does before it evaluates the loss)
    assert r.precond_model._owner() is r
    assert callable(r.sampler) and set(r.state_dict()) == set(m.state_dict())
    p = pickle.loads(pickle.dumps(m))
    p._bind_owner()
    assert p.precond_model._owner() is p and p.precond_model.model is p.denoise_fn
