"""Text encoder + durations + alignment.  CPU: the oracle (oracle/text_oracle.py) and the host mirror's
checkpoint surface against fixtures from the real reference TextEncoder / generate_path (tests/golden/text_*.npz,
manifest_text_*.json, written by oracle/make_golden_text.py).  GPU (-m gpu): dex_text_encode / dex_text_align through the C ABI
against the goldens and the oracle.  fp32 tolerance: exact-fp32 MFMA contractions in another summation order than oneDNN's
through 8 pre-norm layers: max|d| <= 3e-4 * max(1, |ref|max) on mu and logw (measured ~2e-5); durations, lengths and the path
must agree exactly (a log-duration within 1e-4 of an integer boundary could flip a ceil: none in these cases)."""
import json
import os

import numpy as np
import pytest
import torch

from dex_tts_amd import synth, text as T
from oracle import text_oracle as TO

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["gedex_lj", "gedex_vctk", "dex_vctk"]
N_VOCAB = 149


def manifest(name):
    return json.load(open(os.path.join(GOLD, f"manifest_text_{name}.json")))


def ctor_kwargs(name):
    kw = dict(manifest(name)["config"])
    kw["variant"] = "dex" if name.startswith("dex") else "gedex"
    return kw


def weights(name):
    g = np.load(os.path.join(GOLD, f"text_{name}.npz"))
    w = synth.make_text_weights(manifest(name)["keys"])
    w["encoder.retnet_rel_pos.angle"] = g["angle"]; w["encoder.retnet_rel_pos.decay"] = g["decay"]      # the reference's registered buffers
    return w


def inputs(name, B=2, L=23, lengths=(23, 14)):
    tok, lengths = synth.make_text_inputs(B, L, list(lengths), N_VOCAB)
    m = manifest(name)["config"]
    spk = synth.normalish("text_spk", (B, m["spk_emb_dim"]), 5) if m["n_spks"] > 1 else None
    sty = synth.normalish("text_sty", (B, m["n_channels"]), 6) * np.float32(0.5) if name.startswith("dex") else None
    return tok, lengths, spk, sty


def oracle_cfg(name):
    m = manifest(name)["config"]
    return dict(n_channels=m["n_channels"], n_layers=m["n_layers"], n_heads=m["n_heads"], n_spks=m["n_spks"], kernel_size=m["kernel_size"])


@pytest.mark.parametrize("name", CASES)
def test_param_shapes_match_reference_state_dict(name):
    kw = ctor_kwargs(name)
    got = T.param_shapes(kw["n_vocab"], kw["n_feats"], kw["n_channels"], kw["filter_channels"], kw["filter_channels_dp"], kw["n_heads"],
                         kw["n_layers"], kw["kernel_size"], kw["spk_emb_dim"], kw["n_spks"], kw["variant"])
    assert {k: tuple(v) for k, v in manifest(name)["keys"].items()} == got


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    g = dict(np.load(os.path.join(GOLD, f"text_{name}.npz")))
    W = {k: torch.from_numpy(v) for k, v in weights(name).items()}
    tok, lengths, spk, sty = inputs(name)
    t = lambda a: None if a is None else torch.from_numpy(a)
    with torch.no_grad():
        mu, logw, mask = TO.text_encoder_forward(W, oracle_cfg(name), t(tok), t(lengths), spk=t(spk), sty=t(sty))
        al = TO.align(mu, logw, mask)
    assert np.abs(mu.numpy() - g["mu"]).max() <= 1e-6 and np.abs(logw.numpy() - g["logw"]).max() <= 1e-6
    assert np.array_equal(al["w_ceil"].numpy(), g["w_ceil"]) and np.array_equal(al["y_lengths"].numpy(), g["y_lengths"])
    assert np.array_equal(al["attn"].numpy().astype(np.int8), g["attn"])
    assert np.abs(al["mu_y"].numpy() - g["mu_y"]).max() <= 1e-6
    # the alignment is a partition: every valid frame belongs to exactly one token, in order
    a = g["attn"][:, 0].astype(np.int64)
    for b in range(a.shape[0]):
        ylen = int(g["y_lengths"][b])
        assert (a[b].sum(0)[:ylen] == 1).all() and (a[b].sum(0)[ylen:] == 0).all()
        assert (np.diff(a[b].argmax(0)[:ylen]) >= 0).all()


def test_module_checkpoint_surface():
    kw = ctor_kwargs("gedex_lj")
    m = T.TextEncoder(**kw)
    w = weights("gedex_lj")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    assert set(m.state_dict()) == set(w)
    with pytest.raises(RuntimeError):
        m.load_state_dict({"emb.weight": torch.zeros(3, 3)})
    with pytest.raises(RuntimeError):                      # no CPU path
        m(torch.zeros(1, 4, dtype=torch.long), torch.tensor([4]))
    with pytest.raises(NotImplementedError):
        T.TextEncoder(**dict(kw, use_decay=True))


# ---------------------------------------------------------------------------------------------------- GPU
_MOD = {}


def gpu_module(name):
    if name not in _MOD:
        m = T.TextEncoder(**ctor_kwargs(name))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in weights(name).items()})
        _MOD[name] = m.cuda().eval()
    return _MOD[name]


def _close(got, ref, tag, rel=3e-4):
    err = np.abs(got - ref).max()
    assert np.isfinite(got).all() and err <= rel * max(1.0, np.abs(ref).max()), (tag, float(err))


def _run(name, tok, lengths, spk, sty, length_scale=1.0):
    m = gpu_module(name)
    c = lambda a: None if a is None else torch.from_numpy(a).cuda()
    if name.startswith("dex"):
        mu, logw, mask = m(c(tok), c(lengths), c(sty), length_scale=length_scale)
    else:
        mu, logw, mask = m(c(tok), c(lengths), spk=c(spk), length_scale=length_scale)
    mu_y, y_mask, attn, y_len, y_max = m.align()
    return dict(mu=mu.cpu().numpy(), logw=logw.cpu().numpy(), x_mask=mask.cpu().numpy(), w_ceil=m._last["w_ceil"].cpu().numpy(),
                y_lengths=y_len.cpu().numpy(), y_mask=y_mask.cpu().numpy(), attn=attn.cpu().numpy(), mu_y=mu_y.cpu().numpy(), y_max=y_max)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_text_matches_reference_golden(name):
    g = dict(np.load(os.path.join(GOLD, f"text_{name}.npz")))
    o = _run(name, *inputs(name))
    _close(o["mu"], g["mu"], "mu"); _close(o["logw"], g["logw"], "logw")
    assert np.array_equal(o["x_mask"], g["x_mask"])
    assert np.array_equal(o["w_ceil"], g["w_ceil"]) and np.array_equal(o["y_lengths"], g["y_lengths"])
    assert np.array_equal(o["y_mask"], g["y_mask"]) and np.array_equal(o["attn"].astype(np.int8), g["attn"])
    _close(o["mu_y"], g["mu_y"], "mu_y")
    # mu_y is a gather of mu_x columns: exact against the library's own mu_x
    gather = np.einsum("bit,bfi->bft", o["attn"][:, 0], o["mu"])
    assert np.array_equal(gather, o["mu_y"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,L,lens,scale", [("gedex_lj", 1, 1, [1], 1.0), ("gedex_lj", 3, 70, [70, 33, 5], 1.0),
                                                  ("gedex_vctk", 2, 130, [130, 64], 1.0), ("dex_vctk", 4, 40, [40, 39, 2, 17], 1.0),
                                                  ("gedex_lj", 2, 50, [50, 21], 1.3)])
def test_text_matches_oracle(name, B, L, lens, scale):
    tok, lengths, spk, sty = inputs(name, B, L, lens)
    o = _run(name, tok, lengths, spk, sty, scale)
    W = {k: torch.from_numpy(v) for k, v in weights(name).items()}
    t = lambda a: None if a is None else torch.from_numpy(a)
    with torch.no_grad():
        mu, logw, mask = TO.text_encoder_forward(W, oracle_cfg(name), t(tok), t(lengths), spk=t(spk), sty=t(sty))
        al = TO.align(mu, logw, mask, scale)
    _close(o["mu"], mu.numpy(), "mu"); _close(o["logw"], logw.numpy(), "logw")
    # fixed seeds: no ceil lands on an exact integer boundary, so the durations - and everything derived from them - are
    # IDENTICAL (round 2 skipped the alignment checks when one flipped; a flip now fails the test)
    flips = int((o["w_ceil"] != al["w_ceil"].numpy()).sum())
    assert flips == 0, f"{flips} duration ceil flips"
    assert np.array_equal(o["y_lengths"], al["y_lengths"].numpy()) and o["y_max"] == al["y_max_length"]
    assert np.array_equal(o["attn"], al["attn"].numpy())
    _close(o["mu_y"], al["mu_y"].numpy(), "mu_y")


@pytest.mark.gpu
def test_text_feeds_the_decoder():
    """tokens -> mu_y / y_mask -> the sampler (tts.py:34-55): shapes line up, lengths are multiples of 4, output finite."""
    from tests import gpu_util as U
    o = _run("gedex_lj", *inputs("gedex_lj"))
    cfg, eng, w = U.engine_for("gedex_lj")
    Ty = o["mu_y"].shape[2]
    assert Ty % 4 == 0 and Ty >= o["y_max"]
    z = synth.normalish("z_text", o["mu_y"].shape, 3) / np.float32(1.5) + o["mu_y"]
    out = eng.sample(torch.from_numpy(z.astype(np.float32)), torch.from_numpy(o["y_mask"]), torch.from_numpy(o["mu_y"]), 4)
    assert out.shape == o["mu_y"].shape and torch.isfinite(out).all()
