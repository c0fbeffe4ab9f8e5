"""ORACLE TOOLING — generate golden vectors by running the REAL reference (imported from
/root/reference via oracle/ref_import.py) on portable synthetic weights/inputs.

Run only in the build container:   python -m oracle.make_golden
Writes small fixtures (data only: inputs + expected outputs) under tests/golden/.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SIGMAS = [80.0, 1.0, 0.002]


def sub(t: torch.Tensor) -> np.ndarray:
    """Strided subsample of a stage checkpoint [B,C,H,W] (keeps fixtures small)."""
    return t.detach()[:, ::8, ::8, ::8].contiguous().numpy().astype(np.float32)


def tap_hooks(m, cfg, store):
    hs = []
    dn = m.denoise_fn
    for i in range(len(cfg.dim_mults)):
        hs.append(dn.downs[i][2].register_forward_hook(lambda mod, a, o, i=i: store.__setitem__(f"down{i}", sub(o))))
    def vit_hook(mod, a, o):
        store["dit_in"], store["dit_out"] = sub(a[0]), sub(o)
    hs.append(dn.vit.register_forward_hook(vit_hook))
    hs.append(dn.ups[0][2].register_forward_hook(lambda mod, a, o: store.__setitem__("up0", sub(o))))
    if cfg.variant == "dex":
        hs.append(dn.tv_adaptor.register_forward_hook(lambda mod, a, o: store.__setitem__("tv", sub(o))))
        hs.append(dn.tiv_adaptor.register_forward_hook(lambda mod, a, o: store.__setitem__("tiv", sub(o))))
    return hs


def manifest(name, cfg):
    w = synth.make_weights(C.param_shapes(cfg), seed=0)
    m = ref_import.build_reference_diffusion(cfg, w)
    sd = m.state_dict()
    man = {"config": cfg.to_dict(), "keys": {k: list(v.shape) for k, v in sd.items()}}
    if name in C.PRESETS:
        with open(os.path.join(OUT, f"manifest_{name}.json"), "w") as f:
            json.dump(man, f, indent=0, sort_keys=False)
    return m


@torch.no_grad()
def golden_model(name, cfg, B, T, lengths, sampler_steps, dex_dims=None, spk=False):
    m = manifest(name, cfg)
    mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
    tmu, tmask, tz = map(torch.from_numpy, (mu, mask, z))
    eps = torch.from_numpy(synth.normalish("eps", (B, 80, T), 5))
    extra, extra_np = [], {}
    kw = {}
    if cfg.variant == "dex":
        Tr, Ts, sl = dex_dims
        ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
        extra = [[torch.from_numpy(r) for r in ref], torch.from_numpy(ref_len), torch.from_numpy(sty),
                 torch.from_numpy(sty_len)]
        extra_np = {"ref": np.stack(ref), "ref_lengths": ref_len, "sty": sty, "sty_lengths": sty_len}
    if spk:
        kw["spk"] = torch.from_numpy(synth.normalish("spk", (B, cfg.spk_emb_dim), 9))
        extra_np["spk"] = kw["spk"].numpy()
    out = {"mu": mu, "mask": mask, "z": z, "eps": eps.numpy(), "lengths": lengths, **extra_np}
    for s in SIGMAS:
        x = tmu + s * eps
        store = {}
        hooks = tap_hooks(m, cfg, store) if s == 1.0 else []
        # sampler-style call: 0-dim sigma (DEX needs B == 1 in the reference)
        d = m.precond_model(x, torch.tensor(s), tmask, tmu, *extra, **kw)
        for h in hooks:
            h.remove()
        out[f"precond_sigma{s}"] = d.numpy()
        for k, v in store.items():
            out[f"tap_{k}"] = v
    for n in sampler_steps:
        args = (tz, tmask, tmu, *extra, kw.get("spk"), n)
        y = m.sampler(*args)
        out[f"sampler_n{n}"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if k.startswith(("precond", "sampler"))})


def golden_sigmas():
    mod = ref_import.import_reference("GeDEX-TTS")
    edm = sys.modules["model.edm"]

    class Rec:
        sigma_min, sigma_max = 0, float("inf")

        def __init__(self):
            self.s = []

        def round_sigma(self, s):
            return torch.as_tensor(s)

        def __call__(self, x, sigma, mask, mu, spk=None):
            self.s.append(float(sigma))
            return x

    out = {}
    for n in (2, 10, 50, 100):
        r = Rec()
        edm.ablation_sampler(net=r, latents=torch.zeros(1, 1, 1), num_steps=n, solver="euler",
                             discretization="edm", schedule="linear", scaling="none")
        out[f"n{n}"] = np.asarray(r.s, dtype=np.float32)
    np.savez(os.path.join(OUT, "sigma_tables.npz"), **out)
    print("sigmas n50:", out["n50"][:3], out["n50"][-3:])


def golden_audio():
    from scipy.io import wavfile
    stft, tools = ref_import.import_reference_audio("DEX-TTS")
    tac = stft.TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000)
    sr, wav = wavfile.read("/root/reference/DEX-TTS/syn_samples/sample1.wav")
    wav = wav.astype(np.float32) / (32768.0 if wav.dtype == np.int16 else 1.0)
    if wav.ndim > 1:
        wav = wav[:, 0]
    w1 = wav[:22050].astype(np.float32)
    n = np.arange(16000, dtype=np.float64)
    chirp = (0.8 * np.sin(2 * np.pi * (100.0 * n / 22050 + 0.5 * 3000.0 * (n / 22050) ** 2))
             + 0.3 * (synth.uniform01("audio", 16000) * 2 - 1)).astype(np.float32) * 1.2   # exercises the clip
    out = {"sr": np.int64(sr), "n_total": np.int64(len(wav))}
    for tag, w in (("sample1_1s", w1), ("chirp", chirp)):
        mel, energy = tools.get_mel_from_wav(w, tac)
        out[f"{tag}_wav"], out[f"{tag}_mel"], out[f"{tag}_energy"] = w, mel, energy
    melf, _ = tools.get_mel_from_wav(wav, tac)
    out["sample1_full_frames"] = np.int64(melf.shape[1])
    out["mel_basis_rowsum"] = tac.mel_basis.numpy().sum(1).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "audio_mel.npz"), **out)
    print("audio:", sr, len(wav), melf.shape, out["sample1_1s_mel"].shape)


@torch.no_grad()
def golden_heun():
    """Second-order branch of the reference's own ablation_sampler (edm.py:207-214) on the inputs of the
    gedex_lj / dex_vctk fixtures (same seeds, so only the outputs are stored), plus the sigma sequence the
    network is called with (pins t' = t + 1*h in fp32)."""
    out = {}
    for name, cfg, B, T, lengths, steps, dex_dims in (
            ("gedex_lj", C.gedex_lj(), 2, 64, [64, 44], (4, 7), None),
            ("dex_vctk", C.dex_vctk(), 1, 64, [57], (4,), (40, 40, [33]))):
        m = manifest(name + "_heun", cfg)
        edm = sys.modules[type(m.precond_model).__module__]
        mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
        tmu, tmask, tz = map(torch.from_numpy, (mu, mask, z))
        kw = {}
        if cfg.variant == "dex":
            Tr, Ts, sl = dex_dims
            ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, sty_lengths=sl)
            kw = dict(ref=[torch.from_numpy(r) for r in ref], ref_lengths=torch.from_numpy(ref_len),
                      sty=torch.from_numpy(sty), sty_lengths=torch.from_numpy(sty_len))
        for n in steps:
            y = edm.ablation_sampler(net=m.precond_model, latents=tz, mask=tmask, mu=tmu, spk=None, num_steps=n,
                                     solver="heun", discretization="edm", schedule="linear", scaling="none", **kw)
            out[f"{name}_n{n}"] = y.numpy()

    ref_import.import_reference("GeDEX-TTS")
    edm = sys.modules["model.edm"]

    class Rec:
        sigma_min, sigma_max = 0, float("inf")

        def __init__(self):
            self.s = []

        def round_sigma(self, s):
            return torch.as_tensor(s)

        def __call__(self, x, sigma, mask, mu, spk=None):
            self.s.append(float(sigma))
            return x * 0.5

    for n in (6, 50):
        r = Rec()
        edm.ablation_sampler(net=r, latents=torch.ones(1, 1, 1), num_steps=n, solver="heun",
                             discretization="edm", schedule="linear", scaling="none")
        out[f"sigmas_n{n}"] = np.asarray(r.s, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "heun.npz"), **out)
    print("heun:", {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})


@torch.no_grad()
def golden_churn():
    """Stochastic branch of the reference's own ablation_sampler (edm.py:194-196, S_churn > 0) on the gedex_lj fixture
    inputs, with the per-step noise injected through the sampler's ``randn_like`` argument (stored: it is the draw the
    library has to be fed), Euler and Heun, plus the t_hat sequence the network sees."""
    out = {}
    cfg = C.gedex_lj()
    m = manifest("gedex_lj_churn", cfg)
    edm = sys.modules[type(m.precond_model).__module__]
    B, T, lengths = 2, 64, [64, 44]
    mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
    tmu, tmask, tz = map(torch.from_numpy, (mu, mask, z))
    for solver, n, S_churn, S_min, S_max, S_noise in (("euler", 6, 30.0, 0.05, 50.0, 1.003), ("heun", 4, 10.0, 0.0, float("inf"), 1.0)):
        noise = synth.normalish(f"churn_{solver}", (n, B, 80, T), 4321)
        it = iter(torch.from_numpy(noise))
        y = edm.ablation_sampler(net=m.precond_model, latents=tz, mask=tmask, mu=tmu, spk=None, num_steps=n, solver=solver,
                                 discretization="edm", schedule="linear", scaling="none", randn_like=lambda x: next(it),
                                 S_churn=S_churn, S_min=S_min, S_max=S_max, S_noise=S_noise)
        tag = f"{solver}_n{n}"
        out[tag] = y.numpy()
        # (the noise itself is regenerated by the tests from the same portable generator: synth.normalish(f"churn_{solver}", ...))
        out[tag + "_params"] = np.asarray([S_churn, S_min, min(S_max, 3.0e38), S_noise], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "churn.npz"), **out)
    print("churn:", {k: v.shape for k, v in out.items()})


@torch.no_grad()
def golden_dex_stacked():
    """Batched DEX, pinned to the reference (VERDICT r3 Missing #3): the reference cannot run DEX at B > 1
    (ref_encoder.py:157,248 broadcast one time token), so a batched run is DEFINED as its B = 1 runs at one padded
    T / Tr / Ts, stacked.  Three utterances of different lengths: every one goes through the REAL
    reference alone (same padded shapes), the outputs are stacked; the oracle's / the library's batched path must
    reproduce the stack."""
    cfg = C.dex_vctk()
    B, T, lengths = 3, 64, [64, 51, 37]
    Tr = Ts = 40
    ref_lengths, sty_lengths = [40, 31, 25], [40, 33, 21]
    m = manifest("dex_vctk_b3_stacked", cfg)
    mu, mask, z, lengths = synth.make_inputs(B, T, lengths, seed=1234)
    eps = synth.normalish("eps", (B, 80, T), 5)
    ref, ref_len, sty, sty_len = synth.make_dex_style(B, Tr, Ts, cfg.mid_dim, ref_lengths=ref_lengths, sty_lengths=sty_lengths)
    out = {"mu": mu, "mask": mask, "z": z, "eps": eps, "lengths": lengths, "ref": np.stack(ref), "ref_lengths": ref_len,
           "sty": sty, "sty_lengths": sty_len}
    rows = {f"precond_sigma{s}": [] for s in SIGMAS}
    rows["sampler_n4"] = []
    for b in range(B):
        sl = slice(b, b + 1)
        extra = [[torch.from_numpy(r[sl]) for r in ref], torch.from_numpy(ref_len[sl]), torch.from_numpy(sty[sl]), torch.from_numpy(sty_len[sl])]
        tmu, tmask, tz, teps = (torch.from_numpy(a[sl]) for a in (mu, mask, z, eps))
        for s in SIGMAS:
            rows[f"precond_sigma{s}"].append(m.precond_model(tmu + s * teps, torch.tensor(s), tmask, tmu, *extra).numpy())
        rows["sampler_n4"].append(m.sampler(tz, tmask, tmu, *extra, None, 4).numpy())
    for k, v in rows.items():
        out[k] = np.concatenate(v, axis=0)
    np.savez_compressed(os.path.join(OUT, "dex_vctk_b3_stacked.npz"), **out)
    print("dex_vctk_b3_stacked", {k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if k.startswith(("precond", "sampler"))})


def main():
    if "--libritts-only" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_model("dex_libritts", C.dex_libritts(), B=1, T=32, lengths=[29], sampler_steps=[4], dex_dims=(24, 24, [19]))
        return
    if "--stacked-only" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_dex_stacked()
        return
    if "--churn-only" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_churn()
        return
    if "--heun-only" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        golden_heun()
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    golden_sigmas()
    golden_model("gedex_lj", C.gedex_lj(), B=2, T=64, lengths=[64, 44], sampler_steps=[4, 10])
    golden_model("gedex_lj_n50", C.gedex_lj(), B=1, T=48, lengths=[48], sampler_steps=[50])
    golden_model("gedex_vctk", C.gedex_vctk(), B=2, T=32, lengths=[32, 21], sampler_steps=[4], spk=True)
    golden_model("dex_vctk", C.dex_vctk(), B=1, T=64, lengths=[57], sampler_steps=[4, 10], dex_dims=(40, 40, [33]))
    golden_model("dex_libritts", C.dex_libritts(), B=1, T=32, lengths=[29], sampler_steps=[4], dex_dims=(24, 24, [19]))
    golden_audio()
    golden_heun()
    golden_churn()
    golden_dex_stacked()


if __name__ == "__main__":
    main()
