"""Test-infrastructure analysis (not a product path): where does the distance of a reduced-precision mode from the fp32
reference come from?  Runs the oracle's sampler on BASELINE.json configs[1] with the OPERANDS of selected contraction families
rounded to fp16 / bf16 (fp32 accumulation, like the MFMA path), everything else exact, and prints the distance of each variant
from the exact oracle run.  Families: conv3 (U-Net 3x3 convs), conv1 (1x1 convs: shortcuts, linear-attention qkv / out, patch
pointwise), convx (strided / transposed / grouped convs: Down / Upsample, patch depthwise, positional conv), linattn (the two
einsums), lin (every F.linear: DiT qkv / proj / fc1 / fc2 / adaLN / final layer, time MLPs), attn (softmax attention matmuls).
Modes per family: 'xw' both operands rounded, 'x' activations only, 'w' weights only, '-' exact.
    python -m oracle.lowp_emulate [--dtype fp16] [--T 512] [--steps 50]"""
import argparse, sys, os, time
import numpy as np
import torch
import torch.nn.functional as F_real

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dex_tts_amd import synth, config as C          # noqa: E402
from oracle import dex_oracle as O                   # noqa: E402

FAMILIES = ["conv3", "conv1", "convx", "linattn", "lin", "attn"]


class Rounder:
    def __init__(self, dtype):
        self.dtype = dtype
        self.mode = {f: "-" for f in FAMILIES}

    split = False         # weights as hi + lo of the 16-bit type (two MFMAs per product, activations rounded once)
    dither = 0            # K > 0: weights use one of K "twins" per network evaluation (floor / ceil in the 16-bit grid chosen per element
    step = 0              # so that the mean over K consecutive evaluations is the fp32 weight to 1 / (2K) ulp)

    def r(self, t, fam, which):
        m = self.mode[fam]
        if which not in m:
            return t
        if which == "w" and self.split:
            hi = t.to(self.dtype).to(torch.float32)
            lo = (t - hi).to(self.dtype).to(torch.float32)          # fp16: may be subnormal (the MFMA keeps subnormal inputs: tools/mfmadenorm)
            return hi + lo
        if which == "w" and self.dither:
            lo = t.to(self.dtype)                                   # round to nearest, then find floor / ceil neighbours
            lo32 = lo.to(torch.float32)
            up = torch.nextafter(lo, torch.full_like(lo, float("inf"))).to(torch.float32)
            dn = torch.nextafter(lo, torch.full_like(lo, float("-inf"))).to(torch.float32)
            fl = torch.where(lo32 <= t, lo32, dn)
            ce = torch.where(lo32 >= t, lo32, up)
            frac = torch.where(ce > fl, (t - fl) / (ce - fl), torch.zeros_like(t))
            K = self.dither
            k = self.step % K
            # thresholds in a bit-reversed order so that consecutive evaluations alternate
            order = {1: [0], 2: [0, 1], 4: [0, 2, 1, 3], 8: [0, 4, 2, 6, 1, 5, 3, 7]}[K]
            thr = (order[k] + 0.5) / K
            return torch.where(frac > thr, ce, fl)
        return t.to(self.dtype).to(torch.float32)


class FProxy:
    """stands in for torch.nn.functional inside the oracle module"""
    def __init__(self, rd):
        self.rd = rd

    def __getattr__(self, name):
        return getattr(F_real, name)

    def _fam(self, w, kw):
        if kw.get("groups", 1) != 1 or kw.get("stride", 1) != 1:
            return "convx"
        return "conv3" if w.shape[-1] == 3 else "conv1"

    def conv2d(self, x, w, b=None, **kw):
        fam = self._fam(w, kw)
        return F_real.conv2d(self.rd.r(x, fam, "x"), self.rd.r(w, fam, "w"), b, **kw)

    def conv_transpose2d(self, x, w, b=None, **kw):
        return F_real.conv_transpose2d(self.rd.r(x, "convx", "x"), self.rd.r(w, "convx", "w"), b, **kw)

    def linear(self, x, w, b=None):
        return F_real.linear(self.rd.r(x, "lin", "x"), self.rd.r(w, "lin", "w"), b)


def run(rd, W, cfg, mask, mu, z, steps):
    real_einsum, real_matmul = torch.einsum, torch.Tensor.__matmul__
    def einsum(eq, a, b):
        return real_einsum(eq, rd.r(a, "linattn", "x"), rd.r(b, "linattn", "x"))
    def matmul(a, b):
        return real_matmul(rd.r(a, "attn", "x"), rd.r(b, "attn", "x"))
    O.F = FProxy(rd)
    real_dn = O.denoiser_forward
    def dn(*args, **kw):
        out = real_dn(*args, **kw)
        rd.step += 1
        return out
    O.denoiser_forward = dn
    rd.step = 0
    torch.einsum = einsum
    torch.Tensor.__matmul__ = matmul
    try:
        with torch.no_grad():
            return O.diffusion_infer(W, cfg, mask, mu, steps, z).numpy()
    finally:
        O.F = F_real
        O.denoiser_forward = real_dn
        torch.einsum = real_einsum
        torch.Tensor.__matmul__ = real_matmul


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    cfg = C.PRESETS["gedex_lj"]()
    W = O.as_torch(synth.make_weights(C.param_shapes(cfg)))
    mu, mask, z, _ = synth.make_inputs(1, a.T, None, seed=1234)
    mu, mask, z = map(torch.from_numpy, (mu, mask, z))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rd = Rounder(dt)
    t0 = time.time()
    y0 = run(rd, W, cfg, mask, mu, z, a.steps)
    print(f"# exact oracle run: {time.time() - t0:.1f} s; |y| max {np.abs(y0).max():.2f}; operands rounded to {a.dtype}", flush=True)
    variants = [("all xw", {f: "xw" for f in FAMILIES})]
    variants += [(f"only {f} xw", {f: "xw"}) for f in FAMILIES]
    variants += [(f"all but {f}", {g: "xw" for g in FAMILIES if g != f}) for f in FAMILIES]
    variants += [("all: weights only", {f: "w" for f in FAMILIES}), ("all: activations only", {f: "x" for f in FAMILIES}),
                 ("conv3 + lin exact, rest xw", {f: "xw" for f in FAMILIES if f not in ("conv3", "lin")}),
                 ("conv3 + lin + attn exact, rest xw", {f: "xw" for f in FAMILIES if f not in ("conv3", "lin", "attn")})]
    if a.variants == "weights":
        variants = [(f"weights of {f} only", {f: "w"}) for f in ("conv3", "conv1", "convx", "lin")]
        variants += [(f"all xw, weights of {f} exact", {g: ("x" if g == f else "xw") for g in FAMILIES}) for f in ("conv3", "conv1", "convx", "lin")]
        variants += [("all xw, weights of conv3+conv1 exact", {g: ("x" if g in ("conv3", "conv1") else "xw") for g in FAMILIES}),
                     ("all xw, weights of conv3+conv1+lin exact", {g: ("x" if g in ("conv3", "conv1", "lin") else "xw") for g in FAMILIES}),
                     ("all xw, weights of conv3+conv1+convx exact", {g: ("x" if g in ("conv3", "conv1", "convx") else "xw") for g in FAMILIES})]
    if a.variants == "dither":
        variants = [("all xw", {f: "xw" for f in FAMILIES}, 0)] + [(f"all xw, weights dithered over {K} evaluations", {f: "xw" for f in FAMILIES}, K) for K in (2, 4, 8)]
    variants = [v if len(v) == 3 else (v[0], v[1], 0) for v in variants]
    if a.variants == "split":
        variants = [("all xw", {f: "xw" for f in FAMILIES}, 0), ("all xw, weights hi + lo", {f: "xw" for f in FAMILIES}, -1)]
    for name, spec, dith in variants:
        rd.dither, rd.split = max(dith, 0), dith < 0
        rd.mode = {f: spec.get(f, "-") for f in FAMILIES}
        y = run(rd, W, cfg, mask, mu, z, a.steps)
        d = np.abs(y - y0)
        print(f"{name:36s} max {d.max():.3e}  mean {d.mean():.3e}", flush=True)


if __name__ == "__main__":
    main()
