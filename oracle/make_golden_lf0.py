"""ORACLE TOOLING — golden vectors for the deterministic tail of the DEX f0 front-end: ``lf0 = log f0`` on the voiced frames
followed by the reference's own ``normalize_lf0`` (DEX-TTS/synthesize.py:26-38, :55-58).  The reference module cannot be
imported whole here (it imports pyworld / soundfile / librosa / resampy, none of which is in the image or in /root/reference),
so this script pulls the ONE function ``normalize_lf0`` out of the reference source with ``ast`` at generation time and runs it
— nothing of it is stored: the fixture holds synthetic f0 tracks and the function's outputs.

DIO / StoneMask (pyworld, a third-party CPU pitch tracker) stay a host input: "parity unpinned" would be the only honest
status for a restatement, see DESIGN.md §7.

Run only in the build container:   python -m oracle.make_golden_lf0      -> tests/golden/lf0.npz
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dex_tts_amd import synth  # noqa: E402
from oracle import style_oracle as SO  # noqa: E402

REF = "/root/reference/DEX-TTS/synthesize.py"
OUT = os.path.join(ROOT, "tests", "golden")


def reference_normalize_lf0():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "normalize_lf0")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns["normalize_lf0"]


def f0_tracks():
    """Synthetic pitch tracks in Hz, float32, 0 = unvoiced: a contour with unvoiced gaps, an all-unvoiced one, a constant one
    (std == 0 branch), a single voiced frame, and one holding an exact 1.0 Hz frame (log = 0: counted as unvoiced)."""
    T = 348
    u = synth.uniform01("f0_track", T)
    base = (120.0 + 60.0 * np.sin(np.arange(T) / 17.0) + 25.0 * (u - 0.5)).astype(np.float32)
    gaps = (synth.uniform01("f0_gap", T) < 0.35)
    a = np.where(gaps, 0.0, base).astype(np.float32)
    b = np.zeros(T, np.float32)
    c = np.where(np.arange(T) % 3 == 0, 0.0, 200.0).astype(np.float32)
    d = np.zeros(40, np.float32); d[17] = 333.0
    e = a[:100].copy(); e[5] = 1.0
    return {"contour": a, "unvoiced": b, "constant": c, "single": d, "one_hz": e}


def main():
    norm = reference_normalize_lf0()
    out = {}
    for k, f0 in f0_tracks().items():
        lf0 = f0.copy()
        nz = np.nonzero(f0)
        lf0[nz] = np.log(f0[nz])                                  # synthesize.py:55-57
        want = norm(lf0.copy())                                    # synthesize.py:58
        out[f"{k}_f0"], out[f"{k}_lf0"] = f0, want.astype(np.float32)
        got = SO.lf0_from_f0(f0)
        print(f"{k:10s} T={len(f0):4d} voiced={int((f0 != 0).sum()):4d}  oracle vs reference max|d| = {float(np.abs(got - want).max()):.2e}")
    np.savez_compressed(os.path.join(OUT, "lf0.npz"), **out)
    print("wrote lf0.npz")


if __name__ == "__main__":
    main()
