"""Build libdexamd.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m dex_tts_amd.build            # incremental
    python -m dex_tts_amd.build --force

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libdexamd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-variable", "-Wno-unused-value"]


# the MFMA-heavy kernel files of the reduced-precision modes are compiled twice: operands bf16 (namespace dex::bf16) and,
# with -DDEX_LP_F16, fp16 (namespace dex::f16), and -DDEX_LP_F16 -DDEX_LP_WSPLIT (namespace dex::f16w: weights as hi + lo) — csrc/lp_config.h
LP_SOURCES = ("conv3x3_bf16.hip", "conv3x3_stream.hip", "igemm_bf16.hip", "attention_bf16.hip", "attention_direct.hip",
              "attention_q64.hip", "dit_rowchain.hip", "linattn_fused.hip", "pos_conv.hip", "convt_up.hip", "conv3x3_regw.hip", "conv_down.hip", "patch_embed.hip")


# per-file flags.  -fno-slp-vectorize: the SLP vectorizer turns adjacent fp32 multiplies / adds into v_pk_*_f32, and a packed fp32
# instruction between a wave's MFMAs costs the matrix pipe ~11 cycles where a plain one is free (tools/mfmabench)
FILE_FLAGS = {"conv3x3_regw.hip": ["-fno-slp-vectorize"]}


def _env_file_flags():
    """DEX_FILE_FLAGS="a.hip=-DX -fY;b.hip=-DZ": extra per-file flags for experiments (tools/pkab.sh)."""
    out = {}
    for item in os.environ.get("DEX_FILE_FLAGS", "").split(";"):
        if "=" in item:
            f, fl = item.split("=", 1)
            out[f.strip()] = fl.split()
    return out


def sources():
    """(source file, extra flags, object suffix) per compilation."""
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip"):
            ff = FILE_FLAGS.get(f, []) + _env_file_flags().get(f, [])
            out.append((f, list(ff), ""))
            if f in LP_SOURCES:
                out.append((f, ["-DDEX_LP_F16", *ff], ".f16"))
                out.append((f, ["-DDEX_LP_F16", "-DDEX_LP_WSPLIT", *ff], ".f16w"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job, force):
    src, extra, suffix = job
    obj = os.path.join(OBJ, src[:-4] + suffix + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    deps += [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".inc")]
    deps.append(os.path.join(HERE, "..", "include", "dex_amd.h"))
    if force or _stale(obj, deps):
        cmd = [HIPCC, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src} {extra}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


# Kernels whose asm statements (generated instruction streams) keep state in registers ACROSS statement boundaries - loads in flight
# into v64.. / the accumulation file (dit_rowchain64a_kernel), O^T and Q in the accumulation file (attn_q64_kernel) - which hipcc only
# knows as clobbers: nothing the compiler emits between the statements may touch them.  ADVICE r5: the check (tools/audit_rowchain_a.py)
# ran from the test-suite only, and not with the build's flags; it is part of the build now - every compilation of these files, with
# exactly its flags, and a hit fails the build.  (source, kernel, also-forbid-v64-up, object suffixes it exists in)
AUDITS = (("dit_rowchain.hip", "dit_rowchain64a_kernel", True, ("", ".f16", ".f16w")),
          ("attention_q64.hip", "attn_q64_kernel", False, ("", ".f16", ".f16w")))


def _audit(job):
    src, extra, suffix = job
    for a_src, kernel, vgprs, suffixes in AUDITS:
        if src != a_src or suffix not in suffixes:
            continue
        obj = os.path.join(OBJ, src[:-4] + suffix + ".o")
        stamp = obj + ".audit_ok"
        if os.path.exists(stamp) and os.path.getmtime(stamp) >= os.path.getmtime(obj):
            continue
        sys.path.insert(0, os.path.join(HERE, "..", "tools"))
        import audit_rowchain_a
        problems = audit_rowchain_a.audit(list(extra), src, kernel, vgprs)
        if problems:
            raise RuntimeError(f"register audit of {kernel} ({src} {extra}) FAILED - the compiler touches registers the generated streams own:\n  "
                               + "\n  ".join(problems[:20]))
        open(stamp, "w").write("clean\n")


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(12, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
        list(ex.map(_audit, srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
