#!/usr/bin/env python3
"""Audit of dit_rowchain64a_kernel's build: the kernel is two asm statements (tools/gen_rowchain_a.py) around a few lines of C++, and the
first statement leaves loads in flight into v[64:95], v[128:191] and a[0:63].  hipcc knows nothing of that, so this script compiles
dit_rowchain.hip to assembly and checks that no compiler-generated instruction of the kernel (anything outside ;;#ASMSTART .. ;;#ASMEND)
names a vector register from v64 up or an accumulation register, and that the kernel has no scratch.

    python tools/audit_rowchain_a.py [extra hipcc flags, e.g. -DDEX_LP_F16]        exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HI_V64 = r"v(?:6[4-9]|[7-9]\d|1\d\d|2[0-4]\d)\b|v\[(?:6[4-9]|[7-9]\d|1\d\d|2[0-4]\d):|"


def audit(extra=(), source="dit_rowchain.hip", kernel="dit_rowchain64a_kernel", vgprs=True):
    """source / kernel: which generated-stream kernel; vgprs: also forbid v64 and up outside the statements (the row chain leaves loads in
    flight into them; the 64-query attention - attention_q64.hip, attn_q64_kernel - only keeps O^T and Q in the accumulation file)"""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rc.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
               "--offload-device-only", "-S", *extra, os.path.join(ROOT, "dex_tts_amd", "csrc", source), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return [f"hipcc failed: {r.stderr[-2000:]}"]
        text = open(out).read()
    bad = []
    for m in re.finditer(r"^(_ZN\w*" + kernel + r"\w*):\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        in_asm = False
        hi = re.compile(r"\b(?:" + (HI_V64 if vgprs else "") + r"a\d+\b|a\[\d+:)")
        for ln in body.split("\n"):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm and t and not t.startswith((";", ".", "s_")) and not t.endswith(":"):
                code = t.split(";")[0]
                if hi.search(code):
                    bad.append(f"{name}: compiler instruction touches a register the streams own: {code.strip()}")
        if re.search(r"scratch_|buffer_(?:load|store)_dword\s+v\d+, off, s\[0:3\]", body):
            bad.append(f"{name}: scratch access")
    if not re.search(kernel, text):
        bad.append("kernel not found in the assembly")
    return bad


if __name__ == "__main__":
    problems = audit(sys.argv[1:]) + audit(sys.argv[1:], "attention_q64.hip", "attn_q64_kernel", vgprs=False)
    for b in problems:
        print(b)
    print("generated-stream kernels audit:", "FAILED" if problems else "clean")
    sys.exit(1 if problems else 0)
