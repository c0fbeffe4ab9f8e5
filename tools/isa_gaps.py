#!/usr/bin/env python3
"""Per-MFMA-gap instruction census of a gfx950 .s file (hipcc -save-temps): for every v_mfma, the instructions that follow it
up to the next one, by class.  Usage: isa_gaps.py file.s [kernel-substring] [first_mfma last_mfma]"""
import re, sys, collections
path = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else ""
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0; hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
lines = open(path).read().split("\n")
start = 0
if kern:
    for i, l in enumerate(lines):
        if l.startswith("_Z") and kern in l and l.rstrip().endswith(":"): start = i; break
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "bar"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    return "other"
n = -1; cur = None; out = []
for l in lines[start:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"): break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        if t.endswith(":") and cur is not None: cur.append("LBL")
        continue
    op = t.split()[0]
    c = cls(op)
    if c == "MFMA":
        n += 1; cur = []; out.append((n, t, cur)); continue
    if cur is not None: cur.append(c if c != "wait" else "wait:" + t.split(None, 1)[1].split(";")[0].strip())
for n, t, cur in out:
    if n < lo or n > hi: continue
    cnt = collections.Counter(x.split(":")[0] if not x.startswith("wait") else x for x in cur)
    print(f"{n:4d} {t.split()[1]:>12s} | " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
