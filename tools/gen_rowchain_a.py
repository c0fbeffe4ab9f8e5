#!/usr/bin/env python3
"""Generator of dex_tts_amd/csrc/dit_rowchain_a_core.inc - the hand-scheduled gfx950 instruction streams of the 64-row DiT row chain
(dit_rowchain64a_kernel in dit_rowchain.hip: attention projection + gated residual, LayerNorm + modulate, fc1 + GELU, fc2 + gated
residual, LayerNorm + modulate, the next block's qkv projection, for 64 token rows per workgroup).

Why a generator: the compiler-scheduled kernel issues 11 VALU instructions per MFMA, none of them under an MFMA,
and fetches its weight tiles one 16-KB tile ahead (matrix pipe 0.196 busy, profiles/round4_dex_b32_diag_counters.txt).  Here the whole
chain of a workgroup is ONE straight-line instruction stream per wave with a fixed register map:

  * 4 waves, one per SIMD (512 registers each).  Wave w owns the feature tiles 2w, 2w + 1 (64 features) of every 256-wide output and
    both 32-token tiles: a weight fragment feeds two MFMAs (token tiles T0, T1) and an activation fragment two (feature tiles j0, j1)
    - half the LDS fragment reads per MFMA of the 8-wave kernels.
  * every product TRANSPOSED (weights = A operand, activations = B): acc[r] = C[feature (r & 3) + 8 (r >> 2) + 4 hh][token lane & 31];
    the residual stream of the wave's 64 features x 64 tokens stays in 64 registers; only v^T is computed the plain way round (its
    layout wants the feature in the lane).  The residual rows enter and leave through LDS (LDS-DMA in, whole 1-KB rows out): read
    straight into the accumulator layout a load instruction touches 32 rows and costs the CU's address path 4x a contiguous one.
  * weights stream from L2 into a RING of 56 fragments in the accumulation file (a[0:223]); fragment g + 56 is requested the moment
    fragment g has had its last MFMA - ~100 MFMA slots ahead, against one tile (16 slots) in the C++ kernels.  Accumulators start
    from the bias (LDS reads straight into the accumulator registers), so no epilogue adds one.
  * the MLP is a pipeline of QUARTERS: fc1 of 128 hidden columns (one feature tile per wave, 32 MFMAs) with the GELU of the previous
    quarter in its MFMA gaps, then fc2 over the four K chunks with the last quarter's GELU spread under the first three - the GELU
    (the chain's largest VALU block) is under MFMAs of seven of the eight half-passes instead of two of four.
  * GELU, packing and the 16-byte chunk exchange (v_permlane32_swap) run in place in the consumed accumulator registers.
  * s_waitcnt values are COMPUTED: the generator keeps the queues of outstanding VMEM and LDS operations in program order.

    python tools/gen_rowchain_a.py           # rewrites the .inc (committed; tests/test_cabi.py checks it is up to date)

Register map:
  a[0:223]    weight ring, slot s = a[4 s ..]
  v[0:63]     accumulator set 0: tile (j, T) = v[32 j + 16 T ..];  v[64:127] set 1
  v[128:191]  residual stream X: tile (j, T) = v[128 + 32 j + 16 T ..]
  v[192:223]  activation fragment ring, 8 slots
  v[224:233]  lane constants (addresses), v[234:249] temporaries; v250.. are left to the compiler (the statement's inputs)
  s[60:..]    owned scalars
"""
import os
import struct
import sys

OUT = os.environ.get("RCAGEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dex_tts_amd", "csrc", "dit_rowchain_a_core.inc")
GELU_TERMS = int(os.environ.get("RCAGEN_GELU", "3"))       # erf by Abramowitz-Stegun: 3 = 7.1.25 (|err| 2.5e-5: a twentieth of the fp16 rounding the value gets next,
                                                            # 2 VALU instructions per value less), 5 = 7.1.26 (1.5e-7, the C++ kernels' formula; A/B builds)
DROP = set(filter(None, os.environ.get("RCAGEN_DROP", "").split(",")))   # anatomy builds (results wrong): gelu, mfma, wload, xstore, qkvstore
STORE_MOD = os.environ.get("RCAGEN_STORE", "")              # cache-policy bits on the output stores (A/B: nt, sc1, "sc0 sc1")
XS_SPLIT = tuple(int(v) for v in os.environ.get("RCAGEN_XSPLIT", "10,8,0").split(","))   # x2 row-store atoms (18) in the q / k / v passes
TIMING = os.environ.get("RCAGEN_TIMING", "0") == "1"        # s_memtime stamps after every pass / at every barrier -> %[dbg] (tools/rc64bench -DRCA_TIMING)

# ---- LDS map (bytes) - mirrored by dit_rowchain.hip (RCA_*)
AS, A_ROW = 0, 528
HS, H_ROW = 64 * A_ROW, 1040
XS, X_ROW = HS, 1040                 # the fp32 residual rows on their way in / out share the GELU tile's bytes (64 rows x (1024 + 16))
PRM = HS + 64 * H_ROW
P_SHM, P_SC1M, P_SHN, P_SC1N, P_GMSA, P_BP, P_GMLP, P_B2 = range(8)
P_B1 = 8 * 1024
P_BQ = P_B1 + 2048
PRM_BYTES = P_BQ + 3072
ST = PRM + PRM_BYTES
LDS_BYTES = ST + 64 * 4 * 8

WS = False                           # split-weight build of the streams (set per generation in main()): every weight fragment is followed in the
                                     # stream by its lo fragment (what the fp16 rounding of the weight lost; the lo pack sits LO_BYTES behind the
                                     # hi pack, dit_rowchain.hip LO_*), a product is hi then lo into the same accumulator - both ride the same ring
LO_BYTES = {"wp": 256 * 256 * 2, "w1": 256 * 512 * 2, "w2": 512 * 256 * 2, "wq": 256 * 768 * 2}      # bytes of a matrix's hi pack (RC_H = 256, RC_MLP = 512)
NRING = 32                           # weight ring a[0:127].  The ring position of a fragment is (index in the tile's stream) % NRING and the stream wraps
                                     # into the next tile, so NRING must divide every variant's fragment count (256 / 160 / 96): 40 was tried and read the
                                     # previous tile's slots
AG_O, AG_X = 128, 160                # the NEXT tile's O rows (8 quads) and residual rows (16 quads) wait in a[128:159], a[160:223]

# ---- register map
A0, A1 = 0, 64
def X(j, T): return 128 + 32 * j + 16 * T
def RING(slot): return 192 + 4 * slot
V_OFF16, V_OFF32, V_OFFQ, V_AS, V_HS, V_PRM, V_QK0, V_QK1, V_ST, V_GA = range(224, 234)
def T(k): return 234 + k
V_TOP = 253 if TIMING else 249
# owned SGPRs
S_WA, S_WB = 60, 62           # pairs
S_C1, S_P, S_A1, S_A2, S_A3, S_A4 = 64, 65, 66, 67, 68, 69
S_TMP, S_W8, S_W64, S_W128, S_W256, S_C64, S_EPS, S_XSB, S_N0W, S_TMP2 = 70, 71, 72, 73, 74, 75, 76, 77, 78, 79
S_TS = 80                      # timing build: s_memtime lands in s[80:81], leaves through v[252:253] (v251 = 0) to %[dbg] + 8 k
S_WAL, S_WBL = 82, 84          # pairs: the lo halves' bases (split-weight streams)
S_TOP = 85


def vr(i, n=1): return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"
def ar(i, n=1): return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"
def sr(i, n=1): return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"
def f32(x): return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


class Prog:
    def __init__(self):
        self.out = []
        self.n = 0                      # instructions emitted (hazard distances)
        self.vm_issued = 0              # VMEM operations issued / known complete (in-order return on gfx9: loads and stores share vmcnt)
        self.vm_done = 0
        self.lg_issued = 0
        self.lg_done = 0
        self.mfma_at = {}               # first register of an accumulator tile -> instruction index of the last MFMA that wrote it
        self.stats = {"mfma": 0, "valu": 0, "trans": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "nop": 0}
        self.nstamp = 0
        self.names = []

    def e(self, text):
        if text.startswith("MFMA "):
            self.out.append(f'RCA_MFMA " {text[5:]}\\n\\t"')
            kind = "mfma"
        elif text.startswith("PK "):
            self.out.append(f'RCA_PK " {text[3:]}\\n\\t"')
            kind = "valu"
        else:
            self.out.append(f'"{text}\\n\\t"')
            op = text.split()[0]
            kind = ("trans" if op in ("v_exp_f32", "v_rcp_f32", "v_rsq_f32") else "valu" if op.startswith("v_") else
                    "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_")) else
                    "wait" if op == "s_waitcnt" else "nop" if op == "s_nop" else "salu")
        self.stats[kind] += 1
        self.n += int(text.split()[1]) + 1 if text.startswith("s_nop") else 1

    # ---- counted waits
    def vmem(self, text):
        # vmcnt is a 6-bit counter: the wave must never have more than 63 requests in flight (ring 40 + the next tile's 24 rows once did:
        # silently wrong data).  What the generator KNOWS to be complete is a lower bound, so this wait is conservative.
        if self.vm_issued - self.vm_done >= 60:
            self.e("s_waitcnt vmcnt(56)")
            self.vm_done = self.vm_issued - 56
            self.forced_vm_waits = getattr(self, "forced_vm_waits", 0) + 1
        self.e(text)
        self.vm_issued += 1
        self.max_vm = max(getattr(self, "max_vm", 0), self.vm_issued - self.vm_done)
        return self.vm_issued

    def lds(self, text):
        self.e(text)
        self.lg_issued += 1
        return self.lg_issued

    def wait_vm(self, tag):
        if tag is None or tag <= self.vm_done:
            return
        n = min(self.vm_issued - tag, 63)
        self.e(f"s_waitcnt vmcnt({n})")
        self.vm_done = self.vm_issued - n

    def wait_lg(self, tag):
        if tag is None or tag <= self.lg_done:
            return
        n = min(self.lg_issued - tag, 15)
        self.e(f"s_waitcnt lgkmcnt({n})")
        self.lg_done = self.lg_issued - n

    def stamp(self, name=""):
        if TIMING:
            k = self.nstamp
            self.nstamp += 1
            self.names.append(name)
            self.e(f"s_memtime {sr(S_TS, 2)}")          # SMEM shares lgkmcnt and may return out of order: followed by lgkmcnt(0)
            self.e("s_waitcnt lgkmcnt(0)")
            self.lg_done = self.lg_issued
            self.e(f"v_mov_b32 v252, {sr(S_TS)}")
            self.e(f"v_mov_b32 v253, {sr(S_TS + 1)}")
            self.vmem(f"global_store_dwordx2 v251, v[252:253], %[dbg] offset:{8 * k}")
            self.e("s_nop 1")

    def barrier(self, name=""):
        self.stamp(name)
        self.wait_lg(self.lg_issued)
        self.e("s_barrier")

    def nops(self, states):
        while states > 0:
            k = min(states, 16)
            self.e(f"s_nop {k - 1}")
            states -= k

    def acc_read(self, base):
        """a VALU / LDS instruction is about to read accumulator tile `base`: an MFMA's D needs 12 wait states before any reader"""
        at = self.mfma_at.get(base)
        if at is not None and self.n - at < 14:
            self.nops(14 - (self.n - at))
        self.mfma_at.pop(base, None)

    def text(self):
        return "\n    ".join(self.out)


# ------------------------------------------------------------------------------------------------ passes and the weight stream
class Pass:
    """ntile feature tiles (tile index = tiles[jj] + wmul * w) x 2 token tiles x nks K-steps.  accs[(jj, T)] = accumulator tile."""
    def __init__(self, name, mat, kt, tiles, wmul, ks0, nks, act, act_ks0, accs, bias=None, trans=True):
        self.name, self.mat, self.kt, self.tiles, self.wmul, self.ks0, self.nks = name, mat, kt, tiles, wmul, ks0, nks
        self.act, self.act_ks0, self.accs, self.bias, self.trans = act, act_ks0, accs, bias, trans
        self.ntile = len(tiles)
        self.voff = {(2, 16): V_OFF16, (2, 32): V_OFF32, (1, 16): V_OFFQ}[(wmul, kt)]
        self.nmfma = self.ntile * 2 * nks * (2 if WS else 1)


def set_accs(base, ntile=2):
    return {(jj, Tt): base + 32 * jj + 16 * Tt for jj in range(ntile) for Tt in range(2)}


def passes_of(variant):
    P = []
    if variant != "QKV":
        P.append(Pass("proj", "wp", 16, [0, 1], 2, 0, 16, "as", 0, set_accs(A0), bias=P_BP * 1024))
        for q in range(4):       # fc1 quarter q: hidden tile 4 q + w -> accumulator half q & 1 of set 0
            P.append(Pass(f"fc1.{q}", "w1", 16, [4 * q], 1, 0, 16, "as", 0, set_accs(A0 + 32 * (q & 1), 1), bias=P_B1 + 512 * q))
        for c in range(4):       # fc2 K chunk c: hidden columns 128 c .. (set 1 all the way)
            P.append(Pass(f"fc2.{c}", "w2", 32, [0, 1], 2, 8 * c, 8, "hs", 8 * c, set_accs(A1), bias=P_B2 * 1024 if c == 0 else None))
    if variant != "LAST":
        P += [Pass("q", "wq", 16, [0, 1], 2, 0, 16, "as", 0, set_accs(A0), bias=P_BQ),
              Pass("k", "wq", 16, [8, 9], 2, 0, 16, "as", 0, set_accs(A1), bias=P_BQ + 1024),
              Pass("v", "wq", 16, [16, 17], 2, 0, 16, "as", 0, set_accs(A0), trans=False)]
    return P


class Weights:
    """the fragment stream in consumption order -> ring slot g % NRING; loads are issued in stream order"""
    def __init__(self, p, passes):
        self.p = p
        self.frags = []                 # (pass index, ks, jj, part): part 0 = the (hi) fragment, 1 = its lo fragment (split-weight streams)
        self.index = {}
        for pi, ps in enumerate(passes):
            for ks in range(ps.nks):
                for jj in range(ps.ntile):
                    for part in range(2 if WS else 1):
                        self.index[(pi, ks, jj, part)] = len(self.frags)
                        self.frags.append((pi, ks, jj, part))
        self.passes = passes
        assert len(self.frags) % NRING == 0
        self.tag = {}
        self.next = 0

    def issue(self, upto):
        """issue the loads of fragments < upto (the caller guarantees their ring slots are free).  The stream is periodic: fragment
        g >= len(frags) is fragment g - len(frags) of the workgroup's next tile (same weights), requested by this tile's tail."""
        p = self.p
        while self.next < upto:
            g = self.next
            pi, ks, jj, part = self.frags[g % len(self.frags)]
            ps = self.passes[pi]
            if ks % 4 == 0 and jj == 0 and part == 0:
                for j2 in range(ps.ntile):
                    c = (ps.tiles[j2] * ps.kt + ps.ks0 + ks) * 1024
                    for lo in range(2 if WS else 1):
                        pair = ((S_WBL if j2 else S_WAL) if lo else (S_WB if j2 else S_WA))
                        p.e(f"s_add_u32 {sr(pair)}, %[{ps.mat}_lo], {c + (LO_BYTES[ps.mat] if lo else 0)}")
                        p.e(f"s_addc_u32 {sr(pair + 1)}, %[{ps.mat}_hi], 0")
            pair = ((S_WBL if jj else S_WAL) if part else (S_WB if jj else S_WA))
            self.tag[g] = p.vmem(f"global_load_dwordx4 {ar(4 * (g % NRING), 4)}, {vr(ps.voff)}, {sr(pair, 2)} offset:{(ks % 4) * 1024}")
            self.next += 1


# ------------------------------------------------------------------------------------------------ pieces
def act_addr(ps, ks, Tt):
    if ps.act == "as":
        return V_AS, Tt * 32 * A_ROW + (ps.act_ks0 + ks) * 32
    return V_HS, Tt * 32 * H_ROW + (ps.act_ks0 + ks) * 32


def acc_init_reads(p, ps):
    """bias -> the accumulator registers of pass ps (TRANS tiles): 4 ds_read_b128 per tile; returns the tag of the last read.
    The lane's bias quad of feature tile (tiles[jj] + wmul w): PRM row bytes + wmul * w * 128 + jj * 128 + q * 32 + hh * 16."""
    tag = None
    if ps.wmul == 1:
        p.e(f"v_subrev_u32 {vr(T(12))}, {sr(S_W128)}, {vr(V_PRM)}")         # V_PRM carries w * 256: one tile per wave needs w * 128
    base = T(12) if ps.wmul == 1 else V_PRM
    for jj in range(ps.ntile):
        for Tt in range(2):
            for q in range(4):
                tag = p.lds(f"ds_read_b128 {vr(ps.accs[(jj, Tt)] + 4 * q, 4)}, {vr(base)} offset:{ps.bias + jj * 128 + q * 32}")
    return tag


def xhalf_sum(p, val, tmp):
    """val <- val + (val of the other 32-lane half); 2 wait states between a VALU write and the swap"""
    p.e(f"v_mov_b32 {vr(tmp)}, {vr(val)}")
    p.e("s_nop 1")
    p.e(f"v_permlane32_swap_b32 {vr(tmp)}, {vr(val)}")
    p.e(f"v_add_f32 {vr(val)}, {vr(val)}, {vr(tmp)}")


def gelu_ops(a, t0):
    """instruction list (strings) of GELU in place on register a (bias already in); temps t0 .. t0 + 3.
    gelu(x) = max(x, 0) - (0.5 |x| poly(t) t) exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2)   (erf by Abramowitz-Stegun)"""
    u, t, ex, pl = t0, t0 + 1, t0 + 2, t0 + 3
    ops = [f"v_mul_f32_e64 {vr(u)}, |{vr(a)}|, {sr(S_C1)}",
           f"v_fma_f32 {vr(t)}, {vr(u)}, {sr(S_P)}, 1.0",
           f"v_mul_f32_e64 {vr(ex)}, -{vr(u)}, {vr(u)}",
           f"v_rcp_f32 {vr(t)}, {vr(t)}",
           f"v_exp_f32 {vr(ex)}, {vr(ex)}"]
    coefs = [S_A4, S_A3, S_A2, S_A1] if GELU_TERMS == 5 else [S_A2, S_A1]
    ops.append(f"v_fma_f32 {vr(pl)}, {vr(V_GA)}, {vr(t)}, {sr(coefs[0])}")
    for c in coefs[1:]:
        ops.append(f"v_fma_f32 {vr(pl)}, {vr(pl)}, {vr(t)}, {sr(c)}")
    ops += [f"v_mul_f32 {vr(pl)}, {vr(pl)}, {vr(t)}",
            f"v_mul_f32_e64 {vr(u)}, |{vr(a)}|, {vr(pl)}",            # 0.5 |x| poly(t) t   (0.5 folded into the coefficients)
            f"v_max_f32 {vr(a)}, 0, {vr(a)}",
            f"v_fma_f32 {vr(a)}, -{vr(u)}, {vr(ex)}, {vr(a)}"]
    return ops


def interleave(*lists):
    out = []
    for k in range(max(len(l) for l in lists)):
        for l in lists:
            if k < len(l):
                out.append(l[k])
    return out


def chunk_ops(base, pr):
    """packed dwords of the 8 values acc[8 pr .. 8 pr + 7] -> the 16-byte chunk of (token, 8 features 16 pr + 8 hh ..) in acc[8 pr .. + 3]"""
    b = base + 8 * pr
    return [f"PK {vr(b)}, {vr(b)}, {vr(b + 1)}", f"PK {vr(b + 1)}, {vr(b + 2)}, {vr(b + 3)}",
            f"PK {vr(b + 2)}, {vr(b + 4)}, {vr(b + 5)}", f"PK {vr(b + 3)}, {vr(b + 6)}, {vr(b + 7)}",
            "s_nop 1",
            f"v_permlane32_swap_b32 {vr(b)}, {vr(b + 2)}", f"v_permlane32_swap_b32 {vr(b + 1)}, {vr(b + 3)}"]


def emit_strs(strs):
    def fn(p):
        for s in strs:
            if s.startswith("ds_"):
                p.lds(s)
            elif s.startswith(("buffer_", "global_")):
                p.vmem(s)
            else:
                p.e(s)
    return fn


def gelu_atoms(ps_prev, q):
    """GELU of the finished fc1 quarter (one feature tile, two token tiles) -> Hs columns of hidden tile 4 q + w; T(8) = V_HS + 64 w"""
    atoms = []
    for Tt in range(2):
        base = ps_prev.accs[(0, Tt)]
        atoms.append(lambda p, base=base: p.acc_read(base))
        for pr in range(2):
            for k in range(0, 8, 2):
                a0, a1 = base + 8 * pr + k, base + 8 * pr + k + 1
                if "gelu" in DROP:
                    atoms.append(emit_strs([f"v_max_f32 {vr(a0)}, 0, {vr(a0)}", f"v_max_f32 {vr(a1)}, 0, {vr(a1)}"]))
                else:
                    atoms.append(emit_strs(interleave(gelu_ops(a0, T(0)), gelu_ops(a1, T(4)))))
            off = Tt * 32 * H_ROW + (4 * q * 32 + pr * 16) * 2
            atoms.append(emit_strs(chunk_ops(base, pr) + [f"ds_write_b128 {vr(T(8))}, {vr(base + 8 * pr, 4)} offset:{off}"]))
    return atoms


def qk_atoms(ps_prev, rsrc, scale):
    """q / k of the finished pass -> fragment-ordered operand buffers (two 1-KB lane-linear stores per tile)"""
    atoms = []
    for jj in range(2):
        for Tt in range(2):
            base = ps_prev.accs[(jj, Tt)]
            atoms.append(lambda p, base=base: p.acc_read(base))
            for pr in range(2):
                ops = []
                if scale:
                    ops += [f"v_mul_f32 {vr(base + 8 * pr + k)}, %[qscale], {vr(base + 8 * pr + k)}" for k in range(8)]
                ops += chunk_ops(base, pr)
                if "qkvstore" not in DROP:
                    ops.append(f"buffer_store_dwordx4 {vr(base + 8 * pr, 4)}, {vr(V_QK1 if Tt else V_QK0)}, %[{rsrc}], 0 offen offset:{2048 * jj + 1024 * pr} {STORE_MOD}".rstrip())
                atoms.append(emit_strs(ops))
    return atoms


def run_pass(p, W, pi, ps, atoms, first_act_tags, acc_tag, next_ps=None):
    """the pass' MFMAs with `atoms` (callables) spread over its slots.  first_act_tags: {(ks, T): tag} of the activation fragments already
    requested (ks < 4).  With next_ps (same activation tile): requests its first fragments as this pass' last ones die; returns their tags."""
    act_tag = dict(first_act_tags)
    next_tags = {}
    atoms = list(atoms)
    nslot = ps.nmfma
    done = 0
    slot_i = 0
    nparts = 2 if WS else 1
    for ks in range(ps.nks):
      for jj in range(ps.ntile):
        for part in range(nparts):
            for Tt in range(2):
                g = W.index[(pi, ks, jj, part)]
                p.wait_vm(W.tag.get(g))
                p.wait_lg(act_tag[(ks, Tt)])
                if ks == 0:
                    p.wait_lg(acc_tag)
                base = ps.accs[(jj, Tt)]
                slot = RING((2 * ks + Tt) % 8)
                mfma(p, ps, base, g, slot)
                if Tt == 1 and "wload" not in DROP:
                    W.issue(min(g + NRING + 1, W.next + 2))  # fragment g is dead: its slot may take fragment g + NRING (at most two requests per
                                                             # death: the ring fills up over the first passes instead of in one burst at the start)
                if jj == ps.ntile - 1 and part == nparts - 1:   # activation fragment (ks, T) is dead: its slot takes (ks + 4, T)
                    if ks + 4 < ps.nks:
                        va, off = act_addr(ps, ks + 4, Tt)
                        act_tag[(ks + 4, Tt)] = p.lds(f"ds_read_b128 {vr(slot, 4)}, {vr(va)} offset:{off}")
                    elif next_ps is not None:
                        va, off = act_addr(next_ps, ks + 4 - ps.nks, Tt)
                        next_tags[(ks + 4 - ps.nks, Tt)] = p.lds(f"ds_read_b128 {vr(slot, 4)}, {vr(va)} offset:{off}")
                slot_i += 1
                want = (len(atoms) + done) * slot_i // nslot     # atoms spread evenly over the slots
                while done < want and atoms:
                    atoms.pop(0)(p)
                    done += 1
    while atoms:
        atoms.pop(0)(p)
    p.stamp(ps.name)
    return next_tags


def mfma(p, ps, base, g, slot):
    wreg = ar(4 * (g % NRING), 4)
    if "mfma" in DROP:
        pass
    elif ps.trans:
        p.e(f"MFMA {vr(base, 16)}, {wreg}, {vr(slot, 4)}, {vr(base, 16)}")
    else:
        p.e(f"MFMA {vr(base, 16)}, {vr(slot, 4)}, {wreg}, {vr(base, 16)}")
    p.mfma_at[base] = p.n


def first_act_reads(p, ps):
    tags = {}
    for ks in range(4):
        for Tt in range(2):
            va, off = act_addr(ps, ks, Tt)
            tags[(ks, Tt)] = p.lds(f"ds_read_b128 {vr(RING((2 * ks + Tt) % 8), 4)}, {vr(va)} offset:{off}")
    return tags


def x_lds_addr(p, dst):
    p.e(f"v_add_u32 {vr(dst)}, {sr(S_W256)}, {vr(V_HS)}")           # XS + i * 1040 + hh * 16 + w * 256 (XS = HS)


def x_read_atoms(addr):
    """the residual rows, staged in XS by LDS-DMA, into the accumulator layout"""
    atoms = []
    for j in range(2):
        for Tt in range(2):
            atoms.append(emit_strs([f"ds_read_b128 {vr(X(j, Tt) + 4 * q, 4)}, {vr(addr)} offset:{Tt * 32 * X_ROW + j * 128 + q * 32}" for q in range(4)]))
    return atoms


def x_load_pieces(nxt=False, lane16=None):
    """rows 16 w .. 16 w + 15 of a tile as whole 1-KB rows (row clamped to N - 1: the padding rows are finite duplicates) into the
    staging quads a[160:223] (one LDS-DMA per row was tried first: the M0 write in front of each serialises them, ~300 cycles apiece).
    nxt: the tile this workgroup takes next (%[n0_n], %[rx_n]; a descriptor of zero records when there is none: the loads return 0).
    Returns [setup, load 0, .., load 15] as callables."""
    sfx = "_n" if nxt else ""
    lane16 = T(14) if lane16 is None else lane16
    def setup(p):
        p.e(f"v_and_b32 {vr(lane16)}, 63, %[tid]")
        p.e(f"v_lshlrev_b32 {vr(lane16)}, 4, {vr(lane16)}")
    def one(p, k):
        p.e(f"s_lshl_b32 {sr(S_TMP)}, %[w], 4")
        p.e(f"s_add_u32 {sr(S_TMP)}, %[n0{sfx}], {sr(S_TMP)}")
        p.e(f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP)}, {k}")
        p.e(f"s_min_u32 {sr(S_TMP)}, {sr(S_TMP)}, %[nm1]")
        p.e(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 10")
        p.vmem(f"buffer_load_dwordx4 {ar(AG_X + 4 * k, 4)}, {vr(lane16)}, %[rx{sfx}], {sr(S_TMP)} offen")
    return [setup] + [lambda p, k=k: one(p, k) for k in range(16)]


def x_loads(p, nxt=False, lane16=None):
    for f in x_load_pieces(nxt, lane16):
        f(p)


def x_stage_writes(p):
    """... and on into XS"""
    p.e(f"v_and_b32 {vr(T(14))}, 63, %[tid]")
    p.e(f"v_lshlrev_b32 {vr(T(14))}, 4, {vr(T(14))}")
    p.e(f"v_add_u32 {vr(T(13))}, {sr(S_XSB)}, {vr(T(14))}")
    for k in range(16):
        p.lds(f"ds_write_b128 {vr(T(13))}, {ar(AG_X + 4 * k, 4)} offset:{k * X_ROW}")


def x_store_write_atoms(addr):
    """X registers -> XS in the accumulator layout (the caller has made sure every wave is done with the bytes); `addr` = x_lds_addr"""
    return [emit_strs([f"ds_write_b128 {vr(addr)}, {vr(X(j, Tt) + 4 * q, 4)} offset:{Tt * 32 * X_ROW + j * 128 + q * 32}" for Tt in range(2)])
            for j in range(2) for q in range(4)]


def x_store_write(p):
    x_lds_addr(p, T(15))
    for a in x_store_write_atoms(T(15)):
        a(p)


def x_store_atoms():
    """rows 16 w .. 16 w + 15 from XS to memory as whole 1-KB rows (rows >= N are out of the descriptor's range: dropped).
    Temporaries: T(0..7) two quads, T(12) lane * 16, T(13) its LDS address."""
    atoms = []
    def setup(p):
        p.e(f"v_and_b32 {vr(T(12))}, 63, %[tid]")
        p.e(f"v_lshlrev_b32 {vr(T(12))}, 4, {vr(T(12))}")
        p.e(f"v_add_u32 {vr(T(13))}, {sr(S_XSB)}, {vr(T(12))}")
    atoms.append(setup)
    tg = {}
    for k in range(17):                  # row k is read one atom before it is stored (two quads alternate)
        def one(p, k=k):
            if k < 16:
                tg[k] = p.lds(f"ds_read_b128 {vr(T(4 * (k % 2)), 4)}, {vr(T(13))} offset:{k * X_ROW}")
            if k >= 1:
                r = k - 1
                p.e(f"s_add_u32 {sr(S_TMP2)}, {sr(S_N0W)}, {r}")
                p.e(f"s_lshl_b32 {sr(S_TMP2)}, {sr(S_TMP2)}, 10")
                p.wait_lg(tg[r])
                if "xstore" not in DROP:
                    p.vmem(f"buffer_store_dwordx4 {vr(T(4 * (r % 2)), 4)}, {vr(T(12))}, %[rx], {sr(S_TMP2)} offen {STORE_MOD}".rstrip())
        atoms.append(one)
    return atoms


def residual(p, acc_base, gate_row, tiles=(0, 1)):
    """X += gate * acc (the bias is already in acc) for the token tiles `tiles`"""
    G = [T(0), T(4)]                     # two gate quads alternate
    gt = {}
    seq = [(j, q) for j in range(2) for q in range(4)]
    for n, (j, q) in enumerate(seq[:2]):
        gt[(j, q)] = p.lds(f"ds_read_b128 {vr(G[n & 1], 4)}, {vr(V_PRM)} offset:{gate_row * 1024 + j * 128 + q * 32}")
    for n, (j, q) in enumerate(seq):
        p.wait_lg(gt[(j, q)])
        for Tt in tiles:
            a = acc_base + 32 * j + 16 * Tt
            p.acc_read(a)
            for e in range(4):
                r = 4 * q + e
                p.e(f"v_fma_f32 {vr(X(j, Tt) + r)}, {vr(G[n & 1] + e)}, {vr(a + r)}, {vr(X(j, Tt) + r)}")
        if n + 2 < len(seq):
            j2, q2 = seq[n + 2]
            gt[(j2, q2)] = p.lds(f"ds_read_b128 {vr(G[n & 1], 4)}, {vr(V_PRM)} offset:{gate_row * 1024 + j2 * 128 + q2 * 32}")


def wave_stats(p, Tt):
    """per-wave LayerNorm statistics of token tile Tt: mean over the wave's 64 features, M2 = sum of squared deviations from it -> ST"""
    p.e(f"v_add_u32 {vr(T(15))}, {sr(S_W8)}, {vr(V_ST)}")
    s0, s1, tmp = T(0), T(1), T(2)
    regs = [X(j, Tt) + r for j in range(2) for r in range(16)]
    p.e(f"v_add_f32 {vr(s0)}, {vr(regs[0])}, {vr(regs[1])}")
    p.e(f"v_add_f32 {vr(s1)}, {vr(regs[2])}, {vr(regs[3])}")
    for k in range(4, 32, 2):
        p.e(f"v_add_f32 {vr(s0)}, {vr(s0)}, {vr(regs[k])}")
        p.e(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(regs[k + 1])}")
    p.e(f"v_add_f32 {vr(s0)}, {vr(s0)}, {vr(s1)}")
    xhalf_sum(p, s0, tmp)
    mean, m2a, m2b, d0, d1 = T(4), T(5), T(3), T(6), T(7)      # (mean, M2) = T(4), T(5): consecutive for the ds_write_b64
    p.e(f"v_mul_f32 {vr(mean)}, {f32(1.0 / 64)}, {vr(s0)}")
    p.e(f"v_mov_b32 {vr(m2a)}, 0")
    p.e(f"v_mov_b32 {vr(m2b)}, 0")
    for k in range(0, 32, 2):
        p.e(f"v_sub_f32 {vr(d0)}, {vr(regs[k])}, {vr(mean)}")
        p.e(f"v_sub_f32 {vr(d1)}, {vr(regs[k + 1])}, {vr(mean)}")
        p.e(f"v_fmac_f32 {vr(m2a)}, {vr(d0)}, {vr(d0)}")
        p.e(f"v_fmac_f32 {vr(m2b)}, {vr(d1)}, {vr(d1)}")
    p.e(f"v_add_f32 {vr(m2a)}, {vr(m2a)}, {vr(m2b)}")
    xhalf_sum(p, m2a, tmp)
    p.lds(f"ds_write_b64 {vr(T(15))}, {vr(mean, 2)} offset:{Tt * 1024}")
    p.e("s_nop 1")


def layernorm(p, SCR, ln_shift_row, ln_scale_row, between=None, stats_tiles=(0, 1)):
    """LayerNorm statistics of the wave's 64 features -> ST (token tiles `stats_tiles`: the others' are there already); barrier; combine
    the 4 waves (Chan); LayerNorm + modulate -> As (16-bit chunks) [the finished residual rows go to XS between its steps]; barrier.
    SCR: 64 scratch registers (a consumed accumulator set)."""
    for Tt in stats_tiles:
        wave_stats(p, Tt)
    p.barrier("stats")
    # ---- all four waves' partials of the lane's token -> mean, rstd (Chan's combination, equal counts)
    RS, CC = [T(8), T(10)], [T(9), T(11)]
    stt = {}
    for Tt in range(2):
        for h in range(2):
            stt[(Tt, h)] = p.lds(f"ds_read_b128 {vr(SCR + 8 * Tt + 4 * h, 4)}, {vr(V_ST)} offset:{Tt * 1024 + h * 16}")
    for Tt in range(2):
        m = [SCR + 8 * Tt + 2 * k for k in range(4)]          # means at even, M2 at odd registers
        p.wait_lg(stt[(Tt, 1)])
        mean, dv, d = T(0), T(1), T(2)
        p.e(f"v_add_f32 {vr(mean)}, {vr(m[0])}, {vr(m[1])}")
        p.e(f"v_add_f32 {vr(d)}, {vr(m[2])}, {vr(m[3])}")
        p.e(f"v_add_f32 {vr(mean)}, {vr(mean)}, {vr(d)}")
        p.e(f"v_mul_f32 {vr(mean)}, {f32(0.25)}, {vr(mean)}")
        p.e(f"v_add_f32 {vr(dv)}, {vr(m[0] + 1)}, {vr(m[1] + 1)}")
        p.e(f"v_add_f32 {vr(d)}, {vr(m[2] + 1)}, {vr(m[3] + 1)}")
        p.e(f"v_add_f32 {vr(dv)}, {vr(dv)}, {vr(d)}")           # sum of the waves' M2
        p.e(f"v_mov_b32 {vr(T(3))}, 0")
        for k in range(4):
            p.e(f"v_sub_f32 {vr(d)}, {vr(m[k])}, {vr(mean)}")
            p.e(f"v_fmac_f32 {vr(T(3))}, {vr(d)}, {vr(d)}")
        p.e(f"v_fmac_f32 {vr(dv)}, {sr(S_C64)}, {vr(T(3))}")    # + 64 * sum (mean_w - mean)^2
        p.e(f"v_mul_f32 {vr(dv)}, {f32(1.0 / 256)}, {vr(dv)}")
        p.e(f"v_add_f32 {vr(dv)}, {sr(S_EPS)}, {vr(dv)}")
        p.e(f"v_rsq_f32 {vr(RS[Tt])}, {vr(dv)}")
        p.e("s_nop 0")
        p.e(f"v_mul_f32_e64 {vr(CC[Tt])}, -{vr(mean)}, {vr(RS[Tt])}")
    extra = []
    if between:
        x_lds_addr(p, T(14))
        extra = x_store_write_atoms(T(14))       # one per step of the loop below: the writes' LDS time hides under the arithmetic
    # ---- y = ((x - mean) rstd) (1 + scale) + shift -> 16-bit, chunks of 8 features -> As
    p.e(f"v_add_u32 {vr(T(15))}, {sr(S_W128)}, {vr(V_AS)}")
    PQ = [SCR + 16, SCR + 24]               # (1 + scale | shift) quads of a feature quad, two sets alternate
    Y = [SCR + 32, SCR + 48]                # 8 values of (j, pr) per token tile
    seq = [(j, q) for j in range(2) for q in range(4)]
    pt = {}
    def prm_reads(n):
        j, q = seq[n]
        p.lds(f"ds_read_b128 {vr(PQ[n & 1], 4)}, {vr(V_PRM)} offset:{ln_scale_row * 1024 + j * 128 + q * 32}")
        pt[n] = p.lds(f"ds_read_b128 {vr(PQ[n & 1] + 4, 4)}, {vr(V_PRM)} offset:{ln_shift_row * 1024 + j * 128 + q * 32}")
    prm_reads(0)
    prm_reads(1)
    for n, (j, q) in enumerate(seq):
        p.wait_lg(pt[n])
        for Tt in range(2):
            for e in range(4):
                y = Y[Tt] + 4 * (q & 1) + e
                p.e(f"v_fma_f32 {vr(y)}, {vr(X(j, Tt) + 4 * q + e)}, {vr(RS[Tt])}, {vr(CC[Tt])}")
            for e in range(4):
                y = Y[Tt] + 4 * (q & 1) + e
                p.e(f"v_fma_f32 {vr(y)}, {vr(y)}, {vr(PQ[n & 1] + e)}, {vr(PQ[n & 1] + 4 + e)}")
        if n + 2 < len(seq):
            prm_reads(n + 2)
        if extra:
            extra.pop(0)(p)
        if q & 1:
            pr = q >> 1
            for Tt in range(2):
                for s in chunk_ops(Y[Tt], 0):
                    p.e(s)
                p.lds(f"ds_write_b128 {vr(T(15))}, {vr(Y[Tt], 4)} offset:{Tt * 32 * A_ROW + j * 64 + pr * 32}")
            p.e("s_nop 1")
    p.barrier("LN")


def lane_consts(p):
    """lane constants (the statement's only vector input is the thread index)"""
    lane, i_, hh = T(0), T(1), T(2)
    p.e(f"v_and_b32 {vr(lane)}, 63, %[tid]")
    p.e(f"v_and_b32 {vr(i_)}, 31, {vr(lane)}")
    p.e(f"v_lshrrev_b32 {vr(hh)}, 5, {vr(lane)}")
    p.e(f"s_lshl_b32 {sr(S_W8)}, %[w], 3")
    p.e(f"s_lshl_b32 {sr(S_W64)}, %[w], 6")
    p.e(f"s_lshl_b32 {sr(S_W128)}, %[w], 7")
    p.e(f"s_lshl_b32 {sr(S_W256)}, %[w], 8")
    p.e(f"s_lshl_b32 {sr(S_TMP)}, %[w], 14")
    p.e(f"v_lshlrev_b32 {vr(V_OFFQ)}, 4, {vr(lane)}")
    p.e(f"v_add_u32 {vr(V_OFFQ)}, {sr(S_TMP)}, {vr(V_OFFQ)}")                       # lane * 16 + w * 16384
    p.e(f"v_add_u32 {vr(V_OFF16)}, {sr(S_TMP)}, {vr(V_OFFQ)}")                      # lane * 16 + w * 32768
    p.e(f"s_lshl_b32 {sr(S_TMP)}, %[w], 15")
    p.e(f"v_add_u32 {vr(V_OFF32)}, {sr(S_TMP)}, {vr(V_OFF16)}")                     # lane * 16 + w * 65536
    p.e(f"v_mul_u32_u24 {vr(V_AS)}, {A_ROW}, {vr(i_)}")
    p.e(f"v_lshl_add_u32 {vr(V_AS)}, {vr(hh)}, 4, {vr(V_AS)}")                      # AS (= 0) + i * 528 + hh * 16
    p.e(f"v_mul_u32_u24 {vr(V_HS)}, {H_ROW}, {vr(i_)}")
    p.e(f"v_lshl_add_u32 {vr(V_HS)}, {vr(hh)}, 4, {vr(V_HS)}")
    p.e(f"v_add_u32 {vr(V_HS)}, {HS}, {vr(V_HS)}")
    p.e(f"v_lshlrev_b32 {vr(V_PRM)}, 4, {vr(hh)}")
    p.e(f"v_add_u32 {vr(V_PRM)}, {sr(S_W256)}, {vr(V_PRM)}")
    p.e(f"v_add_u32 {vr(V_PRM)}, {PRM}, {vr(V_PRM)}")                                # PRM + hh * 16 + w * 256
    p.e(f"s_lshr_b32 {sr(S_TMP)}, %[n0], 5")
    p.e(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 13")
    p.e(f"v_lshlrev_b32 {vr(V_QK0)}, 4, {vr(lane)}")
    p.e(f"v_add_u32 {vr(V_QK0)}, {sr(S_TMP)}, {vr(V_QK0)}")
    p.e(f"s_and_b32 {sr(S_TMP)}, %[w], 1")
    p.e(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 12")
    p.e(f"v_add_u32 {vr(V_QK0)}, {sr(S_TMP)}, {vr(V_QK0)}")                         # (n0 / 32) * 8192 + lane * 16 + (w & 1) * 4096
    p.e(f"v_add_u32 {vr(V_QK1)}, 8192, {vr(V_QK0)}")
    p.e(f"v_lshlrev_b32 {vr(V_ST)}, 5, {vr(i_)}")
    p.e(f"v_add_u32 {vr(V_ST)}, {ST}, {vr(V_ST)}")                                  # ST + i * 32
    p.e(f"s_lshl_b32 {sr(S_TMP)}, %[w], 4")
    p.e(f"s_add_u32 {sr(S_N0W)}, %[n0], {sr(S_TMP)}")                               # first row of this wave's 16 residual rows
    p.e(f"s_mul_i32 {sr(S_XSB)}, {sr(S_TMP)}, {X_ROW}")
    p.e(f"s_add_u32 {sr(S_XSB)}, {sr(S_XSB)}, {XS}")                                # XS + 16 w * 1040


def o_load_pieces(nxt=False, tmp=None):
    """O rows: thread (tid >> 5, tid & 31) takes the 16-byte chunk tid & 31 of rows (tid >> 5) + 8 m (clamped to N - 1) into a[128 + 4 m ..].
    Returns [setup, load 0, .., load 7] as callables."""
    sfx = "_n" if nxt else ""
    ol, orow, och = tmp or (T(12), T(13), T(15))
    def setup(p):
        p.e(f"v_lshrrev_b32 {vr(orow)}, 5, %[tid]")
        p.e(f"v_and_b32 {vr(och)}, 31, %[tid]")
        p.e(f"v_lshlrev_b32 {vr(och)}, 4, {vr(och)}")
        p.e(f"v_add_u32 {vr(orow)}, %[n0{sfx}], {vr(orow)}")
    def one(p, m):
        p.e(f"v_add_u32 {vr(ol)}, {8 * m}, {vr(orow)}")
        p.e(f"v_min_u32 {vr(ol)}, %[nm1], {vr(ol)}")
        p.e(f"v_lshl_add_u32 {vr(ol)}, {vr(ol)}, 9, {vr(och)}")                  # row * 512 + chunk * 16
        p.vmem(f"buffer_load_dwordx4 {ar(AG_O + 4 * m, 4)}, {vr(ol)}, %[ro{sfx}], 0 offen")
    return [setup] + [lambda p, m=m: one(p, m) for m in range(8)]


def o_loads(p, nxt=False, tmp=None):
    for f in o_load_pieces(nxt, tmp):
        f(p)


def prefetch_atoms(variant):
    """the requests of the workgroup's NEXT tile (O rows, residual rows), ONE request per atom: a burst of 24 HBM reads fills the CU's
    request queue and blocks the wave at its next weight request (4k cycles wherever the burst was put); spread thin they do not"""
    atoms = []
    if variant != "QKV":
        atoms += o_load_pieces(True, (T(9), T(10), T(11)))        # (T(8) is the GELU's Hs address, T(12) the bias address)
    atoms += x_load_pieces(True, T(13))
    return atoms


def spread(main, extra):
    """`extra` atoms dealt evenly into the list `main`"""
    if not extra:
        return list(main)
    out, n, m = [], len(main), len(extra)
    ei = 0
    for k, a in enumerate(main):
        out.append(a)
        while ei < m and (ei + 1) * n <= (k + 1) * m:
            out.append(extra[ei]); ei += 1
    out += extra[ei:]
    return out


def prologue(variant):
    """the FIRST statement of a workgroup: the requests of its first tile that depend on nothing - the O rows, the residual rows, the
    first NRING weight fragments - go out before the caller stages the parameter rows (ONE memory round trip instead of two).  The
    destinations (accumulation file) stay untouched by the code between the statements (tools/audit_rowchain_a.py checks the build)."""
    p = Prog()
    W = Weights(p, passes_of(variant))
    p.e(f"s_lshl_b32 {sr(S_TMP)}, %[w], 14")
    p.e(f"v_and_b32 {vr(T(0))}, 63, %[tid]")
    p.e(f"v_lshlrev_b32 {vr(V_OFFQ)}, 4, {vr(T(0))}")
    p.e(f"v_add_u32 {vr(V_OFFQ)}, {sr(S_TMP)}, {vr(V_OFFQ)}")
    p.e(f"v_add_u32 {vr(V_OFF16)}, {sr(S_TMP)}, {vr(V_OFFQ)}")
    if variant != "QKV":
        o_loads(p)
    W.issue(8)
    x_loads(p)
    W.issue(NRING)
    return p


def core(variant):
    p = Prog()
    passes = passes_of(variant)
    W = Weights(p, passes)
    W.next = NRING                       # requested by the first statement / by the previous tile's tail; everything has landed (the wait below)
    idx = {ps.name: i for i, ps in enumerate(passes)}
    p.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    p.e("s_barrier")                     # every wave is done with the previous tile's LDS (As, XS)
    if TIMING:
        p.e("v_mov_b32 v251, 0")
    lane_consts(p)
    # ---- scalar constants
    c1 = (0.5 * 1.4426950408889634) ** 0.5                  # u = |x| c1: u^2 = (x^2 / 2) log2 e
    if GELU_TERMS == 5:
        pz, co = 0.3275911, [0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429]
    else:
        pz, co = 0.47047, [0.3480242, -0.0958798, 0.7478556]
    p.e(f"s_mov_b32 {sr(S_C1)}, {f32(c1)}")
    p.e(f"s_mov_b32 {sr(S_P)}, {f32(pz * 0.7071067811865476 / c1)}")
    for k, c in enumerate(co[:-1]):
        p.e(f"s_mov_b32 {sr(S_A1 + k)}, {f32(0.5 * c)}")
    p.e(f"v_mov_b32 {vr(V_GA)}, {f32(0.5 * co[-1])}")
    p.e(f"s_mov_b32 {sr(S_C64)}, {f32(64.0)}")
    p.e(f"s_mov_b32 {sr(S_EPS)}, {f32(1e-6)}")
    p.stamp("start")

    if variant == "QKV":
        x_stage_writes(p)
        p.barrier("x staged")                # the residual rows and the parameter rows (written by the statement's caller) are visible
        x_lds_addr(p, T(13))
        for a in x_read_atoms(T(13)):
            a(p)
        p.wait_lg(p.lg_issued)
        layernorm(p, A1, P_SHN, P_SC1N)
    else:
        # ---- O rows (in a[128:159] since the first statement / the previous tile's q pass) -> As
        ow, orow, och = T(5), T(3), T(6)
        p.e(f"v_lshrrev_b32 {vr(orow)}, 5, %[tid]")
        p.e(f"v_and_b32 {vr(och)}, 31, %[tid]")
        p.e(f"v_mul_u32_u24 {vr(ow)}, {A_ROW}, {vr(orow)}")
        p.e(f"v_lshl_add_u32 {vr(ow)}, {vr(och)}, 4, {vr(ow)}")
        for m in range(8):
            p.lds(f"ds_write_b128 {vr(ow)}, {ar(AG_O + 4 * m, 4)} offset:{m * 8 * A_ROW}")
        x_stage_writes(p)                    # the residual rows (a[160:223], requested by the first statement / the previous tile) -> XS
        acc1 = acc_init_reads(p, passes[0])
        p.barrier("stage O")
        # ---- proj (set 0); the residual rows come back from XS in the accumulator layout meanwhile
        tags = first_act_reads(p, passes[0])
        atoms = [lambda p: x_lds_addr(p, T(13))] + x_read_atoms(T(13)) + [lambda p: None] * 3
        run_pass(p, W, 0, passes[0], atoms, tags, acc1)
        p.wait_lg(p.lg_issued)
        # (token-tile-major order - all of T0's MFMAs, then T1's with T0's residual + statistics in their gaps - was built and measured for
        # this pass and for fc2.3: the epilogues shrank by 0.6k / 0.8k cycles, the passes grew by 1.5k / 0.8k; dropped)
        residual(p, A0, P_GMSA)
        layernorm(p, A0, P_SHM, P_SC1M)
        # ---- fc1 quarters 0..3 (set 0 halves), GELU of quarter q - 1 under quarter q; fc2's accumulators (set 1) take b2 meanwhile
        init = {}
        tags = first_act_reads(p, passes[idx["fc1.0"]])
        init[0] = acc_init_reads(p, passes[idx["fc1.0"]])
        pf = prefetch_atoms(variant)
        for q in range(4):
            pi = idx[f"fc1.{q}"]
            atoms = []
            if q == 0:
                atoms.append(emit_strs([f"v_add_u32 {vr(T(8))}, {sr(S_W64)}, {vr(V_HS)}"]))
                atoms.append(lambda p: init.__setitem__(1, acc_init_reads(p, passes[idx["fc1.1"]])))
                atoms.append(lambda p: init.__setitem__("fc2", acc_init_reads(p, passes[idx["fc2.0"]])))
            else:
                # the next tile's rows are requested HERE, one at a time: vmcnt retires in order, so every weight fragment requested after one of
                # these HBM reads waits for it - in the GELU-bound quarters the ring's 32 fragments last ~6k cycles, which covers the latency
                # (requested in the q pass, MFMA-bound, they stalled it for 5k cycles)
                n3 = (len(pf) + 2) // 3
                atoms += spread(gelu_atoms(passes[pi - 1], q - 1), pf[(q - 1) * n3:q * n3])
                if q < 3:
                    atoms.append(lambda p, q=q: init.__setitem__(q + 1, acc_init_reads(p, passes[idx[f"fc1.{q + 1}"]])))
            tags = run_pass(p, W, pi, passes[pi], atoms, tags, init[q], passes[pi + 1] if q < 3 else None)
        p.barrier("B3")                      # Hs[:, 0:384] complete
        g3 = gelu_atoms(passes[idx["fc1.3"]], 3)
        third = (len(g3) + 2) // 3
        for c in range(4):
            pi = idx[f"fc2.{c}"]
            atoms = g3[c * third:(c + 1) * third] if c < 3 else []
            if c == 3 and variant == "FULL":
                atoms = [lambda p: init.__setitem__("q", acc_init_reads(p, passes[idx["q"]]))]
            tags = first_act_reads(p, passes[pi])
            run_pass(p, W, pi, passes[pi], atoms, tags, init["fc2"] if c == 0 else None)
            if c == 2:
                p.barrier("B4")              # Hs[:, 384:512] complete
        p.wait_lg(p.lg_issued)
        residual(p, A1, P_GMLP)
        if variant == "FULL":
            layernorm(p, A1, P_SHN, P_SC1N, between=True)
        else:
            p.barrier("all done with Hs")
            x_store_write(p)
            p.barrier("x2 staged")
            for a in x_store_atoms():
                a(p)
    if variant != "LAST":
        # ---- q (set 0) [+ the finished residual rows leave], k (set 1) + q stores, v^T (set 0, plain product) + k stores, v^T stores
        pq, pk, pv = idx["q"], idx["k"], idx["v"]
        tq = init["q"] if variant == "FULL" else acc_init_reads(p, passes[pq])
        tags = first_act_reads(p, passes[pq])
        init_k = {}
        atoms = [lambda p: init_k.__setitem__("t", acc_init_reads(p, passes[pk]))]
        xs_atoms = x_store_atoms() if variant == "FULL" else []
        if variant == "QKV":
            atoms += prefetch_atoms(variant)
        atoms += xs_atoms[:XS_SPLIT[0]]
        nt = run_pass(p, W, pq, passes[pq], atoms, tags, tq, passes[pk])
        # v^T bias: the lane's feature (32 (16 + 2 w + j) + i) for all 16 registers of a tile; set 0 is free once its q tile has left
        def vbias(p):
            p.e(f"v_and_b32 {vr(T(9))}, 31, %[tid]")
            p.e(f"v_lshlrev_b32 {vr(T(9))}, 2, {vr(T(9))}")
            p.e(f"v_add_u32 {vr(T(9))}, {sr(S_W256)}, {vr(T(9))}")
            p.e(f"v_add_u32 {vr(T(9))}, {PRM}, {vr(T(9))}")
            init_k["b0"] = p.lds(f"ds_read_b32 {vr(T(10))}, {vr(T(9))} offset:{P_BQ + 2048}")
            init_k["b1"] = p.lds(f"ds_read_b32 {vr(T(11))}, {vr(T(9))} offset:{P_BQ + 2048 + 128}")
        def vinit(p):
            p.wait_lg(init_k["b1"])
            for j in range(2):
                for Tt in range(2):
                    for r in range(16):
                        p.e(f"v_mov_b32 {vr(passes[pv].accs[(j, Tt)] + r)}, {vr(T(10 + j))}")
        qa = qk_atoms(passes[pq], "rq", True)
        atoms = [vbias] + interleave(qa, xs_atoms[XS_SPLIT[0]:XS_SPLIT[0] + XS_SPLIT[1]]) + [vinit]
        nt2 = run_pass(p, W, pk, passes[pk], atoms, nt, init_k["t"], passes[pv])
        run_pass(p, W, pv, passes[pv], interleave(qk_atoms(passes[pk], "rk", False), xs_atoms[XS_SPLIT[0] + XS_SPLIT[1]:]), nt2, None)
        for j in range(2):
            for Tt in range(2):
                base = passes[pv].accs[(j, Tt)]
                p.acc_read(base)
                for half in range(2):
                    for k in range(4):
                        p.e(f"PK {vr(base + 8 * half + k)}, {vr(base + 8 * half + 2 * k)}, {vr(base + 8 * half + 2 * k + 1)}")
                    p.vmem(f"buffer_store_dwordx4 {vr(base + 8 * half, 4)}, {vr(V_QK1 if Tt else V_QK0)}, %[rv], 0 offen offset:{2048 * j + 1024 * half} {STORE_MOD}".rstrip())
    assert W.next == len(W.frags) + NRING, (W.next, len(W.frags))      # the next tile finds its first NRING fragments requested
    p.e("s_nop 1")
    p.stamp("end")
    return p


def clobbers():
    items = [f'"v{i}"' for i in range(V_TOP + 1)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(60, S_TOP + 1)]
    items += ['"vcc"', '"scc"', '"memory"']
    lines, cur = [], ""
    for it in items:
        if len(cur) + len(it) > 120:
            lines.append(cur); cur = ""
        cur += it + ", "
    lines.append(cur.rstrip(", "))
    return " \\\n    ".join(lines)


def main():
    global WS, GELU_TERMS
    parts = ["// GENERATED by tools/gen_rowchain_a.py - do not edit (the generator holds the register map, the schedule and the wait counts).\n"
             "// Instruction streams of the 64-row DiT row chain (dit_rowchain64a_kernel, dit_rowchain.hip); RCA_MFMA / RCA_PK are the\n"
             "// mnemonics of the operand type (bf16 / fp16 build).\n",
             f"#define RCA_LDS_BYTES {LDS_BYTES}\n#define RCA_LDS_AS {AS}\n#define RCA_LDS_HS {HS}\n#define RCA_LDS_PRM {PRM}\n#define RCA_LDS_ST {ST}\n"
             f"#define RCA_A_ROW {A_ROW}\n#define RCA_GELU_TERMS {GELU_TERMS}\n"]
    info = []
    gelu_default = GELU_TERMS
    for ws in (False, True):
        # the split-weight build (namespace dex::f16w: -DDEX_LP_WSPLIT) gets streams of its own under the same macro names.  Its GELU is the
        # five-term erf (7.1.26, |err| 1.5e-7 - the formula of the C++ kernels): the mode exists to sit at the fp32 reference's 1e-4, the
        # three-term form's 2.5e-5 is a quarter of that budget, and with twice the MFMA slots the two extra instructions per value hide
        WS = ws
        GELU_TERMS = int(os.environ.get("RCAGEN_GELU_WS", "5")) if ws else gelu_default
        parts.append("#ifdef DEX_LP_WSPLIT\n" if ws else "#ifndef DEX_LP_WSPLIT\n")
        for name, variant in (("RCA_ASM_PRE", "FULL"), ("RCA_ASM_PRE_QKV", "QKV")):
            body = prologue(variant).text().replace("\n", " \\\n")
            parts.append(f"#define {name} \\\n    {body}\n")
        for name, variant in (("RCA_ASM_FULL", "FULL"), ("RCA_ASM_LAST", "LAST"), ("RCA_ASM_QKV", "QKV")):
            p = core(variant)
            body = p.text().replace("\n", " \\\n")
            parts.append(f"#define {name} \\\n    {body}\n")
            info.append(f"{'split-weight ' if ws else ''}{variant}: {p.stats} max VMEM in flight (upper bound) {getattr(p, 'max_vm', 0)}, forced waits {getattr(p, 'forced_vm_waits', 0)}")
            if TIMING:
                parts.append(f"#define {name}_STAMPS " + ", ".join(f'"{n}"' for n in p.names) + "\n")
        parts.append("#endif\n")
    WS = False
    GELU_TERMS = gelu_default
    parts.append("#define RCA_CLOBBER \\\n    " + clobbers() + "\n")
    text = "\n".join(parts)
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {os.path.normpath(OUT)}: {len(text)} bytes")
    for i in info:
        print("  ", i)


if __name__ == "__main__":
    main()
