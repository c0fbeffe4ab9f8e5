#!/usr/bin/env python3
"""Generator of dex_tts_amd/csrc/attention_q64_core.inc — the hand-scheduled gfx950 instruction streams of the
64-queries-per-wave attention (attention_q64.hip): the per-unit core (first tile, software-pipelined tile loop, last tile),
the unit prologue (LDS-DMA of the first tiles + Q straight into the accumulation file) and the output stage.

Why a generator: the core owns the register file — fixed physical registers, every MFMA gap's fillers placed by hand.  As
inline asm with compiler-allocated operands (the first form of this kernel) hipcc copied 16-register score tuples around
element updates, spilled the Q fragments to scratch once the arch VGPRs ran out, padded statements with s_nop and moved
plain C++ fillers out of their gaps; as one statement with a fixed register map none of that can happen.

    python tools/gen_attn_q64.py            # rewrites the .inc (committed; tests/test_cabi.py checks it is up to date)

Round 6: the shipped streams are the v2 ones (class StreamV2 below: one generator for whole units - 64 queries per wave - and half
units - 32 -, plus passive_v2 for waves without a live query block; the loop is unrolled over the four ring slots, see the comment
there).  The v1 functions (core / iteration / core_h ...) are kept for A/B builds: Q64GEN_V1=1.

Register map (core; v2 adds v[184:187] row-sum accumulators, v188 the V^T ring's read base, fragment sets a[192:255]):
  a[0:127]   O^T accumulators: block A d-tile td = a[16 td ..], block B = a[64 + 16 td ..]
  a[128:191] Q fragments: QA[s] = a[128 + 4 s ..], QB[s] = a[160 + 4 s ..]
  a[192:223] K / V^T fragment staging: fr[set][q] = a[192 + 16 set + 4 q ..]
  v[0:63]    score buffer 0: array a = 2 X + kb (query block X, key block kb) = v[16 a ..];  v[64:127] score buffer 1
  v[128:143] -m seeds of block A, v[144:159] of block B
  v160.. scalars (row sums, reference maxima, running maxima, addresses)
"""
import os
import sys

OUT = os.environ.get("Q64GEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dex_tts_amd", "csrc", "attention_q64_core.inc")
# ---- experiment switches (tools/attnq64 A/B builds: Q64GEN_OUT=/tmp/x.inc Q64GEN_DMA=... python tools/gen_attn_q64.py, then
# -DQ64_CORE_INC='"/tmp/x.inc"').  The committed .inc is generated with none of them set.
OPT_DMA = os.environ.get("Q64GEN_DMA", "default")        # where the 4 + 4 LDS-DMA pieces of an iteration sit: see DMA_PLANS
OPT_QK = os.environ.get("Q64GEN_QK", "kb")             # order of the 32 S^T MFMAs: "kb" key block 0 first, "ks" K-step major (4 accumulators rotate)
OPT_LAG = int(os.environ.get("Q64GEN_LAG", "0"))          # 1: a row-sum add follows its exponential one gap later (no dependent issue right behind a transcendental)
OPT_HK = tuple(int(x) for x in os.environ.get("Q64GEN_HK", "8,9,14,15").split(","))      # half units: gaps of phase A that issue the K(T+4) pieces
OPT_HV = tuple(int(x) for x in os.environ.get("Q64GEN_HV", "0,1,2,3").split(","))        # half units: gaps of phase B that issue the V(T+2) pieces
OPT_DEEP = int(os.environ.get("Q64GEN_DEEP", "0"))        # 1: whole units stage K / V^T fragments four sets deep (a[192:255]) and request them TWO groups ahead, as the half units do
OPT_V1 = int(os.environ.get("Q64GEN_V1", "0"))            # 1: the round-4 / 5 streams (two-phase loop, ring addresses in registers) for A/B builds
OPT_DROP = set(filter(None, os.environ.get("Q64GEN_DROP", "").split(",")))     # anatomy: fill, dma, lds, bar (results are wrong without them)
DMA_PLANS = {            # (gaps of phase A for K pieces 0..3, gaps of phase B for V pieces 0..3)
    "default": ((13, 15, 29, 31), (4, 5, 6, 7)),
    "a_late": ((28, 29, 30, 31), (4, 5, 6, 7)),
    "a_spread": ((6, 14, 22, 30), (5, 6, 7, 9)),
    "b_all": ((), (0, 1, 4, 5, 6, 7, 8, 9)),              # all eight in phase B (K first)
    "burst": ((29, 29, 29, 29), (5, 5, 5, 5)),
    "a_all": ((4, 5, 6, 7, 12, 13, 14, 15), ()),          # all eight in phase A (K first)
}

TILE = 16384
V_RING = 4 * TILE
STAGE_ROW = 272

# ---- register map
def OA(td): return 16 * td
def OB(td): return 64 + 16 * td
def QA(s): return 128 + 4 * s
def QB(s): return 160 + 4 * s
def FR(st, q): return 192 + 16 * st + 4 * q
def S(buf, a): return 64 * buf + 16 * a
NEGM = (128, 144)
V_LA, V_LB, V_MA, V_MB = 160, 161, 162, 163
V_C = (164, 165, 166, 167)          # running maxima: c0 (A kb0), c1 (A kb1), c2 (B kb0), c3 (B kb1)
V_MXA, V_MXB, V_T0, V_T1, V_ALA, V_ALB = 168, 169, 170, 171, 172, 173
V_KA, V_VA, V_KA2, V_VOFF, V_L16, V_HH4, V_NINF = 174, 175, 176, 177, 178, 179, 180
V_TOP = 183                          # highest arch VGPR the core owns
# owned SGPRs
S_IT, S_T, S_X0, S_X1, S_X2, S_PEND, S_SOFFK, S_SOFFV, S_KDST, S_VDST = 40, 41, 42, 43, 44, 45, 46, 47, 48, 49
S_VS0, S_VS1, S_VS2, S_HK, S_HV, S_NT32M1 = 50, 51, 52, 53, 54, 55
S_TOP = 57


def vr(i, n=1): return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"
def ar(i, n=1): return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"
def sr(i): return f"s{i}"


class Prog:
    """instruction list; cold(): blocks that almost never run are collected and emitted behind the hot stream (the hot path stays
    dense in the instruction cache and falls through its branches)"""
    def __init__(self):
        self.lag = []            # Q64GEN_LAG: row-sum adds waiting for the next gap
        self.out = []
        self.hot = self.out
        self.cold_blocks = []

    def begin_cold(self, entry, back):
        self._cold = []
        self._back = back
        self.out = self._cold
        self.label(entry)

    def end_cold(self):
        self.br("s_branch", self._back)
        self.cold_blocks.append(self._cold)
        self.out = self.hot

    def finish(self):
        for b in self.cold_blocks:
            self.hot.extend(b)
        self.cold_blocks = []

    def e(self, text):
        """one instruction; MFMA / PK are C macros holding the mnemonic of the operand type"""
        if "lds" in OPT_DROP and text.startswith("ds_read_b128 a["):
            return
        if text.startswith("MFMA "):
            self.out.append(f'Q64_MFMA " {text[5:]}\\n\\t"')
        elif text.startswith("PK "):
            self.out.append(f'Q64_PK " {text[3:]}\\n\\t"')
        else:
            self.out.append(f'"{text}\\n\\t"')

    def label(self, name):
        self.out.append(f'"{name}_%=:\\n\\t"')

    def br(self, op, name):
        self.e(f"{op} {name}_%=")

    def text(self):
        self.finish()
        return "\n    ".join(self.hot)


def VOFF(kk, td): return ((kk >> 1) * 8 + td * 2 + (kk & 1)) * 1024


# ------------------------------------------------------------------------------------------------------------------ phases
def qk_phase(p, dst, first, fill, last4):
    """32 S^T MFMAs into score buffer dst, key block 0 first; group 0's fragments already requested into fr[0]"""
    for n in range(32):
        if OPT_QK == "ks":
            s, kb, x = n >> 2, (n >> 1) & 1, n & 1
        else:
            kb, s, x = n >> 4, (n >> 1) & 7, n & 1
        j = n >> 1                                                       # fragment j is read from byte frag_off(j) of the slot
        grp, q = j >> 2, j & 3
        if n & 7 == 0:
            p.e("s_waitcnt lgkmcnt(0)")
        acc = vr(S(dst, x * 2 + kb), 16)
        c = ("0" if first else vr(NEGM[x], 16)) if s == 0 else acc
        p.e(f"MFMA {acc}, {ar(FR(grp & 1, q), 4)}, {ar((QB if x else QA)(s), 4)}, {c}")
        if (n & 7) < 4:
            if grp < 3:
                p.e(f"ds_read_b128 {ar(FR((grp + 1) & 1, n & 7), 4)}, {vr(V_KA)} offset:{frag_off((grp + 1) * 4 + (n & 7))}")
            else:
                last4(p, n & 7)
        fill(p, n)


def pv_phase(p, pbuf, fill, last4):
    """32 O^T MFMAs; P sits in place in score buffer pbuf (register quads [8 k2, 8 k2 + 4) of array 2 X + kb)"""
    for n in range(32):
        kk, td, x = n >> 3, (n >> 1) & 3, n & 1
        if n & 7 == 0:
            p.e("s_waitcnt lgkmcnt(0)")
        acc = ar((OB if x else OA)(td), 16)
        pb = vr(S(pbuf, x * 2 + (kk >> 1)) + 8 * (kk & 1), 4)
        p.e(f"MFMA {acc}, {ar(FR(kk & 1, td), 4)}, {pb}, {acc}")
        if (n & 7) < 4:
            if kk < 3:
                p.e(f"ds_read_b128 {ar(FR((kk + 1) & 1, n & 7), 4)}, {vr(V_VA)} offset:{VOFF(kk + 1, n & 7)}")
            else:
                last4(p, n & 7)
        fill(p, n)


def qk_phase_d(p, lds, dst, first, fill, ahead):
    """Q64GEN_DEEP: the 32 S^T MFMAs with four fragment sets (FRH) requested two groups ahead; K groups 0 and 1 already requested"""
    for n in range(32):
        kb, s, x = n >> 4, (n >> 1) & 7, n & 1
        j = n >> 1
        grp, q = j >> 2, j & 3
        if n & 7 == 0:
            lds.need(("K", grp))
        acc = vr(S(dst, x * 2 + kb), 16)
        c = ("0" if first else vr(NEGM[x], 16)) if s == 0 else acc
        p.e(f"MFMA {acc}, {ar(FRH(grp, q), 4)}, {ar((QB if x else QA)(s), 4)}, {c}")
        if (n & 7) < 4:
            ahead(grp, n & 7)
        fill(p, n)


def pv_phase_d(p, lds, pbuf, fill, ahead):
    for n in range(32):
        kk, td, x = n >> 3, (n >> 1) & 3, n & 1
        if n & 7 == 0:
            lds.need(("V", kk))
        acc = ar((OB if x else OA)(td), 16)
        pb = vr(S(pbuf, x * 2 + (kk >> 1)) + 8 * (kk & 1), 4)
        p.e(f"MFMA {acc}, {ar(FRH(kk, td), 4)}, {pb}, {acc}")
        if (n & 7) < 4:
            ahead(kk, n & 7)
        fill(p, n)


def frag_off(j):
    """LDS byte offset of the j-th K fragment an S^T phase consumes: (kb, s) at (kb * 8 + s) KB"""
    if OPT_QK == "ks":
        s_, kb = j >> 1, j & 1
        return (kb * 8 + s_) * 1024
    return j * 1024


def k_group0(p, vaddr):
    for q in range(4):
        p.e(f"ds_read_b128 {ar(FR(0, q), 4)}, {vr(vaddr)} offset:{frag_off(q)}")


def v_group0(p, vaddr):
    for q in range(4):
        p.e(f"ds_read_b128 {ar(FR(0, q), 4)}, {vr(vaddr)} offset:{VOFF(0, q)}")


# ---- softmax shadow (see attention_q64.hip for the schedule)
def flush_lag(p):
    for t in p.lag:
        p.e(t)
    p.lag = []


def row_sum(p, l, r):
    t = f"v_add_f32 {vr(l)}, {vr(l)}, {vr(r)}"
    if OPT_LAG:
        p.lag.append(t)
    else:
        p.e(t)


def fill_a(p, buf, n):
    """gap n of phase A: late exponential n (n < 24) + its row sum, pack n (in place)"""
    pa = 0 if n < 8 else 2 if n < 16 else 1 if n < 24 else 3
    pi = n & 7
    lo = S(buf, pa) + 2 * pi
    dst = S(buf, pa) + 8 * (pi >> 2) + (pi & 3)
    if n < 24:
        ea, er = (1, 8 + n) if n < 8 else (3, n - 8)
        ev = S(buf, ea) + er
        p.e(f"v_exp_f32 {vr(ev)}, {vr(ev)}")
        p.e(f"PK {vr(dst)}, {vr(lo)}, {vr(lo + 1)}")
        flush_lag(p)
        row_sum(p, V_LA if ea == 1 else V_LB, ev)
    else:
        p.e(f"PK {vr(dst)}, {vr(lo)}, {vr(lo + 1)}")
        flush_lag(p)


def fill_b(p, buf, e):
    """early exponentials, pair e = 0..19: (A, kb 0) 0..7, (B, kb 0) 8..15, (A, kb 1, registers 0..7) 16..19"""
    a = 0 if e < 8 else 2 if e < 16 else 1
    r = S(buf, a) + 2 * (e & 7)
    l = V_LB if a == 2 else V_LA
    p.e(f"v_exp_f32 {vr(r)}, {vr(r)}")
    p.e(f"v_exp_f32 {vr(r + 1)}, {vr(r + 1)}")
    flush_lag(p)
    row_sum(p, l, r)
    row_sum(p, l, r + 1)
    if e == 19:
        flush_lag(p)             # the last pair of a phase: nothing may stay pending across the iteration's end


def max_step(p, buf, kb, k):
    """running maxima of key block kb (arrays kb and 2 + kb), step k = 0..6 over registers 0..14"""
    for X in (0, 1):
        c = V_C[2 * X + kb]
        b = S(buf, 2 * X + kb)
        if k == 0:
            p.e(f"v_max3_f32 {vr(c)}, {vr(b)}, {vr(b + 1)}, {vr(b + 2)}")
        else:
            p.e(f"v_max3_f32 {vr(c)}, {vr(c)}, {vr(b + 2 * k + 1)}, {vr(b + 2 * k + 2)}")


def max_last(p, buf, dA, dB):
    p.e(f"v_max3_f32 {vr(dA)}, {vr(V_C[0])}, {vr(V_C[1])}, {vr(S(buf, 0) + 15)}")
    p.e(f"v_max3_f32 {vr(dB)}, {vr(V_C[2])}, {vr(V_C[3])}, {vr(S(buf, 2) + 15)}")
    p.e(f"v_max_f32 {vr(dA)}, {vr(dA)}, {vr(S(buf, 1) + 15)}")
    p.e(f"v_max_f32 {vr(dB)}, {vr(dB)}, {vr(S(buf, 3) + 15)}")


def xhalf_max(p, x, tmp):
    """max over the two 32-lane halves, result in every lane"""
    p.e(f"v_mov_b32 {vr(tmp)}, {vr(x)}")
    p.e("s_nop 1")
    p.e(f"v_permlane32_swap_b32 {vr(x)}, {vr(tmp)}")
    p.e("s_nop 1")
    p.e(f"v_max_f32 {vr(x)}, {vr(x)}, {vr(tmp)}")


def mask_tile(p, buf, s_tile64, uid):
    """keys >= N of the tile whose first key is s_tile64 (an SGPR): -inf.  key = tile64 + kb*32 + (r&3) + 8 (r>>2) + 4 hh"""
    p.e(f"v_mov_b32 {vr(V_NINF)}, 0xff800000")
    p.e(f"s_sub_i32 {sr(S_X1)}, %[N], {sr(s_tile64)}")                 # key >= N  <=>  4 hh >= N - tile64 - const
    for kb in range(2):
        for r in range(16):
            const = kb * 32 + (r & 3) + 8 * (r >> 2)
            p.e(f"s_sub_i32 {sr(S_X2)}, {sr(S_X1)}, {const}")
            p.e(f"v_cmp_ge_i32 vcc, {vr(V_HH4)}, {sr(S_X2)}")
            for X in (0, 1):
                reg = S(buf, 2 * X + kb) + r
                p.e(f"v_cndmask_b32 {vr(reg)}, {vr(reg)}, {vr(V_NINF)}, vcc")


def dma_piece(p, rs, s_soff, j):
    p.e(f"buffer_load_dwordx4 {vr(V_VOFF)}, {rs}, {sr(s_soff)} offen offset:{j * 1024} lds")


def soff_of(p, dst, s_tile, add):
    """dst = min(2 (tile + add) + wh, nt32 - 1) * 8192 + wq"""
    p.e(f"s_add_i32 {sr(dst)}, {sr(s_tile)}, {add}")
    p.e(f"s_lshl_b32 {sr(dst)}, {sr(dst)}, 1")
    p.e(f"s_add_i32 {sr(dst)}, {sr(dst)}, %[wh]")
    p.e(f"s_min_i32 {sr(dst)}, {sr(dst)}, {sr(S_NT32M1)}")
    p.e(f"s_lshl_b32 {sr(dst)}, {sr(dst)}, 13")
    p.e(f"s_add_i32 {sr(dst)}, {sr(dst)}, %[wq]")


def kslot_addr(p, vdst, add):
    """vdst = lane16 + ((T + add) & 3) * 16 KB"""
    p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, {add}")
    p.e(f"s_and_b32 {sr(S_X0)}, {sr(S_X0)}, 3")
    p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 14")
    p.e(f"v_add_u32_e32 {vr(vdst)}, {sr(S_X0)}, {vr(V_L16)}")


# ------------------------------------------------------------------------------------------------------------------ iteration
def iteration(p, cb, nb, steady, tag):
    """tile T (scores in buffer cb) is current, T + 1 (buffer nb) is next"""
    # The iteration's addresses and DMA operands are computed INSIDE its MFMA gaps, each just before its first use (as a burst in
    # front of the first MFMA they were ~45 instructions = ~150 cycles per tile with nothing to hide behind): setup_a / setup_b below.
    # K(T+1)'s read address is last iteration's look-ahead address.
    p.e(f"v_mov_b32 {vr(V_KA)}, {vr(V_KA2)}")
    if not steady:
        p.e(f"s_sub_i32 {sr(S_X0)}, %[nt], {sr(S_IT)}")                 # nt - it
        p.e(f"s_cmp_gt_i32 {sr(S_X0)}, 4")
        p.e(f"s_cselect_b32 {sr(S_HK)}, 1, 0")
        p.e(f"s_cmp_gt_i32 {sr(S_X0)}, 2")
        p.e(f"s_cselect_b32 {sr(S_HV)}, 1, 0")

    def setup_a(p, n):
        if n == 4:          # soffset of K(T+4): min(2 (T + 4) + wh, nt32 - 1) * 8192 + wq
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_T)}, 4")
            p.e(f"s_lshl_b32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, 1")
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, %[wh]")
        if n == 5:
            p.e(f"s_min_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, {sr(S_NT32M1)}")
            p.e(f"s_lshl_b32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, 13")
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, %[wq]")
        if n == 6:          # its LDS destination: slot (T + 4) & 3 = T & 3
            p.e(f"s_and_b32 {sr(S_KDST)}, {sr(S_T)}, 3")
            p.e(f"s_lshl_b32 {sr(S_KDST)}, {sr(S_KDST)}, 14")
            p.e(f"s_add_i32 {sr(S_KDST)}, {sr(S_KDST)}, %[dbase]")      # LDS base + wave * 4 KB
        if n == 7:
            p.e(f"s_mov_b32 m0, {sr(S_KDST)}")
            p.e(f"v_add_u32_e32 {vr(V_VA)}, {sr(S_VS0)}, {vr(V_L16)}")  # V(T) read address (first use: gap 24)
        if n == 12:         # look-ahead address: K(T+2) (first use: gap 24 of phase B)
            p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, 2")
            p.e(f"s_and_b32 {sr(S_X0)}, {sr(S_X0)}, 3")
            p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 14")
            p.e(f"v_add_u32_e32 {vr(V_KA2)}, {sr(S_X0)}, {vr(V_L16)}")
        if n == 28:         # soffset of V(T+2)
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_T)}, 2")
            p.e(f"s_lshl_b32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, 1")
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, %[wh]")
        if n == 30:
            p.e(f"s_min_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, {sr(S_NT32M1)}")
            p.e(f"s_lshl_b32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, 13")
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, %[wq]")
            p.e(f"s_add_i32 {sr(S_VDST)}, {sr(S_VS2)}, %[dbase]")

    def dma_guarded(p, rs, soff, j, flag, name):
        if steady:
            dma_piece(p, rs, soff, j)
        else:
            p.e(f"s_cmp_eq_u32 {sr(flag)}, 0")
            p.br("s_cbranch_scc1", name)
            dma_piece(p, rs, soff, j)
            p.label(name)

    ka_gaps, vb_gaps = DMA_PLANS[OPT_DMA]
    # pieces 0..3 = K(T+4), 4..7 = V(T+2); a plan lists the gaps of phase A, then of phase B (M0 switches where the V pieces start)
    seq_a = list(ka_gaps)
    seq_b = list(vb_gaps)
    n_k_in_b = 4 - min(4, len(seq_a))          # K pieces that a plan moved into phase B
    n_v_in_a = max(0, len(seq_a) - 4)          # V pieces that a plan moved into phase A

    def dma_at(p, phase, n):
        if "dma" in OPT_DROP:
            return
        seq = seq_a if phase == 0 else seq_b
        for idx, g in enumerate(seq):
            if g != n:
                continue
            piece = idx if phase == 0 else idx + len(seq_a)          # 0..7 in issue order
            if piece < 4:
                dma_guarded(p, "%[rk]", S_SOFFK, piece, S_HK, f"SKK{tag}{piece}")
            else:
                if piece == 4:
                    p.e(f"s_mov_b32 m0, {sr(S_VDST)}")
                    p.e("s_nop 0")
                dma_guarded(p, "%[rv]", S_SOFFV, piece - 4, S_HV, f"SKV{tag}{piece - 4}")

    def fa(p, n):
        if "fill" not in OPT_DROP:
            fill_a(p, cb, n)
            if 20 <= n <= 26 and OPT_QK == "kb":
                max_step(p, nb, 0, n - 20)
        setup_a(p, n)
        dma_at(p, 0, n)

    def last4_a(p, q):
        p.e(f"ds_read_b128 {ar(FR(0, q), 4)}, {vr(V_VA)} offset:{VOFF(0, q)}")

    if OPT_DEEP:
        lds = LdsQ(p, [("K", 0)] * 4 + [("K", 1)] * 4)
        qk_phase_d(p, lds, nb, False, fa, lambda grp, q: k_read_h(lds, grp + 2, q, V_KA) if grp < 2 else v_read_h(lds, grp - 2, q, V_VA))
    else:
        qk_phase(p, nb, False, fa, last4_a)

    def fb(p, n):
        dma_at(p, 1, n)
        if "fill" in OPT_DROP:
            return
        if n == 2:
            # the next tile is the last of the sequence and ragged: mask its keys >= N (scores are complete: >= 2 gaps behind the MFMAs)
            p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, 1")
            p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 6")
            p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
            p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
            p.br("s_cbranch_scc1", f"MASK{tag}")
            p.label(f"NOMASK{tag}")
            p.begin_cold(f"MASK{tag}", f"NOMASK{tag}")
            p.e("s_nop 7")
            mask_tile(p, nb, S_X0, tag)
            p.end_cold()
        if 2 <= n <= 8:
            if OPT_QK != "kb":
                max_step(p, nb, 0, n - 2)
            max_step(p, nb, 1, n - 2)
        if n == 9:
            max_last(p, nb, V_MXA, V_MXB)
        if n == 10:
            xhalf_max(p, V_MXA, V_T0)
            xhalf_max(p, V_MXB, V_T1)
        if n == 11:
            p.e(f"v_max_f32 {vr(V_T0)}, {vr(V_MXA)}, {vr(V_MXB)}")
            p.e(f"v_cmp_lt_f32 vcc, 0x41000000, {vr(V_T0)}")           # 8.0 < max
            p.br("s_cbranch_vccnz", f"MOVE{tag}")
            p.label(f"NOMOVE{tag}")
            p.begin_cold(f"MOVE{tag}", f"NOMOVE{tag}")
            # ---- the reference maximum moves (rare): scores of tile T+1, seeds, row sums now; O^T after this phase's MFMAs
            p.e(f"v_max_f32 {vr(V_T0)}, 0, {vr(V_MXA)}")
            p.e(f"v_max_f32 {vr(V_T1)}, 0, {vr(V_MXB)}")
            p.e(f"v_add_f32 {vr(V_MA)}, {vr(V_MA)}, {vr(V_T0)}")
            p.e(f"v_add_f32 {vr(V_MB)}, {vr(V_MB)}, {vr(V_T1)}")
            p.e(f"v_exp_f32 {vr(V_ALA)}, -{vr(V_T0)}")
            p.e(f"v_exp_f32 {vr(V_ALB)}, -{vr(V_T1)}")
            p.e("s_nop 0")
            p.e(f"v_mul_f32 {vr(V_LA)}, {vr(V_LA)}, {vr(V_ALA)}")
            p.e(f"v_mul_f32 {vr(V_LB)}, {vr(V_LB)}, {vr(V_ALB)}")
            for a in range(4):
                for r in range(16):
                    reg = S(nb, a) + r
                    p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr(V_T0 if a < 2 else V_T1)}")
            for r in range(16):
                p.e(f"v_sub_f32 {vr(NEGM[0] + r)}, 0, {vr(V_MA)}")
                p.e(f"v_sub_f32 {vr(NEGM[1] + r)}, 0, {vr(V_MB)}")
            p.e(f"s_mov_b32 {sr(S_PEND)}, 1")
            p.end_cold()
        if n >= 12:
            fill_b(p, nb, n - 12)

    def last4_b(p, q):
        p.e(f"ds_read_b128 {ar(FR(0, q), 4)}, {vr(V_KA2)} offset:{frag_off(q)}")

    if OPT_DEEP:
        pv_phase_d(p, lds, cb, fb, lambda kk, q: v_read_h(lds, kk + 2, q, V_VA) if kk < 2 else k_read_h(lds, kk - 2, q, V_KA2))
        assert lds.q == [("K", 0)] * 4 + [("K", 1)] * 4
    else:
        pv_phase(p, cb, fb, last4_b)
    # pending rescale of O^T
    p.e(f"s_cmp_lg_u32 {sr(S_PEND)}, 0")
    p.br("s_cbranch_scc1", f"PEND{tag}")
    p.label(f"NOPEND{tag}")
    p.begin_cold(f"PEND{tag}", f"NOPEND{tag}")
    p.e("s_nop 15")
    p.e("s_nop 7")
    for k in range(128):
        p.e(f"v_accvgpr_read_b32 {vr(V_T0)}, {ar(k)}")
        p.e(f"v_mul_f32 {vr(V_T0)}, {vr(V_T0)}, {vr(V_ALA if k < 64 else V_ALB)}")
        p.e(f"v_accvgpr_write_b32 {ar(k)}, {vr(V_T0)}")
    p.e("s_nop 3")
    p.e(f"s_mov_b32 {sr(S_PEND)}, 0")
    p.end_cold()
    # everything issued before this iteration has landed; this iteration's pieces fly on
    if steady:
        p.e("s_waitcnt vmcnt(8)")
    else:
        p.e(f"s_cmp_eq_u32 {sr(S_HK)}, 0")
        p.br("s_cbranch_scc1", f"W4{tag}")
        p.e("s_waitcnt vmcnt(8)")
        p.br("s_branch", f"WD{tag}")
        p.label(f"W4{tag}")
        p.e(f"s_cmp_eq_u32 {sr(S_HV)}, 0")
        p.br("s_cbranch_scc1", f"W0{tag}")
        p.e("s_waitcnt vmcnt(4)")
        p.br("s_branch", f"WD{tag}")
        p.label(f"W0{tag}")
        p.e("s_waitcnt vmcnt(0)")
        p.label(f"WD{tag}")
    if "bar" not in OPT_DROP:
        p.e("s_barrier")
    p.e(f"s_mov_b32 {sr(S_X0)}, {sr(S_VS0)}")
    p.e(f"s_mov_b32 {sr(S_VS0)}, {sr(S_VS1)}")
    p.e(f"s_mov_b32 {sr(S_VS1)}, {sr(S_VS2)}")
    p.e(f"s_mov_b32 {sr(S_VS2)}, {sr(S_X0)}")
    p.e(f"s_add_i32 {sr(S_IT)}, {sr(S_IT)}, 1")
    p.e(f"s_add_i32 {sr(S_T)}, {sr(S_T)}, 1")


def tail(p, cb):
    """the sequence's last tile: late exponentials + packs, then its O^T MFMAs"""
    for n in range(32):
        fill_a(p, cb, n)
    p.e(f"v_add_u32_e32 {vr(V_VA)}, {sr(S_VS0)}, {vr(V_L16)}")
    if OPT_DEEP:
        lds = LdsQ(p, ["old"] * 8)
        for g in range(2):
            for q in range(4):
                v_read_h(lds, g, q, V_VA)
        pv_phase_d(p, lds, cb, lambda p, n: None, lambda kk, q: v_read_h(lds, kk + 2, q, V_VA) if kk < 2 else None)
        return
    v_group0(p, V_VA)
    pv_phase(p, cb, lambda p, n: None, lambda p, q: None)


def core():
    """whole unit (blocks A and B).  The statement is ONE asm for both unit shapes (a C++ branch around two statements made hipcc
    duplicate them): it starts with the dispatch to the half-unit stream (core_h, appended behind this one's cold blocks)"""
    p = Prog()
    p.e("s_cmp_lg_u32 %[half], 0")
    p.br("s_cbranch_scc1", "HCORE")
    p.e(f"v_mov_b32 {vr(V_L16)}, %[lane16]")
    p.e(f"v_mov_b32 {vr(V_VOFF)}, %[vlane]")
    p.e(f"v_mov_b32 {vr(V_HH4)}, %[hh4]")
    p.e(f"v_mov_b32 {vr(V_LA)}, 0")
    p.e(f"v_mov_b32 {vr(V_LB)}, 0")
    p.e(f"s_mov_b32 {sr(S_IT)}, 0")
    p.e(f"s_mov_b32 {sr(S_T)}, %[tlo]")
    p.e(f"s_mov_b32 {sr(S_PEND)}, 0")
    p.e(f"s_sub_i32 {sr(S_NT32M1)}, %[nt32], 1")
    p.e(f"s_mov_b32 {sr(S_VS0)}, {V_RING}")
    p.e(f"s_mov_b32 {sr(S_VS1)}, {V_RING + TILE}")
    p.e(f"s_mov_b32 {sr(S_VS2)}, {V_RING + 2 * TILE}")
    # ---- the unit's first requests but K(3) have landed (Q too); a finished unit's stores (>= 16, issued after them) may fly on
    p.e("s_cmp_eq_u32 %[first], 0")
    p.br("s_cbranch_scc1", "WNF")
    p.e("s_cmp_gt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "WF4")
    p.e("s_waitcnt vmcnt(0)")
    p.br("s_branch", "WDONE")
    p.label("WF4")
    p.e("s_waitcnt vmcnt(4)")
    p.br("s_branch", "WDONE")
    p.label("WNF")
    p.e("s_cmp_gt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "WN4")
    p.e("s_waitcnt vmcnt(16)")
    p.br("s_branch", "WDONE")
    p.label("WN4")
    p.e("s_waitcnt vmcnt(20)")
    p.label("WDONE")
    p.e("s_barrier")
    # ---- tile 0: scores with a zero seed (the O^T accumulators are zeroed in its gaps), first reference maximum, early exponentials
    kslot_addr(p, V_KA, 0)

    def f0(p, n):
        for k in range(4):
            p.e(f"v_accvgpr_write_b32 {ar(4 * n + k)}, 0")

    if OPT_DEEP:
        lds0 = LdsQ(p)
        for g in range(2):
            for q in range(4):
                k_read_h(lds0, g, q, V_KA)
        qk_phase_d(p, lds0, 0, True, f0, lambda grp, q: k_read_h(lds0, grp + 2, q, V_KA) if grp < 2 else None)
    else:
        k_group0(p, V_KA)
        qk_phase(p, 0, True, f0, lambda p, q: None)
    p.e("s_nop 15")
    p.e("s_nop 7")
    p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_T)}, 6")
    p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
    p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
    p.br("s_cbranch_scc1", "MASK0")
    p.label("NOMASK0")
    p.begin_cold("MASK0", "NOMASK0")
    mask_tile(p, 0, S_X0, "0")
    p.end_cold()
    for k in range(7):
        max_step(p, 0, 0, k)
        max_step(p, 0, 1, k)
    max_last(p, 0, V_MA, V_MB)
    xhalf_max(p, V_MA, V_T0)
    xhalf_max(p, V_MB, V_T1)
    for a in range(4):
        for r in range(16):
            reg = S(0, a) + r
            p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr(V_MA if a < 2 else V_MB)}")
    for r in range(16):
        p.e(f"v_sub_f32 {vr(NEGM[0] + r)}, 0, {vr(V_MA)}")
        p.e(f"v_sub_f32 {vr(NEGM[1] + r)}, 0, {vr(V_MB)}")
    for e in range(20):
        fill_b(p, 0, e)
    p.e("s_barrier")                                                       # every wave has read K(0): its slot may take K(4)
    # group 0 of K(1) for the first iteration
    p.e("s_cmp_lt_i32 %[nt], 2")
    p.br("s_cbranch_scc1", "TAIL_E")
    kslot_addr(p, V_KA2, 1)
    if OPT_DEEP:
        for g in range(2):
            for q in range(4):
                p.e(f"ds_read_b128 {ar(FRH(g, q), 4)}, {vr(V_KA2)} offset:{(4 * g + q) * 1024}")
    else:
        k_group0(p, V_KA2)
    # ---- the loop: even iterations have the current tile in buffer 0, odd ones in buffer 1
    for par, (cb, nb) in enumerate(((0, 1), (1, 0))):
        me, other = "EO"[par], "OE"[par]
        p.label(f"LOOP_{me}")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_IT)}, 1")
        p.e(f"s_cmp_ge_i32 {sr(S_X0)}, %[nt]")
        p.br("s_cbranch_scc1", f"TAIL_{me}")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_IT)}, 4")
        p.e(f"s_cmp_ge_i32 {sr(S_X0)}, %[nt]")
        p.br("s_cbranch_scc1", f"DRAIN_{me}")
        iteration(p, cb, nb, True, f"S{me}")
        p.br("s_branch", f"LOOP_{other}")
        p.label(f"DRAIN_{me}")
        iteration(p, cb, nb, False, f"D{me}")
        p.br("s_branch", f"LOOP_{other}")
    p.label("TAIL_E")
    tail(p, 0)
    p.br("s_branch", "END")
    p.label("TAIL_O")
    tail(p, 1)
    p.label("END")
    p.e("s_barrier")                                                       # the rings are free for the next unit's first requests
    p.e("s_nop 15")
    p.e("s_nop 7")
    p.e(f"v_mov_b32 %[o_la], {vr(V_LA)}")
    p.e(f"v_mov_b32 %[o_lb], {vr(V_LB)}")
    p.e(f"v_mov_b32 %[o_ma], {vr(V_MA)}")
    p.e(f"v_mov_b32 %[o_mb], {vr(V_MB)}")
    p.br("s_branch", "EXIT")
    p.finish()
    return p



# ------------------------------------------------------------------------------------------------------------------ half units
# A HALF unit = 4 waves x ONE 32-query block (block A only): the unit shape of the second, shorter round when the whole units of a
# launch fill the persistent grid once and a bit (DEX B = 32, N = 1300: 41 query blocks per (element, head) = 4 whole units + 9 blocks;
# as whole units the 9 blocks were a second full-length round for half of the chip).  Its rows leave in the ordinary output format -
# unlike a key split, nobody merges anything.  Same rings, same DMA, same lazy reference move; 16 + 16 MFMAs per tile, so every K / V^T
# fragment feeds ONE MFMA: fragment staging is four sets deep (a[192:255]) and requested TWO groups ahead.
def FRH(st, q): return 192 + 16 * st + 4 * q


class LdsQ:
    """LDS reads return in order: the wait count for a group = the number of reads issued after its last one"""
    def __init__(self, p, outstanding=()):
        self.p, self.q = p, list(outstanding)

    def read(self, tag, text):
        self.p.e(text)
        self.q.append(tag)

    def need(self, tag):
        idx = max(i for i, t in enumerate(self.q) if t == tag)
        self.p.e(f"s_waitcnt lgkmcnt({len(self.q) - 1 - idx})")
        self.q = self.q[idx + 1:]


def k_read_h(lds, grp, q, vaddr):
    lds.read(("K", grp), f"ds_read_b128 {ar(FRH(grp, q), 4)}, {vr(vaddr)} offset:{(4 * grp + q) * 1024}")


def v_read_h(lds, grp, q, vaddr):
    lds.read(("V", grp), f"ds_read_b128 {ar(FRH(grp, q), 4)}, {vr(vaddr)} offset:{VOFF(grp, q)}")


def qk_phase_h(p, lds, dst, first, fill, ahead):
    """16 S^T MFMAs of block A into score buffer dst; K groups 0 and 1 already requested; ahead(grp, q) issues the read two groups on"""
    for n in range(16):
        kb, s = n >> 3, n & 7
        grp, q = n >> 2, n & 3
        if q == 0:
            lds.need(("K", grp))
        acc = vr(S(dst, kb), 16)
        c = ("0" if first else vr(NEGM[0], 16)) if s == 0 else acc
        p.e(f"MFMA {acc}, {ar(FRH(grp, q), 4)}, {ar(QA(s), 4)}, {c}")
        ahead(grp, q)
        fill(p, n)


def pv_phase_h(p, lds, pbuf, fill, ahead):
    """16 O^T MFMAs of block A; V^T groups 0 and 1 already requested"""
    for n in range(16):
        kk, td = n >> 2, n & 3
        if td == 0:
            lds.need(("V", kk))
        acc = ar(OA(td), 16)
        pb = vr(S(pbuf, kk >> 1) + 8 * (kk & 1), 4)
        p.e(f"MFMA {acc}, {ar(FRH(kk, td), 4)}, {pb}, {acc}")
        ahead(kk, td)
        fill(p, n)


def max_step_h(p, buf, kb, k):
    c, b = V_C[kb], S(buf, kb)
    if k == 0:
        p.e(f"v_max3_f32 {vr(c)}, {vr(b)}, {vr(b + 1)}, {vr(b + 2)}")
    else:
        p.e(f"v_max3_f32 {vr(c)}, {vr(c)}, {vr(b + 2 * k + 1)}, {vr(b + 2 * k + 2)}")


def max_last_h(p, buf, d):
    p.e(f"v_max3_f32 {vr(d)}, {vr(V_C[0])}, {vr(V_C[1])}, {vr(S(buf, 0) + 15)}")
    p.e(f"v_max_f32 {vr(d)}, {vr(d)}, {vr(S(buf, 1) + 15)}")


def late_h(p, cb, n):
    """gap n = 0..15 of phase A: the late exponentials of tile T (array 0 registers 14, 15, all of array 1) + pack n, in place"""
    a0, a1 = S(cb, 0), S(cb, 1)
    ev = [a0 + 14, a0 + 15] if n == 0 else [a1, a1 + 1] if n == 1 else [a1 + 2, a1 + 3] if n == 2 else [a1 + n + 1] if n <= 14 else []
    arr, pi = (a0, n) if n < 8 else (a1, n - 8)
    lo, dst = arr + 2 * pi, arr + 8 * (pi >> 2) + (pi & 3)
    for r in ev:
        p.e(f"v_exp_f32 {vr(r)}, {vr(r)}")
    p.e(f"PK {vr(dst)}, {vr(lo)}, {vr(lo + 1)}")
    flush_lag(p)
    for r in ev:
        row_sum(p, V_LA, r)


def early_h(p, buf, e):
    """early exponentials, pair e = 0..6: registers 2 e, 2 e + 1 of array 0"""
    r = S(buf, 0) + 2 * e
    p.e(f"v_exp_f32 {vr(r)}, {vr(r)}")
    p.e(f"v_exp_f32 {vr(r + 1)}, {vr(r + 1)}")
    flush_lag(p)
    row_sum(p, V_LA, r)
    row_sum(p, V_LA, r + 1)
    if e == 6:
        flush_lag(p)


def mask_tile_h(p, buf, s_tile64):
    p.e(f"v_mov_b32 {vr(V_NINF)}, 0xff800000")
    p.e(f"s_sub_i32 {sr(S_X1)}, %[N], {sr(s_tile64)}")
    for kb in range(2):
        for r in range(16):
            const = kb * 32 + (r & 3) + 8 * (r >> 2)
            p.e(f"s_sub_i32 {sr(S_X2)}, {sr(S_X1)}, {const}")
            p.e(f"v_cmp_ge_i32 vcc, {vr(V_HH4)}, {sr(S_X2)}")
            reg = S(buf, kb) + r
            p.e(f"v_cndmask_b32 {vr(reg)}, {vr(reg)}, {vr(V_NINF)}, vcc")


H_K_GAPS = OPT_HK               # gaps of phase A that issue the four K(T+4) pieces
H_V_GAPS = OPT_HV               # gaps of phase B that issue the four V(T+2) pieces


def iteration_h(p, cb, nb, steady, tag):
    """half unit: tile T (scores in buffer cb) is current, T + 1 (buffer nb) is next; LDS state at entry: K(T+1) groups 0, 1 requested"""
    lds = LdsQ(p, [("K", 0)] * 4 + [("K", 1)] * 4)
    p.e(f"v_mov_b32 {vr(V_KA)}, {vr(V_KA2)}")
    if not steady:
        p.e(f"s_sub_i32 {sr(S_X0)}, %[nt], {sr(S_IT)}")
        p.e(f"s_cmp_gt_i32 {sr(S_X0)}, 4")
        p.e(f"s_cselect_b32 {sr(S_HK)}, 1, 0")
        p.e(f"s_cmp_gt_i32 {sr(S_X0)}, 2")
        p.e(f"s_cselect_b32 {sr(S_HV)}, 1, 0")

    def dma_guarded(rs, soff, j, flag, name):
        if steady:
            dma_piece(p, rs, soff, j)
        else:
            p.e(f"s_cmp_eq_u32 {sr(flag)}, 0")
            p.br("s_cbranch_scc1", name)
            dma_piece(p, rs, soff, j)
            p.label(name)

    def setup_a(n):
        if n == 1:
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_T)}, 4")
            p.e(f"s_lshl_b32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, 1")
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, %[wh]")
        if n == 2:
            p.e(f"s_min_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, {sr(S_NT32M1)}")
            p.e(f"s_lshl_b32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, 13")
            p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, %[wq]")
        if n == 3:
            p.e(f"s_and_b32 {sr(S_KDST)}, {sr(S_T)}, 3")
            p.e(f"s_lshl_b32 {sr(S_KDST)}, {sr(S_KDST)}, 14")
            p.e(f"s_add_i32 {sr(S_KDST)}, {sr(S_KDST)}, %[dbase]")
        if n == 4:
            p.e(f"s_mov_b32 m0, {sr(S_KDST)}")
            p.e(f"v_add_u32_e32 {vr(V_VA)}, {sr(S_VS0)}, {vr(V_L16)}")      # V(T) read address (first use: gap 8)
        if n == 6:          # look-ahead address: K(T+2) (first use: gap 8 of phase B)
            p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, 2")
            p.e(f"s_and_b32 {sr(S_X0)}, {sr(S_X0)}, 3")
            p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 14")
            p.e(f"v_add_u32_e32 {vr(V_KA2)}, {sr(S_X0)}, {vr(V_L16)}")
        if n == 12:
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_T)}, 2")
            p.e(f"s_lshl_b32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, 1")
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, %[wh]")
        if n == 13:
            p.e(f"s_min_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, {sr(S_NT32M1)}")
            p.e(f"s_lshl_b32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, 13")
            p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, %[wq]")
            p.e(f"s_add_i32 {sr(S_VDST)}, {sr(S_VS2)}, %[dbase]")

    def fa(p_, n):
        late_h(p, cb, n)
        if 10 <= n <= 15:
            max_step_h(p, nb, 0, n - 10)
        if n == 15:
            max_step_h(p, nb, 0, 6)
        setup_a(n)
        for j, g in enumerate(H_K_GAPS):
            if g == n:
                dma_guarded("%[rk]", S_SOFFK, j, S_HK, f"HKK{tag}{j}")

    def ahead_a(grp, q):
        if grp < 2:
            k_read_h(lds, grp + 2, q, V_KA)
        else:
            v_read_h(lds, grp - 2, q, V_VA)

    qk_phase_h(p, lds, nb, False, fa, ahead_a)

    def fb(p_, n):
        for j, g in enumerate(H_V_GAPS):
            if g == n:
                if j == 0:
                    p.e(f"s_mov_b32 m0, {sr(S_VDST)}")
                    p.e("s_nop 0")
                dma_guarded("%[rv]", S_SOFFV, j, S_HV, f"HKV{tag}{j}")
        if n == 2:
            p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, 1")
            p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 6")
            p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
            p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
            p.br("s_cbranch_scc1", f"HMASK{tag}")
            p.label(f"HNOMASK{tag}")
            p.begin_cold(f"HMASK{tag}", f"HNOMASK{tag}")
            p.e("s_nop 7")
            mask_tile_h(p, nb, S_X0)
            p.end_cold()
        if 2 <= n <= 4:
            max_step_h(p, nb, 1, 2 * (n - 2))
            max_step_h(p, nb, 1, 2 * (n - 2) + 1)
        if n == 5:
            max_step_h(p, nb, 1, 6)
            max_last_h(p, nb, V_MXA)
        if n == 6:
            xhalf_max(p, V_MXA, V_T0)
        if n == 7:
            p.e(f"v_cmp_lt_f32 vcc, 0x41000000, {vr(V_MXA)}")          # 8.0 < max
            p.br("s_cbranch_vccnz", f"HMOVE{tag}")
            p.label(f"HNOMOVE{tag}")
            p.begin_cold(f"HMOVE{tag}", f"HNOMOVE{tag}")
            p.e(f"v_max_f32 {vr(V_T0)}, 0, {vr(V_MXA)}")
            p.e(f"v_add_f32 {vr(V_MA)}, {vr(V_MA)}, {vr(V_T0)}")
            p.e(f"v_exp_f32 {vr(V_ALA)}, -{vr(V_T0)}")
            p.e("s_nop 0")
            p.e(f"v_mul_f32 {vr(V_LA)}, {vr(V_LA)}, {vr(V_ALA)}")
            for a in range(2):
                for r in range(16):
                    reg = S(nb, a) + r
                    p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr(V_T0)}")
            for r in range(16):
                p.e(f"v_sub_f32 {vr(NEGM[0] + r)}, 0, {vr(V_MA)}")
            p.e(f"s_mov_b32 {sr(S_PEND)}, 1")
            p.end_cold()
        if 8 <= n <= 14:
            early_h(p, nb, n - 8)

    def ahead_b(kk, td):
        if kk < 2:
            v_read_h(lds, kk + 2, td, V_VA)
        else:
            k_read_h(lds, kk - 2, td, V_KA2)

    pv_phase_h(p, lds, cb, fb, ahead_b)
    assert lds.q == [("K", 0)] * 4 + [("K", 1)] * 4
    p.e(f"s_cmp_lg_u32 {sr(S_PEND)}, 0")
    p.br("s_cbranch_scc1", f"HPEND{tag}")
    p.label(f"HNOPEND{tag}")
    p.begin_cold(f"HPEND{tag}", f"HNOPEND{tag}")
    p.e("s_nop 15")
    p.e("s_nop 7")
    for k in range(64):
        p.e(f"v_accvgpr_read_b32 {vr(V_T0)}, {ar(k)}")
        p.e(f"v_mul_f32 {vr(V_T0)}, {vr(V_T0)}, {vr(V_ALA)}")
        p.e(f"v_accvgpr_write_b32 {ar(k)}, {vr(V_T0)}")
    p.e("s_nop 3")
    p.e(f"s_mov_b32 {sr(S_PEND)}, 0")
    p.end_cold()
    if steady:
        p.e("s_waitcnt vmcnt(8)")
    else:
        p.e(f"s_cmp_eq_u32 {sr(S_HK)}, 0")
        p.br("s_cbranch_scc1", f"HW4{tag}")
        p.e("s_waitcnt vmcnt(8)")
        p.br("s_branch", f"HWD{tag}")
        p.label(f"HW4{tag}")
        p.e(f"s_cmp_eq_u32 {sr(S_HV)}, 0")
        p.br("s_cbranch_scc1", f"HW0{tag}")
        p.e("s_waitcnt vmcnt(4)")
        p.br("s_branch", f"HWD{tag}")
        p.label(f"HW0{tag}")
        p.e("s_waitcnt vmcnt(0)")
        p.label(f"HWD{tag}")
    p.e("s_barrier")
    p.e(f"s_mov_b32 {sr(S_X0)}, {sr(S_VS0)}")
    p.e(f"s_mov_b32 {sr(S_VS0)}, {sr(S_VS1)}")
    p.e(f"s_mov_b32 {sr(S_VS1)}, {sr(S_VS2)}")
    p.e(f"s_mov_b32 {sr(S_VS2)}, {sr(S_X0)}")
    p.e(f"s_add_i32 {sr(S_IT)}, {sr(S_IT)}, 1")
    p.e(f"s_add_i32 {sr(S_T)}, {sr(S_T)}, 1")


def tail_h(p, cb):
    """the sequence's last tile: late exponentials + packs, then its O^T MFMAs (older reads may still be in flight: counts are relative)"""
    for n in range(16):
        late_h(p, cb, n)
    p.e(f"v_add_u32_e32 {vr(V_VA)}, {sr(S_VS0)}, {vr(V_L16)}")
    lds = LdsQ(p, ["old"] * 8)
    for g in range(2):
        for q in range(4):
            v_read_h(lds, g, q, V_VA)

    def ahead(kk, td):
        if kk < 2:
            v_read_h(lds, kk + 2, td, V_VA)

    pv_phase_h(p, lds, cb, lambda p_, n: None, ahead)


def core_h():
    p = Prog()
    p.label("HCORE")
    p.e(f"v_mov_b32 {vr(V_L16)}, %[lane16]")
    p.e(f"v_mov_b32 {vr(V_VOFF)}, %[vlane]")
    p.e(f"v_mov_b32 {vr(V_HH4)}, %[hh4]")
    p.e(f"v_mov_b32 {vr(V_LA)}, 0")
    p.e(f"s_mov_b32 {sr(S_IT)}, 0")
    p.e(f"s_mov_b32 {sr(S_T)}, %[tlo]")
    p.e(f"s_mov_b32 {sr(S_PEND)}, 0")
    p.e(f"s_sub_i32 {sr(S_NT32M1)}, %[nt32], 1")
    p.e(f"s_mov_b32 {sr(S_VS0)}, {V_RING}")
    p.e(f"s_mov_b32 {sr(S_VS1)}, {V_RING + TILE}")
    p.e(f"s_mov_b32 {sr(S_VS2)}, {V_RING + 2 * TILE}")
    # the unit's first requests but K(3) have landed; a finished unit's stores (>= 8: a half unit's one block in 16-bit rows) may fly on
    p.e("s_cmp_eq_u32 %[first], 0")
    p.br("s_cbranch_scc1", "HWNF")
    p.e("s_cmp_gt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "HWF4")
    p.e("s_waitcnt vmcnt(0)")
    p.br("s_branch", "HWDONE")
    p.label("HWF4")
    p.e("s_waitcnt vmcnt(4)")
    p.br("s_branch", "HWDONE")
    p.label("HWNF")
    p.e("s_cmp_gt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "HWN4")
    p.e("s_waitcnt vmcnt(8)")
    p.br("s_branch", "HWDONE")
    p.label("HWN4")
    p.e("s_waitcnt vmcnt(12)")
    p.label("HWDONE")
    p.e("s_barrier")
    # ---- tile 0
    kslot_addr(p, V_KA, 0)
    lds = LdsQ(p)
    for g in range(2):
        for q in range(4):
            k_read_h(lds, g, q, V_KA)

    def f0(p_, n):
        for k in range(4):
            p.e(f"v_accvgpr_write_b32 {ar(4 * n + k)}, 0")

    def ahead0(grp, q):
        if grp < 2:
            k_read_h(lds, grp + 2, q, V_KA)

    qk_phase_h(p, lds, 0, True, f0, ahead0)
    p.e("s_nop 15")
    p.e("s_nop 7")
    p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_T)}, 6")
    p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
    p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
    p.br("s_cbranch_scc1", "HMASK0")
    p.label("HNOMASK0")
    p.begin_cold("HMASK0", "HNOMASK0")
    mask_tile_h(p, 0, S_X0)
    p.end_cold()
    for k in range(7):
        max_step_h(p, 0, 0, k)
        max_step_h(p, 0, 1, k)
    max_last_h(p, 0, V_MA)
    xhalf_max(p, V_MA, V_T0)
    for a in range(2):
        for r in range(16):
            reg = S(0, a) + r
            p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr(V_MA)}")
    for r in range(16):
        p.e(f"v_sub_f32 {vr(NEGM[0] + r)}, 0, {vr(V_MA)}")
    for e in range(7):
        early_h(p, 0, e)
    p.e("s_barrier")                                                       # every wave has read K(0): its slot may take K(4)
    p.e("s_cmp_lt_i32 %[nt], 2")
    p.br("s_cbranch_scc1", "HTAIL_E")
    kslot_addr(p, V_KA2, 1)
    for g in range(2):
        for q in range(4):
            p.e(f"ds_read_b128 {ar(FRH(g, q), 4)}, {vr(V_KA2)} offset:{(4 * g + q) * 1024}")
    for par, (cb, nb) in enumerate(((0, 1), (1, 0))):
        me, other = "EO"[par], "OE"[par]
        p.label(f"HLOOP_{me}")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_IT)}, 1")
        p.e(f"s_cmp_ge_i32 {sr(S_X0)}, %[nt]")
        p.br("s_cbranch_scc1", f"HTAIL_{me}")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_IT)}, 4")
        p.e(f"s_cmp_ge_i32 {sr(S_X0)}, %[nt]")
        p.br("s_cbranch_scc1", f"HDRAIN_{me}")
        iteration_h(p, cb, nb, True, f"S{me}")
        p.br("s_branch", f"HLOOP_{other}")
        p.label(f"HDRAIN_{me}")
        iteration_h(p, cb, nb, False, f"D{me}")
        p.br("s_branch", f"HLOOP_{other}")
    p.label("HTAIL_E")
    tail_h(p, 0)
    p.br("s_branch", "HEND")
    p.label("HTAIL_O")
    tail_h(p, 1)
    p.label("HEND")
    p.e("s_barrier")
    p.e("s_nop 15")
    p.e("s_nop 7")
    p.e(f"v_mov_b32 %[o_la], {vr(V_LA)}")
    p.e(f"v_mov_b32 %[o_ma], {vr(V_MA)}")
    p.e("v_mov_b32 %[o_lb], 0")
    p.e("v_mov_b32 %[o_mb], 0")
    p.br("s_branch", "EXIT")
    p.finish()
    p.label("EXIT")
    return p



# ================================================================================================================== v2 streams
# Round 6.  tools/valubench says what a lone wave pays per instruction between its MFMAs (32 cycles each): a plain VALU ~5.7, a VALU
# that depends on the one before ~8.5, v_exp ~9.5, ANY scalar ALU instruction ~8.5.  The v1 iteration issued ~50 scalar instructions
# per key tile (ring-slot addresses, soffsets, V-slot rotation, two counters, loop dispatch) and its 64 row-sum adds were one dependent
# chain per query block - ~500 cycles over the 2 048 of the tile's MFMAs, exactly what the stamps showed (2 554).  v2:
#   * the loop is unrolled over the FOUR ring slots (phase p = tile & 3, unit-relative): K / V^T slot addresses are immediates of the
#     ds_read / of one s_add into M0 - no address registers; the V^T ring has four slots too (its fourth aliases the output stage, which
#     is only live between a unit's last barrier and the next unit's first), so nothing rotates;
#   * steady iterations (five or more to go) advance the DMA soffsets by one add each, never clamp, never mask; one counter (REM);
#   * the pending O^T rescale after a reference-maximum move is the VCC of the move test itself (s_cbranch_vccnz), not a flag register;
#     the cross-half maximum of the running maxima is only taken inside the (cold) move;
#   * row sums go to two accumulators per block alternately, and an add follows its exponential one gap later;
#   * fragments are staged four sets deep and requested two groups ahead in both unit shapes.
# One generator for both unit shapes: NB = 2 (whole: blocks A and B) / 1 (half: block A).
V2_LA = (184, 185)       # row-sum accumulators of block A
V2_LB = (186, 187)
V2_L16V = 188            # lane16 + 64 KB: base of the V^T ring reads
V2_TOP = 189
S_REM = 40               # iterations left (including the current one at its start)
S2_TOP = 57


class StreamV2:
    def __init__(self, NB, pfx):
        self.NB, self.G, self.pfx = NB, 16 * NB, pfx
        self.sum_i = [0, 0]

    # ---- small helpers
    def L(self, name): return f"{self.pfx}{name}"

    def row_sum(self, p, x, r):
        acc = (V2_LA, V2_LB)[x][self.sum_i[x] & 1]
        self.sum_i[x] += 1
        p.lag.append(f"v_add_f32 {vr(acc)}, {vr(acc)}, {vr(r)}")

    def exp(self, p, x, r):
        p.e(f"v_exp_f32 {vr(r)}, {vr(r)}")
        self.row_sum(p, x, r)

    def k_read(self, lds, slot, grp, q):
        lds.read(("K", grp), f"ds_read_b128 {ar(FRH(grp, q), 4)}, {vr(V_L16)} offset:{slot * TILE + (4 * grp + q) * 1024}")

    def v_read(self, lds, slot, grp, q):
        lds.read(("V", grp), f"ds_read_b128 {ar(FRH(grp, q), 4)}, {vr(V2_L16V)} offset:{slot * TILE + VOFF(grp, q)}")

    def qk(self, p, lds, dst, first, fill, ahead):
        NB = self.NB
        for n in range(self.G):
            if NB == 2:
                kb, s_, x, j = n >> 4, (n >> 1) & 7, n & 1, n >> 1
            else:
                kb, s_, x, j = n >> 3, n & 7, 0, n
            grp, q = j >> 2, j & 3
            if n % (4 * NB) == 0:
                lds.need(("K", grp))
            acc = vr(S(dst, 2 * x + kb), 16)
            c = ("0" if first else vr(NEGM[x], 16)) if s_ == 0 else acc
            p.e(f"MFMA {acc}, {ar(FRH(grp, q), 4)}, {ar((QB if x else QA)(s_), 4)}, {c}")
            if n % (4 * NB) < 4:
                ahead(grp, n % (4 * NB))
            fill(n)

    def pv(self, p, lds, pbuf, fill, ahead):
        NB = self.NB
        for n in range(self.G):
            if NB == 2:
                kk, td, x = n >> 3, (n >> 1) & 3, n & 1
            else:
                kk, td, x = n >> 2, n & 3, 0
            if n % (4 * NB) == 0:
                lds.need(("V", kk))
            acc = ar((OB if x else OA)(td), 16)
            pb = vr(S(pbuf, 2 * x + (kk >> 1)) + 8 * (kk & 1), 4)
            p.e(f"MFMA {acc}, {ar(FRH(kk, td), 4)}, {pb}, {acc}")
            if n % (4 * NB) < 4:
                ahead(kk, n % (4 * NB))
            fill(n)

    def max_step(self, p, buf, kb, k):
        for X in range(self.NB):
            c, b = V_C[2 * X + kb], S(buf, 2 * X + kb)
            if k == 0:
                p.e(f"v_max3_f32 {vr(c)}, {vr(b)}, {vr(b + 1)}, {vr(b + 2)}")
            else:
                p.e(f"v_max3_f32 {vr(c)}, {vr(c)}, {vr(b + 2 * k + 1)}, {vr(b + 2 * k + 2)}")

    def max_last(self, p, buf, dA, dB):
        p.e(f"v_max3_f32 {vr(dA)}, {vr(V_C[0])}, {vr(V_C[1])}, {vr(S(buf, 0) + 15)}")
        if self.NB == 2:
            p.e(f"v_max3_f32 {vr(dB)}, {vr(V_C[2])}, {vr(V_C[3])}, {vr(S(buf, 2) + 15)}")
        p.e(f"v_max_f32 {vr(dA)}, {vr(dA)}, {vr(S(buf, 1) + 15)}")
        if self.NB == 2:
            p.e(f"v_max_f32 {vr(dB)}, {vr(dB)}, {vr(S(buf, 3) + 15)}")

    def mask(self, p, buf, s_tile64):
        p.e(f"v_mov_b32 {vr(V_NINF)}, 0xff800000")
        p.e(f"s_sub_i32 {sr(S_X1)}, %[N], {sr(s_tile64)}")
        for kb in range(2):
            for r in range(16):
                const = kb * 32 + (r & 3) + 8 * (r >> 2)
                p.e(f"s_sub_i32 {sr(S_X2)}, {sr(S_X1)}, {const}")
                p.e(f"v_cmp_ge_i32 vcc, {vr(V_HH4)}, {sr(S_X2)}")
                for X in range(self.NB):
                    reg = S(buf, 2 * X + kb) + r
                    p.e(f"v_cndmask_b32 {vr(reg)}, {vr(reg)}, {vr(V_NINF)}, vcc")

    # ---- softmax shadow
    def late(self, p, cb, n):
        """gap n of phase A: late exponentials of tile T (+ their lagged row sums) and pack n, in place"""
        flush_lag(p)
        if self.NB == 2:
            pa = 0 if n < 8 else 2 if n < 16 else 1 if n < 24 else 3
            pi = n & 7
            if n < 24:
                ea, er = (1, 8 + n) if n < 8 else (3, n - 8)
                self.exp(p, ea >> 1, S(cb, ea) + er)
        else:
            a1 = S(cb, 1)
            for r in ([a1, a1 + 1] if n == 0 else [a1 + n + 1] if n <= 14 else []):
                self.exp(p, 0, r)
            pa, pi = (0, n) if n < 8 else (1, n - 8)
        lo, dst = S(cb, pa) + 2 * pi, S(cb, pa) + 8 * (pi >> 2) + (pi & 3)
        p.e(f"PK {vr(dst)}, {vr(lo)}, {vr(lo + 1)}")
        if n == self.G - 1:
            flush_lag(p)

    N_EARLY = {2: 20, 1: 8}

    def early(self, p, buf, e):
        """early exponentials of tile T + 1, pair e: whole: (A, kb 0) 0..7, (B, kb 0) 8..15, (A, kb 1, registers 0..7) 16..19; half: (A, kb 0) 0..7"""
        flush_lag(p)
        if self.NB == 2:
            a = 0 if e < 8 else 2 if e < 16 else 1
        else:
            a = 0
        r = S(buf, a) + 2 * (e & 7)
        self.exp(p, a >> 1, r)
        self.exp(p, a >> 1, r + 1)
        if e == self.N_EARLY[self.NB] - 1:
            flush_lag(p)

    # ---- schedule tables: gaps of the phases
    def sched(self):
        if self.NB == 2:
            return dict(max0=range(24, 31), k_dma=(13, 15, 29, 31), v_dma=(1, 2, 3, 4), max1=[(g, (g - 2,)) for g in range(2, 9)], last=9, test=10, early0=12)
        return dict(max0=None, k_dma=(8, 9, 14, 15), v_dma=(1, 2, 3, 4), max1=[(2, (0, 1)), (3, (2, 3)), (4, (4, 5)), (5, (6,))], last=5, test=6, early0=7)

    def iteration(self, p, ph, steady):
        """iteration IT with IT & 3 == ph: phase A S(IT+1) = K(IT+1) Q^T - m, phase B O^T += V^T(IT) P(IT)^T"""
        NB, G, sc = self.NB, self.G, self.sched()
        cb, nb = ph & 1, (ph & 1) ^ 1
        tag = f"{'SD'[0 if steady else 1]}{ph}"
        kslot_r, kslot_la, vslot_r = (ph + 1) & 3, (ph + 2) & 3, ph
        kdst, vdst = ph * TILE, V_RING + ((ph + 2) & 3) * TILE
        lds = LdsQ(p, [("K", 0)] * 4 + [("K", 1)] * 4)
        if not steady:
            # T = tlo + nt - 1 - REM (absolute index of the current tile); which of this iteration's requests exist
            p.e(f"s_sub_i32 {sr(S_T)}, %[nt], {sr(S_REM)}")
            p.e(f"s_add_i32 {sr(S_T)}, {sr(S_T)}, %[tlo]")
            p.e(f"s_sub_i32 {sr(S_T)}, {sr(S_T)}, 1")
            p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 3")
            p.e(f"s_cselect_b32 {sr(S_HK)}, 1, 0")
            p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 1")
            p.e(f"s_cselect_b32 {sr(S_HV)}, 1, 0")

        def dma(rs, soff, j, flag, name):
            if steady:
                dma_piece(p, rs, soff, j)
            else:
                p.e(f"s_cmp_eq_u32 {sr(flag)}, 0")
                p.br("s_cbranch_scc1", self.L(name))
                dma_piece(p, rs, soff, j)
                p.label(self.L(name))

        def fa(n):
            self.late(p, cb, n)
            if NB == 2:
                if n in sc["max0"]:
                    self.max_step(p, nb, 0, n - sc["max0"][0])
            else:
                if 10 <= n <= 15:
                    self.max_step(p, nb, 0, n - 10)
                if n == 15:
                    self.max_step(p, nb, 0, 6)
            if steady:
                if n == 0:
                    p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_SOFFK)}, {TILE}")
                if n == 2:
                    p.e(f"s_sub_i32 {sr(S_REM)}, {sr(S_REM)}, 1")
                if n == 3:
                    p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_SOFFV)}, {TILE}")
            else:
                if n == 0:
                    soff_of(p, S_SOFFK, S_T, 4)
                if n == 2:
                    p.e(f"s_sub_i32 {sr(S_REM)}, {sr(S_REM)}, 1")
                if n == 3:
                    soff_of(p, S_SOFFV, S_T, 2)
            if n == 1:
                p.e(f"s_add_i32 m0, %[dbase], {kdst}")
            for j, g in enumerate(sc["k_dma"]):
                if g == n:
                    dma("%[rk]", S_SOFFK, j, S_HK, f"KK{tag}{j}")

        def ahead_a(grp, q):
            if grp < 2:
                self.k_read(lds, kslot_r, grp + 2, q)
            else:
                self.v_read(lds, vslot_r, grp - 2, q)

        self.qk(p, lds, nb, False, fa, ahead_a)

        def fb(n):
            if n == 0:
                p.e(f"s_add_i32 m0, %[dbase], {vdst}")
            for j, g in enumerate(sc["v_dma"]):
                if g == n:
                    dma("%[rv]", S_SOFFV, j, S_HV, f"KV{tag}{j}")
            if n == 2 and not steady:
                # the next tile may be the last of the sequence and ragged: mask its keys >= N
                p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_T)}, 1")
                p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 6")
                p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
                p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
                p.br("s_cbranch_scc1", self.L(f"MASK{tag}"))
                p.label(self.L(f"NOMASK{tag}"))
                p.begin_cold(self.L(f"MASK{tag}"), self.L(f"NOMASK{tag}"))
                p.e("s_nop 7")
                self.mask(p, nb, S_X0)
                p.end_cold()
            for g, ks in sc["max1"]:
                if g == n:
                    for k in ks:
                        self.max_step(p, nb, 1, k)
            if n == sc["last"]:
                self.max_last(p, nb, V_MXA, V_MXB)
            if n == sc["test"]:
                if NB == 2:
                    p.e(f"v_max_f32 {vr(V_T0)}, {vr(V_MXA)}, {vr(V_MXB)}")
                    p.e(f"v_cmp_lt_f32 vcc, 0x41000000, {vr(V_T0)}")           # 8.0 < max (any lane: the halves are only combined in the move)
                else:
                    p.e(f"v_cmp_lt_f32 vcc, 0x41000000, {vr(V_MXA)}")
                p.br("s_cbranch_vccnz", self.L(f"MOVE{tag}"))
                p.label(self.L(f"NOMOVE{tag}"))
                p.begin_cold(self.L(f"MOVE{tag}"), self.L(f"NOMOVE{tag}"))
                # ---- the reference maximum moves (rare): scores of tile T+1, seeds, row sums now; O^T after this phase's MFMAs (VCC stays set)
                xhalf_max(p, V_MXA, V_T0)
                if NB == 2:
                    xhalf_max(p, V_MXB, V_T1)
                p.e(f"v_max_f32 {vr(V_T0)}, 0, {vr(V_MXA)}")
                p.e(f"v_add_f32 {vr(V_MA)}, {vr(V_MA)}, {vr(V_T0)}")
                p.e(f"v_exp_f32 {vr(V_ALA)}, -{vr(V_T0)}")
                if NB == 2:
                    p.e(f"v_max_f32 {vr(V_T1)}, 0, {vr(V_MXB)}")
                    p.e(f"v_add_f32 {vr(V_MB)}, {vr(V_MB)}, {vr(V_T1)}")
                    p.e(f"v_exp_f32 {vr(V_ALB)}, -{vr(V_T1)}")
                p.e("s_nop 0")
                for X in range(NB):
                    for acc in (V2_LA, V2_LB)[X]:
                        p.e(f"v_mul_f32 {vr(acc)}, {vr(acc)}, {vr((V_ALA, V_ALB)[X])}")
                for X in range(NB):
                    for kb in range(2):
                        for r in range(16):
                            reg = S(nb, 2 * X + kb) + r
                            p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr((V_T0, V_T1)[X])}")
                for r in range(16):
                    for X in range(NB):
                        p.e(f"v_sub_f32 {vr(NEGM[X] + r)}, 0, {vr((V_MA, V_MB)[X])}")
                p.end_cold()
            if n >= sc["early0"] and n - sc["early0"] < self.N_EARLY[NB]:
                self.early(p, nb, n - sc["early0"])

        def ahead_b(kk, q):
            if kk < 2:
                self.v_read(lds, vslot_r, kk + 2, q)
            else:
                self.k_read(lds, kslot_la, kk - 2, q)

        self.pv(p, lds, cb, fb, ahead_b)
        assert lds.q == [("K", 0)] * 4 + [("K", 1)] * 4 and not p.lag
        # pending rescale of O^T: the move test's VCC
        p.br("s_cbranch_vccnz", self.L(f"PEND{tag}"))
        p.label(self.L(f"NOPEND{tag}"))
        p.begin_cold(self.L(f"PEND{tag}"), self.L(f"NOPEND{tag}"))
        p.e("s_nop 15")
        p.e("s_nop 7")
        for k in range(64 * NB):
            p.e(f"v_accvgpr_read_b32 {vr(V_T0)}, {ar(k)}")
            p.e(f"v_mul_f32 {vr(V_T0)}, {vr(V_T0)}, {vr(V_ALA if k < 64 else V_ALB)}")
            p.e(f"v_accvgpr_write_b32 {ar(k)}, {vr(V_T0)}")
        p.e("s_nop 3")
        p.end_cold()
        if steady:
            p.e("s_waitcnt vmcnt(8)")
        else:
            p.e(f"s_cmp_eq_u32 {sr(S_HK)}, 0")
            p.br("s_cbranch_scc1", self.L(f"W4{tag}"))
            p.e("s_waitcnt vmcnt(8)")
            p.br("s_branch", self.L(f"WD{tag}"))
            p.label(self.L(f"W4{tag}"))
            p.e(f"s_cmp_eq_u32 {sr(S_HV)}, 0")
            p.br("s_cbranch_scc1", self.L(f"W0{tag}"))
            p.e("s_waitcnt vmcnt(4)")
            p.br("s_branch", self.L(f"WD{tag}"))
            p.label(self.L(f"W0{tag}"))
            p.e("s_waitcnt vmcnt(0)")
            p.label(self.L(f"WD{tag}"))
        p.e("s_barrier")

    def tail(self, p, ph):
        """the sequence's last tile (IT & 3 == ph): late exponentials + packs, then its O^T MFMAs"""
        cb = ph & 1
        self.sum_i = [0, 0]
        for n in range(self.G):
            self.late(p, cb, n)
        lds = LdsQ(p, ["old"] * 8)
        for g in range(2):
            for q in range(4):
                self.v_read(lds, ph, g, q)
        self.pv(p, lds, cb, lambda n: None, lambda kk, q: self.v_read(lds, ph, kk + 2, q) if kk < 2 else None)

    def core(self):
        NB = self.NB
        p = Prog()
        if NB == 1:
            p.label("HCORE")
        else:
            # the stream of this WAVE (%[half]): 0 both query blocks, 1 block A only, 2 no live block (requests and barriers only)
            p.e("s_cmp_eq_u32 %[half], 1")
            p.br("s_cbranch_scc1", "HCORE")
            p.e("s_cmp_eq_u32 %[half], 2")
            p.br("s_cbranch_scc1", "PCORE")
        p.e(f"v_mov_b32 {vr(V_L16)}, %[lane16]")
        p.e(f"v_add_u32_e32 {vr(V2_L16V)}, {V_RING}, {vr(V_L16)}")
        p.e(f"v_mov_b32 {vr(V_VOFF)}, %[vlane]")
        p.e(f"v_mov_b32 {vr(V_HH4)}, %[hh4]")
        for X in range(NB):
            for acc in (V2_LA, V2_LB)[X]:
                p.e(f"v_mov_b32 {vr(acc)}, 0")
        p.e(f"s_sub_i32 {sr(S_REM)}, %[nt], 1")
        p.e(f"s_mov_b32 {sr(S_T)}, %[tlo]")
        p.e(f"s_sub_i32 {sr(S_NT32M1)}, %[nt32], 1")
        # steady soffsets: K(IT + 4) / V(IT + 2) of iteration IT after its add
        p.e(f"s_lshl_b32 {sr(S_X0)}, %[tlo], 14")
        p.e(f"s_lshl_b32 {sr(S_X1)}, %[wh], 13")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_X0)}, {sr(S_X1)}")
        p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_X0)}, %[wq]")
        p.e(f"s_add_i32 {sr(S_SOFFK)}, {sr(S_X0)}, {3 * TILE}")
        p.e(f"s_add_i32 {sr(S_SOFFV)}, {sr(S_X0)}, {1 * TILE}")
        # ---- the unit's first requests but K(3) have landed (Q too); a finished unit's stores may fly on (whole: >= 16 behind a whole
        # unit; half: >= 8, a half unit's one block in 16-bit rows - and a half unit never precedes a whole one)
        st = 8        # stores a finished unit's wave has issued at least (one block in 16-bit rows: the wave may have run the half / passive stream)
        p.e("s_cmp_eq_u32 %[first], 0")
        p.br("s_cbranch_scc1", self.L("WNF"))
        p.e("s_cmp_gt_i32 %[nt], 3")
        p.br("s_cbranch_scc1", self.L("WF4"))
        p.e("s_waitcnt vmcnt(0)")
        p.br("s_branch", self.L("WDONE"))
        p.label(self.L("WF4"))
        p.e("s_waitcnt vmcnt(4)")
        p.br("s_branch", self.L("WDONE"))
        p.label(self.L("WNF"))
        p.e("s_cmp_gt_i32 %[nt], 3")
        p.br("s_cbranch_scc1", self.L("WN4"))
        p.e(f"s_waitcnt vmcnt({st})")
        p.br("s_branch", self.L("WDONE"))
        p.label(self.L("WN4"))
        p.e(f"s_waitcnt vmcnt({st + 4})")
        p.label(self.L("WDONE"))
        p.e("s_barrier")
        # ---- tile 0: scores with a zero seed (the O^T accumulators are zeroed in its gaps), first reference maximum, early exponentials
        lds = LdsQ(p)
        for g in range(2):
            for q in range(4):
                self.k_read(lds, 0, g, q)

        def f0(n):
            for k in range(4):
                p.e(f"v_accvgpr_write_b32 {ar(4 * n + k)}, 0")

        self.qk(p, lds, 0, True, f0, lambda grp, q: self.k_read(lds, 0, grp + 2, q) if grp < 2 else None)
        p.e("s_nop 15")
        p.e("s_nop 7")
        p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_T)}, 6")
        p.e(f"s_add_i32 {sr(S_X1)}, {sr(S_X0)}, 64")
        p.e(f"s_cmp_gt_i32 {sr(S_X1)}, %[N]")
        p.br("s_cbranch_scc1", self.L("MASK0"))
        p.label(self.L("NOMASK0"))
        p.begin_cold(self.L("MASK0"), self.L("NOMASK0"))
        self.mask(p, 0, S_X0)
        p.end_cold()
        for k in range(7):
            self.max_step(p, 0, 0, k)
            self.max_step(p, 0, 1, k)
        self.max_last(p, 0, V_MA, V_MB)
        xhalf_max(p, V_MA, V_T0)
        if NB == 2:
            xhalf_max(p, V_MB, V_T1)
        for X in range(NB):
            for kb in range(2):
                for r in range(16):
                    reg = S(0, 2 * X + kb) + r
                    p.e(f"v_sub_f32 {vr(reg)}, {vr(reg)}, {vr((V_MA, V_MB)[X])}")
        for r in range(16):
            for X in range(NB):
                p.e(f"v_sub_f32 {vr(NEGM[X] + r)}, 0, {vr((V_MA, V_MB)[X])}")
        self.sum_i = [0, 0]
        for e in range(self.N_EARLY[NB]):
            self.early(p, 0, e)
        p.e("s_barrier")                                                       # every wave has read K(0): its slot may take K(4)
        p.e(f"s_cmp_lt_i32 {sr(S_REM)}, 1")
        p.br("s_cbranch_scc1", self.L("TAIL_0"))
        for g in range(2):                                                     # K(1) groups 0, 1 for the first iteration
            for q in range(4):
                p.e(f"ds_read_b128 {ar(FRH(g, q), 4)}, {vr(V_L16)} offset:{1 * TILE + (4 * g + q) * 1024}")
        # ---- the loop, unrolled over the four ring phases; steady bodies fall through into each other
        p.e(f"s_cmp_lt_i32 {sr(S_REM)}, 5")
        p.br("s_cbranch_scc1", self.L("DRAIN_0"))
        for ph in range(4):
            p.label(self.L(f"STEADY_{ph}"))
            self.sum_i = [0, 0]
            self.iteration(p, ph, True)
            if ph < 3:
                p.e(f"s_cmp_lt_i32 {sr(S_REM)}, 5")
                p.br("s_cbranch_scc1", self.L(f"ENTRY_{ph + 1}"))
            else:
                p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 4")
                p.br("s_cbranch_scc1", self.L("STEADY_0"))
        for ph in (0, 1, 2, 3):                                                # (ENTRY_0 first: STEADY_3 falls into it)
            p.label(self.L(f"ENTRY_{ph}"))
            p.e(f"s_cmp_lt_i32 {sr(S_REM)}, 1")
            p.br("s_cbranch_scc1", self.L(f"TAIL_{ph}"))
            p.label(self.L(f"DRAIN_{ph}"))
            self.sum_i = [0, 0]
            self.iteration(p, ph, False)
            if ph == 3:
                p.br("s_branch", self.L("ENTRY_0"))
        for ph in range(4):
            p.label(self.L(f"TAIL_{ph}"))
            self.tail(p, ph)
            if ph < 3:
                p.br("s_branch", self.L("END"))
        p.label(self.L("END"))
        p.e("s_barrier")                                                       # the rings are free for the next unit's first requests
        p.e("s_nop 15")
        p.e("s_nop 7")
        p.e(f"v_add_f32 %[o_la], {vr(V2_LA[0])}, {vr(V2_LA[1])}")
        p.e(f"v_mov_b32 %[o_ma], {vr(V_MA)}")
        if NB == 2:
            p.e(f"v_add_f32 %[o_lb], {vr(V2_LB[0])}, {vr(V2_LB[1])}")
            p.e(f"v_mov_b32 %[o_mb], {vr(V_MB)}")
        else:
            p.e("v_mov_b32 %[o_lb], 0")
            p.e("v_mov_b32 %[o_mb], 0")
        p.br("s_branch", "EXIT")
        p.finish()
        return p


def passive_v2():
    """the stream of a wave without a live query block (a ragged unit's waves past the last block): its share of the tile requests and
    every barrier of the unit, nothing else - no MFMA, no LDS read, no softmax (the chip is power-limited under dense MFMA: garbage
    blocks cost everybody clock)"""
    p = Prog()
    p.label("PCORE")
    p.e(f"v_mov_b32 {vr(V_VOFF)}, %[vlane]")
    p.e(f"s_sub_i32 {sr(S_REM)}, %[nt], 1")
    p.e(f"s_sub_i32 {sr(S_NT32M1)}, %[nt32], 1")
    p.e("s_cmp_eq_u32 %[first], 0")
    p.br("s_cbranch_scc1", "PWNF")
    p.e("s_waitcnt vmcnt(0)")
    p.br("s_branch", "PWDONE")
    p.label("PWNF")
    p.e("s_waitcnt vmcnt(8)")
    p.label("PWDONE")
    p.e("s_barrier")                      # (core start)
    p.e("s_barrier")                      # (tile 0 read)
    p.label("PLOOP")
    p.e(f"s_cmp_lt_i32 {sr(S_REM)}, 1")
    p.br("s_cbranch_scc1", "PEND_")
    # T = tlo + nt - 1 - REM, IT = nt - 1 - REM
    p.e(f"s_sub_i32 {sr(S_X2)}, %[nt], {sr(S_REM)}")
    p.e(f"s_sub_i32 {sr(S_X2)}, {sr(S_X2)}, 1")                    # IT
    p.e(f"s_add_i32 {sr(S_T)}, {sr(S_X2)}, %[tlo]")
    p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 3")
    p.br("s_cbranch_scc0", "PNOK")
    soff_of(p, S_SOFFK, S_T, 4)
    p.e(f"s_and_b32 {sr(S_X0)}, {sr(S_X2)}, 3")
    p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 14")
    p.e(f"s_add_i32 m0, {sr(S_X0)}, %[dbase]")
    p.e("s_nop 0")
    for j in range(4):
        dma_piece(p, "%[rk]", S_SOFFK, j)
    p.label("PNOK")
    p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 1")
    p.br("s_cbranch_scc0", "PNOV")
    soff_of(p, S_SOFFV, S_T, 2)
    p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_X2)}, 2")
    p.e(f"s_and_b32 {sr(S_X0)}, {sr(S_X0)}, 3")
    p.e(f"s_lshl_b32 {sr(S_X0)}, {sr(S_X0)}, 14")
    p.e(f"s_add_i32 {sr(S_X0)}, {sr(S_X0)}, {V_RING}")
    p.e(f"s_add_i32 m0, {sr(S_X0)}, %[dbase]")
    p.e("s_nop 0")
    for j in range(4):
        dma_piece(p, "%[rv]", S_SOFFV, j)
    p.label("PNOV")
    # everything issued before this iteration has landed (the pieces of this iteration: 8, 4 or none)
    p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 3")
    p.br("s_cbranch_scc1", "PW8")
    p.e(f"s_cmp_gt_i32 {sr(S_REM)}, 1")
    p.br("s_cbranch_scc1", "PW4")
    p.e("s_waitcnt vmcnt(0)")
    p.br("s_branch", "PWD")
    p.label("PW4")
    p.e("s_waitcnt vmcnt(4)")
    p.br("s_branch", "PWD")
    p.label("PW8")
    p.e("s_waitcnt vmcnt(8)")
    p.label("PWD")
    p.e("s_barrier")
    p.e(f"s_sub_i32 {sr(S_REM)}, {sr(S_REM)}, 1")
    p.br("s_branch", "PLOOP")
    p.label("PEND_")
    p.e("s_barrier")
    p.e("v_mov_b32 %[o_la], 0")
    p.e("v_mov_b32 %[o_ma], 0")
    p.e("v_mov_b32 %[o_lb], 0")
    p.e("v_mov_b32 %[o_mb], 0")
    p.label("EXIT")
    return p


def prologue_v2():
    """a unit's first requests: K(0), Q (straight into a[128:191]; a half unit - %[half] - block A only), V(0), K(1), V(1), K(2), K(3);
    tile i of the unit goes to ring slot i"""
    p = Prog()
    X0, X2 = 40, 42

    def tile(rs, add, dst):
        p.e(f"s_add_i32 {sr(X0)}, %[tlo], {add}")
        p.e(f"s_lshl_b32 {sr(X0)}, {sr(X0)}, 1")
        p.e(f"s_add_i32 {sr(X0)}, {sr(X0)}, %[wh]")
        p.e(f"s_min_i32 {sr(X0)}, {sr(X0)}, {sr(X2)}")
        p.e(f"s_lshl_b32 {sr(X0)}, {sr(X0)}, 13")
        p.e(f"s_add_i32 {sr(X0)}, {sr(X0)}, %[wq]")
        p.e(f"s_add_i32 m0, %[dbase], {dst}")
        p.e("s_nop 0")
        for j in range(4):
            p.e(f"buffer_load_dwordx4 %[vlane], {rs}, {sr(X0)} offen offset:{j * 1024} lds")

    p.e(f"s_sub_i32 {sr(X2)}, %[nt32], 1")
    tile("%[rk]", 0, 0)
    p.e("v_add_u32_e32 v0, 0x1000, %[qa]")
    p.e("v_add_u32_e32 v1, 0x1000, %[qb]")
    for s_ in range(8):
        p.e(f"global_load_dwordx4 {ar(QA(s_), 4)}, {'%[qa]' if s_ < 4 else 'v0'}, %[qbase] offset:{(s_ & 3) * 1024}")
    p.e("s_cmp_lg_u32 %[half], 0")
    p.br("s_cbranch_scc1", "NOQB")
    for s_ in range(8):
        p.e(f"global_load_dwordx4 {ar(QB(s_), 4)}, {'%[qb]' if s_ < 4 else 'v1'}, %[qbase] offset:{(s_ & 3) * 1024}")
    p.label("NOQB")
    tile("%[rv]", 0, V_RING)
    p.e("s_cmp_lt_i32 %[nt], 2")
    p.br("s_cbranch_scc1", "PDONE")
    tile("%[rk]", 1, TILE)
    tile("%[rv]", 1, V_RING + TILE)
    p.e("s_cmp_lt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "PDONE")
    tile("%[rk]", 2, 2 * TILE)
    p.e("s_cmp_lt_i32 %[nt], 4")
    p.br("s_cbranch_scc1", "PDONE")
    tile("%[rk]", 3, 3 * TILE)
    p.label("PDONE")
    return p


def prologue():
    """a unit's first requests: K(0), Q (straight into a[128:191]; a half unit - %[half] - block A only), V(0), K(1), V(1), K(2), K(3)"""
    p = Prog()
    X0, X1, X2 = 40, 41, 42

    def tile(rs, add, slot_expr_reg):
        # soffset of tile tlo + add for this wave, LDS destination in slot_expr_reg (already computed)
        p.e(f"s_add_i32 {sr(X0)}, %[tlo], {add}")
        p.e(f"s_lshl_b32 {sr(X0)}, {sr(X0)}, 1")
        p.e(f"s_add_i32 {sr(X0)}, {sr(X0)}, %[wh]")
        p.e(f"s_min_i32 {sr(X0)}, {sr(X0)}, {sr(X2)}")
        p.e(f"s_lshl_b32 {sr(X0)}, {sr(X0)}, 13")
        p.e(f"s_add_i32 {sr(X0)}, {sr(X0)}, %[wq]")
        p.e(f"s_mov_b32 m0, {sr(slot_expr_reg)}")
        p.e("s_nop 0")
        for j in range(4):
            p.e(f"buffer_load_dwordx4 %[vlane], {rs}, {sr(X0)} offen offset:{j * 1024} lds")

    def kdst(add):
        p.e(f"s_add_i32 {sr(X1)}, %[tlo], {add}")
        p.e(f"s_and_b32 {sr(X1)}, {sr(X1)}, 3")
        p.e(f"s_lshl_b32 {sr(X1)}, {sr(X1)}, 14")
        p.e(f"s_add_i32 {sr(X1)}, {sr(X1)}, %[dbase]")

    def vdst(slot):
        p.e(f"s_add_i32 {sr(X1)}, %[dbase], {V_RING + slot * TILE}")

    p.e(f"s_sub_i32 {sr(X2)}, %[nt32], 1")
    kdst(0); tile("%[rk]", 0, X1)
    p.e("v_add_u32_e32 v0, 0x1000, %[qa]")
    p.e("v_add_u32_e32 v1, 0x1000, %[qb]")
    for s in range(8):
        p.e(f"global_load_dwordx4 {ar(QA(s), 4)}, {'%[qa]' if s < 4 else 'v0'}, %[qbase] offset:{(s & 3) * 1024}")
    p.e("s_cmp_lg_u32 %[half], 0")
    p.br("s_cbranch_scc1", "NOQB")
    for s in range(8):
        p.e(f"global_load_dwordx4 {ar(QB(s), 4)}, {'%[qb]' if s < 4 else 'v1'}, %[qbase] offset:{(s & 3) * 1024}")
    p.label("NOQB")
    vdst(0); tile("%[rv]", 0, X1)
    p.e("s_cmp_lt_i32 %[nt], 2")
    p.br("s_cbranch_scc1", "PDONE")
    kdst(1); tile("%[rk]", 1, X1)
    vdst(1); tile("%[rv]", 1, X1)
    p.e("s_cmp_lt_i32 %[nt], 3")
    p.br("s_cbranch_scc1", "PDONE")
    kdst(2); tile("%[rk]", 2, X1)
    p.e("s_cmp_lt_i32 %[nt], 4")
    p.br("s_cbranch_scc1", "PDONE")
    kdst(3); tile("%[rk]", 3, X1)
    p.label("PDONE")
    return p


def epilogue(which, lp):
    """block `which` of the finished unit: a[64 which ..] * inv -> this wave's LDS rows -> 256-byte row segments to memory.
    fp32: two passes (64 d each); 16-bit: one pass.  Temps v0..v63.  Block B: skipped by a half unit (%[half])."""
    p = Prog()
    base = 64 * which
    if which:
        p.e("s_cmp_lg_u32 %[half], 0")
        p.br("s_cbranch_scc1", "NOEPI")
    if lp:
        # 16 ds_write_b64: (t, rq) -> row i, bytes (t*32 + 8*rq + 4*hh)*2 ; the per-lane part (i, hh) is in %[sw]
        for t in range(4):
            for rq in range(4):
                k = t * 4 + rq
                tmp = 4 * (k % 8)           # 8 rotating quads v0..v31
                for e in range(4):
                    p.e(f"v_accvgpr_read_b32 {vr(tmp + e)}, {ar(base + 16 * t + 4 * rq + e)}")
                for e in range(4):
                    p.e(f"v_mul_f32 {vr(tmp + e)}, {vr(tmp + e)}, %[inv]")
                p.e(f"PK {vr(32 + 2 * (k % 8))}, {vr(tmp)}, {vr(tmp + 1)}")
                p.e(f"PK {vr(33 + 2 * (k % 8))}, {vr(tmp + 2)}, {vr(tmp + 3)}")
                p.e(f"ds_write_b64 %[sw], {vr(32 + 2 * (k % 8), 2)} offset:{(t * 32 + 8 * rq) * 2}")
                if k % 8 == 7:
                    p.e("s_waitcnt lgkmcnt(0)")
        for k in range(8):
            p.e(f"ds_read_b128 {vr(4 * k, 4)}, %[sr] offset:{k * 4 * STAGE_ROW}")
        p.e("s_waitcnt lgkmcnt(0)")
        for k in range(8):
            p.e(f"buffer_store_dwordx4 {vr(4 * k, 4)}, %[o{k}], %[ro], 0 offen")
        p.e("s_nop 1")
    else:
        for half in range(2):
            for t2 in range(2):
                for rq in range(4):
                    k = t2 * 4 + rq
                    tmp = 32 + 4 * k        # 8 quads v32..v63
                    for e in range(4):
                        p.e(f"v_accvgpr_read_b32 {vr(tmp + e)}, {ar(base + 16 * (2 * half + t2) + 4 * rq + e)}")
                    for e in range(4):
                        p.e(f"v_mul_f32 {vr(tmp + e)}, {vr(tmp + e)}, %[inv]")
                    p.e(f"ds_write_b128 %[sw], {vr(tmp, 4)} offset:{(t2 * 32 + 8 * rq) * 4}")
            p.e("s_waitcnt lgkmcnt(0)")
            for k in range(8):
                p.e(f"ds_read_b128 {vr(4 * k, 4)}, %[sr] offset:{k * 4 * STAGE_ROW}")
            p.e("s_waitcnt lgkmcnt(0)")
            for k in range(8):
                p.e(f"buffer_store_dwordx4 {vr(4 * k, 4)}, %[o{k}], %[ro], 0 offen offset:{half * 256}")
            p.e("s_nop 1")
    if which:
        p.label("NOEPI")
    return p


def clobbers(vtop, stop, agprs, extra=()):
    items = [f'"v{i}"' for i in range(vtop + 1)] + [f'"a{i}"' for i in agprs] + [f'"s{i}"' for i in range(40, stop + 1)]
    items += [f'"{x}"' for x in extra]
    lines, cur = [], ""
    for it in items:
        if len(cur) + len(it) > 120:
            lines.append(cur); cur = ""
        cur += it + ", "
    lines.append(cur.rstrip(", "))
    return " \\\n    ".join(lines)


def main():
    parts = []
    parts.append("// GENERATED by tools/gen_attn_q64.py — do not edit (the generator holds the register map and the schedule).\n"
                 "// Instruction streams of the 64-queries-per-wave attention (attention_q64.hip); Q64_MFMA / Q64_PK are the mnemonics of the\n"
                 "// operand type (bf16 / fp16 build).\n")

    def macro(name, prog):
        body = prog.text().replace("\n", " \\\n")
        parts.append(f"#define {name} \\\n    {body}\n")

    if OPT_V1:
        both = core()
        both.hot.extend(core_h().hot)
        macro("Q64_ASM_CORE", both)
        macro("Q64_ASM_PROLOGUE", prologue())
    else:
        both = StreamV2(2, "").core()
        both.hot.extend(StreamV2(1, "H").core().hot)
        both.hot.extend(passive_v2().hot)
        macro("Q64_ASM_CORE", both)
        macro("Q64_ASM_PROLOGUE", prologue_v2())
    for which in (0, 1):
        macro(f"Q64_ASM_EPI_F32_{'AB'[which]}", epilogue(which, False))
        macro(f"Q64_ASM_EPI_LP_{'AB'[which]}", epilogue(which, True))
    parts.append("#define Q64_CLOBBER_CORE \\\n    " + clobbers(V_TOP if OPT_V1 else V2_TOP, S_TOP, range(256), ("vcc", "scc", "memory")) + "\n")
    parts.append("#define Q64_CLOBBER_PROLOGUE \\\n    " + clobbers(1, 42, range(128, 192), ("scc", "memory")) + "\n")
    parts.append("#define Q64_CLOBBER_EPI \\\n    " + clobbers(63, 39, (), ("scc", "memory")) + "\n")
    text = "\n".join(parts)
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    open(OUT, "w").write(text)
    n_inst = text.count("\\n\\t")
    print(f"wrote {os.path.normpath(OUT)}: {len(text)} bytes, {n_inst} instructions")


if __name__ == "__main__":
    main()
