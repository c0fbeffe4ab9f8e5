// attention_q64.hip — batch / long-form softmax attention of the DiT blocks (timm Attention core, reference model/dit.py:276,
// GeDEX-TTS/model/dit.py:268-290 is the DiTBlock around it) on fragment-ordered 16-bit operands (see attention_direct.hip for
// the layouts the row chain writes): the round-4 form with 64 QUERIES PER WAVE.
//
//   workgroup = 4 waves, ONE wave per SIMD, each wave owns 64 queries (two 32-query blocks A and B) and the whole 512-entry
//   register file: O^T (2 x 4 x 16 = 128), Q (2 x 8 fragments = 64) and the K / V^T fragment staging (2 x 4 fragments = 32) live
//   in the ACCUMULATION file, the two score buffers (2 x 64; the packed P of a tile replaces its scores IN PLACE), the -m
//   accumulator seeds (32) and the softmax state in the arch VGPRs.
//   Every K / V^T fragment read from LDS feeds TWO MFMAs (blocks A and B): 0.5 ds_read_b128 per MFMA (the 32-query forms: 1).
//   Key tiles are 64 keys (two of the producer's 32-key tiles = 16 KB of K + 16 KB of V^T), brought in by LDS-DMA
//   (buffer_load ... lds, 1 KB per wave instruction, 4 + 4 pieces per wave and tile, one M0 / one scalar offset per group: the
//   instruction's immediate offset moves the LDS destination too, tools/dmaprobe) into a 4-slot K ring and a 3-slot V^T ring,
//   two tiles ahead; ONE barrier per 64-key tile = per 64 MFMAs of a wave.
//   Per tile a wave runs two phases of 32 MFMAs, software-pipelined over tiles:
//     phase A(i): S(i+1) = K(i+1) Q^T - m   (accumulators SEEDED with -m: no subtraction in the softmax), key block 0 first
//                 VALU shadow: the rest of the exp2 / row sums of tile i, all 32 packs of P(i), the maxima of key block 0 of S(i+1)
//     phase B(i): O^T += V^T(i) P(i)^T
//                 VALU shadow: the maxima of key block 1, the (rare) reference-maximum move, 40 of the 64 exp2 / row sums of tile i+1
//   The reference maximum moves lazily (threshold 2^8, scores are in the log2 domain as the producer folds log2 e into q),
//   exactly as in the 32-query kernels; the move rescales O^T in the accumulation file after the tile's PV MFMAs.
//
//   THE INSTRUCTION STREAMS ARE GENERATED (tools/gen_attn_q64.py -> attention_q64_core.inc): one asm statement per unit with a
//   FIXED register map, every filler of every MFMA gap placed by hand.  The first form of this kernel left register allocation
//   to hipcc (inline-asm MFMAs with "a" / "v" operands, one statement per gap): it copied 16-register score tuples around
//   element updates, spilled Q fragments to scratch once the arch VGPRs ran out, padded statements with s_nop and, before the
//   fillers were made volatile, moved them out of their gaps altogether (profiles/round4_attention_q64_anatomy.txt).
//
//   Work units (batch element, head, group of 4 x 64 queries, key split) are walked by persistent workgroups; unit u runs on
//   XCD u % 8 and all units of one (element, head) share an XCD (its K and V^T cross the fabric once).  At a unit seam the next
//   unit's first tiles and its Q are requested BEFORE the finished unit's output leaves, and the output goes through a
//   wave-private LDS stage so that every store instruction writes whole 256-byte row segments.
//
//   ROUND 6 (what differs from the description above; DESIGN.md section 4 "Round 6"):
//   * unit shapes: a WHOLE unit is the one above; a HALF unit is 4 waves x ONE 32-query block on a block-A-only stream (16 + 16 MFMAs
//     per tile) - the second, shorter round of the persistent grid when the whole units alone fill it once and a bit (N = 1300: 4 whole
//     + 3 half units per (element, head) instead of 6 whole ones); its rows leave in the ordinary format, nobody merges anything;
//   * a stream per WAVE: whole / half / passive - a wave whose blocks lie past the sequence end only requests its share of the tiles
//     and keeps the barriers (all three streams have the same barrier count and DMA share per tile, so they mix inside a workgroup);
//   * the v2 streams: the loop is unrolled over the four ring slots (every LDS address an immediate; the V^T ring has four slots, the
//     fourth aliases the output stage), soffsets advance by one add, the pending-rescale flag is the VCC of the move test, fragments are
//     staged four sets deep and requested two groups ahead, row sums go to two accumulators per block.
// Output: normalised O per key split (+ (m, l) for the consumer's merge), the format attn_direct_ring_kernel writes.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "kernels.h"
#include "lp_util.h"
#include "kernels_lp.h"

namespace dex {
namespace DEX_LP_NS {

namespace {
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int HD = 128;
constexpr int KSLOTS = 4, VSLOTS = 3, TILE_BYTES = 16384;
constexpr int STAGE = (KSLOTS + VSLOTS) * TILE_BYTES;      // wave-private output stage: 32 rows x (256 + 16) bytes per wave
constexpr int STAGE_ROW = 272, STAGE_WAVE = 32 * STAGE_ROW;
constexpr int LDS_BYTES = STAGE + 4 * STAGE_WAVE;

#ifdef DEX_LP_F16
#define Q64_MFMA "v_mfma_f32_32x32x16_f16"
#define Q64_PK "v_cvt_pk_f16_f32"
#else
#define Q64_MFMA "v_mfma_f32_32x32x16_bf16"
#define Q64_PK "v_cvt_pk_bf16_f32"
#endif
#ifndef Q64_CORE_INC
#define Q64_CORE_INC "attention_q64_core.inc"
#endif
#include Q64_CORE_INC

struct Q64Unit { int bh, g, sp, T_lo, nt, tail, half, blk0; };      // blk0: the unit's first 32-query block

__device__ __forceinline__ float xhalf_sum(float x) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
}  // namespace

#ifdef Q64_STAMP
#define Q64_T(k) do { if (p.dbg) tst[k] = __builtin_readcyclecounter(); } while (0)
#else
#define Q64_T(k) do { } while (0)
#endif

// ng = groups of 256 queries per (element, head); nunits = 2 B ng ksplit; xcd_mode: unit u -> XCD u % 8 owns (element, head) pairs.
// Tail split (p.tail_ks > 1, p.ksplit == 1): the query groups g < p.tail_g are whole units and come FIRST in the unit order, the
// groups from tail_g on are split tail_ks ways over the keys and follow - a persistent workgroup then runs one long and one short unit
// instead of two long ones when the whole units alone fill the chip once (DEX B = 32, N = 1300: 384 whole units were two rounds of
// 256 CUs with the second half empty).  Whole units write O slot 0 (16-bit when o_lp), tail partials write fp32 slots 1 + sp and (m, l).
__global__ __launch_bounds__(256, 1) void attn_q64_kernel(const AttnDirectP p, const int ng, const int nunits, const int xcd_mode) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char q64_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const int N = p.N, ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nt32 = (N + 31) >> 5, nT = (nt32 + 1) >> 1;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)q64_smem;
    const unsigned lane16 = lds_base + lane * 16;
    const unsigned vlane = lane * 16, hh4 = 4 * hh;
    const long bh_bytes = (long)p.Npad * (HD * 2);            // one (element, head) operand: Npad rows x 128 x 16 bit
    const int wh = wave >> 1, wq = (wave & 1) * 4096;         // this wave's DMA share of a 64-key tile: 32-key half wh, 4 KB at wq
    const int dbase = __builtin_amdgcn_readfirstlane((int)lds_base) + wave * 4096;
#ifdef Q64_STAMP
    long long tst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    const int tks = p.tail_ks > 1 ? p.tail_ks : 1;
    const int half_n = p.half_n > 0 ? p.half_n : 0;          // half units (4 waves x ONE 32-query block) per (element, head) behind half_g whole groups
    const int tail_g = half_n ? p.half_g : tks > 1 ? p.tail_g : ng;
    const int nwhole = 2 * p.B * tail_g * ks;                 // units of the whole groups (all of them without a tail split / half units)
    // Unit decode without integer division (round 6): hipcc's signed 32-bit / 64-bit divisions by run-time values were ~400 SCALAR instructions
    // per decode - ~3.4k cycles for a lone wave (tools/valubench: ~8.5 cycles per scalar instruction) at the start of the launch and at every
    // unit seam, 7 % of a workgroup's time at N = 1300.  Quotients of small unsigned values through a float reciprocal + one correction step
    // (a < 2^22, b < 2^12: the truncated product is off by at most one either way).
    auto divmod = [](unsigned a, unsigned b, float rb, unsigned& q, unsigned& r) __attribute__((always_inline)) {
        q = (unsigned)((float)a * rb);
        int rem = (int)(a - q * b);
        if (rem < 0) { --q; rem += (int)b; }
        else if (rem >= (int)b) { ++q; rem -= (int)b; }
        r = (unsigned)rem;
    };
    // the two unit classes: whole groups (key split ks), then tail-split groups or half units
    const unsigned per_w = (unsigned)(tail_g * ks), per_s = (unsigned)(half_n ? half_n : (ng - tail_g) * tks);
    const float rcp_per_w = 1.f / (float)(per_w ? per_w : 1u), rcp_per_s = 1.f / (float)(per_s ? per_s : 1u);
    const float rcp_ks = 1.f / (float)ks, rcp_tks = 1.f / (float)tks;
    auto decode = [&](int unit) __attribute__((always_inline)) -> Q64Unit {
        Q64Unit u;
        const int second = unit >= nwhole ? 1 : 0;            // the unit order's second class: tail-split groups, or half units
        u.half = second && half_n ? 1 : 0;
        u.tail = second && !half_n ? 1 : 0;
        const unsigned uks = u.tail ? (unsigned)tks : u.half ? 1u : (unsigned)ks;
        const float rcp_uks = u.tail ? rcp_tks : u.half ? 1.f : rcp_ks;
        const unsigned per = second ? per_s : per_w;
        const float rcp_per = second ? rcp_per_s : rcp_per_w;
        unsigned r = (unsigned)(second ? unit - nwhole : unit), q;
        if (xcd_mode) { const unsigned x = r & 7u; divmod(r >> 3, per, rcp_per, q, r); u.bh = (int)(x + 8u * q); }
        else { divmod(r, per, rcp_per, q, r); u.bh = (int)q; }
        unsigned g = r, sp = 0;
        if (uks > 1) divmod(r, uks, rcp_uks, g, sp);           // (uniform)
        u.g = (int)g + (u.tail ? tail_g : 0); u.sp = (int)sp;
        u.blk0 = u.half ? 8 * tail_g + 4 * (int)r : 8 * u.g;
        if (uks > 1) {
            unsigned t0, t1, rem;
            divmod((unsigned)nT * sp, uks, rcp_uks, t0, rem);
            divmod((unsigned)nT * (sp + 1), uks, rcp_uks, t1, rem);
            u.T_lo = (int)t0; u.nt = (int)(t1 - t0);
        } else { u.T_lo = 0; u.nt = nT; }
        u.tail = __builtin_amdgcn_readfirstlane(u.tail); u.half = __builtin_amdgcn_readfirstlane(u.half);
        u.bh = __builtin_amdgcn_readfirstlane(u.bh); u.g = __builtin_amdgcn_readfirstlane(u.g); u.sp = __builtin_amdgcn_readfirstlane(u.sp);
        u.blk0 = __builtin_amdgcn_readfirstlane(u.blk0);
        u.T_lo = __builtin_amdgcn_readfirstlane(u.T_lo); u.nt = __builtin_amdgcn_readfirstlane(u.nt);
        return u;
    };
    auto rsrc_of = [&](const void* base, int bh) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(base) + bh * bh_bytes), 0, (int)bh_bytes, 0x00020000);
    };
    // The stream of THIS WAVE in a unit: 0 both of its 32-query blocks are live, 1 only block A is (a half unit's wave, or the wave of a
    // whole unit that holds the (element, head)'s last, odd block), 2 none is (the waves of a ragged unit past the last block: they only
    // issue their share of the tile requests and keep the barriers).  Round 6: before, such waves ran the full stream on clamped rows -
    // 3 to 12 % of a launch's MFMAs at the judged shapes, and under dense MFMA the chip is power-limited.
    auto wave_mode = [&](const Q64Unit& u) __attribute__((always_inline)) -> int {
        const int blkA = u.blk0 + (u.half ? wave : 2 * wave);
        return __builtin_amdgcn_readfirstlane(blkA >= nt32 ? 2 : (u.half || blkA + 1 >= nt32) ? 1 : 0);
    };
    // a unit's first requests (generated stream): K(0), Q -> a[128:191], V(0), K(1), V(1), K(2), K(3)
    auto issue_prologue = [&](const Q64Unit& u) __attribute__((always_inline)) {
        const auto rK = rsrc_of(p.Kh, u.bh), rV = rsrc_of(p.Vt, u.bh);
        const unsigned char* qbase = reinterpret_cast<const unsigned char*>(p.Qh) + u.bh * bh_bytes;
        const int blkA = u.blk0 + (u.half ? wave : 2 * wave);       // one asm for every stream (a half / passive wave skips the block-B loads inside)
        const int qtA = min(blkA, nt32 - 1), qtB = min(blkA + 1, nt32 - 1);
        const unsigned qa = vlane + qtA * 8192, qb = vlane + qtB * 8192;
        const int mode = wave_mode(u);
        asm volatile(Q64_ASM_PROLOGUE
                     :
                     : [rk] "s"(rK), [rv] "s"(rV), [tlo] "s"(u.T_lo), [nt] "s"(u.nt), [nt32] "s"(nt32), [wh] "s"(wh), [wq] "s"(wq), [dbase] "s"(dbase),
                       [vlane] "v"(vlane), [qa] "v"(qa), [qb] "v"(qb), [qbase] "s"(qbase), [half] "s"(mode)
                     : Q64_CLOBBER_PROLOGUE);
    };

    Q64Unit cur = decode(blockIdx.x);
    issue_prologue(cur);
    int first_unit = 1;
    // (no peeling: a peeled first iteration is a second copy of the 13k-instruction core statement)
#pragma clang loop unroll(disable)
    for (int unit = blockIdx.x;;) {
        const int bh = cur.bh, sp = cur.sp, oslot = cur.tail ? 1 + cur.sp : cur.sp;
        const int b = bh >> 1, h = bh & 1;
        const int rowA = (cur.half ? cur.blk0 + wave : cur.blk0 + 2 * wave) * 32;      // this wave's first query row (block B: + 32)
        const int mode = wave_mode(cur);
        Q64_T(0); Q64_T(1); Q64_T(2);
        float lA, lB, mA, mB;
        {
            const auto rK = rsrc_of(p.Kh, bh), rV = rsrc_of(p.Vt, bh);
            asm volatile(Q64_ASM_CORE               // both unit shapes: the statement dispatches on %[half] itself
                         : [o_la] "=&v"(lA), [o_lb] "=&v"(lB), [o_ma] "=&v"(mA), [o_mb] "=&v"(mB)
                         : [rk] "s"(rK), [rv] "s"(rV), [tlo] "s"(cur.T_lo), [nt] "s"(cur.nt), [nt32] "s"(nt32), [N] "s"(N), [wh] "s"(wh), [wq] "s"(wq),
                           [dbase] "s"(dbase), [first] "s"(first_unit), [half] "s"(mode), [lane16] "v"(lane16), [vlane] "v"(vlane), [hh4] "v"(hh4)
                         : Q64_CLOBBER_CORE);
        }
        Q64_T(3); Q64_T(4);
        // ---- seam: the next unit's first requests go out BEFORE this unit's output (the stores then cover their latency).  O^T of the
        // finished unit stays in a[0:127] (nothing between the core and the output statements may touch the accumulation file:
        // tools/isa_gaps.py --acc audits the build)
        const int next_unit = unit + gridDim.x;
        const bool has_next = next_unit < nunits;
        Q64Unit nxt = cur;
        if (has_next) { nxt = decode(next_unit); issue_prologue(nxt); }
        // ---- output: normalise, stage through this wave's LDS rows, store whole 256-byte row segments (fp32, or the mode's 16-bit
        // type), (m, l) for the consumer's merge of key splits.  Every store is a buffer store whose dead lanes point out of range:
        // the number of store instructions is fixed (>= 16: the next unit's first wait lets them fly); "out of range" is 2 GB, which the
        // instruction's immediate offset cannot wrap back into the buffer (0xffffffff + 256 did: 16 stray bytes in row 0).
        lA = xhalf_sum(lA); lB = xhalf_sum(lB);
        {
            const unsigned stg = lds_base + STAGE + wave * STAGE_WAVE;
            const int rsel = lane >> 4, c16 = lane & 15;
            const long orow = (long)N * 1024;                                   // one element's output: N rows x 256 fp32
            const unsigned sread = stg + rsel * STAGE_ROW + c16 * 16;
            const float invA = lA > 0.f ? 1.f / lA : 0.f, invB = lB > 0.f ? 1.f / lB : 0.f;
            if (p.o_lp && !cur.tail) {
                // (key splits, round 6: slot sp of the 16-bit partials sits o_sstride ELEMENTS behind slot sp - 1, as in the fp32 layout)
                const auto rO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.O) + (long)oslot * p.o_sstride * 2 + (long)b * (orow / 2), 0, (int)(orow / 2), 0x00020000);
                const unsigned swrite = stg + i * STAGE_ROW + hh * 8;
                unsigned o[8];
#define Q64_EPI(STREAM, INV, Q0, RB, HB)                                                                                                        \
                {                                                                                                                                \
                    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                                               \
                        const int row = (Q0) + k * 4 + rsel;                                                                                     \
                        o[k] = row < N ? (unsigned)(row * (RB) + h * (HB) + c16 * 16) : 0x80000000u;                                             \
                    }                                                                                                                            \
                    asm volatile(STREAM : : [inv] "v"(INV), [sw] "v"(swrite), [sr] "v"(sread), [ro] "s"(rO), [o0] "v"(o[0]), [o1] "v"(o[1]),   \
                                 [o2] "v"(o[2]), [o3] "v"(o[3]), [o4] "v"(o[4]), [o5] "v"(o[5]), [o6] "v"(o[6]), [o7] "v"(o[7]),                 \
                                 [half] "s"(mode)                                                                                               \
                                 : Q64_CLOBBER_EPI);                                                                                            \
                }
                Q64_EPI(Q64_ASM_EPI_LP_A, invA, rowA, 512, 256)
                Q64_EPI(Q64_ASM_EPI_LP_B, invB, rowA + 32, 512, 256)
            } else {
                const auto rO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.O + (long)oslot * p.o_sstride) + (long)b * orow, 0, (int)orow, 0x00020000);
                const unsigned swrite = stg + i * STAGE_ROW + hh * 16;
                unsigned o[8];
                Q64_EPI(Q64_ASM_EPI_F32_A, invA, rowA, 1024, 512)
                Q64_EPI(Q64_ASM_EPI_F32_B, invB, rowA + 32, 1024, 512)
            }
            if (p.ml && (ks > 1 || cur.tail)) {
                const auto rM = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.ml + ((((long)sp * p.B + b) * 2 + h) * N) * 2), 0, N * 8, 0x00020000);
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const int q = rowA + which * 32 + i;
                    const u32x2 v = {__float_as_uint(which ? mB : mA), __float_as_uint(which ? lB : lA)};
                    __builtin_amdgcn_raw_buffer_store_b64(v, rM, (hh == 0 && q < N) ? (unsigned)(q * 8) : 0x80000000u, 0, 0);
                }
            }
        }
        Q64_T(5);
#ifdef Q64_STAMP
        if (p.dbg && lane == 0) {
            long long* d = p.dbg + ((long)unit * 4 + wave) * 8;
            for (int k = 0; k < 6; ++k) d[k] = tst[k];
            d[6] = cur.nt | (mode << 16); d[7] = __builtin_amdgcn_s_getreg(0x14 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID
        }
#endif
        if (!has_next) break;
        cur = nxt; unit = next_unit; first_unit = 0;
    }
}

// Key split for the 64-query form: the split (1..max_split) that minimises  rounds of workgroups x (key tiles + per-unit overhead)
// + what every extra partial costs the consumer.  The row chain merges the partials while it loads its A tile: per extra partial
// one more fp32 read of the whole O (and O itself leaves as fp32 instead of the 16-bit form a single split allows) - measured at DEX
// B = 32, N = 1300: the 64-row chain 78 -> 94 us with two partials while the attention gained 4 us, so at batch size a split only
// pays when it fills an otherwise idle chip (long-form: one utterance, 40 workgroup-sized query groups for 256 CUs).
int attention_q64_ksplit(int N, int B, int max_split) {
    const int nt32 = (N + 31) / 32, nT = (nt32 + 1) / 2, ng = (nt32 + 7) / 8;
    const double tile_us = 1.35, unit_us = 5.0;                              // per 64-key tile of a unit / per unit (first tile, last tile, output)
    const double merge_us = (double)B * N * 256 * 4 / 4.0e6 + 1.0;           // one more partial through the consumer at ~4 TB/s
    int best = 1; double best_cost = 1e30;
    for (int ks = 1; ks <= max_split && ks <= nT; ++ks) {
        const long units = 2L * B * ng * ks;
        const long rounds = (units + 255) / 256;
        const double cost = rounds * (((nT + ks - 1) / ks) * tile_us + unit_us) + (ks > 1 ? merge_us * ks : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
    }
    return best;
}

// The unit plan: a uniform key split (above), or - at batch size, where a uniform split costs the consumer a merge of every row - whole
// units for the query groups below tail_g and a tail_ks-way split for the rest.  Makespan of the persistent grid (workgroup w runs
// units w, w + 256, ...; whole units first) + the consumer's merge of the tail rows; tail_ks = 1: no tail split.
void attention_q64_plan(int N, int B, int max_split, int* ks_out, int* tail_g_out, int* tail_ks_out) {
    const int nt32 = (N + 31) / 32, nT = (nt32 + 1) / 2, ng = (nt32 + 7) / 8;
    const double tile_us = 1.35, unit_us = 5.0;
    const double merge_us = (double)B * N * 256 * 4 / 4.0e6 + 1.0;
    const int ks = attention_q64_ksplit(N, B, max_split);
    *ks_out = ks; *tail_g_out = ng; *tail_ks_out = 1;
    if (ks > 1 || max_split < 2) return;
    auto makespan = [&](int tg, int tk) {
        const long nwhole = 2L * B * tg, ntail = 2L * B * (ng - tg) * tk;
        const double cw = nT * tile_us + unit_us, ct = ((nT + tk - 1) / tk) * tile_us + unit_us;
        // static deal: workgroup w of min(units, 256) gets units w, w + G, ...
        const long G = std::min<long>(nwhole + ntail, 256);
        double worst = 0;
        for (long w = 0; w < G; ++w) {
            double t = 0;
            for (long u = w; u < nwhole + ntail; u += G) t += u < nwhole ? cw : ct;
            worst = std::max(worst, t);
        }
        return worst;
    };
    double best = makespan(ng, 1);
    for (int tk = 2; tk <= std::min(max_split - 1, 4) && tk <= nT; ++tk)
        for (int tg = 1; tg < ng; ++tg) {
            const double rows_tail = std::min(1.0, std::max(0.0, (double)(N - tg * 256) / N));
            const double cost = makespan(tg, tk) + merge_us * tk * rows_tail;
            if (cost < best - 0.5) { best = cost; *tail_g_out = tg; *tail_ks_out = tk; }
        }
}

// Whole units + HALF units.  The 32-query blocks of an (element, head) are dealt as half_g whole units (8 blocks: 4 waves x 64 queries)
// and half_n half units (4 blocks: 4 waves x 32 queries, the block-A-only streams) behind them in the unit order, when that shortens the
// makespan of the persistent grid: DEX B = 32, N = 1300 has 41 blocks per (element, head) - as 6 whole units (the last one a single ragged
// block) 384 units = two full-length rounds of 256 workgroups with the second half empty; as 4 whole + 3 half units the second round is
// 192 half units at ~0.6 of the length.  Unlike a key split the half units write ordinary rows: the consumer sees no difference.
// Cost model in key tiles of a whole unit (per-unit overhead ~4.5 tiles: first-tile wait, tile 0, last tile, output; half unit ~0.6 x).
bool attention_q64_half_plan(int N, int B, int* half_g, int* half_n) {
    const int nt32 = (N + 31) / 32, nT = (nt32 + 1) / 2, ng = (nt32 + 7) / 8;
    const double cw = nT + 4.5, ch = 0.6 * nT + 3.5;
    auto makespan = [&](long nw, long nh) {
        const long G = std::min<long>(nw + nh, 256);
        double worst = 0;
        for (long w = 0; w < G; ++w) {
            double t = 0;
            for (long u = w; u < nw + nh; u += G) t += u < nw ? cw : ch;
            worst = std::max(worst, t);
        }
        return worst;
    };
    double best = makespan(2L * B * ng, 0);
    bool found = false;
    for (int hg = nt32 / 8; hg >= 0; --hg) {                 // more whole units first: a tie keeps the fewer half units
        const int hn = (nt32 - 8 * hg + 3) / 4;
        if (hn <= 0) continue;
        const double c = makespan(2L * B * hg, 2L * B * hn);
        if (c < best * 0.95) { best = c; *half_g = hg; *half_n = hn; found = true; }
    }
    return found;
}

void launch_attention_q64(const AttnDirectP& p0, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr = true;
    }
    AttnDirectP p = p0;
    const int nt32 = (p.N + 31) / 32, ng = (nt32 + 7) / 8, ks = p.ksplit > 1 ? p.ksplit : 1;
    const int xcd_mode = ((2 * p.B) % 8 == 0) ? 1 : 0;
    const int tks = p.tail_ks > 1 && ks == 1 ? p.tail_ks : 1, tg = tks > 1 ? p.tail_g : ng;
    int nunits = 2 * p.B * (tg * ks + (ng - tg) * tks);
    p.half_g = p.half_n = 0;
    if (ks == 1 && tks == 1) {
        const int env = p0.half_n < 0 ? 0 : knob_or("DEX_ATTN_Q64_HALF", 1);      // 0: whole units only; half_n < 0: the caller forbids them (tools)
        int hg = 0, hn = 0;
        if (p0.half_n > 0) { hg = p0.half_g; hn = p0.half_n; }                      // forced plan (tools / tests)
        else if (!(env && attention_q64_half_plan(p.N, p.B, &hg, &hn))) hn = 0;
        if (hn > 0) { p.half_g = hg; p.half_n = hn; nunits = 2 * p.B * (hg + hn); }
    }
    const int grid = nunits < 256 ? nunits : 256;
    hipLaunchKernelGGL(attn_q64_kernel, dim3(grid), dim3(256), LDS_BYTES, st, p, ng, nunits, xcd_mode);
}

}  // namespace DEX_LP_NS
}  // namespace dex
