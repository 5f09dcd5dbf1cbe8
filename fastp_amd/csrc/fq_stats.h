// fq_stats.h - Stats::statRead (src/stats.cpp:191-266) as its own streaming kernel (gfx950).
//
// The fused kernel kept the four Stats objects' counters in the LDS of ONE 1024-lane workgroup per CU, which
// pinned the whole per-read path to that geometry (12 workgroup barriers per tile, 4 waves per SIMD, the thin
// lane = read phases run by 2 of 16 wavefronts).  In the split plan (KernelArgs::split) the per-read path runs
// as small workgroups, several per CU, and leaves the kept length of every read in HBM (`swin`); this kernel
// then streams the packed rows once more - coalesced, no LDS tile, no workgroup barrier in the loop - and does
// nothing but count.  Legal when no option moves or edits a kept base (DevParams::stats_one_pass): base j of a
// read sits in cycle j before and after filtering, so one pass classifies it as
//     kept    (the read is written out and j < kept length)  -> accumulators of the POST slot
//     dropped (everything else)                               -> accumulators of the PRE slot
// and the slab fold forms PRE = kept + dropped, POST = kept (reduce_body, one_pass).
//
// lane = (read, 8 consecutive bases): one 8-byte quality load, one 2-byte base load (+ the dword / byte before
// them for the 5-mers that start in the previous item).  Counter layout in LDS: per-cycle accumulators sit
// [slot][base-in-item k][class][item h], so the 64 lanes of a DS instruction - consecutive items, the same k -
// land on consecutive 8-byte words when their classes agree and 6 banks apart per class step when not; the
// 40-dword stride of the fused kernel's [slot][cycle][class] layout put a wavefront on 4 banks.
#pragma once
#include "fq_intrin.h"

namespace fq {

struct StatsArgs {
    int n;                  // units (pairs or single reads) of this launch
    int paired;
    int sw_g, qw_g;         // batch row strides in dwords
    int H;                  // 8-base items per row = qw_g / 2
    int Hs;                 // items between two classes of the per-cycle table (>= H; 32 = one full set of banks: lanes
                            // that differ in class then never meet in a bank unless they share the item column)
    u32 magic_H;            // ceil(2^32 / H)
    int Cp;                 // cycles rounded up to a multiple of 4 (canonical slab layout, cyc_index)
    int units_per_block;    // a workgroup takes this many consecutive units (<= CYC_MAX_READS: packed counters)
    const u32* seq[2];
    const u32* qual[2];
    const u32* swin[2];     // per read: original length | kept length << 16 (0 kept = not written out), left by the scan kernel
    // LDS layout (dwords)
    int l_cyc;              // [4][8][N_CLS][Hs] u64
    int l_kmer;             // [4][KMER_BINS] u32
    int l_qh;               // [4][128][ST_QH_COPIES] u32: lane l adds to copy l % ST_QH_COPIES of a bin (same-address atomics
                            // serialise; the copies of a bin sit in consecutive banks)
    int l_lut;              // [256] x 4 dwords: character | kept << 7 -> {increment u64 (bit 0 = "a base"), per-cycle byte offset,
                            // k-mer byte offset}
    int l_mt;               // [9] x 2 dwords: the byte masks "first i of 8 bytes"
    int l_wl, wl_cap;       // work list of the items with an N among their 12 bases: [0] = count, then wl_cap item numbers
    int l_total;
    // slab (dwords): [cyc canonical: 4 * Cp * N_CLS u64][kmer 4 * KMER_BINS][qh 4 * 128]
    u32* slabs;
    int slab_dwords;
    u32 debug_skip;         // profiling only: 64 no per-cycle atomics, 128 no k-mer atomics, 256 no histogram atomics
    int form;               // 4: u32 cells, one mate's tables at a time (stats_body4, round 5); 3: round 3's packed u64 cells (stats_body)
    int kc;                 // form 4: copies of the 5-mer table (1, 2 or 4)
    int merge;              // form 4, DevParams::merge_lane: read 2 has no POST Stats object of its own - every base of it is "dropped" in its
                            // pass; a third pass (stats4_tail_pass) counts what swin[1] marks as kept once more into slot 3, the merged
                            // reads' second parts at the merged reads' cycles (the slab fold adds slot 3 to the POST Stats of read 1)
    int H16;                // form 5 (fq_stats5.h): 16-base items per row = ceil(qw_g / 4), Hs = the item columns of its table
    u32 magic_H16;          // ceil(2^32 / Hs): a lane number -> (unit of the trip, column of the block)
    int l_ovf;              // form 5: [2][Cp][N_CLS] u64 packed cells of the bases its joint table has no cell for (N, qualities above 'K')
    const u32* fr_rec[2];   // form 5: where a read's own front is read from - its result record (stride fr_stride = 3 dwords, low 16 bits)
    int fr_stride;          // with DevParams::front_per_read (--cut_front), else the mate's swin array with stride 0 (a load nobody uses)
    int front_per_read;
    int front[2];           // form 4, DevParams::front_lane: the kept bases of a read of mate m are [front[m], swin >> 16) - the same
                            // front for every read that is written out; they are counted at their ORIGINAL cycle (the slab fold
                            // moves the POST Stats, reduce_body); the 5-mers that end on the first four kept bases exist in the
                            // original read only (stats.cpp:224-227: five bases of the read the Stats object is given)
};

enum { ST_QH_COPIES = 8 };

FQ_DEV u64 stats_inc_of(u32 q) {   // stats.cpp:209-222: q30 ('?') counts into Q30 and Q20, q20 ('5') into Q20
    return 1ull | ((u64)(q >= 53u) << CYC_Q20_SHIFT) | ((u64)(q >= 63u) << CYC_Q30_SHIFT) | ((u64)(q - 33u) << CYC_QSUM_SHIFT);
}

// what a lane needs of item `it` (of one mate) of the workgroup's unit range
struct StatsItem {
    int h, rl0, lk;
    u32 q0, q1, qp, codes, prev8;
    bool act;
};
// qual / seq / swin: the mate's arrays at the workgroup's first unit.  A row has H 8-byte quality items, so item `it`
// of the range is simply the it-th 8-byte word behind `qual`.
FQ_DEV void stats_fetch(const StatsArgs& a, const u32* qual, const u32* seq, const u32* swin, int it, bool tv, StatsItem& s) {
    const u32 ur = fastdiv((u32)(tv ? it : 0), a.magic_H);
    s.h = (tv ? it : 0) - (int)mul24(ur, (u32)a.H);     // (units < 2^14, H < 2^8: the full-rate 24-bit multiply)
    u32 sw = 0;
    s.q0 = s.q1 = s.qp = s.codes = s.prev8 = 0;
    if (tv) {   // the item's 8 quality bytes and 8 bases, the 4 of each before them - all loads independent of each other:
                // the bytes of a row behind its read are zeros (fastp_gpu.h), i.e. "no base" characters, whatever swin says
        sw = swin[ur];
        const u64 qq = ((const u64*)qual)[(u32)it];
        s.q0 = (u32)qq;
        s.q1 = (u32)(qq >> 32);
        const u32 sb = mul24(ur, (u32)(a.sw_g * 4)) + 2u * (u32)s.h;   // byte of the row's packed bases
        s.codes = (u32)*(const u16*)((const u8*)seq + sb);
        if (s.h > 0) {
            s.qp = qual[2u * (u32)it - 1u];
            s.prev8 = (u32)((const u8*)seq)[sb - 1u];
        }
    }
    s.rl0 = (int)(sw & 0xFFFFu);
    s.lk = (int)(sw >> 16);
    s.act = tv && 8 * s.h < s.rl0;
}

// an item with an N among its 8 bases or the 4 before: base by base (dense pass over the work list)
FQ_DEV void stats_item_general(const StatsArgs* ap, u32* lds, int m, int h, int rl0, int lk, u32 q0, u32 q1, u32 qp, u32 codes, u32 prev8,
                               int lane) {
    const StatsArgs& a = *ap;
    u64* cyc = (u64*)(lds + a.l_cyc);
    const u32 nb0 = (q0 >> 7) & 0x01010101u, nb1 = (q1 >> 7) & 0x01010101u, nbp = (qp >> 7) & 0x01010101u;
    // bit i = base j0 - 4 + i is N (before the read start: "invalid" as in the reference, which needs 5 bases)
    u32 n12 = ((nbp | (nbp >> 7) | (nbp >> 14) | (nbp >> 21)) & 0xFu) | (((nb0 | (nb0 >> 7) | (nb0 >> 14) | (nb0 >> 21)) & 0xFu) << 4) |
              (((nb1 | (nb1 >> 7) | (nb1 >> 14) | (nb1 >> 21)) & 0xFu) << 8);
    if (h == 0) n12 |= 0xFu;
    const u32 c24 = prev8 | (codes << 8);
    const int j0 = 8 * h;
    u32* qh = lds + a.l_qh + (lane & (ST_QH_COPIES - 1));
    for (int k = 0; k < 8; k++) {
        const int j = j0 + k;
        if (j >= rl0) break;
        const u32 q = ((k < 4 ? q0 : q1) >> (8 * (k & 3))) & 0x7Fu;
        const bool isn = ((n12 >> (4 + k)) & 1u) != 0;
        const int cls = isn ? (int)CLS_N : (int)((codes >> (2 * k)) & 3u);
        const int slot = 2 * m + (j < lk ? 1 : 0);
        lds_add_u64(&cyc[((slot * 8 + k) * N_CLS + cls) * a.Hs + h], stats_inc_of(q));
        lds_add_u32(&qh[(slot * 128 + (int)q) * ST_QH_COPIES], 1u);
        if (((n12 >> k) & 0x1Fu) == 0u) lds_add_u32(&lds[a.l_kmer + slot * KMER_BINS + (int)((c24 >> (2 * k)) & 0x3FFu)], 1u);
    }
}

FQ_DEV void stats_body(const StatsArgs& a, u32* lds) {
    const int tid = thread_id(), nt = block_threads(), lane = tid & 63;
    const int H = a.H;
    const u32 H8 = (u32)a.Hs * 8u;            // bytes between the classes of one (slot, k)
    const u32 K8 = (u32)N_CLS * H8;           // bytes between the k of one slot
    const u32 S8 = 8u * K8;                   // bytes between slots
    // ---- clear the accumulators, build the character table ----
    for (int i = tid; i < a.l_total; i += nt) lds[i] = 0;
    block_sync();
    for (int e = tid; e < 256; e += nt) {
        const u32 q = (u32)e & 0x7Fu, kept = (u32)e >> 7;
        // characters below '!' never occur in a read (the packers refuse them): 0 stands for "no base here" and adds nothing
        const u64 inc = q < 33u ? 0ull : stats_inc_of(q);
        u32* t = lds + a.l_lut + 4 * e;
        t[0] = (u32)inc;
        t[1] = (u32)(inc >> 32);
        t[2] = kept ? S8 : 0u;
        t[3] = kept ? (u32)(KMER_BINS * 4) : 0u;   // k-mer slot offset; "this is a base" = bit 0 of the increment (its count field)
    }
    for (int i = tid; i < 9; i += nt) {
        lds[a.l_mt + 2 * i] = lowmask32(8 * imin(i, 4));
        lds[a.l_mt + 2 * i + 1] = lowmask32(8 * imax(0, i - 4));
    }
    block_sync();
    const int u0 = block_id() * a.units_per_block;
    const int nu = imax(0, imin(a.units_per_block, a.n - u0));
    const int per_mate = nu * H;
    u8* ldsw = (u8*)lds;
    const u32x4* lut = (const u32x4*)__builtin_assume_aligned(lds + a.l_lut, 16);
    u32* wl = lds + a.l_wl;
    const u32x2* mt = (const u32x2*)__builtin_assume_aligned(lds + a.l_mt, 8);
    for (int m = 0; m < (a.paired ? 2 : 1); m++) {   // uniform: a mate's arrays and accumulator bases sit in scalar registers
        const u32* qual = a.qual[m] + (size_t)u0 * a.qw_g;
        const u32* seq = a.seq[m] + (size_t)u0 * a.sw_g;
        const u32* swin = a.swin[m] + u0;
        const u32 slot_d = 2u * (u32)m;       // dropped bases -> the PRE slot, kept ones -> the POST slot (+1)
        const u32 cyc_m = (u32)a.l_cyc * 4u + slot_d * S8;
        const u32 kmer_m = (u32)a.l_kmer * 4u + slot_d * (KMER_BINS * 4);
        const u32 qh_m = (u32)a.l_qh * 4u + slot_d * (512u * ST_QH_COPIES) + (u32)(lane & (ST_QH_COPIES - 1)) * 4u;
        // The wavefront's mode = the character (with its kept bit) of the first item's first base, fixed at its first
        // appearance: bases that hit it are counted per lane and added once at the end - most characters of a run are
        // one value, and as LDS atomics they would all land on one address and serialise.
        u32 mode_e = 0xFFFFFFFFu;
        u32 agg_cnt = 0;
        for (int base = tid - lane; base < per_mate; base += nt) {   // wave-uniform trip count (ballots inside)
            const int it = base + lane;
            StatsItem s;
            stats_fetch(a, qual, seq, swin, it, it < per_mate, s);
            const u32 nany = (s.q0 | s.q1 | s.qp) & 0x80808080u;   // an N among the 8 bases or the 4 before
            const bool plain = s.act && nany == 0u;
            if (s.act && !plain) {                                  // rare: queued for the base-by-base pass
                const u32 slot = lds_add_ret_u32(wl, 1u);
                if (slot < (u32)a.wl_cap) wl[1 + slot] = (u32)it | ((u32)m << 31);
                else stats_item_general(&a, lds, m, s.h, s.rl0, s.lk, s.q0, s.q1, s.qp, s.codes, s.prev8, lane);   // list full: here and now
            }
            // bytes of the item that hold a base, and which of those are kept: bit 7 of a quality byte (free: no N
            // here) becomes "kept", a byte past the read's end becomes character 0 = "no base": it adds nothing
            const int j0 = 8 * s.h;
            const int nv = plain ? s.rl0 - j0 : 0, nk = s.lk - j0;   // not plain: eight "no base" characters
            const u32x2 vm = mt[imax(0, imin(nv, 8))], km = mt[imax(0, imin(nk, 8))];   // byte masks from a 9-row table
            const u32 v0 = vm.x, v1 = vm.y, k0 = km.x & 0x80808080u, k1 = km.y & 0x80808080u;
            const u32 e0 = (s.q0 | k0) & v0, e1 = (s.q1 | k1) & v1;
            if (mode_e == 0xFFFFFFFFu) {                           // wave-uniform
                const u64 cand = ballot(plain);
                if (cand) mode_e = shfl(e0 & 0xFFu, ffs64(cand) - 1);
            }
            const u32 c24 = s.prev8 | (s.codes << 8);              // bases j0-4 .. j0+7, 2 bits each
            const u32 cyc0 = cyc_m + (u32)s.h * 8u;
            const u32 hpos = s.h > 0 ? 1u : 0u;                    // 5-mers need positions >= 4 (stats.cpp:224-266)
            // the eight table rows first (independent reads, one round trip), then the adds: an add in between would pin
            // every later read behind it (the compiler cannot tell the table from the counters)
            // (four at a time: the workgroup's 64 VGPRs per lane do not hold eight rows)
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4) {
                u32x4 t[4];
#pragma unroll
                for (int i = 0; i < 4; i++) t[i] = lut[bfe(kb ? e1 : e0, 8 * i, 8)];   // character | kept << 7
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int k = kb + i;
                    const u32 e = bfe(kb ? e1 : e0, 8 * i, 8);
                    lds_add_u64((u64*)(ldsw + mul24(bfe(s.codes, 2 * k, 2), H8) + (cyc0 + t[i].z + (u32)k * K8)), (u64)t[i].x | ((u64)t[i].y << 32));
                    const u32 one = k < 4 ? (t[i].x & hpos) : (t[i].x & 1u);
                    lds_add_u32((u32*)(ldsw + ((kmer_m + t[i].w) + (bfe(c24, 2 * k, 10) << 2))), one);
                    const bool is_mode = e == mode_e;
                    agg_cnt += is_mode ? 1u : 0u;
                    // character 0 ("no base") lands in bin 0 of the dropped slot, which the flush leaves out
                    if (!is_mode) lds_add_u32((u32*)(ldsw + (qh_m + e * (4u * ST_QH_COPIES))), 1u);
                }
            }
        }
        if (mode_e != 0xFFFFFFFFu) {
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) agg_cnt += shfl_xor(agg_cnt, sh);
            if (lane == 0 && agg_cnt) lds_add_u32(&lds[a.l_qh + (int)(slot_d * 128u + mode_e) * ST_QH_COPIES], agg_cnt);
        }
    }
    block_sync();
    // ---- the queued items, every lane busy ----
    const int nw = imin((int)wl[0], a.wl_cap);
    for (int i = tid; i < nw; i += nt) {
        const u32 w = wl[1 + i];
        const int m = (int)(w >> 31);
        StatsItem s;
        stats_fetch(a, a.qual[m] + (size_t)u0 * a.qw_g, a.seq[m] + (size_t)u0 * a.sw_g, a.swin[m] + u0, (int)(w & 0x7FFFFFFFu), true, s);
        stats_item_general(&a, lds, m, s.h, s.rl0, s.lk, s.q0, s.q1, s.qp, s.codes, s.prev8, lane);
    }
    block_sync();
    // ---- flush to this workgroup's slab in the canonical order the slab fold reads ([slot][cycle][class]) ----
    u32* slab = a.slabs + (size_t)block_id() * a.slab_dwords;
    const int n_cyc = 4 * a.Cp * N_CLS;
    for (int i = tid; i < n_cyc; i += nt) {
        const int slot = i / (a.Cp * N_CLS);
        const int rem = i - slot * a.Cp * N_CLS;
        const int pos = rem / N_CLS, cls = rem - pos * N_CLS;
        const int h = pos >> 3, k = pos & 7;
        u32 lo = 0, hi = 0;
        if (h < H) {
            const int w = a.l_cyc + 2 * (((slot * 8 + k) * N_CLS + cls) * a.Hs + h);
            lo = lds[w];
            hi = lds[w + 1];
        }
        slab[2 * i] = lo;
        slab[2 * i + 1] = hi;
    }
    for (int i = tid; i < 4 * KMER_BINS; i += nt) slab[2 * n_cyc + i] = lds[a.l_kmer + i];
    for (int i = tid; i < 4 * 128; i += nt) {
        u32 v = 0;
        if (i & 127)   // bin 0 of a slot collects the "no base" characters of the fast path
            for (int c = 0; c < ST_QH_COPIES; c++) v += lds[a.l_qh + i * ST_QH_COPIES + c];
        slab[2 * n_cyc + 4 * KMER_BINS + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same pass with u32 cells (stats_body4).  What round 3/4's form (stats_body, kept as FASTP_GPU_STATS_V=3)
// spends on the LDS pipe per base is one table read, one ds_add_u64 into the packed [cnt|q20|q30|qsum] cell (9.2 cycles
// per wave instruction: twice a 32-bit one, profiles/r02c_issue_rate_microbench.txt), one ds_add_u32 into the 5-mer table
// whose 64 random bins meet in banks (39 % of the pipe's busy cycles, profiles/r04_sq_counters.txt) and the histogram add.
// Here
//   * a per-cycle cell is ONE dword [count : 12 | sum of (quality - 33) : 20] and the Q20 / Q30 counts come from WHICH
//     cell the base lands in: the table has a row per (class, level) with level = (q >= Q20) + (q >= Q30), read off the
//     character's table row like the increment.  cnt = c0 + c1 + c2, q20 = c1 + c2, q30 = c2 when the table is flushed into
//     the slab's packed u64 form (the slab fold and everything behind it are unchanged).  A workgroup therefore takes at
//     most ST4_MAX_READS = 4095 units (93 * 4095 < 2^20),
//   * the tables hold ONE mate at a time (its dropped and kept slots): half the LDS, which pays for
//   * KC copies of the 5-mer table, lane l adds to copy l % KC, the copies of a bin in consecutive banks: the KC lanes groups
//     of a wave instruction meet in 64 / KC banks each instead of all 64 lanes in 64 (KC = 4: 16 balls into 16 bins),
//   * the wavefront's mode character is counted with three SWAR instructions + v_bcnt per dword instead of a compare,
//     a select and an add per base.
// ---------------------------------------------------------------------------------------------------------------------
enum { ST4_ROWS = 3 * N_CLS, ST4_CNT_BITS = 12, ST4_MAX_READS = (1 << ST4_CNT_BITS) - 1 };

FQ_DEV u32 stats4_inc_of(u32 q) { return 1u | ((q - 33u) << ST4_CNT_BITS); }
FQ_DEV u32 stats4_level_of(u32 q) { return (q >= 53u ? 1u : 0u) + (q >= 63u ? 1u : 0u); }   // stats.cpp:209-222

// an item with an N among its 8 bases or the 4 before: base by base (dense pass over the work list), one mate's tables
template <int KC>
FQ_DEV void stats4_item_general(const StatsArgs* ap, u32* lds, int F, int h, int rl0, int lk, u32 q0, u32 q1, u32 qp, u32 codes, u32 prev8, int lane) {
    const StatsArgs& a = *ap;
    u32* cyc = lds + a.l_cyc;
    const u32 nb0 = (q0 >> 7) & 0x01010101u, nb1 = (q1 >> 7) & 0x01010101u, nbp = (qp >> 7) & 0x01010101u;
    u32 n12 = ((nbp | (nbp >> 7) | (nbp >> 14) | (nbp >> 21)) & 0xFu) | (((nb0 | (nb0 >> 7) | (nb0 >> 14) | (nb0 >> 21)) & 0xFu) << 4) |
              (((nb1 | (nb1 >> 7) | (nb1 >> 14) | (nb1 >> 21)) & 0xFu) << 8);
    if (h == 0) n12 |= 0xFu;
    const u32 c24 = prev8 | (codes << 8);
    const int j0 = 8 * h;
    u32* qh = lds + a.l_qh + (lane & (ST_QH_COPIES - 1));
    for (int k = 0; k < 8; k++) {
        const int j = j0 + k;
        if (j >= rl0) break;
        const u32 q = ((k < 4 ? q0 : q1) >> (8 * (k & 3))) & 0x7Fu;
        const bool isn = ((n12 >> (4 + k)) & 1u) != 0;
        const int cls = isn ? (int)CLS_N : (int)((codes >> (2 * k)) & 3u);
        const int slot = (j >= F && j < lk) ? 1 : 0;   // (5-mers of the first four kept bases: stats4_front_kmers moves them)
        lds_add_u32(&cyc[((slot * 8 + k) * ST4_ROWS + cls * 3 + (int)stats4_level_of(q)) * a.Hs + h], stats4_inc_of(q));
        lds_add_u32(&qh[(slot * 128 + (int)q) * ST_QH_COPIES], 1u);
        if (((n12 >> k) & 0x1Fu) == 0u)
            lds_add_u32(&lds[a.l_kmer + (slot * KMER_BINS + (int)((c24 >> (2 * k)) & 0x3FFu)) * KC + (lane & (KC - 1))], 1u);
    }
}

// DevParams::front_lane: the 5-mers that end on the first four kept bases [F, F + 4) of a read that is written out were counted
// into the kept slot like every 5-mer of a kept base (fast path and work list alike).  They are 5-mers of the ORIGINAL read
// only - the read the POST Stats object is given starts at F, and a 5-mer needs four bases in front of it (stats.cpp:224-227) -
// so each of them moves to the dropped slot (PRE = kept + dropped is unchanged).  One lane per read, at most four 5-mers.
template <int KC>
FQ_DEV void stats4_front_kmers(const StatsArgs& a, u32* lds, const u32* qual, const u32* seq, const u32* swin, int nu, int F, int tid, int nt) {
    const int copy = tid & (KC - 1);
    // the eight bases [b0, b0 + 8) hold every such 5-mer; the front is the same for every read, so where they sit in a row is
    // uniform: three quality dwords and two dwords of packed bases per read, asked for together (the first form read them byte by
    // byte, forty dependent loads a read, each a cache line of its own per lane: the -f 5 -F 5 line's Stats kernel took 2.35 ms
    // instead of 1.07, profiles/r05_front_line_kernels.txt)
    const int b0 = imax(F - 4, 0);
    const int qd = b0 >> 2, qs = 8 * (b0 & 3);
    const int sd = b0 >> 4, ss = 2 * (b0 & 15);
    for (int u = tid; u < nu; u += nt) {
        const u32 sw = swin[u];
        const int lk = (int)(sw >> 16);
        if (lk <= F) continue;                                  // not written out (or nothing kept)
        const u32* qrow = qual + (size_t)u * a.qw_g;
        const u32* srow = seq + (size_t)u * a.sw_g;
        const u32 w0 = qrow[qd], w1 = qd + 1 < a.qw_g ? qrow[qd + 1] : 0u, w2 = qd + 2 < a.qw_g ? qrow[qd + 2] : 0u;
        const u32 s0 = srow[sd], s1 = sd + 1 < a.sw_g ? srow[sd + 1] : 0u;
        const u64 lo = (u64)w0 | ((u64)w1 << 32);
        const u64 q8 = qs ? ((lo >> qs) | ((u64)w2 << (64 - qs))) : lo;   // the quality bytes of bases b0 .. b0 + 7 (bit 7: an N)
        const u32 c16 = ss ? ((s0 >> ss) | (s1 << (32 - ss))) : s0;       // their codes, the earliest base in the low bits (reduce_body)
        u32 nm = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) nm |= (u32)((q8 >> (8 * i + 7)) & 1ull) << i;
        for (int j = imax(F, 4); j < imin(F + 4, lk); j++) {    // the 5-mer of bases j - 4 .. j
            const int r = j - 4 - b0;
            if ((nm >> r) & 0x1Fu) continue;                    // an N: no 5-mer here in either Stats
            const u32 idx = (c16 >> (2 * r)) & 0x3FFu;
            lds_add_u32(&lds[a.l_kmer + (1 * KMER_BINS + (int)idx) * KC + copy], 0xFFFFFFFFu);   // kept - 1
            lds_add_u32(&lds[a.l_kmer + (0 * KMER_BINS + (int)idx) * KC + copy], 1u);            // dropped + 1
        }
    }
}

// DevParams::merge_lane, the third pass: what read 2 gives to the POST Stats object of read 1 (peprocessor.cpp:529, :554).
// swin[1] >> 16 of a unit = kept bases k2 | flags: bit 15 = the unit merged and r2[0, k2) is the merged read's second part, i.e.
// base j sits at the merged read's position m1 + k2 - 1 - j (m1 = swin[0] >> 16, the first part) as its complement; no bit 15
// (--include_unmerged): read 2 as it is.  Bit 14: the lane kernel counted the part itself (lane_merge_tail_slow).
// lane = (read 2, 8 bases); the table is ONE slot with twice the columns (a merged read is up to two reads long): it takes the place
// of the two slots of a mate's pass, [8][ST4_ROWS][2 Hs]; 5-mers and histogram use slot 0's regions.  5-mers: the ones whose five
// bases lie inside the part (the lane kernel adds the four that straddle the junction, lane_merge_junction_kmers).
template <int KC>
FQ_DEV void stats4_tail_pass(const StatsArgs& a, u32* lds, int u0, int nu, int tid, int nt, int lane) {
    const int H = a.H, Hs2 = 2 * a.Hs;
    const u32* qual = a.qual[1] + (size_t)u0 * a.qw_g;
    const u32* seq = a.seq[1] + (size_t)u0 * a.sw_g;
    const u32* swin1 = a.swin[0] + u0;
    const u32* swin2 = a.swin[1] + u0;
    u32* cyc = lds + a.l_cyc;
    u32* kmer = lds + a.l_kmer + (lane & (KC - 1));
    u32* qh = lds + a.l_qh + (lane & (ST_QH_COPIES - 1));
    const int items = nu * H;
    for (int it = tid; it < items; it += nt) {
        const u32 ur = fastdiv((u32)it, a.magic_H);
        const int h = it - (int)mul24(ur, (u32)H);
        const u32 hi = swin2[ur] >> 16;
        if (!hi || (hi & 0x4000u)) continue;
        const int lk = (int)(hi & 0x3FFFu), j0 = 8 * h;
        if (j0 >= lk) continue;
        const bool rcf = (hi & 0x8000u) != 0;
        const int base_c = rcf ? (int)(swin1[ur] >> 16) + lk - 1 : 0;   // merged position of r2[j] = base_c - j
        const u32* qrow = qual + (size_t)ur * a.qw_g;
        const u8* sb = (const u8*)(seq + (size_t)ur * a.sw_g) + 2 * h;
        const u32 q0 = qrow[2 * h], q1 = qrow[2 * h + 1];
        const u32 qp = h > 0 ? qrow[2 * h - 1] : 0u, qn = 2 * h + 2 < a.qw_g ? qrow[2 * h + 2] : 0u;
        // bases j0 - 4 .. j0 + 11, two bits each; bit i of n16 = base j0 - 4 + i is an N (or lies in front of the read)
        const u32 c32 = (h > 0 ? (u32)sb[-1] : 0u) | ((u32)sb[0] << 8) | ((u32)sb[1] << 16) | (2 * h + 2 < 4 * a.sw_g ? (u32)sb[2] << 24 : 0u);
        auto nib = [](u32 q) { const u32 b = (q >> 7) & 0x01010101u; return (b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xFu; };
        u32 n16 = nib(qp) | (nib(q0) << 4) | (nib(q1) << 8) | (nib(qn) << 12);
        if (h == 0) n16 |= 0xFu;
        for (int k = 0; k < 8; k++) {
            const int j = j0 + k;
            if (j >= lk) break;
            const u32 q = ((k < 4 ? q0 : q1) >> (8 * (k & 3))) & 0x7Fu;
            const bool isn = ((n16 >> (4 + k)) & 1u) != 0;
            const u32 code = (c32 >> (2 * (k + 4))) & 3u;
            const int c = rcf ? base_c - j : j;
            const int cls = isn ? (int)CLS_N : (int)(rcf ? code ^ 1u : code);   // the complement: A0 <-> T1, C2 <-> G3
            lds_add_u32(&cyc[(((c & 7) * ST4_ROWS) + cls * 3 + (int)stats4_level_of(q)) * Hs2 + (c >> 3)], stats4_inc_of(q));
            lds_add_u32(&qh[(int)q * ST_QH_COPIES], 1u);
            if (rcf) {   // the 5-mer that ends at merged position c: r2[j + 4] .. r2[j], complemented - inside the part when j + 4 < lk
                if (j + 4 < lk && ((n16 >> (4 + k)) & 0x1Fu) == 0u) {
                    const u32 x = (c32 >> (2 * (k + 4))) & 0x3FFu;   // r2[j] in the low bits
                    const u32 rv = ((x & 3u) << 8) | (((x >> 2) & 3u) << 6) | (x & 0x30u) | (((x >> 6) & 3u) >> 0 << 2) | ((x >> 8) & 3u);
                    lds_add_u32(&kmer[(int)(rv ^ 0x155u) * KC], 1u);   // earliest base of the merged read in the low bits (reduce_body)
                }
            } else if (((n16 >> k) & 0x1Fu) == 0u) {
                lds_add_u32(&kmer[(int)((c32 >> (2 * k)) & 0x3FFu) * KC], 1u);
            }
        }
    }
}

// ABL: the profiling instantiation (FASTP_GPU_DEBUG_SKIP 64 / 128 / 256 leave out the per-cycle / 5-mer / histogram adds:
// what is left is the measured floor of the pass - results are meaningless then); the product instantiation has no such branch
template <int KC, bool ABL>
FQ_DEV void stats_body4(const StatsArgs& a, u32* lds) {
    const int tid = thread_id(), nt = block_threads(), lane = tid & 63;
    const int H = a.H;
    const u32 H4 = (u32)a.Hs * 4u;            // bytes between the rows of one (slot, k)
    const u32 C4 = 3u * H4;                   // bytes between the classes
    const u32 K4 = (u32)ST4_ROWS * H4;        // bytes between the k of one slot
    const u32 S4 = 8u * K4;                   // bytes between the two slots
    const int u0 = block_id() * a.units_per_block;
    const int nu = imax(0, imin(a.units_per_block, a.n - u0));
    const int per_mate = nu * H;
    u8* ldsw = (u8*)lds;
    const u32x4* lut = (const u32x4*)__builtin_assume_aligned(lds + a.l_lut, 16);
    u32* wl = lds + a.l_wl;
    const u32x2* mt = (const u32x2*)__builtin_assume_aligned(lds + a.l_mt, 8);
    u32* slab = a.slabs + (size_t)block_id() * a.slab_dwords;
    const int n_cyc = 4 * a.Cp * N_CLS;       // u64 items of the slab's per-cycle part
    const int nm = a.paired ? 2 : 1;
    for (int m = 0; m < 2; m++) {             // uniform
        // ---- clear this mate's accumulators; the character table and the byte masks once ----
        for (int i = tid; i < a.l_lut; i += nt) lds[i] = 0;          // [cyc | kmer | qh] sit in front of the table
        if (tid == 0) wl[0] = 0;
        if (m == 0) {
            for (int e = tid; e < 256; e += nt) {
                const u32 q = (u32)e & 0x7Fu, kept = (u32)e >> 7;
                u32* t = lds + a.l_lut + 4 * e;
                // characters below '!' never occur in a read (the packers refuse them): 0 stands for "no base here" and adds nothing
                t[0] = q < 33u ? 0u : stats4_inc_of(q);
                t[1] = (kept ? S4 : 0u) + (q < 33u ? 0u : stats4_level_of(q) * H4);
                t[2] = kept ? (u32)(KMER_BINS * KC * 4) : 0u;
                t[3] = q < 33u ? 0u : 1u;                        // what a 5-mer add takes
            }
            for (int i = tid; i < 9; i += nt) {
                lds[a.l_mt + 2 * i] = lowmask32(8 * imin(i, 4));
                lds[a.l_mt + 2 * i + 1] = lowmask32(8 * imax(0, i - 4));
            }
        }
        block_sync();
        if (m < nm) {
            const u32* qual = a.qual[m] + (size_t)u0 * a.qw_g;
            const u32* seq = a.seq[m] + (size_t)u0 * a.sw_g;
            const u32* swin = a.swin[m] + u0;
            const u32 cyc_m = (u32)a.l_cyc * 4u;
            const u32 kmer_m = (u32)a.l_kmer * 4u + (u32)(lane & (KC - 1)) * 4u;
            const u32 qh_m = (u32)a.l_qh * 4u + (u32)(lane & (ST_QH_COPIES - 1)) * 4u;
            const int F = a.front[m];             // (uniform; 0 unless DevParams::front_lane)
            const bool no_kept = a.merge != 0 && m == 1;   // (uniform) merge mode: read 2's swin word is for the third pass
            // The wavefront's mode = the character (with its kept bit) of the first item's first base, fixed at its first
            // appearance: bases that hit it are counted per lane and added once at the end.
            u32 mode_e = 0xFFFFFFFFu, mode4 = 0, mode_tries = 0;
            u32 n_other = 0, n_items = 0;     // bytes that are not the mode / items looked at since the mode was fixed
            for (int base = tid - lane; base < per_mate; base += nt) {   // wave-uniform trip count (ballots inside)
                const int it = base + lane;
                StatsItem s;
                stats_fetch(a, qual, seq, swin, it, it < per_mate, s);
                if (no_kept) s.lk = 0;
                const u32 nany = (s.q0 | s.q1 | s.qp) & 0x80808080u;   // an N among the 8 bases or the 4 before
                const bool plain = s.act && nany == 0u;
                if (s.act && !plain) {                                  // rare: queued for the base-by-base pass
                    const u32 slot = lds_add_ret_u32(wl, 1u);
                    if (slot < (u32)a.wl_cap) wl[1 + slot] = (u32)it;
                    else stats4_item_general<KC>(&a, lds, F, s.h, s.rl0, s.lk, s.q0, s.q1, s.qp, s.codes, s.prev8, lane);   // list full: here and now
                }
                const int j0 = 8 * s.h;
                const int nv = plain ? s.rl0 - j0 : 0, nk = s.lk - j0;   // not plain: eight "no base" characters
                const u32x2 vm = mt[imax(0, imin(nv, 8))];
                u32x2 km = mt[imax(0, imin(nk, 8))];
                if (F > 0) {                                           // (uniform) kept = [F, lk): not the bytes in front of F
                    const u32x2 fm = mt[imax(0, imin(F - j0, 8))];
                    km.x &= ~fm.x;
                    km.y &= ~fm.y;
                }
                const u32 e0 = (s.q0 | (km.x & 0x80808080u)) & vm.x, e1 = (s.q1 | (km.y & 0x80808080u)) & vm.y;
                if (mode_e == 0xFFFFFFFFu) {                           // wave-uniform
                    // (a KEPT character: most bases of a run are kept ones.  The first item's first base as it came used to fix a
                    // dropped character for the whole pass when the wavefront's first read was filtered out - and for every wavefront
                    // when a front trim drops each read's first bases: the kept bases' histogram adds then all met on a few addresses,
                    // the -f 5 -F 5 line's Stats kernel took 2.35 ms instead of 1.07, profiles/r05_front_line_kernels.txt.  No such
                    // lane in this trip: tried again in the next, every byte goes through its atomic meanwhile)
                    // (read 2 in merge mode has no kept base in this pass; after a few trips without one any character will do)
                    const u64 cand = ballot(plain && ((e0 & 0x80u) != 0u || no_kept || mode_tries >= 4u));
                    mode_tries++;
                    if (cand) {
                        mode_e = shfl(e0 & 0xFFu, ffs64(cand) - 1);
                        mode4 = mode_e * 0x01010101u;
                    }
                }
                if (mode_e != 0xFFFFFFFFu) {                           // wave-uniform: bytes of the item that differ from the mode
                    const u32 x0 = e0 ^ mode4, x1 = e1 ^ mode4;
                    n_other = (u32)popc32((((x0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x0) & 0x80808080u) + n_other;
                    n_other = (u32)popc32((((x1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x1) & 0x80808080u) + n_other;
                    n_items++;
                }
                const u32 c24 = s.prev8 | (s.codes << 8);              // bases j0-4 .. j0+7, 2 bits each
                const u32 cyc0 = cyc_m + (u32)s.h * 4u;
                const u32 hpos = s.h > 0 ? 1u : 0u;                    // 5-mers need positions >= 4 (stats.cpp:224-266)
#pragma unroll
                for (int kb = 0; kb < 8; kb += 4) {
                    u32x4 t[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) t[i] = lut[bfe(kb ? e1 : e0, 8 * i, 8)];   // character | kept << 7
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int k = kb + i;
                        const u32 e = bfe(kb ? e1 : e0, 8 * i, 8);
                        if (!ABL || !(a.debug_skip & 64u))
                            lds_add_u32((u32*)(ldsw + mul24(bfe(s.codes, 2 * k, 2), C4) + (cyc0 + t[i].y + (u32)k * K4)), t[i].x);
                        const u32 one = k < 4 ? (t[i].w & hpos) : t[i].w;
                        if (!ABL || !(a.debug_skip & 128u))
                            lds_add_u32((u32*)(ldsw + ((kmer_m + t[i].z) + bfe(c24, 2 * k, 10) * (u32)(4 * KC))), one);
                        // character 0 ("no base") lands in bin 0 of the dropped slot, which the flush leaves out
                        if ((!ABL || !(a.debug_skip & 256u)) && e != mode_e) lds_add_u32((u32*)(ldsw + (qh_m + e * (4u * ST_QH_COPIES))), 1u);
                    }
                }
            }
            if (mode_e != 0xFFFFFFFFu) {
                u32 agg = 8u * n_items - n_other;
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) agg += shfl_xor(agg, sh);
                if (lane == 0 && agg) lds_add_u32(&lds[a.l_qh + (int)mode_e * ST_QH_COPIES], agg);
            }
            block_sync();
            // ---- the queued items, every lane busy ----
            const int nw = imin((int)wl[0], a.wl_cap);
            for (int i = tid; i < nw; i += nt) {
                StatsItem s;
                stats_fetch(a, qual, seq, swin, (int)wl[1 + i], true, s);
                stats4_item_general<KC>(&a, lds, F, s.h, s.rl0, no_kept ? 0 : s.lk, s.q0, s.q1, s.qp, s.codes, s.prev8, lane);
            }
            if (F > 0) {   // (uniform)
                block_sync();
                stats4_front_kmers<KC>(a, lds, a.qual[m] + (size_t)u0 * a.qw_g, a.seq[m] + (size_t)u0 * a.sw_g, swin, nu, F, tid, nt);
            }
        }
        block_sync();
        // ---- flush the mate's two slots to the slab in its canonical packed form ([slot][cycle][class] u64, reduce_body) ----
        const int per_slot = a.Cp * N_CLS;
        const int nsl = (a.merge && m == 1) ? 1 : 2;   // (merge mode: slot 3 is the third pass')
        for (int i = tid; i < nsl * per_slot; i += nt) {
            const int sl = i >= per_slot ? 1 : 0;
            const int rem = i - sl * per_slot;
            const int pos = rem / N_CLS, cls = rem - pos * N_CLS;
            const int h = pos >> 3, k = pos & 7;
            u64 v = 0;
            if (m < nm && h < H) {
                const int w = a.l_cyc + ((sl * 8 + k) * ST4_ROWS + cls * 3) * a.Hs + h;
                const u32 c0 = lds[w], c1 = lds[w + a.Hs], c2 = lds[w + 2 * a.Hs];
                const u32 M = (1u << ST4_CNT_BITS) - 1u;
                const u64 cnt = (u64)((c0 & M) + (c1 & M) + (c2 & M)), q20 = (u64)((c1 & M) + (c2 & M)), q30 = (u64)(c2 & M);
                const u64 qs = (u64)((c0 >> ST4_CNT_BITS) + (c1 >> ST4_CNT_BITS) + (c2 >> ST4_CNT_BITS));
                v = cnt | (q20 << CYC_Q20_SHIFT) | (q30 << CYC_Q30_SHIFT) | (qs << CYC_QSUM_SHIFT);
            }
            const int o = 2 * ((2 * m + sl) * per_slot + rem);
            slab[o] = (u32)v;
            slab[o + 1] = (u32)(v >> 32);
        }
        for (int i = tid; i < nsl * KMER_BINS; i += nt) {
            u32 v = 0;
            if (m < nm)
                for (int c = 0; c < KC; c++) v += lds[a.l_kmer + i * KC + c];
            slab[2 * n_cyc + 2 * m * KMER_BINS + i] = v;
        }
        for (int i = tid; i < nsl * 128; i += nt) {
            u32 v = 0;
            if (m < nm && (i & 127))   // bin 0 of a slot collects the "no base" characters of the fast path
                for (int c = 0; c < ST_QH_COPIES; c++) v += lds[a.l_qh + i * ST_QH_COPIES + c];
            slab[2 * n_cyc + 4 * KMER_BINS + 2 * m * 128 + i] = v;
        }
        block_sync();
    }
    if (a.merge) {   // (uniform) the third pass -> slot 3
        for (int i = tid; i < a.l_lut; i += nt) lds[i] = 0;
        block_sync();
        stats4_tail_pass<KC>(a, lds, u0, nu, tid, nt, lane);
        block_sync();
        const int per_slot = a.Cp * N_CLS, Hs2 = 2 * a.Hs;
        for (int i = tid; i < per_slot; i += nt) {
            const int pos = i / N_CLS, cls = i - pos * N_CLS;
            const int h = pos >> 3, k = pos & 7;
            u64 v = 0;
            if (h < Hs2) {
                const int w = a.l_cyc + (k * ST4_ROWS + cls * 3) * Hs2 + h;
                const u32 c0 = lds[w], c1 = lds[w + Hs2], c2 = lds[w + 2 * Hs2];
                const u32 M = (1u << ST4_CNT_BITS) - 1u;
                const u64 cnt = (u64)((c0 & M) + (c1 & M) + (c2 & M)), q20 = (u64)((c1 & M) + (c2 & M)), q30 = (u64)(c2 & M);
                const u64 qs = (u64)((c0 >> ST4_CNT_BITS) + (c1 >> ST4_CNT_BITS) + (c2 >> ST4_CNT_BITS));
                v = cnt | (q20 << CYC_Q20_SHIFT) | (q30 << CYC_Q30_SHIFT) | (qs << CYC_QSUM_SHIFT);
            }
            const int o = 2 * (3 * per_slot + i);
            slab[o] = (u32)v;
            slab[o + 1] = (u32)(v >> 32);
        }
        for (int i = tid; i < KMER_BINS; i += nt) {
            u32 v = 0;
            for (int c = 0; c < KC; c++) v += lds[a.l_kmer + i * KC + c];
            slab[2 * n_cyc + 3 * KMER_BINS + i] = v;
        }
        for (int i = tid; i < 128; i += nt) {
            u32 v = 0;
            for (int c = 0; c < ST_QH_COPIES; c++) v += lds[a.l_qh + i * ST_QH_COPIES + c];
            slab[2 * n_cyc + 4 * KMER_BINS + 3 * 128 + i] = v;
        }
    }
}

}  // namespace fq
