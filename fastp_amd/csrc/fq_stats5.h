// fq_stats5.h - Stats::statRead (src/stats.cpp:191-266), form 5 of the streaming Stats kernel (round 6, gfx950).
//
// Form 4 (fq_stats.h) spends, per base, one table read and three LDS atomics (per-cycle cell, 5-mer, quality histogram) and
// 23.5 VALU wave-instructions per 64 bases - it is bound by VALU issue (profiles/r05_stats_forms_and_floor.txt).  Per-cycle
// count, Q20 / Q30 counts, quality sum AND the quality histogram are all projections of ONE joint histogram
//     J[slot][cycle][class][quality]
// so a base costs ONE atomic for all of them (plus the 5-mer add), no table read, and the address is two multiply-adds:
//   * item = 16 consecutive bases of a read (four quality dwords + one dword of packed bases + the byte of bases in front);
//   * a cell is 16 bits wide: the dword [slot][k & 7][class][quality - 33][h] holds base k of item column h in its low half and
//     base k + 8 in its high half, so the increment (1 or 1 << 16) and the k-part of the address are compile-time constants of
//     the unrolled loop.  A workgroup takes at most CYC_MAX_READS (16383) units, a cell sees at most one base per unit: no
//     half ever carries into the other;
//   * the FAST path takes an item whose 16 bases all exist, hold no N (nor do the 4 in front), have qualities below
//     '!' + ST5_QN, and are all kept or all dropped (the slot is then part of the item's base address).  Everything else - a
//     read's last, ragged item, the item a trim ends in, items with an N, exotic qualities - is queued in a per-WAVEFRONT list
//     (ballot + prefix count: no atomic, no workgroup barrier) and taken base by base by all 64 lanes whenever 64 are queued;
//   * N bases and qualities outside the table go to a small packed-u64 table [slot][cycle][class] and the histogram proper;
//   * flush: cnt = sum over q, q20 = sum over q >= 20, q30 = sum over q >= 30, qsum = sum of q x cell, histogram = sum over
//     (cycle, class) - once per workgroup and mate, into the slab's canonical packed form (reduce_body is unchanged).
// A front trim (DevParams::front_lane): kept bases are [F, lk) at their ORIGINAL cycle (the slab fold moves the POST Stats);
// the 5-mer that ends on base j belongs to the POST Stats only when j - 4 >= F (stats.cpp:224-227 on the read that starts at
// F) - items that touch [0, F + 4) take the base-by-base path, which applies exactly that rule (form 4 needs a fix-up pass).
#pragma once
#include "fq_stats.h"

#ifndef FQ_ST5_BOUNDARY
#define FQ_ST5_BOUNDARY 1   // the clean item a trim ends in takes the masked path (a per-base slot) instead of the base-by-base one (A/B: 0.927 -> 0.916 ms, profiles/r06_n_*)
#endif
#ifndef FQ_ST5_DEPTH
#define FQ_ST5_DEPTH 3   // trips whose loads a wavefront keeps in flight (A/B: tools/gpu_r6_c.sh)
#endif

namespace fq {

enum { ST5_QN = 43,            // quality rows of the joint table: '!' .. 'K' (Q0 .. Q42); anything above takes the slow path
       ST5_QADD = 127 - 32 - ST5_QN,   // q + ST5_QADD has bit 7 set exactly when q >= '!' + ST5_QN (q < 128: no N flag)
       ST5_WL = 256 };         // queued items (u16) per wavefront and list: <= 63 left over + 60 per trip of the FQ_ST5_DEPTH <= 3 between two drains

FQ_DEV int lane_rank(u64 mask) {   // set bits of `mask` below this lane
#ifdef FQ_HOSTSIM
    const int l = lane_id();
    return popc64(mask & (l ? (~0ull >> (64 - l)) : 0ull));
#else
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
#endif
}

struct Stats5Item {
    u32 q[4], qp, codes, prev8;
    int rl0, lk;
    int F;          // the start of the kept range: the mate's front, or the read's own (DevParams::front_per_read)
};

// LDS addresses (bytes) of the joint table
struct Stats5Geo {
    u32 HS4;     // bytes between two quality rows (= 4 * item columns)
    u32 C4;      // bytes between classes  = ST5_QN * HS4
    u32 K4;      // bytes between the k & 7 = 4 * C4
    u32 S4;      // bytes between the two slots = 8 * K4
};

// item column h of unit u of the workgroup's range (qual / seq / swin = the mate's arrays at the workgroup's first unit).  Every
// load is unconditional - a lane without an item passes u = 0, and what a column does not have (the dword and the byte in front
// of column 0, the second half of a row's last, half vector) is read from a valid neighbour and masked off - so the six of them
// leave together and no lane waits behind a branch.  Issue and use are apart: the main loop keeps FQ_ST5_DEPTH trips' loads in
// flight (a wavefront that waits for one trip's 1.2 KB at a time leaves the CU with < 20 KB on its way: the pass then runs at the
// memory system's latency, 1.7 TB/s - profiles/r06_b_stats5_lane_mapping_ab_and_sq_counters.txt: 54 % of the wave cycles waiting).
struct Stats5Raw {
    u64 q01, q23;
    u32 sw, qp, cd, p8, fr;
};
struct Stats5Src {   // one mate's arrays at the workgroup's first unit
    const u32* qual;
    const u32* seq;
    const u32* swin;
    const u32* frec;   // StatsArgs::fr_rec
    int F0;            // StatsArgs::front of the mate
};
template <bool NF = false>   // NF: no front trim of any kind in this run (the kept range starts at base 0): no record is read
FQ_DEV void stats5_issue(const StatsArgs& a, const Stats5Src& src, u32 u, u32 h, Stats5Raw& r) {
    const u32* qual = src.qual;
    const u32* seq = src.seq;
    const u32* swin = src.swin;
    r.fr = NF ? 0u : src.frec[mul24(u, (u32)a.fr_stride)];
    const bool has23 = 4u * h + 4u <= (u32)a.qw_g;    // (the last item of a row may be half a vector)
    const u32 qd = mul24(u, (u32)a.qw_g) + 4u * h;      // dword of the row's quality bytes (rows are 8-byte aligned)
    const u32 sd = mul24(u, (u32)a.sw_g) + h;
    r.sw = swin[u];
    r.q01 = *(const u64*)(qual + qd);
    r.q23 = *(const u64*)(qual + qd + (has23 ? 2u : 0u));
    r.qp = qual[qd - (h > 0 ? 1u : 0u)];
    r.cd = seq[sd];
    r.p8 = (u32)((const u8*)seq)[4u * sd - (h > 0 ? 1u : 0u)];
}
template <bool NF = false>
FQ_DEV void stats5_finish(const StatsArgs& a, const Stats5Src& src, u32 h, const Stats5Raw& r, Stats5Item& s) {
    s.F = NF ? 0 : (a.front_per_read ? (int)(r.fr & 0xFFFFu) : src.F0);
    const bool has23 = 4u * h + 4u <= (u32)a.qw_g;
    const u32 m23 = has23 ? 0xFFFFFFFFu : 0u, mh = h > 0 ? 0xFFFFFFFFu : 0u;
    s.q[0] = (u32)r.q01;
    s.q[1] = (u32)(r.q01 >> 32);
    s.q[2] = (u32)r.q23 & m23;
    s.q[3] = (u32)(r.q23 >> 32) & m23;
    s.qp = r.qp & mh;
    s.codes = r.cd;
    s.prev8 = r.p8 & mh;
    s.rl0 = (int)(r.sw & 0xFFFFu);
    s.lk = (int)(r.sw >> 16);
}
template <bool NF = false>
FQ_DEV void stats5_fetch(const StatsArgs& a, const Stats5Src& src, u32 u, u32 h, Stats5Item& s) {
    Stats5Raw r;
    stats5_issue<NF>(a, src, u, h, r);
    stats5_finish<NF>(a, src, h, r, s);
}

// The cells of one item whose bases [0, nv) are clean (no N among them or the four in front, qualities the table has rows for)
// and all in one slot: 6 VALU + 2 DS instructions per base - two field extracts and two multiply-adds for the cell, an extract
// and a shift-add for the 5-mer.  MASKED: nv < 16 (a read's last item), the lanes of a wavefront stop at their own nv.
// FQ_ST5_BOUNDARY, MASKED only: bases [0, nk) of the item are kept ones (ib / kb then name the DROPPED slot)
template <int KC, bool MASKED, bool ABL>
FQ_DEV void stats5_cells(const StatsArgs& a, const Stats5Geo& g, u32 ib, u32 kb, const Stats5Item& s, u32 h, int nv, int nk = 16, u32 slotK = 0u) {
    const u32 qadd = 0x01010101u * (u32)ST5_QADD;
    const u32 ev[4] = {s.q[0] + qadd, s.q[1] + qadd, s.q[2] + qadd, s.q[3] + qadd};   // byte q + ST5_QADD: row q - 33 (ib holds the difference)
    const u32 c_lo = s.prev8 | (s.codes << 8);    // bases j0 - 4 .. j0 + 11
    const u32 c_hi = s.codes >> 8;                // bases j0 + 4 .. j0 + 15
    const u32 hpos = h > 0 ? 1u : 0u;             // 5-mers need positions >= 4 (stats.cpp:224-266)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (MASKED && (k & 3) == 0 && ballot(k < nv) == 0ull) return;   // (uniform) nobody has a base left
        if (!MASKED || k < nv) {
            const u32 e = bfe(ev[k >> 2], 8 * (k & 3), 8);
            const u32 cl = bfe(s.codes, 2 * k, 2);
            const bool BD = MASKED && FQ_ST5_BOUNDARY;
            const bool kp = BD && k < nk;
            const u32 ibk = BD ? ib + (kp ? g.S4 : 0u) : ib, kbk = BD ? kb + (kp ? slotK : 0u) : kb;
            const u32 t = mad24_su(cl, g.C4, ibk);
            const u32 ca = mad24_su(e, g.HS4, t);
            if (!ABL || !(a.debug_skip & 64u)) lds_add_u32_at(ca + (u32)(k & 7) * g.K4, k < 8 ? 1u : 0x10000u);
            const u32 x = k < 8 ? bfe(c_lo, 2 * k, 10) : bfe(c_hi, 2 * k - 16, 10);
            if (!ABL || !(a.debug_skip & 128u)) lds_add_u32_at(lshl_add<KC == 2 ? 3 : 2>(x, kbk), k < 4 ? hpos : 1u);
        }
    }
}

// one item, base by base: any state (an N, a quality the table has no row for, a kept range that ends inside the item)
template <int KC>
FQ_DEV void stats5_item_general(const StatsArgs& a, u32* lds, const Stats5Geo& g, const Stats5Item& s, int h, int hq, int lane) {   // hq: h inside its block
    const int F = s.F;
    u8* ldsw = (u8*)lds;
    const u32 nbp = (s.qp >> 7) & 0x01010101u;
    u32 n20 = (nbp | (nbp >> 7) | (nbp >> 14) | (nbp >> 21)) & 0xFu;   // bit i = base j0 - 4 + i is an N
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const u32 nb = (s.q[d] >> 7) & 0x01010101u;
        n20 |= ((nb | (nb >> 7) | (nb >> 14) | (nb >> 21)) & 0xFu) << (4 + 4 * d);
    }
    if (h == 0) n20 |= 0xFu;                                    // in front of the read: no 5-mer (stats.cpp:224-227)
    const u64 c40 = (u64)s.prev8 | ((u64)s.codes << 8);         // bases j0 - 4 .. j0 + 15, two bits each
    const int j0 = 16 * h;
    const int Fk = F > 0 ? F + 4 : 0;
    u64* ovf = (u64*)(lds + a.l_ovf);
    u32* qh = lds + a.l_qh;
    u32* kmer = lds + a.l_kmer + (lane & (KC - 1));
    for (int k = 0; k < 16; k++) {
        const int j = j0 + k;
        if (j >= s.rl0) break;
        const u32 qb = (s.q[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        const u32 q = qb & 0x7Fu;
        const bool isn = (qb & 0x80u) != 0;
        const u32 code = (s.codes >> (2 * k)) & 3u;
        const u32 slot = (j >= F && j < s.lk) ? 1u : 0u;
        const u32 e = q - 33u;
        if (!isn && e < (u32)ST5_QN) {
            lds_add_u32((u32*)(ldsw + ((u32)a.l_cyc * 4u + slot * g.S4 + (u32)(k & 7) * g.K4 + code * g.C4 + e * g.HS4 + (u32)hq * 4u)), k < 8 ? 1u : 0x10000u);
        } else {
            lds_add_u64(&ovf[(slot * (u32)a.Cp + (u32)j) * N_CLS + (isn ? (u32)CLS_N : code)], stats_inc_of(q));
            lds_add_u32(&qh[slot * 128u + q], 1u);
        }
        if (((n20 >> k) & 0x1Fu) == 0u) {
            const u32 ks = (j >= Fk && j < s.lk) ? 1u : 0u;
            lds_add_u32(&kmer[(ks * KMER_BINS + (u32)((c40 >> (2 * k)) & 0x3FFull)) * KC], 1u);
        }
    }
}

// HS: the item columns of the table as a compile-time constant (the k-part of a cell's address then sits in the DS instruction's
// offset field, the class and quality strides are literals of the two multiply-adds); 0 = read from the arguments
// ONE: the table holds every column of the reads (H16 <= HS) - one block, known at compile time: the lane's column, its first
// base and the block loop are constants again (the block form cost the 150-base configuration 9 %: 1.014 against 0.927 ms per
// 4,194,304 pairs on the same box, profiles/r06_n_compile_time_ab.txt).  NF: no front trim of any kind (uniform) - the kept
// range starts at base 0 and no result record is read (a seventh load per item otherwise)
// TC (with ONE, H16 == HS): the reads' LAST column - a ragged item for every read of the usual length (6 bases of a 150-base read),
// which no lane could ever take on the fast path - leaves the lane mapping: lane = (unit, one of the HS - 1 FULL columns), 7 x 9 =
// 63 lanes with fast work per trip instead of 6 x 9 = 54 of 60, and the last column is swept behind the trips, a lane per unit,
// straight through the masked body (no list, no second fetch for the clean ones)
template <int KC, int HS, bool ABL, bool ONE = false, bool NF = false, bool TC = false>
FQ_DEV void stats_body5(const StatsArgs& a, u32* lds) {
    const int tid = thread_id(), nt = block_threads(), lane = tid & 63;
    const int H16 = a.H16;
    Stats5Geo g;
    g.HS4 = HS ? (u32)HS * 4u : (u32)a.Hs * 4u;
    g.C4 = (u32)ST5_QN * g.HS4;
    g.K4 = 4u * g.C4;
    g.S4 = 8u * g.K4;
    const int u0 = block_id() * a.units_per_block;
    const int nu = imax(0, imin(a.units_per_block, a.n - u0));
    // lane = (unit of the wavefront's trip, item column): both are constants of the lane for the whole kernel - no division, no
    // per-item address arithmetic beyond a multiply-add per array; 64 / H16 units per trip (60 of 64 lanes at ten columns)
    // Reads longer than the table has columns for (HB = a.Hs < H16: 250-base reads are 16 columns, the table holds 8) are taken
    // in column BLOCKS: a pass over a mate's rows per block, each reading only its own columns' bytes; the per-cycle cells of a
    // block are flushed and cleared behind it, the 5-mer table, the packed cells and the histogram stay for the whole mate.
    const int HB = HS ? HS : a.Hs, nblk = ONE ? 1 : (H16 + HB - 1) / HB;
    const int HM = TC ? HS - 1 : HB;                              // the columns of the lane mapping
    const int upw = 64 / HM;
    const u32 lu = TC ? (u32)lane / (u32)(HS - 1) : HS ? (u32)lane / (u32)HS : (u32)lane / (u32)HB;
    const u32 hl = (u32)lane - lu * (u32)HM;                      // the lane's column inside a block
    const int ustride = (nt >> 6) * upw;
    u16* wlT = (u16*)(lds + a.l_wl) + (tid >> 6) * (2 * ST5_WL);   // this wavefront's two lists: a read's clean last item ...
    u16* wlG = wlT + ST5_WL;                                      // ... and everything else the fast path does not take
    u32* slab = a.slabs + (size_t)block_id() * a.slab_dwords;
    const int n_cyc = 4 * a.Cp * N_CLS;                           // u64 items of the slab's per-cycle part
    const int nm = a.paired ? 2 : 1;
    for (int m = 0; m < 2; m++) {                                 // uniform
        for (int i = tid; i < a.l_wl; i += nt) lds[i] = 0;       // [cyc | kmer | ovf | qh] sit in front of the lists
        block_sync();
      for (int cb = 0; cb < nblk; cb++) {                         // uniform
        const bool hcol = cb * HB + (int)hl < H16;                // (the last block may hold fewer columns)
        const u32 h = hcol ? (u32)(cb * HB) + hl : 0u;            // the lane's item column
        const bool used = (int)lu < upw && hcol;
        const int j0 = 16 * (int)h;
        if (m < nm) {
            Stats5Src src;
            src.qual = a.qual[m] + (size_t)u0 * a.qw_g;
            src.seq = a.seq[m] + (size_t)u0 * a.sw_g;
            src.swin = a.swin[m] + u0;
            src.frec = a.fr_rec[m] + (size_t)u0 * a.fr_stride;
            src.F0 = a.front[m];                                  // (uniform; 0 unless DevParams::front_lane)
            const u32 lds0 = lds_addr_of(lds);                    // (DS addresses, not generic pointers: no aperture add per access)
            const u32 cyc_b = lds0 + (u32)a.l_cyc * 4u - (u32)(33 + ST5_QADD) * g.HS4;   // the row of byte value q + ST5_QADD is q - 33
            const u32 kmer_b = lds0 + (u32)a.l_kmer * 4u + (u32)(lane & (KC - 1)) * 4u;
            const u32 slotK = (u32)(KMER_BINS * KC * 4);
            int wnT = 0, wnG = 0;                                 // queued items of this wavefront (uniform)
            // The loads of FQ_ST5_DEPTH trips are in flight: the trip loop is unrolled by that many with one register set per
            // position (a rotating set would have to MOVE registers that loads are still on their way to, i.e. wait for them).
            Stats5Raw ring[FQ_ST5_DEPTH];
#pragma unroll
            for (int d = 0; d < FQ_ST5_DEPTH; d++) {
                const u32 ud = (u32)((tid >> 6) * upw + d * ustride) + lu;
                stats5_issue<NF>(a, src, (used && (int)ud < nu) ? ud : 0u, h, ring[d]);
            }
            const u32 ent_lane = (u32)lane;                       // a queued item: trip << 6 | the lane that found it (its unit and column)
            const int ub0 = (tid >> 6) * upw;
            int tr = 0;
            bool more = ub0 < nu;
            for (;;) {                                            // wave-uniform (ballots inside)
              // the steady loop: blocks of FQ_ST5_DEPTH trips while no list holds a wavefront's worth.  The lists are emptied
              // OUTSIDE it: with their loads (and the fences around them) inside, the compiler's s_waitcnt at the loop's head was
              // vmcnt(0) - every block began by waiting for the loads just issued for the NEXT ones
#pragma unroll 1
              while (more && wnT < 64 && wnG < 64) {
#pragma unroll
                for (int d = 0; d < FQ_ST5_DEPTH; d++) {
                    const int ub = ub0 + (tr + d) * ustride;
                    {   // (no branch around a trip behind the range's end - its lanes have no item: the loads of the other positions
                        // stay countable for the compiler's s_waitcnt only on a straight path)
                        const u32 u = (u32)ub + lu;
                        const bool tv = used && (int)u < nu;
                        Stats5Item s;
                        stats5_finish<NF>(a, src, h, ring[d], s);
                        const int nv = s.rl0 - j0, nk = s.lk - j0;
                        const bool act = tv && nv > 0;
                        const u32 qadd = 0x01010101u * (u32)ST5_QADD;
                        const u32 dirty = (s.q[0] | s.q[1] | s.q[2] | s.q[3] | s.qp |                          // an N among the 16 bases or the 4 before,
                                           (s.q[0] + qadd) | (s.q[1] + qadd) | (s.q[2] + qadd) | (s.q[3] + qadd)) & 0x80808080u;   // a quality without a row
                        const int F = s.F, Fk = F > 0 ? F + 4 : 0;
                        const bool kept = j0 >= Fk && nk >= imin(nv, 16);              // every base of the item (and its 5-mer) is a kept one
                        const bool drop = nk <= 0 || j0 + 16 <= F;                     // ... a dropped one
                        const bool mixed = FQ_ST5_BOUNDARY && F == 0 && !kept && !drop;   // a trim ends inside the item (no front trim)
                        const bool clean = act && dirty == 0u && (kept || drop || mixed);
                        const bool fast = clean && nv >= 16 && !mixed;
                        const u64 mT = ballot(clean && !fast), mG = ballot(act && !clean);
                        const u32 ent = ((u32)(tr + d) << 6) | ent_lane;
                        if (mT) {                                 // (uniform) positions by a prefix count over the ballot: no atomic
                            if (clean && !fast) wlT[wnT + lane_rank(mT)] = (u16)ent;
                            wnT += popc64(mT);
                        }
                        if (mG) {
                            if (act && !clean) wlG[wnG + lane_rank(mG)] = (u16)ent;
                            wnG += popc64(mG);
                        }
                        if (fast) stats5_cells<KC, false, ABL>(a, g, opaque(cyc_b + (kept ? g.S4 : 0u) + hl * 4u), opaque(kmer_b + (kept ? slotK : 0u)), s, h, 16);
                        {   // this position's registers are free again: the trip FQ_ST5_DEPTH ahead
                            const u32 un = u + (u32)(FQ_ST5_DEPTH * ustride);
                            stats5_issue<NF>(a, src, (used && (int)un < nu) ? un : 0u, h, ring[d]);
                        }
                    }
                }
                tr += FQ_ST5_DEPTH;
                more = ub0 + tr * ustride < nu;
              }
                // a full wavefront of queued items (behind the last trip: what is left)
                while (wnT >= 64 || (!more && wnT > 0)) {         // (uniform)
                    const int cnt = imin(wnT, 64);
                    wave_sync();
                    wnT -= cnt;
                    const bool on = lane < cnt;
                    const u32 w = on ? (u32)wlT[wnT + lane] : 0u;
                    const u32 wl_ = TC ? (w & 63u) / (u32)(HS - 1) : HS ? (w & 63u) / (u32)HS : fastdiv(w & 63u, a.magic_H16);
                    const u32 hq = (w & 63u) - wl_ * (u32)HM, hh = (u32)(cb * HB) + hq, uu = (u32)(ub0 + (int)(w >> 6) * ustride) + wl_;
                    Stats5Item t;
                    stats5_fetch<NF>(a, src, on ? uu : 0u, hh, t);
                    const int tj0 = 16 * (int)hh;
                    const int tnv = t.rl0 - tj0;
                    const bool tk = tj0 >= (t.F > 0 ? t.F + 4 : 0) && t.lk - tj0 >= imin(tnv, 16);   // (queued as all kept or all dropped: `kept` above)
                    if (FQ_ST5_BOUNDARY) {   // ... or (no front trim) kept up to the trim's end inside the item: bases [0, tnk) are kept
                        const int tnk = t.F == 0 ? imax(0, imin(16, t.lk - tj0)) : (tk ? 16 : 0);
                        stats5_cells<KC, true, ABL>(a, g, opaque(cyc_b + hq * 4u), opaque(kmer_b), t, hh, on ? tnv : 0, tnk, slotK);
                    } else
                    stats5_cells<KC, true, ABL>(a, g, opaque(cyc_b + (tk ? g.S4 : 0u) + hq * 4u), opaque(kmer_b + (tk ? slotK : 0u)), t, hh, on ? tnv : 0);
                    wave_sync();
                }
                while (wnG >= 64 || (!more && wnG > 0)) {         // (uniform)
                    const int cnt = imin(wnG, 64);
                    wave_sync();
                    wnG -= cnt;
                    const bool on = lane < cnt;
                    const u32 w = on ? (u32)wlG[wnG + lane] : 0u;
                    const u32 wl_ = TC ? (w & 63u) / (u32)(HS - 1) : HS ? (w & 63u) / (u32)HS : fastdiv(w & 63u, a.magic_H16);
                    const u32 hq = (w & 63u) - wl_ * (u32)HM, hh = (u32)(cb * HB) + hq, uu = (u32)(ub0 + (int)(w >> 6) * ustride) + wl_;
                    Stats5Item t;
                    stats5_fetch<NF>(a, src, on ? uu : 0u, hh, t);
                    if (on) stats5_item_general<KC>(a, lds, g, t, (int)hh, (int)hq, lane);
                    wave_sync();
                }
                if (!more) break;
            }
            if (TC) {   // the last column, a lane per unit (both lists are empty here: wlG now holds UNIT indexes of what is not clean)
                const u32 hh = (u32)(HS - 1);
                const int tj0 = 16 * (HS - 1);
                const int wstep = (nt >> 6) * 64;
                int wnD = 0;
                for (int ut = (tid >> 6) * 64; ut < nu; ut += wstep) {   // wave-uniform
                    const u32 u = (u32)(ut + lane);
                    const bool on = (int)u < nu;
                    Stats5Item t;
                    stats5_fetch<NF>(a, src, on ? u : 0u, hh, t);
                    const int tnv = t.rl0 - tj0, tnkr = t.lk - tj0;
                    const bool act = on && tnv > 0;
                    const u32 qadd = 0x01010101u * (u32)ST5_QADD;
                    const u32 dirty = (t.q[0] | t.q[1] | t.q[2] | t.q[3] | t.qp | (t.q[0] + qadd) | (t.q[1] + qadd) | (t.q[2] + qadd) | (t.q[3] + qadd)) & 0x80808080u;
                    const int F = t.F, Fk = F > 0 ? F + 4 : 0;
                    const bool kept = tj0 >= Fk && tnkr >= imin(tnv, 16);
                    const bool drop = tnkr <= 0 || tj0 + 16 <= F;
                    const bool mixed = FQ_ST5_BOUNDARY && F == 0 && !kept && !drop;
                    const bool clean = act && dirty == 0u && (kept || drop || mixed);
                    const u64 mD = ballot(act && !clean);
                    if (mD) {
                        if (act && !clean) wlG[wnD + lane_rank(mD)] = (u16)u;
                        wnD += popc64(mD);
                    }
                    if (FQ_ST5_BOUNDARY) {
                        const int tnk = F == 0 ? imax(0, imin(16, tnkr)) : (kept ? 16 : 0);
                        stats5_cells<KC, true, ABL>(a, g, opaque(cyc_b + hh * 4u), opaque(kmer_b), t, hh, clean ? tnv : 0, tnk, slotK);
                    } else {
                        stats5_cells<KC, true, ABL>(a, g, opaque(cyc_b + (kept ? g.S4 : 0u) + hh * 4u), opaque(kmer_b + (kept ? slotK : 0u)), t, hh, clean ? tnv : 0);
                    }
                    const bool last = ut + wstep >= nu;
                    while (wnD >= 64 || (last && wnD > 0)) {          // (uniform)
                        const int cnt = imin(wnD, 64);
                        wave_sync();
                        wnD -= cnt;
                        const bool on2 = lane < cnt;
                        const u32 w = on2 ? (u32)wlG[wnD + lane] : 0u;
                        Stats5Item t2;
                        stats5_fetch<NF>(a, src, w, hh, t2);
                        if (on2) stats5_item_general<KC>(a, lds, g, t2, (int)hh, (int)hh, lane);
                        wave_sync();
                    }
                }
            }
        }
        block_sync();
        // ---- flush the block's cycles of the mate's two slots to the slab in its canonical packed form ([slot][cycle][class] u64,
        // reduce_body); the last block also takes the cycles behind the last column ----
        const int per_slot = a.Cp * N_CLS;
        const u64* ovf = (const u64*)(lds + a.l_ovf);
        const int pos_lo = cb * HB * 16, pos_hi = cb + 1 < nblk ? (cb + 1) * HB * 16 : a.Cp;
        const int blk_items = imax(0, imin(pos_hi, a.Cp) - pos_lo) * N_CLS;
        for (int i = tid; i < 2 * blk_items; i += nt) {
            const int sl = i >= blk_items ? 1 : 0;
            const int rem = i - sl * blk_items + pos_lo * N_CLS;
            const int pos = rem / N_CLS, cls = rem - pos * N_CLS;
            u64 v = 0;
            if (m < nm) {
                u32 cnt = 0, q20 = 0, q30 = 0, qs = 0;
                const int hq = (pos - pos_lo) >> 4, k = pos & 15;
                if (cls < 4 && hq < HB && (pos >> 4) < H16) {
                    const u32* row = lds + a.l_cyc + ((sl * 8 + (k & 7)) * 4 + cls) * ST5_QN * HB + hq;
                    const int sh = k < 8 ? 0 : 16;
                    for (int e = 0; e < ST5_QN; e++) {
                        const u32 c = (row[e * HB] >> sh) & 0xFFFFu;
                        cnt += c;
                        qs += c * (u32)e;
                        if (e >= 20) q20 += c;                   // stats.cpp:209-222: '5' counts into Q20, '?' into Q30 and Q20
                        if (e >= 30) q30 += c;
                    }
                }
                const u64 o = ovf[(sl * a.Cp + pos) * N_CLS + cls];
                const u64 M = (1ull << CYC_CNT_BITS) - 1ull;
                v = ((u64)cnt + (o & M)) | (((u64)q20 + ((o >> CYC_Q20_SHIFT) & M)) << CYC_Q20_SHIFT) |
                    (((u64)q30 + ((o >> CYC_Q30_SHIFT) & M)) << CYC_Q30_SHIFT) | (((u64)qs + (o >> CYC_QSUM_SHIFT)) << CYC_QSUM_SHIFT);
            }
            const int o2 = 2 * ((2 * m + sl) * per_slot + rem);
            slab[o2] = (u32)v;
            slab[o2 + 1] = (u32)(v >> 32);
        }
        // the histogram of the block's table bases - the sum over (cycle, class) - joins the histogram proper
        for (int i = tid; i < 2 * ST5_QN; i += nt) {
            if (m >= nm) break;
            const int sl = i >= ST5_QN ? 1 : 0, e = i - sl * ST5_QN;
            u32 v = 0;
            for (int kc = 0; kc < 8 * 4; kc++) {
                const u32* row = lds + a.l_cyc + (((sl * 8 * 4 + kc) * ST5_QN) + e) * HB;
                for (int hq = 0; hq < HB; hq++) v += (row[hq] & 0xFFFFu) + (row[hq] >> 16);
            }
            lds[a.l_qh + sl * 128 + 33 + e] += v;                 // (one thread per bin)
        }
        block_sync();
        if (cb + 1 < nblk) {                                      // (uniform) the next block starts on an empty table
            for (int i = tid; i < a.l_kmer - a.l_cyc; i += nt) lds[a.l_cyc + i] = 0;
            block_sync();
        }
      }
        for (int i = tid; i < 2 * KMER_BINS; i += nt) {
            u32 v = 0;
            if (m < nm)
                for (int c = 0; c < KC; c++) v += lds[a.l_kmer + i * KC + c];
            slab[2 * n_cyc + 2 * m * KMER_BINS + i] = v;
        }
        for (int i = tid; i < 2 * 128; i += nt) slab[2 * n_cyc + 4 * KMER_BINS + 2 * m * 128 + i] = m < nm ? lds[a.l_qh + i] : 0u;
        block_sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DevParams::front_per_read (--cut_front on the lane plan): the Stats kernel counted a written-out read's kept bases at their
// ORIGINAL cycle and the slab fold moved the POST Stats by the mate's common front F0; a read whose own front F is larger (its
// forward quality cut found a bad window: few reads) has its kept bases F - F0 cycles too late there.  One lane per such read
// moves them: every per-cycle array of the POST Stats object (stats.cpp:206-222) from cycle j - F0 to cycle j - F.  Deltas are
// gathered in LDS as signed 32-bit sums ([34 * cycles] per mate) and added to the int64 block once per workgroup.
// ---------------------------------------------------------------------------------------------------------------------
struct FrontStatsArgs {
    int n, paired;
    int sw_g, qw_g;
    const u32* seq[2];
    const u32* qual[2];
    const u32* swin[2];       // original length | END of the kept range << 16 (0: not written out), after --dedup's decisions
    const u32* rec[2];        // result records, 3 dwords a read: the low 16 bits of the first = the read's front
    int front0[2];            // DevParams::lane_front*
    int64_t* post[2];         // the POST Stats object of mate m in the counter block
    int64_t st_cycle, cycles;
};
FQ_DEV void front_stats_body(const FrontStatsArgs& c, u32* lds) {
    const int tid = thread_id(), nt = block_threads();
    const int CC = (int)c.cycles;
    const int per_mate = 34 * CC, mates = c.paired ? 2 : 1;
    for (int i = tid; i < mates * per_mate; i += nt) lds[i] = 0;
    block_sync();
    const int reads = c.paired ? 2 * c.n : c.n;
    for (int t = block_id() * nt + tid; t < reads; t += grid_blocks() * nt) {
        const int g = c.paired ? t >> 1 : t, m = c.paired ? (t & 1) : 0;
        const u32 sw = c.swin[m][g];
        const int lk = (int)(sw >> 16), F0 = c.front0[m];
        if (lk == 0) continue;                                 // not written out: no POST Stats
        const int F = (int)(c.rec[m][(size_t)g * 3] & 0xFFFFu);
        if (F <= F0 || lk <= F) continue;
        const u32* srow = c.seq[m] + (size_t)g * c.sw_g;
        const u8* qrow = (const u8*)(c.qual[m] + (size_t)g * c.qw_g);
        u32* cyc = lds + m * per_mate;
        for (int j = F; j < lk; j++) {
            const u32 q = (u32)qrow[j] & 0x7Fu;
            const int b = (int)sym_bin(row_sym(srow, qrow, j));
            const int co = j - F0, cn = j - F;
            if (q >= 63u) { lds_add_u32(&cyc[(0 * 8 + b) * CC + co], (u32)-1); lds_add_u32(&cyc[(0 * 8 + b) * CC + cn], 1u); }   // stats.cpp:209-222
            if (q >= 53u) { lds_add_u32(&cyc[(1 * 8 + b) * CC + co], (u32)-1); lds_add_u32(&cyc[(1 * 8 + b) * CC + cn], 1u); }
            lds_add_u32(&cyc[(2 * 8 + b) * CC + co], (u32)-1);
            lds_add_u32(&cyc[(2 * 8 + b) * CC + cn], 1u);
            lds_add_u32(&cyc[(3 * 8 + b) * CC + co], 0u - (q - 33u));
            lds_add_u32(&cyc[(3 * 8 + b) * CC + cn], q - 33u);
            lds_add_u32(&cyc[32 * CC + co], (u32)-1);          // mCycleTotalBase
            lds_add_u32(&cyc[32 * CC + cn], 1u);
            lds_add_u32(&cyc[33 * CC + co], 0u - (q - 33u));   // mCycleTotalQual
            lds_add_u32(&cyc[33 * CC + cn], q - 33u);
        }
    }
    block_sync();
    for (int i = tid; i < mates * per_mate; i += nt) {
        const int v = (int)lds[i];
        if (!v) continue;
        const int m = i / per_mate, k = i - m * per_mate;
        g_atomic_add_i64(c.post[m] + c.st_cycle + k, (int64_t)v);
    }
}

}  // namespace fq
