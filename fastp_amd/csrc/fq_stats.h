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
    u32 magic_H;            // ceil(2^32 / H)
    int Cp;                 // cycles rounded up to a multiple of 4 (canonical slab layout, cyc_index)
    int units_per_block;    // a workgroup takes this many consecutive units (<= CYC_MAX_READS: packed counters)
    const u32* seq[2];
    const u32* qual[2];
    const u32* swin[2];     // per read: original length | kept length << 16 (0 kept = not written out), left by the scan kernel
    // LDS layout (dwords)
    int l_cyc;              // [4][8][N_CLS][H] u64
    int l_kmer;             // [4][KMER_BINS] u32
    int l_qh;               // [4][128] u32
    int l_lut;              // [128] u64: packed per-cycle increment of a quality character
    int l_total;
    // slab (dwords): [cyc canonical: 4 * Cp * N_CLS u64][kmer 4 * KMER_BINS][qh 4 * 128]
    u32* slabs;
    int slab_dwords;
    u32 debug_skip;         // profiling only: 64 no per-cycle atomics, 128 no k-mer atomics, 256 no histogram atomics
};

FQ_DEV u64 stats_inc_of(u32 q) {   // stats.cpp:209-222: q30 ('?') counts into Q30 and Q20, q20 ('5') into Q20
    return 1ull | ((u64)(q >= 53u) << CYC_Q20_SHIFT) | ((u64)(q >= 63u) << CYC_Q30_SHIFT) | ((u64)(q - 33u) << CYC_QSUM_SHIFT);
}

// one base through the general path (N, read end or kept boundary inside the item)
FQ_DEV void stats_base_general(const StatsArgs& a, u32* lds, int slot, int h, int k, u32 q, u32 cls, bool kmer_ok, u32 km) {
    u64* cyc = (u64*)(lds + a.l_cyc);
    lds_add_u64(&cyc[((slot * 8 + k) * N_CLS + (int)cls) * a.H + h], stats_inc_of(q));
    lds_add_u32(&lds[a.l_qh + slot * 128 + (int)q], 1u);
    if (kmer_ok) lds_add_u32(&lds[a.l_kmer + slot * KMER_BINS + (int)km], 1u);
}

FQ_DEV void stats_body(const StatsArgs& a, u32* lds) {
    const int tid = thread_id(), nt = block_threads(), lane = tid & 63;
    const int H = a.H;
    // ---- clear the accumulators, build the increment table ----
    for (int i = tid; i < a.l_total; i += nt) lds[i] = 0;
    block_sync();
    for (int q = tid; q < 128; q += nt) {
        const u64 inc = q < 33 ? 0ull : stats_inc_of((u32)q);
        lds[a.l_lut + 2 * q] = (u32)inc;
        lds[a.l_lut + 2 * q + 1] = (u32)(inc >> 32);
    }
    block_sync();
    const int u0 = block_id() * a.units_per_block;
    const int nu = imax(0, imin(a.units_per_block, a.n - u0));
    const int per_mate = nu * H;
    const int total = (a.paired ? 2 : 1) * per_mate;
    const u8* lds_b = (const u8*)lds;
    u8* ldsw = (u8*)lds;
    const int cyc_b = a.l_cyc * 4, kmer_b = a.l_kmer * 4, qh_b = a.l_qh * 4, lut_b = a.l_lut * 4;
    const u32 dbg = a.debug_skip;
    const u32 H8 = (u32)H * 8u;               // bytes between the classes of one (slot, k)
    const u32 K8 = (u32)N_CLS * H8;           // bytes between the k of one slot
    const u32 S8 = 8u * K8;                   // bytes between slots
    // The wavefront's mode = the histogram bin (Stats slot AND character) of the first plain item's first base, fixed
    // at its first appearance: bases that hit it are counted per lane and added once at the end - most characters of a
    // run are one value, and as LDS atomics they would all land on one address and serialise.
    u32 mode_bin = 0xFFFFFFFFu;
    u32 agg_cnt = 0;
    for (int base = tid - lane; base < total; base += nt) {   // wave-uniform trip count (ballots inside)
        const int it = base + lane;
        const bool tv = it < total;
        const int m = (tv && it >= per_mate) ? 1 : 0;
        const int r = tv ? it - m * per_mate : 0;
        const int ur = (int)fastdiv((u32)r, a.magic_H);
        const int h = r - ur * H;
        const int g = u0 + ur;
        const u32* qrow = (m ? a.qual[1] : a.qual[0]) + (size_t)g * a.qw_g;
        const u8* srow = (const u8*)((m ? a.seq[1] : a.seq[0]) + (size_t)g * a.sw_g);
        const u32 sw = tv ? (m ? a.swin[1] : a.swin[0])[g] : 0u;
        const int rl0 = (int)(sw & 0xFFFFu), lk = (int)(sw >> 16);
        const int j0 = 8 * h;
        const bool act = tv && j0 < rl0;
        // the item's 8 quality bytes and 8 bases, the 4 bases before them
        u32 q0 = 0, q1 = 0, qp = 0, codes = 0, prev8 = 0;
        if (act) {
            const u64 qq = *(const u64*)(qrow + 2 * h);
            q0 = (u32)qq;
            q1 = (u32)(qq >> 32);
            codes = (u32)*(const u16*)(srow + 2 * h);
            if (h > 0) {
                qp = qrow[2 * h - 1];
                prev8 = (u32)srow[2 * h - 1];
            }
        }
        const int slot_d = 2 * m;
        const bool full = j0 + 8 <= rl0;                       // all 8 bases exist
        const bool kept = j0 + 8 <= lk;                        // ... and are kept
        const bool drop = j0 >= lk;                            // ... or all dropped
        const u32 nany = (q0 | q1 | qp) & 0x80808080u;         // an N among the 8 bases or the 4 before
        const bool plain = (int)act & (int)full & (int)(nany == 0u) & ((int)kept | (int)drop);
        const u32 c24 = prev8 | (codes << 8);                  // bases j0-4 .. j0+7, 2 bits each
        if (mode_bin == 0xFFFFFFFFu) {                         // wave-uniform
            const u64 cand = ballot(plain);
            if (cand) {
                const int src = ffs64(cand) - 1;
                mode_bin = shfl((u32)((kept ? slot_d + 1 : slot_d) * 128) + (q0 & 0x7Fu), src);
            }
        }
        if (plain) {
            const u32 slot = (u32)(kept ? slot_d + 1 : slot_d);
            u8* cyc = ldsw + (cyc_b + (int)(slot * S8) + h * 8);
            u8* kmer = ldsw + (kmer_b + (int)(slot * (KMER_BINS * 4)));
            u8* qh = ldsw + (qh_b + (int)(slot * 512));
            const u32 bin0 = slot * 128u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 q = bfe(k < 4 ? q0 : q1, 8 * (k & 3), 7);
                if (!(dbg & 64u)) {
                    const u64 inc = *(const u64*)(lds_b + lut_b + (q << 3));
                    lds_add_u64((u64*)(cyc + (u32)k * K8 + bfe(codes, 2 * k, 2) * H8), inc);
                }
                // 5-mer ending at base j0 + k (stats.cpp:224-266): positions >= 4 only
                if (!(dbg & 128u) && (k >= 4 || h > 0)) lds_add_u32((u32*)(kmer + (bfe(c24, 2 * k, 10) << 2)), 1u);
                if (!(dbg & 256u)) {
                    const bool is_mode = bin0 + q == mode_bin;
                    agg_cnt += is_mode ? 1u : 0u;
                    if (!is_mode) lds_add_u32((u32*)(qh + (q << 2)), 1u);
                }
            }
        } else if (act) {                                       // rare: N, the read's last item, the item the kept length cuts
            const u32 nb0 = (q0 >> 7) & 0x01010101u, nb1 = (q1 >> 7) & 0x01010101u, nbp = (qp >> 7) & 0x01010101u;
            // bit i = base j0 - 4 + i is N (before the read start: "invalid" as in the reference, which needs 5 bases)
            u32 n12 = ((nbp | (nbp >> 7) | (nbp >> 14) | (nbp >> 21)) & 0xFu) | (((nb0 | (nb0 >> 7) | (nb0 >> 14) | (nb0 >> 21)) & 0xFu) << 4) |
                      (((nb1 | (nb1 >> 7) | (nb1 >> 14) | (nb1 >> 21)) & 0xFu) << 8);
            if (h == 0) n12 |= 0xFu;
            for (int k = 0; k < 8; k++) {
                const int j = j0 + k;
                if (j >= rl0) break;
                const u32 q = ((k < 4 ? q0 : q1) >> (8 * (k & 3))) & 0x7Fu;
                const bool isn = ((n12 >> (4 + k)) & 1u) != 0;
                const u32 cls = isn ? (u32)CLS_N : ((codes >> (2 * k)) & 3u);
                const int slot = slot_d + (j < lk ? 1 : 0);
                stats_base_general(a, lds, slot, h, k, q, cls, ((n12 >> k) & 0x1Fu) == 0u, (c24 >> (2 * k)) & 0x3FFu);
            }
        }
    }
    if (mode_bin != 0xFFFFFFFFu) {
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) agg_cnt += shfl_xor(agg_cnt, sh);
        if (lane == 0 && agg_cnt) lds_add_u32(&lds[a.l_qh + (int)mode_bin], agg_cnt);
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
            const int w = a.l_cyc + 2 * (((slot * 8 + k) * N_CLS + cls) * H + h);
            lo = lds[w];
            hi = lds[w + 1];
        }
        slab[2 * i] = lo;
        slab[2 * i + 1] = hi;
    }
    for (int i = tid; i < 4 * KMER_BINS; i += nt) slab[2 * n_cyc + i] = lds[a.l_kmer + i];
    for (int i = tid; i < 4 * 128; i += nt) slab[2 * n_cyc + 4 * KMER_BINS + i] = lds[a.l_qh + i];
}

}  // namespace fq
