// fq_device.h - the fused per-read kernel of the MI355X FASTQ engine (gfx950).
//
// One workgroup owns a TILE of P read pairs (or P single reads) staged in LDS as
// 2-bit bases + N mask + quality bytes, runs every per-read step of fastp's
// worker loop on it (reference: src/peprocessor.cpp:383-643,
// src/seprocessor.cpp:204-296) and accumulates the Stats / FilterResult
// counters in LDS-privatised histograms that are flushed once per launch.
//
// Phases (block_sync between them), each with the lane mapping that keeps all wavefronts busy:
//   load    : flat 16-byte vector copies HBM -> LDS (registers filled one tile ahead), N masks
//   masks   : lane = (read, 32-position mask word)   trimAndCut's windows for every start position
//   hash    : 8 lanes per read                        Duplicate::seq2intvector on the original read
//   trim    : lane = read                             UMI front trim, trimAndCut as bit scans
//   polyG   : lane = read (when enabled)
//   overlap : lane = (pair, direction, quarter)       OverlapAnalysis::analyze, prefilter + verify,
//             LDS atomic-min over scan-order keys; optional one-gap pass; again after trimming in
//             merge mode
//   decide  : lane = pair                             isize, BaseCorrector, AdapterTrimmer, polyX, max_len
//   metrics : 8 lanes per read                        countQualityMetrics / countAdjacentDiffs
//   filter  : lane = pair                             Filter::passFilter, merge bookkeeping, records
//   stats   : lane = (read, 4 cycles)                 Stats::statRead for all four Stats objects - one
//             kept/dropped pass when no option moves or edits a kept base, else pre + post passes
// After the fused kernel: slab fold (reduce_body), duplicate probe/resolve in input order, and the
// overrepresentation analysis (ovr_*).  Before it, for FASTQ text resident in HBM: parse_*.
//
// Integer / byte work only: no MFMA.  Every function cites the reference lines
// whose behaviour it reproduces (paths relative to the reference root).
#pragma once
#include "fq_intrin.h"

namespace fq {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
FQ_DEV u32 lowmask32(int nbits) { return nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u); }
FQ_DEV int imin(int a, int b) { return a < b ? a : b; }
FQ_DEV int imax(int a, int b) { return a > b ? a : b; }

// 32-bit window (16 bases) of a packed row starting at base position bp >= 0
FQ_DEV u32 window16(const u32* row, int bp) {
    const int w = bp >> 4;
    return alignbit(row[w + 1], row[w], (u32)((bp & 15) * 2));
}
// same for any bp: bases at negative positions read as zero
FQ_DEV u32 window16_signed(const u32* row, int bp) {
    if (bp >= 0) return window16(row, bp);
    return bp <= -16 ? 0u : row[0] << (u32)(-bp * 2);
}
// 2-bit groups differ -> bit 2k set
FQ_DEV u32 fold_diff(u32 x) { return (x | (x >> 1)) & 0x55555555u; }
// reverse the order of the 16 2-bit groups of a word
FQ_DEV u32 reverse_groups(u32 x) {
    const u32 r = brev32(x);
    return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}
// exact division of small non-negative n by d via a host-computed reciprocal
FQ_DEV u32 fastdiv(u32 n, u32 magic) { return __umulhi(n, magic); }

struct Rows {  // LDS views of one read
    const u32* s;  // packed bases
    const u32* n;  // N mask (bit 2k = base k is N)
    const u8* q;   // quality bytes (bit 7 = N)
};

// LDS indices are far below 2^24: row offsets use the full-rate 24-bit multiply (v_mul_lo_u32 is half rate)
FQ_DEV int rowoff(int R, int stride) { return (int)mul24((u32)R, (u32)stride); }
FQ_DEV u32* lds_seq(const LdsLayout& L, u32* lds, int R) { return lds + L.seq + rowoff(R, L.SW); }
FQ_DEV u32* lds_nmk(const LdsLayout& L, u32* lds, int R) { return lds + L.nmk + rowoff(R, L.SW); }
FQ_DEV u32* lds_qual(const LdsLayout& L, u32* lds, int R) { return lds + L.qual + rowoff(R, L.QW); }
FQ_DEV int* lds_i(u32* lds, int off) { return (int*)(lds + off); }

// barrier over the wavefronts that work on one tile: the whole workgroup, or - two tiles in flight - the half that
// owns the tile (`lds` is that half's LDS base, `tid` / `nthreads` its view of itself)
FQ_DEV void tile_sync(const KernelArgs& a, u32* lds, int nthreads) {
    if (a.L.halves == 2) half_sync(lds + a.L.bar, thread_id() >= nthreads ? 1 : 0, nthreads, a.half_naps);
    else block_sync();
}

FQ_DEV u32 code_at(const u32* srow, int j) { return (srow[j >> 4] >> ((j & 15) * 2)) & 3u; }
FQ_DEV u32 qchar_at(const u8* q, int j) { return q[j] & 0x7Fu; }
FQ_DEV bool isn_at(const u8* q, int j) { return (q[j] & 0x80u) != 0; }
// symbol 0..3 = A,T,C,G ; 4 = N
FQ_DEV u32 sym_at(const u32* srow, const u8* q, int j) { return isn_at(q, j) ? 4u : code_at(srow, j); }
// util.h:16-33 on symbols: A<->T, C<->G, N->N
FQ_DEV u32 sym_complement(u32 s) { return s == 4u ? 4u : (s ^ 1u); }

// ---------------------------------------------------------------------------
// Phase A: load one tile
// ---------------------------------------------------------------------------
// per-read state of a fresh tile
FQ_DEV void tile_init_read(const LdsLayout& L, u32* lds, int R, int len) {
    lds_i(lds, L.rlen0)[R] = len;
    lds_i(lds, L.front)[R] = 0;
    lds_i(lds, L.len)[R] = len;
    lds_i(lds, L.flags)[R] = 0;
    lds_i(lds, L.ft)[R] = 0;
    lds_i(lds, L.apos)[R] = 0;
    lds_i(lds, L.alen)[R] = 0;
    lds_i(lds, L.code)[R] = 0;
    lds[L.swin + R] = 0;
    if (R < L.P) {
        lds_i(lds, L.ov_off)[R] = (int)OV_KEY_NONE;
        lds_i(lds, L.ov_len)[R] = (int)OV_KEY_NONE;
    }
    if (R == 0) lds[L.wl] = 0;
}
// N masks (bit 2k of word w = base 16w + k is N) of every read, from bit 7 of its quality bytes, and the read's
// RS_HAS_N flag: lane = (read, 16-base word).  Every word is written, so nothing has to be cleared beforehand.
FQ_DEV void phase_nmask(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int SW = L.SW, QW = L.QW;
    const int total = L.NR * SW;
    const u32 magic = a.magic_sw;
    int* flags = lds_i(lds, L.flags);
    for (int i = tid; i < total; i += nthreads) {
        const int R = (int)fastdiv((u32)i, magic), w = i - R * SW;
        u32 m = 0;
        if (4 * w < QW) {  // quality dwords 4w .. 4w+3 (rows are 8-byte aligned: QW is even)
            const u64* q2 = (const u64*)(lds + L.qual + rowoff(R, QW) + 4 * w);
            const u64 a01 = q2[0];
            const u64 a23 = 4 * w + 2 < QW ? q2[1] : 0ull;
            if ((a01 | a23) & 0x8080808080808080ull) {
                const u32 qd[4] = {(u32)a01, (u32)(a01 >> 32), (u32)a23, (u32)(a23 >> 32)};
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const u32 nb = (qd[d] >> 7) & 0x01010101u;
                    m |= ((nb | (nb >> 6) | (nb >> 12) | (nb >> 18)) & 0x55u) << (8 * d);
                }
                lds_or_i32(&flags[R], RS_HAS_N);
            }
        }
        lds[L.nmk + i] = m;
    }
}

// Phase A, scalar form (any alignment / tile shape).  The LDS rows have the batch's own strides,
// so each mate's part of a tile is one contiguous run of dwords in both places.
FQ_DEV void phase_load(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int mates = a.p.paired ? 2 : 1;
    const int P = L.P;
    const int rows = imax(0, imin(P, a.n - tile_first));  // rows that exist
    const u16* len0 = a.len[0];
    const u16* len1 = a.len[1];
    for (int R = tid; R < L.NR; R += nthreads) {
        const int m = R >= P ? 1 : 0;
        const int gp = tile_first + (R - m * P);
        tile_init_read(L, lds, R, gp < a.n ? (int)(m ? len1 : len0)[gp] : 0);
    }
    for (int m = 0; m < mates; m++) {
        const u32* g = a.seq[m] + (size_t)tile_first * L.SW;
        for (int i = tid; i < P * L.SW; i += nthreads) lds[L.seq + m * P * L.SW + i] = i < rows * L.SW ? g[i] : 0u;
    }
    for (int m = 0; m < mates; m++) {
        const u32* g = a.qual[m] + (size_t)tile_first * L.QW;
        for (int i = tid; i < P * L.QW; i += nthreads) lds[L.qual + m * P * L.QW + i] = i < rows * L.QW ? g[i] : 0u;
    }
}

// ---------------------------------------------------------------------------
// Phase A, vector form with software prefetch: 16-byte chunks; the global loads of tile i+1 are
// issued right after tile i has been staged and stay in flight (registers) while tile i is
// processed, so the HBM latency of a tile is hidden behind the compute of the previous one.
// ---------------------------------------------------------------------------

// Stage a FULL tile (all P rows exist): 16-byte chunks HBM -> registers -> LDS, every chunk's load issued before
// the first store waits for one.  No bounds logic (a ragged last tile goes through phase_load).  The mates' row
// arrays are scalars: a per-lane choice between them as POINTERS from the argument block would be a vector load
// followed by a wait.
FQ_DEV void tile_stage(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int nq = L.NR * L.QW / 4, ns = L.NR * L.SW / 4;
    typedef u32 v4u __attribute__((vector_size(16)));   // a native 16-byte vector: stays in registers
    const v4u* q0 = (const v4u*)(a.qual[0] + (size_t)tile_first * L.QW);
    const v4u* q1 = (const v4u*)(a.qual[1] + (size_t)tile_first * L.QW);
    const v4u* s0 = (const v4u*)(a.seq[0] + (size_t)tile_first * L.SW);
    const v4u* s1 = (const v4u*)(a.seq[1] + (size_t)tile_first * L.SW);
    const int qpm = L.P * L.QW / 4, spm = L.P * L.SW / 4;   // chunks per mate
    v4u q[PF_Q], sq[PF_S];
    u32 len = 0;
#pragma unroll
    for (int i = 0; i < PF_Q; i++) {
        const int ci = imin(tid + i * nthreads, nq - 1);   // no branch around a load: lanes past the end re-read the last chunk
        q[i] = *(ci >= qpm ? q1 + (ci - qpm) : q0 + ci);
    }
#pragma unroll
    for (int i = 0; i < PF_S; i++) {
        const int ci = imin(tid + i * nthreads, ns - 1);
        sq[i] = *(ci >= spm ? s1 + (ci - spm) : s0 + ci);
    }
    {
        const u16* l0 = scalar_ptr(a.len[0]) + tile_first;
        const u16* l1 = scalar_ptr(a.len[1]) + tile_first;
        const int rr = imin(tid, L.NR - 1);
        len = *(rr >= L.P ? l1 + (rr - L.P) : l0 + rr);
    }
    if (tid < L.NR) tile_init_read(L, lds, tid, (int)len);
#pragma unroll
    for (int i = 0; i < PF_S; i++) {
        const int ci = tid + i * nthreads;
        if (ci < ns) *(v4u*)(lds + L.seq + 4 * ci) = sq[i];
    }
#pragma unroll
    for (int i = 0; i < PF_Q; i++) {
        const int ci = tid + i * nthreads;
        if (ci < nq) *(v4u*)(lds + L.qual + 4 * ci) = q[i];
    }
}

// one dword of every 128-byte line of the tile that starts at unit tile_first (rows that exist only): pulls the
// tile into L2 / the Infinity Cache while the current one is processed.  Returns the touch's register (touch_done).
FQ_DEV u32 tile_warm(const KernelArgs& a, int tile_first, int tid) {
    const LdsLayout& L = a.L;
    const int rows = imax(0, imin(L.P, a.n - tile_first));
    const int ql = (rows * L.QW * 4 + 127) >> 7, sl = (rows * L.SW * 4 + 127) >> 7;   // lines per mate
    const int per = ql + sl;
    const int lines = a.p.paired ? 2 * per : per;
    const u32* q0 = scalar_ptr(a.qual[0]) + (size_t)tile_first * L.QW;
    const u32* s0 = scalar_ptr(a.seq[0]) + (size_t)tile_first * L.SW;
    const u32* q1 = a.p.paired ? scalar_ptr(a.qual[1]) + (size_t)tile_first * L.QW : q0;
    const u32* s1 = a.p.paired ? scalar_ptr(a.seq[1]) + (size_t)tile_first * L.SW : s0;
    const int t = imin(tid, lines - 1);    // lanes past the last line touch it again: no branch around the load
    const int m = t >= per ? 1 : 0;
    const int k = t - (m ? per : 0);
    const u32* p = k < ql ? (m ? q1 : q0) + 32 * k : (m ? s1 : s0) + 32 * (k - ql);
    return touch_begin(lines > 0 ? p : q0);
}

// total quality (N flag masked off) of the windows [4c+k, 4c+k+w), k = 0..3, of one row:
// v_alignbit_b32 + v_sad_u8 per 4 bases.  ncols = dwords of the LDS row (pad columns are zero).
FQ_DEV void window_sums4(const u32* qrow, int c, int ncols, int w, u32 out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    u32 lo = qrow[c] & 0x7F7F7F7Fu;
    int rem = w;
    for (int i = c + 1; rem > 0; i++, rem -= 4) {
        const u32 hi = i < ncols ? (qrow[i] & 0x7F7F7F7Fu) : 0u;
        const u32 keep = lowmask32(8 * rem);
        out[0] = sum_bytes(lo & keep, out[0]);
        out[1] = sum_bytes(alignbit(hi, lo, 8) & keep, out[1]);
        out[2] = sum_bytes(alignbit(hi, lo, 16) & keep, out[2]);
        out[3] = sum_bytes(alignbit(hi, lo, 24) & keep, out[3]);
        lo = hi;
    }
}

// bit j (j = 0..31) <=> the window [32*word + j, 32*word + j + w) of the row has total quality < thr, for the
// positions below `limit` (bits at or past it are unspecified: no scan of trim_and_cut looks at a window that
// leaves the read).  Windows of up to 8 bases - the default is 4 - take three instructions per start position:
// v_alignbit (the window's bytes), v_sad_u8 with -thr as the addend (sum - thr), v_alignbit again to shift the
// sign into the mask; a wider window goes through window_sums4.
FQ_DEV u32 bad_window_word(const u32* qrow, int word, int QW, int limit, int w, int thr) {
    u32 m = 0;
    if (w == 4) {  // the default window: the four bytes at a position ARE the window
        const u32 nthr = (u32)(-thr);
        const int c0 = 8 * word;
        u32 q[9];
#pragma unroll
        for (int d = 0; d < 9; d++) q[d] = (c0 + d < QW && 4 * (c0 + d) < limit + 4) ? (qrow[c0 + d] & 0x7F7F7F7Fu) : 0u;
#pragma unroll
        for (int d = 7; d >= 0; d--) {
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                const u32 x = k ? alignbit(q[d + 1], q[d], 8 * k) : q[d];
                m = alignbit(m, sum_bytes(x, nthr), 31);  // m = m << 1 | (sum < thr)
            }
        }
        return m;
    }
    if (w <= 8) {
        const u32 nthr = (u32)(-thr);
        const u32 keep_lo = lowmask32(8 * imin(w, 4)), keep_hi = w > 4 ? lowmask32(8 * (w - 4)) : 0u;
        // quality dwords 8*word .. 8*word + 9 (two of look-ahead), high to low so that the mask fills from bit 31 down
        const int c0 = 8 * word;
        u32 q[10];
#pragma unroll
        for (int d = 0; d < 10; d++) q[d] = (c0 + d < QW && 4 * (c0 + d) < limit + 8) ? (qrow[c0 + d] & 0x7F7F7F7Fu) : 0u;
#pragma unroll
        for (int d = 7; d >= 0; d--) {
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                const u32 x = k ? alignbit(q[d + 1], q[d], 8 * k) : q[d];
                u32 sdiff = sum_bytes(x & keep_lo, nthr);
                if (w > 4) {
                    const u32 y = k ? alignbit(q[d + 2], q[d + 1], 8 * k) : q[d + 1];
                    sdiff = sum_bytes(y & keep_hi, sdiff);
                }
                m = alignbit(m, sdiff, 31);  // m = m << 1 | (sum < thr)
            }
        }
        return m;
    }
    for (int d = 0; d < 8; d++) {
        const int c = 8 * word + d;
        if (c >= QW || 4 * c >= limit) break;
        u32 s4[4];
        window_sums4(qrow, c, QW, w, s4);
        m |= ((u32)((int)s4[0] < thr) | ((u32)((int)s4[1] < thr) << 1) | ((u32)((int)s4[2] < thr) << 2) |
              ((u32)((int)s4[3] < thr) << 3)) << (4 * d);
    }
    return m;
}

// The sliding windows of Filter::trimAndCut (filter.cpp:97-194) evaluated for EVERY start
// position at once, as per-read bit masks that trim_and_cut() then only bit-scans.  A window sum
// is position-absolute, so the reference's rolling sum at position s equals the mask's window
// [s, s+w).  Item = (mask word w, read R): one lane builds bits 32w..32w+31 of every mask of its
// read from 8 quality dwords (+ the look-ahead the windows need) and stores the words - no
// atomics, and nothing to clear beforehand.  The read index runs fastest across lanes.
// (cut_right's per-base test "quality < 33+Q", filter.cpp:159, is not a mask: the scan it belongs to starts at
// the window found and ends within a few bases, so trim_and_cut reads the quality bytes themselves.)
FQ_DEV void phase_masks(const KernelArgs& a, u32* lds, int n_valid, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    if (!L.wm_stride) return;
    const DevParams& p = a.p;
    // argument-block fields fetched once, not at every use inside the loops
    const int NR = L.NR, QW = L.QW, WW = L.wm_words, wm_stride = L.wm_stride;
    const int oF = L.wm_badF, oR = L.wm_badR, oT = L.wm_badT, oN = L.wm_isN;
    const int wF = p.wF, wR = p.wR, wT = p.wT, thrF = p.thrF, thrR = p.thrR, thrT = p.thrT;
    const int max_len = p.max_len;
    const int* rlen0_v = lds_i(lds, L.rlen0);
    const u32* qual_v = lds + L.qual;
    u32* wm_v = lds + L.wm;
    const int total = NR * WW;
    const int dW = nthreads / NR, dR = nthreads - dW * NR;
    int w = tid / NR, R = tid - w * NR;
    for (int i = tid; i < total; i += nthreads) {
        const u32* qrow = qual_v + rowoff(R, QW);
        const int rl0 = rlen0_v[R];
        u32* wm = wm_v + rowoff(R, wm_stride) + w;
        if (oF >= 0) wm[oF] = wF <= max_len ? bad_window_word(qrow, w, QW, rl0, wF, thrF) : 0u;
        if (oR >= 0) wm[oR] = wR <= max_len ? bad_window_word(qrow, w, QW, rl0, wR, thrR) : 0u;
        if (oT >= 0) wm[oT] = wT <= max_len ? bad_window_word(qrow, w, QW, rl0, wT, thrT) : 0u;
        if (oN >= 0) {
            u32 mN = 0;
            for (int d = 0; d < 8; d++) {
                const int c = 8 * w + d;
                if (c >= QW || 4 * c >= rl0) break;
                const u32 nb = (qrow[c] >> 7) & 0x01010101u;
                mN |= ((nb | (nb >> 7) | (nb >> 14) | (nb >> 21)) & 0xFu) << (4 * d);
            }
            wm[oN] = mN;
        }
        w += dW;
        R += dR;
        if (R >= NR) { R -= NR; w++; }
    }
    (void)n_valid;
}

// ---------------------------------------------------------------------------
// Stats::statRead (stats.cpp:191-291) over a tile.
// lane = (read R, quality dword c) = cycles 4c..4c+3 of that read; the loop has a wave-uniform
// trip count (ballots inside).
//   ST_PRE  : every read, window [0, rlen0)                          -> slots PRE1 / PRE2
//   ST_POST : reads flagged RS_STAT_POST, window [front, front+len)  -> slots POST1 / POST2
//   ST_BOTH : one pass doing both, legal when no option can move or edit a base that is kept
//             (DevParams::stats_one_pass): front == 0 always, so base j of a read that is
//             written out sits in cycle j before and after filtering.  Base j goes to the
//             "kept" accumulators (stored in the POST slot) when the read is written out and
//             j < len, else to the "dropped" ones (stored in the PRE slot); the slab fold
//             forms PRE = kept + dropped, POST = kept.
// ---------------------------------------------------------------------------
enum { ST_PRE = 0, ST_POST = 1, ST_BOTH = 2 };

// per-cycle accumulator of (Stats slot, class, cycle), in u64 units from LdsLayout::acc_cyc
FQ_DEV int cyc_index(int Cp, int slot, int cls, int pos) { return (slot * Cp + pos) * N_CLS + cls; }

// one (read R, quality dword c) item of Stats::statRead; BALLOT: aggregate the quality histogram
// over the wavefront (every lane of the wave must call this, `act` says whether it has an item)
template <int mode, bool MERGE, bool BALLOT>
FQ_DEV void stats_item(const KernelArgs& a, u32* lds, bool act, int R, int c, int n_valid, int lane, u32 copy,
                       bool count_read = true) {
    const LdsLayout& L = a.L;
    const int Cp = L.Cp;
    u64* cyc_all = (u64*)(lds + L.acc_cyc);
    u32* kmer_all = lds + L.acc_kmer;
    u32* qh_all = lds + L.acc_qh;
    u32* misc = lds + L.acc_misc;
    (void)copy;
    const int m = R >= L.P ? 1 : 0;
    const int rl0 = lds_i(lds, L.rlen0)[R];
    const int rflags = lds_i(lds, L.flags)[R];
    const bool out_ok = (rflags & RS_STAT_POST) != 0;
    const bool rc = MERGE && mode == ST_POST && (rflags & RS_POST_RC) != 0;  // tail of a merged read
    int f = 0, l = rl0, lk = 0, rc_base = 0;
    if (mode == ST_POST) {
        act = act && out_ok;
        f = lds_i(lds, L.front)[R];
        l = lds_i(lds, MERGE ? L.mlen : L.len)[R];
        if (rc) rc_base = lds_i(lds, L.mlen)[R - L.P] + l - 1;  // merged position of r2'[i] = len1 + len2 - 1 - i
    } else {
        act = act && (R - m * L.P < n_valid);  // rows past the end of the batch do not exist
        if (mode == ST_BOTH && out_ok) lk = lds_i(lds, L.len)[R];
    }
    int slot0 = m * 2 + (mode == ST_POST ? 1 : 0);  // PRE1=0 POST1=1 PRE2=2 POST2=3
    if (MERGE && mode == ST_POST && (rflags & RS_POST_TO1)) slot0 = 1;
    if (act && c == 0 && count_read) {  // mReads++, mLengthSum += len (stats.cpp:194, 290)
        lds_add_u32(&misc[MISC_STAT_READS + slot0], rc ? 0u : 1u);  // a merged read is ONE read
        lds_add_u32(&misc[MISC_STAT_LENSUM + slot0], (u32)l);
        if (mode == ST_BOTH && out_ok) {
            lds_add_u32(&misc[MISC_STAT_READS + slot0 + 1], 1u);
            lds_add_u32(&misc[MISC_STAT_LENSUM + slot0 + 1], (u32)lk);
        }
    }
    const int j0 = c * 4;
    act = act && j0 < f + l && j0 + 4 > f;
    u32 qd = 0, codes = 0, nbits = 0xFFu;
    if (act) {
        const u32* srow = lds + L.seq + rowoff(R, L.SW);
        qd = lds[L.qual + rowoff(R, L.QW) + c];
        const u32 cur8 = (srow[c >> 2] >> ((c & 3) * 8)) & 0xFFu;
        u32 prev8 = 0, nprev = 0xFu;  // before the read start: "invalid"
        if (c > 0) {
            prev8 = (srow[(c - 1) >> 2] >> (((c - 1) & 3) * 8)) & 0xFFu;
            const u32 nb = (lds[L.qual + rowoff(R, L.QW) + c - 1] >> 7) & 0x01010101u;
            nprev = (nb | (nb >> 7) | (nb >> 14) | (nb >> 21)) & 0xFu;  // bit k = base j0-4+k is N
        }
        const u32 nbc = (qd >> 7) & 0x01010101u;
        codes = prev8 | (cur8 << 8);  // bases j0-4 .. j0+3, 2 bits each
        nbits = nprev | (((nbc | (nbc >> 7) | (nbc >> 14) | (nbc >> 21)) & 0xFu) << 4);  // same 8 bases, 1 bit each
    }
    // ---- per base: per-cycle counters and 5-mers; collect the quality-histogram keys ----
    u32 key[4];
    bool val[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = j0 + k;
        val[k] = act && j >= f && j < f + l;
        const int slot = slot0 + ((mode == ST_BOTH && j < lk) ? 1 : 0);
        const u32 q = (qd >> (k * 8)) & 0x7Fu;
        key[k] = (u32)slot * 128u + q;
        if (val[k]) {
            const int wpos = j - f;                      // index inside the window
            const int pos = rc ? rc_base - wpos : wpos;  // cycle
            const u32 isn = (nbits >> (4 + k)) & 1u;
            const u32 cls = isn ? (u32)CLS_N : (((codes >> (8 + 2 * k)) & 3u) ^ (rc ? 1u : 0u));
            // stats.cpp:209-222: q30 ('?') counts into Q30 and Q20, q20 ('5') into Q20
            const u64 inc = 1ull | ((u64)(q >= 53u) << CYC_Q20_SHIFT) | ((u64)(q >= 63u) << CYC_Q30_SHIFT) |
                            ((u64)(q - 33u) << CYC_QSUM_SHIFT);
            lds_add_u64(&cyc_all[cyc_index(Cp, slot, (int)cls, pos)], inc);
            // 5-mer ending at this base (stats.cpp:224-266): counted iff the five bases
            // pos-4..pos all exist in the window and none of them is N
            if (wpos >= 4 && ((nbits >> k) & 0x1Fu) == 0u) {
                u32 km = (codes >> (2 * k)) & 0x3FFu;  // earliest base in the low bits
                if (rc) km = (reverse_groups(km) >> 22) ^ 0x155u;  // the same five bases on the merged strand
                lds_add_u32(&kmer_all[slot * KMER_BINS + km], 1u);
            }
        }
    }
    // ---- mBaseQualHistogram[qual]++ (:207).  Qualities cluster on a few values, so the
    // wave first counts the bases equal to one lane's (slot, quality) key with ballots and
    // lets that lane add the total; only the other bases pay an LDS atomic each.
    if (!BALLOT) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (val[k]) lds_add_u32(&qh_all[key[k] * QT_DWORDS + QT_COUNT], 1u);
        return;
    }
    const bool have = val[0] | val[1] | val[2] | val[3];
    const u32 mine = val[0] ? key[0] : val[1] ? key[1] : val[2] ? key[2] : key[3];
    const u64 hv = ballot(have);
    if (hv) {  // wave-uniform
        const int src = ffs64(hv) - 1;
        const u32 modek = shfl(mine, src);
        u32 cnt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool mk = val[k] && key[k] == modek;
            cnt += (u32)popc64(ballot(mk));
            if (val[k] && !mk) lds_add_u32(&qh_all[key[k] * QT_DWORDS + QT_COUNT], 1u);
        }
        if (lane == src) lds_add_u32(&qh_all[modek * QT_DWORDS + QT_COUNT], cnt);
    }
}

template <int mode, bool MERGE>
FQ_DEV void phase_stats(const KernelArgs& a, u32* lds, int n_valid, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int qwg = a.p.qw_g;
    const int total = L.NR * qwg;
    const int lane = tid & 63;
    const u32 copy = 0u;
    for (int base = tid - lane; base < total; base += nthreads) {  // wave-uniform trip count (ballots inside)
        const int idx = base + lane;
        const bool act = idx < total;
        const int R = act ? (int)fastdiv((u32)idx, a.magic_qwg) : 0;
        const int c = act ? idx - R * qwg : 0;
        stats_item<mode, MERGE, true>(a, lds, act, R, c, n_valid, lane, copy);
    }
}

// ---------------------------------------------------------------------------
// One-pass Stats, fast form.  lane = (read R, quarter of its quality dwords): everything that only depends
// on the read - its two windows, the kept / dropped accumulator bases, the row addresses - is set up once per
// lane, then the lane walks its ~QW/4 dwords.  The walk starts at a read-dependent dword and wraps, so the
// lanes of a wavefront sit on different cycles (no same-address atomics) and their row reads fall into
// different banks.
// Almost every (read, quality dword) item is "plain": no N among its four bases or the four before, all of
// its existing bases kept or all dropped.  A plain item is branch-free:
//   * the four quality characters index the quality table of the item's Stats slot (one 16-byte entry per
//     character: packed per-cycle increment, histogram counter, the constant 1); character 0 = a byte past
//     the read's end (phase_trim zeroes those) has increment 0 and constant 0, so the partial last dword of a
//     read needs no special case,
//   * per-cycle counters sit [slot][cycle][class]: the four cycles of the dword are at fixed distances (DS
//     offset field) and the class adds 8 bytes,
//   * the histogram counter shares the address of the table entry already computed; dwords whose four
//     characters all equal the wavefront's mode are counted with one ballot instead.
// The others - the dword a kept length cuts, dwords with N - are queued in an LDS work list and run through
// stats_item afterwards, so that no wavefront pays for both paths.
// ---------------------------------------------------------------------------
// the work list is full (a tile of reads riddled with N): the item is done in place - out of line, so that this
// rare case does not drag the general path into the fast loop's code
__device__ __attribute__((noinline)) void stats_item_overflow(const KernelArgs* a, u32* lds, int R, int c, int n_valid) {
    stats_item<ST_BOTH, false, false>(*a, lds, true, R, c, n_valid, 0, 0u, false);
}

FQ_DEV void phase_stats_both(const KernelArgs& a, u32* lds, int n_valid, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int QW = L.QW, NR = L.NR, P = L.P, SW4 = L.SW * 4, Cp = L.Cp;
    const int S = (QW + 3) >> 2;                // dwords per lane
    const int rot_mask = S >= 8 ? 7 : (S >= 4 ? 3 : 0);
    u32* misc = lds + L.acc_misc;
    u32* wl_count = lds + L.wl;
    u16* wl_items = (u16*)(lds + L.wl + 1);
    const int wl_cap = L.wl_cap;
    const u32* swin_v = lds + L.swin;           // rlen0 | kept length << 16, left by the filter phase
    const u8* lds_b = (const u8*)lds;
    // byte offsets from this half's LDS base; the shared accumulators sit BELOW the base of the second tile slot
    const int cyc_b = L.acc_cyc * 4, kmer_b = L.acc_kmer * 4, qt_b = L.acc_qh * 4;
    const int qual_b = L.qual * 4, seq_b = L.seq * 4;
    const int lane = tid & 63;
    const int total = NR * 4;
#ifdef FQ_PROFILE_ABLATION
    const u32 dbg = a.debug_skip;   // profiling build only: 64 no per-cycle atomics, 128 no k-mer atomics, 256 no histogram atomics
#else
    const u32 dbg = 0;
#endif
    for (int base = tid - lane; base < total; base += nthreads) {  // wave-uniform trip count (ballots inside)
        const int task = base + lane;
        const bool tv = task < total;
        const int R = tv ? (task >> 2) : 0, seg = task & 3;
        const int m = R >= P ? 1 : 0;
        const bool rv = tv && (R - m * P < n_valid);
        const u32 sw = rv ? swin_v[R] : 0u;
        const int rl0 = (int)(sw & 0xFFFFu);
        const int lk = (int)(sw >> 16);         // > 0 exactly for the reads that are written out
        const int slot_d = 2 * m;               // dropped bases -> the PRE slot, kept ones -> the POST slot
        if (rv && seg == 0) {                   // mReads++, mLengthSum += len (stats.cpp:194, 290)
            lds_add_u32(&misc[MISC_STAT_READS + slot_d], 1u);
            lds_add_u32(&misc[MISC_STAT_LENSUM + slot_d], (u32)rl0);
            if (lk) {
                lds_add_u32(&misc[MISC_STAT_READS + slot_d + 1], 1u);
                lds_add_u32(&misc[MISC_STAT_LENSUM + slot_d + 1], (u32)lk);
            }
        }
        const int cany = (rl0 + 3) >> 2;        // dwords holding at least one base
        const int ck = lk >= rl0 ? cany : (lk >> 2);   // dwords below ck: every existing base is kept
        const int cd = (lk + 3) >> 2;           // dwords from cd on: every base is dropped
        const int cbeg = seg * S;
        const int cmax = imin(imin(cbeg + S, QW), cany);
        // this read's rows and the two accumulator sets (kept / dropped) as LDS pointers, set up once per lane
        const u8* qrow_p = lds_b + (qual_b + rowoff(R, QW * 4));
        const u8* srow_p = lds_b + (seq_b + rowoff(R, SW4));
        u8* ldsw = (u8*)lds;
        u8* cyc_d = ldsw + (cyc_b + slot_d * Cp * (N_CLS * 8));
        u8* cyc_k = cyc_d + Cp * (N_CLS * 8);
        u8* kmer_d = ldsw + (kmer_b + slot_d * (KMER_BINS * 4));
        u8* kmer_k = kmer_d + KMER_BINS * 4;
        u8* qt_d = ldsw + (qt_b + slot_d * (128 * QT_DWORDS * 4));
        u8* qt_k = qt_d + 128 * QT_DWORDS * 4;
        int cc = R & rot_mask;
        u32 mode_ta = 0x7FFFFFFFu;   // wave-uniform: LDS byte offset (from this half's base) of the mode's table entry
        u32 agg_cnt = 0;    // per lane: bases that hit it
        for (int t = 0; t < S; t++) {
            const int c = cbeg + cc;
            const bool act = rv && c < cmax;
            const int cr = act ? c : 1;
            // one DS instruction per row: the quality dword and the one before it (ds_read2_b32), and the two base
            // dwords that hold base bytes cr - 1 and cr.  (For cr == 0 "the one before" is whatever precedes the row:
            // read, never used.)  The LDS pipe takes one wave-instruction per ~4.5 cycles per CU whatever its width,
            // so the number of DS instructions is what this loop pays for.
            const u32* qw = (const u32*)(qrow_p + 4 * cr);
            const u32 qp = qw[-1], qd = qw[0];
            const u32* sw2 = (const u32*)(srow_p + ((cr - 1) & ~3));
            const u32 pc16 = alignbit(sw2[1], sw2[0], (u32)((cr - 1) & 3) * 8u);
            const u32 cur8 = (pc16 >> 8) & 0xFFu;   // (pc16's low byte: the base byte in front, part of the 5-mer word below)
            const bool kept = c < ck;
            const u32 nany = (qd | (c > 0 ? qp : 0u)) & 0x80808080u;   // an N among the four bases or the four before
            const bool plain = (int)act & (int)(nany == 0u) & ((int)kept | (int)(c >= cd));
            if (act && !plain) {                // rare: hand it to the general path
                const u32 slot = lds_add_ret_u32(wl_count, 1u);
                if (slot < (u32)wl_cap) wl_items[slot] = (u16)(R * QW + c);
                else stats_item_overflow(&a, lds, R, c, n_valid);  // list full: do it here
            }
            // The wavefront's mode = the table entry (Stats slot AND character) of the first plain kept item's first
            // character, fixed at its first appearance.  Bases that hit that entry are counted with a ballot and added
            // once per wavefront: most of a tile's characters are one value, and as LDS atomics they would all land on
            // one address and serialise.
            if (mode_ta == 0x7FFFFFFFu) {       // wave-uniform
                const u64 cand = ballot(plain && kept);
                if (cand) {
                    const int src = ffs64(cand) - 1;
                    mode_ta = shfl((u32)(int)(qt_k - ldsw) + ((qd & 0x7Fu) << 4), src);
                }
            }
            if (plain) {
                u8* qt = kept ? qt_k : qt_d;
                u8* cyc = (kept ? cyc_k : cyc_d) + mul24((u32)c, 4u * N_CLS * 8u);
                u8* kmer = kept ? kmer_k : kmer_d;
                u8* ta[4];
                u64 inc[4];
                u32 one[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {   // table entry of each character: {increment u64, counter}
                    ta[k] = qt + (bfe(qd, 8 * k, 7) << 4);
                    inc[k] = *(const u64*)ta[k];
                    one[k] = (u32)inc[k] & 1u;  // the count field's increment: 1 for a base, 0 for character 0
                }
                if (!(dbg & 64u)) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    lds_add_u64((u64*)(cyc + (bfe(cur8, 2 * k, 2) << 3) + k * (N_CLS * 8)), inc[k]);
                }
                if (c > 0 && !(dbg & 128u)) {  // 5-mers ending at 4c..4c+3 (positions >= 4, no N); a base past the read's end adds 0
                    const u32 codes = pc16 & 0xFFFFu;
#pragma unroll
                    for (int k = 0; k < 4; k++) lds_add_u32((u32*)(kmer + (bfe(codes, 2 * k, 10) << 2)), one[k]);
                }
                // a dword of four equal characters (the tail of a read after its quality dropped, for one) adds 4 once
                const bool same4 = ta[0] == ta[1] && ta[0] == ta[2] && ta[0] == ta[3];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool is_mode = (u32)(int)(ta[k] - ldsw) == mode_ta;
                    agg_cnt += is_mode ? 1u : 0u;   // per lane; folded over the wavefront after the loop
                    if (!is_mode && !(dbg & 256u) && (k == 0 || !same4))
                        lds_add_u32((u32*)(ta[k] + QT_COUNT * 4), (k == 0 && same4) ? 4u * one[0] : one[k]);
                }
            }
            cc = cc + 1 == S ? 0 : cc + 1;
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) agg_cnt += shfl_xor(agg_cnt, sh);
        if (lane == 0 && agg_cnt) lds_add_u32((u32*)(ldsw + (int)mode_ta + QT_COUNT * 4), agg_cnt);
    }
    tile_sync(a, lds, nthreads);
    // the queued items, all through the general path
    const int nw = (int)imin((int)*wl_count, L.wl_cap);
    for (int base = tid - lane; base < nw; base += nthreads) {
        const int i = base + lane;
        const bool act = i < nw;
        const int idx = act ? (int)wl_items[i] : 0;
        const int Rw = (int)fastdiv((u32)idx, a.magic_qwg);
        stats_item<ST_BOTH, false, true>(a, lds, act, Rw, idx - Rw * QW, n_valid, lane, 0u, false);  // reads were counted above
    }
}

// ---------------------------------------------------------------------------
// bit scans over a position mask (bit j of word j>>5 = predicate at base j)
// ---------------------------------------------------------------------------
// first j in [lo, hi) whose bit == want; hi if there is none
FQ_DEV int scan_first(const u32* m, int lo, int hi, bool want) {
    if (lo >= hi) return hi;
    int w = lo >> 5;
    const int wend = (hi - 1) >> 5;
    // two words per LDS round trip (the second index clamped: a duplicate of the last word is harmless)
    u32 r0 = m[w], r1 = m[w < wend ? w + 1 : wend];
    u32 x = (want ? r0 : ~r0) & ~lowmask32(lo & 31);
    for (;;) {
        if (x) {
            const int j = (w << 5) + ffs32(x) - 1;
            return j < hi ? j : hi;
        }
        if (++w > wend) return hi;
        x = want ? r1 : ~r1;
        if (x) continue;
        if (++w > wend) return hi;
        r0 = m[w];
        r1 = m[w < wend ? w + 1 : wend];
        x = want ? r0 : ~r0;
    }
}
FQ_DEV int scan_last(const u32* m, int lo, int hi, bool want) {
    if (lo >= hi) return lo - 1;
    int w = (hi - 1) >> 5;
    const int wbeg = lo >> 5;
    u32 x = (want ? m[w] : ~m[w]) & lowmask32(((hi - 1) & 31) + 1);
    for (;;) {
        if (x) {
            const int j = (w << 5) + 31 - clz32(x);
            return j >= lo ? j : lo - 1;
        }
        if (--w < wbeg) return lo - 1;
        x = want ? m[w] : ~m[w];
    }
}

// ---------------------------------------------------------------------------
// Filter::trimAndCut (filter.cpp:68-207) on one read.  Returns false for NULL.
// wm = the read's predicate masks (phase_masks), u = row position of the read's
// first base as trimAndCut sees it (after the UMI front trim), len = its length.
// Every loop of the reference becomes one bit scan; comments give the loop it replaces.
// ---------------------------------------------------------------------------
// first j in [lo, hi) whose quality character (row position j) is below qmin; hi if there is none.
// qmin4 = qmin in every byte (0..127); four characters per step: bit 7 of (q|0x80) - qmin survives iff q >= qmin
FQ_DEV int scan_first_lowq(const u32* qrow, int lo, int hi, u32 qmin4) {
    if (lo >= hi) return hi;
    int w = lo >> 2;
    const int wend = (hi - 1) >> 2;
    u32 x = ~(((qrow[w] & 0x7F7F7F7Fu) | 0x80808080u) - qmin4) & 0x80808080u & ~lowmask32(8 * (lo & 3));
    for (;;) {
        if (x) {
            const int j = 4 * w + ((ffs32(x) - 1) >> 3);
            return j < hi ? j : hi;
        }
        if (++w > wend) return hi;
        x = ~(((qrow[w] & 0x7F7F7F7Fu) | 0x80808080u) - qmin4) & 0x80808080u;
    }
}

FQ_DEV bool trim_and_cut(const DevParams& p, const LdsLayout& L, const u32* wm, const u32* qrow, int u, int len, int front, int tail,
                         int& out_front, int& out_len) {
    out_front = 0;
    out_len = len;
    const bool enF = p.cut_front, enT = p.cut_tail, enR = p.cut_right;
    if (front == 0 && tail == 0 && !enF && !enT && !enR) return true;  // :71-72
    int rlen = len - front - tail;
    if (rlen < 0) return false;  // :76-77
    if (!enF && !enT && !enR) {  // :79-89
        out_front = front;
        out_len = rlen;
        return true;
    }
    const int l = len;
    if (enF) {  // :97-127
        const int w = p.wF;
        if (l - front - tail - w <= 0) return false;
        const int end = l - tail - w;  // for (s = front; s + w < l - tail; s++) ... break at the first good window
        int s = scan_first(wm + L.wm_badF, u + front, u + end, false) - u;  // no break: s == end
        if (s > 0) s = s + w - 1;
        s = scan_first(wm + L.wm_isN, u + s, u + l, false) - u;  // while (s < l && seq[s] == 'N') s++
        front = s;
        rlen = l - front - tail;
    }
    if (enR) {  // :130-163
        const int w = p.wR;
        if (l - front - tail - w <= 0) return false;
        const int end = l - tail - w;
        int s = scan_first(wm + L.wm_badR, u + front, u + end, true) - u;  // first window below the threshold
        if (s < end) {  // foundLowQualWindow
            const u32 qmin4 = (u32)imin(imax(p.qRmin, 0), 127) * 0x01010101u;
            s = scan_first_lowq(qrow, u + s, u + l - 1, qmin4) - u;  // while (s < l-1 && qual[s] >= 33+Q) s++
            rlen = s - front;
        }
    }
    if (!enR && enT) {  // :166-194
        const int w = p.wT;
        if (l - front - tail - w <= 0) return false;
        // for (t = l-tail-1; t - w >= front; t--): window [t-w+1, t]; break at the first good one.
        // In window-start coordinates s' = t-w+1 runs from l-tail-w down to front+1.
        const int sp = scan_last(wm + L.wm_badT, u + front + 1, u + l - tail - w + 1, false) - u;  // none: front
        int t = sp + w - 1;  // no break: t == front + w - 1, the loop's exit value
        if (t < l - 1) t = t - w + 1;
        t = scan_last(wm + L.wm_isN, u, u + t + 1, false) - u;  // while (t >= 0 && seq[t] == 'N') t--
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return false;  // :196-197
    out_front = front;
    out_len = rlen;
    return true;
}

// PolyX::trimPolyG (polyx.cpp:16-42): new length of the read [f, f+rlen)
FQ_DEV int trim_poly_g(const u32* srow, const u8* q, int f, int rlen, int compareReq) {
    int mismatch = 0, i = 0, firstGPos = rlen - 1;
    for (i = 0; i < rlen; i++) {
        const int j = f + rlen - i - 1;
        if (sym_at(srow, q, j) != (u32)CODE_G) mismatch++;
        else firstGPos = rlen - i - 1;
        const int allowed = (i + 1) / 8;
        if (mismatch > 5 || (mismatch > allowed && i >= compareReq - 1)) break;
    }
    if (i >= compareReq && firstGPos >= 0) return firstGPos;
    return rlen;
}

// PolyX::trimPolyX (polyx.cpp:49-116).  poly = -1 when nothing is recorded.
FQ_DEV int trim_poly_x(const u32* srow, const u8* q, int f, int rlen, int compareReq, int& poly, int& trimmed) {
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    int pos = 0;
    poly = -1;
    trimmed = 0;
    for (pos = 0; pos < rlen; pos++) {
        const u32 s = sym_at(srow, q, f + rlen - pos - 1);
        cnt0 += (s == 0u) | (s == 4u);  // N counts toward all (:79-85)
        cnt1 += (s == 1u) | (s == 4u);
        cnt2 += (s == 2u) | (s == 4u);
        cnt3 += (s == 3u) | (s == 4u);
        const int cmp = pos + 1;
        const int allowed = imin(5, cmp / 8);
        const bool needToBreak = !((cmp - cnt0 <= allowed) | (cmp - cnt1 <= allowed) | (cmp - cnt2 <= allowed) |
                                   (cmp - cnt3 <= allowed));
        if (needToBreak && (pos >= 8 || pos + 1 >= compareReq - 1)) break;
    }
    if (pos + 1 >= compareReq) {  // :98-115
        int best = 0, mx = cnt0;  // first maximum in the order A,T,C,G
        if (cnt1 > mx) { mx = cnt1; best = 1; }
        if (cnt2 > mx) { mx = cnt2; best = 2; }
        if (cnt3 > mx) { mx = cnt3; best = 3; }
        // :109  while(data[rlen-pos-1] != polyBase && pos>=0) pos--;
        // index -1 (scan never broke) and index rlen (terminating 0) never equal the poly base
        for (;;) {
            const int idx = rlen - pos - 1;
            const bool is_poly = (idx >= 0 && idx < rlen) && sym_at(srow, q, f + idx) == (u32)best;
            if (!(!is_poly && pos >= 0)) break;
            pos--;
        }
        poly = best;
        trimmed = pos + 1;
        const int newlen = rlen - pos - 1;
        if (newlen < 0 || newlen > rlen) return rlen;  // Read::resize ignores (read.cpp:62-64)
        return newlen;
    }
    return rlen;
}

// ---------------------------------------------------------------------------
// Phase C1: Duplicate::seq2intvector (duplicate.cpp:111-120) on the ORIGINAL reads
// (peprocessor.cpp:398, quirk #11).  8 lanes per read, lane s summing quality dwords s, s+8, ...
// (consecutive lanes -> consecutive primes: conflict-free table reads), two buffers at a
// time; a 3-step shuffle folds the 8 partial sums.
//   h_i = sum_p prime[((p+off)*B+i) & mask] * val(base_p)     (val: A7 T222 C74 G31 else 13)
// the position part sum_p prime[...]*(p+off) only depends on the lengths and comes from a
// host-built prefix table (DevLuts::dup_posum).
// ---------------------------------------------------------------------------
// Duplicate's primes into LDS: the byte-plane table of the dot-product hash, or the plain list (generic path)
FQ_DEV void stage_primes(const KernelArgs& a, u32* lds, int tid, int nt) {
    const LdsLayout& L = a.L;
    if (!a.p.dup_enabled) return;
    if (L.has_hp) {
        const int n = 4 * L.hp_nq * a.p.dup_bufnum * a.p.dup_npl;
        for (int i = tid; i < n; i += nt) lds[L.hp + i] = a.lut.dup_planes[i];
    } else {
        for (int i = tid; i < 512 * a.p.dup_bufnum; i += nt) lds[L.primes + i] = a.lut.dup_primes[i];
    }
}

// The same sum as byte-plane dot products.  prime * val = sum_b 2^(8b) * byte_b(prime) * val, so for the four bases
// of one packed byte the contribution to h_i is sum_b 2^(8b) * dot4(vals, plane_b) with vals = the four base
// values (one byte each, from the 256-entry LUT) and plane_b = the b-th bytes of the four primes - ONE
// v_dot4_u32_u8 per (buffer, plane) and 4 bases instead of a multiply and a 64-bit add per (buffer, base).  The
// planes come from a host-built table indexed by stream position (DevLuts::dup_planes), which absorbs read 2's
// start offset (duplicate.cpp:139) for any read-1 length.  4 lanes per read, base dwords interleaved; the partial
// dot products of a read stay far below 2^32 (150 * 4 * 255 * 222).
template <int B, int NPL>
FQ_DEV void phase_hash_dot(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const int P = L.P, SW4 = L.SW * 4, QW = L.QW, NQ = L.hp_nq;
    const u32* hp = lds + L.hp;
    const u32* val4 = lds + L.val4_lut;
    const int* rlen0_v = lds_i(lds, L.rlen0);
    const int* flags_v = lds_i(lds, L.flags);
    const u8* seq_bytes = (const u8*)(lds + L.seq);
    const u32* qual_v = lds + L.qual;
    u64* hash_v = (u64*)(lds + L.hash);
    const int total = L.NR * 4;
    const int lane = tid & 63;
    for (int t0 = tid - lane; t0 < total; t0 += nthreads) {  // wave-uniform trip count (lane exchanges inside)
        const int t = t0 + lane;
        const bool valid = t < total;
        const int R = valid ? (t >> 2) : 0, seg = t & 3;
        const int len = valid ? rlen0_v[R] : 0;
        const int off = R >= P ? rlen0_v[R - P] : 0;  // r2 continues at r1->length() (duplicate.cpp:139)
        const bool hasN = (flags_v[R] & RS_HAS_N) != 0;
        const int rq = rowoff(R, QW), rs = rowoff(R, SW4);
        const u32* tb = hp + ((off & 3) * NQ + (off >> 2)) * (B * NPL);
        u32 acc[B * NPL];
#pragma unroll
        for (int k = 0; k < B * NPL; k++) acc[k] = 0;
        for (int c = seg; 4 * c < len; c += 4) {
            u32 vals = val4[seq_bytes[rs + c]];  // the four base values, one byte each
            if (hasN) {
                const u32 qd = qual_v[rq + c];
                const u32 mN = ((qd >> 7) & 0x01010101u) * 0xFFu;  // N -> 13 (duplicate.cpp:92-109)
                vals = (vals & ~mN) | (0x0D0D0D0Du & mN);
            }
            const int rem = len - 4 * c;
            if (rem < 4) vals &= lowmask32(8 * rem);  // bases past the read end contribute 0
            const u32* e = tb + c * (B * NPL);
#pragma unroll
            for (int k = 0; k < B * NPL; k++) acc[k] = dot4_u8(vals, e[k], acc[k]);
        }
#pragma unroll
        for (int i = 0; i < B; i++) {
            u64 h = (u64)acc[i * NPL] + ((u64)acc[i * NPL + 1] << 8) + ((u64)acc[i * NPL + 2] << 16);
            if (NPL > 3) h += (u64)acc[i * NPL + 3] << 24;
            h = sum4_u64(h);
            if (valid && seg == 0) hash_v[(size_t)R * B + i] = h;
        }
    }
}

FQ_DEV void phase_hash_generic(const KernelArgs& a, u32* lds, int tid, int nthreads);

FQ_DEV void phase_hash(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const DevParams& p = a.p;
    if (!p.dup_enabled || !a.dup_pos) return;
    if (a.L.has_hp) {
        if (p.dup_bufnum == 2 && p.dup_npl == 3) return phase_hash_dot<2, 3>(a, lds, tid, nthreads);
        if (p.dup_bufnum == 4 && p.dup_npl == 3) return phase_hash_dot<4, 3>(a, lds, tid, nthreads);
        if (p.dup_bufnum == 2 && p.dup_npl == 4) return phase_hash_dot<2, 4>(a, lds, tid, nthreads);
        if (p.dup_bufnum == 4 && p.dup_npl == 4) return phase_hash_dot<4, 4>(a, lds, tid, nthreads);
    }
    phase_hash_generic(a, lds, tid, nthreads);
}

FQ_DEV void phase_hash_generic(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    // argument-block fields fetched once, not at every use inside the loops
    const int B = p.dup_bufnum, P = L.P, SW4 = L.SW * 4, QW = L.QW;
    const u32 mask = (u32)(512 * B - 1);
    const u32* primes = lds + L.primes;
    const u32* val4 = lds + L.val4_lut;
    const int* rlen0_v = lds_i(lds, L.rlen0);
    const u8* seq_bytes = (const u8*)(lds + L.seq);
    const u32* qual_v = lds + L.qual;
    u64* hash_v = (u64*)(lds + L.hash);
    const int D = (p.qw_g + 7) >> 3;  // quality dwords per lane
    const int total = L.NR * 8;
    const int lane = tid & 63;
    for (int t0 = tid - lane; t0 < total; t0 += nthreads) {  // wave-uniform trip count (shuffles inside)
        const int t = t0 + lane;
        const bool valid = t < total;
        const int R = valid ? (t >> 3) : 0, seg = t & 7;
        const int len = valid ? rlen0_v[R] : 0;
        const int off = R >= P ? rlen0_v[R - P] : 0;  // r2 continues at r1->length() (duplicate.cpp:139)
        const int rq = rowoff(R, QW), rs = rowoff(R, SW4);
        for (int i0 = 0; i0 < B; i0 += 2) {
            u64 acc0 = 0, acc1 = 0;
            for (int d = 0; d < D; d++) {
                const int c = seg + 8 * d;
                const int rem = len - 4 * c;  // bases of this dword that exist
                if (rem <= 0) break;
                const u32 qd = qual_v[rq + c];
                u32 vals = val4[seq_bytes[rs + c]];  // the four base values, one byte each
                if (qd & 0x80808080u) {                  // N -> 13
                    const u32 mN = ((qd >> 7) & 0x01010101u) * 0xFFu;
                    vals = (vals & ~mN) | (0x0D0D0D0Du & mN);
                }
                vals &= lowmask32(8 * rem);              // bases past the read end contribute 0
                const u32 pi0 = mul24((u32)(4 * c + off), (u32)B) + (u32)i0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 v = (vals >> (8 * k)) & 0xFFu;
                    const u32 pi = (pi0 + (k == 0 ? 0u : k == 1 ? (u32)B : k == 2 ? 2u * (u32)B : 3u * (u32)B)) & mask;
                    acc0 += (u64)mul24(primes[pi], v);      // prime < 2^24, value < 2^8: exact in 32 bits
                    acc1 += (u64)mul24(primes[pi + 1], v);
                }
            }
            u32 lo0 = (u32)acc0, hi0 = (u32)(acc0 >> 32), lo1 = (u32)acc1, hi1 = (u32)(acc1 >> 32);
#pragma unroll
            for (int sh = 1; sh < 8; sh <<= 1) {
                const u64 o0 = (u64)shfl_xor(lo0, sh) | ((u64)shfl_xor(hi0, sh) << 32);
                const u64 o1 = (u64)shfl_xor(lo1, sh) | ((u64)shfl_xor(hi1, sh) << 32);
                const u64 n0 = (((u64)hi0 << 32) | lo0) + o0, n1 = (((u64)hi1 << 32) | lo1) + o1;
                lo0 = (u32)n0; hi0 = (u32)(n0 >> 32);
                lo1 = (u32)n1; hi1 = (u32)(n1 >> 32);
            }
            if (valid && seg == 0) {
                u64* h = hash_v + (size_t)R * B;
                h[i0] = ((u64)hi0 << 32) | lo0;
                h[i0 + 1] = ((u64)hi1 << 32) | lo1;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Phase C2: lane = one read.  UMI front trim (umiprocessor.cpp:19-49 / read.cpp:69-73),
// then Filter::trimAndCut on the predicate masks.
// ---------------------------------------------------------------------------
FQ_DEV void phase_trim(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    for (int R = tid; R < L.NR; R += nthreads) {
        const int m = R >= L.P ? 1 : 0;
        const int rl0 = lds_i(lds, L.rlen0)[R];
        if (p.stats_one_pass && !a.split) {
            // the one-pass Stats path reads quality character 0 as "no base here": make sure that is what the
            // row holds behind the read's end, whatever the caller's buffer had there
            u32* qrow = lds_qual(L, lds, R);
            int c0 = rl0 >> 2;
            if (rl0 & 3) {
                qrow[c0] &= lowmask32(8 * (rl0 & 3));
                c0++;
            }
            for (int c = c0; c < L.QW; c++) qrow[c] = 0;
        }
        if (a.dupflag) {  // --dedup: Duplicate::checkPair/checkRead already ran for this batch
            const int gp = tile_first + R - m * L.P;
            if (gp < a.n && a.dupflag[gp]) lds_or_i32(&lds_i(lds, L.flags)[R], RS_DUP);
        }
        int len = rl0;
        int front = 0;
        const int umi = m ? p.umi_len2 : p.umi_len1;
        if (umi > 0) {  // Read::trimFront(min(len,umi)+skip): len = min(length()-1, len)
            int t = imin(len, umi) + p.umi_skip;
            t = imin(len - 1, t);
            if (t > 0) { front = t; len -= t; }
        }
        int f2 = 0, l2 = len;
        const bool alive = trim_and_cut(p, L, lds + L.wm + rowoff(R, L.wm_stride), lds_qual(L, lds, R), front, len, m ? p.trim_front2 : p.trim_front1,
                                        m ? p.trim_tail2 : p.trim_tail1, f2, l2);
        if (alive) {
            lds_i(lds, L.front)[R] = front + f2;
            lds_i(lds, L.len)[R] = l2;
            lds_i(lds, L.ft)[R] = f2;  // frontTrimmed (without the UMI part)
        } else {
            lds_i(lds, L.front)[R] = front;
            lds_i(lds, L.len)[R] = len;
            lds_or_i32(&lds_i(lds, L.flags)[R], RS_NULL);
        }
    }
}

// polyG needs to know that BOTH mates survived trimAndCut (peprocessor.cpp:428-431)
FQ_DEV void phase_polyg(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    if (!p.poly_g) return;
    for (int R = tid; R < L.NR; R += nthreads) {
        const int m = R >= L.P ? 1 : 0;
        const int pr = R - m * L.P;
        int dead = lds_i(lds, L.flags)[pr] & RS_NULL;
        if (p.paired) dead |= lds_i(lds, L.flags)[L.P + pr] & RS_NULL;
        if (dead) continue;
        const int f = lds_i(lds, L.front)[R];
        const int l = lds_i(lds, L.len)[R];
        lds_i(lds, L.len)[R] = trim_poly_g(lds_seq(L, lds, R), (const u8*)lds_qual(L, lds, R), f, l, p.poly_g_min);
    }
}

// ---------------------------------------------------------------------------
// Phase D: OverlapAnalysis::analyze (overlapanalysis.cpp:17-89, the no-gap part).
//
// The reference slides rc(r2) along r1: forward offsets o = 0..len1-require-1 compare
// r1[o+i] with rc2[i], then reverse offsets o = 0..len2-require-1 compare r1[i] with
// rc2[o+i]; the FIRST offset in that order whose first min(ol,50) bases have at most
// limit(ol) mismatches wins (:34-44).  Both directions are the same problem "slide X over
// Y" with (X,Y) = (r1,rc2) / (rc2,r1), so a task is (pair, direction, quarter): 8 lanes per
// pair, the direction wave-uniform.  A lane takes blocks of 16 consecutive offsets; per
// offset a 16-base 2-bit XOR/popcount prefilter (a lower bound of the true count on a prefix
// every legal offset has: it can only over-accept) costs 8 VALU instructions; the rare
// survivors are verified exactly by the same lane, and the pair's winner is an LDS
// atomic-min over keys ordered like the reference's scan.
// ---------------------------------------------------------------------------
struct PairView {
    const u32 *s1, *n1;   // LDS rows of read 1: packed bases, N mask
    const u32 *rc, *rcn;  // LDS rows of rc(whole read 2) and its N mask (phase_rc)
    int f1, l1, z2, l2;   // r1' = row1[f1, f1+l1);  rc(r2') = rc[z2, z2+l2) with z2 = rlen0(r2) - (front2 + len2)
    bool hasN;
};
FQ_DEV void pair_view(const LdsLayout& L, u32* lds, int pr, PairView& v) {
    const int R1 = pr, R2 = L.P + pr;
    v.s1 = lds_seq(L, lds, R1);
    v.n1 = lds_nmk(L, lds, R1);
    v.rc = lds + L.rc + rowoff(pr, L.SW);
    v.rcn = lds + L.rcn + rowoff(pr, L.SW);
    v.f1 = lds_i(lds, L.front)[R1];
    v.l1 = lds_i(lds, L.len)[R1];
    v.l2 = lds_i(lds, L.len)[R2];
    v.z2 = lds_i(lds, L.rlen0)[R2] - (lds_i(lds, L.front)[R2] + v.l2);
    v.hasN = ((lds_i(lds, L.flags)[R1] | lds_i(lds, L.flags)[R2]) & RS_HAS_N) != 0;
}
// 16 bases of r1' starting at t (t >= 0; bases past l1 are whatever the row holds)
FQ_DEV u32 ov_r1(const PairView& v, int t) { return window16(v.s1, v.f1 + t); }
FQ_DEV u32 ov_r1n(const PairView& v, int t) { return window16(v.n1, v.f1 + t); }
// 16 bases of rc(r2') starting at t: rc[k] = comp(r2'[l2-1-k]); complement of a code is code^1,
// an N keeps code 0 on both strands (the rc of N is N, overlapanalysis.cpp:19-22 / simd.cpp:129)
FQ_DEV u32 ov_rc2(const PairView& v, int t) { return window16(v.rc, v.z2 + t); }
FQ_DEV u32 ov_rc2n(const PairView& v, int t) { return window16(v.rcn, v.z2 + t); }
template <int DIR> FQ_DEV u32 ov_x(const PairView& v, int t) { return DIR ? ov_rc2(v, t) : ov_r1(v, t); }
template <int DIR> FQ_DEV u32 ov_y(const PairView& v, int t) { return DIR ? ov_r1(v, t) : ov_rc2(v, t); }
template <int DIR> FQ_DEV u32 ov_xn(const PairView& v, int t) { return DIR ? ov_rc2n(v, t) : ov_r1n(v, t); }
template <int DIR> FQ_DEV u32 ov_yn(const PairView& v, int t) { return DIR ? ov_r1n(v, t) : ov_rc2n(v, t); }

// acceptNoGapOverlap (:34-44) for X shifted by o against Y; returns the key payload or -1
template <int DIR>
FQ_DEV int ov_verify(const PairView& v, int o, int lenX, int lenY, const short* lut) {
    const int ol = imin(lenX - o, lenY);
    const int limit = lut ? lut[ol] : 0;   // no table: diffPercentLimit 0 (--overlapped_out's analysis, peprocessor.cpp:489)
    const int pre = imin(ol, 50);  // complete_compare_require (:28)
    int cnt_pre = 0, cnt_full = 0;
    for (int t = 0; t < ol; t += 16) {
        u32 dd = fold_diff(ov_x<DIR>(v, o + t) ^ ov_y<DIR>(v, t));
        if (v.hasN) dd |= ov_xn<DIR>(v, o + t) ^ ov_yn<DIR>(v, t);
        cnt_full += popc32(dd & lowmask32(2 * (ol - t)));
        if (t < pre) {
            cnt_pre += popc32(dd & lowmask32(2 * (pre - t)));
            if (cnt_pre > limit) return -1;
        }
    }
    return ol > 50 ? cnt_full : cnt_pre;
}

// key of (direction, offset, diff) in the reference's scan order
FQ_DEV u32 ov_key(int dir, int o, int diff) {
    return ((u32)dir << (OV_KEY_OFF_BITS + OV_KEY_DIFF_BITS)) | ((u32)o << OV_KEY_DIFF_BITS) | (u32)diff;
}
// one offset that passed the prefilter: exact test, winner by atomic-min
template <int DIR>
FQ_DEV void overlap_check(const LdsLayout& L, u32* lds, const PairView& v, int pr, int o, bool exact) {
    const int lenX = DIR ? v.l2 : v.l1, lenY = DIR ? v.l1 : v.l2;
    const int diff = ov_verify<DIR>(v, o, lenX, lenY, exact ? nullptr : (const short*)(lds + L.lut_ov));
    if (diff >= 0) lds_min_u32((u32*)&lds_i(lds, L.ov_off)[pr], ov_key(DIR, o, diff));
}

// Pass 1, task = (pair, direction, quarter): the prefilter over blocks of 16 offsets.  Per offset: v_alignbit
// (X shifted), xor, shift, and-or (2-bit groups that differ), v_bcnt with -(lmax+1) as the addend (negative <=>
// at most lmax mismatches), v_alignbit to shift that sign into the block's candidate mask - six instructions.
// Survivors go to the tile's candidate list; pass 2 verifies them with every lane busy instead of one lane per
// wavefront verifying while 63 wait.
template <int DIR>
FQ_DEV void overlap_scan(const KernelArgs& a, u32* lds, int pr, int part, bool exact) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    const int R1 = pr, R2 = L.P + pr;
    if ((lds_i(lds, L.flags)[R1] | lds_i(lds, L.flags)[R2]) & RS_NULL) return;  // r1 != NULL && r2 != NULL
    PairView v;
    pair_view(L, lds, pr, v);
    const int lenX = DIR ? v.l2 : v.l1, lenY = DIR ? v.l1 : v.l2;
    const int nvalid = lenX - p.overlap_require;  // offsets 0 .. nvalid-1 (:48, :73)
    if (nvalid <= 0) return;
    // every legal offset compares at least min(require+1, lenY) bases; prefilter on <= 16 of them
    const int npre = imin(16, imin(p.overlap_require + 1, lenY));
    const u32 premask = lowmask32(2 * npre) & 0x55555555u;
    const u32 y0 = ov_y<DIR>(v, 0);
    const u32 nlim = (u32)(-((exact ? 0 : p.ov_limit_max) + 1));
    const u32* xrow = DIR ? v.rc : v.s1;
    const int xoff = DIR ? v.z2 : v.f1;
    u32* cl = lds + L.cand;
    for (int b = part; 16 * b < nvalid; b += 4) {
        const int o0 = 16 * b;
        const int bp = xoff + o0;
        const u32* xw = xrow + (bp >> 4);
        const u32 sh = (u32)(bp & 15) * 2u;
        const u32 d0 = xw[0], d1 = xw[1], d2 = xw[2];
        const u32 w0 = alignbit(d1, d0, sh), w1 = alignbit(d2, d1, sh);
        u32 cand = 0;  // bit (15 - t) <=> offset o0 + t survives the prefilter
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const u32 x = t ? alignbit(w1, w0, 2 * t) : w0;
            const u32 d = x ^ y0;
            const u32 sd = (u32)popc32((d | (d >> 1)) & premask) + nlim;  // negative <=> count <= lmax
            cand = alignbit(cand, sd, 31);
        }
        cand &= 0xFFFFu;
        if (nvalid - o0 < 16) cand &= ~lowmask32(16 - (nvalid - o0));
        while (cand) {
            const int t = clz32(cand) - 16;  // smallest surviving offset first
            cand &= ~(0x8000u >> t);
            const u32 slot = lds_add_ret_u32(cl, 1u);
            if (slot < (u32)L.cand_cap) cl[1 + slot] = ((u32)pr << 11) | ((u32)DIR << 10) | (u32)(o0 + t);
            else overlap_check<DIR>(L, lds, v, pr, o0 + t, exact);  // list full (low-complexity reads): verify here
        }
    }
}

// ---- the one-gap pass (overlapanalysis.cpp:91-139), only for pairs the no-gap scan left
// without a result.  Matcher::diffWithOneInsertion (matcher.cpp:56-100) in closed form: with
//   D0[k] = ins[k] != nor[k],  D1[k] = ins[k+1] != nor[k],  P0/P1 their prefix counts, c = cmplen:
//   result = -1                              if P0(c-1) + D1[c-1] > limit   (the final loop's exit,
//                                            taken even when an earlier split was good)
//          = min_{1<=i<c} P0(i) - P1(i) + P1(c)   otherwise; the capped entries of
//            accMismatchFromRight are all > limit and cannot be that minimum when it is accepted.
// ins_is_x: ins = X shifted by o, nor = Y;  else ins = Y, nor = X shifted by o.
template <int DIR>
FQ_DEV int ov_gap_diff(const PairView& v, int o, int c, int limit, bool ins_is_x) {
    if (c <= 1) return 100000000;  // the loops never run (matcher.cpp:88)
    int p0 = 0, p1 = 0, p0_cm1 = 0, d1_last = 0;
    for (int t = 0; t < c; t += 16) {
        const u32 x = ov_x<DIR>(v, o + t), y = ov_y<DIR>(v, t);
        u32 d0 = fold_diff(x ^ y);
        u32 d1 = ins_is_x ? fold_diff(ov_x<DIR>(v, o + t + 1) ^ y) : fold_diff(ov_y<DIR>(v, t + 1) ^ x);
        if (v.hasN) {
            const u32 xn = ov_xn<DIR>(v, o + t), yn = ov_yn<DIR>(v, t);
            d0 |= xn ^ yn;
            d1 |= ins_is_x ? (ov_xn<DIR>(v, o + t + 1) ^ yn) : (ov_yn<DIR>(v, t + 1) ^ xn);
        }
        p0 += popc32(d0 & lowmask32(2 * (c - t)));
        p0_cm1 += popc32(d0 & lowmask32(2 * imax(0, c - 1 - t)));
        p1 += popc32(d1 & lowmask32(2 * (c - t)));
        if (c - 1 >= t && c - 1 < t + 16) d1_last = (int)((d1 >> (2 * (c - 1 - t))) & 1u);
    }
    if (p0_cm1 + d1_last > limit) return -1;
    // min over i in [1, c) of P0(i) - P1(i)
    int g = 0, gmin = 0x7FFFFFFF;
    for (int t = 0; t < c - 1; t += 16) {
        const u32 x = ov_x<DIR>(v, o + t), y = ov_y<DIR>(v, t);
        u32 d0 = fold_diff(x ^ y);
        u32 d1 = ins_is_x ? fold_diff(ov_x<DIR>(v, o + t + 1) ^ y) : fold_diff(ov_y<DIR>(v, t + 1) ^ x);
        if (v.hasN) {
            const u32 xn = ov_xn<DIR>(v, o + t), yn = ov_yn<DIR>(v, t);
            d0 |= xn ^ yn;
            d1 |= ins_is_x ? (ov_xn<DIR>(v, o + t + 1) ^ yn) : (ov_yn<DIR>(v, t + 1) ^ xn);
        }
        const int nb = imin(16, c - 1 - t);  // positions k = t .. t+nb-1 give i = k+1 in [1, c)
        for (int k = 0; k < nb; k++) {
            g += (int)((d0 >> (2 * k)) & 1u) - (int)((d1 >> (2 * k)) & 1u);
            gmin = imin(gmin, g);
        }
    }
    return gmin + p1;
    (void)p0;
}

template <int DIR>
FQ_DEV void overlap_gap_task(const KernelArgs& a, u32* lds, int pr, int part) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    const int R1 = pr, R2 = L.P + pr;
    if ((lds_i(lds, L.flags)[R1] | lds_i(lds, L.flags)[R2]) & RS_NULL) return;
    if ((u32)lds_i(lds, L.ov_off)[pr] != OV_KEY_NONE) return;  // the no-gap passes return first (:48-89)
    PairView v;
    pair_view(L, lds, pr, v);
    const int lenX = DIR ? v.l2 : v.l1, lenY = DIR ? v.l1 : v.l2;
    const int nvalid = lenX - p.overlap_require;
    if (nvalid <= 0) return;
    const short* lut = (const short*)(lds + L.lut_ov);
    // acceptance needs P0(c-1) <= limit: the ungapped mismatches of the first ol-2 positions
    const int npre = imax(0, imin(16, imin(p.overlap_require - 1, lenY - 2)));
    const u32 premask = lowmask32(2 * npre) & 0x55555555u;
    const u32 y0 = ov_y<DIR>(v, 0);
    const int lmax = p.ov_limit_max;
    u32* keyp = (u32*)&lds_i(lds, L.ov_len)[pr];
    for (int b = part; 16 * b < nvalid; b += 4) {
        const int o0 = 16 * b;
        if (*(volatile u32*)keyp < ((u32)DIR << (OV_KEY_OFF_BITS + OV_KEY_DIFF_BITS) | ((u32)o0 << OV_KEY_DIFF_BITS))) break;
        const u32 w0 = ov_x<DIR>(v, o0), w1 = ov_x<DIR>(v, o0 + 16);
        u32 cand = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const u32 x = t ? alignbit(w1, w0, 2 * t) : w0;
            const u32 d = x ^ y0;
            const int cnt = popc32((d | (d >> 1)) & premask);
            cand = cand + cand + (u32)(cnt <= lmax);
        }
        if (nvalid - o0 < 16) cand &= ~lowmask32(16 - (nvalid - o0));
        while (cand) {
            const int t = clz32(cand) - 16;
            cand &= ~(0x8000u >> t);
            const int o = o0 + t;
            const int ol = imin(lenX - o, lenY);
            const int limit = lut[ol];
            // forward: diff(str1+o, str2) then diff(str2, str1+o); reverse: diff(str1, str2+o) then diff(str2+o, str1)
            int d = ov_gap_diff<DIR>(v, o, ol - 1, limit, DIR == 0);
            if (d < 0 || d > limit) d = ov_gap_diff<DIR>(v, o, ol - 1, limit, DIR != 0);
            if (d <= limit && d >= 0) {
                lds_min_u32(keyp, ((u32)DIR << (OV_KEY_OFF_BITS + OV_KEY_DIFF_BITS)) | ((u32)o << OV_KEY_DIFF_BITS) | (u32)d);
                return;
            }
        }
    }
}

FQ_DEV void phase_overlap_gap(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    if (!p.paired || !p.allow_gap || !p.need_overlap) return;  // peprocessor.cpp:443-447
    const int half = 4 * L.P;
    for (int t = tid; t < 2 * half; t += nthreads) {
        const int dir = t >= half ? 1 : 0;
        const int u = t - dir * half;
        if (dir) overlap_gap_task<1>(a, lds, u >> 2, u & 3);
        else overlap_gap_task<0>(a, lds, u >> 2, u & 3);
    }
}

// rc(whole read 2) and its N mask into the rc rows: lane = (pair, 16-base word).  Runs on the reads as loaded;
// BaseCorrector's edits of read 2 are mirrored into the rows by store_base.
FQ_DEV void phase_rc(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    if (!a.p.paired) return;
    const int SW = L.SW, P = L.P;
    const int total = P * SW;
    const u32 magic = a.magic_sw;
    for (int i = tid; i < total; i += nthreads) {
        const int pr = (int)fastdiv((u32)i, magic), w = i - pr * SW;
        const int R2 = P + pr;
        const int l2 = lds_i(lds, L.rlen0)[R2];
        // rc[16w + k] = comp(r2[l2 - 1 - 16w - k]): the 16 bases that END at l2 - 16w, reversed
        const int bp = l2 - 16 - 16 * w;
        u32 x = reverse_groups(window16_signed(lds_seq(L, lds, R2), bp)) ^ 0x55555555u;
        u32 n = 0;
        if (lds_i(lds, L.flags)[R2] & RS_HAS_N) {
            n = reverse_groups(window16_signed(lds_nmk(L, lds, R2), bp));
            x &= ~(n | (n << 1));   // N stays code 0 on both strands
        }
        lds[L.rc + i] = x;
        lds[L.rcn + i] = n;
    }
    if (tid == 0) lds[L.cand] = 0;
}

FQ_DEV void phase_overlap(const KernelArgs& a, u32* lds, int tid, int nthreads, bool exact = false) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    if (!p.paired) return;
    const bool thread0 = (a.batch_flags & 1u) != 0;  // FASTP_GPU_BATCH_STAT_ISIZE
    if (!(p.need_overlap || thread0 || p.merge || exact)) return;  // peprocessor.cpp:438
    // tasks [0, 4P): forward, [4P, 8P): reverse -> the direction is uniform per wavefront when 4P % 64 == 0
    const int half = 4 * L.P;
    for (int t = tid; t < 2 * half; t += nthreads) {
        const int dir = t >= half ? 1 : 0;
        const int u = t - dir * half;
        if (dir) overlap_scan<1>(a, lds, u >> 2, u & 3, exact);
        else overlap_scan<0>(a, lds, u >> 2, u & 3, exact);
    }
    tile_sync(a, lds, nthreads);
    // pass 2: lane = candidate
    const int nc = imin((int)lds[L.cand], L.cand_cap);
    for (int i = tid; i < nc; i += nthreads) {
        const u32 e = lds[L.cand + 1 + i];
        const int pr = (int)(e >> 11), o = (int)(e & 0x3FFu);
        PairView v;
        pair_view(L, lds, pr, v);
        if (e & 0x400u) overlap_check<1>(L, lds, v, pr, o, exact);
        else overlap_check<0>(L, lds, v, pr, o, exact);
    }
    tile_sync(a, lds, nthreads);
    if (tid == 0) lds[L.cand] = 0;   // ready for the next use (merge mode analyzes twice per tile)
}

// decode the packed scan key of a pair (lengths = the mates' lengths at analyze time)
FQ_DEV void decode_overlap(u32 key, int l1, int l2, int& ovl, int& off, int& ol, int& diff) {
    ovl = key != OV_KEY_NONE;
    off = ol = diff = 0;
    if (!ovl) return;
    const int dir = (int)(key >> (OV_KEY_OFF_BITS + OV_KEY_DIFF_BITS));
    const int o = (int)((key >> OV_KEY_DIFF_BITS) & ((1u << OV_KEY_OFF_BITS) - 1u));
    diff = (int)(key & ((1u << OV_KEY_DIFF_BITS) - 1u));
    off = dir ? -o : o;
    ol = dir ? imin(l1, l2 - o) : imin(l1 - o, l2);
}

// ---------------------------------------------------------------------------
// AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157) on the read [f, f+rlen)
// aw = packed adapter words (ACGT only, options.cpp:369-399), alen <= FASTP_GPU_MAX_ADAPTER_LEN (16 words).
// ---------------------------------------------------------------------------
FQ_DEV int count_mismatch_words(const u32* srow, const u32* nrow, bool hasN, int rp, const u32* aw, int ap,
                                int n, int allowed) {
    int mm = 0;
    for (int k = 0; k < n; k += 16) {
        u32 d = fold_diff(window16(srow, rp + k) ^ window16(aw, ap + k));
        if (hasN) d |= window16(nrow, rp + k);
        mm += popc32(d & lowmask32(2 * imin(16, n - k)));
        if (mm > allowed) break;
    }
    return mm;
}

// Matcher::matchWithOneInsertion (matcher.cpp:10-54) for every compare length at once.
// For fixed strings the function only depends on (cmplen, diffLimit):
//   matched <=> min_{1<=i<cmplen} (L[i-1] + R[i]) <= diffLimit,
//   L[j] = #mismatch ins[0..j] vs nor[0..j],  R[i] = #mismatch ins[i+1..cmplen] vs nor[i..cmplen-1]
// (the early exits of the reference never change that outcome: they fire exactly when
// no later i can succeed).  With P0/P1 the prefix counts of D0[k]=(ins[k]!=nor[k]) and
// D1[k]=(ins[k+1]!=nor[k]):  L[i-1]+R[i] = (P0(i)-P1(i)) + P1(cmplen), so one pass
// gives the answer for all cmplen.  Returns a bit mask: bit (c-1) set <=> matched for
// cmplen == c with diffLimit == c/8 - 1 (adaptertrimmer.cpp:108,125).
// ins_is_read: insertion case (ins = read, nor = adapter) else deletion case.
struct GapMask {   // one bit per compare length 1 .. 64 * MAX_ADAPTER_WORDS / 4; words picked by compares, not by address
    u64 w[MAX_ADAPTER_WORDS / 4];
    FQ_DEV void clear() {
#pragma unroll
        for (int i = 0; i < MAX_ADAPTER_WORDS / 4; i++) w[i] = 0;
    }
    FQ_DEV void set(int b) {
#pragma unroll
        for (int i = 0; i < MAX_ADAPTER_WORDS / 4; i++) w[i] |= (b >> 6) == i ? 1ull << (b & 63) : 0ull;
    }
    FQ_DEV bool test(int b) const {
        u64 x = 0;
#pragma unroll
        for (int i = 0; i < MAX_ADAPTER_WORDS / 4; i++) x |= (b >> 6) == i ? w[i] : 0ull;
        return ((x >> (b & 63)) & 1ull) != 0;
    }
    FQ_DEV bool any() const {
        u64 x = 0;
#pragma unroll
        for (int i = 0; i < MAX_ADAPTER_WORDS / 4; i++) x |= w[i];
        return x != 0;
    }
};
FQ_DEV GapMask one_gap_match_mask(const u32* srow, const u8* q, int f, int rlen, const u32* aw, int alen,
                                  bool ins_is_read, int cmax) {
    GapMask ok;
    ok.clear();
    int p0 = 0, p1 = 0;           // P0(i), P1(i)
    int gmin = 0x7FFFFFFF;        // min_{1<=i'<=i-1}... running min of P0(i)-P1(i) over i in [1, c-1]
    // walk i = 1..cmax ; at step i we know P0(i), P1(i) (prefix over k < i)
    for (int i = 1; i <= cmax; i++) {
        const int k = i - 1;
        u32 d0, d1;
        if (ins_is_read) {
            const u32 a_k = code_at(aw, k);
            d0 = sym_at(srow, q, f + k) != a_k;
            d1 = sym_at(srow, q, f + k + 1) != a_k;
        } else {
            const u32 r_k = sym_at(srow, q, f + k);
            d0 = code_at(aw, k) != r_k;
            d1 = code_at(aw, k + 1) != r_k;
        }
        p0 += (int)d0;
        p1 += (int)d1;
        // now p0 = P0(i), p1 = P1(i).  cmplen c = i: uses min over i' in [1, c-1] (previous gmin) and P1(c)=p1
        const int c = i;
        const int limit = c / 8 - 1;
        if (c >= 2 && gmin != 0x7FFFFFFF && gmin + p1 <= limit) ok.set(c - 1);
        // extend the running min with i' = i (valid for cmplen > i)
        const int g = p0 - p1;
        if (g < gmin) gmin = g;
    }
    (void)rlen;
    (void)alen;
    return ok;
}

FQ_DEV bool trim_by_sequence(const u32* srow, const u32* nrow, const u8* q, bool hasN, int f, int rlen,
                             const u32* aw, int alen, int matchReq, int& out_pos) {
    if (alen < matchReq) return false;
    int start = 0;
    if (alen >= 16) start = -4;
    else if (alen >= 12) start = -3;
    else if (alen >= 8) start = -2;
    int pos;
    for (pos = start; pos < rlen - matchReq; pos++) {  // :87-100
        const int cmplen = imin(rlen - pos, alen);
        const int allowed = cmplen / 8;
        const int so = imax(0, -pos);
        const int mm = count_mismatch_words(srow, nrow, hasN, f + so + pos, aw, so, cmplen - so, allowed);
        if (mm <= allowed) { out_pos = pos; return true; }
    }
    // one insertion in the read (:105-118) - rdata/adata are NOT advanced by pos (quirk #7)
    if (rlen - matchReq - 1 > 0) {
        const int cmax = imin(rlen - 1, alen);
        const GapMask ok = one_gap_match_mask(srow, q, f, rlen, aw, alen, true, cmax);
        if (ok.any()) {
            for (pos = 0; pos < rlen - matchReq - 1; pos++) {
                const int c = imin(rlen - pos - 1, alen);
                if (c >= 1 && ok.test(c - 1)) { out_pos = pos; return true; }
            }
        }
    }
    // one deletion in the read (:122-135)
    if (rlen - matchReq > 0) {
        const int cmax = imin(rlen, alen - 1);
        const GapMask ok = one_gap_match_mask(srow, q, f, rlen, aw, alen, false, cmax);
        if (ok.any()) {
            for (pos = 0; pos < rlen - matchReq; pos++) {
                const int c = imin(rlen - pos, alen - 1);
                if (c >= 1 && ok.test(c - 1)) { out_pos = pos; return true; }
            }
        }
    }
    return false;
}

// apply trimBySequence to read R (state in LDS); returns trimmed?
FQ_DEV bool apply_trim_by_sequence(const KernelArgs& a, u32* lds, int R, const u32* aw, int alen, u32* misc) {
    const LdsLayout& L = a.L;
    const int f = lds_i(lds, L.front)[R];
    const int rlen = lds_i(lds, L.len)[R];
    const bool hasN = (lds_i(lds, L.flags)[R] & RS_HAS_N) != 0;
    int pos = 0;
    if (!trim_by_sequence(lds_seq(L, lds, R), lds_nmk(L, lds, R), (const u8*)lds_qual(L, lds, R), hasN, f, rlen, aw,
                          alen, 4, pos))
        return false;
    int adapter_len;
    if (pos < 0) {  // adaptertrimmer.cpp:138-145
        adapter_len = alen + pos;
        lds_i(lds, L.len)[R] = 0;
    } else {
        adapter_len = rlen - pos;
        lds_i(lds, L.len)[R] = pos;
    }
    if (adapter_len > 0) lds_add_u32(&misc[MISC_ADAPTER_BASES], (u32)adapter_len);  // filterresult.cpp:127
    lds_i(lds, L.apos)[R] = pos;
    lds_i(lds, L.alen)[R] = adapter_len;
    return true;
}

// AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62) on read R: every --adapter_fasta
// sequence in turn on the (shrinking) read; each cut is reported as a fastp_gpu_adapter_event
FQ_DEV bool apply_fasta_trims(const KernelArgs& a, u32* lds, int R, u32 read_index, u32* misc) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    const int f = lds_i(lds, L.front)[R];
    bool trimmed = false;
    for (int i = 0; i < p.n_fasta; i++) {
        const int rlen = lds_i(lds, L.len)[R];
        const int alen = a.lut.fasta_len[i];
        const bool hasN = (lds_i(lds, L.flags)[R] & RS_HAS_N) != 0;
        int pos = 0;
        if (!trim_by_sequence(lds_seq(L, lds, R), lds_nmk(L, lds, R), (const u8*)lds_qual(L, lds, R), hasN, f, rlen,
                              a.lut.fasta_words + (size_t)i * ADAPT_WORDS, alen, p.fasta_match_req, pos))
            continue;
        trimmed = true;
        int adapter_len;
        if (pos < 0) {  // adaptertrimmer.cpp:138-145
            adapter_len = alen + pos;
            lds_i(lds, L.len)[R] = 0;
        } else {
            adapter_len = rlen - pos;
            lds_i(lds, L.len)[R] = pos;
        }
        if (adapter_len > 0) lds_add_u32(&misc[MISC_ADAPTER_BASES], (u32)adapter_len);  // filterresult.cpp:127
        if (a.adapter_events) {
            const int slot = g_atomic_add_i32(a.n_adapter_events, 1);
            if (slot < a.adapter_events_capacity) {
                a.adapter_events[3 * slot] = read_index;
                a.adapter_events[3 * slot + 1] = ((u32)pos & 0xFFFFu) | (((u32)adapter_len & 0xFFFFu) << 16);
                a.adapter_events[3 * slot + 2] = (u32)i;
            }
        }
    }
    return trimmed;
}

// ---------------------------------------------------------------------------
// fastp_simd::countQualityMetrics (simd.cpp:54-119) and countAdjacentDiffs (:162-185) on the
// final window [front, front+len) of every read: 4 lanes per read, lane s takes the window's quality
// dwords s, s+4, ... with byte-parallel arithmetic, two DPP steps fold the partial sums.
//   met[R][0] = total(qual-33) | lowQualNum << 16      met[R][1] = nBaseNum | adjacentDiffs << 16
// ---------------------------------------------------------------------------
FQ_DEV void phase_metrics(const KernelArgs& a, u32* lds, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    // argument-block fields fetched once, not at every use inside the loops
    const int QW = L.QW, SW = L.SW;
    const bool cplx = p.complexity_filter != 0;
    const int* front_v = lds_i(lds, L.front);
    const int* wlen_v = lds_i(lds, p.merge ? L.mlen : L.len);
    const u32* qual_v = lds + L.qual;
    const u32* seq_v = lds + L.seq;
    u32* met_v = lds + L.met;
    const int total = L.NR * 4;   // 4 lanes per read, quality dwords interleaved
    const int lane = tid & 63;
    const u32 thr4 = (u32)p.qual_thr * 0x01010101u;
    for (int t0 = tid - lane; t0 < total; t0 += nthreads) {  // wave-uniform trip count (lane exchanges inside)
        const int t = t0 + lane;
        const bool valid = t < total;
        const int R = valid ? (t >> 2) : 0, seg = t & 3;
        const int f = front_v[R];
        const int e = valid ? f + wlen_v[R] : f;
        const u32* qrow = qual_v + rowoff(R, QW);
        const u32* srow = seq_v + rowoff(R, SW);
        u32 ma = 0, mb = 0;
        for (int c = (f >> 2) + seg; 4 * c < e; c += 4) {  // only the dwords that touch the window
            const int j0 = 4 * c;
            const u32 qd = qrow[c];
            const u32 q7 = qd & 0x7F7F7F7Fu;
            const u32 ge = ((q7 | 0x80808080u) - thr4) & 0x80808080u;  // bit 7 of a byte: qual >= threshold
            int lo = 0, hi = 4;
            if (j0 >= f && j0 + 4 <= e) {  // the whole dword is inside the window (the common case)
                ma += sum_bytes(q7, 0u) - 132u + ((u32)popc32(ge ^ 0x80808080u) << 16);
                mb += (u32)popc32(qd & 0x80808080u);
            } else {
                lo = imax(f - j0, 0);
                hi = imin(e - j0, 4);  // bytes [lo, hi) of this dword are in the window
                const u32 M = lowmask32(8 * hi) & ~lowmask32(8 * lo);
                ma += sum_bytes(q7 & M, 0u) - 33u * (u32)(hi - lo);
                ma += (u32)popc32(~ge & 0x80808080u & M) << 16;
                mb += (u32)popc32(qd & 0x80808080u & M);
            }
            if (cplx) {
                // symbol j differs from symbol j-1 (N is its own symbol; its stored code is 0)
                const u32 cur8 = (srow[c >> 2] >> ((c & 3) * 8)) & 0xFFu;
                const u32 nb = (qd >> 7) & 0x01010101u;
                const u32 ncur = (nb | (nb >> 7) | (nb >> 14) | (nb >> 21)) & 0xFu;
                u32 pcode = 0, pn = 0;
                if (c > 0) {
                    pcode = (srow[(c - 1) >> 2] >> (((c - 1) & 3) * 8 + 6)) & 3u;
                    pn = (qrow[c - 1] >> 31) & 1u;
                }
                const u32 ecodes = pcode | (cur8 << 2);
                const u32 dc = ecodes ^ (ecodes >> 2);           // group k: code(j0+k-1) ^ code(j0+k)
                const u32 df = (dc | (dc >> 1)) & 0x55u;
                const u32 d4 = (df & 1u) | ((df >> 1) & 2u) | ((df >> 2) & 4u) | ((df >> 3) & 8u);
                const u32 en = pn | (ncur << 1);
                const u32 dn = (en ^ (en >> 1)) & 0xFu;          // bit k: N(j0+k-1) ^ N(j0+k)
                const int lo1 = imax(f + 1 - j0, 0);             // pairs (j-1, j) with j in [f+1, e)
                const u32 M4 = hi > lo1 ? (lowmask32(hi) & ~lowmask32(lo1)) : 0u;
                mb += (u32)popc32((d4 | dn) & M4) << 16;
            }
        }
        ma = sum4(ma);
        mb = sum4(mb);
        if (valid && seg == 0) {
            met_v[2 * R] = ma;
            met_v[2 * R + 1] = mb;
        }
    }
}

// Filter::passFilter (filter.cpp:15-57) on the read [f, f+rlen) from its metrics; alive == non-NULL
FQ_DEV int filter_code(const KernelArgs& a, u32* lds, int rlen, int tot, int low, int nb, int diff) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    if (rlen == 0) return 16;  // FAIL_LENGTH (:16-18)
    if (p.qual_filter) {  // :35-42
        const u16* lowq = (const u16*)(lds + L.lut_lowq);
        if (low > (int)lowq[rlen]) return 20;                                   // FAIL_QUALITY
        else if (p.avg_qual_req > 0 && (tot / rlen) < p.avg_qual_req) return 20;  // FAIL_QUALITY
        else if (nb > p.n_base_limit) return 12;                                // FAIL_N_BASE
    }
    if (p.length_filter) {  // :44-49
        if (rlen < p.length_required) return 16;
        if (p.length_limit > 0 && rlen > p.length_limit) return 17;
    }
    if (p.complexity_filter) {  // :51-54, 59-66
        if (rlen <= 1) return 24;
        const u16* cmin = (const u16*)(lds + L.lut_cplx);
        if (diff < (int)cmin[rlen]) return 24;  // FAIL_COMPLEXITY
    }
    return 0;
}
FQ_DEV int pass_filter(const KernelArgs& a, u32* lds, int R, bool alive) {
    const LdsLayout& L = a.L;
    if (!alive) return 16;  // r == NULL (:16-18)
    const u32 ma = lds[L.met + 2 * R], mb = lds[L.met + 2 * R + 1];
    return filter_code(a, lds, lds_i(lds, L.len)[R], (int)(ma & 0xFFFFu), (int)(ma >> 16), (int)(mb & 0xFFFFu), (int)(mb >> 16));
}

FQ_DEV void store_base(const KernelArgs& a, u32* lds, int R, int j, u32 sym, u32 qchar) {
    const LdsLayout& L = a.L;
    u32* srow = lds_seq(L, lds, R);
    u32* nrow = lds_nmk(L, lds, R);
    u8* q = (u8*)lds_qual(L, lds, R);
    const int w = j >> 4, sh = (j & 15) * 2;
    const u32 code = sym == 4u ? 0u : sym;
    srow[w] = (srow[w] & ~(3u << sh)) | (code << sh);
    nrow[w] = (nrow[w] & ~(1u << sh)) | ((sym == 4u ? 1u : 0u) << sh);
    q[j] = (u8)(qchar | (sym == 4u ? 0x80u : 0u));
    if (sym == 4u) lds_or_i32(&lds_i(lds, L.flags)[R], RS_HAS_N);
    if (R >= L.P && L.rc >= 0) {  // read 2: the same base in the rc rows (position rlen0 - 1 - j, complemented; N -> 0)
        const int k = lds_i(lds, L.rlen0)[R] - 1 - j;
        if (k >= 0) {
            u32* rc = lds + L.rc + rowoff(R - L.P, L.SW);
            u32* rcn = lds + L.rcn + rowoff(R - L.P, L.SW);
            const int w2 = k >> 4, sh2 = (k & 15) * 2;
            const u32 c2 = sym == 4u ? 0u : (sym ^ 1u);
            rc[w2] = (rc[w2] & ~(3u << sh2)) | (c2 << sh2);
            rcn[w2] = (rcn[w2] & ~(1u << sh2)) | ((sym == 4u ? 1u : 0u) << sh2);
        }
    }
}

FQ_DEV void write_read_result(const KernelArgs& a, u32* lds, int m, int R, int gp) {
    const LdsLayout& L = a.L;
    // fastp_gpu_read_result: u16 front, u16 len | u8 code, u8 flags, i16 adapter_pos | u16 adapter_len, u16 reserved
    const u32 front = (u32)lds_i(lds, L.front)[R] & 0xFFFFu;
    const u32 len = (u32)lds_i(lds, L.len)[R] & 0xFFFFu;
    const u32 code = (u32)lds_i(lds, L.code)[R] & 0xFFu;
    const u32 flags = (u32)lds_i(lds, L.flags)[R] & 0xFFu;
    const u32 apos = (u32)lds_i(lds, L.apos)[R] & 0xFFFFu;
    const u32 alen = (u32)lds_i(lds, L.alen)[R] & 0xFFFFu;
    // reserved: merge mode, overlapped pair: bases of this mate in the merged read (the name tag merged_L1_L2)
    const int pr1 = m ? R - L.P : R;
    // --overlapped_out: read 1 carries 0x8000 | first printed position, read 2 the number of printed bases
    // (with merge AS WELL the fields carry the --overlapped_out values: the merged part lengths follow from the pair
    // record, len1 = ov_len + max(0, ov_offset), len2 = ov_offset > 0 ? len(read 2) - ov_len : 0)
    const u32 rsv = a.p.overlapped_out ? ((u32)lds_i(lds, L.olen)[R] & 0xFFFFu)
                    : (lds_i(lds, L.flags)[pr1] & RS_MERGE_OV) ? ((u32)lds_i(lds, L.mlen)[R] & 0xFFFFu) : 0u;
    u32* out = a.res[m] + (size_t)gp * 3;
    out[0] = front | (len << 16);
    out[1] = code | (flags << 8) | (apos << 16);
    out[2] = alen | (rsv << 16);
}

// ASCII & 7 of a symbol (A=1 T=4 C=3 G=7 N=6): FilterResult::addCorrection filterresult.cpp:99-103
FQ_DEV u32 sym_bin(u32 s) { return (0x67341u >> (s * 4)) & 0xFu; }
FQ_DEV u32 sym_ascii(u32 s) { return (u32)("ATCGN"[s]); }

// fastp_gpu_pair_result: i16 ov_offset, u16 ov_len | u16 ov_diff, u16 flags
FQ_DEV void write_pair_result(const KernelArgs& a, int gp, int ovl, int off, int ol, int diff, bool isize_done) {
    a.pair[2 * (size_t)gp] = ((u32)off & 0xFFFFu) | (((u32)ol & 0xFFFFu) << 16);
    a.pair[2 * (size_t)gp + 1] = ((u32)diff & 0xFFFFu) | ((u32)((ovl ? 1 : 0) | (isize_done ? 4 : 0)) << 16);
}

// ---------------------------------------------------------------------------
// Phase E (paired): lane = one pair.  peprocessor.cpp:443-573.
// ---------------------------------------------------------------------------
FQ_DEV void phase_decide_pe(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    u32* misc = lds + L.acc_misc;
    const bool thread0 = (a.batch_flags & 1u) != 0;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        const int gp = tile_first + pr;
        if (gp >= a.n) continue;
        const int R1 = pr, R2 = L.P + pr;
        int* flags = lds_i(lds, L.flags);
        int* lenv = lds_i(lds, L.len);
        const bool a1 = !(flags[R1] & RS_NULL), a2 = !(flags[R2] & RS_NULL);
        const bool both = a1 && a2;
        const int ft1 = lds_i(lds, L.ft)[R1], ft2 = lds_i(lds, L.ft)[R2];
        // the two lengths as they are now, kept in registers (a re-read behind every store is an LDS round trip, and two
        // wavefronts run this phase while fourteen wait); reloaded after the helpers that edit them in LDS
        int cur1 = lenv[R1], cur2 = lenv[R2];
        int ovl, ov_off, ov_len, ov_diff;  // the OverlapResult of the pair as it is now (quirk #6)
        decode_overlap((u32)lds_i(lds, L.ov_off)[pr], cur1, cur2, ovl, ov_off, ov_len, ov_diff);
        // ovForAdapter (peprocessor.cpp:445-447): with allow_gap, the one-gap result where no-gap found nothing
        int aovl = ovl, aoff = ov_off, aol = ov_len, adiff = ov_diff;
        bool agap = false;
        if (p.allow_gap && !ovl) {
            decode_overlap((u32)lds_i(lds, L.ov_len)[pr], cur1, cur2, aovl, aoff, aol, adiff);
            agap = aovl != 0;
        }
        bool isize_done = false, dimer = false;
        // statInsertSize (peprocessor.cpp:710-723), thread 0 only (:449, :497)
        if (both && thread0) {
            int isize = p.isize_max;
            if (ovl) {
                if (ov_off > 0) isize = cur1 + cur2 - ov_len + ft1 + ft2;
                else isize = ov_len + ft1 + ft2;
            }
            if (isize > p.isize_max) isize = p.isize_max;
            if (isize >= 0) lds_add_u32(&misc[MISC_ISIZE + isize], 1u);
            isize_done = true;
        }
        if (both && p.need_overlap) {
            // BaseCorrector::correctByOverlapAnalysis (basecorrector.cpp:16-83)
            if (p.correction && aovl && !agap && adiff != 0) {
                const int f1 = lds_i(lds, L.front)[R1], f2 = lds_i(lds, L.front)[R2];
                const int start1 = imax(0, aoff);
                const int start2 = cur2 - imax(0, -aoff) - 1;
                const u32* s1 = lds_seq(L, lds, R1);
                const u32* s2 = lds_seq(L, lds, R2);
                const u8* q1 = (const u8*)lds_qual(L, lds, R1);
                const u8* q2 = (const u8*)lds_qual(L, lds, R2);
                int corrected = 0;
                bool r1c = false, r2c = false;
                for (int i = 0; i < aol; i++) {
                    const int j1 = f1 + start1 + i, j2 = f2 + start2 - i;
                    const u32 b1 = sym_at(s1, q1, j1), b2 = sym_at(s2, q2, j2);
                    if (b1 != sym_complement(b2)) {
                        const u32 c1 = qchar_at(q1, j1), c2 = qchar_at(q2, j2);
                        int which = -1, cpos = 0;
                        u32 nb = 0, nq = 0, from = 0;
                        if (c1 >= 63u && c2 <= 47u) {  // GOOD_QUAL = Q30, BAD_QUAL = Q14 (:32-33)
                            from = b2; nb = sym_complement(b1); nq = c1;
                            store_base(a, lds, R2, j2, nb, nq);
                            which = 1; cpos = j2; r2c = true;
                        } else if (c2 >= 63u && c1 <= 47u) {
                            from = b1; nb = sym_complement(b2); nq = c2;
                            store_base(a, lds, R1, j1, nb, nq);
                            which = 0; cpos = j1; r1c = true;
                        }
                        if (which >= 0) {
                            corrected++;
                            lds_add_u32(&misc[MISC_CORRECTION + sym_bin(from) * 8 + sym_bin(nb)], 1u);
                            if (a.corrections) {
                                const int slot = g_atomic_add_i32(a.n_corrections, 1);
                                if (slot < a.corr_capacity) {
                                    a.corrections[2 * slot] = (u32)(2 * (a.first + gp) + which);
                                    a.corrections[2 * slot + 1] = (u32)cpos | (sym_ascii(nb) << 16) | (nq << 24);
                                }
                            }
                            if (a.corr_int) {   // the engine's own list (-c with the Stats kernel as its own launch: fq_corr_stats_kernel reads it)
                                const int slot = g_atomic_add_i32(a.n_corr_int, 1);
                                if (slot < a.corr_int_cap) {
                                    a.corr_int[2 * slot] = (u32)(2 * (a.first + gp) + which);
                                    a.corr_int[2 * slot + 1] = (u32)cpos | (sym_ascii(nb) << 16) | (nq << 24);
                                }
                            }
                        }
                    }
                }
                if (corrected > 0) {  // :75-80
                    lds_add_u32(&misc[MISC_CORRECTED_READS], (r1c && r2c) ? 2u : 1u);
                    if (r1c) lds_or_i32(&flags[R1], RS_CORRECTED);
                    if (r2c) lds_or_i32(&flags[R2], RS_CORRECTED);
                }
            }
            if (p.adapter_enabled) {
                bool trimmed = false;
                if (aovl && aoff < 0) {  // trimByOverlapAnalysis adaptertrimmer.cpp:17-46
                    const int len1 = imin(cur1, aol + ft2);
                    const int len2 = imin(cur2, aol + ft1);
                    lds_i(lds, L.apos)[R1] = len1;
                    lds_i(lds, L.alen)[R1] = cur1 - len1;
                    lds_i(lds, L.apos)[R2] = len2;
                    lds_i(lds, L.alen)[R2] = cur2 - len2;
                    lds_add_u32(&misc[MISC_ADAPTER_BASES], (u32)((cur1 - len1) + (cur2 - len2)));
                    lenv[R1] = cur1 = len1;
                    lenv[R2] = cur2 = len2;
                    trimmed = true;
                    lds_or_i32(&flags[R1], RS_ADAPTER_OV);
                    lds_or_i32(&flags[R2], RS_ADAPTER_OV);
                }
                bool t1 = trimmed, t2 = trimmed;
                if (!trimmed) {  // peprocessor.cpp:460-466
                    if (p.has_a1) { t1 = apply_trim_by_sequence(a, lds, R1, lds + L.adapt, p.alen1, misc); cur1 = lenv[R1]; }
                    if (p.has_a2) { t2 = apply_trim_by_sequence(a, lds, R2, lds + L.adapt + ADAPT_WORDS, p.alen2, misc); cur2 = lenv[R2]; }
                }
                if (p.n_fasta) {  // :467-470
                    t1 |= apply_fasta_trims(a, lds, R1, 2u * (u32)(a.first + gp), misc);
                    t2 |= apply_fasta_trims(a, lds, R2, 2u * (u32)(a.first + gp) + 1u, misc);
                    cur1 = lenv[R1];
                    cur2 = lenv[R2];
                }
                if (t1) { lds_add_u32(&misc[MISC_ADAPTER_READS], 1u); lds_or_i32(&flags[R1], RS_ADAPTER); }  // :472-475
                if (t2) { lds_add_u32(&misc[MISC_ADAPTER_READS], 1u); lds_or_i32(&flags[R2], RS_ADAPTER); }
                if ((t1 || t2) && cur1 <= p.dimer_max_len && cur2 <= p.dimer_max_len) dimer = true;  // :480-484
            }
        }
        (void)isize_done;
        if (p.overlapped_out) {  // the reads as --overlapped_out's analysis sees them (:488, before polyX and max_len)
            lds_i(lds, L.olen)[R1] = cur1;
            lds_i(lds, L.olen)[R2] = cur2;
        }
        if (both && p.poly_x) {  // :506-509
            for (int k = 0; k < 2; k++) {
                const int R = k ? R2 : R1;
                int poly, trimmed;
                const int nl = trim_poly_x(lds_seq(L, lds, R), (const u8*)lds_qual(L, lds, R), lds_i(lds, L.front)[R],
                                           k ? cur2 : cur1, p.poly_x_min, poly, trimmed);
                if (poly >= 0) {  // addPolyXTrimmed filterresult.cpp:186-189
                    lds_add_u32(&misc[MISC_POLYX_READS + poly], 1u);
                    lds_add_u32(&misc[MISC_POLYX_BASES + poly], (u32)trimmed);
                    lds_or_i32(&flags[R], RS_POLYX);
                }
                lenv[R] = nl;
                if (k) cur2 = nl; else cur1 = nl;
            }
        }
        if (both) {  // :511-516
            if (p.max_len1 > 0 && p.max_len1 < cur1) lenv[R1] = cur1 = p.max_len1;
            if (p.max_len2 > 0 && p.max_len2 < cur2) lenv[R2] = cur2 = p.max_len2;
        }
        if (dimer | isize_done) lds_or_i32(&flags[R1], (dimer ? RS_DIMER : 0) | (isize_done ? RS_ISIZE : 0));
        lds_i(lds, L.mlen)[R1] = cur1;
        lds_i(lds, L.mlen)[R2] = cur2;
        if (p.merge && both) {
            // merge mode analyzes the post-trim reads again (peprocessor.cpp:523); phase_merge writes the record
            lds_i(lds, L.ov_off)[pr] = (int)OV_KEY_NONE;
        } else {
            write_pair_result(a, gp, ovl, ov_off, ov_len, ov_diff, isize_done);
        }
    }
}

// ---------------------------------------------------------------------------
// --overlapped_out (peprocessor.cpp:488-495): a third OverlapAnalysis::analyze with diffPercentLimit 0 on the reads
// as they are right after adapter trimming.  What the reference prints for an overlapped pair is
//   string(r1->mSeq->substr(max(0, offset)), overlap_len)   (:491)
// - std::string's (str, pos) constructor, i.e. the bases of read 1 BEHIND the overlapped region,
// r1'[max(0, offset) + overlap_len, len1').  phase_decide_pe left the post-adapter lengths in olen[] (an array of its
// own: mlen[] belongs to merge mode, which may be on as well and runs AFTER this, :518); the analysis runs on them,
// then the record's `reserved` fields take
//   olen[R1] = 0x8000 | pos (overlapped; pos = max(0, offset) + overlap_len) or 0, olen[R2] = len1' - pos.
// ---------------------------------------------------------------------------
FQ_DEV void phase_ovout_begin(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        if (tile_first + pr >= a.n) continue;
        for (int k = 0; k < 2; k++) {
            const int R = k ? L.P + pr : pr;
            const int fin = lds_i(lds, L.len)[R];
            lds_i(lds, L.len)[R] = lds_i(lds, L.olen)[R];
            lds_i(lds, L.olen)[R] = fin;
        }
        lds_i(lds, L.ov_off)[pr] = (int)OV_KEY_NONE;
    }
}
FQ_DEV void phase_ovout_end(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        if (tile_first + pr >= a.n) continue;
        const int R1 = pr, R2 = L.P + pr;
        int ovl, off, ol, diff;
        const int l1 = lds_i(lds, L.len)[R1];
        decode_overlap((u32)lds_i(lds, L.ov_off)[pr], l1, lds_i(lds, L.len)[R2], ovl, off, ol, diff);
        lds_i(lds, L.len)[R1] = lds_i(lds, L.olen)[R1];
        lds_i(lds, L.len)[R2] = lds_i(lds, L.olen)[R2];
        const int pos = imax(0, off) + ol;   // <= l1: string(substr(start), overlap_len) never throws
        lds_i(lds, L.olen)[R1] = ovl ? (0x8000 | pos) : 0;
        lds_i(lds, L.olen)[R2] = ovl ? l1 - pos : 0;
        if (a.p.merge) lds_i(lds, L.ov_off)[pr] = (int)OV_KEY_NONE;   // merge mode's analysis of the final reads comes next
    }
}

// ---------------------------------------------------------------------------
// Merge mode (peprocessor.cpp:518-527, OverlapAnalysis::merge overlapanalysis.cpp:148-179):
// lane = one pair, after the second overlap analysis.  merged = r1[0, len1) + rc(r2)[ol, ol+len2),
// i.e. the first len1 bases of r1' followed by the first len2 bases of r2' reversed and
// complemented; mlen[] records the two part lengths for the metrics / filter / stats steps.
// ---------------------------------------------------------------------------
FQ_DEV void phase_merge(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        const int gp = tile_first + pr;
        if (gp >= a.n) continue;
        const int R1 = pr, R2 = L.P + pr;
        int* flags = lds_i(lds, L.flags);
        if ((flags[R1] | flags[R2]) & RS_NULL) continue;
        const int l1 = lds_i(lds, L.len)[R1], l2 = lds_i(lds, L.len)[R2];
        int ovl, off, ol, diff;
        decode_overlap((u32)lds_i(lds, L.ov_off)[pr], l1, l2, ovl, off, ol, diff);
        if (ovl) {
            const int len1 = ol + imax(0, off);
            const int len2 = off > 0 ? l2 - ol : 0;
            lds_i(lds, L.mlen)[R1] = imin(len1, l1);          // substr(0, len1) clamps
            lds_i(lds, L.mlen)[R2] = imax(0, imin(len2, l2 - ol));
            flags[R1] |= RS_MERGE_OV;
        }
        write_pair_result(a, gp, ovl, off, ol, diff, (flags[R1] & RS_ISIZE) != 0);
    }
}

// Duplicate::checkPair / checkRead hash values (duplicate.cpp:122-148) of unit u (pair or single
// read) for the dup kernels: per-read base parts + the host-built position part
FQ_DEV void write_dup_pos(const KernelArgs& a, u32* lds, int u, int gp) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    if (!p.dup_enabled || !a.dup_pos) return;
    // the base-value part only: the length-dependent position part (DevLuts::dup_posum) is added
    // by the duplicate kernels, so that this thin phase does not wait for a global table read
    const u64* h1 = (const u64*)(lds + L.hash) + (size_t)u * p.dup_bufnum;
    for (int i = 0; i < p.dup_bufnum; i++) {
        u64 h = h1[i];
        if (p.paired) h += ((const u64*)(lds + L.hash) + (size_t)(L.P + u) * p.dup_bufnum)[i];
        a.dup_pos[(size_t)gp * p.dup_bufnum + i] = h;
    }
}

// split plan: mReads++ / mLengthSum += len of the four Stats objects (stats.cpp:194, 290) for one unit - the one-pass
// convention of phase_stats_both (every read counts in its PRE slot, a read that is written out also in its POST
// slot with its kept length) - and the unit's entry of the swin arrays the Stats kernel reads.  The addresses are
// uniform: the compiler folds each of these into one LDS atomic per wavefront.
// (kept1 / kept2: the lengths of the reads that are written out - what swin's upper half holds unless the option set has a
// front, DevParams::front_lane: then it holds the END of the kept range, front + length)
FQ_DEV void split_stat_reads(const KernelArgs& a, u32* misc, int gp, u32 sw1, u32 sw2, u32 kept1, u32 kept2) {
    lds_add_u32(&misc[MISC_STAT_READS + 0], 1u);
    lds_add_u32(&misc[MISC_STAT_LENSUM + 0], sw1 & 0xFFFFu);
    lds_add_u32(&misc[MISC_STAT_READS + 1], (sw1 >> 16) ? 1u : 0u);
    lds_add_u32(&misc[MISC_STAT_LENSUM + 1], kept1);
    a.swin_out[0][gp] = sw1;
    if (a.p.paired) {
        lds_add_u32(&misc[MISC_STAT_READS + 2], 1u);
        lds_add_u32(&misc[MISC_STAT_LENSUM + 2], sw2 & 0xFFFFu);
        lds_add_u32(&misc[MISC_STAT_READS + 3], (sw2 >> 16) ? 1u : 0u);
        lds_add_u32(&misc[MISC_STAT_LENSUM + 3], kept2);
        a.swin_out[1][gp] = sw2;
    }
}
FQ_DEV void split_stat_reads(const KernelArgs& a, u32* misc, int gp, u32 sw1, u32 sw2) { split_stat_reads(a, misc, gp, sw1, sw2, sw1 >> 16, sw2 >> 16); }

// Phase E3 (paired): lane = one pair.  Filter::passFilter and routing, peprocessor.cpp:563-591.
// passFilter (filter.cpp:15-66) with the two LUT entries of the read's length already in registers
FQ_DEV int filter_code_pre(const DevParams& p, int rlen, int tot, int low, int nb, int diff, int lowq_v, int cmin_v) {
    if (rlen == 0) return 16;
    if (p.qual_filter) {
        if (low > lowq_v) return 20;
        else if (p.avg_qual_req > 0 && (tot / rlen) < p.avg_qual_req) return 20;
        else if (nb > p.n_base_limit) return 12;
    }
    if (p.length_filter) {
        if (rlen < p.length_required) return 16;
        if (p.length_limit > 0 && rlen > p.length_limit) return 17;
    }
    if (p.complexity_filter) {
        if (rlen <= 1) return 24;
        if (diff < cmin_v) return 24;
    }
    return 0;
}

// phase_filter_pe without merge mode, written for latency: two wavefronts run this phase while fourteen wait, and a
// lane's LDS reads were a chain of ~12 round trips because the stores in between pin their order.  Here everything
// the pair needs is read first (independent loads: one round trip), then the two LUT rows, then it is all registers.
FQ_DEV void phase_filter_pe_plain(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    u32* misc = lds + L.acc_misc;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        const int gp = tile_first + pr;
        if (gp >= a.n) continue;
        const int R1 = pr, R2 = L.P + pr;
        int* flags = lds_i(lds, L.flags);
        int f1 = flags[R1], f2 = flags[R2];
        const u32 ma1 = lds[L.met + 2 * R1], mb1 = lds[L.met + 2 * R1 + 1];
        const u32 ma2 = lds[L.met + 2 * R2], mb2 = lds[L.met + 2 * R2 + 1];
        const int len1 = lds_i(lds, L.len)[R1], len2 = lds_i(lds, L.len)[R2];
        const u32 front1 = (u32)lds_i(lds, L.front)[R1], front2 = (u32)lds_i(lds, L.front)[R2];
        const u32 rl1 = (u32)lds_i(lds, L.rlen0)[R1], rl2 = (u32)lds_i(lds, L.rlen0)[R2];
        const u32 apos1 = (u32)lds_i(lds, L.apos)[R1], apos2 = (u32)lds_i(lds, L.apos)[R2];
        const u32 alen1 = (u32)lds_i(lds, L.alen)[R1], alen2 = (u32)lds_i(lds, L.alen)[R2];
        const u32 rsv1 = p.overlapped_out ? (u32)lds_i(lds, L.olen)[R1] & 0xFFFFu : 0u;   // phase_ovout_end
        const u32 rsv2 = p.overlapped_out ? (u32)lds_i(lds, L.olen)[R2] & 0xFFFFu : 0u;
        write_dup_pos(a, lds, pr, gp);   // LDS reads + global stores only
        const u16* lowq = (const u16*)(lds + L.lut_lowq);
        const u16* cmin = (const u16*)(lds + L.lut_cplx);
        const int lq1 = lowq[len1], lq2 = lowq[len2], cm1 = cmin[len1], cm2 = cmin[len2];
        const bool a1 = !(f1 & RS_NULL), a2 = !(f2 & RS_NULL);
        int code1 = a1 ? filter_code_pre(p, len1, (int)(ma1 & 0xFFFFu), (int)(ma1 >> 16), (int)(mb1 & 0xFFFFu), (int)(mb1 >> 16), lq1, cm1) : 16;
        int code2 = a2 ? filter_code_pre(p, len2, (int)(ma2 & 0xFFFFu), (int)(ma2 >> 16), (int)(mb2 & 0xFFFFu), (int)(mb2 >> 16), lq2, cm2) : 16;
        if (f1 & RS_DIMER) { code1 = 28; code2 = 28; }     // :568-571
        lds_add_u32(&misc[MISC_FILTER + imax(code1, code2)], 2u);  // addFilterResult(max, 2) :573
        const bool dedup_out = p.dedup && (f1 & RS_DUP);
        if (!dedup_out && a1 && a2 && code1 == 0 && code2 == 0) {   // written to out1 / out2 (:577-591)
            f1 |= RS_STAT_POST;
            f2 |= RS_STAT_POST;
            flags[R1] = f1;
            flags[R2] = f2;
        }
        lds_i(lds, L.code)[R1] = code1;
        lds_i(lds, L.code)[R2] = code2;
        const u32 sw1 = rl1 | ((f1 & RS_STAT_POST) ? (u32)len1 << 16 : 0u);
        const u32 sw2 = rl2 | ((f2 & RS_STAT_POST) ? (u32)len2 << 16 : 0u);
        lds[L.swin + R1] = sw1;
        lds[L.swin + R2] = sw2;
        if (a.split && p.front_lane)   // the Stats kernel's kept range ends at front + length (fq_stats.h)
            split_stat_reads(a, misc, gp, rl1 | ((f1 & RS_STAT_POST) ? (front1 + (u32)len1) << 16 : 0u), rl2 | ((f2 & RS_STAT_POST) ? (front2 + (u32)len2) << 16 : 0u),
                             (f1 & RS_STAT_POST) ? (u32)len1 : 0u, (f2 & RS_STAT_POST) ? (u32)len2 : 0u);
        else if (a.split) split_stat_reads(a, misc, gp, sw1, sw2);
        u32* o1 = a.res[0] + (size_t)gp * 3;
        u32* o2 = a.res[1] + (size_t)gp * 3;
        o1[0] = (front1 & 0xFFFFu) | ((u32)len1 << 16);
        o1[1] = ((u32)code1 & 0xFFu) | (((u32)f1 & 0xFFu) << 8) | (apos1 << 16);
        o1[2] = (alen1 & 0xFFFFu) | (rsv1 << 16);
        o2[0] = (front2 & 0xFFFFu) | ((u32)len2 << 16);
        o2[1] = ((u32)code2 & 0xFFu) | (((u32)f2 & 0xFFu) << 8) | (apos2 << 16);
        o2[2] = (alen2 & 0xFFFFu) | (rsv2 << 16);
    }
}

FQ_DEV void phase_filter_pe(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    u32* misc = lds + L.acc_misc;
    for (int pr = tid; pr < L.P; pr += nthreads) {
        const int gp = tile_first + pr;
        if (gp >= a.n) continue;
        const int R1 = pr, R2 = L.P + pr;
        int* flags = lds_i(lds, L.flags);
        const bool a1 = !(flags[R1] & RS_NULL), a2 = !(flags[R2] & RS_NULL);
        const bool dimer = (flags[R1] & RS_DIMER) != 0;
        const bool dedup_out = p.dedup && (flags[R1] & RS_DUP);
        int code1, code2;
        if (p.merge && a1 && a2 && (flags[R1] & RS_MERGE_OV)) {
            // passFilter(merged) (:526-535): the metrics of the two parts add up; the adjacent-difference
            // count gets the junction r1[len1-1] | comp(r2'[len2-1])
            const int m1 = lds_i(lds, L.mlen)[R1], m2 = lds_i(lds, L.mlen)[R2];
            const u32 ma1 = lds[L.met + 2 * R1], mb1 = lds[L.met + 2 * R1 + 1];
            const u32 ma2 = lds[L.met + 2 * R2], mb2 = lds[L.met + 2 * R2 + 1];
            const int f1 = lds_i(lds, L.front)[R1], f2 = lds_i(lds, L.front)[R2];
            const u32* s1 = lds_seq(L, lds, R1);
            const u32* s2 = lds_seq(L, lds, R2);
            const u8* q1 = (const u8*)lds_qual(L, lds, R1);
            const u8* q2 = (const u8*)lds_qual(L, lds, R2);
            int diff = (int)(mb1 >> 16) + (int)(mb2 >> 16);
            if (m1 > 0 && m2 > 0 && sym_at(s1, q1, f1 + m1 - 1) != sym_complement(sym_at(s2, q2, f2 + m2 - 1))) diff++;
            const int result = filter_code(a, lds, m1 + m2, (int)(ma1 & 0xFFFFu) + (int)(ma2 & 0xFFFFu),
                                           (int)(ma1 >> 16) + (int)(ma2 >> 16), (int)(mb1 & 0xFFFFu) + (int)(mb2 & 0xFFFFu), diff);
            lds_add_u32(&misc[MISC_FILTER + result], 2u);
            code1 = code2 = result;
            if (result == 0) {
                flags[R1] |= RS_MERGED | RS_STAT_POST;                               // postStats1->statRead(merged)
                flags[R2] |= RS_MERGED | RS_STAT_POST | RS_POST_TO1 | RS_POST_RC;
                lds_add_u32(&misc[MISC_MERGED], 1u);                                 // mergedCount++
                // 5-mers that straddle the junction (merged positions len1 .. len1+3); the stats pass
                // counts the ones inside either part
                u32* kmer1 = lds + L.acc_kmer + 1 * KMER_BINS;  // slot POST1
                for (int qj = 0; qj < 4 && qj < m2; qj++) {
                    const int pe = m1 + qj;
                    if (pe < 4) continue;
                    u32 km = 0;
                    bool ok = true;
                    for (int t = 0; t < 5; t++) {
                        const int x = pe - 4 + t;
                        const u32 sy = x < m1 ? sym_at(s1, q1, f1 + x) : sym_complement(sym_at(s2, q2, f2 + (m1 + m2 - 1 - x)));
                        if (sy == 4u) ok = false;
                        km |= (sy & 3u) << (2 * t);
                    }
                    if (ok && m1 > pe - 4) lds_add_u32(&kmer1[km], 1u);
                }
            }
        } else if (p.merge && a1 && a2 && p.merge_include_unmerged) {  // :536-559
            code1 = pass_filter(a, lds, R1, a1);
            code2 = pass_filter(a, lds, R2, a2);
            if (dimer) { code1 = 28; code2 = 28; }
            lds_add_u32(&misc[MISC_FILTER + code1], 1u);
            lds_add_u32(&misc[MISC_FILTER + code2], 1u);
            if (code1 == 0 && !dedup_out) flags[R1] |= RS_STAT_POST;
            if (code2 == 0 && !dedup_out) flags[R2] |= RS_STAT_POST | RS_POST_TO1;
        } else {
            code1 = pass_filter(a, lds, R1, a1);  // :565-566
            code2 = pass_filter(a, lds, R2, a2);
            if (dimer) { code1 = 28; code2 = 28; }     // :568-571
            const int worst = imax(code1, code2);
            lds_add_u32(&misc[MISC_FILTER + worst], 2u);  // addFilterResult(max, 2) :573
            // post-filtering Stats only see pairs that are written to out1/out2 (:577-591), and not
            // in merge mode (:588); with dedup the duplicate decision was taken beforehand
            if (!p.merge && !dedup_out && a1 && a2 && code1 == 0 && code2 == 0) {
                flags[R1] |= RS_STAT_POST;
                flags[R2] |= RS_STAT_POST;
            }
        }
        lds_i(lds, L.code)[R1] = code1;
        lds_i(lds, L.code)[R2] = code2;
        lds[L.swin + R1] = (u32)lds_i(lds, L.rlen0)[R1] | ((flags[R1] & RS_STAT_POST) ? (u32)lds_i(lds, L.len)[R1] << 16 : 0u);
        lds[L.swin + R2] = (u32)lds_i(lds, L.rlen0)[R2] | ((flags[R2] & RS_STAT_POST) ? (u32)lds_i(lds, L.len)[R2] << 16 : 0u);
        write_dup_pos(a, lds, pr, gp);
        write_read_result(a, lds, 0, R1, gp);
        write_read_result(a, lds, 1, R2, gp);
    }
}

// ---------------------------------------------------------------------------
// Phase E (single-end): lane = one read.  seprocessor.cpp:244-290.
// ---------------------------------------------------------------------------
FQ_DEV void phase_decide_se(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    u32* misc = lds + L.acc_misc;
    for (int R = tid; R < L.P; R += nthreads) {
        const int gp = tile_first + R;
        if (gp >= a.n) continue;
        int* flags = lds_i(lds, L.flags);
        int* lenv = lds_i(lds, L.len);
        const bool alive = !(flags[R] & RS_NULL);
        bool dimer = false;
        if (alive && p.adapter_enabled) {  // :244-261
            bool trimmed = false;
            if (p.has_a1) trimmed = apply_trim_by_sequence(a, lds, R, lds + L.adapt, p.alen1, misc);
            if (p.n_fasta) trimmed |= apply_fasta_trims(a, lds, R, (u32)(a.first + gp), misc);  // seprocessor.cpp:249-251
            if (trimmed) { lds_add_u32(&misc[MISC_ADAPTER_READS], 1u); flags[R] |= RS_ADAPTER; }
            if (trimmed && lenv[R] <= p.dimer_max_len) dimer = true;
        }
        if (alive && p.poly_x) {  // :263-266
            int poly, trimmed;
            const int nl = trim_poly_x(lds_seq(L, lds, R), (const u8*)lds_qual(L, lds, R), lds_i(lds, L.front)[R], lenv[R],
                                       p.poly_x_min, poly, trimmed);
            if (poly >= 0) {
                lds_add_u32(&misc[MISC_POLYX_READS + poly], 1u);
                lds_add_u32(&misc[MISC_POLYX_BASES + poly], (u32)trimmed);
                flags[R] |= RS_POLYX;
            }
            lenv[R] = nl;
        }
        if (alive && p.max_len1 > 0 && p.max_len1 < lenv[R]) lenv[R] = p.max_len1;  // :268-271
        if (dimer) flags[R] |= RS_DIMER;
    }
}

// Phase E3 (single-end): lane = one read.  Filter::passFilter and routing, seprocessor.cpp:273-290.
FQ_DEV void phase_filter_se(const KernelArgs& a, u32* lds, int tile_first, int tid, int nthreads) {
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    u32* misc = lds + L.acc_misc;
    for (int R = tid; R < L.P; R += nthreads) {
        const int gp = tile_first + R;
        if (gp >= a.n) continue;
        int* flags = lds_i(lds, L.flags);
        const bool alive = !(flags[R] & RS_NULL);
        int code = pass_filter(a, lds, R, alive);  // :273
        if (flags[R] & RS_DIMER) code = 28;
        lds_add_u32(&misc[MISC_FILTER + code], 1u);  // :278
        lds_i(lds, L.code)[R] = code;
        const bool dedup_out = p.dedup && (flags[R] & RS_DUP);
        if (!dedup_out && alive && code == 0) flags[R] |= RS_STAT_POST;  // :280-286
        const u32 sw = (u32)lds_i(lds, L.rlen0)[R] | ((flags[R] & RS_STAT_POST) ? (u32)lds_i(lds, L.len)[R] << 16 : 0u);
        lds[L.swin + R] = sw;
        if (a.split && p.front_lane) {
            const u32 kept = (flags[R] & RS_STAT_POST) ? (u32)lds_i(lds, L.len)[R] : 0u;
            split_stat_reads(a, misc, gp, (u32)lds_i(lds, L.rlen0)[R] | (kept ? ((u32)lds_i(lds, L.front)[R] + kept) << 16 : 0u), 0u, kept, 0u);
        } else if (a.split) split_stat_reads(a, misc, gp, sw, 0u);
        write_dup_pos(a, lds, R, gp);
        write_read_result(a, lds, 0, R, gp);
    }
}


// ---------------------------------------------------------------------------
// The fused kernel: persistent workgroups, grid-stride over tiles.
// ---------------------------------------------------------------------------
// Two tiles are in flight per workgroup: waves [0, W/2) own tile slot 0, waves [W/2, W) slot 1, each with its own
// barrier (tile_sync) and its own copy of the argument block (same but for the LDS layout, LdsLayout::halves).
// The halves run the same phase sequence half a tile apart (half 1 starts late), so that while one sits in a phase
// bound by the LDS pipe (Stats), by latency (the per-read / per-pair phases, the barriers) or by memory (staging),
// the other one's VALU-bound phases (overlap, hash, masks, metrics) have the SIMDs.  The accumulators are shared:
// both halves add to the same LDS counters.
// Duplicate's claim step (see dup_claim_body) issued from inside the fused kernel: the unit's thread fires the
// atomic_or of its one or two bloom bits at the start of the metrics phase and looks at what they returned only after
// the filter phase, so the device-scope atomics' latency hides behind two phases of other work.  `raw` keeps the
// returned words and the bit numbers; nothing is decided here.
struct ClaimRegs {
    u32 old0, old1, shifts;   // shifts: bit number of buffer 0 | buffer 1 << 8 | valid << 16
};
FQ_DEV void dup_claim_issue(const KernelArgs& a, u32* lds, int tile_first, int tid, ClaimRegs& c) {   // tid = unit of the tile
    const LdsLayout& L = a.L;
    const DevParams& p = a.p;
    c.old0 = c.old1 = c.shifts = 0;
    const int gp = tile_first + tid;
    if (tid < 0 || tid >= L.P || gp >= a.n) return;
    const int B = p.dup_bufnum;   // 1 or 2 here
    int tl = lds_i(lds, L.rlen0)[tid];
    if (p.paired) tl += lds_i(lds, L.rlen0)[L.P + tid];
    const u64* h1 = (const u64*)(lds + L.hash) + (size_t)tid * B;
    const u64* h2 = (const u64*)(lds + L.hash) + (size_t)(L.P + tid) * B;
    const u64 words = a.dup_bits >> 5;
    u64 h = h1[0] + (p.paired ? h2[0] : 0ull) + a.lut.dup_posum[(size_t)tl * B];
    u64 pos = h & (a.dup_bits - 1);
    c.shifts = (u32)(pos & 31) | (1u << 16);
    c.old0 = g_atomic_or_u32(&a.dup_bitmap[pos >> 5], 1u << (pos & 31));
    if (B > 1) {
        h = h1[1] + (p.paired ? h2[1] : 0ull) + a.lut.dup_posum[(size_t)tl * B + 1];
        pos = h & (a.dup_bits - 1);
        c.shifts |= (u32)(pos & 31) << 8;
        c.old1 = g_atomic_or_u32(&a.dup_bitmap[words + (pos >> 5)], 1u << (pos & 31));
    }
}
FQ_DEV void dup_claim_collect(const KernelArgs& a, int tile_first, int tid, const ClaimRegs& c) {
    if (!(c.shifts >> 16)) return;
    u32 won = ((c.old0 >> (c.shifts & 31u)) & 1u) ^ 1u;
    if (a.p.dup_bufnum > 1) won |= (((c.old1 >> ((c.shifts >> 8) & 31u)) & 1u) ^ 1u) << 1;
    a.claim_won[tile_first + tid] = (u8)won;
}

// SPLIT: the per-read kernel of the split plan (KernelArgs::split is set; no Stats code in this instantiation)
template <bool SPLIT>
FQ_DEV void fused_body(const FusedArgs& fa, u32* lds0) {
    const int tid0 = thread_id(), nt0 = block_threads();
    const KernelArgs& a0 = fa.h[0];
    {   // one-time per workgroup, all waves, slot 0's view: clear the accumulators, stage LUTs / primes / adapters
        const LdsLayout& L = a0.L;
        u32* lds = lds0;
        const int tid = tid0, nt = nt0;
        for (int i = tid; i < L.acc_end - L.acc_cyc; i += nt) lds[L.acc_cyc + i] = 0;
        const int lw = (a0.p.cycles + 2) / 2;  // u16 tables of cycles+1 entries, in dwords
        const u32* g0 = (const u32*)a0.lut.ov_limit;
        const u32* g1 = (const u32*)a0.lut.lowq_limit;
        const u32* g2 = (const u32*)a0.lut.cplx_min;
        for (int i = tid; i < lw; i += nt) {
            lds[L.lut_ov + i] = g0[i];
            lds[L.lut_lowq + i] = g1[i];
            lds[L.lut_cplx + i] = g2[i];
        }
        stage_primes(a0, lds, tid, nt);
        for (int i = tid; i < 2 * ADAPT_WORDS; i += nt) {
            const int which = i >= ADAPT_WORDS ? 1 : 0;
            const int w = i - which * ADAPT_WORDS;
            u32 v = 0;
            if (w < MAX_ADAPTER_WORDS) v = which ? a0.p.a2w[w] : a0.p.a1w[w];
            lds[L.adapt + i] = v;
        }
        if (a0.p.dup_enabled)
            for (int i = tid; i < 256; i += nt) {  // duplicate.cpp:92-109: A=7 T=222 C=74 G=31 (codes A0 T1 C2 G3)
                u32 v = 0;
                for (int k = 0; k < 4; k++) v |= ((0x1F4ADE07u >> (((i >> (2 * k)) & 3) * 8)) & 0xFFu) << (8 * k);
                lds[L.val4_lut + i] = v;
            }
        if (tid < 2 * L.halves) lds[L.bar + (tid >> 1) * L.tile_stride + (tid & 1)] = 0;   // the halves' barrier words
        block_sync();
        for (int i = tid; i < (SPLIT ? 0 : 4 * 128); i += nt) {  // quality table constants (the counters in between stay zero)
            const int q = i & 127;
            // stats.cpp:209-222: q30 ('?') counts into Q30 and Q20, q20 ('5') into Q20.  Character 0 = no base.
            const u64 inc = q == 0 ? 0ull
                                   : (1ull | ((u64)(q >= 53) << CYC_Q20_SHIFT) | ((u64)(q >= 63) << CYC_Q30_SHIFT) |
                                      ((u64)(u32)(q - 33) << CYC_QSUM_SHIFT));
            u32* e = lds + L.acc_qh + i * QT_DWORDS;
            e[QT_INC] = (u32)inc;
            e[QT_INC + 1] = (u32)(inc >> 32);
            e[QT_ONE] = q == 0 ? 0u : 1u;
        }
        block_sync();
    }
    // from here on each half is its own little workgroup
    const int nh = a0.L.halves;
    const int nt = nt0 / nh;
    const int half = nh == 2 ? (int)uniform((u32)(tid0 >= nt ? 1 : 0)) : 0;
    const int tid = tid0 - half * nt;
    const KernelArgs& a = fa.h[half];
    const LdsLayout& L = a.L;
    u32* lds = lds0 + half * a0.L.tile_stride;
    const int vblock = block_id() * nh + half, vgrid = grid_blocks() * nh;
    if (half) for (int i = 0; i < a.half_skew; i++) nap();
    const bool timing_on = a.phase_cycles != nullptr;  // uniform
    const bool timing = timing_on && tid0 == 0;
    u64 tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool vec = a.prefetch != 0;       // uniform: 16-byte tile copies
    const bool prefetch = a.prefetch == 1;  // ... issued one tile ahead
    // Only full tiles take the vector path.  The tile is fetched (all chunks in flight, one wait) when the loop
    // reaches it; what runs ahead is a one-dword-per-cache-line touch of the NEXT tile (tile_warm: one register for
    // the length of the N-mask pass instead of the 17 a register-held prefetch keeps for a whole tile - under the
    // 128-register cap of a 1024-lane workgroup those were spilled, and a spill behind a load is a wait for it),
    // so the fetch hits L2 / the Infinity Cache.
    for (int tile = vblock; tile < a.tiles; tile += vgrid) {
        const int tile_first = tile * L.P;
        const int n_valid = imin(L.P, a.n - tile_first);
        u64 t0 = timing ? cycle_counter() : 0, t1;
#define FQ_STAMP(k) if (timing) { t1 = cycle_counter(); tacc[k] += t1 - t0; t0 = t1; }
        if (vec && n_valid == L.P) {
            tile_stage(a, lds, tile_first, tid, nt);
        } else {
            phase_load(a, lds, tile_first, tid, nt);
        }
        tile_sync(a, lds, nt);
        const int next = tile + vgrid;
        const bool touch = prefetch && next < a.tiles;   // uniform
        u32 warm = 0;
        if (touch) warm = tile_warm(a, next * L.P, tid);
        phase_nmask(a, lds, tid, nt);
        if (touch) touch_done(warm);
        tile_sync(a, lds, nt);
        FQ_STAMP(0)
#ifdef FQ_PROFILE_ABLATION
        const u32 skip = a.debug_skip;   // profiling build only: 0 in any real run
#else
        const u32 skip = 0;
#endif
        if (!SPLIT && !a.p.stats_one_pass) phase_stats<ST_PRE, false>(a, lds, n_valid, tid, nt);  // Stats::statRead on the original reads
        if (!(skip & 1u)) {
            phase_masks(a, lds, n_valid, tid, nt);
            phase_rc(a, lds, tid, nt);
        }
        if (timing_on) { tile_sync(a, lds, nt); FQ_STAMP(8) }
        if (!(skip & 2u)) phase_hash(a, lds, tid, nt);
        tile_sync(a, lds, nt);
        FQ_STAMP(1)
        // Duplicate's claim: fired here by threads that idle through the thin phases (trim runs on the first NR lanes), the
        // hash values are final since the last barrier; collected after the filter phase
        ClaimRegs claim;
        claim.shifts = 0;
        const int claim_unit = tid - (nt >= 2 * L.NR ? nt - L.P : 0);
        if (a.claim_won) dup_claim_issue(a, lds, tile_first, claim_unit, claim);
        if (!(skip & 32u)) phase_trim(a, lds, tile_first, tid, nt);
        tile_sync(a, lds, nt);
        FQ_STAMP(2)
        if (a.p.poly_g) {
            phase_polyg(a, lds, tid, nt);
            tile_sync(a, lds, nt);
        }
        FQ_STAMP(3)
        if (!(skip & 4u)) phase_overlap(a, lds, tid, nt);
        if (a.p.allow_gap) {
            phase_overlap_gap(a, lds, tid, nt);
            tile_sync(a, lds, nt);
        }
        FQ_STAMP(4)
        if (!(skip & 32u)) {
            if (a.p.paired) phase_decide_pe(a, lds, tile_first, tid, nt);
            else phase_decide_se(a, lds, tile_first, tid, nt);
        }
        tile_sync(a, lds, nt);
        if (a.p.overlapped_out) {   // :488-495, in front of merge mode's own analysis (:518) as in the reference
            phase_ovout_begin(a, lds, tile_first, tid, nt);
            tile_sync(a, lds, nt);
            phase_overlap(a, lds, tid, nt, true);
            phase_ovout_end(a, lds, tile_first, tid, nt);
            tile_sync(a, lds, nt);
        }
        if (a.p.merge) {
            phase_overlap(a, lds, tid, nt);
            phase_merge(a, lds, tile_first, tid, nt);
            tile_sync(a, lds, nt);
        }
        FQ_STAMP(5)
        if (!(skip & 8u)) phase_metrics(a, lds, tid, nt);
        tile_sync(a, lds, nt);
        FQ_STAMP(9)
        if (!(skip & 32u)) {
            if (a.p.paired && !a.p.merge) phase_filter_pe_plain(a, lds, tile_first, tid, nt);
            else if (a.p.paired) phase_filter_pe(a, lds, tile_first, tid, nt);
            else phase_filter_se(a, lds, tile_first, tid, nt);
        }
        if (a.claim_won) dup_claim_collect(a, tile_first, claim_unit, claim);
        tile_sync(a, lds, nt);
        FQ_STAMP(6)
        // Stats::statRead on what is written out (+ on the original reads in one-pass mode)
        if (!SPLIT && !(skip & 16u)) {
            if (a.p.stats_one_pass) phase_stats_both(a, lds, n_valid, tid, nt);
            else if (a.p.merge) phase_stats<ST_POST, true>(a, lds, n_valid, tid, nt);
            else phase_stats<ST_POST, false>(a, lds, n_valid, tid, nt);
        }
        if (!SPLIT) tile_sync(a, lds, nt);   // split: the barrier behind the filter phase already closed the tile
        FQ_STAMP(7)
#undef FQ_STAMP
    }
    if (timing)
        for (int k = 0; k < 10; k++) g_atomic_add_u64(&a.phase_cycles[k], tacc[k]);
    // both halves are done: flush this workgroup's accumulators to its slab (plain coalesced stores)
    block_sync();
    u32* slab = a0.slabs + (size_t)block_id() * a0.slab_dwords;
    for (int i = tid0; i < a0.slab_dwords; i += nt0) slab[i] = lds0[a0.L.acc_cyc + i];
}

// ---------------------------------------------------------------------------
// --dedup needs the duplicate decision BEFORE routing (peprocessor.cpp:396-402, 575): a light
// first pass stages each tile, hashes the original reads and writes the hash values; the dup
// kernels then decide, and the fused kernel reads the decision (KernelArgs::dupflag).
// ---------------------------------------------------------------------------
FQ_DEV void hash_body(const KernelArgs& a, u32* lds) {
    const LdsLayout& L = a.L;
    const int tid = thread_id(), nt = block_threads();
    if (a.p.dup_enabled)
        for (int i = tid; i < 256; i += nt) {  // duplicate.cpp:92-109: A=7 T=222 C=74 G=31 (codes A0 T1 C2 G3)
            u32 v = 0;
            for (int k = 0; k < 4; k++) v |= ((0x1F4ADE07u >> (((i >> (2 * k)) & 3) * 8)) & 0xFFu) << (8 * k);
            lds[L.val4_lut + i] = v;
        }
    stage_primes(a, lds, tid, nt);
    block_sync();
    for (int tile = block_id(); tile < a.tiles; tile += grid_blocks()) {
        const int tile_first = tile * L.P;
        phase_load(a, lds, tile_first, tid, nt);
        block_sync();
        phase_nmask(a, lds, tid, nt);
        block_sync();
        phase_hash(a, lds, tid, nt);
        block_sync();
        for (int u = tid; u < L.P; u += nt)
            if (tile_first + u < a.n) write_dup_pos(a, lds, u, tile_first + u);
        block_sync();
    }
}

// ---------------------------------------------------------------------------
// Slab reduction: the device analogue of Stats::merge / FilterResult::merge
// (stats.cpp:877-955, filterresult.cpp:38-89) over the workgroups of one launch,
// unpacking the LDS formats into the int64 counter block of include/fastp_gpu.h.
// ---------------------------------------------------------------------------
struct ReduceArgs {
    const u32* slabs;
    int slab_dwords, nblocks;
    LdsLayout L;       // Cp / C of the per-cycle accumulators
    // where the regions sit inside a slab (dwords): per-cycle u64s at 0, then
    int off_kmer, off_qh, off_misc;
    int qh_stride, qh_count;   // histogram counter of (slot, character) = slab[off_qh + (slot * 128 + q) * qh_stride + qh_count]
    // which items this launch folds: 1 = per-cycle + k-mer + histogram, 2 = the MISC_* counters, 3 = both (one slab set)
    int parts;
    int isize_max;
    int one_pass;      // slabs hold kept (POST slot) / dropped (PRE slot): PRE = kept + dropped
    int front[2];      // one_pass with a uniform front trim (DevParams::front_lane): the kept bases of mate m sit at their
                       // ORIGINAL cycle; in the POST Stats a read starts behind its front
    int merge_tail;    // one_pass, DevParams::merge_lane: slot 3 of the Stats slabs = what read 2 gives to the POST Stats object of read 1
                       // (stats4_tail_pass) - added to that object only; the per-read kernel's slabs carry KMER_BINS more counters
                       // behind the MISC_* ones: the merged reads' junction 5-mers (LaneLds::jkmer, fastp's index)
    int64_t* ctr;      // counter block
    // fastp_gpu_counter_layout offsets
    int64_t o_filter, o_adapter_reads, o_adapter_bases, o_polyx_reads, o_polyx_bases, o_correction,
        o_corrected_reads, o_merged, o_isize, o_stats[4], st_reads, st_length_sum, st_qual_hist, st_kmer,
        st_cycle, cycles;
};

// Work item = (slab element, group of REDUCE_GROUP workgroup slabs): consecutive lanes read
// consecutive slab dwords (coalesced), sum them over the group's slabs and add the partial
// sums to the int64 counters with device atomics.
// One-pass Stats mode: the POST slots hold "kept", the PRE slots "dropped": a kept element is
// added to its own (POST) Stats and to the PRE Stats of the same mate.
#ifndef FQ_REDUCE_GROUP
#define FQ_REDUCE_GROUP 64   // 16: 0.050 ms per fold of 256 slabs (16-way contention on every counter), 64: see profiles/r02t
#endif
enum { REDUCE_GROUP = FQ_REDUCE_GROUP };

FQ_DEV void reduce_body(const ReduceArgs& r) {
    const LdsLayout& L = r.L;
    const int C = L.C, Cp = L.Cp;
    const int n_cyc = 4 * N_CLS * Cp, n_kmer = 4 * KMER_BINS, n_qh = 4 * 128;
    const int n_misc = MISC_ISIZE + r.isize_max + 1 + (r.merge_tail ? (int)KMER_BINS : 0);
    const int n_stats = n_cyc + n_kmer + n_qh;
    const int lo = (r.parts & 1) ? 0 : n_stats;
    const int total = ((r.parts & 2) ? n_stats + n_misc : n_stats) - lo;
    const int chunks = (total + block_threads() - 1) / block_threads();  // workgroups per slab group
    const int g = block_id() / chunks;
    int item = (block_id() - g * chunks) * block_threads() + thread_id();
    if (item >= total) return;
    item += lo;
    const int b0 = g * REDUCE_GROUP, b1 = imin(r.nblocks, b0 + REDUCE_GROUP);
    if (b0 >= b1) return;
    const int64_t CC = r.cycles;
    if (item < n_cyc) {
        const int slot = item / (N_CLS * Cp);   // cyc_index: [slot][cycle][class]
        const int rem = item - slot * N_CLS * Cp;
        const int c = rem / N_CLS, cls = rem - c * N_CLS;
        if (c >= C) return;
        int64_t cnt = 0, q20 = 0, q30 = 0, qs = 0;
        for (int b = b0; b < b1; b++) {
            const u32* s = r.slabs + (size_t)b * r.slab_dwords + 2 * item;
            const u64 v = (u64)s[0] | ((u64)s[1] << 32);
            cnt += (int64_t)(v & 0x3FFFu);
            q20 += (int64_t)((v >> CYC_Q20_SHIFT) & 0x3FFFu);
            q30 += (int64_t)((v >> CYC_Q30_SHIFT) & 0x3FFFu);
            qs += (int64_t)(v >> CYC_QSUM_SHIFT);
        }
        if (!(cnt | qs)) return;
        const int bin = (int)sym_bin((u32)cls);  // 'A'&7=1 'T'&7=4 'C'&7=3 'G'&7=7 'N'&7=6
        const bool tail = r.merge_tail && slot == 3;
        for (int tgt = tail ? 1 : slot; tgt >= 0; tgt -= 1) {
            int64_t* st = r.ctr + r.o_stats[tgt] + r.st_cycle;  // Stats::mCycleBuffer layout (stats.cpp:54-63)
            // the POST Stats of a front-trimmed mate: cycle c of the original read is cycle c - front of the read that is written out
            const int cc = (r.one_pass && (tgt & 1)) ? c - r.front[tgt >> 1] : c;
            if (cc >= 0) {
                if (q30) g_atomic_add_i64(&st[(0 * 8 + bin) * CC + cc], q30);  // mCycleQ30Bases
                if (q20) g_atomic_add_i64(&st[(1 * 8 + bin) * CC + cc], q20);  // mCycleQ20Bases
                g_atomic_add_i64(&st[(2 * 8 + bin) * CC + cc], cnt);           // mCycleBaseContents
                g_atomic_add_i64(&st[(3 * 8 + bin) * CC + cc], qs);            // mCycleBaseQual
                g_atomic_add_i64(&st[32 * CC + cc], cnt);                      // mCycleTotalBase
                g_atomic_add_i64(&st[33 * CC + cc], qs);                       // mCycleTotalQual
            }
            if (tail || !(r.one_pass && (tgt & 1))) break;  // kept -> also the PRE Stats of the mate
        }
        return;
    }
    int src;   // dword of the slab that holds the item
    if (item < n_cyc + n_kmer) src = r.off_kmer + (item - n_cyc);
    else if (item < n_stats) src = r.off_qh + (item - n_cyc - n_kmer) * r.qh_stride + r.qh_count;
    else src = r.off_misc + (item - n_stats);
    int64_t sum = 0;
    for (int b = b0; b < b1; b++) sum += (int64_t)r.slabs[(size_t)b * r.slab_dwords + src];
    if (!sum) return;
    if (item < n_cyc + n_kmer) {
        const int k = item - n_cyc;
        const int slot = k / KMER_BINS, km = k - slot * KMER_BINS;
        // LDS index has the earliest base in the low bits; fastp's has it in the high bits
        const u32 fk = ((km & 3u) << 8) | (((km >> 2) & 3u) << 6) | (((km >> 4) & 3u) << 4) | (((km >> 6) & 3u) << 2) |
                       ((km >> 8) & 3u);
        if (r.merge_tail && slot == 3) { g_atomic_add_i64(&r.ctr[r.o_stats[1] + r.st_kmer + fk], sum); return; }
        g_atomic_add_i64(&r.ctr[r.o_stats[slot] + r.st_kmer + fk], sum);
        if (r.one_pass && (slot & 1)) g_atomic_add_i64(&r.ctr[r.o_stats[slot - 1] + r.st_kmer + fk], sum);
    } else if (item < n_stats) {
        const int k = item - n_cyc - n_kmer;
        const int slot = k / 128, q = k - slot * 128;
        if (q == 0) return;  // character 0 = "no base" (bytes past a read's end): never a real quality (>= '!')
        if (r.merge_tail && slot == 3) { g_atomic_add_i64(&r.ctr[r.o_stats[1] + r.st_qual_hist + q], sum); return; }
        g_atomic_add_i64(&r.ctr[r.o_stats[slot] + r.st_qual_hist + q], sum);
        if (r.one_pass && (slot & 1)) g_atomic_add_i64(&r.ctr[r.o_stats[slot - 1] + r.st_qual_hist + q], sum);
    } else {
        const int k = item - n_stats;
        int64_t dst;
        if (k < MISC_ADAPTER_READS) dst = r.o_filter + k;
        else if (k == MISC_ADAPTER_READS) dst = r.o_adapter_reads;
        else if (k == MISC_ADAPTER_BASES) dst = r.o_adapter_bases;
        else if (k < MISC_POLYX_BASES) dst = r.o_polyx_reads + (k - MISC_POLYX_READS);
        else if (k < MISC_CORRECTION) dst = r.o_polyx_bases + (k - MISC_POLYX_BASES);
        else if (k < MISC_CORRECTED_READS) dst = r.o_correction + (k - MISC_CORRECTION);
        else if (k == MISC_CORRECTED_READS) dst = r.o_corrected_reads;
        else if (k == MISC_MERGED) dst = r.o_merged;
        else if (k < MISC_STAT_LENSUM) dst = r.o_stats[k - MISC_STAT_READS] + r.st_reads;
        else if (k < MISC_ISIZE) dst = r.o_stats[k - MISC_STAT_LENSUM] + r.st_length_sum;
        else if (k <= MISC_ISIZE + r.isize_max) dst = r.o_isize + (k - MISC_ISIZE);
        else dst = r.o_stats[1] + r.st_kmer + (k - (MISC_ISIZE + r.isize_max + 1));   // (merge_tail) the junction 5-mers
        g_atomic_add_i64(&r.ctr[dst], sum);
    }
}


// ---------------------------------------------------------------------------
// Duplicate::applyBloomFilter (duplicate.cpp:150-163) with the reference's
// SEQUENTIAL semantics, evaluated in parallel:
//   pair g is a duplicate  <=>  for every buffer i the bit pos_i(g) was set by a
//   pair with a smaller index (in an earlier batch: "committed", or earlier in
//   this batch).
// probe  : read the committed bitmaps; pairs with a missing bit register
//          (buffer, bit) -> min pair index in an open-addressing table.
// resolve: a missing bit counts as set iff the table's min index is smaller than
//          the pair's own; then commit the bits (atomic OR) and count.
// Bitmaps are u32 words: bit `pos` lives in word pos>>5, bit pos&31 (the same bit
// identity as the reference's byte array: byte pos>>3, bit pos&7).
// ---------------------------------------------------------------------------
struct DupArgs {
    const u64* dup_pos;   // [n][B] Duplicate::seq2intvector values, base-value part (write_dup_pos)
    const u64* posum;     // [(2*max_len+1)][B] position part by total length (DevLuts::dup_posum)
    const u16* len[2];    // read lengths of the launch (len[1] only when paired)
    int n, B;
    u64 bits;             // mBufLenInBits
    u32* bitmap;          // [B][bits/32]
    u64* table;           // open addressing, EMPTY = ~0, entry = key(38 bit) << 25 | pair index
                          // (bit 63 of a real entry is always 0, so it never equals EMPTY)
    int table_log2;
    u8* need;             // [n] mask of buffers whose bit was not committed
    u32* res[2];          // result records (flags byte gets RS_DUP) - used when dupflag == nullptr
    u8* dupflag;          // [n] --dedup: the decision goes here instead (the records do not exist yet)
    int paired;
    int64_t* ctr_total;
    int64_t* ctr_dups;
    // sharded runs, pass 1 (fastp_gpu_dup_scan_device): keep the bit positions and the mask of
    // buffers an earlier unit of this stream had set; no decision, no counters
    u64* scan_pos;        // [n][B]
    u8* scan_mask;        // [n]
    // claim / winners / finish form (dup_claim_body ...)
    u8* setw;             // [n] buffers a unit claimed but that turn out to have had an earlier unit of this launch
    u32* cfilter;         // 2^DUP_CF_LOG2 bits: keys some unit of this launch lost (a one-hash Bloom filter in front of the table)
};

// sharded runs, pass 2: the decision from the scan state and the preceding shards' bitmaps
struct DupFinalArgs {
    const u64* scan_pos;  // [n][B]
    const u8* scan_mask;  // [n]
    const u32* prefix;    // [B][bits/32] OR of the preceding shards' bitmaps, or nullptr
    u64 bits;
    int n, B;
    u8* dupflag;          // [n] --dedup: the fused kernel reads the decision ...
    u32* res[2];          // ... otherwise the records exist already: their flags byte gets RS_DUP
    int paired;
    int64_t* ctr_total;
    int64_t* ctr_dups;
};

// --dedup on the lane plan (round 5): the per-read kernel hashes and claims as it does without --dedup and writes its records
// as if no unit were a duplicate; Duplicate's tail then leaves the decisions in dupflag, and this kernel applies them BEFORE the
// Stats kernel runs.  What `dedupOut` changes in the worker loop is only the last step (peprocessor.cpp:574-591,
// seprocessor.cpp:280-286): the unit is not written out and not given to the post-filtering Stats objects - so a duplicate
// gets its RS_DUP flag, and one that had been marked as written out loses its kept range (swin: the Stats kernel then counts
// every base as dropped) and is taken out of the POST Stats' read count and length sum.
struct DedupApplyArgs {
    int n, paired;
    const u8* dupflag;
    u32* res[2];
    u32* swin[2];
    int64_t* st_reads[2];   // POST Stats of mate m: mReads, mLengthSum (counter block)
    int64_t* st_lensum[2];
};
FQ_DEV void dedup_apply_body(const DedupApplyArgs& d, u32* lds) {
    if (thread_id() < 4) lds[thread_id()] = 0;
    block_sync();
    const int g = block_id() * block_threads() + thread_id();
    u32 rd[2] = {0, 0}, ln[2] = {0, 0};
    if (g < d.n && d.dupflag[g]) {
        for (int m = 0; m < (d.paired ? 2 : 1); m++) {
            d.res[m][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
            const u32 sw = d.swin[m][g];
            if (sw >> 16) {   // had been counted as written out
                rd[m] = 1;
                ln[m] = d.res[m][(size_t)g * 3] >> 16;   // the record's length
                d.swin[m][g] = sw & 0xFFFFu;
            }
        }
    }
    for (int m = 0; m < 2; m++) {   // (uniform)
        const u64 any = ballot(rd[m] != 0);
        if (any) {
            u32 l = ln[m];
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) l += shfl_xor(l, sh);
            if (lane_id() == 0) {
                lds_add_u32(&lds[2 * m], (u32)popc64(any));
                lds_add_u32(&lds[2 * m + 1], l);
            }
        }
    }
    block_sync();
    if (thread_id() < 2 && lds[2 * thread_id()]) {
        g_atomic_add_i64(d.st_reads[thread_id()], -(int64_t)lds[2 * thread_id()]);
        g_atomic_add_i64(d.st_lensum[thread_id()], -(int64_t)lds[2 * thread_id() + 1]);
    }
}

// -c on the lane plan (round 5): the Stats kernel has counted every kept base of a corrected read with its ORIGINAL letter and
// quality (into the kept slot: PRE and POST alike).  The PRE Stats are right that way (statRead runs before BaseCorrector,
// peprocessor.cpp:393-394); the POST Stats see the corrected read (:583-586).  One lane per read that has corrections (their
// chain: ovr_corr_link_body) and was written out: for each corrected position inside the kept range the POST Stats' per-cycle
// arrays and quality histogram move from the old (class, quality) to the new, and the 5-mers that cover a corrected position
// are taken out as the original read gives them and put back as the corrected read gives them.  Sparse: a few global atomics
// per corrected read.
struct CorrStatsArgs {
    int n, paired;
    int sw_g, qw_g;
    const u32* seq[2];
    const u32* qual[2];
    const u32* swin[2];       // original length | END of the kept range << 16 (0: not written out), after --dedup's decisions
    int front[2];             // start of the kept range (DevParams::lane_front*)
    int merge;                // DevParams::merge_lane: post[1] is the POST Stats object of read 1 as well (--include_unmerged)
    const u32* corr;          // the launch's corrections (fastp_gpu_correction: read | pos, base << 16, qual << 24)
    const u32* corr_head;     // [n * (paired ? 2 : 1)] chain heads (entry + 1), corr_next [capacity]
    const u32* corr_next;
    int64_t* post[2];         // the POST Stats object of mate m in the counter block
    int64_t st_qual_hist, st_kmer, st_cycle, cycles;
};
FQ_DEV u32 ascii_sym(u32 b) { return b == 'A' ? 0u : b == 'T' ? 1u : b == 'C' ? 2u : b == 'G' ? 3u : 4u; }
FQ_DEV u32 row_sym(const u32* srow, const u8* qrow, int j) { return (qrow[j] & 0x80u) ? 4u : ((srow[j >> 4] >> ((j & 15) * 2)) & 3u); }   // A0 T1 C2 G3 N4
// Persistent workgroups (a few per launch), the reads dealt out grid-stride; the deltas are gathered in LDS as signed 32-bit
// sums and folded into the int64 block once per workgroup - the first form added every delta to the block with a global
// atomic: a thousand per counter and launch, 0.69 ms (profiles/r05_other_configs_kernels.txt).
// LDS per mate: [33 * cycles] = the Stats object's arrays 0..31 (Q30 / Q20 / content / quality sum x 8 bins) + mCycleTotalQual,
// [128] the quality histogram, [1024] the 5-mers.
FQ_DEV void corr_stats_body(const CorrStatsArgs& c, u32* lds) {
    const int tid = thread_id(), nt = block_threads();
    const int CC = (int)c.cycles;
    const int per_mate = 33 * CC + 128 + KMER_BINS, mates = c.paired ? 2 : 1;
    for (int i = tid; i < mates * per_mate; i += nt) lds[i] = 0;
    block_sync();
    const int reads = c.paired ? 2 * c.n : c.n;
    for (int t = block_id() * nt + tid; t < reads; t += grid_blocks() * nt) {
        const u32 head = c.corr_head[t];
        if (!head) continue;
        const int g = c.paired ? t >> 1 : t, m = c.paired ? (t & 1) : 0;
        const u32 sw = c.swin[m][g];
        // (DevParams::merge_lane, read 2: a merged read's second part - bit 15 - holds no edit, or the lane kernel counted it itself)
        if (c.merge && m == 1 && (sw >> 31)) continue;
        const int lk = (int)((sw >> 16) & (c.merge ? 0x3FFFu : 0xFFFFu)), F = c.front[m];
        if (lk <= F) continue;                                 // not written out: no POST Stats
        const u32* srow = c.seq[m] + (size_t)g * c.sw_g;
        const u8* qrow = (const u8*)(c.qual[m] + (size_t)g * c.qw_g);
        u32* cyc = lds + m * per_mate;
        u32* hist = cyc + 33 * CC;
        u32* kmer = hist + 128;
        // (a read can hold any number of edits - only the first 50 bases of an overlap are held to the mismatch limit: for every
        // edit the chain is walked once more, for the neighbours within four bases; such reads are few)
        const int rl0 = (int)(sw & 0xFFFFu);
        for (u32 e = head; e; e = c.corr_next[e - 1]) {
            const u32 w1 = c.corr[2 * (size_t)(e - 1) + 1];
            const int P = (int)(w1 & 0xFFFFu);
            if (P < F || P >= lk) continue;                    // trimmed away afterwards: the POST Stats never saw the base
            const int cc = P - F;
            // the nine bases P - 4 .. P + 4 of the original read (independent loads: one round trip), 4 = N / outside the read
            u32 so[9], sn[9];
#pragma unroll
            for (int d = 0; d < 9; d++) {
                const int j = P - 4 + d;
                so[d] = (j >= 0 && j < rl0) ? row_sym(srow, qrow, j) : 4u;
                sn[d] = so[d];
            }
            const u32 qo = (u32)qrow[P] & 0x7Fu, qn = w1 >> 24;
            // ... as the corrected read has them; and which of the five 5-mers that cover P belong to an edit at a smaller position
            u32 other = 0;                                     // bit jj: the 5-mer that ends at P + jj is a smaller edit's
            for (u32 e2 = head; e2; e2 = c.corr_next[e2 - 1]) {
                const u32 v1 = c.corr[2 * (size_t)(e2 - 1) + 1];
                const int P2 = (int)(v1 & 0xFFFFu), d2 = P2 - (P - 4);
                const u32 s2 = ascii_sym((v1 >> 16) & 0xFFu);
#pragma unroll
                for (int d = 0; d < 9; d++) sn[d] = d == d2 ? s2 : sn[d];
                if (P2 < P) {
#pragma unroll
                    for (int jj = 0; jj < 5; jj++) other |= (P + jj <= P2 + 4) ? (1u << jj) : 0u;
                }
            }
            const int bo = (int)sym_bin(so[4]), bn = (int)sym_bin(sn[4]);
            if (qo >= 63u) lds_add_u32(&cyc[(0 * 8 + bo) * CC + cc], (u32)-1);     // stats.cpp:209-222
            if (qo >= 53u) lds_add_u32(&cyc[(1 * 8 + bo) * CC + cc], (u32)-1);
            lds_add_u32(&cyc[(2 * 8 + bo) * CC + cc], (u32)-1);
            lds_add_u32(&cyc[(3 * 8 + bo) * CC + cc], 0u - (qo - 33u));
            if (qn >= 63u) lds_add_u32(&cyc[(0 * 8 + bn) * CC + cc], 1u);
            if (qn >= 53u) lds_add_u32(&cyc[(1 * 8 + bn) * CC + cc], 1u);
            lds_add_u32(&cyc[(2 * 8 + bn) * CC + cc], 1u);
            lds_add_u32(&cyc[(3 * 8 + bn) * CC + cc], qn - 33u);
            lds_add_u32(&cyc[32 * CC + cc], qn - qo);          // mCycleTotalQual (mCycleTotalBase is unchanged)
            lds_add_u32(&hist[qo], (u32)-1);
            lds_add_u32(&hist[qn], 1u);
            // 5-mers: the one that ends at j = P + jj is bases jj .. jj + 4 of the window
#pragma unroll
            for (int jj = 0; jj < 5; jj++) {
                const int j = P + jj;
                if (j - 4 < F || j >= lk || ((other >> jj) & 1u)) continue;   // a 5-mer of the read that is written out needs j - 4 >= F
                u32 ko = 0, kn = 0;
                bool vo = true, vn = true;
#pragma unroll
                for (int t = 0; t < 5; t++) {                  // fastp's index: the earliest base in the high bits (stats.cpp:236, :250)
                    vo = vo && so[jj + t] < 4u;
                    vn = vn && sn[jj + t] < 4u;
                    ko = (ko << 2) | (so[jj + t] & 3u);
                    kn = (kn << 2) | (sn[jj + t] & 3u);
                }
                if (vo) lds_add_u32(&kmer[ko], (u32)-1);
                if (vn) lds_add_u32(&kmer[kn], 1u);
            }
        }
    }
    block_sync();
    for (int i = tid; i < mates * per_mate; i += nt) {
        const int v = (int)lds[i];
        if (!v) continue;
        const int m = i / per_mate, k = i - m * per_mate;
        int64_t* st = c.post[m];
        int64_t* dst = k < 32 * CC ? st + c.st_cycle + k : k < 33 * CC ? st + c.st_cycle + 33 * (int64_t)CC + (k - 32 * CC)
                     : k < 33 * CC + 128 ? st + c.st_qual_hist + (k - 33 * CC) : st + c.st_kmer + (k - 33 * CC - 128);
        g_atomic_add_i64(dst, (int64_t)v);
    }
}

// images[k] <- OR of images[j], j < k (exclusive prefix, in place; images[0] <- 0); `chunks` 16-byte chunks each.
// With dst != nullptr instead: dst <- OR of all n_images (images untouched).
struct OrArgs {
    u32x4* images;
    u32x4* dst;
    u64 chunks;
    int n_images;
};

// Read::convertPhred64To33 (src/read.cpp, called from FastqReader::read when --phred64 is given): the quality characters of
// the n parsed records become max(33, q - 31) - in the TEXT (what the formatter copies and the text kernel reads) and in
// the packed rows (bit 7 of a row byte marks an N base and stays).  One wavefront per record.
struct Phred64Args {
    int n;
    int qs;                 // bytes between two packed quality rows
    u8* text;
    const u32* line_off;    // [4 * n]
    const u32* line_len;
    u8* qual;
};
FQ_DEV void phred64_body(const Phred64Args& a) {
    const int waves = block_threads() >> 6;
    for (int r = block_id() * waves + wave_id(); r < a.n; r += grid_blocks() * waves) {
        const u32 off = a.line_off[4 * (size_t)r + 3], len = a.line_len[4 * (size_t)r + 3];
        u8* t = a.text + off;
        u8* q = a.qual + (size_t)r * (size_t)a.qs;
        for (u32 j = (u32)lane_id(); j < len; j += 64u) {
            const u32 c = t[j];
            const u32 v = c > 64u ? c - 31u : 33u;
            t[j] = (u8)v;
            q[j] = (u8)((q[j] & 0x80u) | v);
        }
    }
}

enum { DUP_IDX_BITS = 25 };  // pairs per launch < 2^25
// bit position of unit g in buffer i: (base-value part + position part) mod mBufLenInBits
FQ_DEV u64 dup_bit(const DupArgs& d, int g, int i) {
    int tl = (int)d.len[0][g];
    if (d.paired) tl += (int)d.len[1][g];
    const u64 h = d.dup_pos[(size_t)g * d.B + i] + d.posum[(size_t)tl * d.B + i];
    return h & (d.bits - 1);  // mBufLenInBits is a power of two (duplicate.cpp:13-47)
}
FQ_DEV u64 dup_key(int i, u64 pos) { return ((u64)i << 35) | pos; }
FQ_DEV u32 dup_slot(u64 key, int log2) { return (u32)((key * 0x9E3779B97F4A7C15ull) >> (64 - log2)); }

FQ_DEV void dup_probe_body(const DupArgs& d) {
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    const u64 words = d.bits >> 5;
    const u32 tmask = (1u << d.table_log2) - 1u;
    for (int g = gid; g < d.n; g += gstride) {
        u32 need = 0;
        for (int i = 0; i < d.B; i++) {
            const u64 pos = dup_bit(d, g, i);
            const u32 w = d.bitmap[(size_t)i * words + (pos >> 5)];
            if (!((w >> (pos & 31)) & 1u)) {
                need |= 1u << i;
                const u64 key = dup_key(i, pos);
                const u64 entry = (key << DUP_IDX_BITS) | (u64)g;
                u32 slot = dup_slot(key, d.table_log2);
                for (;;) {
                    u64 cur = g_atomic_cas_u64(&d.table[slot], ~0ull, entry);
                    if (cur == ~0ull) break;              // claimed an empty slot
                    if ((cur >> DUP_IDX_BITS) == key) {   // same bit: keep the smallest pair index
                        g_atomic_min_u64(&d.table[slot], entry);
                        break;
                    }
                    slot = (slot + 1) & tmask;
                }
            }
        }
        d.need[g] = (u8)need;
    }
}

FQ_DEV void dup_resolve_body(const DupArgs& d, u32* block_count) {
    if (thread_id() == 0) *block_count = 0;
    block_sync();
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    const u64 words = d.bits >> 5;
    const u32 tmask = (1u << d.table_log2) - 1u;
    const int rounds = (d.n + gstride - 1) / gstride;
    for (int it = 0; it < rounds; it++) {
        const int g = gid + it * gstride;
        bool is_dup = false;
        if (g < d.n) {
            const u32 need = d.need[g];
            is_dup = true;
            u32 set_before = 0;
            for (int i = 0; i < d.B; i++) {
                const bool committed = !((need >> i) & 1u);  // by an earlier batch
                if (committed) set_before |= 1u << i;
                if (committed && !d.scan_pos) continue;
                const u64 pos = dup_bit(d, g, i);
                if (d.scan_pos) d.scan_pos[(size_t)g * d.B + i] = pos;
                if (committed) continue;
                const u64 key = dup_key(i, pos);
                u32 slot = dup_slot(key, d.table_log2);
                u64 cur;
                for (;;) {
                    cur = d.table[slot];
                    if ((cur >> DUP_IDX_BITS) == key) break;
                    slot = (slot + 1) & tmask;
                }
                const int first = (int)(cur & ((1ull << DUP_IDX_BITS) - 1));
                if (first < g) set_before |= 1u << i;
                else is_dup = false;  // nobody earlier in this batch set it
                g_atomic_or_u32(&d.bitmap[(size_t)i * words + (pos >> 5)], 1u << (pos & 31));
            }
            if (d.scan_mask) {
                d.scan_mask[g] = (u8)set_before;
                is_dup = false;  // pass 1 decides nothing
            } else if (d.dupflag) {
                d.dupflag[g] = is_dup ? 1 : 0;
            } else if (is_dup) {
                d.res[0][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
                if (d.paired) d.res[1][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
            }
        }
        // one same-address device atomic per WORKGROUP (they serialize at ~10 ns each): lanes -> ballot,
        // waves -> LDS counter, workgroup -> global
        const u64 m = ballot(is_dup);
        if (lane_id() == 0 && m) lds_add_u32(block_count, (u32)popc64(m));
    }
    block_sync();
    if (d.scan_mask) return;
    if (thread_id() == 0 && *block_count) g_atomic_add_i64(d.ctr_dups, (int64_t)*block_count);
    if (gid == 0) g_atomic_add_i64(d.ctr_total, (int64_t)d.n);
}

// ---------------------------------------------------------------------------
// The same sequential semantics with one scattered atomic per (unit, buffer) instead of four scattered accesses:
//   claim   : old = atomic_or(bit).  Exactly one unit of the launch sees a clear bit - it WON the bit (in execution
//             order, which is not input order).  A unit that lost registers (bit -> its index, min) in the table
//             and marks the key in a small filter; on fresh data almost nobody loses.
//   winners : a unit that won looks its key up only when the filter says somebody lost it: if the smallest loser
//             is earlier in the input than the winner, the winner was not first - it counts the bit as set, and
//             flags the table entry so that the smallest loser knows it was.
//   finish  : a lost bit counts as set unless the unit is the smallest loser of a flagged entry (it was first; a
//             bit set before this launch has no winner, hence no flag).  Then Duplicate's decision as before.
// Per bit with same-launch units S and input-order first g1: every unit of S \ {g1} ends with "set", g1 with
// "set" iff the bit was set before the launch - what the sequential loop gives (duplicate.cpp:122-163).
// ---------------------------------------------------------------------------
enum { DUP_CF_LOG2 = 23 };
constexpr u64 DUP_FLAG = 1ull << 63;
FQ_DEV u32 dup_cf_bit(u64 key) { return (u32)((key * 0xD6E8FEB86659FD93ull) >> (64 - DUP_CF_LOG2)); }

FQ_DEV void dup_claim_body(const DupArgs& d) {
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    const u64 words = d.bits >> 5;
    const u32 tmask = (1u << d.table_log2) - 1u;
    for (int g = gid; g < d.n; g += gstride) {
        u32 won = 0;
        for (int i = 0; i < d.B; i++) {
            const u64 pos = dup_bit(d, g, i);
            if (d.scan_pos) d.scan_pos[(size_t)g * d.B + i] = pos;
            const u32 bit = 1u << (pos & 31);
            const u32 old = g_atomic_or_u32(&d.bitmap[(size_t)i * words + (pos >> 5)], bit);
            if (!(old & bit)) {
                won |= 1u << i;
                continue;
            }
            const u64 key = dup_key(i, pos);
            const u32 cf = dup_cf_bit(key);
            g_atomic_or_u32(&d.cfilter[cf >> 5], 1u << (cf & 31));
            const u64 entry = (key << DUP_IDX_BITS) | (u64)g;
            u32 slot = dup_slot(key, d.table_log2);
            for (;;) {
                u64 cur = g_atomic_cas_u64(&d.table[slot], ~0ull, entry);
                if (cur == ~0ull) break;
                if ((cur >> DUP_IDX_BITS) == key) {
                    g_atomic_min_u64(&d.table[slot], entry);
                    break;
                }
                slot = (slot + 1) & tmask;
            }
        }
        d.need[g] = (u8)won;
    }
}

// after a launch whose fused kernel did the claiming: the units that lost a bit register in the table and the filter
FQ_DEV void dup_losers_body(const DupArgs& d) {
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    const u32 tmask = (1u << d.table_log2) - 1u;
    const u32 all = (1u << d.B) - 1u;
    for (int g = gid; g < d.n; g += gstride) {
        u32 lost = all & ~(u32)d.need[g];
        while (lost) {
            const int i = ffs32(lost) - 1;
            lost &= lost - 1u;
            const u64 key = dup_key(i, dup_bit(d, g, i));
            const u32 cf = dup_cf_bit(key);
            g_atomic_or_u32(&d.cfilter[cf >> 5], 1u << (cf & 31));
            const u64 entry = (key << DUP_IDX_BITS) | (u64)g;
            u32 slot = dup_slot(key, d.table_log2);
            for (;;) {
                u64 cur = g_atomic_cas_u64(&d.table[slot], ~0ull, entry);
                if (cur == ~0ull) break;
                if ((cur >> DUP_IDX_BITS) == key) {
                    g_atomic_min_u64(&d.table[slot], entry);
                    break;
                }
                slot = (slot + 1) & tmask;
            }
        }
    }
}

// the table entry of `key`, or ~0 when nobody lost it
FQ_DEV u64 dup_find(const DupArgs& d, u64 key, u32& slot_out) {
    const u32 tmask = (1u << d.table_log2) - 1u;
    u32 slot = dup_slot(key, d.table_log2);
    for (;;) {
        const u64 cur = d.table[slot];
        if (cur == ~0ull) return ~0ull;
        if (((cur & ~DUP_FLAG) >> DUP_IDX_BITS) == key) {
            slot_out = slot;
            return cur;
        }
        slot = (slot + 1) & tmask;
    }
}

FQ_DEV void dup_winners_body(const DupArgs& d) {
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    for (int g = gid; g < d.n; g += gstride) {
        const u32 won = d.need[g];
        u32 setw = 0;
        for (int i = 0; i < d.B; i++) {
            if (!((won >> i) & 1u)) continue;
            const u64 key = dup_key(i, dup_bit(d, g, i));
            const u32 cf = dup_cf_bit(key);
            if (!((d.cfilter[cf >> 5] >> (cf & 31)) & 1u)) continue;   // nobody lost this bit
            u32 slot = 0;
            const u64 e = dup_find(d, key, slot);
            if (e == ~0ull) continue;                                   // a filter collision
            if ((int)(e & ((1ull << DUP_IDX_BITS) - 1)) < g) {
                setw |= 1u << i;
                __hip_atomic_fetch_or(&d.table[slot], DUP_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        d.setw[g] = (u8)setw;
    }
}

FQ_DEV void dup_finish_body(const DupArgs& d, u32* block_count) {
    if (thread_id() == 0) *block_count = 0;
    block_sync();
    const int gid = block_id() * block_threads() + thread_id();
    const int gstride = grid_blocks() * block_threads();
    const int rounds = (d.n + gstride - 1) / gstride;
    const u32 all = (1u << d.B) - 1u;
    for (int it = 0; it < rounds; it++) {
        const int g = gid + it * gstride;
        bool is_dup = false;
        if (g < d.n) {
            const u32 won = d.need[g];
            u32 set_before = d.setw[g];
            u32 lost = all & ~won;
            while (lost) {
                const int i = ffs32(lost) - 1;
                lost &= lost - 1u;
                u32 slot = 0;
                const u64 e = dup_find(d, dup_key(i, dup_bit(d, g, i)), slot);
                const bool first = (int)(e & ((1ull << DUP_IDX_BITS) - 1)) == g && (e & DUP_FLAG);
                if (!first) set_before |= 1u << i;
            }
            is_dup = set_before == all;
            if (d.scan_mask) {
                d.scan_mask[g] = (u8)set_before;
                is_dup = false;  // pass 1 decides nothing
            } else if (d.dupflag) {
                d.dupflag[g] = is_dup ? 1 : 0;
            } else if (is_dup) {
                d.res[0][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
                if (d.paired) d.res[1][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
            }
        }
        const u64 m = ballot(is_dup);
        if (lane_id() == 0 && m) lds_add_u32(block_count, (u32)popc64(m));
    }
    block_sync();
    if (d.scan_mask) return;
    if (thread_id() == 0 && *block_count) g_atomic_add_i64(d.ctr_dups, (int64_t)*block_count);
    if (gid == 0) g_atomic_add_i64(d.ctr_total, (int64_t)d.n);
}

// duplicate = AND_i (bit i set earlier in this shard OR set by a preceding shard)   
FQ_DEV void dup_final_body(const DupFinalArgs& d, u32* block_count) {
    if (thread_id() == 0) *block_count = 0;
    block_sync();
    const int g = block_id() * block_threads() + thread_id();
    const u64 words = d.bits >> 5;
    bool is_dup = false;
    if (g < d.n) {
        const u32 mask = d.scan_mask[g];
        is_dup = true;
        for (int i = 0; i < d.B; i++) {
            if ((mask >> i) & 1u) continue;
            bool set = false;
            if (d.prefix) {
                const u64 pos = d.scan_pos[(size_t)g * d.B + i];
                set = ((d.prefix[(size_t)i * words + (pos >> 5)] >> (pos & 31)) & 1u) != 0;
            }
            if (!set) { is_dup = false; break; }
        }
        if (d.dupflag) {
            d.dupflag[g] = is_dup ? 1 : 0;
        } else if (is_dup) {
            d.res[0][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
            if (d.paired) d.res[1][(size_t)g * 3 + 1] |= (u32)RS_DUP << 8;
        }
    }
    const u64 m = ballot(is_dup);
    if (lane_id() == 0 && m) lds_add_u32(block_count, (u32)popc64(m));
    block_sync();
    if (thread_id() == 0 && *block_count) g_atomic_add_i64(d.ctr_dups, (int64_t)*block_count);
    if (g == 0) g_atomic_add_i64(d.ctr_total, (int64_t)d.n);
}

FQ_DEV void or_images_body(const OrArgs& o) {
    const u64 stride = (u64)grid_blocks() * block_threads();
    for (u64 c = (u64)block_id() * block_threads() + thread_id(); c < o.chunks; c += stride) {
        u32x4 v;
        v.x = v.y = v.z = v.w = 0u;
        for (int k = 0; k < o.n_images; k++) {
            u32x4* at = o.images + (u64)k * o.chunks + c;
            const u32x4 w = *at;
            if (!o.dst) *at = v;
            v.x |= w.x; v.y |= w.y; v.z |= w.z; v.w |= w.w;
        }
        if (o.dst) o.dst[c] = v;
    }
}


// ---------------------------------------------------------------------------
// Overrepresentation analysis (Stats::statRead stats.cpp:270-288), after the fused kernel.
// Every `sampling`-th read a Stats object sees is analysed; "sees" is a position in the run's
// read stream: the unit index for the pre-filtering Stats, the rank among the units that are
// written out for the post-filtering ones.  Four small kernels:
//   ovr_pass  : which units reach the post-filtering Stats (from the result records) -> count per
//               256-unit block
//   ovr_scan  : running rank at the start of every block, continued across launches
//   ovr_tasks : the sampled (unit, mate, pre/post) reads, appended to a dense task list
//   ovr_count : one lane per task: the five step lengths slide over the read with a rolling
//               hash; a window whose key is in the seed table (and compares equal) is a hit
// ---------------------------------------------------------------------------
FQ_DEV bool ovr_unit_passes(const OvrArgs& o, int g) {
    // the unit is routed to out1[/out2] and statRead by the post-filtering Stats
    // (peprocessor.cpp:575-591, seprocessor.cpp:280-286)
    const u32 w1 = o.res[0][(size_t)g * 3 + 1];
    u32 code = w1 & 0xFFu, flags = (w1 >> 8) & 0xFFu;
    if (o.paired) {
        const u32 w2 = o.res[1][(size_t)g * 3 + 1];
        code |= w2 & 0xFFu;
        flags |= (w2 >> 8) & (u32)RS_NULL;
    }
    if (flags & RS_NULL) return false;
    if (o.dedup && (flags & RS_DUP)) return false;
    return code == 0u;
}

// what a post-filtering task reads
enum { OVR_SRC_MATE = 0, OVR_SRC_MERGED = 1, OVR_SRC_R1 = 2, OVR_SRC_R2 = 3 };

// the reads of unit g that reach the post-filtering Stats, in statRead order: 0..2 of them.  Outside merge
// mode both mates of a passing unit are analysed at the same stream position (one each for the two Stats).
FQ_DEV int ovr_post_reads(const OvrArgs& o, int g, u32 src[2]) {
    if (!o.merge) {
        src[0] = OVR_SRC_MATE;
        return ovr_unit_passes(o, g) ? 1 : 0;
    }
    const u32 w1 = o.res[0][(size_t)g * 3 + 1], w2 = o.res[1][(size_t)g * 3 + 1];
    const u32 code1 = w1 & 0xFFu, code2 = w2 & 0xFFu;
    const u32 flags1 = (w1 >> 8) & 0xFFu, flags2 = (w2 >> 8) & 0xFFu;
    if ((flags1 | flags2) & RS_NULL) return 0;  // merge.enabled && r1 && r2 (:521); otherwise no post-filtering Stats in merge mode (:584)
    const bool dedup_out = o.dedup && (flags1 & RS_DUP);
    const u32 pflags = o.pair[(size_t)g * 2 + 1] >> 16;
    if (pflags & 1u) {  // FASTP_GPU_PF_OVERLAPPED: the merged read, if it passes (:525-537)
        src[0] = OVR_SRC_MERGED;
        return code1 == 0u ? 1 : 0;
    }
    if (!o.merge_include_unmerged) return 0;
    int k = 0;  // :538-560
    if (code1 == 0u && !dedup_out) src[k++] = OVR_SRC_R1;
    if (code2 == 0u && !dedup_out) src[k++] = OVR_SRC_R2;
    return k;
}

FQ_DEV void ovr_pass_body(const OvrArgs& o, u32* lds) {
    if (thread_id() == 0) lds[0] = 0;
    block_sync();
    const int g = block_id() * block_threads() + thread_id();
    u32 src[2];
    const int k = g < o.n ? ovr_post_reads(o, g, src) : 0;
    const u32 c = (u32)popc64(ballot(k >= 1)) + (u32)popc64(ballot(k == 2));
    if (lane_id() == 0 && c) lds_add_u32(&lds[0], c);
    block_sync();
    if (thread_id() == 0) o.blocksum[block_id()] = lds[0];
}

FQ_DEV void ovr_scan_body(const OvrArgs& o, int nblocks, u32* lds) {
    // one workgroup: every lane sums a contiguous run of block counts, the lane totals are scanned through LDS,
    // then each lane writes the running stream position (mod sampling) at the start of each of its blocks
    if (block_id() != 0) return;
    const int tid = thread_id(), nt = block_threads();
    const int per = (nblocks + nt - 1) / nt;
    const int b0 = imin(tid * per, nblocks), b1 = imin(b0 + per, nblocks);
    u32 sum = 0;
    for (int b = b0; b < b1; b++) sum += o.blocksum[b];
    lds[tid] = sum;
    block_sync();
    const u64 seen = *o.post_seen;
    block_sync();
    if (tid == 0) {
        u32 run = 0;
        for (int i = 0; i < nt; i++) {
            const u32 v = lds[i];
            lds[i] = run;
            run += v;
        }
        *o.post_seen = seen + run;
    }
    block_sync();
    u64 run = seen + lds[tid];
    for (int b = b0; b < b1; b++) {
        o.blockbase[b] = (u32)(run % (u64)o.sampling);
        run += o.blocksum[b];
    }
}

// task = unit << 4 | source << 2 | Stats slot (PRE1=0 POST1=1 PRE2=2 POST2=3)
FQ_DEV void ovr_tasks_body(const OvrArgs& o, u32* lds) {
    const int g = block_id() * block_threads() + thread_id();
    u32 src[2] = {0u, 0u};
    const int k = g < o.n ? ovr_post_reads(o, g, src) : 0;
    // stream position of this unit's first post-filtering read: block base + waves before + lanes before
    const u64 m1 = ballot(k >= 1), m2 = ballot(k == 2);
    const int wave = wave_id(), nw = block_threads() >> 6;
    if (lane_id() == 0) lds[wave] = (u32)popc64(m1) + (u32)popc64(m2);
    block_sync();
    u32 before = 0;
    for (int w = 0; w < nw; w++)
        if (w < wave) before += lds[w];
    const u64 lower = (1ull << lane_id()) - 1ull;
    before += (u32)popc64(m1 & lower) + (u32)popc64(m2 & lower);
    const int mates = o.paired ? 2 : 1;
    const bool pre = g < o.n && (o.pre_mod + (u32)g) % (u32)o.sampling == 0u;  // mReads % sampling == 0 (:272)
    bool post[2];
    int npost = 0;
    for (int j = 0; j < 2; j++) {
        post[j] = g < o.n && j < k && (o.blockbase[block_id()] + before + (u32)j) % (u32)o.sampling == 0u;
        npost += post[j] ? 1 : 0;
    }
    const int total = (pre ? mates : 0) + (o.merge ? npost : npost * mates);   // 0 .. 6
    // The list's order is free: the workgroup takes the slots of all its tasks with ONE atomic (a returning atomic per sampled
    // unit on one word was 0.36 ms of configs[4]'s 4.4 ms step - 2 M pairs, every twentieth sampled twice).  A lane's offset:
    // the totals' bit planes as ballots.
    const u64 t0 = ballot((total & 1) != 0), t1 = ballot((total & 2) != 0), t2 = ballot((total & 4) != 0);
    if (lane_id() == 0) lds[4 + wave] = (u32)popc64(t0) + 2u * (u32)popc64(t1) + 4u * (u32)popc64(t2);
    block_sync();
    if (thread_id() == 0) {
        u32 sum = 0;
        for (int w = 0; w < nw; w++) sum += lds[4 + w];
        lds[8] = sum ? g_atomic_add_u32(o.n_tasks, sum) : 0u;
    }
    block_sync();
    if (!total) return;
    u32 slot = lds[8] + (u32)popc64(t0 & lower) + 2u * (u32)popc64(t1 & lower) + 4u * (u32)popc64(t2 & lower);
    for (int w = 0; w < nw; w++)
        if (w < wave) slot += lds[4 + w];
    auto put = [&](u32 t) {
        if (slot < (u32)o.task_cap) o.tasks[slot] = t;
        slot++;
    };
    if (pre)
        for (int mt = 0; mt < mates; mt++) put(((u32)g << 4) | ((u32)mt << 1));
    for (int j = 0; j < 2; j++) {
        if (!post[j]) continue;
        if (o.merge) put(((u32)g << 4) | (src[j] << 2) | 1u);
        else
            for (int mt = 0; mt < mates; mt++) put(((u32)g << 4) | ((u32)mt << 1) | 1u);
    }
}

FQ_DEV u32 ovr_sym(const u32* srow, const u8* qrow, int j) {  // 0..3 = A,T,C,G ; 4 = N
    return (qrow[j] & 0x80u) ? 4u : ((srow[j >> 4] >> ((j & 15) * 2)) & 3u);
}

// thread the correction entries that belong to this launch's reads into per-read chains
FQ_DEV void ovr_corr_link_body(const OvrArgs& o) {
    const int e = block_id() * block_threads() + thread_id();
    const int ne = (int)imin(*o.n_corr, o.corr_cap);
    if (e >= ne) return;
    const u32 read = o.corr[2 * (size_t)e];          // pair * 2 + mate (PE) / read (SE), from the batch start
    const int unit = (int)(o.paired ? read >> 1 : read) - o.first;
    if (unit < 0 || unit >= o.n) return;
    const u32 key = o.paired ? (u32)unit * 2u + (read & 1u) : (u32)unit;
    o.corr_next[e] = g_atomic_exch_u32(&o.corr_head[key], (u32)e + 1u);
}

// the same over a list whose capacity is far larger than its fill (the engine's own list of -c on the lane plan): grid-stride
FQ_DEV void corr_link_stride_body(const OvrArgs& o) {
    const int ne = (int)imin(*o.n_corr, o.corr_cap);
    for (int e = block_id() * block_threads() + thread_id(); e < ne; e += grid_blocks() * block_threads()) {
        const u32 read = o.corr[2 * (size_t)e];
        const int unit = (int)(o.paired ? read >> 1 : read) - o.first;
        if (unit < 0 || unit >= o.n) continue;
        const u32 key = o.paired ? (u32)unit * 2u + (read & 1u) : (u32)unit;
        o.corr_next[e] = g_atomic_exch_u32(&o.corr_head[key], (u32)e + 1u);
    }
}

// symbol j of a read as the post-filtering Stats see it: BaseCorrector's edits applied (basecorrector.cpp:45-63)
FQ_DEV u32 ovr_sym_corrected(const OvrArgs& o, const u32* srow, const u8* qrow, int j, u32 head) {
    u32 s = ovr_sym(srow, qrow, j);
    for (u32 e = head; e; e = o.corr_next[e - 1]) {
        const u32 w1 = o.corr[2 * (size_t)(e - 1) + 1];  // u16 pos | u8 base | u8 qual
        if ((int)(w1 & 0xFFFFu) == j) {
            const u32 b = (w1 >> 16) & 0xFFu;            // ASCII; engine code order A0 T1 C2 G3
            s = b == 'A' ? 0u : b == 'T' ? 1u : b == 'C' ? 2u : b == 'G' ? 3u : 4u;
        }
    }
    return s;
}

// a read as one of the Stats objects sees it
struct OvrRead {
    const u32* srow[2];
    const u8* qrow[2];
    const u8* raw[2];   // the mate's sequence text when the unit holds letters outside ACGTN (else null)
    u32 chain[2];   // correction chains of the two mates (0 = none)
    int f[2];       // merged: front of r1 / index of r2'[last] ; plain: f[0] = front
    int m1;         // merged: bases taken from r1
    int ol;         // merged: overlap length
    int len;
    int src;
    int mt;         // plain: which mate
};
// the text of a unit with letters outside ACGTN: the byte at `at` of mate mt with BaseCorrector's edits applied
FQ_DEV u32 ovr_raw_byte(const OvrArgs& o, const OvrRead& r, int mt, int at) {
    u32 b = r.raw[mt][at];
    for (u32 e = r.chain[mt]; e; e = o.corr_next[e - 1]) {
        const u32 w1 = o.corr[2 * (size_t)(e - 1) + 1];
        if ((int)(w1 & 0xFFFFu) == at) b = (w1 >> 16) & 0xFFu;
    }
    return b;
}
// A T C G N as the packed rows give them (0..4); any other byte is its own symbol, as in the seed tables (fq_host.cpp)
FQ_DEV u32 ovr_byte_sym(u32 b) { return b == 'A' ? 0u : b == 'T' ? 1u : b == 'C' ? 2u : b == 'G' ? 3u : b == 'N' ? 4u : b; }
FQ_DEV u32 ovr_byte_complement(u32 b) {   // util.h:16-33
    switch (b) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}
// the raw sequence text of unit g's mate m, or null (binary search in the batch's list)
FQ_DEV const u8* ovr_raw(const OvrArgs& o, int g, int m) {
    const int unit = o.first + g;
    int lo = 0, hi = o.x_n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (o.x_unit[mid] < unit) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= o.x_n || o.x_unit[lo] != unit) return nullptr;
    return o.x_text[m] + (o.x_dense ? o.x_off[m][4 * (size_t)unit + 1] : o.x_off[m][lo]);
}

FQ_DEV u32 ovr_fetch(const OvrArgs& o, const OvrRead& r, int j) {
    if (r.src != OVR_SRC_MERGED || j < r.m1) {
        const int mt = r.src == OVR_SRC_MERGED ? 0 : r.mt;
        const int at = r.f[0] + j;
        if (r.raw[mt]) return ovr_byte_sym(ovr_raw_byte(o, r, mt, at));
        return r.chain[mt] ? ovr_sym_corrected(o, r.srow[mt], r.qrow[mt], at, r.chain[mt]) : ovr_sym(r.srow[mt], r.qrow[mt], at);
    }
    // merged tail: rc(r2')[ol + k], k = j - m1  =  complement of r2'[len2 - 1 - ol - k]   (overlapanalysis.cpp:148-179)
    const int at = r.f[1] - r.ol - (j - r.m1);
    if (r.raw[1]) return ovr_byte_sym(ovr_byte_complement(ovr_raw_byte(o, r, 1, at)));
    const u32 s = r.chain[1] ? ovr_sym_corrected(o, r.srow[1], r.qrow[1], at, r.chain[1]) : ovr_sym(r.srow[1], r.qrow[1], at);
    return s < 4u ? (s ^ 1u) : s;  // A0<->T1, C2<->G3
}

FQ_DEV void ovr_count_body(const OvrArgs& o, u32* lds) {
    // the seed hash tables into LDS (every lane probes them at every window position)
    const int tid = thread_id();
    for (int m = 0; m < 2; m++)
        if (o.table_lds[m] >= 0)
            for (u32 i = (u32)tid; i < 2u * (o.mate[m].table_mask + 1u); i += (u32)block_threads()) lds[o.table_lds[m] + (int)i] = o.mate[m].table[i];
    block_sync();
    // lane = (task, window length): the five lengths of a read run side by side on the read's symbols in LDS (one lane
    // per read walked 5 x len windows in turn with two wavefronts per SIMD to hide its LDS round trips)
    const int tk = tid / OVR_STEPS, my_step = tid - tk * OVR_STEPS;
    const int t = block_id() * OVR_TPB + tk;
    const int nt = (int)imin((int)*o.n_tasks, o.task_cap);
    const bool live = tk < OVR_TPB && t < nt;
    const u32 task = live ? o.tasks[t] : 0u;
    const int g = (int)(task >> 4), src = (int)((task >> 2) & 3u), slot = (int)(task & 3u);
    const int post = slot & 1;
    const int mt = src == OVR_SRC_R1 ? 0 : src == OVR_SRC_R2 ? 1 : (slot >> 1);
    const OvrMate& M = o.mate[slot >> 1];   // the seed list belongs to the Stats object, not to the read
    const bool work = live && M.n_seeds != 0;
    OvrRead r;
    r.src = src;
    r.mt = mt;
    for (int m = 0; m < (o.paired ? 2 : 1); m++) {
        r.srow[m] = o.seq[m] + (size_t)g * o.sw_g;
        r.qrow[m] = (const u8*)(o.qual[m] + (size_t)g * o.qw_g);
        // the pre-filtering Stats saw the read before BaseCorrector touched it (peprocessor.cpp:419-432 vs :447-460)
        r.chain[m] = (post && o.corr) ? o.corr_head[o.paired ? (u32)g * 2u + (u32)m : (u32)g] : 0u;
        r.raw[m] = (work && o.x_n) ? ovr_raw(o, g, m) : nullptr;
    }
    if (!o.paired) { r.srow[1] = r.srow[0]; r.qrow[1] = r.qrow[0]; r.chain[1] = 0u; r.raw[1] = r.raw[0]; }
    r.m1 = 0;
    r.ol = 0;
    r.f[0] = r.f[1] = 0;
    int len = work ? (int)o.len[mt][g] : 0;
    if (!work) {
    } else if (src == OVR_SRC_MERGED) {
        const u32 a0 = o.res[0][(size_t)g * 3], b0 = o.res[1][(size_t)g * 3];
        // the merged parts from the pair record (merge mode's own analysis): len1 = overlap_len + max(0, offset), len2 =
        // offset > 0 ? len(r2') - overlap_len : 0 (overlapanalysis.cpp:152-156) - the records' reserved fields hold the same
        // numbers unless --overlapped_out occupies them
        const u32 pw = o.pair[(size_t)g * 2];
        const int pol = (int)(pw >> 16), poff = (int)(int16_t)(pw & 0xFFFFu);
        const int m1 = pol + imax(0, poff), m2 = poff > 0 ? (int)(b0 >> 16) - pol : 0;
        r.f[0] = (int)(a0 & 0xFFFFu);
        r.f[1] = (int)(b0 & 0xFFFFu) + (int)(b0 >> 16) - 1;  // r2'[last]
        r.m1 = m1;
        r.ol = (int)(o.pair[(size_t)g * 2] >> 16);
        len = m1 + m2;
    } else if (post) {  // the read as it is written out: [front, front + len)
        const u32 w0 = o.res[mt][(size_t)g * 3];
        r.f[0] = (int)(w0 & 0xFFFFu);
        len = (int)(w0 >> 16);
    }
    r.len = len;
    // the read's symbols once into LDS, [position][lane]: the five window lengths below then slide over LDS bytes
    // instead of re-reading (and re-correcting / re-complementing) global memory twice per position
    u8* symv = (u8*)lds + tk;
    const bool staged = o.sym_cap > 0;   // (uniform; launch_overrep sizes the rows for the longest read or not at all)
    if (staged) {
        // the plain case (a mate's packed row, no correction chain, no text): eight symbols' loads in flight at a time - as one
        // ovr_fetch per trip every symbol was a round trip to memory of its own, fifty in a row for a lane of a 250-base read
        const bool simple = src != OVR_SRC_MERGED && !r.raw[mt] && !r.chain[mt];
        if (ballot(work && !simple) == 0ull) {   // (wave-uniform)
            const u32* srow = r.srow[mt];
            const u8* qrow = r.qrow[mt];
            for (int j0 = my_step; j0 < len; j0 += 8 * OVR_STEPS) {
                u32 w[8], qb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = j0 + u * OVR_STEPS, at = r.f[0] + imin(j, len - 1);
                    w[u] = srow[at >> 4];
                    qb[u] = qrow[at];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int j = j0 + u * OVR_STEPS, at = r.f[0] + j;
                    if (j < len) symv[(size_t)j * OVR_SYM_STRIDE] = (u8)((qb[u] & 0x80u) ? 4u : ((w[u] >> ((at & 15) * 2)) & 3u));
                }
            }
        } else {
            for (int j = my_step; j < len; j += OVR_STEPS) symv[(size_t)j * OVR_SYM_STRIDE] = (u8)ovr_fetch(o, r, j);
        }
        block_sync();
    }
    if (!work) return;
#define OVR_SYM(j) (staged ? (u32)symv[(size_t)(j) * OVR_SYM_STRIDE] : ovr_fetch(o, r, (j)))
    const u32* tab = o.table_lds[slot >> 1] >= 0 ? lds + o.table_lds[slot >> 1] : M.table;
    const int f = 0;
    int64_t* cnt = o.ctr + o.o_count[slot];
    int64_t* dist = o.ctr + o.o_dist[slot];
    if (staged && o.table_lds[0] >= 0 && (o.table_lds[1] >= 0 || o.mate[1].n_seeds == 0)) {   // (uniform)
        // Round 6: symbols and seed tables are both in LDS (the usual case) - read through DS addresses (the generic pointers of
        // the form below are FLAT loads: 6.3e7 of them per launch, and 5.9e8 SALU + 2.3e8 branch instructions against 4.0e8 VALU,
        // profiles/r06_i_config4_sq_counters.txt: the kernel - configs[4]'s critical path behind the lane kernel - was bound by
        // instruction issue, most of it exec-mask bookkeeping around the per-position guards).  The symbol rows are four
        // positions longer than the longest read, so the four slides of a trip read unguarded; positions behind the loop's range
        // are masked out of ONE "any first slot occupied" test, the only branch of a trip without a hit.
        const int s = my_step;
        const int L = M.steps[s];
        if (L <= 0 || len - L <= 0) return;
        const u32 salt = (u32)L * OVR_SALT_MUL, pw = M.pw[s];
        const u32 sym0 = lds_addr_of(symv);
        const u32 tab0 = lds_addr_of(lds + o.table_lds[slot >> 1]);
#define SYMF(j) lds_read_u8_at(sym0 + (u32)(j) * (u32)OVR_SYM_STRIDE)
#define TABF(w) lds_read_u32_at(tab0 + 4u * (u32)(w))
        int i = 0;
        u32 h = 0;
        bool fresh = true;
        while (i < len - L) {  // for (i = 0; i < len - step; i++) (:276)
            if (fresh) {
                h = 0;
                for (int k = 0; k < L; k++) h = h * OVR_HASH_MUL + SYMF(i + k) + 1u;
                fresh = false;
            }
            const int nb = imin(4, len - L - i);
            u32 so[4], si[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                so[b] = SYMF(i + b);
                si[b] = SYMF(i + b + L);
            }
            u32 hh[5];
            hh[0] = h;
#pragma unroll
            for (int b = 0; b < 4; b++) hh[b + 1] = (hh[b] - (so[b] + 1u) * pw) * OVR_HASH_MUL + si[b] + 1u;
            u32 s0[4], id0[4], anyid = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                s0[b] = ((hh[b] ^ salt) * OVR_SALT_MUL) & M.table_mask;
                id0[b] = TABF(2u * s0[b] + 1u);
                anyid |= b < nb ? id0[b] : 0u;
            }
            int hit = -1, hit_b = 0;
            if (anyid != 0u) {   // an occupied first slot somewhere in the trip (rare): the probe as the table defines it
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if (b < nb && hit < 0 && id0[b] != 0u) {
                        const u32 key = hh[b] ^ salt;
                        for (u32 sl = s0[b];; sl = (sl + 1) & M.table_mask) {
                            const u32 id1 = TABF(2u * sl + 1u);
                            if (id1 == 0u) break;
                            if (TABF(2u * sl) == key && M.seed_len[id1 - 1] == L) {
                                const u8* sd = M.seed_sym + (size_t)(id1 - 1) * OVR_SEED_STRIDE;
                                bool same = true;
                                for (int k = 0; k < L && same; k++) same = sd[k] == (u8)SYMF(i + b + k);
                                if (same) { hit = (int)id1 - 1; hit_b = b; break; }
                            }
                        }
                    }
                }
            }
            if (hit >= 0) {  // mOverRepSeq[seq]++, the covered positions of mOverRepSeqDist, i += step (:279-284)
                const int at = i + hit_b;
                g_atomic_add_i64(&cnt[hit], 1);
                if (o.dist_diff[slot]) {
                    const int e0 = imin(at, M.eval_len), e1 = imin(at + L, M.eval_len);
                    if (e0 < e1) {
                        int* dd = o.dist_diff[slot] + (size_t)hit * (M.eval_len + 1);
                        g_atomic_add_i32(&dd[e0], 1);
                        g_atomic_add_i32(&dd[e1], -1);
                    }
                } else {
                    for (int pp = at; pp < at + L && pp < M.eval_len; pp++) g_atomic_add_i64(&dist[(size_t)hit * M.eval_len + pp], 1);
                }
                i = at + L + 1;
                fresh = true;
            } else {  // the window has slid by nb bases
                h = nb == 4 ? hh[4] : nb == 3 ? hh[3] : nb == 2 ? hh[2] : hh[1];
                i += nb;
            }
        }
#undef SYMF
#undef TABF
    } else {
        const int s = my_step;
        const int L = M.steps[s];
        if (L <= 0 || len - L <= 0) return;
        const u32 salt = (u32)L * OVR_SALT_MUL, pw = M.pw[s];
        int i = 0;
        u32 h = 0;
        bool fresh = true;  // h has to be computed from scratch for the window at i
        // Four positions per trip (round 5): the symbols the next three slides need and the first table slot of all four
        // windows are asked for together - independent LDS reads, two round trips per four positions instead of three per
        // position on a lane's critical path - then looked at in order; a hit (rare) throws the rest of the block away.
        while (i < len - L) {  // for (i = 0; i < len - step; i++) (:276)
            if (fresh) {
                h = 0;
                for (int k = 0; k < L; k++) h = h * OVR_HASH_MUL + OVR_SYM(f + i + k) + 1u;
                fresh = false;
            }
            const int nb = imin(4, len - L - i);   // positions i .. i + nb - 1 are inside the loop's range
            u32 so[4], si[4];                      // the symbols that leave / enter when the window slides from i + b to i + b + 1
#pragma unroll
            for (int b = 0; b < 4; b++) {
                so[b] = b < nb ? OVR_SYM(f + i + b) : 0u;
                si[b] = b < nb ? OVR_SYM(f + i + b + L) : 0u;   // (i + b + L <= len - 1)
            }
            u32 hh[5];
            hh[0] = h;
#pragma unroll
            for (int b = 0; b < 4; b++) hh[b + 1] = (hh[b] - (so[b] + 1u) * pw) * OVR_HASH_MUL + si[b] + 1u;
            u32 s0[4], id0[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                s0[b] = ((hh[b] ^ salt) * OVR_SALT_MUL) & M.table_mask;
                id0[b] = b < nb ? tab[2 * s0[b] + 1] : 0u;
            }
            int hit = -1, hit_b = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (b < nb && hit < 0 && id0[b] != 0u) {   // an occupied first slot: the probe as the table defines it
                    const u32 key = hh[b] ^ salt;
                    for (u32 sl = s0[b];; sl = (sl + 1) & M.table_mask) {
                        const u32 id1 = tab[2 * sl + 1];
                        if (id1 == 0u) break;
                        if (tab[2 * sl] == key && M.seed_len[id1 - 1] == L) {
                            const u8* sd = M.seed_sym + (size_t)(id1 - 1) * OVR_SEED_STRIDE;
                            bool same = true;
                            for (int k = 0; k < L && same; k++) same = sd[k] == (u8)OVR_SYM(f + i + b + k);
                            if (same) { hit = (int)id1 - 1; hit_b = b; break; }
                        }
                    }
                }
            }
            if (hit >= 0) {  // mOverRepSeq[seq]++, the covered positions of mOverRepSeqDist, i += step (:279-284)
                const int at = i + hit_b;
                g_atomic_add_i64(&cnt[hit], 1);
                if (o.dist_diff[slot]) {
                    const int e0 = imin(at, M.eval_len), e1 = imin(at + L, M.eval_len);
                    if (e0 < e1) {
                        int* dd = o.dist_diff[slot] + (size_t)hit * (M.eval_len + 1);
                        g_atomic_add_i32(&dd[e0], 1);
                        g_atomic_add_i32(&dd[e1], -1);
                    }
                } else {
                    for (int pp = at; pp < at + L && pp < M.eval_len; pp++) g_atomic_add_i64(&dist[(size_t)hit * M.eval_len + pp], 1);
                }
                i = at + L + 1;
                fresh = true;
            } else {  // the window has slid by nb bases
                h = nb == 4 ? hh[4] : nb == 3 ? hh[3] : nb == 2 ? hh[2] : hh[1];
                i += nb;
            }
        }
    }
}
#undef OVR_SYM


// one lane per (slot, seed) row of the difference arrays: running sums -> mOverRepSeqDist in the counter block, array cleared
FQ_DEV void ovr_dist_body(const OvrArgs& o) {
    int t = block_id() * block_threads() + thread_id();
    for (int slot = 0; slot < 4; slot++) {
        const OvrMate& M = o.mate[slot >> 1];
        if (!o.dist_diff[slot]) continue;
        if (t >= 0 && t < M.n_seeds) {
            int* dd = o.dist_diff[slot] + (size_t)t * (M.eval_len + 1);
            int64_t* dist = o.ctr + o.o_dist[slot] + (size_t)t * M.eval_len;
            int run = 0;
            for (int p = 0; p < M.eval_len; p++) {
                const int v = dd[p];
                if (v) dd[p] = 0;
                run += v;
                if (run) g_atomic_add_i64(&dist[p], (int64_t)run);
            }
            dd[M.eval_len] = 0;
        }
        t -= M.n_seeds;
    }
}

// ---------------------------------------------------------------------------
// FASTQ text -> packed batch: FastqReader::getLine / read
// (fastqreader.cpp:240-368) for a well-formed chunk resident in HBM.
//   parse_count : line terminators per 4 KiB block.  A terminator STARTS at every '\r' and at every
//                 '\n' that does not follow a '\r' ("\r\n" is one terminator, :257-259)
//   parse_scan  : terminators before each block
//   parse_index : position / length of the terminator that ends line k
//   parse_pack  : one wavefront per record (4 lines): validate, 2-bit pack, N flags, zero padding
// ---------------------------------------------------------------------------
// 0x80 in every byte of v that is zero (exact, no borrow across bytes)
FQ_DEV u32 zero_bytes(u32 v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }

// bit i: a line terminator starts at byte base + i of the 16 bytes v (prev = the byte before them).  Byte-parallel:
// '\r' always starts one, '\n' does unless the byte before it is '\r' (FastqReader::getLine, fastqreader.cpp:240-268)
FQ_DEV u32 parse_term_mask16(const ParseArgs& p, u32 base, const u32x4& v, u32 prev) {
    if (base >= p.nbytes) return 0;
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    u32 m = 0;
    u32 carry = prev == 13u ? 0x80u : 0u;  // "the byte before this dword is '\r'", positioned for its first byte
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 cr = zero_bytes(w[j] ^ 0x0D0D0D0Du), lf = zero_bytes(w[j] ^ 0x0A0A0A0Au);
        const u32 t = (cr | (lf & ~((cr << 8) | carry))) >> 7;  // bits 0, 8, 16, 24
        m |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu) << (4 * j);
        carry = cr >> 24;
    }
    const u32 left = p.nbytes - base;  // bytes of this chunk that exist
    if (left < 16u) m &= (1u << left) - 1u;
    // a '\r' that ends a non-final chunk may be half of a "\r\n": leave its line to the next chunk
    if (!p.is_last && left <= 16u && ((m >> (left - 1u)) & 1u) && p.text[p.nbytes - 1] == 13) m &= ~(1u << (left - 1u));
    return m;
}

// the workgroup's PARSE_SUB sub-blocks: lane t of sub-block k owns bytes (block * PARSE_SUB + k) * 4096 + 16 t .. +15.
// All loads are issued before any mask is built (four independent 16-byte loads in flight per lane).
FQ_DEV void parse_masks(const ParseArgs& p, u32 m[PARSE_SUB], u32 bases[PARSE_SUB]) {
    u32x4 v[PARSE_SUB];
    u32 prev[PARSE_SUB];
#pragma unroll
    for (int k = 0; k < PARSE_SUB; k++) {
        bases[k] = (((u32)block_id() * PARSE_SUB + (u32)k) * PARSE_BLOCK + (u32)thread_id()) * PARSE_BYTES_PER_LANE;
        const u32x4 z = {0u, 0u, 0u, 0u};
        const bool in = bases[k] < p.nbytes;
        v[k] = in ? *(const u32x4*)(p.text + bases[k]) : z;
        prev[k] = (in && bases[k]) ? (u32)p.text[bases[k] - 1] : 0u;
    }
#pragma unroll
    for (int k = 0; k < PARSE_SUB; k++) m[k] = parse_term_mask16(p, bases[k], v[k], prev[k]);
}

FQ_DEV void parse_count_body(const ParseArgs& p, u32* lds) {
    if (thread_id() == 0) lds[0] = 0;
    block_sync();
    u32 m[PARSE_SUB], bases[PARSE_SUB];
    parse_masks(p, m, bases);
    u32 wsum = 0;
#pragma unroll
    for (int k = 0; k < PARSE_SUB; k++) wsum += (u32)popc32(m[k]);
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) wsum += shfl_xor(wsum, sh);
    if (lane_id() == 0 && wsum) lds_add_u32(&lds[0], wsum);
    block_sync();
    if (thread_id() == 0) p.blockcount[block_id()] = lds[0];
}

FQ_DEV void parse_scan_body(const ParseArgs& p, int nblocks, u32* lds) {
    // one workgroup: every lane sums a contiguous run of block counts, the lane totals are
    // scanned through LDS, then each lane writes the running prefix of its run
    const int tid = thread_id(), nt = block_threads();
    const int per = (nblocks + nt - 1) / nt;
    const int b0 = imin(tid * per, nblocks), b1 = imin(b0 + per, nblocks);
    u32 sum = 0;
    for (int b = b0; b < b1; b++) sum += p.blockcount[b];
    lds[tid] = sum;
    block_sync();
    if (tid == 0) {
        u32 run = 0;
        for (int i = 0; i < nt; i++) {
            const u32 v = lds[i];
            lds[i] = run;
            run += v;
        }
        p.totals[0] = run;
        // an unterminated last line of the file still is a line (getLine's bufferFinished() branch)
        u32 lines = run;
        if (p.is_last && p.nbytes > 0) {
            const u32 last = (u32)p.text[p.nbytes - 1];
            if (last != 10u && last != 13u) lines++;
        }
        p.totals[2] = lines;
    }
    block_sync();
    u32 run = lds[tid];
    for (int b = b0; b < b1; b++) {
        p.blockbase[b] = run;
        run += p.blockcount[b];
    }
}

FQ_DEV void parse_index_body(const ParseArgs& p, u32* lds) {
    u32 m[PARSE_SUB], bases[PARSE_SUB];
    parse_masks(p, m, bases);
    const int wave = wave_id(), nw = block_threads() >> 6;
    u32 run = p.blockbase[block_id()];  // terminators before this workgroup's first sub-block
    for (int k = 0; k < PARSE_SUB; k++) {
        u32 mk = m[k];
        const u32 c = (u32)popc32(mk);
        // exclusive prefix of c inside the sub-block: lanes via shuffles, waves via LDS
        u32 incl = c;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const u32 o = shfl(incl, lane_id() - sh);
            if (lane_id() >= sh) incl += o;
        }
        if (lane_id() == 63) lds[wave] = incl;
        block_sync();
        u32 rank = run + incl - c, total = 0;
        for (int w = 0; w < nw; w++) {
            if (w < wave) rank += lds[w];
            total += lds[w];
        }
        block_sync();
        run += total;
        while (mk) {
            const int i = ffs32(mk) - 1;
            mk &= mk - 1;
            if (rank < p.max_lines) {
                const u32 pos = bases[k] + (u32)i;
                p.term_pos[rank] = pos;
                p.term_len[rank] = (u8)((p.text[pos] == 13 && pos + 1 < p.nbytes && p.text[pos + 1] == 10) ? 2 : 1);
            }
            rank++;
        }
    }
}

// totals[3] = records that will be packed, totals[4] = bytes they cover
FQ_DEV void parse_finish_body(const ParseArgs& p) {
    if (block_id() != 0 || thread_id() != 0) return;
    const u32 lines = p.totals[2], terms = p.totals[0];
    u32 nrec = lines / 4u;
    if (nrec > (u32)p.max_records) nrec = (u32)p.max_records;
    u32 consumed = 0;
    if (nrec) {
        const u32 k = 4u * nrec - 1u;  // last line of the last record
        consumed = k < terms ? p.term_pos[k] + p.term_len[k] : p.nbytes;
    }
    p.totals[3] = nrec;
    p.totals[4] = consumed;
}

enum { PACK_GROUP = 16 };  // lanes per record: four records per wavefront
enum { PARSE_BAD_MALFORMED = 1, PARSE_BAD_TOO_LONG = 2, PARSE_BAD_ALPHABET = 3 };  // = FASTP_GPU_PARSE_BAD_* (fastp_gpu.h)

FQ_DEV void parse_pack_body(const ParseArgs& p) {
    // 16 lanes per record; lane c packs bases 4c .. 4c+3 from ONE dword of the sequence line and one of the
    // quality line (four letters at a time, byte-parallel), three trips for a 150-base read
    const int lane = lane_id();
    const int grp = lane >> 4, gl = lane & (PACK_GROUP - 1);
    const int r = (block_id() * (block_threads() >> 6) + wave_id()) * 4 + grp;
    const u32 lines = p.totals[2], terms = p.totals[0];
    int nrec = (int)(lines / 4u);
    if (nrec > p.max_records) nrec = p.max_records;
    const bool have = r < nrec;
    const int rr = have ? r : 0;
    u32 start[4], len[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 k = 4u * (u32)rr + (u32)j;
        start[j] = (have && k) ? p.term_pos[k - 1] + p.term_len[k - 1] : 0u;
        const u32 end = (have && k < terms) ? p.term_pos[k] : p.nbytes;  // the unterminated last line ends at EOF
        len[j] = have ? end - start[j] : 0u;
    }
    bool bad = false;
    u32 kind = 0;  // why the record is refused: PARSE_BAD_* (the smallest bad record's reason reaches the host)
    if (have && gl == 0) {  // FastqReader::read's checks (:338-362)
        const bool malformed = len[0] == 0 || p.text[start[0]] != '@' || len[2] == 0 || p.text[start[2]] != '+' || len[1] != len[3];
        const bool too_long = len[1] > (u32)p.max_len;
        bad = malformed || too_long;
        kind = malformed ? (u32)PARSE_BAD_MALFORMED : too_long ? (u32)PARSE_BAD_TOO_LONG : 0u;
        if (!malformed) g_atomic_max_u32(&p.totals[5], len[1]);  // the longest sequence line: what a re-plan sizes max_len from
    }
    const u32 gmask_shift = 16u * (u32)grp;
    bad = ((ballot(bad) >> gmask_shift) & 0xFFFFull) != 0ull;
    const u32 L = bad ? 0u : len[1];
    if (have && gl < 4) {
        const u32 st = gl == 0 ? start[0] : gl == 1 ? start[1] : gl == 2 ? start[2] : start[3];
        const u32 ln = gl == 0 ? len[0] : gl == 1 ? len[1] : gl == 2 ? len[2] : len[3];
        p.line_off[4 * (size_t)rr + gl] = st;
        p.line_len[4 * (size_t)rr + gl] = ln;
    }
    if (have && gl == 0) p.len_out[rr] = (u16)L;
    const u8* sp = p.text + start[1];
    const u8* qp = p.text + start[3];
    u32* qrow = p.qual_out + (size_t)rr * p.qw_g;
    u8* srow = (u8*)(p.seq_out + (size_t)rr * p.sw_g);
    bool alpha_bad = false, foreign = false;
    const int ncol = p.qw_g > p.sw_g * 4 ? p.qw_g : p.sw_g * 4;
    for (int c = gl; c < ncol; c += PACK_GROUP) {
        u32 qd = 0, sb = 0;
        const int rem = (int)L - 4 * c;  // letters of this dword that exist
        if (have && rem > 0) {
            // unaligned dword reads; the up to three bytes behind the line are its terminator / the next line (the
            // text is padded behind its end) and are masked off
            u32 w, qw;
            __builtin_memcpy(&w, sp + 4 * c, 4);
            __builtin_memcpy(&qw, qp + 4 * c, 4);
            const u32 keep = lowmask32(8 * rem);
            w &= keep;
            qw &= keep;
            // ASCII bits 2:1 tell the letters apart: A 00, C 01, T 10, G 11 (N also 11); engine codes A0 T1 C2 G3 swap them
            const u32 x = (w >> 1) & 0x03030303u;
            const u32 code = ((x & 0x01010101u) << 1) | ((x >> 1) & 0x01010101u);
            // the letter each code stands for, to compare with: 'A' + {0, 0x13, 0x02, 0x06}
            const u32 c0 = code & 0x01010101u, c1 = (code >> 1) & 0x01010101u, both = c0 & c1;
            const u32 expect = 0x41414141u + (c1 << 1) + (c0 << 4) + (c0 << 1) + c0 - ((both << 4) - both);
            const u32 is_letter = zero_bytes(w ^ expect);
            const u32 is_n = zero_bytes(w ^ 0x4E4E4E4Eu);
            const u32 live = (0x80808080u & keep);
            // quality characters outside '!'..'~' are refused like foreign letters: the kernels take (q - 33) as an
            // unsigned field of a packed counter, the reference adds a negative long (stats.cpp:223,226)
            const u32 q_lt33 = ~(((qw & 0x7F7F7F7Fu) | 0x80808080u) - 0x21212121u) & 0x80808080u;
            const u32 q_127 = zero_bytes(qw ^ 0x7F7F7F7Fu);
            if ((qw & 0x80808080u) | ((q_lt33 | q_127) & live)) alpha_bad = true;
            // a letter outside ACGTN: the record is listed for the text kernel (fq_text.h) - its packed row is a placeholder
            const u32 fm = ~(is_letter | is_n) & live;
            if (fm) foreign = true;
            const u32 zero = is_n | fm;
            const u32 codes = code & ~(zero >> 7) & ~(zero >> 6);  // N (and a foreign letter) packs as code 0
            sb = (codes & 3u) | ((codes >> 6) & 0xCu) | ((codes >> 12) & 0x30u) | ((codes >> 18) & 0xC0u);
            qd = (qw & 0x7F7F7F7Fu) | is_n;
        }
        if (have && c < p.qw_g) qrow[c] = qd;
        if (have && c < p.sw_g * 4) srow[c] = (u8)sb;
    }
    const bool any_bad = ((ballot(alpha_bad) >> gmask_shift) & 0xFFFFull) != 0ull || bad;
    if (have && any_bad && gl == 0) g_atomic_min_u32(&p.totals[1], ((u32)r << 2) | (kind ? kind : (u32)PARSE_BAD_ALPHABET));
    const bool any_foreign = ((ballot(foreign) >> gmask_shift) & 0xFFFFull) != 0ull;
    if (have && any_foreign && !any_bad && gl == 0) {
        const u32 slot = g_atomic_add_u32(&p.totals[6], 1u);
        if (slot < p.exotic_cap) p.exotic_list[slot] = (u32)r;
    }
}


// ---------------------------------------------------------------------------
// Result records -> output FASTQ text: Read::appendToString (read.cpp:119-134)
// for the units routed to out1 [/ out2] (peprocessor.cpp:577-591, seprocessor.cpp:280-286).
//   fmt_len   : bytes each block's units add to each stream
//   fmt_scan  : running offsets
//   fmt_write : one wavefront per unit and mate copies the four lines
//   fmt_fix   : BaseCorrector's edits patched into the copied text
// ---------------------------------------------------------------------------
FQ_DEV bool fmt_unit_out(const FmtArgs& f, int g) {
    const u32 w1 = f.m[0].res[(size_t)g * 3 + 1];
    u32 code = w1 & 0xFFu, flags = (w1 >> 8) & 0xFFu;
    if (f.paired) {
        const u32 w2 = f.m[1].res[(size_t)g * 3 + 1];
        code |= w2 & 0xFFu;
        flags |= (w2 >> 8) & (u32)RS_NULL;
    }
    if (flags & RS_NULL) return false;
    if (f.dedup && (flags & RS_DUP)) return false;
    return code == 0u;
}
// name + '\n' + seq + '\n' + strand + '\n' + qual + '\n'
FQ_DEV u32 fmt_record_bytes(const FmtMate& M, int g) {
    const u32 len = M.res[(size_t)g * 3] >> 16;
    return M.line_len[4 * (size_t)g] + M.line_len[4 * (size_t)g + 2] + 2u * len + 4u;
}

FQ_DEV void fmt_len_body(const FmtArgs& f, u32* lds) {
    if (thread_id() < 2) lds[thread_id()] = 0;
    block_sync();
    const int g = block_id() * block_threads() + thread_id();
    const bool out = g < f.n && fmt_unit_out(f, g);
    for (int mt = 0; mt < (f.paired ? 2 : 1); mt++) {
        u32 b = out ? fmt_record_bytes(f.m[mt], g) : 0u;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) b += shfl_xor(b, sh);
        if (lane_id() == 0 && b) lds_add_u32(&lds[mt], b);
    }
    block_sync();
    if (thread_id() < 2) f.blocksum[(size_t)thread_id() * f.nblocks + block_id()] = lds[thread_id()];
}

FQ_DEV void fmt_scan_body(const FmtArgs& f, u64* lds) {
    // one workgroup per stream (block_id = stream): lanes sum runs of blocks, scan through LDS
    const int mt = block_id();
    const int tid = thread_id(), nt = block_threads();
    const int per = (f.nblocks + nt - 1) / nt;
    const int b0 = imin(tid * per, f.nblocks), b1 = imin(b0 + per, f.nblocks);
    const u64* sums = f.blocksum + (size_t)mt * f.nblocks;
    u64* base = f.blockbase + (size_t)mt * f.nblocks;
    u64 sum = 0;
    for (int b = b0; b < b1; b++) sum += sums[b];
    lds[tid] = sum;
    block_sync();
    if (tid == 0) {
        u64 run = 0;
        for (int i = 0; i < nt; i++) {
            const u64 v = lds[i];
            lds[i] = run;
            run += v;
        }
        f.totals[mt] = run;
    }
    block_sync();
    u64 run = lds[tid];
    for (int b = b0; b < b1; b++) {
        base[b] = run;
        run += sums[b];
    }
}

// one line of a record by a 16-lane group: four bytes per lane and trip (source and destination sit at arbitrary
// byte offsets: global memory takes unaligned dwords), the 0-3 byte tail and the '\n' by the group's first lane
FQ_DEV void fmt_copy(u8* dst, const u8* src, u32 n, int gl) {
    for (u32 i = 4u * (u32)gl; i + 4u <= n; i += 4u * 16u) {
        u32 w;
        __builtin_memcpy(&w, src + i, 4);
        __builtin_memcpy(dst + i, &w, 4);
    }
    if (gl == 0) {
        for (u32 i = n & ~3u; i < n; i++) dst[i] = src[i];
        dst[n] = 10;  // '\n'
    }
}

FQ_DEV void fmt_write_body(const FmtArgs& f, u32* lds) {
    // offsets of the block's units: block base + exclusive prefix of the record sizes (per stream)
    const int tid = thread_id();
    const int g0 = block_id() * block_threads();
    const int mates = f.paired ? 2 : 1;
    const int g = g0 + tid;
    const bool out = g < f.n && fmt_unit_out(f, g);
    for (int mt = 0; mt < mates; mt++) {
        const u32 b = out ? fmt_record_bytes(f.m[mt], g) : 0u;
        u32 incl = b;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const u32 o = shfl(incl, lane_id() - sh);
            if (lane_id() >= sh) incl += o;
        }
        const int wave = wave_id(), nw = block_threads() >> 6;
        if (lane_id() == 63) lds[mt * 16 + wave] = incl;
        block_sync();
        u64 off = f.blockbase[(size_t)mt * f.nblocks + block_id()] + (incl - b);
        for (int w = 0; w < nw; w++)
            if (w < wave) off += lds[mt * 16 + w];
        if (g < f.n) f.m[mt].unit_off[g] = out ? off : ~0ull;
        block_sync();
    }
    // copy: a 16-lane group per (unit, mate), looping over the block's units
    const int lane = lane_id() & 15, ngroups = block_threads() >> 4;
    for (int t = tid >> 4; t < block_threads() * mates; t += ngroups) {
        const int u = f.paired ? t >> 1 : t;
        const int gu = g0 + u;
        if (gu >= f.n) break;
        {
            const int mt = f.paired ? t & 1 : 0;
            const FmtMate& M = f.m[mt];
            const u64 off = M.unit_off[gu];
            if (off == ~0ull) continue;
            const u32 w0 = M.res[(size_t)gu * 3];
            const u32 front = w0 & 0xFFFFu, len = w0 >> 16;
            const u32 nl = M.line_len[4 * (size_t)gu], sl = M.line_len[4 * (size_t)gu + 2];
            if (off + nl + sl + 2u * len + 4u > M.out_cap) continue;  // the host sees the needed size in totals
            u8* o = M.out + off;
            fmt_copy(o, M.text + M.line_off[4 * (size_t)gu], nl, lane);
            o += nl + 1;
            fmt_copy(o, M.text + M.line_off[4 * (size_t)gu + 1] + front, len, lane);
            o += len + 1;
            fmt_copy(o, M.text + M.line_off[4 * (size_t)gu + 2], sl, lane);
            o += sl + 1;
            fmt_copy(o, M.text + M.line_off[4 * (size_t)gu + 3] + front, len, lane);
        }
    }
}

FQ_DEV void fmt_fix_body(const FmtArgs& f) {
    const int i = block_id() * block_threads() + thread_id();
    if (!f.corrections || !f.n_corrections || i >= *f.n_corrections) return;
    const u32 rd = f.corrections[2 * (size_t)i], w = f.corrections[2 * (size_t)i + 1];
    const int mt = f.paired ? (int)(rd & 1u) : 0;
    const int g = (int)(f.paired ? rd >> 1 : rd) - f.corr_first;
    if (g < 0 || g >= f.n) return;
    const FmtMate& M = f.m[mt];
    const u64 off = M.unit_off[g];
    if (off == ~0ull) return;
    const u32 w0 = M.res[(size_t)g * 3];
    const u32 front = w0 & 0xFFFFu, len = w0 >> 16, pos = w & 0xFFFFu;
    if (pos < front || pos >= front + len) return;  // the edited base was trimmed away
    const u32 nl = M.line_len[4 * (size_t)g], sl = M.line_len[4 * (size_t)g + 2];
    if (off + nl + sl + 2u * len + 4u > M.out_cap) return;
    M.out[off + nl + 1 + (pos - front)] = (u8)((w >> 16) & 0xFFu);
    M.out[off + nl + 1 + len + 1 + sl + 1 + (pos - front)] = (u8)(w >> 24);
}

// ---------------------------------------------------------------------------
// Every output stream on the device (second half): the routing of
// peprocessor.cpp:518-621 / seprocessor.cpp:280-290 from the result records, Read::appendToString /
// appendToStringWithTag (read.cpp:119-154), OverlapAnalysis::merge's string assembly
// (overlapanalysis.cpp:148-179) and UmiProcessor::addUmiToName (umiprocessor.cpp:62-81).
// A unit (pair / single read) makes at most two records ("emissions"); an emission is
//   stream | source << 3 | tag kind << 5 | filter code << 8      source: 0 read 1, 1 read 2, 2 the merged read
//                                                                  tag kind: 0 none, 1 FAILED_TYPES[code], 2 "paired_read_is_failing"
//   fmts_corr  : BaseCorrector's edits patched into the parsed text (the reference edits reads in place)
//   fmts_len   : bytes each block's units add to each stream
//   fmts_scan  : running offsets per stream
//   fmts_write : a 16-lane group per emission assembles the record
// ---------------------------------------------------------------------------
FQ_DEV u32 fmts_em(int stream, int src, int tagkind, u32 code) { return (u32)stream | ((u32)src << 3) | ((u32)tagkind << 5) | (code << 8); }

FQ_DEV void fmts_route(const FmtsArgs& f, int g, u32 e[2]) {
    e[0] = e[1] = FMTS_NONE;
    const u32 w1 = f.m[0].res[(size_t)g * 3 + 1];
    const u32 code1 = w1 & 0xFFu, fl1 = (w1 >> 8) & 0xFFu;
    const bool alive1 = !(fl1 & RS_NULL);
    const bool dedup_out = f.dedup && (fl1 & RS_DUP);
    if (!f.paired) {  // seprocessor.cpp:280-290
        if (dedup_out) return;
        if (alive1 && code1 == 0u) e[0] = fmts_em(0, 0, 0, 0);
        else if (f.want_failed) e[0] = fmts_em(2, 0, 1, code1);
        return;
    }
    const u32 w2 = f.m[1].res[(size_t)g * 3 + 1];
    const u32 code2 = w2 & 0xFFu, fl2 = (w2 >> 8) & 0xFFu;
    const bool alive2 = !(fl2 & RS_NULL);
    if (f.merge && alive1 && alive2) {  // peprocessor.cpp:518-561
        const u32 pflags = f.pair[2 * (size_t)g + 1] >> 16;
        if (pflags & 1u) {  // overlapped
            if (code1 == 0u) e[0] = fmts_em(3, 2, 0, 0);
            return;
        }
        if (f.merge_include_unmerged) {
            int k = 0;
            if (code1 == 0u && !dedup_out) e[k++] = fmts_em(3, 0, 0, 0);
            if (code2 == 0u && !dedup_out) e[k++] = fmts_em(3, 1, 0, 0);
            return;
        }
    }
    if (dedup_out) return;
    const bool p1 = alive1 && code1 == 0u, p2 = alive2 && code2 == 0u;
    const bool wf = f.want_failed != 0;
    if (p1 && p2) {  // :577-594
        e[0] = fmts_em(0, 0, 0, 0);
        e[1] = fmts_em(1, 1, 0, 0);
    } else if (p1) {  // :595-605
        if (f.want_u1) {
            e[0] = fmts_em(4, 0, 0, 0);
            if (wf) e[1] = fmts_em(2, 1, 1, code2);
        } else if (wf) {
            e[0] = fmts_em(2, 0, 2, 0);
            e[1] = fmts_em(2, 1, 1, code2);
        }
    } else if (p2) {  // :606-621
        if (f.want_u2) {
            e[0] = fmts_em(5, 1, 0, 0);
            if (wf) e[1] = fmts_em(2, 0, 1, code1);
        } else if (f.want_u1) {
            e[0] = fmts_em(4, 1, 0, 0);
            if (wf) e[1] = fmts_em(2, 0, 1, code1);
        } else if (wf) {
            e[0] = fmts_em(2, 0, 1, code1);
            e[1] = fmts_em(2, 1, 2, 0);
        }
    }
}

// FAILED_TYPES (common.h:57-66) / "paired_read_is_failing": character i of the tag, its length
FQ_DEV const char* fmts_tag_text(int kind, u32 code) {
    if (kind == 2) return "paired_read_is_failing";
    switch (code) {
        case 0: return "passed";
        case 4: return "failed_polyx_filter";
        case 8: return "failed_bad_overlap";
        case 12: return "failed_too_many_n_bases";
        case 16: return "failed_too_short";
        case 17: return "failed_too_long";
        case 20: return "failed_quality_filter";
        case 24: return "failed_low_complexity";
        case 28: return "failed_adapter_dimer";
        default: return "";
    }
}
FQ_DEV u32 fmts_strlen(const char* t) {
    u32 n = 0;
    while (t[n]) n++;
    return n;
}
FQ_DEV u32 fmts_digits(u32 v) { return v >= 1000u ? 4u : v >= 100u ? 3u : v >= 10u ? 2u : 1u; }

// UmiProcessor::process + addUmiToName: length of the text inserted into the names of this unit (0 = no edit),
// and the lengths of its two UMI parts
FQ_DEV u32 fmts_umi(const FmtsArgs& f, int g, u32& u1, u32& u2) {
    u1 = u2 = 0;
    if (f.umi_loc == 0) return 0;
    const u32 ul = (u32)imax(0, f.umi_len);
    const u32 l1 = f.m[0].line_len[4 * (size_t)g + 1];
    const u32 l2 = f.paired ? f.m[1].line_len[4 * (size_t)g + 1] : 0u;
    bool tag = true;
    if (f.umi_loc == 1) u1 = l1 < ul ? l1 : ul;
    else if (f.umi_loc == 2) { if (f.paired) u2 = l2 < ul ? l2 : ul; else tag = false; }
    else { u1 = l1 < ul ? l1 : ul; if (f.paired) u2 = l2 < ul ? l2 : ul; }
    if (f.umi_loc != 3 && u1 + u2 == 0u) tag = false;
    if (!tag) return 0;
    return f.delim_len + (f.prefix_len ? f.prefix_len + 1u : 0u) + u1 + u2 + ((f.umi_loc == 3 && f.paired) ? 1u : 0u);
}

// geometry of one emission of unit g
struct FmtsRec {
    int mt;            // the mate whose name / strand lines are used
    u32 name_len, strand_len, seq_len, tag_len, umi_len, mtag_len;   // mtag: " merged_L1_L2"
    u32 m1, m2, ol;    // merged parts
    bool strand_tagged;
    u32 bytes;
};
FQ_DEV void fmts_rec(const FmtsArgs& f, int g, u32 em, u32 umi_len, FmtsRec& r) {
    const int src = (int)((em >> 3) & 3u);
    r.mt = src == 1 ? 1 : 0;
    const FmtsMate& M = f.m[r.mt];
    r.name_len = M.line_len[4 * (size_t)g];
    r.strand_len = M.line_len[4 * (size_t)g + 2];
    r.umi_len = umi_len;
    const int tagkind = (int)((em >> 5) & 7u);
    r.tag_len = tagkind ? fmts_strlen(fmts_tag_text(tagkind, (em >> 8) & 0xFFu)) : 0u;
    r.mtag_len = 0;
    r.m1 = r.m2 = r.ol = 0;
    r.strand_tagged = false;
    if (src == 2) {
        // bases of each mate in the merged read, from the pair record (see ovr_count_body)
        const u32 pw = f.pair[2 * (size_t)g];
        const int poff = (int)(int16_t)(pw & 0xFFFFu);
        r.ol = pw >> 16;
        r.m1 = r.ol + (u32)imax(0, poff);
        r.m2 = poff > 0 ? (f.m[1].res[(size_t)g * 3] >> 16) - r.ol : 0u;
        r.seq_len = r.m1 + r.m2;
        r.mtag_len = 8u + fmts_digits(r.m1) + 1u + fmts_digits(r.m2);   // " merged_" L1 "_" L2
        // the strand line gets the tag too unless it is just "+" (overlapanalysis.cpp:170-173)
        const u8* st = M.text + M.line_off[4 * (size_t)g + 2];
        r.strand_tagged = !(r.strand_len == 1u && st[0] == '+');
    } else {
        r.seq_len = M.res[(size_t)g * 3] >> 16;
    }
    r.bytes = r.name_len + r.umi_len + r.mtag_len + (r.tag_len ? 1u + r.tag_len : 0u) + 1u + r.seq_len + 1u + r.strand_len +
              (r.strand_tagged ? r.mtag_len : 0u) + 1u + r.seq_len + 1u;
}

FQ_DEV void fmts_corr_body(const FmtsArgs& f) {
    const int i = block_id() * block_threads() + thread_id();
    if (!f.corrections || !f.n_corrections || i >= *f.n_corrections) return;
    const u32 rd = f.corrections[2 * (size_t)i], w = f.corrections[2 * (size_t)i + 1];
    const int mt = f.paired ? (int)(rd & 1u) : 0;
    const int g = (int)(f.paired ? rd >> 1 : rd) - f.corr_first;
    if (g < 0 || g >= f.n) return;
    const FmtsMate& M = f.m[mt];
    const u32 pos = w & 0xFFFFu;
    if (pos >= M.line_len[4 * (size_t)g + 1]) return;
    M.text[M.line_off[4 * (size_t)g + 1] + pos] = (u8)((w >> 16) & 0xFFu);
    M.text[M.line_off[4 * (size_t)g + 3] + pos] = (u8)(w >> 24);
}

// per-unit byte counts per stream -> LDS block sums -> blocksum
FQ_DEV void fmts_len_body(const FmtsArgs& f, u32* lds) {
    if (thread_id() < FMTS_STREAMS) lds[thread_id()] = 0;
    block_sync();
    const int g = block_id() * block_threads() + thread_id();
    u32 bytes[FMTS_STREAMS] = {0, 0, 0, 0, 0, 0};
    if (g < f.n) {
        u32 e[2];
        fmts_route(f, g, e);
        u32 u1, u2;
        const u32 ul = (e[0] != FMTS_NONE) ? fmts_umi(f, g, u1, u2) : 0u;
        for (int k = 0; k < 2; k++) {
            if (e[k] == FMTS_NONE) continue;
            FmtsRec r;
            fmts_rec(f, g, e[k], ul, r);
            const int st = (int)(e[k] & 7u);
#pragma unroll
            for (int q = 0; q < FMTS_STREAMS; q++) bytes[q] += q == st ? r.bytes : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < FMTS_STREAMS; q++) {
        u32 b = bytes[q];
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) b += shfl_xor(b, sh);
        if (lane_id() == 0 && b) lds_add_u32(&lds[q], b);
    }
    block_sync();
    if (thread_id() < FMTS_STREAMS) f.blocksum[(size_t)thread_id() * f.nblocks + block_id()] = lds[thread_id()];
}

FQ_DEV void fmts_scan_body(const FmtsArgs& f, u64* lds) {
    // one workgroup per stream (block_id = stream): lanes sum runs of blocks, scan through LDS
    const int q = block_id();
    const int tid = thread_id(), nt = block_threads();
    const int per = (f.nblocks + nt - 1) / nt;
    const int b0 = imin(tid * per, f.nblocks), b1 = imin(b0 + per, f.nblocks);
    const u64* sums = f.blocksum + (size_t)q * f.nblocks;
    u64* base = f.blockbase + (size_t)q * f.nblocks;
    u64 sum = 0;
    for (int b = b0; b < b1; b++) sum += sums[b];
    lds[tid] = sum;
    block_sync();
    if (tid == 0) {
        u64 run = 0;
        for (int i = 0; i < nt; i++) {
            const u64 v = lds[i];
            lds[i] = run;
            run += v;
        }
        f.totals[q] = run;
    }
    block_sync();
    u64 run = lds[tid];
    for (int b = b0; b < b1; b++) {
        base[b] = run;
        run += sums[b];
    }
}

// n bytes by a 16-lane group (no terminator)
FQ_DEV void fmts_put(u8* dst, const u8* src, u32 n, int gl) {
    for (u32 i = 4u * (u32)gl; i + 4u <= n; i += 64u) {  // global memory takes unaligned dwords
        u32 w;
        __builtin_memcpy(&w, src + i, 4);
        __builtin_memcpy(dst + i, &w, 4);
    }
    if (gl == 15)
        for (u32 i = n & ~3u; i < n; i++) dst[i] = src[i];
}
FQ_DEV u8 fmts_complement(u8 c) {  // util.h:16-33: anything outside ACGTacgt -> 'N'
    switch (c) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}
// " merged_L1_L2" by the group's first lane; returns its length
FQ_DEV u32 fmts_put_mtag(u8* dst, u32 m1, u32 m2, int gl) {
    const u32 d1 = fmts_digits(m1), d2 = fmts_digits(m2);
    if (gl == 0) {
        const char* pre = " merged_";
        for (int i = 0; i < 8; i++) dst[i] = (u8)pre[i];
        u32 v = m1;
        for (u32 i = 0; i < d1; i++) { dst[8 + d1 - 1 - i] = (u8)('0' + v % 10u); v /= 10u; }
        dst[8 + d1] = '_';
        v = m2;
        for (u32 i = 0; i < d2; i++) { dst[9 + d1 + d2 - 1 - i] = (u8)('0' + v % 10u); v /= 10u; }
    }
    return 8u + d1 + 1u + d2;
}

FQ_DEV void fmts_write_body(const FmtsArgs& f, u32* lds) {
    // lds: [FMTS_STREAMS][16 waves] wave sums, then per unit [2] u64 offsets of its emissions (block_threads * 2 * 2 dwords)
    const int tid = thread_id(), nw = block_threads() >> 6, wave = wave_id();
    const int g0 = block_id() * block_threads();
    const int g = g0 + tid;
    u32 e[2] = {FMTS_NONE, FMTS_NONE};
    u32 sz[2] = {0, 0};
    u32 ul = 0, u1 = 0, u2 = 0;
    if (g < f.n) {
        fmts_route(f, g, e);
        if (e[0] != FMTS_NONE) ul = fmts_umi(f, g, u1, u2);
        for (int k = 0; k < 2; k++)
            if (e[k] != FMTS_NONE) {
                FmtsRec r;
                fmts_rec(f, g, e[k], ul, r);
                sz[k] = r.bytes;
            }
    }
    u64* eoff = (u64*)(lds + FMTS_STREAMS * 16);
    u64 off[2] = {~0ull, ~0ull};
    u32 excl[FMTS_STREAMS], mine0[FMTS_STREAMS];
#pragma unroll
    for (int q = 0; q < FMTS_STREAMS; q++) {
        const u32 b0 = (e[0] != FMTS_NONE && (int)(e[0] & 7u) == q) ? sz[0] : 0u;
        const u32 b1 = (e[1] != FMTS_NONE && (int)(e[1] & 7u) == q) ? sz[1] : 0u;
        const u32 b = b0 + b1;
        u32 incl = b;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const u32 o = shfl(incl, lane_id() - sh);
            if (lane_id() >= sh) incl += o;
        }
        if (lane_id() == 63) lds[q * 16 + wave] = incl;
        excl[q] = incl - b;
        mine0[q] = b0;
    }
    block_sync();
#pragma unroll
    for (int q = 0; q < FMTS_STREAMS; q++) {
        u64 base = f.blockbase[(size_t)q * f.nblocks + block_id()] + excl[q];
        for (int w = 0; w < nw; w++)
            if (w < wave) base += lds[q * 16 + w];
        if (e[0] != FMTS_NONE && (int)(e[0] & 7u) == q) off[0] = base;
        if (e[1] != FMTS_NONE && (int)(e[1] & 7u) == q) off[1] = base + mine0[q];
    }
    eoff[2 * tid] = off[0];
    eoff[2 * tid + 1] = off[1];
    block_sync();
    // a 16-lane group per (unit, emission slot)
    const int gl = lane_id() & 15, ngroups = block_threads() >> 4;
    for (int t = tid >> 4; t < block_threads() * 2; t += ngroups) {
        const int u = t >> 1, k = t & 1;
        const int gu = g0 + u;
        if (gu >= f.n) break;
        const u64 o64 = eoff[2 * u + k];
        if (o64 == ~0ull) continue;
        u32 ee[2];
        fmts_route(f, gu, ee);
        const u32 em = ee[k];
        u32 v1, v2;
        const u32 uml = fmts_umi(f, gu, v1, v2);
        FmtsRec r;
        fmts_rec(f, gu, em, uml, r);
        const int st = (int)(em & 7u), src = (int)((em >> 3) & 3u), tagkind = (int)((em >> 5) & 7u);
        if (!f.out[st] || o64 + r.bytes > f.out_cap[st]) continue;  // the host sees the needed size in totals
        const FmtsMate& M = f.m[r.mt];
        u8* o = f.out[st] + o64;
        // ---- name line: name [up to its first space] + UMI tag + rest + merged tag + failed tag ----
        const u8* name = M.text + M.line_off[4 * (size_t)gu];
        u32 sp = r.name_len;
        if (uml) {  // position of the first space (one lane scans: names are short)
            for (u32 i = 0; i < r.name_len; i++)
                if (name[i] == ' ') { sp = i; break; }
        }
        fmts_put(o, name, sp, gl);
        o += sp;
        if (uml) {
            fmts_put(o, f.delim, f.delim_len, gl);
            o += f.delim_len;
            if (f.prefix_len) {
                fmts_put(o, f.prefix, f.prefix_len, gl);
                if (gl == 0) o[f.prefix_len] = '_';
                o += f.prefix_len + 1u;
            }
            fmts_put(o, f.m[0].text + f.m[0].line_off[4 * (size_t)gu + 1], v1, gl);
            o += v1;
            if (f.umi_loc == 3 && f.paired) {
                if (gl == 0) o[0] = '_';
                o += 1;
            }
            if (v2) fmts_put(o, f.m[1].text + f.m[1].line_off[4 * (size_t)gu + 1], v2, gl);
            o += v2;
        }
        fmts_put(o, name + sp, r.name_len - sp, gl);
        o += r.name_len - sp;
        if (src == 2) o += fmts_put_mtag(o, r.m1, r.m2, gl);
        if (tagkind) {
            if (gl == 0) o[0] = ' ';
            fmts_put(o + 1, (const u8*)fmts_tag_text(tagkind, (em >> 8) & 0xFFu), r.tag_len, gl);
            o += 1u + r.tag_len;
        }
        if (gl == 0) o[0] = 10;
        o += 1;
        // ---- sequence, strand, quality ----
        const u32 front = M.res[(size_t)gu * 3] & 0xFFFFu;
        const u8* seq = M.text + M.line_off[4 * (size_t)gu + 1] + front;
        const u8* qual = M.text + M.line_off[4 * (size_t)gu + 3] + front;
        const u8 *seq2 = nullptr, *qual2 = nullptr;
        u32 t2l = 0;
        if (src == 2) {  // merged: r1'[0, m1) then rc(r2')[ol + k] = comp(r2'[len2 - 1 - ol - k]), k < m2
            const FmtsMate& M2 = f.m[1];
            const u32 w2 = M2.res[(size_t)gu * 3];
            t2l = w2 >> 16;
            seq2 = M2.text + M2.line_off[4 * (size_t)gu + 1] + (w2 & 0xFFFFu);
            qual2 = M2.text + M2.line_off[4 * (size_t)gu + 3] + (w2 & 0xFFFFu);
        }
        const u32 n1 = src == 2 ? r.m1 : r.seq_len;
        fmts_put(o, seq, n1, gl);
        if (src == 2)
            for (u32 i = (u32)gl; i < r.m2; i += 16u) o[n1 + i] = fmts_complement(seq2[t2l - 1u - r.ol - i]);
        o += r.seq_len;
        if (gl == 0) o[0] = 10;
        o += 1;
        fmts_put(o, M.text + M.line_off[4 * (size_t)gu + 2], r.strand_len, gl);
        o += r.strand_len;
        if (r.strand_tagged) o += fmts_put_mtag(o, r.m1, r.m2, gl);
        if (gl == 0) o[0] = 10;
        o += 1;
        fmts_put(o, qual, n1, gl);
        if (src == 2)
            for (u32 i = (u32)gl; i < r.m2; i += 16u) o[n1 + i] = qual2[t2l - 1u - r.ol - i];
        o += r.seq_len;
        if (gl == 0) o[0] = 10;
    }
}

}  // namespace fq
