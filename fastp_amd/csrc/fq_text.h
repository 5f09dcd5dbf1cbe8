// fq_text.h - the worker loop on the reads' TEXT, one WAVEFRONT per unit ("text kernel", round 6).
//
// The lane / split / fused plans work on the packed batch: 2 bits per base + an N flag.  That is everything the loop
// ever looks at while the letters are A, C, G, T, N - and not enough for any other byte (lower case, IUPAC codes, '.'):
// the reference treats such a byte differently in nearly every consumer (DESIGN.md section 1 has the table):
//   Stats::statRead bins it by `base & 7` (stats.cpp:206-222), the 5-mer window drops it (:229-266), Duplicate hashes it
//   as 13 (duplicate.cpp:92-109), reverseComplement turns a/c/g/t into T/G/C/A and everything else into N
//   (util.h:16-33, simd.cpp:296-310), overlap analysis / adapter matching / polyG / polyX / the complexity filter compare
//   raw bytes, passFilter and trimAndCut count only a literal 'N'.
// Units that hold such a byte ("exotic" units, fastp_gpu_batch::exotic_*) go through THIS kernel.
//
// Shape (it replaces round 2-5's fq_exact.h, which walked a unit byte by byte on ONE lane with its text in HBM scratch:
// 3.4 ms for 4,194 units - the longest kernel of a launch that held one soft-masked pair in a thousand):
//   * a wavefront owns a unit; the unit's texts (both reads, their qualities, rc(read 2), the merged read, the decoded
//     adapters) live in the wavefront's stretch of LDS - no HBM scratch;
//   * every step of the loop body is evaluated ACROSS the lanes, never as a walk:
//       - the scans of Filter::trimAndCut, PolyX::trimPolyG / trimPolyX: lane = position, the reference's running sums as
//         window sums / ballot prefix counts, "first position where ..." = the lowest set bit of a ballot;
//       - OverlapAnalysis::analyze, AdapterTrimmer::trimBySequence: lane = CANDIDATE (an offset / an adapter position), each
//         lane counts its candidate's mismatches over the LDS bytes, the accepted candidate = the first set bit of the ballot
//         in the reference's order of trial;
//       - Matcher's one-insertion forms need no L / R tables: with Lt[i] = mismatches of ins[0..i] against nor[0..i] and
//         Rt[i] = mismatches of ins[i+1..c] against nor[i..c-1], the reference's early breaks only ever hide entries that
//         are not read before it returns, and values that are above the limit either way (proof in t_gap_best), so
//         matchWithOneInsertion = (min_i Lt[i-1] + Rt[i] <= limit) and diffWithOneInsertion = that minimum unless
//         Lt[c-2] + Rt[c-1] > limit: two running sums per lane;
//       - Stats::statRead, BaseCorrector, Duplicate's hash, passFilter, the merged read: lane = base (the 5-mer at base i is
//         counted iff the five letters i-4..i are all ACGT - what the reference's rolling window with its `needFullCompute`
//         restarts comes to);
//   * Stats' per-base counters of a workgroup's units are gathered in LDS tables and added to the int64 block once.
//
// How it composes with the plan's kernels (fastp_gpu.hip launch_chunk): those still sweep the whole launch, but on a
// copy of the length arrays in which the listed units are EMPTY reads.  What an empty unit adds to the counters is
// exactly what this loop computes for an empty unit - so ONE wavefront of the launch runs the loop on an empty unit with
// sign -(number of listed units) (the "ghost" pass: counters only; nothing in it depends on which unit it stands for), every
// listed unit is then run with sign +1 and overwrites its records and hash values.  Duplicate's bloom semantics stay with the
// fq_dup_* kernels, which run once over the whole launch afterwards: this kernel leaves the COMPLETE hash values of its units
// (the position part included - the dup kernels add the position sum of the length they see, zero).
//
// Each function cites the reference lines it answers for.
#pragma once
#include "fq_device.h"

namespace fq {

struct TextCtr {   // offsets into the int64 counter block (fastp_gpu_counter_layout)
    long long filter, adapter_reads, adapter_bases, polyx_reads, polyx_bases, correction, corrected_reads, merged, isize;
    long long stats[4];
    long long st_reads, st_length_sum, st_qual_hist, st_kmer, st_cycle, cycles;
};

struct TextArgs {
    KernelArgs k;        // parameters, LUTs, the launch's rows / records / lists (pointers already at `first`)
    TextCtr c;
    int64_t* ctr;
    // which units: the entries [x_k0, x_k0 + x_count) of the batch's list, or (x_all) the launch's units 0 .. x_count - 1
    int x_all, x_k0, x_count;
    // raw sequence bytes of the exotic units
    const int* x_unit;   // [x_n] ascending unit indexes inside the submitted batch
    int x_n;
    int x_dense;         // 1: x_off[m] is fastp_gpu_parse_fastq's line table of mate m ([4 * unit + 1] = the sequence line)
    const u8* x_text[2];
    const u32* x_off[2]; // [x_n] byte offset of the unit's sequence in x_text[m] (or the dense table)
    int ML;              // bytes per text buffer of a wavefront (max_len + slack, multiple of 8)
    int hash_only;       // --dedup's pre-pass: leave the hash values, nothing else
    // Stats::statRead's per-base counters of the workgroup's units are gathered in LDS (u32: [slot][34 * cycles | 1024 5-mers |
    // 128 quality characters]) and added to the block once at the end.  0: the tables do not fit (long reads)
    int lds_slot_dwords;
    int lds_slots;       // Stats objects that can be touched: 2 single-end, 3 merge mode (nothing reaches POST2), else 4
};

enum { TEXT_BUFS = 9, TEXT_ADAPTER_BYTES = 3 * 264, TEXT_WAVES = 8 };   // text buffers of ML bytes per wavefront; three decoded adapters
static inline __host__ __device__ int text_wave_bytes(int ML) { return TEXT_BUFS * ML + TEXT_ADAPTER_BYTES; }

struct XRead {
    u8* s;
    u8* q;
    int len;
    int front;
};

struct TextMaskArgs {
    const int* units;   // listed units of this launch (batch indexes)
    int count, first;
    u16* len[2];        // the launch's copies of the length arrays
    u8* skip;           // [n] or nullptr: 1 for the listed units (KernelArgs::xskip)
};
FQ_DEV void text_mask_body(const TextMaskArgs& m) {
    const int i = block_id() * block_threads() + thread_id();
    if (i >= m.count) return;
    const int gp = m.units[i] - m.first;
    m.len[0][gp] = 0;
    if (m.len[1]) m.len[1][gp] = 0;
    if (m.skip) m.skip[gp] = 1;
}

// what a wavefront carries through the loop body; everything but `lane` is the same in all its lanes
struct TWave {
    const TextArgs* E;
    u32* tables;         // the workgroup's Stats tables in LDS, or null
    u8 *s1, *q1, *s2, *q2, *rc, *ms, *mq, *ad1, *ad2, *adf;
    int lane;
    long long sign;      // +1; the ghost pass: -(units of the launch)
    bool ghost;
};

// ---- wave helpers ----
FQ_DEV u64 lanes_upto(int lane) { return lane >= 63 ? ~0ull : ((2ull << lane) - 1ull); }   // lanes 0 .. lane
FQ_DEV int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (int)shfl_xor((u32)v, m);
    return v;
}
FQ_DEV u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const u64 o = (u64)shfl_xor((u32)v, m) | ((u64)shfl_xor((u32)(v >> 32), m) << 32);
        v += o;
    }
    return v;
}
// the smallest i in [lo, hi) with pred(i), or T_NONE (far below any position); lo / hi the same in every lane
enum { T_NONE = -(1 << 30) };
template <class P> FQ_DEV int first_up(int lo, int hi, int lane, P pred) {
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const u64 m = ballot(i < hi && pred(i));
        if (m) return base + ffs64(m) - 1;
    }
    return T_NONE;
}
// the largest i in [lo, hi] with pred(i), or lo - 1
template <class P> FQ_DEV int first_down(int lo, int hi, int lane, P pred) {
    for (int base = hi; base >= lo; base -= 64) {
        const int i = base - lane;
        const u64 m = ballot(i >= lo && pred(i));
        if (m) return base - (ffs64(m) - 1);
    }
    return lo - 1;
}

// counters: a value every lane holds (lane 0 adds it), or a lane's own
FQ_DEV void t_add(const TWave& W, long long off, long long v) {
    if (W.lane == 0) g_atomic_add_i64(W.E->ctr + off, (int64_t)(v * W.sign));
}
FQ_DEV void t_add_lane(const TWave& W, long long off, long long v) { g_atomic_add_i64(W.E->ctr + off, (int64_t)(v * W.sign)); }

FQ_DEV u8 t_complement(u8 b) {   // util.h:16-33
    switch (b) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}
FQ_DEV int t_base2val(u8 b) {   // stats.cpp:294-311
    switch (b) {
        case 'A': return 0;
        case 'T': return 1;
        case 'C': return 2;
        case 'G': return 3;
        default: return -1;
    }
}
FQ_DEV u64 t_hash_val(u8 b) {   // duplicate.cpp:92-109
    switch (b) {
        case 'A': return 7;
        case 'T': return 222;
        case 'C': return 74;
        case 'G': return 31;
        default: return 13;
    }
}
FQ_DEV u8 t_code_ascii(u32 code) { return (u8)("ATCG"[code & 3u]); }

// the raw sequence bytes of unit gp's mate m, or null when the unit is not exotic
FQ_DEV const u8* t_raw(const TextArgs& E, int gp, int m) {
    const int unit = E.k.first + gp;
    int lo = 0, hi = E.x_n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (E.x_unit[mid] < unit) lo = mid + 1;
        else hi = mid;
    }
    if (lo >= E.x_n || E.x_unit[lo] != unit) return nullptr;
    const u32 off = E.x_dense ? E.x_off[m][4 * (size_t)unit + 1] : E.x_off[m][lo];
    return E.x_text[m] + off;
}

// the unit's text into the wavefront's buffers: a lane per base
FQ_DEV int t_load(const TWave& W, int gp, int m, u8* s, u8* q) {
    if (W.ghost) return 0;
    const TextArgs& E = *W.E;
    const int len = (int)E.k.len[m][gp];
    const u8* qrow = (const u8*)(E.k.qual[m] + (size_t)gp * E.k.p.qw_g);
    const u32* srow = E.k.seq[m] + (size_t)gp * E.k.p.sw_g;
    const u8* raw = t_raw(E, gp, m);
    for (int i = W.lane; i < len; i += 64) {
        const u8 qq = qrow[i];
        q[i] = (u8)(qq & 0x7Fu);
        s[i] = raw ? raw[i] : ((qq & 0x80u) ? (u8)'N' : t_code_ascii(srow[i >> 4] >> ((i & 15) * 2)));
    }
    wave_sync();
    return len;
}

FQ_DEV void t_decode_adapter(const TWave& W, const u32* words, int alen, u8* out) {
    for (int i = W.lane; i < alen; i += 64) out[i] = t_code_ascii(words[i >> 4] >> ((i & 15) * 2));
    wave_sync();
}

// ---- Stats::statRead (stats.cpp:191-291, without the overrepresentation part): a lane per base.  The reference's 5-mer
// window restarts (`needFullCompute`) after an N or a letter base2val() does not know and is valid again once five such letters
// have gone by: base i >= 4 counts its 5-mer iff the letters i-4 .. i are all ACGT ----
FQ_DEV void t_stat_read(const TWave& W, int slot, const u8* s, const u8* q, int len) {
    const TextArgs& E = *W.E;
    const long long st = E.c.stats[slot];
    const int C = (int)E.c.cycles;
    t_add(W, st + E.c.st_length_sum, len);
    t_add(W, st + E.c.st_reads, 1);
    if (len <= 0) return;
    const long long cyc = st + E.c.st_cycle;
    u32* t = W.tables ? W.tables + (size_t)slot * E.lds_slot_dwords : nullptr;
    for (int i = W.lane; i < len; i += 64) {
        const u8 base = s[i], qc = q[i];
        const int b = base & 7;
        int kmer = -1;
        if (i >= 4) {
            kmer = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int v = t_base2val(s[i - 4 + k]);
                kmer = (v < 0 || kmer < 0) ? -1 : ((kmer << 2) | v);
            }
        }
        const u32 qv = (u32)((int)qc - 33);
        if (t) {   // [q30 8C | q20 8C | content 8C | quality 8C | total bases C | total quality C | 1024 5-mers | 128 characters]
            lds_add_u32(&t[34 * C + 1024 + qc], 1u);
            if (qc >= '?') lds_add_u32(&t[b * C + i], 1u);
            if (qc >= '5') lds_add_u32(&t[8 * C + b * C + i], 1u);
            lds_add_u32(&t[16 * C + b * C + i], 1u);
            lds_add_u32(&t[24 * C + b * C + i], qv);
            lds_add_u32(&t[32 * C + i], 1u);
            lds_add_u32(&t[33 * C + i], qv);
            if (kmer >= 0) lds_add_u32(&t[34 * C + kmer], 1u);
        } else {
            t_add_lane(W, st + E.c.st_qual_hist + qc, 1);
            if (qc >= '?') t_add_lane(W, cyc + (long long)b * C + i, 1);
            if (qc >= '5') t_add_lane(W, cyc + 8LL * C + (long long)b * C + i, 1);
            t_add_lane(W, cyc + 16LL * C + (long long)b * C + i, 1);
            t_add_lane(W, cyc + 24LL * C + (long long)b * C + i, (long long)qv);
            t_add_lane(W, cyc + 32LL * C + i, 1);
            t_add_lane(W, cyc + 33LL * C + i, (long long)qv);
            if (kmer >= 0) t_add_lane(W, st + E.c.st_kmer + kmer, 1);
        }
    }
}

// sum of the w quality characters from q[i]
FQ_DEV int t_window(const u8* q, int i, int w) {
    int t = 0;
    for (int k = 0; k < w; k++) t += (int)q[i + k];
    return t;
}

// ---- Filter::trimAndCut (filter.cpp:68-207); false = NULL.  The reference slides a running total and stops at the first window
// on the wrong side of the threshold: here every lane sums the window that starts at its position ----
FQ_DEV bool t_trim_and_cut(const TWave& W, const DevParams& p, const u8* seq, const u8* q, int l, int front, int tail, int& out_front, int& out_len) {
    const bool enF = p.cut_front != 0, enT = p.cut_tail != 0, enR = p.cut_right != 0;
    const int lane = W.lane;
    out_front = 0;
    out_len = l;
    if (front == 0 && tail == 0 && !enF && !enT && !enR) return true;   // :71-72
    int rlen = l - front - tail;
    if (rlen < 0) return false;   // :76-77
    if (!enF && !enT && !enR) { out_front = front; out_len = rlen; return true; }   // :79-89
    if (enF) {   // :97-127: the first window AT the threshold, or the loop's last position
        const int w = p.wF;
        if (l - front - tail - w <= 0) return false;
        const int hi = l - tail - w;
        int s = first_up(front, hi, lane, [&](int i) { return t_window(q, i, w) >= p.thrF; });
        if (s < 0) s = hi;
        if (s > 0) s = s + w - 1;
        const int k = first_up(s, l, lane, [&](int i) { return seq[i] != 'N'; });
        front = k < 0 ? (s < l ? l : s) : k;
        rlen = l - front - tail;
    }
    if (enR) {   // :130-163: the first window BELOW the threshold, then on to the first base below the minimum
        const int w = p.wR;
        if (l - front - tail - w <= 0) return false;
        const int hi = l - tail - w;
        int s = first_up(front, hi, lane, [&](int i) { return t_window(q, i, w) < p.thrR; });
        if (s >= 0) {
            const int k = first_up(s, l - 1, lane, [&](int i) { return (int)q[i] < p.qRmin; });
            s = k < 0 ? (s < l - 1 ? l - 1 : s) : k;
            rlen = s - front;
        }
    }
    if (!enR && enT) {   // :166-194: from the tail, the first window AT the threshold
        const int w = p.wT;
        if (l - front - tail - w <= 0) return false;
        // t runs from l - tail - 1 down while t - w >= front; its window is q[t - w + 1 .. t]
        int t = first_down(front + w, l - tail - 1, lane, [&](int i) { return t_window(q, i - w + 1, w) >= p.thrT; });
        if (t < l - 1) t = t - w + 1;
        t = t < 0 ? t : first_down(0, t, lane, [&](int i) { return seq[i] != 'N'; });
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return false;   // :196-197
    out_front = front;
    out_len = rlen;
    return true;
}

FQ_DEV bool t_apply_trim_and_cut(const TWave& W, const DevParams& p, XRead& r, int front, int tail, int& ft) {
    int f = 0, l = 0;
    ft = 0;
    if (!t_trim_and_cut(W, p, r.s, r.q, r.len, front, tail, f, l)) return false;
    r.s += f;
    r.q += f;
    r.front += f;
    r.len = l;
    ft = f;
    return true;
}

FQ_DEV void t_trim_front(XRead& r, int len) {   // Read::trimFront read.cpp:69-73
    len = imin(r.len - 1, len);
    if (len < 0) return;
    r.s += len;
    r.q += len;
    r.len -= len;
    r.front += len;
}
FQ_DEV void t_resize(XRead& r, int len) {   // Read::resize read.cpp:62-67
    if (len > r.len || len < 0) return;
    r.len = len;
}

// ---- PolyX::trimPolyG (polyx.cpp:16-42): walk step i looks at base rlen - 1 - i; the mismatch count of a step is a ballot prefix
// count, the step the walk breaks at the first set bit of the break condition's ballot ----
FQ_DEV int t_trim_poly_g(const TWave& W, const u8* d, int rlen, int compare_req) {
    int carry = 0, ib = rlen, ig = -1;   // ib: the step the walk stops at (rlen: it ran out); ig: the last step <= ib that saw a G
    for (int base = 0; base < rlen; base += 64) {
        const int i = base + W.lane;
        const bool in = i < rlen;
        const bool notg = in && d[rlen - 1 - i] != 'G';
        const u64 m = ballot(notg), gm = ballot(in && !notg);
        const int mismatch = carry + popc64(m & lanes_upto(W.lane));
        const int allowed = (i + 1) / 8;
        const u64 bm = ballot(in && (mismatch > 5 || (mismatch > allowed && i >= compare_req - 1)));
        if (bm) {
            const int k = ffs64(bm) - 1;
            ib = base + k;
            const u64 g = gm & lanes_upto(k);
            if (g) ig = base + 63 - clz64(g);
            break;
        }
        if (gm) ig = base + 63 - clz64(gm);
        carry += popc64(m);
    }
    const int first_g = ig >= 0 ? rlen - 1 - ig : rlen - 1;
    if (ib >= compare_req && first_g >= 0 && first_g <= rlen) return first_g;
    return rlen;
}

// ---- PolyX::trimPolyX (polyx.cpp:49-116): four prefix counts (an N counts for every base) ----
FQ_DEV int t_trim_poly_x(const TWave& W, const u8* d, int rlen, int compare_req, int& poly_base, int& trimmed) {
    int cnt[4] = {0, 0, 0, 0};
    int pos = rlen;
    poly_base = -1;
    trimmed = 0;
    for (int base = 0; base < rlen; base += 64) {
        const int i = base + W.lane;
        const bool in = i < rlen;
        const u8 c = in ? d[rlen - i - 1] : (u8)0;
        const u64 below = lanes_upto(W.lane);
        const u64 mn = ballot(c == 'N');
        const u64 m0 = ballot(c == 'A') | mn, m1 = ballot(c == 'T') | mn, m2 = ballot(c == 'C') | mn, m3 = ballot(c == 'G') | mn;
        const int c0 = cnt[0] + popc64(m0 & below), c1 = cnt[1] + popc64(m1 & below), c2 = cnt[2] + popc64(m2 & below), c3 = cnt[3] + popc64(m3 & below);
        const int cmp = i + 1;
        const int allowed = imin(5, cmp / 8);
        const bool need_break = cmp - c0 > allowed && cmp - c1 > allowed && cmp - c2 > allowed && cmp - c3 > allowed;
        const u64 bm = ballot(in && need_break && (i >= 8 || i + 1 >= compare_req - 1));
        if (bm) {   // the counts as they stand when the walk breaks: its own step included
            const int k = ffs64(bm) - 1;
            const u64 upto = lanes_upto(k);
            pos = base + k;
            cnt[0] += popc64(m0 & upto); cnt[1] += popc64(m1 & upto); cnt[2] += popc64(m2 & upto); cnt[3] += popc64(m3 & upto);
            break;
        }
        cnt[0] += popc64(m0); cnt[1] += popc64(m1); cnt[2] += popc64(m2); cnt[3] += popc64(m3);
    }
    if (pos + 1 >= compare_req) {   // :98-115
        int poly = 0, max_count = -1;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (cnt[b] > max_count) { max_count = cnt[b]; poly = b; }
        const u8 pb = t_code_ascii((u32)poly);
        // :109 while (data[rlen - pos - 1] != polyBase && pos >= 0) pos--;  a step outside the read never matches: the last
        // step <= pos inside the read whose base is polyBase, or -1
        pos = first_down(0, imin(pos, rlen - 1), W.lane, [&](int i) { return d[rlen - i - 1] == pb; });
        const int newlen = rlen - pos - 1;
        poly_base = poly;
        trimmed = pos + 1;
        if (newlen < 0 || newlen > rlen) return rlen;
        return newlen;
    }
    return rlen;
}

// mismatches of a[0 .. len) against b[0 .. len), given up on above `limit` (simd.cpp:209-233, :319-324)
FQ_DEV int t_mismatches_bounded(const u8* a, const u8* b, int len, int limit) {
    int d = 0;
    for (int i = 0; i < len && d <= limit; i++) d += (a[i] != b[i]);
    return d;
}
FQ_DEV int t_mismatches(const u8* a, const u8* b, int len) {
    int d = 0;
    for (int i = 0; i < len; i++) d += (a[i] != b[i]);
    return d;
}

// ---- Matcher::matchWithOneInsertion / diffWithOneInsertion (matcher.cpp:10-100) without their tables.
// D0[i] = ins[i] != nor[i], D1[i] = ins[i + 1] != nor[i]; Lt[i] = D0[0] + .. + D0[i], Rt[i] = D1[i] + .. + D1[c - 1], r = Rt[c - 1].
// The reference fills L up to the first i whose Lt[i] + r is above the limit and leaves the rest zero; its final scan returns
// at the first i with L[i - 1] + r above the limit - one step behind that break, so it never reads a zeroed entry.  It fills R
// downwards to the first i whose Rt[i] + Lt[0] is above the limit and sets the entries below to limit + 1: for those i the true
// Lt[i - 1] + Rt[i] >= Lt[0] + Rt[i] is above the limit as well.  A sum at or below the limit at i implies L[j - 1] + r at or
// below it for every j <= i (both parts only grow), so the scan reaches it.  Hence, as far as the callers look (they compare
// with the limit): match = (best <= limit); diff = best when best <= limit and Lt[c - 2] + r <= limit, "no" otherwise.
// Returns best = min over i in [1, c) of Lt[i - 1] + Rt[i] (a large number for c < 2) and Lt[c - 2] + r in `edge` ----
FQ_DEV int t_gap_best(const u8* ins, const u8* nor, int c, int& edge) {
    int R = 0;
    for (int j = 0; j < c; j++) R += (ins[j + 1] != nor[j]);
    const int r = c > 0 ? (ins[c] != nor[c - 1] ? 1 : 0) : 0;
    int L = 0, best = 1 << 28;
    for (int i = 1; i < c; i++) {
        L += (ins[i - 1] != nor[i - 1]);
        R -= (ins[i] != nor[i - 1]);
        best = imin(best, L + R);
    }
    edge = L + r;
    return best;
}
FQ_DEV bool t_match_one_insertion(const u8* ins, const u8* nor, int cmplen, int limit) {   // :10-54
    if (cmplen <= 0) return false;
    int edge;
    return t_gap_best(ins, nor, cmplen, edge) <= limit;
}
// diffWithOneInsertion as its caller reads it: the difference when it is inside [0, limit], else -1
FQ_DEV int t_diff_one_insertion(const u8* ins, const u8* nor, int cmplen, int limit) {   // :56-100
    if (cmplen <= 0) return -1;
    int edge;
    const int best = t_gap_best(ins, nor, cmplen, edge);
    if (cmplen >= 2 && edge > limit) return -1;
    return best <= limit ? best : -1;
}

// ---- OverlapAnalysis::analyze (overlapanalysis.cpp:17-146): a lane per offset, in the reference's order of trial ----
struct XOverlap {
    int overlapped, offset, overlap_len, diff, has_gap;
};
FQ_DEV bool t_accept_nogap(const u8* a, const u8* b, int len, int limit, int& diff) {   // :34-44
    const int prefix = imin(len, 50);
    diff = t_mismatches_bounded(a, b, prefix, limit);
    if (diff > limit) return false;
    if (len > 50) diff = t_mismatches(a, b, len);
    return true;
}
// one direction of a scan: candidate k of [0, count) is accepted by `trial(k, ol, diff)`; the first accepted one wins
template <class T> FQ_DEV bool t_scan_offsets(int lane, int count, T trial, int& k_out, int& ol_out, int& diff_out) {
    for (int base = 0; base < count; base += 64) {
        const int k = base + lane;
        int ol = 0, diff = 0;
        const bool ok = k < count && trial(k, ol, diff);
        const u64 m = ballot(ok);
        if (m) {
            const int w = ffs64(m) - 1;
            k_out = base + w;
            ol_out = (int)shfl((u32)ol, w);
            diff_out = (int)shfl((u32)diff, w);
            return true;
        }
    }
    return false;
}
// `lut` = min(diffLimit, (int)(overlap_len * diffPercentLimit)) per overlap length, or null for diffPercentLimit 0
FQ_DEV XOverlap t_analyze(const TWave& W, const u8* r1, int len1, const u8* r2, int len2, u8* rc, const u16* lut, int overlap_require, bool allow_gap) {
    XOverlap ov = {0, 0, 0, 0, 0};
    wave_sync();
    for (int i = W.lane; i < len2; i += 64) rc[len2 - 1 - i] = t_complement(r2[i]);   // :19-22
    wave_sync();
    int k = 0, ol = 0, diff = 0;
    const int nf = imax(0, len1 - overlap_require), nr = imax(0, len2 - overlap_require);
    // :48-64 offset = k
    if (t_scan_offsets(W.lane, nf, [&](int o, int& l, int& d) {
            l = imin(len1 - o, len2);
            return t_accept_nogap(r1 + o, rc, l, lut ? (int)lut[l] : 0, d);
        }, k, ol, diff)) {
        ov.overlapped = 1; ov.offset = k; ov.overlap_len = ol; ov.diff = diff;
        return ov;
    }
    // :72-89 offset = -k
    if (t_scan_offsets(W.lane, nr, [&](int o, int& l, int& d) {
            l = imin(len1, len2 - o);
            return t_accept_nogap(r1, rc + o, l, lut ? (int)lut[l] : 0, d);
        }, k, ol, diff)) {
        ov.overlapped = 1; ov.offset = -k; ov.overlap_len = ol; ov.diff = diff;
        return ov;
    }
    if (allow_gap) {   // :91-139
        if (t_scan_offsets(W.lane, nf, [&](int o, int& l, int& d) {
                l = imin(len1 - o, len2);
                const int limit = lut ? (int)lut[l] : 0;
                d = t_diff_one_insertion(r1 + o, rc, l - 1, limit);
                if (d < 0) d = t_diff_one_insertion(rc, r1 + o, l - 1, limit);
                return d >= 0;
            }, k, ol, diff)) {
            ov.overlapped = 1; ov.offset = k; ov.overlap_len = ol; ov.diff = diff; ov.has_gap = 1;
            return ov;
        }
        if (t_scan_offsets(W.lane, nr, [&](int o, int& l, int& d) {
                l = imin(len1, len2 - o);
                const int limit = lut ? (int)lut[l] : 0;
                d = t_diff_one_insertion(r1, rc + o, l - 1, limit);
                if (d < 0) d = t_diff_one_insertion(rc + o, r1, l - 1, limit);
                return d >= 0;
            }, k, ol, diff)) {
            ov.overlapped = 1; ov.offset = -k; ov.overlap_len = ol; ov.diff = diff; ov.has_gap = 1;
            return ov;
        }
    }
    return ov;
}

// ---- AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157): a lane per adapter position ----
FQ_DEV bool t_trim_by_sequence(const TWave& W, const u8* rd, int rlen, const u8* ad, int alen, int match_req, int& out_pos) {
    if (alen < match_req) return false;
    int start = 0;
    if (alen >= 16) start = -4;
    else if (alen >= 12) start = -3;
    else if (alen >= 8) start = -2;
    int pos = first_up(start, rlen - match_req, W.lane, [&](int ps) {   // :87-100
        const int cmplen = imin(rlen - ps, alen);
        const int allowed = cmplen / 8;
        const int so = imax(0, -ps);
        return t_mismatches_bounded(ad + so, rd + so + ps, cmplen - so, allowed) <= allowed;
    });
    bool found = pos != T_NONE;
    if (!found)   // :105-118, rdata / adata WITHOUT + pos as in the reference
        found = (pos = first_up(0, rlen - match_req - 1, W.lane, [&](int ps) {
                     const int cmplen = imin(rlen - ps - 1, alen);
                     return t_match_one_insertion(rd, ad, cmplen, cmplen / 8 - 1);
                 })) >= 0;
    if (!found)   // :122-135
        found = (pos = first_up(0, rlen - match_req, W.lane, [&](int ps) {
                     const int cmplen = imin(rlen - ps, alen - 1);
                     return t_match_one_insertion(ad, rd, cmplen, cmplen / 8 - 1);
                 })) >= 0;
    out_pos = pos;
    return found;
}

// trimBySequence on the read + FilterResult::addAdapterTrimmed (filterresult.cpp:124-152)
FQ_DEV bool t_apply_trim_by_sequence(const TWave& W, XRead& r, const u8* ad, int alen, int match_req, int& pos, int& adapter_len) {
    pos = 0;
    if (!t_trim_by_sequence(W, r.s, r.len, ad, alen, match_req, pos)) return false;
    if (pos < 0) {   // adaptertrimmer.cpp:138-145
        adapter_len = alen + pos;
        r.len = 0;
    } else {
        adapter_len = r.len - pos;
        t_resize(r, pos);
    }
    if (adapter_len > 0) t_add(W, W.E->c.adapter_bases, adapter_len);
    return true;
}

// AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62): one event per cut for the host's adapter-string replay
FQ_DEV bool t_trim_by_multi(const TWave& W, XRead& r, u32 read_index) {
    const KernelArgs& a = W.E->k;
    bool trimmed = false;
    for (int i = 0; i < a.p.n_fasta; i++) {
        const int alen = a.lut.fasta_len[i];
        wave_sync();
        t_decode_adapter(W, a.lut.fasta_words + (size_t)i * ADAPT_WORDS, alen, W.adf);
        int pos, adapter_len;
        if (!t_apply_trim_by_sequence(W, r, W.adf, alen, a.p.fasta_match_req, pos, adapter_len)) continue;
        trimmed = true;
        if (a.adapter_events && W.lane == 0) {
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

// ---- Filter::passFilter (filter.cpp:15-66), thresholds through the host-built LUTs (the reference's double expressions) ----
FQ_DEV int t_pass_filter(const TWave& W, const KernelArgs& a, const u8* s, const u8* q, int rlen) {
    const DevParams& p = a.p;
    if (rlen == 0) return 16;
    int low = 0, nb = 0, total = 0;
    if (p.qual_filter || p.length_filter) {   // :26-33
        for (int base = 0; base < rlen; base += 64) {
            const int i = base + W.lane;
            const bool in = i < rlen;
            const int qv = in ? (int)q[i] : 33;
            total += qv - 33;
            low += popc64(ballot(in && qv < p.qual_thr));
            nb += popc64(ballot(in && s[i] == 'N'));
        }
        total = wave_sum_i32(total);
    }
    if (p.qual_filter) {   // :35-42
        if (low > (int)a.lut.lowq_limit[rlen]) return 20;
        else if (p.avg_qual_req > 0 && (total / rlen) < p.avg_qual_req) return 20;
        else if (nb > p.n_base_limit) return 12;
    }
    if (p.length_filter) {   // :44-49
        if (rlen < p.length_required) return 16;
        if (p.length_limit > 0 && rlen > p.length_limit) return 17;
    }
    if (p.complexity_filter) {   // :51-54, 59-66
        if (rlen <= 1) return 24;
        int diff = 0;
        for (int base = 0; base < rlen - 1; base += 64) {
            const int i = base + W.lane;
            diff += popc64(ballot(i < rlen - 1 && s[i] != s[i + 1]));
        }
        if (diff < (int)a.lut.cplx_min[rlen]) return 24;
    }
    return 0;
}

// ---- Duplicate::seq2intvector (duplicate.cpp:111-120), the whole value (the dup kernels add the position sum of the
// length THEY see for this unit: zero, it is an empty read in their length arrays) ----
FQ_DEV void t_hash(const TWave& W, const KernelArgs& a, const u8* s, int len, int pos_offset, u64 (&out)[MAX_DUP_BUFS]) {
    const int B = a.p.dup_bufnum;
    const u32 mask = (u32)(512 * B - 1);
    u64 part[MAX_DUP_BUFS];
#pragma unroll
    for (int i = 0; i < MAX_DUP_BUFS; i++) part[i] = 0;
    for (int p = W.lane; p < len; p += 64) {
        const u64 v = t_hash_val(s[p]) + (u64)(p + pos_offset);
#pragma unroll
        for (int i = 0; i < MAX_DUP_BUFS; i++)
            if (i < B) part[i] += (u64)a.lut.dup_primes[((u32)((p + pos_offset) * B + i)) & mask] * v;
    }
#pragma unroll
    for (int i = 0; i < MAX_DUP_BUFS; i++)
        if (i < B) out[i] += wave_sum_u64(part[i]);
}

// the unit's hash values for Duplicate's kernels - and, in a launch whose claim step is fused into the lane kernel
// (KernelArgs::claim_won), the unit's claim of its bloom bits (lane_claim of fq_lane.h: that kernel fires none for a listed unit;
// the value here is complete, the position sum of the empty read the other kernels see is zero)
FQ_DEV void t_leave_hash(const TWave& W, const KernelArgs& a, int gp, const u64 (&h)[MAX_DUP_BUFS]) {
    if (W.lane != 0) return;
    const int B = a.p.dup_bufnum;
    for (int i = 0; i < B; i++) a.dup_pos[(size_t)gp * B + i] = h[i];
    if (a.claim_won) {
        const u64 words = a.dup_bits >> 5;
        u32 won = 0;
        for (int i = 0; i < B; i++) {
            const u64 pos = h[i] & (a.dup_bits - 1);
            const u32 bit = 1u << (pos & 31);
            const u32 old = g_atomic_or_u32(&a.dup_bitmap[(size_t)i * words + (pos >> 5)], bit);
            if (!(old & bit)) won |= 1u << i;
        }
        a.claim_won[gp] = (u8)won;
    }
}

struct XRes {
    u32 flags, apos, alen, rsv;
};
FQ_DEV void t_write_read(const TWave& W, const KernelArgs& a, int m, int gp, const XRead& r, int code, const XRes& x) {
    if (W.lane != 0) return;
    u32* out = a.res[m] + (size_t)gp * 3;
    out[0] = ((u32)r.front & 0xFFFFu) | (((u32)r.len & 0xFFFFu) << 16);
    out[1] = ((u32)code & 0xFFu) | ((x.flags & 0xFFu) << 8) | ((x.apos & 0xFFFFu) << 16);
    out[2] = (x.alen & 0xFFFFu) | ((x.rsv & 0xFFFFu) << 16);
}

FQ_DEV void t_stat_isize(const TWave& W, int l1, int l2, const XOverlap& ov, int ft1, int ft2) {   // peprocessor.cpp:710-723
    const TextArgs& E = *W.E;
    int isize = E.k.p.isize_max;
    if (ov.overlapped) {
        if (ov.offset > 0) isize = l1 + l2 - ov.overlap_len + ft1 + ft2;
        else isize = ov.overlap_len + ft1 + ft2;
    }
    if (isize > E.k.p.isize_max) isize = E.k.p.isize_max;
    if (isize < 0) return;
    t_add(W, E.c.isize + isize, 1);
}

// ---- BaseCorrector::correctByOverlapAnalysis (basecorrector.cpp:16-83): a lane per position of the overlap (position i only
// touches base start1 + i of read 1 and base start2 - i of read 2) ----
FQ_DEV void t_correct(const TWave& W, XRead& r1, XRead& r2, const XOverlap& ov, int gp, XRes& x1, XRes& x2) {
    const TextArgs& E = *W.E;
    const KernelArgs& a = E.k;
    if (ov.diff == 0 || !ov.overlapped) return;
    const int ol = ov.overlap_len;
    const int start1 = imax(0, ov.offset);
    const int start2 = r2.len - imax(0, -ov.offset) - 1;
    const u8 GOOD = (u8)(30 + 33), BAD = (u8)(14 + 33);
    int corrected = 0;
    bool r1c = false, r2c = false;
    wave_sync();
    for (int i = W.lane; i < ol; i += 64) {
        const int p1 = start1 + i, p2 = start2 - i;
        if (r1.s[p1] == t_complement(r2.s[p2])) continue;
        int which = -1, pos = 0;
        u8 nb = 0, nq = 0, from = 0;
        if (r1.q[p1] >= GOOD && r2.q[p2] <= BAD) {
            from = r2.s[p2];
            nb = t_complement(r1.s[p1]);
            nq = r1.q[p1];
            r2.s[p2] = nb;
            r2.q[p2] = nq;
            which = 1;
            pos = r2.front + p2;
            r2c = true;
        } else if (r2.q[p2] >= GOOD && r1.q[p1] <= BAD) {
            from = r1.s[p1];
            nb = t_complement(r2.s[p2]);
            nq = r2.q[p2];
            r1.s[p1] = nb;
            r1.q[p1] = nq;
            which = 0;
            pos = r1.front + p1;
            r1c = true;
        }
        if (which < 0) continue;
        corrected++;
        t_add_lane(W, E.c.correction + (from & 7) * 8 + (nb & 7), 1);   // addCorrection filterresult.cpp:99-103
        if (a.corrections) {
            const int slot = g_atomic_add_i32(a.n_corrections, 1);
            if (slot < a.corr_capacity) {
                a.corrections[2 * slot] = 2u * (u32)(a.first + gp) + (u32)which;
                a.corrections[2 * slot + 1] = ((u32)pos & 0xFFFFu) | ((u32)nb << 16) | ((u32)nq << 24);
            }
        }
    }
    wave_sync();
    const bool any = ballot(corrected > 0) != 0ull, any1 = ballot(r1c) != 0ull, any2 = ballot(r2c) != 0ull;
    if (any) {   // :75-80
        t_add(W, E.c.corrected_reads, (any1 && any2) ? 2 : 1);
        if (any1) x1.flags |= RS_CORRECTED;
        if (any2) x2.flags |= RS_CORRECTED;
    }
}

FQ_DEV void t_poly_x(const TWave& W, XRead& r, XRes& x) {
    const TextArgs& E = *W.E;
    int poly, trimmed;
    const int nl = t_trim_poly_x(W, r.s, r.len, E.k.p.poly_x_min, poly, trimmed);
    if (poly >= 0) {   // addPolyXTrimmed filterresult.cpp:186-189
        t_add(W, E.c.polyx_reads + poly, 1);
        t_add(W, E.c.polyx_bases + poly, trimmed);
        x.flags |= RS_POLYX;
    }
    r.len = nl;
}

// ---- single-end loop body: seprocessor.cpp:204-296 ----
FQ_DEV void t_process_se(const TWave& W, int gp) {
    const TextArgs& E = *W.E;
    const KernelArgs& a = E.k;
    const DevParams& p = a.p;
    XRead r = {W.s1, W.q1, t_load(W, gp, 0, W.s1, W.q1), 0};
    XRes x = {0, 0, 0, 0};
    if (p.dup_enabled && a.dup_pos && !W.ghost) {   // :213-218, checkRead duplicate.cpp:122-134
        u64 h[MAX_DUP_BUFS] = {0, 0, 0, 0, 0, 0, 0, 0};
        t_hash(W, a, r.s, r.len, 0, h);
        t_leave_hash(W, a, gp, h);
    }
    if (E.hash_only) return;
    t_stat_read(W, 0, r.s, r.q, r.len);   // :210
    bool dedup_out = false;
    if (!W.ghost && a.dupflag && a.dupflag[gp]) {
        x.flags |= RS_DUP;
        dedup_out = p.dedup != 0;
    }
    if (p.umi_len1 > 0) t_trim_front(r, imin(r.len, p.umi_len1) + p.umi_skip);   // :232-233, umiprocessor.cpp:19-22
    int ft = 0;
    const bool alive = t_apply_trim_and_cut(W, p, r, p.trim_front1, p.trim_tail1, ft);   // :237
    if (alive && p.poly_g) r.len = t_trim_poly_g(W, r.s, r.len, p.poly_g_min);   // :239-242
    bool dimer = false;
    if (alive && p.adapter_enabled) {   // :244-261
        bool trimmed = false;
        if (p.has_a1) {
            int pos, al;
            if (t_apply_trim_by_sequence(W, r, W.ad1, p.alen1, 4, pos, al)) {
                trimmed = true;
                x.apos = (u32)pos;
                x.alen = (u32)al;
            }
        }
        if (p.n_fasta) trimmed |= t_trim_by_multi(W, r, (u32)(a.first + gp));   // :249-251
        if (trimmed) { t_add(W, E.c.adapter_reads, 1); x.flags |= RS_ADAPTER; }
        if (trimmed && r.len <= p.dimer_max_len) dimer = true;
    }
    if (alive && p.poly_x) t_poly_x(W, r, x);   // :263-266
    if (alive && p.max_len1 > 0 && p.max_len1 < r.len) t_resize(r, p.max_len1);   // :268-271
    int result = alive ? t_pass_filter(W, a, r.s, r.q, r.len) : 16;
    if (dimer) result = 28;
    t_add(W, E.c.filter + result, 1);   // :278
    if (!dedup_out && alive && result == 0) t_stat_read(W, 1, r.s, r.q, r.len);   // :280-290
    if (!alive) x.flags |= RS_NULL;
    if (!W.ghost) t_write_read(W, a, 0, gp, r, result, x);
}

// ---- paired-end loop body: peprocessor.cpp:383-643 ----
FQ_DEV void t_process_pe(const TWave& W, int gp) {
    const TextArgs& E = *W.E;
    const KernelArgs& a = E.k;
    const DevParams& p = a.p;
    const bool thread0 = (a.batch_flags & 1u) != 0;
    XRead r1 = {W.s1, W.q1, t_load(W, gp, 0, W.s1, W.q1), 0};
    XRead r2 = {W.s2, W.q2, t_load(W, gp, 1, W.s2, W.q2), 0};
    XRes x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
    if (p.dup_enabled && a.dup_pos && !W.ghost) {   // :397-402, checkPair duplicate.cpp:136-148
        u64 h[MAX_DUP_BUFS] = {0, 0, 0, 0, 0, 0, 0, 0};
        t_hash(W, a, r1.s, r1.len, 0, h);
        t_hash(W, a, r2.s, r2.len, r1.len, h);
        t_leave_hash(W, a, gp, h);
    }
    if (E.hash_only) return;
    t_stat_read(W, 0, r1.s, r1.q, r1.len);   // :393
    t_stat_read(W, 2, r2.s, r2.q, r2.len);   // :394
    bool dedup_out = false;
    if (!W.ghost && a.dupflag && a.dupflag[gp]) {
        x1.flags |= RS_DUP;
        x2.flags |= RS_DUP;
        dedup_out = p.dedup != 0;
    }
    if (p.umi_len1 > 0) t_trim_front(r1, imin(r1.len, p.umi_len1) + p.umi_skip);   // :419-420
    if (p.umi_len2 > 0) t_trim_front(r2, imin(r2.len, p.umi_len2) + p.umi_skip);
    int ft1 = 0, ft2 = 0;
    const bool a1 = t_apply_trim_and_cut(W, p, r1, p.trim_front1, p.trim_tail1, ft1);   // :425
    const bool a2 = t_apply_trim_and_cut(W, p, r2, p.trim_front2, p.trim_tail2, ft2);   // :426
    const bool both = a1 && a2;
    if (both && p.poly_g) {   // :428-431
        r1.len = t_trim_poly_g(W, r1.s, r1.len, p.poly_g_min);
        r2.len = t_trim_poly_g(W, r2.s, r2.len, p.poly_g_min);
    }
    bool isize_done = false, dimer = false;
    XOverlap ov = {0, 0, 0, 0, 0};
    bool ov_computed = false;
    const u16* lut = a.lut.ov_limit;
    if (both && (p.adapter_enabled || p.correction || thread0 || p.merge)) {   // :438-441
        ov = t_analyze(W, r1.s, r1.len, r2.s, r2.len, W.rc, lut, p.overlap_require, false);
        ov_computed = true;
    }
    if (both && (p.adapter_enabled || p.correction)) {   // :443-485
        const XOverlap ova = p.allow_gap ? t_analyze(W, r1.s, r1.len, r2.s, r2.len, W.rc, lut, p.overlap_require, true) : ov;
        if (thread0) { t_stat_isize(W, r1.len, r2.len, ov, ft1, ft2); isize_done = true; }
        if (p.correction && !ova.has_gap) t_correct(W, r1, r2, ova, gp, x1, x2);
        if (p.adapter_enabled) {
            bool trimmed = false;
            if (ova.overlapped && ova.offset < 0) {   // trimByOverlapAnalysis adaptertrimmer.cpp:17-46
                const int ol = ova.overlap_len;
                const int len1 = imin(r1.len, ol + ft2);
                const int len2 = imin(r2.len, ol + ft1);
                x1.apos = (u32)len1; x1.alen = (u32)(r1.len - len1);
                x2.apos = (u32)len2; x2.alen = (u32)(r2.len - len2);
                t_add(W, E.c.adapter_bases, (r1.len - len1) + (r2.len - len2));   // filterresult.cpp:154-155
                t_resize(r1, len1);
                t_resize(r2, len2);
                trimmed = true;
                x1.flags |= RS_ADAPTER_OV;
                x2.flags |= RS_ADAPTER_OV;
            }
            bool t1 = trimmed, t2 = trimmed;
            if (!trimmed) {   // :460-466
                int pos, al;
                if (p.has_a1) {
                    t1 = t_apply_trim_by_sequence(W, r1, W.ad1, p.alen1, 4, pos, al);
                    if (t1) { x1.apos = (u32)pos; x1.alen = (u32)al; }
                }
                if (p.has_a2) {
                    t2 = t_apply_trim_by_sequence(W, r2, W.ad2, p.alen2, 4, pos, al);
                    if (t2) { x2.apos = (u32)pos; x2.alen = (u32)al; }
                }
            }
            if (p.n_fasta) {   // :467-470
                t1 |= t_trim_by_multi(W, r1, 2u * (u32)(a.first + gp));
                t2 |= t_trim_by_multi(W, r2, 2u * (u32)(a.first + gp) + 1u);
            }
            if (t1) { t_add(W, E.c.adapter_reads, 1); x1.flags |= RS_ADAPTER; }   // :472-475
            if (t2) { t_add(W, E.c.adapter_reads, 1); x2.flags |= RS_ADAPTER; }
            if ((t1 || t2) && r1.len <= p.dimer_max_len && r2.len <= p.dimer_max_len) dimer = true;   // :480-484
        }
    }
    if (p.overlapped_out && both) {   // :488-495
        const XOverlap ovx = t_analyze(W, r1.s, r1.len, r2.s, r2.len, W.rc, nullptr, p.overlap_require, false);
        if (ovx.overlapped) {
            const int pos = imax(0, ovx.offset) + ovx.overlap_len;   // string(substr(start), overlap_len): the (str, pos) constructor
            x1.rsv = 0x8000u | (u32)pos;
            x2.rsv = (u32)(r1.len - pos);
        }
    }
    if (thread0 && !isize_done && both) {   // :497-504
        if (!ov_computed) { ov = t_analyze(W, r1.s, r1.len, r2.s, r2.len, W.rc, lut, p.overlap_require, false); ov_computed = true; }
        t_stat_isize(W, r1.len, r2.len, ov, ft1, ft2);
        isize_done = true;
    }
    if (both && p.poly_x) {   // :506-509
        t_poly_x(W, r1, x1);
        t_poly_x(W, r2, x2);
    }
    if (both) {   // :511-516
        if (p.max_len1 > 0 && p.max_len1 < r1.len) t_resize(r1, p.max_len1);
        if (p.max_len2 > 0 && p.max_len2 < r2.len) t_resize(r2, p.max_len2);
    }
    bool merge_done = false;
    int code1 = 0, code2 = 0;
    if (p.merge && both) {   // :518-561
        ov = t_analyze(W, r1.s, r1.len, r2.s, r2.len, W.rc, lut, p.overlap_require, false);
        if (ov.overlapped) {   // OverlapAnalysis::merge overlapanalysis.cpp:148-179
            const int ol = ov.overlap_len;
            const int len1 = ol + imax(0, ov.offset);
            const int len2 = ov.offset > 0 ? r2.len - ol : 0;
            const int m1 = imin(len1, r1.len);
            const int m2 = ov.offset > 0 ? imax(0, imin(len2, r2.len - ol)) : 0;
            const int mlen = m1 + m2;
            wave_sync();
            for (int k = W.lane; k < m1; k += 64) { W.ms[k] = r1.s[k]; W.mq[k] = r1.q[k]; }
            for (int k = W.lane; k < m2; k += 64) {   // rc(r2)[ol + k] = comp(r2[len2 - 1 - ol - k])
                const int src = r2.len - 1 - ol - k;
                W.ms[m1 + k] = t_complement(r2.s[src]);
                W.mq[m1 + k] = r2.q[src];
            }
            wave_sync();
            const int result = t_pass_filter(W, a, W.ms, W.mq, mlen);
            t_add(W, E.c.filter + result, 2);
            if (result == 0) {
                t_stat_read(W, 1, W.ms, W.mq, mlen);
                t_add(W, E.c.merged, 1);   // :688-690
                x1.flags |= RS_MERGED;
                x2.flags |= RS_MERGED;
            }
            code1 = code2 = result;
            if (!p.overlapped_out) { x1.rsv = (u32)m1; x2.rsv = (u32)m2; }
            merge_done = true;
        } else if (p.merge_include_unmerged) {
            code1 = t_pass_filter(W, a, r1.s, r1.q, r1.len);
            code2 = t_pass_filter(W, a, r2.s, r2.q, r2.len);
            if (dimer) code1 = code2 = 28;
            t_add(W, E.c.filter + code1, 1);
            if (code1 == 0 && !dedup_out) t_stat_read(W, 1, r1.s, r1.q, r1.len);
            t_add(W, E.c.filter + code2, 1);
            if (code2 == 0 && !dedup_out) t_stat_read(W, 1, r2.s, r2.q, r2.len);
            merge_done = true;
        }
    }
    if (!merge_done) {   // :563-621
        code1 = a1 ? t_pass_filter(W, a, r1.s, r1.q, r1.len) : 16;
        code2 = a2 ? t_pass_filter(W, a, r2.s, r2.q, r2.len) : 16;
        if (dimer) code1 = code2 = 28;
        t_add(W, E.c.filter + imax(code1, code2), 2);
        if (!dedup_out && a1 && code1 == 0 && a2 && code2 == 0 && !p.merge) {   // :588-591
            t_stat_read(W, 1, r1.s, r1.q, r1.len);
            t_stat_read(W, 3, r2.s, r2.q, r2.len);
        }
    }
    if (!a1) x1.flags |= RS_NULL;
    if (!a2) x2.flags |= RS_NULL;
    if (W.ghost) return;
    t_write_read(W, a, 0, gp, r1, code1, x1);
    t_write_read(W, a, 1, gp, r2, code2, x2);
    if (W.lane == 0) {
        a.pair[2 * (size_t)gp] = ((u32)ov.offset & 0xFFFFu) | (((u32)ov.overlap_len & 0xFFFFu) << 16);
        a.pair[2 * (size_t)gp + 1] = ((u32)ov.diff & 0xFFFFu) | ((u32)((ov.overlapped ? 1 : 0) | (ov.has_gap ? 2 : 0) | (isize_done ? 4 : 0)) << 16);
    }
}

// LDS: [the workgroup's Stats tables: lds_slots x lds_slot_dwords][a stretch of text_wave_bytes(ML) per wavefront]
FQ_DEV void text_body(const TextArgs& E, u32* lds) {
    const int lds_total = E.lds_slots * E.lds_slot_dwords;
    if (lds_total) {
        for (int i = thread_id(); i < lds_total; i += block_threads()) lds[i] = 0u;
        block_sync();
    }
    const int waves = block_threads() >> 6;
    const int ML = E.ML;
    u8* base = (u8*)(lds + lds_total) + (size_t)wave_id() * text_wave_bytes(ML);
    TWave W;
    W.E = &E;
    W.tables = lds_total ? lds : nullptr;
    W.s1 = base;
    W.q1 = base + ML;
    W.s2 = base + 2 * ML;
    W.q2 = base + 3 * ML;
    W.rc = base + 4 * ML;
    W.ms = base + 5 * ML;    // 2 ML
    W.mq = base + 7 * ML;    // 2 ML
    W.ad1 = base + (size_t)TEXT_BUFS * ML;
    W.ad2 = W.ad1 + 264;
    W.adf = W.ad2 + 264;
    W.lane = lane_id();
    W.sign = 1;
    W.ghost = false;
    if (!E.hash_only) {
        if (E.k.p.has_a1) t_decode_adapter(W, E.k.p.a1w, E.k.p.alen1, W.ad1);
        if (E.k.p.has_a2) t_decode_adapter(W, E.k.p.a2w, E.k.p.alen2, W.ad2);
    }
    const int gw = block_id() * waves + wave_id(), nw = grid_blocks() * waves;
    if (gw == 0 && !E.hash_only && E.x_count > 0) {   // what the plan's kernels counted for the empty units they saw in these places
        TWave G = W;
        G.sign = -(long long)E.x_count;
        G.ghost = true;
        G.tables = nullptr;
        if (E.k.p.paired) t_process_pe(G, 0);
        else t_process_se(G, 0);
    }
    for (int i = gw; i < E.x_count; i += nw) {
        const int gp = E.x_all ? i : E.x_unit[E.x_k0 + i] - E.k.first;
        wave_sync();
        if (E.k.p.paired) t_process_pe(W, gp);
        else t_process_se(W, gp);
    }
    if (lds_total) {   // the workgroup's tables into the counter block: [slot][34 * cycles | 5-mers | quality characters]
        block_sync();
        const int C34 = 34 * (int)E.c.cycles;
        for (int i = thread_id(); i < lds_total; i += block_threads()) {
            const u32 v = lds[i];
            if (!v) continue;
            const int slot = i / E.lds_slot_dwords, r = i - slot * E.lds_slot_dwords;
            const long long st = E.c.stats[slot];
            const long long off = r < C34 ? st + E.c.st_cycle + r : r < C34 + 1024 ? st + E.c.st_kmer + (r - C34) : st + E.c.st_qual_hist + (r - C34 - 1024);
            g_atomic_add_i64(E.ctr + off, (int64_t)v);
        }
    }
}

}  // namespace fq
