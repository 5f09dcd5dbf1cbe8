// fq_exact.h - the worker loop on the reads' TEXT, one lane per unit ("exact plan").
//
// The lane / split / fused plans work on the packed batch: 2 bits per base + an N flag.  That is everything the loop
// ever looks at while the letters are A, C, G, T, N - and not enough for any other byte (lower case, IUPAC codes, '.'):
// the reference treats such a byte differently in nearly every consumer (DESIGN.md section 1 has the table):
//   Stats::statRead bins it by `base & 7` (stats.cpp:206-222), the 5-mer window drops it (:229-266), Duplicate hashes it
//   as 13 (duplicate.cpp:92-109), reverseComplement turns a/c/g/t into T/G/C/A and everything else into N
//   (util.h:16-33, simd.cpp:296-310), overlap analysis / adapter matching / polyG / polyX / the complexity filter compare
//   raw bytes, passFilter and trimAndCut count only a literal 'N'.
// Units that hold such a byte ("exotic" units, fastp_gpu_batch::exotic_*) therefore go through THIS kernel: a lane owns a
// unit, rebuilds its text in a private stretch of HBM (exotic units: the raw bytes; any other unit: decoded from the
// packed rows) and runs the loop body as the reference wrote it, byte by byte, counters straight into the context's
// int64 block with global atomics.  Nothing here is fast and nothing has to be: one launch per batch launch, over the
// listed units only (FASTP_GPU_EXACT=1, tests: over every unit).
//
// How it composes with the plan's kernels (fastp_gpu.hip launch_chunk): those still sweep the whole launch, but on a
// copy of the length arrays in which the listed units are EMPTY reads.  What an empty unit adds to the counters is
// exactly what this loop computes for an empty unit - so a lane first runs the loop on an empty unit with sign -1
// (the "ghost" pass: counters only), then on the real unit with sign +1, and overwrites the unit's records and hash
// values.  Duplicate's bloom semantics stay with the fq_dup_* kernels, which run once over the whole launch afterwards:
// this kernel leaves the COMPLETE hash values of its units (the position part included - the dup kernels add the
// position sum of the length they see, zero).
//
// Each function cites the reference lines it follows.
#pragma once
#include "fq_device.h"

namespace fq {

struct ExactCtr {   // offsets into the int64 counter block (fastp_gpu_counter_layout)
    long long filter, adapter_reads, adapter_bases, polyx_reads, polyx_bases, correction, corrected_reads, merged, isize;
    long long stats[4];
    long long st_reads, st_length_sum, st_qual_hist, st_kmer, st_cycle, cycles;
};

struct ExactArgs {
    KernelArgs k;        // parameters, LUTs, the launch's rows / records / lists (pointers already at `first`)
    ExactCtr c;
    int64_t* ctr;
    // which units: the entries [x_k0, x_k0 + x_count) of the batch's list, or (x_all) the launch's units 0 .. x_count - 1
    int x_all, x_k0, x_count;
    int sign, ghost;     // ghost pass: an empty unit, counters only, sign -1 (set per lane, see exact_body)
    // raw sequence bytes of the exotic units
    const int* x_unit;   // [x_n] ascending unit indexes inside the submitted batch
    int x_n;
    int x_dense;         // 1: x_off[m] is fastp_gpu_parse_fastq's line table of mate m ([4 * unit + 1] = the sequence line)
    const u8* x_text[2];
    const u32* x_off[2]; // [x_n] byte offset of the unit's sequence in x_text[m] (or the dense table)
    u8* scratch;         // [lanes][lane_bytes]
    u32 lane_bytes;
    int ML;              // bytes per text buffer of a lane (max_len + slack, multiple of 8)
    int hash_only;       // --dedup's pre-pass: leave the hash values, nothing else
    // Stats::statRead's per-base counters of the workgroup's units are gathered in LDS (u32: [slot][34 * cycles | 1024 5-mers |
    // 128 quality characters]) and added to the block once at the end - with every lane of a launch adding to the same few
    // thousand int64s in HBM the kernel ran at the atomics' rate.  0: the tables do not fit (merge mode of long reads)
    int lds_slot_dwords;
    int lds_slots;       // Stats objects that can be touched: 2 single-end, 3 merge mode (nothing reaches POST2), else 4
};

enum { EXACT_BUFS = 14, EXACT_ADAPTER_BYTES = 3 * 264 };   // text buffers of ML bytes per lane; three decoded adapters

struct XRead {
    u8* s;
    u8* q;
    int len;
    int front;
};

struct ExactMaskArgs {
    const int* units;   // listed units of this launch (batch indexes)
    int count, first;
    u16* len[2];        // the launch's copies of the length arrays
    u8* skip;           // [n] or nullptr: 1 for the listed units (KernelArgs::xskip)
};
FQ_DEV void exact_mask_body(const ExactMaskArgs& m) {
    const int i = block_id() * block_threads() + thread_id();
    if (i >= m.count) return;
    const int gp = m.units[i] - m.first;
    m.len[0][gp] = 0;
    if (m.len[1]) m.len[1][gp] = 0;
    if (m.skip) m.skip[gp] = 1;
}

FQ_DEV void x_add(const ExactArgs& E, long long off, long long v) { g_atomic_add_i64(E.ctr + off, (int64_t)(v * E.sign)); }

FQ_DEV u8 x_complement(u8 b) {   // util.h:16-33
    switch (b) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}
FQ_DEV int x_base2val(u8 b) {   // stats.cpp:294-311
    switch (b) {
        case 'A': return 0;
        case 'T': return 1;
        case 'C': return 2;
        case 'G': return 3;
        default: return -1;
    }
}
FQ_DEV u64 x_hash_val(u8 b) {   // duplicate.cpp:92-109
    switch (b) {
        case 'A': return 7;
        case 'T': return 222;
        case 'C': return 74;
        case 'G': return 31;
        default: return 13;
    }
}
FQ_DEV u8 x_code_ascii(u32 code) { return (u8)("ATCG"[code & 3u]); }

// the raw sequence bytes of unit gp's mate m, or null when the unit is not exotic
FQ_DEV const u8* x_raw(const ExactArgs& E, int gp, int m) {
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

FQ_DEV int x_load(const ExactArgs& E, int gp, int m, u8* s, u8* q) {
    if (E.ghost) { s[0] = 0; q[0] = 0; return 0; }
    const int len = (int)E.k.len[m][gp];
    const u8* qrow = (const u8*)(E.k.qual[m] + (size_t)gp * E.k.p.qw_g);
    const u32* srow = E.k.seq[m] + (size_t)gp * E.k.p.sw_g;
    const u8* raw = x_raw(E, gp, m);
    for (int i = 0; i < len; i++) {
        const u8 qq = qrow[i];
        q[i] = (u8)(qq & 0x7Fu);
        s[i] = raw ? raw[i] : ((qq & 0x80u) ? (u8)'N' : x_code_ascii(srow[i >> 4] >> ((i & 15) * 2)));
    }
    s[len] = 0;
    q[len] = 0;
    return len;
}

FQ_DEV void x_decode_adapter(const u32* words, int alen, u8* out) {
    for (int i = 0; i < alen; i++) out[i] = x_code_ascii(words[i >> 4] >> ((i & 15) * 2));
    out[alen] = 0;
}

// ---- Stats::statRead (stats.cpp:191-291, without the overrepresentation part) ----
FQ_DEV void x_stat_read_lds(const ExactArgs& E, u32* lds, int slot, const u8* s, const u8* q, int len);
FQ_DEV void x_stat_read(const ExactArgs& E, u32* lds, int slot, const u8* s, const u8* q, int len) {
    const long long st = E.c.stats[slot];
    const long long C = E.c.cycles;
    const long long cyc = st + E.c.st_cycle;
    const long long q30 = cyc, q20 = cyc + 8 * C, cont = cyc + 16 * C, qual = cyc + 24 * C, tot_base = cyc + 32 * C, tot_qual = cyc + 33 * C;
    x_add(E, st + E.c.st_length_sum, len);
    if (E.lds_slot_dwords && E.sign > 0) {
        x_stat_read_lds(E, lds, slot, s, q, len);
        x_add(E, st + E.c.st_reads, 1);
        return;
    }
    int kmer = 0;
    bool need_full = true;
    for (int i = 0; i < len; i++) {
        const u8 base = s[i], qc = q[i];
        const int b = base & 7;
        x_add(E, st + E.c.st_qual_hist + qc, 1);
        if (qc >= '?') { x_add(E, q30 + b * C + i, 1); x_add(E, q20 + b * C + i, 1); }
        else if (qc >= '5') x_add(E, q20 + b * C + i, 1);
        x_add(E, cont + b * C + i, 1);
        x_add(E, qual + b * C + i, (int)qc - 33);
        x_add(E, tot_base + i, 1);
        x_add(E, tot_qual + i, (int)qc - 33);
        if (base == 'N') { need_full = true; continue; }
        if (i < 4) continue;
        if (!need_full) {
            const int v = x_base2val(base);
            if (v < 0) { need_full = true; continue; }
            kmer = ((kmer << 2) & 0x3FC) | v;
            x_add(E, st + E.c.st_kmer + kmer, 1);
        } else {
            bool valid = true;
            kmer = 0;
            for (int k = 0; k < 5; k++) {
                const int v = x_base2val(s[i - 4 + k]);
                if (v < 0) { valid = false; break; }
                kmer = ((kmer << 2) & 0x3FC) | v;
            }
            if (!valid) { need_full = true; continue; }
            x_add(E, st + E.c.st_kmer + kmer, 1);
            need_full = false;
        }
    }
    x_add(E, st + E.c.st_reads, 1);
}

// the same walk with the per-base counters in the workgroup's LDS tables (exact_body folds them into the block)
FQ_DEV void x_stat_read_lds(const ExactArgs& E, u32* lds, int slot, const u8* s, const u8* q, int len) {
    const int C = (int)E.c.cycles;
    u32* t = lds + (size_t)slot * E.lds_slot_dwords;
    u32* q30 = t, *q20 = t + 8 * C, *cont = t + 16 * C, *qual = t + 24 * C, *tot_base = t + 32 * C, *tot_qual = t + 33 * C;
    u32* kmers = t + 34 * C;
    u32* hist = kmers + 1024;
    int kmer = 0;
    bool need_full = true;
    for (int i = 0; i < len; i++) {
        const u8 base = s[i], qc = q[i];
        const int b = base & 7;
        lds_add_u32(&hist[qc], 1u);
        if (qc >= '?') { lds_add_u32(&q30[b * C + i], 1u); lds_add_u32(&q20[b * C + i], 1u); }
        else if (qc >= '5') lds_add_u32(&q20[b * C + i], 1u);
        lds_add_u32(&cont[b * C + i], 1u);
        lds_add_u32(&qual[b * C + i], (u32)((int)qc - 33));
        lds_add_u32(&tot_base[i], 1u);
        lds_add_u32(&tot_qual[i], (u32)((int)qc - 33));
        if (base == 'N') { need_full = true; continue; }
        if (i < 4) continue;
        if (!need_full) {
            const int v = x_base2val(base);
            if (v < 0) { need_full = true; continue; }
            kmer = ((kmer << 2) & 0x3FC) | v;
            lds_add_u32(&kmers[kmer], 1u);
        } else {
            bool valid = true;
            kmer = 0;
            for (int k = 0; k < 5; k++) {
                const int v = x_base2val(s[i - 4 + k]);
                if (v < 0) { valid = false; break; }
                kmer = ((kmer << 2) & 0x3FC) | v;
            }
            if (!valid) { need_full = true; continue; }
            lds_add_u32(&kmers[kmer], 1u);
            need_full = false;
        }
    }
}

// ---- Filter::trimAndCut (filter.cpp:68-207); false = NULL ----
FQ_DEV bool x_trim_and_cut(const DevParams& p, const u8* seq, const u8* q, int l, int front, int tail, int& out_front, int& out_len) {
    const bool enF = p.cut_front != 0, enT = p.cut_tail != 0, enR = p.cut_right != 0;
    out_front = 0;
    out_len = l;
    if (front == 0 && tail == 0 && !enF && !enT && !enR) return true;   // :71-72
    int rlen = l - front - tail;
    if (rlen < 0) return false;   // :76-77
    if (!enF && !enT && !enR) { out_front = front; out_len = rlen; return true; }   // :79-89
    if (enF) {   // :97-127
        const int w = p.wF;
        int s = front;
        if (l - front - tail - w <= 0) return false;
        int total = 0;
        for (int i = 0; i < w - 1; i++) total += q[s + i];
        for (s = front; s + w < l - tail; s++) {
            total += q[s + w - 1];
            if (s > front) total -= q[s - 1];
            if (total >= p.thrF) break;
        }
        if (s > 0) s = s + w - 1;
        while (s < l && seq[s] == 'N') s++;
        front = s;
        rlen = l - front - tail;
    }
    if (enR) {   // :130-163
        const int w = p.wR;
        int s = front;
        if (l - front - tail - w <= 0) return false;
        int total = 0;
        for (int i = 0; i < w - 1; i++) total += q[s + i];
        bool found = false;
        for (s = front; s + w < l - tail; s++) {
            total += q[s + w - 1];
            if (s > front) total -= q[s - 1];
            if (total < p.thrR) { found = true; break; }
        }
        if (found) {
            while (s < l - 1 && q[s] >= p.qRmin) s++;
            rlen = s - front;
        }
    }
    if (!enR && enT) {   // :166-194
        const int w = p.wT;
        if (l - front - tail - w <= 0) return false;
        int total = 0;
        int t = l - tail - 1;
        for (int i = 0; i < w - 1; i++) total += q[t - i];
        for (t = l - tail - 1; t - w >= front; t--) {
            total += q[t - w + 1];
            if (t < l - tail - 1) total -= q[t + 1];
            if (total >= p.thrT) break;
        }
        if (t < l - 1) t = t - w + 1;
        while (t >= 0 && seq[t] == 'N') t--;
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return false;   // :196-197
    out_front = front;
    out_len = rlen;
    return true;
}

FQ_DEV bool x_apply_trim_and_cut(const DevParams& p, XRead& r, int front, int tail, int& ft) {
    int f = 0, l = 0;
    ft = 0;
    if (!x_trim_and_cut(p, r.s, r.q, r.len, front, tail, f, l)) return false;
    r.s += f;
    r.q += f;
    r.front += f;
    r.len = l;
    ft = f;
    return true;
}

FQ_DEV void x_trim_front(XRead& r, int len) {   // Read::trimFront read.cpp:69-73
    len = imin(r.len - 1, len);
    if (len < 0) return;
    r.s += len;
    r.q += len;
    r.len -= len;
    r.front += len;
}
FQ_DEV void x_resize(XRead& r, int len) {   // Read::resize read.cpp:62-67
    if (len > r.len || len < 0) return;
    r.len = len;
}

// ---- PolyX::trimPolyG (polyx.cpp:16-42) ----
FQ_DEV int x_trim_poly_g(const u8* d, int rlen, int compare_req) {
    int mismatch = 0, i = 0, first_g = rlen - 1;
    for (i = 0; i < rlen; i++) {
        if (d[rlen - i - 1] != 'G') mismatch++;
        else first_g = rlen - i - 1;
        const int allowed = (i + 1) / 8;
        if (mismatch > 5 || (mismatch > allowed && i >= compare_req - 1)) break;
    }
    if (i >= compare_req && first_g >= 0 && first_g <= rlen) return first_g;
    return rlen;
}

// ---- PolyX::trimPolyX (polyx.cpp:49-116) ----
FQ_DEV int x_trim_poly_x(const u8* d, int rlen, int compare_req, int& poly_base, int& trimmed) {
    int cnt[4] = {0, 0, 0, 0};
    int pos = 0;
    poly_base = -1;
    trimmed = 0;
    for (pos = 0; pos < rlen; pos++) {
        const u8 c = d[rlen - pos - 1];
        if (c == 'A') cnt[0]++;
        else if (c == 'T') cnt[1]++;
        else if (c == 'C') cnt[2]++;
        else if (c == 'G') cnt[3]++;
        else if (c == 'N') { cnt[0]++; cnt[1]++; cnt[2]++; cnt[3]++; }
        const int cmp = pos + 1;
        const int allowed = imin(5, cmp / 8);
        bool need_break = true;
        for (int b = 0; b < 4; b++)
            if (cmp - cnt[b] <= allowed) need_break = false;
        if (need_break && (pos >= 8 || pos + 1 >= compare_req - 1)) break;
    }
    if (pos + 1 >= compare_req) {   // :98-115
        int poly = 0, max_count = -1;
        for (int b = 0; b < 4; b++)
            if (cnt[b] > max_count) { max_count = cnt[b]; poly = b; }
        const u8 pb = x_code_ascii((u32)poly);
        // :109 while (data[rlen - pos - 1] != polyBase && pos >= 0) pos--;  index -1 counts as "not the base" (the
        // scan never broke), index rlen is the string's terminating zero
        for (;;) {
            const int idx = rlen - pos - 1;
            const u8 c = (idx < 0 || idx >= rlen) ? (u8)0 : d[idx];
            if (!(c != pb && pos >= 0)) break;
            pos--;
        }
        const int newlen = rlen - pos - 1;
        poly_base = poly;
        trimmed = pos + 1;
        if (newlen < 0 || newlen > rlen) return rlen;
        return newlen;
    }
    return rlen;
}

FQ_DEV int x_mismatches(const u8* a, const u8* b, int len) {   // simd.cpp:319-324
    int d = 0;
    for (int i = 0; i < len; i++) d += (a[i] != b[i]);
    return d;
}
FQ_DEV int x_mismatches_bounded(const u8* a, const u8* b, int len, int limit) {   // simd.cpp:209-233
    int d = 0;
    for (int i = 0; i < len; i++) {
        d += (a[i] != b[i]);
        if (d > limit) return d;
    }
    return d;
}

// ---- Matcher (matcher.cpp:10-100) ----
FQ_DEV void x_matcher_tables(const u8* ins, const u8* nor, int cmplen, int limit, short* L, short* R) {
    for (int i = 0; i < cmplen; i++) { L[i] = 0; R[i] = 0; }
    L[0] = ins[0] == nor[0] ? 0 : 1;
    R[cmplen - 1] = ins[cmplen] == nor[cmplen - 1] ? 0 : 1;
    for (int i = 1; i < cmplen; i++) {
        L[i] = (short)(L[i - 1] + (ins[i] != nor[i] ? 1 : 0));
        if (L[i] + R[cmplen - 1] > limit) break;
    }
    for (int i = cmplen - 2; i >= 0; i--) {
        R[i] = (short)(R[i + 1] + (ins[i + 1] != nor[i] ? 1 : 0));
        if (R[i] + L[0] > limit) {
            for (int p = 0; p < i; p++) R[p] = (short)(limit + 1);
            break;
        }
    }
}
FQ_DEV bool x_match_one_insertion(const u8* ins, const u8* nor, int cmplen, int limit, short* L, short* R) {   // :10-54
    if (cmplen <= 0) return false;
    x_matcher_tables(ins, nor, cmplen, limit, L, R);
    for (int i = 1; i < cmplen; i++) {
        if (L[i - 1] + R[cmplen - 1] > limit) return false;
        if (L[i - 1] + R[i] <= limit) return true;
    }
    return false;
}
FQ_DEV int x_diff_one_insertion(const u8* ins, const u8* nor, int cmplen, int limit, short* L, short* R) {   // :56-100
    if (cmplen <= 0) return 100000000;
    x_matcher_tables(ins, nor, cmplen, limit, L, R);
    int min_diff = 100000000;
    for (int i = 1; i < cmplen; i++) {
        if (L[i - 1] + R[cmplen - 1] > limit) return -1;
        const int d = L[i - 1] + R[i];
        if (d <= min_diff) min_diff = d;
    }
    return min_diff;
}

// ---- OverlapAnalysis::analyze (overlapanalysis.cpp:17-146) ----
struct XOverlap {
    int overlapped, offset, overlap_len, diff, has_gap;
};
FQ_DEV bool x_accept_nogap(const u8* a, const u8* b, int len, int limit, int& diff) {   // :34-44
    const int prefix = imin(len, 50);
    diff = x_mismatches_bounded(a, b, prefix, limit);
    if (diff > limit) return false;
    if (len > 50) diff = x_mismatches(a, b, len);
    return true;
}
// `lut` = min(diffLimit, (int)(overlap_len * diffPercentLimit)) per overlap length, or null for diffPercentLimit 0
FQ_DEV XOverlap x_analyze(const u8* r1, int len1, const u8* r2, int len2, u8* rc, const u16* lut, int overlap_require, bool allow_gap,
                          short* L, short* R) {
    XOverlap ov = {0, 0, 0, 0, 0};
    for (int i = 0; i < len2; i++) rc[len2 - 1 - i] = x_complement(r2[i]);   // :19-22
    rc[len2] = 0;
    int overlap_len = 0, offset = 0, diff = 0;
    while (offset < len1 - overlap_require) {   // :48-64
        overlap_len = imin(len1 - offset, len2);
        const int limit = lut ? (int)lut[overlap_len] : 0;
        if (x_accept_nogap(r1 + offset, rc, overlap_len, limit, diff)) {
            ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = diff;
            return ov;
        }
        offset += 1;
    }
    offset = 0;
    while (offset > -(len2 - overlap_require)) {   // :72-89
        overlap_len = imin(len1, len2 + offset);
        const int limit = lut ? (int)lut[overlap_len] : 0;
        if (x_accept_nogap(r1, rc - offset, overlap_len, limit, diff)) {
            ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = diff;
            return ov;
        }
        offset -= 1;
    }
    if (allow_gap) {   // :91-139
        offset = 0;
        while (offset < len1 - overlap_require) {
            overlap_len = imin(len1 - offset, len2);
            const int limit = lut ? (int)lut[overlap_len] : 0;
            int d = x_diff_one_insertion(r1 + offset, rc, overlap_len - 1, limit, L, R);
            if (d < 0 || d > limit) d = x_diff_one_insertion(rc, r1 + offset, overlap_len - 1, limit, L, R);
            if (d <= limit && d >= 0) {
                ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = d; ov.has_gap = 1;
                return ov;
            }
            offset += 1;
        }
        offset = 0;
        while (offset > -(len2 - overlap_require)) {
            overlap_len = imin(len1, len2 + offset);
            const int limit = lut ? (int)lut[overlap_len] : 0;
            int d = x_diff_one_insertion(r1, rc - offset, overlap_len - 1, limit, L, R);
            if (d < 0 || d > limit) d = x_diff_one_insertion(rc - offset, r1, overlap_len - 1, limit, L, R);
            if (d <= limit && d >= 0) {
                ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = d; ov.has_gap = 1;
                return ov;
            }
            offset -= 1;
        }
    }
    return ov;
}

// ---- AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157) ----
FQ_DEV bool x_trim_by_sequence(const u8* rd, int rlen, const u8* ad, int alen, int match_req, int& out_pos, short* L, short* R) {
    if (alen < match_req) return false;
    int pos = 0, start = 0;
    bool found = false;
    if (alen >= 16) start = -4;
    else if (alen >= 12) start = -3;
    else if (alen >= 8) start = -2;
    for (pos = start; pos < rlen - match_req; pos++) {   // :87-100
        const int cmplen = imin(rlen - pos, alen);
        const int allowed = cmplen / 8;
        const int so = imax(0, -pos);
        const int mm = x_mismatches_bounded(ad + so, rd + so + pos, cmplen - so, allowed);
        if (mm <= allowed) { found = true; break; }
    }
    if (!found) {   // :105-118, rdata / adata WITHOUT + pos as in the reference
        for (pos = 0; pos < rlen - match_req - 1; pos++) {
            const int cmplen = imin(rlen - pos - 1, alen);
            const int allowed = cmplen / 8 - 1;
            if (x_match_one_insertion(rd, ad, cmplen, allowed, L, R)) { found = true; break; }
        }
    }
    if (!found) {   // :122-135
        for (pos = 0; pos < rlen - match_req; pos++) {
            const int cmplen = imin(rlen - pos, alen - 1);
            const int allowed = cmplen / 8 - 1;
            if (x_match_one_insertion(ad, rd, cmplen, allowed, L, R)) { found = true; break; }
        }
    }
    out_pos = pos;
    return found;
}

// trimBySequence on the read + FilterResult::addAdapterTrimmed (filterresult.cpp:124-152)
FQ_DEV bool x_apply_trim_by_sequence(const ExactArgs& E, XRead& r, const u8* ad, int alen, int match_req, int& pos, int& adapter_len,
                                     short* L, short* R) {
    pos = 0;
    if (!x_trim_by_sequence(r.s, r.len, ad, alen, match_req, pos, L, R)) return false;
    if (pos < 0) {   // adaptertrimmer.cpp:138-145
        adapter_len = alen + pos;
        r.len = 0;
    } else {
        adapter_len = r.len - pos;
        x_resize(r, pos);
    }
    if (adapter_len > 0) x_add(E, E.c.adapter_bases, adapter_len);
    return true;
}

// AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62): one event per cut for the host's adapter-string replay
FQ_DEV bool x_trim_by_multi(const ExactArgs& E, XRead& r, u32 read_index, u8* abuf, short* L, short* R) {
    const KernelArgs& a = E.k;
    bool trimmed = false;
    for (int i = 0; i < a.p.n_fasta; i++) {
        const int alen = a.lut.fasta_len[i];
        x_decode_adapter(a.lut.fasta_words + (size_t)i * ADAPT_WORDS, alen, abuf);
        int pos, adapter_len;
        if (!x_apply_trim_by_sequence(E, r, abuf, alen, a.p.fasta_match_req, pos, adapter_len, L, R)) continue;
        trimmed = true;
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

// ---- Filter::passFilter (filter.cpp:15-66), thresholds through the host-built LUTs (the reference's double expressions) ----
FQ_DEV int x_pass_filter(const KernelArgs& a, const u8* s, const u8* q, int rlen) {
    const DevParams& p = a.p;
    if (rlen == 0) return 16;
    int low = 0, nb = 0, total = 0;
    if (p.qual_filter || p.length_filter) {   // :26-33
        for (int i = 0; i < rlen; i++) {
            total += (int)q[i] - 33;
            if ((int)q[i] < p.qual_thr) low++;
            if (s[i] == 'N') nb++;
        }
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
        for (int i = 0; i < rlen - 1; i++) diff += (s[i] != s[i + 1]);
        if (diff < (int)a.lut.cplx_min[rlen]) return 24;
    }
    return 0;
}

// ---- Duplicate::seq2intvector (duplicate.cpp:111-120), the whole value (the dup kernels add the position sum of the
// length THEY see for this unit: zero, it is an empty read in their length arrays) ----
FQ_DEV void x_hash(const KernelArgs& a, const u8* s, int len, int pos_offset, u64* out) {
    const int B = a.p.dup_bufnum;
    const u32 mask = (u32)(512 * B - 1);
    for (int p = 0; p < len; p++) {
        const u64 v = x_hash_val(s[p]) + (u64)(p + pos_offset);
        for (int i = 0; i < B; i++) out[i] += (u64)a.lut.dup_primes[((u32)((p + pos_offset) * B + i)) & mask] * v;
    }
}

struct XRes {
    u32 flags, apos, alen, rsv;
};
FQ_DEV void x_write_read(const KernelArgs& a, int m, int gp, const XRead& r, int code, const XRes& x) {
    u32* out = a.res[m] + (size_t)gp * 3;
    out[0] = ((u32)r.front & 0xFFFFu) | (((u32)r.len & 0xFFFFu) << 16);
    out[1] = ((u32)code & 0xFFu) | ((x.flags & 0xFFu) << 8) | ((x.apos & 0xFFFFu) << 16);
    out[2] = (x.alen & 0xFFFFu) | ((x.rsv & 0xFFFFu) << 16);
}

FQ_DEV void x_stat_isize(const ExactArgs& E, int l1, int l2, const XOverlap& ov, int ft1, int ft2) {   // peprocessor.cpp:710-723
    int isize = E.k.p.isize_max;
    if (ov.overlapped) {
        if (ov.offset > 0) isize = l1 + l2 - ov.overlap_len + ft1 + ft2;
        else isize = ov.overlap_len + ft1 + ft2;
    }
    if (isize > E.k.p.isize_max) isize = E.k.p.isize_max;
    if (isize < 0) return;
    x_add(E, E.c.isize + isize, 1);
}

// ---- BaseCorrector::correctByOverlapAnalysis (basecorrector.cpp:16-83) ----
FQ_DEV void x_correct(const ExactArgs& E, XRead& r1, XRead& r2, const XOverlap& ov, int gp, XRes& x1, XRes& x2) {
    const KernelArgs& a = E.k;
    if (ov.diff == 0 || !ov.overlapped) return;
    const int ol = ov.overlap_len;
    const int start1 = imax(0, ov.offset);
    const int start2 = r2.len - imax(0, -ov.offset) - 1;
    const u8 GOOD = (u8)(30 + 33), BAD = (u8)(14 + 33);
    int corrected = 0;
    bool r1c = false, r2c = false;
    for (int i = 0; i < ol; i++) {
        const int p1 = start1 + i, p2 = start2 - i;
        if (r1.s[p1] == x_complement(r2.s[p2])) continue;
        int which = -1, pos = 0;
        u8 nb = 0, nq = 0, from = 0;
        if (r1.q[p1] >= GOOD && r2.q[p2] <= BAD) {
            from = r2.s[p2];
            nb = x_complement(r1.s[p1]);
            nq = r1.q[p1];
            r2.s[p2] = nb;
            r2.q[p2] = nq;
            which = 1;
            pos = r2.front + p2;
            r2c = true;
        } else if (r2.q[p2] >= GOOD && r1.q[p1] <= BAD) {
            from = r1.s[p1];
            nb = x_complement(r2.s[p2]);
            nq = r2.q[p2];
            r1.s[p1] = nb;
            r1.q[p1] = nq;
            which = 0;
            pos = r1.front + p1;
            r1c = true;
        }
        if (which < 0) continue;
        corrected++;
        x_add(E, E.c.correction + (from & 7) * 8 + (nb & 7), 1);   // addCorrection filterresult.cpp:99-103
        if (a.corrections) {
            const int slot = g_atomic_add_i32(a.n_corrections, 1);
            if (slot < a.corr_capacity) {
                a.corrections[2 * slot] = 2u * (u32)(a.first + gp) + (u32)which;
                a.corrections[2 * slot + 1] = ((u32)pos & 0xFFFFu) | ((u32)nb << 16) | ((u32)nq << 24);
            }
        }
    }
    if (corrected > 0) {   // :75-80
        x_add(E, E.c.corrected_reads, (r1c && r2c) ? 2 : 1);
        if (r1c) x1.flags |= RS_CORRECTED;
        if (r2c) x2.flags |= RS_CORRECTED;
    }
}

struct XBufs {
    u8 *s1, *q1, *s2, *q2, *rc, *ms, *mq, *ad1, *ad2, *adf;
    short *L, *R;
};

FQ_DEV void x_poly_x(const ExactArgs& E, XRead& r, XRes& x) {
    int poly, trimmed;
    const int nl = x_trim_poly_x(r.s, r.len, E.k.p.poly_x_min, poly, trimmed);
    if (poly >= 0) {   // addPolyXTrimmed filterresult.cpp:186-189
        x_add(E, E.c.polyx_reads + poly, 1);
        x_add(E, E.c.polyx_bases + poly, trimmed);
        x.flags |= RS_POLYX;
    }
    r.len = nl;
}

// ---- single-end loop body: seprocessor.cpp:204-296 ----
FQ_DEV void x_process_se(const ExactArgs& E, u32* lds, int gp, const XBufs& b) {
    const KernelArgs& a = E.k;
    const DevParams& p = a.p;
    XRead r = {b.s1, b.q1, x_load(E, gp, 0, b.s1, b.q1), 0};
    XRes x = {0, 0, 0, 0};
    if (p.dup_enabled && a.dup_pos && !E.ghost) {   // :213-218, checkRead duplicate.cpp:122-134
        u64 h[MAX_DUP_BUFS] = {0, 0, 0, 0, 0, 0, 0, 0};
        x_hash(a, r.s, r.len, 0, h);
        for (int i = 0; i < p.dup_bufnum; i++) a.dup_pos[(size_t)gp * p.dup_bufnum + i] = h[i];
    }
    if (E.hash_only) return;
    x_stat_read(E, lds, 0, r.s, r.q, r.len);   // :210
    bool dedup_out = false;
    if (a.dupflag && a.dupflag[gp]) {
        x.flags |= RS_DUP;
        dedup_out = p.dedup != 0;
    }
    if (p.umi_len1 > 0) x_trim_front(r, imin(r.len, p.umi_len1) + p.umi_skip);   // :232-233, umiprocessor.cpp:19-22
    int ft = 0;
    const bool alive = x_apply_trim_and_cut(p, r, p.trim_front1, p.trim_tail1, ft);   // :237
    if (alive && p.poly_g) r.len = x_trim_poly_g(r.s, r.len, p.poly_g_min);   // :239-242
    bool dimer = false;
    if (alive && p.adapter_enabled) {   // :244-261
        bool trimmed = false;
        if (p.has_a1) {
            int pos, al;
            if (x_apply_trim_by_sequence(E, r, b.ad1, p.alen1, 4, pos, al, b.L, b.R)) {
                trimmed = true;
                x.apos = (u32)pos;
                x.alen = (u32)al;
            }
        }
        if (p.n_fasta) trimmed |= x_trim_by_multi(E, r, (u32)(a.first + gp), b.adf, b.L, b.R);   // :249-251
        if (trimmed) { x_add(E, E.c.adapter_reads, 1); x.flags |= RS_ADAPTER; }
        if (trimmed && r.len <= p.dimer_max_len) dimer = true;
    }
    if (alive && p.poly_x) x_poly_x(E, r, x);   // :263-266
    if (alive && p.max_len1 > 0 && p.max_len1 < r.len) x_resize(r, p.max_len1);   // :268-271
    int result = alive ? x_pass_filter(a, r.s, r.q, r.len) : 16;
    if (dimer) result = 28;
    x_add(E, E.c.filter + result, 1);   // :278
    if (!dedup_out && alive && result == 0) x_stat_read(E, lds, 1, r.s, r.q, r.len);   // :280-290
    if (!alive) x.flags |= RS_NULL;
    if (!E.ghost) x_write_read(a, 0, gp, r, result, x);
}

// ---- paired-end loop body: peprocessor.cpp:383-643 ----
FQ_DEV void x_process_pe(const ExactArgs& E, u32* lds, int gp, const XBufs& b) {
    const KernelArgs& a = E.k;
    const DevParams& p = a.p;
    const bool thread0 = (a.batch_flags & 1u) != 0;
    XRead r1 = {b.s1, b.q1, x_load(E, gp, 0, b.s1, b.q1), 0};
    XRead r2 = {b.s2, b.q2, x_load(E, gp, 1, b.s2, b.q2), 0};
    XRes x1 = {0, 0, 0, 0}, x2 = {0, 0, 0, 0};
    if (p.dup_enabled && a.dup_pos && !E.ghost) {   // :397-402, checkPair duplicate.cpp:136-148
        u64 h[MAX_DUP_BUFS] = {0, 0, 0, 0, 0, 0, 0, 0};
        x_hash(a, r1.s, r1.len, 0, h);
        x_hash(a, r2.s, r2.len, r1.len, h);
        for (int i = 0; i < p.dup_bufnum; i++) a.dup_pos[(size_t)gp * p.dup_bufnum + i] = h[i];
    }
    if (E.hash_only) return;
    x_stat_read(E, lds, 0, r1.s, r1.q, r1.len);   // :393
    x_stat_read(E, lds, 2, r2.s, r2.q, r2.len);   // :394
    bool dedup_out = false;
    if (a.dupflag && a.dupflag[gp]) {
        x1.flags |= RS_DUP;
        x2.flags |= RS_DUP;
        dedup_out = p.dedup != 0;
    }
    if (p.umi_len1 > 0) x_trim_front(r1, imin(r1.len, p.umi_len1) + p.umi_skip);   // :419-420
    if (p.umi_len2 > 0) x_trim_front(r2, imin(r2.len, p.umi_len2) + p.umi_skip);
    int ft1 = 0, ft2 = 0;
    const bool a1 = x_apply_trim_and_cut(p, r1, p.trim_front1, p.trim_tail1, ft1);   // :425
    const bool a2 = x_apply_trim_and_cut(p, r2, p.trim_front2, p.trim_tail2, ft2);   // :426
    const bool both = a1 && a2;
    if (both && p.poly_g) {   // :428-431
        r1.len = x_trim_poly_g(r1.s, r1.len, p.poly_g_min);
        r2.len = x_trim_poly_g(r2.s, r2.len, p.poly_g_min);
    }
    bool isize_done = false, dimer = false;
    XOverlap ov = {0, 0, 0, 0, 0};
    bool ov_computed = false;
    const u16* lut = a.lut.ov_limit;
    if (both && (p.adapter_enabled || p.correction || thread0 || p.merge)) {   // :438-441
        ov = x_analyze(r1.s, r1.len, r2.s, r2.len, b.rc, lut, p.overlap_require, false, b.L, b.R);
        ov_computed = true;
    }
    if (both && (p.adapter_enabled || p.correction)) {   // :443-485
        const XOverlap ova = p.allow_gap ? x_analyze(r1.s, r1.len, r2.s, r2.len, b.rc, lut, p.overlap_require, true, b.L, b.R) : ov;
        if (thread0) { x_stat_isize(E, r1.len, r2.len, ov, ft1, ft2); isize_done = true; }
        if (p.correction && !ova.has_gap) x_correct(E, r1, r2, ova, gp, x1, x2);
        if (p.adapter_enabled) {
            bool trimmed = false;
            if (ova.overlapped && ova.offset < 0) {   // trimByOverlapAnalysis adaptertrimmer.cpp:17-46
                const int ol = ova.overlap_len;
                const int len1 = imin(r1.len, ol + ft2);
                const int len2 = imin(r2.len, ol + ft1);
                x1.apos = (u32)len1; x1.alen = (u32)(r1.len - len1);
                x2.apos = (u32)len2; x2.alen = (u32)(r2.len - len2);
                x_add(E, E.c.adapter_bases, (r1.len - len1) + (r2.len - len2));   // filterresult.cpp:154-155
                x_resize(r1, len1);
                x_resize(r2, len2);
                trimmed = true;
                x1.flags |= RS_ADAPTER_OV;
                x2.flags |= RS_ADAPTER_OV;
            }
            bool t1 = trimmed, t2 = trimmed;
            if (!trimmed) {   // :460-466
                int pos, al;
                if (p.has_a1) {
                    t1 = x_apply_trim_by_sequence(E, r1, b.ad1, p.alen1, 4, pos, al, b.L, b.R);
                    if (t1) { x1.apos = (u32)pos; x1.alen = (u32)al; }
                }
                if (p.has_a2) {
                    t2 = x_apply_trim_by_sequence(E, r2, b.ad2, p.alen2, 4, pos, al, b.L, b.R);
                    if (t2) { x2.apos = (u32)pos; x2.alen = (u32)al; }
                }
            }
            if (p.n_fasta) {   // :467-470
                t1 |= x_trim_by_multi(E, r1, 2u * (u32)(a.first + gp), b.adf, b.L, b.R);
                t2 |= x_trim_by_multi(E, r2, 2u * (u32)(a.first + gp) + 1u, b.adf, b.L, b.R);
            }
            if (t1) { x_add(E, E.c.adapter_reads, 1); x1.flags |= RS_ADAPTER; }   // :472-475
            if (t2) { x_add(E, E.c.adapter_reads, 1); x2.flags |= RS_ADAPTER; }
            if ((t1 || t2) && r1.len <= p.dimer_max_len && r2.len <= p.dimer_max_len) dimer = true;   // :480-484
        }
    }
    if (p.overlapped_out && both) {   // :488-495
        const XOverlap ovx = x_analyze(r1.s, r1.len, r2.s, r2.len, b.rc, nullptr, p.overlap_require, false, b.L, b.R);
        if (ovx.overlapped) {
            const int pos = imax(0, ovx.offset) + ovx.overlap_len;   // string(substr(start), overlap_len): the (str, pos) constructor
            x1.rsv = 0x8000u | (u32)pos;
            x2.rsv = (u32)(r1.len - pos);
        }
    }
    if (thread0 && !isize_done && both) {   // :497-504
        if (!ov_computed) { ov = x_analyze(r1.s, r1.len, r2.s, r2.len, b.rc, lut, p.overlap_require, false, b.L, b.R); ov_computed = true; }
        x_stat_isize(E, r1.len, r2.len, ov, ft1, ft2);
        isize_done = true;
    }
    if (both && p.poly_x) {   // :506-509
        x_poly_x(E, r1, x1);
        x_poly_x(E, r2, x2);
    }
    if (both) {   // :511-516
        if (p.max_len1 > 0 && p.max_len1 < r1.len) x_resize(r1, p.max_len1);
        if (p.max_len2 > 0 && p.max_len2 < r2.len) x_resize(r2, p.max_len2);
    }
    bool merge_done = false;
    int code1 = 0, code2 = 0;
    if (p.merge && both) {   // :518-561
        ov = x_analyze(r1.s, r1.len, r2.s, r2.len, b.rc, lut, p.overlap_require, false, b.L, b.R);
        if (ov.overlapped) {   // OverlapAnalysis::merge overlapanalysis.cpp:148-179
            const int ol = ov.overlap_len;
            const int len1 = ol + imax(0, ov.offset);
            const int len2 = ov.offset > 0 ? r2.len - ol : 0;
            const int m1 = imin(len1, r1.len);
            const int m2 = ov.offset > 0 ? imax(0, imin(len2, r2.len - ol)) : 0;
            const int mlen = m1 + m2;
            for (int k = 0; k < m1; k++) { b.ms[k] = r1.s[k]; b.mq[k] = r1.q[k]; }
            for (int k = 0; k < m2; k++) {   // rc(r2)[ol + k] = comp(r2[len2 - 1 - ol - k])
                const int src = r2.len - 1 - ol - k;
                b.ms[m1 + k] = x_complement(r2.s[src]);
                b.mq[m1 + k] = r2.q[src];
            }
            b.ms[mlen] = 0;
            const int result = x_pass_filter(a, b.ms, b.mq, mlen);
            x_add(E, E.c.filter + result, 2);
            if (result == 0) {
                x_stat_read(E, lds, 1, b.ms, b.mq, mlen);
                x_add(E, E.c.merged, 1);   // :688-690
                x1.flags |= RS_MERGED;
                x2.flags |= RS_MERGED;
            }
            code1 = code2 = result;
            if (!p.overlapped_out) { x1.rsv = (u32)m1; x2.rsv = (u32)m2; }
            merge_done = true;
        } else if (p.merge_include_unmerged) {
            code1 = x_pass_filter(a, r1.s, r1.q, r1.len);
            code2 = x_pass_filter(a, r2.s, r2.q, r2.len);
            if (dimer) code1 = code2 = 28;
            x_add(E, E.c.filter + code1, 1);
            if (code1 == 0 && !dedup_out) x_stat_read(E, lds, 1, r1.s, r1.q, r1.len);
            x_add(E, E.c.filter + code2, 1);
            if (code2 == 0 && !dedup_out) x_stat_read(E, lds, 1, r2.s, r2.q, r2.len);
            merge_done = true;
        }
    }
    if (!merge_done) {   // :563-621
        code1 = a1 ? x_pass_filter(a, r1.s, r1.q, r1.len) : 16;
        code2 = a2 ? x_pass_filter(a, r2.s, r2.q, r2.len) : 16;
        if (dimer) code1 = code2 = 28;
        x_add(E, E.c.filter + imax(code1, code2), 2);
        if (!dedup_out && a1 && code1 == 0 && a2 && code2 == 0 && !p.merge) {   // :588-591
            x_stat_read(E, lds, 1, r1.s, r1.q, r1.len);
            x_stat_read(E, lds, 3, r2.s, r2.q, r2.len);
        }
    }
    if (!a1) x1.flags |= RS_NULL;
    if (!a2) x2.flags |= RS_NULL;
    if (E.ghost) return;
    x_write_read(a, 0, gp, r1, code1, x1);
    x_write_read(a, 1, gp, r2, code2, x2);
    a.pair[2 * (size_t)gp] = ((u32)ov.offset & 0xFFFFu) | (((u32)ov.overlap_len & 0xFFFFu) << 16);
    a.pair[2 * (size_t)gp + 1] = ((u32)ov.diff & 0xFFFFu) | ((u32)((ov.overlapped ? 1 : 0) | (ov.has_gap ? 2 : 0) | (isize_done ? 4 : 0)) << 16);
}

FQ_DEV void exact_body(const ExactArgs& E, u32* lds) {
    const int lds_total = E.lds_slots * E.lds_slot_dwords;
    if (lds_total) {
        for (int i = thread_id(); i < lds_total; i += block_threads()) lds[i] = 0u;
        block_sync();
    }
    const int lane = block_id() * block_threads() + thread_id();
    const int lanes = grid_blocks() * block_threads();
    u8* base = E.scratch + (size_t)lane * E.lane_bytes;
    const int ML = E.ML;
    XBufs b;
    b.s1 = base;
    b.q1 = base + ML;
    b.s2 = base + 2 * ML;
    b.q2 = base + 3 * ML;
    b.rc = base + 4 * ML;
    b.ms = base + 5 * ML;    // 2 ML
    b.mq = base + 7 * ML;    // 2 ML
    b.L = (short*)(base + 9 * ML);    // 2 ML
    b.R = (short*)(base + 11 * ML);   // 2 ML (+ one spare buffer)
    b.ad1 = base + (size_t)EXACT_BUFS * ML;
    b.ad2 = b.ad1 + 264;
    b.adf = b.ad2 + 264;
    if (!E.hash_only) {
        if (E.k.p.has_a1) x_decode_adapter(E.k.p.a1w, E.k.p.alen1, b.ad1);
        if (E.k.p.has_a2) x_decode_adapter(E.k.p.a2w, E.k.p.alen2, b.ad2);
    }
    for (int i = lane; i < E.x_count; i += lanes) {
        const int gp = E.x_all ? i : E.x_unit[E.x_k0 + i] - E.k.first;
        if (!E.hash_only) {   // what the plan's kernels counted for the empty unit they saw in this place
            ExactArgs G = E;
            G.sign = -1;
            G.ghost = 1;
            if (G.k.p.paired) x_process_pe(G, lds, gp, b);
            else x_process_se(G, lds, gp, b);
        }
        if (E.k.p.paired) x_process_pe(E, lds, gp, b);
        else x_process_se(E, lds, gp, b);
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
