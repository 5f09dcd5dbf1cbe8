// fq_eval.h - the Evaluator pre-pass on the device: the two counting loops of
// /root/reference/src/evaluator.cpp that run over a prefix of the input before any worker starts -
//   * the 4^10 ten-mer histogram of evalAdapterAndReadNum (evaluator.cpp:377-399), and
//   * the substring census of computeOverRepSeq (evaluator.cpp:78-142): every substring of 5 lengths of the
//     first 1.51 Mbases, counted exactly (the reference keeps a std::map<string,long>).
// Both read the packed rows a batch already holds in HBM (2-bit bases + the N flag in the quality byte).
#pragma once
#include "fq_intrin.h"
#include "fq_types.h"

namespace fq {

struct EvalReads {
    const u8* seq;
    const u8* qual;
    const u16* len;
    int n;               // reads the limits of the reference loop admit (computed by the host from `len`)
    int seq_stride, qual_stride;
};

// ---- ten-mer histogram --------------------------------------------------------------------------------------
// Evaluator::seq2int (evaluator.cpp:573-625) is a pure function of the window: the rolling update and the
// from-scratch computation give the same key, and a window holding a non-ACGT base gives -1 either way.
// key = first base in the most significant pair; rows hold base j in bits [2(j%4), 2(j%4)+1] of byte j/4.
struct EvalKmerArgs {
    EvalReads r;
    int shift_tail;      // max(1, trim_tail1)
    int span;            // positions per read one row of lanes covers (max_len rounded up)
    u32* counts;         // [1 << 20]
};

FQ_DEV u32 eval_reverse_pairs20(u32 v) {   // 10 two-bit groups, first group to the top
    u32 x = brev32(v);                     // bit i -> 31 - i : groups reversed, bits inside a group swapped
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return x >> 12;
}

FQ_DEV void eval_kmer_body(const EvalKmerArgs& a) {
    const long long t = (long long)block_id() * block_threads() + thread_id();
    const int rd = (int)(t / a.span), pos = 20 + (int)(t % a.span);
    if (rd >= a.r.n) return;
    const int rlen = a.r.len[rd];
    if (pos > rlen - 10 - a.shift_tail) return;
    const u8* s = a.r.seq + (size_t)rd * a.r.seq_stride;
    const u8* q = a.r.qual + (size_t)rd * a.r.qual_stride;
    u32 nflag = 0;
    for (int k = 0; k < 10; k++) nflag |= q[pos + k];
    if (nflag & 0x80u) return;   // an N in the window
    // 20 bits starting at base `pos`: bytes pos/4 .. (pos + 9)/4 hold them (3 or 4 bytes)
    const int b = pos >> 2, nb = ((pos + 9) >> 2) - b;
    u32 w = 0;
    for (int k = 0; k <= nb; k++) w |= (u32)s[b + k] << (8 * k);
    const u32 key = eval_reverse_pairs20((w >> (2 * (pos & 3))) & 0xFFFFFu);
    if (key != 0u) g_atomic_add_u32(&a.counts[key], 1u);          // "set AAAAAAAAAA = 0" (evaluator.cpp:401-402)
}

// ---- substring census ---------------------------------------------------------------------------------------
// An exact multiset count without strings: an open-addressing table whose slot names ONE occurrence of its
// substring (the representative); a later occurrence that hashes there is compared base by base with the
// representative's text and either counts or moves on to the next slot.
//   slot word: (read + 1) << 40 | pos << 24 | step index << 21 | 21 hash bits        0 = empty
enum { EVAL_STEPS = 5 };
struct EvalCensusArgs {
    EvalReads r;
    int span;
    int step[EVAL_STEPS];      // 10, 20, 40, 100, min(150, seqlen - 2); <= 0: skipped
    u64* slot;                 // [cap]
    u32* count;                // [cap]
    u32 mask;                  // cap - 1
    // harvest
    int seqlen;
    u64* hot;                  // [hot_cap] slot words of the substrings over their threshold
    u32* hot_count;            // [hot_cap]
    u32* n_hot;                // [1]
    u32 hot_cap;
    // text
    u8* text;                  // [hot_cap * 152] a row of characters per hot substring
};

FQ_DEV u32 eval_base(const u8* s, const u8* q, int j) {   // 0..3 = A T C G, 4 = N
    return (q[j] & 0x80u) ? 4u : (u32)((s[j >> 2] >> (2 * (j & 3))) & 3u);
}

FQ_DEV void eval_census_body(const EvalCensusArgs& a) {
    const long long t = (long long)block_id() * block_threads() + thread_id();
    const int rd = (int)(t / a.span), pos = (int)(t % a.span);
    if (rd >= a.r.n) return;
    const int rlen = a.r.len[rd];
    const u8* s = a.r.seq + (size_t)rd * a.r.seq_stride;
    const u8* q = a.r.qual + (size_t)rd * a.r.qual_stride;
    u64 h = 0x9E3779B97F4A7C15ull;
    int done = 0;
    // the steps grow (the last one may be smaller than the others: hash restarted then)
    for (int si = 0; si < EVAL_STEPS; si++) {
        const int step = a.step[si];
        if (step <= 0 || pos >= rlen - step) continue;   // for(i = 0; i < rlen - step; i++)
        if (step < done) { h = 0x9E3779B97F4A7C15ull; done = 0; }
        for (; done < step; done++) h = (h ^ (u64)(eval_base(s, q, pos + done) + 1u)) * 0x100000001B3ull;
        u64 hs = (h ^ (u64)step) * 0xD6E8FEB86659FD93ull;
        hs ^= hs >> 32;
        const u64 mine = ((u64)(rd + 1) << 40) | ((u64)pos << 24) | ((u64)si << 21) | (hs & 0x1FFFFFull);
        u32 at = (u32)(hs >> 21) & a.mask;
        for (;;) {
            u64 cur = __hip_atomic_load(&a.slot[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                cur = g_atomic_cas_u64(&a.slot[at], 0ull, mine);
                if (cur == 0ull) {
                    g_atomic_add_u32(&a.count[at], 1u);
                    break;
                }
            }
            if (((cur ^ mine) & 0x1FFFFFull) == 0ull) {
                // same hash bits: same length and same text?
                const int rd2 = (int)(cur >> 40) - 1, pos2 = (int)((cur >> 24) & 0xFFFFu);
                const int step2 = a.step[(cur >> 21) & 7u];
                bool same = step2 == step;
                if (same && !(rd2 == rd && pos2 == pos)) {
                    const u8* s2 = a.r.seq + (size_t)rd2 * a.r.seq_stride;
                    const u8* q2 = a.r.qual + (size_t)rd2 * a.r.qual_stride;
                    for (int k = 0; k < step; k++)
                        if (eval_base(s, q, pos + k) != eval_base(s2, q2, pos2 + k)) {
                            same = false;
                            break;
                        }
                }
                if (same) {
                    g_atomic_add_u32(&a.count[at], 1u);
                    break;
                }
            }
            at = (at + 1u) & a.mask;
        }
    }
}

// thresholds of evaluator.cpp:115-137
FQ_DEV bool eval_is_hot(int len, u32 count, int seqlen) {
    if (len >= seqlen - 1) return count >= 3u;
    if (len >= 100) return count >= 5u;
    if (len >= 40) return count >= 20u;
    if (len >= 20) return count >= 100u;
    if (len >= 10) return count >= 500u;
    return false;
}

FQ_DEV void eval_harvest_body(const EvalCensusArgs& a) {
    const u32 at = (u32)block_id() * (u32)block_threads() + (u32)thread_id();
    if (at > a.mask) return;
    const u64 w = a.slot[at];
    if (w == 0ull) return;
    const u32 c = a.count[at];
    if (!eval_is_hot(a.step[(w >> 21) & 7u], c, a.seqlen)) return;
    const u32 k = g_atomic_add_u32(a.n_hot, 1u);
    if (k < a.hot_cap) {
        a.hot[k] = w;
        a.hot_count[k] = c;
    }
}

FQ_DEV void eval_text_body(const EvalCensusArgs& a) {
    // 16 lanes per hot substring
    const u32 t = (u32)block_id() * (u32)block_threads() + (u32)thread_id();
    const u32 k = t >> 4, gl = t & 15u;
    const u32 n = *a.n_hot < a.hot_cap ? *a.n_hot : a.hot_cap;
    if (k >= n) return;
    const u64 w = a.hot[k];
    const int rd = (int)(w >> 40) - 1, pos = (int)((w >> 24) & 0xFFFFu), step = a.step[(w >> 21) & 7u];
    const u8* s = a.r.seq + (size_t)rd * a.r.seq_stride;
    const u8* q = a.r.qual + (size_t)rd * a.r.qual_stride;
    u8* o = a.text + (size_t)k * 152;
    for (int j = (int)gl; j < step; j += 16) o[j] = (u8)"ATCGN"[eval_base(s, q, pos + j)];
}

}  // namespace fq
