/* fastp_oracle.c - CPU restatement of fastp's per-read worker loop (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see fastp_oracle.h).  Every function names the
 * reference lines it follows (paths relative to /root/reference).  The
 * behaviour of the reference - including its quirks (SURVEY.md section 8a
 * "quirk ledger") - is the specification; nothing here is "improved".
 */
#include "fastp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------------ */
/* layout of the counter block (shared declaration, own implementation)     */
/* ------------------------------------------------------------------------ */
int fastp_oracle_cycles_for(const fastp_gpu_params* p) { return p->merge ? 2 * p->max_len : p->max_len; }

void fastp_oracle_counter_layout(int cycles, int insert_size_max, fastp_gpu_counter_layout* L) {
    int64_t o = 0;
    const int max_len = cycles;
    memset(L, 0, sizeof(*L));
    L->cycles = cycles;
    o += 4; /* header: [0]=abi version [1]=cycles [2]=insert_size_max [3]=reserved */
    L->filter_stats = o;    o += FASTP_FILTER_RESULT_TYPES;
    L->adapter_reads = o;   o += 1;
    L->adapter_bases = o;   o += 1;
    L->polyx_reads = o;     o += 4;
    L->polyx_bases = o;     o += 4;
    L->correction = o;      o += 64;
    L->corrected_reads = o; o += 1;
    L->merged_pairs = o;    o += 1;
    L->dup_total = o;       o += 1;
    L->dup_count = o;       o += 1;
    L->isize = o;           o += (int64_t)insert_size_max + 1;
    L->st_reads = 0;
    L->st_length_sum = 1;
    L->st_qual_hist = 2;
    L->st_kmer = 2 + 128;
    L->st_cycle = 2 + 128 + 1024;
    L->st_size = L->st_cycle + 34 * (int64_t)max_len;
    for (int s = 0; s < 4; s++) { L->stats[s] = o; o += L->st_size; }
    for (int s = 0; s < 4; s++) { L->overrep_count[s] = o; L->overrep_dist[s] = o; }
    L->total = o;
}

void fastp_oracle_counter_layout_params(const fastp_gpu_params* p, fastp_gpu_counter_layout* L) {
    fastp_oracle_counter_layout(fastp_oracle_cycles_for(p), p->insert_size_max, L);
    if (!p->overrep_enabled) return;
    int64_t o = L->total;
    for (int s = 0; s < 4; s++) { /* Stats::initOverRepSeq stats.cpp:956-971: read-2 Stats use overRepSeqs2 */
        const int r2 = s >= 2;
        L->n_overrep[s] = r2 ? p->n_overrep_seqs2 : p->n_overrep_seqs1;
        L->eval_len[s] = r2 ? p->eval_seq_len2 : p->eval_seq_len1;
        if (!p->paired && r2) { L->n_overrep[s] = 0; L->eval_len[s] = 0; }
        L->overrep_count[s] = o; o += L->n_overrep[s];
        L->overrep_dist[s] = o;  o += L->n_overrep[s] * L->eval_len[s];
    }
    L->total = o;
}

/* ------------------------------------------------------------------------ */
/* the five byte kernels: scalar semantics of src/simd.h:12-31               */
/* (scalar references: simd.cpp:281-324)                                     */
/* ------------------------------------------------------------------------ */
static char orc_complement(char b) { /* util.h:16-33 */
    switch (b) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}

static void orc_revcomp(const char* src, char* dst, int len) { /* simd.cpp:296-309 */
    for (int i = 0; i < len; i++) dst[len - 1 - i] = orc_complement(src[i]);
}

static int orc_mismatches(const char* a, const char* b, int len) { /* simd.cpp:319-324 */
    int d = 0;
    for (int i = 0; i < len; i++) d += (a[i] != b[i]);
    return d;
}

/* simd.cpp:209-233: exact count when <= limit, otherwise "some value > limit" */
static int orc_mismatches_bounded(const char* a, const char* b, int len, int limit) {
    int d = 0;
    for (int i = 0; i < len; i++) {
        d += (a[i] != b[i]);
        if (d > limit) return d;
    }
    return d;
}

/* ------------------------------------------------------------------------ */
/* Matcher (matcher.cpp:10-100)                                              */
/* ------------------------------------------------------------------------ */
static void orc_matcher_tables(const char* ins, const char* nor, int cmplen, int limit,
                               int* L, int* R) {
    /* matcher.cpp:12-36 / 58-82; arrays zero-filled first so that the entries
     * the reference leaves uninitialised (and never reads) are defined */
    memset(L, 0, sizeof(int) * (size_t)cmplen);
    memset(R, 0, sizeof(int) * (size_t)cmplen);
    L[0] = ins[0] == nor[0] ? 0 : 1;
    R[cmplen - 1] = ins[cmplen] == nor[cmplen - 1] ? 0 : 1;
    for (int i = 1; i < cmplen; i++) {
        L[i] = L[i - 1] + (ins[i] != nor[i] ? 1 : 0);
        if (L[i] + R[cmplen - 1] > limit) break;
    }
    for (int i = cmplen - 2; i >= 0; i--) {
        R[i] = R[i + 1] + (ins[i + 1] != nor[i] ? 1 : 0);
        if (R[i] + L[0] > limit) {
            for (int p = 0; p < i; p++) R[p] = limit + 1;
            break;
        }
    }
}

int fastp_oracle_match_one_insertion(const char* ins, const char* nor, int cmplen, int limit) {
    if (cmplen <= 0) return 0;
    int* L = (int*)malloc(sizeof(int) * (size_t)cmplen * 2);
    int* R = L + cmplen;
    orc_matcher_tables(ins, nor, cmplen, limit, L, R);
    int ret = 0;
    for (int i = 1; i < cmplen; i++) { /* matcher.cpp:43-50 */
        if (L[i - 1] + R[cmplen - 1] > limit) { ret = 0; break; }
        if (L[i - 1] + R[i] <= limit) { ret = 1; break; }
    }
    free(L);
    return ret;
}

static int orc_diff_one_insertion(const char* ins, const char* nor, int cmplen, int limit) {
    if (cmplen <= 0) return 100000000; /* matcher.cpp:88 initial minDiff, loop never runs */
    int* L = (int*)malloc(sizeof(int) * (size_t)cmplen * 2);
    int* R = L + cmplen;
    orc_matcher_tables(ins, nor, cmplen, limit, L, R);
    int minDiff = 100000000;
    for (int i = 1; i < cmplen; i++) { /* matcher.cpp:90-97 */
        if (L[i - 1] + R[cmplen - 1] > limit) { minDiff = -1; break; }
        int d = L[i - 1] + R[i];
        if (d <= minDiff) minDiff = d;
    }
    free(L);
    return minDiff;
}

/* ------------------------------------------------------------------------ */
/* OverlapAnalysis::analyze (overlapanalysis.cpp:17-146)                     */
/* ------------------------------------------------------------------------ */
static int orc_accept_nogap(const char* a, const char* b, int len, int limit, int* diff) {
    /* overlapanalysis.cpp:34-44; complete_compare_require = 50 (:28) */
    const int prefix = ORC_MIN(len, 50);
    *diff = orc_mismatches_bounded(a, b, prefix, limit);
    if (*diff > limit) return 0;
    if (len > 50) *diff = orc_mismatches(a, b, len);
    return 1;
}

fastp_oracle_overlap fastp_oracle_analyze(const char* r1, int len1, const char* r2, int len2,
                                          int diffLimit, int overlapRequire,
                                          double diffPercentLimit, int allowGap) {
    fastp_oracle_overlap ov = {0, 0, 0, 0, 0};
    char* rc = (char*)malloc((size_t)len2 + 1);
    orc_revcomp(r2, rc, len2); /* :19-22 */
    rc[len2] = 0;
    const char* str1 = r1;
    const char* str2 = rc;
    int overlap_len = 0, offset = 0, diff = 0;

    /* forward, no gap (:48-64) */
    while (offset < len1 - overlapRequire) {
        overlap_len = ORC_MIN(len1 - offset, len2);
        int limit = ORC_MIN(diffLimit, (int)(overlap_len * diffPercentLimit));
        if (orc_accept_nogap(str1 + offset, str2, overlap_len, limit, &diff)) {
            ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = diff;
            ov.has_gap = 0;
            free(rc);
            return ov;
        }
        offset += 1;
    }
    /* reverse, no gap (:72-89) */
    offset = 0;
    while (offset > -(len2 - overlapRequire)) {
        overlap_len = ORC_MIN(len1, len2 - abs(offset));
        int limit = ORC_MIN(diffLimit, (int)(overlap_len * diffPercentLimit));
        if (orc_accept_nogap(str1, str2 + (-offset), overlap_len, limit, &diff)) {
            ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = diff;
            ov.has_gap = 0;
            free(rc);
            return ov;
        }
        offset -= 1;
    }
    if (allowGap) { /* :91-139 */
        offset = 0;
        while (offset < len1 - overlapRequire) {
            overlap_len = ORC_MIN(len1 - offset, len2);
            int limit = ORC_MIN(diffLimit, (int)(overlap_len * diffPercentLimit));
            int d = orc_diff_one_insertion(str1 + offset, str2, overlap_len - 1, limit);
            if (d < 0 || d > limit) d = orc_diff_one_insertion(str2, str1 + offset, overlap_len - 1, limit);
            if (d <= limit && d >= 0) {
                ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = d;
                ov.has_gap = 1;
                free(rc);
                return ov;
            }
            offset += 1;
        }
        offset = 0;
        while (offset > -(len2 - overlapRequire)) {
            overlap_len = ORC_MIN(len1, len2 - abs(offset));
            int limit = ORC_MIN(diffLimit, (int)(overlap_len * diffPercentLimit));
            int d = orc_diff_one_insertion(str1, str2 - offset, overlap_len - 1, limit);
            if (d < 0 || d > limit) d = orc_diff_one_insertion(str2 - offset, str1, overlap_len - 1, limit);
            if (d <= limit && d >= 0) {
                ov.overlapped = 1; ov.offset = offset; ov.overlap_len = overlap_len; ov.diff = d;
                ov.has_gap = 1;
                free(rc);
                return ov;
            }
            offset -= 1;
        }
    }
    free(rc);
    return ov; /* :141-145 */
}

/* ------------------------------------------------------------------------ */
/* Filter::trimAndCut (filter.cpp:68-207)                                    */
/* ------------------------------------------------------------------------ */
int fastp_oracle_trim_and_cut(const fastp_gpu_params* p, const char* seq, const char* qualstr,
                              int len, int front, int tail, int* out_front, int* out_len) {
    const int enF = p->cut_front, enT = p->cut_tail, enR = p->cut_right;
    *out_front = 0;
    *out_len = len;
    if (front == 0 && tail == 0 && !enF && !enT && !enR) return 1; /* :71-72 */
    int rlen = len - front - tail;
    if (rlen < 0) return 0; /* :76-77 */
    if (!enF && !enT && !enR) { /* :79-89 */
        *out_front = front;
        *out_len = rlen;
        return 1;
    }
    const int l = len;
    if (enF) { /* :97-127 */
        int w = p->cut_front_window;
        int s = front;
        if (l - front - tail - w <= 0) return 0;
        int totalQual = 0;
        for (int i = 0; i < w - 1; i++) totalQual += qualstr[s + i];
        for (s = front; s + w < l - tail; s++) {
            totalQual += qualstr[s + w - 1];
            if (s > front) totalQual -= qualstr[s - 1];
            if (totalQual >= w * (33 + p->cut_front_quality)) break;
        }
        if (s > 0) s = s + w - 1;
        while (s < l && seq[s] == 'N') s++;
        front = s;
        rlen = l - front - tail;
    }
    if (enR) { /* :130-163 */
        int w = p->cut_right_window;
        int s = front;
        if (l - front - tail - w <= 0) return 0;
        int totalQual = 0;
        for (int i = 0; i < w - 1; i++) totalQual += qualstr[s + i];
        int found = 0;
        for (s = front; s + w < l - tail; s++) {
            totalQual += qualstr[s + w - 1];
            if (s > front) totalQual -= qualstr[s - 1];
            if (totalQual < w * (33 + p->cut_right_quality)) { found = 1; break; }
        }
        if (found) {
            while (s < l - 1 && qualstr[s] >= 33 + p->cut_right_quality) s++;
            rlen = s - front;
        }
    }
    if (!enR && enT) { /* :166-194 */
        int w = p->cut_tail_window;
        if (l - front - tail - w <= 0) return 0;
        int totalQual = 0;
        int t = l - tail - 1;
        for (int i = 0; i < w - 1; i++) totalQual += qualstr[t - i];
        for (t = l - tail - 1; t - w >= front; t--) {
            totalQual += qualstr[t - w + 1];
            if (t < l - tail - 1) totalQual -= qualstr[t + 1];
            if (totalQual >= w * (33 + p->cut_tail_quality)) break;
        }
        if (t < l - 1) t = t - w + 1;
        while (t >= 0 && seq[t] == 'N') t--;
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return 0; /* :196-197 */
    *out_front = front;
    *out_len = rlen;
    return 1;
}

/* ------------------------------------------------------------------------ */
/* PolyX (polyx.cpp:16-116)                                                  */
/* ------------------------------------------------------------------------ */
int fastp_oracle_trim_poly_g(const char* data, int rlen, int compareReq) { /* :16-42 */
    const int allowOneMismatchForEach = 8, maxMismatch = 5;
    int mismatch = 0, i = 0, firstGPos = rlen - 1;
    for (i = 0; i < rlen; i++) {
        if (data[rlen - i - 1] != 'G') mismatch++;
        else firstGPos = rlen - i - 1;
        int allowed = (i + 1) / allowOneMismatchForEach;
        if (mismatch > maxMismatch || (mismatch > allowed && i >= compareReq - 1)) break;
    }
    if (i >= compareReq) {
        /* Read::resize(firstGPos) (read.cpp:62-67) ignores len<0 / len>length */
        if (firstGPos >= 0 && firstGPos <= rlen) return firstGPos;
    }
    return rlen;
}

static int orc_polyx_idx(char c) { /* polyx.cpp:59-68: A0 T1 C2 G3 N4 else 5 */
    switch (c) {
        case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3;
        case 'N': return 4; default: return 5;
    }
}

int fastp_oracle_trim_poly_x(const char* data, int rlen, int compareReq, int* poly_base,
                             int* trimmed) { /* :49-116 */
    const int allowOneMismatchForEach = 8, maxMismatch = 5;
    static const char ATCG[4] = {'A', 'T', 'C', 'G'}; /* common.h:25 */
    int cnt[4] = {0, 0, 0, 0};
    int pos = 0;
    *poly_base = -1;
    *trimmed = 0;
    for (pos = 0; pos < rlen; pos++) {
        int idx = orc_polyx_idx(data[rlen - pos - 1]);
        if (idx < 4) cnt[idx]++;
        else if (idx == 4) { cnt[0]++; cnt[1]++; cnt[2]++; cnt[3]++; }
        int cmp = pos + 1;
        int allowed = ORC_MIN(maxMismatch, cmp / allowOneMismatchForEach);
        int needToBreak = 1;
        for (int b = 0; b < 4; b++)
            if (cmp - cnt[b] <= allowed) needToBreak = 0;
        if (needToBreak && (pos >= allowOneMismatchForEach || pos + 1 >= compareReq - 1)) break;
    }
    if (pos + 1 >= compareReq) { /* :98-115 */
        int poly = 0, maxCount = -1;
        for (int b = 0; b < 4; b++)
            if (cnt[b] > maxCount) { maxCount = cnt[b]; poly = b; }
        char polyBase = ATCG[poly];
        /* :109  while(data[rlen-pos-1] != polyBase && pos>=0) pos--;
         * index -1 (pos == rlen, the scan never broke) is out of bounds in the
         * reference; it is treated as "not the poly base" (quirk ledger #4);
         * index rlen is the string's terminating 0. */
        for (;;) {
            int idx = rlen - pos - 1;
            char c = (idx < 0) ? 0 : (idx >= rlen ? 0 : data[idx]);
            if (!(c != polyBase && pos >= 0)) break;
            pos--;
        }
        int newlen = rlen - pos - 1;
        *poly_base = poly;
        *trimmed = pos + 1; /* addPolyXTrimmed(poly, pos+1) */
        if (newlen < 0 || newlen > rlen) return rlen; /* Read::resize ignores */
        return newlen;
    }
    return rlen;
}

/* ------------------------------------------------------------------------ */
/* AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157)                */
/* ------------------------------------------------------------------------ */
int fastp_oracle_trim_by_sequence(const char* rdata, int rlen, const char* adata, int alen,
                                  int matchReq, int* out_pos) {
    const int allowOneMismatchForEach = 8;
    if (alen < matchReq) return 0;
    int pos = 0, found = 0, start = 0;
    if (alen >= 16) start = -4;
    else if (alen >= 12) start = -3;
    else if (alen >= 8) start = -2;
    for (pos = start; pos < rlen - matchReq; pos++) { /* :87-100 */
        int cmplen = ORC_MIN(rlen - pos, alen);
        int allowed = cmplen / allowOneMismatchForEach;
        int startOffset = ORC_MAX(0, -pos);
        int mm = orc_mismatches_bounded(adata + startOffset, rdata + startOffset + pos,
                                        cmplen - startOffset, allowed);
        if (mm <= allowed) { found = 1; break; }
    }
    if (!found) { /* :105-118 - note rdata/adata WITHOUT +pos (quirk #7) */
        for (pos = 0; pos < rlen - matchReq - 1; pos++) {
            int cmplen = ORC_MIN(rlen - pos - 1, alen);
            int allowed = cmplen / allowOneMismatchForEach - 1;
            if (fastp_oracle_match_one_insertion(rdata, adata, cmplen, allowed)) { found = 1; break; }
        }
    }
    if (!found) { /* :122-135 */
        for (pos = 0; pos < rlen - matchReq; pos++) {
            int cmplen = ORC_MIN(rlen - pos, alen - 1);
            int allowed = cmplen / allowOneMismatchForEach - 1;
            if (fastp_oracle_match_one_insertion(adata, rdata, cmplen, allowed)) { found = 1; break; }
        }
    }
    *out_pos = pos;
    return found;
}

/* ------------------------------------------------------------------------ */
/* Filter::passFilter (filter.cpp:15-66)                                     */
/* ------------------------------------------------------------------------ */
static char orc_num2qual(int num) { /* util.h:260-268 */
    if (num > 127 - 33) num = 127 - 33;
    if (num < 0) num = 0;
    return (char)(num + 33);
}

int fastp_oracle_pass_filter(const fastp_gpu_params* p, const char* seq, const char* qual, int rlen) {
    if (seq == NULL || rlen == 0) return FASTP_FAIL_LENGTH; /* :16-18 */
    int lowQualNum = 0, nBaseNum = 0, totalQual = 0;
    if (p->qual_filter || p->length_filter) { /* :26-33, simd.cpp:281-294 */
        const unsigned char thr = (unsigned char)orc_num2qual(p->qualified_qual);
        for (int i = 0; i < rlen; i++) {
            unsigned char q = (unsigned char)qual[i];
            totalQual += q - 33;
            if (q < thr) lowQualNum++;
            if (seq[i] == 'N') nBaseNum++;
        }
    }
    if (p->qual_filter) { /* :35-42 */
        if (lowQualNum > (p->unqualified_percent_limit * rlen / 100.0)) return FASTP_FAIL_QUALITY;
        else if (p->avg_qual_req > 0 && (totalQual / rlen) < p->avg_qual_req) return FASTP_FAIL_QUALITY;
        else if (nBaseNum > p->n_base_limit) return FASTP_FAIL_N_BASE;
    }
    if (p->length_filter) { /* :44-49 */
        if (rlen < p->length_required) return FASTP_FAIL_LENGTH;
        if (p->length_limit > 0 && rlen > p->length_limit) return FASTP_FAIL_TOO_LONG;
    }
    if (p->complexity_filter) { /* :51-54, 59-66 */
        if (rlen <= 1) return FASTP_FAIL_COMPLEXITY;
        int diff = 0;
        for (int i = 0; i < rlen - 1; i++) diff += (seq[i] != seq[i + 1]);
        if (!((double)diff / (double)(rlen - 1) >= p->complexity_threshold)) return FASTP_FAIL_COMPLEXITY;
    }
    return FASTP_PASS_FILTER;
}

/* ------------------------------------------------------------------------ */
/* Duplicate (duplicate.cpp:9-163)                                           */
/* ------------------------------------------------------------------------ */
#define ORC_PRIME_ARRAY_LEN 512

typedef struct orc_dup {
    uint64_t bufLenInBytes, bufLenInBits, offsetMask;
    int bufNum;
    unsigned char* buf;
    uint64_t* primes;
    uint64_t total, dups;
} orc_dup;

static void orc_dup_geometry(int level, uint64_t* bytes, int* num) { /* :13-47 */
    uint64_t b = 1ULL << 29;
    int n = 2;
    switch (level) {
        case 2: b *= 2; break;
        case 3: b *= 2; n *= 2; break;
        case 4: b *= 4; n *= 2; break;
        case 5: b *= 8; n *= 2; break;
        case 6: b *= 8; n *= 4; break;
        default: break;
    }
    *bytes = b;
    *num = n;
}

static uint64_t* orc_dup_primes(int bufNum) { /* initPrimeArrays :66-84 */
    uint64_t* arr = (uint64_t*)calloc((size_t)bufNum * ORC_PRIME_ARRAY_LEN, sizeof(uint64_t));
    uint64_t number = 10000, count = 0;
    while (count < (uint64_t)bufNum * ORC_PRIME_ARRAY_LEN) {
        number++;
        int isPrime = 1;
        for (uint64_t i = 2; (double)i <= sqrt((double)number); i++) {
            if (number % i == 0) { isPrime = 0; break; }
        }
        if (isPrime) { arr[count++] = number; number += 10000; }
    }
    return arr;
}

static uint64_t orc_hash_val(char c) { /* SEQ_HASH_VAL :92-109 */
    switch (c) {
        case 'A': return 7; case 'T': return 222; case 'C': return 74; case 'G': return 31;
        default: return 13;
    }
}

static void orc_seq2intvector(const uint64_t* primes, int bufNum, uint64_t mask, const char* data,
                              int len, uint64_t* out, int posOffset) { /* :111-120 */
    for (int p = 0; p < len; p++) {
        uint64_t base = orc_hash_val(data[p]);
        for (int i = 0; i < bufNum; i++) {
            int offset = (p + posOffset) * bufNum + i;
            offset &= (int)mask;
            out[i] += primes[offset] * (base + (uint64_t)(p + posOffset));
        }
    }
}

int fastp_oracle_dup_hash(int level, const char* s1, int l1, const char* s2, int l2, uint64_t* out) {
    uint64_t bytes; int num;
    orc_dup_geometry(level, &bytes, &num);
    uint64_t* primes = orc_dup_primes(num);
    uint64_t mask = (uint64_t)ORC_PRIME_ARRAY_LEN * num - 1;
    for (int i = 0; i < 8; i++) out[i] = 0;
    orc_seq2intvector(primes, num, mask, s1, l1, out, 0);
    if (s2) orc_seq2intvector(primes, num, mask, s2, l2, out, l1);
    free(primes);
    return num;
}

/* the bit positions (hash mod mBufLenInBits) of n pairs / reads: out[n][bufNum].  The stateless half of
 * Duplicate::checkPair / checkRead, callable from several threads - lets a test at BASELINE scale run the
 * per-read part of the oracle on chunks in parallel and still check the (stream-ordered) duplicate decisions. */
int fastp_oracle_dup_bits_batch(int level, int n, int row_stride, const char* seq1, const int32_t* len1,
                                const char* seq2, const int32_t* len2, uint64_t* out) {
    uint64_t bytes; int num;
    orc_dup_geometry(level, &bytes, &num);
    uint64_t* primes = orc_dup_primes(num);
    const uint64_t mask = (uint64_t)ORC_PRIME_ARRAY_LEN * num - 1, bits = bytes << 3;
    for (int g = 0; g < n; g++) {
        uint64_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        orc_seq2intvector(primes, num, mask, seq1 + (size_t)g * row_stride, len1[g], h, 0);
        if (seq2) orc_seq2intvector(primes, num, mask, seq2 + (size_t)g * row_stride, len2[g], h, len1[g]);
        for (int i = 0; i < num; i++) out[(size_t)g * num + i] = h[i] % bits;
    }
    free(primes);
    return num;
}

static orc_dup* orc_dup_create(int level) {
    orc_dup* d = (orc_dup*)calloc(1, sizeof(orc_dup));
    orc_dup_geometry(level, &d->bufLenInBytes, &d->bufNum);
    d->offsetMask = (uint64_t)ORC_PRIME_ARRAY_LEN * d->bufNum - 1;
    d->bufLenInBits = d->bufLenInBytes << 3;
    d->buf = (unsigned char*)calloc(d->bufLenInBytes * d->bufNum, 1); /* lazily committed */
    d->primes = orc_dup_primes(d->bufNum);
    if (!d->buf) { free(d->primes); free(d); return NULL; }
    return d;
}

static int orc_dup_apply(orc_dup* d, const uint64_t* positions) { /* applyBloomFilter :150-163 */
    int isDup = 1;
    for (int i = 0; i < d->bufNum; i++) {
        uint64_t pos = positions[i] % d->bufLenInBits;
        uint64_t bytePos = pos >> 3;
        unsigned char byte = (unsigned char)(1u << (pos & 7));
        unsigned char* cell = d->buf + (uint64_t)i * d->bufLenInBytes + bytePos;
        unsigned char old = *cell;
        *cell = old | byte;
        isDup &= (old & byte) != 0;
    }
    d->total++;
    if (isDup) d->dups++;
    return isDup;
}

/* ------------------------------------------------------------------------ */
/* Stats::statRead (stats.cpp:191-291; overrepresentation part not restated) */
/* ------------------------------------------------------------------------ */
static int orc_base2val(char b) { /* BASE2VAL stats.cpp:294-311 */
    switch (b) {
        case 'A': return 0; case 'T': return 1; case 'C': return 2; case 'G': return 3;
        default: return -1;
    }
}

static void orc_stat_read(const fastp_gpu_counter_layout* L, int64_t* st, const char* seqstr,
                          const char* qualstr, int len) {
    const int64_t C = L->cycles;
    int64_t* cyc = st + L->st_cycle;
    int64_t* q30 = cyc + 0 * 8 * C;   /* setCyclePointers stats.cpp:54-63 */
    int64_t* q20 = cyc + 1 * 8 * C;
    int64_t* cont = cyc + 2 * 8 * C;
    int64_t* qual = cyc + 3 * 8 * C;
    int64_t* totBase = cyc + 32 * C;
    int64_t* totQual = cyc + 33 * C;
    st[L->st_length_sum] += len;
    int kmer = 0;
    int needFullCompute = 1;
    for (int i = 0; i < len; i++) {
        char base = seqstr[i];
        char q = qualstr[i];
        int b = base & 0x07;
        st[L->st_qual_hist + (unsigned char)q]++;
        if (q >= '?') { q30[b * C + i]++; q20[b * C + i]++; }
        else if (q >= '5') { q20[b * C + i]++; }
        cont[b * C + i]++;
        qual[b * C + i] += (q - 33);
        totBase[i]++;
        totQual[i] += (q - 33);
        if (base == 'N') { needFullCompute = 1; continue; }
        if (i < 4) continue;
        if (!needFullCompute) {
            int val = orc_base2val(base);
            if (val < 0) { needFullCompute = 1; continue; }
            kmer = ((kmer << 2) & 0x3FC) | val;
            st[L->st_kmer + kmer]++;
        } else {
            int valid = 1;
            kmer = 0;
            for (int k = 0; k < 5; k++) {
                int val = orc_base2val(seqstr[i - 4 + k]);
                if (val < 0) { valid = 0; break; }
                kmer = ((kmer << 2) & 0x3FC) | val;
            }
            if (!valid) { needFullCompute = 1; continue; }
            st[L->st_kmer + kmer]++;
            needFullCompute = 0;
        }
    }
    st[L->st_reads]++;
}

/* ------------------------------------------------------------------------ */
/* the engine object                                                         */
/* ------------------------------------------------------------------------ */
struct fastp_oracle;
static void orc_stat_read_slot(struct fastp_oracle* o, int slot, const char* seqstr, const char* qualstr, int len);

struct fastp_oracle {
    fastp_gpu_params p;
    char adapter1[FASTP_GPU_MAX_ADAPTER_LEN + 1];
    char adapter2[FASTP_GPU_MAX_ADAPTER_LEN + 1];
    int has_seq_r1, has_seq_r2, alen1, alen2;
    int n_fasta;          /* AdapterOptions::seqsInFasta (options.h:213) */
    char** fasta;
    int* fasta_len;
    int n_ovr[2];         /* Options::overRepSeqs1/2 (options.h:364-365), map order */
    char** ovr[2];
    int* ovr_len[2];
    fastp_gpu_counter_layout L;
    int64_t* ctr;
    orc_dup* dup;
};

fastp_oracle* fastp_oracle_create(const fastp_gpu_params* params) {
    if (!params || params->max_len <= 0) return NULL;
    fastp_oracle* o = (fastp_oracle*)calloc(1, sizeof(*o));
    o->p = *params;
    if (params->adapter_seq_r1 && params->adapter_seq_r1[0]) {
        o->alen1 = (int)strlen(params->adapter_seq_r1);
        if (o->alen1 > FASTP_GPU_MAX_ADAPTER_LEN) { free(o); return NULL; }
        memcpy(o->adapter1, params->adapter_seq_r1, (size_t)o->alen1);
        o->has_seq_r1 = 1;
    }
    if (params->adapter_seq_r2 && params->adapter_seq_r2[0]) {
        o->alen2 = (int)strlen(params->adapter_seq_r2);
        if (o->alen2 > FASTP_GPU_MAX_ADAPTER_LEN) { free(o); return NULL; }
        memcpy(o->adapter2, params->adapter_seq_r2, (size_t)o->alen2);
        o->has_seq_r2 = 1;
    }
    o->p.adapter_seq_r1 = o->adapter1;
    o->p.adapter_seq_r2 = o->adapter2;
    if (params->n_adapter_fasta > 0 && params->adapter_fasta) {
        o->n_fasta = params->n_adapter_fasta;
        o->fasta = (char**)calloc((size_t)o->n_fasta, sizeof(char*));
        o->fasta_len = (int*)calloc((size_t)o->n_fasta, sizeof(int));
        for (int i = 0; i < o->n_fasta; i++) {
            o->fasta_len[i] = (int)strlen(params->adapter_fasta[i]);
            o->fasta[i] = (char*)malloc((size_t)o->fasta_len[i] + 1);
            memcpy(o->fasta[i], params->adapter_fasta[i], (size_t)o->fasta_len[i] + 1);
        }
    }
    o->p.adapter_fasta = (const char* const*)o->fasta;
    if (params->overrep_enabled) {
        const char* const* lists[2] = {params->overrep_seqs1, params->overrep_seqs2};
        const int ns[2] = {params->n_overrep_seqs1, params->paired ? params->n_overrep_seqs2 : 0};
        for (int m = 0; m < 2; m++) {
            o->n_ovr[m] = ns[m];
            o->ovr[m] = (char**)calloc((size_t)ns[m] + 1, sizeof(char*));
            o->ovr_len[m] = (int*)calloc((size_t)ns[m] + 1, sizeof(int));
            for (int i = 0; i < ns[m]; i++) {
                o->ovr_len[m][i] = (int)strlen(lists[m][i]);
                o->ovr[m][i] = (char*)malloc((size_t)o->ovr_len[m][i] + 1);
                memcpy(o->ovr[m][i], lists[m][i], (size_t)o->ovr_len[m][i] + 1);
            }
        }
    }
    fastp_oracle_counter_layout_params(params, &o->L);
    o->ctr = (int64_t*)calloc((size_t)o->L.total, sizeof(int64_t));
    o->ctr[0] = FASTP_GPU_ABI_VERSION;
    o->ctr[1] = o->L.cycles;
    o->ctr[2] = params->insert_size_max;
    if (params->dup_enabled) {
        o->dup = orc_dup_create(params->dup_accuracy_level);
        if (!o->dup) { free(o->ctr); free(o); return NULL; }
    }
    return o;
}

void fastp_oracle_destroy(fastp_oracle* o) {
    if (!o) return;
    if (o->dup) { free(o->dup->buf); free(o->dup->primes); free(o->dup); }
    free(o->ctr);
    for (int m = 0; m < 2; m++) {
        for (int i = 0; i < o->n_ovr[m]; i++) free(o->ovr[m][i]);
        free(o->ovr[m]);
        free(o->ovr_len[m]);
    }
    for (int i = 0; i < o->n_fasta; i++) free(o->fasta[i]);
    free(o->fasta);
    free(o->fasta_len);
    free(o);
}

int fastp_oracle_counters(fastp_oracle* o, int64_t* out, int64_t n) {
    if (!o || !out || n != o->L.total) return FASTP_GPU_E_INVALID;
    memcpy(out, o->ctr, sizeof(int64_t) * (size_t)n);
    if (o->dup) {
        out[o->L.dup_total] = (int64_t)o->dup->total;
        out[o->L.dup_count] = (int64_t)o->dup->dups;
    }
    return FASTP_GPU_OK;
}

/* a mutable read: points into a scratch copy of the original (Read, read.h:14-47) */
typedef struct orc_read {
    char* seq;
    char* qual;
    int len;
    int front; /* bases erased at the 5' end of the ORIGINAL so far */
} orc_read;

static void orc_add_filter_result(fastp_oracle* o, int result, int n) { /* filterresult.cpp:28-32 */
    if (result < FASTP_PASS_FILTER || result >= FASTP_FILTER_RESULT_TYPES) return;
    o->ctr[o->L.filter_stats + result] += n;
}

static void orc_trim_front(orc_read* r, int len) { /* Read::trimFront read.cpp:69-73 */
    len = ORC_MIN(r->len - 1, len);
    if (len < 0) { /* erase(0, npos) on an empty string */
        return;
    }
    r->seq += len; r->qual += len; r->len -= len; r->front += len;
}

static void orc_resize(orc_read* r, int len) { /* Read::resize read.cpp:62-67 */
    if (len > r->len || len < 0) return;
    r->len = len;
}

/* apply trimAndCut to a mutable read; returns 0 for NULL (read untouched) */
static int orc_apply_trim_and_cut(fastp_oracle* o, orc_read* r, int front, int tail, int* ft) {
    int f = 0, l = 0;
    *ft = 0;
    if (!fastp_oracle_trim_and_cut(&o->p, r->seq, r->qual, r->len, front, tail, &f, &l)) return 0;
    r->seq += f; r->qual += f; r->front += f; r->len = l;
    *ft = f;
    return 1;
}

/* trimBySequence on a mutable read + bookkeeping for the result record */
static int orc_apply_trim_by_sequence_req(fastp_oracle* o, orc_read* r, const char* adapter, int alen, int matchReq,
                                          int* out_pos, int* out_len) {
    int pos = 0;
    if (!fastp_oracle_trim_by_sequence(r->seq, r->len, adapter, alen, matchReq, &pos)) return 0;
    int adapter_len;
    if (pos < 0) { /* adaptertrimmer.cpp:138-145 */
        adapter_len = alen + pos;
        r->len = 0;
    } else {
        adapter_len = r->len - pos;
        orc_resize(r, pos);
    }
    /* FilterResult::addAdapterTrimmed(string,bool) filterresult.cpp:124-152 */
    if (adapter_len > 0) o->ctr[o->L.adapter_bases] += adapter_len;
    *out_pos = pos;
    *out_len = adapter_len;
    return 1;
}

/* AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62): every --adapter_fasta sequence in
 * turn on the (shrinking) read; one event per cut for the host's adapter-string replay */
static int orc_trim_by_multi_sequences(fastp_oracle* o, orc_read* r, uint32_t read_index, fastp_gpu_results* res, int* err) {
    int matchReq = 4;
    if (o->n_fasta > 16) matchReq = 5;
    if (o->n_fasta > 256) matchReq = 6;
    int trimmed = 0;
    for (int i = 0; i < o->n_fasta; i++) {
        int pos, alen;
        if (!orc_apply_trim_by_sequence_req(o, r, o->fasta[i], o->fasta_len[i], matchReq, &pos, &alen)) continue;
        trimmed = 1;
        if (res->adapter_events && res->n_adapter_events) {
            int k = *res->n_adapter_events;
            if (k < res->adapter_events_capacity) {
                fastp_gpu_adapter_event* e = &res->adapter_events[k];
                e->read = read_index; e->pos = (int16_t)pos; e->len = (uint16_t)alen; e->adapter = (uint16_t)i; e->reserved = 0;
                *res->n_adapter_events = k + 1;
            } else {
                *err = FASTP_GPU_E_OVERFLOW;
            }
        } else {
            *err = FASTP_GPU_E_INVALID; /* adapter_fasta needs an event list */
        }
    }
    return trimmed;
}

static int orc_apply_trim_by_sequence(fastp_oracle* o, orc_read* r, const char* adapter, int alen,
                                      fastp_gpu_read_result* rr) {
    int pos = 0;
    if (!fastp_oracle_trim_by_sequence(r->seq, r->len, adapter, alen, 4, &pos)) return 0;
    int adapter_len;
    if (pos < 0) { /* adaptertrimmer.cpp:138-145 */
        adapter_len = alen + pos;
        r->len = 0;
    } else {
        adapter_len = r->len - pos;
        orc_resize(r, pos);
    }
    /* FilterResult::addAdapterTrimmed(string,bool) filterresult.cpp:124-152 */
    if (adapter_len > 0) o->ctr[o->L.adapter_bases] += adapter_len;
    rr->adapter_pos = (int16_t)pos;
    rr->adapter_len = (uint16_t)adapter_len;
    return 1;
}

static void orc_trim_poly_x(fastp_oracle* o, orc_read* r, fastp_gpu_read_result* rr) {
    int poly, trimmed;
    int nl = fastp_oracle_trim_poly_x(r->seq, r->len, o->p.poly_x_min_len, &poly, &trimmed);
    if (poly >= 0) { /* addPolyXTrimmed filterresult.cpp:186-189 */
        o->ctr[o->L.polyx_reads + poly] += 1;
        o->ctr[o->L.polyx_bases + poly] += trimmed;
        rr->flags |= FASTP_GPU_RF_POLYX;
    }
    r->len = nl;
}


static void orc_finish_result(fastp_gpu_read_result* rr, const orc_read* r, int code) {
    rr->front = (uint16_t)r->front;
    rr->len = (uint16_t)r->len;
    rr->code = (uint8_t)code;
}

/* Stats::statRead incl. the overrepresentation analysis (stats.cpp:270-288): for every
 * `sampling`-th read THIS Stats object sees, slide the five step lengths over the read and count
 * the substrings that are seed sequences, skipping `step` bases after a hit */
static void orc_stat_read_slot(fastp_oracle* o, int slot, const char* seqstr, const char* qualstr, int len) {
    int64_t* st = o->ctr + o->L.stats[slot];
    if (o->p.overrep_enabled && o->p.overrep_sampling > 0 && st[o->L.st_reads] % o->p.overrep_sampling == 0) {
        const int m = slot >= 2 ? 1 : 0;
        const int evalLen = (int)o->L.eval_len[slot];
        const int steps[5] = {10, 20, 40, 100, ORC_MIN(150, evalLen - 2)};
        int64_t* cnt = o->ctr + o->L.overrep_count[slot];
        int64_t* dist = o->ctr + o->L.overrep_dist[slot];
        for (int s = 0; s < 5; s++) {
            const int step = steps[s];
            if (step <= 0) continue; /* substr(i, 0) is empty and never a seed; negative lengths are not modelled */
            for (int i = 0; i < len - step; i++) {
                int hit = -1;
                for (int k = 0; k < o->n_ovr[m]; k++)
                    if (o->ovr_len[m][k] == step && memcmp(o->ovr[m][k], seqstr + i, (size_t)step) == 0) { hit = k; break; }
                if (hit >= 0) {
                    cnt[hit]++;
                    for (int p = i; p < step + i && p < evalLen; p++) dist[(int64_t)hit * evalLen + p]++;
                    i += step;
                }
            }
        }
    }
    orc_stat_read(&o->L, st, seqstr, qualstr, len);
}

/* ---- single-end loop body: seprocessor.cpp:204-296 ---------------------- */
static void orc_process_se(fastp_oracle* o, int read_index, char* seq, char* qual, int len, fastp_gpu_read_result* rr,
                           fastp_gpu_results* res, int* err) {
    const fastp_gpu_params* p = &o->p;
    memset(rr, 0, sizeof(*rr));
    orc_read or1 = {seq, qual, len, 0};
    orc_stat_read_slot(o, FASTP_GPU_STATS_PRE1, or1.seq, or1.qual, or1.len); /* :210 */
    int dedupOut = 0;
    if (o->dup) { /* :213-218 checkRead duplicate.cpp:122-134 */
        uint64_t pos[8] = {0};
        orc_seq2intvector(o->dup->primes, o->dup->bufNum, o->dup->offsetMask, or1.seq, or1.len, pos, 0);
        int isDup = orc_dup_apply(o->dup, pos);
        if (isDup) rr->flags |= FASTP_GPU_RF_DUP;
        if (p->dedup && isDup) dedupOut = 1;
    }
    if (p->umi_len1 > 0) /* :232-233, umiprocessor.cpp:19-22 */
        orc_trim_front(&or1, ORC_MIN(or1.len, p->umi_len1) + p->umi_skip);
    int ft = 0;
    int alive = orc_apply_trim_and_cut(o, &or1, p->trim_front1, p->trim_tail1, &ft); /* :237 */
    if (alive && p->poly_g) /* :239-242 */
        or1.len = fastp_oracle_trim_poly_g(or1.seq, or1.len, p->poly_g_min_len);
    int isAdapterDimer = 0;
    if (alive && p->adapter_enabled) { /* :244-261 */
        int trimmed = 0;
        if (o->has_seq_r1) trimmed = orc_apply_trim_by_sequence(o, &or1, o->adapter1, o->alen1, rr);
        if (o->n_fasta) trimmed |= orc_trim_by_multi_sequences(o, &or1, (uint32_t)read_index, res, err); /* :249-251 */
        if (trimmed) { o->ctr[o->L.adapter_reads] += 1; rr->flags |= FASTP_GPU_RF_ADAPTER; }
        if (trimmed && or1.len <= p->dimer_max_len) isAdapterDimer = 1;
    }
    if (alive && p->poly_x) orc_trim_poly_x(o, &or1, rr); /* :263-266 */
    if (alive && p->max_len1 > 0 && p->max_len1 < or1.len) orc_resize(&or1, p->max_len1); /* :268-271 */
    int result = alive ? fastp_oracle_pass_filter(p, or1.seq, or1.qual, or1.len) : FASTP_FAIL_LENGTH;
    if (isAdapterDimer) result = FASTP_FAIL_ADAPTER_DIMER;
    orc_add_filter_result(o, result, 1); /* :278 */
    if (!dedupOut && alive && result == FASTP_PASS_FILTER) /* :280-290 */
        orc_stat_read_slot(o, FASTP_GPU_STATS_POST1, or1.seq, or1.qual, or1.len);
    if (!alive) rr->flags |= FASTP_GPU_RF_NULL;
    orc_finish_result(rr, &or1, result);
}

/* statInsertSize peprocessor.cpp:710-723 */
static void orc_stat_isize(fastp_oracle* o, int l1, int l2, const fastp_oracle_overlap* ov, int ft1, int ft2) {
    int isize = o->p.insert_size_max;
    if (ov->overlapped) {
        if (ov->offset > 0) isize = l1 + l2 - ov->overlap_len + ft1 + ft2;
        else isize = ov->overlap_len + ft1 + ft2;
    }
    if (isize > o->p.insert_size_max) isize = o->p.insert_size_max;
    if (isize < 0) return; /* cannot happen; guards the histogram */
    o->ctr[o->L.isize + isize]++;
}

/* BaseCorrector::correctByOverlapAnalysis basecorrector.cpp:16-83 */
static void orc_correct(fastp_oracle* o, orc_read* r1, orc_read* r2, const fastp_oracle_overlap* ov,
                        int pair_index, fastp_gpu_results* res, fastp_gpu_read_result* rr1,
                        fastp_gpu_read_result* rr2, int* err) {
    if (ov->diff == 0 || !ov->overlapped) return;
    int ol = ov->overlap_len;
    int start1 = ORC_MAX(0, ov->offset);
    int start2 = r2->len - ORC_MAX(0, -ov->offset) - 1;
    const char GOOD = orc_num2qual(30), BAD = orc_num2qual(14);
    int corrected = 0, r1c = 0, r2c = 0;
    for (int i = 0; i < ol; i++) {
        int p1 = start1 + i, p2 = start2 - i;
        if (r1->seq[p1] != orc_complement(r2->seq[p2])) {
            int which = -1, pos = 0;
            char nb = 0, nq = 0, from = 0;
            if (r1->qual[p1] >= GOOD && r2->qual[p2] <= BAD) {
                from = r2->seq[p2];
                nb = orc_complement(r1->seq[p1]); nq = r1->qual[p1];
                r2->seq[p2] = nb; r2->qual[p2] = nq;
                which = 1; pos = r2->front + p2; r2c = 1;
            } else if (r2->qual[p2] >= GOOD && r1->qual[p1] <= BAD) {
                from = r1->seq[p1];
                nb = orc_complement(r2->seq[p2]); nq = r2->qual[p2];
                r1->seq[p1] = nb; r1->qual[p1] = nq;
                which = 0; pos = r1->front + p1; r1c = 1;
            }
            if (which >= 0) {
                corrected++;
                /* addCorrection filterresult.cpp:99-103 */
                o->ctr[o->L.correction + (from & 7) * 8 + (nb & 7)]++;
                if (res->corrections && res->n_corrections) {
                    if (*res->n_corrections < res->corrections_capacity) {
                        fastp_gpu_correction* c = &res->corrections[*res->n_corrections];
                        c->read = (uint32_t)(2 * pair_index + which);
                        c->pos = (uint16_t)pos; c->base = (uint8_t)nb; c->qual = (uint8_t)nq;
                        (*res->n_corrections)++;
                    } else {
                        *err = FASTP_GPU_E_OVERFLOW;
                    }
                }
            }
        }
    }
    if (corrected > 0) { /* :75-80 */
        o->ctr[o->L.corrected_reads] += (r1c && r2c) ? 2 : 1;
        if (r1c) rr1->flags |= FASTP_GPU_RF_CORRECTED;
        if (r2c) rr2->flags |= FASTP_GPU_RF_CORRECTED;
    }
}

static fastp_oracle_overlap orc_analyze_reads(fastp_oracle* o, const orc_read* r1, const orc_read* r2,
                                              int allowGap) {
    return fastp_oracle_analyze(r1->seq, r1->len, r2->seq, r2->len, o->p.overlap_diff_limit,
                                o->p.overlap_require, o->p.overlap_diff_percent_limit / 100.0, allowGap);
}

/* ---- paired-end loop body: peprocessor.cpp:383-643 ----------------------- */
static void orc_process_pe(fastp_oracle* o, int pair_index, uint32_t batch_flags, char* s1, char* q1,
                           int l1, char* s2, char* q2, int l2, fastp_gpu_results* res, int* err) {
    const fastp_gpu_params* p = &o->p;
    fastp_gpu_read_result* rr1 = &res->r1[pair_index];
    fastp_gpu_read_result* rr2 = &res->r2[pair_index];
    fastp_gpu_pair_result* pr = &res->pair[pair_index];
    memset(rr1, 0, sizeof(*rr1));
    memset(rr2, 0, sizeof(*rr2));
    memset(pr, 0, sizeof(*pr));
    const int thread0 = (batch_flags & FASTP_GPU_BATCH_STAT_ISIZE) != 0;
    orc_read or1 = {s1, q1, l1, 0}, or2 = {s2, q2, l2, 0};

    orc_stat_read_slot(o, FASTP_GPU_STATS_PRE1, or1.seq, or1.qual, or1.len); /* :393 */
    orc_stat_read_slot(o, FASTP_GPU_STATS_PRE2, or2.seq, or2.qual, or2.len); /* :394 */

    int dedupOut = 0;
    if (o->dup) { /* :397-402, checkPair duplicate.cpp:136-148 */
        uint64_t pos[8] = {0};
        orc_seq2intvector(o->dup->primes, o->dup->bufNum, o->dup->offsetMask, or1.seq, or1.len, pos, 0);
        orc_seq2intvector(o->dup->primes, o->dup->bufNum, o->dup->offsetMask, or2.seq, or2.len, pos, or1.len);
        int isDup = orc_dup_apply(o->dup, pos);
        if (isDup) { rr1->flags |= FASTP_GPU_RF_DUP; rr2->flags |= FASTP_GPU_RF_DUP; }
        if (p->dedup && isDup) dedupOut = 1;
    }
    /* umi processing :419-420 (umiprocessor.cpp:19-49, in-read part) */
    if (p->umi_len1 > 0) orc_trim_front(&or1, ORC_MIN(or1.len, p->umi_len1) + p->umi_skip);
    if (p->umi_len2 > 0) orc_trim_front(&or2, ORC_MIN(or2.len, p->umi_len2) + p->umi_skip);

    int ft1 = 0, ft2 = 0;
    int a1 = orc_apply_trim_and_cut(o, &or1, p->trim_front1, p->trim_tail1, &ft1); /* :425 */
    int a2 = orc_apply_trim_and_cut(o, &or2, p->trim_front2, p->trim_tail2, &ft2); /* :426 */
    const int both = a1 && a2;
    if (both && p->poly_g) { /* :428-431 */
        or1.len = fastp_oracle_trim_poly_g(or1.seq, or1.len, p->poly_g_min_len);
        or2.len = fastp_oracle_trim_poly_g(or2.seq, or2.len, p->poly_g_min_len);
    }
    int isizeEvaluated = 0, isAdapterDimer = 0;
    fastp_oracle_overlap ov = {0, 0, 0, 0, 0};
    int ovComputed = 0;
    if (both && (p->adapter_enabled || p->correction || thread0 || p->merge)) { /* :438-441 */
        ov = orc_analyze_reads(o, &or1, &or2, 0);
        ovComputed = 1;
    }
    if (both && (p->adapter_enabled || p->correction)) { /* :443-485 */
        fastp_oracle_overlap ovA = p->allow_gap_overlap_trimming ? orc_analyze_reads(o, &or1, &or2, 1) : ov;
        if (thread0) { orc_stat_isize(o, or1.len, or2.len, &ov, ft1, ft2); isizeEvaluated = 1; }
        if (p->correction && !ovA.has_gap) orc_correct(o, &or1, &or2, &ovA, pair_index, res, rr1, rr2, err);
        if (p->adapter_enabled) {
            int trimmed = 0;
            if (ovA.overlapped && ovA.offset < 0) { /* trimByOverlapAnalysis adaptertrimmer.cpp:17-46 */
                int ol = ovA.overlap_len;
                int len1 = ORC_MIN(or1.len, ol + ft2);
                int len2 = ORC_MIN(or2.len, ol + ft1);
                rr1->adapter_pos = (int16_t)len1; rr1->adapter_len = (uint16_t)(or1.len - len1);
                rr2->adapter_pos = (int16_t)len2; rr2->adapter_len = (uint16_t)(or2.len - len2);
                /* addAdapterTrimmed(a1,a2) filterresult.cpp:154-155 */
                o->ctr[o->L.adapter_bases] += (or1.len - len1) + (or2.len - len2);
                orc_resize(&or1, len1);
                orc_resize(&or2, len2);
                trimmed = 1;
                rr1->flags |= FASTP_GPU_RF_ADAPTER_OV; rr2->flags |= FASTP_GPU_RF_ADAPTER_OV;
            }
            int t1 = trimmed, t2 = trimmed;
            if (!trimmed) { /* :460-466 */
                if (o->has_seq_r1) t1 = orc_apply_trim_by_sequence(o, &or1, o->adapter1, o->alen1, rr1);
                if (o->has_seq_r2) t2 = orc_apply_trim_by_sequence(o, &or2, o->adapter2, o->alen2, rr2);
            }
            if (o->n_fasta) { /* :467-470 */
                t1 |= orc_trim_by_multi_sequences(o, &or1, 2u * (uint32_t)pair_index, res, err);
                t2 |= orc_trim_by_multi_sequences(o, &or2, 2u * (uint32_t)pair_index + 1u, res, err);
            }
            if (t1) { o->ctr[o->L.adapter_reads] += 1; rr1->flags |= FASTP_GPU_RF_ADAPTER; } /* :472-475 */
            if (t2) { o->ctr[o->L.adapter_reads] += 1; rr2->flags |= FASTP_GPU_RF_ADAPTER; }
            if ((t1 || t2) && or1.len <= p->dimer_max_len && or2.len <= p->dimer_max_len) /* :480-484 */
                isAdapterDimer = 1;
        }
    }
    if (p->overlapped_out && both) { /* :488-495, mOverlappedWriter: analyze(r1, r2, diffLimit, require, 0) */
        fastp_oracle_overlap ovx = fastp_oracle_analyze(or1.seq, or1.len, or2.seq, or2.len, p->overlap_diff_limit,
                                                        p->overlap_require, 0.0, 0);
        if (ovx.overlapped) {
            /* :491 new string(r1->mSeq->substr(max(0, offset)), overlap_len): the (str, pos) constructor, so the
             * stream gets what FOLLOWS the overlapped region of read 1, r1[start + overlap_len, len1) */
            int pos = ORC_MAX(0, ovx.offset) + ovx.overlap_len;
            rr1->reserved = (uint16_t)(FASTP_GPU_OVOUT_HIT | (unsigned)pos);
            rr2->reserved = (uint16_t)(or1.len - pos);
        }
    }
    if (thread0 && !isizeEvaluated && both) { /* :497-504 */
        if (!ovComputed) { ov = orc_analyze_reads(o, &or1, &or2, 0); ovComputed = 1; }
        orc_stat_isize(o, or1.len, or2.len, &ov, ft1, ft2);
        isizeEvaluated = 1;
    }
    if (both && p->poly_x) { /* :506-509 */
        orc_trim_poly_x(o, &or1, rr1);
        orc_trim_poly_x(o, &or2, rr2);
    }
    if (both) { /* :511-516 */
        if (p->max_len1 > 0 && p->max_len1 < or1.len) orc_resize(&or1, p->max_len1);
        if (p->max_len2 > 0 && p->max_len2 < or2.len) orc_resize(&or2, p->max_len2);
    }
    int mergeProcessed = 0;
    int code1 = 0, code2 = 0;
    if (p->merge && both) { /* :518-561 */
        ov = orc_analyze_reads(o, &or1, &or2, 0);
        ovComputed = 1;
        if (ov.overlapped) {
            /* OverlapAnalysis::merge overlapanalysis.cpp:148-179 */
            int ol = ov.overlap_len;
            int len1 = ol + ORC_MAX(0, ov.offset);
            int len2 = 0;
            if (ov.offset > 0) len2 = or2.len - ol;
            int m1 = ORC_MIN(len1, or1.len); /* substr(0,len1) clamps */
            int m2 = (ov.offset > 0) ? ORC_MAX(0, ORC_MIN(len2, or2.len - ol)) : 0;
            int mlen = m1 + m2;
            char* ms = (char*)malloc((size_t)mlen + 1);
            char* mq = (char*)malloc((size_t)mlen + 1);
            memcpy(ms, or1.seq, (size_t)m1);
            memcpy(mq, or1.qual, (size_t)m1);
            for (int k = 0; k < m2; k++) { /* rc(r2)[ol+k] = comp(r2[len2r-1-ol-k]) */
                int src = or2.len - 1 - ol - k;
                ms[m1 + k] = orc_complement(or2.seq[src]);
                mq[m1 + k] = or2.qual[src];
            }
            int result = fastp_oracle_pass_filter(p, ms, mq, mlen);
            orc_add_filter_result(o, result, 2);
            if (result == FASTP_PASS_FILTER) {
                orc_stat_read_slot(o, FASTP_GPU_STATS_POST1, ms, mq, mlen);
                o->ctr[o->L.merged_pairs] += 1; /* mergedCount -> addMergedPairs :688-690 */
                rr1->flags |= FASTP_GPU_RF_MERGED; rr2->flags |= FASTP_GPU_RF_MERGED;
            }
            free(ms); free(mq);
            code1 = code2 = result;
            if (!p->overlapped_out) { /* with --overlapped_out the fields keep its values; m1 / m2 follow from the pair record */
                rr1->reserved = (uint16_t)m1; /* merged_<len1>_<len2> for the host's name tag */
                rr2->reserved = (uint16_t)m2;
            }
            mergeProcessed = 1;
        } else if (p->merge_include_unmerged) {
            code1 = fastp_oracle_pass_filter(p, or1.seq, or1.qual, or1.len);
            code2 = fastp_oracle_pass_filter(p, or2.seq, or2.qual, or2.len);
            if (isAdapterDimer) { code1 = code2 = FASTP_FAIL_ADAPTER_DIMER; }
            orc_add_filter_result(o, code1, 1);
            if (code1 == FASTP_PASS_FILTER && !dedupOut)
                orc_stat_read_slot(o, FASTP_GPU_STATS_POST1, or1.seq, or1.qual, or1.len);
            orc_add_filter_result(o, code2, 1);
            if (code2 == FASTP_PASS_FILTER && !dedupOut)
                orc_stat_read_slot(o, FASTP_GPU_STATS_POST1, or2.seq, or2.qual, or2.len);
            mergeProcessed = 1;
        }
    }
    if (!mergeProcessed) { /* :563-621 */
        code1 = a1 ? fastp_oracle_pass_filter(p, or1.seq, or1.qual, or1.len) : FASTP_FAIL_LENGTH;
        code2 = a2 ? fastp_oracle_pass_filter(p, or2.seq, or2.qual, or2.len) : FASTP_FAIL_LENGTH;
        if (isAdapterDimer) { code1 = code2 = FASTP_FAIL_ADAPTER_DIMER; }
        orc_add_filter_result(o, ORC_MAX(code1, code2), 2);
        if (!dedupOut && a1 && code1 == FASTP_PASS_FILTER && a2 && code2 == FASTP_PASS_FILTER) {
            if (!p->merge) { /* :588-591 */
                orc_stat_read_slot(o, FASTP_GPU_STATS_POST1, or1.seq, or1.qual, or1.len);
                orc_stat_read_slot(o, FASTP_GPU_STATS_POST2, or2.seq, or2.qual, or2.len);
            }
        }
    }
    if (!a1) rr1->flags |= FASTP_GPU_RF_NULL;
    if (!a2) rr2->flags |= FASTP_GPU_RF_NULL;
    orc_finish_result(rr1, &or1, code1);
    orc_finish_result(rr2, &or2, code2);
    pr->ov_offset = (int16_t)ov.offset;
    pr->ov_len = (uint16_t)ov.overlap_len;
    pr->ov_diff = (uint16_t)ov.diff;
    pr->flags = (uint16_t)((ov.overlapped ? FASTP_GPU_PF_OVERLAPPED : 0) | (ov.has_gap ? FASTP_GPU_PF_HAS_GAP : 0) |
                           (isizeEvaluated ? FASTP_GPU_PF_ISIZE : 0));
}

int fastp_oracle_process(fastp_oracle* o, int n, uint32_t batch_flags, int row_stride,
                         const char* seq1, const char* qual1, const int32_t* len1,
                         const char* seq2, const char* qual2, const int32_t* len2,
                         fastp_gpu_results* res) {
    if (!o || !res || n < 0 || !res->r1) return FASTP_GPU_E_INVALID;
    const int paired = o->p.paired;
    if (paired && (!seq2 || !qual2 || !len2 || !res->r2 || !res->pair)) return FASTP_GPU_E_INVALID;
    int err = FASTP_GPU_OK;
    if (res->n_corrections) *res->n_corrections = 0;
    if (res->n_adapter_events) *res->n_adapter_events = 0;
    const int cap = o->p.max_len + 1;
    char* buf = (char*)malloc((size_t)cap * 4);
    for (int i = 0; i < n; i++) {
        int l1 = len1[i];
        if (l1 < 0 || l1 > o->p.max_len || l1 > row_stride) { err = FASTP_GPU_E_TOO_LONG; break; }
        char* s1 = buf; char* q1 = buf + cap;
        memcpy(s1, seq1 + (size_t)i * row_stride, (size_t)l1); s1[l1] = 0;
        memcpy(q1, qual1 + (size_t)i * row_stride, (size_t)l1); q1[l1] = 0;
        if (!paired) {
            orc_process_se(o, i, s1, q1, l1, &res->r1[i], res, &err);
        } else {
            int l2 = len2[i];
            if (l2 < 0 || l2 > o->p.max_len || l2 > row_stride) { err = FASTP_GPU_E_TOO_LONG; break; }
            char* s2 = buf + 2 * cap; char* q2 = buf + 3 * cap;
            memcpy(s2, seq2 + (size_t)i * row_stride, (size_t)l2); s2[l2] = 0;
            memcpy(q2, qual2 + (size_t)i * row_stride, (size_t)l2); q2[l2] = 0;
            orc_process_pe(o, i, batch_flags, s1, q1, l1, s2, q2, l2, res, &err);
        }
    }
    free(buf);
    return err;
}
