/* fastp_oracle.h - CPU restatement of fastp's per-read worker loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product path (fastp_amd/, include/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (1) the reference's own known-answer tests (Filter::test,
 * OverlapAnalysis::test, AdapterTrimmer::test, BaseCorrector::test,
 * PolyX::test), (2) outputs of the real reference binary built from
 * /root/reference by oracle/build_ref.sh (oracle/_ref/fastp_ref -w 1) on
 * testdata/R1.fq+R2.fq and on synthetic inputs, committed as fixtures under
 * tests/golden/.
 *
 * The oracle works on plain ASCII reads (any alphabet), exactly like the
 * reference; it shares only the parameter / result / counter *declarations*
 * of include/fastp_gpu.h with the product.
 */
#ifndef FASTP_ORACLE_H
#define FASTP_ORACLE_H

#include "../include/fastp_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fastp_oracle fastp_oracle;

fastp_oracle* fastp_oracle_create(const fastp_gpu_params* params);
void fastp_oracle_destroy(fastp_oracle* o);

/* ASCII batch: row i of seq/qual starts at i*row_stride and holds len[i]
 * characters.  seq2/qual2/len2 are NULL for single-end.  Result pointers as in
 * fastp_gpu_results (host memory). Returns 0 or a FASTP_GPU_E_* code. */
int fastp_oracle_process(fastp_oracle* o, int n, uint32_t batch_flags, int row_stride,
                         const char* seq1, const char* qual1, const int32_t* len1,
                         const char* seq2, const char* qual2, const int32_t* len2,
                         fastp_gpu_results* res);

/* counter block; cycles = max_len (2*max_len in merge mode), see fastp_gpu.h */
int fastp_oracle_counters(fastp_oracle* o, int64_t* out, int64_t n);
int fastp_oracle_cycles_for(const fastp_gpu_params* params);
void fastp_oracle_counter_layout(int cycles, int insert_size_max, fastp_gpu_counter_layout* out);
void fastp_oracle_counter_layout_params(const fastp_gpu_params* params, fastp_gpu_counter_layout* out);

/* ---- individual functions, exported for the known-answer tests ---------- */
typedef struct fastp_oracle_overlap {
    int overlapped, offset, overlap_len, diff, has_gap;
} fastp_oracle_overlap;

/* OverlapAnalysis::analyze (overlapanalysis.cpp:17-146) */
fastp_oracle_overlap fastp_oracle_analyze(const char* r1, int len1, const char* r2, int len2,
                                          int diff_limit, int overlap_require,
                                          double diff_percent_limit, int allow_gap);
/* Filter::trimAndCut (filter.cpp:68-207): returns 1 and (front,rlen), or 0 for NULL */
int fastp_oracle_trim_and_cut(const fastp_gpu_params* p, const char* seq, const char* qual, int len,
                              int front, int tail, int* out_front, int* out_len);
/* PolyX::trimPolyG (polyx.cpp:16-42): returns new length */
int fastp_oracle_trim_poly_g(const char* seq, int len, int compare_req);
/* PolyX::trimPolyX (polyx.cpp:49-116): returns new length; *poly_base in 0..3 (A,T,C,G) or -1 */
int fastp_oracle_trim_poly_x(const char* seq, int len, int compare_req, int* poly_base, int* trimmed);
/* AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157): returns 1 if found, *pos */
int fastp_oracle_trim_by_sequence(const char* seq, int len, const char* adapter, int alen,
                                  int match_req, int* pos);
/* Matcher::matchWithOneInsertion (matcher.cpp:10-54) */
int fastp_oracle_match_one_insertion(const char* ins, const char* normal, int cmplen, int diff_limit);
/* Filter::passFilter (filter.cpp:15-57) on a non-NULL read */
int fastp_oracle_pass_filter(const fastp_gpu_params* p, const char* seq, const char* qual, int len);
/* Duplicate::seq2intvector for a read / pair (duplicate.cpp:111-148): out[0..bufnum) */
int fastp_oracle_dup_hash(int accuracy_level, const char* s1, int l1, const char* s2, int l2,
                          uint64_t* out);

/* bit positions of n units in every bloom buffer, out[n][bufnum] (stateless; returns bufnum) */
int fastp_oracle_dup_bits_batch(int accuracy_level, int n, int row_stride, const char* seq1, const int32_t* len1,
                                const char* seq2, const int32_t* len2, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
