// fq_types.h - plain-old-data shared by the host side of the C ABI and the
// gfx950 kernels: the device parameter block, the LDS tile layout and the
// kernel argument block.  No HIP types in here.
#pragma once
#include <stdint.h>

namespace fq {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
struct alignas(16) u32x4 { u32 x, y, z, w; };
struct alignas(8) u32x2 { u32 x, y; };

// ---- 2-bit base code (also the order of fastp's k-mer code, stats.cpp:294-311,
//      so complement(code) == code ^ 1 and the 5-mer index needs no remapping
//      of symbols, only of significance order) --------------------------------
enum { CODE_A = 0, CODE_T = 1, CODE_C = 2, CODE_G = 3 };
// statistics class of a base: the four codes, then N
enum { CLS_N = 4, N_CLS = 5 };

// ---- per-cycle accumulator word in LDS ------------------------------------
// one u64 per (Stats slot, class, cycle): [cnt:14][q20:14][q30:14][qsum:22]
// -> a workgroup may accumulate at most CYC_MAX_READS reads per Stats slot
// between two flushes (enforced by the host when it sizes a launch).
enum { CYC_CNT_BITS = 14, CYC_Q20_SHIFT = 14, CYC_Q30_SHIFT = 28, CYC_QSUM_SHIFT = 42 };
enum { CYC_MAX_READS = (1 << CYC_CNT_BITS) - 1 };

// OverlapAnalysis::analyze result of a pair as one u32 whose order is the reference's scan order
// (all forward offsets ascending, then the reverse ones): [dir:1][offset:10][diff:16]; ~0 = none
enum { OV_KEY_DIFF_BITS = 16, OV_KEY_OFF_BITS = 10 };
static const u32 OV_KEY_NONE = 0xFFFFFFFFu;

// per-lane registers that hold the NEXT tile's input while the current tile is processed
enum { PF_Q = 3, PF_S = 1 };   // in 16-byte chunks

// quality table entry (one per Stats slot and quality character): the packed per-cycle increment of that
// character (u64), its histogram counter, and the constant 1 (0 for character 0 = "no base here") that the
// one-pass Stats path adds to the k-mer / histogram counters so that bases past a read's end count nothing
enum { QT_INC = 0, QT_COUNT = 2, QT_ONE = 3, QT_DWORDS = 4 };
enum { KMER_BINS = 1024 };
enum { MAX_DUP_BUFS = 8 };
enum { MAX_ADAPTER_WORDS = 16 }; // 256 bases = FASTP_GPU_MAX_ADAPTER_LEN
enum { ADAPT_WORDS = 18 };       // LDS words per adapter (16 + zero padding for window reads)

// read flags kept in LDS while a tile is processed (low byte == FASTP_GPU_RF_*)
enum {
    RS_NULL = 0x01, RS_DUP = 0x02, RS_ADAPTER = 0x04, RS_ADAPTER_OV = 0x08, RS_CORRECTED = 0x10,
    RS_MERGED = 0x20, RS_POLYX = 0x40,
    RS_HAS_N = 0x100,   // the read contains at least one 'N'
    RS_STAT_POST = 0x200, // goes into the post-filtering Stats
    RS_DIMER = 0x400,    // adapter dimer evidence (peprocessor.cpp:480-484), carried from the trim to the filter step
    RS_ISIZE = 0x800,    // statInsertSize ran for this pair (kept on R1 until the pair record is written)
    RS_MERGE_OV = 0x1000,// merge mode: the post-trim overlap analysis found an overlap (kept on R1)
    RS_POST_TO1 = 0x2000,// merge mode: this read's post-filtering stats go to slot POST1 (peprocessor.cpp:529,548,554)
    RS_POST_RC = 0x4000  // merge mode: this mate is the reverse-complemented tail of the merged read
};

struct DevParams {
    int paired, max_len, cycles;
    int sw_g, qw_g;         // global row strides in dwords (seq, qual)
    int trim_front1, trim_tail1, trim_front2, trim_tail2, max_len1, max_len2;
    int cut_front, cut_tail, cut_right;
    int wF, thrF;           // window, w*(33+Q)
    int wT, thrT;
    int wR, thrR, qRmin;    // qRmin = 33+Q (filter.cpp:159)
    int poly_g, poly_g_min, poly_x, poly_x_min;
    int adapter_enabled, dimer_max_len;
    int has_a1, has_a2, alen1, alen2;
    int n_fasta, fasta_match_req;   // --adapter_fasta list (adaptertrimmer.cpp:48-55)
    int fasta_max_len;              // its longest sequence (the lane plan takes lists of sequences <= 64 bases, round 6)
    u32 a1w[MAX_ADAPTER_WORDS], a2w[MAX_ADAPTER_WORDS];
    int correction;
    int merge, merge_include_unmerged;  // MergeOptions (peprocessor.cpp:518-561)
    int overlapped_out;     // --overlapped_out: the diffPercentLimit-0 analysis after adapter trimming (peprocessor.cpp:488-495)
    int allow_gap;          // AdapterOptions::allowGapOverlapTrimming (overlapanalysis.cpp:91-139)
    int overlap_require, overlap_diff_limit;
    int ov_limit_max;       // largest per-length mismatch limit (the LUT is non-decreasing): prefilter bound
    int qual_filter, qual_thr, n_base_limit, avg_qual_req;
    int length_filter, length_required, length_limit;
    int complexity_filter;
    int dup_enabled, dedup, dup_bufnum;
    int dup_npl;            // byte planes a prime of the hash needs (3: all primes a position can select are < 2^24; else 4)
    u64 dup_bits;           // Duplicate::mBufLenInBits
    int isize_max;
    int umi_len1, umi_len2, umi_skip;
    int need_overlap;       // adapter_enabled || correction  (peprocessor.cpp:438,443)
    int overrep, overrep_sampling;  // OverrepresentedSequenceAnasysOptions (-p, -P)
    int stats_one_pass;     // no option can move or edit a kept base (no front trim, no correction):
                            // one Stats pass classifies each base as kept / dropped (see phase_stats)
    int front_lane;         // round 5: the only such option is a front trim that is the SAME for every read that is written out
                            // (-f / -F, the UMI of --umi_loc read1 / read2 / per_read): kept bases stay at their original
                            // cycle in the Stats kernel's tables and the slab fold moves the POST Stats by the mate's front
                            // (lane plan only; fq_lane.h, fq_stats.h)
    int lane_front1, lane_front2;   // that front: UMI length + skip + --trim_front of the mate
    int front_per_read;     // round 6: --cut_front on the lane plan - a front of its own for every read (Filter::trimAndCut's forward
                            // quality cut, filter.cpp:97-127, on the window predicate the enabled tail / right cut already builds:
                            // same window and quality).  The Stats kernel (form 5) reads a read's front from its result record and
                            // counts kept bases at their ORIGINAL cycle; the slab fold moves the POST Stats by lane_front*, and
                            // fq_front_stats_kernel moves what a read's own cut adds to that, read by read (few reads have one)
    int merge_lane;         // round 5: --merge on the lane plan (no front trims, no UMI, no --cut_front) - the lane kernel runs the second
                            // overlap analysis on the trimmed reads, filters the merged read and leaves the part lengths in swin; the
                            // Stats kernel counts read 2's kept bases once more as the merged read's tail (reverse-complemented, at
                            // the merged read's cycles: slot 3 of its slabs, which the fold adds to the POST Stats of read 1)
    int corr_lane;          // round 5: -c on the lane plan - BaseCorrector's edits are applied to the reads in registers, the Stats
                            // kernel counts the ORIGINAL reads (kept -> PRE and POST), fq_corr_stats_kernel then moves the corrected
                            // positions' contributions to the POST Stats from the old base / quality to the new one
};

// LUTs living in global memory (built on the host with the reference's own
// double expressions so the device only does integer compares)
struct DevLuts {
    const u16* ov_limit;    // [max_len+1]  min(diffLimit, (int)(ol*pct/100.0))  overlapanalysis.cpp:51
    const u16* lowq_limit;  // [max_len+1]  floor(unqualPct*rlen/100.0)          filter.cpp:36
    const u16* cplx_min;    // [max_len+1]  least adjacent-diff count that passes filter.cpp:65
    const u32* dup_primes;  // [bufnum*512]                                      duplicate.cpp:66-84
    const u64* dup_posum;   // [(2*max_len+1)*bufnum]  sum_{p<n} prime[(p*B+i)&mask]*p
    const u32* dup_planes;  // [4][hp_nq][bufnum][dup_npl] the primes of four consecutive positions, one byte plane per
                            // dword (phase_hash_dot); entry (r, q) starts at position 4q + r.  nullptr: not built
    const u32* fasta_words; // [n_fasta][ADAPT_WORDS] packed --adapter_fasta sequences (zero padded)
    const int* fasta_len;   // [n_fasta]
};

// LDS layout, all offsets in dwords from the start of dynamic LDS
struct LdsLayout {
    // Shared part first (accumulators, tables), then the per-tile arrays: a workgroup stages `halves` tiles at a
    // time, tile h at +h * tile_stride dwords.  Half h of the workgroup works on an LDS base moved up by that much
    // and on a copy of this layout whose SHARED offsets are moved down by it (KernelArgs per half).
    int halves;     // 1 or 2 tiles in flight per workgroup (2: each half of the waves owns one, see fused_body)
    int tile_begin, tile_stride;   // dwords
    int bar;        // [2] per tile: arrival count and generation of the half-workgroup barrier
    int has_hp;     // the byte-plane prime table exists (hp itself may be negative in a half's copy)
    int P;          // pairs (PE) or reads (SE) per tile
    int NR;         // rows per tile: 2P (PE) or P (SE)
    int SW, QW;     // LDS row strides in dwords = the global row strides: a tile in LDS is a flat copy of
                    // the batch rows (bases past a read's length are masked by every consumer)
    int C;          // cycles
    int Cp;         // C rounded up to a multiple of 4: per-cycle accumulators are stored [slot][cycle][class]
                    // (cyc_index), so the four cycles of a quality dword sit at constant distances and the
                    // class only adds 8 bytes
    int seq, nmk, qual;            // [NR][SW], [NR][SW], [NR][QW]
    int rlen0, front, len, flags, ft, apos, alen, code;   // [NR] ints
    int swin;       // [NR] one-pass Stats window of a read: rlen0 | kept length << 16 (0 = not written out)
    int mlen;       // [NR] merge mode: bases of this mate that enter the merged read (else = len)
    int olen;       // [NR] --overlapped_out: post-adapter length, then the record's `reserved` value (phase_ovout_*); else -1
    int met;        // [NR][2] countQualityMetrics / countAdjacentDiffs of the final window (phase_metrics)
    int ov_off, ov_len, ov_diff, ov_flags;                // [P]; ov_off holds the packed no-gap scan key (OV_KEY_*),
                                                          // ov_len the key of the one-gap pass (allow_gap)
    int rc, rcn;    // [P][SW] paired: the reverse complement of the WHOLE read 2 (rc[k] = comp(r2[rlen0-1-k]), N kept as
                    // code 0) and its N mask - OverlapAnalysis::analyze's rc(r2') is a forward window of this row
    int cand, cand_cap;  // overlap candidates that passed the prefilter: [0] = count, then cand_cap u32 entries
    int hash;       // [NR][bufnum] u64 (2 dwords each): per-read part of Duplicate::seq2intvector
    // per-read position bit masks (bit j of a mask = predicate at base j of the row), built in
    // the pre-stats pass and bit-scanned by Filter::trimAndCut's resolver; an offset is -1 when
    // the option that needs the mask is off.  [NR][wm_stride] dwords, each mask wm_words long.
    int wm, wm_words, wm_stride;
    int wm_badF;    // window [j, j+wF) has total quality <  thrF   (filter.cpp:116)
    int wm_badR;    // window [j, j+wR) has total quality <  thrR   (filter.cpp:151)
    int wm_badT;    // window [j, j+wT) has total quality <  thrT   (filter.cpp:185)
    int wm_isN;     // base j is 'N'                                (filter.cpp:123,191)
    int adapt;      // [2][ADAPT_WORDS] packed adapter words
    int lut_ov, lut_lowq, lut_cplx;                       // u16 tables, (max_len+1+1)/2 dwords each
    int primes;     // [bufnum*512] (generic hash path only)
    int hp, hp_nq;  // [4][hp_nq][bufnum][dup_npl] byte planes of the primes (DevLuts::dup_planes); see has_hp
    int val4_lut;   // [256] u32: Duplicate's base values (A7 T222 C74 G31, duplicate.cpp:92-109) of the four
                    // bases a packed byte holds, one byte each
    int wl, wl_cap; // work list of (read, quality dword) items the fast Stats path hands to the general one:
                    // [0] = count, then wl_cap u16 entries
    int acc_cyc;    // [4][Cp][N_CLS] u64  (2 dwords each)
    int acc_kmer;   // [4][KMER_BINS] u32
    int acc_qh;     // [4][128][QT_DWORDS] u32: quality table = mBaseQualHistogram counters next to the constants
    int acc_misc;   // MISC_* u32 counters
    int acc_end;    // end of the accumulator region (acc_cyc..acc_end is flushed)
    int total;      // dwords of dynamic LDS
};

// misc counter indices (u32 each) inside acc_misc
enum {
    MISC_FILTER = 0,            // [32]
    MISC_ADAPTER_READS = 32,
    MISC_ADAPTER_BASES = 33,
    MISC_POLYX_READS = 34,      // [4]
    MISC_POLYX_BASES = 38,      // [4]
    MISC_CORRECTION = 42,       // [64]
    MISC_CORRECTED_READS = 106,
    MISC_MERGED = 107,
    MISC_STAT_READS = 108,      // [4]
    MISC_STAT_LENSUM = 112,     // [4]
    MISC_ISIZE = 116,           // [isize_max+1]
};

// ---- overrepresentation analysis (Stats::statRead stats.cpp:270-288), kernels fq_ovr_* ----
enum { OVR_STEPS = 5, OVR_SEED_STRIDE = 152 };
static const u32 OVR_HASH_MUL = 0x01000193u;   // odd: the rolling hash is a polynomial mod 2^32
static const u32 OVR_SALT_MUL = 0x9E3779B1u;   // key = hash ^ (length * OVR_SALT_MUL)

struct OvrMate {                 // one mate's seed set (Options::overRepSeqs1 / overRepSeqs2)
    int n_seeds;
    int eval_len;                // Stats::mEvaluatedSeqLen
    int steps[OVR_STEPS];        // {10, 20, 40, 100, min(150, eval_len - 2)}
    u32 pw[OVR_STEPS];           // OVR_HASH_MUL ^ (step - 1)
    const u32* table;            // open addressing, pairs {key, seed index + 1}; index 0 = empty
    u32 table_mask;              // slots - 1
    const u8* seed_sym;          // [n_seeds][OVR_SEED_STRIDE] symbols 0..4 (A T C G N); any other byte stands for itself
    const int* seed_len;
};

struct OvrArgs {
    int n, paired, dedup, sampling;
    u32 pre_mod;                 // (reads this Stats object saw before this launch) % sampling
    int sw_g, qw_g;              // batch row strides in dwords
    const u32* seq[2];
    const u32* qual[2];
    const u16* len[2];
    const u32* res[2];           // result records of this launch (3 dwords each)
    u32* blocksum;               // [ceil(n/256)] units of the block that reach the post-filtering Stats
    u32* blockbase;              // [ceil(n/256)] (units that reached them before the block) % sampling
    u64* post_seen;              // running total across launches
    u32* tasks;                  // [(unit << 2) | mate << 1 | post]
    u32* n_tasks;
    int task_cap;
    OvrMate mate[2];
    int64_t* ctr;
    int64_t o_count[4], o_dist[4];   // fastp_gpu_counter_layout::overrep_count / overrep_dist
    // mOverRepSeqDist of a launch as a DIFFERENCE array (round 5): a hit of a seed at `at` covers positions [at, at + L) -
    // +1 at its start, -1 behind its end, two atomics instead of up to 150; ovr_dist_body takes the running sums into the
    // int64 block and clears the array.  [slot][n_seeds of the slot's mate][eval_len + 1] i32; nullptr: straight into the block
    int* dist_diff[4];
    // LDS staging of the counting kernel: the task's symbols ([position][lane] bytes, sym_cap positions) and, when
    // they fit, the seed hash tables (table_lds[m] = dword offset in LDS, or -1: probe the global copy)
    int sym_cap;
    int table_lds[2];
    // merge mode (peprocessor.cpp:518-561): every post-filtering read goes to the read-1 Stats - the merged read,
    // or with --include_unmerged r1 then r2 of a pair that did not overlap
    int merge, merge_include_unmerged;
    const u32* pair;             // pair records of this launch (2 dwords each)
    // --correction: the post-filtering Stats see the corrected bases.  The batch's correction list is threaded
    // into one chain per read (corr_head[read key] -> entry index + 1, corr_next[entry]) before the counting.
    const u32* corr;             // fastp_gpu_correction entries, 2 dwords each (nullptr: no correction)
    const int32_t* n_corr;
    int corr_cap;
    int first;                   // index of this launch's first unit inside the batch (the entries' read field counts from the batch start)
    u32* corr_head;              // [(paired ? 2 : 1) * n]
    u32* corr_next;              // [corr_cap]
    // units with letters outside ACGTN (fastp_gpu_batch::exotic_*): their symbols come from the raw text - a foreign byte
    // is its own symbol (seeds cut from such reads hold it the same way), its complement follows util.h:16-33
    const int* x_unit;
    int x_n, x_dense;
    const u8* x_text[2];
    const u32* x_off[2];
};

// ---- FASTQ text -> packed rows on the device (fq_parse_* kernels) ----
enum { PARSE_BYTES_PER_LANE = 16, PARSE_BLOCK = 256, PARSE_SUB = 4 };  // a workgroup scans PARSE_SUB consecutive 4 KiB sub-blocks
struct ParseArgs {
    const u8* text;       // 16-byte aligned
    u32 nbytes;           // bytes to scan (a trailing lone '\r' of a non-final chunk is left out)
    int is_last;          // the chunk ends the file: an unterminated last line counts
    u32* blockcount;      // [nblocks] line terminators starting in the block
    u32* blockbase;       // [nblocks] ... before the block
    u32* term_pos;        // [max_lines] offset of the terminator that ends line k
    u8* term_len;         // [max_lines] 1 or 2 ("\r\n")
    u32 max_lines;
    u32* totals;          // [0] terminators, [1] first bad record (atomic min), [2] lines incl. unterminated tail,
                          // [3] records, [4] bytes consumed, [5] longest sequence line, [6] records with letters outside ACGTN
    u32* exotic_list;     // [exotic_cap] those records (unordered), for the text kernel (fq_text.h)
    u32 exotic_cap;
    // packing
    int max_len, sw_g, qw_g, max_records;
    u32* seq_out;
    u32* qual_out;
    u16* len_out;
    u32* line_off;
    u32* line_len;
};

// ---- result records -> output FASTQ text on the device (fq_fmt_* kernels) ----
enum { FMT_BLOCK = 256 };
enum { OVR_BLOCK = 128 };  // threads of the overrepresentation counting kernel
// a task (one sampled read) takes OVR_STEPS lanes, one per window length; OVR_TPB tasks per workgroup share its LDS
// symbol rows ([position][OVR_SYM_STRIDE] bytes)
enum { OVR_TPB = OVR_BLOCK / 5, OVR_SYM_STRIDE = 32 };
struct FmtMate {
    const u8* text;
    const u32* line_off;
    const u32* line_len;
    const u32* res;      // 3 dwords per record
    u8* out;
    u64 out_cap;
    u64* unit_off;       // [n] offset of the unit's record in `out`
};
struct FmtArgs {
    int n, paired, dedup;
    FmtMate m[2];
    u64* blocksum;       // [2][nblocks] bytes the block's units add to each stream
    u64* blockbase;      // [2][nblocks]
    u64* totals;         // [2]
    int nblocks;
    const u32* corrections;  // fastp_gpu_correction, 2 dwords each
    const int* n_corrections;
    int corr_first;      // unit index the correction list's `read` field is relative to
};

// ---- every output stream on the device (fq_fmts_* kernels, fastp_gpu_format_streams) ----
enum { FMTS_STREAMS = 6 };
constexpr u32 FMTS_NONE = 0xFFFFFFFFu;
struct FmtsMate {
    u8* text;                // corrections are patched in
    const u32* line_off;
    const u32* line_len;
    const u32* res;          // 3 dwords per record
};
struct FmtsArgs {
    int n, paired, dedup, merge, merge_include_unmerged;
    int want_failed, want_u1, want_u2;
    int umi_loc, umi_len;
    u32 delim_len, prefix_len;
    u8 delim[8], prefix[32];
    FmtsMate m[2];
    const u32* pair;         // 2 dwords per pair record
    u8* out[FMTS_STREAMS];
    u64 out_cap[FMTS_STREAMS];
    u64* blocksum;           // [FMTS_STREAMS][nblocks] bytes the block's units add to each stream
    u64* blockbase;          // [FMTS_STREAMS][nblocks]
    u64* totals;             // [FMTS_STREAMS]
    int nblocks;
    const u32* corrections;  // fastp_gpu_correction, 2 dwords each
    const int* n_corrections;
    int corr_first;
};

struct KernelArgs {
    DevParams p;
    DevLuts lut;
    LdsLayout L;
    u32 magic_sw, magic_qwg;   // ceil(2^32 / L.SW), ceil(2^32 / p.qw_g) for exact small divisions
    u32 magic_swg;             // ceil(2^32 / p.sw_g)
    int prefetch;              // a tile's input fits the per-lane prefetch registers (TileRegs) and the
                               // batch buffers are 16-byte aligned (vector loads)
    // batch (device pointers)
    int n;
    int first;          // index of this launch's first read/pair inside the submitted batch
    u32 batch_flags;
    const u32* seq[2];
    const u32* qual[2];
    const u16* len[2];
    // results
    u32* res[2];        // fastp_gpu_read_result, 3 dwords each
    u32* pair;          // fastp_gpu_pair_result, 2 dwords each
    u32* corrections;   // fastp_gpu_correction, 2 dwords each
    int corr_capacity;
    int* n_corrections;
    // -c on the lane plan: the engine's own list of the launch's corrections (same entries; sized for the mismatch limit of
    // every pair, so it never overflows) - what the Stats fix-up and the overrepresentation analysis read, whether or not the
    // caller asked for the list
    u32* corr_int;
    int corr_int_cap;
    int* n_corr_int;
    u32* adapter_events;   // fastp_gpu_adapter_event, 3 dwords each
    int adapter_events_capacity;
    int* n_adapter_events;
    const u8* xskip;    // lane plan, units with letters outside ACGTN: [n] 1 = the text kernel writes this unit's records and hash
                        // values (it runs BESIDE this kernel: nothing of such a unit may be written here, only counted)
    u64* dup_pos;       // [n][bufnum] hash values (Duplicate::seq2intvector), for the dup kernels
    // the claim step of Duplicate inside this kernel (dup_claim_issue; plain stream mode, at most two buffers):
    u8* claim_won;      // [n] or null: mask of the buffers whose bloom bit this unit found clear and set
    u32* dup_bitmap;    // [bufnum][dup_bits / 32]
    u64 dup_bits;       // mBufLenInBits
    const u8* dupflag;  // [n] --dedup: the duplicate decision, taken by the dup kernels BEFORE this launch
    u64* phase_cycles;  // optional [16]: cycles per phase summed over workgroups (debug)
    int half_skew;      // half 1 starts this many ~3 us sleeps late, so the halves sit in different phases
    int half_naps;      // poll interval class of the half barrier (0: 64 cycles ... 3: 4096)
    u32 debug_skip;     // profiling only (FASTP_GPU_DEBUG_SKIP): phases left out, results are then meaningless.
                        // 1 masks+rc, 2 hash, 4 overlap, 8 metrics, 16 stats, 32 trim/decide/filter
    // per-workgroup counter slabs: [gridDim][slab_dwords]
    u32* slabs;
    int slab_dwords;
    int tiles;          // ceil(n / P)
    // split plan (fq_stats.h): Stats::statRead runs in its own kernel afterwards.  This kernel then has no per-cycle /
    // k-mer / histogram accumulators (their LDS regions are empty, the slab holds the MISC_* counters only), counts the
    // reads and lengths of the four Stats objects itself, and leaves every read's original and kept length in swin_out.
    int split;
    u32* swin_out[2];   // [n] per mate: rlen0 | kept length << 16
};

// argument block of the fused kernel: one KernelArgs per half-workgroup (identical but for the LDS layout)
struct FusedArgs {
    KernelArgs h[2];
};

}  // namespace fq
