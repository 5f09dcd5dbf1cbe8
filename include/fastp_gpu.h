/* fastp_gpu.h - C ABI of the MI355X-native per-read engine that drops in for the
 * body of fastp's worker loop.
 *
 * What it replaces (reference = OpenGene/fastp v1.3.6, paths relative to the
 * reference root):
 *   bool PairEndProcessor::processPairEnd(ReadPack*, ReadPack*, ThreadConfig*)
 *        src/peprocessor.h:33, src/peprocessor.cpp:362-708
 *   bool SingleEndProcessor::processSingleEnd(ReadPack*, ThreadConfig*)
 *        src/seprocessor.h:32, src/seprocessor.cpp:197-325
 * i.e. everything the loop body computes per read / per pair:
 *   Stats::statRead (stats.cpp:191-291), Duplicate::checkRead/checkPair
 *   (duplicate.cpp:111-163), Filter::trimAndCut (filter.cpp:68-207),
 *   PolyX::trimPolyG/trimPolyX (polyx.cpp:16-116), OverlapAnalysis::analyze
 *   (overlapanalysis.cpp:17-146), BaseCorrector (basecorrector.cpp:16-83),
 *   AdapterTrimmer (adaptertrimmer.cpp:17-157) + Matcher (matcher.cpp:10-54),
 *   Filter::passFilter (filter.cpp:15-66), statInsertSize
 *   (peprocessor.cpp:710-723) and the integer side of FilterResult
 *   (filterresult.cpp:28-36,99-107,182-189).
 * The host keeps FASTQ decode/encode, read names, output routing, the
 * adapter-string map and the JSON/HTML reporters.
 *
 * Plain C: pointers and sizes only.  No function here ever calls exit();
 * every failure is a negative return code (reference: error_exit, util.h:270).
 */
#ifndef FASTP_GPU_H
#define FASTP_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FASTP_GPU_ABI_VERSION 4

/* ---- limits ------------------------------------------------------------ */
#define FASTP_GPU_MAX_READ_LEN 512    /* padded read length the kernels tile in LDS */
#define FASTP_GPU_MAX_ADAPTER_LEN 256 /* -a / --adapter_sequence_r2 / --adapter_fasta entries (detected ones are <= 60) */

/* ---- error codes ------------------------------------------------------- */
#define FASTP_GPU_OK 0
#define FASTP_GPU_E_INVALID (-1)     /* bad argument / inconsistent params       */
#define FASTP_GPU_E_NO_DEVICE (-2)   /* no HIP device / kernels not loadable     */
#define FASTP_GPU_E_HIP (-3)         /* a HIP runtime call failed                */
#define FASTP_GPU_E_ALPHABET (-4)    /* base outside {A,C,G,T,N} or quality outside '!'..'~' in pack */
#define FASTP_GPU_E_TOO_LONG (-5)    /* read longer than params.max_len          */
#define FASTP_GPU_E_UNSUPPORTED (-6) /* option outside the device path's scope   */
#define FASTP_GPU_E_OVERFLOW (-7)    /* correction / adapter-event list capacity exceeded */
#define FASTP_GPU_E_NOMEM (-8)

/* ---- filter result codes (src/common.h:43-51) --------------------------- */
#define FASTP_PASS_FILTER 0
#define FASTP_FAIL_POLY_X 4
#define FASTP_FAIL_OVERLAP 8
#define FASTP_FAIL_N_BASE 12
#define FASTP_FAIL_LENGTH 16
#define FASTP_FAIL_TOO_LONG 17
#define FASTP_FAIL_QUALITY 20
#define FASTP_FAIL_COMPLEXITY 24
#define FASTP_FAIL_ADAPTER_DIMER 28
#define FASTP_FILTER_RESULT_TYPES 32

/* ---- parameter block ----------------------------------------------------
 * Flattened copy of the hot-path subset of `Options` (src/options.h); values
 * are the ones main.cpp:176-427 + Options::validate (options.cpp:85-446)
 * derive, i.e. AFTER defaulting (front2 follows front1, merge forces
 * correction, ...).  Field comments give the Options member.
 */
typedef struct fastp_gpu_params {
    int32_t abi_version; /* FASTP_GPU_ABI_VERSION */
    int32_t paired;      /* Options::isPaired()                                  */
    int32_t max_len;     /* longest read any batch may carry (cycles capacity)   */

    /* TrimmingOptions options.h:223-246 */
    int32_t trim_front1, trim_tail1, trim_front2, trim_tail2;
    int32_t max_len1, max_len2;

    /* QualityCutOptions options.h:132-170 */
    int32_t cut_front, cut_tail, cut_right;
    int32_t cut_front_window, cut_front_quality;
    int32_t cut_tail_window, cut_tail_quality;
    int32_t cut_right_window, cut_right_quality;

    /* PolyGTrimmerOptions / PolyXTrimmerOptions options.h:82-102 */
    int32_t poly_g, poly_g_min_len;
    int32_t poly_x, poly_x_min_len;

    /* AdapterOptions options.h:197-221.  adapter_seq_rN == NULL or "" means
     * hasSeqRN == false.  Sequences must be over {A,C,G,T} (options.cpp:369-399). */
    int32_t adapter_enabled;
    int32_t allow_gap_overlap_trimming;
    int32_t dimer_max_len;
    const char* adapter_seq_r1;
    const char* adapter_seq_r2;

    /* CorrectionOptions / MergeOptions, overlap knobs options.h:376-379 */
    int32_t correction;
    int32_t merge, merge_include_unmerged;
    int32_t overlap_require, overlap_diff_limit, overlap_diff_percent_limit;

    /* QualityFilteringOptions options.h:248-268 (qualified_qual is the PHRED
     * number, e.g. 15; the engine applies num2qual itself, util.h:260) */
    int32_t qual_filter;
    int32_t qualified_qual, unqualified_percent_limit, n_base_limit, avg_qual_req;

    /* ReadLengthFilteringOptions / LowComplexityFilterOptions */
    int32_t length_filter, length_required, length_limit;
    int32_t complexity_filter;
    double complexity_threshold; /* 0..1, as Options stores it (main.cpp:342) */

    /* DuplicationOptions options.h:32-45 */
    int32_t dup_enabled, dedup, dup_accuracy_level;

    /* insertSizeMax options.cpp:23 (histogram has insert_size_max+1 bins) */
    int32_t insert_size_max;

    /* UMI taken from the read itself (umiprocessor.cpp:19-49): number of UMI
     * bases per mate (0 = that mate carries none) and the skip after it.  The
     * name edit stays on the host; the engine only reproduces
     * Read::trimFront (read.cpp:69-73). */
    int32_t umi_len1, umi_len2, umi_skip;

    /* --adapter_fasta (AdapterOptions::seqsInFasta, options.cpp:50-83): the sequences in the
     * order Options::loadFastaAdapters leaves them (sorted by contig name, >= 6 bp, duplicates
     * removed).  A..T only, each at most FASTP_GPU_MAX_ADAPTER_LEN long.  Applied to both mates
     * after the other adapter trimming (AdapterTrimmer::trimByMultiSequences,
     * adaptertrimmer.cpp:48-62; call sites peprocessor.cpp:467-470, seprocessor.cpp:249-251). */
    int32_t n_adapter_fasta;
    const char* const* adapter_fasta;

    /* OverrepresentedSequenceAnasysOptions (-p / -P, options.h:363-365; Stats::statRead
     * stats.cpp:270-288).  The seed sequences are what Evaluator::computeOverRepSeq left in
     * Options::overRepSeqs1/2 (host logic), in std::map order; eval_seq_lenN = Options::seqLenN
     * (Evaluator::computeSeqLen).  Letters A,C,G,T,N.  Not available together with merge or
     * correction (FASTP_GPU_E_UNSUPPORTED). */
    int32_t overrep_enabled, overrep_sampling;
    int32_t eval_seq_len1, eval_seq_len2;
    int32_t n_overrep_seqs1, n_overrep_seqs2;
    const char* const* overrep_seqs1;
    const char* const* overrep_seqs2;

    /* --overlapped_out given (paired input; Options::overlappedOut, peprocessor.cpp:488-495): the engine runs the
     * third OverlapAnalysis::analyze (diffPercentLimit 0, on the pair right after adapter trimming) and reports the
     * overlapped part of read 1 in the records' `reserved` fields.  Together with merge the fields keep THESE values and
     * the merged part lengths follow from the pair record (merge's own analysis, peprocessor.cpp:523): len1 = ov_len +
     * max(0, ov_offset), len2 = ov_offset > 0 ? len(read 2) - ov_len : 0 (overlapanalysis.cpp:152-156). */
    int32_t overlapped_out;
    int32_t reserved[3];
} fastp_gpu_params;

/* fills *p with the values an un-flagged `fastp -i R1 [-I R2]` run uses
 * (main.cpp defaults). */
void fastp_gpu_default_params(fastp_gpu_params* p, int paired, int max_len);

/* ---- batch layout in memory (host or device) ----------------------------
 * SoA, one row per read, rows padded to a fixed stride:
 *   seq : 2 bits/base, base j of a read in bits [2*(j%4), 2*(j%4)+1] of byte
 *         j/4 of its row; code A=0 T=1 C=2 G=3 - fastp's own k-mer code
 *         (stats.cpp:294-311), so complement == code^1 ('N' is stored as code 0
 *         and flagged in qual).  Row stride = fastp_gpu_seq_stride(max_len).
 *   qual: 1 byte/base, bits 0..6 = the phred33 ASCII character (33..126),
 *         bit 7 = 1 iff the base is 'N'.  Row stride = fastp_gpu_qual_stride.
 *   len : uint16 read length (0..max_len).
 * Bytes past a read's length inside its row must be zero (the packer does it).
 */
size_t fastp_gpu_seq_stride(int max_len);  /* bytes: ceil(max_len/4) rounded up to 8 */
size_t fastp_gpu_qual_stride(int max_len); /* bytes: max_len rounded up to 8          */

typedef struct fastp_gpu_batch {
    int32_t n;                  /* reads (SE) or pairs (PE) in this batch         */
    uint32_t flags;             /* FASTP_GPU_BATCH_*                              */
    const uint8_t* seq1;        /* n * seq_stride                                 */
    const uint8_t* qual1;       /* n * qual_stride                                */
    const uint16_t* len1;       /* n                                              */
    const uint8_t* seq2;        /* PE only                                        */
    const uint8_t* qual2;
    const uint16_t* len2;
    /* ---- units with letters outside ACGTN ("exotic" units; ABI v4) --------------------------------
     * The packed rows cannot carry such a byte, and the reference treats it differently in nearly every
     * consumer: `base & 7` bins in Stats::statRead (stats.cpp:206-222), hash value 13 (duplicate.cpp:92-109),
     * a/c/g/t complement to T/G/C/A and everything else to N (util.h:16-33), raw-byte compares in the overlap
     * analysis, adapter matching, polyG / polyX and the complexity filter, only a literal 'N' counts as N.
     * The packer (fastp_gpu_pack_reads_x / fastp_gpu_parse_fastq) stores such a letter as code 0 WITHOUT the N
     * flag and names the unit; the engine runs the listed units - and a few of their neighbours - through a
     * kernel that works on the text itself.  All zero / NULL: no such unit (what a zeroed struct says). */
    int32_t n_exotic;               /* units listed in exotic_unit                                        */
    int32_t exotic_dense;           /* 1: exotic_off[m] is fastp_gpu_parse_fastq's line_off table of mate m
                                     * (entry [4 * unit + 1] = the sequence line); 0: one entry per listed unit */
    const int32_t* exotic_unit;     /* HOST memory (always): n_exotic ascending unit indexes              */
    const uint8_t* exotic_text[2];  /* same memory space as seq1: text holding the listed units' sequence bytes of mate m */
    const uint32_t* exotic_off[2];  /* same memory space: byte offset of each listed unit's sequence in exotic_text[m]     */
    int64_t exotic_text_bytes[2];   /* host submits: size of exotic_text[m] (it is copied to the device)  */
} fastp_gpu_batch;

/* the reference only samples insert size on worker thread 0
 * (peprocessor.cpp:449,497); the host sets this flag on the packs that thread
 * 0 would have received.  With -w 1 (the parity configuration) every pack. */
#define FASTP_GPU_BATCH_STAT_ISIZE 1u
/* leave Stats::statRead's overrepresentation analysis (stats.cpp:270-288) of this batch to a later
 * fastp_gpu_overrep_device call - a sharded run needs the stream positions of the preceding
 * shards first (see "Sharded runs" below) */
#define FASTP_GPU_BATCH_DEFER_OVERREP 2u

/* ---- per-read / per-pair results ---------------------------------------- */
typedef struct fastp_gpu_read_result {
    uint16_t front;      /* bases removed at the 5' end of the ORIGINAL read      */
    uint16_t len;        /* final length: output seq = orig[front, front+len)     */
    uint8_t code;        /* passFilter code after the dimer override (common.h)   */
    uint8_t flags;       /* FASTP_GPU_RF_*                                        */
    int16_t adapter_pos; /* where the adapter starts, in coordinates of the read  */
                         /* as it was just before adapter trimming (may be < 0    */
                         /* for trimBySequence, adaptertrimmer.cpp:138-145)       */
    uint16_t adapter_len;/* length of the string handed to addAdapterTrimmed:     */
                         /* pos>=0: read[pos, pos+adapter_len) (after correction); */
                         /* pos<0 : adapterseq.substr(0, adapter_len)              */
    uint16_t reserved;   /* merge mode, merged pair: bases of this mate in the merged read (unless:)      */
                         /* overlapped_out: read 1 = FASTP_GPU_OVOUT_HIT | pos, read 2 = count: the extra  */
                         /* stream gets orig_r1[front + pos, front + pos + count) - the bases BEHIND the   */
                         /* overlapped region, pos = max(0, offset) + overlap_len, as peprocessor.cpp:491  */
                         /* builds it with std::string's (str, pos) constructor                           */
} fastp_gpu_read_result; /* 12 bytes */
#define FASTP_GPU_OVOUT_HIT 0x8000u

#define FASTP_GPU_RF_NULL 0x01        /* trimAndCut returned NULL (filter.cpp:68)      */
#define FASTP_GPU_RF_DUP 0x02         /* Duplicate::check* said duplicate               */
#define FASTP_GPU_RF_ADAPTER 0x04     /* trimmedN == true (peprocessor.cpp:472-475)     */
#define FASTP_GPU_RF_ADAPTER_OV 0x08  /* ... by trimByOverlapAnalysis                   */
#define FASTP_GPU_RF_CORRECTED 0x10   /* BaseCorrector edited this mate                 */
#define FASTP_GPU_RF_MERGED 0x20      /* pair was merged (merge mode)                   */
#define FASTP_GPU_RF_POLYX 0x40       /* trimPolyX cut this read                        */

typedef struct fastp_gpu_pair_result {
    int16_t ov_offset;   /* OverlapResult (overlapanalysis.h:15-22) used for       */
    uint16_t ov_len;     /* isize/correction/adapter trimming; in merge mode the   */
    uint16_t ov_diff;    /* recomputed one (peprocessor.cpp:523).  ov_len==0 and   */
                         /* ov_offset==0 when not overlapped                       */
    uint16_t flags;      /* bit0 overlapped, bit1 hasGap, bit2 isize evaluated     */
} fastp_gpu_pair_result; /* 8 bytes */

#define FASTP_GPU_PF_OVERLAPPED 0x1
#define FASTP_GPU_PF_HAS_GAP 0x2
#define FASTP_GPU_PF_ISIZE 0x4

/* one base edited by BaseCorrector (basecorrector.cpp:39-57) */
typedef struct fastp_gpu_correction {
    uint32_t read;  /* 2*pair + (0 for R1, 1 for R2), index within the batch */
    uint16_t pos;   /* position in the ORIGINAL read                          */
    uint8_t base;   /* new base, ASCII                                        */
    uint8_t qual;   /* new quality, ASCII                                     */
} fastp_gpu_correction; /* 8 bytes */

/* one trim by an --adapter_fasta sequence (a read can be cut by several of them in turn; the
 * host replays FilterResult::addAdapterTrimmed for each, per read in `adapter` order) */
typedef struct fastp_gpu_adapter_event {
    uint32_t read;      /* 2*unit + (0 for R1, 1 for R2) for PE, the read index for SE        */
    int16_t pos;        /* as fastp_gpu_read_result::adapter_pos                              */
    uint16_t len;       /* as fastp_gpu_read_result::adapter_len                              */
    uint16_t adapter;   /* index into params.adapter_fasta                                    */
    uint16_t reserved;
} fastp_gpu_adapter_event; /* 12 bytes */

typedef struct fastp_gpu_results {
    fastp_gpu_read_result* r1;      /* n                                       */
    fastp_gpu_read_result* r2;      /* n (PE)                                  */
    fastp_gpu_pair_result* pair;    /* n (PE)                                  */
    fastp_gpu_correction* corrections; /* capacity entries, may be NULL        */
    int32_t corrections_capacity;
    int32_t* n_corrections;         /* out: number written                     */
    fastp_gpu_adapter_event* adapter_events; /* capacity entries; needed with adapter_fasta */
    int32_t adapter_events_capacity;
    int32_t* n_adapter_events;      /* out: number written (unordered)         */
} fastp_gpu_results;

/* ---- counter block -------------------------------------------------------
 * All int64.  Mirrors Stats (stats.h:75-99), FilterResult (filterresult.h) and
 * the insert-size histogram so the host can load it back into those objects
 * and let the reference's own JsonReporter/HtmlReporter run unchanged.
 * The per-cycle part of each Stats slot has the exact layout of
 * Stats::mCycleBuffer with bufLen = cycles (stats.cpp:54-63):
 *   [Q30[8] | Q20[8] | Content[8] | Qual[8] | TotalBase | TotalQual] x cycles
 */
enum {
    FASTP_GPU_STATS_PRE1 = 0,
    FASTP_GPU_STATS_POST1 = 1,
    FASTP_GPU_STATS_PRE2 = 2,
    FASTP_GPU_STATS_POST2 = 3
};

typedef struct fastp_gpu_counter_layout {
    int64_t total;            /* number of int64 in the block                   */
    int64_t cycles;           /* = fastp_gpu_cycles_for(params)                 */
    int64_t filter_stats;     /* [32]   FilterResult::mFilterReadStats          */
    int64_t adapter_reads;    /* [1]    mTrimmedAdapterRead                     */
    int64_t adapter_bases;    /* [1]    mTrimmedAdapterBases                    */
    int64_t polyx_reads;      /* [4]    mTrimmedPolyXReads (A,T,C,G)            */
    int64_t polyx_bases;      /* [4]                                            */
    int64_t correction;       /* [64]   mCorrectionMatrix                       */
    int64_t corrected_reads;  /* [1]                                            */
    int64_t merged_pairs;     /* [1]                                            */
    int64_t dup_total;        /* [1]    Duplicate::mTotalReads                  */
    int64_t dup_count;        /* [1]    Duplicate::mDupReads                    */
    int64_t isize;            /* [insert_size_max+1]                            */
    int64_t stats[4];         /* base offset of each Stats slot                 */
    /* offsets inside a Stats slot */
    int64_t st_reads;         /* [1]    mReads                                  */
    int64_t st_length_sum;    /* [1]    mLengthSum                              */
    int64_t st_qual_hist;     /* [128]  mBaseQualHistogram                      */
    int64_t st_kmer;          /* [1024] mKmer (the used half, stats.cpp:45)     */
    int64_t st_cycle;         /* [34*cycles]                                    */
    int64_t st_size;
    /* overrepresentation analysis (absolute offsets, one pair per Stats slot):
     * mOverRepSeq counts [n_overrep[slot]] and mOverRepSeqDist [n_overrep[slot]][eval_len[slot]]
     * in the seed order of the parameter block; all zero-sized without overrep_enabled */
    int64_t n_overrep[4];
    int64_t eval_len[4];
    int64_t overrep_count[4];
    int64_t overrep_dist[4];
} fastp_gpu_counter_layout;

/* per-cycle capacity of the Stats slots: max_len, or 2*max_len in merge mode
 * (a merged read can be as long as both mates; the reference grows its buffers
 * on demand, Stats::extendBuffer stats.cpp:65-83) */
int fastp_gpu_cycles_for(const fastp_gpu_params* params);
void fastp_gpu_counter_layout_for(int cycles, int insert_size_max, fastp_gpu_counter_layout* out);
/* the layout an engine created from `params` uses (adds the overrepresentation arrays) */
void fastp_gpu_counter_layout_for_params(const fastp_gpu_params* params, fastp_gpu_counter_layout* out);

/* ---- engine -------------------------------------------------------------- */
typedef struct fastp_gpu_ctx fastp_gpu_ctx;

/* device = HIP device ordinal.  Allocates the duplicate bitmaps, the counter
 * block and the per-workgroup counter slabs in HBM. */
int fastp_gpu_create(const fastp_gpu_params* params, int device, fastp_gpu_ctx** out);
void fastp_gpu_destroy(fastp_gpu_ctx* ctx);
const char* fastp_gpu_last_error(const fastp_gpu_ctx* ctx); /* ctx may be NULL */

/* ASCII -> packed SoA rows (host code; the repack a patched worker loop does
 * on its ReadPack before submit).  seqs[i]/quals[i] need not be 0-terminated.
 * Returns FASTP_GPU_E_ALPHABET (and the index in *bad_read if non-NULL) when a
 * base outside {A,C,G,T,N} or a quality character outside '!'..'~' (33..126) is met (letters:
 * see fastp_gpu_pack_reads_x, which lists such reads instead), FASTP_GPU_E_TOO_LONG when lens[i] > max_len. */
int fastp_gpu_pack_reads(int max_len, int n, const char* const* seqs, const char* const* quals,
                         const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out,
                         uint16_t* len_out, int32_t* bad_read);
/* The same, but a letter outside {A,C,G,T,N} is not an error: the read's entry of exotic[n] (caller-zeroed) is set
 * to 1, the letter is stored as code 0 without the N flag, and the caller lists the unit in the batch's exotic_* fields
 * (fastp_gpu_batch) with its raw sequence bytes.  Quality characters outside '!'..'~' stay an error. */
int fastp_gpu_pack_reads_x(int max_len, int n, const char* const* seqs, const char* const* quals,
                           const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out,
                           uint16_t* len_out, int32_t* bad_read, uint8_t* exotic);

/* ---- FASTQ text -> packed batch ON THE DEVICE -----------------------
 * The step before the path: FastqReader::getLine / read (src/fastqreader.cpp:240-368) + the
 * packer above, for a chunk of plain FASTQ text resident in HBM.  A line ends at the first '\r' or
 * '\n'; "\r\n" is one terminator (:246-262).  Records are four consecutive lines; the chunk must
 * be well formed the way FastqReader::read expects its input - name lines start with '@' and are
 * not empty, third lines start with '+', sequence and quality have equal length (:338-362) -
 * otherwise FASTP_GPU_E_INVALID is returned with info->first_bad set and the host falls back to its
 * own reader for that chunk (the reference's tolerant resynchronisation on '@' is host logic).
 * Only complete records are produced; info->consumed tells where the next chunk must start.
 * All pointers except `info` are DEVICE pointers; synchronous. */
#define FASTP_GPU_PARSE_BAD_MALFORMED 1 /* what FastqReader::read refuses (:338-362): the reference stops reading there      */
#define FASTP_GPU_PARSE_BAD_TOO_LONG 2  /* well formed, but longer than the context's max_len: re-plan with max_seq_len     */
#define FASTP_GPU_PARSE_BAD_ALPHABET 3  /* well formed, a quality character outside '!'..'~' (a letter outside ACGTN is no error: n_exotic) */
typedef struct fastp_gpu_parse_info {
    int32_t n_records;   /* records packed                                              */
    int32_t first_bad;   /* index of the first malformed / over-long / quality outside '!'..'~' record, or -1           */
    int64_t consumed;    /* bytes of text the records cover (offset of the next record)  */
    int64_t n_lines;     /* line terminators seen (+1 for an unterminated last line)     */
    int32_t bad_kind;    /* FASTP_GPU_PARSE_BAD_* of record first_bad, 0 when there is none (ABI v3)                */
    int32_t max_seq_len; /* the longest sequence line among the well-formed records looked at (ABI v3)              */
    int32_t n_exotic;    /* records with a letter outside ACGTN (ABI v4): not an error - fastp_gpu_parse_exotic lists  */
    int32_t reserved;    /* them, and the batch built from this chunk names them in its exotic_* fields              */
} fastp_gpu_parse_info;

int fastp_gpu_parse_fastq(fastp_gpu_ctx* ctx, const uint8_t* text, int64_t nbytes, int is_last_chunk,
                          int32_t max_records,
                          uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out, /* packed rows, strides as above */
                          uint32_t* line_off,  /* [4*max_records] offset of each record line in `text`      */
                          uint32_t* line_len,  /* [4*max_records] its length without the terminator        */
                          fastp_gpu_parse_info* info);
/* --overlapped_out (src/peprocessor.cpp:488-495) makes a seventh stream that the device formatter does not write: by
 * default fastp_gpu_format_streams refuses a context with that option (FASTP_GPU_E_UNSUPPORTED) so that nobody loses the
 * stream unnoticed.  A caller that assembles it itself from the records (read 1's `reserved` = FASTP_GPU_OVOUT_HIT | first
 * printed position, read 2's = the number of printed bases; what fastp_gpu_host.h and fastp_gpu_stream.h do) says so
 * here; the other six streams are then formatted as usual (merged part lengths come from the pair records). */
int fastp_gpu_host_writes_overlapped(fastp_gpu_ctx* ctx, int on);

/* --phred64 (Read::convertPhred64To33 src/read.cpp, applied by FastqReader::read src/fastqreader.cpp:309-368 to every read of
 * such a run): the quality characters of the n records fastp_gpu_parse_fastq found become max(33, q - 31), in place, in
 * `text` (what the formatter copies and the text kernel reads) and in the packed rows `qual_rows` (the N marks stay).
 * line_off / line_len are the parser's tables.  DEVICE pointers; synchronous. */
int fastp_gpu_phred64_to_33(fastp_gpu_ctx* ctx, int32_t n, uint8_t* text, const uint32_t* line_off, const uint32_t* line_len,
                            uint8_t* qual_rows);

/* The records of the LAST fastp_gpu_parse_fastq call on this context that hold a letter outside ACGTN
 * (info->n_exotic of them): ascending record indexes into units[0, capacity); returns their number.
 * They go into fastp_gpu_batch::exotic_unit (the union of both mates' lists for paired input), with
 * exotic_dense = 1, exotic_text[m] = the chunk and exotic_off[m] = its line_off table. */
int32_t fastp_gpu_parse_exotic(const fastp_gpu_ctx* ctx, int32_t* units, int32_t capacity);

/* ---- BGZF-compressed FASTQ -> text ON THE DEVICE ----------------------
 * The reference reads a bgzip-written .gz with BgzfMtReader (src/bgzf.h:36-239): a reader thread
 * cuts the file into its independent <= 64 KiB gzip members by the BSIZE field of each header
 * (bgzf.h:29-32) and a pool of igzip workers inflates them.  Same split here: the header walk stays
 * on the host (fastp_gpu_bgzf_index: HOST memory in, a few bytes per block out), the inflation runs
 * on the GPU, one lane per block (fastp_gpu_inflate_bgzf).  A plain (single-member) gzip stream has
 * no block structure to exploit and stays on the host, as in the reference (fastqreader.cpp:88-149).
 *
 * fastp_gpu_bgzf_index walks the complete members of bytes[0, nbytes) (at most max_blocks of them,
 * and only while their text fits max_text_bytes): for block k, pay_off/pay_len locate the raw
 * deflate payload inside the chunk, isize/crc come from the member trailer, out_off is the running
 * sum of isize (where its text goes).  info->consumed = bytes of whole members walked (the rest is
 * carried into the next chunk), info->out_bytes their total text size.  FASTP_GPU_E_INVALID when the
 * bytes at a member boundary are not a BGZF header (info->first_bad = that block's index). */
typedef struct fastp_gpu_inflate_info {
    int32_t n_blocks;
    int32_t first_bad;   /* index of the first malformed block, or -1 */
    int64_t consumed;
    int64_t out_bytes;
} fastp_gpu_inflate_info;

int fastp_gpu_bgzf_index(const uint8_t* host_bytes, int64_t nbytes, int32_t max_blocks, int64_t max_text_bytes,
                         uint32_t* pay_off, uint32_t* pay_len, uint32_t* isize, uint32_t* crc, uint64_t* out_off,
                         fastp_gpu_inflate_info* info);

/* Inflate n_blocks indexed blocks of the compressed chunk `comp` (DEVICE memory, readable for 16 bytes
 * past the last member) into `out` (DEVICE).  The five index arrays are DEVICE copies of what
 * fastp_gpu_bgzf_index produced.  Every block's size is checked against its trailer, and its CRC-32
 * when check_crc != 0 (the reference's igzip does both).  first_bad (HOST, may be NULL) receives the
 * first failing block or -1; FASTP_GPU_E_INVALID if any block failed.  Synchronous. */
int fastp_gpu_inflate_bgzf(fastp_gpu_ctx* ctx, const uint8_t* comp, int32_t n_blocks, const uint32_t* pay_off,
                           const uint32_t* pay_len, const uint32_t* isize, const uint32_t* crc, const uint64_t* out_off,
                           uint8_t* out, int64_t out_capacity, int check_crc, int32_t* first_bad);

/* ---- result records -> output FASTQ text ON THE DEVICE ---------------
 * The step after the path for the main output streams: Read::appendToString (src/read.cpp:119-134)
 * for every unit the worker loop routes to out1 [and out2] (peprocessor.cpp:577-591,
 * seprocessor.cpp:280-286): name line, seq[front, front+len), strand line, qual[front, front+len),
 * each followed by '\n', in input order, BaseCorrector edits applied.  Inputs are what
 * fastp_gpu_parse_fastq and fastp_gpu_submit_device left on the device.  This entry point writes
 * the two main streams only and refuses merge mode and umi_len* > 0 (FASTP_GPU_E_UNSUPPORTED):
 * fastp_gpu_format_streams below writes every stream, name edits included.  FASTP_GPU_E_OVERFLOW when an output buffer is too small
 * (out_len still reports the needed sizes).  All pointers except out_len are DEVICE pointers. */
typedef struct fastp_gpu_format_in {
    const uint8_t* text;            /* the FASTQ chunk the records were parsed from      */
    const uint32_t* line_off;       /* [4n] from fastp_gpu_parse_fastq                    */
    const uint32_t* line_len;       /* [4n]                                               */
    const fastp_gpu_read_result* res; /* [n] this mate's result records                   */
} fastp_gpu_format_in;

int fastp_gpu_format_fastq(fastp_gpu_ctx* ctx, int32_t n, const fastp_gpu_format_in* mate1,
                           const fastp_gpu_format_in* mate2 /* NULL for single-end */,
                           const fastp_gpu_correction* corrections, const int32_t* n_corrections /* may be NULL */,
                           uint8_t* out1, int64_t out1_capacity, uint8_t* out2, int64_t out2_capacity,
                           int64_t out_len[2] /* host: bytes written (needed) per stream */);

/* The same for EVERY output stream of the worker loop (src/peprocessor.cpp:518-621, src/seprocessor.cpp:280-290):
 * out1 / out2, --failed_out (Read::appendToStringWithTag src/read.cpp:136-154 with the FAILED_TYPES tag,
 * src/common.h:57-66), --unpaired1 / --unpaired2, the merged stream (OverlapAnalysis::merge's string assembly
 * src/overlapanalysis.cpp:148-179: r1 part + reverse complement of the r2 part, " merged_L1_L2" appended to the
 * name, and to the strand line unless that is "+") and the UMI name edit (UmiProcessor::addUmiToName
 * src/umiprocessor.cpp:62-81).  Routing decisions all come from the result records.  BaseCorrector's edits are
 * patched INTO `text` first (the reference edits its reads in place, so every stream prints the corrected bases):
 * `text` is therefore written to.  Streams the options do not ask for get no bytes (their buffers may be NULL).
 * Output order inside a stream = input order.  FASTP_GPU_E_OVERFLOW + needed sizes in out_len when a buffer is
 * too small.  All pointers except opts / out / out_capacity / out_len themselves are DEVICE pointers. */
enum { FASTP_GPU_OUT1 = 0, FASTP_GPU_OUT2 = 1, FASTP_GPU_FAILED = 2, FASTP_GPU_MERGED = 3,
       FASTP_GPU_UNPAIRED1 = 4, FASTP_GPU_UNPAIRED2 = 5, FASTP_GPU_N_OUTPUTS = 6,
       /* host glue only (fastp_gpu_host.h): --overlapped_out's stream; fastp_gpu_format_streams refuses that option */
       FASTP_GPU_OVERLAPPED = 6, FASTP_GPU_N_HOST_OUTPUTS = 7 };
#define FASTP_GPU_UMI_NONE 0
#define FASTP_GPU_UMI_READ1 1     /* UMI_LOC_READ1    */
#define FASTP_GPU_UMI_READ2 2     /* UMI_LOC_READ2    */
#define FASTP_GPU_UMI_PER_READ 3  /* UMI_LOC_PER_READ */

typedef struct fastp_gpu_format_options {
    int32_t want_failed;      /* --failed_out given   */
    int32_t want_unpaired1;   /* --unpaired1 given    */
    int32_t want_unpaired2;   /* --unpaired2 given (and different from --unpaired1) */
    int32_t umi_loc;          /* FASTP_GPU_UMI_*      */
    int32_t umi_len;
    const char* umi_prefix;   /* host string, may be NULL, at most 32 characters */
    const char* umi_delimiter;/* host string, NULL = ":", at most 8 characters   */
    int32_t corrections_capacity; /* entries the correction list can hold; 0 = the capacity given to the submit that
                                     filled it on this context.  A list whose counter ran past it is not applied:
                                     FASTP_GPU_E_OVERFLOW, nothing is read behind the buffer */
} fastp_gpu_format_options;

typedef struct fastp_gpu_format_io {
    uint8_t* text;                  /* the FASTQ chunk the records were parsed from (corrections are patched in) */
    const uint32_t* line_off;       /* [4n] from fastp_gpu_parse_fastq                    */
    const uint32_t* line_len;       /* [4n]                                               */
    const fastp_gpu_read_result* res; /* [n] this mate's result records                   */
} fastp_gpu_format_io;

int fastp_gpu_format_streams(fastp_gpu_ctx* ctx, int32_t n, const fastp_gpu_format_io* mate1,
                             const fastp_gpu_format_io* mate2 /* NULL for single-end */,
                             const fastp_gpu_pair_result* pair /* [n] paired; NULL for single-end */,
                             const fastp_gpu_correction* corrections, const int32_t* n_corrections /* may be NULL */,
                             const fastp_gpu_format_options* opts /* NULL = main streams only */,
                             uint8_t* const out[FASTP_GPU_N_OUTPUTS], const int64_t out_capacity[FASTP_GPU_N_OUTPUTS],
                             int64_t out_len[FASTP_GPU_N_OUTPUTS] /* host: bytes written (needed) per stream */);

/* ---- device memory for callers that do not link the HIP runtime themselves ----------------
 * The entry points that take DEVICE pointers (submit_device, parse / format / deflate / inflate, eval_*)
 * are meant for hosts that manage HBM with HIP; a host that does not (the reference is plain C++ built
 * with g++) gets the four calls it needs from the engine.  Synchronous; pointers are hipMalloc'ed memory
 * on the context's device. */
int fastp_gpu_device_alloc(fastp_gpu_ctx* ctx, int64_t bytes, void** dev_ptr);
int fastp_gpu_device_free(fastp_gpu_ctx* ctx, void* dev_ptr);
int fastp_gpu_device_upload(fastp_gpu_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);
int fastp_gpu_device_download(fastp_gpu_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);

/* ---- output text -> gzip members ON THE DEVICE (second half) -----------
 * The reference's .gz outputs: every worker compresses its pack as one independent gzip member
 * (libdeflate_gzip_compress, src/writer.cpp:110-133) and the members land at ordered offsets
 * (src/writerthread.cpp:118-168); concatenated members are one valid gzip file (src/common.h:27-30).
 * Here a stream of text (what fastp_gpu_format_streams wrote) is cut into blocks of 65280 bytes and each
 * becomes one gzip member with the BGZF extra field (the block's size): any gzip reads the result, and a BGZF
 * reader - the reference's BgzfMtReader, fastp_gpu_bgzf_index + fastp_gpu_inflate_bgzf - inflates it block by
 * block.  LZ77 (hash of 4 bytes + distance 1) with dynamic Huffman codes, or a stored block when that is
 * smaller; one wavefront per block.  DEFLATE has no canonical output: the bytes differ from libdeflate's, the
 * text they inflate to is identical.  write_eof != 0 appends bgzip's 28-byte end-of-file member (for the last
 * call of a file).  text / out: DEVICE pointers; *out_len (HOST) = bytes written, or needed when
 * FASTP_GPU_E_OVERFLOW is returned.  An upper bound for out_capacity: nbytes + 31 * (nbytes / 65280 + 1) + 28. */
int fastp_gpu_deflate_bgzf(fastp_gpu_ctx* ctx, const uint8_t* text, int64_t nbytes, int write_eof, uint8_t* out,
                           int64_t out_capacity, int64_t* out_len);

/* ---- the Evaluator pre-pass ON THE DEVICE ------------------------------
 * The loops of src/evaluator.cpp that scan a prefix of the input before the workers start, run on packed
 * rows already in HBM (DEVICE pointers seq / qual / len as in fastp_gpu_batch, one mate at a time; n reads
 * are available - the reference's own read / base limits are applied inside).  What the reference does with
 * the numbers afterwards (top-10 seeds, NucleotideTree walks, known-adapter names) stays in its Evaluator.
 *
 * fastp_gpu_eval_seq_len: Evaluator::computeSeqLen (evaluator.cpp:54-76) - the longest of the first 1000 reads.
 *
 * fastp_gpu_eval_adapter_kmers: the 4^10 ten-mer histogram of Evaluator::evalAdapterAndReadNum
 * (evaluator.cpp:377-402) over the reads its loading loop admits (:326-341: at most 256 Ki reads and while
 * fewer than 151 * 256 Ki bases were loaded): positions 20 .. len - 10 - max(1, trim_tail1), keys by
 * Evaluator::seq2int (A=0 T=1 C=2 G=3, first base most significant), windows with an N skipped,
 * counts[0] ("AAAAAAAAAA") = 0.  counts: DEVICE uint32[1 << 20], overwritten.  *records (HOST) = reads used;
 * the reference does not evaluate below 10000 of them (:357), the caller applies that.
 *
 * fastp_gpu_eval_overrep: Evaluator::computeOverRepSeq (evaluator.cpp:78-169) - every substring of lengths
 * 10, 20, 40, 100 and min(150, seq_len - 2) of the reads read while fewer than 151 * 10000 bases were seen,
 * counted exactly, kept when over the threshold of its length class (:115-137), minus the ones contained in a
 * kept longer one with count / count2 < 10 (:140-160).  Result in HOST memory in std::map order (byte order
 * of the text): sequence i is text[off[i], off[i+1]), count[i] its occurrences.  FASTP_GPU_E_OVERFLOW when
 * max_seqs or text_capacity is too small (*n_seqs = the number found). */
int fastp_gpu_eval_seq_len(fastp_gpu_ctx* ctx, const uint16_t* len, int32_t n, int32_t* seq_len);
int fastp_gpu_eval_adapter_kmers(fastp_gpu_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint16_t* len,
                                 int32_t n, int32_t trim_tail1, uint32_t* counts, int64_t* records);
int fastp_gpu_eval_overrep(fastp_gpu_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint16_t* len, int32_t n,
                           int32_t seq_len, char* text, int64_t text_capacity, int64_t* off /* [max_seqs + 1] */,
                           int64_t* count /* [max_seqs] */, int32_t max_seqs, int32_t* n_seqs);

/* Process one batch whose buffers (and result buffers) live in HOST memory:
 * H2D copy, kernels, D2H copy, synchronous. */
int fastp_gpu_submit_host(fastp_gpu_ctx* ctx, const fastp_gpu_batch* batch, fastp_gpu_results* res);

/* Process one batch whose buffers AND result buffers are DEVICE pointers.
 * Asynchronous on `hip_stream` (a hipStream_t passed as void*; NULL = the
 * context's own stream).  res->n_corrections must be a device pointer too. */
int fastp_gpu_submit_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* batch,
                            fastp_gpu_results* res, void* hip_stream);

/* wait for everything submitted on the context's stream */
int fastp_gpu_synchronize(fastp_gpu_ctx* ctx);

/* Device pointer to the reduced counter block (layout above) - the buffer a
 * multi-GPU host all-reduces (RCCL sum, int64) in place of Stats::merge /
 * FilterResult::merge (stats.cpp:877-955, filterresult.cpp:38-89).
 * Folds any pending per-workgroup slabs first. */
int fastp_gpu_counters_device(fastp_gpu_ctx* ctx, int64_t** dev_ptr, int64_t* n, void* hip_stream);

/* copy the counter block to the host (n must equal layout.total) */
int fastp_gpu_counters(fastp_gpu_ctx* ctx, int64_t* out, int64_t n);

/* Multi-GPU merge without aliasing engine memory: export copies the counter block into a
 * caller-owned DEVICE buffer of layout.total int64 (e.g. a torch tensor the host then
 * all-reduces with RCCL); import replaces the engine's block with the merged one (the
 * non-additive header words are restored).  Both are synchronous. */
int fastp_gpu_counters_export(fastp_gpu_ctx* ctx, int64_t* dst_device, int64_t n);
int fastp_gpu_counters_import(fastp_gpu_ctx* ctx, const int64_t* src_device, int64_t n);

/* ---- Sharded runs: one engine per GPU, results identical to ONE stream -------------
 * The input is cut into contiguous shards in input order, one engine per shard.  Everything the
 * worker loop computes is per read except two things that depend on what came EARLIER in the
 * stream, and both have an exact sharded form:
 *
 *  (1) Duplicate::checkPair / checkRead (duplicate.cpp:122-163): a unit is a duplicate iff, in every
 *      bloom buffer, its bit was already set by an earlier unit.  "Earlier" = earlier in this shard,
 *      or anywhere in a preceding shard.  Two passes over the shard's batches with one exchange:
 *        pass 1  fastp_gpu_submit_pass1_device: inserts the units into this engine's bitmaps in input
 *                order and keeps, per unit, the bit positions and the mask of buffers whose bit an
 *                earlier unit of THIS shard had set (`scan_state`: fastp_gpu_dup_scan_bytes(n) bytes
 *                of device memory per batch, kept until pass 2).  Without --dedup nothing else
 *                depends on the decision, so the whole worker loop runs here as in
 *                fastp_gpu_submit_device (records, counters) - only RF_DUP and the duplicate
 *                counters are left open.  With --dedup routing depends on it: pass 1 only hashes
 *                and inserts (res may be NULL).
 *        exchange  the exclusive prefix-OR of the bitmaps in shard order.  Either
 *                all-gather the images (fastp_gpu_dup_bitmap_export) and hand the preceding ones to
 *                fastp_gpu_dup_prefix_set, or - 1/4 of the traffic at 8 shards - all-to-all slices
 *                of the images, fastp_gpu_prefix_or_images on the slice owner, all-to-all back
 *                (fastp_amd/multigpu.py does the latter over RCCL).
 *        pass 2  fastp_gpu_submit_pass2_device: duplicate = AND_i (set earlier in this shard OR set
 *                in the prefix).  Without --dedup the decision is patched into pass 1's records
 *                (RF_DUP) and counted; with --dedup the worker loop runs now, with the decision.
 *
 *  (2) the overrepresentation sampling (stats.cpp:272: every `sampling`-th read a Stats object has
 *      seen): run the worker loop with FASTP_GPU_BATCH_DEFER_OVERREP, exchange the per-shard read
 *      counts of the post-filtering Stats (counter st_reads of slot 1), fastp_gpu_stream_set_origin
 *      with the sums over the preceding shards, then fastp_gpu_overrep_device on the same batches.
 *
 * All pointers are DEVICE pointers; calls are asynchronous on the stream like submit_device unless
 * stated.  fastp_amd/multigpu.py drives this over torch.distributed. */
int64_t fastp_gpu_dup_scan_bytes(const fastp_gpu_ctx* ctx, int32_t n);
int fastp_gpu_submit_pass1_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* batch, void* scan_state,
                                  fastp_gpu_results* res, void* hip_stream);
int fastp_gpu_submit_pass2_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* batch, const void* scan_state,
                                  fastp_gpu_results* res, void* hip_stream);
/* size of one image of the engine's bloom bitmaps (mBufNum * mBufLenInBytes, duplicate.cpp:13-47) */
int64_t fastp_gpu_dup_bitmap_bytes(const fastp_gpu_ctx* ctx);
int fastp_gpu_dup_bitmap_export(fastp_gpu_ctx* ctx, void* dst_device);              /* synchronous */
/* this engine's bitmaps <- an image of the same size (a context that continues another one's stream); synchronous */
int fastp_gpu_dup_bitmap_import(fastp_gpu_ctx* ctx, const void* src_device);
/* OR of `n_images` consecutive images becomes this engine's prefix (0 = no preceding shard); synchronous */
int fastp_gpu_dup_prefix_set(fastp_gpu_ctx* ctx, const void* images_device, int32_t n_images);
/* in place: images[k] <- OR of images[j], j < k (images[0] <- 0), each `bytes_each` long; synchronous */
int fastp_gpu_prefix_or_images(fastp_gpu_ctx* ctx, void* images_device, int32_t n_images, int64_t bytes_each);
/* stream position of the next unit this engine will see: units of the preceding shards, and the
 * reads their post-filtering Stats saw; synchronous */
int fastp_gpu_stream_set_origin(fastp_gpu_ctx* ctx, int64_t units_before, int64_t post_reads_before);
int fastp_gpu_overrep_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* batch, const fastp_gpu_results* res,
                             void* hip_stream);

/* ---- Collectives: RCCL over xGMI behind the C ABI ------------------------------------------------
 * What a C++ host calls where reference fastp merges its worker threads at the end of a run:
 * PairEndProcessor::process src/peprocessor.cpp:217-234 (SingleEndProcessor::process
 * src/seprocessor.cpp:104-116) -> Stats::merge src/stats.cpp:877-955 + FilterResult::merge
 * src/filterresult.cpp:38-89: an int64 sum over per-worker counter arrays = ONE ncclAllReduce(ncclInt64,
 * ncclSum) of the counter block (~0.25 MB: latency bound).  And the exchange step of the sharded duplicate
 * protocol above (the shared Duplicate bitmaps, src/duplicate.h:34-37).
 *
 * One communicator rank per context (= per GPU).  Either one process per GPU - rank 0 calls
 * fastp_gpu_comm_id, the host program hands the 128 bytes to the other processes (MPI, a file, a socket:
 * its own business), every process calls fastp_gpu_comm_init - or one process that owns n contexts on n
 * different GPUs calls fastp_gpu_comm_init_local (rank = position in the array).  Both collectives take the
 * contexts the CALLING PROCESS owns (n = 1 with one process per GPU) and are synchronous; every rank of the
 * communicator must make the same call.  librccl is loaded at the first use (FASTP_GPU_E_UNSUPPORTED if it
 * cannot be); errors of this section are reported by fastp_gpu_comm_last_error (thread local). */
#define FASTP_GPU_COMM_ID_BYTES 128
int fastp_gpu_comm_id(uint8_t id[FASTP_GPU_COMM_ID_BYTES]);
int fastp_gpu_comm_init(fastp_gpu_ctx* ctx, const uint8_t id[FASTP_GPU_COMM_ID_BYTES], int nranks, int rank);
int fastp_gpu_comm_init_local(fastp_gpu_ctx* const* ctxs, int n);
void fastp_gpu_comm_destroy(fastp_gpu_ctx* ctx);   /* also done by fastp_gpu_destroy */
/* Stats::merge / FilterResult::merge: every rank's counter block <- the sum over all ranks */
/* Contract of both collectives: a context's communicator carries ONE collective at a time.  The registry lock is not held
 * across the RCCL calls (two ranks living in one process must be able to meet), the entries are pinned instead:
 * fastp_gpu_comm_destroy / fastp_gpu_destroy of one of the contexts from another thread blocks until the collective in
 * flight has returned (so it must not be called from the thread the other ranks are waiting for). */
int fastp_gpu_allreduce(fastp_gpu_ctx* const* ctxs, int n);
/* between pass 1 and pass 2 of a sharded run: rank r's prefix <- OR of the bitmaps of ranks 0..r-1
 * (slices all-to-all, fastp_gpu_prefix_or_images on the slice owner, all-to-all back, fastp_gpu_dup_prefix_set) */
int fastp_gpu_exchange_dup_prefix(fastp_gpu_ctx* const* ctxs, int n);
const char* fastp_gpu_comm_last_error(void);

/* ---- Pipelined host submits -------------------------------------------------
 * What a patched fastp does instead of calling the worker-loop body once per 1000-read pack
 * (src/peprocessor.cpp:1021-1033 processorTask -> processPairEnd): its worker threads pack their packs, in input
 * order, into one large batch in page-locked memory (fastp_gpu_host_alloc), one thread submits the batch - copies,
 * kernels and result copies are queued on the context's stream and the call returns - and the threads go on packing
 * the next batch while it runs; fastp_gpu_wait / fastp_gpu_poll say when a batch's records have arrived.
 * Batches run in submission order (one stream): Duplicate's and the overrepresentation sampling's stream order is
 * the submission order.  A slot owns the device staging of one batch in flight. */
#define FASTP_GPU_ASYNC_SLOTS 4
int fastp_gpu_host_alloc(fastp_gpu_ctx* ctx, int64_t bytes, void** ptr);   /* page-locked host memory */
int fastp_gpu_host_free(fastp_gpu_ctx* ctx, void* ptr);
/* `b` and the arrays of `res` (r1, r2, pair, corrections, adapter_events) must stay valid and untouched until the slot
 * has been waited for; n_corrections / n_adapter_events are written by fastp_gpu_wait / fastp_gpu_poll. */
int fastp_gpu_submit_host_async(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, fastp_gpu_results* res, int slot);
int fastp_gpu_wait(fastp_gpu_ctx* ctx, int slot);   /* blocks; FASTP_GPU_OK, or the error of the batch (list overflow) */
int fastp_gpu_poll(fastp_gpu_ctx* ctx, int slot);   /* 1 = arrived (slot free), 0 = still running, < 0 = error */

/* Bring the HIP runtime up on `device` (context, this library's code object) ahead of fastp_gpu_create - for a helper
 * thread started as soon as the host program knows a GPU run is coming, so that the runtime's start-up overlaps its own
 * (reference: nothing - the cost has no CPU counterpart). */
int fastp_gpu_warmup(int device);

/* HIP device ordinal the context lives on */
int fastp_gpu_device(const fastp_gpu_ctx* ctx);

/* Start a new run on the same context: counter block, duplicate bitmaps and stream positions as right after
 * fastp_gpu_create - what constructing fresh Stats / FilterResult / Duplicate objects is for the reference
 * (src/peprocessor.cpp:26-60).  Synchronous. */
int fastp_gpu_reset(fastp_gpu_ctx* ctx);

/* time spent inside the kernels of the per-read path (the fused kernel, or - split / lane plan - the per-read kernel
 * plus the Stats kernel) for the launches since the last call, measured with HIP events on the launch stream: total
 * milliseconds and number of launches (used by bench.py for the roofline object). */
int fastp_gpu_kernel_time(fastp_gpu_ctx* ctx, double* total_ms, int64_t* launches);

/* which kernels run the worker loop for this context's options (diagnostic; the results do not depend on it):
 * 0 = fq_fused_kernel (one 1024-lane workgroup per CU, Stats::statRead inside),
 * 1 = fq_scan_kernel + fq_stats_kernel (tile kernel as 256-lane workgroups, streaming Stats kernel),
 * 2 = fq_lane_kernel + fq_stats_kernel (one lane per pair, reads in registers). */
#define FASTP_GPU_PLAN_FUSED 0
#define FASTP_GPU_PLAN_SPLIT 1
#define FASTP_GPU_PLAN_LANE 2
int fastp_gpu_plan(const fastp_gpu_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* FASTP_GPU_H */
