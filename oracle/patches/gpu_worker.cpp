// gpu_worker.cpp - see gpu_worker.h.  TEST INFRASTRUCTURE: the reference-side binding, built into
// oracle/_ref/fastp_ref_gpu only.  It reaches into the reference's classes the way a maintainer's patch would
// (a friend declaration per class); here the access specifiers are lifted for this one translation unit instead,
// so that the reference headers are compiled untouched.
// every standard / system header the reference headers pull in, BEFORE the access specifiers are lifted
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <deque>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cctype>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <time.h>
#include <vector>
#include <zlib.h>
#include <libdeflate.h>
#include "isa-l/igzip_lib.h"
#include "../../fastp_amd/csrc/fq_timeline.h"   // FASTP_GPU_TIMELINE=1: where the start-up goes

#define private public
#define protected public
#include "src/peprocessor.h"
#include "src/seprocessor.h"
#include "src/threadconfig.h"
#include "src/stats.h"
#include "src/filterresult.h"
#include "src/duplicate.h"
#include "src/writerthread.h"
#include "src/options.h"
#include "src/read.h"
#include "src/util.h"
#include "src/fastqreader.h"
#include "src/evaluator.h"
#undef private
#undef protected

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "fastp_gpu.h"
#include "fastp_gpu_host.h"
#include "fastp_gpu_stream.h"
#include "gpu_worker.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// The reference deals its 1000-read packs round-robin to W worker threads (pack k -> thread k % W,
// src/peprocessor.cpp:787-792) and each thread runs the loop body once per pack.  Here the threads do what is left
// of the body on the host - packing a pack's strings into the engine's rows, and later turning the records into the
// output strings - while the engine sees WINDOWS of K consecutive packs as one batch:
//   * window j = packs [jK, (j+1)K).  A thread writes its pack straight into the window's page-locked rows at
//     (pack - jK) * PACK_SIZE, so the rows of a window are the input stream in input order whatever the threads'
//     timing: Duplicate and the overrepresentation sampling see exactly the `-w 1` stream.
//   * the thread that fills the last missing pack of the oldest unsubmitted window submits it
//     (fastp_gpu_submit_host_async: copies + kernels + result copies queued on one stream, returns at once).
//   * NSLOT windows are in flight: while window j runs, the threads fill j+1 .. j+NSLOT-1.
//   * a thread keeps its packs in a FIFO; whenever it comes by (next pack, or the drain at the end of processorTask)
//     it applies the records of its packs whose window has arrived - fastp_gpu_host_apply into the strings the
//     WriterThreads take (per-thread order = pack order, as WriterThread::input expects) - and recycles the reads.
// The reader's backpressure counter (mPackProcessedCounter) is advanced when a pack is ACCEPTED: the windows are the
// bound on memory now (NSLOT * K packs), and the reader must be allowed to run K packs ahead.
// ---------------------------------------------------------------------------------------------------------------
enum { NSLOT = 3 };

struct Window {
    // page-locked rows of K * PACK_SIZE units
    uint8_t *pseq[2] = {nullptr, nullptr}, *pqual[2] = {nullptr, nullptr};
    uint16_t* plen[2] = {nullptr, nullptr};
    fastp_gpu_read_result* rr[2] = {nullptr, nullptr};
    fastp_gpu_pair_result* pr = nullptr;
    fastp_gpu_correction* corr = nullptr;
    fastp_gpu_adapter_event* ev = nullptr;
    int32_t corr_cap = 0, ev_cap = 0;
    int32_t ncorr = 0, nev = 0;
    std::vector<int32_t> count;            // reads of pack i of the window (PACK_SIZE but for the stream's last pack)
    std::atomic<int> filled{0};            // packs packed so far
    std::atomic<int> applied{0};           // packs whose records have been turned into output
    std::atomic<long> index{-1};           // which window of the stream the slot holds (-1: free)
    std::atomic<long> next{-1};            // the window the slot serves next (slot, slot + NSLOT, ...): set by make_state / the free
    std::atomic<int> state{0};             // 0 filling, 1 submitted, 2 arrived
    int packs = 0;                         // packs in the submitted batch
    // units with letters outside ACGTN: mask per row (the packing threads set it), their raw sequence bytes at row * max_len
    std::vector<uint8_t> xmask, xraw[2];
    std::atomic<int> xany{0};
    std::vector<int32_t> xunit;
    std::vector<uint32_t> xoff;
    fastp_gpu_batch batch;
    fastp_gpu_results res;
};

// the strings a parameter block points at, and the host-side options that go with it
struct ParamBlock {
    fastp_gpu_params params;
    fastp_gpu_host_options ho;
    std::vector<std::string> seeds[2], fasta;        // own copies of the strings the parameter block points at
    std::vector<const char*> seedp[2], fastap;
    std::string a1, a2, umi_prefix, umi_delim;
};

struct State {
    std::mutex mu;               // submission order = window order; also guards fastp_gpu_* calls that touch the stream
    fastp_gpu_ctx* ctx = nullptr;
    ParamBlock B;
    fastp_gpu_counter_layout lay;
    bool paired = false;
    int max_len = 0;
    int W = 1, K = 256;
    size_t ss = 0, qs = 0;
    Window win[NSLOT];
    long next_submit = 0;                  // (under mu) the oldest window not yet submitted
    std::atomic<long> total_packs{-1};     // known once the reader has finished
    std::vector<fastp_gpu_host*> hosts;              // per worker thread: output strings + adapter maps
};
State* G = nullptr;
std::once_flag g_once;

bool enabled() {
    static const int e = [] { const char* v = getenv("FASTP_GPU"); return (v && *v) ? atoi(v) : 0; }();
    return e != 0;
}

// the HIP runtime comes up on a helper thread from the moment the binary is loaded: it overlaps main()'s option parsing
// and the Evaluator pre-pass instead of preceding the first chunk (fastp_gpu_warmup)
struct WarmUp {
    WarmUp() {
        if (!enabled()) return;
        fq::timeline("binding: static initialisers (the binary and its libraries are mapped)");
        atexit([] { fq::timeline("binding: atexit (reports written)"); });
        std::thread([] { (void)fastp_gpu_warmup(0); }).detach();
    }
} g_warm_up;

void refuse(const char* what) { error_exit(std::string("FASTP_GPU=1: ") + what + " is outside the engine's scope"); }

// Options (already validated, Evaluator results applied) -> the engine's flat parameter block
void fill_params(Options* o, bool paired, int max_len, ParamBlock* s) {
    if (o->indexFilter.enabled) refuse("--filter_by_index");
    if (o->fixMGI) refuse("--fix_mgi_id");
    if (o->split.enabled) refuse("--split");
    if (o->outputToSTDOUT) refuse("--stdout");
    if (max_len > FASTP_GPU_MAX_READ_LEN) refuse("reads longer than FASTP_GPU_MAX_READ_LEN");
    fastp_gpu_params& p = s->params;
    fastp_gpu_default_params(&p, paired ? 1 : 0, max_len);
    p.trim_front1 = o->trim.front1;  p.trim_tail1 = o->trim.tail1;
    p.trim_front2 = o->trim.front2;  p.trim_tail2 = o->trim.tail2;
    p.max_len1 = o->trim.maxLen1;    p.max_len2 = o->trim.maxLen2;
    p.cut_front = o->qualityCut.enabledFront;  p.cut_front_window = o->qualityCut.windowSizeFront;
    p.cut_front_quality = o->qualityCut.qualityFront;
    p.cut_tail = o->qualityCut.enabledTail;    p.cut_tail_window = o->qualityCut.windowSizeTail;
    p.cut_tail_quality = o->qualityCut.qualityTail;
    p.cut_right = o->qualityCut.enabledRight;  p.cut_right_window = o->qualityCut.windowSizeRight;
    p.cut_right_quality = o->qualityCut.qualityRight;
    p.poly_g = o->polyGTrim.enabled;  p.poly_g_min_len = o->polyGTrim.minLen;
    p.poly_x = o->polyXTrim.enabled;  p.poly_x_min_len = o->polyXTrim.minLen;
    p.adapter_enabled = o->adapter.enabled;
    s->a1 = o->adapter.sequence;  s->a2 = o->adapter.sequenceR2;
    p.adapter_seq_r1 = o->adapter.hasSeqR1 ? s->a1.c_str() : NULL;
    p.adapter_seq_r2 = (paired && o->adapter.hasSeqR2) ? s->a2.c_str() : NULL;
    if (o->adapter.hasFasta) {
        s->fasta = o->adapter.seqsInFasta;
        for (auto& f : s->fasta) s->fastap.push_back(f.c_str());
        p.adapter_fasta = s->fastap.data();
        p.n_adapter_fasta = (int)s->fastap.size();
    }
    p.allow_gap_overlap_trimming = o->adapter.allowGapOverlapTrimming;
    p.dimer_max_len = o->adapter.dimerMaxLen;
    p.correction = o->correction.enabled;  p.merge = o->merge.enabled;
    p.overlapped_out = (paired && !o->overlappedOut.empty()) ? 1 : 0;   // mOverlappedWriter (peprocessor.cpp:94-97)
    p.merge_include_unmerged = o->merge.includeUnmerged;
    p.overlap_require = o->overlapRequire;  p.overlap_diff_limit = o->overlapDiffLimit;
    p.overlap_diff_percent_limit = o->overlapDiffPercentLimit;
    p.qual_filter = o->qualfilter.enabled;
    p.qualified_qual = o->qualfilter.qualifiedQual - 33;   // the engine applies num2qual itself
    p.unqualified_percent_limit = o->qualfilter.unqualifiedPercentLimit;
    p.n_base_limit = o->qualfilter.nBaseLimit;  p.avg_qual_req = o->qualfilter.avgQualReq;
    p.length_filter = o->lengthFilter.enabled;  p.length_required = o->lengthFilter.requiredLength;
    p.length_limit = o->lengthFilter.maxLength;
    p.complexity_filter = o->complexityFilter.enabled;
    p.complexity_threshold = o->complexityFilter.threshold;
    p.dup_enabled = o->duplicate.enabled;  p.dedup = o->duplicate.dedup;
    p.dup_accuracy_level = o->duplicate.accuracyLevel;
    p.insert_size_max = o->insertSizeMax;
    if (o->umi.enabled && (o->umi.location == UMI_LOC_READ1 || o->umi.location == UMI_LOC_PER_READ)) p.umi_len1 = o->umi.length;
    if (o->umi.enabled && paired && (o->umi.location == UMI_LOC_READ2 || o->umi.location == UMI_LOC_PER_READ)) p.umi_len2 = o->umi.length;
    if (o->umi.enabled && (o->umi.location == UMI_LOC_INDEX1 || o->umi.location == UMI_LOC_INDEX2 || o->umi.location == UMI_LOC_PER_INDEX))
        refuse("--umi_loc index1/index2/per_index");
    p.umi_skip = o->umi.skip;
    if (o->overRepAnalysis.enabled) {   // seeds in std::map order = the order Stats keeps its own counters in
        p.overrep_enabled = 1;
        p.overrep_sampling = o->overRepAnalysis.sampling;
        for (auto& kv : o->overRepSeqs1) s->seeds[0].push_back(kv.first);
        for (auto& kv : o->overRepSeqs2) s->seeds[1].push_back(kv.first);
        for (int m = 0; m < 2; m++) for (auto& q : s->seeds[m]) s->seedp[m].push_back(q.c_str());
        p.overrep_seqs1 = s->seedp[0].data();  p.n_overrep_seqs1 = (int)s->seedp[0].size();
        p.overrep_seqs2 = s->seedp[1].data();  p.n_overrep_seqs2 = paired ? (int)s->seedp[1].size() : 0;
        p.eval_seq_len1 = o->seqLen1;  p.eval_seq_len2 = o->seqLen2;
    }
    fastp_gpu_host_options& ho = s->ho;
    memset(&ho, 0, sizeof(ho));
    ho.want_failed = !o->failedOut.empty();
    ho.want_unpaired1 = !o->unpaired1.empty();
    ho.want_unpaired2 = !o->unpaired2.empty() && o->unpaired2 != o->unpaired1;
    if (o->umi.enabled) {
        ho.umi_loc = o->umi.location == UMI_LOC_READ1 ? FASTP_GPU_UMI_READ1
                   : o->umi.location == UMI_LOC_READ2 ? FASTP_GPU_UMI_READ2 : FASTP_GPU_UMI_PER_READ;
        ho.umi_len = o->umi.length;
        s->umi_prefix = o->umi.prefix;
        s->umi_delim = o->umi.delimiter;
        ho.umi_prefix = s->umi_prefix.empty() ? NULL : s->umi_prefix.c_str();
        ho.umi_delimiter = s->umi_delim.empty() ? NULL : s->umi_delim.c_str();
    }
}

void make_state(Options* o, bool paired) {
    State* s = new State();
    s->paired = paired;
    // pack mode is what is left for the option sets the stream binding below does not take (none any more: FASTP_GPU_STREAM=0 and the FASTP_GPU_STREAM_* switches select it); it is bound by the reference's reader thread, so the rows are sized generously rather
    // than by the first 1000 reads (Evaluator::computeSeqLen evaluator.cpp:54-76): a longer read later in the file is
    // what the reference takes in its stride (Stats::extendBuffer stats.cpp:65-83)
    s->max_len = std::max(o->seqLen1, paired ? o->seqLen2 : 0);
    if (s->max_len <= 0) s->max_len = 151;
    s->max_len = std::min<int>(FASTP_GPU_MAX_READ_LEN, std::max(s->max_len + s->max_len / 2, 256));
    if (const char* v = getenv("FASTP_GPU_MAX_LEN")) s->max_len = atoi(v);
    fill_params(o, paired, s->max_len, &s->B);
    fastp_gpu_params& p = s->B.params;
    int rc = fastp_gpu_create(&p, 0, &s->ctx);
    if (rc != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_create: ") + fastp_gpu_last_error(NULL));   // never a silent CPU fallback
    fastp_gpu_counter_layout_for_params(&p, &s->lay);
    const fastp_gpu_host_options& ho = s->B.ho;
    for (int t = 0; t < o->thread; t++) {
        fastp_gpu_host* h = NULL;
        if (fastp_gpu_host_create(&p, &ho, &h) != FASTP_GPU_OK) error_exit("fastp_gpu_host_create failed");
        s->hosts.push_back(h);
    }
    // the windows: K packs each (FASTP_GPU_PACKS).  K = PACK_IN_MEM_LIMIT / 2 (16): the reader thread pauses whenever a
    // WriterThread holds more than that many strings (peprocessor.cpp:842-849), and a window's outputs reach the writers
    // in one burst - with 256-pack windows the reader sat out the writer's drain after every window and the run was
    // slower than the CPU loop (profiles/r03_dropin_first.txt); 16 beat 32 and 64 at every thread count (r03_dropin.txt).
    // 16 K pairs per batch still take the engine ~0.1 ms.
    s->W = std::max(1, o->thread);
    s->K = PACK_IN_MEM_LIMIT / 2;
    if (const char* v = getenv("FASTP_GPU_PACKS")) s->K = std::max(1, atoi(v));
    s->ss = fastp_gpu_seq_stride(s->max_len);
    s->qs = fastp_gpu_qual_stride(s->max_len);
    const size_t units = (size_t)s->K * PACK_SIZE;
    auto pinned = [&](size_t bytes) {
        void* q = nullptr;
        if (fastp_gpu_host_alloc(s->ctx, (int64_t)bytes, &q) != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_host_alloc: ") + fastp_gpu_last_error(s->ctx));
        return q;
    };
    for (Window& w : s->win) {
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            w.pseq[m] = (uint8_t*)pinned(units * s->ss);
            w.pqual[m] = (uint8_t*)pinned(units * s->qs);
            w.plen[m] = (uint16_t*)pinned(units * 2);
            w.rr[m] = (fastp_gpu_read_result*)pinned(units * sizeof(fastp_gpu_read_result));
        }
        if (paired) w.pr = (fastp_gpu_pair_result*)pinned(units * sizeof(fastp_gpu_pair_result));
        if (p.correction) { w.corr_cap = (int32_t)std::min<size_t>(units * 8 + 1024, (size_t)1 << 26); w.corr = (fastp_gpu_correction*)pinned((size_t)w.corr_cap * sizeof(fastp_gpu_correction)); }
        if (p.n_adapter_fasta) { w.ev_cap = (int32_t)(units * 2 * std::min(p.n_adapter_fasta, 8) + 16); w.ev = (fastp_gpu_adapter_event*)pinned((size_t)w.ev_cap * sizeof(fastp_gpu_adapter_event)); }
        w.count.assign((size_t)s->K, 0);
        w.xmask.assign(units, 0);
        for (int m = 0; m < (paired ? 2 : 1); m++) w.xraw[m].assign(units * (size_t)s->max_len, 0);   // virtual until a unit with such letters turns up
    }
    for (int k = 0; k < NSLOT; k++) s->win[k].next.store(k);
    G = s;
}

// a worker thread's packs that the engine has not answered yet
struct Pending {
    long seq;                    // pack number in the stream
    ReadPack *left, *right;      // right == NULL for single-end
    int n;
};
struct Scratch {
    std::vector<const char*> name[2], seq[2], qual[2], strand[2];
    std::vector<int32_t> name_len[2], len[2], strand_len[2];
    std::deque<Pending> pending;
    long npacks = 0;             // packs this thread has accepted
    std::vector<fastp_gpu_correction> corr;
    std::vector<fastp_gpu_adapter_event> ev;
};
thread_local Scratch T;

void gather(Read** data, int n, int m) {
    T.name[m].resize(n); T.seq[m].resize(n); T.qual[m].resize(n); T.strand[m].resize(n);
    T.name_len[m].resize(n); T.len[m].resize(n); T.strand_len[m].resize(n);
    for (int i = 0; i < n; i++) {   // Read = four heap strings (read.h)
        Read* r = data[i];
        T.name[m][i] = r->mName->data();     T.name_len[m][i] = (int32_t)r->mName->size();
        T.seq[m][i] = r->mSeq->data();       T.len[m][i] = (int32_t)r->mSeq->size();
        T.qual[m][i] = r->mQuality->data();
        T.strand[m][i] = r->mStrand->data(); T.strand_len[m][i] = (int32_t)r->mStrand->size();
    }
}

fastp_gpu_reads reads_of(int n, int m) {
    fastp_gpu_reads r;
    r.n = n;
    r.name = T.name[m].data(); r.name_len = T.name_len[m].data();
    r.seq = T.seq[m].data(); r.qual = T.qual[m].data(); r.len = T.len[m].data();
    r.strand = T.strand[m].data(); r.strand_len = T.strand_len[m].data();
    return r;
}

std::string* take(int tid, int which) {
    size_t len = 0;
    const char* s = fastp_gpu_host_output(G->hosts[tid], which, &len);
    return new std::string(s ? s : "", s ? len : 0);
}

long packs_expected(long j) {   // packs window j will hold, or -1 while the stream's length is unknown and the window may be its last
    const long total = G->total_packs.load(std::memory_order_acquire);
    if (total < 0) return -1;
    return std::max<long>(0, std::min<long>(G->K, total - j * G->K));
}

// submit every complete window at the head of the stream, in order; poll the ones in flight
void pump() {
    std::unique_lock<std::mutex> lk(G->mu, std::try_to_lock);
    if (!lk.owns_lock()) return;          // somebody else is pumping
    for (;;) {
        Window& w = G->win[G->next_submit % NSLOT];
        if (w.index.load(std::memory_order_acquire) != G->next_submit || w.state.load(std::memory_order_acquire) != 0) break;
        const long exp = packs_expected(G->next_submit);
        const int have = w.filled.load(std::memory_order_acquire);
        if (!(have == G->K || (exp >= 0 && have == exp)) || have == 0) {
            if (exp == 0 && have == 0) { /* the stream ended on a window boundary: nothing to submit */ }
            break;
        }
        // rows are contiguous when every pack in front of the last non-empty one is full: the reader leaves the stream's
        // last pack short, and --reads_to_process / unequal files put an empty terminal pack behind a short one
        // (peprocessor.cpp:757-771, :775-778)
        long n = 0;
        bool contiguous = true, ended = false;
        for (int i = 0; i < have; i++) {
            const int cnt = w.count[(size_t)i];
            if (ended && cnt != 0) contiguous = false;
            if (cnt != PACK_SIZE) ended = true;
            if (cnt > 0) n = (long)i * PACK_SIZE + cnt;
        }
        if (!contiguous) error_exit("FASTP_GPU=1: a short pack in the middle of the stream (unexpected reader behaviour)");
        memset(&w.batch, 0, sizeof(w.batch));
        w.batch.n = (int32_t)n;
        w.batch.flags = FASTP_GPU_BATCH_STAT_ISIZE;   // the `-w 1` semantics: every pair's insert size (the reference samples thread 0's packs)
        w.batch.seq1 = w.pseq[0]; w.batch.qual1 = w.pqual[0]; w.batch.len1 = w.plen[0];
        if (G->paired) { w.batch.seq2 = w.pseq[1]; w.batch.qual2 = w.pqual[1]; w.batch.len2 = w.plen[1]; }
        if (w.xany.load(std::memory_order_acquire)) {
            w.xunit.clear();
            w.xoff.clear();
            for (long u = 0; u < n; u++)
                if (w.xmask[(size_t)u]) { w.xunit.push_back((int32_t)u); w.xoff.push_back((uint32_t)((size_t)u * (size_t)G->max_len)); }
            w.batch.n_exotic = (int32_t)w.xunit.size();
            w.batch.exotic_unit = w.xunit.data();
            for (int m = 0; m < (G->paired ? 2 : 1); m++) {
                w.batch.exotic_text[m] = w.xraw[m].data();
                w.batch.exotic_off[m] = w.xoff.data();
                w.batch.exotic_text_bytes[m] = (int64_t)w.xraw[m].size();
            }
            std::fill(w.xmask.begin(), w.xmask.end(), 0);
            w.xany.store(0, std::memory_order_relaxed);
        }
        memset(&w.res, 0, sizeof(w.res));
        w.res.r1 = w.rr[0];
        if (G->paired) { w.res.r2 = w.rr[1]; w.res.pair = w.pr; }
        w.res.corrections = w.corr; w.res.corrections_capacity = w.corr_cap; w.res.n_corrections = &w.ncorr;
        w.res.adapter_events = w.ev; w.res.adapter_events_capacity = w.ev_cap; w.res.n_adapter_events = &w.nev;
        w.packs = have;
        if (n > 0) {
            const int rc = fastp_gpu_submit_host_async(G->ctx, &w.batch, &w.res, (int)(G->next_submit % NSLOT));
            if (rc != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_submit_host_async: ") + fastp_gpu_last_error(G->ctx));
            w.state.store(1, std::memory_order_release);
        } else {
            w.state.store(2, std::memory_order_release);
        }
        G->next_submit++;
    }
    for (int s = 0; s < NSLOT; s++) {
        Window& w = G->win[s];
        if (w.state.load(std::memory_order_acquire) == 1) {
            const int r = fastp_gpu_poll(G->ctx, s);
            if (r < 0) error_exit(std::string("fastp_gpu_poll: ") + fastp_gpu_last_error(G->ctx));
            if (r == 1) w.state.store(2, std::memory_order_release);
        }
    }
}

// the slot of window j, once it is this window's turn to use it
Window* window_for(long j) {
    Window& w = G->win[j % NSLOT];
    for (;;) {
        long idx = w.index.load(std::memory_order_acquire);
        if (idx == j) return &w;
        if (idx == -1) {   // free: claim it for window j (several threads may try; one wins, the others see idx == j)
            // ... but only when it is window j's turn: a thread that runs ahead must not take the slot of a window whose
            // packs have not been picked up yet (window 3 in front of window 0: nobody could ever submit window 0)
            if (w.next.load(std::memory_order_acquire) != j) return nullptr;
            long expect = -1;
            if (w.index.compare_exchange_strong(expect, j, std::memory_order_acq_rel)) return &w;
            continue;
        }
        return nullptr;    // still holds window j - NSLOT
    }
}

template <class Proc>
void emit_pe(Proc* pp, int tid) {
    if (pp->mMergedWriter) pp->mMergedWriter->input(tid, take(tid, FASTP_GPU_MERGED));
    if (pp->mFailedWriter) pp->mFailedWriter->input(tid, take(tid, FASTP_GPU_FAILED));
    if (pp->mOverlappedWriter) pp->mOverlappedWriter->input(tid, take(tid, FASTP_GPU_OVERLAPPED));   // :662-664
    if (pp->mRightWriter && pp->mLeftWriter) {
        pp->mLeftWriter->input(tid, take(tid, FASTP_GPU_OUT1));
        pp->mRightWriter->input(tid, take(tid, FASTP_GPU_OUT2));
    } else if (pp->mLeftWriter) {
        pp->mLeftWriter->input(tid, new std::string());   // the interleaved-to-one-stream form needs --stdout (refused above)
    }
    if (pp->mUnpairedLeftWriter && pp->mUnpairedRightWriter) {
        pp->mUnpairedLeftWriter->input(tid, take(tid, FASTP_GPU_UNPAIRED1));
        pp->mUnpairedRightWriter->input(tid, take(tid, FASTP_GPU_UNPAIRED2));
    } else if (pp->mUnpairedLeftWriter) {
        pp->mUnpairedLeftWriter->input(tid, take(tid, FASTP_GPU_UNPAIRED1));
    }
}

// apply the records of this thread's packs whose window has arrived (in pack order); returns packs applied
template <class Emit, class Recycle>
int drain_ready(int tid, Emit emit, Recycle recycle) {
    int done = 0;
    while (!T.pending.empty()) {
        Pending& pd = T.pending.front();
        const long j = pd.seq / G->K;
        Window& w = G->win[j % NSLOT];
        if (w.index.load(std::memory_order_acquire) != j || w.state.load(std::memory_order_acquire) != 2) break;
        const int n = pd.n;
        const size_t row = (size_t)(pd.seq - j * G->K) * PACK_SIZE;
        gather(pd.left->data, n, 0);
        if (pd.right) gather(pd.right->data, n, 1);
        fastp_gpu_results res;
        memset(&res, 0, sizeof(res));
        res.r1 = w.rr[0] + row;
        if (pd.right) { res.r2 = w.rr[1] + row; res.pair = w.pr + row; }
        // the sparse lists of the window, narrowed to this pack and re-based to it
        int32_t ncorr = 0, nev = 0;
        const uint32_t lo = (uint32_t)row * (pd.right ? 2u : 1u), hi = (uint32_t)(row + (size_t)n) * (pd.right ? 2u : 1u);
        if (w.corr && w.ncorr > 0) {
            T.corr.clear();
            for (int32_t i = 0; i < w.ncorr; i++)
                if (w.corr[i].read >= lo && w.corr[i].read < hi) { T.corr.push_back(w.corr[i]); T.corr.back().read -= lo; }
            res.corrections = T.corr.data(); ncorr = (int32_t)T.corr.size(); res.corrections_capacity = ncorr; res.n_corrections = &ncorr;
        }
        if (w.ev && w.nev > 0) {
            T.ev.clear();
            for (int32_t i = 0; i < w.nev; i++)
                if (w.ev[i].read >= lo && w.ev[i].read < hi) { T.ev.push_back(w.ev[i]); T.ev.back().read -= lo; }
            res.adapter_events = T.ev.data(); nev = (int32_t)T.ev.size(); res.adapter_events_capacity = nev; res.n_adapter_events = &nev;
        }
        fastp_gpu_reads r1 = reads_of(n, 0), r2;
        if (pd.right) r2 = reads_of(n, 1);
        fastp_gpu_host_clear_outputs(G->hosts[tid]);
        if (fastp_gpu_host_apply(G->hosts[tid], &r1, pd.right ? &r2 : NULL, &res) != FASTP_GPU_OK) error_exit("fastp_gpu_host_apply failed");
        emit(tid);
        recycle(pd);
        // the last pack of a window to be applied frees the slot
        if (w.applied.fetch_add(1, std::memory_order_acq_rel) + 1 == w.packs) {
            w.applied.store(0, std::memory_order_relaxed);
            w.filled.store(0, std::memory_order_relaxed);
            w.state.store(0, std::memory_order_relaxed);
            w.next.store(j + NSLOT, std::memory_order_release);
            w.index.store(-1, std::memory_order_release);
        }
        T.pending.pop_front();
        done++;
    }
    return done;
}

// pack the reads of one ReadPack pair into its rows of its window; false = the engine refuses a read of it
bool pack_into(Window& w, size_t row, int n, bool paired) {
    for (int m = 0; m < (paired ? 2 : 1); m++) {
        int32_t bad = -1;
        if (fastp_gpu_pack_reads_x(G->max_len, n, T.seq[m].data(), T.qual[m].data(), T.len[m].data(), w.pseq[m] + row * G->ss,
                                   w.pqual[m] + row * G->qs, w.plen[m] + row, &bad, w.xmask.data() + row) != FASTP_GPU_OK)
            return false;
    }
    // letters outside ACGTN: the unit's text travels with the batch (fastp_gpu_batch::exotic_*), the engine's text kernel takes it
    for (int i = 0; i < n; i++) {
        if (!w.xmask[row + (size_t)i]) continue;
        for (int m = 0; m < (paired ? 2 : 1); m++) {
            memcpy(w.xraw[m].data() + (row + (size_t)i) * (size_t)G->max_len, T.seq[m][(size_t)i], (size_t)T.len[m][(size_t)i]);
        }
        w.xany.store(1, std::memory_order_release);
    }
    return true;
}

// the engine's counter block added onto one Stats object (the per-cycle part has Stats::mCycleBuffer's layout)
const fastp_gpu_counter_layout* LAY = nullptr;   // the layout of the block being loaded (pack mode: G->lay, stream mode: the stream's)

void load_stats(Stats* st, const std::vector<int64_t>& c, int slot, const std::vector<std::string>& seeds) {
    const fastp_gpu_counter_layout& L = *LAY;
    const int64_t base = L.stats[slot];
    st->mReads += c[base + L.st_reads];
    st->mLengthSum += c[base + L.st_length_sum];
    for (int q = 0; q < 128; q++) st->mBaseQualHistogram[q] += c[base + L.st_qual_hist + q];
    for (int k = 0; k < 1024; k++) st->mKmer[k] += c[base + L.st_kmer + k];
    const int cycles = (int)L.cycles;
    if (st->mBufLen < cycles) st->extendBuffer(cycles);
    for (int a = 0; a < 34; a++)   // CYCLE_ARRAY_COUNT arrays of mBufLen longs each
        for (int i = 0; i < cycles; i++) st->mCycleBuffer[(size_t)a * st->mBufLen + i] += c[base + L.st_cycle + (int64_t)a * cycles + i];
    const int ns = (int)L.n_overrep[slot], el = (int)L.eval_len[slot];
    for (int k = 0; k < ns && k < (int)seeds.size(); k++) {
        st->mOverRepSeq[seeds[k]] += c[L.overrep_count[slot] + k];
        long* dist = st->mOverRepSeqDist[seeds[k]];
        for (int i = 0; i < el && dist; i++) dist[i] += c[L.overrep_dist[slot] + (int64_t)k * el + i];
    }
}

void load_filter_result(FilterResult* fr, const std::vector<int64_t>& c) {
    const fastp_gpu_counter_layout& L = *LAY;
    for (int i = 0; i < FILTER_RESULT_TYPES; i++) fr->mFilterReadStats[i] += c[L.filter_stats + i];
    fr->mTrimmedAdapterRead += c[L.adapter_reads];
    fr->mTrimmedAdapterBases += c[L.adapter_bases];
    for (int i = 0; i < 4; i++) { fr->mTrimmedPolyXReads[i] += c[L.polyx_reads + i]; fr->mTrimmedPolyXBases[i] += c[L.polyx_bases + i]; }
    for (int i = 0; i < 64; i++) fr->mCorrectionMatrix[i] += c[L.correction + i];
    fr->mCorrectedReads += c[L.corrected_reads];
    fr->mMergedPairs += c[L.merged_pairs];
}

void load_adapters(ThreadConfig** configs, const std::vector<fastp_gpu_host*>& hosts, int threads) {
    for (int t = 0; t < threads && t < (int)hosts.size(); t++)
        for (int m = 0; m < 2; m++) {
            const int64_t n = fastp_gpu_host_adapter_entries(hosts[t], m);
            auto& dst = m ? configs[t]->getFilterResult()->mAdapter2 : configs[t]->getFilterResult()->mAdapter1;
            for (int64_t i = 0; i < n; i++) {
                const char* s; int32_t len; int64_t cnt;
                fastp_gpu_host_adapter_entry(hosts[t], m, i, &s, &len, &cnt);
                dst[std::string(s, (size_t)len)] += cnt;
            }
        }
}

std::vector<int64_t> fetch_counters() {
    if (fastp_gpu_synchronize(G->ctx) != FASTP_GPU_OK) error_exit("fastp_gpu_synchronize failed");
    std::vector<int64_t> c((size_t)G->lay.total);
    if (fastp_gpu_counters(G->ctx, c.data(), (int64_t)c.size()) != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_counters: ") + fastp_gpu_last_error(G->ctx));
    return c;
}

void shutdown() {
    for (Window& w : G->win) {
        void* bufs[] = {w.pseq[0], w.pseq[1], w.pqual[0], w.pqual[1], w.plen[0], w.plen[1], w.rr[0], w.rr[1], w.pr, w.corr, w.ev};
        for (void* b : bufs) if (b) fastp_gpu_host_free(G->ctx, b);
    }
    for (auto* h : G->hosts) fastp_gpu_host_destroy(h);
    fastp_gpu_destroy(G->ctx);
    delete G;
    G = nullptr;
}

}  // namespace

// the stream's length is known once the reader has handed out its last pack (a thread sees that on its own input list)
// Paired: a worker's loop can also end because ITS read-2 list is exhausted while the read-1 reader is still producing
// packs for other threads (processorTask's second exit, peprocessor.cpp:1034-1037) - the pack counters are final only
// when both readers have closed their lists, and the stream holds min(read-1 packs, read-2 packs) pack pairs.
void note_total(PairEndProcessor* pp, ThreadConfig* config) {
    if (G->total_packs.load(std::memory_order_acquire) >= 0) return;
    if (config->getLeftInput()->isProducerFinished() && config->getRightInput()->isProducerFinished())
        G->total_packs.store(std::min((long)pp->mLeftPackReadCounter, (long)pp->mRightPackReadCounter), std::memory_order_release);
}
void note_total_se(SingleEndProcessor* sp, SingleProducerSingleConsumerList<ReadPack*>* in) {
    if (G->total_packs.load(std::memory_order_acquire) >= 0) return;
    if (in->isProducerFinished()) G->total_packs.store((long)sp->mPackReadCounter, std::memory_order_release);
}

void refuse_pack() {
    error_exit("FASTP_GPU=1: a pack holds reads the engine refuses (quality characters outside '!'..'~', or a read longer than the rows "
               "pack mode sized - FASTP_GPU_MAX_LEN); rerun without FASTP_GPU=1");
}

int fastp_gpu_worker_pe(PairEndProcessor* pp, ReadPack* left, ReadPack* right, ThreadConfig* config) {
    if (!enabled()) return -1;
    Options* o = pp->mOptions;
    std::call_once(g_once, [&] { make_state(o, true); });
    if (left->count != right->count) {   // peprocessor.cpp:363-370
        cerr << endl << "WARNING: different read numbers of the " << pp->mPackProcessedCounter << " pack" << endl;
        cerr << "Read1 pack size: " << left->count << endl << "Read2 pack size: " << right->count << endl;
        cerr << "Ignore the unmatched reads" << endl << endl;
        pp->shouldStopReading = true;
    }
    const int tid = config->getThreadId();
    const int n = std::min(left->count, right->count);
    auto emit = [&](int t) { emit_pe(pp, t); };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) pp->recycleToPool1(tid, pd.left->data[i]);
        for (int i = 0; i < pd.right->count; i++) pp->recycleToPool2(tid, pd.right->data[i]);
        delete[] pd.left->data;
        delete[] pd.right->data;
        delete pd.left;
        delete pd.right;
    };
    const long seq = T.npacks * G->W + tid;   // pack k goes to thread k % W and a thread's packs arrive in order (peprocessor.cpp:787-792)
    T.npacks++;
    const long j = seq / G->K;
    Window* w;
    while (!(w = window_for(j))) {            // the slot still holds window j - NSLOT: help it along
        note_total(pp, config);
        pump();
        if (!drain_ready(tid, emit, recycle)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const size_t slot = (size_t)(seq - j * G->K);
    gather(left->data, n, 0);
    gather(right->data, n, 1);
    if (n > 0 && !pack_into(*w, slot * PACK_SIZE, n, true)) refuse_pack();   // the stream's terminal pack may be empty (reads % 1000 == 0)
    w->count[slot] = n;
    w->filled.fetch_add(1, std::memory_order_acq_rel);
    T.pending.push_back(Pending{seq, left, right, n});
    config->markProcessed(left->count);
    pp->mPackProcessedCounter.fetch_add(1, std::memory_order_release);   // accepted: the windows bound the memory now
    pp->mBackpressureCV.notify_all();
    note_total(pp, config);
    pump();
    drain_ready(tid, emit, recycle);
    return 1;
}

// end of processorTask: everything this thread still holds goes out before the writers are told the input is complete
void fastp_gpu_worker_drain_pe(PairEndProcessor* pp, ThreadConfig* config) {
    if (!G) return;
    const int tid = config->getThreadId();
    auto emit = [&](int t) { emit_pe(pp, t); };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) pp->recycleToPool1(tid, pd.left->data[i]);
        for (int i = 0; i < pd.right->count; i++) pp->recycleToPool2(tid, pd.right->data[i]);
        delete[] pd.left->data;
        delete[] pd.right->data;
        delete pd.left;
        delete pd.right;
    };
    // this thread's loop has ended; the stream's length is final once BOTH readers are done (see note_total)
    long spins = 0;
    while (!T.pending.empty()) {
        note_total(pp, config);
        pump();
        if (!drain_ready(tid, emit, recycle)) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (getenv("FASTP_GPU_DEBUG_BINDING") && ++spins % 20000 == 0) {
            fprintf(stderr, "drain tid %d: pending front %ld (n=%zu) next_submit %ld total %ld L %ld R %ld lf %d rf %d\n", tid, T.pending.front().seq, T.pending.size(), G->next_submit,
                    G->total_packs.load(), (long)pp->mLeftPackReadCounter, (long)pp->mRightPackReadCounter, (int)config->getLeftInput()->isProducerFinished(), (int)config->getRightInput()->isProducerFinished());
            for (int k = 0; k < NSLOT; k++) fprintf(stderr, "   slot %d: index %ld state %d filled %d applied %d packs %d\n", k, G->win[k].index.load(), G->win[k].state.load(), G->win[k].filled.load(), G->win[k].applied.load(), G->win[k].packs);
        }
    }
    note_total(pp, config);
    pump();
}

void fastp_gpu_worker_idle_pe(PairEndProcessor* pp, ThreadConfig* config) {
    if (!G || T.pending.empty()) return;
    const int tid = config->getThreadId();
    auto emit = [&](int t) { emit_pe(pp, t); };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) pp->recycleToPool1(tid, pd.left->data[i]);
        for (int i = 0; i < pd.right->count; i++) pp->recycleToPool2(tid, pd.right->data[i]);
        delete[] pd.left->data;
        delete[] pd.right->data;
        delete pd.left;
        delete pd.right;
    };
    note_total(pp, config);
    pump();
    drain_ready(tid, emit, recycle);
}

int fastp_gpu_worker_se(SingleEndProcessor* sp, ReadPack* pack, ThreadConfig* config) {
    if (!enabled()) return -1;
    Options* o = sp->mOptions;
    std::call_once(g_once, [&] { make_state(o, false); });
    const int tid = config->getThreadId();
    const int n = pack->count;
    auto emit = [&](int t) {
        if (sp->mLeftWriter) sp->mLeftWriter->input(t, take(t, FASTP_GPU_OUT1));     // seprocessor.cpp:299-304
        if (sp->mFailedWriter) sp->mFailedWriter->input(t, take(t, FASTP_GPU_FAILED));
    };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) sp->recycleToPool(tid, pd.left->data[i]);
        delete[] pd.left->data;
        delete pd.left;
    };
    const long seq = T.npacks * G->W + tid;
    T.npacks++;
    const long j = seq / G->K;
    Window* w;
    while (!(w = window_for(j))) {
        note_total_se(sp, config->getLeftInput());
        pump();
        if (!drain_ready(tid, emit, recycle)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const size_t slot = (size_t)(seq - j * G->K);
    gather(pack->data, n, 0);
    if (n > 0 && !pack_into(*w, slot * PACK_SIZE, n, false)) refuse_pack();
    w->count[slot] = n;
    w->filled.fetch_add(1, std::memory_order_acq_rel);
    T.pending.push_back(Pending{seq, pack, nullptr, n});
    config->markProcessed(pack->count);
    sp->mPackProcessedCounter.fetch_add(1, std::memory_order_release);
    sp->mBackpressureCV.notify_all();
    note_total_se(sp, config->getLeftInput());
    pump();
    drain_ready(tid, emit, recycle);
    return 1;
}

void fastp_gpu_worker_drain_se(SingleEndProcessor* sp, ThreadConfig* config) {
    if (!G) return;
    const int tid = config->getThreadId();
    auto emit = [&](int t) {
        if (sp->mLeftWriter) sp->mLeftWriter->input(t, take(t, FASTP_GPU_OUT1));
        if (sp->mFailedWriter) sp->mFailedWriter->input(t, take(t, FASTP_GPU_FAILED));
    };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) sp->recycleToPool(tid, pd.left->data[i]);
        delete[] pd.left->data;
        delete pd.left;
    };
    G->total_packs.store((long)sp->mPackReadCounter, std::memory_order_release);
    while (!T.pending.empty()) {
        pump();
        if (!drain_ready(tid, emit, recycle)) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    pump();
}

void fastp_gpu_worker_idle_se(SingleEndProcessor* sp, ThreadConfig* config) {
    if (!G || T.pending.empty()) return;
    const int tid = config->getThreadId();
    auto emit = [&](int t) {
        if (sp->mLeftWriter) sp->mLeftWriter->input(t, take(t, FASTP_GPU_OUT1));
        if (sp->mFailedWriter) sp->mFailedWriter->input(t, take(t, FASTP_GPU_FAILED));
    };
    auto recycle = [&](Pending& pd) {
        for (int i = 0; i < pd.left->count; i++) sp->recycleToPool(tid, pd.left->data[i]);
        delete[] pd.left->data;
        delete pd.left;
    };
    note_total_se(sp, config->getLeftInput());
    pump();
    drain_ready(tid, emit, recycle);
}

// the counter block (layout *LAY) into the first worker's Stats / FilterResult objects, Duplicate's totals and the
// insert-size histogram; the reference's own merge + reporters take it from there
template <class Proc>
void load_block(Proc* pp, ThreadConfig** configs, const std::vector<int64_t>& c, const ParamBlock& B, bool paired) {
    load_stats(configs[0]->getPreStats1(), c, FASTP_GPU_STATS_PRE1, B.seeds[0]);
    load_stats(configs[0]->getPostStats1(), c, FASTP_GPU_STATS_POST1, B.seeds[0]);
    if (paired) {
        load_stats(configs[0]->getPreStats2(), c, FASTP_GPU_STATS_PRE2, B.seeds[1]);
        load_stats(configs[0]->getPostStats2(), c, FASTP_GPU_STATS_POST2, B.seeds[1]);
    }
    load_filter_result(configs[0]->getFilterResult(), c);
    if (pp->mDuplicate) {
        pp->mDuplicate->mTotalReads += (unsigned long)c[LAY->dup_total];
        pp->mDuplicate->mDupReads += (unsigned long)c[LAY->dup_count];
    }
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// STREAM MODE (the default with FASTP_GPU=1): the reference's reader threads, worker threads and the string side of its
// writers are bypassed by ONE loop - include/fastp_gpu_stream.h: raw chunks of the input files -> page-locked memory ->
// fastp_gpu_parse_fastq -> fastp_gpu_submit_device -> fastp_gpu_format_streams (-> fastp_gpu_deflate_bgzf for ".gz") ->
// the files the reference's own WriterThread objects opened.  The hook sits at the top of readerTask: the read-1 reader
// thread runs the loop, the read-2 reader thread returns at once, and when the loop is done both sets of input lists
// are closed - the (idle) worker threads then leave processorTask and the last of them completes the writers, exactly
// as after a normal run.  Evaluator, Options, WriterThread (file ownership, gz truncation), Stats / FilterResult /
// Duplicate objects, JSON and HTML reporters stay the reference's.
//   FASTP_GPU_STREAM=0       pack mode (below) for every option set
//   FASTP_GPU_WRITER=input   hand the text to WriterThread::input as strings (one per chunk, threads in turn) instead
//                            of writing it into the writers' file descriptors with positional writes
// ---------------------------------------------------------------------------------------------------------------
struct StreamState {
    ParamBlock B;
    fastp_gpu_stream* st = nullptr;
    fastp_gpu_host* host = nullptr;
    WriterThread* writer[FASTP_GPU_N_HOST_OUTPUTS] = {};   // [FASTP_GPU_OVERLAPPED]: --overlapped_out's writer, always fed through input()
    long fed[FASTP_GPU_N_HOST_OUTPUTS] = {};     // strings handed to each writer so far (FASTP_GPU_WRITER=input)
    int W = 1;
    bool paired = false, ran = false;
};
StreamState* SG = nullptr;
std::once_flag sg_once;

// (".gz" inputs included: the stream inflates bgzip-written files on the device and other gzip streams with zlib,
// fastp_gpu_stream.h; FASTP_GPU_STREAM_GZ=0 sends them through the reference's reader - pack mode - instead)
bool plain_regular_file(const std::string& path) {
    if (path.empty()) return false;
    if (ends_with(path, ".gz"))
        if (const char* v = getenv("FASTP_GPU_STREAM_GZ")) if (atoi(v) == 0) return false;
    struct stat sb;   // (a pipe as well: --stdin = "/dev/stdin", a FIFO)
    return stat(path.c_str(), &sb) == 0 && (S_ISREG(sb.st_mode) || S_ISFIFO(sb.st_mode));
}

// the option sets the stream loop takes; everything else goes through pack mode
bool stream_mode(Options* o, bool paired) {
    if (!enabled()) return false;
    if (const char* v = getenv("FASTP_GPU_STREAM")) if (atoi(v) == 0) return false;
    if (!o->overlappedOut.empty()) {
        if (const char* v = getenv("FASTP_GPU_STREAM_OVERLAPPED")) if (atoi(v) == 0) return false;   // (pack mode for comparison)
    }
    if (o->interleavedInput) {
        if (const char* v = getenv("FASTP_GPU_STREAM_INTERLEAVED")) if (atoi(v) == 0) return false;   // (pack mode for comparison)
    }
    if (!plain_regular_file(o->in1) || (paired && !o->interleavedInput && !plain_regular_file(o->in2))) return false;
    if (paired && !o->out1.empty() && o->out2.empty()) return false;   // two reads into one stream: --stdout's form
    return true;
}

int emit_to_writer(void* user, int stream, const char* data, int64_t len) {
    StreamState* S = (StreamState*)user;
    WriterThread* w = S->writer[stream];
    if (!w) return 0;
    while (w->bufferLength() > PACK_IN_MEM_LIMIT) usleep(200);   // the reader's own backpressure rule (peprocessor.cpp:842-849)
    w->input((int)(S->fed[stream]++ % S->W), new std::string(data, (size_t)len));
    return 0;
}

void stream_setup(Options* o, bool paired, WriterThread* const writers[FASTP_GPU_N_OUTPUTS], WriterThread* overlapped_writer = NULL) {
    StreamState* S = new StreamState();
    fq::timeline("binding: stream_setup begin (options parsed, Evaluator pre-pass done, threads started)");
    S->paired = paired;
    S->W = std::max(1, o->thread);
    int max_len = std::max(o->seqLen1, paired ? o->seqLen2 : 0);   // Evaluator::computeSeqLen: the first 1000 reads
    if (max_len <= 0) max_len = 151;
    if (const char* v = getenv("FASTP_GPU_MAX_LEN")) max_len = atoi(v);
    fill_params(o, paired, max_len, &S->B);
    if (fastp_gpu_host_create(&S->B.params, &S->B.ho, &S->host) != FASTP_GPU_OK) error_exit("fastp_gpu_host_create failed");
    fastp_gpu_stream_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.in1 = o->in1.c_str();
    cfg.in2 = paired && !o->interleavedInput ? o->in2.c_str() : NULL;
    cfg.interleaved = paired && o->interleavedInput ? 1 : 0;
    cfg.phred64 = o->phred64 ? 1 : 0;
    cfg.reads_to_process = o->readsToProcess;
    cfg.format.want_failed = S->B.ho.want_failed;
    cfg.format.want_unpaired1 = S->B.ho.want_unpaired1;
    cfg.format.want_unpaired2 = S->B.ho.want_unpaired2;
    cfg.format.umi_loc = S->B.ho.umi_loc;
    cfg.format.umi_len = S->B.ho.umi_len;
    cfg.format.umi_prefix = S->B.ho.umi_prefix;
    cfg.format.umi_delimiter = S->B.ho.umi_delimiter;
    const char* wm = getenv("FASTP_GPU_WRITER");
    const bool via_input = wm && std::string(wm) == "input";
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
        WriterThread* w = writers[q];
        S->writer[q] = w;
        cfg.out_fd[q] = -1;
        if (!w) continue;
        cfg.want[q] = 1;
        if (via_input) continue;   // text to WriterThread::input, which compresses ".gz" itself
        cfg.compress[q] = ends_with(w->getFilename(), ".gz") ? 1 : 0;
        cfg.out_fd[q] = w->mPwriteMode ? w->mFd : fileno(w->mWriter1->mFP);
    }
    S->writer[FASTP_GPU_OVERLAPPED] = overlapped_writer;   // its records are assembled on the host by the stream (peprocessor.cpp:488-495)
    cfg.want_overlapped = overlapped_writer && S->B.params.overlapped_out ? 1 : 0;
    cfg.emit = emit_to_writer;
    cfg.user = S;
    cfg.host = S->host;
    if (fastp_gpu_stream_create(&S->B.params, &cfg, &S->st) != FASTP_GPU_OK)
        error_exit(std::string("fastp_gpu_stream_create: ") + fastp_gpu_stream_last_error(NULL));   // never a silent CPU fallback
    fq::timeline("binding: stream_setup end");
    SG = S;
}

void stream_run() {
    StreamState* S = SG;
    if (fastp_gpu_stream_run(S->st) != FASTP_GPU_OK) error_exit(std::string("FASTP_GPU=1: ") + fastp_gpu_stream_last_error(S->st));
    fastp_gpu_stream_stats st;
    fastp_gpu_stream_get_stats(S->st, &st);
    // a pwrite-mode writer (".gz", several threads) cuts its file at the offset the last pack was written to
    // (WriterThread::setInputCompletedPwrite): tell it where the stream's bytes end
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
        WriterThread* w = S->writer[q];
        if (!w || !w->mPwriteMode || S->fed[q] > 0) continue;
        w->mOffsetRing[0].cumulative_offset.store((size_t)st.bytes_out[q], std::memory_order_relaxed);
        w->mOffsetRing[0].published_seq.store(0, std::memory_order_release);
        w->mNextSeq[0] = (size_t)S->W;   // "worker 0 wrote pack 0"
    }
    if (st.truncated) cerr << "WARNING: the input ended at a malformed FASTQ record; the reads in front of it were processed" << endl;
    if (getenv("FASTP_GPU_VERBOSE"))
        fprintf(stderr, "fastp_gpu: stream mode: %lld units in %lld chunks, %.3f s (setup %.3f, wait-read %.3f, parse %.3f, engine %.3f, format %.3f, deflate %.3f, "
                        "copies %.3f, wait-write %.3f; writer %.3f, adapter replay %.3f), max_len %d, %lld re-plan(s)\n",
                (long long)st.units, (long long)st.chunks, st.wall_s, st.setup_s, st.wait_read_s, st.parse_s, st.engine_s, st.format_s, st.deflate_s, st.d2h_s,
                st.wait_write_s, st.write_s, st.replay_s, (int)st.max_len, (long long)st.replans);
    if (getenv("FASTP_GPU_VERBOSE"))
        for (int m = 0; m < (S->paired ? 2 : 1); m++)
            if (st.input_kind[m])
                fprintf(stderr, "fastp_gpu: stream mode: input %d is %s: %lld bytes of the file -> %lld bytes of text (inflate + its copy to the host %.3f s)\n", m + 1,
                        st.input_kind[m] == 2 ? "BGZF, inflated on the device" : "gzip, inflated on host threads of the stream (fq_pgunzip.h; a pipe: one thread, fq_gunzip.h)", (long long)st.bytes_file[m],
                        (long long)st.bytes_in[m], st.inflate_s);
    S->ran = true;
}

// The run's results are in the reference's objects by now; what is left of the process is the reference's report writing and
// its exit.  Taking the stream apart (page-locked buffers: 34 ms) and then the HIP runtime's own exit handlers (~ 100 ms) is a
// quarter of a 4 M-pair run's wall clock (profiles/r06_r_dropin_probe.txt) and releases nothing that the end of the process
// does not release: once main() has returned - every output file closed, both reports written - the process leaves through
// _exit with main's status.  FASTP_GPU_FAST_EXIT=0: the orderly way.
void fast_exit_handler(int status, void*) {
    fq::timeline("binding: exit (reports written)");
    fflush(NULL);
    std::cout.flush();
    std::cerr.flush();
    _exit(status);
}

void stream_shutdown() {
    fq::timeline("binding: counters loaded into the reference's objects");
    static const bool fast = [] { const char* v = getenv("FASTP_GPU_FAST_EXIT"); return !(v && *v && atoi(v) == 0); }();
    if (fast && on_exit(fast_exit_handler, NULL) == 0) {   // (registered last: it runs before the runtime's handlers)
        SG = nullptr;   // the stream, the host object and their buffers stay as they are until the process ends
        return;
    }
    fastp_gpu_stream_destroy(SG->st);
    fastp_gpu_host_destroy(SG->host);
    delete SG;
    SG = nullptr;
}

bool stream_finish_pe(PairEndProcessor* pp, ThreadConfig** configs) {
    if (!SG) return false;
    fastp_gpu_counter_layout lay;
    fastp_gpu_stream_layout(SG->st, &lay);
    std::vector<int64_t> c((size_t)lay.total);
    if (fastp_gpu_stream_counters(SG->st, c.data(), lay.total) != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_stream_counters: ") + fastp_gpu_stream_last_error(SG->st));
    LAY = &lay;
    load_block(pp, configs, c, SG->B, true);
    load_adapters(configs, std::vector<fastp_gpu_host*>(1, SG->host), 1);
    for (int i = 0; i <= pp->mOptions->insertSizeMax; i++) pp->mInsertSizeHist[i] += (long)c[lay.isize + i];
    LAY = nullptr;
    stream_shutdown();
    return true;
}

bool stream_finish_se(SingleEndProcessor* sp, ThreadConfig** configs) {
    if (!SG) return false;
    fastp_gpu_counter_layout lay;
    fastp_gpu_stream_layout(SG->st, &lay);
    std::vector<int64_t> c((size_t)lay.total);
    if (fastp_gpu_stream_counters(SG->st, c.data(), lay.total) != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_stream_counters: ") + fastp_gpu_stream_last_error(SG->st));
    LAY = &lay;
    load_block(sp, configs, c, SG->B, false);
    load_adapters(configs, std::vector<fastp_gpu_host*>(1, SG->host), 1);
    LAY = nullptr;
    stream_shutdown();
    return true;
}

}  // namespace

// the reader-thread hooks (top of readerTask): 1 = stream mode took the run, -1 = the reference's reader runs (pack mode)
int fastp_gpu_stream_reader_pe(PairEndProcessor* pp, bool isLeft) {
    if (!stream_mode(pp->mOptions, true)) return -1;
    if (!isLeft) return 1;   // the read-1 reader thread drives both files
    WriterThread* const writers[FASTP_GPU_N_OUTPUTS] = {pp->mLeftWriter, pp->mRightWriter, pp->mFailedWriter, pp->mMergedWriter,
                                                        pp->mUnpairedLeftWriter, pp->mUnpairedRightWriter};
    stream_setup(pp->mOptions, true, writers, pp->mOverlappedWriter);
    stream_run();
    for (int t = 0; t < pp->mOptions->thread; t++) {   // what both reader threads do when their file is exhausted (peprocessor.cpp:866-872)
        pp->mLeftInputLists[t]->setProducerFinished();
        pp->mRightInputLists[t]->setProducerFinished();
    }
    pp->mBackpressureCV.notify_all();
    return 1;
}

// --interleaved_in: one reader thread (PairEndProcessor::interleavedReaderTask, src/peprocessor.cpp:890-1013) feeds both lists
int fastp_gpu_stream_reader_interleaved(PairEndProcessor* pp) {
    if (!stream_mode(pp->mOptions, true)) return -1;
    WriterThread* const writers[FASTP_GPU_N_OUTPUTS] = {pp->mLeftWriter, pp->mRightWriter, pp->mFailedWriter, pp->mMergedWriter,
                                                        pp->mUnpairedLeftWriter, pp->mUnpairedRightWriter};
    stream_setup(pp->mOptions, true, writers, pp->mOverlappedWriter);
    stream_run();
    for (int t = 0; t < pp->mOptions->thread; t++) {   // the task's own tail (:1001-1012)
        pp->mLeftInputLists[t]->setProducerFinished();
        pp->mRightInputLists[t]->setProducerFinished();
    }
    pp->mBackpressureCV.notify_all();
    pp->mLeftReaderFinished.store(true, std::memory_order_release);
    pp->mRightReaderFinished.store(true, std::memory_order_release);
    return 1;
}

int fastp_gpu_stream_reader_se(SingleEndProcessor* sp) {
    if (!stream_mode(sp->mOptions, false)) return -1;
    WriterThread* const writers[FASTP_GPU_N_OUTPUTS] = {sp->mLeftWriter, NULL, sp->mFailedWriter, NULL, NULL, NULL};
    stream_setup(sp->mOptions, false, writers);
    stream_run();
    for (int t = 0; t < sp->mOptions->thread; t++) sp->mInputLists[t]->setProducerFinished();
    sp->mBackpressureCV.notify_all();
    return 1;
}

void fastp_gpu_worker_finish_pe(PairEndProcessor* pp, ThreadConfig** configs) {
    if (stream_finish_pe(pp, configs)) return;
    if (!G) return;
    const std::vector<int64_t> c = fetch_counters();
    LAY = &G->lay;
    load_block(pp, configs, c, G->B, true);
    load_adapters(configs, G->hosts, pp->mOptions->thread);
    for (int i = 0; i <= pp->mOptions->insertSizeMax; i++) pp->mInsertSizeHist[i] += (long)c[G->lay.isize + i];
    shutdown();
}

void fastp_gpu_worker_finish_se(SingleEndProcessor* sp, ThreadConfig** configs) {
    if (stream_finish_se(sp, configs)) return;
    if (!G) return;
    const std::vector<int64_t> c = fetch_counters();
    LAY = &G->lay;
    load_block(sp, configs, c, G->B, false);
    load_adapters(configs, G->hosts, sp->mOptions->thread);
    shutdown();
}

// Evaluator::computeOverRepSeq (evaluator.cpp:78-169) with the counting on the device: the reads are taken from the
// file with the reference's own FastqReader and its own stopping rule, packed, and handed to fastp_gpu_eval_overrep
// (thresholds and the "remove substrings" pass are inside it).  A temporary single-end engine provides the context:
// the run's engine does not exist yet when the Evaluator runs.
int fastp_gpu_worker_overrep(const std::string& filename, std::map<std::string, long>& hotseqs, int seqlen) {
    if (!enabled() || seqlen < 1) return -1;
    std::vector<Read*> reads;
    {
        FastqReader reader(filename);
        const long BASE_LIMIT = 151 * 10000;
        long bases = 0;
        while (bases < BASE_LIMIT) {
            Read* r = reader.read();
            if (!r) break;
            bases += r->length();
            reads.push_back(r);
        }
    }
    auto drop = [&] { for (Read* r : reads) delete r; };
    const int n = (int)reads.size();
    int max_len = 1;
    for (Read* r : reads) max_len = std::max(max_len, r->length());
    if (n == 0 || max_len > 65535) { drop(); return -1; }
    const size_t ss = fastp_gpu_seq_stride(max_len), qs = fastp_gpu_qual_stride(max_len);
    std::vector<uint8_t> seq((size_t)n * ss), qual((size_t)n * qs);
    std::vector<uint16_t> len((size_t)n);
    {
        std::vector<const char*> sp((size_t)n), qp((size_t)n);
        std::vector<int32_t> ln((size_t)n);
        for (int i = 0; i < n; i++) { sp[i] = reads[i]->mSeq->data(); qp[i] = reads[i]->mQuality->data(); ln[i] = reads[i]->length(); }
        int32_t bad = -1;
        const int rc = fastp_gpu_pack_reads(max_len, n, sp.data(), qp.data(), ln.data(), seq.data(), qual.data(), len.data(), &bad);
        drop();
        if (rc != FASTP_GPU_OK) return -1;   // a letter outside ACGTN: the reference's own loop handles the file
    }
    fastp_gpu_params prm;
    fastp_gpu_default_params(&prm, 0, max_len);
    prm.dup_enabled = 0;   // no bloom bitmaps for this short-lived context
    fastp_gpu_ctx* ctx = nullptr;
    if (fastp_gpu_create(&prm, 0, &ctx) != FASTP_GPU_OK) return -1;
    void *d_seq = nullptr, *d_qual = nullptr, *d_len = nullptr;
    int rc = fastp_gpu_device_alloc(ctx, (int64_t)seq.size(), &d_seq);
    if (!rc) rc = fastp_gpu_device_alloc(ctx, (int64_t)qual.size(), &d_qual);
    if (!rc) rc = fastp_gpu_device_alloc(ctx, (int64_t)len.size() * 2, &d_len);
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_seq, seq.data(), (int64_t)seq.size());
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_qual, qual.data(), (int64_t)qual.size());
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_len, len.data(), (int64_t)len.size() * 2);
    int32_t n_seqs = 0;
    const int32_t max_seqs = 1 << 16;
    std::vector<char> text((size_t)8 << 20);
    std::vector<int64_t> off((size_t)max_seqs + 1), cnt((size_t)max_seqs);
    if (!rc)
        rc = fastp_gpu_eval_overrep(ctx, (const uint8_t*)d_seq, (const uint8_t*)d_qual, (const uint16_t*)d_len, n, seqlen, text.data(),
                                    (int64_t)text.size(), off.data(), cnt.data(), max_seqs, &n_seqs);
    fastp_gpu_device_free(ctx, d_seq);
    fastp_gpu_device_free(ctx, d_qual);
    fastp_gpu_device_free(ctx, d_len);
    fastp_gpu_destroy(ctx);
    if (rc != FASTP_GPU_OK) return -1;
    hotseqs.clear();
    for (int i = 0; i < n_seqs; i++) hotseqs[std::string(text.data() + off[i], (size_t)(off[i + 1] - off[i]))] = (long)cnt[i];
    if (getenv("FASTP_GPU_VERBOSE")) fprintf(stderr, "fastp_gpu: computeOverRepSeq on the device: %d reads, %d sequences\n", n, (int)n_seqs);
    return 1;
}

// The 4^10 ten-mer histogram of Evaluator::evalAdapterAndReadNum (evaluator.cpp:384-396) through
// fastp_gpu_eval_adapter_kmers: the reads are the ones the reference's own loading loop admitted (:326-341), the
// top-10 selection and the NucleotideTree walks that follow stay the reference's.  1 = counts filled, -1 = not handled
// (engine disabled, or a letter outside ACGTN: the reference's loop counts then).
// Duplicate::Duplicate (duplicate.cpp:46-52): with the engine on, Duplicate's bitmaps live in HBM and the reference's own
// checkPair / checkRead are never called - a token buffer instead of 1 .. 32 GiB allocated and cleared on the host
long fastp_gpu_worker_dup_bytes(long bytes) { return enabled() ? 64 : bytes; }

int fastp_gpu_reader_scan_eol(const char* buf, int from, int to) {
    if (!enabled() || from >= to) return from;
    const char* p = buf + from;
    const size_t n = (size_t)(to - from);
    const char* lf = (const char*)memchr(p, '\n', n);
    const char* cr = (const char*)memchr(p, '\r', lf ? (size_t)(lf - p) : n);   // a '\r' only counts in front of that '\n'
    const char* e = cr ? cr : lf;
    return e ? (int)(e - buf) : to;
}

int fastp_gpu_worker_adapter_kmers(Evaluator* ev, Read** reads, long records, int shiftTail, unsigned int* counts) {
    if (!enabled() || records <= 0 || records > (1 << 30)) return -1;
    const int n = (int)records;
    int max_len = 1;
    for (int i = 0; i < n; i++) max_len = std::max(max_len, reads[i]->length());
    if (max_len > 65535) return -1;
    const size_t ss = fastp_gpu_seq_stride(max_len), qs = fastp_gpu_qual_stride(max_len);
    std::vector<uint8_t> seq((size_t)n * ss), qual((size_t)n * qs);
    std::vector<uint16_t> len((size_t)n);
    {
        std::vector<const char*> sp((size_t)n), qp((size_t)n);
        std::vector<int32_t> ln((size_t)n);
        for (int i = 0; i < n; i++) { sp[i] = reads[i]->mSeq->data(); qp[i] = reads[i]->mQuality->data(); ln[i] = reads[i]->length(); }
        int32_t bad = -1;
        if (fastp_gpu_pack_reads(max_len, n, sp.data(), qp.data(), ln.data(), seq.data(), qual.data(), len.data(), &bad) != FASTP_GPU_OK) return -1;
    }
    fastp_gpu_params prm;
    fastp_gpu_default_params(&prm, 0, max_len);
    prm.dup_enabled = 0;   // no bloom bitmaps for this short-lived context
    fastp_gpu_ctx* ctx = nullptr;
    if (fastp_gpu_create(&prm, 0, &ctx) != FASTP_GPU_OK) return -1;
    void *d_seq = nullptr, *d_qual = nullptr, *d_len = nullptr, *d_cnt = nullptr;
    const int64_t cnt_bytes = (int64_t)4 << 20;
    int rc = fastp_gpu_device_alloc(ctx, (int64_t)seq.size(), &d_seq);
    if (!rc) rc = fastp_gpu_device_alloc(ctx, (int64_t)qual.size(), &d_qual);
    if (!rc) rc = fastp_gpu_device_alloc(ctx, (int64_t)len.size() * 2, &d_len);
    if (!rc) rc = fastp_gpu_device_alloc(ctx, cnt_bytes, &d_cnt);
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_seq, seq.data(), (int64_t)seq.size());
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_qual, qual.data(), (int64_t)qual.size());
    if (!rc) rc = fastp_gpu_device_upload(ctx, d_len, len.data(), (int64_t)len.size() * 2);
    int64_t used = 0;
    if (!rc) rc = fastp_gpu_eval_adapter_kmers(ctx, (const uint8_t*)d_seq, (const uint8_t*)d_qual, (const uint16_t*)d_len, n, shiftTail,
                                               (uint32_t*)d_cnt, &used);
    if (!rc && used != records) rc = FASTP_GPU_E_INVALID;   // the device admits exactly the reads the reference loaded
    if (!rc) rc = fastp_gpu_device_download(ctx, counts, d_cnt, cnt_bytes);
    fastp_gpu_device_free(ctx, d_seq);
    fastp_gpu_device_free(ctx, d_qual);
    fastp_gpu_device_free(ctx, d_len);
    fastp_gpu_device_free(ctx, d_cnt);
    fastp_gpu_destroy(ctx);
    if (rc != FASTP_GPU_OK) return -1;
    if (getenv("FASTP_GPU_VERBOSE")) fprintf(stderr, "fastp_gpu: evalAdapterAndReadNum's ten-mer histogram on the device: %d reads\n", n);
    if (getenv("FASTP_GPU_EVAL_CHECK")) {   // tests: the reference's own counting loop (Evaluator::seq2int) beside it
        std::vector<unsigned int> want((size_t)1 << 20, 0u);
        for (int i = 0; i < n; i++) {
            Read* r = reads[i];
            int key = -1;
            for (int pos = 20; pos <= r->length() - 10 - shiftTail; pos++) {
                key = ev->seq2int(r->mSeq, pos, 10, key);
                if (key >= 0) want[(size_t)key]++;
            }
        }
        want[0] = 0;
        size_t bad = 0, nonzero = 0;
        for (size_t k = 0; k < want.size(); k++) { bad += want[k] != counts[k]; nonzero += want[k] != 0; }
        fprintf(stderr, "fastp_gpu: ten-mer histogram check vs Evaluator::seq2int: %zu of %zu bins differ (%zu non-zero)\n", bad, want.size(), nonzero);
        if (bad) error_exit("FASTP_GPU_EVAL_CHECK: the device histogram differs from the reference's counting loop");
    }
    return 1;
}
