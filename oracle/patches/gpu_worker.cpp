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
#include "gpu_worker.h"

namespace {

struct State {
    std::mutex mu;               // one engine, packs submitted one at a time (stream order = submission order)
    fastp_gpu_ctx* ctx = nullptr;
    fastp_gpu_params params;
    fastp_gpu_counter_layout lay;
    bool paired = false;
    int max_len = 0;
    std::vector<fastp_gpu_host*> hosts;              // per worker thread: output strings + adapter maps
    std::vector<std::string> seeds[2], fasta;        // own copies of the strings the parameter block points at
    std::vector<const char*> seedp[2], fastap;
    std::string a1, a2, umi_prefix;
    bool warned_fallback = false;
};
State* G = nullptr;
std::once_flag g_once;

bool enabled() {
    static const int e = [] { const char* v = getenv("FASTP_GPU"); return (v && *v) ? atoi(v) : 0; }();
    return e != 0;
}

void refuse(const char* what) { error_exit(std::string("FASTP_GPU=1: ") + what + " is outside the engine's scope"); }

// Options (already validated, Evaluator results applied) -> the engine's flat parameter block
void make_state(Options* o, bool paired) {
    State* s = new State();
    s->paired = paired;
    if (o->indexFilter.enabled) refuse("--filter_by_index");
    if (o->fixMGI) refuse("--fix_mgi_id");
    if (o->split.enabled) refuse("--split");
    if (!o->overlappedOut.empty()) refuse("--overlapped_out");
    if (o->outputToSTDOUT) refuse("--stdout");
    s->max_len = std::max(o->seqLen1, paired ? o->seqLen2 : 0);
    if (s->max_len <= 0) s->max_len = 151;
    if (s->max_len > FASTP_GPU_MAX_READ_LEN) refuse("reads longer than FASTP_GPU_MAX_READ_LEN");
    fastp_gpu_params& p = s->params;
    fastp_gpu_default_params(&p, paired ? 1 : 0, s->max_len);
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
    int rc = fastp_gpu_create(&p, 0, &s->ctx);
    if (rc != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_create: ") + fastp_gpu_last_error(NULL));   // never a silent CPU fallback
    fastp_gpu_counter_layout_for_params(&p, &s->lay);
    fastp_gpu_host_options ho;
    memset(&ho, 0, sizeof(ho));
    ho.want_failed = !o->failedOut.empty();
    ho.want_unpaired1 = !o->unpaired1.empty();
    ho.want_unpaired2 = !o->unpaired2.empty() && o->unpaired2 != o->unpaired1;
    if (o->umi.enabled) {
        ho.umi_loc = o->umi.location == UMI_LOC_READ1 ? FASTP_GPU_UMI_READ1
                   : o->umi.location == UMI_LOC_READ2 ? FASTP_GPU_UMI_READ2 : FASTP_GPU_UMI_PER_READ;
        ho.umi_len = o->umi.length;
        s->umi_prefix = o->umi.prefix;
        ho.umi_prefix = s->umi_prefix.empty() ? NULL : s->umi_prefix.c_str();
        ho.umi_delimiter = o->umi.delimiter.empty() ? NULL : o->umi.delimiter.c_str();
    }
    for (int t = 0; t < o->thread; t++) {
        fastp_gpu_host* h = NULL;
        if (fastp_gpu_host_create(&p, &ho, &h) != FASTP_GPU_OK) error_exit("fastp_gpu_host_create failed");
        s->hosts.push_back(h);
    }
    G = s;
}

// per worker thread scratch: the packed pack, the records
struct Scratch {
    std::vector<const char*> name[2], seq[2], qual[2], strand[2];
    std::vector<int32_t> name_len[2], len[2], strand_len[2];
    std::vector<uint8_t> pseq[2], pqual[2];
    std::vector<uint16_t> plen[2];
    std::vector<fastp_gpu_read_result> rr[2];
    std::vector<fastp_gpu_pair_result> pr;
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

bool pack(int n, int m) {
    const size_t ss = fastp_gpu_seq_stride(G->max_len), qs = fastp_gpu_qual_stride(G->max_len);
    T.pseq[m].resize((size_t)n * ss); T.pqual[m].resize((size_t)n * qs); T.plen[m].resize(n);
    int32_t bad = -1;
    return fastp_gpu_pack_reads(G->max_len, n, T.seq[m].data(), T.qual[m].data(), T.len[m].data(), T.pseq[m].data(),
                                T.pqual[m].data(), T.plen[m].data(), &bad) == FASTP_GPU_OK;
}

fastp_gpu_reads reads_of(int n, int m) {
    fastp_gpu_reads r;
    r.n = n;
    r.name = T.name[m].data(); r.name_len = T.name_len[m].data();
    r.seq = T.seq[m].data(); r.qual = T.qual[m].data(); r.len = T.len[m].data();
    r.strand = T.strand[m].data(); r.strand_len = T.strand_len[m].data();
    return r;
}

// submit one pack and apply its records; false = not handled
bool run_pack(int tid, int n, bool paired, bool thread0) {
    for (int m = 0; m < (paired ? 2 : 1); m++)
        if (!pack(n, m)) {
            if (!G->warned_fallback) {
                G->warned_fallback = true;
                fprintf(stderr, "FASTP_GPU: a pack holds reads the engine refuses (alphabet / length); such packs run through the CPU loop\n");
            }
            return false;
        }
    for (int m = 0; m < (paired ? 2 : 1); m++) T.rr[m].assign(n, fastp_gpu_read_result());
    T.pr.assign(paired ? n : 0, fastp_gpu_pair_result());
    T.corr.resize(G->params.correction ? (size_t)n * 64 + 16 : 0);
    T.ev.resize(G->params.n_adapter_fasta ? (size_t)n * 2 * std::min(G->params.n_adapter_fasta, 8) + 16 : 0);
    int32_t ncorr = 0, nev = 0;
    fastp_gpu_batch b;
    memset(&b, 0, sizeof(b));
    b.n = n;
    b.flags = thread0 ? FASTP_GPU_BATCH_STAT_ISIZE : 0u;   // statInsertSize runs on worker thread 0 only (peprocessor.cpp:449)
    b.seq1 = T.pseq[0].data(); b.qual1 = T.pqual[0].data(); b.len1 = T.plen[0].data();
    if (paired) { b.seq2 = T.pseq[1].data(); b.qual2 = T.pqual[1].data(); b.len2 = T.plen[1].data(); }
    fastp_gpu_results res;
    memset(&res, 0, sizeof(res));
    res.r1 = T.rr[0].data();
    if (paired) { res.r2 = T.rr[1].data(); res.pair = T.pr.data(); }
    res.corrections = T.corr.empty() ? NULL : T.corr.data();
    res.corrections_capacity = (int32_t)T.corr.size();
    res.n_corrections = &ncorr;
    res.adapter_events = T.ev.empty() ? NULL : T.ev.data();
    res.adapter_events_capacity = (int32_t)T.ev.size();
    res.n_adapter_events = &nev;
    {
        std::lock_guard<std::mutex> lk(G->mu);
        const int rc = fastp_gpu_submit_host(G->ctx, &b, &res);
        if (rc != FASTP_GPU_OK) error_exit(std::string("fastp_gpu_submit_host: ") + fastp_gpu_last_error(G->ctx));
    }
    fastp_gpu_reads r1 = reads_of(n, 0), r2;
    if (paired) r2 = reads_of(n, 1);
    fastp_gpu_host_clear_outputs(G->hosts[tid]);
    if (fastp_gpu_host_apply(G->hosts[tid], &r1, paired ? &r2 : NULL, &res) != FASTP_GPU_OK) error_exit("fastp_gpu_host_apply failed");
    return true;
}

std::string* take(int tid, int which) {
    size_t len = 0;
    const char* s = fastp_gpu_host_output(G->hosts[tid], which, &len);
    return new std::string(s ? s : "", s ? len : 0);
}

// the engine's counter block added onto one Stats object (the per-cycle part has Stats::mCycleBuffer's layout)
void load_stats(Stats* st, const std::vector<int64_t>& c, int slot, const std::vector<std::string>& seeds) {
    const fastp_gpu_counter_layout& L = G->lay;
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
    const fastp_gpu_counter_layout& L = G->lay;
    for (int i = 0; i < FILTER_RESULT_TYPES; i++) fr->mFilterReadStats[i] += c[L.filter_stats + i];
    fr->mTrimmedAdapterRead += c[L.adapter_reads];
    fr->mTrimmedAdapterBases += c[L.adapter_bases];
    for (int i = 0; i < 4; i++) { fr->mTrimmedPolyXReads[i] += c[L.polyx_reads + i]; fr->mTrimmedPolyXBases[i] += c[L.polyx_bases + i]; }
    for (int i = 0; i < 64; i++) fr->mCorrectionMatrix[i] += c[L.correction + i];
    fr->mCorrectedReads += c[L.corrected_reads];
    fr->mMergedPairs += c[L.merged_pairs];
}

void load_adapters(ThreadConfig** configs, int threads) {
    for (int t = 0; t < threads && t < (int)G->hosts.size(); t++)
        for (int m = 0; m < 2; m++) {
            const int64_t n = fastp_gpu_host_adapter_entries(G->hosts[t], m);
            auto& dst = m ? configs[t]->getFilterResult()->mAdapter2 : configs[t]->getFilterResult()->mAdapter1;
            for (int64_t i = 0; i < n; i++) {
                const char* s; int32_t len; int64_t cnt;
                fastp_gpu_host_adapter_entry(G->hosts[t], m, i, &s, &len, &cnt);
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
    for (auto* h : G->hosts) fastp_gpu_host_destroy(h);
    fastp_gpu_destroy(G->ctx);
    delete G;
    G = nullptr;
}

}  // namespace

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
    gather(left->data, n, 0);
    gather(right->data, n, 1);
    if (!run_pack(tid, n, true, tid == 0)) return -1;
    // hand the strings to the writer threads exactly as peprocessor.cpp:644-686 does
    if (pp->mMergedWriter) pp->mMergedWriter->input(tid, take(tid, FASTP_GPU_MERGED));
    if (pp->mFailedWriter) pp->mFailedWriter->input(tid, take(tid, FASTP_GPU_FAILED));
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
    for (int i = 0; i < left->count; i++) pp->recycleToPool1(tid, left->data[i]);
    for (int i = 0; i < right->count; i++) pp->recycleToPool2(tid, right->data[i]);
    config->markProcessed(left->count);
    delete[] left->data;
    delete[] right->data;
    delete left;
    delete right;
    pp->mPackProcessedCounter.fetch_add(1, std::memory_order_release);
    pp->mBackpressureCV.notify_all();
    return 1;
}

int fastp_gpu_worker_se(SingleEndProcessor* sp, ReadPack* pack, ThreadConfig* config) {
    if (!enabled()) return -1;
    Options* o = sp->mOptions;
    std::call_once(g_once, [&] { make_state(o, false); });
    const int tid = config->getThreadId();
    const int n = pack->count;
    gather(pack->data, n, 0);
    if (!run_pack(tid, n, false, false)) return -1;
    if (sp->mLeftWriter) sp->mLeftWriter->input(tid, take(tid, FASTP_GPU_OUT1));     // seprocessor.cpp:299-304
    if (sp->mFailedWriter) sp->mFailedWriter->input(tid, take(tid, FASTP_GPU_FAILED));
    for (int i = 0; i < n; i++) sp->recycleToPool(tid, pack->data[i]);
    config->markProcessed(pack->count);
    delete pack->data;
    delete pack;
    sp->mPackProcessedCounter.fetch_add(1, std::memory_order_release);
    sp->mBackpressureCV.notify_all();
    return 1;
}

void fastp_gpu_worker_finish_pe(PairEndProcessor* pp, ThreadConfig** configs) {
    if (!G) return;
    const std::vector<int64_t> c = fetch_counters();
    load_stats(configs[0]->getPreStats1(), c, FASTP_GPU_STATS_PRE1, G->seeds[0]);
    load_stats(configs[0]->getPostStats1(), c, FASTP_GPU_STATS_POST1, G->seeds[0]);
    load_stats(configs[0]->getPreStats2(), c, FASTP_GPU_STATS_PRE2, G->seeds[1]);
    load_stats(configs[0]->getPostStats2(), c, FASTP_GPU_STATS_POST2, G->seeds[1]);
    load_filter_result(configs[0]->getFilterResult(), c);
    load_adapters(configs, pp->mOptions->thread);
    if (pp->mDuplicate) {
        pp->mDuplicate->mTotalReads += (unsigned long)c[G->lay.dup_total];
        pp->mDuplicate->mDupReads += (unsigned long)c[G->lay.dup_count];
    }
    for (int i = 0; i <= pp->mOptions->insertSizeMax; i++) pp->mInsertSizeHist[i] += (long)c[G->lay.isize + i];
    shutdown();
}

void fastp_gpu_worker_finish_se(SingleEndProcessor* sp, ThreadConfig** configs) {
    if (!G) return;
    const std::vector<int64_t> c = fetch_counters();
    load_stats(configs[0]->getPreStats1(), c, FASTP_GPU_STATS_PRE1, G->seeds[0]);
    load_stats(configs[0]->getPostStats1(), c, FASTP_GPU_STATS_POST1, G->seeds[0]);
    load_filter_result(configs[0]->getFilterResult(), c);
    load_adapters(configs, sp->mOptions->thread);
    if (sp->mDuplicate) {
        sp->mDuplicate->mTotalReads += (unsigned long)c[G->lay.dup_total];
        sp->mDuplicate->mDupReads += (unsigned long)c[G->lay.dup_count];
    }
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
