// fq_glue.cpp - include/fastp_gpu_host.h: the string side of the patched worker loop in C++.
// Host code only.  Mirrors fastp_amd/hostloop.py (apply_results, AdapterMaps, UmiNameEditor).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fastp_gpu_host.h"

namespace {

const int MAX_ADAPTER_REC = 20000;      // filterresult.cpp:7
const int LOW_COMPLEXITY_SKIP = 5000;   // filterresult.cpp:8

const char* failed_type(int code) {  // src/common.h:57-66
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

char complement(char c) {  // util.h:16-33: anything outside ACGTacgt -> 'N'
    switch (c) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 'N';
    }
}

static FILE* dump_file() {   // debugging aid: FASTP_GPU_DUMP_ADAPTERS=<path> logs every string handed to a map
    static FILE* f = [] { const char* p = getenv("FASTP_GPU_DUMP_ADAPTERS"); return p ? fopen(p, "w") : (FILE*)nullptr; }();
    return f;
}
struct AdapterMap {
    std::map<std::string, long> m;
    static bool low_complexity(const std::string& a) {  // filterresult.cpp:115-122
        int diff = 0;
        for (size_t i = 0; i + 1 < a.size(); i++) diff += a[i] != a[i + 1];
        return diff < (int)(a.size() / 2);
    }
    bool add(const std::string& a) {  // the per-map body of addAdapterTrimmed (:128-151)
        if (FILE* f = dump_file()) fprintf(f, "%p %s\n", (void*)this, a.c_str());
        auto it = m.find(a);
        if (it != m.end()) { it->second++; return true; }
        if ((int)m.size() > MAX_ADAPTER_REC || ((int)m.size() > LOW_COMPLEXITY_SKIP && low_complexity(a))) return false;
        m[a] = 1;
        return true;
    }
};

}  // namespace

struct fastp_gpu_host {
    fastp_gpu_params p;
    fastp_gpu_host_options o;
    std::string a1seq, a2seq, umi_prefix, umi_delim;
    std::vector<std::string> fasta;
    std::string out[FASTP_GPU_N_HOST_OUTPUTS];
    bool want[FASTP_GPU_N_HOST_OUTPUTS];
    AdapterMap amap[2];
    std::vector<std::map<std::string, long>::const_iterator> index[2];  // for adapter_entry()
    bool index_valid[2] = {false, false};

    // a string as (pointer, length): the pack's own buffers are routed without an intermediate copy
    struct View {
        const char* p;
        size_t n;
    };
    void record(int which, View name, const char* seq, const char* qual, int len, View strand,
                const char* tag = nullptr) {  // Read::appendToString / appendToStringWithTag read.cpp:119-154
        std::string& s = out[which];
        const size_t tl = tag ? strlen(tag) + 1 : 0;
        const size_t at = s.size();
        s.resize(at + name.n + tl + strand.n + 2 * (size_t)len + 4);
        char* w = &s[at];
        memcpy(w, name.p, name.n); w += name.n;
        if (tag) { *w++ = ' '; memcpy(w, tag, tl - 1); w += tl - 1; }
        *w++ = '\n';
        memcpy(w, seq, (size_t)len); w += len;
        *w++ = '\n';
        memcpy(w, strand.p, strand.n); w += strand.n;
        *w++ = '\n';
        memcpy(w, qual, (size_t)len); w += len;
        *w++ = '\n';
    }
    void record(int which, const std::string& name, const char* seq, const char* qual, int len, const std::string& strand,
                const char* tag = nullptr) {
        record(which, View{name.data(), name.size()}, seq, qual, len, View{strand.data(), strand.size()}, tag);
    }
    std::string umi_tagged(const std::string& name, const std::string& umi) const {  // addUmiToName umiprocessor.cpp:62-81
        std::string tag = umi_delim + (umi_prefix.empty() ? std::string() : umi_prefix + "_") + umi;
        const size_t sp = name.find(' ');
        return sp == std::string::npos ? name + tag : name.substr(0, sp) + tag + name.substr(sp);
    }
};

extern "C" {

int fastp_gpu_host_create(const fastp_gpu_params* params, const fastp_gpu_host_options* opts, fastp_gpu_host** out) {
    if (!params || !out) return FASTP_GPU_E_INVALID;
    fastp_gpu_host* h = new fastp_gpu_host();
    h->p = *params;
    memset(&h->o, 0, sizeof(h->o));
    if (opts) h->o = *opts;
    if (params->adapter_seq_r1) h->a1seq = params->adapter_seq_r1;
    if (params->adapter_seq_r2) h->a2seq = params->adapter_seq_r2;
    for (int i = 0; i < params->n_adapter_fasta && params->adapter_fasta; i++) h->fasta.push_back(params->adapter_fasta[i]);
    if (opts && opts->umi_prefix) h->umi_prefix = opts->umi_prefix;
    h->umi_delim = (opts && opts->umi_delimiter) ? opts->umi_delimiter : ":";
    h->p.adapter_seq_r1 = h->p.adapter_seq_r2 = nullptr;  // own copies above
    h->p.adapter_fasta = nullptr;
    h->p.overrep_seqs1 = h->p.overrep_seqs2 = nullptr;
    h->want[FASTP_GPU_OUT1] = true;
    h->want[FASTP_GPU_OUT2] = params->paired != 0;
    h->want[FASTP_GPU_FAILED] = opts && opts->want_failed;
    h->want[FASTP_GPU_MERGED] = params->merge != 0;
    h->want[FASTP_GPU_UNPAIRED1] = opts && opts->want_unpaired1;
    h->want[FASTP_GPU_UNPAIRED2] = opts && opts->want_unpaired2;
    h->want[FASTP_GPU_OVERLAPPED] = params->paired && params->overlapped_out;
    *out = h;
    return FASTP_GPU_OK;
}

void fastp_gpu_host_destroy(fastp_gpu_host* h) { delete h; }

const char* fastp_gpu_host_output(fastp_gpu_host* h, int which, size_t* len) {
    if (!h || which < 0 || which >= FASTP_GPU_N_HOST_OUTPUTS || !h->want[which]) { if (len) *len = 0; return nullptr; }
    if (len) *len = h->out[which].size();
    return h->out[which].data();
}

void fastp_gpu_host_clear_outputs(fastp_gpu_host* h) {
    if (h) for (auto& s : h->out) s.clear();
}

int64_t fastp_gpu_host_adapter_entries(fastp_gpu_host* h, int is_r2) {
    return h ? (int64_t)h->amap[is_r2 ? 1 : 0].m.size() : 0;
}

int fastp_gpu_host_adapter_entry(fastp_gpu_host* h, int is_r2, int64_t index, const char** seq, int32_t* len, int64_t* count) {
    if (!h) return FASTP_GPU_E_INVALID;
    const int k = is_r2 ? 1 : 0;
    auto& m = h->amap[k].m;
    if (!h->index_valid[k] || h->index[k].size() != m.size()) {
        h->index[k].clear();
        for (auto it = m.cbegin(); it != m.cend(); ++it) h->index[k].push_back(it);
        h->index_valid[k] = true;
    }
    if (index < 0 || index >= (int64_t)h->index[k].size()) return FASTP_GPU_E_INVALID;
    auto it = h->index[k][(size_t)index];
    if (seq) *seq = it->first.data();
    if (len) *len = (int32_t)it->first.size();
    if (count) *count = it->second;
    return FASTP_GPU_OK;
}

// FilterResult::addAdapterTrimmed for hosts that cut the adapter strings out of their own text (fastp_gpu_stream.h):
// the single-read form (filterresult.cpp:124-152) and the pair form, which skips read 2's string when read 1's was
// refused by a cap (:154-180, quirk #8)
int fastp_gpu_host_add_adapter(fastp_gpu_host* h, int is_r2, const char* a, int32_t len) {
    if (!h || len < 0 || (len && !a)) return FASTP_GPU_E_INVALID;
    if (len == 0) return FASTP_GPU_OK;
    h->index_valid[0] = h->index_valid[1] = false;
    h->amap[is_r2 ? 1 : 0].add(std::string(a, (size_t)len));
    return FASTP_GPU_OK;
}

int fastp_gpu_host_add_adapter_pair(fastp_gpu_host* h, const char* a1, int32_t len1, const char* a2, int32_t len2) {
    if (!h || len1 < 0 || len2 < 0 || (len1 && !a1) || (len2 && !a2)) return FASTP_GPU_E_INVALID;
    h->index_valid[0] = h->index_valid[1] = false;
    bool go = true;
    if (len1) go = h->amap[0].add(std::string(a1, (size_t)len1));
    if (go && len2) h->amap[1].add(std::string(a2, (size_t)len2));
    return FASTP_GPU_OK;
}

int fastp_gpu_host_apply(fastp_gpu_host* h, const fastp_gpu_reads* b1, const fastp_gpu_reads* b2, const fastp_gpu_results* res) {
    if (!h || !b1 || !res || !res->r1) return FASTP_GPU_E_INVALID;
    const bool paired = b2 != nullptr;
    if (paired != (h->p.paired != 0)) return FASTP_GPU_E_INVALID;
    if (paired && (b2->n != b1->n || !res->r2 || !res->pair)) return FASTP_GPU_E_INVALID;
    const int n = b1->n;
    h->index_valid[0] = h->index_valid[1] = false;
    // sparse edits / events of this pack, keyed by read (2*unit + mate for PE)
    std::unordered_map<uint32_t, std::vector<const fastp_gpu_correction*>> corr;
    if (res->corrections && res->n_corrections)
        for (int i = 0; i < *res->n_corrections; i++) corr[res->corrections[i].read].push_back(&res->corrections[i]);
    std::unordered_map<uint32_t, std::vector<const fastp_gpu_adapter_event*>> events;
    if (res->adapter_events && res->n_adapter_events) {
        for (int i = 0; i < *res->n_adapter_events; i++) events[res->adapter_events[i].read].push_back(&res->adapter_events[i]);
        for (auto& kv : events)  // the device emits them unordered; per read they apply in adapter order
            std::sort(kv.second.begin(), kv.second.end(),
                      [](const fastp_gpu_adapter_event* x, const fastp_gpu_adapter_event* y) { return x->adapter < y->adapter; });
    }
    // The common pack - no UMI, no base correction, no adapter string to replay, no merge - is routed straight from
    // the pack's buffers (pointer + length views, one memcpy per field into the writer's string); everything else
    // takes the general path below on copies it may edit.
    const bool simple_opts = h->o.umi_loc == FASTP_GPU_UMI_NONE && corr.empty() && events.empty() && !h->p.merge && !h->p.overlapped_out;
    std::string s1, q1, s2, q2, name1, name2, strand1, strand2;
    typedef fastp_gpu_host::View View;
    for (int i = 0; i < n; i++) {
        const fastp_gpu_read_result& rr1 = res->r1[i];
        const fastp_gpu_read_result* prr2 = paired ? &res->r2[i] : nullptr;
        if (simple_opts && !((rr1.flags | (paired ? prr2->flags : 0)) & (FASTP_GPU_RF_ADAPTER | FASTP_GPU_RF_ADAPTER_OV))) {
            const View n1{b1->name[i], (size_t)b1->name_len[i]}, st1{b1->strand[i], (size_t)b1->strand_len[i]};
            const char* t1s = b1->seq[i] + rr1.front; const char* t1q = b1->qual[i] + rr1.front;
            const int t1l = rr1.len, code1 = rr1.code;
            const bool dedup_out = h->p.dedup && (rr1.flags & FASTP_GPU_RF_DUP);
            const bool alive1 = !(rr1.flags & FASTP_GPU_RF_NULL);
            if (!paired) {  // seprocessor.cpp:280-290
                if (!dedup_out) {
                    if (alive1 && code1 == FASTP_PASS_FILTER) h->record(FASTP_GPU_OUT1, n1, t1s, t1q, t1l, st1);
                    else if (h->want[FASTP_GPU_FAILED]) h->record(FASTP_GPU_FAILED, n1, t1s, t1q, t1l, st1, failed_type(code1));
                }
                continue;
            }
            if (dedup_out) continue;
            const fastp_gpu_read_result& rr2 = *prr2;
            const View n2{b2->name[i], (size_t)b2->name_len[i]}, st2{b2->strand[i], (size_t)b2->strand_len[i]};
            const char* t2s = b2->seq[i] + rr2.front; const char* t2q = b2->qual[i] + rr2.front;
            const int t2l = rr2.len, code2 = rr2.code;
            const bool alive2 = !(rr2.flags & FASTP_GPU_RF_NULL);
            const bool p1 = alive1 && code1 == FASTP_PASS_FILTER, p2 = alive2 && code2 == FASTP_PASS_FILTER;
            const bool wf = h->want[FASTP_GPU_FAILED];
            if (p1 && p2) {  // peprocessor.cpp:577-594
                h->record(FASTP_GPU_OUT1, n1, t1s, t1q, t1l, st1);
                h->record(FASTP_GPU_OUT2, n2, t2s, t2q, t2l, st2);
            } else if (p1) {  // :595-605
                if (h->want[FASTP_GPU_UNPAIRED1]) {
                    h->record(FASTP_GPU_UNPAIRED1, n1, t1s, t1q, t1l, st1);
                    if (wf) h->record(FASTP_GPU_FAILED, n2, t2s, t2q, t2l, st2, failed_type(code2));
                } else if (wf) {
                    h->record(FASTP_GPU_FAILED, n1, t1s, t1q, t1l, st1, "paired_read_is_failing");
                    h->record(FASTP_GPU_FAILED, n2, t2s, t2q, t2l, st2, failed_type(code2));
                }
            } else if (p2) {  // :606-621
                if (h->want[FASTP_GPU_UNPAIRED2]) {
                    h->record(FASTP_GPU_UNPAIRED2, n2, t2s, t2q, t2l, st2);
                    if (wf) h->record(FASTP_GPU_FAILED, n1, t1s, t1q, t1l, st1, failed_type(code1));
                } else if (h->want[FASTP_GPU_UNPAIRED1]) {
                    h->record(FASTP_GPU_UNPAIRED1, n2, t2s, t2q, t2l, st2);
                    if (wf) h->record(FASTP_GPU_FAILED, n1, t1s, t1q, t1l, st1, failed_type(code1));
                } else if (wf) {
                    h->record(FASTP_GPU_FAILED, n1, t1s, t1q, t1l, st1, failed_type(code1));
                    h->record(FASTP_GPU_FAILED, n2, t2s, t2q, t2l, st2, "paired_read_is_failing");
                }
            }
            continue;
        }
        name1.assign(b1->name[i], (size_t)b1->name_len[i]);
        strand1.assign(b1->strand[i], (size_t)b1->strand_len[i]);
        s1.assign(b1->seq[i], (size_t)b1->len[i]);
        q1.assign(b1->qual[i], (size_t)b1->len[i]);
        if (paired) {
            name2.assign(b2->name[i], (size_t)b2->name_len[i]);
            strand2.assign(b2->strand[i], (size_t)b2->strand_len[i]);
            s2.assign(b2->seq[i], (size_t)b2->len[i]);
            q2.assign(b2->qual[i], (size_t)b2->len[i]);
        }
        // UMI name edit on the ORIGINAL reads (umiprocessor.cpp:19-61), before anything is routed
        if (h->o.umi_loc != FASTP_GPU_UMI_NONE) {
            std::string umi;
            bool tag = true;
            const size_t ul = (size_t)std::max(0, h->o.umi_len);
            if (h->o.umi_loc == FASTP_GPU_UMI_READ1) umi = s1.substr(0, ul);
            else if (h->o.umi_loc == FASTP_GPU_UMI_READ2) { if (paired) umi = s2.substr(0, ul); else tag = false; }
            else { umi = s1.substr(0, ul); if (paired) umi += "_" + s2.substr(0, ul); }
            if (h->o.umi_loc != FASTP_GPU_UMI_PER_READ && umi.empty()) tag = false;
            if (tag) { name1 = h->umi_tagged(name1, umi); if (paired) name2 = h->umi_tagged(name2, umi); }
        }
        // BaseCorrector edits (basecorrector.cpp:39-57)
        auto apply_corr = [&](uint32_t key, std::string& s, std::string& q) {
            auto it = corr.find(key);
            if (it == corr.end()) return;
            for (const fastp_gpu_correction* c : it->second)
                if (c->pos < s.size()) { s[c->pos] = (char)c->base; q[c->pos] = (char)c->qual; }
        };
        apply_corr(paired ? 2u * (uint32_t)i : (uint32_t)i, s1, q1);
        if (paired) apply_corr(2u * (uint32_t)i + 1u, s2, q2);
        // FilterResult::addAdapterTrimmed replay, in input order
        auto adapter_string = [&](const fastp_gpu_read_result& rr, const std::string& s, const std::string& aseq) {
            if (rr.adapter_pos < 0) return aseq.substr(0, rr.adapter_len);
            return s.substr((size_t)rr.front + (size_t)rr.adapter_pos, rr.adapter_len);
        };
        auto replay_fasta = [&](uint32_t key, const fastp_gpu_read_result& rr, const std::string& s, int is_r2) {
            auto it = events.find(key);  // trimByMultiSequences adaptertrimmer.cpp:48-62
            if (it == events.end()) return;
            for (const fastp_gpu_adapter_event* e : it->second) {
                std::string a = e->pos < 0 ? h->fasta[e->adapter].substr(0, e->len)
                                           : s.substr((size_t)rr.front + (size_t)e->pos, e->len);
                if (!a.empty()) h->amap[is_r2].add(a);
            }
        };
        if (paired) {
            const fastp_gpu_read_result& rr2 = *prr2;
            if (rr1.flags & FASTP_GPU_RF_ADAPTER_OV) {  // addAdapterTrimmed(a1, a2) filterresult.cpp:154-180, quirk #8
                const std::string a1 = adapter_string(rr1, s1, h->a1seq), a2 = adapter_string(rr2, s2, h->a2seq);
                bool go = true;
                if (!a1.empty()) go = h->amap[0].add(a1);
                if (go && !a2.empty()) h->amap[1].add(a2);
            } else {
                if ((rr1.flags & FASTP_GPU_RF_ADAPTER) && rr1.adapter_len) { std::string a = adapter_string(rr1, s1, h->a1seq); if (!a.empty()) h->amap[0].add(a); }
                if ((rr2.flags & FASTP_GPU_RF_ADAPTER) && rr2.adapter_len) { std::string a = adapter_string(rr2, s2, h->a2seq); if (!a.empty()) h->amap[1].add(a); }
            }
            replay_fasta(2u * (uint32_t)i, rr1, s1, 0);       // peprocessor.cpp:467-470
            replay_fasta(2u * (uint32_t)i + 1u, rr2, s2, 1);
        } else {
            if ((rr1.flags & FASTP_GPU_RF_ADAPTER) && rr1.adapter_len) { std::string a = adapter_string(rr1, s1, h->a1seq); if (!a.empty()) h->amap[0].add(a); }
            replay_fasta((uint32_t)i, rr1, s1, 0);            // seprocessor.cpp:249-251
        }
        // the trimmed reads: only prefix / suffix removal (filter.cpp:199-202, read.cpp:62-67)
        const char* t1s = s1.data() + rr1.front; const char* t1q = q1.data() + rr1.front;
        const int t1l = rr1.len;
        const bool dedup_out = h->p.dedup && (rr1.flags & FASTP_GPU_RF_DUP);
        const int code1 = rr1.code;
        const bool alive1 = !(rr1.flags & FASTP_GPU_RF_NULL);
        if (!paired) {  // seprocessor.cpp:280-290
            if (!dedup_out) {
                if (alive1 && code1 == FASTP_PASS_FILTER) h->record(FASTP_GPU_OUT1, name1, t1s, t1q, t1l, strand1);
                else if (h->want[FASTP_GPU_FAILED]) h->record(FASTP_GPU_FAILED, name1, t1s, t1q, t1l, strand1, failed_type(code1));
            }
            continue;
        }
        const fastp_gpu_read_result& rr2 = *prr2;
        const char* t2s = s2.data() + rr2.front; const char* t2q = q2.data() + rr2.front;
        const int t2l = rr2.len;
        const int code2 = rr2.code;
        const bool alive2 = !(rr2.flags & FASTP_GPU_RF_NULL);
        if (h->want[FASTP_GPU_OVERLAPPED] && (rr1.reserved & FASTP_GPU_OVOUT_HIT)) {  // peprocessor.cpp:488-495
            // string(substr(max(0, offset)), overlap_len) is std::string's (str, pos) constructor: the reference prints
            // the bases of read 1 BEHIND the overlapped region.  The engine analysed the pair as it was right after
            // adapter trimming; the later polyX / max_len cuts do not shorten what is printed here.
            const int pos = rr1.reserved & 0x7FFF, cnt = rr2.reserved;
            h->record(FASTP_GPU_OVERLAPPED, name1, t1s + pos, t1q + pos, cnt, strand1);
        }
        if (h->p.merge && alive1 && alive2) {  // peprocessor.cpp:518-561
            if (res->pair[i].flags & FASTP_GPU_PF_OVERLAPPED) {
                if (code1 == FASTP_PASS_FILTER) {  // OverlapAnalysis::merge overlapanalysis.cpp:148-179
                    // the part lengths: in the records' reserved fields, unless --overlapped_out occupies them - then
                    // from the pair record (merge mode's own analysis): len1 = overlap_len + max(0, offset), len2 =
                    // offset > 0 ? len(r2') - overlap_len : 0 (overlapanalysis.cpp:152-156)
                    const int ol = res->pair[i].ov_len, off = res->pair[i].ov_offset;
                    const int m1 = h->p.overlapped_out ? ol + (off > 0 ? off : 0) : rr1.reserved;
                    const int m2 = h->p.overlapped_out ? (off > 0 ? t2l - ol : 0) : rr2.reserved;
                    std::string ms(t1s, (size_t)m1), mq(t1q, (size_t)m1);
                    for (int k = 0; k < m2; k++) {  // rc(r2')[ol + k] = comp(r2'[len2 - 1 - ol - k])
                        ms.push_back(complement(t2s[t2l - 1 - ol - k]));
                        mq.push_back(t2q[t2l - 1 - ol - k]);
                    }
                    const std::string tag = " merged_" + std::to_string(m1) + "_" + std::to_string(m2);
                    const std::string strand = strand1 == "+" ? strand1 : strand1 + tag;
                    h->record(FASTP_GPU_MERGED, name1 + tag, ms.data(), mq.data(), (int)ms.size(), strand);
                }
                continue;
            }
            if (h->p.merge_include_unmerged) {
                if (code1 == FASTP_PASS_FILTER && !dedup_out) h->record(FASTP_GPU_MERGED, name1, t1s, t1q, t1l, strand1);
                if (code2 == FASTP_PASS_FILTER && !dedup_out) h->record(FASTP_GPU_MERGED, name2, t2s, t2q, t2l, strand2);
                continue;
            }
        }
        if (dedup_out) continue;
        const bool p1 = alive1 && code1 == FASTP_PASS_FILTER, p2 = alive2 && code2 == FASTP_PASS_FILTER;
        const bool wf = h->want[FASTP_GPU_FAILED];
        if (p1 && p2) {  // :577-594
            h->record(FASTP_GPU_OUT1, name1, t1s, t1q, t1l, strand1);
            h->record(FASTP_GPU_OUT2, name2, t2s, t2q, t2l, strand2);
        } else if (p1) {  // :595-605
            if (h->want[FASTP_GPU_UNPAIRED1]) {
                h->record(FASTP_GPU_UNPAIRED1, name1, t1s, t1q, t1l, strand1);
                if (wf) h->record(FASTP_GPU_FAILED, name2, t2s, t2q, t2l, strand2, failed_type(code2));
            } else if (wf) {
                h->record(FASTP_GPU_FAILED, name1, t1s, t1q, t1l, strand1, "paired_read_is_failing");
                h->record(FASTP_GPU_FAILED, name2, t2s, t2q, t2l, strand2, failed_type(code2));
            }
        } else if (p2) {  // :606-621
            if (h->want[FASTP_GPU_UNPAIRED2]) {
                h->record(FASTP_GPU_UNPAIRED2, name2, t2s, t2q, t2l, strand2);
                if (wf) h->record(FASTP_GPU_FAILED, name1, t1s, t1q, t1l, strand1, failed_type(code1));
            } else if (h->want[FASTP_GPU_UNPAIRED1]) {
                h->record(FASTP_GPU_UNPAIRED1, name2, t2s, t2q, t2l, strand2);
                if (wf) h->record(FASTP_GPU_FAILED, name1, t1s, t1q, t1l, strand1, failed_type(code1));
            } else if (wf) {
                h->record(FASTP_GPU_FAILED, name1, t1s, t1q, t1l, strand1, failed_type(code1));
                h->record(FASTP_GPU_FAILED, name2, t2s, t2q, t2l, strand2, "paired_read_is_failing");
            }
        }
    }
    return FASTP_GPU_OK;
}

}  // extern "C"
