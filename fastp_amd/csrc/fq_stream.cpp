// fq_stream.cpp - include/fastp_gpu_stream.h: FASTQ files -> output streams around the engine, one host loop in C++.
//
// The loop a GPU-enabled fastp runs in place of its reader threads + worker threads + the string side of its writers
// (readerTask src/peprocessor.cpp:725-888 / src/seprocessor.cpp:327-442, FastqReader::read src/fastqreader.cpp:240-368,
// processorTask :1021-1033, the output strings of :652-686, WriterThread::inputPwrite src/writerthread.cpp:118-168).
// Host code; the per-read work is the device entry points of fastp_gpu.h.  Four threads + two small I/O pools:
//
//   reader thread   pread pieces of the next chunk of each file into page-locked memory, H2D copy on its own stream
//                   (".gz" inputs, FastqReader::init src/fastqreader.cpp:169-199: a bgzip-written file - isBgzf,
//                   src/bgzf.h:17-27 - is shipped COMPRESSED, cut at member boundaries by fastp_gpu_bgzf_index, and
//                   inflated on the device by the caller thread, which stands in for BgzfMtReader src/bgzf.h:36-239;
//                   any other gzip stream is inflated here, by several threads per file (fq_pgunzip.h; a pipe: one thread,
//                   fq_gunzip.h), where FastqReader::readToBufIgzip src/fastqreader.cpp:88-149 has ISA-L on the reference's
//                   one reader thread)
//   caller thread   parse -> worker loop -> format (-> deflate) on the context's stream, D2H of the output text
//   writer thread   pwrite pieces of a chunk's output (or the emit callback), in chunk order
//   replay thread   FilterResult::addAdapterTrimmed for the reads the records flag, in input order
//
// The text of a chunk that the records of the trip do not cover (a partial record at the end, the surplus of the
// mate whose records are shorter) is copied to the front of the other page-locked slot and the next trip's fresh bytes
// are read in behind it: a slot always holds the whole text of its trip ([carried | fresh], at most chunk_bytes), which
// is what the adapter replay cuts its strings from.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/fastp_gpu_stream.h"
#include "fq_pgunzip.h"
#include "fq_timeline.h"

namespace {

thread_local std::string g_stream_error;

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// isBgzf (src/bgzf.h:17-27) on the file's first bytes; FastqReader::init looks at the name first (".gz")
// 0 plain text, 1 gzip, 2 BGZF (SRC_* below)
int input_kind(const std::string& path) {
    if (path.size() < 3 || path.compare(path.size() - 3, 3, ".gz") != 0) return 0;
    struct stat sb;
    if (stat(path.c_str(), &sb) == 0 && !S_ISREG(sb.st_mode)) return 1;   // a pipe cannot be looked into twice: the host inflater takes any gzip stream
    unsigned char h[18];
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return 1;   // the run reports the open error
    const ssize_t n = pread(fd, h, 18, 0);
    close(fd);
    if (n == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 0x42 && h[13] == 0x43 && (h[14] | (h[15] << 8)) == 2) return 2;
    return 1;
}

// a few threads that run positional reads / writes; wait() returns when everything submitted so far is done
class IoPool {
  public:
    explicit IoPool(int n) {
        for (int i = 0; i < std::max(1, n); i++) th_.emplace_back([this] { work(); });
    }
    ~IoPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); pending_++; }
        cv_.notify_one();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

  private:
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
            { std::lock_guard<std::mutex> lk(mu_); pending_--; }
            done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<std::function<void()>> q_;
    int pending_ = 0;
    bool stop_ = false;
};

template <class T>
class Channel {
  public:
    void put(T v) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(v)); }
        cv_.notify_one();
    }
    T get() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        return v;
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
};

// a chunk's read is cut into pieces for the reader's threads (FASTP_GPU_STREAM_READ_PIECE_KB; 8 MiB pieces used 4 of the 16
// threads on a 16 MiB chunk per file)
const int64_t IO_PIECE = (int64_t)std::max(64, env_int("FASTP_GPU_STREAM_READ_PIECE_KB", 2048)) << 10;
const unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

struct ReadReq {
    int slot = -1;                 // -1: stop
    int64_t carry[2] = {0, 0};     // bytes already at the front of the slot's page-locked text
    int64_t budget[2] = {0, 0};    // fresh bytes wanted behind them
};
struct ReadDone {
    int slot = 0;
    int64_t nb[2] = {0, 0};        // fresh TEXT bytes behind the carried ones (of a BGZF file: what its blocks inflate to)
    bool eof[2] = {false, false};
    int err = 0;                   // 1 file I/O, 2 host-to-device copy, 4 a gzip stream is damaged, 5 a BGZF file is damaged
    int32_t n_blocks = 0;          // BGZF members of this trip (both files), indexed in d_idx[slot], bytes in d_comp[slot]
    int64_t file_bytes[2] = {0, 0};
};
enum { SRC_PLAIN = 0, SRC_GZIP = 1, SRC_BGZF = 2 };
struct WriteJob {
    int oslot = -1;                // -1: stop; 3: only the host-built stream below (a trip whose device streams are all unwanted)
    int64_t len[FASTP_GPU_N_OUTPUTS] = {0, 0, 0, 0, 0, 0};
    bool copy_pending = false;     // the device-to-host copy into pin_out[oslot] is in flight: wait for ev_out[oslot] first
    bool has_ov = false;           // --overlapped_out's records of this chunk (may be empty: the writer takes one string per chunk)
    std::string ov;
};
// the adapter strings of one chunk, in input order: entries [kind u8][len1 u16][len2 u16][bytes1][bytes2]
struct ReplayJob {
    bool stop = false;
    std::vector<uint8_t> blob;
};
enum { RP_SINGLE_R1 = 0, RP_SINGLE_R2 = 1, RP_PAIR = 2 };

}  // namespace

struct fastp_gpu_stream {
    fastp_gpu_params p;
    fastp_gpu_stream_config cfg;
    std::string in1, in2, a1, a2, umi_prefix, umi_delim;
    std::vector<std::string> fasta, seeds[2];
    std::vector<const char*> fastap, seedp[2];
    fastp_gpu_ctx* ctx = nullptr;
    bool paired = false;
    bool interleaved = false;      // both mates' records alternate in ONE file (--interleaved_in: FastqReaderPair::read src/fastqreader.cpp:470-478)
    int nm = 1;                    // mates of a unit
    int nf = 1;                    // input files (= nm unless interleaved)
    std::string err;
    fastp_gpu_stream_stats st;
    // counter blocks of the contexts a re-plan replaced
    std::vector<std::pair<fastp_gpu_counter_layout, std::vector<int64_t>>> segments;

    // ---- buffers ------------------------------------------------------------------------------------------
    int64_t chunk = 0, text_cap = 0;
    int32_t max_records = 0;
    uint8_t* d_text[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [slot][mate]; the mates of a slot are one allocation
    uint8_t* pin_in[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    // compressed inputs (src_kind per file: SRC_*)
    int src_kind[2] = {SRC_PLAIN, SRC_PLAIN};
    bool any_bgzf = false;
    int64_t comp_cap = 0;                            // compressed bytes of one file per trip (staging + device)
    int32_t max_blocks = 0;                          // BGZF members of one file per trip
    uint8_t* pin_comp[2] = {nullptr, nullptr};       // [mate]: the file's bytes not yet handed to the device, from the front
    uint8_t* d_comp[2] = {nullptr, nullptr};         // [slot]: mate m's members at m * comp_cap
    uint8_t* pin_idx[2] = {nullptr, nullptr};        // [slot]: pay_off | pay_len | isize | crc (u32 each) | out_off (u64), n entries each
    uint8_t* d_idx[2] = {nullptr, nullptr};
    uint8_t *d_seq[2] = {nullptr, nullptr}, *d_qual[2] = {nullptr, nullptr};
    uint16_t* d_len[2] = {nullptr, nullptr};
    uint32_t *d_loff[2] = {nullptr, nullptr}, *d_llen[2] = {nullptr, nullptr};
    // interleaved input: the parser's output for the file's records (2 x max_records), dealt out to the mates' arrays above
    uint8_t *il_seq = nullptr, *il_qual = nullptr;
    uint16_t* il_len = nullptr;
    uint32_t *il_loff = nullptr, *il_llen = nullptr;
    fastp_gpu_read_result* d_res[2] = {nullptr, nullptr};
    fastp_gpu_pair_result* d_pair = nullptr;
    fastp_gpu_correction* d_corr = nullptr;
    fastp_gpu_adapter_event* d_ev = nullptr;
    int32_t *d_nc = nullptr, *d_nev = nullptr;
    int32_t corr_cap = 0, ev_cap = 0;
    uint8_t* d_zero = nullptr;
    // the output streams' text (and its gzip members) on the device, one set per page-locked output slot: the copy of chunk k
    // to the host runs on its own stream under the parser / worker loop / formatter of chunk k + 1 (round 5; the caller's
    // thread used to wait for it - 2 of the 6.5 s of a 100 M-pair run, profiles/r04_dropin_100M_reader_ab.txt)
    uint8_t* d_out[2][FASTP_GPU_N_OUTPUTS] = {};
    uint8_t* d_gz[2][FASTP_GPU_N_OUTPUTS] = {};
    hipStream_t cp_out = nullptr;
    hipEvent_t ev_out[2] = {nullptr, nullptr};   // the copy into pin_out[slot] has landed (the writer thread waits for it)
    int64_t out_cap[FASTP_GPU_N_OUTPUTS] = {}, gz_cap[FASTP_GPU_N_OUTPUTS] = {};
    uint8_t* pin_out[2][FASTP_GPU_N_OUTPUTS] = {};
    // host copies for the adapter replay
    fastp_gpu_read_result* h_res[2] = {nullptr, nullptr};
    uint32_t* h_loff[2] = {nullptr, nullptr};
    fastp_gpu_correction* h_corr = nullptr;
    fastp_gpu_adapter_event* h_ev = nullptr;
    int32_t* h_counts = nullptr;   // pinned: n_corrections, n_adapter_events
    hipStream_t sx = nullptr, cp_in = nullptr;
    bool any_out = false;

    int fail(int code, const std::string& m) {
        err = m;
        g_stream_error = m;
        return code;
    }
    int fail_ctx(int code, const char* what) {
        return fail(code, std::string(what) + ": " + fastp_gpu_last_error(ctx));
    }
};

namespace {

#define S_HIP(s, call)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return (s)->fail(FASTP_GPU_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

void free_buffers(fastp_gpu_stream* s) {
    auto dfree = [](void* p) { if (p) (void)hipFree(p); };
    auto hfree = [](void* p) { if (p) (void)hipHostFree(p); };
    for (int sl = 0; sl < 2; sl++)
        for (int m = 0; m < 2; m++) {
            if (m == 0) dfree(s->d_text[sl][0]);   // mate 2's text lies behind mate 1's in the same allocation
            s->d_text[sl][m] = nullptr;
            hfree(s->pin_in[sl][m]);
            s->pin_in[sl][m] = nullptr;
        }
    for (int k = 0; k < 2; k++) {
        hfree(s->pin_comp[k]); dfree(s->d_comp[k]); hfree(s->pin_idx[k]); dfree(s->d_idx[k]);
        s->pin_comp[k] = s->d_comp[k] = s->pin_idx[k] = s->d_idx[k] = nullptr;
    }
    for (int m = 0; m < 2; m++) {
        dfree(s->d_seq[m]); dfree(s->d_qual[m]); dfree(s->d_len[m]); dfree(s->d_loff[m]); dfree(s->d_llen[m]); dfree(s->d_res[m]);
        s->d_seq[m] = s->d_qual[m] = nullptr; s->d_len[m] = nullptr; s->d_loff[m] = s->d_llen[m] = nullptr; s->d_res[m] = nullptr;
        hfree(s->h_res[m]); hfree(s->h_loff[m]);
        s->h_res[m] = nullptr; s->h_loff[m] = nullptr;
    }
    dfree(s->il_seq); dfree(s->il_qual); dfree(s->il_len); dfree(s->il_loff); dfree(s->il_llen);
    s->il_seq = s->il_qual = nullptr; s->il_len = nullptr; s->il_loff = s->il_llen = nullptr;
    dfree(s->d_pair); dfree(s->d_corr); dfree(s->d_ev); dfree(s->d_nc); dfree(s->d_nev); dfree(s->d_zero);
    s->d_pair = nullptr; s->d_corr = nullptr; s->d_ev = nullptr; s->d_nc = s->d_nev = nullptr; s->d_zero = nullptr;
    hfree(s->h_corr); hfree(s->h_ev); hfree(s->h_counts);
    s->h_corr = nullptr; s->h_ev = nullptr; s->h_counts = nullptr;
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
        if (s->d_out[1][q] == s->d_out[0][q]) s->d_out[1][q] = nullptr;   // (an unwanted stream's two sets are one buffer)
        for (int sl = 0; sl < 2; sl++) {
            dfree(s->d_out[sl][q]); dfree(s->d_gz[sl][q]);
            s->d_out[sl][q] = s->d_gz[sl][q] = nullptr;
        }
        for (int sl = 0; sl < 2; sl++) { hfree(s->pin_out[sl][q]); s->pin_out[sl][q] = nullptr; }
    }
    for (int sl = 0; sl < 2; sl++)
        if (s->ev_out[sl]) { (void)hipEventDestroy(s->ev_out[sl]); s->ev_out[sl] = nullptr; }
    if (s->cp_out) { (void)hipStreamDestroy(s->cp_out); s->cp_out = nullptr; }
    if (s->sx) { (void)hipStreamDestroy(s->sx); s->sx = nullptr; }
    if (s->cp_in) { (void)hipStreamDestroy(s->cp_in); s->cp_in = nullptr; }
}

// the packed rows depend on max_len: (re)allocated for every context
int alloc_rows(fastp_gpu_stream* s) {
    const size_t ss = fastp_gpu_seq_stride(s->p.max_len), qs = fastp_gpu_qual_stride(s->p.max_len);
    for (int m = 0; m < s->nm; m++) {
        if (s->d_seq[m]) (void)hipFree(s->d_seq[m]);
        if (s->d_qual[m]) (void)hipFree(s->d_qual[m]);
        s->d_seq[m] = s->d_qual[m] = nullptr;
        S_HIP(s, hipMalloc((void**)&s->d_seq[m], (size_t)s->max_records * ss));
        S_HIP(s, hipMalloc((void**)&s->d_qual[m], (size_t)s->max_records * qs));
    }
    if (s->interleaved) {
        if (s->il_seq) (void)hipFree(s->il_seq);
        if (s->il_qual) (void)hipFree(s->il_qual);
        s->il_seq = s->il_qual = nullptr;
        S_HIP(s, hipMalloc((void**)&s->il_seq, 2 * (size_t)s->max_records * ss));
        S_HIP(s, hipMalloc((void**)&s->il_qual, 2 * (size_t)s->max_records * qs));
    }
    return FASTP_GPU_OK;
}

int alloc_buffers(fastp_gpu_stream* s) {
    S_HIP(s, hipSetDevice(s->cfg.device));
    S_HIP(s, hipStreamCreateWithFlags(&s->sx, hipStreamNonBlocking));
    S_HIP(s, hipStreamCreateWithFlags(&s->cp_in, hipStreamNonBlocking));
    S_HIP(s, hipStreamCreateWithFlags(&s->cp_out, hipStreamNonBlocking));
    for (int sl = 0; sl < 2; sl++) S_HIP(s, hipEventCreateWithFlags(&s->ev_out[sl], hipEventDisableTiming));
    const int nm = s->nm, nf = s->nf;
    s->text_cap = (s->chunk + 4096 + 255) / 256 * 256;   // a trip's text never exceeds chunk bytes (see the loop)
    s->max_records = (int32_t)std::max<int64_t>(1024, s->chunk / 32);
    if (s->interleaved) s->max_records = (s->max_records + 1) / 2;   // units: the file's text holds two records per unit
    for (int sl = 0; sl < 2; sl++) {
        // one allocation per slot: a single fastp_gpu_inflate_bgzf launch writes the text of both files
        S_HIP(s, hipMalloc((void**)&s->d_text[sl][0], (size_t)(nf * s->text_cap)));
        for (int m = 0; m < nf; m++) {
            s->d_text[sl][m] = s->d_text[sl][0] + (size_t)m * (size_t)s->text_cap;
            S_HIP(s, hipHostMalloc((void**)&s->pin_in[sl][m], (size_t)s->text_cap));
        }
    }
    if (s->any_bgzf) {
        // a member holds at most 64 KiB of text and is at worst a stored block + 26 bytes of framing, so the members
        // that fill a trip's text are never larger than this; one more member may sit incomplete at the end
        s->comp_cap = (s->chunk + s->chunk / 512 + (1 << 17) + 255) / 256 * 256;
        s->max_blocks = (int32_t)std::min<int64_t>(s->chunk / 2048 + 64, 1 << 20);
        const size_t idx_bytes = (size_t)nf * (size_t)s->max_blocks * 24 + 64;
        for (int m = 0; m < nf; m++)
            if (s->src_kind[m] == SRC_BGZF) S_HIP(s, hipHostMalloc((void**)&s->pin_comp[m], (size_t)s->comp_cap + 64));
        for (int sl = 0; sl < 2; sl++) {
            S_HIP(s, hipMalloc((void**)&s->d_comp[sl], (size_t)(nf * s->comp_cap) + 64));
            S_HIP(s, hipMemsetAsync(s->d_comp[sl], 0, (size_t)(nf * s->comp_cap) + 64, s->sx));
            S_HIP(s, hipHostMalloc((void**)&s->pin_idx[sl], idx_bytes));
            S_HIP(s, hipMalloc((void**)&s->d_idx[sl], idx_bytes));
        }
    }
    S_HIP(s, hipMalloc((void**)&s->d_zero, 64));
    S_HIP(s, hipMemsetAsync(s->d_zero, 0, 64, s->sx));
    for (int m = 0; m < nm; m++) {
        S_HIP(s, hipMalloc((void**)&s->d_len[m], (size_t)s->max_records * 2));
        S_HIP(s, hipMalloc((void**)&s->d_loff[m], (size_t)s->max_records * 16));
        S_HIP(s, hipMalloc((void**)&s->d_llen[m], (size_t)s->max_records * 16));
        S_HIP(s, hipMalloc((void**)&s->d_res[m], (size_t)s->max_records * sizeof(fastp_gpu_read_result)));
        S_HIP(s, hipMemsetAsync(s->d_res[m], 0, (size_t)s->max_records * sizeof(fastp_gpu_read_result), s->sx));
    }
    if (s->interleaved) {
        S_HIP(s, hipMalloc((void**)&s->il_len, 2 * (size_t)s->max_records * 2));
        S_HIP(s, hipMalloc((void**)&s->il_loff, 2 * (size_t)s->max_records * 16));
        S_HIP(s, hipMalloc((void**)&s->il_llen, 2 * (size_t)s->max_records * 16));
    }
    int rc = alloc_rows(s);
    if (rc) return rc;
    if (s->paired) S_HIP(s, hipMalloc((void**)&s->d_pair, (size_t)s->max_records * sizeof(fastp_gpu_pair_result)));
    S_HIP(s, hipMalloc((void**)&s->d_nc, 16));
    S_HIP(s, hipMalloc((void**)&s->d_nev, 16));
    S_HIP(s, hipMemsetAsync(s->d_nc, 0, 16, s->sx));
    S_HIP(s, hipMemsetAsync(s->d_nev, 0, 16, s->sx));
    if (s->p.correction) {
        s->corr_cap = 1 << 22;
        S_HIP(s, hipMalloc((void**)&s->d_corr, (size_t)s->corr_cap * sizeof(fastp_gpu_correction)));
    }
    if (s->p.n_adapter_fasta) {
        s->ev_cap = (int32_t)std::min<int64_t>((int64_t)s->max_records * 2 * std::min(s->p.n_adapter_fasta, 8) + 16, (int64_t)1 << 28);
        S_HIP(s, hipMalloc((void**)&s->d_ev, (size_t)s->ev_cap * sizeof(fastp_gpu_adapter_event)));
    }
    // what a record can grow by over its input text: the UMI tag on the name (delimiter + prefix + '_' + the UMI of one or
    // both mates joined by '_', UmiProcessor::addUmiToName), the failed / merged tags on the name and the strand line
    int64_t grow = 0;
    if (s->cfg.format.umi_loc != FASTP_GPU_UMI_NONE)
        grow = (int64_t)s->umi_delim.size() + (s->umi_prefix.empty() ? 0 : (int64_t)s->umi_prefix.size() + 1) + 2 * (int64_t)s->cfg.format.umi_len + 1;
    const int64_t both = nf * s->text_cap + (int64_t)s->max_records * (96 + 2 * grow);   // every record of both mates + tags
    const int64_t one = s->text_cap + (int64_t)s->max_records * grow + 64;              // out1 / out2: records only shrink, but for the UMI tag
    const int64_t caps[FASTP_GPU_N_OUTPUTS] = {one, one, both, both, both, both};
    s->any_out = false;
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) s->any_out = s->any_out || s->cfg.want[q];
    // fastp_gpu_format_streams wants a buffer for every stream the options can route to
    const bool need[FASTP_GPU_N_OUTPUTS] = {true, s->paired, s->cfg.format.want_failed != 0, s->paired && s->p.merge,
                                            s->paired && s->cfg.format.want_unpaired1, s->paired && s->cfg.format.want_unpaired2};
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
        if (!s->any_out || !(need[q] || s->cfg.want[q])) continue;
        s->out_cap[q] = caps[q];
        // (a stream nobody asked for is written by the formatter and never copied: its two sets are one buffer)
        for (int sl = 0; sl < 2; sl++) {
            if (sl == 1 && !s->cfg.want[q]) { s->d_out[1][q] = s->d_out[0][q]; continue; }
            S_HIP(s, hipMalloc((void**)&s->d_out[sl][q], (size_t)caps[q]));
        }
        int64_t host_bytes = caps[q];
        if (s->cfg.want[q] && s->cfg.compress[q]) {
            s->gz_cap[q] = caps[q] + 31 * (caps[q] / 65280 + 1) + 64;
            for (int sl = 0; sl < 2; sl++) S_HIP(s, hipMalloc((void**)&s->d_gz[sl][q], (size_t)s->gz_cap[q]));
            host_bytes = s->gz_cap[q];
        }
        if (s->cfg.want[q])
            for (int sl = 0; sl < 2; sl++) S_HIP(s, hipHostMalloc((void**)&s->pin_out[sl][q], (size_t)host_bytes));
    }
    if (s->cfg.host || s->cfg.want_overlapped) {
        for (int m = 0; m < nm; m++) {
            S_HIP(s, hipHostMalloc((void**)&s->h_res[m], (size_t)s->max_records * sizeof(fastp_gpu_read_result)));
            S_HIP(s, hipHostMalloc((void**)&s->h_loff[m], (size_t)s->max_records * 16));
        }
        if (s->corr_cap) S_HIP(s, hipHostMalloc((void**)&s->h_corr, (size_t)s->corr_cap * sizeof(fastp_gpu_correction)));
        if (s->ev_cap) S_HIP(s, hipHostMalloc((void**)&s->h_ev, (size_t)s->ev_cap * sizeof(fastp_gpu_adapter_event)));
    }
    S_HIP(s, hipHostMalloc((void**)&s->h_counts, 64));
    S_HIP(s, hipStreamSynchronize(s->sx));
    return FASTP_GPU_OK;
}

// dst (layout dl) += src (layout sl): same options, only the per-cycle capacity differs (dl.cycles >= sl.cycles)
void add_counters(const fastp_gpu_counter_layout& dl, std::vector<int64_t>& dst, const fastp_gpu_counter_layout& sl,
                  const std::vector<int64_t>& src, int insert_size_max) {
    auto add = [&](int64_t d, int64_t s0, int64_t n) { for (int64_t i = 0; i < n; i++) dst[(size_t)(d + i)] += src[(size_t)(s0 + i)]; };
    add(dl.filter_stats, sl.filter_stats, FASTP_FILTER_RESULT_TYPES);
    add(dl.adapter_reads, sl.adapter_reads, 1);
    add(dl.adapter_bases, sl.adapter_bases, 1);
    add(dl.polyx_reads, sl.polyx_reads, 4);
    add(dl.polyx_bases, sl.polyx_bases, 4);
    add(dl.correction, sl.correction, 64);
    add(dl.corrected_reads, sl.corrected_reads, 1);
    add(dl.merged_pairs, sl.merged_pairs, 1);
    add(dl.dup_total, sl.dup_total, 1);
    add(dl.dup_count, sl.dup_count, 1);
    add(dl.isize, sl.isize, (int64_t)insert_size_max + 1);
    for (int k = 0; k < 4; k++) {
        const int64_t d = dl.stats[k], s0 = sl.stats[k];
        add(d + dl.st_reads, s0 + sl.st_reads, 1);
        add(d + dl.st_length_sum, s0 + sl.st_length_sum, 1);
        add(d + dl.st_qual_hist, s0 + sl.st_qual_hist, 128);
        add(d + dl.st_kmer, s0 + sl.st_kmer, 1024);
        for (int a = 0; a < 34; a++) add(d + dl.st_cycle + a * dl.cycles, s0 + sl.st_cycle + a * sl.cycles, sl.cycles);
        add(dl.overrep_count[k], sl.overrep_count[k], sl.n_overrep[k]);
        add(dl.overrep_dist[k], sl.overrep_dist[k], sl.n_overrep[k] * sl.eval_len[k]);
    }
}

int fetch_ctx_counters(fastp_gpu_stream* s, fastp_gpu_counter_layout* lay, std::vector<int64_t>* c) {
    fastp_gpu_counter_layout_for_params(&s->p, lay);
    c->assign((size_t)lay->total, 0);
    if (fastp_gpu_synchronize(s->ctx) != FASTP_GPU_OK) return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_synchronize");
    if (fastp_gpu_counters(s->ctx, c->data(), lay->total) != FASTP_GPU_OK) return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_counters");
    return FASTP_GPU_OK;
}

// a read of `needed` bases has turned up: the work so far is kept (counters folded on the host, duplicate bitmaps and
// stream positions carried over) and the run goes on with a context sized for it
int replan(fastp_gpu_stream* s, int needed) {
    if (needed > FASTP_GPU_MAX_READ_LEN)
        return s->fail(FASTP_GPU_E_TOO_LONG, "a read of " + std::to_string(needed) + " bases is longer than FASTP_GPU_MAX_READ_LEN");
    int target = std::max(needed, s->p.max_len + s->p.max_len / 4);   // some headroom: re-plans are rare but not free
    target = std::min<int>(FASTP_GPU_MAX_READ_LEN, (target + 7) / 8 * 8);
    fastp_gpu_counter_layout lay;
    std::vector<int64_t> c;
    int rc = fetch_ctx_counters(s, &lay, &c);
    if (rc) return rc;
    const int64_t post_reads = c[(size_t)(lay.stats[FASTP_GPU_STATS_POST1] + lay.st_reads)];
    s->segments.emplace_back(lay, std::move(c));
    void* image = nullptr;
    const int64_t image_bytes = fastp_gpu_dup_bitmap_bytes(s->ctx);
    if (image_bytes > 0) {
        S_HIP(s, hipMalloc(&image, (size_t)image_bytes));
        if (fastp_gpu_dup_bitmap_export(s->ctx, image) != FASTP_GPU_OK) { (void)hipFree(image); return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_dup_bitmap_export"); }
    }
    fastp_gpu_destroy(s->ctx);
    s->ctx = nullptr;
    s->p.max_len = target;
    rc = fastp_gpu_create(&s->p, s->cfg.device, &s->ctx);
    if (rc != FASTP_GPU_OK) { if (image) (void)hipFree(image); return s->fail(rc, std::string("fastp_gpu_create (re-plan): ") + fastp_gpu_last_error(nullptr)); }
    if (s->cfg.want_overlapped) (void)fastp_gpu_host_writes_overlapped(s->ctx, 1);
    if (image) {
        rc = fastp_gpu_dup_bitmap_import(s->ctx, image);
        (void)hipFree(image);
        if (rc != FASTP_GPU_OK) return s->fail_ctx(rc, "fastp_gpu_dup_bitmap_import");
    }
    if (fastp_gpu_stream_set_origin(s->ctx, s->st.units, post_reads) != FASTP_GPU_OK) return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_stream_set_origin");
    s->st.replans++;
    s->st.max_len = target;
    return alloc_rows(s);
}

// ---- FilterResult::addAdapterTrimmed: the strings of one chunk, cut out of the chunk's text on the host ---------------
struct Extractor {
    fastp_gpu_stream* s;
    std::unordered_map<uint32_t, std::vector<const fastp_gpu_correction*>> corr;
    std::unordered_map<uint32_t, std::vector<const fastp_gpu_adapter_event*>> events;
    std::string tmp[2];

    // the read as the trimmer saw it: BaseCorrector's edits applied (basecorrector.cpp:39-57)
    const char* read_text(int m, int i, uint32_t key, const uint8_t* text, int32_t len_hint, int* out_len) {
        const uint32_t off = s->h_loff[m][4 * (size_t)i + 1];
        const char* base = (const char*)text + off;
        *out_len = len_hint;
        auto it = corr.find(key);
        if (it == corr.end()) return base;
        tmp[m].assign(base, (size_t)len_hint);
        for (const fastp_gpu_correction* c : it->second)
            if (c->pos < tmp[m].size()) tmp[m][c->pos] = (char)c->base;
        return tmp[m].data();
    }
    static void put(std::vector<uint8_t>& blob, int kind, const char* a, size_t la, const char* b, size_t lb) {
        const size_t at = blob.size();
        blob.resize(at + 5 + la + lb);
        uint8_t* w = blob.data() + at;
        w[0] = (uint8_t)kind;
        w[1] = (uint8_t)(la & 0xFF); w[2] = (uint8_t)(la >> 8);
        w[3] = (uint8_t)(lb & 0xFF); w[4] = (uint8_t)(lb >> 8);
        if (la) memcpy(w + 5, a, la);
        if (lb) memcpy(w + 5 + la, b, lb);
    }
    // the string handed to addAdapterTrimmed for one record (fastp_gpu_read_result::adapter_pos / adapter_len)
    static void cut(const fastp_gpu_read_result& rr, const char* read, int read_len, const std::string& aseq, const char** a, size_t* la) {
        if (rr.adapter_pos < 0) {
            *a = aseq.data();
            *la = std::min<size_t>(aseq.size(), rr.adapter_len);
            return;
        }
        const size_t from = (size_t)rr.front + (size_t)rr.adapter_pos;
        if (from >= (size_t)read_len) { *a = read; *la = 0; return; }
        *a = read + from;
        *la = std::min<size_t>(rr.adapter_len, (size_t)read_len - from);
    }

    // the chunk's sparse lists by read
    void index(int32_t ncorr, int32_t nev) {
        corr.clear();
        events.clear();
        for (int32_t i = 0; i < ncorr; i++) corr[s->h_corr[i].read].push_back(&s->h_corr[i]);
        for (int32_t i = 0; i < nev; i++) events[s->h_ev[i].read].push_back(&s->h_ev[i]);
        for (auto& kv : events)   // the device emits them unordered; per read they apply in adapter order
            std::sort(kv.second.begin(), kv.second.end(),
                      [](const fastp_gpu_adapter_event* x, const fastp_gpu_adapter_event* y) { return x->adapter < y->adapter; });
    }

    // --overlapped_out's stream (src/peprocessor.cpp:488-495): for a pair the third analysis finds overlapped, a record with
    // read 1's name (after the UMI edit), strand line, and the bases / qualities of read 1 - as BaseCorrector left them - that
    // the reference's string(substr(max(0, offset)), overlap_len) prints: std::string's (str, pos) constructor, i.e. what lies
    // BEHIND the overlapped region.  The engine leaves first position | FASTP_GPU_OVOUT_HIT and count in the records.
    void overlapped(int n, const uint8_t* const text[2], std::string& out) {
        auto line_end = [](const char* p, const char* lim) { while (p < lim && *p != '\r' && *p != '\n') p++; return p; };
        const fastp_gpu_format_options& fo = s->cfg.format;
        for (int i = 0; i < n; i++) {
            const fastp_gpu_read_result& r1 = s->h_res[0][i];
            if (!(r1.reserved & FASTP_GPU_OVOUT_HIT)) continue;
            const fastp_gpu_read_result& r2 = s->h_res[1][i];
            const size_t pos = (size_t)(r1.reserved & 0x7FFF) + (size_t)r1.front, cnt = r2.reserved;
            const uint32_t* lo = s->h_loff[0] + 4 * (size_t)i;
            const char* t = (const char*)text[0];
            const char* name = t + lo[0];
            const char* name_end = line_end(name, t + lo[1]);
            const char* strand = t + lo[2];
            const char* strand_end = line_end(strand, t + lo[3]);
            // the name after UmiProcessor::process (src/umiprocessor.cpp:19-81): the tag goes in front of the first space
            if (fo.umi_loc != FASTP_GPU_UMI_NONE) {
                const size_t ul = (size_t)std::max(0, fo.umi_len);
                const char* s1 = t + lo[1];
                const size_t l1 = (size_t)(line_end(s1, t + lo[2]) - s1);
                const uint32_t* lo2 = s->h_loff[1] + 4 * (size_t)i;
                const char* s2 = (const char*)text[1] + lo2[1];
                const size_t l2 = (size_t)(line_end(s2, (const char*)text[1] + lo2[2]) - s2);
                std::string umi;
                if (fo.umi_loc == FASTP_GPU_UMI_READ1) umi.assign(s1, std::min(ul, l1));
                else if (fo.umi_loc == FASTP_GPU_UMI_READ2) umi.assign(s2, std::min(ul, l2));
                else { umi.assign(s1, std::min(ul, l1)); umi += "_"; umi.append(s2, std::min(ul, l2)); }
                if (fo.umi_loc == FASTP_GPU_UMI_PER_READ || !umi.empty()) {
                    const char* sp = name;
                    while (sp < name_end && *sp != ' ') sp++;
                    out.append(name, (size_t)(sp - name));
                    out += s->umi_delim;
                    if (!s->umi_prefix.empty()) { out += s->umi_prefix; out += '_'; }
                    out += umi;
                    out.append(sp, (size_t)(name_end - sp));
                } else {
                    out.append(name, (size_t)(name_end - name));
                }
            } else {
                out.append(name, (size_t)(name_end - name));
            }
            out += '\n';
            const size_t at_seq = out.size();
            out.append(t + lo[1] + pos, cnt);
            out += '\n';
            out.append(strand, (size_t)(strand_end - strand));
            out += '\n';
            const size_t at_qual = out.size();
            out.append(t + lo[3] + pos, cnt);
            if (s->cfg.phred64)   // the host's text is the file's: convertPhred64To33 as the device did for everything else
                for (size_t k = at_qual; k < out.size(); k++) out[k] = (char)std::max(33, (int)(unsigned char)out[k] - 31);
            out += '\n';
            auto it = corr.find(2u * (uint32_t)i);   // BaseCorrector's edits of read 1 (base and quality, basecorrector.cpp:39-57)
            if (it != corr.end())
                for (const fastp_gpu_correction* c : it->second)
                    if (c->pos >= pos && c->pos < pos + cnt) {
                        out[at_seq + (c->pos - pos)] = (char)c->base;
                        out[at_qual + (c->pos - pos)] = (char)c->qual;
                    }
        }
    }

    void run(int n, const uint8_t* const text[2], std::vector<uint8_t>& blob) {
        const bool paired = s->paired;
        const int32_t nev = (int32_t)events.size();
        const uint8_t flagmask = FASTP_GPU_RF_ADAPTER | FASTP_GPU_RF_ADAPTER_OV;
        for (int i = 0; i < n; i++) {
            const fastp_gpu_read_result& r1 = s->h_res[0][i];
            const fastp_gpu_read_result* r2 = paired ? &s->h_res[1][i] : nullptr;
            const uint32_t k1 = paired ? 2u * (uint32_t)i : (uint32_t)i, k2 = 2u * (uint32_t)i + 1u;
            const bool ev1 = nev && events.count(k1), ev2 = paired && nev && events.count(k2);
            if (!((r1.flags | (r2 ? r2->flags : 0)) & flagmask) && !ev1 && !ev2) continue;
            // line lengths are not downloaded: a sequence line reaches to the strand line's offset minus its terminator;
            // front + len of the record bound what is cut, so the distance to the next line is a safe upper bound
            int l1 = 0, l2 = 0;
            const int cap1 = (int)(s->h_loff[0][4 * (size_t)i + 2] - s->h_loff[0][4 * (size_t)i + 1]);
            const char* t1 = read_text(0, i, k1, text[0], cap1, &l1);
            const char* t2 = nullptr;
            if (paired) {
                const int cap2 = (int)(s->h_loff[1][4 * (size_t)i + 2] - s->h_loff[1][4 * (size_t)i + 1]);
                t2 = read_text(1, i, k2, text[1], cap2, &l2);
            }
            const char *x1 = nullptr, *x2 = nullptr;
            size_t n1 = 0, n2 = 0;
            if (paired) {
                if (r1.flags & FASTP_GPU_RF_ADAPTER_OV) {   // addAdapterTrimmed(a1, a2) filterresult.cpp:154-180
                    cut(r1, t1, l1, s->a1, &x1, &n1);
                    cut(*r2, t2, l2, s->a2, &x2, &n2);
                    put(blob, RP_PAIR, x1, n1, x2, n2);
                } else {
                    if ((r1.flags & FASTP_GPU_RF_ADAPTER) && r1.adapter_len) { cut(r1, t1, l1, s->a1, &x1, &n1); if (n1) put(blob, RP_SINGLE_R1, x1, n1, nullptr, 0); }
                    if ((r2->flags & FASTP_GPU_RF_ADAPTER) && r2->adapter_len) { cut(*r2, t2, l2, s->a2, &x2, &n2); if (n2) put(blob, RP_SINGLE_R2, x2, n2, nullptr, 0); }
                }
            } else if ((r1.flags & FASTP_GPU_RF_ADAPTER) && r1.adapter_len) {
                cut(r1, t1, l1, s->a1, &x1, &n1);
                if (n1) put(blob, RP_SINGLE_R1, x1, n1, nullptr, 0);
            }
            auto fasta_events = [&](uint32_t key, const fastp_gpu_read_result& rr, const char* t, int tl, int is_r2) {   // trimByMultiSequences adaptertrimmer.cpp:48-62
                auto it = events.find(key);
                if (it == events.end()) return;
                for (const fastp_gpu_adapter_event* e : it->second) {
                    fastp_gpu_read_result q = rr;
                    q.adapter_pos = e->pos;
                    q.adapter_len = e->len;
                    const char* a; size_t la;
                    static const std::string none;
                    const std::string& aseq = e->adapter < s->fasta.size() ? s->fasta[e->adapter] : none;   // (a reference: cut() keeps a pointer into it)
                    cut(q, t, tl, aseq, &a, &la);
                    if (la) put(blob, is_r2 ? RP_SINGLE_R2 : RP_SINGLE_R1, a, la, nullptr, 0);
                }
            };
            if (ev1) fasta_events(k1, r1, t1, l1, 0);
            if (ev2) fasta_events(k2, *r2, t2, l2, 1);
        }
    }
};

void replay_blob(fastp_gpu_host* h, const std::vector<uint8_t>& blob) {
    size_t at = 0;
    while (at + 5 <= blob.size()) {
        const uint8_t* w = blob.data() + at;
        const int kind = w[0];
        const size_t la = (size_t)w[1] | ((size_t)w[2] << 8), lb = (size_t)w[3] | ((size_t)w[4] << 8);
        const char* a = (const char*)w + 5;
        if (kind == RP_PAIR) fastp_gpu_host_add_adapter_pair(h, a, (int32_t)la, a + la, (int32_t)lb);
        else fastp_gpu_host_add_adapter(h, kind == RP_SINGLE_R2, a, (int32_t)la);
        at += 5 + la + lb;
    }
}

}  // namespace

extern "C" {

const char* fastp_gpu_stream_last_error(const fastp_gpu_stream* s) { return s ? s->err.c_str() : g_stream_error.c_str(); }

int fastp_gpu_stream_create(const fastp_gpu_params* params, const fastp_gpu_stream_config* cfg, fastp_gpu_stream** out) {
    if (!params || !cfg || !out || !cfg->in1) { g_stream_error = "null argument"; return FASTP_GPU_E_INVALID; }
    std::unique_ptr<fastp_gpu_stream> s(new fastp_gpu_stream());
    memset(&s->st, 0, sizeof(s->st));
    s->p = *params;
    s->cfg = *cfg;
    s->paired = params->paired != 0;
    s->nm = s->paired ? 2 : 1;
    s->interleaved = cfg->interleaved != 0;
    if (s->interleaved && (!s->paired || cfg->in2)) { g_stream_error = "interleaved input is one file of a paired run"; return FASTP_GPU_E_INVALID; }
    s->nf = s->interleaved ? 1 : s->nm;
    if (!s->interleaved && s->paired != (cfg->in2 != nullptr)) { g_stream_error = "a paired engine needs two input files (or one interleaved file), a single-end engine one"; return FASTP_GPU_E_INVALID; }
    if (params->overlapped_out && !(cfg->want_overlapped && cfg->emit && s->paired)) {
        g_stream_error = "--overlapped_out's stream is assembled on the host: config.want_overlapped and an emit callback are needed (paired input)";
        return FASTP_GPU_E_UNSUPPORTED;
    }
    if (cfg->want_overlapped && !params->overlapped_out) { g_stream_error = "want_overlapped without params->overlapped_out"; return FASTP_GPU_E_INVALID; }
    s->in1 = cfg->in1;
    if (cfg->in2) s->in2 = cfg->in2;
    // own copies of every string the parameter block points at: a re-plan creates a context again
    if (params->adapter_seq_r1) { s->a1 = params->adapter_seq_r1; s->p.adapter_seq_r1 = s->a1.c_str(); }
    if (params->adapter_seq_r2) { s->a2 = params->adapter_seq_r2; s->p.adapter_seq_r2 = s->a2.c_str(); }
    for (int i = 0; i < params->n_adapter_fasta && params->adapter_fasta; i++) s->fasta.push_back(params->adapter_fasta[i]);
    for (auto& f : s->fasta) s->fastap.push_back(f.c_str());
    s->p.adapter_fasta = s->fastap.empty() ? nullptr : s->fastap.data();
    for (int i = 0; i < params->n_overrep_seqs1 && params->overrep_seqs1; i++) s->seeds[0].push_back(params->overrep_seqs1[i]);
    for (int i = 0; i < params->n_overrep_seqs2 && params->overrep_seqs2; i++) s->seeds[1].push_back(params->overrep_seqs2[i]);
    for (int m = 0; m < 2; m++) for (auto& q : s->seeds[m]) s->seedp[m].push_back(q.c_str());
    s->p.overrep_seqs1 = s->seedp[0].empty() ? nullptr : s->seedp[0].data();
    s->p.overrep_seqs2 = s->seedp[1].empty() ? nullptr : s->seedp[1].data();
    if (cfg->format.umi_prefix) s->umi_prefix = cfg->format.umi_prefix;
    s->umi_delim = cfg->format.umi_delimiter ? cfg->format.umi_delimiter : ":";
    s->cfg.format.umi_prefix = s->umi_prefix.empty() ? nullptr : s->umi_prefix.c_str();
    s->cfg.format.umi_delimiter = s->umi_delim.c_str();
    s->cfg.in1 = s->in1.c_str();
    s->cfg.in2 = s->nf > 1 ? s->in2.c_str() : nullptr;
    s->chunk = cfg->chunk_bytes > 0 ? cfg->chunk_bytes : (int64_t)env_int("FASTP_GPU_STREAM_CHUNK_MB", 16) << 20;
    if (cfg->chunk_bytes <= 0 && getenv("FASTP_GPU_STREAM_CHUNK_BYTES")) s->chunk = atoll(getenv("FASTP_GPU_STREAM_CHUNK_BYTES"));   // tests: many small trips
    s->chunk = std::max<int64_t>(4096, std::min<int64_t>(s->chunk, (int64_t)1 << 30)) / 256 * 256;
    if (s->cfg.io_threads <= 0) s->cfg.io_threads = env_int("FASTP_GPU_STREAM_IO_THREADS", 8);
    for (int m = 0; m < s->nf; m++) {
        s->src_kind[m] = input_kind(m ? s->in2 : s->in1);
        s->st.input_kind[m] = s->src_kind[m];
        s->any_bgzf = s->any_bgzf || s->src_kind[m] == SRC_BGZF;
    }
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
        const bool possible = q == FASTP_GPU_OUT1 || (s->paired && q == FASTP_GPU_OUT2) || (q == FASTP_GPU_FAILED && cfg->format.want_failed) ||
                              (q == FASTP_GPU_MERGED && s->paired && params->merge) || (q == FASTP_GPU_UNPAIRED1 && s->paired && cfg->format.want_unpaired1) ||
                              (q == FASTP_GPU_UNPAIRED2 && s->paired && cfg->format.want_unpaired2);
        if (s->cfg.want[q] && !possible) { g_stream_error = "a stream is wanted that the options never write to"; return FASTP_GPU_E_INVALID; }
        if (s->cfg.want[q] && s->cfg.out_fd[q] < 0 && !s->cfg.emit) { g_stream_error = "a wanted stream has neither a file descriptor nor an emit callback"; return FASTP_GPU_E_INVALID; }
    }
    if (params->merge && s->paired && !s->cfg.want[FASTP_GPU_MERGED] && (s->cfg.want[FASTP_GPU_OUT1] || s->cfg.want[FASTP_GPU_OUT2])) {
        // merge mode without --merged_out: legal for the reference (the merged reads are dropped); nothing to check
    }
    const double t0 = now_s();
    // the stream's buffers (the page-locked ones are the slow part: 40 ms of a 4 M-pair run's 0.57 s) are allocated on a helper
    // thread BESIDE the engine's creation (66 ms: runtime, second-stream probe, tables) - neither needs anything of the other
    int rc_buf = FASTP_GPU_OK;
    std::thread buf_thread([&] {
        rc_buf = alloc_buffers(s.get());
        fq::timeline("stream: buffers allocated (device + page-locked)");
    });
    int rc = fastp_gpu_create(&s->p, s->cfg.device, &s->ctx);
    std::string create_err;
    if (rc != FASTP_GPU_OK) create_err = std::string("fastp_gpu_create: ") + fastp_gpu_last_error(nullptr);
    fq::timeline("stream: engine created");
    buf_thread.join();
    if (rc != FASTP_GPU_OK) { free_buffers(s.get()); g_stream_error = create_err; return rc; }
    s->st.max_len = s->p.max_len;
    if (s->cfg.want_overlapped) (void)fastp_gpu_host_writes_overlapped(s->ctx, 1);
    if (rc_buf != FASTP_GPU_OK) {
        g_stream_error = s->err;
        free_buffers(s.get());
        fastp_gpu_destroy(s->ctx);
        return rc_buf;
    }
    s->st.setup_s = now_s() - t0;
    *out = s.release();
    return FASTP_GPU_OK;
}

void fastp_gpu_stream_destroy(fastp_gpu_stream* s) {
    if (!s) return;
    fq::timeline("stream: destroy begin");
    (void)hipSetDevice(s->cfg.device);
    free_buffers(s);
    fq::timeline("stream: buffers freed");
    if (s->ctx) fastp_gpu_destroy(s->ctx);
    delete s;
    fq::timeline("stream: destroy end");
}

int fastp_gpu_stream_get_stats(const fastp_gpu_stream* s, fastp_gpu_stream_stats* out) {
    if (!s || !out) return FASTP_GPU_E_INVALID;
    *out = s->st;
    return FASTP_GPU_OK;
}

int fastp_gpu_stream_layout(const fastp_gpu_stream* s, fastp_gpu_counter_layout* out) {
    if (!s || !out) return FASTP_GPU_E_INVALID;
    fastp_gpu_counter_layout_for_params(&s->p, out);
    return FASTP_GPU_OK;
}

int fastp_gpu_stream_counters(fastp_gpu_stream* s, int64_t* out, int64_t n) {
    if (!s || !out || !s->ctx) return FASTP_GPU_E_INVALID;
    fastp_gpu_counter_layout lay;
    std::vector<int64_t> c;
    int rc = fetch_ctx_counters(s, &lay, &c);
    if (rc) return rc;
    if (n != lay.total) return s->fail(FASTP_GPU_E_INVALID, "counter block size mismatch");
    for (auto& seg : s->segments) add_counters(lay, c, seg.first, seg.second, s->p.insert_size_max);
    memcpy(out, c.data(), (size_t)n * 8);
    return FASTP_GPU_OK;
}

}  // extern "C"

namespace {

// everything the threads of one run share
struct Run {
    fastp_gpu_stream* s;
    int fds[2] = {-1, -1};
    int64_t sizes[2] = {0, 0};
    bool pipe[2] = {false, false};   // not a regular file (--stdin, a FIFO): read() in sequence, no size
    std::atomic<int> io_err{0}, emit_err{0};
    Channel<ReadReq> q_req;
    Channel<ReadDone> q_done;
    Channel<WriteJob> q_write;
    Channel<int> q_ofree;
    Channel<ReplayJob> q_replay;
    int64_t out_pos[FASTP_GPU_N_OUTPUTS];
    explicit Run(fastp_gpu_stream* st) : s(st) {}
};

// a gzip stream that is not bgzip's: inflated on the host, member after member, as FastqReader::readToBufIgzip does
// (src/fastqreader.cpp:88-149: a stream may hold several members; anything but a gzip header behind a member, or a
// file that ends inside one, is an error there too)
struct GzSource {
    z_stream z;
    bool open = false, in_member = false, at_eof = false;
    int fd = -1;
    int64_t fpos = 0, fsize = 0;
    std::vector<uint8_t> in;
    size_t in_at = 0, in_len = 0;
    bool seekable = true;
    std::unique_ptr<fqgz::Gunzip> fast;   // the inflater of fq_gunzip.h: pipes, and FASTP_GPU_STREAM_GUNZIP_THREADS=1; FASTP_GPU_STREAM_GUNZIP=zlib: zlib's, for comparison
    std::unique_ptr<fqgz::ParallelGunzip> par;   // regular files (default): several threads on the one stream (fq_pgunzip.h)
    ~GzSource() { if (open) inflateEnd(&z); }
    // threads per ".gz" file: a quarter of the host's, at most 12 (a paired run has two such files, and the loop its own pools)
    static int gunzip_threads() {
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t set;   // a process confined to fewer CPUs (taskset, a container's cpuset) counts those
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) hw = std::min(hw > 0 ? hw : CPU_COUNT(&set), CPU_COUNT(&set));
        return env_int("FASTP_GPU_STREAM_GUNZIP_THREADS", std::max(1, std::min(12, hw / 4)));
    }
    // up to `want` bytes of text to dst; < 0: damaged stream / read error
    int64_t fill(uint8_t* dst, int64_t want, int* err) {
        static const bool use_zlib = getenv("FASTP_GPU_STREAM_GUNZIP") && !strcmp(getenv("FASTP_GPU_STREAM_GUNZIP"), "zlib");
        if (!use_zlib && seekable && !fast && (par || gunzip_threads() > 1)) {
            if (!par) {
                par.reset(new fqgz::ParallelGunzip());
                par->fd = fd;
                par->fsize = fsize;
                par->threads = gunzip_threads();
                par->chunk = (size_t)std::max(1, env_int("FASTP_GPU_STREAM_GUNZIP_CHUNK_KB", 2048)) << 10;   // (the tests: a few KiB, so that small files cross chunks)
            }
            const int64_t made = par->read(dst, want, err);
            fpos = par->fpos;
            at_eof = par->at_eof;
            return made;
        }
        if (!use_zlib || !seekable) {
            if (!fast) {
                fast.reset(new fqgz::Gunzip());
                fast->fd = fd;
                fast->fsize = fsize;
                fast->seekable = seekable;
            }
            const int64_t made = fast->read(dst, want, err);
            fpos = fast->fpos;
            at_eof = fast->at_eof;
            return made;
        }
        if (!open) {
            memset(&z, 0, sizeof(z));
            if (inflateInit2(&z, 15 + 16) != Z_OK) { *err = 4; return -1; }
            open = true;
            in.resize(4 << 20);
        }
        int64_t made = 0;
        while (made < want && !at_eof) {
            if (in_at == in_len) {
                const int64_t ask = std::min<int64_t>((int64_t)in.size(), fsize - fpos);
                int64_t got = 0;
                while (got < ask) {
                    const ssize_t r = pread(fd, in.data() + got, (size_t)(ask - got), (off_t)(fpos + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { *err = 1; return -1; }
                    got += r;
                }
                fpos += got;
                in_at = 0;
                in_len = (size_t)got;
                if (got == 0) {
                    if (in_member) { *err = 4; return -1; }   // "igzip: unexpected eof"
                    at_eof = true;
                    break;
                }
            }
            z.next_in = in.data() + in_at;
            z.avail_in = (uInt)(in_len - in_at);
            z.next_out = dst + made;
            z.avail_out = (uInt)std::min<int64_t>(want - made, 1 << 30);
            const uInt out0 = z.avail_out;
            const int rc = inflate(&z, Z_NO_FLUSH);
            in_at = in_len - z.avail_in;
            made += (int64_t)(out0 - z.avail_out);
            if (rc == Z_STREAM_END) {
                in_member = false;
                if (inflateReset(&z) != Z_OK) { *err = 4; return -1; }
            } else if (rc == Z_OK || rc == Z_BUF_ERROR) {
                in_member = true;
            } else {
                *err = 4;
                return -1;
            }
        }
        if (!at_eof && !in_member && in_at == in_len && fpos >= fsize) at_eof = true;
        return made;
    }
};

void reader_main(Run* R) {
    fastp_gpu_stream* s = R->s;
    (void)hipSetDevice(s->cfg.device);
    // reads scale with threads (page-cache copies into page-locked memory), writes do not (writer_main): twice the pool here
    IoPool pool(env_int("FASTP_GPU_STREAM_READ_THREADS", 2 * s->cfg.io_threads));
    int64_t pos[2] = {0, 0};
    GzSource gz[2];
    int64_t comp_have[2] = {0, 0};        // BGZF: bytes at the front of pin_comp[m] that no trip has taken yet
    bool src_eof[2] = {false, false};
    std::vector<uint32_t> ix32[4];
    std::vector<uint64_t> ix64;
    for (int m = 0; m < s->nf; m++) { gz[m].fd = R->fds[m]; gz[m].fsize = R->sizes[m]; gz[m].seekable = !R->pipe[m]; }
    auto pread_pieces = [&](int fd, uint8_t* dst, int64_t off0, int64_t want) {
        for (int64_t a = 0; a < want; a += IO_PIECE) {
            const int64_t e = std::min(want, a + IO_PIECE);
            const int64_t off = off0 + a;
            std::atomic<int>* err = &R->io_err;
            pool.submit([fd, dst, a, e, off, err] {
                int64_t got = 0;
                while (got < e - a) {
                    const ssize_t r = pread(fd, dst + a + got, (size_t)(e - a - got), (off_t)(off + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { err->store(1); return; }
                    got += r;
                }
            });
        }
    };
    for (;;) {
        ReadReq rq = R->q_req.get();
        if (rq.slot < 0) return;
        ReadDone d;
        d.slot = rq.slot;
        int64_t gz_made[2] = {0, 0};
        int gz_err[2] = {0, 0};
        for (int m = 0; m < s->nf; m++) {
            uint8_t* dst = s->pin_in[rq.slot][m] + rq.carry[m];
            if (s->src_kind[m] == SRC_PLAIN && R->pipe[m]) {
                if (rq.budget[m] > 0 && !src_eof[m]) {   // what the pipe has, up to the trip's size; 0 bytes = its writer is done
                    const int fd = R->fds[m];
                    const int64_t want = rq.budget[m];
                    int64_t* made = &gz_made[m];
                    pool.submit([fd, dst, want, made] {
                        int64_t got = 0;
                        while (got < want) {
                            const ssize_t r = read(fd, dst + got, (size_t)(want - got));
                            if (r < 0 && errno == EINTR) continue;
                            if (r < 0) { *made = -1; return; }
                            if (r == 0) { *made = -2 - got; return; }   // end of the pipe after `got` bytes
                            got += r;
                        }
                        *made = got;
                    });
                }
            } else if (s->src_kind[m] == SRC_PLAIN) {
                const int64_t want = std::max<int64_t>(0, std::min(rq.budget[m], R->sizes[m] - pos[m]));
                pread_pieces(R->fds[m], dst, pos[m], want);
                pos[m] += want;
                d.nb[m] = want;
                d.file_bytes[m] = want;
                src_eof[m] = pos[m] >= R->sizes[m];
            } else if (s->src_kind[m] == SRC_GZIP) {
                if (rq.budget[m] > 0 && !src_eof[m]) {
                    GzSource* g = &gz[m];
                    const int64_t want = rq.budget[m];
                    int64_t* made = &gz_made[m];
                    int* err = &gz_err[m];
                    pool.submit([g, dst, want, made, err] { *made = g->fill(dst, want, err); });
                }
            } else if (rq.budget[m] > 0) {   // SRC_BGZF: top the staging buffer up
                const int64_t want = std::max<int64_t>(0, std::min(s->comp_cap - comp_have[m], R->sizes[m] - pos[m]));
                pread_pieces(R->fds[m], s->pin_comp[m] + comp_have[m], pos[m], want);
                pos[m] += want;
                comp_have[m] += want;
                d.file_bytes[m] = want;
            }
        }
        pool.wait();
        if (R->io_err.load()) d.err = 1;
        for (int m = 0; m < s->nf && !d.err; m++)
            if (s->src_kind[m] == SRC_PLAIN && R->pipe[m] && rq.budget[m] > 0 && !src_eof[m]) {
                if (gz_made[m] == -1) { d.err = 1; break; }
                if (gz_made[m] <= -2) { src_eof[m] = true; gz_made[m] = -2 - gz_made[m]; }
                d.nb[m] = d.file_bytes[m] = gz_made[m];
                pos[m] += gz_made[m];
            }
        for (int m = 0; m < s->nf && !d.err; m++)
            if (s->src_kind[m] == SRC_GZIP && rq.budget[m] > 0 && !src_eof[m]) {
                if (gz_made[m] < 0) { d.err = gz_err[m] ? gz_err[m] : 4; break; }
                d.nb[m] = gz_made[m];
                src_eof[m] = gz[m].at_eof;
                d.file_bytes[m] = gz[m].fpos - pos[m];
                pos[m] = gz[m].fpos;
            }
        // ---- BGZF: the whole members at the front of the staging buffers whose text fits the trip ----
        int64_t consumed[2] = {0, 0};
        if (s->any_bgzf && !d.err) {
            int32_t nblk[2] = {0, 0};
            for (int k = 0; k < 4; k++) ix32[k].clear();
            ix64.clear();
            for (int m = 0; m < s->nf && !d.err; m++) {
                if (s->src_kind[m] != SRC_BGZF || rq.budget[m] <= 0) continue;
                const size_t at = ix64.size(), room = (size_t)s->max_blocks;
                for (int k = 0; k < 4; k++) ix32[k].resize(at + room);
                ix64.resize(at + room);
                fastp_gpu_inflate_info info;
                const int rc = fastp_gpu_bgzf_index(s->pin_comp[m], comp_have[m], s->max_blocks, rq.budget[m], ix32[0].data() + at, ix32[1].data() + at,
                                                    ix32[2].data() + at, ix32[3].data() + at, ix64.data() + at, &info);
                if (rc != FASTP_GPU_OK) { d.err = 5; break; }
                nblk[m] = info.n_blocks;
                for (int k = 0; k < 4; k++) ix32[k].resize(at + (size_t)info.n_blocks);
                ix64.resize(at + (size_t)info.n_blocks);
                for (int32_t k = 0; k < info.n_blocks; k++) {
                    ix32[0][at + (size_t)k] += (uint32_t)((int64_t)m * s->comp_cap);
                    ix64[at + (size_t)k] += (uint64_t)((int64_t)m * s->text_cap + rq.carry[m]);
                }
                consumed[m] = info.consumed;
                d.nb[m] = info.out_bytes;
                src_eof[m] = pos[m] >= R->sizes[m] && info.consumed == comp_have[m];
                // the file has ended and what is left of it is not a whole member: BgzfMtReader stops there silently
                // (src/bgzf.h:138-142); a run that drops reads unnoticed is worse
                if (pos[m] >= R->sizes[m] && info.consumed < comp_have[m]) {
                    const uint8_t* p = s->pin_comp[m] + info.consumed;
                    const int64_t rest = comp_have[m] - info.consumed;
                    if (rest < 18 || (int64_t)((uint32_t)p[16] | ((uint32_t)p[17] << 8)) + 1 > rest) { d.err = 5; break; }
                }
            }
            d.n_blocks = nblk[0] + nblk[1];
            if (!d.err && d.n_blocks > 0) {
                const size_t n = (size_t)d.n_blocks;
                uint8_t* w = s->pin_idx[rq.slot];
                for (int k = 0; k < 4; k++) memcpy(w + 4 * n * (size_t)k, ix32[k].data(), 4 * n);
                memcpy(w + 16 * n, ix64.data(), 8 * n);
                if (hipMemcpyAsync(s->d_idx[rq.slot], w, 24 * n, hipMemcpyHostToDevice, s->cp_in) != hipSuccess) d.err = 2;
                for (int m = 0; m < s->nf && !d.err; m++)
                    if (consumed[m] > 0 &&
                        hipMemcpyAsync(s->d_comp[rq.slot] + (size_t)m * (size_t)s->comp_cap, s->pin_comp[m], (size_t)consumed[m], hipMemcpyHostToDevice, s->cp_in) != hipSuccess)
                        d.err = 2;
            }
        }
        for (int m = 0; m < s->nf && !d.err; m++) {
            d.eof[m] = src_eof[m];
            // a BGZF file's fresh text is made on the device (run_loop): only the carried text goes up from here
            const int64_t total = rq.carry[m] + (s->src_kind[m] == SRC_BGZF ? 0 : d.nb[m]);
            memset(s->pin_in[rq.slot][m] + total, 0, 32);   // the parser reads 16-byte vectors past the end
            if (hipMemcpyAsync(s->d_text[rq.slot][m], s->pin_in[rq.slot][m], (size_t)total + 32, hipMemcpyHostToDevice, s->cp_in) != hipSuccess) d.err = 2;
        }
        if (!d.err && hipStreamSynchronize(s->cp_in) != hipSuccess) d.err = 2;
        for (int m = 0; m < s->nf; m++)
            if (consumed[m] > 0) {   // (after the copy: the staging buffer is its source)
                memmove(s->pin_comp[m], s->pin_comp[m] + consumed[m], (size_t)(comp_have[m] - consumed[m]));
                comp_have[m] -= consumed[m];
            }
        R->q_done.put(d);
    }
}

void writer_main(Run* R) {
    fastp_gpu_stream* s = R->s;
    IoPool pool(s->cfg.io_threads);
    for (;;) {
        WriteJob j = R->q_write.get();
        if (j.oslot < 0) return;
        if (j.copy_pending && j.oslot < 2 && hipEventSynchronize(s->ev_out[j.oslot]) != hipSuccess) R->io_err.store(4);
        if (R->io_err.load()) {   // the chunk's bytes may not have landed (or an earlier write failed): nothing of it - and of the
            if (j.oslot < 2) R->q_ofree.put(j.oslot);   // chunks behind it - goes to the files; the slot still returns so that the loop ends
            continue;
        }
        const double t0 = now_s();
        if (j.has_ov && !R->emit_err.load()) {   // --overlapped_out's records of the chunk, assembled on the host
            if (s->cfg.emit(s->cfg.user, FASTP_GPU_OVERLAPPED, j.ov.data(), (int64_t)j.ov.size()) != 0) R->emit_err.store(1);
            s->st.bytes_overlapped += (int64_t)j.ov.size();
        }
        for (int q = 0; q < FASTP_GPU_N_OUTPUTS && j.oslot != 3; q++) {
            if (!s->cfg.want[q]) continue;
            const uint8_t* src = j.oslot >= 2 ? BGZF_EOF : s->pin_out[j.oslot][q];   // oslot 2: the end-of-file members of the compressed streams
            if (j.oslot >= 2 && !j.len[q]) continue;
            if (s->cfg.out_fd[q] >= 0) {
                // one positional write per stream and chunk, the streams side by side: on tmpfs eight writers into one file
                // were slower than one (7.8 vs 10 GB/s, profiles/r04_dropin.txt); FASTP_GPU_STREAM_WRITE_PIECE_MB cuts them up
                static const int64_t piece = (int64_t)env_int("FASTP_GPU_STREAM_WRITE_PIECE_MB", 0) << 20;
                const int64_t step = piece > 0 ? piece : std::max<int64_t>(j.len[q], 1);
                for (int64_t a = 0; a < j.len[q]; a += step) {
                    const int64_t e = std::min(j.len[q], a + step);
                    const int fd = s->cfg.out_fd[q];
                    const int64_t off = R->out_pos[q] + a;
                    std::atomic<int>* err = &R->io_err;
                    pool.submit([fd, src, a, e, off, err] {
                        int64_t done = 0;
                        while (done < e - a) {
                            const ssize_t r = pwrite(fd, src + a + done, (size_t)(e - a - done), (off_t)(off + done));
                            if (r < 0 && errno == EINTR) continue;
                            if (r <= 0) { err->store(3); return; }
                            done += r;
                        }
                    });
                }
            } else if (!R->emit_err.load()) {
                if (s->cfg.emit(s->cfg.user, q, (const char*)src, j.len[q]) != 0) R->emit_err.store(1);
            }
            R->out_pos[q] += j.len[q];
            s->st.bytes_out[q] += j.len[q];
        }
        pool.wait();
        s->st.write_s += now_s() - t0;
        if (j.oslot < 2) R->q_ofree.put(j.oslot);
    }
}

void replay_main(Run* R) {
    for (;;) {
        ReplayJob j = R->q_replay.get();
        if (j.stop) return;
        const double t0 = now_s();
        replay_blob(R->s->cfg.host, j.blob);
        R->s->st.replay_s += now_s() - t0;
    }
}

// the caller's thread: one trip per chunk
int run_loop(Run* R) {
    fastp_gpu_stream* s = R->s;
    const int nm = s->nm, nf = s->nf;
    const auto file_of = [s](int m) { return s->interleaved ? 0 : m; };   // the file (text) a mate's records lie in
    Extractor ex{s, {}, {}, {}};
    int64_t carry[2] = {0, 0};
    int slot = 0;
    bool done = false, drain = false;
    {
        ReadReq rq;
        rq.slot = 0;
        for (int m = 0; m < nf; m++) rq.budget[m] = s->chunk;
        R->q_req.put(rq);
    }
    while (!done) {
        double t0 = now_s();
        ReadDone d = R->q_done.get();
        s->st.wait_read_s += now_s() - t0;
        if (d.err) {
            static const char* const what[6] = {"", "reading an input file failed", "host-to-device copy of a chunk failed", "", 
                                                "a gzip input is damaged or ends inside a member",
                                                "a BGZF input is damaged: not a whole bgzip member where one must start, or the file ends inside one"};
            return s->fail(d.err == 2 ? FASTP_GPU_E_HIP : FASTP_GPU_E_INVALID, what[d.err < 6 ? d.err : 1]);
        }
        slot = d.slot;
        if (d.n_blocks > 0) {   // BgzfMtReader's pool of inflaters (src/bgzf.h:165-195): one launch for the members of both files
            t0 = now_s();
            const size_t nb = (size_t)d.n_blocks;
            const uint8_t* ix = s->d_idx[slot];
            int32_t bad = -1;
            static const int check_crc = env_int("FASTP_GPU_STREAM_CHECK_CRC", 1);
            if (fastp_gpu_inflate_bgzf(s->ctx, s->d_comp[slot], d.n_blocks, (const uint32_t*)ix, (const uint32_t*)(ix + 4 * nb), (const uint32_t*)(ix + 8 * nb),
                                       (const uint32_t*)(ix + 12 * nb), (const uint64_t*)(ix + 16 * nb), s->d_text[slot][0], (int64_t)nf * s->text_cap, check_crc,
                                       &bad) != FASTP_GPU_OK)
                return s->fail_ctx(FASTP_GPU_E_INVALID, bad >= 0 ? "a BGZF member of an input does not inflate to what its trailer says (damaged file)" : "fastp_gpu_inflate_bgzf");
            // the text comes to the host as well: the part no record covers is carried from there, the adapter replay cuts
            // its strings from it
            for (int m = 0; m < nf; m++) {
                if (s->src_kind[m] != SRC_BGZF) continue;
                S_HIP(s, hipMemsetAsync(s->d_text[slot][m] + carry[m] + d.nb[m], 0, 32, s->sx));
                if (d.nb[m] > 0)
                    S_HIP(s, hipMemcpyAsync(s->pin_in[slot][m] + carry[m], s->d_text[slot][m] + carry[m], (size_t)d.nb[m], hipMemcpyDeviceToHost, s->sx));
            }
            S_HIP(s, hipStreamSynchronize(s->sx));
            s->st.inflate_s += now_s() - t0;
        }
        int64_t total[2] = {0, 0};
        bool all_eof = true, any_fresh = false;
        for (int m = 0; m < nf; m++) {
            total[m] = carry[m] + d.nb[m];
            s->st.bytes_in[m] += d.nb[m];
            s->st.bytes_file[m] += d.file_bytes[m];
            all_eof = all_eof && d.eof[m];
            any_fresh = any_fresh || d.nb[m] > 0;
        }
        // ---- parse: both mates to the same number of records ----
        int32_t cap = s->max_records;
        if (s->cfg.reads_to_process > 0) cap = (int32_t)std::min<int64_t>(cap, s->cfg.reads_to_process - s->st.units);
        fastp_gpu_parse_info info[2];
        memset(info, 0, sizeof(info));
        int n = 0;
        int32_t first_n[2] = {0, 0};   // complete records each file's text of this trip holds (up to cap)
        std::vector<int32_t> exotic[2];
        bool stop_after = false;
        t0 = now_s();
        for (int attempt = 0; s->interleaved; attempt++) {
            // one file, the mates' records in turn (FastqReaderPair::read, src/fastqreader.cpp:470-478): 2 * cap records at most;
            // an odd record at the end waits for its mate in the next trip (at the end of the file it has none: the reference's
            // pair is then incomplete, which is its end of input as well, src/peprocessor.cpp:906-909)
            if (cap <= 0) { n = 0; break; }
            if (attempt > 16) return s->fail(FASTP_GPU_E_INVALID, "the parser does not settle on a record count");
            int32_t want = 2 * cap;
            int prc = fastp_gpu_parse_fastq(s->ctx, s->d_text[slot][0], total[0], d.eof[0] ? 1 : 0, want, s->il_seq, s->il_qual, s->il_len, s->il_loff, s->il_llen, &info[0]);
            if (prc == FASTP_GPU_OK && (info[0].n_records & 1)) {
                want = info[0].n_records - 1;
                if (want == 0) { n = 0; break; }
                prc = fastp_gpu_parse_fastq(s->ctx, s->d_text[slot][0], total[0], d.eof[0] ? 1 : 0, want, s->il_seq, s->il_qual, s->il_len, s->il_loff, s->il_llen, &info[0]);
                if (prc == FASTP_GPU_OK && info[0].n_records != want) return s->fail(FASTP_GPU_E_INVALID, "the parser does not settle on a record count");
            }
            if (prc == FASTP_GPU_OK) {
                n = info[0].n_records / 2;
                first_n[0] = n;
                exotic[0].resize((size_t)info[0].n_exotic);
                if (info[0].n_exotic) fastp_gpu_parse_exotic(s->ctx, exotic[0].data(), info[0].n_exotic);
                for (int32_t& u : exotic[0]) u >>= 1;   // record -> unit
                break;
            }
            if (prc != FASTP_GPU_E_INVALID || info[0].first_bad < 0) return s->fail_ctx(prc, "fastp_gpu_parse_fastq");
            if (info[0].bad_kind == FASTP_GPU_PARSE_BAD_TOO_LONG) {
                const int rc = replan(s, info[0].max_seq_len);
                if (rc) return rc;
            } else if (info[0].bad_kind == FASTP_GPU_PARSE_BAD_ALPHABET) {
                return s->fail(FASTP_GPU_E_ALPHABET, "record " + std::to_string(2 * s->st.units + info[0].first_bad) + " of file 1 has a quality character outside '!'..'~'");
            } else {   // FastqReader::read returns NULL there, for either mate: the stream ends in front of that pair
                cap = info[0].first_bad / 2;
                s->st.truncated = 1;
                stop_after = true;
            }
        }
        for (int attempt = 0; !s->interleaved; attempt++) {
            if (cap <= 0) { n = 0; break; }
            if (attempt > 16) return s->fail(FASTP_GPU_E_INVALID, "the parser does not settle on a record count");
            bool again = false;
            int32_t want = cap;
            for (int pass = 0; pass < 2 && !again; pass++) {
                for (int m = 0; m < nm && !again; m++) {
                    if (pass == 1 && info[m].n_records == want) continue;
                    const int prc = fastp_gpu_parse_fastq(s->ctx, s->d_text[slot][m], total[m], d.eof[m] ? 1 : 0, want, s->d_seq[m], s->d_qual[m], s->d_len[m],
                                                          s->d_loff[m], s->d_llen[m], &info[m]);
                    if (prc == FASTP_GPU_OK) {
                        if (pass == 0) first_n[m] = info[m].n_records;
                        exotic[m].resize((size_t)info[m].n_exotic);   // records with letters outside ACGTN, for the text kernel
                        if (info[m].n_exotic) fastp_gpu_parse_exotic(s->ctx, exotic[m].data(), info[m].n_exotic);
                        continue;
                    }
                    if (prc != FASTP_GPU_E_INVALID || info[m].first_bad < 0) return s->fail_ctx(prc, "fastp_gpu_parse_fastq");
                    if (info[m].bad_kind == FASTP_GPU_PARSE_BAD_TOO_LONG) {
                        const int rc = replan(s, info[m].max_seq_len);
                        if (rc) return rc;
                        again = true;
                    } else if (info[m].bad_kind == FASTP_GPU_PARSE_BAD_ALPHABET) {
                        return s->fail(FASTP_GPU_E_ALPHABET, "record " + std::to_string(s->st.units + info[m].first_bad) + " of file " + std::to_string(m + 1) +
                                                                 " has a quality character outside '!'..'~'");
                    } else {   // FastqReader::read returns NULL there: the stream ends in front of this record
                        cap = info[m].first_bad;
                        s->st.truncated = 1;
                        stop_after = true;
                        again = true;
                    }
                }
                if (again) break;
                want = info[0].n_records;
                for (int m = 1; m < nm; m++) want = std::min(want, info[m].n_records);
                if (want == 0) break;
            }
            if (again) continue;
            n = want;
            break;
        }
        // a file that is at its end and has handed out its last complete record ends the stream, whatever the other file
        // still holds: the reference pairs packs up and stops at the shorter file (peprocessor.cpp:363-370, :1034-1037)
        bool exhausted = false;
        for (int m = 0; m < nf; m++)
            if (d.eof[m] && cap > 0 && (total[m] == 0 || (info[m].n_records == n && first_n[m] == n && first_n[m] < cap))) exhausted = true;
        s->st.parse_s += now_s() - t0;
        int64_t left[2] = {0, 0};
        bool any_left = false;
        for (int m = 0; m < nf; m++) {
            left[m] = total[m] - (n > 0 ? info[m].consumed : 0);
            any_left = any_left || left[m] > 0;
        }
        const bool limit_hit = s->cfg.reads_to_process > 0 && s->st.units + n >= s->cfg.reads_to_process;
        if (stop_after || limit_hit || (exhausted && nf > 1)) {
            done = true;
        } else if (all_eof) {
            // a trip takes at most max_records records: when the cap was hit, complete records may remain in the carried
            // text - keep parsing without reading until a trip comes back short.  What is left then is a trailing partial
            // record / the longer mate's surplus: the reference stops there too
            drain = n > 0 && n >= cap && any_left;
            done = !drain;
        } else if (n == 0) {
            for (int m = 0; m < nf; m++)
                if (left[m] >= s->chunk) return s->fail(FASTP_GPU_E_INVALID, "a record does not fit the chunk size (FASTP_GPU_STREAM_CHUNK_MB)");
            // nothing parsed and nothing new arrived: a BGZF member (up to 64 KiB of text) takes whole-member room behind the carried
            // text.  (Also with nothing carried: a chunk smaller than the file's first member would ask for the same trip for ever.)
            if (!any_fresh)
                return s->fail(FASTP_GPU_E_INVALID, "a record (or a BGZF member) does not fit the chunk size (FASTP_GPU_STREAM_CHUNK_MB)");
        }
        // the text the records do not cover moves to the front of the other slot; the reader fills in behind it while
        // the device works on this trip
        if (!done) {
            ReadReq rq;
            rq.slot = 1 - slot;
            for (int m = 0; m < nf; m++) {
                if (left[m] > 0) memcpy(s->pin_in[1 - slot][m], s->pin_in[slot][m] + (total[m] - left[m]), (size_t)left[m]);
                rq.carry[m] = left[m];
                rq.budget[m] = drain ? 0 : std::max<int64_t>(0, s->chunk - left[m]);
            }
            R->q_req.put(rq);
        }
        if (n > 0 && s->interleaved) {   // deal the file's records out: even ones are read 1, odd ones read 2 (strided device copies)
            t0 = now_s();
            const size_t ss = fastp_gpu_seq_stride(s->p.max_len), qs = fastp_gpu_qual_stride(s->p.max_len);
            for (int m = 0; m < 2; m++) {
                S_HIP(s, hipMemcpy2DAsync(s->d_seq[m], ss, s->il_seq + (size_t)m * ss, 2 * ss, ss, (size_t)n, hipMemcpyDeviceToDevice, s->sx));
                S_HIP(s, hipMemcpy2DAsync(s->d_qual[m], qs, s->il_qual + (size_t)m * qs, 2 * qs, qs, (size_t)n, hipMemcpyDeviceToDevice, s->sx));
                S_HIP(s, hipMemcpy2DAsync(s->d_len[m], 2, s->il_len + m, 4, 2, (size_t)n, hipMemcpyDeviceToDevice, s->sx));
                S_HIP(s, hipMemcpy2DAsync(s->d_loff[m], 16, s->il_loff + 4 * m, 32, 16, (size_t)n, hipMemcpyDeviceToDevice, s->sx));
                S_HIP(s, hipMemcpy2DAsync(s->d_llen[m], 16, s->il_llen + 4 * m, 32, 16, (size_t)n, hipMemcpyDeviceToDevice, s->sx));
            }
            S_HIP(s, hipStreamSynchronize(s->sx));
            s->st.parse_s += now_s() - t0;
        }
        if (n > 0 && s->cfg.phred64) {   // FastqReader::read's convertPhred64To33, before anything looks at a quality
            t0 = now_s();
            for (int m = 0; m < nm; m++)
                if (fastp_gpu_phred64_to_33(s->ctx, n, s->d_text[slot][file_of(m)], s->d_loff[m], s->d_llen[m], s->d_qual[m]) != FASTP_GPU_OK)
                    return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_phred64_to_33");
            s->st.parse_s += now_s() - t0;
        }
        if (n > 0) {
            // ---- the worker loop ----
            t0 = now_s();
            fastp_gpu_batch b;
            memset(&b, 0, sizeof(b));
            b.n = n;
            b.flags = FASTP_GPU_BATCH_STAT_ISIZE;
            b.seq1 = s->d_seq[0]; b.qual1 = s->d_qual[0]; b.len1 = s->d_len[0];
            if (s->paired) { b.seq2 = s->d_seq[1]; b.qual2 = s->d_qual[1]; b.len2 = s->d_len[1]; }
            std::vector<int32_t> xunits;   // units with letters outside ACGTN in either mate: the engine reads their text in place
            for (int m = 0; m < nf; m++)
                for (int32_t u : exotic[m])
                    if (u < n) xunits.push_back(u);
            if (!xunits.empty()) {
                std::sort(xunits.begin(), xunits.end());
                xunits.erase(std::unique(xunits.begin(), xunits.end()), xunits.end());
                b.n_exotic = (int32_t)xunits.size();
                b.exotic_dense = 1;
                b.exotic_unit = xunits.data();
                for (int m = 0; m < nm; m++) { b.exotic_text[m] = s->d_text[slot][file_of(m)]; b.exotic_off[m] = s->d_loff[m]; }
            }
            fastp_gpu_results r;
            memset(&r, 0, sizeof(r));
            r.r1 = s->d_res[0];
            if (s->paired) { r.r2 = s->d_res[1]; r.pair = s->d_pair; }
            r.corrections = s->d_corr; r.corrections_capacity = s->corr_cap; r.n_corrections = s->d_nc;
            r.adapter_events = s->d_ev; r.adapter_events_capacity = s->ev_cap; r.n_adapter_events = s->d_nev;
            const int src = fastp_gpu_submit_device(s->ctx, &b, &r, nullptr);
            if (src != FASTP_GPU_OK) return s->fail_ctx(src == FASTP_GPU_E_ALPHABET ? src : FASTP_GPU_E_HIP, "fastp_gpu_submit_device");
            if (fastp_gpu_synchronize(s->ctx) != FASTP_GPU_OK) return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_synchronize");
            s->st.engine_s += now_s() - t0;
            // ---- the sparse lists' fill counts and, for the adapter replay, the records come to the host ----
            t0 = now_s();
            S_HIP(s, hipMemcpyAsync(&s->h_counts[0], s->d_nc, 4, hipMemcpyDeviceToHost, s->sx));
            S_HIP(s, hipMemcpyAsync(&s->h_counts[1], s->d_nev, 4, hipMemcpyDeviceToHost, s->sx));
            const bool host_records = s->cfg.host || s->cfg.want_overlapped;
            if (host_records)
                for (int m = 0; m < nm; m++) {
                    S_HIP(s, hipMemcpyAsync(s->h_res[m], s->d_res[m], (size_t)n * sizeof(fastp_gpu_read_result), hipMemcpyDeviceToHost, s->sx));
                    S_HIP(s, hipMemcpyAsync(s->h_loff[m], s->d_loff[m], (size_t)n * 16, hipMemcpyDeviceToHost, s->sx));
                }
            S_HIP(s, hipStreamSynchronize(s->sx));
            const int32_t ncorr = s->corr_cap ? s->h_counts[0] : 0, nev = s->ev_cap ? s->h_counts[1] : 0;
            if (ncorr > s->corr_cap) return s->fail(FASTP_GPU_E_OVERFLOW, "correction list overflow: lower the chunk size");
            if (nev > s->ev_cap) return s->fail(FASTP_GPU_E_OVERFLOW, "adapter event list overflow: lower the chunk size");
            WriteJob wj;
            if (host_records) {
                if (ncorr) S_HIP(s, hipMemcpyAsync(s->h_corr, s->d_corr, (size_t)ncorr * sizeof(fastp_gpu_correction), hipMemcpyDeviceToHost, s->sx));
                if (nev) S_HIP(s, hipMemcpyAsync(s->h_ev, s->d_ev, (size_t)nev * sizeof(fastp_gpu_adapter_event), hipMemcpyDeviceToHost, s->sx));
                if (ncorr || nev) S_HIP(s, hipStreamSynchronize(s->sx));
                const uint8_t* text[2] = {s->pin_in[slot][0], nm > 1 ? s->pin_in[slot][file_of(1)] : nullptr};
                ex.index(ncorr, nev);
                if (s->cfg.host) {
                    ReplayJob job;
                    ex.run(n, text, job.blob);
                    if (!job.blob.empty()) R->q_replay.put(std::move(job));
                }
                if (s->cfg.want_overlapped) {
                    wj.has_ov = true;
                    ex.overlapped(n, text, wj.ov);
                }
            }
            s->st.d2h_s += now_s() - t0;
            // ---- records -> the text of every output stream (-> gzip members) ----
            if (!s->any_out && wj.has_ov) {
                wj.oslot = 3;
                R->q_write.put(std::move(wj));
            }
            if (s->any_out) {
                t0 = now_s();
                fastp_gpu_format_io io[2];
                for (int m = 0; m < nm; m++) { io[m].text = s->d_text[slot][file_of(m)]; io[m].line_off = s->d_loff[m]; io[m].line_len = s->d_llen[m]; io[m].res = s->d_res[m]; }
                fastp_gpu_format_options fo = s->cfg.format;
                fo.corrections_capacity = s->corr_cap;
                // the output slot first: its device buffers and its page-locked ones are free again when the writer has
                // written what the slot held (which also says that the slot's last copy has landed)
                const double tw = now_s();
                wj.oslot = R->q_ofree.get();
                s->st.wait_write_s += now_s() - tw;
                t0 = now_s();
                uint8_t* const* d_out = s->d_out[wj.oslot];
                int64_t lens[FASTP_GPU_N_OUTPUTS];
                if (fastp_gpu_format_streams(s->ctx, n, &io[0], s->paired ? &io[1] : nullptr, s->d_pair, s->d_corr, s->corr_cap ? s->d_nc : nullptr, &fo,
                                             const_cast<uint8_t**>(d_out), s->out_cap, lens) != FASTP_GPU_OK)
                    return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_format_streams");
                s->st.format_s += now_s() - t0;
                t0 = now_s();
                const uint8_t* src[FASTP_GPU_N_OUTPUTS];
                for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
                    src[q] = d_out[q];
                    if (!s->cfg.want[q]) { lens[q] = 0; continue; }
                    if (s->cfg.compress[q] && lens[q] > 0) {
                        int64_t glen = 0;
                        if (fastp_gpu_deflate_bgzf(s->ctx, d_out[q], lens[q], 0, s->d_gz[wj.oslot][q], s->gz_cap[q], &glen) != FASTP_GPU_OK)
                            return s->fail_ctx(FASTP_GPU_E_HIP, "fastp_gpu_deflate_bgzf");
                        lens[q] = glen;
                        src[q] = s->d_gz[wj.oslot][q];
                    }
                }
                s->st.deflate_s += now_s() - t0;
                t0 = now_s();
                // (the formatter / deflater have returned their lengths: the text is complete on the device)  The copy runs on its
                // own stream; the writer thread waits for its event, this thread goes on to the next chunk
                for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) {
                    wj.len[q] = lens[q];
                    if (lens[q] > 0) S_HIP(s, hipMemcpyAsync(s->pin_out[wj.oslot][q], src[q], (size_t)lens[q], hipMemcpyDeviceToHost, s->cp_out));
                }
                S_HIP(s, hipEventRecord(s->ev_out[wj.oslot], s->cp_out));
                wj.copy_pending = true;
                s->st.d2h_s += now_s() - t0;
                R->q_write.put(std::move(wj));
            }
            s->st.units += n;
            s->st.chunks++;
        }
        if (R->io_err.load()) return s->fail(FASTP_GPU_E_INVALID, "writing an output file failed");
        if (R->emit_err.load()) return s->fail(FASTP_GPU_E_INVALID, "the emit callback stopped the run");
        for (int m = 0; m < nf; m++) carry[m] = left[m];
    }
    return FASTP_GPU_OK;
}

}  // namespace

template <class G>
static int gunzip_file_with(G* g, int fd, uint8_t* out, int64_t capacity, int64_t piece, int64_t* out_len);

extern "C" int fastp_gpu_stream_gunzip_file(const char* path, uint8_t* out, int64_t capacity, int64_t piece, int64_t* out_len) {
    return fastp_gpu_stream_gunzip_file_mt(path, out, capacity, piece, 1, 0, out_len);
}

extern "C" int fastp_gpu_stream_gunzip_file_mt(const char* path, uint8_t* out, int64_t capacity, int64_t piece, int threads, int64_t chunk_bytes,
                                               int64_t* out_len) {
    if (!path || !out || !out_len || capacity < 0 || threads < 1 || chunk_bytes < 0) return FASTP_GPU_E_INVALID;
    *out_len = 0;
    const int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0) {
        if (fd >= 0) close(fd);
        g_stream_error = std::string("cannot open ") + path;
        return FASTP_GPU_E_INVALID;
    }
    if (threads > 1 || chunk_bytes > 0) {   // (one thread with a chunk size: fq_pgunzip.h's decoder on its own)
        std::unique_ptr<fqgz::ParallelGunzip> g(new fqgz::ParallelGunzip());
        g->fd = fd;
        g->fsize = (int64_t)sb.st_size;
        g->threads = threads;
        if (chunk_bytes > 0) g->chunk = (size_t)chunk_bytes;
        const int rc = gunzip_file_with(g.get(), fd, out, capacity, piece, out_len);
        g.reset();   // joins the threads that read ahead
        close(fd);
        return rc;
    }
    std::unique_ptr<fqgz::Gunzip> g(new fqgz::Gunzip());
    g->fd = fd;
    g->fsize = (int64_t)sb.st_size;
    const int rc = gunzip_file_with(g.get(), fd, out, capacity, piece, out_len);
    close(fd);
    return rc;
}

template <class G>
static int gunzip_file_with(G* g, int fd, uint8_t* out, int64_t capacity, int64_t piece, int64_t* out_len) {
    if (piece <= 0) piece = 1 << 20;
    int rc = FASTP_GPU_OK;
    for (;;) {
        const int64_t want = std::min(piece, capacity - *out_len);
        if (want <= 0) {   // full: fine if the stream has ended as well
            uint8_t probe;
            int err = 0;
            const int64_t more = g->read(&probe, 1, &err);
            if (more != 0) { rc = more < 0 ? FASTP_GPU_E_INVALID : FASTP_GPU_E_OVERFLOW; g_stream_error = more < 0 ? "damaged gzip stream" : "output buffer too small"; }
            break;
        }
        int err = 0;
        const int64_t made = g->read(out + *out_len, want, &err);
        if (made < 0) { rc = FASTP_GPU_E_INVALID; g_stream_error = err == 1 ? "reading the file failed" : "damaged gzip stream"; break; }
        *out_len += made;
        if (made < want) break;   // fewer than asked: the file has ended
    }
    (void)fd;   // closed by the caller once the inflater - whose threads may still be reading ahead - is gone
    return rc;
}

extern "C" int fastp_gpu_stream_run(fastp_gpu_stream* s) {
    if (!s || !s->ctx) return FASTP_GPU_E_INVALID;
    const double t_start = now_s();
    fq::timeline("stream: run begin");
    S_HIP(s, hipSetDevice(s->cfg.device));
    Run R(s);
    const char* paths[2] = {s->cfg.in1, s->cfg.in2};
    for (int m = 0; m < s->nf; m++) {
        R.fds[m] = open(paths[m], O_RDONLY);
        struct stat sb;
        if (R.fds[m] < 0 || fstat(R.fds[m], &sb) != 0 || S_ISDIR(sb.st_mode)) {
            for (int k = 0; k <= m; k++) if (R.fds[k] >= 0) close(R.fds[k]);
            return s->fail(FASTP_GPU_E_INVALID, std::string("cannot open as a file or a pipe: ") + paths[m]);
        }
        R.pipe[m] = !S_ISREG(sb.st_mode);   // --stdin, a FIFO: read in sequence (a bgzip-written stream then goes through the host inflater)
        R.sizes[m] = R.pipe[m] ? ((int64_t)1 << 62) : (int64_t)sb.st_size;
    }
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) R.out_pos[q] = s->cfg.out_offset[q];
    R.q_ofree.put(0);
    R.q_ofree.put(1);
    std::thread reader(reader_main, &R), writer(writer_main, &R), replayer(replay_main, &R);
    int rc = run_loop(&R);
    if (rc == FASTP_GPU_OK) {   // bgzip's empty last member ends every compressed stream
        WriteJob eofs;
        eofs.oslot = 2;
        bool any = false;
        for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++)
            if (s->cfg.want[q] && s->cfg.compress[q]) { eofs.len[q] = (int64_t)sizeof(BGZF_EOF); any = true; }
        if (any) R.q_write.put(eofs);
    }
    ReadReq stop_r;
    R.q_req.put(stop_r);
    WriteJob stop_w;
    R.q_write.put(stop_w);
    ReplayJob stop_p;
    stop_p.stop = true;
    R.q_replay.put(std::move(stop_p));
    reader.join();
    writer.join();
    replayer.join();
    for (int m = 0; m < s->nf; m++) close(R.fds[m]);
    if (rc == FASTP_GPU_OK && R.io_err.load()) rc = s->fail(FASTP_GPU_E_INVALID, "file I/O failed");
    if (rc == FASTP_GPU_OK && R.emit_err.load()) rc = s->fail(FASTP_GPU_E_INVALID, "the emit callback stopped the run");
    s->st.wall_s = now_s() - t_start;
    fq::timeline("stream: run end");
    return rc;
}
