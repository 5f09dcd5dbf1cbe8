// fastp_gpu.hip - C ABI of the engine (include/fastp_gpu.h) on the HIP runtime:
// context, HBM buffers, kernel launches.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "fq_device.h"
#include "fq_timeline.h"
#include "fq_stats.h"
#include "fq_stats5.h"
#include "fq_lane.h"
#include "fq_text.h"
#include "fq_inflate.h"
#include "fq_eval.h"
#include "fq_deflate.h"
#include "fq_inflate_wave.h"
#include "fq_host.h"

using namespace fq;

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(1024) fq_fused_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    // read the argument block through the kernarg segment pointer (scalar loads where a field is
    // used) instead of holding all ~150 dwords in SGPRs for the whole persistent loop
    fused_body<false>(*kernel_args(&a), fq_lds);
}
// the per-read kernel of the split plan as one large workgroup per CU (A/B against the small-workgroup form)
extern "C" __global__ void __launch_bounds__(1024) fq_scan_wide_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    fused_body<true>(*kernel_args(&a), fq_lds);
}
// the per-read kernel of the split plan: 256-lane workgroups, four to a CU (fq_stats_kernel counts afterwards)
extern "C" __global__ void __launch_bounds__(256, 4) fq_scan_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    fused_body<true>(*kernel_args(&a), fq_lds);
}
// the per-read path with one lane per pair, reads in registers (fq_lane.h): SWM base words per read (10: up to 160
// bases, 16: up to 256), B bloom buffers hashed (0: no hashing in this launch), byte planes per prime, paired / single
#ifndef FQ_LANE_WAVES
#define FQ_LANE_WAVES 3   // wavefronts per SIMD the lane kernel is compiled for (168 VGPRs: no spills; 4 = 128 VGPRs spills ~26 dwords)
#endif
// (reads of up to 256 bases, SWM = 16: the row stage alone is 16 KB per wavefront, two workgroups fit a CU whatever the
// register count - compiled for two wavefronts per SIMD, no spills)
// One workgroup per CU that owns the whole LDS: the tables every wavefront reads (hash planes, counters, LUTs: 12 KB) are then
// held once per CU instead of once per 256-lane workgroup, which is what makes room for twelve wavefronts' row stages PLUS
// read 1's partial quality sums (three 256-lane workgroups of 51 KB fitted, of 56 KB they do not: two per CU, a third of
// the wavefronts gone - profiles/r04_lane_metrics_ab.txt).  FQ_LANE_WAVES wavefronts per SIMD: 168 VGPRs.
template <int SWM> struct LaneGeom { enum { MAX_THREADS = SWM > 10 ? 512 : 256 * FQ_LANE_WAVES }; };
template <int SWM, int B, int NPL, bool PAIRED, int EXT>
__global__ void __launch_bounds__(LaneGeom<SWM>::MAX_THREADS, 1) fq_lane_kernel(LaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    lane_body<SWM, B, NPL, PAIRED, EXT>(*kernel_args(&a), fq_lds);
}
// EXT >= 2 (fronts, -c, --merge) at 168 VGPRs spills ~150 dwords per lane; the same body compiled for two wavefronts per SIMD
// (512 lanes, 256 VGPRs) - FASTP_GPU_LANE_EXT_WAVES picks (A/B, profiles/r05_lane_ext_waves_ab.txt)
template <int B, int EXT>
__global__ void __launch_bounds__(512, 1) fq_lane_pair2w_kernel(LaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    lane_body<10, B, 3, true, EXT>(*kernel_args(&a), fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024, 8) fq_stats_kernel(StatsArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    if (a.form == 4) {   // (uniform)
#ifdef FQ_PROFILE_ABLATION
        if ((a.debug_skip & 0x1C0u) && a.kc == 4) { stats_body4<4, true>(a, fq_lds); return; }   // profiling build only (FASTP_GPU_DEBUG_SKIP)
#endif
        if (a.kc == 4) stats_body4<4, false>(a, fq_lds);
        else if (a.kc == 2) stats_body4<2, false>(a, fq_lds);
        else stats_body4<1, false>(a, fq_lds);
    } else {
        stats_body(a, fq_lds);
    }
}
// form 5 of the Stats kernel (fq_stats5.h): one 1024-lane workgroup per CU that owns the joint table (110 KB at ten item columns)
extern "C" __global__ void __launch_bounds__(1024) fq_stats5_kernel(StatsArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
#ifdef FQ_PROFILE_ABLATION
    if ((a.debug_skip & 0xC0u) && a.Hs == 10 && a.kc == 2) { stats_body5<2, 10, true>(a, fq_lds); return; }   // profiling build only
#endif
#ifndef FQ_ST5_ONEBLK
#define FQ_ST5_ONEBLK 1   // (A/B: 0 = the block form for every length, tools/gpu_r6_n.sh)
#endif
#ifndef FQ_ST5_NOFRONT
#define FQ_ST5_NOFRONT 1   // (A/B: 0 = the front read from the arguments / the records whatever the options)
#endif
    const bool nofront = FQ_ST5_NOFRONT && !a.front_per_read && a.front[0] == 0 && a.front[1] == 0;   // (uniform)
#ifndef FQ_ST5_TAILCOL
#define FQ_ST5_TAILCOL 0   // (A/B: 1 = the reads' last column out of the lane mapping: visit o -2.3 %, visit p +1.4 % - no gain, off)
#endif
    if (FQ_ST5_TAILCOL && FQ_ST5_ONEBLK && a.Hs == 10 && a.kc == 2 && a.H16 == 10 && nofront) stats_body5<2, 10, false, true, true, true>(a, fq_lds);   // 145 - 160 bases, no front trim
    else if (FQ_ST5_ONEBLK && a.Hs == 10 && a.kc == 2 && a.H16 <= 10 && nofront) stats_body5<2, 10, false, true, true>(a, fq_lds);   // reads of up to 160 bases, no front trim
    else if (FQ_ST5_ONEBLK && a.Hs == 10 && a.kc == 2 && a.H16 <= 10) stats_body5<2, 10, false, true>(a, fq_lds);   // reads of up to 160 bases (uniform)
    else if (a.Hs == 10 && a.kc == 2) stats_body5<2, 10, false>(a, fq_lds);
    else if (a.Hs == 8 && a.kc == 2) stats_body5<2, 8, false>(a, fq_lds);  // two blocks of eight columns: reads of up to 256 bases
    else if (a.kc == 2) stats_body5<2, 0, false>(a, fq_lds);
    else stats_body5<1, 0, false>(a, fq_lds);
}
// Do two streams run BESIDE each other?  HIP hands a stream one of a few hardware queues when it is created; two streams of one
// engine can land on the same queue (other engines, torch's and the caller's streams hold the others) and then run one behind the
// other whatever the events between them say - the kernels the engine puts beside the lane / Stats kernels on its second stream (the
// text kernel, Duplicate's tail, the overrepresentation analysis) lost exactly that inside a process that held a second engine
// (profiles/r06_e_*: one soft-masked pair in 1000, 4.48 ms alone, 6.43 ms inside bench.py).  The probe: one kernel on the first
// stream waits (bounded) for a word that a kernel on the second stream sets.
extern "C" __global__ void __launch_bounds__(64) fq_probe_wait_kernel(int* flag, long long budget_cycles) {
#ifndef FQ_HOSTSIM
    if (threadIdx.x == 0) {
        const long long t0 = clock64();
        int seen = 0;
        while (clock64() - t0 < budget_cycles) {
            if (__hip_atomic_load(&flag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { seen = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        flag[1] = seen;
    }
#else
    (void)flag; (void)budget_cycles;
#endif
}
extern "C" __global__ void __launch_bounds__(64) fq_probe_set_kernel(int* flag) {
    if (threadIdx.x == 0) __hip_atomic_store(&flag[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" __global__ void __launch_bounds__(256) fq_front_stats_kernel(FrontStatsArgs c) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    front_stats_body(c, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_hash_kernel(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    hash_body(*kernel_args(&a), fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_ovr_pass_kernel(OvrArgs o) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    ovr_pass_body(o, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_ovr_scan_kernel(OvrArgs o, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    ovr_scan_body(o, nblocks, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_ovr_tasks_kernel(OvrArgs o) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    ovr_tasks_body(o, fq_lds);
}
extern "C" __global__ void __launch_bounds__(OVR_BLOCK) fq_ovr_count_kernel(OvrArgs o) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    ovr_count_body(o, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_ovr_corr_link_kernel(OvrArgs o) { ovr_corr_link_body(o); }
extern "C" __global__ void __launch_bounds__(64) fq_ovr_dist_kernel(OvrArgs o) { ovr_dist_body(o); }
extern "C" __global__ void __launch_bounds__(256) fq_parse_count_kernel(ParseArgs p) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    parse_count_body(p, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_parse_scan_kernel(ParseArgs p, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    parse_scan_body(p, nblocks, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_parse_index_kernel(ParseArgs p) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    parse_index_body(p, fq_lds);
}
extern "C" __global__ void __launch_bounds__(64) fq_parse_finish_kernel(ParseArgs p) { parse_finish_body(p); }
extern "C" __global__ void __launch_bounds__(256) fq_parse_pack_kernel(ParseArgs p) { parse_pack_body(p); }
extern "C" __global__ void __launch_bounds__(64) fq_inflate_kernel(InflateArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    inflate_body(a, (u16*)fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_corr_stats_kernel(CorrStatsArgs c) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    corr_stats_body(c, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_corr_link_kernel(OvrArgs o) { corr_link_stride_body(o); }
extern "C" __global__ void __launch_bounds__(256) fq_dedup_apply_kernel(DedupApplyArgs d) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    dedup_apply_body(d, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_dup_final_kernel(DupFinalArgs d) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    dup_final_body(d, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_or_images_kernel(OrArgs o) { or_images_body(o); }
extern "C" __global__ void __launch_bounds__(256) fq_phred64_kernel(Phred64Args a) { phred64_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_fmt_len_kernel(FmtArgs f) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    fmt_len_body(f, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_fmt_scan_kernel(FmtArgs f) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    fmt_scan_body(f, (u64*)fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_fmt_write_kernel(FmtArgs f) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    fmt_write_body(f, fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_fmt_fix_kernel(FmtArgs f) { fmt_fix_body(f); }
extern "C" __global__ void __launch_bounds__(64) fq_inflate_wave_kernel(InflateArgs a) {
    extern __shared__ u32 fq_lds[];
    inflate_wave_body(a, fq_lds);
}
extern "C" __global__ void __launch_bounds__(64) fq_deflate_kernel(DeflateArgs a) {
    extern __shared__ u32 fq_lds[];
    deflate_body(a, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_deflate_scan_kernel(DeflateArgs a) {
    extern __shared__ u32 fq_lds[];
    deflate_scan_body(a, (u64*)fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_deflate_gather_kernel(DeflateArgs a) { deflate_gather_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_eval_kmer_kernel(EvalKmerArgs a) { eval_kmer_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_eval_census_kernel(EvalCensusArgs a) { eval_census_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_eval_harvest_kernel(EvalCensusArgs a) { eval_harvest_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_eval_text_kernel(EvalCensusArgs a) { eval_text_body(a); }
extern "C" __global__ void __launch_bounds__(256) fq_fmts_corr_kernel(FmtsArgs f) { fmts_corr_body(f); }
extern "C" __global__ void __launch_bounds__(256) fq_fmts_len_kernel(FmtsArgs f) {
    extern __shared__ u32 fq_lds[];
    fmts_len_body(f, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_fmts_scan_kernel(FmtsArgs f) {
    extern __shared__ u32 fq_lds[];
    fmts_scan_body(f, (u64*)fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_fmts_write_kernel(FmtsArgs f) {
    extern __shared__ u32 fq_lds[];
    fmts_write_body(f, fq_lds);
}
extern "C" __global__ void __launch_bounds__(64 * TEXT_WAVES) fq_text_kernel(TextArgs e) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];
    text_body(*kernel_args(&e), fq_lds);
}
extern "C" __global__ void __launch_bounds__(256) fq_text_mask_kernel(TextMaskArgs m) { text_mask_body(m); }
extern "C" __global__ void __launch_bounds__(256) fq_reduce_kernel(ReduceArgs r) { reduce_body(r); }
extern "C" __global__ void __launch_bounds__(256) fq_dup_probe_kernel(DupArgs d) { dup_probe_body(d); }
extern "C" __global__ void __launch_bounds__(256) fq_dup_claim_kernel(DupArgs d) { dup_claim_body(d); }
extern "C" __global__ void __launch_bounds__(256) fq_dup_losers_kernel(DupArgs d) { dup_losers_body(d); }
extern "C" __global__ void __launch_bounds__(256) fq_dup_winners_kernel(DupArgs d) { dup_winners_body(d); }
extern "C" __global__ void __launch_bounds__(1024) fq_dup_finish_kernel(DupArgs d) {
    extern __shared__ u32 fq_lds[];
    dup_finish_body(d, fq_lds);
}
extern "C" __global__ void __launch_bounds__(1024) fq_dup_resolve_kernel(DupArgs d) {
    extern __shared__ __attribute__((aligned(16))) u32 fq_lds[];  // 1 dword: the workgroup's duplicate count
    dup_resolve_body(d, fq_lds);
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

struct fastp_gpu_ctx {
    fastp_gpu_params params;
    std::string adapter1, adapter2;
    DevParams dp;
    HostLuts luts;
    TileConfig cfg;
    LdsLayout L;
    fastp_gpu_counter_layout cl;
    int device = 0;
    int cus = 0;
    int blocks = 0;           // persistent workgroups per launch
    int max_pairs_per_launch = 0;
    // split plan (fq_stats.h): the per-read kernel as small workgroups, Stats::statRead as its own streaming kernel
    bool split = false;
    int st_threads = 0, st_blocks = 0;     // the Stats kernel's workgroup size and the most workgroups it is launched with
    int st_H = 0, st_Hs = 0, st_lds_dwords = 0, st_slab_dwords = 0;
    u32* d_corr_int = nullptr; size_t corr_int_cap = 0;      // -c on the lane plan: the launch's corrections (KernelArgs::corr_int) + 1 counter word
    u32* d_corr_chain = nullptr; size_t corr_chain_cap = 0;  // their per-read chains: head[reads] | next[capacity]
    int ln_prefetch = 0;   // FASTP_GPU_LANE_PREFETCH (round 6): LaneArgs::prefetch
    int ln_grab = 1;       // FASTP_GPU_LANE_GRAB (round 6): LaneArgs::grab
    int ln_glds = 0;   // FASTP_GPU_LANE_GLDS (A/B, measured null: profiles/r05_lane_glds_ab.txt): LaneArgs::glds
    bool ln_2w = false;   // the lane kernel's EXT >= 2 instantiation compiled for two wavefronts per SIMD (FASTP_GPU_LANE_EXT_WAVES=2)
    int st_form = 4, st_kc = 4, st_max_reads = CYC_MAX_READS, st_max_grid = 0;   // FASTP_GPU_STATS_V / _KC: the Stats kernel's form (fq_stats.h)
    int st_l_cyc = 0, st_l_kmer = 0, st_l_qh = 0, st_l_lut = 0, st_l_mt = 0, st_l_wl = 0, st_wl_cap = 0;
    int st_H16 = 0, st_l_ovf = 0;          // form 5 (fq_stats5.h)
    u32* d_st_slabs = nullptr;
    // lane plan (fq_lane.h): one lane per pair, reads in registers - the option family lane_plan_supported() admits
    bool lane = false;
    int ln_swm = 0, ln_blocks = 0, ln_threads = 256;
    LaneLds ln_lds;
    u32* d_ln_slabs = nullptr;
    int* d_ln_ctr = nullptr;       // the lane kernel's chunk counter
    int ln_pool_base = 0;          // ... as the pool's counter (LaneArgs::pool_base): what it may have reached by the next launch
    // split plans: Duplicate's losers / winners / finish kernels of a launch run on this stream beside its Stats kernel
    hipStream_t tail = nullptr;
    hipEvent_t ev_k1 = nullptr, ev_tail = nullptr;
    u32* d_swin[2] = {nullptr, nullptr}; size_t swin_cap = 0;
    hipStream_t stream = nullptr;
    // Duplicate's probe + resolve of launch k run beside the fused kernel of launch k + 1 (see launch_chunk): the fused
    // kernel holds every VGPR of the CUs it sits on, so the pair of streams is confined to disjoint CU sets
    hipStream_t aux = nullptr;
    hipEvent_t ev_fused[2] = {nullptr, nullptr}, ev_dup[2] = {nullptr, nullptr};
    bool ev_dup_set[2] = {false, false};
    bool aux_pending = false;      // the aux stream has work the main stream has not joined yet
    int aux_last = 0;
    uint64_t launch_seq = 0;
    u64* d_dup_pos2[2] = {nullptr, nullptr}; size_t dup_pos2_cap[2] = {0, 0};
    const void* last_res[2] = {nullptr, nullptr};   // result rows [begin, end) the aux stream's last resolve writes to
    // device buffers
    int16_t* d_ov_limit = nullptr;
    u16* d_lowq = nullptr;
    u16* d_cplx = nullptr;
    u32* d_primes = nullptr;
    u32* d_planes = nullptr;  // byte planes of the primes (DevLuts::dup_planes)
    u64* d_posum = nullptr;
    u32* d_fasta_words = nullptr;
    int* d_fasta_len = nullptr;
    std::vector<std::string> fasta_strings;
    std::vector<const char*> fasta_ptrs;
    int64_t* d_ctr = nullptr;
    u32* d_slabs = nullptr;
    int slab_dwords = 0;
    u32* d_bitmap = nullptr;  // Duplicate::mDupBuf as u32 words
    u64* d_dup_pos = nullptr; size_t dup_pos_cap = 0;
    u64* d_table = nullptr; size_t table_cap = 0;
    u8* d_need = nullptr; size_t need_cap = 0;
    u8* d_setw = nullptr; size_t setw_cap = 0;
    u32* d_cfilter = nullptr;
    u8* d_dupflag = nullptr; size_t dupflag_cap = 0;   // --dedup: per-unit duplicate decision
    u32* d_prefix = nullptr;                           // sharded runs: OR of the preceding shards' bitmaps
    bool has_prefix = false;
    bool host_overlapped = false;   // fastp_gpu_host_writes_overlapped: the caller assembles --overlapped_out's stream from the records
    // overrepresentation analysis
    u32* d_ovr_table[2] = {nullptr, nullptr};
    u8* d_ovr_sym[2] = {nullptr, nullptr};
    int* d_ovr_len[2] = {nullptr, nullptr};
    u64* d_post_seen = nullptr;
    u32* d_ovr_work = nullptr; size_t ovr_work_cap = 0;   // blocksum | blockbase | n_tasks | tasks
    u8* d_def = nullptr; size_t def_cap = 0;              // deflate scratch: member slots | tokens | sizes | offsets
    u8* d_eval = nullptr; size_t eval_cap = 0;            // Evaluator pre-pass: census table | hot list | text
    u32* d_ovr_corr = nullptr; size_t ovr_corr_cap = 0;   // correction chains: head[reads] | next[capacity]
    u32* d_parse = nullptr; size_t parse_cap = 0;         // FASTQ parse scratch
    u64* d_fmt = nullptr; size_t fmt_cap = 0;             // FASTQ format scratch
    u8* d_inf = nullptr; size_t inf_cap = 0;              // inflate scratch: code lengths | status | first_bad
    uint64_t units_seen = 0;                               // units submitted so far (the pre-filtering Stats' mReads)
    const void* last_corr = nullptr; int32_t last_corr_cap = 0;   // the correction list of the last submit and its capacity
    std::vector<std::string> ovr_strings[2];
    std::vector<const char*> ovr_ptrs[2];
    // exact plan (fq_text.h): the worker loop on the text, for units with letters outside ACGTN
    bool exact_all = false;                                // FASTP_GPU_EXACT=1: every unit takes it (tests)
    int* d_x_unit = nullptr; size_t x_unit_cap = 0;        // the submitted batch's exotic unit list
    u8* d_x_skip = nullptr; size_t x_skip_cap = 0;          // KernelArgs::xskip of the launch
    u32* d_al[4] = {nullptr, nullptr, nullptr, nullptr}; size_t al_cap[4] = {0, 0, 0, 0};   // merge mode: 16-byte aligned copies of a launch's rows
    int* d_ovr_diff = nullptr; size_t ovr_diff_cap = 0;     // OvrArgs::dist_diff, the four slots one behind the other (zero between launches)
    u16* d_x_len = nullptr; size_t x_len_cap = 0;          // the launch's length arrays with the text kernel's units zeroed (what the plan's kernels see)
    void* d_x_text[2] = {nullptr, nullptr}; size_t x_text_cap[2] = {0, 0};   // host submits: the raw text + offsets staged in HBM
    void* d_x_off[2] = {nullptr, nullptr}; size_t x_off_cap[2] = {0, 0};
    std::vector<int32_t> parse_exotic;                     // records of the last fastp_gpu_parse_fastq call with letters outside ACGTN
    // staging for submit_host
    void* d_stage = nullptr; size_t stage_cap = 0;
    // fastp_gpu_submit_host_async: device staging, completion event and what to finish per slot
    struct AsyncSlot {
        void* d_stage = nullptr; size_t cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
        int32_t* counts = nullptr;            // pinned: n_corrections, n_adapter_events as the device left them
        fastp_gpu_results res;                // the caller's result block (host pointers)
        const void* d_corr = nullptr;         // the batch's sparse lists in the slot's device staging: copied out when the batch
        const void* d_ev = nullptr;           // has arrived, as many entries as were written (not their whole capacity per batch)
        hipStream_t copy = nullptr;
    } aslot[FASTP_GPU_ASYNC_SLOTS];
    u64* d_phase = nullptr;   // optional per-phase cycle counters (FASTP_GPU_PHASE_TIMING=1)
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    double kernel_ms = 0.0;
    int64_t kernel_launches = 0;
    std::string err;
};

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            if (ctx) (ctx)->err = m_;                                                           \
            g_last_error = m_;                                                                  \
            return FASTP_GPU_E_HIP;                                                             \
        }                                                                                       \
    } while (0)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static int fail(fastp_gpu_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return code;
}

extern "C" const char* fastp_gpu_last_error(const fastp_gpu_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

static void drain_events(fastp_gpu_ctx* ctx) {
    for (auto& pr : ctx->pending_events) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            ctx->kernel_ms += ms;
            ctx->kernel_launches += 1;
        }
        ctx->free_events.push_back(pr);
    }
    ctx->pending_events.clear();
}

// set by fq_comm.cpp once a communicator exists: lets fastp_gpu_destroy drop the context's rank
extern "C" { void (*fastp_gpu_comm_destroy_hook)(fastp_gpu_ctx*) = nullptr; }

extern "C" void fastp_gpu_destroy(fastp_gpu_ctx* ctx) {
    if (!ctx) return;
    if (fastp_gpu_comm_destroy_hook) fastp_gpu_comm_destroy_hook(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drain_events(ctx);
    for (int k = 0; k < 2; k++) {
        if (ctx->ev_fused[k]) (void)hipEventDestroy(ctx->ev_fused[k]);
        if (ctx->ev_dup[k]) (void)hipEventDestroy(ctx->ev_dup[k]);
        if (ctx->d_dup_pos2[k]) (void)hipFree(ctx->d_dup_pos2[k]);
    }
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    for (auto& sl : ctx->aslot) {
        if (sl.d_stage) (void)hipFree(sl.d_stage);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.copy) (void)hipStreamDestroy(sl.copy);
        if (sl.counts) (void)hipHostFree(sl.counts);
    }
    for (auto& pr : ctx->free_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    void* bufs[] = {ctx->d_ov_limit, ctx->d_lowq, ctx->d_cplx, ctx->d_primes, ctx->d_planes, ctx->d_posum, ctx->d_fasta_words, ctx->d_fasta_len, ctx->d_ctr, ctx->d_slabs,
                    ctx->d_bitmap, ctx->d_dup_pos, ctx->d_table, ctx->d_need, ctx->d_dupflag, ctx->d_stage, ctx->d_phase,
                    ctx->d_ovr_table[0], ctx->d_ovr_table[1], ctx->d_ovr_sym[0], ctx->d_ovr_sym[1], ctx->d_ovr_len[0],
                    ctx->d_ovr_len[1], ctx->d_post_seen, ctx->d_ovr_work, ctx->d_parse, ctx->d_fmt, ctx->d_prefix, ctx->d_inf, ctx->d_ovr_corr, ctx->d_eval, ctx->d_def, ctx->d_setw, ctx->d_cfilter,
                    ctx->d_st_slabs, ctx->d_swin[0], ctx->d_swin[1], ctx->d_ln_slabs, ctx->d_ln_ctr,
                    ctx->d_corr_int, ctx->d_corr_chain, ctx->d_x_unit, ctx->d_x_len, ctx->d_x_skip, ctx->d_ovr_diff, ctx->d_al[0], ctx->d_al[1], ctx->d_al[2], ctx->d_al[3], ctx->d_x_text[0], ctx->d_x_text[1], ctx->d_x_off[0], ctx->d_x_off[1]};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->tail) (void)hipStreamDestroy(ctx->tail);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->ev_tail) (void)hipEventDestroy(ctx->ev_tail);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// the option family the lane kernel covers: nothing moves or edits a kept base (split plan), no step that needs an
// unbounded indexed walk along a read (adapter sequences, one-gap overlap, polyX, complexity), windows of up to 8
// bases, reads of up to 256 bases, a duplicate hash with the byte-plane table and 3-byte primes
// form 5 of the Stats kernel (fq_stats5.h): the copies of the 5-mer table its LDS layout has room for (0: it does not fit)
// (and, through hb, the columns its table holds: reads with more 16-base columns than fit are taken in blocks of hb columns)
static int stats5_copies(const DevParams& p, int lds_bytes, int* hb_out = nullptr) {
    if (env_int("FASTP_GPU_STATS_V", 5) < 5 || p.merge_lane) return 0;
    const int H16 = (p.qw_g + 3) / 4, Cp = (p.cycles + 3) & ~3;
    if (H16 > p.sw_g || H16 > 64) return 0;
    for (int nblk = 1; nblk <= 4; nblk++) {
        const int hb = (H16 + nblk - 1) / nblk;
        for (int kc = 2; kc >= 1; kc--) {
            int o = 2 * 8 * 4 * ST5_QN * hb + 2 * KMER_BINS * kc;
            o = (o + 1) & ~1;
            o += 2 * Cp * N_CLS * 2 + 2 * 128 + (1024 / 64) * 2 * ST5_WL / 2;
            if (o * 4 <= lds_bytes) {
                if (hb_out) *hb_out = hb;
                return kc;
            }
        }
    }
    return 0;
}
static bool lane_plan_supported(const DevParams& p, const HostLuts& luts) {
    if (p.front_per_read && !stats5_copies(p, 160 * 1024)) return false;   // a front per read: form 5 of the Stats kernel only
    if (!(p.stats_one_pass || p.front_lane || p.corr_lane || p.merge_lane) || p.allow_gap || p.overlapped_out) return false;
    if (p.n_fasta && p.fasta_max_len > 64) return false;   // (--adapter_fasta: each sequence like -a, in four uniform words)
    if (p.merge && !p.merge_lane) return false;
    if ((p.has_a1 && p.alen1 > 64) || (p.has_a2 && p.alen2 > 64)) return false;   // the lane kernel keeps an adapter in four uniform words
    if (p.max_len > 256 || p.sw_g > 16 || (p.qw_g & 1)) return false;
    if (p.cut_right && (p.wR < 1 || p.wR > 8)) return false;
    if (p.cut_tail && !p.cut_right && (p.wT < 1 || p.wT > 8)) return false;
    if (p.dup_enabled && !(luts.dup_nq > 0 && (p.dup_bufnum == 2 || p.dup_bufnum == 4) && p.dup_npl == 3)) return false;
    return true;
}

typedef void (*lane_kernel_fn)(LaneArgs);
template <int EXT>
static lane_kernel_fn lane_kernel_pick(int swm, int B, bool paired) {
    if (swm == 10) {
        if (B == 0) return paired ? fq_lane_kernel<10, 0, 3, true, EXT> : fq_lane_kernel<10, 0, 3, false, EXT>;
        if (B == 2) return paired ? fq_lane_kernel<10, 2, 3, true, EXT> : fq_lane_kernel<10, 2, 3, false, EXT>;
        return paired ? fq_lane_kernel<10, 4, 3, true, EXT> : fq_lane_kernel<10, 4, 3, false, EXT>;
    }
    if (B == 0) return paired ? fq_lane_kernel<16, 0, 3, true, EXT> : fq_lane_kernel<16, 0, 3, false, EXT>;
    if (B == 2) return paired ? fq_lane_kernel<16, 2, 3, true, EXT> : fq_lane_kernel<16, 2, 3, false, EXT>;
    return paired ? fq_lane_kernel<16, 4, 3, true, EXT> : fq_lane_kernel<16, 4, 3, false, EXT>;
}
// ext: 1 = adapter sequences, polyX trimming or the complexity filter are on (the instantiation that carries those steps);
// 2 = a front trim or -c as well (round 5; an instantiation of its own: with their code in it the first one spilled 77 dwords)
// 3 = --merge as well (paired only)
static int lane_ext(const DevParams& p) {
    if (p.merge_lane) return 3;
    if (p.front_lane || p.corr_lane) return 2;
    return (p.has_a1 || p.has_a2 || p.n_fasta || p.poly_x || p.complexity_filter) ? 1 : 0;
}
static lane_kernel_fn lane_kernel_merge(int swm, int B) {
    if (swm == 10) return B == 0 ? fq_lane_kernel<10, 0, 3, true, 3> : B == 2 ? fq_lane_kernel<10, 2, 3, true, 3> : fq_lane_kernel<10, 4, 3, true, 3>;
    return B == 0 ? fq_lane_kernel<16, 0, 3, true, 3> : B == 2 ? fq_lane_kernel<16, 2, 3, true, 3> : fq_lane_kernel<16, 4, 3, true, 3>;
}
static lane_kernel_fn lane_kernel_for(int swm, int B, bool paired, int ext, bool two_waves = false) {
    if (two_waves && swm == 10 && paired && ext == 3) return B == 0 ? fq_lane_pair2w_kernel<0, 3> : B == 2 ? fq_lane_pair2w_kernel<2, 3> : fq_lane_pair2w_kernel<4, 3>;
    if (two_waves && swm == 10 && paired && ext == 2) return B == 0 ? fq_lane_pair2w_kernel<0, 2> : B == 2 ? fq_lane_pair2w_kernel<2, 2> : fq_lane_pair2w_kernel<4, 2>;
    if (ext == 3) return lane_kernel_merge(swm, B);
    return ext == 2 ? lane_kernel_pick<2>(swm, B, paired) : ext == 1 ? lane_kernel_pick<1>(swm, B, paired) : lane_kernel_pick<0>(swm, B, paired);
}

extern "C" int fastp_gpu_create(const fastp_gpu_params* params, int device, fastp_gpu_ctx** out) {
    if (!params || !out) return fail(nullptr, FASTP_GPU_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = 0;
    fq::timeline("create: begin");
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FASTP_GPU_E_NO_DEVICE, "no HIP device visible - the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, FASTP_GPU_E_NO_DEVICE, "device ordinal out of range");
    fastp_gpu_ctx* ctx = new fastp_gpu_ctx();
    ctx->params = *params;
    if (params->adapter_seq_r1) ctx->adapter1 = params->adapter_seq_r1;
    if (params->adapter_seq_r2) ctx->adapter2 = params->adapter_seq_r2;
    ctx->params.adapter_seq_r1 = ctx->adapter1.c_str();
    ctx->params.adapter_seq_r2 = ctx->adapter2.c_str();
    if (params->n_adapter_fasta > 0 && params->adapter_fasta) {  // own copies: the caller's strings may go away
        for (int i = 0; i < params->n_adapter_fasta; i++) ctx->fasta_strings.push_back(params->adapter_fasta[i] ? params->adapter_fasta[i] : "");
        for (auto& f : ctx->fasta_strings) ctx->fasta_ptrs.push_back(f.c_str());
        ctx->params.adapter_fasta = ctx->fasta_ptrs.data();
    }
    if (params->overrep_enabled) {  // own copies of the seed strings
        const char* const* lists[2] = {params->overrep_seqs1, params->overrep_seqs2};
        const int ns[2] = {params->n_overrep_seqs1, params->n_overrep_seqs2};
        for (int m = 0; m < 2; m++) {
            for (int i = 0; i < ns[m] && lists[m]; i++) ctx->ovr_strings[m].push_back(lists[m][i] ? lists[m][i] : "");
            for (auto& q : ctx->ovr_strings[m]) ctx->ovr_ptrs[m].push_back(q.c_str());
        }
        ctx->params.overrep_seqs1 = ctx->ovr_ptrs[0].data();
        ctx->params.overrep_seqs2 = ctx->ovr_ptrs[1].data();
    }
    ctx->device = device;
    std::string err;
    int rc = build_dev_params(ctx->params, ctx->dp, ctx->luts, err);
    if (rc) { delete ctx; return fail(nullptr, rc, err); }
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) { delete ctx; return fail(nullptr, FASTP_GPU_E_HIP, "hipSetDevice failed"); }
    hipDeviceProp_t prop;
    he = hipGetDeviceProperties(&prop, device);
    if (he != hipSuccess) { delete ctx; return fail(nullptr, FASTP_GPU_E_HIP, "hipGetDeviceProperties failed"); }
    ctx->cus = prop.multiProcessorCount;
    // tile / launch geometry (tunable without a rebuild)
    const int lds_kb_default = (int)(prop.sharedMemPerBlock / 1024) >= 160 ? 160 : (int)(prop.sharedMemPerBlock / 1024);
    // The split plan (per-read kernel as 256-lane workgroups, several per CU, + the streaming Stats kernel) whenever no
    // option moves or edits a kept base; FASTP_GPU_SPLIT=0 keeps Stats inside the one-workgroup-per-CU fused kernel.
    // the Stats kernel as its own launch: options that leave every kept base where it was, or (lane plan only) move it by the
    // same front for every read that is written out (DevParams::front_lane)
    const bool lane_wanted = env_int("FASTP_GPU_LANE", 1) != 0 && lane_plan_supported(ctx->dp, ctx->luts);
    ctx->split = (ctx->dp.stats_one_pass || ((ctx->dp.front_lane || ctx->dp.corr_lane || ctx->dp.merge_lane) && lane_wanted)) && env_int("FASTP_GPU_SPLIT", 1) != 0;
    ctx->cfg.split = ctx->split ? 1 : 0;
    ctx->cfg.threads = env_int("FASTP_GPU_THREADS", ctx->split ? 256 : 1024);
    ctx->cfg.P = env_int("FASTP_GPU_TILE", 0);
    ctx->cfg.lds_budget = env_int("FASTP_GPU_LDS_KB", ctx->split ? std::min(40, lds_kb_default) : lds_kb_default) * 1024;
    // two tiles in flight per workgroup (each half of the waves owns one) when the halves are whole wavefronts
    ctx->cfg.halves = (env_int("FASTP_GPU_HALVES", 1) == 2 && ctx->cfg.threads % 128 == 0) ? 2 : 1;
    if (ctx->cfg.threads < 64 || ctx->cfg.threads > 1024 || (ctx->cfg.threads & 63)) {
        delete ctx;
        return fail(nullptr, FASTP_GPU_E_INVALID, "FASTP_GPU_THREADS must be a multiple of 64 in 64..1024");
    }
    if (env_int("FASTP_GPU_HASH_GENERIC", 0)) {  // tests: the multiply form of the duplicate hash (what B = 8 uses)
        ctx->luts.dup_planes.clear();
        ctx->luts.dup_nq = 0;
    }
    rc = compute_lds_layout(ctx->dp, ctx->cfg, ctx->L, err, ctx->luts.dup_nq);
    if (rc && ctx->cfg.halves == 2 && ctx->cfg.P > 0) {  // an explicit tile size that only fits once: one tile in flight
        ctx->cfg.halves = 1;
        rc = compute_lds_layout(ctx->dp, ctx->cfg, ctx->L, err, ctx->luts.dup_nq);
    }
    if (rc && ctx->split && !getenv("FASTP_GPU_LDS_KB")) {  // long reads (or a tile size asked for): the small budget holds no such tile
        ctx->cfg.lds_budget = lds_kb_default * 1024;
        rc = compute_lds_layout(ctx->dp, ctx->cfg, ctx->L, err, ctx->luts.dup_nq);
    }
    if (rc) { delete ctx; return fail(nullptr, rc, err); }
    int blocks_per_cu = env_int("FASTP_GPU_BLOCKS_PER_CU", std::max(1, (int)((160 * 1024) / (ctx->L.total * 4))));
    blocks_per_cu = std::max(1, std::min(blocks_per_cu, 2048 / ctx->cfg.threads));   // 32 wavefronts per CU
#ifndef FQ_HOSTSIM
    if (!getenv("FASTP_GPU_BLOCKS_PER_CU")) {
        // persistent workgroups: the grid must not exceed what is resident at once (registers bound it, not only LDS)
        int nb = 0;
        const void* kfn = ctx->split ? (ctx->cfg.threads > 256 ? (const void*)fq_scan_wide_kernel : (const void*)fq_scan_kernel) : (const void*)fq_fused_kernel;
        (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->L.total * 4);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, ctx->cfg.threads, (size_t)ctx->L.total * 4) == hipSuccess && nb > 0)
            blocks_per_cu = std::min(blocks_per_cu, nb);
        (void)hipGetLastError();
    }
#endif
    ctx->blocks = ctx->cus * blocks_per_cu;
    // a workgroup's packed per-cycle counters hold CYC_MAX_READS reads per Stats slot
    int tiles_per_block = CYC_MAX_READS / ctx->L.P;
    const int cap_tiles = env_int("FASTP_GPU_MAX_TILES_PER_BLOCK", 0);  // tests: force several launches
    if (cap_tiles > 0 && cap_tiles < tiles_per_block) tiles_per_block = cap_tiles;
    if (ctx->cfg.halves == 2 && tiles_per_block > 1) tiles_per_block &= ~1;  // a workgroup's two halves take tiles in pairs
    if (tiles_per_block < 1) { delete ctx; return fail(nullptr, FASTP_GPU_E_INVALID, "tile too large for the packed counters"); }
    if (ctx->split) {
        // the Stats kernel: [4][8][N_CLS][H] u64 per-cycle accumulators, k-mer and histogram counters, the increment table
        ctx->st_H = ctx->dp.qw_g / 2;
        const int st_want = env_int("FASTP_GPU_STATS_V", 5);
        ctx->st_form = (st_want == 3 && !ctx->dp.front_lane && !ctx->dp.corr_lane && !ctx->dp.merge_lane) ? 3 : 4;   // (a front / -c / --merge: form 4 only)
        if (st_want >= 5 && !ctx->dp.merge_lane) {
            // round 6's form (fq_stats5.h): the joint table [2][8][4][ST5_QN][H16] of 16-bit cell pairs, KC copies of the mate's 5-mer
            // counters, the packed cells of what the table has no cell for, the histogram of those, a list per wavefront - where it
            // fits one workgroup's LDS (reads of up to 176 bases; merge mode's third pass exists in form 4 only)
            ctx->st_H16 = (ctx->dp.qw_g + 3) / 4;
            int hb = 0;
            const int kc = stats5_copies(ctx->dp, (int)prop.sharedMemPerBlock, &hb);
            if (kc) {
                int o = 0;
                ctx->st_l_cyc = o; o += 2 * 8 * 4 * ST5_QN * hb;
                ctx->st_l_kmer = o; o += 2 * KMER_BINS * kc;
                o = (o + 1) & ~1;
                ctx->st_l_ovf = o; o += 2 * ctx->L.Cp * N_CLS * 2;
                ctx->st_l_qh = o; o += 2 * 128;
                ctx->st_l_wl = o; o += (1024 / 64) * 2 * ST5_WL / 2;   // (u16 entries)
                ctx->st_form = 5;
                ctx->st_kc = kc;
                ctx->st_Hs = hb;                     // the table's columns: all of a read's (st_H16), or a block of them
                ctx->st_lds_dwords = o;
                ctx->st_max_reads = CYC_MAX_READS;   // (a list entry holds the trip in 10 bits: 16 wavefronts x (64 / hb >= 4) units per trip, <= 256 trips)
            }
        }
        if (ctx->st_form == 5) {
        } else if (ctx->st_form == 4) {
            // round 5's form: [2][8][ST4_ROWS][Hs] u32 per-cycle cells of ONE mate, KC copies of its 5-mer counters, its histogram
            ctx->st_kc = env_int("FASTP_GPU_STATS_KC", 4);
            if (ctx->st_kc != 1 && ctx->st_kc != 2 && ctx->st_kc != 4) ctx->st_kc = 4;
            ctx->st_Hs = env_int("FASTP_GPU_STATS_HS", 0);
            if (ctx->st_Hs < ctx->st_H) ctx->st_Hs = ctx->st_H < 32 ? 32 : ctx->st_H;
            ctx->st_max_reads = ST4_MAX_READS;
            for (;;) {
                ctx->st_wl_cap = 511;
                int o = 0;
                ctx->st_l_cyc = o; o += 2 * 8 * ST4_ROWS * ctx->st_Hs;
                ctx->st_l_kmer = o; o += 2 * KMER_BINS * ctx->st_kc;
                ctx->st_l_qh = o; o += ST_QH_COPIES * 2 * 128;
                o = (o + 3) & ~3;
                ctx->st_l_lut = o; o += 4 * 256;
                ctx->st_l_mt = o; o += 18 + 2;
                ctx->st_l_wl = o; o += 1 + ctx->st_wl_cap;
                ctx->st_lds_dwords = o;
                if (2 * o * 4 <= 160 * 1024) break;                       // two workgroups per CU
                if (ctx->st_Hs > ctx->st_H) ctx->st_Hs = ctx->st_H;       // first the padding,
                else if (ctx->st_kc > 1) ctx->st_kc /= 2;                 // then the copies
                else break;
            }
        } else {
        // class stride of the per-cycle table: 32 items (= all 64 banks) where two workgroups per CU still fit
        ctx->st_max_reads = CYC_MAX_READS;
        ctx->st_Hs = ctx->st_H;
        if (env_int("FASTP_GPU_STATS_PAD", 1) && ctx->st_H < 32) ctx->st_Hs = 32;
        ctx->st_wl_cap = ctx->st_Hs > ctx->st_H ? 511 : 2047;
        for (;;) {
            const int bytes = (4 * 8 * N_CLS * ctx->st_Hs * 2 + 4 * KMER_BINS + ST_QH_COPIES * 4 * 128 + 4 * 256 + 20 + 1 + ctx->st_wl_cap + 3) * 4;
            if (ctx->st_Hs == ctx->st_H || 2 * bytes <= 160 * 1024) break;
            ctx->st_Hs = ctx->st_H;
            ctx->st_wl_cap = 2047;
        }
        int o = 0;
        ctx->st_l_cyc = o; o += 4 * 8 * N_CLS * ctx->st_Hs * 2;
        ctx->st_l_kmer = o; o += 4 * KMER_BINS;
        ctx->st_l_qh = o; o += ST_QH_COPIES * 4 * 128;
        o = (o + 3) & ~3;
        ctx->st_l_lut = o; o += 4 * 256;
        ctx->st_l_mt = o; o += 18 + 2;
        ctx->st_l_wl = o; o += 1 + ctx->st_wl_cap;
        ctx->st_lds_dwords = o;
        }
        ctx->st_slab_dwords = 4 * ctx->L.Cp * N_CLS * 2 + 4 * KMER_BINS + 4 * 128;
        ctx->st_threads = env_int("FASTP_GPU_STATS_THREADS", 1024);
        if (ctx->st_threads < 64 || ctx->st_threads > 1024 || (ctx->st_threads & 63) || ctx->st_form == 5) ctx->st_threads = 1024;
        if (ctx->st_lds_dwords * 4 > (int)prop.sharedMemPerBlock) { delete ctx; return fail(nullptr, FASTP_GPU_E_INVALID, "reads too long for the Stats kernel's LDS"); }
        int st_per_cu = std::min(2048 / ctx->st_threads, (int)((160 * 1024) / (ctx->st_lds_dwords * 4)));
        st_per_cu = env_int("FASTP_GPU_STATS_BLOCKS_PER_CU", std::max(1, st_per_cu));
        ctx->st_blocks = ctx->cus * std::max(1, st_per_cu);
        ctx->lane = env_int("FASTP_GPU_LANE", 1) != 0 && lane_plan_supported(ctx->dp, ctx->luts);
        ctx->ln_glds = env_int("FASTP_GPU_LANE_GLDS", 0);
        ctx->ln_prefetch = env_int("FASTP_GPU_LANE_PREFETCH", 0);
        ctx->ln_grab = env_int("FASTP_GPU_LANE_GRAB", 4);   // (1.400 -> 1.350 ms per 4 Mi pairs, profiles/r06_d_lane_grab_prefetch_ab.txt)
        if (ctx->lane) {
            ctx->ln_swm = ctx->dp.sw_g <= 10 ? 10 : 16;
            LaneLds& l = ctx->ln_lds;
            int o = 0;
            l.n_misc = MISC_ISIZE + ctx->dp.isize_max + 1;
            l.jkmer = 0;
            if (ctx->dp.merge_lane) {   // the merged reads' junction 5-mers: counters behind the MISC_* ones (ReduceArgs::merge_tail)
                l.jkmer = o + l.n_misc;
                l.n_misc += KMER_BINS;
            }
            l.misc = o; o += l.n_misc;
            const int lw = (ctx->dp.cycles + 2) / 2;
            l.lut_ov = o; o += lw;
            l.lut_lowq = o; o += lw;
            l.lut_cplx = o; o += lw;
            l.val4 = o; o += 256;
            o = (o + 1) & ~1;
            l.planes = o;
            l.n_planes = ctx->dp.dup_enabled ? (int)ctx->luts.dup_planes.size() : 0;
            o += l.n_planes;
            o = (o + 3) & ~3;
            l.stage = o;
            l.stage_dwords = (64 * std::max(ctx->dp.qw_g, ctx->dp.sw_g) + 4 * ctx->ln_swm + 3) & ~3;   // + the over-read of the last row
            l.part_dwords = ctx->dp.paired ? (ctx->ln_swm / 2) * 64 : 0;   // read 1 of a pair: [ln_swm / 2 words][64 lanes]
            // as many wavefronts per workgroup as the LDS holds (up to three per SIMD for reads <= 160 bases, two above)
            ctx->ln_2w = ctx->ln_swm == 10 && ctx->dp.paired && lane_ext(ctx->dp) >= 2 && env_int("FASTP_GPU_LANE_EXT_WAVES", 3) == 2;
            const int max_waves = (ctx->ln_swm > 10 || ctx->ln_2w ? 512 : 256 * FQ_LANE_WAVES) / 64;
            l.clist_dwords = (ctx->dp.corr_lane && ctx->dp.paired) ? (ctx->ln_swm / 2) * 64 : 0;   // -c: read 1's edited positions, a bit mask per lane
            int waves = (int)(((long long)prop.sharedMemPerBlock / 4 - o) / (l.stage_dwords + l.part_dwords + l.clist_dwords));
            waves = std::max(1, std::min(waves, max_waves));
            const int env_threads = env_int("FASTP_GPU_LANE_THREADS", 0);   // A/B: 256 = round 3's geometry (several workgroups per CU)
            if (env_threads >= 64 && env_threads <= max_waves * 64) waves = env_threads / 64;
            ctx->ln_threads = waves * 64;
            o += waves * l.stage_dwords;
            l.part = o;
            o += waves * l.part_dwords;
            l.clist = o;
            o += waves * l.clist_dwords;
            o = (o + 3) & ~3;
            l.ctr = o;           // the workgroup's chunk counter (LaneArgs::local_ctr)
            o += 4;
            l.sink = o;          // where the row prefetches land (LaneArgs::prefetch): 64 lanes x 4 bytes, every wavefront's
            o += 64;
            l.total = o;
            int per_cu = env_int("FASTP_GPU_LANE_BLOCKS_PER_CU", 0);
#ifndef FQ_HOSTSIM
            if (per_cu <= 0) {
                int nb = 0;
                lane_kernel_fn fn = lane_kernel_for(ctx->ln_swm, ctx->dp.dup_enabled ? ctx->dp.dup_bufnum : 0, ctx->dp.paired != 0, lane_ext(ctx->dp), ctx->ln_2w);
                (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, l.total * 4);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, ctx->ln_threads, (size_t)l.total * 4) == hipSuccess && nb > 0) per_cu = nb;
                (void)hipGetLastError();
            }
#endif
            if (per_cu <= 0) per_cu = 4;
            ctx->ln_blocks = ctx->cus * per_cu;
        }
    }
    auto set_launch_size = [&]() {
        long long mp = (long long)ctx->blocks * tiles_per_block * ctx->L.P;
        if (ctx->split) {   // the per-read kernel has no packed counters; a Stats workgroup takes <= CYC_MAX_READS units
            mp = (long long)ctx->st_blocks * CYC_MAX_READS;   // (form 4 takes such a launch as several rounds of workgroups)
            if (ctx->st_form == 5) mp *= 2;                   // (one workgroup per CU: two rounds, as many units per launch as before)
            if (cap_tiles > 0) mp = std::min(mp, (long long)ctx->blocks * cap_tiles * ctx->L.P);
        }
        if (mp > (1ll << DUP_IDX_BITS) - 1) mp = (1ll << DUP_IDX_BITS) - 1;
        if (ctx->split && ctx->dp.corr_lane) {   // the launch's correction list (launch_chunk): an entry for every base of every pair
            // 2^29 entries (6 GiB with the chains) - or, on a card with 64 GiB to spare, just under 2^30: a -c / --merge step of 4 Mi
            // pairs of 2x150 is then ONE launch instead of two (2 * index + 1 still fits an int)
            long long entries = 1ll << 29;
            size_t mfree = 0, mtotal = 0;
            const int want = env_int("FASTP_GPU_CORR_LIST_LOG2", 0);
            if (want == 30 || (want == 0 && hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mfree >= ((size_t)64 << 30))) entries = (1ll << 30) - 64;
            (void)hipGetLastError();
            mp = std::min(mp, entries / std::max(1, ctx->dp.max_len));
        }
        mp = mp / ctx->L.P * ctx->L.P;
        ctx->max_pairs_per_launch = (int)mp;
    };
    set_launch_size();
    fastp_gpu_counter_layout_for_params(&ctx->params, &ctx->cl);
    ctx->exact_all = env_int("FASTP_GPU_EXACT", 0) != 0;   // tests: every unit through the text kernel (fq_text.h)
    if (ctx->exact_all && env_int("FASTP_GPU_VERBOSE", 0)) fprintf(stderr, "fastp_gpu: FASTP_GPU_EXACT=1, every unit takes the text kernel\n");
    if (env_int("FASTP_GPU_VERBOSE", 0))
        fprintf(stderr, "fastp_gpu: %s, tile P=%d (%d rows), %d threads, LDS %d bytes, %d workgroups, %d units/launch; stats kernel form %d, %d x %d threads, LDS %d bytes\n",
                ctx->lane ? "lane plan" : (ctx->split ? "split plan" : "fused plan"), ctx->L.P, ctx->L.NR, ctx->cfg.threads, ctx->L.total * 4, ctx->blocks,
                ctx->max_pairs_per_launch, ctx->st_form, ctx->st_blocks, ctx->st_threads, ctx->st_lds_dwords * 4);
    if (env_int("FASTP_GPU_VERBOSE", 0) && ctx->lane)
        fprintf(stderr, "fastp_gpu: lane kernel %d x %d threads (%d per CU), LDS %d bytes per workgroup (stage %d + read-1 sums %d per wavefront), SWM %d, ext %d\n",
                ctx->ln_blocks, ctx->ln_threads, ctx->ln_blocks / std::max(1, ctx->cus), ctx->ln_lds.total * 4, ctx->ln_lds.stage_dwords * 4,
                ctx->ln_lds.part_dwords * 4, ctx->ln_swm, (int)lane_ext(ctx->dp));
    ctx->slab_dwords = ctx->L.acc_end - ctx->L.acc_cyc;

    *out = ctx;  // from here on errors go through destroy
    fq::timeline("create: plan chosen (device properties, occupancy queries)");
#define CREATE_TRY(call)                                               \
    do {                                                               \
        int r_ = [&]() -> int { HIP_TRY(ctx, call); return 0; }();     \
        if (r_) { std::string m = ctx->err; fastp_gpu_destroy(ctx); *out = nullptr; return fail(nullptr, r_, m); } \
    } while (0)
    {
        // FASTP_GPU_AUX_CUS compute units are set aside for the duplicate kernels (0, the default: one stream, everything
        // in order).  Measured (profiles/r02k_cu_mask_sweep.txt): only a whole SE-even group of 32 CUs keeps the fused
        // kernel's workgroups evenly placed; it hides the duplicate kernels (+4.5 % Mreads/s) but the fused kernel then
        // runs on 224 CUs (-13 % of its own rate), so it stays an option.
        const int aux_cus = (ctx->dp.dup_enabled && !ctx->dp.dedup) ? env_int("FASTP_GPU_AUX_CUS", 0) : 0;
        bool split = false;
        if (aux_cus > 0 && aux_cus * 4 <= ctx->cus && blocks_per_cu == 1) {
            const int words = (ctx->cus + 31) / 32;
            std::vector<uint32_t> m_main((size_t)words, 0u), m_aux((size_t)words, 0u);
            // workgroups are dealt evenly to the XCDs, so every XCD gives up the same number of CUs; bit i of a CU mask
            // belongs to XCD i % n_xcd (measured: profiles/r02k_cu_mask_sweep.txt)
            const int n_xcd = env_int("FASTP_GPU_XCDS", ctx->cus % 8 == 0 && ctx->cus >= 64 ? 8 : 1);
            const int per_xcd = std::max(1, aux_cus / n_xcd), slots = ctx->cus / n_xcd;
            int n_aux = 0;
            u64 slot_mask = 0;   // which CU slots of an XCD go to the aux stream
            if (const char* e = getenv("FASTP_GPU_AUX_SLOTS")) {
                for (const char* q = e; *q;) {
                    slot_mask |= 1ull << (strtol(q, (char**)&q, 10) & 63);
                    if (*q == ',') q++;
                }
            } else {
                for (int j = slots - per_xcd; j < slots; j++) slot_mask |= 1ull << j;
            }
            for (int cu = 0; cu < ctx->cus; cu++) {
                const bool to_aux = (slot_mask >> (cu / n_xcd)) & 1ull;
                if (to_aux) n_aux++;
                (to_aux ? m_aux : m_main)[(size_t)cu >> 5] |= 1u << (cu & 31);
            }
            if (hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)words, m_main.data()) == hipSuccess) {
                if (hipExtStreamCreateWithCUMask(&ctx->aux, (uint32_t)words, m_aux.data()) == hipSuccess) {
                    split = true;
                    ctx->blocks = ctx->cus - n_aux;
                    set_launch_size();
                } else {
                    (void)hipStreamDestroy(ctx->stream);
                    ctx->stream = nullptr;
                    ctx->aux = nullptr;
                }
            } else {
                ctx->stream = nullptr;
            }
            (void)hipGetLastError();
        }
        if (split) {
            for (int k = 0; k < 2; k++) {
                CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_fused[k], hipEventDisableTiming));
                CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_dup[k], hipEventDisableTiming));
            }
            if (env_int("FASTP_GPU_VERBOSE", 0)) fprintf(stderr, "fastp_gpu: %d CUs for the fused kernel, %d for Duplicate\n", ctx->blocks, ctx->cus - ctx->blocks);
        } else {
            CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        }
    }
    fq::timeline("create: launch stream created");
    CREATE_TRY(hipFuncSetAttribute(ctx->split ? (ctx->cfg.threads > 256 ? (const void*)fq_scan_wide_kernel : (const void*)fq_scan_kernel) : (const void*)fq_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->L.total * 4));
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_hash_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->L.total * 4));
    if (ctx->split) {
        CREATE_TRY(hipFuncSetAttribute((const void*)fq_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->st_lds_dwords * 4));
        if (ctx->dp.front_per_read)
            CREATE_TRY(hipFuncSetAttribute((const void*)fq_front_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((ctx->dp.paired ? 2 : 1) * 34 * (size_t)ctx->cl.cycles * 4)));
        if (ctx->dp.corr_lane)
            CREATE_TRY(hipFuncSetAttribute((const void*)fq_corr_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((ctx->dp.paired ? 2 : 1) * (33 * (size_t)ctx->cl.cycles + 128 + KMER_BINS) * 4)));
        // a slab per workgroup of the largest launch: st_blocks of them for the packed u64 form, rounds of st_blocks for the u32 form
        ctx->st_max_grid = ctx->st_form == 4 ? (ctx->max_pairs_per_launch + ST4_MAX_READS - 1) / ST4_MAX_READS + ctx->st_blocks
                                             : (ctx->st_form == 5 ? 2 * ctx->st_blocks + 2 : ctx->st_blocks);
        if (ctx->st_form == 5) CREATE_TRY(hipFuncSetAttribute((const void*)fq_stats5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->st_lds_dwords * 4));
        CREATE_TRY(hipMalloc((void**)&ctx->d_st_slabs, (size_t)ctx->st_max_grid * ctx->st_slab_dwords * 4));
        if (env_int("FASTP_GPU_DUP_OVERLAP", 1)) {
#ifdef FQ_HOSTSIM
            CREATE_TRY(hipStreamCreateWithFlags(&ctx->tail, hipStreamNonBlocking));
#else
            // a second stream that really runs beside the first (fq_probe_wait_kernel): up to six candidates, the ones that share the
            // first stream's queue are held until the search ends (the next one is then given another queue) and closed afterwards
            int* d_flag = nullptr;
            CREATE_TRY(hipMalloc((void**)&d_flag, 2 * sizeof(int)));
            std::vector<hipStream_t> rejected;
            const int tries = env_int("FASTP_GPU_TAIL_TRIES", 6);
            const bool tail_prio = env_int("FASTP_GPU_TAIL_PRIORITY", 0) != 0;
            for (int t = 0; t < std::max(1, tries) && !ctx->tail; t++) {
                hipStream_t cand = nullptr;
                if (tail_prio) {   // (A/B, FASTP_GPU_TAIL_PRIORITY=1: the tail stream's workgroups first when a CU frees up)
                    int lo = 0, hi = 0;
                    CREATE_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
                    CREATE_TRY(hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, hi));
                } else
                CREATE_TRY(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
                bool beside = tries <= 1;
                if (!beside) {
                    CREATE_TRY(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), ctx->stream));
                    CREATE_TRY(hipStreamSynchronize(ctx->stream));
                    hipLaunchKernelGGL(fq_probe_wait_kernel, dim3(1), dim3(64), 0, ctx->stream, d_flag, 2000000ll);
                    hipLaunchKernelGGL(fq_probe_set_kernel, dim3(1), dim3(64), 0, cand, d_flag);
                    CREATE_TRY(hipStreamSynchronize(cand));
                    CREATE_TRY(hipStreamSynchronize(ctx->stream));
                    int h[2] = {0, 0};
                    CREATE_TRY(hipMemcpy(h, d_flag, sizeof(h), hipMemcpyDeviceToHost));
                    beside = h[1] != 0;
                }
                if (beside) ctx->tail = cand;
                else rejected.push_back(cand);
            }
            if (!ctx->tail) { ctx->tail = rejected.back(); rejected.pop_back(); }   // (none runs beside it: correct, one behind the other)
            if (env_int("FASTP_GPU_VERBOSE", 0)) fprintf(stderr, "fastp_gpu: second stream: candidate %d of %d runs beside the first\n", (int)rejected.size() + 1, tries);
            for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
            (void)hipFree(d_flag);
#endif
            fq::timeline("create: second stream probed");
            CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_k1, hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&ctx->ev_tail, hipEventDisableTiming));
        }
        if (ctx->lane) {
            CREATE_TRY(hipMalloc((void**)&ctx->d_ln_slabs, (size_t)ctx->ln_blocks * ctx->ln_lds.n_misc * 4));
            if (env_int("FASTP_GPU_LANE_DYNAMIC", 1)) {
                CREATE_TRY(hipMalloc((void**)&ctx->d_ln_ctr, sizeof(int)));
                CREATE_TRY(hipMemset(ctx->d_ln_ctr, 0, sizeof(int)));
            }
            for (int Bh : {0, ctx->dp.dup_enabled ? ctx->dp.dup_bufnum : 0})
                CREATE_TRY(hipFuncSetAttribute((const void*)lane_kernel_for(ctx->ln_swm, Bh, ctx->dp.paired != 0, lane_ext(ctx->dp), ctx->ln_2w),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ctx->ln_lds.total * 4));
        }
    }
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_ovr_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_text_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DefLds)));
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_inflate_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(IwLds)));
    CREATE_TRY(hipFuncSetAttribute((const void*)fq_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   INF_ENTRIES * INF_LANES * 2 + INF_SBUF * INF_LANES * 4));
    fq::timeline("create: kernel attributes set");
    auto upload = [&](void** dptr, const void* src, size_t bytes) -> int {
        if (bytes == 0) { *dptr = nullptr; return 0; }
        HIP_TRY(ctx, hipMalloc(dptr, bytes));
        HIP_TRY(ctx, hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
#define CREATE_RC(expr)                                                                                    \
    do {                                                                                                   \
        int r_ = (expr);                                                                                   \
        if (r_) { std::string m = ctx->err; fastp_gpu_destroy(ctx); *out = nullptr; return fail(nullptr, r_, m); } \
    } while (0)
    CREATE_RC(upload((void**)&ctx->d_ov_limit, ctx->luts.ov_limit.data(), ctx->luts.ov_limit.size() * 2));
    CREATE_RC(upload((void**)&ctx->d_lowq, ctx->luts.lowq_limit.data(), ctx->luts.lowq_limit.size() * 2));
    CREATE_RC(upload((void**)&ctx->d_cplx, ctx->luts.cplx_min.data(), ctx->luts.cplx_min.size() * 2));
    CREATE_RC(upload((void**)&ctx->d_primes, ctx->luts.dup_primes.data(), ctx->luts.dup_primes.size() * 4));
    CREATE_RC(upload((void**)&ctx->d_planes, ctx->luts.dup_planes.data(), ctx->luts.dup_planes.size() * 4));
    CREATE_RC(upload((void**)&ctx->d_posum, ctx->luts.dup_posum.data(), ctx->luts.dup_posum.size() * 8));
    CREATE_RC(upload((void**)&ctx->d_fasta_words, ctx->luts.fasta_words.data(), ctx->luts.fasta_words.size() * 4));
    CREATE_RC(upload((void**)&ctx->d_fasta_len, ctx->luts.fasta_len.data(), ctx->luts.fasta_len.size() * 4));
    if (ctx->dp.overrep) {
        for (int m = 0; m < 2; m++) {
            CREATE_RC(upload((void**)&ctx->d_ovr_table[m], ctx->luts.ovr_table[m].data(), ctx->luts.ovr_table[m].size() * 4));
            CREATE_RC(upload((void**)&ctx->d_ovr_sym[m], ctx->luts.ovr_sym[m].data(), ctx->luts.ovr_sym[m].size()));
            CREATE_RC(upload((void**)&ctx->d_ovr_len[m], ctx->luts.ovr_len[m].data(), ctx->luts.ovr_len[m].size() * 4));
        }
        const uint64_t zero = 0;
        CREATE_RC(upload((void**)&ctx->d_post_seen, &zero, sizeof(zero)));
    }
    {
        std::vector<int64_t> zero((size_t)ctx->cl.total, 0);
        zero[0] = FASTP_GPU_ABI_VERSION;
        zero[1] = ctx->cl.cycles;
        zero[2] = ctx->dp.isize_max;
        CREATE_RC(upload((void**)&ctx->d_ctr, zero.data(), zero.size() * 8));
    }
    CREATE_TRY(hipMalloc((void**)&ctx->d_slabs, (size_t)ctx->blocks * ctx->slab_dwords * 4));
    if (ctx->dp.dup_enabled) {
        const size_t bytes = (size_t)(ctx->dp.dup_bits >> 3) * ctx->dp.dup_bufnum;
        CREATE_TRY(hipMalloc((void**)&ctx->d_bitmap, bytes));
        CREATE_TRY(hipMemsetAsync(ctx->d_bitmap, 0, bytes, ctx->stream));
    }
    if (env_int("FASTP_GPU_PHASE_TIMING", 0)) {
        CREATE_TRY(hipMalloc((void**)&ctx->d_phase, 16 * sizeof(u64)));
        CREATE_TRY(hipMemsetAsync(ctx->d_phase, 0, 16 * sizeof(u64), ctx->stream));
    }
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
    fq::timeline("create: end (tables uploaded, bloom filter cleared)");
    return FASTP_GPU_OK;
}

// debugging aid (not part of the drop-in surface): cycles spent per phase of the fused kernel,
// summed over workgroups, when the context was created with FASTP_GPU_PHASE_TIMING=1
extern "C" int fastp_gpu_debug_phase_cycles(fastp_gpu_ctx* ctx, uint64_t* out16) {
    if (!ctx || !out16 || !ctx->d_phase) return FASTP_GPU_E_INVALID;
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(out16, ctx->d_phase, 16 * sizeof(u64), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemset(ctx->d_phase, 0, 16 * sizeof(u64)));
    return FASTP_GPU_OK;
}

// make `st` wait for what the aux stream still has in flight (duplicate flags, counters, bitmap)
static int join_aux(fastp_gpu_ctx* ctx, hipStream_t st) {
    if (!ctx->aux_pending) return 0;
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_dup[ctx->aux_last], 0));
    if (st == ctx->stream) ctx->aux_pending = false;
    return 0;
}
static int sync_main(fastp_gpu_ctx* ctx) {
    int rc = join_aux(ctx, ctx->stream);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int ensure(fastp_gpu_ctx* ctx, void** buf, size_t* cap, size_t need) {
    if (*cap >= need) return 0;
    if (*buf && ctx->aux_pending) { int rj = join_aux(ctx, ctx->stream); if (rj) return rj; }
    if (*buf) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(*buf)); *buf = nullptr; *cap = 0; }
    size_t want = need + need / 4;
    HIP_TRY(ctx, hipMalloc(buf, want));
    *cap = want;
    return 0;
}

static int get_events(fastp_gpu_ctx* ctx, hipEvent_t* a, hipEvent_t* b) {
    if (ctx->pending_events.size() >= 64) drain_events(ctx);
    if (!ctx->free_events.empty()) {
        *a = ctx->free_events.back().first;
        *b = ctx->free_events.back().second;
        ctx->free_events.pop_back();
        return 0;
    }
    HIP_TRY(ctx, hipEventCreate(a));
    HIP_TRY(ctx, hipEventCreate(b));
    return 0;
}

// one launch: at most ctx->max_pairs_per_launch units
// Stats::statRead's overrepresentation analysis (stats.cpp:270-288) of one launch, after its records exist
static int launch_overrep(fastp_gpu_ctx* ctx, const KernelArgs& a, int n, hipStream_t st, const fastp_gpu_batch* b = nullptr) {
    const fastp_gpu_counter_layout& cl = ctx->cl;
    int rc;
    if (ctx->dp.overrep) {
        OvrArgs o;
        memset(&o, 0, sizeof(o));
        o.n = n;
        o.paired = ctx->dp.paired;
        o.dedup = ctx->dp.dedup;
        o.sampling = ctx->dp.overrep_sampling;
        o.pre_mod = (u32)(ctx->units_seen % (uint64_t)o.sampling);
        o.sw_g = ctx->dp.sw_g;
        o.qw_g = ctx->dp.qw_g;
        o.merge = ctx->dp.merge;
        o.merge_include_unmerged = ctx->dp.merge_include_unmerged;
        o.pair = a.pair;
        for (int m = 0; m < 2; m++) { o.seq[m] = a.seq[m]; o.qual[m] = a.qual[m]; o.len[m] = a.len[m]; o.res[m] = a.res[m]; }
        if (b && b->n_exotic > 0) {   // units with letters outside ACGTN: the counting kernel reads their symbols from the text
            o.first = a.first;
            o.x_unit = ctx->d_x_unit;
            o.x_n = b->n_exotic;
            o.x_dense = b->exotic_dense;
            for (int m = 0; m < 2; m++) { o.x_text[m] = b->exotic_text[m]; o.x_off[m] = b->exotic_off[m]; }
        }
        const int nb = (n + 255) / 256;
        const int task_cap = 4 * (n / o.sampling + 2);
        rc = ensure(ctx, (void**)&ctx->d_ovr_work, &ctx->ovr_work_cap, ((size_t)2 * nb + 1 + task_cap) * 4);
        if (rc) return rc;
        o.blocksum = ctx->d_ovr_work;
        o.blockbase = ctx->d_ovr_work + nb;
        o.n_tasks = ctx->d_ovr_work + 2 * nb;
        o.tasks = ctx->d_ovr_work + 2 * nb + 1;
        o.task_cap = task_cap;
        o.post_seen = ctx->d_post_seen;
        for (int m = 0; m < 2; m++) {
            OvrMate& M = o.mate[m];
            M.n_seeds = m == 0 ? ctx->params.n_overrep_seqs1 : (ctx->dp.paired ? ctx->params.n_overrep_seqs2 : 0);
            M.eval_len = m == 0 ? ctx->params.eval_seq_len1 : ctx->params.eval_seq_len2;
            for (int k = 0; k < OVR_STEPS; k++) { M.steps[k] = ctx->luts.ovr_steps[m][k]; M.pw[k] = ctx->luts.ovr_pw[m][k]; }
            M.table = ctx->d_ovr_table[m];
            M.table_mask = (u32)(ctx->luts.ovr_table[m].size() / 2 - 1);
            M.seed_sym = ctx->d_ovr_sym[m];
            M.seed_len = ctx->d_ovr_len[m];
        }
        o.ctr = ctx->d_ctr;
        for (int k = 0; k < 4; k++) { o.o_count[k] = cl.overrep_count[k]; o.o_dist[k] = cl.overrep_dist[k]; }
        int diff_rows = 0;
        // mOverRepSeqDist as a difference array (OvrArgs::dist_diff): measured, no gain - configs[4] 4.90 ms without, 5.11 ms with
        // (the fold's 130 lanes walk 251 positions each; profiles/r05_ovr_diff_ab.txt) - off unless asked for
        if (env_int("FASTP_GPU_OVR_DIFF", 0)) {
            size_t words = 0;
            for (int k = 0; k < 4; k++) words += (size_t)o.mate[k >> 1].n_seeds * (size_t)(o.mate[k >> 1].eval_len + 1);
            const size_t had = ctx->ovr_diff_cap;
            rc = ensure(ctx, (void**)&ctx->d_ovr_diff, &ctx->ovr_diff_cap, words * 4 + 4);
            if (rc) return rc;
            if (ctx->ovr_diff_cap != had) HIP_TRY(ctx, hipMemsetAsync(ctx->d_ovr_diff, 0, ctx->ovr_diff_cap, st));   // (a new buffer: the fold leaves it zero)
            size_t at = 0;
            for (int k = 0; k < 4; k++) {
                const OvrMate& M = o.mate[k >> 1];
                o.dist_diff[k] = M.n_seeds ? ctx->d_ovr_diff + at : nullptr;
                at += (size_t)M.n_seeds * (size_t)(M.eval_len + 1);
                diff_rows += M.n_seeds;
            }
        }
        // the post-filtering Stats analyse the corrected reads: from the engine's own list of this launch where it keeps one (-c with
        // the Stats kernel as its own launch: sized for an edit at every base, it cannot overflow), else from the caller's
        // (a launch with units for the text kernel: that kernel's edits are in the caller's list only - it counts its units' POST
        // Stats itself, fq_corr_stats_kernel must not see them - so such a launch reads the caller's list as before)
        const bool own_list = a.corr_int != nullptr && a.corr_int_cap > 0 && !(b && b->n_exotic > 0) && !ctx->exact_all;
        if (ctx->dp.correction && !own_list && !(a.corrections && a.corr_capacity > 0))
            return fail(ctx, FASTP_GPU_E_INVALID, "overrepresentation analysis with correction needs the correction list in the results");
        if (own_list || (a.corrections && a.corr_capacity > 0)) {
            const size_t reads = (size_t)(ctx->dp.paired ? 2 : 1) * n;
            const int list_cap = own_list ? a.corr_int_cap : a.corr_capacity;
            rc = ensure(ctx, (void**)&ctx->d_ovr_corr, &ctx->ovr_corr_cap, (reads + (size_t)list_cap) * 4);
            if (rc) return rc;
            o.corr = own_list ? a.corr_int : a.corrections;
            o.n_corr = own_list ? a.n_corr_int : a.n_corrections;
            o.corr_cap = list_cap;
            o.first = a.first;
            o.corr_head = ctx->d_ovr_corr;
            o.corr_next = ctx->d_ovr_corr + reads;
            HIP_TRY(ctx, hipMemsetAsync(o.corr_head, 0, reads * 4, st));
            if (own_list)   // (capacity far above the fill: a grid-stride walk over what the list holds)
                hipLaunchKernelGGL(fq_corr_link_kernel, dim3(std::min(4096, (list_cap + 255) / 256)), dim3(256), 0, st, o);
            else
                hipLaunchKernelGGL(fq_ovr_corr_link_kernel, dim3((list_cap + 255) / 256), dim3(256), 0, st, o);
            HIP_TRY(ctx, hipGetLastError());
        }
        HIP_TRY(ctx, hipMemsetAsync(o.n_tasks, 0, 4, st));
        hipLaunchKernelGGL(fq_ovr_pass_kernel, dim3(nb), dim3(256), 16, st, o);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(fq_ovr_scan_kernel, dim3(1), dim3(1024), 1024 * 4, st, o, nb);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(fq_ovr_tasks_kernel, dim3(nb), dim3(256), 64, st, o);
        HIP_TRY(ctx, hipGetLastError());
        {   // LDS plan of the counting kernel: symbols of one task per lane, then the seed tables that still fit
            const int longest = ctx->dp.max_len * (ctx->dp.merge ? 2 : 1);
            int lds_bytes = (((longest + 4) * OVR_SYM_STRIDE + 15) / 16) * 16;   // (+ 4 rows: a trip's four slides read unguarded)
            o.sym_cap = longest;
            if (lds_bytes > 120 * 1024) { o.sym_cap = 0; lds_bytes = 0; }   // reads too long to stage: the global path
            for (int m = 0; m < 2; m++) {
                const size_t tb = ctx->luts.ovr_table[m].size() * 4;
                o.table_lds[m] = -1;
                if (o.mate[m].n_seeds > 0 && tb > 0 && lds_bytes + (int)tb <= 150 * 1024) {
                    o.table_lds[m] = lds_bytes / 4;
                    lds_bytes += (int)tb;
                }
            }
            hipLaunchKernelGGL(fq_ovr_count_kernel, dim3((task_cap + OVR_TPB - 1) / OVR_TPB), dim3(OVR_BLOCK), (size_t)lds_bytes, st, o);
        }
        HIP_TRY(ctx, hipGetLastError());
        if (diff_rows > 0) {
            hipLaunchKernelGGL(fq_ovr_dist_kernel, dim3((diff_rows + 63) / 64), dim3(64), 0, st, o);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    ctx->units_seen += (uint64_t)n;
    return FASTP_GPU_OK;
}

// what a launch does with Duplicate::checkPair/checkRead
enum ChunkMode {
    CHUNK_STREAM,    // one stream: decide inside this launch (the normal path)
    CHUNK_PASS1,     // sharded run, pass 1: insert + scan state, no decision (--dedup: nothing else; otherwise
                     // the worker loop runs here, its records just lack the RF_DUP flag)
    CHUNK_PASS2,     // sharded run, pass 2: decide from the scan state and the prefix bitmaps (--dedup: then the
                     // worker loop; otherwise the decision is patched into pass 1's records)
    CHUNK_OVERREP,   // the deferred overrepresentation analysis only
};

static int launch_chunk(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, int first, int n, const fastp_gpu_results* res,
                        hipStream_t st, ChunkMode mode = CHUNK_STREAM, u8* scan_state = nullptr) {
    KernelArgs a;
    memset(&a, 0, sizeof(a));
    a.p = ctx->dp;
    a.lut.ov_limit = (const u16*)ctx->d_ov_limit;
    a.lut.lowq_limit = ctx->d_lowq;
    a.lut.cplx_min = ctx->d_cplx;
    a.lut.dup_primes = ctx->d_primes;
    a.lut.dup_planes = ctx->d_planes;
    a.lut.dup_posum = ctx->d_posum;
    a.lut.fasta_words = ctx->d_fasta_words;
    a.lut.fasta_len = ctx->d_fasta_len;
    a.L = ctx->L;
    a.magic_sw = magic_for((u32)ctx->L.SW);
    a.magic_qwg = magic_for((u32)ctx->dp.qw_g);
    a.magic_swg = magic_for((u32)ctx->dp.sw_g);
    {   // vector (16-byte) tile copies + register prefetch need aligned rows and a tile that fits the registers
        const size_t qchunks = (size_t)ctx->L.NR * ctx->dp.qw_g / 4, schunks = (size_t)ctx->L.NR * ctx->dp.sw_g / 4;
        const size_t tile_threads = (size_t)ctx->cfg.threads / ctx->L.halves;   // the waves that stage one tile
        bool ok = (ctx->L.P % 2 == 0) && (first % 2 == 0) && qchunks <= (size_t)PF_Q * tile_threads &&
                  schunks <= (size_t)PF_S * tile_threads && (size_t)ctx->L.NR <= tile_threads &&
                  !env_int("FASTP_GPU_NO_PREFETCH", 0);
        const void* ptrs[4] = {b->seq1, b->qual1, ctx->dp.paired ? b->seq2 : b->seq1, ctx->dp.paired ? b->qual2 : b->qual1};
        for (const void* q : ptrs) ok = ok && (((uintptr_t)q & 15u) == 0);
        // the L2 warm-up of the next tile (tile_warm) is off by default: it paid while the tile fetch waited per chunk,
        // now it costs 2.5 % and makes every line cross HBM 1.35 times (profiles/r02p_prefetch_ab.txt)
        a.prefetch = ok ? (env_int("FASTP_GPU_PREFETCH_AHEAD", 0) ? 1 : 2) : 0;
    }
    a.n = n;
    a.first = first;
    a.batch_flags = b->flags;
    const size_t swg = ctx->dp.sw_g, qwg = ctx->dp.qw_g;
    a.seq[0] = (const u32*)b->seq1 + (size_t)first * swg;
    a.qual[0] = (const u32*)b->qual1 + (size_t)first * qwg;
    a.len[0] = b->len1 + first;
    if (res) a.res[0] = (u32*)res->r1 + (size_t)first * 3;
    if (ctx->dp.paired) {
        a.seq[1] = (const u32*)b->seq2 + (size_t)first * swg;
        a.qual[1] = (const u32*)b->qual2 + (size_t)first * qwg;
        a.len[1] = b->len2 + first;
        if (res) {
            a.res[1] = (u32*)res->r2 + (size_t)first * 3;
            a.pair = (u32*)res->pair + (size_t)first * 2;
        }
    }
    if (res) {
        a.corrections = (ctx->dp.correction && res->corrections && res->n_corrections) ? (u32*)res->corrections : nullptr;
        a.corr_capacity = res->corrections_capacity;
        a.n_corrections = res->n_corrections;
        a.adapter_events = (ctx->dp.n_fasta && res->adapter_events && res->n_adapter_events) ? (u32*)res->adapter_events : nullptr;
        a.adapter_events_capacity = res->adapter_events_capacity;
        a.n_adapter_events = res->n_adapter_events;
    }
    // Units with letters outside ACGTN (fastp_gpu_batch::exotic_*; FASTP_GPU_EXACT=1: every unit) take the text kernel
    // (fq_text.h).  The plan's kernels still run over the whole launch, on a copy of the length arrays in which those units
    // are EMPTY; the text kernel then runs once over the listed units: it takes back what an empty unit added to the
    // counters (the same loop on an empty unit, sign -1), adds the real unit, and overwrites the unit's records and hash
    // values.  Duplicate's kernels run once over the whole launch afterwards: input order holds across both kinds.
    int xk0 = 0, xk1 = 0;
    if (b->n_exotic > 0) {
        xk0 = (int)(std::lower_bound(b->exotic_unit, b->exotic_unit + b->n_exotic, first) - b->exotic_unit);
        xk1 = (int)(std::lower_bound(b->exotic_unit, b->exotic_unit + b->n_exotic, first + n) - b->exotic_unit);
    }
    const bool exact = n > 0 && (ctx->exact_all || xk1 > xk0);
    const u16* true_len[2] = {a.len[0], a.len[1]};
    if (exact) {
        const int mates = ctx->dp.paired ? 2 : 1;
        int rx = ensure(ctx, (void**)&ctx->d_x_len, &ctx->x_len_cap, (size_t)2 * n * sizeof(u16));
        if (rx) return rx;
        for (int m = 0; m < mates; m++) {
            u16* copy = ctx->d_x_len + (size_t)m * n;
            if (ctx->exact_all) HIP_TRY(ctx, hipMemsetAsync(copy, 0, (size_t)n * sizeof(u16), st));
            else HIP_TRY(ctx, hipMemcpyAsync(copy, a.len[m], (size_t)n * sizeof(u16), hipMemcpyDeviceToDevice, st));
            a.len[m] = copy;
        }
        // the lane plan runs the text kernel BESIDE its kernels: the lane kernel is told which units not to write (xskip)
        u8* skip = nullptr;
        if (ctx->lane && ctx->split && ctx->tail && mode == CHUNK_STREAM && !ctx->dp.dedup) {
            rx = ensure(ctx, (void**)&ctx->d_x_skip, &ctx->x_skip_cap, (size_t)n);
            if (rx) return rx;
            skip = ctx->d_x_skip;
            HIP_TRY(ctx, hipMemsetAsync(skip, ctx->exact_all ? 1 : 0, (size_t)n, st));
            a.xskip = skip;
        }
        if (!ctx->exact_all) {
            TextMaskArgs mk;
            mk.units = ctx->d_x_unit + xk0;
            mk.count = xk1 - xk0;
            mk.first = first;
            mk.skip = skip;
            mk.len[0] = ctx->d_x_len;
            mk.len[1] = mates == 2 ? ctx->d_x_len + n : nullptr;
            hipLaunchKernelGGL(fq_text_mask_kernel, dim3((mk.count + 255) / 256), dim3(256), 0, st, mk);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    // scan state of the whole batch: positions [b->n][B] u64, then masks [b->n] u8
    u64* scan_pos = scan_state ? (u64*)scan_state + (size_t)first * ctx->dp.dup_bufnum : nullptr;
    u8* scan_mask = scan_state ? scan_state + (size_t)b->n * ctx->dp.dup_bufnum * 8 + first : nullptr;
    // the worker loop on the context's own streams: Duplicate's kernels go to the aux stream and overlap the next launch
    const bool piped = ctx->aux && st == ctx->stream && mode == CHUNK_STREAM && ctx->dp.dup_enabled && !ctx->dp.dedup && !exact;
    const int par = (int)(ctx->launch_seq & 1);
    if (!piped) {
        int rj = join_aux(ctx, st);
        if (rj) return rj;
    }
    u64* dup_pos_buf = nullptr;
    if (ctx->dp.dup_enabled && mode != CHUNK_OVERREP) {
        if (piped) {
            // this buffer was last read by the resolve of launch k - 2
            if (ctx->ev_dup_set[par]) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_dup[par], 0));
            int rc = ensure(ctx, (void**)&ctx->d_dup_pos2[par], &ctx->dup_pos2_cap[par], (size_t)n * ctx->dp.dup_bufnum * 8);
            if (rc) return rc;
            dup_pos_buf = ctx->d_dup_pos2[par];
            // the resolve of launch k - 1 ORs duplicate flags into its result rows: wait for it if this launch writes the same rows
            const char* lo = (const char*)a.res[0];
            const char* hi = lo + (size_t)n * sizeof(fastp_gpu_read_result);
            if (ctx->aux_pending && ctx->ev_dup_set[par ^ 1] && lo < (const char*)ctx->last_res[1] && (const char*)ctx->last_res[0] < hi)
                HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_dup[par ^ 1], 0));
        } else {
            int rc = ensure(ctx, (void**)&ctx->d_dup_pos, &ctx->dup_pos_cap, (size_t)n * ctx->dp.dup_bufnum * 8);
            if (rc) return rc;
            dup_pos_buf = ctx->d_dup_pos;
        }
        a.dup_pos = dup_pos_buf;
    }
    a.split = ctx->split ? 1 : 0;
    const bool corr_lane = ctx->split && ctx->dp.corr_lane && mode != CHUNK_OVERREP && !(mode == CHUNK_PASS1 && ctx->dp.dedup);
    if (corr_lane) {
        // the engine's own correction list of this launch.  A pair can have as many edits as its overlap is long (only the first
        // 50 bases are held to the mismatch limit, overlapanalysis.cpp:34-44): the list is sized for that - it cannot overflow,
        // and set_launch_size keeps it below 2^29 entries (2^30 where the card has the memory to spare: laid out for 288 GB of HBM)
        const size_t cap = (size_t)n * (size_t)ctx->dp.max_len;
        const int rc0 = ensure(ctx, (void**)&ctx->d_corr_int, &ctx->corr_int_cap, (cap * 2 + 4) * 4);
        if (rc0) return rc0;
        a.n_corr_int = (int*)ctx->d_corr_int;
        a.corr_int = ctx->d_corr_int + 4;
        a.corr_int_cap = (int)std::min<size_t>(cap, 0x7FFFFFFF);
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_corr_int, 0, 16, st));
    }
    if (ctx->split) {
        const size_t need = ((size_t)n * 4 + 255) & ~(size_t)255;
        if (need > ctx->swin_cap) {
            for (int m = 0; m < 2; m++) {
                if (ctx->d_swin[m]) HIP_TRY(ctx, hipFree(ctx->d_swin[m]));
                ctx->d_swin[m] = nullptr;
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_swin[m], need + need / 4));
            }
            ctx->swin_cap = need + need / 4;
        }
        a.swin_out[0] = ctx->d_swin[0];
        a.swin_out[1] = ctx->d_swin[1];
    }
    a.phase_cycles = ctx->d_phase;
    // The product library has no switch that changes a result: FASTP_GPU_DEBUG_SKIP (steps left out of the kernels, for the measured
    // floors under profiles/) exists only in a library built with -DFQ_PROFILE_ABLATION (tools/build_ablation.sh).  Bit 512 is a TEST
    // switch that leaves every result as it is (each merged read's second part counted by the lane kernel, fq_lane.h).
#ifdef FQ_PROFILE_ABLATION
    a.debug_skip = (u32)env_int("FASTP_GPU_DEBUG_SKIP", 0) & ~512u;
#else
    a.debug_skip = 0;
#endif
    if (env_int("FASTP_GPU_TEST_MERGE_SLOW", 0)) a.debug_skip |= 512u;
    a.slabs = ctx->d_slabs;
    a.slab_dwords = ctx->slab_dwords;
    a.tiles = (n + ctx->L.P - 1) / ctx->L.P;
    a.half_skew = env_int("FASTP_GPU_HALF_SKEW", 6);
    a.half_naps = env_int("FASTP_GPU_HALF_NAPS", 1);
    const int wg_tiles = (a.tiles + ctx->L.halves - 1) / ctx->L.halves;   // tiles are dealt to workgroups `halves` at a time
    const int grid = wg_tiles < ctx->blocks ? wg_tiles : ctx->blocks;
    // the stage + hash pre-pass of --dedup runs the whole workgroup on one tile at a time
    auto whole = [](KernelArgs k) { k.L.halves = 1; return k; };
    hipStream_t st_main = st;
    const fastp_gpu_counter_layout& cl = ctx->cl;
    int rc;

    // the lane kernel copies rows with 16-byte accesses: a batch whose arrays are not 16-byte aligned takes the tile kernel
    bool use_lane = ctx->lane;
    {
        const void* ptrs[4] = {a.seq[0], a.qual[0], ctx->dp.paired ? a.seq[1] : a.seq[0], ctx->dp.paired ? a.qual[1] : a.qual[0]};
        for (const void* q : ptrs) use_lane = use_lane && (((uintptr_t)q & 15u) == 0);
    }
    if (ctx->lane && !use_lane && ctx->dp.merge_lane) {
        // --merge has this plan only in its lane form: the launch's rows move to 16-byte aligned arrays of the engine
        for (int k = 0; k < 4; k++) {
            const size_t bytes = (size_t)n * (size_t)((k & 1) ? ctx->dp.qw_g : ctx->dp.sw_g) * 4;
            rc = ensure(ctx, (void**)&ctx->d_al[k], &ctx->al_cap[k], bytes + 256);
            if (rc) return rc;
            const u32* src = (k & 1) ? a.qual[k >> 1] : a.seq[k >> 1];
            if (bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_al[k], src, bytes, hipMemcpyDeviceToDevice, st));
            if (k & 1) a.qual[k >> 1] = ctx->d_al[k]; else a.seq[k >> 1] = ctx->d_al[k];
        }
        use_lane = true;
    }
    // Duplicate::checkPair/checkRead over this chunk, in input order (probe + resolve)
    // --dedup without the hash pre-pass (round 5, lane plan, plain stream mode): the lane kernel hashes and claims as without
    // --dedup, Duplicate's tail decides, fq_dedup_apply_kernel takes the duplicates out again before the Stats kernel counts
    // (not in merge mode: a pair that merges is written out whatever Duplicate says, peprocessor.cpp:523-535)
    const bool dedup_folded = ctx->dp.dedup && use_lane && mode == CHUNK_STREAM && !piped && !exact && !env_int("FASTP_GPU_DUP_TABLE", 0) &&
                              !ctx->dp.merge_lane && env_int("FASTP_GPU_DEDUP_FOLD", 1) &&
                              env_int("FASTP_GPU_CLAIM_FUSED", 1);   // (the fold IS the fused claim: without it the hash pre-pass decides)
    // the claim step inside the fused kernel: plain stream mode, one or two bloom buffers (the lane kernel: four as well), the
    // context's own stream order
    // ... and a launch with units for the text kernel when that kernel runs beside the lane kernel (exact_early below): the lane kernel
    // claims nothing for such a unit (KernelArgs::xskip), the text kernel claims its units' bits itself (fq_text.h t_claim) and
    // Duplicate's tail - which orders the claims by unit index, whoever fired them first - runs behind both
    const bool exact_early = exact && use_lane && a.xskip != nullptr && !piped && env_int("FASTP_GPU_EXACT_EARLY", 1) != 0;
    const bool claim_fused = ctx->dp.dup_enabled && (!ctx->dp.dedup || dedup_folded) && mode == CHUNK_STREAM && !piped &&
                             (ctx->dp.dup_bufnum <= 2 || dedup_folded) && !env_int("FASTP_GPU_DUP_TABLE", 0) && env_int("FASTP_GPU_CLAIM_FUSED", 1) &&
                             (!exact || (exact_early && env_int("FASTP_GPU_EXACT_CLAIM", 1)));
    // the text kernel (fq_text.h): a wavefront per listed unit, its texts in the wavefront's stretch of LDS, Stats' per-base
    // counters in the workgroup's LDS tables (added to d_ctr once), everything else straight into d_ctr
    auto launch_exact = [&](int hash_only, hipStream_t xst = nullptr) -> int {
        hipStream_t st = xst ? xst : st_main;   // (shadows the launch stream: the text kernel may run beside the Stats kernel)
        TextArgs e;
        memset(&e, 0, sizeof(e));
        e.k = a;
        const fastp_gpu_counter_layout& c = ctx->cl;
        e.c.filter = c.filter_stats; e.c.adapter_reads = c.adapter_reads; e.c.adapter_bases = c.adapter_bases;
        e.c.polyx_reads = c.polyx_reads; e.c.polyx_bases = c.polyx_bases; e.c.correction = c.correction;
        e.c.corrected_reads = c.corrected_reads; e.c.merged = c.merged_pairs; e.c.isize = c.isize;
        for (int k = 0; k < 4; k++) e.c.stats[k] = c.stats[k];
        e.c.st_reads = c.st_reads; e.c.st_length_sum = c.st_length_sum; e.c.st_qual_hist = c.st_qual_hist;
        e.c.st_kmer = c.st_kmer; e.c.st_cycle = c.st_cycle; e.c.cycles = c.cycles;
        e.ctr = ctx->d_ctr;
        e.k.len[0] = true_len[0];
        e.k.len[1] = true_len[1];
        e.x_n = b->n_exotic;
        e.x_unit = ctx->d_x_unit;
        e.x_all = ctx->exact_all ? 1 : 0;
        e.x_k0 = xk0;
        e.x_count = ctx->exact_all ? n : xk1 - xk0;
        e.x_dense = b->exotic_dense;
        for (int m = 0; m < 2; m++) { e.x_text[m] = b->exotic_text[m]; e.x_off[m] = b->exotic_off[m]; }
        e.ML = (ctx->dp.max_len + 8 + 7) & ~7;
        e.hash_only = hash_only;
        // Stats' per-base tables of a workgroup in LDS when they fit beside the wavefronts' texts (not the hash pre-pass, which
        // counts nothing; reads too long for it add to the block with global atomics)
        const size_t text_bytes = (size_t)TEXT_WAVES * text_wave_bytes(e.ML);
        const int slot_dwords = 34 * (int)c.cycles + 1024 + 128;
        const int slots = !ctx->dp.paired ? 2 : ctx->dp.merge ? 3 : 4;   // the Stats objects a unit can reach
        const bool lds_tables = !hash_only && (size_t)slots * slot_dwords * 4 + text_bytes <= (size_t)150 * 1024 && env_int("FASTP_GPU_EXACT_LDS", 1);
        e.lds_slot_dwords = lds_tables ? slot_dwords : 0;
        e.lds_slots = lds_tables ? slots : 0;
        const int blocks = std::max(1, std::min((e.x_count + TEXT_WAVES - 1) / TEXT_WAVES, env_int("FASTP_GPU_EXACT_BLOCKS", ctx->cus)));
        hipLaunchKernelGGL(fq_text_kernel, dim3(blocks), dim3(64 * TEXT_WAVES), (size_t)e.lds_slots * e.lds_slot_dwords * 4 + text_bytes, st, e);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    bool dup_prepared = false;
    auto launch_dup = [&](u8* dupflag, bool scan = false, hipStream_t st = nullptr, int stage = 0) -> int {
        // stage 0: everything; 1: only the buffers + clears (before a fused kernel that claims); 2: what follows that kernel;
        // 3: of that only fq_dup_losers_kernel, 4: only the two kernels behind it (the two on different streams, launch_chunk)
        if (!st) st = st_main;
        DupArgs d;
        memset(&d, 0, sizeof(d));
        if (scan) { d.scan_pos = scan_pos; d.scan_mask = scan_mask; }
        d.dup_pos = dup_pos_buf;
        d.posum = ctx->d_posum;
        d.len[0] = a.len[0];
        d.len[1] = a.len[1];
        d.n = n;
        d.B = ctx->dp.dup_bufnum;
        d.bits = ctx->dp.dup_bits;
        d.bitmap = ctx->d_bitmap;
        int lg = 10;
        while ((1ull << lg) < (size_t)n * d.B * 2) lg++;
        int r2 = ensure(ctx, (void**)&ctx->d_table, &ctx->table_cap, (size_t)8 << lg);
        if (r2) return r2;
        r2 = ensure(ctx, (void**)&ctx->d_need, &ctx->need_cap, (size_t)n);
        if (r2) return r2;
        d.table = ctx->d_table;
        d.table_log2 = lg;
        d.need = ctx->d_need;
        d.res[0] = a.res[0];
        d.res[1] = a.res[1];
        d.dupflag = dupflag;
        d.paired = ctx->dp.paired;
        d.ctr_total = ctx->d_ctr + cl.dup_total;
        d.ctr_dups = ctx->d_ctr + cl.dup_count;
        // Stage 1's clears (128 MB of table for 4 Mi pairs: 21 + 6 us) are needed by the kernels of stage 2 only, not by the kernel
        // that claims: where stage 2 will run on the tail stream they go there, beside the per-read kernel instead of in front of it
        // (that stream is past the previous launch's tail by then; the launch stream joins it at the end of every launch)
        hipStream_t cst = (stage == 1 && ctx->split && ctx->tail && !dedup_folded && n > 0 && env_int("FASTP_GPU_DUP_CLEAR_TAIL", 1)) ? ctx->tail : st;
        if (stage < 2) HIP_TRY(ctx, hipMemsetAsync(d.table, 0xFF, (size_t)8 << lg, cst));
        const int g2 = std::max(1, (n + 255) / 256);  // one unit per lane: the kernels are chains of dependent random accesses
        if (env_int("FASTP_GPU_DUP_TABLE", 0)) {      // the first form: probe (read + table insert for every unit) -> resolve
            hipLaunchKernelGGL(fq_dup_probe_kernel, dim3(g2), dim3(256), 0, st, d);
            HIP_TRY(ctx, hipGetLastError());
            hipLaunchKernelGGL(fq_dup_resolve_kernel, dim3(std::max(1, (n + 1023) / 1024)), dim3(1024), 16, st, d);
            HIP_TRY(ctx, hipGetLastError());
            return 0;
        }
        r2 = ensure(ctx, (void**)&ctx->d_setw, &ctx->setw_cap, (size_t)n);
        if (r2) return r2;
        if (!ctx->d_cfilter) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cfilter, (size_t)1 << (DUP_CF_LOG2 - 3)));
        d.setw = ctx->d_setw;
        d.cfilter = ctx->d_cfilter;
        if (stage < 2) HIP_TRY(ctx, hipMemsetAsync(d.cfilter, 0, (size_t)1 << (DUP_CF_LOG2 - 3), cst));
        if (stage == 1) {
            a.claim_won = ctx->d_need;
            a.dup_bitmap = ctx->d_bitmap;
            a.dup_bits = ctx->dp.dup_bits;
            dup_prepared = true;
            return 0;
        }
        if (stage == 2 || stage == 3) hipLaunchKernelGGL(fq_dup_losers_kernel, dim3(g2), dim3(256), 0, st, d);
        else if (stage != 4) hipLaunchKernelGGL(fq_dup_claim_kernel, dim3(g2), dim3(256), 0, st, d);
        HIP_TRY(ctx, hipGetLastError());
        if (stage == 3) return 0;
        hipLaunchKernelGGL(fq_dup_winners_kernel, dim3(g2), dim3(256), 0, st, d);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(fq_dup_finish_kernel, dim3(std::max(1, (n + 1023) / 1024)), dim3(1024), 16, st, d);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };

    if (mode == CHUNK_PASS1 && ctx->dp.dedup) {
        hipLaunchKernelGGL(fq_hash_kernel, dim3(grid), dim3(ctx->cfg.threads), (size_t)ctx->L.total * 4, st, whole(a));
        HIP_TRY(ctx, hipGetLastError());
        if (exact) {
            rc = launch_exact(1);
            if (rc) return rc;
        }
        return launch_dup(nullptr, true);
    }
    // the overrepresentation analysis reads the rows by their TRUE lengths, and the listed units' symbols from their text
    auto overrep = [&](hipStream_t ost = nullptr) -> int {
        KernelArgs ao = a;
        ao.len[0] = true_len[0];
        ao.len[1] = true_len[1];
        return launch_overrep(ctx, ao, n, ost ? ost : st, b);
    };
    if (mode == CHUNK_OVERREP) return overrep();
    if (mode == CHUNK_PASS2) {
        // the decision comes from pass 1's scan state + the preceding shards' bitmaps; nothing is hashed again
        if (ctx->dp.dedup) {
            rc = ensure(ctx, (void**)&ctx->d_dupflag, &ctx->dupflag_cap, (size_t)n);
            if (rc) return rc;
        }
        DupFinalArgs d;
        memset(&d, 0, sizeof(d));
        d.scan_pos = scan_pos;
        d.scan_mask = scan_mask;
        d.prefix = ctx->has_prefix ? ctx->d_prefix : nullptr;
        d.bits = ctx->dp.dup_bits;
        d.n = n;
        d.B = ctx->dp.dup_bufnum;
        d.dupflag = ctx->dp.dedup ? ctx->d_dupflag : nullptr;
        d.res[0] = a.res[0];
        d.res[1] = a.res[1];
        d.paired = ctx->dp.paired;
        d.ctr_total = ctx->d_ctr + cl.dup_total;
        d.ctr_dups = ctx->d_ctr + cl.dup_count;
        hipLaunchKernelGGL(fq_dup_final_kernel, dim3((n + 255) / 256), dim3(256), 16, st, d);
        HIP_TRY(ctx, hipGetLastError());
        if (!ctx->dp.dedup) return FASTP_GPU_OK;  // the records are pass 1's
        a.dup_pos = nullptr;
        a.dupflag = ctx->d_dupflag;
    } else if (ctx->dp.dedup && !dedup_folded) {
        // --dedup: hash pass -> duplicate decision -> fused kernel reads the decision
        rc = ensure(ctx, (void**)&ctx->d_dupflag, &ctx->dupflag_cap, (size_t)n);
        if (rc) return rc;
        hipLaunchKernelGGL(fq_hash_kernel, dim3(grid), dim3(ctx->cfg.threads), (size_t)ctx->L.total * 4, st, whole(a));
        HIP_TRY(ctx, hipGetLastError());
        if (exact) {
            rc = launch_exact(1);
            if (rc) return rc;
        }
        rc = launch_dup(ctx->d_dupflag);
        if (rc) return rc;
        a.dup_pos = nullptr;
        a.dupflag = ctx->d_dupflag;
    }

    if (claim_fused) {
        rc = launch_dup(nullptr, false, nullptr, 1);
        if (rc) return rc;
    }
    int ln_grid = 0;
    // The text kernel of the units with letters outside ACGTN, lane plan (round 5): it runs on the tail stream from here on,
    // BESIDE the lane kernel (which counts such a unit as the empty unit it sees and writes nothing of it, KernelArgs::xskip)
    // and the Stats kernel (to which it is an empty read); Duplicate's kernels wait for both.  A launch with a handful of such
    // units used to wait 2.6 - 3.1 ms for one lane's walk between the two kernels.
    if (!exact_early) a.xskip = nullptr;
    if (exact_early) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail, ctx->ev_k1, 0));
        rc = launch_exact(0, ctx->tail);
        if (rc) return rc;
    }
    hipEvent_t e0, e1;
    rc = get_events(ctx, &e0, &e1);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(e0, st));
    {
        FusedArgs fa;
        fa.h[0] = a;
        fa.h[1] = a;
        fa.h[1].L = layout_for_half(a.L, 1);
        if (use_lane) {
            LaneArgs la;
            la.k = a;
            la.k.slabs = ctx->d_ln_slabs;
            la.k.slab_dwords = ctx->ln_lds.n_misc;
            la.l = ctx->ln_lds;
            la.chunk_ctr = ctx->d_ln_ctr;
            la.glds = ctx->ln_glds;
            la.prefetch = ctx->ln_prefetch;
            la.grab = ctx->ln_grab;
            la.local_ctr = env_int("FASTP_GPU_LANE_DYNAMIC", 2) == 2 ? 1 : 0;
            la.pool = 0;
            la.pool_base = 0;
            la.pool_grab = 1;
            if (la.local_ctr && la.chunk_ctr) {   // the chunks no workgroup owns (LaneArgs::pool)
                // A/B, off: profiles/r06_x_lane_chunk_pool_ab.txt, r06_y_*: an 8th of the chunks -2 %, a 16th 0, a 32nd .. a 128th +0.2 % on the
                // headline (noise), the single-end lines, -c and configs[4] 0.4 - 2 % SLOWER - the CUs' equal shares already end
                // together; the asks of the pool cost what little imbalance there is
                const int lg = env_int("FASTP_GPU_LANE_POOL_LOG2", 0);
                const int chunks = (n + 63) >> 6;
                la.pool = lg > 0 && lg < 31 ? chunks >> lg : 0;
                if (ctx->ln_pool_base > 0x60000000) {   // (the counter only counts up: back to zero long before it could wrap)
                    HIP_TRY(ctx, hipMemsetAsync(la.chunk_ctr, 0, sizeof(int), st));
                    ctx->ln_pool_base = 0;
                }
                la.pool_base = ctx->ln_pool_base;
                la.pool_grab = std::max(1, la.pool >> 11);   // at most ~2 k asks per launch
                // every wavefront asks once more than it gets: the counter ends at most (asks that get chunks) + wavefronts beyond the base
                ctx->ln_pool_base += (la.pool + la.pool_grab - 1) / la.pool_grab + ctx->ln_blocks * (ctx->ln_threads >> 6) + 64;
            }
            la.post1 = ctx->d_ctr + cl.stats[1];
            la.st_qual_hist = cl.st_qual_hist; la.st_kmer = cl.st_kmer; la.st_cycle = cl.st_cycle; la.cycles = cl.cycles;
            if (la.chunk_ctr && !la.local_ctr) HIP_TRY(ctx, hipMemsetAsync(la.chunk_ctr, 0, sizeof(int), st));
            const int Bh = (ctx->dp.dup_enabled && (a.dup_pos || a.claim_won) && !(a.debug_skip & 2u)) ? ctx->dp.dup_bufnum : 0;
            lane_kernel_fn fn = lane_kernel_for(ctx->ln_swm, Bh, ctx->dp.paired != 0, lane_ext(ctx->dp), ctx->ln_2w);
            ln_grid = std::max(1, std::min(ctx->ln_blocks, (n + 255) / 256));
            hipLaunchKernelGGL(fn, dim3(ln_grid), dim3(ctx->ln_threads), (size_t)ctx->ln_lds.total * 4, st, la);
        } else if (ctx->split && ctx->cfg.threads > 256) hipLaunchKernelGGL(fq_scan_wide_kernel, dim3(grid), dim3(ctx->cfg.threads), (size_t)ctx->L.total * 4, st, fa);
        else if (ctx->split) hipLaunchKernelGGL(fq_scan_kernel, dim3(grid), dim3(ctx->cfg.threads), (size_t)ctx->L.total * 4, st, fa);
        else hipLaunchKernelGGL(fq_fused_kernel, dim3(grid), dim3(ctx->cfg.threads), (size_t)ctx->L.total * 4, st, fa);
    }
    HIP_TRY(ctx, hipGetLastError());
    // The text kernel of the units with letters outside ACGTN runs behind the plan's kernel (the records and hash values of its
    // units are overwritten) - and, in the split plans, BESIDE the Stats kernel (round 5): the Stats kernel needs nothing of it
    // (the listed units are empty reads to it, the text kernel adds their Stats itself), only Duplicate's kernels do, so both go
    // to the tail stream.  A launch with a handful of such units used to wait 2.6 - 3.1 ms for one lane's walk before anything else ran.
    bool exact_on_tail = false;
    if (exact_early) {
        // (launched in front of the plan's kernel) Duplicate's kernels need that kernel's hash values as well: the tail stream joins it
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail, ctx->ev_k1, 0));
        exact_on_tail = true;
    } else if (exact && ctx->split && ctx->tail && mode == CHUNK_STREAM && !piped && !ctx->dp.dedup && n > 0 && env_int("FASTP_GPU_EXACT_TAIL", 1)) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail, ctx->ev_k1, 0));
        rc = launch_exact(0, ctx->tail);
        if (rc) return rc;
        exact_on_tail = true;
    } else if (exact) {
        rc = launch_exact(0);
        if (rc) return rc;
    }
    // the slab folds (set up here: the per-read kernel's MISC_* fold may go to the tail stream in front of Duplicate's kernels)
    ReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.L = ctx->L;
    r.isize_max = ctx->dp.isize_max;
    r.one_pass = ctx->dp.stats_one_pass || (ctx->split && (ctx->dp.front_lane || ctx->dp.corr_lane || ctx->dp.merge_lane));
    r.merge_tail = (ctx->split && ctx->dp.merge_lane) ? 1 : 0;
    if (ctx->split && ctx->dp.front_lane) { r.front[0] = ctx->dp.lane_front1; r.front[1] = ctx->dp.lane_front2; }
    r.ctr = ctx->d_ctr;
    r.o_filter = cl.filter_stats; r.o_adapter_reads = cl.adapter_reads; r.o_adapter_bases = cl.adapter_bases;
    r.o_polyx_reads = cl.polyx_reads; r.o_polyx_bases = cl.polyx_bases; r.o_correction = cl.correction;
    r.o_corrected_reads = cl.corrected_reads; r.o_merged = cl.merged_pairs; r.o_isize = cl.isize;
    for (int s = 0; s < 4; s++) r.o_stats[s] = cl.stats[s];
    r.st_reads = cl.st_reads; r.st_length_sum = cl.st_length_sum; r.st_qual_hist = cl.st_qual_hist;
    r.st_kmer = cl.st_kmer; r.st_cycle = cl.st_cycle; r.cycles = cl.cycles;
    const int n_stats = 4 * N_CLS * ctx->L.Cp + 4 * KMER_BINS + 4 * 128, n_misc = MISC_ISIZE + ctx->dp.isize_max + 1 + (r.merge_tail ? (int)KMER_BINS : 0);
    auto fold = [&](int parts, int nblocks, hipStream_t st) -> int {
        if (nblocks <= 0) return 0;
        r.parts = parts;
        r.nblocks = nblocks;
        const int items = ((parts & 1) ? n_stats : 0) + ((parts & 2) ? n_misc : 0);
        const int rgroups = (nblocks + REDUCE_GROUP - 1) / REDUCE_GROUP;
        hipLaunchKernelGGL(fq_reduce_kernel, dim3(((items + 255) / 256) * rgroups), dim3(256), 0, st, r);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    };
    // the claim ran inside that kernel: what is left of Duplicate (losers / winners / finish) needs nothing of the Stats
    // kernel and runs beside it on its own stream; the launch stream joins it before anything else touches the records
    bool dup_tail_launched = false;
    bool dedup_applied = false;
    bool misc_folded = false;
    if (dedup_folded && dup_prepared && n > 0) {
        // --dedup: the decisions are needed before the Stats kernel classifies a base as kept - on the launch stream
        rc = ensure(ctx, (void**)&ctx->d_dupflag, &ctx->dupflag_cap, (size_t)n);
        if (rc) return rc;
        rc = launch_dup(ctx->d_dupflag, false, st, 2);
        if (rc) return rc;
        DedupApplyArgs da;
        memset(&da, 0, sizeof(da));
        da.n = n;
        da.paired = ctx->dp.paired;
        da.dupflag = ctx->d_dupflag;
        for (int m = 0; m < 2; m++) {
            da.res[m] = a.res[m];
            da.swin[m] = ctx->d_swin[m];
            da.st_reads[m] = ctx->d_ctr + cl.stats[2 * m + 1] + cl.st_reads;
            da.st_lensum[m] = ctx->d_ctr + cl.stats[2 * m + 1] + cl.st_length_sum;
        }
        hipLaunchKernelGGL(fq_dedup_apply_kernel, dim3((n + 255) / 256), dim3(256), 16, st, da);
        HIP_TRY(ctx, hipGetLastError());
        dedup_applied = true;
    } else if (ctx->split && ctx->tail && dup_prepared && n > 0) {
        // The Stats kernel's workgroups own every CU (a 1024-lane workgroup at 128 VGPRs is the whole register file): a kernel on
        // the tail stream gets through when one of the Stats kernel's two rounds of workgroups ends, ONE kernel per such moment
        // (profiles/r06_s_step_timeline.txt, r06_t_step_timeline.txt: each tail kernel "takes" 0.43 - 0.50 ms, 0.01 - 0.03 alone).
        // A/B, FASTP_GPU_DUP_LOSERS_FIRST=1: fq_dup_losers_kernel on the launch stream IN FRONT of the Stats kernel, so that
        // winners gets through between the rounds - measured SLOWER (profiles/r06_u_losers_first_ab.txt: step 2.409 -> 2.440 ms):
        // winners is then dispatched together with the Stats kernel's first round and shares the CUs with it from the start (the
        // Stats kernel 0.89 -> 0.95 ms), finish still waits for its end.  Off.
        // (never with units for the text kernel on the tail stream: the Stats kernel would wait for that kernel too)
        const bool losers_first = env_int("FASTP_GPU_DUP_LOSERS_FIRST", 0) != 0 && mode != CHUNK_PASS1 && !exact;
        if (losers_first) {
            // (the tail stream holds the clears of this launch's table and filter: the launch stream waits for them first)
            HIP_TRY(ctx, hipEventRecord(ctx->ev_tail, ctx->tail));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_tail, 0));
            rc = launch_dup(nullptr, false, st, 3);
            if (rc) return rc;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail, ctx->ev_k1, 0));
        // The MISC_* fold needs the per-read kernel only: FIRST on the tail stream.  Behind Duplicate's kernels - which the Stats
        // kernel's workgroups starve until its last round ends (profiles/r06_s_step_timeline.txt: losers 0.45 ms, winners 0.50 ms
        // beside it, 0.01 / 0.03 alone) - it was 27 + 9 us at the very end of every step, with nothing else on the chip.
        if (env_int("FASTP_GPU_MISC_FOLD_FIRST", losers_first ? 0 : 1)) {
            r.slabs = use_lane ? ctx->d_ln_slabs : ctx->d_slabs;
            r.slab_dwords = use_lane ? ctx->ln_lds.n_misc : ctx->slab_dwords;
            r.off_misc = use_lane ? 0 : ctx->L.acc_misc - ctx->L.acc_cyc;
            rc = fold(2, use_lane ? ln_grid : grid, ctx->tail);
            if (rc) return rc;
            misc_folded = true;
        }
        rc = launch_dup(nullptr, mode == CHUNK_PASS1, ctx->tail, losers_first ? 4 : 2);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_tail, ctx->tail));
        dup_tail_launched = true;
    } else if (exact_on_tail) {
        // Duplicate's kernels behind the text kernel, on its stream (the claim is not fused into a launch that has such units)
        if (ctx->dp.dup_enabled) {
            rc = launch_dup(nullptr, false, ctx->tail, 0);
            if (rc) return rc;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_tail, ctx->tail));
        dup_tail_launched = true;
    }
    // The overrepresentation analysis needs the records (and --dedup's decisions), nothing of the Stats kernel: on the tail stream
    // BESIDE it (round 5; behind Duplicate's tail / the text kernel when they are there - they write record flags), joined at the end
    bool ovr_early = false;
    if (ctx->dp.overrep && !(b->flags & FASTP_GPU_BATCH_DEFER_OVERREP) && ctx->split && ctx->tail && mode == CHUNK_STREAM && !piped && n > 0 &&
        env_int("FASTP_GPU_OVR_TAIL", 1)) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail, ctx->ev_k1, 0));
        rc = overrep(ctx->tail);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_tail, ctx->tail));
        ovr_early = true;           // (the launch stream waits for the tail stream below)
    }
    int st_grid = 0;
    if (ctx->split && n > 0) {
        // Stats::statRead of the launch's units: a workgroup takes a run of consecutive units (at most CYC_MAX_READS:
        // packed counters; at least 64 so that small launches do not pay a 43 KB slab per handful of reads)
        StatsArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.n = n;
        sa.paired = ctx->dp.paired;
        sa.sw_g = ctx->dp.sw_g;
        sa.qw_g = ctx->dp.qw_g;
        sa.H = ctx->st_H; sa.Hs = ctx->st_Hs;
        sa.magic_H = magic_for((u32)ctx->st_H);
        sa.Cp = ctx->L.Cp;
        // whole rounds of the resident workgroups: 4.19 M pairs = 3 rounds of 512 workgroups with 2731 units each in form 4
        const int rounds = (int)(((long long)n + (long long)ctx->st_blocks * ctx->st_max_reads - 1) / ((long long)ctx->st_blocks * ctx->st_max_reads));
        int upb = (n + ctx->st_blocks * rounds - 1) / (ctx->st_blocks * rounds);
        upb = std::max(upb, std::min(n, 64));
        if (upb > ctx->st_max_reads) return fail(ctx, FASTP_GPU_E_INVALID, "launch too large for the Stats kernel");
        sa.units_per_block = upb;
        st_grid = (n + upb - 1) / upb;
        if (st_grid > ctx->st_max_grid) return fail(ctx, FASTP_GPU_E_INVALID, "launch too large for the Stats kernel's slabs");
        sa.form = ctx->st_form;
        sa.kc = ctx->st_kc;
        if (ctx->dp.front_lane) { sa.front[0] = ctx->dp.lane_front1; sa.front[1] = ctx->dp.lane_front2; }
        sa.merge = ctx->dp.merge_lane;
        for (int m = 0; m < 2; m++) { sa.seq[m] = a.seq[m]; sa.qual[m] = a.qual[m]; sa.swin[m] = ctx->d_swin[m]; }
        sa.l_cyc = ctx->st_l_cyc; sa.l_kmer = ctx->st_l_kmer; sa.l_qh = ctx->st_l_qh; sa.l_lut = ctx->st_l_lut; sa.l_mt = ctx->st_l_mt;
        sa.l_wl = ctx->st_l_wl; sa.wl_cap = ctx->st_wl_cap;
        sa.l_total = ctx->st_lds_dwords;
        sa.H16 = ctx->st_H16; sa.magic_H16 = (ctx->st_form == 5 && ctx->st_Hs) ? magic_for((u32)ctx->st_Hs) : 0u; sa.l_ovf = ctx->st_l_ovf;   // (the magic of the table's columns)
        sa.front_per_read = ctx->dp.front_per_read;
        sa.fr_stride = ctx->dp.front_per_read ? 3 : 0;
        for (int m = 0; m < 2; m++) sa.fr_rec[m] = ctx->dp.front_per_read ? (const u32*)a.res[m] : ctx->d_swin[m];
        sa.slabs = ctx->d_st_slabs;
        sa.slab_dwords = ctx->st_slab_dwords;
        sa.debug_skip = a.debug_skip;
        if (!(a.debug_skip & 16u)) {
            if (ctx->st_form == 5) hipLaunchKernelGGL(fq_stats5_kernel, dim3(st_grid), dim3(ctx->st_threads), (size_t)ctx->st_lds_dwords * 4, st, sa);
            else hipLaunchKernelGGL(fq_stats_kernel, dim3(st_grid), dim3(ctx->st_threads), (size_t)ctx->st_lds_dwords * 4, st, sa);
            HIP_TRY(ctx, hipGetLastError());
        } else {
            st_grid = 0;
        }
    }
    if (ctx->dp.front_per_read && ctx->lane && ctx->split && n > 0 && st_grid > 0) {
        // --cut_front on the lane plan: the reads whose own front is beyond the mate's common one (fq_stats5.h, front_stats_body)
        FrontStatsArgs fs;
        memset(&fs, 0, sizeof(fs));
        fs.n = n;
        fs.paired = ctx->dp.paired;
        fs.sw_g = ctx->dp.sw_g;
        fs.qw_g = ctx->dp.qw_g;
        for (int m = 0; m < 2; m++) {
            fs.seq[m] = a.seq[m];
            fs.qual[m] = a.qual[m];
            fs.swin[m] = ctx->d_swin[m];
            fs.rec[m] = (const u32*)a.res[m];
            fs.post[m] = ctx->d_ctr + ctx->cl.stats[2 * m + 1];
        }
        fs.front0[0] = ctx->dp.lane_front1;
        fs.front0[1] = ctx->dp.lane_front2;
        fs.st_cycle = ctx->cl.st_cycle;
        fs.cycles = ctx->cl.cycles;
        const size_t reads = (size_t)n * (ctx->dp.paired ? 2 : 1);
        const size_t lds_bytes = (size_t)(ctx->dp.paired ? 2 : 1) * 34 * (size_t)ctx->cl.cycles * 4;
        const int fgrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)ctx->cus * 4, (reads + 255) / 256));
        hipLaunchKernelGGL(fq_front_stats_kernel, dim3(fgrid), dim3(256), lds_bytes, st, fs);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (corr_lane && ctx->split && n > 0 && st_grid > 0) {
        // -c: the corrected positions' share of the POST Stats moves from the original base / quality to the corrected one
        const size_t reads = (size_t)n * (ctx->dp.paired ? 2 : 1);
        rc = ensure(ctx, (void**)&ctx->d_corr_chain, &ctx->corr_chain_cap, (reads + (size_t)a.corr_int_cap) * 4);
        if (rc) return rc;
        OvrArgs lo;
        memset(&lo, 0, sizeof(lo));
        lo.n = n;
        lo.first = a.first;
        lo.paired = ctx->dp.paired;
        lo.corr = a.corr_int;
        lo.n_corr = a.n_corr_int;
        lo.corr_cap = a.corr_int_cap;
        lo.corr_head = ctx->d_corr_chain;
        lo.corr_next = ctx->d_corr_chain + reads;
        HIP_TRY(ctx, hipMemsetAsync(lo.corr_head, 0, reads * 4, st));
        // (one lane per possible entry would be n x limit lanes; the list is short - a grid-stride walk over what it holds)
        hipLaunchKernelGGL(fq_corr_link_kernel, dim3(std::min(4096, (a.corr_int_cap + 255) / 256)), dim3(256), 0, st, lo);
        HIP_TRY(ctx, hipGetLastError());
        CorrStatsArgs cs;
        memset(&cs, 0, sizeof(cs));
        cs.n = n;
        cs.paired = ctx->dp.paired;
        cs.sw_g = ctx->dp.sw_g;
        cs.qw_g = ctx->dp.qw_g;
        for (int m = 0; m < 2; m++) {
            cs.seq[m] = a.seq[m];
            cs.qual[m] = a.qual[m];
            cs.swin[m] = ctx->d_swin[m];
            cs.post[m] = ctx->d_ctr + cl.stats[ctx->dp.merge_lane ? 1 : 2 * m + 1];
        }
        cs.merge = ctx->dp.merge_lane;
        cs.front[0] = ctx->dp.front_lane ? ctx->dp.lane_front1 : 0;
        cs.front[1] = ctx->dp.front_lane ? ctx->dp.lane_front2 : 0;
        cs.corr = a.corr_int;
        cs.corr_head = lo.corr_head;
        cs.corr_next = lo.corr_next;
        cs.st_qual_hist = cl.st_qual_hist;
        cs.st_kmer = cl.st_kmer;
        cs.st_cycle = cl.st_cycle;
        cs.cycles = cl.cycles;
        {   // persistent workgroups: the deltas of a workgroup's reads meet in its LDS tables first (corr_stats_body)
            const size_t lds_bytes = (size_t)(ctx->dp.paired ? 2 : 1) * (33 * (size_t)cl.cycles + 128 + KMER_BINS) * 4;
            const int cgrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)ctx->cus * 2, (reads + 1023) / 1024));
            hipLaunchKernelGGL(fq_corr_stats_kernel, dim3(cgrid), dim3(1024), lds_bytes, st, cs);
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(e1, st));
    ctx->pending_events.push_back({e0, e1});

    if (ctx->split) {
        // the Stats kernel's slabs: per-cycle u64s, k-mer counters, one histogram counter per (slot, character)
        r.slabs = ctx->d_st_slabs;
        r.slab_dwords = ctx->st_slab_dwords;
        r.off_kmer = 4 * N_CLS * ctx->L.Cp * 2;
        r.off_qh = r.off_kmer + 4 * KMER_BINS;
        r.qh_stride = 1;
        r.qh_count = 0;
        rc = fold(1, st_grid, st);
        if (rc) return rc;
        // the per-read kernel's slabs: the MISC_* counters only
        r.slabs = use_lane ? ctx->d_ln_slabs : ctx->d_slabs;
        r.slab_dwords = use_lane ? ctx->ln_lds.n_misc : ctx->slab_dwords;
        r.off_misc = use_lane ? 0 : ctx->L.acc_misc - ctx->L.acc_cyc;
        // (needs nothing of the Stats kernel either: beside it, behind Duplicate's tail, when that stream is in use)
        if (!misc_folded) {
            rc = fold(2, use_lane ? ln_grid : grid, dup_tail_launched ? ctx->tail : st);
            if (rc) return rc;
        }
        if (dup_tail_launched) HIP_TRY(ctx, hipEventRecord(ctx->ev_tail, ctx->tail));
    } else {
        r.slabs = ctx->d_slabs;
        r.slab_dwords = ctx->slab_dwords;
        r.off_kmer = ctx->L.acc_kmer - ctx->L.acc_cyc;
        r.off_qh = ctx->L.acc_qh - ctx->L.acc_cyc;
        r.qh_stride = QT_DWORDS;
        r.qh_count = QT_COUNT;
        r.off_misc = ctx->L.acc_misc - ctx->L.acc_cyc;
        rc = fold(3, grid, st);
        if (rc) return rc;
    }

    if (piped) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fused[par], st));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fused[par], 0));
        rc = launch_dup(nullptr, false, ctx->aux);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_dup[par], ctx->aux));
        ctx->ev_dup_set[par] = true;
        ctx->aux_pending = true;
        ctx->aux_last = par;
        ctx->last_res[0] = a.res[0];
        ctx->last_res[1] = (const char*)a.res[0] + (size_t)n * sizeof(fastp_gpu_read_result);
        ctx->launch_seq++;
    } else if (dup_tail_launched) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_tail, 0));
    } else if (dedup_applied) {
        // (Duplicate's tail ran in front of the Stats kernel)
    } else if (ctx->dp.dup_enabled && !ctx->dp.dedup) {
        rc = launch_dup(nullptr, mode == CHUNK_PASS1, nullptr, dup_prepared ? 2 : 0);
        if (rc) return rc;
    }
    if (ovr_early && !dup_tail_launched && !piped) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_tail, 0));
    if ((b->flags & FASTP_GPU_BATCH_DEFER_OVERREP) || ovr_early) return FASTP_GPU_OK;
    return overrep();
}


static int submit_chunks(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, const fastp_gpu_results* res, void* hip_stream,
                         ChunkMode mode, u8* scan_state) {
    if (!ctx || !b) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    const bool need_res = !(mode == CHUNK_PASS1 && ctx->dp.dedup);
    if (need_res && !res) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    if (b->n < 0) return fail(ctx, FASTP_GPU_E_INVALID, "negative batch size");
    if (!b->seq1 || !b->qual1 || !b->len1 || (need_res && !res->r1)) {
        if (b->n > 0) return fail(ctx, FASTP_GPU_E_INVALID, "missing read-1 buffers");
    }
    if (ctx->dp.paired && b->n > 0 && (!b->seq2 || !b->qual2 || !b->len2 || (need_res && (!res->r2 || !res->pair))))
        return fail(ctx, FASTP_GPU_E_INVALID, "paired engine needs read-2 buffers and pair results");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    const bool worker_loop = mode == CHUNK_STREAM || (mode == CHUNK_PASS1 && !ctx->dp.dedup) || (mode == CHUNK_PASS2 && ctx->dp.dedup);
    if (worker_loop) {
        ctx->last_corr = res->corrections;
        ctx->last_corr_cap = res->corrections ? res->corrections_capacity : 0;
        if (res->n_corrections) HIP_TRY(ctx, hipMemsetAsync(res->n_corrections, 0, sizeof(int32_t), st));
        if (ctx->dp.n_fasta && b->n > 0 && (!res->adapter_events || !res->n_adapter_events))
            return fail(ctx, FASTP_GPU_E_INVALID, "adapter_fasta needs an adapter event list in the results");
        if (res->n_adapter_events) HIP_TRY(ctx, hipMemsetAsync(res->n_adapter_events, 0, sizeof(int32_t), st));
    }
    if (b->n_exotic > 0) {
        // units with letters outside ACGTN: the text kernel (fq_text.h) takes them, launch by launch (launch_chunk)
        if (!b->exotic_unit || !b->exotic_text[0] || !b->exotic_off[0] || (ctx->dp.paired && (!b->exotic_text[1] || !b->exotic_off[1])))
            return fail(ctx, FASTP_GPU_E_INVALID, "n_exotic > 0 needs exotic_unit, exotic_text and exotic_off");
        for (int k = 0; k < b->n_exotic; k++)
            if (b->exotic_unit[k] < 0 || b->exotic_unit[k] >= b->n || (k && b->exotic_unit[k] <= b->exotic_unit[k - 1]))
                return fail(ctx, FASTP_GPU_E_INVALID, "exotic_unit must be ascending unit indexes of the batch");
        int rc = ensure(ctx, (void**)&ctx->d_x_unit, &ctx->x_unit_cap, (size_t)b->n_exotic * sizeof(int));
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_x_unit, b->exotic_unit, (size_t)b->n_exotic * sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));   // the list is the caller's (pageable) memory
    }
    // split into equally sized launches (each a multiple of the tile size)
    const int launches = (b->n + ctx->max_pairs_per_launch - 1) / ctx->max_pairs_per_launch;
    int per = launches ? (b->n + launches - 1) / launches : 0;
    per = (per + ctx->L.P - 1) / ctx->L.P * ctx->L.P;
    for (int first = 0; first < b->n; first += per) {
        const int n = std::min(per, b->n - first);
        int rc = launch_chunk(ctx, b, first, n, res, st, mode, scan_state);
        if (rc) return rc;
    }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_submit_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, fastp_gpu_results* res,
                                       void* hip_stream) {
    return submit_chunks(ctx, b, res, hip_stream, CHUNK_STREAM, nullptr);
}

// ---- sharded runs (include/fastp_gpu.h "Sharded runs") -------------------------------------
extern "C" int64_t fastp_gpu_dup_scan_bytes(const fastp_gpu_ctx* ctx, int32_t n) {
    if (!ctx || n < 0) return -1;
    return (int64_t)n * ctx->dp.dup_bufnum * 8 + (((int64_t)n + 15) & ~(int64_t)15);
}

extern "C" int fastp_gpu_submit_pass1_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, void* scan_state,
                                             fastp_gpu_results* res, void* hip_stream) {
    if (!ctx) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    if (!ctx->dp.dup_enabled) return submit_chunks(ctx, b, res, hip_stream, CHUNK_STREAM, nullptr);  // nothing depends on earlier units
    if (!scan_state || ((uintptr_t)scan_state & 7u)) return fail(ctx, FASTP_GPU_E_INVALID, "scan state must be an 8-byte aligned device buffer");
    return submit_chunks(ctx, b, res, hip_stream, CHUNK_PASS1, (u8*)scan_state);
}

extern "C" int fastp_gpu_submit_pass2_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, const void* scan_state,
                                             fastp_gpu_results* res, void* hip_stream) {
    if (!ctx) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    if (!ctx->dp.dup_enabled) return FASTP_GPU_OK;  // pass 1 did everything
    if (!scan_state || ((uintptr_t)scan_state & 7u)) return fail(ctx, FASTP_GPU_E_INVALID, "scan state must be an 8-byte aligned device buffer");
    return submit_chunks(ctx, b, res, hip_stream, CHUNK_PASS2, (u8*)scan_state);
}

extern "C" int fastp_gpu_overrep_device(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, const fastp_gpu_results* res,
                                        void* hip_stream) {
    return submit_chunks(ctx, b, res, hip_stream, CHUNK_OVERREP, nullptr);
}

extern "C" int64_t fastp_gpu_dup_bitmap_bytes(const fastp_gpu_ctx* ctx) {
    if (!ctx) return -1;
    return ctx->dp.dup_enabled ? (int64_t)ctx->dp.dup_bufnum * (int64_t)(ctx->dp.dup_bits / 8) : 0;
}

extern "C" int fastp_gpu_dup_bitmap_export(fastp_gpu_ctx* ctx, void* dst_device) {
    if (!ctx) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    const int64_t bytes = fastp_gpu_dup_bitmap_bytes(ctx);
    if (bytes == 0) return FASTP_GPU_OK;
    if (!dst_device) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rj_ = join_aux(ctx, ctx->stream); if (rj_) return rj_; }
    HIP_TRY(ctx, hipMemcpyAsync(dst_device, ctx->d_bitmap, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

// the counterpart of the export: this engine's bitmaps <- an image (a context that takes over another's stream,
// fastp_gpu_stream.h's re-plan; Duplicate's state is nothing but these bits, duplicate.h:34-37)
extern "C" int fastp_gpu_dup_bitmap_import(fastp_gpu_ctx* ctx, const void* src_device) {
    if (!ctx) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    const int64_t bytes = fastp_gpu_dup_bitmap_bytes(ctx);
    if (bytes == 0) return FASTP_GPU_OK;
    if (!src_device) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rj_ = join_aux(ctx, ctx->stream); if (rj_) return rj_; }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bitmap, src_device, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_dup_prefix_set(fastp_gpu_ctx* ctx, const void* images_device, int32_t n_images) {
    if (!ctx || n_images < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    const int64_t bytes = fastp_gpu_dup_bitmap_bytes(ctx);
    if (bytes == 0 || n_images == 0) {
        ctx->has_prefix = false;
        return FASTP_GPU_OK;
    }
    if (!images_device || ((uintptr_t)images_device & 15u)) return fail(ctx, FASTP_GPU_E_INVALID, "bitmap images must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_prefix) {
        if (hipMalloc((void**)&ctx->d_prefix, (size_t)bytes) != hipSuccess) return fail(ctx, FASTP_GPU_E_NOMEM, "hipMalloc(prefix bitmaps) failed");
    }
    OrArgs o;
    o.images = (u32x4*)images_device;
    o.dst = (u32x4*)ctx->d_prefix;
    o.chunks = (u64)bytes / 16;
    o.n_images = n_images;
    hipLaunchKernelGGL(fq_or_images_kernel, dim3(ctx->cus * 8), dim3(256), 0, ctx->stream, o);
    HIP_TRY(ctx, hipGetLastError());
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    ctx->has_prefix = true;
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_prefix_or_images(fastp_gpu_ctx* ctx, void* images_device, int32_t n_images, int64_t bytes_each) {
    if (!ctx || n_images < 0 || bytes_each < 0 || (bytes_each & 15)) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    if (n_images == 0 || bytes_each == 0) return FASTP_GPU_OK;
    if (!images_device || ((uintptr_t)images_device & 15u)) return fail(ctx, FASTP_GPU_E_INVALID, "images must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    OrArgs o;
    o.images = (u32x4*)images_device;
    o.dst = nullptr;
    o.chunks = (u64)bytes_each / 16;
    o.n_images = n_images;
    hipLaunchKernelGGL(fq_or_images_kernel, dim3(ctx->cus * 8), dim3(256), 0, ctx->stream, o);
    HIP_TRY(ctx, hipGetLastError());
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_host_writes_overlapped(fastp_gpu_ctx* ctx, int on) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    ctx->host_overlapped = on != 0;
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_phred64_to_33(fastp_gpu_ctx* ctx, int32_t n, uint8_t* text, const uint32_t* line_off, const uint32_t* line_len, uint8_t* qual_rows) {
    if (!ctx || n < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    if (n == 0) return FASTP_GPU_OK;
    if (!text || !line_off || !line_len || !qual_rows) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Phred64Args a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.qs = (int)fastp_gpu_qual_stride(ctx->dp.max_len);
    a.text = text;
    a.line_off = line_off;
    a.line_len = line_len;
    a.qual = qual_rows;
    hipLaunchKernelGGL(fq_phred64_kernel, dim3(std::min((n + 3) / 4, ctx->cus * 32)), dim3(256), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_stream_set_origin(fastp_gpu_ctx* ctx, int64_t units_before, int64_t post_reads_before) {
    if (!ctx || units_before < 0 || post_reads_before < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->units_seen = (uint64_t)units_before;
    const u64 v = (u64)post_reads_before;
    if (!ctx->d_post_seen) return FASTP_GPU_OK;  // no overrepresentation analysis: nothing else reads positions
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_post_seen, &v, sizeof(v), hipMemcpyHostToDevice, ctx->stream));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_parse_fastq(fastp_gpu_ctx* ctx, const uint8_t* text, int64_t nbytes, int is_last_chunk,
                                     int32_t max_records, uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out,
                                     uint32_t* line_off, uint32_t* line_len, fastp_gpu_parse_info* info) {
    if (!ctx || !info || nbytes < 0 || max_records < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    memset(info, 0, sizeof(*info));
    info->first_bad = -1;
    if (nbytes == 0 || max_records == 0) return FASTP_GPU_OK;
    if (!text || !seq_out || !qual_out || !len_out || !line_off || !line_len) return fail(ctx, FASTP_GPU_E_INVALID, "null buffer");
    if (((uintptr_t)text & 15u) != 0) return fail(ctx, FASTP_GPU_E_INVALID, "text must be 16-byte aligned (and padded to 16 bytes)");
    if (nbytes >= (1ll << 32)) return fail(ctx, FASTP_GPU_E_INVALID, "chunk of 4 GiB or more");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ParseArgs p;
    memset(&p, 0, sizeof(p));
    p.text = text;
    p.nbytes = (u32)nbytes;
    p.is_last = is_last_chunk ? 1 : 0;
    p.max_lines = 4u * (u32)max_records + 1u;
    p.max_len = ctx->dp.max_len;
    p.sw_g = ctx->dp.sw_g;
    p.qw_g = ctx->dp.qw_g;
    p.max_records = max_records;
    p.seq_out = (u32*)seq_out;
    p.qual_out = (u32*)qual_out;
    p.len_out = len_out;
    p.line_off = line_off;
    p.line_len = line_len;
    const int per_block = PARSE_BLOCK * PARSE_BYTES_PER_LANE * PARSE_SUB;
    const int nblocks = (int)((nbytes + per_block - 1) / per_block);
    const size_t dwords = 8 + (size_t)2 * nblocks + p.max_lines + (p.max_lines + 3) / 4 + 4 + (size_t)max_records;
    int rc = ensure(ctx, (void**)&ctx->d_parse, &ctx->parse_cap, dwords * 4);
    if (rc) return rc;
    p.totals = ctx->d_parse;
    p.blockcount = ctx->d_parse + 8;
    p.blockbase = p.blockcount + nblocks;
    p.term_pos = p.blockbase + nblocks;
    p.term_len = (u8*)(p.term_pos + p.max_lines);
    p.exotic_list = p.term_pos + p.max_lines + (p.max_lines + 3) / 4 + 1;
    p.exotic_cap = (u32)max_records;
    ctx->parse_exotic.clear();
    const u32 init[8] = {0, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(p.totals, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fq_parse_count_kernel, dim3(nblocks), dim3(PARSE_BLOCK), 16, st, p);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_parse_scan_kernel, dim3(1), dim3(1024), 1024 * 4, st, p, nblocks);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_parse_index_kernel, dim3(nblocks), dim3(PARSE_BLOCK), 64, st, p);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_parse_finish_kernel, dim3(1), dim3(64), 0, st, p);
    HIP_TRY(ctx, hipGetLastError());
    u32 totals[8];
    // the record count comes back before the packer is launched: its grid then covers the records that exist, not the
    // caller's capacity (a 16 MiB chunk of 150-base reads holds 50 K records of the 512 K it may hold)
    HIP_TRY(ctx, hipMemcpyAsync(totals, p.totals, sizeof(totals), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const u32 nrec = totals[3];
    if (nrec > 0) {
        hipLaunchKernelGGL(fq_parse_pack_kernel, dim3((nrec + 15) / 16), dim3(256), 0, st, p);  // 4 records per wavefront
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(totals, p.totals, sizeof(totals), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    info->n_records = (int32_t)totals[3];
    info->consumed = (int64_t)totals[4];
    info->n_lines = (int64_t)totals[2];
    info->first_bad = totals[1] == 0xFFFFFFFFu ? -1 : (int32_t)(totals[1] >> 2);
    info->bad_kind = totals[1] == 0xFFFFFFFFu ? 0 : (int32_t)(totals[1] & 3u);
    info->max_seq_len = (int32_t)totals[5];
    if (totals[6] > 0 && nrec > 0) {   // records with letters outside ACGTN: their indexes, ascending (fastp_gpu_parse_exotic)
        const u32 cnt = std::min(totals[6], p.exotic_cap);
        ctx->parse_exotic.resize(cnt);
        HIP_TRY(ctx, hipMemcpy(ctx->parse_exotic.data(), p.exotic_list, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        std::sort(ctx->parse_exotic.begin(), ctx->parse_exotic.end());
    }
    info->n_exotic = (int32_t)ctx->parse_exotic.size();
    if (info->first_bad >= 0)
        return fail(ctx, FASTP_GPU_E_INVALID,
                    info->bad_kind == FASTP_GPU_PARSE_BAD_TOO_LONG ? "a read of the chunk is longer than the context's max_len (see first_bad, max_seq_len)"
                    : info->bad_kind == FASTP_GPU_PARSE_BAD_ALPHABET ? "a record of the chunk has a quality character outside '!'..'~' (see first_bad)"
                                                                     : "malformed FASTQ record in the chunk (see first_bad)");
    return FASTP_GPU_OK;
}

extern "C" int32_t fastp_gpu_parse_exotic(const fastp_gpu_ctx* ctx, int32_t* units, int32_t capacity) {
    if (!ctx) return 0;
    const int32_t n = (int32_t)ctx->parse_exotic.size();
    for (int32_t k = 0; units && k < n && k < capacity; k++) units[k] = ctx->parse_exotic[k];
    return n;
}

// BgzfMtReader::readerLoop's header walk (src/bgzf.h:29-32, 150-200): gzip member header with the BC extra
// subfield, BSIZE = member size - 1; deflate payload; CRC32; ISIZE
extern "C" int fastp_gpu_bgzf_index(const uint8_t* h, int64_t nbytes, int32_t max_blocks, int64_t max_text_bytes,
                                    uint32_t* pay_off, uint32_t* pay_len, uint32_t* isize, uint32_t* crc, uint64_t* out_off,
                                    fastp_gpu_inflate_info* info) {
    if (!info) return FASTP_GPU_E_INVALID;
    info->n_blocks = 0;
    info->first_bad = -1;
    info->consumed = 0;
    info->out_bytes = 0;
    if (!h || nbytes < 0 || max_blocks < 0 || !pay_off || !pay_len || !isize || !crc || !out_off) return FASTP_GPU_E_INVALID;
    int64_t pos = 0, text = 0;
    int32_t k = 0;
    while (k < max_blocks && pos + 18 <= nbytes) {
        const uint8_t* p = h + pos;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { info->first_bad = k; break; }
        const uint32_t xlen = (uint32_t)p[10] | ((uint32_t)p[11] << 8);
        if (pos + 12 + xlen > nbytes) break;  // header not complete yet
        // find the BC subfield among the extra subfields (bgzip writes it first; others may precede it)
        int64_t bsize = -1;
        for (uint32_t o = 0; o + 4 <= xlen;) {
            const uint8_t* e = p + 12 + o;
            const uint32_t slen = (uint32_t)e[2] | ((uint32_t)e[3] << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && o + 6 <= xlen) { bsize = ((int64_t)e[4] | ((int64_t)e[5] << 8)) + 1; break; }
            o += 4 + slen;
        }
        if (bsize < 0 || (p[3] & ~4)) { info->first_bad = k; break; }  // not BGZF (or name/comment/hcrc fields: bgzip never writes them)
        const int64_t hdr = 12 + (int64_t)xlen;
        if (bsize < hdr + 8) { info->first_bad = k; break; }
        if (pos + bsize > nbytes) break;  // member not complete yet
        if (pos + bsize > 0xFFFFFFFFll) break;  // payload offsets are 32 bit: the caller indexes the rest from `consumed`
        const uint8_t* t = p + bsize - 8;
        const uint32_t c = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t n = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        if (n > 65536u) { info->first_bad = k; break; }
        if (text + n > max_text_bytes) break;
        pay_off[k] = (uint32_t)(pos + hdr);
        pay_len[k] = (uint32_t)(bsize - hdr - 8);
        isize[k] = n;
        crc[k] = c;
        out_off[k] = (uint64_t)text;
        text += n;
        pos += bsize;
        k++;
    }
    info->n_blocks = k;
    info->consumed = pos;
    info->out_bytes = text;
    return info->first_bad >= 0 ? FASTP_GPU_E_INVALID : FASTP_GPU_OK;
}

extern "C" int fastp_gpu_inflate_bgzf(fastp_gpu_ctx* ctx, const uint8_t* comp, int32_t n_blocks, const uint32_t* pay_off,
                                      const uint32_t* pay_len, const uint32_t* isize, const uint32_t* crc, const uint64_t* out_off,
                                      uint8_t* out, int64_t out_capacity, int check_crc, int32_t* first_bad) {
    if (first_bad) *first_bad = -1;
    if (!ctx || n_blocks < 0 || out_capacity < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    if (n_blocks == 0) return FASTP_GPU_OK;
    if (!comp || !pay_off || !pay_len || !isize || !crc || !out_off || !out) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t scratch = (size_t)n_blocks * INF_SCRATCH, status = (size_t)n_blocks * 4;
    int rc = ensure(ctx, (void**)&ctx->d_inf, &ctx->inf_cap, scratch + status + 16);
    if (rc) return rc;
    InflateArgs a;
    memset(&a, 0, sizeof(a));
    a.comp = comp;
    a.pay_off = pay_off; a.pay_len = pay_len; a.isize = isize; a.crc = crc; a.out_off = out_off;
    a.n = n_blocks;
    a.out = out;
    a.out_cap = (u64)out_capacity;
    a.scratch = ctx->d_inf;
    a.status = (u32*)(ctx->d_inf + scratch);
    a.first_bad = (u32*)(ctx->d_inf + scratch + status);
    a.check_crc = check_crc;
    HIP_TRY(ctx, hipMemsetAsync(a.first_bad, 0xFF, 4, st));
    // one wavefront per block (fq_inflate_wave.h: 4.6 ms per launch whatever the size, 2 blocks per CU in flight) up to
    // 6144 blocks; beyond that blocks in flight decide and one lane per block (fq_inflate.h: ~50 ms per launch, 64 blocks
    // per wavefront) overtakes it (profiles/r02l_inflate.txt).  FASTP_GPU_INFLATE=lane|wave forces one.
    const char* how = getenv("FASTP_GPU_INFLATE");
    const bool lane_kernel = how ? !strcmp(how, "lane") : n_blocks > 6144;
    if (lane_kernel) {
        const int lds_bytes = INF_ENTRIES * INF_LANES * 2 + INF_SBUF * INF_LANES * 4;
        hipLaunchKernelGGL(fq_inflate_kernel, dim3((n_blocks + INF_LANES - 1) / INF_LANES), dim3(INF_LANES), lds_bytes, st, a);
    } else {
        const int per_cu = std::max(1, (160 * 1024) / (int)sizeof(IwLds));
        hipLaunchKernelGGL(fq_inflate_wave_kernel, dim3(std::min(n_blocks, ctx->cus * per_cu)), dim3(64), sizeof(IwLds), st, a);
    }
    HIP_TRY(ctx, hipGetLastError());
    u32 bad = 0xFFFFFFFFu;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, a.first_bad, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (bad != 0xFFFFFFFFu) {
        if (first_bad) *first_bad = (int32_t)bad;
        return fail(ctx, FASTP_GPU_E_INVALID, "a BGZF block failed to inflate (see first_bad)");
    }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_format_fastq(fastp_gpu_ctx* ctx, int32_t n, const fastp_gpu_format_in* m1, const fastp_gpu_format_in* m2,
                                      const fastp_gpu_correction* corrections, const int32_t* n_corrections, uint8_t* out1,
                                      int64_t out1_capacity, uint8_t* out2, int64_t out2_capacity, int64_t out_len[2]) {
    if (!ctx || !m1 || !out_len || n < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    out_len[0] = out_len[1] = 0;
    const bool paired = ctx->dp.paired != 0;
    if (paired != (m2 != nullptr)) return fail(ctx, FASTP_GPU_E_INVALID, "mate 2 must be given exactly for a paired engine");
    if (ctx->dp.merge) return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "merge mode writes the merged stream on the host");
    if (ctx->dp.umi_len1 || ctx->dp.umi_len2) return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "UMI name edits are host logic");
    if (n == 0) return FASTP_GPU_OK;
    if (!out1 || (paired && !out2) || out1_capacity < 0 || out2_capacity < 0) return fail(ctx, FASTP_GPU_E_INVALID, "null output buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    FmtArgs f;
    memset(&f, 0, sizeof(f));
    f.n = n;
    f.paired = paired ? 1 : 0;
    f.dedup = ctx->dp.dedup;
    f.nblocks = (n + FMT_BLOCK - 1) / FMT_BLOCK;
    const size_t words = (size_t)4 * f.nblocks + 2 + (size_t)2 * n;
    int rc = ensure(ctx, (void**)&ctx->d_fmt, &ctx->fmt_cap, words * 8);
    if (rc) return rc;
    f.blocksum = ctx->d_fmt;
    f.blockbase = f.blocksum + (size_t)2 * f.nblocks;
    f.totals = f.blockbase + (size_t)2 * f.nblocks;
    const fastp_gpu_format_in* ins[2] = {m1, m2};
    uint8_t* outs[2] = {out1, out2};
    const int64_t caps[2] = {out1_capacity, out2_capacity};
    for (int m = 0; m < (paired ? 2 : 1); m++) {
        if (!ins[m]->text || !ins[m]->line_off || !ins[m]->line_len || !ins[m]->res) return fail(ctx, FASTP_GPU_E_INVALID, "null input");
        f.m[m].text = ins[m]->text;
        f.m[m].line_off = ins[m]->line_off;
        f.m[m].line_len = ins[m]->line_len;
        f.m[m].res = (const u32*)ins[m]->res;
        f.m[m].out = outs[m];
        f.m[m].out_cap = (u64)caps[m];
        f.m[m].unit_off = f.totals + 2 + (size_t)m * n;
    }
    f.corrections = (const u32*)corrections;
    f.n_corrections = n_corrections;
    HIP_TRY(ctx, hipMemsetAsync(f.totals, 0, 16, st));
    hipLaunchKernelGGL(fq_fmt_len_kernel, dim3(f.nblocks), dim3(FMT_BLOCK), 16, st, f);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_fmt_scan_kernel, dim3(paired ? 2 : 1), dim3(1024), 1024 * 8, st, f);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_fmt_write_kernel, dim3(f.nblocks), dim3(FMT_BLOCK), 128, st, f);
    HIP_TRY(ctx, hipGetLastError());
    int32_t ncorr = 0;
    if (corrections && n_corrections) HIP_TRY(ctx, hipMemcpyAsync(&ncorr, n_corrections, 4, hipMemcpyDeviceToHost, st));
    u64 totals[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(totals, f.totals, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ncorr > 0) {
        hipLaunchKernelGGL(fq_fmt_fix_kernel, dim3((ncorr + 255) / 256), dim3(256), 0, st, f);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    out_len[0] = (int64_t)totals[0];
    out_len[1] = (int64_t)totals[1];
    if (out_len[0] > out1_capacity || (paired && out_len[1] > out2_capacity))
        return fail(ctx, FASTP_GPU_E_OVERFLOW, "output buffer too small (see out_len for the needed sizes)");
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_format_streams(fastp_gpu_ctx* ctx, int32_t n, const fastp_gpu_format_io* m1, const fastp_gpu_format_io* m2,
                                        const fastp_gpu_pair_result* pair, const fastp_gpu_correction* corrections,
                                        const int32_t* n_corrections, const fastp_gpu_format_options* opts,
                                        uint8_t* const out[FASTP_GPU_N_OUTPUTS], const int64_t out_capacity[FASTP_GPU_N_OUTPUTS],
                                        int64_t out_len[FASTP_GPU_N_OUTPUTS]) {
    if (!ctx || !m1 || !out || !out_capacity || !out_len || n < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    for (int q = 0; q < FASTP_GPU_N_OUTPUTS; q++) out_len[q] = 0;
    const bool paired = ctx->dp.paired != 0;
    if (paired != (m2 != nullptr)) return fail(ctx, FASTP_GPU_E_INVALID, "mate 2 must be given exactly for a paired engine");
    if (paired && !pair) return fail(ctx, FASTP_GPU_E_INVALID, "a paired engine needs the pair records");
    if (ctx->dp.overlapped_out && !ctx->host_overlapped)
        return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "--overlapped_out's stream is written by the host (fastp_gpu_host.h, or fastp_gpu_host_writes_overlapped)");
    FmtsArgs f;
    memset(&f, 0, sizeof(f));
    f.n = n;
    f.paired = paired ? 1 : 0;
    f.dedup = ctx->dp.dedup;
    f.merge = paired ? ctx->dp.merge : 0;
    f.merge_include_unmerged = ctx->dp.merge_include_unmerged;
    f.delim[0] = ':';
    f.delim_len = 1;
    if (opts) {
        f.want_failed = opts->want_failed != 0;
        f.want_u1 = opts->want_unpaired1 != 0;
        f.want_u2 = opts->want_unpaired2 != 0;
        if (opts->umi_loc < FASTP_GPU_UMI_NONE || opts->umi_loc > FASTP_GPU_UMI_PER_READ)
            return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "UMIs taken from the index part of the name are host logic");
        f.umi_loc = opts->umi_loc;
        f.umi_len = opts->umi_len;
        if (opts->umi_delimiter) {
            const size_t dl = strlen(opts->umi_delimiter);
            if (dl > sizeof(f.delim)) return fail(ctx, FASTP_GPU_E_INVALID, "UMI delimiter longer than 8 characters");
            memcpy(f.delim, opts->umi_delimiter, dl);
            f.delim_len = (u32)dl;
        }
        if (opts->umi_prefix) {
            const size_t pl = strlen(opts->umi_prefix);
            if (pl > sizeof(f.prefix)) return fail(ctx, FASTP_GPU_E_INVALID, "UMI prefix longer than 32 characters");
            memcpy(f.prefix, opts->umi_prefix, pl);
            f.prefix_len = (u32)pl;
        }
    }
    if (n == 0) return FASTP_GPU_OK;
    // the streams this configuration can write to need a buffer
    bool need[FMTS_STREAMS] = {true, paired, f.want_failed != 0, f.merge != 0, paired && f.want_u1, paired && f.want_u2};
    for (int q = 0; q < FMTS_STREAMS; q++) {
        if (out_capacity[q] < 0 || (need[q] && !out[q])) return fail(ctx, FASTP_GPU_E_INVALID, "null output buffer for a stream the options ask for");
        f.out[q] = out[q];
        f.out_cap[q] = out[q] ? (u64)out_capacity[q] : 0;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    f.nblocks = (n + FMT_BLOCK - 1) / FMT_BLOCK;
    const size_t words = (size_t)2 * FMTS_STREAMS * f.nblocks + FMTS_STREAMS;
    int rc = ensure(ctx, (void**)&ctx->d_fmt, &ctx->fmt_cap, words * 8);
    if (rc) return rc;
    f.blocksum = ctx->d_fmt;
    f.blockbase = f.blocksum + (size_t)FMTS_STREAMS * f.nblocks;
    f.totals = f.blockbase + (size_t)FMTS_STREAMS * f.nblocks;
    const fastp_gpu_format_io* ins[2] = {m1, m2};
    for (int m = 0; m < (paired ? 2 : 1); m++) {
        if (!ins[m]->text || !ins[m]->line_off || !ins[m]->line_len || !ins[m]->res) return fail(ctx, FASTP_GPU_E_INVALID, "null input");
        f.m[m].text = ins[m]->text;
        f.m[m].line_off = ins[m]->line_off;
        f.m[m].line_len = ins[m]->line_len;
        f.m[m].res = (const u32*)ins[m]->res;
    }
    f.pair = (const u32*)pair;
    f.corrections = (const u32*)corrections;
    f.n_corrections = n_corrections;
    int32_t ncorr = 0;
    if (corrections && n_corrections) {
        HIP_TRY(ctx, hipMemcpyAsync(&ncorr, n_corrections, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        // the engine's counter keeps counting past the list's capacity: such a list is incomplete and reading
        // n_corrections entries would run behind the buffer
        int32_t cap = (opts && opts->corrections_capacity > 0) ? opts->corrections_capacity
                                                                : (corrections == ctx->last_corr ? ctx->last_corr_cap : 0);
        if (cap > 0 && ncorr > cap) return fail(ctx, FASTP_GPU_E_OVERFLOW, "correction list capacity exceeded: the list is incomplete");
        if (ncorr > 0) {
            hipLaunchKernelGGL(fq_fmts_corr_kernel, dim3((ncorr + 255) / 256), dim3(256), 0, st, f);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    hipLaunchKernelGGL(fq_fmts_len_kernel, dim3(f.nblocks), dim3(FMT_BLOCK), FMTS_STREAMS * 4, st, f);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_fmts_scan_kernel, dim3(FMTS_STREAMS), dim3(1024), 1024 * 8, st, f);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_fmts_write_kernel, dim3(f.nblocks), dim3(FMT_BLOCK), FMTS_STREAMS * 16 * 4 + FMT_BLOCK * 16, st, f);
    HIP_TRY(ctx, hipGetLastError());
    u64 totals[FMTS_STREAMS];
    HIP_TRY(ctx, hipMemcpyAsync(totals, f.totals, sizeof(totals), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    bool overflow = false;
    for (int q = 0; q < FMTS_STREAMS; q++) {
        out_len[q] = (int64_t)totals[q];
        if (totals[q] > f.out_cap[q]) overflow = true;
    }
    if (overflow) return fail(ctx, FASTP_GPU_E_OVERFLOW, "output buffer too small (see out_len for the needed sizes)");
    return FASTP_GPU_OK;
}

// ---- Evaluator pre-pass (fq_eval.h) ---------------------------------------------------------------------
// the reads a reference loading loop `while (records < read_limit && bases < base_limit)` admits
static int eval_admit(fastp_gpu_ctx* ctx, const uint16_t* len, int32_t n, int64_t read_limit, int64_t base_limit,
                      std::vector<u16>& lens) {
    const size_t take = (size_t)std::min<int64_t>(n, read_limit);
    lens.resize(take);
    if (take) {
        HIP_TRY(ctx, hipMemcpyAsync(lens.data(), len, take * 2, hipMemcpyDeviceToHost, ctx->stream));
        { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    }
    int64_t bases = 0;
    size_t used = 0;
    while (used < take && bases < base_limit) bases += lens[used++];
    lens.resize(used);
    return 0;
}

extern "C" int fastp_gpu_eval_seq_len(fastp_gpu_ctx* ctx, const uint16_t* len, int32_t n, int32_t* seq_len) {
    if (!ctx || !seq_len || n < 0 || (n > 0 && !len)) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<u16> lens;
    int rc = eval_admit(ctx, len, n, 1000, INT64_MAX, lens);   // evaluator.cpp:63-74
    if (rc) return rc;
    int best = 0;
    for (u16 v : lens) best = std::max(best, (int)v);
    *seq_len = best;
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_eval_adapter_kmers(fastp_gpu_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint16_t* len,
                                            int32_t n, int32_t trim_tail1, uint32_t* counts, int64_t* records) {
    if (!ctx || !counts || n < 0 || (n > 0 && (!seq || !qual || !len))) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t READ_LIMIT = 256 * 1024, BASE_LIMIT = 151 * READ_LIMIT;   // evaluator.cpp:313-314
    std::vector<u16> lens;
    int rc = eval_admit(ctx, len, n, READ_LIMIT, BASE_LIMIT, lens);
    if (rc) return rc;
    if (records) *records = (int64_t)lens.size();
    HIP_TRY(ctx, hipMemsetAsync(counts, 0, sizeof(u32) << 20, st));
    EvalKmerArgs a;
    memset(&a, 0, sizeof(a));
    a.r.seq = seq;
    a.r.qual = qual;
    a.r.len = len;
    a.r.n = (int)lens.size();
    a.r.seq_stride = (int)fastp_gpu_seq_stride(ctx->dp.max_len);
    a.r.qual_stride = (int)fastp_gpu_qual_stride(ctx->dp.max_len);
    a.shift_tail = std::max(1, trim_tail1);
    a.span = std::max(1, ctx->dp.max_len - 29);   // positions 20 .. len - 10 - shift_tail
    a.counts = counts;
    const long long lanes = (long long)a.r.n * a.span;
    if (lanes > 0) {
        hipLaunchKernelGGL(fq_eval_kmer_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, a);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_eval_overrep(fastp_gpu_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint16_t* len, int32_t n,
                                      int32_t seq_len, char* text, int64_t text_capacity, int64_t* off, int64_t* count,
                                      int32_t max_seqs, int32_t* n_seqs) {
    if (!ctx || !n_seqs || !off || n < 0 || max_seqs < 0 || text_capacity < 0 || (n > 0 && (!seq || !qual || !len)) ||
        (max_seqs > 0 && (!text || !count)))
        return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    *n_seqs = 0;
    off[0] = 0;
    if (seq_len < 1) return fail(ctx, FASTP_GPU_E_INVALID, "seq_len (Evaluator::computeSeqLen) must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t BASE_LIMIT = 151 * 10000;   // evaluator.cpp:83
    std::vector<u16> lens;
    int rc = eval_admit(ctx, len, n, (int64_t)(1 << 24) - 2, BASE_LIMIT, lens);
    if (rc) return rc;
    EvalCensusArgs a;
    memset(&a, 0, sizeof(a));
    a.r.seq = seq;
    a.r.qual = qual;
    a.r.len = len;
    a.r.n = (int)lens.size();
    a.r.seq_stride = (int)fastp_gpu_seq_stride(ctx->dp.max_len);
    a.r.qual_stride = (int)fastp_gpu_qual_stride(ctx->dp.max_len);
    const int steps[EVAL_STEPS] = {10, 20, 40, 100, std::min(150, seq_len - 2)};   // :97
    int min_step = 1 << 30;
    for (int i = 0; i < EVAL_STEPS; i++) {
        a.step[i] = steps[i];
        if (steps[i] > 0) min_step = std::min(min_step, steps[i]);
    }
    a.span = std::max(1, ctx->dp.max_len - min_step);
    a.seqlen = seq_len;
    uint64_t items = 0;
    for (u16 l : lens)
        for (int i = 0; i < EVAL_STEPS; i++)
            if (steps[i] > 0 && (int)l > steps[i]) items += (uint64_t)((int)l - steps[i]);
    if (items == 0) return FASTP_GPU_OK;
    uint64_t cap = 1024;
    while (cap < 2 * items) cap <<= 1;
    if (cap > (1ull << 31)) return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "substring census larger than 2^30 items");
    // a substring over its threshold occurs >= 3 times
    a.hot_cap = (u32)std::min<uint64_t>(items / 3 + 16, 1u << 22);
    const size_t b_slot = cap * 8, b_count = cap * 4, b_hot = (size_t)a.hot_cap * 8, b_hc = (size_t)a.hot_cap * 4,
                 b_text = (size_t)a.hot_cap * 152;
    rc = ensure(ctx, (void**)&ctx->d_eval, &ctx->eval_cap, b_slot + b_count + b_hot + b_hc + 16 + b_text);
    if (rc) return rc;
    u8* base = ctx->d_eval;
    a.slot = (u64*)base;
    a.count = (u32*)(base + b_slot);
    a.hot = (u64*)(base + b_slot + b_count);
    a.hot_count = (u32*)(base + b_slot + b_count + b_hot);
    a.n_hot = (u32*)(base + b_slot + b_count + b_hot + b_hc);
    a.text = base + b_slot + b_count + b_hot + b_hc + 16;
    a.mask = (u32)(cap - 1);
    HIP_TRY(ctx, hipMemsetAsync(base, 0, b_slot + b_count, st));
    HIP_TRY(ctx, hipMemsetAsync(a.n_hot, 0, 16, st));
    const long long lanes = (long long)a.r.n * a.span;
    hipLaunchKernelGGL(fq_eval_census_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(fq_eval_harvest_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, a);
    HIP_TRY(ctx, hipGetLastError());
    u32 n_hot = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_hot, a.n_hot, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (n_hot > a.hot_cap) return fail(ctx, FASTP_GPU_E_UNSUPPORTED, "more than 4 Mi overrepresented substrings");
    std::map<std::string, long> hot;   // Options::overRepSeqs (evaluator.cpp:112-137)
    if (n_hot) {
        hipLaunchKernelGGL(fq_eval_text_kernel, dim3((unsigned)(((size_t)n_hot * 16 + 255) / 256)), dim3(256), 0, st, a);
        HIP_TRY(ctx, hipGetLastError());
        std::vector<u64> hw(n_hot);
        std::vector<u32> hc(n_hot);
        std::vector<u8> ht((size_t)n_hot * 152);
        HIP_TRY(ctx, hipMemcpyAsync(hw.data(), a.hot, (size_t)n_hot * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hc.data(), a.hot_count, (size_t)n_hot * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(ht.data(), a.text, ht.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        for (u32 k = 0; k < n_hot; k++)
            hot[std::string((const char*)&ht[(size_t)k * 152], (size_t)steps[(hw[k] >> 21) & 7u])] = (long)hc[k];
    }
    // "remove substrings" (evaluator.cpp:139-160), erasing while iterating as the reference does
    for (auto it = hot.begin(); it != hot.end();) {
        bool is_sub = false;
        for (auto it2 = hot.begin(); it2 != hot.end(); ++it2)
            if (it->first != it2->first && it2->first.find(it->first) != std::string::npos && it->second / it2->second < 10) {
                is_sub = true;
                break;
            }
        if (is_sub) it = hot.erase(it);
        else ++it;
    }
    *n_seqs = (int32_t)hot.size();
    if ((int64_t)hot.size() > max_seqs) return fail(ctx, FASTP_GPU_E_OVERFLOW, "more sequences than max_seqs");
    int64_t at = 0;
    int32_t i = 0;
    for (const auto& kv : hot) {
        if (at + (int64_t)kv.first.size() > text_capacity) return fail(ctx, FASTP_GPU_E_OVERFLOW, "text buffer too small");
        memcpy(text + at, kv.first.data(), kv.first.size());
        at += (int64_t)kv.first.size();
        count[i] = kv.second;
        off[++i] = at;
    }
    return FASTP_GPU_OK;
}

// ---- output text -> BGZF-framed gzip members (fq_deflate.h) ------------------------------------------------
extern "C" int fastp_gpu_deflate_bgzf(fastp_gpu_ctx* ctx, const uint8_t* text, int64_t nbytes, int write_eof, uint8_t* out,
                                      int64_t out_capacity, int64_t* out_len) {
    if (!ctx || !out_len || nbytes < 0 || out_capacity < 0 || (nbytes > 0 && !text) || (out_capacity > 0 && !out))
        return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    *out_len = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    static const uint8_t eof_member[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t total_blocks = (nbytes + DEF_BLOCK - 1) / DEF_BLOCK;
    // a round = what is in flight at once (a wavefront per block, eight per CU): the scratch is sized by it - 4 bytes of
    // tokens per input byte + a member slot per block - so a larger round only costs HBM
    const int round = (int)std::min<int64_t>(total_blocks, (int64_t)ctx->cus * 8);
    if (round > 0) {
        const size_t b_slots = (size_t)round * DEF_SLOT, b_tok = (size_t)round * DEF_BLOCK * 4, b_sizes = (size_t)round * 4 + 8,
                     b_offs = ((size_t)round + 1) * 8;
        int rc = ensure(ctx, (void**)&ctx->d_def, &ctx->def_cap, b_slots + b_tok + b_sizes + b_offs);
        if (rc) return rc;
    }
    u64 written = 0;   // bytes needed so far
    for (int64_t b0 = 0; b0 < total_blocks; b0 += round) {
        DeflateArgs a;
        memset(&a, 0, sizeof(a));
        a.nblocks = (int)std::min<int64_t>(round, total_blocks - b0);
        a.text = text + (size_t)b0 * DEF_BLOCK;
        a.nbytes = (u64)std::min<int64_t>(nbytes - b0 * DEF_BLOCK, (int64_t)a.nblocks * DEF_BLOCK);
        a.slots = ctx->d_def;
        a.tokens = (u32*)(ctx->d_def + (size_t)round * DEF_SLOT);
        a.sizes = (u32*)((u8*)a.tokens + (size_t)round * DEF_BLOCK * 4);
        a.offs = (u64*)((u8*)a.sizes + (size_t)round * 4 + 8);
        a.out = out;
        a.out_base = written;
        a.out_cap = (u64)out_capacity;
        const int grid = std::min(a.nblocks, ctx->cus * std::max(1, (160 * 1024) / (int)sizeof(DefLds)));
        hipLaunchKernelGGL(fq_deflate_kernel, dim3(grid), dim3(64), sizeof(DefLds), st, a);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(fq_deflate_scan_kernel, dim3(1), dim3(1024), 1024 * 8, st, a);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(fq_deflate_gather_kernel, dim3(std::min(a.nblocks, ctx->cus * 16)), dim3(256), 0, st, a);
        HIP_TRY(ctx, hipGetLastError());
        u64 bytes = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&bytes, a.offs + a.nblocks, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        written += bytes;
    }
    if (write_eof) {
        if (written + sizeof(eof_member) <= (u64)out_capacity) {
            HIP_TRY(ctx, hipMemcpyAsync(out + written, eof_member, sizeof(eof_member), hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        written += sizeof(eof_member);
    }
    *out_len = (int64_t)written;
    if (written > (u64)out_capacity) return fail(ctx, FASTP_GPU_E_OVERFLOW, "output buffer too small (see out_len for the needed size)");
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_device_alloc(fastp_gpu_ctx* ctx, int64_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr || bytes < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    *dev_ptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (hipMalloc(dev_ptr, (size_t)std::max<int64_t>(bytes, 16)) != hipSuccess) {
        *dev_ptr = nullptr;
        return fail(ctx, FASTP_GPU_E_NOMEM, "hipMalloc failed");
    }
    return FASTP_GPU_OK;
}
extern "C" int fastp_gpu_device_free(fastp_gpu_ctx* ctx, void* dev_ptr) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    if (!dev_ptr) return FASTP_GPU_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    HIP_TRY(ctx, hipFree(dev_ptr));
    return FASTP_GPU_OK;
}
extern "C" int fastp_gpu_device_upload(fastp_gpu_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes > 0 && (!dst_dev || !src_host))) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    if (bytes == 0) return FASTP_GPU_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}
extern "C" int fastp_gpu_device_download(fastp_gpu_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes > 0 && (!dst_host || !src_dev))) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    if (bytes == 0) return FASTP_GPU_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rj_ = join_aux(ctx, ctx->stream); if (rj_) return rj_; }
    HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_synchronize(fastp_gpu_ctx* ctx) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

// host submits: the raw text and offsets of the batch's exotic units (fastp_gpu_batch::exotic_*) go to HBM with it
static int stage_exotic(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, fastp_gpu_batch* db, hipStream_t st) {
    if (b->n_exotic <= 0) return FASTP_GPU_OK;
    if (b->exotic_dense) return fail(ctx, FASTP_GPU_E_INVALID, "exotic_dense describes text that is already in HBM (fastp_gpu_submit_device)");
    const int mates = ctx->dp.paired ? 2 : 1;
    for (int m = 0; m < mates; m++) {
        if (!b->exotic_text[m] || !b->exotic_off[m] || b->exotic_text_bytes[m] <= 0)
            return fail(ctx, FASTP_GPU_E_INVALID, "n_exotic > 0 needs exotic_text, exotic_off and exotic_text_bytes");
        int rc = ensure(ctx, &ctx->d_x_text[m], &ctx->x_text_cap[m], (size_t)b->exotic_text_bytes[m] + 16);
        if (rc) return rc;
        rc = ensure(ctx, &ctx->d_x_off[m], &ctx->x_off_cap[m], (size_t)b->n_exotic * 4);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_x_text[m], b->exotic_text[m], (size_t)b->exotic_text_bytes[m], hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_x_off[m], b->exotic_off[m], (size_t)b->n_exotic * 4, hipMemcpyHostToDevice, st));
        db->exotic_text[m] = (const uint8_t*)ctx->d_x_text[m];
        db->exotic_off[m] = (const uint32_t*)ctx->d_x_off[m];
    }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_submit_host(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, fastp_gpu_results* res) {
    if (!ctx || !b || !res) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    if (b->n < 0) return fail(ctx, FASTP_GPU_E_INVALID, "negative batch size");
    if (b->n == 0) { if (res->n_corrections) *res->n_corrections = 0; return FASTP_GPU_OK; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)b->n;
    const size_t ss = fastp_gpu_seq_stride(ctx->dp.max_len), qs = fastp_gpu_qual_stride(ctx->dp.max_len);
    const int mates = ctx->dp.paired ? 2 : 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t per_mate_in = al(n * ss) + al(n * qs) + al(n * 2);
    const size_t per_mate_out = al(n * sizeof(fastp_gpu_read_result));
    const size_t pair_out = ctx->dp.paired ? al(n * sizeof(fastp_gpu_pair_result)) : 0;
    const size_t corr_out = (res->corrections && res->corrections_capacity > 0)
                                ? al((size_t)res->corrections_capacity * sizeof(fastp_gpu_correction)) : 0;
    const size_t ev_out = (res->adapter_events && res->adapter_events_capacity > 0)
                              ? al((size_t)res->adapter_events_capacity * sizeof(fastp_gpu_adapter_event)) : 0;
    const size_t total = mates * (per_mate_in + per_mate_out) + pair_out + corr_out + ev_out + 512;
    int rc = ensure(ctx, &ctx->d_stage, &ctx->stage_cap, total);
    if (rc) return rc;
    char* base = (char*)ctx->d_stage;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += al(bytes); return p; };
    fastp_gpu_batch db = *b;
    fastp_gpu_results dr = *res;
    hipStream_t st = ctx->stream;
    const void* hs[2][3] = {{b->seq1, b->qual1, b->len1}, {b->seq2, b->qual2, b->len2}};
    void* dsq[2][3];
    for (int m = 0; m < mates; m++) {
        if (!hs[m][0] || !hs[m][1] || !hs[m][2]) return fail(ctx, FASTP_GPU_E_INVALID, "missing input buffer");
        dsq[m][0] = take(n * ss);
        dsq[m][1] = take(n * qs);
        dsq[m][2] = take(n * 2);
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][0], hs[m][0], n * ss, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][1], hs[m][1], n * qs, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][2], hs[m][2], n * 2, hipMemcpyHostToDevice, st));
    }
    db.seq1 = (const uint8_t*)dsq[0][0]; db.qual1 = (const uint8_t*)dsq[0][1]; db.len1 = (const uint16_t*)dsq[0][2];
    if (mates == 2) {
        db.seq2 = (const uint8_t*)dsq[1][0]; db.qual2 = (const uint8_t*)dsq[1][1]; db.len2 = (const uint16_t*)dsq[1][2];
    }
    dr.r1 = (fastp_gpu_read_result*)take(n * sizeof(fastp_gpu_read_result));
    if (mates == 2) {
        dr.r2 = (fastp_gpu_read_result*)take(n * sizeof(fastp_gpu_read_result));
        dr.pair = (fastp_gpu_pair_result*)take(n * sizeof(fastp_gpu_pair_result));
    }
    if (corr_out) dr.corrections = (fastp_gpu_correction*)take((size_t)res->corrections_capacity * sizeof(fastp_gpu_correction));
    else { dr.corrections = nullptr; dr.corrections_capacity = 0; }
    dr.n_corrections = (int32_t*)take(sizeof(int32_t));
    if (ev_out) dr.adapter_events = (fastp_gpu_adapter_event*)take((size_t)res->adapter_events_capacity * sizeof(fastp_gpu_adapter_event));
    else { dr.adapter_events = nullptr; dr.adapter_events_capacity = 0; }
    dr.n_adapter_events = (int32_t*)take(sizeof(int32_t));
    rc = stage_exotic(ctx, b, &db, st);
    if (!rc) rc = fastp_gpu_submit_device(ctx, &db, &dr, st);
    if (!rc) rc = join_aux(ctx, st);   // the duplicate flags of the last launch
    if (rc) {   // copies from the caller's pinned buffers are queued: they must have drained before the caller may reuse them
        (void)hipStreamSynchronize(st);
        return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(res->r1, dr.r1, n * sizeof(fastp_gpu_read_result), hipMemcpyDeviceToHost, st));
    if (mates == 2) {
        HIP_TRY(ctx, hipMemcpyAsync(res->r2, dr.r2, n * sizeof(fastp_gpu_read_result), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(res->pair, dr.pair, n * sizeof(fastp_gpu_pair_result), hipMemcpyDeviceToHost, st));
    }
    int32_t ncorr = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ncorr, dr.n_corrections, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    int32_t nev = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nev, dr.n_adapter_events, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (res->n_adapter_events) *res->n_adapter_events = nev < res->adapter_events_capacity ? nev : res->adapter_events_capacity;
    if (ev_out && nev > 0) {
        const int32_t take_n = nev < res->adapter_events_capacity ? nev : res->adapter_events_capacity;
        HIP_TRY(ctx, hipMemcpy(res->adapter_events, dr.adapter_events, (size_t)take_n * sizeof(fastp_gpu_adapter_event),
                               hipMemcpyDeviceToHost));
        if (nev > res->adapter_events_capacity) return fail(ctx, FASTP_GPU_E_OVERFLOW, "adapter event list capacity exceeded");
    }
    if (res->n_corrections) *res->n_corrections = ncorr;
    if (corr_out && ncorr > 0) {
        if (ncorr > res->corrections_capacity) {
            if (res->n_corrections) *res->n_corrections = res->corrections_capacity;
            HIP_TRY(ctx, hipMemcpy(res->corrections, dr.corrections,
                                   (size_t)res->corrections_capacity * sizeof(fastp_gpu_correction), hipMemcpyDeviceToHost));
            return fail(ctx, FASTP_GPU_E_OVERFLOW, "correction list capacity exceeded");
        }
        HIP_TRY(ctx, hipMemcpy(res->corrections, dr.corrections, (size_t)ncorr * sizeof(fastp_gpu_correction),
                               hipMemcpyDeviceToHost));
    }
    return FASTP_GPU_OK;
}


// ---- asynchronous host submits (include/fastp_gpu.h "Pipelined host submits") ----
extern "C" int fastp_gpu_host_alloc(fastp_gpu_ctx* ctx, int64_t bytes, void** ptr) {
    if (!ctx || !ptr || bytes < 0) return fail(ctx, FASTP_GPU_E_INVALID, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(ptr, (size_t)std::max<int64_t>(bytes, 16)));
    return FASTP_GPU_OK;
}
extern "C" int fastp_gpu_host_free(fastp_gpu_ctx* ctx, void* ptr) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    if (ptr) HIP_TRY(ctx, hipHostFree(ptr));
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_submit_host_async(fastp_gpu_ctx* ctx, const fastp_gpu_batch* b, fastp_gpu_results* res, int slot) {
    if (!ctx || !b || !res) return fail(ctx, FASTP_GPU_E_INVALID, "null argument");
    if (slot < 0 || slot >= FASTP_GPU_ASYNC_SLOTS) return fail(ctx, FASTP_GPU_E_INVALID, "slot out of range");
    if (b->n <= 0) return fail(ctx, FASTP_GPU_E_INVALID, "empty batch");
    fastp_gpu_ctx::AsyncSlot& sl = ctx->aslot[slot];
    if (sl.busy) return fail(ctx, FASTP_GPU_E_INVALID, "slot still in flight: fastp_gpu_wait it first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!sl.done) HIP_TRY(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.counts) HIP_TRY(ctx, hipHostMalloc((void**)&sl.counts, 64));
    const size_t n = (size_t)b->n;
    const size_t ss = fastp_gpu_seq_stride(ctx->dp.max_len), qs = fastp_gpu_qual_stride(ctx->dp.max_len);
    const int mates = ctx->dp.paired ? 2 : 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t corr_cap = (res->corrections && res->corrections_capacity > 0) ? (size_t)res->corrections_capacity : 0;
    const size_t ev_cap = (res->adapter_events && res->adapter_events_capacity > 0) ? (size_t)res->adapter_events_capacity : 0;
    const size_t total = mates * (al(n * ss) + al(n * qs) + al(n * 2) + al(n * sizeof(fastp_gpu_read_result))) +
                         al(n * sizeof(fastp_gpu_pair_result)) + al(corr_cap * sizeof(fastp_gpu_correction)) +
                         al(ev_cap * sizeof(fastp_gpu_adapter_event)) + 1024;
    int rc = ensure(ctx, &sl.d_stage, &sl.cap, total);
    if (rc) return rc;
    char* base = (char*)sl.d_stage;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += al(bytes); return p; };
    hipStream_t st = ctx->stream;
    fastp_gpu_batch db = *b;
    fastp_gpu_results dr;
    memset(&dr, 0, sizeof(dr));
    const void* hs[2][3] = {{b->seq1, b->qual1, b->len1}, {b->seq2, b->qual2, b->len2}};
    void* dsq[2][3];
    for (int m = 0; m < mates; m++) {
        if (!hs[m][0] || !hs[m][1] || !hs[m][2]) return fail(ctx, FASTP_GPU_E_INVALID, "missing input buffer");
        dsq[m][0] = take(n * ss);
        dsq[m][1] = take(n * qs);
        dsq[m][2] = take(n * 2);
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][0], hs[m][0], n * ss, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][1], hs[m][1], n * qs, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(dsq[m][2], hs[m][2], n * 2, hipMemcpyHostToDevice, st));
    }
    db.seq1 = (const uint8_t*)dsq[0][0]; db.qual1 = (const uint8_t*)dsq[0][1]; db.len1 = (const uint16_t*)dsq[0][2];
    if (mates == 2) {
        db.seq2 = (const uint8_t*)dsq[1][0]; db.qual2 = (const uint8_t*)dsq[1][1]; db.len2 = (const uint16_t*)dsq[1][2];
    }
    dr.r1 = (fastp_gpu_read_result*)take(n * sizeof(fastp_gpu_read_result));
    if (mates == 2) {
        dr.r2 = (fastp_gpu_read_result*)take(n * sizeof(fastp_gpu_read_result));
        dr.pair = (fastp_gpu_pair_result*)take(n * sizeof(fastp_gpu_pair_result));
    }
    if (corr_cap) { dr.corrections = (fastp_gpu_correction*)take(corr_cap * sizeof(fastp_gpu_correction)); dr.corrections_capacity = (int32_t)corr_cap; }
    dr.n_corrections = (int32_t*)take(sizeof(int32_t));
    if (ev_cap) { dr.adapter_events = (fastp_gpu_adapter_event*)take(ev_cap * sizeof(fastp_gpu_adapter_event)); dr.adapter_events_capacity = (int32_t)ev_cap; }
    dr.n_adapter_events = (int32_t*)take(sizeof(int32_t));
    rc = stage_exotic(ctx, b, &db, st);
    if (!rc) rc = fastp_gpu_submit_device(ctx, &db, &dr, st);
    if (!rc) rc = join_aux(ctx, st);   // the duplicate flags of the last launch
    if (rc) {   // copies from the caller's pinned buffers are queued: they must have drained before the caller may reuse them
        (void)hipStreamSynchronize(st);
        return rc;
    }
    HIP_TRY(ctx, hipMemcpyAsync(res->r1, dr.r1, n * sizeof(fastp_gpu_read_result), hipMemcpyDeviceToHost, st));
    if (mates == 2) {
        HIP_TRY(ctx, hipMemcpyAsync(res->r2, dr.r2, n * sizeof(fastp_gpu_read_result), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(res->pair, dr.pair, n * sizeof(fastp_gpu_pair_result), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipMemcpyAsync(&sl.counts[0], dr.n_corrections, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(&sl.counts[1], dr.n_adapter_events, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // the sparse lists (--correction / --adapter_fasta) come out in finish_slot: their fill is only known once the batch has run,
    // and copying their whole capacity with every batch multiplied the D2H traffic of a window by ten
    sl.d_corr = corr_cap ? dr.corrections : nullptr;
    sl.d_ev = ev_cap ? dr.adapter_events : nullptr;
    HIP_TRY(ctx, hipEventRecord(sl.done, st));
    sl.res = *res;
    sl.busy = true;
    return FASTP_GPU_OK;
}

static int finish_slot(fastp_gpu_ctx* ctx, fastp_gpu_ctx::AsyncSlot& sl) {
    sl.busy = false;
    const int32_t ncorr = sl.counts[0], nev = sl.counts[1];
    // the entries that were written, on the slot's own stream (the launch stream may already hold the next batches)
    const int32_t ccopy = sl.d_corr ? std::min(ncorr, sl.res.corrections_capacity) : 0, ecopy = sl.d_ev ? std::min(nev, sl.res.adapter_events_capacity) : 0;
    if (ccopy > 0 || ecopy > 0) {
        if (!sl.copy) HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking));
        if (ccopy > 0) HIP_TRY(ctx, hipMemcpyAsync(sl.res.corrections, sl.d_corr, (size_t)ccopy * sizeof(fastp_gpu_correction), hipMemcpyDeviceToHost, sl.copy));
        if (ecopy > 0) HIP_TRY(ctx, hipMemcpyAsync(sl.res.adapter_events, sl.d_ev, (size_t)ecopy * sizeof(fastp_gpu_adapter_event), hipMemcpyDeviceToHost, sl.copy));
        HIP_TRY(ctx, hipStreamSynchronize(sl.copy));
    }
    if (sl.res.n_corrections) *sl.res.n_corrections = std::min(ncorr, sl.res.corrections ? sl.res.corrections_capacity : 0);
    if (sl.res.n_adapter_events) *sl.res.n_adapter_events = std::min(nev, sl.res.adapter_events ? sl.res.adapter_events_capacity : 0);
    if (sl.res.corrections && ncorr > sl.res.corrections_capacity) return fail(ctx, FASTP_GPU_E_OVERFLOW, "correction list capacity exceeded");
    if (sl.res.adapter_events && nev > sl.res.adapter_events_capacity) return fail(ctx, FASTP_GPU_E_OVERFLOW, "adapter event list capacity exceeded");
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_wait(fastp_gpu_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= FASTP_GPU_ASYNC_SLOTS) return fail(ctx, FASTP_GPU_E_INVALID, "slot out of range");
    fastp_gpu_ctx::AsyncSlot& sl = ctx->aslot[slot];
    if (!sl.busy) return FASTP_GPU_OK;
    HIP_TRY(ctx, hipEventSynchronize(sl.done));
    return finish_slot(ctx, sl);
}

extern "C" int fastp_gpu_poll(fastp_gpu_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot >= FASTP_GPU_ASYNC_SLOTS) return fail(ctx, FASTP_GPU_E_INVALID, "slot out of range");
    fastp_gpu_ctx::AsyncSlot& sl = ctx->aslot[slot];
    if (!sl.busy) return 1;
    const hipError_t e = hipEventQuery(sl.done);
    if (e == hipErrorNotReady) return 0;
    if (e != hipSuccess) { HIP_TRY(ctx, e); }
    const int rc = finish_slot(ctx, sl);
    return rc ? rc : 1;
}

extern "C" int fastp_gpu_counters_device(fastp_gpu_ctx* ctx, int64_t** dev_ptr, int64_t* n, void* hip_stream) {
    if (!ctx || !dev_ptr || !n) return FASTP_GPU_E_INVALID;
    (void)hip_stream;  // slabs are folded right after every launch, on the launch stream
    *dev_ptr = ctx->d_ctr;
    *n = ctx->cl.total;
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_counters(fastp_gpu_ctx* ctx, int64_t* out, int64_t n) {
    if (!ctx || !out || n != ctx->cl.total) return fail(ctx, FASTP_GPU_E_INVALID, "counter block size mismatch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());  // submits may have used a caller-provided stream
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_ctr, (size_t)n * 8, hipMemcpyDeviceToHost));
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_counters_export(fastp_gpu_ctx* ctx, int64_t* dst_device, int64_t n) {
    if (!ctx || !dst_device || n != ctx->cl.total) return fail(ctx, FASTP_GPU_E_INVALID, "counter block size mismatch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(dst_device, ctx->d_ctr, (size_t)n * 8, hipMemcpyDeviceToDevice));
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_counters_import(fastp_gpu_ctx* ctx, const int64_t* src_device, int64_t n) {
    if (!ctx || !src_device || n != ctx->cl.total) return fail(ctx, FASTP_GPU_E_INVALID, "counter block size mismatch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipMemcpy(ctx->d_ctr, src_device, (size_t)n * 8, hipMemcpyDeviceToDevice));
    const int64_t hdr[4] = {FASTP_GPU_ABI_VERSION, ctx->cl.cycles, ctx->dp.isize_max, 0};
    HIP_TRY(ctx, hipMemcpy(ctx->d_ctr, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    return FASTP_GPU_OK;
}

// Bring the HIP runtime up on `device` (context creation, the library's code object) without creating an engine: a
// host program calls it from a helper thread as early as it knows a GPU run is coming, so that the ~0.2 s this takes
// overlap its own start-up (option parsing, the Evaluator pre-pass) instead of preceding the first batch.
extern "C" int fastp_gpu_warmup(int device) {
    int n = 0;
    fq::timeline("warmup: begin");
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(nullptr, FASTP_GPU_E_NO_DEVICE, "no such HIP device");
    fq::timeline("warmup: hipGetDeviceCount (the runtime is up)");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, FASTP_GPU_E_HIP, "hipSetDevice failed");
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return fail(nullptr, FASTP_GPU_E_HIP, "hipMalloc failed");
    fq::timeline("warmup: first hipMalloc (the device's context)");
    (void)hipFuncSetAttribute((const void*)fq_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 0);   // looks the kernel up: loads the code object
    fq::timeline("warmup: code object loaded");
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    (void)hipFree(p);
    fq::timeline("warmup: end");
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_device(const fastp_gpu_ctx* ctx) { return ctx ? ctx->device : -1; }
extern "C" int fastp_gpu_plan(const fastp_gpu_ctx* ctx) {
    if (!ctx) return -1;
    return ctx->lane ? FASTP_GPU_PLAN_LANE : (ctx->split ? FASTP_GPU_PLAN_SPLIT : FASTP_GPU_PLAN_FUSED);
}

extern "C" int fastp_gpu_reset(fastp_gpu_ctx* ctx) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());  // submits may have used a caller-provided stream
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctr, 0, (size_t)ctx->cl.total * 8, ctx->stream));
    const int64_t hdr[4] = {FASTP_GPU_ABI_VERSION, ctx->cl.cycles, ctx->dp.isize_max, 0};
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctr, hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->d_bitmap)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_bitmap, 0, (size_t)(ctx->dp.dup_bits >> 3) * ctx->dp.dup_bufnum, ctx->stream));
    if (ctx->d_post_seen) HIP_TRY(ctx, hipMemsetAsync(ctx->d_post_seen, 0, sizeof(u64), ctx->stream));
    ctx->units_seen = 0;
    ctx->has_prefix = false;
    { int rs_ = sync_main(ctx); if (rs_) return rs_; }
    return FASTP_GPU_OK;
}

extern "C" int fastp_gpu_kernel_time(fastp_gpu_ctx* ctx, double* total_ms, int64_t* launches) {
    if (!ctx) return FASTP_GPU_E_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    drain_events(ctx);
    if (total_ms) *total_ms = ctx->kernel_ms;
    if (launches) *launches = ctx->kernel_launches;
    ctx->kernel_ms = 0.0;
    ctx->kernel_launches = 0;
    return FASTP_GPU_OK;
}
