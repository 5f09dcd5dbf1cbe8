// fq_lane.h - the per-read path with one LANE per read pair (or single read), everything in registers (gfx950).
//
// The tile kernels (fq_device.h) stage a tile of pairs in LDS and walk it phase by phase behind workgroup barriers;
// measured (profiles/r03b_*): the wavefronts are parked 70 % of the time, the three lane = read phases run on one
// wavefront of the workgroup, and a pair costs ~230 VALU + ~145 SALU instructions of which most are index
// arithmetic, LDS addressing and loop control.  Here a lane owns a pair from the first load to the result record:
//   * its reads live in VGPRs (2-bit bases: SWM dwords per read; N mask; the sliding-window predicate of
//     Filter::trimAndCut as a bit mask); the wavefront copies its 64 rows coalesced into a private LDS stage (all the
//     vectors of a stage in flight at once) and every lane reads its own row back from there,
//   * every scan is a fully unrolled, branch-free instruction stream over STATIC register indices - the overlap
//     prefilter of OverlapAnalysis::analyze is 6 instructions per offset with no address arithmetic at all,
//   * a dynamic offset (the surviving overlap candidates, the shift of rc(read 2) by its trimmed tail) is applied
//     with a log-step word shifter over the register array instead of an indexed memory access,
//   * no LDS tile, no workgroup barrier in the loop, no phase with idle wavefronts; LDS holds the row stages, the
//     MISC_* counters, two threshold tables and the duplicate hash's tables,
//   * chunks of 64 pairs are handed to the persistent wavefronts by an atomic counter.
// Stats::statRead runs afterwards in fq_stats.h (split plan).  The kernel covers the option family in which nothing
// needs an indexed walk along a read that cannot be bounded (see lane_plan_supported in fastp_gpu.hip); everything
// else stays on the tile kernels.  Reference: src/peprocessor.cpp:383-643, src/seprocessor.cpp:204-296.
#pragma once
#include "fq_device.h"

#ifndef FQ_LANE_STAGE_BATCH
#define FQ_LANE_STAGE_BATCH 10
#endif
#ifndef FQ_LANE_METRICS      // A/B switch: 2 = read 1 from the load sweep's partial sums + the one cut word, the last-staged read straight from the
#define FQ_LANE_METRICS 2    // stage (default); 0 = round 3's second staging of the quality rows with every dword masked by the window
#endif
#ifdef FQ_LANE_NO_FENCE      // A/B switch (tools/gpu_lane_ab3.sh)
#define FQ_LANE_FENCE() ((void)0)
#else
#define FQ_LANE_FENCE() sched_fence()
#endif

namespace fq {

// LDS of the lane kernel (dwords from the start of dynamic LDS)
struct LaneLds {
    int misc;       // MISC_* counters (u32), flushed to the workgroup's slab
    int n_misc;
    int lut_ov;     // u16 [cycles + 1] min(diffLimit, ol * pct)          overlapanalysis.cpp:51
    int lut_lowq;   // u16 [cycles + 1] floor(unqualPct * rlen / 100.0)   filter.cpp:36
    int lut_cplx;   // u16 [cycles + 1] least adjacent-difference count that passes filter.cpp:65
    int val4;       // u32 [256] Duplicate's base values of the four bases of a packed byte
    int planes;     // u32 [4][hp_nq][B][NPL] byte planes of the primes (DevLuts::dup_planes), 8-byte aligned
    int n_planes;
    int stage;      // per wavefront: 64 rows of one mate's quality (then base) rows, copied from HBM with coalesced 16-byte
    int stage_dwords;   // loads and read back one row per lane (16-byte aligned, 64 * qw_g dwords each)
    int part;       // per wavefront: countQualityMetrics' per-32-base partial sums of READ 1 of a pair [word][lane] (lane-contiguous:
    int part_dwords;    // conflict-free; registers do not hold them - ten more live VGPRs spilled 270 dwords at the cap of 168)
    int clist;      // -c on the lane plan, per wavefront: the positions of read 1 that BaseCorrector edited, one bit per base,
    int clist_dwords;   // [SWM / 2 words][lane]: read 1's rows are gone when countQualityMetrics runs (lane_apply_corrected)
    int ctr;        // the workgroup's own chunk counter (LaneArgs::local_ctr)
    int sink;       // 64 dwords every wavefront's row prefetches are written to (never read): global_load_lds needs a destination
    int jkmer;      // --merge on the lane plan (DevParams::merge_lane): [KMER_BINS] behind the MISC_* counters (inside n_misc, so in the
                    // slab) - the merged reads' 5-mers that straddle the junction of the two parts (fastp's index: earliest base high)
    int total;
};

struct LaneArgs {
    KernelArgs k;   // parameters, batch, result arrays (the LDS layout inside is not used)
    LaneLds l;
    int* chunk_ctr; // zero at launch: chunks beyond every wave's first are handed out by this counter (a static stride
                    // leaves a third of the waves one chunk short at 21.3 chunks per wave); nullptr = static stride
    int glds;       // read 2's quality rows come into the stage by global_load_lds while read 1 is hashed (FASTP_GPU_LANE_GLDS, A/B)
    int local_ctr;  // round 6, the default (FASTP_GPU_LANE_DYNAMIC=2): a workgroup owns a contiguous share of the launch's chunks and
                    // hands them to its wavefronts from a counter in ITS LDS - the balance of a shared counter inside a CU, none of
                    // its traffic: the kernel's loads-only skeleton takes 0.58 ms with a static stride, 0.92 with one returning
                    // global atomic per chunk on one word, 0.78 with one per four chunks (profiles/r06_e_*)
    int grab;       // FASTP_GPU_LANE_GRAB (round 6): chunks a wavefront takes from the counter at a time
    int pool;       // with local_ctr: the launch's LAST `pool` chunks belong to no workgroup - a wavefront whose workgroup's share is
                    // used up takes them one at a time from chunk_ctr (a word that only ever counts up: a chunk's number in the pool
                    // is what the atomic returns minus pool_base, the host adds `pool` to pool_base per launch - no reset between
                    // launches).  The shares are equal, the work per chunk is not (overlap verifications, trims): the kernel ended
                    // when its slowest CU did.  FASTP_GPU_LANE_POOL_LOG2 (0: no pool, the default - measured: no gain; 5: a 32nd of the chunks)
    int pool_base;
    int pool_grab;  // chunks per ask of the pool (>= 1): a launch of many small chunks (single-end runs: 156 k of them) would otherwise
                    // put ten thousand returning atomics on the one word while its last part runs
    int prefetch;   // FASTP_GPU_LANE_PREFETCH (round 6), bit mask: 1 = read 2's rows of the chunk are pulled into L2 while read 1 is
                    // staged and swept, 2 = read 1's rows of the wavefront's NEXT chunk while read 2 is - one dword per 128-byte line
                    // by global_load_lds into a sink (no register, no wait): the four row stagings of a chunk then find their lines
                    // in L2 / the Infinity Cache instead of paying a trip to HBM each
    // --merge with -c: the POST Stats object of read 1 in the counter block, for the (rare) merged read whose tail holds an edited
    // base - the lane that has the corrected tail in registers counts it itself (lane_merge_tail_slow)
    int64_t* post1;
    int64_t st_qual_hist, st_kmer, st_cycle, cycles;
};

// ---------------------------------------------------------------------------
// register arrays with static indices
// ---------------------------------------------------------------------------
// words [0, N) shifted towards index 0 by a per-lane number of words (0 .. 31): out[w] = in[w + dw], zero behind the end
template <int N>
FQ_DEV void word_shift_down(u32 (&x)[N], u32 dw) {
#pragma unroll
    for (int b = 1; b <= 16; b <<= 1) {
        const bool on = (dw & (u32)b) != 0;
#pragma unroll
        for (int w = 0; w < N; w++) {
            const u32 from = w + b < N ? x[w + b] : 0u;
            x[w] = on ? from : x[w];
        }
    }
    if (dw >= 32u) {
#pragma unroll
        for (int w = 0; w < N; w++) x[w] = 0;
    }
}
// the 2-bit row x (N words, word N - 1 followed by zeros) shifted down by `bases` positions
template <int N>
FQ_DEV void base_shift_down(u32 (&x)[N], u32 bases) {
    word_shift_down<N>(x, bases >> 4);
    const u32 sh = (bases & 15u) * 2u;
#pragma unroll
    for (int w = 0; w < N; w++) x[w] = alignbit(w + 1 < N ? x[w + 1] : 0u, x[w], sh);
}

// a 1-bit-per-base row (NW words of 32 bases, zeros behind) shifted down by `bits` positions
template <int NW>
FQ_DEV void bit_shift_down(u32 (&x)[NW], u32 bits) {
    word_shift_down<NW>(x, bits >> 5);
    const u32 sh = bits & 31u;
#pragma unroll
    for (int w = 0; w < NW; w++) x[w] = alignbit(w + 1 < NW ? x[w + 1] : 0u, x[w], sh);
}
// bit k of the low 16 bits -> bit 2k
FQ_DEV u32 spread16(u32 x) {
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    return (x | (x << 1)) & 0x55555555u;
}
// the 16 bases of 2-bit word w of a 1-bit-per-base row, spread to bit 2k
template <int NW>
FQ_DEV u32 nmask_word(const u32 (&n)[NW], int w) { return spread16(n[w >> 1] >> (16 * (w & 1))); }

// first set bit at or above `lo` and below `hi` of a mask held in NW words; hi if there is none
template <int NW>
FQ_DEV int mask_first(const u32 (&m)[NW], int lo, int hi, bool want) {
    int r = hi;
    if (lo >= hi) return hi;
#pragma unroll
    for (int w = NW - 1; w >= 0; w--) {   // descending: the smallest position is what remains
        u32 x = want ? m[w] : ~m[w];
        const int base = 32 * w;
        if (lo > base) x &= lo - base >= 32 ? 0u : ~lowmask32(lo - base);
        const int j = base + ffs32(x) - 1;
        if (x && j < hi) r = j;
    }
    return r;
}
// last position in [lo, hi) whose bit == want; lo - 1 if there is none
template <int NW>
FQ_DEV int mask_last(const u32 (&m)[NW], int lo, int hi, bool want) {
    int r = lo - 1;
    if (lo >= hi) return r;
#pragma unroll
    for (int w = 0; w < NW; w++) {        // ascending: the largest position is what remains
        u32 x = want ? m[w] : ~m[w];
        const int base = 32 * w;
        if (hi - base < 32) x &= hi - base <= 0 ? 0u : lowmask32(hi - base);
        const int j = base + 31 - clz32(x);
        if (x && j >= lo) r = j;
    }
    return r;
}

// ---------------------------------------------------------------------------
// one read in registers
// ---------------------------------------------------------------------------
template <int SWM>
struct LaneRead {
    u32 s[SWM];         // packed bases (N = code 0); bits past the read's end are whatever the row held
    u32 n[SWM / 2];     // N mask, bit j = base j is N (zero unless hasN)
    u32 bad[SWM / 2];   // Filter::trimAndCut: bit j = the window [j, j + w) has total quality < threshold
    int rl0, len;       // original length, length after the steps so far
    u32 flags;          // RS_*
};

// Load read `g` of one mate: bases, N mask, the window predicate of cut_right / cut_tail (the one that is enabled),
// and - in the same sweep over the quality row - the read's part of Duplicate::seq2intvector as byte-plane dot
// products (phase_hash_dot of the tile kernel: duplicate.cpp:111-120, N counts as 13).
// `off` = stream position of the read's first base (0 for read 1, read 1's length for read 2, duplicate.cpp:139).
// Duplicate::seq2intvector (duplicate.cpp:111-120) of the original read as byte-plane dot products (phase_hash_dot of
// the tile kernel; N counts as 13, duplicate.cpp:92-109).  `off` = stream position of the read's first base (0 for
// read 1, read 1's length for read 2, duplicate.cpp:139); the planes of the primes of the read's positions come from
// the LDS copy of the table (the lanes of a wavefront mostly share `off`: broadcast reads).
template <int SWM, int B, int NPL>
FQ_DEV void lane_hash(const KernelArgs& a, const u32* lds, const LaneLds& ll, const LaneRead<SWM>& r, int off, u64 (&h)[B]) {
    const u32* val4 = lds + ll.val4;
    u32 acc[B * NPL];
#pragma unroll
    for (int k = 0; k < B * NPL; k++) acc[k] = 0;
    const u32* tb = lds + ll.planes + ((off & 3) * a.L.hp_nq + (off >> 2)) * (B * NPL);
    // A ROLLED loop over the read's 32-base groups (8 packed bytes, 48 table words each): fully unrolled, the
    // compiler issued all 240 table reads of a read before the first dot product and the kernel needed 256 VGPRs.
    // The base words rotate through sw[0..1] and the N words through nw[0] so that every register index stays static.
    u32 sw[SWM], nw[SWM / 2];
#pragma unroll
    for (int w = 0; w < SWM; w++) sw[w] = r.s[w];
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) nw[w] = r.n[w];
    int rem = r.rl0;
#pragma unroll 1
    for (int grp = 0; grp < SWM / 2; grp++) {
        // a group every lane's read covers whole (four of five for 150-base reads) needs no mask by the read's length
        const bool cut = ballot(rem < 32) != 0ull;   // wave-uniform
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const u32 byte = (sw[d >> 2] >> (8 * (d & 3))) & 0xFFu;
            u32 vals = val4[byte];
            const u32 nb = (nw[0] >> (4 * d)) & 0xFu;
            if (nb) {   // N -> 13
                const u32 mN = ((nb & 1u) | ((nb & 2u) << 7) | ((nb & 4u) << 14) | ((nb & 8u) << 21)) * 0xFFu;
                vals = (vals & ~mN) | (0x0D0D0D0Du & mN);
            }
            if (cut) {
                const int left = rem - 4 * d;
                vals = left >= 4 ? vals : (left <= 0 ? 0u : (vals & lowmask32(8 * left)));   // bases past the read's end add nothing
            }
#pragma unroll
            for (int k = 0; k < B * NPL; k++) acc[k] = dot4_u8(vals, tb[d * (B * NPL) + k], acc[k]);
        }
        tb += 8 * (B * NPL);
        rem -= 32;
#pragma unroll
        for (int w = 0; w + 2 < SWM; w++) sw[w] = sw[w + 2];
#pragma unroll
        for (int w = 0; w + 1 < SWM / 2; w++) nw[w] = nw[w + 1];
    }
#pragma unroll
    for (int i = 0; i < B; i++) {
        u64 v = (u64)acc[i * NPL] + ((u64)acc[i * NPL + 1] << 8) + ((u64)acc[i * NPL + 2] << 16);
        if (NPL > 3) v += (u64)acc[i * NPL + 3] << 24;
        h[i] = v;
    }
}

// 64 consecutive rows of `stride` dwords (the wavefront's chunk of one mate) HBM -> this wavefront's LDS buffer with
// coalesced 16-byte loads.  A lane reading its own 152-byte row straight from HBM touches a fresh 128-byte line per
// 8-byte load: the rows of 12 resident wavefronts do not fit the 32 KB L1, so nearly every load refetched its line
// from L2 (profiles/r03e: the load sweep alone took 1.0 ms per 4 M pairs).  Row r of the buffer starts at dword
// r * stride: stride / 2 is odd for the common read lengths, so the 64-bit row reads of a half-wave are conflict-free.
template <int NB>   // vectors per lane in flight: what one stage of the caller's rows needs, so that it is one round trip
FQ_DEV void lane_stage_rows(u32* buf, const u32* src, int rows, int stride, int lane) {
    wave_order();   // the buffer's previous contents have been read
    const int bytes = rows * stride * 4;
    typedef u32 vec16 __attribute__((vector_size(16)));   // (a register value: an array of the u32x4 struct stays in scratch)
    const vec16* s4 = (const vec16*)src;
    vec16* d4 = (vec16*)buf;
    const int n16 = bytes >> 4;
    // NB vectors per lane are in flight before the first is stored (clamped indices instead of
    // branches): as a plain copy loop every 1 KB piece waited for its own round trip to memory - ten per quality stage
    for (int base = lane; base - lane < n16; base += 64 * NB) {
        vec16 v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) v[k] = s4[imin(base + 64 * k, n16 - 1)];
        sched_fence();   // (the scheduler sinks each load to its store otherwise)
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (base + 64 * k < n16) d4[base + 64 * k] = v[k];
    }
    if ((bytes & 8) && lane == 0) ((u64*)buf)[2 * n16] = ((const u64*)src)[2 * n16];
    wave_order();
}

// Pull `bytes` bytes behind `src` into L2: one dword of every 128-byte line by global_load_lds_dword into `sink` (64 dwords of LDS
// nobody reads).  Nothing waits for it; the later loads of the same lines by lane_stage_rows are L2 hits.
FQ_DEV void lane_prefetch_lines(u32* sink, const u32* src, int bytes, int lane, int step) {
    for (int off = lane * step; off < bytes; off += 64 * step) glds4((const char*)src + off, (char*)sink, lane);
}

// The same copy without registers (round 5): global_load_lds_dwordx4, 16 bytes per lane straight into the stage, asynchronous.
// Issued for read 2's quality rows as soon as read 1 is done with the stage; Duplicate's hash of read 1 runs while they are on
// their way (lane_body); lane_stage_rows_done() before the sweep.  The image is the same flat copy (lane-linear: piece k of the
// wave = vectors 64 k .. 64 k + 63).
FQ_DEV void lane_stage_rows_async(u32* buf, const u32* src, int rows, int stride, int lane) {
    wave_order();   // the buffer's previous contents have been read
    const int n16 = (rows * stride * 4) >> 4;
    for (int base = 0; base < n16; base += 64)
        if (base + lane < n16) glds16((const char*)src + 16 * (size_t)(base + lane), (char*)buf + 16 * (size_t)base, lane);
}
FQ_DEV void lane_stage_rows_done(u32* buf, const u32* src, int rows, int stride, int lane) {
    glds_wait();
    const int bytes = rows * stride * 4, n16 = bytes >> 4;
    if ((bytes & 8) && lane == 0) ((u64*)buf)[2 * n16] = ((const u64*)src)[2 * n16];
    wave_order();
}

// bit j of the result = the window that starts at base j of this 32-base word (q[0..7] its quality dwords, q[8..9] the
// two behind them) has total quality < threshold: v_alignbit (the window's bytes), v_sad_u8 with -threshold as the
// addend, v_alignbit to shift the sign into the mask.  WIDE: windows of 5..8 bases need a second v_sad_u8.
// windows of exactly four bases (the default --cut_right_window_size): v_qsad_pk_u16_u8 sums four consecutive windows per
// instruction; nthr4 = the negated threshold in each 16-bit field, so a field's sign bit says "window sum < threshold"
FQ_DEV u32 lane_window_word4(const u32 (&q)[10], u64 nthr4) {
    u32 m = 0;
#pragma unroll
    for (int d = 7; d >= 0; d--) {
        const u64 s = sum_bytes_sliding4(q[d] & 0x7F7F7F7Fu, q[d + 1] & 0x7F7F7F7Fu, nthr4);
        const u32 lo = (u32)s, hi = (u32)(s >> 32);
        m = alignbit(m, hi, 31);         // window 4d + 3
        m = alignbit(m, hi << 16, 31);   //        4d + 2
        m = alignbit(m, lo, 31);         //        4d + 1
        m = alignbit(m, lo << 16, 31);   //        4d
    }
    return m;
}
template <bool WIDE>
FQ_DEV u32 lane_window_word(const u32 (&q)[10], u32 keep_lo, u32 keep_hi, u32 nthr) {
    u32 m = 0;
#pragma unroll
    for (int d = 7; d >= 0; d--) {
        const u32 q0 = q[d] & 0x7F7F7F7Fu, q1 = q[d + 1] & 0x7F7F7F7Fu, q2 = q[d + 2 < 10 ? d + 2 : 9] & 0x7F7F7F7Fu;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const u32 x = k ? alignbit(q1, q0, 8 * k) : q0;
            u32 sdiff = sum_bytes(x & keep_lo, nthr);
            if (WIDE) {
                const u32 y = k ? alignbit(q2, q1, 8 * k) : q1;
                sdiff = sum_bytes(y & keep_hi, sdiff);
            }
            m = alignbit(m, sdiff, 31);   // m = m << 1 | (sum < thr)
        }
    }
    return m;
}

// Load read `g` of one mate: bases, N mask, and in ONE sweep over the quality row the window predicate of cut_right /
// cut_tail (the one that is enabled) and the per-32-base partial sums of countQualityMetrics.  `stage` = this
// wavefront's LDS buffer, chunk0 = first unit of its chunk, rows = units the chunk has.
// the read's length and packed bases (the mate's base rows through the stage)
template <int SWM>
FQ_DEV void lane_load_bases(const KernelArgs& a, u32* stage, const u32* seq, const u16* lenp, int chunk0, int rows, int lane, bool valid, LaneRead<SWM>& r) {
    const int swg = a.p.sw_g;
    const int g = chunk0 + lane;
    r.rl0 = valid ? (int)lenp[g] : 0;
    r.len = r.rl0;
    r.flags = 0;
    lane_stage_rows<(SWM + 3) / 4>(stage, seq + (size_t)chunk0 * swg, rows, swg, lane);
    {
        const u64* srow = (const u64*)(stage + lane * swg);
#pragma unroll
        for (int w = 0; w < SWM; w += 2) {
            // (a word past the row's stride holds the next row's bases: every consumer masks by the read's length)
            const u64 v = srow[w >> 1];
            r.s[w] = (u32)v;
            r.s[w + 1] = (u32)(v >> 32);
        }
    }
}
template <int SWM, bool KEEP>
FQ_DEV void lane_sweep_quality(const KernelArgs& a, u32* stage, u32* part, int lane, int win, int thr, u32 thr4, LaneRead<SWM>& r);
template <int SWM, bool KEEP>   // KEEP: leave countQualityMetrics' per-word sums in `part` (read 1 of a pair: its rows leave the stage)
FQ_DEV void lane_load_read(const KernelArgs& a, u32* stage, u32* part, const u32* seq, const u32* qual, const u16* lenp, int chunk0, int rows, int lane,
                           bool valid, int win, int thr, u32 thr4, LaneRead<SWM>& r) {
    lane_load_bases<SWM>(a, stage, seq, lenp, chunk0, rows, lane, valid, r);
    lane_stage_rows<FQ_LANE_STAGE_BATCH>(stage, qual + (size_t)chunk0 * a.p.qw_g, rows, a.p.qw_g, lane);
    lane_sweep_quality<SWM, KEEP>(a, stage, part, lane, win, thr, thr4, r);
}
// the sweep over the mate's quality rows in the stage: N mask, window predicate, read 1's partial sums
template <int SWM, bool KEEP>
FQ_DEV void lane_sweep_quality(const KernelArgs& a, u32* stage, u32* part, int lane, int win, int thr, u32 thr4, LaneRead<SWM>& r) {
    const int qwg = a.p.qw_g;
    const u64* qrow = (const u64*)(stage + lane * qwg);
    const u32 nthr = (u32)(-thr);
    const u64 nthr4 = 0x0001000100010001ull * (u64)(nthr & 0xFFFFu);
    const u32 keep_lo = lowmask32(8 * imin(imax(win, 1), 4)), keep_hi = win > 4 ? lowmask32(8 * (win - 4)) : 0u;
    u32 anyn = 0;
    // Eight dwords (one 32-position mask word) at a time from the END of the row: the window predicate of word W looks
    // two dwords into word W + 1, which the previous step left in nx0 / nx1.
    u32 nx0 = 0, nx1 = 0;
#pragma unroll
    for (int W = SWM / 2 - 1; W >= 0; W--) {
        u32 q[10];
#pragma unroll
        for (int d = 0; d < 8; d += 2) {
            const int c = 8 * W + d;
            const u64 v = qrow[c >> 1];
            q[d] = (u32)v;
            q[d + 1] = (u32)(v >> 32);
        }
        q[8] = nx0;
        q[9] = nx1;
        nx0 = q[0];
        nx1 = q[1];
        // ---- N mask of the word's 32 bases (bit 7 of the quality bytes), branch-free ----
        u32 nw = 0;
        // ... and countQualityMetrics' sums over the word's 32 bases (bytes behind the read are zeros: they add nothing)
        u32 ts = 0, gs = 0;
#pragma unroll
        for (int d = 7; d >= 0; d--) {
            const u32 b1 = (q[d] >> 7) & 0x01010101u;   // bits 0, 8, 16, 24
            nw = dot4_u8(b1, 0x08040201u, nw << 4);     // the four flags as a nibble behind the ones gathered so far
            if (KEEP && FQ_LANE_METRICS == 2) {
                ts = sum_bytes(q[d] & 0x7F7F7F7Fu, ts);
                gs += (u32)popc32(((q[d] | 0x80808080u) - thr4) & 0x80808080u);
            }
        }
        r.n[W] = nw;
        anyn |= nw;
        if (KEEP && FQ_LANE_METRICS == 2) part[W * 64 + lane] = ts | (gs << 16);   // sum of the quality characters | bases at or above the qualified quality << 16
        else { (void)ts; (void)gs; (void)part; (void)thr4; }
        // ---- window predicate (bad_window_word of the tile kernel, windows of up to 8 bases) ----
        u32 m = 0;
        if (win == 4) m = lane_window_word4(q, nthr4);                             // uniform
        else if (win > 4) m = lane_window_word<true>(q, keep_lo, keep_hi, nthr);
        else if (win > 0) m = lane_window_word<false>(q, keep_lo, keep_hi, nthr);
        r.bad[W] = m;
        FQ_LANE_FENCE();   // one mask word at a time: the scheduler would otherwise keep every word's dwords in flight
    }
    if (anyn) r.flags |= RS_HAS_N;
}

// quality character (7 bits) / N flag of base j of a row in global memory (indexed walks that end after a few bases)
FQ_DEV u32 g_qbyte(const u32* qual, int qwg, int g, int j) { return (u32)((const u8*)(qual + (size_t)g * qwg))[j]; }
FQ_DEV u32 g_code(const u32* seq, int swg, int g, int j) { return (u32)(((const u8*)(seq + (size_t)g * swg))[j >> 2] >> ((j & 3) * 2)) & 3u; }

// Filter::trimAndCut (filter.cpp:68-207) for the option family of this kernel: no front trim, no cut_front; cut_right
// or cut_tail (windows <= 8) and a fixed tail trim.  Returns false for NULL; `len` in / out.
// qrow: the lane's quality row in the wave's LDS stage (lane_load_read leaves it there) - the two short walks below
// are chains of dependent byte reads, a round trip to memory each when they went to the global row
template <int SWM>
FQ_DEV bool lane_trim_and_cut(const KernelArgs& a, const LaneRead<SWM>& r, const u8* qrow, int tail, int& len) {
    const DevParams& p = a.p;
    const bool enT = p.cut_tail, enR = p.cut_right;
    const int l = len;
    if (tail == 0 && !enT && !enR) return true;                 // :71-72
    int rlen = l - tail;
    if (rlen < 0) return false;                                 // :76-77
    if (!enT && !enR) { len = rlen; return true; }              // :79-89
    if (enR) {                                                  // :130-163
        const int w = p.wR;
        if (l - tail - w <= 0) return false;
        const int end = l - tail - w;
        int s = mask_first<SWM / 2>(r.bad, 0, end, true);       // first window below the threshold
        if (s < end) {                                          // foundLowQualWindow: while (s < l-1 && qual[s] >= 33+Q) s++
            const u32 qmin = (u32)imin(imax(p.qRmin, 0), 127);
            while (s < l - 1 && ((u32)qrow[s] & 0x7Fu) >= qmin) s++;
            rlen = s;
        }
    }
    if (!enR && enT) {                                          // :166-194
        const int w = p.wT;
        if (l - tail - w <= 0) return false;
        const int sp = mask_last<SWM / 2>(r.bad, 1, l - tail - w + 1, false);   // none: 0 (= front)
        int t = sp + w - 1;
        if (t < l - 1) t = t - w + 1;
        while (t >= 0 && ((u32)qrow[t] & 0x80u)) t--;                           // while (t >= 0 && seq[t] == 'N') t--
        rlen = t + 1;
    }
    if (rlen <= 0 || 0 >= l - 1) return false;                  // :196-197 (front == 0)
    len = rlen;
    return true;
}

// Filter::trimAndCut (filter.cpp:68-207) with a front trim and / or behind a UMI (DevParams::front_lane: no cut_front): the read
// as trimAndCut sees it is [u, u + l) of the row - u = what Read::trimFront took for the UMI (read.cpp:69-73) - `front` / `tail`
// are -f / -t.  out_front = frontTrimmed (without u), out_len the new length; false = NULL.  trim_and_cut() of the tile kernel
// on the register masks.
template <int SWM>
FQ_DEV bool lane_trim_and_cut_front(const KernelArgs& a, const LaneRead<SWM>& r, const u8* qrow, int u, int l, int front, int tail, int& out_front,
                                    int& out_len) {
    const DevParams& p = a.p;
    const bool enT = p.cut_tail, enR = p.cut_right, enF = p.front_per_read != 0;   // (--cut_front reaches this kernel only as front_per_read)
    out_front = 0;
    out_len = l;
    if (front == 0 && tail == 0 && !enT && !enR && !enF) return true;   // :71-72
    int rlen = l - front - tail;
    if (rlen < 0) return false;                                 // :76-77
    if (!enT && !enR && !enF) {                                 // :79-89
        out_front = front;
        out_len = rlen;
        return true;
    }
    if (enF) {                                                  // :97-127 quality cutting forward, on the same window predicate
        const int w = p.wF;
        if (l - front - tail - w <= 0) return false;
        const int end = l - tail - w;
        int s = mask_first<SWM / 2>(r.bad, u + front, u + end, false) - u;   // first window AT the threshold (none: s = end, the loop's exit)
        if (s > 0) s = s + w - 1;                               // "the trimming in front is forwarded"
        while (s < l && ((u32)qrow[u + s] & 0x80u)) s++;        // while (s < l && seq[s] == 'N') s++
        front = s;
        rlen = l - front - tail;
    }
    if (enR) {                                                  // :130-163
        const int w = p.wR;
        if (l - front - tail - w <= 0) return false;
        const int end = l - tail - w;
        int s = mask_first<SWM / 2>(r.bad, u + front, u + end, true) - u;   // first window below the threshold
        if (s < end) {                                          // foundLowQualWindow: while (s < l-1 && qual[s] >= 33+Q) s++
            const u32 qmin = (u32)imin(imax(p.qRmin, 0), 127);
            while (s < l - 1 && ((u32)qrow[u + s] & 0x7Fu) >= qmin) s++;
            rlen = s - front;
        }
    }
    if (!enR && enT) {                                          // :166-194
        const int w = p.wT;
        if (l - front - tail - w <= 0) return false;
        const int sp = mask_last<SWM / 2>(r.bad, u + front + 1, u + l - tail - w + 1, false) - u;   // none: front
        int t = sp + w - 1;
        if (t < l - 1) t = t - w + 1;
        while (t >= 0 && ((u32)qrow[u + t] & 0x80u)) t--;       // while (t >= 0 && seq[t] == 'N') t--
        rlen = t - front + 1;
    }
    if (rlen <= 0 || front >= l - 1) return false;              // :196-197
    out_front = front;
    out_len = rlen;
    return true;
}

// word idx of a register array, idx per lane (0 outside the array): a compare + select per word
template <int N>
FQ_DEV u32 lane_word_at(const u32 (&a)[N], int idx) {
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < N; k++) v = idx == k ? a[k] : v;
    return v;
}
// the 32 bases [end - 32, end) of the read in registers as "is a G" flags: bit 2 * (31 - t) <=> base end - 1 - t is a G
// (bases before the read's start: 0)
template <int SWM>
FQ_DEV u64 lane_g_window(const LaneRead<SWM>& r, int end) {
    const int b0 = end - 32;
    const int w0 = b0 >> 4;                        // floor: negative for windows that start before the read
    const u32 sh = (u32)(b0 & 15) * 2u;
    const u32 A = lane_word_at<SWM>(r.s, w0), B = lane_word_at<SWM>(r.s, w0 + 1), C = lane_word_at<SWM>(r.s, w0 + 2);
    const u32 lo = alignbit(B, A, sh), hi = alignbit(C, B, sh);
    // G = code 3 = both bits of the group; an N is stored as code 0 (fastp_gpu.h), so its group never shows a G
    const u32 glo = lo & (lo >> 1) & 0x55555555u, ghi = hi & (hi >> 1) & 0x55555555u;
    return (u64)glo | ((u64)ghi << 32);
}
// PolyX::trimPolyG (polyx.cpp:16-42) on [0, rlen): new length.  The walk from the tail runs on 32-base windows taken
// from the read's registers (as a walk over the global rows it was a round trip to memory per base, ten at least)
template <int SWM>
FQ_DEV int lane_trim_poly_g(const LaneRead<SWM>& r, bool active, int rlen, int compareReq) {
    int mismatch = 0, i = 0, firstGPos = rlen - 1;
    bool done = !active || rlen <= 0;
    for (int k = 0; ballot(!done) != 0ull; k++) {   // wave-uniform
        const u64 gw = lane_g_window<SWM>(r, rlen - 32 * k);
        for (int t = 0; t < 32; t++) {
            if (ballot(!done) == 0ull) break;
            if (!done) {
                if (i >= rlen) {
                    done = true;                     // the loop ran out: i == rlen
                } else {
                    const bool isg = ((gw >> (62 - 2 * t)) & 1ull) != 0ull;
                    if (!isg) mismatch++;
                    else firstGPos = rlen - i - 1;
                    const int allowed = (i + 1) / 8;
                    if (mismatch > 5 || (mismatch > allowed && i >= compareReq - 1)) done = true;   // break: i stays
                    else i++;
                }
            }
        }
    }
    if (!active) return rlen;
    if (i >= compareReq && firstGPos >= 0) return firstGPos;
    return rlen;
}

// ---------------------------------------------------------------------------
// OverlapAnalysis::analyze (overlapanalysis.cpp:17-89), the no-gap part.  X slides over Y: forward X = r1', Y = rc(r2');
// reverse X = rc(r2'), Y = r1'.  Prefilter: per offset the 2-bit XOR / popcount of 16 bases against Y's first 16 -
// X's window at a STATIC offset is one v_alignbit of two registers - survivors are verified exactly, smallest
// offset first; the first one that passes is the reference's answer for the direction.
// ---------------------------------------------------------------------------
// candidate offsets of one direction: bit (15 - t) of half-word b of cm = offset 16 b + t survives
template <int SWM>
FQ_DEV void lane_scan(const u32 (&X)[SWM], u32 y0, int nvalid, u32 premask, u32 nlim, u32 (&cm)[SWM / 2]) {
#pragma unroll
    for (int W = 0; W < SWM / 2; W++) cm[W] = 0;
#pragma unroll
    for (int b = 0; b < SWM; b++) {
        const int o0 = 16 * b;
        if (ballot(nvalid > o0) != 0ull) {   // uniform: some lane has offsets this far out
            const u32 w0 = X[b], w1 = b + 1 < SWM ? X[b + 1] : 0u;
            u32 cand = 0;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const u32 x = t ? alignbit(w1, w0, 2 * t) : w0;
                const u32 d = x ^ y0;
                const u32 sd = (u32)popc32((d | (d >> 1)) & premask) + nlim;   // negative <=> count <= lmax
                cand = alignbit(cand, sd, 31);
            }
            cand &= 0xFFFFu;
            const int left = nvalid - o0;
            if (left < 16) cand = left <= 0 ? 0u : (cand & ~lowmask32(16 - left));
            cm[b >> 1] |= cand << (16 * (b & 1));
        }
        FQ_LANE_FENCE();
    }
}
// smallest candidate offset left in cm (removed from it), or -1
template <int SWM>
FQ_DEV int lane_next_candidate(u32 (&cm)[SWM / 2]) {
    int o = -1;
#pragma unroll
    for (int W = SWM / 2 - 1; W >= 0; W--) {
        const u32 lo = cm[W] & 0xFFFFu, hi = cm[W] >> 16;
        if (hi) o = 32 * W + 16 + (clz32(hi) - 16);
        if (lo) o = 32 * W + (clz32(lo) - 16);
    }
    if (o >= 0) {
        const int b = o >> 4, t = o & 15;
        const u32 bit = (0x8000u >> t) << (16 * (b & 1));
#pragma unroll
        for (int W = 0; W < SWM / 2; W++)
            if (W == (b >> 1)) cm[W] &= ~bit;
    }
    return o;
}
// acceptNoGapOverlap (:34-44) of X shifted by o against Y; -1 or the reported difference count
template <int SWM>
FQ_DEV int lane_verify(const u32 (&X)[SWM], const u32 (&XN)[SWM / 2], const u32 (&Y)[SWM], const u32 (&YN)[SWM / 2], bool hasN, int o,
                       int lenX, int lenY, const u16* lut) {
    const int ol = imin(lenX - o, lenY);
    const int limit = (int)lut[ol];
    const int pre = imin(ol, 50);   // complete_compare_require (:28)
    u32 xs[SWM];
#pragma unroll
    for (int w = 0; w < SWM; w++) xs[w] = X[w];
    base_shift_down<SWM>(xs, (u32)o);
    u32 dn[SWM / 2];   // positions where exactly one side is N
    if (hasN) {
#pragma unroll
        for (int w = 0; w < SWM / 2; w++) dn[w] = XN[w];
        bit_shift_down<SWM / 2>(dn, (u32)o);
#pragma unroll
        for (int w = 0; w < SWM / 2; w++) dn[w] ^= YN[w];
    }
    int cnt_pre = 0, cnt_full = 0;
#pragma unroll
    for (int w = 0; w < SWM; w++) {
        const int t = 16 * w;
        u32 dd = fold_diff(xs[w] ^ Y[w]);
        if (hasN) dd |= nmask_word<SWM / 2>(dn, w);
        const int rem = ol - t, remp = pre - t;
        cnt_full += popc32(rem >= 16 ? dd : (rem <= 0 ? 0u : (dd & lowmask32(2 * rem))));
        cnt_pre += popc32(remp >= 16 ? dd : (remp <= 0 ? 0u : (dd & lowmask32(2 * remp))));
    }
    if (cnt_pre > limit) return -1;
    return ol > 50 ? cnt_full : cnt_pre;
}

// ---------------------------------------------------------------------------
// fastp_simd::countQualityMetrics (simd.cpp:54-119) of [0, len): total (qual - 33), bases below the qualified quality, N.
// Round 3 staged every quality row a second time through LDS and masked all 38 dwords by the window (two more round trips
// per chunk, 32 instructions per dword).  Now nothing is staged again:
//   * the read staged LAST (read 2 of a pair, the read of a single-end run) still has its quality rows in the wavefront's
//     stage when the final window is known: a dword wholly inside the window is summed as it is and committed by one
//     select, the one dword the window's end cuts is read by its index and masked (9 instructions per dword);
//   * read 1 of a pair (its rows are gone by then) gets the 32-base words its window covers whole from the partial sums
//     the load sweep left in LDS, and the ONE word the window's end cuts from its row in memory (32 bytes per lane, an
//     L2 hit: the row was staged moments ago).
// The N count comes from the N mask in registers.
// ---------------------------------------------------------------------------
template <int SWM>
FQ_DEV int lane_count_n(const LaneRead<SWM>& r, int len) {
    u32 n = 0;
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) {
        const int left = len - 32 * w;
        n += (u32)popc32(left >= 32 ? r.n[w] : (left <= 0 ? 0u : (r.n[w] & lowmask32(left))));
    }
    return (int)n;
}
// the read whose quality rows are still in the stage (row = this lane's row there)
template <int SWM>
FQ_DEV void lane_metrics_stage(const KernelArgs& a, const u32* row, const LaneRead<SWM>& r, int len, int& tot, int& low, int& nb) {
    const int qwg = a.p.qw_g;
    const u64* qrow = (const u64*)row;
    const u32 thr4 = (u32)a.p.qual_thr * 0x01010101u;
    const int full = len >> 2, part = len & 3;
    const u32 cutmask = lowmask32(8 * part);   // 0 when the window ends on a dword boundary
    const u32 qb = row[imin(full, qwg - 1)] & 0x7F7F7F7Fu & cutmask;
    u32 t = sum_bytes(qb, 0u);
    u32 ge = (u32)popc32(((qb | 0x80808080u) - thr4) & 0x80808080u & cutmask);
#pragma unroll
    for (int c = 0; c < 4 * SWM; c += 2) {
        const u64 v = qrow[c >> 1];   // (dwords behind the row's stride are never inside the window)
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const u32 q7 = (hlf ? (u32)(v >> 32) : (u32)v) & 0x7F7F7F7Fu;
            const bool in = c + hlf < full;
            const u32 t2 = sum_bytes(q7, t);
            const u32 g2 = ge + (u32)popc32(((q7 | 0x80808080u) - thr4) & 0x80808080u);
            t = in ? t2 : t;
            ge = in ? g2 : ge;
        }
    }
    tot = (int)t - 33 * len;
    low = len - (int)ge;
    nb = lane_count_n<SWM>(r, len);
}
// The same with a front (DevParams::front_lane): the window is [f, f + len) of the row = the prefix up to f + len minus the
// prefix up to f (the N count comes from the N mask, which has been moved down by f with the bases).
template <int SWM>
FQ_DEV void lane_metrics_stage_front(const KernelArgs& a, const u32* row, const LaneRead<SWM>& r, int f, int len, int& tot, int& low, int& nb) {
    int t1, l1, n1, t0, l0, n0;
    lane_metrics_stage<SWM>(a, row, r, f + len, t1, l1, n1);   // tot = sum - 33 x, low = x - ge
    lane_metrics_stage<SWM>(a, row, r, f, t0, l0, n0);
    tot = t1 - t0;
    low = l1 - l0;
    nb = lane_count_n<SWM>(r, len);
}
// read 1 of a pair: partial sums + the cut word
struct LaneCutWord {
    u64 v[4];   // the 32 quality bytes of the word the window's end cuts
};
FQ_DEV void lane_cut_fetch(const KernelArgs& a, const u32* qual, int g, bool valid, int len, LaneCutWord& cw) {
    const int qwg = a.p.qw_g;
    const u64* row = (const u64*)(qual + (size_t)g * qwg);
    const int w8 = 4 * (len >> 5);   // first u64 of the cut word
#pragma unroll
    for (int k = 0; k < 4; k++) cw.v[k] = valid ? row[imin(w8 + k, (qwg >> 1) - 1)] : 0ull;   // (a u64 behind the row is never inside the window)
}
template <int SWM>
FQ_DEV void lane_metrics_part(const KernelArgs& a, const LaneRead<SWM>& r, const u32* part, int lane, const LaneCutWord& cw, int len, int& tot,
                              int& low, int& nb) {
    const u32 thr4 = (u32)a.p.qual_thr * 0x01010101u;
    const int wfull = len >> 5, rem = len & 31;
    u32 acc = 0;   // sum | count << 16, as the partial sums
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) {
        const u32 pw = part[w * 64 + lane];
        acc += w < wfull ? pw : 0u;
    }
    const int fullb = rem >> 2, partb = rem & 3;
    const u32 cutmask = lowmask32(8 * partb);   // 0 when the window ends on a dword boundary
    u32 t = 0, ge = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const u32 qd = (d & 1) ? (u32)(cw.v[d >> 1] >> 32) : (u32)cw.v[d >> 1];
        const u32 M = d < fullb ? 0xFFFFFFFFu : (d == fullb ? cutmask : 0u);
        const u32 q7 = qd & 0x7F7F7F7Fu & M;
        t = sum_bytes(q7, t);
        ge += (u32)popc32(((q7 | 0x80808080u) - thr4) & 0x80808080u & M);
    }
    tot = (int)((acc & 0xFFFFu) + t) - 33 * len;
    low = len - (int)((acc >> 16) + ge);
    nb = lane_count_n<SWM>(r, len);
}

template <int SWM>
FQ_DEV void lane_metrics_part_front(const KernelArgs& a, const LaneRead<SWM>& r, const u32* part, int lane, const LaneCutWord& cwe, const LaneCutWord& cwf,
                                    int f, int len, int& tot, int& low, int& nb) {
    int t1, l1, n1, t0, l0, n0;
    lane_metrics_part<SWM>(a, r, part, lane, cwe, f + len, t1, l1, n1);
    lane_metrics_part<SWM>(a, r, part, lane, cwf, f, t0, l0, n0);
    tot = t1 - t0;
    low = l1 - l0;
    nb = lane_count_n<SWM>(r, len);
}

// ---------------------------------------------------------------------------
// AdapterTrimmer::trimBySequence (adaptertrimmer.cpp:64-157) with the read in registers, for adapters of at most 64
// bases (four uniform words: -a / --adapter_sequence_r2 and the ones the Evaluator detects are <= 60; longer ones keep
// the tile kernels).  Same three passes as trim_by_sequence() of the tile kernel: the no-gap scan over pos = start ..
// rlen - matchReq - 1 (here: negative positions by four static compares against the shifted adapter, the positions with
// 16 bases to look at by the overlap prefilter's 6 instructions per position + an exact check of the survivors in
// ascending order, the last positions on a 16-base window cut out of the read's tail), then Matcher's one-insertion and
// one-deletion forms in closed form (one_gap_match_mask: one walk gives the answer for every compare length).
// ---------------------------------------------------------------------------
enum { LANE_ADAPTER_WORDS = 4 };
// 16 bases (and their N flags, spread to bit 2k) of the read starting at base bp (any sign; outside the words: zeros)
template <int SWM>
FQ_DEV void lane_window16(const LaneRead<SWM>& r, int bp, u32& codes, u32& nsp) {
    const int w0 = bp >> 4;   // floor
    const u32 sh = (u32)(bp & 15) * 2u;
    const u32 A = lane_word_at<SWM>(r.s, w0), B = lane_word_at<SWM>(r.s, w0 + 1);
    codes = alignbit(B, A, sh);
    const int b0 = bp >> 5;
    const u32 nsh = (u32)(bp & 31);
    const u32 NA = lane_word_at<SWM / 2>(r.n, b0), NB = lane_word_at<SWM / 2>(r.n, b0 + 1);
    nsp = spread16(alignbit(NB, NA, nsh));
}
// mismatches of n (<= 64) bases: xs (the read moved so that the compare starts at base 0; xn = its N flags, 1 bit per base)
// against the uniform words aw
template <int SWM>
FQ_DEV int lane_adapter_mm(const u32 (&xs)[SWM], const u32 (&xn)[SWM / 2], bool hasN, const u32 (&aw)[LANE_ADAPTER_WORDS], int n) {
    int mm = 0;
#pragma unroll
    for (int w = 0; w < LANE_ADAPTER_WORDS; w++) {
        u32 d = fold_diff(xs[w] ^ aw[w]);
        if (hasN) d |= nmask_word<SWM / 2>(xn, w);
        const int rem = n - 16 * w;
        mm += popc32(rem >= 16 ? d : (rem <= 0 ? 0u : (d & lowmask32(2 * rem))));
    }
    return mm;
}
// Matcher::matchWithOneInsertion for every compare length at once (one_gap_match_mask of the tile kernel): D0 / D1 =
// "position k differs" of the aligned / the shifted comparison, 2 bits per position as fold_diff leaves them; returns
// bit (c - 1) <=> matched for compare length c (c <= cmax <= 64)
FQ_DEV u64 lane_gap_mask(const u32 (&D0)[LANE_ADAPTER_WORDS], const u32 (&D1)[LANE_ADAPTER_WORDS], int cmax, int bound) {
    u32 ok_lo = 0, ok_hi = 0;
    int p0 = 0, p1 = 0, gmin = 4096;   // 4096 = "no split point yet": never passes a limit
#pragma unroll
    for (int k = 0; k < 16 * LANE_ADAPTER_WORDS; k++) {
        if (k < bound) {   // uniform: the adapter's length
            const int c = k + 1;
            p0 += (int)((D0[k >> 4] >> (2 * (k & 15))) & 1u);
            p1 += (int)((D1[k >> 4] >> (2 * (k & 15))) & 1u);
            const bool hit = c <= cmax && gmin + p1 <= c / 8 - 1;
            if (c <= 32) ok_lo |= hit ? (1u << ((c - 1) & 31)) : 0u;
            else ok_hi |= hit ? (1u << ((c - 1) & 31)) : 0u;
            gmin = imin(gmin, p0 - p1);
        }
    }
    return (u64)ok_lo | ((u64)ok_hi << 32);
}
// the position the reference's loop over pos picks from such a mask: compare length cfirst at pos 0, one less per
// position once the read's end is nearer than the adapter's (quirk #7: the strings are not advanced by pos); -1 = none
FQ_DEV int lane_gap_pick(u64 ok, int cfirst, int npos, int rlen_minus) {
    if (cfirst < 1 || npos <= 0) return -1;
    const u64 m = cfirst >= 64 ? ok : (ok & ((1ull << cfirst) - 1ull));
    if (m == 0ull) return -1;
    if ((m >> (cfirst - 1)) & 1ull) return 0;
    const int c = 64 - clz64(m);             // the largest compare length that matches
    const int pos = rlen_minus - c;          // where the loop reaches it
    return pos < npos ? pos : -1;
}
template <int SWM>
FQ_DEV bool lane_trim_by_sequence(const LaneRead<SWM>& r, int rlen, const u32 (&aw)[LANE_ADAPTER_WORDS], int alen, int matchReq, int& out_pos) {
    if (alen < matchReq) return false;                      // (uniform)
    const bool hasN = (r.flags & RS_HAS_N) != 0;
    const int last = rlen - matchReq;                       // pos runs to last - 1
    bool found = false;
    int pos_found = 0;
    // ---- negative positions: the read's head against the adapter without its first -pos bases (:79-86) ----
    const int start = alen >= 16 ? -4 : (alen >= 12 ? -3 : (alen >= 8 ? -2 : 0));
#pragma unroll
    for (int pos = -4; pos < 0; pos++) {
        if (pos >= start) {   // uniform
            const int so = -pos;
            u32 as[LANE_ADAPTER_WORDS];
#pragma unroll
            for (int w = 0; w < LANE_ADAPTER_WORDS; w++) as[w] = alignbit(w + 1 < LANE_ADAPTER_WORDS ? aw[w + 1] : 0u, aw[w], 2u * (u32)so);
            const int cmplen = imin(rlen - pos, alen);
            const int mm = lane_adapter_mm<SWM>(r.s, r.n, hasN, as, cmplen - so);
            if (!found && pos < last && mm <= cmplen / 8) { found = true; pos_found = pos; }
        }
    }
    // ---- positions with npre bases to look at: prefilter + exact check, ascending ----
    const int npre = imin(16, alen);
    const int nmain = imin(rlen - npre + 1, last);          // positions [0, nmain)
    if (ballot(!found && nmain > 0) != 0ull) {
        u32 cm[SWM / 2];
        const u32 premask = lowmask32(2 * npre) & 0x55555555u;
        lane_scan<SWM>(r.s, aw[0], found ? 0 : nmain, premask, (u32)(-(alen / 8 + 1)), cm);
        for (;;) {
            const int o = found ? -1 : lane_next_candidate<SWM>(cm);
            if (ballot(o >= 0) == 0ull) break;
            if (o >= 0) {
                u32 xs[SWM], xn[SWM / 2];
#pragma unroll
                for (int w = 0; w < SWM; w++) xs[w] = r.s[w];
                base_shift_down<SWM>(xs, (u32)o);
#pragma unroll
                for (int w = 0; w < SWM / 2; w++) xn[w] = r.n[w];
                if (hasN) bit_shift_down<SWM / 2>(xn, (u32)o);
                const int cmplen = imin(rlen - o, alen);
                if (lane_adapter_mm<SWM>(xs, xn, hasN, aw, cmplen) <= cmplen / 8) { found = true; pos_found = o; }
            }
        }
    }
    // ---- the last positions: fewer than npre bases left (compare length k = rlen - pos, matchReq < k < npre) ----
    if (ballot(!found && rlen > matchReq) != 0ull) {
        u32 tc, tn;
        lane_window16<SWM>(r, rlen - 16, tc, tn);           // bases [rlen - 16, rlen)
#pragma unroll
        for (int k = 15; k > 0; k--) {
            if (k < npre && k > matchReq) {   // uniform
                const u32 sh = 2u * (u32)(16 - k);
                u32 d = fold_diff((tc >> sh) ^ aw[0]);
                if (hasN) d |= tn >> sh;
                const int mm = popc32(d & lowmask32(2 * k));
                const int pos = rlen - k;
                if (!found && pos >= 0 && pos >= nmain && mm <= k / 8) { found = true; pos_found = pos; }
            }
        }
    }
    // ---- one insertion / one deletion in the read (:105-135), both on the strings as they start (quirk #7) ----
    if (ballot(!found && rlen - matchReq > 0) != 0ull) {
        u32 D0[LANE_ADAPTER_WORDS], D1[LANE_ADAPTER_WORDS];
        u32 n1[SWM / 2];
#pragma unroll
        for (int w = 0; w < SWM / 2; w++) n1[w] = alignbit(w + 1 < SWM / 2 ? r.n[w + 1] : 0u, r.n[w], 1);   // N flags of the read moved down one base
#pragma unroll
        for (int w = 0; w < LANE_ADAPTER_WORDS; w++) {
            const u32 s1 = alignbit(r.s[w + 1], r.s[w], 2);                  // read[k + 1]
            D0[w] = fold_diff(r.s[w] ^ aw[w]);
            D1[w] = fold_diff(s1 ^ aw[w]);
            if (hasN) { D0[w] |= nmask_word<SWM / 2>(r.n, w); D1[w] |= nmask_word<SWM / 2>(n1, w); }
        }
        if (!found && rlen - matchReq - 1 > 0) {             // insertion: ins = read, nor = adapter
            const int cmax = imin(rlen - 1, alen);
            const int pos = lane_gap_pick(lane_gap_mask(D0, D1, cmax, imin(alen, 64)), cmax, rlen - matchReq - 1, rlen - 1);
            if (pos >= 0) { found = true; pos_found = pos; }
        } else {
            (void)lane_gap_mask(D0, D1, 0, 0);
        }
        if (ballot(!found && rlen - matchReq > 0) != 0ull) {
#pragma unroll
            for (int w = 0; w < LANE_ADAPTER_WORDS; w++) {               // deletion: ins = adapter, nor = read: D1 = adapter[k + 1] != read[k]
                const u32 a1 = alignbit(w + 1 < LANE_ADAPTER_WORDS ? aw[w + 1] : 0u, aw[w], 2);
                D1[w] = fold_diff(r.s[w] ^ a1);
                if (hasN) D1[w] |= nmask_word<SWM / 2>(r.n, w);
            }
            if (!found && rlen - matchReq > 0) {
                const int cmax = imin(rlen, alen - 1);
                const int pos = lane_gap_pick(lane_gap_mask(D0, D1, cmax, imin(alen - 1, 64)), cmax, rlen - matchReq, rlen);
                if (pos >= 0) { found = true; pos_found = pos; }
            }
        }
    }
    out_pos = pos_found;
    return found;
}
// the bookkeeping of AdapterTrimmer::trimBySequence's hit (:138-156): new length, the string handed to addAdapterTrimmed
FQ_DEV void lane_apply_adapter(u32* misc, int pos, int alen, int& len, u32& apos, u32& alen_out) {
    int adapter_len;
    if (pos < 0) { adapter_len = alen + pos; len = 0; }
    else { adapter_len = len - pos; len = pos; }
    if (adapter_len > 0) lds_add_u32(&misc[MISC_ADAPTER_BASES], (u32)adapter_len);
    apos = (u32)pos & 0xFFFFu;
    alen_out = (u32)adapter_len;
}

// AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62) on a read in registers (round 6: --adapter_fasta lists whose
// sequences are <= 64 bases): every sequence of the list in turn on the shrinking read - trimBySequence as for -a - each cut
// reported as a fastp_gpu_adapter_event (the host replays FilterResult's adapter map from them).  A wave collective: `go` lanes
// take part, `cur` = the lane's read length in / out.  which: the read's index in the run's stream (2 g + mate, or g).
template <int SWM>
FQ_DEV bool lane_fasta_trims(const KernelArgs& a, u32* misc, const LaneRead<SWM>& r, bool go, int& cur, u32 read_index) {
    const DevParams& p = a.p;
    bool trimmed = false;
    for (int i = 0; i < p.n_fasta; i++) {                      // (uniform)
        if (ballot(go && cur > 0) == 0ull) break;
        u32 fw[LANE_ADAPTER_WORDS];
#pragma unroll
        for (int w = 0; w < LANE_ADAPTER_WORDS; w++) fw[w] = a.lut.fasta_words[(size_t)i * ADAPT_WORDS + w];   // (uniform loads)
        const int alen = a.lut.fasta_len[i];
        int pos = 0;
        const int rlen = cur;
        const bool hit = lane_trim_by_sequence<SWM>(r, go ? rlen : 0, fw, alen, p.fasta_match_req, pos);
        if (go && hit) {
            u32 ap, al;
            lane_apply_adapter(misc, pos, alen, cur, ap, al);
            trimmed = true;
            if (a.adapter_events) {
                const int slot = g_atomic_add_i32(a.n_adapter_events, 1);
                if (slot < a.adapter_events_capacity) {
                    a.adapter_events[3 * slot] = read_index;
                    a.adapter_events[3 * slot + 1] = ((u32)pos & 0xFFFFu) | ((al & 0xFFFFu) << 16);
                    a.adapter_events[3 * slot + 2] = (u32)i;
                }
            }
        }
    }
    return trimmed;
}

// ---------------------------------------------------------------------------
// PolyX::trimPolyX (polyx.cpp:49-116) on [0, rlen) of a read in registers: the walk from the tail over 32-base windows
// (codes and N flags cut out of the registers), then the reference's step back to the first base of the winning letter.
// ---------------------------------------------------------------------------
template <int SWM>
FQ_DEV u32 lane_sym_at(const LaneRead<SWM>& r, int j) {   // A0 T1 C2 G3 N4; 5 outside the read's words
    const u32 w = lane_word_at<SWM>(r.s, j >> 4), nw = lane_word_at<SWM / 2>(r.n, j >> 5);
    const u32 code = (w >> (2 * (j & 15))) & 3u;
    return ((nw >> (j & 31)) & 1u) ? 4u : code;
}
template <int SWM>
FQ_DEV int lane_trim_poly_x(const LaneRead<SWM>& r, bool active, int rlen, int compareReq, int& poly, int& trimmed) {
    int cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0, pos = 0;
    bool done = !active || rlen <= 0;
    poly = -1;
    trimmed = 0;
    for (int k = 0; ballot(!done) != 0ull; k++) {   // wave-uniform
        const int end = rlen - 32 * k;              // the window holds bases [end - 32, end)
        u32 lo, hi, nlo, nhi;
        lane_window16<SWM>(r, end - 32, lo, nlo);
        lane_window16<SWM>(r, end - 16, hi, nhi);
        for (int t = 0; t < 32; t++) {
            if (ballot(!done) == 0ull) break;
            if (!done) {
                if (pos >= rlen) {
                    done = true;                    // the loop ran out: pos == rlen
                } else {
                    const int g = 15 - (t & 15);    // base end - 1 - t
                    const u32 cw = t < 16 ? hi : lo, nw = t < 16 ? nhi : nlo;
                    const u32 code = (cw >> (2 * g)) & 3u;
                    const bool isn = ((nw >> (2 * g)) & 1u) != 0u;
                    cnt0 += (isn || code == 0u) ? 1 : 0;   // N counts toward all (:79-85)
                    cnt1 += (isn || code == 1u) ? 1 : 0;
                    cnt2 += (isn || code == 2u) ? 1 : 0;
                    cnt3 += (isn || code == 3u) ? 1 : 0;
                    const int cmp = pos + 1;
                    const int allowed = imin(5, cmp / 8);
                    const bool brk = !((cmp - cnt0 <= allowed) | (cmp - cnt1 <= allowed) | (cmp - cnt2 <= allowed) | (cmp - cnt3 <= allowed));
                    if (brk && (pos >= 8 || pos + 1 >= compareReq - 1)) done = true;   // break: pos stays
                    else pos++;
                }
            }
        }
    }
    const bool hit = active && pos + 1 >= compareReq;        // :98 (no early return: the loop below is a wave collective)
    int best = 0, mx = cnt0;                                 // first maximum in the order A, T, C, G
    if (cnt1 > mx) { mx = cnt1; best = 1; }
    if (cnt2 > mx) { mx = cnt2; best = 2; }
    if (cnt3 > mx) { mx = cnt3; best = 3; }
    // :109  while(data[rlen-pos-1] != polyBase && pos>=0) pos--;   index -1 and index rlen never equal the poly base
    bool walking = hit;
    while (ballot(walking) != 0ull) {
        if (walking) {
            const int idx = rlen - pos - 1;
            const bool is_poly = idx >= 0 && idx < rlen && lane_sym_at<SWM>(r, idx) == (u32)best;
            if (!is_poly && pos >= 0) pos--;
            else walking = false;
        }
    }
    if (!hit) return rlen;
    poly = best;
    trimmed = pos + 1;
    const int newlen = rlen - pos - 1;
    if (newlen < 0 || newlen > rlen) return rlen;   // Read::resize ignores (read.cpp:62-64)
    return newlen;
}

// fastp_simd::countAdjacentDiffs (simd.cpp:121-185) of [0, len): positions j in [1, len) whose symbol differs from the one before
template <int SWM>
FQ_DEV int lane_adjacent_diffs(const LaneRead<SWM>& r, int len) {
    const bool hasN = (r.flags & RS_HAS_N) != 0;
    u32 nd[SWM / 2];   // bit j: N(j) != N(j - 1)
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) nd[w] = r.n[w] ^ ((r.n[w] << 1) | (w ? r.n[w - 1] >> 31 : 0u));
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < SWM; w++) {
        const u32 prev = (r.s[w] << 2) | (w ? r.s[w - 1] >> 30 : 0u);
        u32 d = fold_diff(r.s[w] ^ prev);
        if (hasN) d |= nmask_word<SWM / 2>(nd, w);
        const int hi = len - 16 * w;                         // bases [16 w, 16 w + hi) of this word are inside
        u32 M = hi >= 16 ? 0x55555555u : (hi <= 0 ? 0u : (lowmask32(2 * hi) & 0x55555555u));
        if (w == 0) M &= ~1u;                                // base 0 has nothing before it
        cnt += popc32(d & M);
    }
    return cnt;
}

// round 3's form (FQ_LANE_METRICS == 0): the mate's quality rows are staged once more and every dword masked by the window
template <int SWM>
FQ_DEV void lane_metrics_staged(const KernelArgs& a, u32* stage, const u32* qual, int chunk0, int rows, int lane, int len, int& tot, int& low, int& nb) {
    const int qwg = a.p.qw_g;
    lane_stage_rows<FQ_LANE_STAGE_BATCH>(stage, qual + (size_t)chunk0 * qwg, rows, qwg, lane);
    const u64* qrow = (const u64*)(stage + lane * qwg);
    const u32 thr4 = (u32)a.p.qual_thr * 0x01010101u;
    u32 t = 0, lo = 0, n = 0;
#pragma unroll
    for (int c = 0; c < 4 * SWM; c += 2) {
        const u64 v = qrow[c >> 1];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const u32 qd = hlf ? (u32)(v >> 32) : (u32)v;
            const int rem = len - 4 * (c + hlf);
            const u32 M = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : lowmask32(8 * rem));
            const u32 q7 = qd & 0x7F7F7F7Fu & M;
            const u32 ge = ((q7 | 0x80808080u) - thr4) & 0x80808080u;
            t = sum_bytes(q7, t);
            lo += (u32)popc32(~ge & 0x80808080u & M);
            n += (u32)popc32(qd & 0x80808080u & M);
        }
    }
    tot = (int)t - 33 * len;
    low = (int)lo;
    nb = (int)n;
}

// ---------------------------------------------------------------------------
// BaseCorrector::correctByOverlapAnalysis (basecorrector.cpp:16-83) with both reads in registers (-c on the lane plan, round 5).
// The mismatch positions of the accepted overlap come from the same XOR words as the verification; they are taken one per
// round (a wave loop: at most the mismatch limit, five by default).  A corrected base changes the read's registers (what the
// adapter scan by sequence, polyX, the complexity filter and the N count look at), its quality changes what
// countQualityMetrics will add up: read 2's rows are still in the wavefront's stage - the byte is rewritten there; read 1's
// are not - its edited positions are marked in a per-lane bit mask (LaneLds::clist) and added up when the final window is
// known (lane_apply_corrected: the new quality of such a position is read 2's at the position it was compared with - any
// number of them: only the first 50 bases of an overlap are held to the mismatch limit, overlapanalysis.cpp:34-44).  Every edit goes
// to the engine's own correction list (the Stats fix-up, fq_corr_stats_kernel, and the overrepresentation analysis read it)
// and to the caller's, if there is one.
// ---------------------------------------------------------------------------
// symbol (A0 T1 C2 G3 N4) of base j of a row in registers
template <int SWM>
FQ_DEV u32 lane_sym_of(const u32 (&s)[SWM], const u32 (&n)[SWM / 2], int j) {
    const u32 w = lane_word_at<SWM>(s, j >> 4), nw = lane_word_at<SWM / 2>(n, j >> 5);
    return ((nw >> (j & 31)) & 1u) ? 4u : ((w >> (2 * (j & 15))) & 3u);
}
// base j of the read becomes symbol sym
template <int SWM>
FQ_DEV void lane_set_sym(LaneRead<SWM>& r, int j, u32 sym) {
    const u32 code = sym < 4u ? sym : 0u, sh = 2u * (u32)(j & 15);
#pragma unroll
    for (int w = 0; w < SWM; w++) {
        const u32 v = (r.s[w] & ~(3u << sh)) | (code << sh);
        r.s[w] = (j >> 4) == w ? v : r.s[w];
    }
    const u32 bit = 1u << (j & 31);
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) {
        const u32 v = sym == 4u ? (r.n[w] | bit) : (r.n[w] & ~bit);
        r.n[w] = (j >> 5) == w ? v : r.n[w];
    }
    if (sym == 4u) r.flags |= RS_HAS_N;
}
// the round's edits of a wavefront go to the lists together: ONE atomic per list and round takes the slots of all lanes that
// have one (a million atomics on one counter per launch serialise at L2; which < 0: this lane has none).  A wave collective.
FQ_DEV void lane_emit_corrections(const KernelArgs& a, int lane, int gp, int which, int rowpos, u32 nb, u32 nq) {
    const u64 em = ballot(which >= 0);
    if (em == 0ull) return;
    const int cnt = popc64(em), leader = ffs64(em) - 1;
    const int rank = popc64(em & ((1ull << lane) - 1ull));
    const u32 w0 = (u32)(2 * (a.first + gp) + which), w1 = (u32)rowpos | (sym_ascii(nb) << 16) | (nq << 24);
    int base = 0, cb = 0;
    if (lane == leader) {   // both lists' slots in ONE round trip (the two atomics are in flight together)
        base = g_atomic_add_i32(a.n_corr_int, cnt);   // never full: sized for an edit at every base of every pair
        if (a.corrections) cb = g_atomic_add_i32(a.n_corrections, cnt);
    }
    base = (int)shfl((u32)base, leader);
    cb = (int)shfl((u32)cb, leader);
    if (which >= 0 && base + rank < a.corr_int_cap) { a.corr_int[2 * (base + rank)] = w0; a.corr_int[2 * (base + rank) + 1] = w1; }
    if (a.corrections && which >= 0 && cb + rank < a.corr_capacity) { a.corrections[2 * (cb + rank)] = w0; a.corrections[2 * (cb + rank) + 1] = w1; }
}
#ifndef FQ_CORR_FLUSH
#define FQ_CORR_FLUSH 1   // (A/B: 0 = a list entry per round, tools/gpu_r6_n.sh)
#endif
// The same for a whole chunk: a lane keeps its first two edits (the list's second word each; which read: bits 2 / 3 of `st`, how
// many: its low two bits) and the wavefront takes the slots of all of them with ONE atomic per list when the rounds are over.
// Measured on the -c line (profiles/r06_l_corr_rounds_ablation.txt): the per-round form - a ballot, a returning atomic per
// list and the stores in EVERY round that edits - was 0.12 of the correction rounds' 0.20 ms per 2 Mi pairs.  A lane's third
// and later edits (rare) still go out round by round.
FQ_DEV void lane_flush_corrections(const KernelArgs& a, int lane, int gp, u32 st, u32 e0, u32 e1) {
    const u32 n = st & 3u;
    const u64 m1 = ballot(n >= 1u), m2 = ballot(n >= 2u);
    if (m1 == 0ull) return;
    const int cnt = popc64(m1) + popc64(m2), leader = ffs64(m1) - 1;
    const u64 below = (1ull << lane) - 1ull;
    const int rank = popc64(m1 & below) + popc64(m2 & below);
    int base = 0, cb = 0;
    if (lane == leader) {
        base = g_atomic_add_i32(a.n_corr_int, cnt);   // never full: sized for an edit at every base of every pair
        if (a.corrections) cb = g_atomic_add_i32(a.n_corrections, cnt);
    }
    base = (int)shfl((u32)base, leader);
    cb = (int)shfl((u32)cb, leader);
    const u32 r0 = (u32)(2 * (a.first + gp));
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (n > (u32)k) {
            const u32 w0 = r0 + ((st >> (2 + k)) & 1u), w1 = k ? e1 : e0;
            const int si = base + rank + k, sc = cb + rank + k;
            if (si < a.corr_int_cap) { a.corr_int[2 * si] = w0; a.corr_int[2 * si + 1] = w1; }
            if (a.corrections && sc < a.corr_capacity) { a.corrections[2 * sc] = w0; a.corrections[2 * sc + 1] = w1; }
        }
    }
}
// key = the pair's accepted overlap (no gap), l1 / l2 the lengths it was found on; rc / rcn = rc(r2') as the scan built it.
// q1row: read 1's quality row in memory, q2row: read 2's in the stage (both at the ORIGINAL read's start).  nc1 = entries of clist.
template <int SWM>
FQ_DEV void lane_correct(const KernelArgs& a, u32* misc, LaneRead<SWM>& r1, LaneRead<SWM>& r2, const u32 (&rc)[SWM], const u32 (&rcn)[SWM / 2], bool go,
                         u32 key, int l1, int l2, int fr1, int fr2, const u8* q1row, u8* q2row, u32* clist, int lane, int gp, int& geom, int& r2min) {
    int ovl, off, ol, diff;
    decode_overlap(key, l1, l2, ovl, off, ol, diff);
    go = go && ovl && diff != 0;                               // :18-19
    if (ballot(go) == 0ull) return;
    const bool hasN = ((r1.flags | r2.flags) & RS_HAS_N) != 0;
    const int o1 = imax(0, off), o2 = imax(0, -off);           // start1, and where the overlap starts in rc(r2')
    // D = the overlap's mismatches, bit 2 i = position i of the overlap: r1'[o1 + i] against rc(r2')[o2 + i]
    u32 D[SWM];
    {
        u32 xs[SWM], ys[SWM];
#pragma unroll
        for (int w = 0; w < SWM; w++) { xs[w] = r1.s[w]; ys[w] = rc[w]; }
        base_shift_down<SWM>(xs, (u32)o1);
        base_shift_down<SWM>(ys, (u32)o2);
        u32 dn[SWM / 2];
        if (hasN) {
            u32 yn[SWM / 2];
#pragma unroll
            for (int w = 0; w < SWM / 2; w++) { dn[w] = r1.n[w]; yn[w] = rcn[w]; }
            bit_shift_down<SWM / 2>(dn, (u32)o1);
            bit_shift_down<SWM / 2>(yn, (u32)o2);
#pragma unroll
            for (int w = 0; w < SWM / 2; w++) dn[w] ^= yn[w];
        }
#pragma unroll
        for (int w = 0; w < SWM; w++) {
            u32 dd = fold_diff(xs[w] ^ ys[w]);
            if (hasN) dd |= nmask_word<SWM / 2>(dn, w);
            const int rem = ol - 16 * w;
            D[w] = !go ? 0u : (rem >= 16 ? dd : (rem <= 0 ? 0u : (dd & lowmask32(2 * rem))));
        }
    }
#ifdef FQ_PROFILE_ABLATION
    const u32 abl = a.debug_skip;   // profiling build only: 1024 no list entries, 2048 no edits, 4096 the mismatch words only
    if (abl & 4096u) { if (D[0] == 0xFFFFFFFFu) geom = 2; return; }
#else
    const u32 abl = 0;
#endif
    int corrected = 0;
    bool r1c = false, r2c = false;
    auto next_mismatch = [&]() -> int {                        // the smallest mismatch position left (taken out of D), or -1
        int i = -1;
#pragma unroll
        for (int w = SWM - 1; w >= 0; w--)
            if (D[w]) i = 16 * w + ((ffs32(D[w]) - 1) >> 1);
        if (i >= 0) {
#pragma unroll
            for (int w = 0; w < SWM; w++)
                if ((i >> 4) == w) D[w] &= ~(1u << (2 * (i & 15)));
        }
        return i;
    };
    int em_which = -1, em_pos = 0;                             // this round's edit of the lane, for the lists
    u32 em_nb = 0, em_nq = 0;
    auto edit = [&](int i, u32 c1) {                           // one mismatch: :38-66
        const int p1 = o1 + i, k = o2 + i, p2 = l2 - 1 - k;    // :24-25, :38-39
        const u32 b1 = lane_sym_of<SWM>(r1.s, r1.n, p1), brc = lane_sym_of<SWM>(rc, rcn, k);
        const u32 b2 = sym_complement(brc);                    // r2'[p2] itself
        const u32 c2 = (u32)q2row[fr2 + p2] & 0x7Fu;
        if (c1 >= 63u && c2 <= 47u) {                          // GOOD_QUAL = Q30, BAD_QUAL = Q14 (:32-33): use R1
            const u32 nb = sym_complement(b1);
            lane_set_sym<SWM>(r2, p2, nb);
            q2row[fr2 + p2] = (u8)(c1 | (nb == 4u ? 0x80u : 0u));
            lds_add_u32(&misc[MISC_CORRECTION + sym_bin(b2) * 8 + sym_bin(nb)], 1u);
            em_which = 1; em_pos = fr2 + p2; em_nb = nb; em_nq = c1;
            r2min = imin(r2min, p2);
            corrected++;
            r2c = true;
        } else if (c2 >= 63u && c1 <= 47u) {                   // use R2
            const u32 nb = brc;                                // complement(seq2[p2])
            lane_set_sym<SWM>(r1, p1, nb);
            clist[(p1 >> 5) * 64 + lane] |= 1u << (p1 & 31);
            lds_add_u32(&misc[MISC_CORRECTION + sym_bin(b1) * 8 + sym_bin(nb)], 1u);
            em_which = 0; em_pos = fr1 + p1; em_nb = nb; em_nq = c2;
            corrected++;
            r1c = true;
        }
    };
    u32 rec_st = 0, rec0 = 0, rec1 = 0;                         // the lane's first two edits, for the lists (lane_flush_corrections)
    auto record = [&]() {                                      // this round's edit: kept, or (a third one) sent right away
        const u32 n = rec_st & 3u;
        const bool have = em_which >= 0;
        const u32 w1 = (u32)em_pos | (sym_ascii(em_nb) << 16) | (em_nq << 24);
        if (have && n == 0u) { rec0 = w1; rec_st = 1u | ((u32)em_which << 2); }
        else if (have && n == 1u) { rec1 = w1; rec_st = (rec_st & ~3u) | 2u | ((u32)em_which << 3); }
        if (ballot(have && n >= 2u) != 0ull) lane_emit_corrections(a, lane, gp, (have && n >= 2u) ? em_which : -1, em_pos, em_nb, em_nq);
    };
    // read 1's quality at a mismatch comes from memory (its rows have left the stage): the first four positions' bytes are
    // asked for together - one round trip for nearly every pair - the rest one per round
    int pi[4];
    u32 pq[4];
#pragma unroll
    for (int t = 0; t < 4; t++) pi[t] = next_mismatch();
#pragma unroll
    for (int t = 0; t < 4; t++) pq[t] = pi[t] >= 0 ? ((u32)q1row[fr1 + o1 + pi[t]] & 0x7Fu) : 0u;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (ballot(pi[t] >= 0) == 0ull) break;
        em_which = -1;
        if (pi[t] >= 0 && !(abl & 2048u)) edit(pi[t], pq[t]);
        if (!(abl & 1024u)) { if (FQ_CORR_FLUSH) record(); else lane_emit_corrections(a, lane, gp, em_which, em_pos, em_nb, em_nq); }
    }
    for (;;) {
        const int i = next_mismatch();
        if (ballot(i >= 0) == 0ull) break;
        em_which = -1;
        if (i >= 0 && !(abl & 2048u)) edit(i, (u32)q1row[fr1 + o1 + i] & 0x7Fu);
        if (!(abl & 1024u)) { if (FQ_CORR_FLUSH) record(); else lane_emit_corrections(a, lane, gp, em_which, em_pos, em_nb, em_nq); }
    }
    if (FQ_CORR_FLUSH) lane_flush_corrections(a, lane, gp, rec_st, rec0, rec1);
    if (corrected > 0) {                                           // :75-80
        lds_add_u32(&misc[MISC_CORRECTED_READS], (r1c && r2c) ? 2u : 1u);
        if (r1c) r1.flags |= RS_CORRECTED;
        if (r2c) r2.flags |= RS_CORRECTED;
    }
    // where read 1's position p1 was compared with read 2: p2 = geom's high half - p1 (= l2 - 1 - o2 + o1 - p1)
    if (r1c) geom = 1 | ((l2 - 1 - o2 + o1) << 1);
}
// read 1's edited positions inside its final window [0, len) (registers' coordinates; the rows': + fr1): what
// countQualityMetrics adds up changes from the old quality (read 1's row in memory) to the new one (read 2's at the position
// it was compared with, still in the stage: a position where read 1 was edited is one where read 2 was not)
template <int SWM>
FQ_DEV void lane_apply_corrected(const KernelArgs& a, const u32* clist, int lane, int geom, const u8* q1row, const u8* q2row, int fr1, int fr2, int len,
                                 int& tot, int& low) {
    const u32 thr = (u32)a.p.qual_thr;
    const bool any = (geom & 1) != 0;
    const int pivot = geom >> 1;
    u32 m[SWM / 2];
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) {
        const int left = len - 32 * w;
        const u32 v = any ? clist[w * 64 + lane] : 0u;
        m[w] = left >= 32 ? v : (left <= 0 ? 0u : (v & lowmask32(left)));
    }
    for (;;) {
        int p1 = -1;
#pragma unroll
        for (int w = SWM / 2 - 1; w >= 0; w--)
            if (m[w]) p1 = 32 * w + ffs32(m[w]) - 1;
        if (ballot(p1 >= 0) == 0ull) break;
        if (p1 >= 0) {
#pragma unroll
            for (int w = 0; w < SWM / 2; w++)
                if ((p1 >> 5) == w) m[w] &= m[w] - 1u;
            const u32 qo = (u32)q1row[fr1 + p1] & 0x7Fu, qn = (u32)q2row[fr2 + pivot - p1] & 0x7Fu;
            tot += (int)qn - (int)qo;
            low += (qn < thr ? 1 : 0) - (qo < thr ? 1 : 0);
        }
    }
}

// UMI front trim (umiprocessor.cpp:19-49 -> Read::trimFront, read.cpp:69-73), then Filter::trimAndCut with -f / -t: phase_trim of
// the tile kernel.  r.len = the new length (the length behind the UMI for a NULL read), fr / ft as in lane_body.
template <int SWM>
FQ_DEV void lane_front_trim(const KernelArgs& a, LaneRead<SWM>& r, const u8* qrow, int umi, int front, int tail, int& fr, int& ft) {
    int u = 0, l = r.rl0;
    if (umi > 0) {   // Read::trimFront(min(len, umi) + skip): len = min(length() - 1, len)
        int t = imin(l, umi) + a.p.umi_skip;
        t = imin(l - 1, t);
        if (t > 0) { u = t; l -= t; }
    }
    int f2 = 0, l2 = l;
    if (lane_trim_and_cut_front<SWM>(a, r, qrow, u, l, front, tail, f2, l2)) {
        fr = u + f2;
        ft = f2;
        r.len = l2;
    } else {
        fr = u;
        ft = 0;
        r.len = l;
        r.flags |= RS_NULL;
    }
}

FQ_DEV void lane_claim(const KernelArgs& a, int gp, int tl, const u64* h, int B, u32& won) {
    // Duplicate's claim (dup_claim_issue / dup_claim_collect of the tile kernel) for this unit
    const u64 words = a.dup_bits >> 5;
    won = 0;
    for (int i = 0; i < B; i++) {
        const u64 hh = h[i] + a.lut.dup_posum[(size_t)tl * B + i];
        const u64 pos = hh & (a.dup_bits - 1);
        const u32 bit = 1u << (pos & 31);
        const u32 old = g_atomic_or_u32(&a.dup_bitmap[(size_t)i * words + (pos >> 5)], bit);
        if (!(old & bit)) won |= 1u << i;
    }
}

// split_stat_reads() with a front: the swin word holds the END of the kept range, the Stats objects' length sums the kept LENGTH
FQ_DEV void lane_stat_reads_front(const KernelArgs& a, u32* misc, int gp, u32 sw1, u32 sw2, int kept1, int kept2) {
    lds_add_u32(&misc[MISC_STAT_READS + 0], 1u);
    lds_add_u32(&misc[MISC_STAT_LENSUM + 0], sw1 & 0xFFFFu);
    lds_add_u32(&misc[MISC_STAT_READS + 1], (sw1 >> 16) ? 1u : 0u);
    lds_add_u32(&misc[MISC_STAT_LENSUM + 1], (u32)kept1);
    a.swin_out[0][gp] = sw1;
    if (a.p.paired) {
        lds_add_u32(&misc[MISC_STAT_READS + 2], 1u);
        lds_add_u32(&misc[MISC_STAT_LENSUM + 2], sw2 & 0xFFFFu);
        lds_add_u32(&misc[MISC_STAT_READS + 3], (sw2 >> 16) ? 1u : 0u);
        lds_add_u32(&misc[MISC_STAT_LENSUM + 3], (u32)kept2);
        a.swin_out[1][gp] = sw2;
    }
}

// ---------------------------------------------------------------------------
// --merge on the lane plan (DevParams::merge_lane, peprocessor.cpp:518-561)
// ---------------------------------------------------------------------------
// OverlapAnalysis::analyze once more, on the reads as trimmed (peprocessor.cpp:522): lane_body's analysis with the lengths of now
// (a wave collective; the bases behind l1 / l2 in the registers are masked / shifted out as there)
template <int SWM>
FQ_DEV u32 lane_overlap_again(const DevParams& p, const LaneRead<SWM>& r1, const LaneRead<SWM>& r2, int l1, int l2, bool both, const u16* lut_ov) {
    u32 key = OV_KEY_NONE;
    const bool hasN = ((r1.flags | r2.flags) & RS_HAS_N) != 0;
    u32 rc[SWM], rcn[SWM / 2];
#pragma unroll
    for (int w = 0; w < SWM; w++) rc[w] = reverse_groups(r2.s[SWM - 1 - w]) ^ 0x55555555u;
#pragma unroll
    for (int w = 0; w < SWM / 2; w++) rcn[w] = 0;
    const u32 D = (u32)(16 * SWM - l2);
    base_shift_down<SWM>(rc, D);
    if (hasN) {
#pragma unroll
        for (int w = 0; w < SWM / 2; w++) rcn[w] = brev32(r2.n[SWM / 2 - 1 - w]);
        bit_shift_down<SWM / 2>(rcn, D);
#pragma unroll
        for (int w = 0; w < SWM; w++) {
            const u32 sp = nmask_word<SWM / 2>(rcn, w);
            rc[w] &= ~(sp | (sp << 1));
        }
    }
    const u32 nlim = (u32)(-(p.ov_limit_max + 1));
    u32 cm[SWM / 2];
    {
        const int nvalid = both ? l1 - p.overlap_require : 0;
        const int npre = imin(16, imin(p.overlap_require + 1, l2));
        const u32 premask = lowmask32(2 * npre) & 0x55555555u;
        lane_scan<SWM>(r1.s, rc[0], nvalid, premask, nlim, cm);
        for (;;) {
            const int o = key == OV_KEY_NONE ? lane_next_candidate<SWM>(cm) : -1;
            if (ballot(o >= 0) == 0ull) break;
            if (o >= 0) {
                const int diff = lane_verify<SWM>(r1.s, r1.n, rc, rcn, hasN, o, l1, l2, lut_ov);
                if (diff >= 0) key = ov_key(0, o, diff);
            }
        }
    }
    if (ballot(both && key == OV_KEY_NONE) != 0ull) {
        const int nvalid = (both && key == OV_KEY_NONE) ? l2 - p.overlap_require : 0;
        const int npre = imin(16, imin(p.overlap_require + 1, l1));
        const u32 premask = lowmask32(2 * npre) & 0x55555555u;
        lane_scan<SWM>(rc, r1.s[0], nvalid, premask, nlim, cm);
        for (;;) {
            const int o = key == OV_KEY_NONE ? lane_next_candidate<SWM>(cm) : -1;
            if (ballot(o >= 0) == 0ull) break;
            if (o >= 0) {
                const int diff = lane_verify<SWM>(rc, rcn, r1.s, r1.n, hasN, o, l2, l1, lut_ov);
                if (diff >= 0) key = ov_key(1, o, diff);
            }
        }
    }
    return key;
}
// the merged read = r1[0, m1) + reverse complement of r2[0, m2) (OverlapAnalysis::merge, overlapanalysis.cpp:148-179): its 5-mers
// that end on the first four bases of the second part hold bases of both (the Stats kernel counts the 5-mers inside either
// part) - Stats::statRead's index, the earliest base in the high bits (stats.cpp:236, :250)
template <int SWM>
FQ_DEV void lane_merge_junction_kmers(u32* jk, const LaneRead<SWM>& r1, const LaneRead<SWM>& r2, int m1, int m2) {
    u32 sy[8];   // merged[m1 - 4 + i]; 5 = no such base
#pragma unroll
    for (int i = 0; i < 4; i++) sy[i] = m1 - 4 + i >= 0 ? lane_sym_at<SWM>(r1, m1 - 4 + i) : 5u;
#pragma unroll
    for (int i = 0; i < 4; i++) sy[4 + i] = i < m2 ? sym_complement(lane_sym_at<SWM>(r2, m2 - 1 - i)) : 5u;
#pragma unroll
    for (int qj = 0; qj < 4; qj++) {
        u32 idx = 0;
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            ok = ok && sy[qj + t] < 4u;
            idx = (idx << 2) | (sy[qj + t] & 3u);
        }
        if (ok) lds_add_u32(&jk[idx], 1u);
    }
}
// A merged read that passed and whose second part holds a base BaseCorrector edited (possible only when the two overlap analyses
// disagree about the geometry: the edits sit inside the FIRST analysis' overlap, the second part is what lies outside the
// second's): the Stats kernel would count the original bases there, so this lane - it has the corrected read 2 in registers and
// its corrected qualities in the stage - counts the part itself, straight into the POST Stats object of read 1
// (Stats::statRead, stats.cpp:191-266, positions m1 .. m1 + m2 - 1 of the merged read; the 5-mers that lie inside the part).
template <int SWM>
FQ_DEV void lane_merge_tail_slow(const LaneArgs& la, const LaneRead<SWM>& r2, const u8* q2row, int m1, int m2) {
    int64_t* st = la.post1 + la.st_cycle;
    const int64_t CC = la.cycles;
    for (int j = 0; j < m2; j++) {
        const int c = m1 + m2 - 1 - j;
        const u32 sym = sym_complement(lane_sym_at<SWM>(r2, j));
        const u32 q = (u32)q2row[j] & 0x7Fu;
        const int bin = (int)sym_bin(sym);
        if (q >= 63u) g_atomic_add_i64(&st[(0 * 8 + bin) * CC + c], 1);
        if (q >= 53u) g_atomic_add_i64(&st[(1 * 8 + bin) * CC + c], 1);
        g_atomic_add_i64(&st[(2 * 8 + bin) * CC + c], 1);
        g_atomic_add_i64(&st[(3 * 8 + bin) * CC + c], (int64_t)(q - 33u));
        g_atomic_add_i64(&st[32 * CC + c], 1);
        g_atomic_add_i64(&st[33 * CC + c], (int64_t)(q - 33u));
        g_atomic_add_i64(&la.post1[la.st_qual_hist + q], 1);
        if (j + 4 < m2) {   // the 5-mer that ends at merged position c: r2[j + 4] .. r2[j], complemented
            u32 idx = 0;
            bool ok = true;
            for (int t = 0; t < 5; t++) {
                const u32 sy = sym_complement(lane_sym_at<SWM>(r2, j + 4 - t));
                ok = ok && sy < 4u;
                idx = (idx << 2) | (sy & 3u);
            }
            if (ok) g_atomic_add_i64(&la.post1[la.st_kmer + idx], 1);
        }
    }
}

// ---------------------------------------------------------------------------
// the kernel body: persistent wavefronts, a wavefront takes 64 consecutive units at a time
// ---------------------------------------------------------------------------
// EXT: the option family with adapter sequences, polyX trimming or the complexity filter - a second instantiation, so that
// the registers those steps need (+30) are not taken from the kernel of the options that do not use them
// a chunk of the pool (LaneArgs::pool): its number is what the counter returns minus the launch's base.  The base is where the counter
// CAN be at most when the launch starts (the host adds the pool and one ask per wavefront per launch); where fewer wavefronts asked, the
// first askers of the next launch get numbers below its base and ask again.
FQ_DEV int lane_pool_take(const LaneArgs& la, int lane) {
    int gx;
    do {
        gx = 0;
        if (lane == 0) gx = g_atomic_add_i32(la.chunk_ctr, 1) - la.pool_base;
        gx = (int)shfl((u32)gx, 0);
    } while (gx < 0);   // (wave-uniform)
    return gx;
}

template <int SWM, int B, int NPL, bool PAIRED, int EXT>
FQ_DEV void lane_body(const LaneArgs& la, u32* lds) {
    const KernelArgs& a = la.k;
    const LaneLds& ll = la.l;
    const DevParams& p = a.p;
    const int tid = thread_id(), nt = block_threads(), lane = tid & 63;
    {   // one-time per workgroup
        for (int i = tid; i < ll.n_misc; i += nt) lds[ll.misc + i] = 0;
        if (tid == 0) lds[ll.ctr] = 0;
        const int lw = (p.cycles + 2) / 2;
        const u32* g0 = (const u32*)a.lut.ov_limit;
        const u32* g1 = (const u32*)a.lut.lowq_limit;
        const u32* g2 = (const u32*)a.lut.cplx_min;
        for (int i = tid; i < lw; i += nt) {
            lds[ll.lut_ov + i] = g0[i];
            lds[ll.lut_lowq + i] = g1[i];
            lds[ll.lut_cplx + i] = g2[i];
        }
        if (B > 0)
            for (int i = tid; i < 256; i += nt) {  // duplicate.cpp:92-109: A=7 T=222 C=74 G=31 (codes A0 T1 C2 G3)
                u32 v = 0;
                for (int k = 0; k < 4; k++) v |= ((0x1F4ADE07u >> (((i >> (2 * k)) & 3) * 8)) & 0xFFu) << (8 * k);
                lds[ll.val4 + i] = v;
            }
        if (B > 0)
            for (int i = tid; i < ll.n_planes; i += nt) lds[ll.planes + i] = a.lut.dup_planes[i];
        block_sync();
    }
    u32* misc = lds + ll.misc;
    const u16* lut_ov = (const u16*)(lds + ll.lut_ov);
    const u16* lut_lowq = (const u16*)(lds + ll.lut_lowq);
    const u16* lut_cplx = (const u16*)(lds + ll.lut_cplx);
    u32 aw1[LANE_ADAPTER_WORDS], aw2[LANE_ADAPTER_WORDS];   // -a / --adapter_sequence_r2 (uniform)
#pragma unroll
    for (int w = 0; w < LANE_ADAPTER_WORDS; w++) { aw1[w] = p.a1w[w]; aw2[w] = p.a2w[w]; }
    const bool thread0 = (a.batch_flags & 1u) != 0;   // FASTP_GPU_BATCH_STAT_ISIZE
    const bool FR = EXT >= 2 && p.front_lane != 0;         // (uniform) -f / -F / a UMI at the reads' start
    const bool CR = EXT >= 2 && PAIRED && p.corr_lane != 0;   // (uniform) -c
    const bool MG = EXT >= 3 && PAIRED && p.merge_lane != 0;  // (uniform) --merge
#ifdef FQ_PROFILE_ABLATION
    const u32 skip = a.debug_skip;   // profiling build only (FASTP_GPU_DEBUG_SKIP): 1 window predicate, 4 overlap, 8 metrics; results are then meaningless
#else
    const u32 skip = a.debug_skip & 512u;   // (512: a test switch that leaves the results as they are, see `slow` below)
#endif
    const int win = (skip & 1u) ? 0 : (p.cut_right ? p.wR : (p.cut_tail ? p.wT : (p.front_per_read ? p.wF : 0)));
    const int thr = p.cut_right ? p.thrR : (p.cut_tail ? p.thrT : p.thrF);
    const u32 thr4 = (u32)p.qual_thr * 0x01010101u;
    const int chunks = (a.n + 63) >> 6;
    const int wpb = nt >> 6;
    const int nstatic = grid_blocks() * wpb;
    // Chunks beyond a wavefront's first come from the counter, `grab` at a time while the end of the launch is far (one returning atomic
    // per chunk is 62 500 of them on ONE word per 4 Mi pairs: a word takes ~88 per microsecond - MI355X_MICROARCH.md, "dequeue" - which
    // bounds the kernel at 0.7 ms whatever else it does), one at a time over the last chunks so that the wavefronts still end together
    int nx = 0, gsz = 1;
    const bool LC = la.local_ctr != 0;                          // (uniform)
    const int grab = (la.chunk_ctr && !LC) ? imax(1, la.grab) : 1;
    // the workgroup's share of the chunks (local_ctr): the first (chunks % workgroups) workgroups take one more
    const int pool = (LC && la.chunk_ctr) ? imin(la.pool, chunks) : 0;   // (uniform) the chunks behind the workgroups' shares
    const int shared = chunks - pool;
    const int pg = imax(1, la.pool_grab), pgroups = (pool + pg - 1) / pg;
    const int sq = shared / grid_blocks(), sr = shared - sq * grid_blocks();
    const int share_lo = LC ? block_id() * sq + imin(block_id(), sr) : 0;
    const int share_hi = LC ? share_lo + sq + (block_id() < sr ? 1 : 0) : chunks;
    int chunk_end = (LC ? share_lo + (tid >> 6) : block_id() * wpb + (tid >> 6)) + 1;   // (the range in hand: [chunk, chunk_end))
    int chunk0 = chunk_end - 1;
    bool in_pool = false;                                       // (uniform)
    if (pool && chunk0 >= share_hi) {                           // a share smaller than the workgroup: straight to the pool
        const int gx = lane_pool_take(la, lane);
        in_pool = true;
        chunk0 = gx < pgroups ? shared + gx * pg : chunks;
        chunk_end = imin(chunk0 + pg, chunks);
    }
    for (int chunk = chunk0; chunk < (in_pool ? chunks : share_hi);) {       // wave-uniform
        const bool last_in_hand = chunk + 1 == chunk_end;
        if (LC) {
            if (in_pool) {   // (the pool's next range is asked for here as well, when the one in hand ends with this chunk)
                if (last_in_hand && lane == 0) nx = g_atomic_add_i32(la.chunk_ctr, 1) - la.pool_base;
            } else if (lane == 0) nx = (int)lds_add_ret_u32(&lds[ll.ctr], 1u);   // the workgroup's next chunk, looked at at the loop's end
        } else if (la.chunk_ctr && last_in_hand) {
            gsz = (chunk + 4 * grab * nstatic < chunks) ? grab : 1;   // (uniform)
            if (lane == 0) nx = g_atomic_add_i32(la.chunk_ctr, gsz);   // the next range's first chunk, looked at at the loop's end
        }
        const int gp = chunk * 64 + lane;
        const bool valid = gp < a.n;
        const int rows = imin(64, a.n - chunk * 64);
        u32* stage = lds + ll.stage + (tid >> 6) * ll.stage_dwords;
        u32* part = lds + ll.part + (tid >> 6) * ll.part_dwords;
        u32* clist = lds + ll.clist + (tid >> 6) * ll.clist_dwords;
        int geom = 0;  // -c: bit 0 = read 1 has edited positions (their mask: clist), the rest: lane_correct
        int r2min = 0x7FFF;   // -c: the smallest edited position of read 2
        int m1 = 0, m2 = 0;   // --merge: the two parts of the merged read (mov: the second analysis found an overlap)
        bool mov = false;
        if (CR) {
#pragma unroll
            for (int w = 0; w < SWM / 2; w++) clist[w * 64 + lane] = 0;
        }
        const int g = valid ? gp : 0;
        // a unit of the text kernel (which runs beside this kernel): an empty unit here, counted and not written
        const bool xs = a.xskip != nullptr && valid && a.xskip[g] != 0;
        LaneRead<SWM> r1, r2;
        // each read is trimmed (Filter::trimAndCut) as soon as it is loaded: its window predicate is dead after that
        // DevParams::front_lane (EXT): fr = the read's front in the row (UMI + -f), ft = trimAndCut's part of it (frontTrimmed)
        int fr1 = 0, ft1 = 0, fr2 = 0, ft2 = 0;
        if (PAIRED && (la.prefetch & 1)) {   // (uniform) read 2's rows on their way to L2 while read 1 is staged and swept
            const int step = 128 >> ((la.prefetch >> 2) & 3);   // (A/B: bits 2-3 = a dword every 128 / 64 / 32 bytes)
            lane_prefetch_lines(lds + ll.sink, a.seq[1] + (size_t)(chunk * 64) * p.sw_g, rows * p.sw_g * 4, lane, step);
            lane_prefetch_lines(lds + ll.sink, a.qual[1] + (size_t)(chunk * 64) * p.qw_g, rows * p.qw_g * 4, lane, step);
        }
        lane_load_read<SWM, PAIRED>(a, stage, part, a.seq[0], a.qual[0], a.len[0], chunk * 64, rows, lane, valid, win, thr, thr4, r1);
        if (FR) {
            if (valid) lane_front_trim<SWM>(a, r1, (const u8*)(stage + lane * p.qw_g), p.umi_len1, p.trim_front1, p.trim_tail1, fr1, ft1);
        } else if (valid && !lane_trim_and_cut<SWM>(a, r1, (const u8*)(stage + lane * p.qw_g), p.trim_tail1, r1.len)) r1.flags |= RS_NULL;
        sched_fence();
        if ((la.prefetch & 2) && (la.chunk_ctr || LC)) {   // (uniform) the next chunk's read 1 (its number came back long ago)
            const int nxc = LC ? share_lo + wpb + (int)uniform((u32)nx) : last_in_hand ? nstatic + (int)uniform((u32)nx) : chunk + 1;
            if (nxc < share_hi) {
                const int nrows = imin(64, a.n - nxc * 64);
                const int step = 128 >> ((la.prefetch >> 2) & 3);
                lane_prefetch_lines(lds + ll.sink, a.seq[0] + (size_t)(nxc * 64) * p.sw_g, nrows * p.sw_g * 4, lane, step);
                lane_prefetch_lines(lds + ll.sink, a.qual[0] + (size_t)(nxc * 64) * p.qw_g, nrows * p.qw_g * 4, lane, step);
            }
        }
        // Duplicate::seq2intvector of read 1 in front of read 2's sweep (B > 0, the asynchronous stage): what it needs of read 1
        // is final, and it covers the round trip of read 2's quality rows
        const bool GL = PAIRED && B > 0 && la.glds != 0;   // (uniform)
        u64 hs_early[B > 0 ? B : 1];
        if (PAIRED) {
            if (GL) {
                if constexpr (B > 0) {
                    lane_load_bases<SWM>(a, stage, a.seq[1], a.len[1], chunk * 64, rows, lane, valid, r2);
                    lane_stage_rows_async(stage, a.qual[1] + (size_t)(chunk * 64) * p.qw_g, rows, p.qw_g, lane);
                    lane_hash<SWM, B, NPL>(a, lds, ll, r1, 0, hs_early);
                    lane_stage_rows_done(stage, a.qual[1] + (size_t)(chunk * 64) * p.qw_g, rows, p.qw_g, lane);
                    lane_sweep_quality<SWM, false>(a, stage, part, lane, win, thr, thr4, r2);
                }
            } else {
                lane_load_read<SWM, false>(a, stage, part, a.seq[1], a.qual[1], a.len[1], chunk * 64, rows, lane, valid, win, thr, thr4, r2);
            }
            if (FR) {
                if (valid) lane_front_trim<SWM>(a, r2, (const u8*)(stage + lane * p.qw_g), p.umi_len2, p.trim_front2, p.trim_tail2, fr2, ft2);
            } else if (valid && !lane_trim_and_cut<SWM>(a, r2, (const u8*)(stage + lane * p.qw_g), p.trim_tail2, r2.len)) r2.flags |= RS_NULL;
            sched_fence();
        }
        if (a.dupflag && valid && a.dupflag[g]) {   // --dedup: Duplicate::checkPair/checkRead already ran for this batch
            r1.flags |= RS_DUP;
            if (PAIRED) r2.flags |= RS_DUP;
        }
        // Duplicate's claim: fired now, looked at when the record is written
        u32 won = 0;
        const bool claim = B > 0 && a.claim_won != nullptr;
        if constexpr (B > 0) {
            u64 hs[B], h2[B];
            if (GL) {
#pragma unroll
                for (int i = 0; i < B; i++) hs[i] = hs_early[i];
            } else {
                lane_hash<SWM, B, NPL>(a, lds, ll, r1, 0, hs);
            }
            if (PAIRED) {
                lane_hash<SWM, B, NPL>(a, lds, ll, r2, r1.rl0, h2);
#pragma unroll
                for (int i = 0; i < B; i++) hs[i] += h2[i];
            }
            if (valid && !xs) {
                if (a.dup_pos)
                    for (int i = 0; i < B; i++) a.dup_pos[(size_t)g * B + i] = hs[i];
                if (claim) lane_claim(a, g, r1.rl0 + (PAIRED ? r2.rl0 : 0), hs, B, won);
            }
        }
        if (FR) {   // from here on a read starts at base 0 of its registers (Duplicate hashed the original read, duplicate.cpp:111-148)
            base_shift_down<SWM>(r1.s, (u32)fr1);
            bit_shift_down<SWM / 2>(r1.n, (u32)fr1);
            if (PAIRED) {
                base_shift_down<SWM>(r2.s, (u32)fr2);
                bit_shift_down<SWM / 2>(r2.n, (u32)fr2);
            }
        }
        // ---- PolyX::trimPolyG ----
        const bool a1 = valid && !(r1.flags & RS_NULL), a2 = PAIRED ? (valid && !(r2.flags & RS_NULL)) : a1;
        const bool both = a1 && a2;
        if (p.poly_g) {   // (uniform) a pair's mates are trimmed when both survived trimAndCut (peprocessor.cpp:428-431)
            r1.len = lane_trim_poly_g<SWM>(r1, both, r1.len, p.poly_g_min);
            if (PAIRED) r2.len = lane_trim_poly_g<SWM>(r2, both, r2.len, p.poly_g_min);
        }
        u32 apos1 = 0, alen1 = 0, apos2 = 0, alen2 = 0;
        bool dimer = false;
        if (PAIRED) {
            // ---- OverlapAnalysis::analyze ----
            u32 key = OV_KEY_NONE;
            const bool want_ov = (p.need_overlap || thread0) && !(skip & 4u);   // peprocessor.cpp:438
            if (want_ov && ballot(both) != 0ull) {
                const int l1 = r1.len, l2 = r2.len;
                const bool hasN = ((r1.flags | r2.flags) & RS_HAS_N) != 0;
                // rc(r2') in registers: reverse the word order and the groups, complement, move the frame's unused head out
                u32 rc[SWM], rcn[SWM / 2];
#pragma unroll
                for (int w = 0; w < SWM; w++) rc[w] = reverse_groups(r2.s[SWM - 1 - w]) ^ 0x55555555u;
#pragma unroll
                for (int w = 0; w < SWM / 2; w++) rcn[w] = 0;
                const u32 D = (u32)(16 * SWM - l2);
                base_shift_down<SWM>(rc, D);
                if (hasN) {
#pragma unroll
                    for (int w = 0; w < SWM / 2; w++) rcn[w] = brev32(r2.n[SWM / 2 - 1 - w]);
                    bit_shift_down<SWM / 2>(rcn, D);
#pragma unroll
                    for (int w = 0; w < SWM; w++) {   // N stays code 0 on both strands
                        const u32 sp = nmask_word<SWM / 2>(rcn, w);
                        rc[w] &= ~(sp | (sp << 1));
                    }
                }
                // bases past l2 of rc are zeros by construction; bases of r1 past l1 are masked by every consumer
                const u32 nlim = (u32)(-(p.ov_limit_max + 1));
                u32 cm[SWM / 2];
                // forward: X = r1', Y = rc(r2')
                {
                    const int nvalid = both ? l1 - p.overlap_require : 0;
                    const int npre = imin(16, imin(p.overlap_require + 1, l2));
                    const u32 premask = lowmask32(2 * npre) & 0x55555555u;
                    lane_scan<SWM>(r1.s, rc[0], nvalid, premask, nlim, cm);
                    for (;;) {
                        const int o = key == OV_KEY_NONE ? lane_next_candidate<SWM>(cm) : -1;
                        if (ballot(o >= 0) == 0ull) break;
                        if (o >= 0) {
                            const int diff = lane_verify<SWM>(r1.s, r1.n, rc, rcn, hasN, o, l1, l2, lut_ov);
                            if (diff >= 0) key = ov_key(0, o, diff);
                        }
                    }
                }
                // reverse: X = rc(r2'), Y = r1'
                if (ballot(both && key == OV_KEY_NONE) != 0ull) {
                    const int nvalid = (both && key == OV_KEY_NONE) ? l2 - p.overlap_require : 0;
                    const int npre = imin(16, imin(p.overlap_require + 1, l1));
                    const u32 premask = lowmask32(2 * npre) & 0x55555555u;
                    lane_scan<SWM>(rc, r1.s[0], nvalid, premask, nlim, cm);
                    for (;;) {
                        const int o = key == OV_KEY_NONE ? lane_next_candidate<SWM>(cm) : -1;
                        if (ballot(o >= 0) == 0ull) break;
                        if (o >= 0) {
                            const int diff = lane_verify<SWM>(rc, rcn, r1.s, r1.n, hasN, o, l2, l1, lut_ov);
                            if (diff >= 0) key = ov_key(1, o, diff);
                        }
                    }
                }
                // ---- BaseCorrector::correctByOverlapAnalysis (peprocessor.cpp:453-456; no gap on this plan) ----
                if (CR && p.need_overlap && !(skip & 32u))   // (32: profiling build only - the correction rounds left out)
                    lane_correct<SWM>(a, misc, r1, r2, rc, rcn, both, key, l1, l2, fr1, fr2, (const u8*)(a.qual[0] + (size_t)g * p.qw_g),
                                      (u8*)(stage + lane * p.qw_g), clist, lane, g, geom, r2min);
            }
            // ---- peprocessor.cpp:443-516: insert size, adapter trimming by overlap, max_len ----
            int cur1 = r1.len, cur2 = r2.len;
            int ovl, ov_off, ov_len, ov_diff;
            decode_overlap(key, cur1, cur2, ovl, ov_off, ov_len, ov_diff);
            bool isize_done = false;
            if (both && thread0) {   // statInsertSize (peprocessor.cpp:710-723)
                int isize = p.isize_max;
                if (ovl) isize = (ov_off > 0 ? cur1 + cur2 - ov_len : ov_len) + ft1 + ft2;
                if (isize > p.isize_max) isize = p.isize_max;
                if (isize >= 0) lds_add_u32(&misc[MISC_ISIZE + isize], 1u);
                isize_done = true;
            }
            {
                const bool adapt = both && p.need_overlap && p.adapter_enabled;   // per lane; the sequence scans below are wave collectives
                bool trimmed = false;
                if (adapt && ovl && ov_off < 0) {   // trimByOverlapAnalysis adaptertrimmer.cpp:17-46
                    const int len1 = imin(cur1, ov_len + ft2), len2 = imin(cur2, ov_len + ft1);
                    apos1 = (u32)len1; alen1 = (u32)(cur1 - len1);
                    apos2 = (u32)len2; alen2 = (u32)(cur2 - len2);
                    lds_add_u32(&misc[MISC_ADAPTER_BASES], (u32)((cur1 - len1) + (cur2 - len2)));
                    cur1 = len1;
                    cur2 = len2;
                    trimmed = true;
                    r1.flags |= RS_ADAPTER_OV;
                    r2.flags |= RS_ADAPTER_OV;
                }
                bool t1 = trimmed, t2 = trimmed;
                if (EXT && (p.has_a1 | p.has_a2)) {   // (uniform) peprocessor.cpp:460-466: the pairs the overlap did not trim
                    const bool go = adapt && !trimmed;
                    int pos = 0;
                    if (p.has_a1 && ballot(go) != 0ull) {
                        const bool hit = lane_trim_by_sequence<SWM>(r1, go ? cur1 : 0, aw1, p.alen1, 4, pos);
                        if (go && hit) { lane_apply_adapter(misc, pos, p.alen1, cur1, apos1, alen1); t1 = true; }
                    }
                    if (p.has_a2 && ballot(go) != 0ull) {
                        const bool hit = lane_trim_by_sequence<SWM>(r2, go ? cur2 : 0, aw2, p.alen2, 4, pos);
                        if (go && hit) { lane_apply_adapter(misc, pos, p.alen2, cur2, apos2, alen2); t2 = true; }
                    }
                }
                if (EXT && p.n_fasta) {   // (uniform) :467-470: every pair, whatever trimmed it before
                    t1 |= lane_fasta_trims<SWM>(a, misc, r1, adapt, cur1, 2u * (u32)(a.first + gp));
                    t2 |= lane_fasta_trims<SWM>(a, misc, r2, adapt, cur2, 2u * (u32)(a.first + gp) + 1u);
                }
                if (t1) { lds_add_u32(&misc[MISC_ADAPTER_READS], 1u); r1.flags |= RS_ADAPTER; }   // :472-475
                if (t2) { lds_add_u32(&misc[MISC_ADAPTER_READS], 1u); r2.flags |= RS_ADAPTER; }
                if ((t1 || t2) && cur1 <= p.dimer_max_len && cur2 <= p.dimer_max_len) dimer = true;   // :480-484
            }
            if (EXT && p.poly_x && ballot(both) != 0ull) {   // :506-509
                int poly, cut;
                cur1 = lane_trim_poly_x<SWM>(r1, both, cur1, p.poly_x_min, poly, cut);
                if (both && poly >= 0) {   // addPolyXTrimmed filterresult.cpp:186-189
                    lds_add_u32(&misc[MISC_POLYX_READS + poly], 1u);
                    lds_add_u32(&misc[MISC_POLYX_BASES + poly], (u32)cut);
                    r1.flags |= RS_POLYX;
                }
                cur2 = lane_trim_poly_x<SWM>(r2, both, cur2, p.poly_x_min, poly, cut);
                if (both && poly >= 0) {
                    lds_add_u32(&misc[MISC_POLYX_READS + poly], 1u);
                    lds_add_u32(&misc[MISC_POLYX_BASES + poly], (u32)cut);
                    r2.flags |= RS_POLYX;
                }
            }
            if (both) {   // :511-516
                if (p.max_len1 > 0 && p.max_len1 < cur1) cur1 = p.max_len1;
                if (p.max_len2 > 0 && p.max_len2 < cur2) cur2 = p.max_len2;
            }
            r1.len = cur1;
            r2.len = cur2;
            if (MG && !(skip & 64u) && ballot(both) != 0ull) {   // peprocessor.cpp:518-527: the analysis of the reads as they are now decides about merging
                                                                  // (64: profiling build only - the second analysis left out)
                const u32 key2 = lane_overlap_again<SWM>(p, r1, r2, cur1, cur2, both, lut_ov);
                if (both) {
                    decode_overlap(key2, cur1, cur2, ovl, ov_off, ov_len, ov_diff);
                    if (ovl) {   // OverlapAnalysis::merge, overlapanalysis.cpp:148-179 (substr clamps)
                        m1 = imin(ov_len + imax(0, ov_off), cur1);
                        m2 = imax(0, imin(ov_off > 0 ? cur2 - ov_len : 0, cur2 - ov_len));
                        mov = true;
                    }
                }
            }
            if (valid && !xs) write_pair_result(a, g, ovl, ov_off, ov_len, ov_diff, isize_done);
        } else {
            if (EXT && p.adapter_enabled && (p.has_a1 || p.n_fasta) && ballot(a1) != 0ull) {   // seprocessor.cpp:244-261
                int pos = 0, cur = r1.len;
                bool trimmed = false;
                if (p.has_a1) {
                    const bool hit = lane_trim_by_sequence<SWM>(r1, a1 ? cur : 0, aw1, p.alen1, 4, pos);
                    if (a1 && hit) {
                        lane_apply_adapter(misc, pos, p.alen1, cur, apos1, alen1);
                        trimmed = true;
                    }
                }
                if (p.n_fasta) trimmed |= lane_fasta_trims<SWM>(a, misc, r1, a1, cur, (u32)(a.first + gp));   // :249-251
                if (a1 && trimmed) {
                    r1.len = cur;
                    lds_add_u32(&misc[MISC_ADAPTER_READS], 1u);
                    r1.flags |= RS_ADAPTER;
                    if (cur <= p.dimer_max_len) dimer = true;
                }
            }
            if (EXT && p.poly_x && ballot(a1) != 0ull) {   // :263-266
                int poly, cut;
                r1.len = lane_trim_poly_x<SWM>(r1, a1, r1.len, p.poly_x_min, poly, cut);
                if (a1 && poly >= 0) {
                    lds_add_u32(&misc[MISC_POLYX_READS + poly], 1u);
                    lds_add_u32(&misc[MISC_POLYX_BASES + poly], (u32)cut);
                    r1.flags |= RS_POLYX;
                }
            }
            if (a1 && p.max_len1 > 0 && p.max_len1 < r1.len) r1.len = p.max_len1;   // seprocessor.cpp:268-271
        }
        // ---- Filter::passFilter (filter.cpp:15-57), routing, records ----
        int tot1 = 0, low1 = 0, nb1 = 0, tot2 = 0, low2 = 0, nb2 = 0;
        // (--merge: a pair that merges is filtered as its merged read - the two parts' metrics add up, peprocessor.cpp:524-526)
        const int ml1 = (MG && mov) ? m1 : r1.len, ml2 = (MG && mov) ? m2 : (PAIRED ? r2.len : 0);
#if FQ_LANE_METRICS == 0
        if (!(skip & 8u)) {
            lane_metrics_staged<SWM>(a, stage, a.qual[0], chunk * 64, rows, lane, r1.len, tot1, low1, nb1);
            if (PAIRED) lane_metrics_staged<SWM>(a, stage, a.qual[1], chunk * 64, rows, lane, r2.len, tot2, low2, nb2);
        }
#else
        if (!(skip & 8u)) {
            const u32* row = stage + lane * p.qw_g;   // the quality row of the read that was staged last
            if (FR) {   // (uniform) the windows start at the reads' fronts
                if (PAIRED) {
                    LaneCutWord c1e, c1f;
                    lane_cut_fetch(a, a.qual[0], g, a1, fr1 + r1.len, c1e);
                    lane_cut_fetch(a, a.qual[0], g, a1, fr1, c1f);
                    lane_metrics_stage_front<SWM>(a, row, r2, fr2, r2.len, tot2, low2, nb2);
                    lane_metrics_part_front<SWM>(a, r1, part, lane, c1e, c1f, fr1, r1.len, tot1, low1, nb1);
                } else {
                    lane_metrics_stage_front<SWM>(a, row, r1, fr1, r1.len, tot1, low1, nb1);
                }
            } else if (PAIRED) {
                LaneCutWord c1;
                lane_cut_fetch(a, a.qual[0], g, a1, ml1, c1);
                lane_metrics_stage<SWM>(a, row, r2, ml2, tot2, low2, nb2);      // (read 1's cut word is on its way meanwhile)
                lane_metrics_part<SWM>(a, r1, part, lane, c1, ml1, tot1, low1, nb1);
            } else {
                lane_metrics_stage<SWM>(a, row, r1, r1.len, tot1, low1, nb1);
            }
        }
#endif
        if (CR && !(skip & 8u) && ballot((geom & 1) != 0) != 0ull)   // read 1's edited positions inside its final window
            lane_apply_corrected<SWM>(a, clist, lane, geom, (const u8*)(a.qual[0] + (size_t)g * p.qw_g), (const u8*)(stage + lane * p.qw_g), fr1, fr2, ml1,
                                      tot1, low1);
        int dif1 = 0, dif2 = 0;
        if (EXT && p.complexity_filter) {   // (uniform) filter.cpp:51-54: countAdjacentDiffs of the final window
            dif1 = lane_adjacent_diffs<SWM>(r1, ml1);
            if (PAIRED) dif2 = lane_adjacent_diffs<SWM>(r2, ml2);
        }
        bool merged_out = false;   // --merge: this pair's merged read passed the filter
        if (MG && valid) {   // peprocessor.cpp:518-591 in merge mode
            const bool dedup_out = p.dedup && (r1.flags & RS_DUP);
            int code1, code2;
            bool post1 = false, post2 = false;   // given to the POST Stats object (of read 1: all of them, :529, :548, :554)
            if (both && mov) {                   // :523-535 (neither the dimer evidence nor --dedup's decision is looked at here)
                int dif = dif1 + dif2;
                if (EXT && p.complexity_filter && m1 > 0 && m2 > 0 && lane_sym_at<SWM>(r1, m1 - 1) != sym_complement(lane_sym_at<SWM>(r2, m2 - 1))) dif++;
                const int ml = m1 + m2;
                code1 = code2 = filter_code_pre(p, ml, tot1 + tot2, low1 + low2, nb1 + nb2, dif, (int)lut_lowq[ml], (int)lut_cplx[ml]);
                lds_add_u32(&misc[MISC_FILTER + code1], 2u);
                if (code1 == 0) {
                    r1.flags |= RS_MERGED;
                    r2.flags |= RS_MERGED;
                    lds_add_u32(&misc[MISC_MERGED], 1u);
                    post1 = post2 = merged_out = true;
                }
            } else {
                code1 = a1 ? filter_code_pre(p, r1.len, tot1, low1, nb1, dif1, (int)lut_lowq[r1.len], (int)lut_cplx[r1.len]) : 16;
                code2 = a2 ? filter_code_pre(p, r2.len, tot2, low2, nb2, dif2, (int)lut_lowq[r2.len], (int)lut_cplx[r2.len]) : 16;
                if (dimer) { code1 = 28; code2 = 28; }
                if (both && p.merge_include_unmerged) {   // :536-559
                    lds_add_u32(&misc[MISC_FILTER + code1], 1u);
                    lds_add_u32(&misc[MISC_FILTER + code2], 1u);
                    post1 = code1 == 0 && !dedup_out;
                    post2 = code2 == 0 && !dedup_out;
                } else {                                  // :563-591: written to out1 / out2, no Stats object sees them (:588)
                    lds_add_u32(&misc[MISC_FILTER + imax(code1, code2)], 2u);
                }
            }
            // what the Stats kernel reads: read 1's kept range is the first part / the whole read; read 2's the second part (bit 15:
            // reverse-complemented behind the first; bit 14: counted here already, lane_merge_tail_slow) / the whole read
            const bool slow = merged_out && ((CR && r2min < m2) || (skip & 512u));   // (512: tests - every merged read's second part this way)
            const u32 k1 = (u32)(merged_out ? m1 : r1.len), k2 = (u32)(merged_out ? m2 : r2.len);
            const u32 sw1 = (u32)r1.rl0 | (post1 ? k1 << 16 : 0u);
            const u32 sw2 = (u32)r2.rl0 | (post2 ? (k2 | (merged_out ? 0x8000u : 0u) | (slow ? 0x4000u : 0u)) << 16 : 0u);
            lds_add_u32(&misc[MISC_STAT_READS + 0], 1u);
            lds_add_u32(&misc[MISC_STAT_LENSUM + 0], (u32)r1.rl0);
            lds_add_u32(&misc[MISC_STAT_READS + 2], 1u);
            lds_add_u32(&misc[MISC_STAT_LENSUM + 2], (u32)r2.rl0);
            lds_add_u32(&misc[MISC_STAT_READS + 1], merged_out ? 1u : (post1 ? 1u : 0u) + (post2 ? 1u : 0u));   // a merged read is ONE read
            lds_add_u32(&misc[MISC_STAT_LENSUM + 1], (post1 ? k1 : 0u) + (post2 ? k2 : 0u));
            a.swin_out[0][g] = sw1;
            a.swin_out[1][g] = sw2;
            if (!xs) {
                // reserved: the bases of the mate in the merged read when the second analysis found an overlap (write_read_result)
                u32* o1 = a.res[0] + (size_t)g * 3;
                o1[0] = (u32)r1.len << 16;
                o1[1] = ((u32)code1 & 0xFFu) | ((r1.flags & 0xFFu) << 8) | (apos1 << 16);
                o1[2] = (alen1 & 0xFFFFu) | (mov ? (u32)m1 << 16 : 0u);
                u32* o2 = a.res[1] + (size_t)g * 3;
                o2[0] = (u32)r2.len << 16;
                o2[1] = ((u32)code2 & 0xFFu) | ((r2.flags & 0xFFu) << 8) | (apos2 << 16);
                o2[2] = (alen2 & 0xFFFFu) | (mov ? (u32)m2 << 16 : 0u);
                if (claim) a.claim_won[g] = (u8)won;
            }
            if (slow) lane_merge_tail_slow<SWM>(la, r2, (const u8*)(stage + lane * p.qw_g), m1, m2);
        }
        if (MG) {
            if (ballot(merged_out) != 0ull && merged_out) lane_merge_junction_kmers<SWM>(lds + ll.jkmer, r1, r2, m1, m2);
        } else if (valid) {
            int code1 = a1 ? filter_code_pre(p, r1.len, tot1, low1, nb1, dif1, (int)lut_lowq[r1.len], (int)lut_cplx[r1.len]) : 16;
            int code2 = 0;
            if (PAIRED) {
                code2 = a2 ? filter_code_pre(p, r2.len, tot2, low2, nb2, dif2, (int)lut_lowq[r2.len], (int)lut_cplx[r2.len]) : 16;
                if (dimer) { code1 = 28; code2 = 28; }                          // :568-571
                lds_add_u32(&misc[MISC_FILTER + imax(code1, code2)], 2u);       // addFilterResult(max, 2) :573
            } else {
                if (dimer) code1 = 28;                                          // seprocessor.cpp:273-277
                lds_add_u32(&misc[MISC_FILTER + code1], 1u);                    // seprocessor.cpp:278
            }
            const bool dedup_out = p.dedup && (r1.flags & RS_DUP);
            const bool post = !dedup_out && a1 && a2 && code1 == 0 && code2 == 0;   // written to out1 / out2 (:577-591)
            // what the Stats kernel reads: original length | END of the kept range << 16 (the range starts at the mate's front)
            const u32 sw1 = (u32)r1.rl0 | (post ? (u32)(fr1 + r1.len) << 16 : 0u);
            const u32 sw2 = PAIRED ? ((u32)r2.rl0 | (post ? (u32)(fr2 + r2.len) << 16 : 0u)) : 0u;
            if (FR) lane_stat_reads_front(a, misc, g, sw1, sw2, post ? r1.len : 0, post ? r2.len : 0);
            else split_stat_reads(a, misc, g, sw1, sw2);
            if (!xs) {
                u32* o1 = a.res[0] + (size_t)g * 3;
                o1[0] = ((u32)fr1 & 0xFFFFu) | ((u32)r1.len << 16);
                o1[1] = ((u32)code1 & 0xFFu) | ((r1.flags & 0xFFu) << 8) | (apos1 << 16);
                o1[2] = alen1 & 0xFFFFu;
                if (PAIRED) {
                    u32* o2 = a.res[1] + (size_t)g * 3;
                    o2[0] = ((u32)fr2 & 0xFFFFu) | ((u32)r2.len << 16);
                    o2[1] = ((u32)code2 & 0xFFu) | ((r2.flags & 0xFFu) << 8) | (apos2 << 16);
                    o2[2] = alen2 & 0xFFFFu;
                }
                if (claim) a.claim_won[g] = (u8)won;
            }
        }
        // the next chunk: the range in hand, then the counter's (or the static stride's) next range
        if (LC) {
            if (in_pool && !last_in_hand) chunk++;
            else {
                if (!in_pool) { chunk = share_lo + wpb + (int)shfl((u32)nx, 0); chunk_end = chunk + 1; }
                if (pool && (in_pool || chunk >= share_hi)) {   // the share is used up: the pool's next chunks (a global atomic, few of them)
                    int gx = in_pool ? (int)shfl((u32)nx, 0) : -1;   // (asked for at the loop's head; the first one here)
                    if (gx < 0) gx = lane_pool_take(la, lane);
                    in_pool = true;
                    chunk = gx < pgroups ? shared + gx * pg : chunks;
                    chunk_end = imin(chunk + pg, chunks);
                }
            }
        }
        else if (!last_in_hand) chunk++;
        else if (la.chunk_ctr) { chunk = nstatic + (int)shfl((u32)nx, 0); chunk_end = chunk + gsz; }
        else { chunk += nstatic; chunk_end = chunk + 1; }
    }
    block_sync();
    u32* slab = a.slabs + (size_t)block_id() * a.slab_dwords;
    for (int i = tid; i < a.slab_dwords; i += nt) slab[i] = lds[ll.misc + i];
}

}  // namespace fq
