// fq_host.cpp - see fq_host.h.  Host code only (no HIP runtime, no kernels).
#include "fq_host.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace fq {

u32 magic_for(u32 d) { return (u32)((0x100000000ull + d - 1) / d); }

void dup_geometry(int level, u64& bytes, int& num) {  // Duplicate::Duplicate duplicate.cpp:13-47
    bytes = 1ull << 29;
    num = 2;
    switch (level) {
        case 2: bytes *= 2; break;
        case 3: bytes *= 2; num *= 2; break;
        case 4: bytes *= 4; num *= 2; break;
        case 5: bytes *= 8; num *= 2; break;
        case 6: bytes *= 8; num *= 4; break;
        default: break;
    }
}

void dup_primes(int bufnum, std::vector<u32>& out) {  // Duplicate::initPrimeArrays duplicate.cpp:66-84
    out.clear();
    u64 number = 10000;
    while ((int)out.size() < bufnum * 512) {
        number++;
        bool is_prime = true;
        for (u64 i = 2; (double)i <= sqrt((double)number); i++) {
            if (number % i == 0) { is_prime = false; break; }
        }
        if (is_prime) {
            out.push_back((u32)number);
            number += 10000;
        }
    }
}

static int base_code(char c) {
    switch (c) {
        case 'A': return CODE_A;
        case 'T': return CODE_T;
        case 'C': return CODE_C;
        case 'G': return CODE_G;
        default: return -1;
    }
}

static int pack_adapter(const char* s, u32* words, int& alen, std::string& err) {
    alen = 0;
    for (int i = 0; i < MAX_ADAPTER_WORDS; i++) words[i] = 0;
    if (!s || !s[0]) return FASTP_GPU_OK;
    const int n = (int)strlen(s);
    if (n > FASTP_GPU_MAX_ADAPTER_LEN) {
        err = "adapter sequence longer than FASTP_GPU_MAX_ADAPTER_LEN";
        return FASTP_GPU_E_UNSUPPORTED;
    }
    for (int i = 0; i < n; i++) {
        const int c = base_code(s[i]);
        if (c < 0) {  // options.cpp:375-381: only A/T/C/G are legal
            err = "adapter sequence may only contain A, T, C, G";
            return FASTP_GPU_E_INVALID;
        }
        words[i >> 4] |= (u32)c << ((i & 15) * 2);
    }
    alen = n;
    return FASTP_GPU_OK;
}

static char num2qual(int num) {  // util.h:260-268
    if (num > 127 - 33) num = 127 - 33;
    if (num < 0) num = 0;
    return (char)(num + 33);
}

int build_dev_params(const fastp_gpu_params& in, DevParams& p, HostLuts& luts, std::string& err) {
    memset(&p, 0, sizeof(p));
    if (in.abi_version != FASTP_GPU_ABI_VERSION) { err = "abi_version mismatch"; return FASTP_GPU_E_INVALID; }
    if (in.max_len <= 0 || in.max_len > FASTP_GPU_MAX_READ_LEN) {
        err = "max_len must be in 1..FASTP_GPU_MAX_READ_LEN";
        return FASTP_GPU_E_TOO_LONG;
    }
    // options outside the device path's scope for now: fail loudly, never fall back
    if (in.insert_size_max < 0 || in.insert_size_max > 4096) { err = "insert_size_max out of range"; return FASTP_GPU_E_INVALID; }
    if (in.overlap_diff_limit < 0 || in.overlap_require < 0) { err = "negative overlap knobs"; return FASTP_GPU_E_INVALID; }
    p.paired = in.paired ? 1 : 0;
    p.max_len = in.max_len;
    p.cycles = fastp_gpu_cycles_for(&in);
    p.sw_g = (int)(fastp_gpu_seq_stride(in.max_len) / 4);
    p.qw_g = (int)(fastp_gpu_qual_stride(in.max_len) / 4);
    p.trim_front1 = in.trim_front1; p.trim_tail1 = in.trim_tail1;
    p.trim_front2 = in.trim_front2; p.trim_tail2 = in.trim_tail2;
    p.max_len1 = in.max_len1; p.max_len2 = in.max_len2;
    if (in.trim_front1 < 0 || in.trim_tail1 < 0 || in.trim_front2 < 0 || in.trim_tail2 < 0) {
        err = "negative trim"; return FASTP_GPU_E_INVALID;
    }
    p.cut_front = in.cut_front != 0; p.cut_tail = in.cut_tail != 0; p.cut_right = in.cut_right != 0;
    if (p.cut_front || p.cut_tail || p.cut_right) {  // options.cpp:350-367
        const int ws[3] = {in.cut_front_window, in.cut_tail_window, in.cut_right_window};
        for (int i = 0; i < 3; i++)
            if (ws[i] < 1 || ws[i] > 1000) { err = "cut window size must be 1..1000"; return FASTP_GPU_E_INVALID; }
    }
    p.wF = in.cut_front_window; p.thrF = in.cut_front_window * (33 + in.cut_front_quality);   // filter.cpp:116
    p.wT = in.cut_tail_window;  p.thrT = in.cut_tail_window * (33 + in.cut_tail_quality);     // filter.cpp:185
    p.wR = in.cut_right_window; p.thrR = in.cut_right_window * (33 + in.cut_right_quality);   // filter.cpp:151
    p.qRmin = 33 + in.cut_right_quality;                                                      // filter.cpp:159
    p.poly_g = in.poly_g != 0; p.poly_g_min = in.poly_g_min_len;
    p.poly_x = in.poly_x != 0; p.poly_x_min = in.poly_x_min_len;
    p.adapter_enabled = in.adapter_enabled != 0;
    p.dimer_max_len = in.dimer_max_len;
    int rc = pack_adapter(in.adapter_seq_r1, p.a1w, p.alen1, err);
    if (rc) return rc;
    rc = pack_adapter(in.adapter_seq_r2, p.a2w, p.alen2, err);
    if (rc) return rc;
    p.has_a1 = p.alen1 > 0;
    p.has_a2 = p.alen2 > 0;
    luts.fasta_words.clear();
    luts.fasta_len.clear();
    if (in.n_adapter_fasta < 0 || (in.n_adapter_fasta > 0 && !in.adapter_fasta)) { err = "adapter_fasta list missing"; return FASTP_GPU_E_INVALID; }
    if (in.n_adapter_fasta > 65535) { err = "more than 65535 adapter_fasta sequences"; return FASTP_GPU_E_UNSUPPORTED; }
    for (int i = 0; i < in.n_adapter_fasta; i++) {
        u32 w[MAX_ADAPTER_WORDS];
        int alen = 0;
        rc = pack_adapter(in.adapter_fasta[i], w, alen, err);
        if (rc) return rc;
        for (int k = 0; k < ADAPT_WORDS; k++) luts.fasta_words.push_back(k < MAX_ADAPTER_WORDS ? w[k] : 0u);
        luts.fasta_len.push_back(alen);
    }
    p.n_fasta = in.n_adapter_fasta;
    p.fasta_max_len = 0;
    for (int v : luts.fasta_len) p.fasta_max_len = std::max(p.fasta_max_len, v);
    p.fasta_match_req = p.n_fasta > 256 ? 6 : (p.n_fasta > 16 ? 5 : 4);  // adaptertrimmer.cpp:49-53
    p.correction = (in.correction != 0) && p.paired;  // options.cpp:401-404
    p.allow_gap = (in.allow_gap_overlap_trimming != 0) && p.paired;
    p.merge = (in.merge != 0) && p.paired;
    p.merge_include_unmerged = in.merge_include_unmerged != 0;
    p.overlapped_out = (in.overlapped_out != 0) && p.paired;   // mOverlappedWriter exists for paired input only (peprocessor.cpp:94-97)
    p.overlap_require = in.overlap_require;
    p.overlap_diff_limit = in.overlap_diff_limit;
    p.qual_filter = in.qual_filter != 0;
    p.qual_thr = (int)(unsigned char)num2qual(in.qualified_qual);
    p.n_base_limit = in.n_base_limit;
    p.avg_qual_req = in.avg_qual_req;
    p.length_filter = in.length_filter != 0;
    p.length_required = in.length_required;
    p.length_limit = in.length_limit;
    p.complexity_filter = in.complexity_filter != 0;
    p.dup_enabled = in.dup_enabled != 0;
    p.dedup = (in.dedup != 0) && p.dup_enabled;
    p.isize_max = in.insert_size_max;
    p.umi_len1 = in.umi_len1 > 0 ? in.umi_len1 : 0;
    p.umi_len2 = (p.paired && in.umi_len2 > 0) ? in.umi_len2 : 0;
    p.umi_skip = in.umi_skip > 0 ? in.umi_skip : 0;
    p.need_overlap = p.paired && (p.adapter_enabled || p.correction);
    p.stats_one_pass = !p.correction && !p.merge && !p.cut_front && !p.trim_front1 && !p.trim_front2 && !p.umi_len1 && !p.umi_len2;
    // Front trims the lane plan takes: what moves a kept base is the same number of positions for every read that is written
    // out.  -f / -F are (Filter::trimAndCut :74-77: a shorter read is NULL); Read::trimFront of the UMI step is
    // min(length() - 1, umi + skip) (read.cpp:69-73) - a read too short for its UMI is left with one base, which no run with a
    // length filter of two or more writes out.  --cut_front moves every read by its own amount: the tile kernels.
    const bool any_umi = p.umi_len1 > 0 || p.umi_len2 > 0;
    // --cut_front (round 6): a front of its own per read - on the lane plan when its window predicate is the one the kernel builds
    // anyway (the enabled right / tail cut's window and quality, or its own when it is the only cut), without -c / --merge
    const bool cutf_same = p.cut_front && (p.cut_right ? (p.wF == p.wR && p.thrF == p.thrR) : p.cut_tail ? (p.wF == p.wT && p.thrF == p.thrT) : true) &&
                           p.wF >= 1 && p.wF <= 8;
    p.front_per_read = (cutf_same && !p.correction && !p.merge && (!any_umi || (p.length_filter && p.length_required >= 2))) ? 1 : 0;
    p.front_lane = !p.stats_one_pass && !p.correction && !p.merge && (!p.cut_front || p.front_per_read) && (!any_umi || (p.length_filter && p.length_required >= 2));
    // -c on the lane plan: same condition on the fronts
    // --merge on the lane plan: nothing in front of a read at all (merge mode switches -c on, options.cpp:119-121)
    p.merge_lane = p.merge && p.paired && !p.cut_front && !any_umi && !p.trim_front1 && !p.trim_front2;
    p.corr_lane = p.correction && (!p.merge || p.merge_lane) && !p.cut_front && (!any_umi || (p.length_filter && p.length_required >= 2));
    if (p.corr_lane) p.front_lane = (p.trim_front1 || p.trim_front2 || any_umi) ? 1 : 0;
    p.lane_front1 = (p.umi_len1 > 0 ? p.umi_len1 + p.umi_skip : 0) + p.trim_front1;
    p.lane_front2 = p.paired ? (p.umi_len2 > 0 ? p.umi_len2 + p.umi_skip : 0) + p.trim_front2 : 0;

    // ---- overrepresentation analysis seeds -> hash tables ----
    p.overrep = in.overrep_enabled != 0;
    p.overrep_sampling = in.overrep_sampling;
    for (int m = 0; m < 2; m++) { luts.ovr_table[m].clear(); luts.ovr_sym[m].clear(); luts.ovr_len[m].clear(); }
    if (p.overrep) {
        if (in.overrep_sampling <= 0) { err = "overrep_sampling must be positive"; return FASTP_GPU_E_INVALID; }
        const char* const* lists[2] = {in.overrep_seqs1, in.overrep_seqs2};
        const int ns[2] = {in.n_overrep_seqs1, p.paired ? in.n_overrep_seqs2 : 0};
        const int evl[2] = {in.eval_seq_len1, in.eval_seq_len2};
        for (int m = 0; m < 2; m++) {
            if (ns[m] < 0 || (ns[m] > 0 && !lists[m])) { err = "overrep seed list missing"; return FASTP_GPU_E_INVALID; }
            const int steps[OVR_STEPS] = {10, 20, 40, 100, std::min(150, evl[m] - 2)};  // stats.cpp:273
            for (int s2 = 0; s2 < OVR_STEPS; s2++) {
                luts.ovr_steps[m][s2] = steps[s2];
                u32 pw = 1;
                for (int k = 1; k < steps[s2]; k++) pw *= OVR_HASH_MUL;
                luts.ovr_pw[m][s2] = pw;
            }
            u32 slots = 16;
            while (slots < (u32)ns[m] * 4u) slots <<= 1;
            luts.ovr_table[m].assign((size_t)slots * 2, 0u);
            luts.ovr_sym[m].assign((size_t)std::max(ns[m], 1) * OVR_SEED_STRIDE, 0);
            luts.ovr_len[m].assign((size_t)std::max(ns[m], 1), 0);
            for (int i = 0; i < ns[m]; i++) {
                const char* q = lists[m][i];
                const int L = (int)strlen(q);
                if (L <= 0 || L > 150) { err = "overrep seed length out of range"; return FASTP_GPU_E_INVALID; }
                u32 h = 0;
                for (int k = 0; k < L; k++) {
                    int sy = base_code(q[k]);
                    // a seed the Evaluator cut from reads with letters outside ACGTN: such a byte is its own symbol (> 4), equal
                    // only to the same byte of a listed unit's text (ovr_byte_sym); the packed rows never produce it
                    if (sy < 0) sy = q[k] == 'N' ? 4 : (int)(unsigned char)q[k];
                    luts.ovr_sym[m][(size_t)i * OVR_SEED_STRIDE + k] = (u8)sy;
                    h = h * OVR_HASH_MUL + (u32)(sy + 1);
                }
                luts.ovr_len[m][i] = L;
                const u32 key = h ^ ((u32)L * OVR_SALT_MUL);
                u32 slot = (key * OVR_SALT_MUL) & (slots - 1);
                while (luts.ovr_table[m][2 * slot + 1] != 0) slot = (slot + 1) & (slots - 1);
                luts.ovr_table[m][2 * slot] = key;
                luts.ovr_table[m][2 * slot + 1] = (u32)i + 1;
            }
        }
    }

    // ---- LUTs: the reference's floating point thresholds, evaluated on the host ----
    const int n = p.cycles + 2;  // a merged read can be as long as both mates (cycles = 2*max_len in merge mode)
    luts.ov_limit.assign(n, 0);
    luts.lowq_limit.assign(n, 0);
    luts.cplx_min.assign(n, 0);
    const double diffPercentLimit = in.overlap_diff_percent_limit / 100.0;  // peprocessor.cpp:440
    for (int ol = 0; ol <= p.cycles; ol++) {
        int lim = (int)(ol * diffPercentLimit);  // overlapanalysis.cpp:51,76
        if (in.overlap_diff_limit < lim) lim = in.overlap_diff_limit;
        luts.ov_limit[ol] = (int16_t)lim;
        if (lim > p.ov_limit_max) p.ov_limit_max = lim;
        // filter.cpp:36: FAIL iff lowQualNum > (pct * rlen / 100.0)  <=> lowQualNum > floor(x)
        const double x = in.unqualified_percent_limit * ol / 100.0;
        double fl = floor(x);
        if (fl < 0) fl = -1;  // any count (>= 0) exceeds a negative bound
        if (fl > 65534) fl = 65534;
        luts.lowq_limit[ol] = (u16)(fl < 0 ? 0xFFFF : (int)fl);
        // filter.cpp:65: pass iff (double)diff/(double)(len-1) >= threshold
        int need = ol;  // impossible -> always fails
        if (ol >= 2) {
            for (int d = 0; d <= ol - 1; d++) {
                if ((double)d / (double)(ol - 1) >= in.complexity_threshold) { need = d; break; }
            }
        }
        luts.cplx_min[ol] = (u16)need;
    }
    // a negative lowq bound cannot be expressed as "low > lut"; only reachable with a negative
    // percent limit, which the reference's CLI does not validate either - reject it.
    if (in.unqualified_percent_limit < 0) { err = "unqualified_percent_limit < 0"; return FASTP_GPU_E_INVALID; }

    if (p.dup_enabled) {
        u64 bytes;
        int num;
        dup_geometry(in.dup_accuracy_level, bytes, num);
        p.dup_bufnum = num;
        p.dup_bits = bytes << 3;
        dup_primes(num, luts.dup_primes);
        // position part of Duplicate::seq2intvector: sum_{t<n} prime[(t*B+i)&mask] * t
        const int maxpos = 2 * in.max_len;
        luts.dup_posum.assign((size_t)(maxpos + 1) * num, 0);
        const u32 mask = (u32)(512 * num - 1);
        for (int i = 0; i < num; i++) {
            u64 acc = 0;
            luts.dup_posum[i] = 0;
            for (int t = 0; t < maxpos; t++) {
                acc += (u64)luts.dup_primes[((u32)t * (u32)num + (u32)i) & mask] * (u64)t;
                luts.dup_posum[(size_t)(t + 1) * num + i] = acc;
            }
        }
        // phase_hash_dot: the value part sum_p prime[...] * val(base_p) as byte-plane dot products (v_dot4_u32_u8).
        // Entry (r, q) holds, for every buffer i and byte plane b, the b-th bytes of the primes of the four
        // consecutive stream positions 4q + r .. 4q + r + 3.  A read whose first base sits at stream position
        // `off` (0 for read 1, read 1's length for read 2, duplicate.cpp:139) reads entry (off & 3, (off >> 2) + c)
        // for its base dword c.  Only built for the geometries where it fits LDS comfortably.
        luts.dup_planes.clear();
        luts.dup_nq = 0;
        p.dup_npl = 3;
        if (num <= 4) {
            const int nq = (maxpos + 3) / 4 + 2;
            u32 pmax = 0;
            for (int t = 0; t < 4 * nq + 3; t++)
                for (int i = 0; i < num; i++) pmax = std::max(pmax, luts.dup_primes[((u32)t * (u32)num + (u32)i) & mask]);
            p.dup_npl = pmax < (1u << 24) ? 3 : 4;
            luts.dup_nq = nq;
            luts.dup_planes.assign((size_t)4 * nq * num * p.dup_npl, 0u);
            for (int r = 0; r < 4; r++)
                for (int q = 0; q < nq; q++)
                    for (int i = 0; i < num; i++)
                        for (int b = 0; b < p.dup_npl; b++) {
                            u32 w = 0;
                            for (int k = 0; k < 4; k++) {
                                const u32 pr = luts.dup_primes[((u32)(4 * q + r + k) * (u32)num + (u32)i) & mask];
                                w |= ((pr >> (8 * b)) & 0xFFu) << (8 * k);
                            }
                            luts.dup_planes[(((size_t)r * nq + q) * num + i) * p.dup_npl + b] = w;
                        }
        }
    } else {
        p.dup_bufnum = 0;
        p.dup_bits = 0;
        luts.dup_primes.clear();
        luts.dup_posum.clear();
        luts.dup_planes.clear();
        luts.dup_nq = 0;
    }
    return FASTP_GPU_OK;
}

static int round_odd(int x) { return (x & 1) ? x : x + 1; }
static int imin_i(int a, int b) { return a < b ? a : b; }

int compute_lds_layout(const DevParams& p, TileConfig& cfg, LdsLayout& L, std::string& err, int hp_nq) {
    const int mates = p.paired ? 2 : 1;
    const int halves = cfg.halves == 2 ? 2 : 1;
    auto build = [&](int P, LdsLayout& out) {
        memset(&out, 0, sizeof(out));
        out.halves = halves;
        out.P = P;
        out.NR = mates * P;
        out.SW = p.sw_g;
        out.QW = p.qw_g;
        out.C = p.cycles;
        out.Cp = (p.cycles + 3) / 4 * 4;
        int o = 0;
        auto take = [&](int n) { int at = o; o += n; return at; };
        // ---- shared by the tiles in flight: accumulators first (u64 part 8-byte aligned at offset 0), tables ----
        out.acc_cyc = take(cfg.split ? 0 : 4 * N_CLS * out.Cp * 2);
        out.acc_kmer = take(cfg.split ? 0 : 4 * KMER_BINS);
        out.acc_qh = take(cfg.split ? 0 : 4 * 128 * QT_DWORDS);
        out.acc_misc = take(MISC_ISIZE + p.isize_max + 1);
        out.acc_end = o;
        out.val4_lut = take(p.dup_bufnum > 0 ? 256 : 0);
        out.adapt = take(2 * ADAPT_WORDS);
        const int lw = (p.cycles + 2) / 2;
        out.lut_ov = take(lw);
        out.lut_lowq = take(lw);
        out.lut_cplx = take(lw);
        // Duplicate's primes: as byte planes for the dot-product hash when that table was built, else the plain list
        out.hp = 0;
        out.has_hp = 0;
        out.hp_nq = 0;
        out.primes = 0;
        if (p.dup_bufnum > 0 && hp_nq > 0) {
            if (o & 1) o++;
            out.hp_nq = hp_nq;
            out.has_hp = 1;
            out.hp = take(4 * hp_nq * p.dup_bufnum * p.dup_npl);
        } else {
            out.primes = take(p.dup_bufnum > 0 ? 512 * p.dup_bufnum : 0);
        }
        o = (o + 3) & ~3;
        out.tile_begin = o;
        // ---- per tile ----
        out.bar = take(2);
        if (o & 1) o++;
        out.hash = take(out.NR * (p.dup_bufnum > 0 ? p.dup_bufnum : 0) * 2);
        {   // trimAndCut predicate masks, only those the options need
            int k = 0;
            out.wm_words = (p.max_len + 31) / 32;
            out.wm_badF = p.cut_front ? (k++) * out.wm_words : -1;
            out.wm_badR = p.cut_right ? (k++) * out.wm_words : -1;
            out.wm_badT = (p.cut_tail && !p.cut_right) ? (k++) * out.wm_words : -1;   // filter.cpp:166
            out.wm_isN = (p.cut_front || (p.cut_tail && !p.cut_right)) ? (k++) * out.wm_words : -1;
            out.wm_stride = k * out.wm_words;
            out.wm = take(out.NR * out.wm_stride);
        }
        o = (o + 3) & ~3;  // 16-byte aligned rows (vector tile copies)
        out.seq = take(out.NR * out.SW);
        o = (o + 3) & ~3;
        out.nmk = take(out.NR * out.SW);
        o = (o + 3) & ~3;
        out.qual = take(out.NR * out.QW);
        out.rc = out.rcn = out.cand = -1;
        out.cand_cap = 0;
        if (p.paired) {
            out.rc = take(P * out.SW + 2);    // + the dword a 16-base window read may touch behind the last row
            out.rcn = take(P * out.SW + 2);
            out.cand_cap = 4 * P;
            out.cand = take(1 + out.cand_cap);
        }
        out.rlen0 = take(out.NR);
        out.front = take(out.NR);
        out.len = take(out.NR);
        out.flags = take(out.NR);
        out.ft = take(out.NR);
        out.apos = take(out.NR);
        out.alen = take(out.NR);
        out.code = take(out.NR);
        out.swin = take(out.NR);
        out.mlen = take(out.NR);
        out.olen = p.overlapped_out ? take(out.NR) : -1;
        out.wl_cap = imin_i(2046, 8 * out.NR + 62);
        out.wl_cap -= out.wl_cap & 1;
        out.wl = take(1 + out.wl_cap / 2);
        out.met = take(out.NR * 2);
        out.ov_off = take(P);
        out.ov_len = take(P);
        out.ov_diff = take(P);
        out.ov_flags = take(P);
        o = (o + 3) & ~3;
        out.tile_stride = o - out.tile_begin;
        out.total = out.tile_begin + halves * out.tile_stride;
    };
    if (cfg.P > 0) {
        build(cfg.P, L);
        if (L.total * 4 > cfg.lds_budget) { err = "tile does not fit the LDS budget"; return FASTP_GPU_E_INVALID; }
        return FASTP_GPU_OK;
    }
    // largest P (multiple of 32 once >= 32, else multiple of 4) that fits
    int best = 0;
    for (int P = 4; P <= 1024; P += (P >= 32 ? 32 : 4)) {
        LdsLayout t;
        build(P, t);
        if (t.total * 4 <= cfg.lds_budget) best = P;
        else break;
    }
    if (!best) { err = "no tile size fits the LDS budget"; return FASTP_GPU_E_INVALID; }
    cfg.P = best;
    build(best, L);
    return FASTP_GPU_OK;
}

// the layout half h of a workgroup works with: its LDS base sits h * tile_stride dwords higher, so the offsets of
// everything SHARED between the tiles move down by that much; per-tile offsets stay
LdsLayout layout_for_half(const LdsLayout& L, int h) {
    LdsLayout o = L;
    const int d = h * L.tile_stride;
    o.acc_cyc -= d; o.acc_kmer -= d; o.acc_qh -= d; o.acc_misc -= d; o.acc_end -= d;
    o.val4_lut -= d; o.adapt -= d; o.lut_ov -= d; o.lut_lowq -= d; o.lut_cplx -= d; o.hp -= d; o.primes -= d;
    return o;
}

}  // namespace fq

// ---------------------------------------------------------------------------
// C ABI functions that need no device
// ---------------------------------------------------------------------------
extern "C" {

size_t fastp_gpu_seq_stride(int max_len) { return (size_t)(((max_len + 3) / 4 + 7) / 8 * 8); }
size_t fastp_gpu_qual_stride(int max_len) { return (size_t)((max_len + 7) / 8 * 8); }

int fastp_gpu_cycles_for(const fastp_gpu_params* p) { return p->merge ? 2 * p->max_len : p->max_len; }

void fastp_gpu_default_params(fastp_gpu_params* p, int paired, int max_len) {
    memset(p, 0, sizeof(*p));
    p->abi_version = FASTP_GPU_ABI_VERSION;
    p->paired = paired ? 1 : 0;
    p->max_len = max_len;
    p->cut_front_window = p->cut_tail_window = p->cut_right_window = 4;       // main.cpp:90
    p->cut_front_quality = p->cut_tail_quality = p->cut_right_quality = 20;   // main.cpp:91
    p->poly_g_min_len = 10;            // main.cpp:79
    p->poly_x_min_len = 10;            // main.cpp:84
    p->adapter_enabled = 1;            // main.cpp:55
    p->dimer_max_len = 2;              // main.cpp:62
    p->overlap_require = 30;           // main.cpp:123
    p->overlap_diff_limit = 5;         // main.cpp:124
    p->overlap_diff_percent_limit = 20;  // main.cpp:125
    p->qual_filter = 1;                // main.cpp:101-105
    p->qualified_qual = 15;
    p->unqualified_percent_limit = 40;
    p->n_base_limit = 5;
    p->length_filter = 1;              // main.cpp:108-110
    p->length_required = 15;
    p->complexity_threshold = 0.30;    // main.cpp:114, 342
    p->dup_enabled = 1;                // main.cpp:201-210
    p->dup_accuracy_level = 1;
    p->insert_size_max = 512;          // options.cpp:23
}

void fastp_gpu_counter_layout_for(int cycles, int insert_size_max, fastp_gpu_counter_layout* L) {
    int64_t o = 0;
    memset(L, 0, sizeof(*L));
    L->cycles = cycles;
    o += 4;  // header: abi version, cycles, insert_size_max, reserved
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
    L->st_size = L->st_cycle + 34 * (int64_t)cycles;
    for (int s = 0; s < 4; s++) { L->stats[s] = o; o += L->st_size; }
    for (int s = 0; s < 4; s++) { L->overrep_count[s] = o; L->overrep_dist[s] = o; }
    L->total = o;
}

void fastp_gpu_counter_layout_for_params(const fastp_gpu_params* p, fastp_gpu_counter_layout* L) {
    fastp_gpu_counter_layout_for(fastp_gpu_cycles_for(p), p->insert_size_max, L);
    if (!p->overrep_enabled) return;
    int64_t o = L->total;
    for (int s = 0; s < 4; s++) {  // PRE1, POST1 use read-1 seeds; PRE2, POST2 read-2 seeds (stats.cpp:956-971)
        const bool r2 = s >= 2;
        L->n_overrep[s] = r2 ? p->n_overrep_seqs2 : p->n_overrep_seqs1;
        L->eval_len[s] = r2 ? p->eval_seq_len2 : p->eval_seq_len1;
        if (!p->paired && r2) { L->n_overrep[s] = 0; L->eval_len[s] = 0; }
        L->overrep_count[s] = o;
        o += L->n_overrep[s];
        L->overrep_dist[s] = o;
        o += L->n_overrep[s] * L->eval_len[s];
    }
    L->total = o;
}

// Eight bases / quality characters at a time (the packer runs on the host's worker threads between the reader and the
// submit: one byte at a time through a switch it was half of what a worker did per pack).
//   letters:  x = (c >> 1) & 3 is A 0, C 1, T 2, G 3 (and 3 for N); fastp's code A 0, T 1, C 2, G 3 is x with its two
//             bits swapped; an N becomes code 0 + the flag; anything that is not one of the five letters is refused
//   quality:  '!'..'~' only (the kernels take q - 33 as an unsigned field, stats.cpp:223,226)
namespace {
inline uint64_t load8(const char* p, int n) {   // n <= 8 bytes, zero padded
    uint64_t v = 0;
    memcpy(&v, p, (size_t)n);
    return v;
}
inline uint64_t zero_bytes(uint64_t v) {   // 0x80 in every byte of v that is zero (exact: no borrow between bytes)
    const uint64_t m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((v & m) + m) | v | m);
}
inline uint64_t bytes_equal(uint64_t v, unsigned char c) { return zero_bytes(v ^ (0x0101010101010101ull * c)); }
}  // namespace

static int pack_reads(int max_len, int n, const char* const* seqs, const char* const* quals,
                      const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out,
                      int32_t* bad_read, uint8_t* exotic);

int fastp_gpu_pack_reads(int max_len, int n, const char* const* seqs, const char* const* quals,
                         const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out,
                         int32_t* bad_read) {
    return pack_reads(max_len, n, seqs, quals, lens, seq_out, qual_out, len_out, bad_read, nullptr);
}

int fastp_gpu_pack_reads_x(int max_len, int n, const char* const* seqs, const char* const* quals,
                           const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out,
                           int32_t* bad_read, uint8_t* exotic) {
    if (!exotic) return FASTP_GPU_E_INVALID;
    return pack_reads(max_len, n, seqs, quals, lens, seq_out, qual_out, len_out, bad_read, exotic);
}

static int pack_reads(int max_len, int n, const char* const* seqs, const char* const* quals,
                      const int32_t* lens, uint8_t* seq_out, uint8_t* qual_out, uint16_t* len_out,
                      int32_t* bad_read, uint8_t* exotic) {
    if (n < 0 || max_len <= 0 || !seqs || !quals || !lens || !seq_out || !qual_out || !len_out)
        return FASTP_GPU_E_INVALID;
    const size_t ss = fastp_gpu_seq_stride(max_len), qs = fastp_gpu_qual_stride(max_len);
    for (int i = 0; i < n; i++) {
        const int len = lens[i];
        if (len < 0 || len > max_len) { if (bad_read) *bad_read = i; return FASTP_GPU_E_TOO_LONG; }
        uint8_t* so = seq_out + (size_t)i * ss;
        uint8_t* qo = qual_out + (size_t)i * qs;
        const char* s = seqs[i];
        const char* q = quals[i];
        int j = 0;
        for (; j < len; j += 8) {   // both strides are multiples of 8 bytes of quality / 2 bytes of bases: whole groups fit
            const int m = len - j < 8 ? len - j : 8;
            const uint64_t live = m == 8 ? ~0ull : ((1ull << (8 * m)) - 1ull);
            const uint64_t c = load8(s + j, m), qq = load8(q + j, m);
            const uint64_t isn = bytes_equal(c, 'N');
            const uint64_t known = bytes_equal(c, 'A') | bytes_equal(c, 'C') | bytes_equal(c, 'G') | bytes_equal(c, 'T') | isn;
            // quality in '!'..'~': q - 33 does not borrow and q + 1 does not reach bit 7
            const uint64_t qbad = ((qq | 0x8080808080808080ull) - 0x2121212121212121ull) ^ 0x8080808080808080ull;   // bit 7 set <=> q < 33 (for q < 128)
            const uint64_t qhigh = (qq | ((qq & 0x7F7F7F7F7F7F7F7Full) + 0x0101010101010101ull)) & 0x8080808080808080ull;   // q >= 127
            const uint64_t foreign = ~known & 0x8080808080808080ull & live;
            if (((((qbad & 0x8080808080808080ull) | qhigh) & 0x8080808080808080ull & live) != 0) || (foreign && !exotic)) {
                if (bad_read) *bad_read = i;
                return FASTP_GPU_E_ALPHABET;
            }
            if (foreign) exotic[i] = 1;   // the unit goes through the text kernel (fq_text.h); its packed row is a placeholder
            uint64_t code = (c >> 1) & 0x0303030303030303ull;
            code = ((code >> 1) | (code << 1)) & 0x0303030303030303ull;   // swap the two bits: A 0, T 1, C 2, G 3
            code &= ~(((isn | foreign) >> 7) * 3ull);                     // an N (and a foreign letter) is code 0
            code &= live;
            // gather the eight 2-bit codes into 16 bits
            code = (code | (code >> 6)) & 0x000F000F000F000Full;
            code = (code | (code >> 12)) & 0x000000FF000000FFull;
            code = (code | (code >> 24)) & 0xFFFFull;
            so[j >> 2] = (uint8_t)code;
            so[(j >> 2) + 1] = (uint8_t)(code >> 8);
            const uint64_t qv = (qq | isn) & live;
            memcpy(qo + j, &qv, 8);
        }
        // the rest of the row is zero (fastp_gpu.h: bytes past a read's length)
        const size_t sdone = (size_t)((len + 7) / 8) * 2, qdone = (size_t)((len + 7) / 8) * 8;
        if (sdone < ss) memset(so + sdone, 0, ss - sdone);
        if (qdone < qs) memset(qo + qdone, 0, qs - qdone);
        len_out[i] = (uint16_t)len;
    }
    return FASTP_GPU_OK;
}

}  // extern "C"
