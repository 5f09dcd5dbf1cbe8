// fq_host.h - host-side logic of the engine that does not touch the HIP runtime:
// parameter flattening, LUT construction (with the reference's own double
// expressions), LDS layout, ASCII -> 2-bit packing, counter-block layout.
#pragma once
#include <string>
#include <vector>

#include "../../include/fastp_gpu.h"
#include "fq_types.h"

namespace fq {

struct HostLuts {
    std::vector<int16_t> ov_limit;   // [max_len+2]
    std::vector<u16> lowq_limit;     // [max_len+2]
    std::vector<u16> cplx_min;       // [max_len+2]
    std::vector<u32> dup_primes;     // [bufnum*512]
    std::vector<u64> dup_posum;      // [(2*max_len+1)*bufnum]
    std::vector<u32> dup_planes;     // [4][dup_nq][bufnum][dup_npl], empty when the table would not pay (see build_dev_params)
    int dup_nq = 0;
    std::vector<u32> fasta_words;    // [n_fasta][ADAPT_WORDS]
    std::vector<int> fasta_len;      // [n_fasta]
    // overrepresentation analysis seeds, per mate
    std::vector<u32> ovr_table[2];   // pairs {key, index + 1}
    std::vector<u8> ovr_sym[2];      // [n][OVR_SEED_STRIDE]
    std::vector<int> ovr_len[2];
    int ovr_steps[2][OVR_STEPS];
    u32 ovr_pw[2][OVR_STEPS];
};

// device geometry the layout is sized for
struct TileConfig {
    int threads;      // workgroup size (multiple of 64)
    int P;            // pairs / reads per tile
    int lds_budget;   // bytes of LDS one workgroup may use
    int halves;       // tiles in flight per workgroup: 2 = each half of the waves owns one (fused_body), 1 = one tile
    int split;        // split plan: no per-cycle / k-mer / histogram accumulators in this kernel's LDS (fq_stats.h)
};

// returns FASTP_GPU_OK or an error code; err receives a human readable reason
int build_dev_params(const fastp_gpu_params& in, DevParams& out, HostLuts& luts, std::string& err);

// picks P (if cfg.P == 0) so that the tile fits cfg.lds_budget; fills L
// hp_nq = HostLuts::dup_nq (0: no byte-plane prime table, the generic hash path is used)
int compute_lds_layout(const DevParams& p, TileConfig& cfg, LdsLayout& L, std::string& err, int hp_nq = 0);

LdsLayout layout_for_half(const LdsLayout& L, int h);   // see LdsLayout::halves

u32 magic_for(u32 d);  // ceil(2^32 / d)

void dup_geometry(int level, u64& bytes_per_buf, int& bufnum);  // duplicate.cpp:13-47
void dup_primes(int bufnum, std::vector<u32>& out);              // duplicate.cpp:66-84

}  // namespace fq
