// Host side of the drop-in on one core: the packer and the record application per 1000-pair pack (no GPU needed).
//   g++ -O2 -std=c++17 -Iinclude tools/host_bench.cpp -Lfastp_amd -lfastp_gpu -Wl,-rpath,$PWD/fastp_amd -o /tmp/host_bench && /tmp/host_bench
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "fastp_gpu.h"
#include "fastp_gpu_host.h"
int main() {
    const int n = 1000, L = 150, packs = 200;
    std::mt19937 rng(1);
    std::vector<std::string> name(n), seq(n), qual(n), name2(n), seq2(n), qual2(n);
    for (int i = 0; i < n; i++) {
        name[i] = "@SIM:1:FC:1:1101:" + std::to_string(i) + " 1:N:0:ATCG"; name2[i] = "@SIM:1:FC:1:1101:" + std::to_string(i) + " 2:N:0:ATCG";
        seq[i].resize(L); qual[i].assign(L, 'I'); seq2[i].resize(L); qual2[i].assign(L, 'I');
        for (int j = 0; j < L; j++) { seq[i][j] = "ACGT"[rng() & 3]; seq2[i][j] = "ACGT"[rng() & 3]; }
    }
    std::vector<const char*> np(n), sp(n), qp(n), st(n), np2(n), sp2(n), qp2(n);
    std::vector<int32_t> nl(n), ll(n), sl(n), nl2(n);
    for (int i = 0; i < n; i++) { np[i] = name[i].data(); nl[i] = (int)name[i].size(); sp[i] = seq[i].data(); qp[i] = qual[i].data(); ll[i] = L; st[i] = "+"; sl[i] = 1;
                                  np2[i] = name2[i].data(); nl2[i] = (int)name2[i].size(); sp2[i] = seq2[i].data(); qp2[i] = qual2[i].data(); }
    // packer
    const size_t ss = fastp_gpu_seq_stride(L), qs = fastp_gpu_qual_stride(L);
    std::vector<uint8_t> so(n * ss), qo(n * qs); std::vector<uint16_t> lo(n);
    auto t0 = std::chrono::steady_clock::now();
    int32_t bad;
    for (int r = 0; r < packs; r++) { fastp_gpu_pack_reads(L, n, sp.data(), qp.data(), ll.data(), so.data(), qo.data(), lo.data(), &bad);
                                      fastp_gpu_pack_reads(L, n, sp2.data(), qp2.data(), ll.data(), so.data(), qo.data(), lo.data(), &bad); }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("fastp_gpu_pack_reads: %.0f ns per pair (2 x 150 bp)\n", dt / (packs * n) * 1e9);
    // apply
    fastp_gpu_params p; fastp_gpu_default_params(&p, 1, L);
    fastp_gpu_host_options ho; memset(&ho, 0, sizeof(ho)); ho.want_failed = 1;
    fastp_gpu_host* h = nullptr; fastp_gpu_host_create(&p, &ho, &h);
    std::vector<fastp_gpu_read_result> r1(n), r2(n); std::vector<fastp_gpu_pair_result> pr(n);
    memset(r1.data(), 0, n * sizeof(r1[0])); memset(r2.data(), 0, n * sizeof(r2[0])); memset(pr.data(), 0, n * sizeof(pr[0]));
    for (int i = 0; i < n; i++) { r1[i].len = (uint16_t)(i % 7 == 0 ? 120 : L); r2[i].len = L; if (i % 50 == 0) r1[i].code = 12; }
    fastp_gpu_reads b1{n, np.data(), nl.data(), sp.data(), qp.data(), ll.data(), st.data(), sl.data()};
    fastp_gpu_reads b2{n, np2.data(), nl2.data(), sp2.data(), qp2.data(), ll.data(), st.data(), sl.data()};
    fastp_gpu_results res; memset(&res, 0, sizeof(res)); res.r1 = r1.data(); res.r2 = r2.data(); res.pair = pr.data();
    t0 = std::chrono::steady_clock::now();
    size_t bytes = 0;
    for (int r = 0; r < packs; r++) { fastp_gpu_host_apply(h, &b1, &b2, &res); size_t l; fastp_gpu_host_output(h, FASTP_GPU_OUT1, &l); bytes += l; fastp_gpu_host_clear_outputs(h); }
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("fastp_gpu_host_apply: %.0f ns per pair (%zu bytes of out1 per pack)\n", dt / (packs * n) * 1e9, bytes / packs);
    return 0;
}
