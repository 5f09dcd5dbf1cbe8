// stats_mfma.hip - the measurement the round-4 review asked for instead of a costing on paper: Stats::statRead's per-cycle
// table (stats.cpp:206-222: per cycle and base class the count, the Q20 / Q30 counts and the quality sum) as
// v_mfma_i32_16x16x64_i8 contractions over 64 reads,
//     C[class | kept][feature] += A[class | kept][read] x B[read][feature],   A = one-hot(class | kept << 2),
//     B = [1, q >= Q20, q >= Q30, q - 33],
// fed from CYCLE-MAJOR planes (code[cycle][read], qv[cycle][read]: one byte per base each) - the layout a contraction over reads
// needs and the engine's row-major batches do not have.  One MFMA takes two cycles (rows 0-7 / 8-15 of A, columns 0-3 / 4-7 of
// B; the off-diagonal blocks are thrown away).  What is measured: the operand build (v_perm one-hots, SWAR threshold bytes) per
// (cycle, 64 reads) cell group, with the planes already written - i.e. WITHOUT what producing them would cost the lane kernel
// (300 byte stores + the extraction per pair).  Self-checked against a plain atomics kernel on the same planes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 stats_mfma.hip -o stats_mfma && ./stats_mfma [reads]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
enum { C = 152, CPW = 8 /* cycles per wavefront: CPW / 2 accumulators of 4 VGPRs */, ROWS = 8, FEAT = 4 };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// table[cycle][row][feature] (int32 per launch; the engine would fold them into its int64 block)
__global__ void __launch_bounds__(256) k_mfma(const uint8_t* __restrict__ code, const uint8_t* __restrict__ qv, int N, int slices, int* table) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
    const int groups = C / CPW;
    const int cg = wave % groups, slice = wave / groups;
    if (slice >= slices) return;
    const int c0 = cg * CPW;
    const int per = ((N / 64 + slices - 1) / slices) * 64;
    const int n0 = slice * per, n1 = min(N, n0 + per);
    const int r = l & 15, g = l >> 4;
    const uint64_t tab = 1ull << (8 * (r & 7));                 // the one-hot table of this lane's row: byte (r & 7) = 1
    const uint32_t tab_lo = (uint32_t)tab, tab_hi = (uint32_t)(tab >> 32);
    const int f = r & 3;                                        // this lane's feature column
    const uint32_t thr4 = (f == 1 ? 20u : f == 2 ? 30u : 0u) * 0x01010101u;
    const bool is_sum = f == 3, live_col = r < 8;
    v4i acc[CPW / 2];
    for (int i = 0; i < CPW / 2; i++) acc[i] = v4i{0, 0, 0, 0};
    for (int n = n0; n < n1; n += 64) {
#pragma unroll
        for (int jp = 0; jp < CPW / 2; jp++) {
            const int cycA = c0 + 2 * jp + (r >> 3), cycB = c0 + 2 * jp + ((r >> 2) & 1);
            const uint4 cd = *(const uint4*)(code + (size_t)cycA * N + n + 16 * g);
            const uint4 qd = *(const uint4*)(qv + (size_t)cycB * N + n + 16 * g);
            v4i a, b;
            a.x = (int)__builtin_amdgcn_perm(tab_hi, tab_lo, cd.x);   // byte i = (code byte i == r & 7); code 12 ("no base") -> 0
            a.y = (int)__builtin_amdgcn_perm(tab_hi, tab_lo, cd.y);
            a.z = (int)__builtin_amdgcn_perm(tab_hi, tab_lo, cd.z);
            a.w = (int)__builtin_amdgcn_perm(tab_hi, tab_lo, cd.w);
            const uint32_t q[4] = {qd.x, qd.y, qd.z, qd.w};
            uint32_t bb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t t = (((q[k] | 0x80808080u) - thr4) & 0x80808080u) >> 7;   // 1 where q - 33 >= threshold
                bb[k] = live_col ? (is_sum ? q[k] : t) : 0u;
            }
            b.x = (int)bb[0]; b.y = (int)bb[1]; b.z = (int)bb[2]; b.w = (int)bb[3];
            acc[jp] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[jp], 0, 0, 0);
        }
    }
    // C[i = 4 * (l / 16) + t][j = l % 16]; rows 0-7 x columns 0-3 = cycle c0 + 2 jp, rows 8-15 x columns 4-7 = the next one
#pragma unroll
    for (int jp = 0; jp < CPW / 2; jp++)
        for (int t = 0; t < 4; t++) {
            const int i = 4 * g + t, j = r;
            const int v = acc[jp][t];
            if (i < 8 && j < 4 && v) atomicAdd(&table[((c0 + 2 * jp) * ROWS + i) * FEAT + j], v);
            if (i >= 8 && j >= 4 && j < 8 && v) atomicAdd(&table[((c0 + 2 * jp + 1) * ROWS + (i - 8)) * FEAT + (j - 4)], v);
        }
}

// the checker: one lane per (cycle, read), global atomics
__global__ void k_ref(const uint8_t* code, const uint8_t* qv, int N, int* table) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)C * N) return;
    const int c = (int)(idx / N);
    const uint32_t cd = code[idx], q = qv[idx];
    if (cd >= 8) return;
    int* t = table + (c * ROWS + cd) * FEAT;
    atomicAdd(&t[0], 1);
    if (q >= 20) atomicAdd(&t[1], 1);
    if (q >= 30) atomicAdd(&t[2], 1);
    atomicAdd(&t[3], (int)q);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4 * 1024 * 1024;   // reads of ONE mate (a bench batch has 2 x 4,194,304)
    const size_t bytes = (size_t)C * N;
    std::vector<uint8_t> hc(bytes), hq(bytes);
    uint32_t s = 12345;
    for (size_t i = 0; i < bytes; i++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = s >> 8;
        const int c = (int)(i / N);
        const bool none = c >= 150 || (r & 0xFF) < 3;            // behind the read's end / an N: "no base"
        hc[i] = none ? 12 : (uint8_t)((r >> 8) & 7);             // class | kept << 2
        hq[i] = (uint8_t)(((r >> 12) & 63) % 42);                // q - 33 in 0 .. 41
    }
    uint8_t *dc, *dq;
    int *t1, *t2;
    CHECK(hipMalloc(&dc, bytes));
    CHECK(hipMalloc(&dq, bytes));
    CHECK(hipMalloc(&t1, C * ROWS * FEAT * 4));
    CHECK(hipMalloc(&t2, C * ROWS * FEAT * 4));
    CHECK(hipMemcpy(dc, hc.data(), bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dq, hq.data(), bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int groups = C / CPW;                                   // 19 cycle groups
    const int waves_wanted = prop.multiProcessorCount * 4 * 4;    // four waves per SIMD
    const int slices = (waves_wanted + groups - 1) / groups;
    const int waves = slices * groups, blocks = (waves * 64 + 255) / 256;
    CHECK(hipMemset(t1, 0, C * ROWS * FEAT * 4));
    CHECK(hipMemset(t2, 0, C * ROWS * FEAT * 4));
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, dc, dq, N, slices, t1);
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, 0, dc, dq, N, t2);
    CHECK(hipDeviceSynchronize());
    std::vector<int> a(C * ROWS * FEAT), b(C * ROWS * FEAT);
    CHECK(hipMemcpy(a.data(), t1, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), t2, b.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t i = 0; i < a.size(); i++)
        if (a[i] != b[i] && bad++ < 5) printf("MISMATCH cell %zu (cycle %zu row %zu feature %zu): mfma %d atomics %d\n", i, i / 32, (i / 4) % 8, i % 4, a[i], b[i]);
    printf("self-check against the atomics kernel: %s (%d of %zu cells differ)\n", bad ? "FAILED" : "equal", bad, a.size());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, dc, dq, N, slices, t1);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double cells = (double)C * N / 64.0;                    // (cycle, 64 reads) cell groups
    printf("k_mfma: %d reads x %d cycles of ONE mate (cycle-major planes resident), %d waves: %.4f ms per launch = %.3f ns per (cycle, 64 reads) cell group,\n"
           "        %.1f GB/s of plane bytes; a pair of the bench batch (2 mates) would take %.4f ms per 4,194,304 pairs for the per-cycle table alone\n",
           N, C, waves, ms, ms * 1e6 / cells, 2.0 * bytes / (ms * 1e-3) / 1e9, 2.0 * ms * (4194304.0 / N));
    return bad ? 1 : 0;
}
