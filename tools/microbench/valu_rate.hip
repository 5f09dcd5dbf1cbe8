// valu_rate.hip - integer VALU issue rate of gfx950, the bound that actually limits the fused
// kernel.  Each kernel runs a long chain-free stream of one instruction kind on
// 8 independent registers per lane; rate = wave-instructions * 64 lanes / time.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define UNROLL 8

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; i++) r[i] = seed * (i + 1) + threadIdx.x;
    const uint32_t c = seed | 1u;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) r[i] = r[i] + r[(i + 3) & 7];                                          // v_add_u32
            else if (OP == 1) r[i] = (r[i] & c) | (r[i] >> 1);                        // v_and_or / shifts (2 ops)
            else if (OP == 2) r[i] = __popc(r[i]) + r[i];                             // v_bcnt_u32_b32 (with add operand)
            else if (OP == 3) r[i] = __builtin_amdgcn_alignbit(r[i], c, r[i] & 31);   // v_alignbit_b32 (+ v_and)
            else if (OP == 4) r[i] = r[i] * (r[(i + 3) & 7] | 1u);                                      // v_mul_lo_u32
            else if (OP == 5) r[i] = __umul24(r[i], c) + c;                           // v_mad_u32_u24
            else if (OP == 6) r[i] = __builtin_amdgcn_sad_u8(r[i], c, r[i]);          // v_sad_u8
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, int ops_per_stmt, uint32_t* d) {
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<OP><<<blocks, threads>>>(d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, threads>>>(d, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double lane_ops = (double)blocks * threads * ITERS * UNROLL * ops_per_stmt;
    const double rate = lane_ops / (ms * 1e-3);
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  (%.1f lanes/clk/SIMD at 2.1 GHz, 1024 SIMDs)\n", name, ms, rate / 1e12,
           rate / (1024.0 * 2.1e9));
    return rate;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", 1, d);
    run<1>("and/or/shift (2 ops)", 2, d);
    run<2>("v_bcnt_u32_b32", 1, d);
    run<3>("v_alignbit_b32 (+and)", 2, d);
    run<4>("v_mul_lo_u32", 1, d);
    run<5>("v_mad_u32_u24", 1, d);
    run<6>("v_sad_u8", 1, d);
    return 0;
}
