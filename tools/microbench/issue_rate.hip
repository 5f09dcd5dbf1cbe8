// issue_rate.hip - what one wave64 VALU / LDS instruction costs on gfx950, measured with inline asm so that the
// instruction under test is exactly what runs: 16 independent destination registers per lane, 8 waves per SIMD,
// no dependency between consecutive instructions.  Reported: lanes/clk/SIMD (32 = full rate, a wave64
// instruction issues over 2 cycles) for VALU; LDS-pipe cycles per wave-instruction per CU for the DS forms the
// fused kernel uses (conflict-free, bank-conflicting and same-address address patterns).
//   hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 2048

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define VALU_KERNEL(NAME, ASM)                                                              \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {             \
        uint32_t r[16];                                                                     \
        for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;                   \
        uint32_t a = seed | 1u, b = threadIdx.x * 2654435761u;                              \
        for (int it = 0; it < ITERS; it++) {                                                \
            REP16(ASM)                                                                      \
        }                                                                                   \
        uint32_t s = 0;                                                                     \
        for (int i = 0; i < 16; i++) s ^= r[i];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                     \
    }

#define A_ADD(i) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_AND(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(r[i]) : "v"(b));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %1, 8, 7" : "=v"(r[i]) : "v"(b));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_SAD(i) asm volatile("v_sad_u8 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %1, %2, 8" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(a), "v"(b) : "vcc");
#define A_CMP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define A_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(*(uint64_t*)&r[(i) & 14]) : "v"(*(uint64_t*)&a));
#define A_MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r[i]) : "v"(b));
#define A_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r[i]) : "v"(a), "v"(b));

VALU_KERNEL(k_add, A_ADD)
VALU_KERNEL(k_and, A_AND)
VALU_KERNEL(k_lshl, A_LSHL)
VALU_KERNEL(k_bfe, A_BFE)
VALU_KERNEL(k_lshladd, A_LSHLADD)
VALU_KERNEL(k_add3, A_ADD3)
VALU_KERNEL(k_andor, A_ANDOR)
VALU_KERNEL(k_bcnt, A_BCNT)
VALU_KERNEL(k_sad, A_SAD)
VALU_KERNEL(k_dot4, A_DOT4)
VALU_KERNEL(k_mul24, A_MUL24)
VALU_KERNEL(k_mad24, A_MAD24)
VALU_KERNEL(k_mullo, A_MULLO)
VALU_KERNEL(k_alignbit, A_ALIGNBIT)
VALU_KERNEL(k_perm, A_PERM)
VALU_KERNEL(k_cndmask, A_CNDMASK)
VALU_KERNEL(k_cmp, A_CMP)
VALU_KERNEL(k_movdpp, A_MOVDPP)
VALU_KERNEL(k_adddpp, A_ADDDPP)

typedef void (*valu_fn)(uint32_t*, uint32_t);

static void run_valu(const char* name, valu_fn f, uint32_t* d) {
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_insts = (double)blocks * (threads / 64) * ITERS * 16;
    const double per_simd = wave_insts / 1024.0;                  // 256 CUs x 4 SIMDs
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd;              // cycles per wave-instruction per SIMD at 2.4 GHz
    printf("%-22s %8.3f ms  %6.2f cycles/wave-inst/SIMD (at 2.4 GHz)  %6.1f T lane-ops/s\n", name, ms, cyc,
           wave_insts * 64 / (ms * 1e-3) / 1e12);
}

// ---- LDS: MODE selects the address pattern of the 64 lanes -------------------------------------------------
//  0 consecutive (lane * size)        1 random bank (hash of lane and iteration)      2 one address for all lanes
//  3 16 distinct addresses (4 lanes each)   4 stride of 40 dwords (the [cycle][class] rows of the Stats counters)
template <int OP, int MODE>
__global__ void __launch_bounds__(1024) k_lds(uint32_t* out, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    const uint32_t base = wave * 1024 * 4 % 32768;   // byte offset of the wave's own 4 KB window (32 KB used)
    for (int it = 0; it < ITERS; it++) {
        uint32_t idx;
        if (MODE == 0) idx = lane;
        else if (MODE == 1) idx = ((lane * 2654435761u + it * 40503u + seed) >> 7) & 511u;
        else if (MODE == 2) idx = 5;
        else if (MODE == 3) idx = (lane & 15) * 33;
        else idx = (lane * 40 + it) & 511u;
        const uint32_t sz = (OP == 1 || OP == 4) ? 8 : (OP == 5 ? 16 : 4);
        const uint32_t addr = base + ((idx * sz) & 4095u & ~(sz - 1));
        if (OP == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(lane) : "memory");
        else if (OP == 1) { uint64_t v = lane; asm volatile("ds_add_u64 %0, %1" : : "v"(addr), "v"(v) : "memory"); }
        else if (OP == 2) { uint32_t r; asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(lane) : "memory"); acc += r; }
        else if (OP == 3) { uint32_t r; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); acc += r; }
        else if (OP == 4) { uint64_t r; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); acc += (uint32_t)r; }
        else if (OP == 5) { uint32_t r0, r1, r2, r3; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(4))) uint32_t*)&r0) : "v"(addr) : "memory"); acc += r0; (void)r1; (void)r2; (void)r3; }
        else if (OP == 6) { uint32_t r; asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory"); acc += r; }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int OP, int MODE>
static void run_lds(const char* name, uint32_t* d) {
    const int blocks = 256, threads = 1024;  // one 16-wave workgroup per CU, like the fused kernel
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_lds<OP, MODE>), dim3(blocks), dim3(threads), 65536, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k_lds<OP, MODE>), dim3(blocks), dim3(threads), 65536, 0, d, 12345u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double per_cu = 16.0 * ITERS;                            // wave-instructions per CU
    printf("%-44s %8.3f ms  %7.2f cycles per wave-instruction per CU (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 2);
    run_valu("v_add_u32", k_add, d);
    run_valu("v_and_b32", k_and, d);
    run_valu("v_lshlrev_b32", k_lshl, d);
    run_valu("v_bfe_u32", k_bfe, d);
    run_valu("v_lshl_add_u32", k_lshladd, d);
    run_valu("v_add3_u32", k_add3, d);
    run_valu("v_and_or_b32", k_andor, d);
    run_valu("v_bcnt_u32_b32", k_bcnt, d);
    run_valu("v_sad_u8", k_sad, d);
    run_valu("v_dot4_u32_u8", k_dot4, d);
    run_valu("v_mul_u32_u24", k_mul24, d);
    run_valu("v_mad_u32_u24", k_mad24, d);
    run_valu("v_mul_lo_u32", k_mullo, d);
    run_valu("v_alignbit_b32", k_alignbit, d);
    run_valu("v_perm_b32", k_perm, d);
    run_valu("v_cndmask_b32", k_cndmask, d);
    run_valu("v_cmp_lt_u32", k_cmp, d);
    run_valu("v_mov_b32_dpp", k_movdpp, d);
    run_valu("v_add_u32_dpp", k_adddpp, d);
    run_lds<0, 0>("ds_add_u32  consecutive", d);
    run_lds<0, 1>("ds_add_u32  random", d);
    run_lds<0, 2>("ds_add_u32  one address", d);
    run_lds<0, 3>("ds_add_u32  16 addresses x 4 lanes", d);
    run_lds<1, 0>("ds_add_u64  consecutive", d);
    run_lds<1, 1>("ds_add_u64  random", d);
    run_lds<1, 2>("ds_add_u64  one address", d);
    run_lds<1, 3>("ds_add_u64  16 addresses x 4 lanes", d);
    run_lds<1, 4>("ds_add_u64  stride 40 dwords", d);
    run_lds<2, 0>("ds_add_rtn_u32 consecutive (+wait)", d);
    run_lds<2, 2>("ds_add_rtn_u32 one address (+wait)", d);
    run_lds<3, 0>("ds_read_b32 consecutive (+wait)", d);
    run_lds<3, 1>("ds_read_b32 random (+wait)", d);
    run_lds<3, 2>("ds_read_b32 broadcast (+wait)", d);
    run_lds<4, 0>("ds_read_b64 consecutive (+wait)", d);
    run_lds<4, 1>("ds_read_b64 random (+wait)", d);
    run_lds<5, 0>("ds_read_b128 consecutive (+wait)", d);
    run_lds<5, 1>("ds_read_b128 random (+wait)", d);
    run_lds<5, 2>("ds_read_b128 broadcast (+wait)", d);
    run_lds<6, 1>("ds_read_u8 random (+wait)", d);
    return 0;
}
