// fetch_calib.hip - what rocprofv3's FETCH_SIZE reports on gfx950 for a streaming read of a KNOWN byte count, by load width.
// The guide (MI355X_MICROARCH.md, "HBM") calibrates the 16-byte-per-lane case only (FETCH_SIZE = half the bytes) and calls the
// other widths uncalibrated; the Stats kernel's form 5 reads its rows as 8-byte and 4-byte loads per lane.  Each kernel reads
// every byte of a 1 GiB buffer (four times the Infinity Cache) exactly once, coalesced, consecutive lanes = consecutive words.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o pmc -- ./fetch_calib     (FETCH_SIZE is in KiB)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename T>
__global__ void __launch_bounds__(256) stream_read(const T* __restrict__ src, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t* w = (const uint32_t*)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;   // (keeps the loads)
}

// the Stats kernel's pattern: a lane = (row, 16-byte column) over rows of 152 bytes - two 8-byte loads and the dword in front
__global__ void __launch_bounds__(256) rows_read(const uint32_t* __restrict__ src, size_t rows, uint32_t* out) {
    uint32_t acc = 0;
    const size_t lanes = rows * 10;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < lanes; i += (size_t)gridDim.x * blockDim.x) {
        const size_t u = i / 10, h = i % 10;
        const size_t qd = u * 38 + 4 * h;
        const uint64_t a = *(const uint64_t*)(src + qd);
        const uint64_t b = *(const uint64_t*)(src + qd + (h < 9 ? 2 : 0));
        const uint32_t c = src[qd - (h > 0 ? 1 : 0)];
        acc ^= (uint32_t)a ^ (uint32_t)(a >> 32) ^ (uint32_t)b ^ (uint32_t)(b >> 32) ^ c;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf = nullptr;
    uint32_t* out = nullptr;
    if (hipMalloc(&buf, bytes + 256) != hipSuccess || hipMalloc((void**)&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, bytes + 256);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto timed = [&](const char* name, auto launch) {
        launch();   // (warm-up: page tables)
        (void)hipEventRecord(e0, 0);
        launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, (double)bytes / ms / 1e6);
    };
    timed("stream_read<uint4>  (16 B)", [&] { hipLaunchKernelGGL(stream_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out); });
    timed("stream_read<uint2>  (8 B)", [&] { hipLaunchKernelGGL(stream_read<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out); });
    timed("stream_read<uint>   (4 B)", [&] { hipLaunchKernelGGL(stream_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out); });
    timed("rows_read (152-byte rows)", [&] { hipLaunchKernelGGL(rows_read, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 152, out); });
    (void)hipDeviceSynchronize();
    printf("bytes per launch: %zu (every kernel above is launched twice)\n", bytes);
    return 0;
}
