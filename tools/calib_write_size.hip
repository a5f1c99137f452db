// Calibration of the TCC WRITE_SIZE counter on gfx950 (MI355X_MICROARCH.md lists it as uncalibrated): kernels with a
// KNOWN number of stored bytes / device-scope atomics, run under `rocprofv3 --pmc WRITE_SIZE` (tools/calib_write_size.sh).
//   k_store16 / k_store32   every thread stores 16 / 32 contiguous bytes once (coalesced, whole cache lines)
//   k_store4_strided        every thread stores 4 bytes at a 32-byte stride (partial lines, like a flag field)
//   k_atomic64              N device-scope 64-bit atomicAdd spread over M addresses (the packed hypothesis counters)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_store16(uint4* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(1, 2, 3, 4); }
__global__ void k_store32(uint4* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { p[2 * i] = make_uint4(1, 2, 3, 4); p[2 * i + 1] = make_uint4(5, 6, 7, 8); } }
__global__ void k_store4_strided(uint32_t* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[8 * i] = 7u; }
__global__ void k_atomic64(unsigned long long* p, size_t n, size_t m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&p[(i * 2654435761ull) % m], (1ull << 32) | 1ull);
}

int main() {
    const size_t N = 8u << 20;              // 8 Mi threads
    void* buf = nullptr;
    if (hipMalloc(&buf, N * 32) != hipSuccess || hipMemset(buf, 0, N * 32) != hipSuccess) return 1;
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_store16, grid, block, 0, 0, (uint4*)buf, N);          // 128 MiB
    hipLaunchKernelGGL(k_store32, grid, block, 0, 0, (uint4*)buf, N);          // 256 MiB
    hipLaunchKernelGGL(k_store4_strided, grid, block, 0, 0, (uint32_t*)buf, N);// 32 MiB of payload, 256 MiB of lines touched
    hipLaunchKernelGGL(k_atomic64, grid, block, 0, 0, (unsigned long long*)buf, N, (size_t)128000);   // 8 Mi atomics, 1 MB of counters
    hipLaunchKernelGGL(k_atomic64, grid, block, 0, 0, (unsigned long long*)buf, N, (size_t)N);        // 8 Mi atomics, 64 MiB of counters
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("threads %zu\n", N);
    (void)hipFree(buf);
    return 0;
}
