// =============================================================================
// oracle/ref_shim_cuda/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A host-only stand-in for the slice of the CUDA runtime the reference uses (dataArray.h, sparsematrix.cc,
// cudawrapper.cu), so that its CUDA path can be compiled by g++ and executed on the CPU, kernel by kernel:
//   * vector types float2/3/4, int2, uint3, dim3 and their make_* constructors (helper_math.h of the reference builds
//     its operators on these);
//   * "device memory" = host memory (cudaMallocPitch / cudaMemcpy2D / cudaFree on malloc'd blocks);
//   * __global__ / __device__ / __host__ expand to nothing; blockIdx / threadIdx / blockDim / gridDim are thread-local
//     variables set by L3D_SHIM_LAUNCH, which runs the grid as nested loops -- one "thread" after the other.  The
//     reference's kernels use neither shared memory nor barriers nor atomics, and every thread writes only its own
//     outputs, so sequential execution gives what any parallel schedule gives;
//   * oracle/Makefile turns `K <<< grid, block >>> (args)` of cudawrapper.cu into `L3D_SHIM_LAUNCH(K, grid, block)(args)`
//     with sed while piping the file into the compiler (nothing of the reference is stored in this repository).
// Arithmetic note: the kernels are float code; g++ is told -ffp-contract=off, nvcc would contract a*b+c into FMAs.
// The diffusion kernels only multiply-accumulate in one place (mul += d1.z*d2.z); tests compare at 1e-4 and report
// the largest difference.
// =============================================================================
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct uint4 { unsigned int x, y, z, w; };
typedef unsigned int uint;
typedef unsigned short ushort;
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int a = 1, unsigned int b = 1, unsigned int c = 1) : x(a), y(b), z(c) {}
};
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
static inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
static inline int3 make_int3(int x, int y, int z) { int3 r = {x, y, z}; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r = {x, y, z, w}; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = {x, y}; return r; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { uint3 r = {x, y, z}; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r = {x, y, z, w}; return r; }

// (fminf / fmaxf / rsqrtf for host compilation come from the reference's own helper_math.h)

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline const char* cudaGetErrorString(cudaError_t) { return "host shim"; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? cudaSuccess : 2; }
static inline cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t height) {
    *pitch = (width_bytes + 511) & ~(size_t)511;                      // cudaMallocPitch pads rows; keep that visible
    *p = std::calloc((*pitch) * (height ? height : 1), 1);
    return *p ? cudaSuccess : 2;
}
static inline cudaError_t cudaFree(void* p) { std::free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { std::memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width_bytes, size_t height,
                                       cudaMemcpyKind) {
    for (size_t r = 0; r < height; ++r) std::memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width_bytes);
    return cudaSuccess;
}
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { std::memset(d, v, n); return cudaSuccess; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }

// ---- kernel launches as loops over the grid ----
extern thread_local uint3 blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;
template <class F>
struct L3DShimLauncher {
    F f; dim3 g, b;
    template <class... A>
    void operator()(A... a) const {
        gridDim = g; blockDim = b;
        for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx)
            for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
                blockIdx = make_uint3(bx, by, bz); threadIdx = make_uint3(tx, ty, tz);
                f(a...);
            }
    }
};
template <class F>
static inline L3DShimLauncher<F> l3d_shim_launcher(F f, dim3 g, dim3 b) { return L3DShimLauncher<F>{f, g, b}; }
#define L3D_SHIM_LAUNCH(K, G, B) l3d_shim_launcher(K, G, B)
