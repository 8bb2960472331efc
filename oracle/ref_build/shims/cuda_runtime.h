/*
 * Host shim for <cuda_runtime.h> (ORACLE / test infrastructure only).
 * Lets g++ compile the reference's CUDA device code (src/cuda/volrend.cu +
 * include/volrend/cuda/rt_core.cuh) unmodified, from where it lies under
 * /root/reference, as ordinary host C++: one "thread" = one call with
 * blockIdx.x = pixel index, blockDim.x = 1, threadIdx.x = 0.
 */
#ifndef VR_REF_SHIM_CUDA_RUNTIME_H_
#define VR_REF_SHIM_CUDA_RUNTIME_H_
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __host__
#define __device__
#define __global__

struct VrShimDim3 {
    unsigned x = 0, y = 0, z = 0;
};
static thread_local VrShimDim3 blockIdx, threadIdx;
static thread_local VrShimDim3 blockDim;

typedef int cudaError_t;
enum { cudaSuccess = 0 };
typedef void* cudaStream_t;

/* a pitched 2-D byte image standing in for both cudaArray and its surface */
struct VrShimSurface {
    uint8_t* base;
    size_t pitch;   /* bytes per row */
    int width_bytes;
    int height;
};
typedef VrShimSurface* cudaArray_t;
typedef VrShimSurface* cudaSurfaceObject_t;
enum cudaSurfaceBoundaryMode { cudaBoundaryModeZero = 0 };
enum cudaResourceType { cudaResourceTypeArray = 0 };
struct cudaResourceDesc {
    cudaResourceType resType;
    struct {
        struct {
            cudaArray_t array;
        } array;
    } res;
};
static inline cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t* out,
                                                  const cudaResourceDesc* d) {
    *out = d->res.array.array;
    return cudaSuccess;
}
static inline cudaError_t cudaMalloc(void* pp, size_t sz) {
    *reinterpret_cast<void**>(pp) = std::malloc(sz);
    return cudaSuccess;
}
template <typename T>
static inline cudaError_t cudaMalloc(T** pp, size_t sz) {
    *pp = static_cast<T*>(std::malloc(sz));
    return cudaSuccess;
}
static inline cudaError_t cudaFree(void* p) {
    std::free(p);
    return cudaSuccess;
}

template <typename T>
static inline void surf2Dread(T* out, cudaSurfaceObject_t s, int xbytes, int y,
                              cudaSurfaceBoundaryMode) {
    if (!s || xbytes < 0 || y < 0 || xbytes + (int)sizeof(T) > s->width_bytes || y >= s->height) {
        std::memset(out, 0, sizeof(T)); /* cudaBoundaryModeZero */
        return;
    }
    std::memcpy(out, s->base + (size_t)y * s->pitch + xbytes, sizeof(T));
}
template <typename T>
static inline void surf2Dwrite(T v, cudaSurfaceObject_t s, int xbytes, int y,
                               cudaSurfaceBoundaryMode) {
    if (!s || xbytes < 0 || y < 0 || xbytes + (int)sizeof(T) > s->width_bytes || y >= s->height)
        return; /* squelched */
    std::memcpy(s->base + (size_t)y * s->pitch + xbytes, &v, sizeof(T));
}

/* CUDA's global float min/max: PTX min.f32/max.f32 (NaN loses, -0 < +0) */
static inline float min(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline float max(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
/* cos/sin(float): <math.h> in C++ already exposes std::cos(float)/std::sin(float)
 * globally, i.e. the float overloads -- the same overloads CUDA resolves to. */
#include <math.h>

#endif
