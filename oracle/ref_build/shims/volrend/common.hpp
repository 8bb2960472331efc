/* Stand-in for the cmake-generated volrend/common.hpp (common.hpp.in) with the
 * CUDA backend selected, as a CUDA build of the reference sees it.
 * ORACLE / test infrastructure only. */
#pragma once
#define VOLREND_VERSION_MAJOR 0
#define VOLREND_VERSION_MINOR 0
#define VOLREND_VERSION_PATCH 0
#define VOLREND_CUDA
#include <cuda_runtime.h>
/* the __CUDACC__ branch of common.hpp.in:13-17 */
#define VOLREND_COMMON_FUNCTION __host__ __device__ __inline__
#define VOLREND_RESTRICT __restrict__
#define VOLREND_MIN(a, b) min(a, b)
#define VOLREND_MAX(a, b) max(a, b)
