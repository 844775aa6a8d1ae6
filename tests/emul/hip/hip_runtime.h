// tests/emul/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the product's .hip sources be compiled as plain host C++ (amdclang++ -x c++) so that the kernels'
// index arithmetic can be exercised against the oracle in the GPU-less build container before spending
// MI355X time.  A launch becomes a loop nest over (block, thread).  Never shipped, never loaded by cotnet_amd/.
#pragma once
#include <cstdint>
#include <cstring>
#include <initializer_list>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 blockIdx, threadIdx;
inline dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)        \
    do {                                                                   \
        const dim3 g_ = (grid), b_ = (block);                              \
        gridDim = g_;                                                      \
        blockDim = b_;                                                     \
        _Pragma("omp parallel for schedule(static)")                       \
        for (long bx_ = 0; bx_ < (long)g_.x; ++bx_) {                      \
            blockIdx = dim3((unsigned)bx_, 0, 0);                          \
            for (unsigned tx_ = 0; tx_ < b_.x; ++tx_) {                    \
                threadIdx = dim3(tx_, 0, 0);                               \
                kernel(__VA_ARGS__);                                       \
            }                                                              \
        }                                                                  \
    } while (0)
