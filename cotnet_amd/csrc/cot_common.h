// cot_common.h -- shared device helpers for the cotnet_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/cotnet_amd.h"

namespace cot {

typedef __bf16 bf16_t;   // native bf16: conversions lower to v_cvt_pk_bf16_f32 on gfx950 (RNE)
typedef _Float16 f16_t;

// accumulate type: fp32 for fp32/bf16/fp16 storage, fp64 for fp64 (the reference accumulates in the
// storage type, cupy_layers/aggregation_zeropad.py:31; fp32 accumulation of bf16 data is the documented
// extension)
template <typename T> struct AccOf { typedef float type; };
template <> struct AccOf<double> { typedef double type; };

template <typename T> __device__ __forceinline__ typename AccOf<T>::type ld(const T* p) {
    return (typename AccOf<T>::type)(*p);
}
template <typename T, typename A> __device__ __forceinline__ void st(T* p, A v) { *p = (T)v; }

// POD vector of V elements with natural (V*sizeof(T)) alignment so the compiler emits one wide access
template <typename T, int V> struct alignas(sizeof(T) * V) Vec { T v[V]; };
// 7 elements with ELEMENT alignment (7 x 7 planes: rows of 49 elements start on odd multiples of 2 bytes): the access is split by
// the compiler into the widest pieces an unaligned address allows (global memory takes unaligned dword accesses on gfx950)
template <typename T> struct alignas(sizeof(T)) Vec<T, 7> { T v[7]; };

template <typename T, int V> __device__ __forceinline__ Vec<T, V> ldv(const T* p) {
    return *reinterpret_cast<const Vec<T, V>*>(p);
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const Vec<T, V>& v) {
    *reinterpret_cast<Vec<T, V>*>(p) = v;
}

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4): per-lane source address, destination is the
// WAVE-UNIFORM LDS address `lds_wave_base` + lane*16.  No VGPR round trip; completion is tracked by vmcnt and
// drained by the s_waitcnt the compiler places before the next __syncthreads().
#ifndef COT_ASYNC_COPY16  // (tests/emul pre-defines this one primitive for its host build)
#define COT_ASYNC_COPY16(gptr, lds_wave_base)                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#endif

// 1 / x by v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division sequence: the sigmoid inside SiLU and its derivative sit in
// HBM-bound element loops that the division made VALU-bound (radix_gap_t_bn 56 x 56: 24 us for a 12 us read, profiles/r06_bn_tail_kernels.log).
// ONE definition for every kernel that forms silu(z), so that a tensor activated by one kernel and re-formed by another agrees bit for bit.
#ifndef COT_RCP  // (tests/emul pre-defines it for its host build)
#define COT_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ float silu_fwd(float z) { return z * COT_RCP(1.f + __expf(-z)); }

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace cot

// host-side error plumbing (thread-local message, int status codes from cotnet_amd.h)
namespace cot {
int set_error(int code, const char* fmt, ...);
int check_launch(const char* what);

// optional per-launch device timing (cot_profile_begin/_end): the start/stop events are attached to the kernel
// dispatch itself (hipExtLaunchKernelGGL), so the interval is the kernel's execution, not host launch gaps.
namespace prof {
bool enabled();
bool enabled_for(const char* kernel);  // enabled() and (all kernels requested or an aggregation kernel)
void begin_launch(hipEvent_t* e0, hipEvent_t* e1);
void end_launch(const char* name, hipEvent_t e0, hipEvent_t e1);
}  // namespace prof
}  // namespace cot

// Dry run (cot_set_tuning key 26 = 1): nothing is launched and no HIP API is called; every would-be launch is appended to a
// thread-local log instead -- the launcher's instantiation (__PRETTY_FUNCTION__ carries the template arguments), the kernel's
// source name, grid and block -- which cot_launch_log() hands out.  This is how the dispatch table (which kernel variant
// every layer of the BASELINE configurations takes) is pinned by a test that needs no GPU (tests/test_dispatch_table.py).
namespace cot {
extern int g_dry_run;
void dry_note(const char* where, const char* kernel, dim3 grid, dim3 block, size_t shmem);
}  // namespace cot

#define COT_LAUNCH(KERNEL, GRID, BLOCK, SHMEM, STREAM, ...)                                             \
    do {                                                                                                \
        if (cot::g_dry_run) {                                                                           \
            cot::dry_note(__PRETTY_FUNCTION__, #KERNEL, GRID, BLOCK, SHMEM);                            \
        } else if (cot::prof::enabled_for(#KERNEL)) {                                                   \
            hipEvent_t e0_, e1_;                                                                        \
            cot::prof::begin_launch(&e0_, &e1_);                                                        \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, e0_, e1_, 0, __VA_ARGS__);        \
            cot::prof::end_launch(#KERNEL, e0_, e1_);                                                   \
        } else {                                                                                        \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, STREAM, __VA_ARGS__);                        \
        }                                                                                               \
    } while (0)
