// mfma_common.h -- pieces shared by the MFMA convolution kernels (conv1x1.hip, conv3x3g.hip): operand types, the one
// MFMA primitive, wide (possibly unaligned) accesses with a wave-uniform fast path, wave -> work-item mapping.
//
// MFMA operand maps used (v_mfma_f32_16x16x32_bf16, D = A*B + C, one wave):
//       A: lane l holds A[i = l&15][k = 8*(l>>4) .. +7]          (8 bf16, K-contiguous)
//       B: lane l holds B[k = 8*(l>>4) .. +7][j = l&15]
//     C/D: lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3
// (sources: cdna_hip_programming.md "Fragment layout" for C/D; the K split -- four lane blocks of 8 consecutive k -- is
// what composable_kernel's mfma_type<mfma_f32_16x16x32bf16> encodes: num_input_blks = 4, k_per_blk = 8, is_k_reduction,
// /opt/rocm/include/ck/tensor_operation/gpu/warp/xdlops_gemm.hpp.  Any K split that is the same for A and B gives the
// same product, so only the row / column / accumulator maps can be wrong; tests/emul implements exactly these maps.)
#pragma once
#include "cot_common.h"

namespace cot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#ifndef COT_MFMA_16X16X32_BF16  // (tests/emul pre-defines this primitive for its host build)
#define COT_MFMA_16X16X32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

template <int V> struct BFVec {
    typedef __attribute__((ext_vector_type(V))) __bf16 type;
};

// dst[0..V) = p[0..V) for the first `cnt` elements, zero beyond (cnt may be <= 0 or > V).
// `wide` is a WAVE-UNIFORM promise that reading all V elements at p is inside the tensor for every lane (scalar branch:
// the one wide -- possibly unaligned -- access is not entangled with the element-wise path).  With `wide`, elements past
// `cnt` hold whatever follows in memory unless `zero_tail` asks for them to be cleared (by selection, never by
// multiplication: they may be Inf/NaN).  AL = what is known about p's alignment.
template <int V, int AL>
__device__ __forceinline__ void load_piece(bf16_t (&dst)[V], const bf16_t* p, int cnt, bool wide, bool zero_tail = true) {
    if (wide) {
        typename BFVec<V>::type t;
        __builtin_memcpy(&t, __builtin_assume_aligned(p, AL), sizeof(t));
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = t[i];
        if (zero_tail) {
#pragma unroll
            for (int i = 0; i < V; ++i) dst[i] = (i < cnt) ? dst[i] : (bf16_t)0.0f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) dst[i] = (i < cnt) ? p[i] : (bf16_t)0.0f;
    }
}
// p[0..min(cnt,V)) = src: one wide store for lanes that own all V elements, element-wise for the lane at the row's end
template <int V, int AL>
__device__ __forceinline__ void store_piece(bf16_t* p, const bf16_t (&src)[V], int cnt) {
    if (cnt >= V) {
        typename BFVec<V>::type t;
#pragma unroll
        for (int i = 0; i < V; ++i) t[i] = src[i];
        __builtin_memcpy(__builtin_assume_aligned(p, AL), &t, sizeof(t));
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i)
            if (i < cnt) p[i] = src[i];
    }
}
// wave-uniform value -> SGPR (so that conditions on it become scalar branches)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int64_t wave_work_id(int xcd_remap) {
    unsigned b = blockIdx.x;
    const unsigned nblk = gridDim.x;
    if (xcd_remap && (nblk & 7u) == 0) b = (b & 7u) * (nblk >> 3) + (b >> 3);  // consecutive ids -> same XCD
    return (int64_t)b * (blockDim.x >> 6) + (threadIdx.x >> 6);
}

// grid size (4 waves per workgroup) rounded up to a multiple of 8 so that the XCD-aware order applies
__host__ inline int wave_grid_blocks(int64_t waves) {
    const int64_t b = (waves + 3) / 4;
    return (int)((b + 7) / 8 * 8);
}

}  // namespace cot
