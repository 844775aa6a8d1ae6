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
#include <type_traits>

#include "cot_common.h"

namespace cot {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#ifndef COT_MFMA_16X16X32_BF16  // (tests/emul pre-defines this primitive for its host build)
#define COT_MFMA_16X16X32_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
// COT_KEEP_PACKED(u): the 32-bit register `u` is "redefined" here as far as the optimiser knows, so nothing computed from
// it earlier (say its two bf16 halves widened to fp32) can be kept alive in its place.  Used where a tile is held in
// registers across a reduction: only the packed form should occupy registers while the workgroup waits.
// COT_WAIT_LOADS(): s_waitcnt vmcnt(0) -- closes an edge-of-tensor path that issues a data-dependent number of loads
// (predicated element loads).  The compiler places its own waits for the worst case over all paths; a path that might
// have issued no loads at all makes it wait, on the common path too, for loads it could have left in flight.  Ending
// the rare path with "everything has arrived" takes it out of that calculation.
#ifndef COT_KEEP_PACKED  // (tests/emul pre-defines both as no-ops)
#define COT_KEEP_PACKED(u) asm volatile("" : "+v"(u))
#define COT_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0f70)  // gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt = no wait
#endif

template <int V> struct BFVec {
    typedef __attribute__((ext_vector_type(V))) __bf16 type;
};

// ---- pieces: V consecutive bf16 of one row, held PACKED (two per 32-bit register) from the load to the point of use.
// Why packed matters: a piece arrives either by one wide load or, at the edges of a tensor, element by element; if the two
// arms met in an unpacked form the wide arm would have to split its registers right after the load -- i.e. WAIT for it --
// and a ring of loads in flight would degenerate into one round trip per 16 bytes.  So nothing here touches the loaded
// bits: clearing a row's tail (mask_packed) and picking halves (packed_pair, packed_lo/hi) happen where the data is used.
//
// q[0..V/2) = p[0..V): `wide` is a WAVE-UNIFORM promise that reading all V elements at p is inside the tensor for every
// lane (scalar branch); elements past a row's end then hold whatever follows in memory.  Without it, elements at index
// >= cnt are not read and come back as zero bits (cnt may be <= 0 or > V).  AL = what is known about p's alignment.
template <int V, int AL>
__device__ __forceinline__ void load_packed(uint32_t (&q)[V / 2], const bf16_t* p, int cnt, bool wide) {
    if (wide) {
        __builtin_memcpy(q, __builtin_assume_aligned(p, AL), 2 * V);
    } else {
#pragma unroll
        for (int i = 0; i < V / 2; ++i) {
            uint16_t lo = 0, hi = 0;
            if (2 * i < cnt) __builtin_memcpy(&lo, p + 2 * i, 2);
            if (2 * i + 1 < cnt) __builtin_memcpy(&hi, p + 2 * i + 1, 2);
            q[i] = (uint32_t)lo | ((uint32_t)hi << 16);
        }
        COT_WAIT_LOADS();
    }
}
// q = t[off .. off+V) with every element bounds-checked against the tensor [0, elems) (zero bits outside): the edge-of-
// tensor path of the 3x3 kernels, whose shifted windows start before / end after the tensor for the first / last waves
template <int V>
__device__ __forceinline__ void load_packed_checked(uint32_t (&q)[V / 2], const bf16_t* t, int64_t off, int64_t elems) {
#pragma unroll
    for (int i = 0; i < V / 2; ++i) {
        uint16_t lo = 0, hi = 0;
        if (off + 2 * i >= 0 && off + 2 * i < elems) __builtin_memcpy(&lo, t + off + 2 * i, 2);
        if (off + 2 * i + 1 >= 0 && off + 2 * i + 1 < elems) __builtin_memcpy(&hi, t + off + 2 * i + 1, 2);
        q[i] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    COT_WAIT_LOADS();
}
// per-lane variant: wide when this lane owns all V elements (cnt >= V), element-wise otherwise (divergent branch)
template <int V, int AL> __device__ __forceinline__ void load_packed_lane(uint32_t (&q)[V / 2], const bf16_t* p, int cnt) {
    load_packed<V, AL>(q, p, cnt, cnt >= V);
}
// clear elements at index >= cnt (by selection: they may be Inf/NaN bit patterns of a neighbouring row)
template <int V> __device__ __forceinline__ void mask_packed(uint32_t (&q)[V / 2], int cnt) {
#pragma unroll
    for (int i = 0; i < V / 2; ++i) q[i] &= (2 * i < cnt ? 0x0000ffffu : 0u) | (2 * i + 1 < cnt ? 0xffff0000u : 0u);
}
// element e of a packed piece as fp32 (bf16 -> fp32 is a shift)
__device__ __forceinline__ float packed_lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float packed_hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float packed_get(const uint32_t* q, int e) {
    return (e & 1) ? packed_hi(q[e >> 1]) : packed_lo(q[e >> 1]);
}
// one register holding element c of piece `a` (low half) and element c of piece `b` (high half): the in-register
// transposition step of the NCHW kernels (one v_perm_b32)
__device__ __forceinline__ uint32_t packed_pair(const uint32_t* a, const uint32_t* b, int c) {
    const uint32_t ua = a[c >> 1], ub = b[c >> 1];
    return (c & 1) ? ((ua >> 16) | (ub & 0xffff0000u)) : ((ua & 0x0000ffffu) | (ub << 16));
}
__device__ __forceinline__ bf16x8_t packed_as_frag(const uint32_t (&q)[4]) {
    bf16x8_t f;
    __builtin_memcpy(&f, q, 16);
    return f;
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
// ---- D-slot register ring over `nsteps` steps: step s is multiplied from slot s % D while the loads of steps s+1 ..
// s+D-1 are in flight.  load_step(slot, step) issues a step's loads, multiply_slot(slot) consumes them; both are called
// with compile-time slot numbers (after unrolling), so the slots are plain registers.
// The steady state -- every step issues a load D-1 steps ahead -- is its own loop, apart from the drain at the end: in
// one shared body the compiler has to place waits that are also right for the drain (nothing newer in flight), which
// in the steady state means waiting for the loads just issued, i.e. no overlap at all.
template <int D, typename LoadStep, typename MultiplySlot>
__device__ __forceinline__ void ring_loop(int nsteps, LoadStep&& load_step, MultiplySlot&& multiply_slot) {
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nsteps) load_step(d, d);
    int s = 0;
    for (; s + 2 * D - 2 < nsteps; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_step((d + D - 1) % D, s + d + D - 1);
            multiply_slot(d);
        }
    }
    for (; s < nsteps; s += D) {  // the last (up to 2D-2) steps: s is a multiple of D, so the slots line up
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (s + d < nsteps) {
                if (s + d + D - 1 < nsteps) load_step((d + D - 1) % D, s + d + D - 1);
                multiply_slot(d);
            }
        }
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
