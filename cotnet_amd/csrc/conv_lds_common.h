// conv_lds_common.h -- pieces shared by the LDS-DMA pipelined convolution kernels (conv_lds.hip, conv_lds2.hip): the
// hand-counted copy / wait / barrier primitives, the epilogue through LDS, the 1x1 kernels' argument block.
#pragma once
#include <atomic>

#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

// Opt a kernel in to more than the default 64 KB of dynamic LDS (160 KB per CU on gfx950), once per DEVICE: `raised` is the
// calling launch function's own static bit mask (one per kernel instantiation).  Returns false when the runtime refuses --
// the caller then reports "not covered" and the previous kernel generation runs instead of an opaque launch failure.
inline bool raise_dynamic_lds_once(std::atomic<uint32_t>& raised, const void* func) {
    if (g_dry_run) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    if ((raised.load(std::memory_order_relaxed) >> dev) & 1u) return true;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    raised.fetch_or(1u << dev, std::memory_order_relaxed);
    return true;
}

// ---- hand-counted LDS-DMA pipeline primitives ----------------------------------------------------------------------
// The compiler's own bookkeeping drains every outstanding LDS-DMA (s_waitcnt vmcnt(0)) in front of the first LDS read it
// cannot prove disjoint from the DMA's destination -- i.e. in front of every K step's fragment reads -- which turns a
// multi-stage pipeline into load-wait-compute (seen in the first version's ISA).  So the copies are issued from an asm
// statement the compiler does not count (cdna_hip_programming.md 5.7: M0 = wave-uniform LDS base, set and restored in the
// same statement), completion is tracked by hand with counted s_waitcnt vmcnt(N), and the workgroup barrier is the raw
// s_barrier behind an lgkmcnt(0) (LDS reads of the step done; nothing else pending).
#ifndef COT_GLDS16  // (tests/emul pre-defines the three primitives for its host build)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds_wave_base));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
// ds_read_b64_tr_b16: every lane passes the address of 4 consecutive bf16 (8-byte aligned); within each group of 16 lanes,
// lane L (0..15) pointing at row (L>>2), columns 4*(L&3).. of a 4 x 16 block receives rows 0..3 of column L -- a free 4x4
// transposition (mapping confirmed on the MI355X by scripts/ubench_trprobe.py).  Two of them build the 8-deep K fragment of
// a K-strided operand that otherwise takes eight 2-byte reads and four permutes.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
}
// Scalar-base form (round 3): source = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset, destination =
// wave-uniform LDS byte address (M0) + lane*16.  A K step then moves only the SCALAR base -- the per-lane offsets are loop
// constants -- and M0 is written by the compiler's own scalar code ("{m0}" binds the operand to the register; the s_nop covers
// the M0-write -> LDS-DMA hazard the compiler cannot see inside an asm statement).  The pointer form above costs ~30
// instructions per copy in 64-bit vector address arithmetic, readfirstlanes and M0 save/restore: with two waves per SIMD
// that WAS the K step of the deep layers (profiles/r03_*: step time independent of the tile's size).
// Probed on the MI355X (scripts/probe3.py): aligned, 2-byte-aligned and scattered sources land as expected.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "{m0}"(lds_addr) : "memory");
}
// LDS byte address of a pointer into the dynamic LDS array (what M0 / DS instructions take)
#define COT_LDS_ADDR(p) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(p))
#define COT_GLDS16S(sbase, voff, lds_addr) cot::glds16_s((sbase), (voff), (lds_addr))
#define COT_LDS_READ_TR16(p) cot::lds_read_tr16((p))
#define COT_GLDS16(gptr, lds_wave_base) cot::glds16((gptr), (lds_wave_base))
#define COT_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// nothing is scheduled across this point (register-only instructions -- MFMAs -- move freely across asm statements otherwise:
// the fragment-prefetch loop needs "issue the next step's LDS reads, THEN multiply", and got the opposite without it)
#define COT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// lgkmcnt(0) + s_barrier in ONE asm statement (no LDS access can be scheduled between the two), followed by the same wait as
// a builtin: a no-op for the hardware, but it tells the compiler's own wait-count bookkeeping that every earlier LDS read has
// returned -- without it the compiler, which cannot see inside the asm, guards the first use of registers loaded BEFORE the
// barrier with an lgkmcnt(N) that also drains the reads issued AFTER it (seen in the ISA of the fragment-prefetch loop: the
// MFMAs of step k waited for most of step k+1's fragments).
#define COT_LDS_BARRIER()                                                   \
    do {                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
        __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0), vmcnt / expcnt: no wait */ \
    } while (0)
#else
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#endif

// DIAGNOSTIC time stamps (cot_debug_stamps): lane 0 of wave 0 stores s_memtime into slot `i` of its workgroup's record
// (8 x 8 bytes per workgroup; slot 7 = XCC id).  `p` is NULL in production: one scalar test per stamp.
#ifndef COT_STAMP
#define COT_STAMP(p, i)                                                                                    \
    do {                                                                                                   \
        if ((p) && threadIdx.x == 0) {                                                                     \
            (p)[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();                              \
            if ((i) == 0) {                                                                                \
                unsigned xcc_;                                                                             \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                        \
                (p)[(size_t)blockIdx.x * 8 + 7] = xcc_ & 15u;                                              \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
#endif

// `ahead` stages (G copies each, per wave) were issued after the one about to be read: leave exactly those in flight
template <int G, int A> struct WaitBehind {
    static __device__ __forceinline__ void go(int ahead) {
        if (ahead >= A) COT_WAIT_VM(A * G);
        else WaitBehind<G, A - 1>::go(ahead);
    }
};
template <int G> struct WaitBehind<G, 0> {
    static __device__ __forceinline__ void go(int) { COT_WAIT_VM(0); }
};
// the same when GX further copies (the next chunk's X stage, issued ahead of its stage's own G copies) are among those in flight
template <int G, int GX, int A> struct WaitBehindX {
    static __device__ __forceinline__ void go(int ahead) {
        if (ahead >= A) COT_WAIT_VM(A * G + GX);
        else WaitBehindX<G, GX, A - 1>::go(ahead);
    }
};
template <int G, int GX> struct WaitBehindX<G, GX, 0> {
    static __device__ __forceinline__ void go(int) { COT_WAIT_VM(0); }
};

// ---- epilogue through LDS (shared by the 1x1 and the grouped 3x3 kernels).  In the C/D map a lane holds 4 consecutive
// pixels of ONE channel and the 16 lanes of a group 16 different channels: stored directly that is 32 contiguous bytes per
// channel row and instruction.  Instead the tile goes to LDS in its memory order (rounded to bf16, bias added) and is
// copied out in full 16-byte pieces, 256 contiguous bytes per 16 lanes: BIG = rows of BPX pixels; FLAT = per image one
// contiguous [channels][H*W] block (whatever H*W is -- the channel block of an image IS one flat range of y).
struct EpiArgs {
    bf16_t* y1;
    bf16_t* y2;          // second channel slab of the output (NULL: m1 == M)
    const bf16_t* bias;  // indexed by the global channel (may be NULL)
    int m1, M, HW, N, ni;
    int n0, p0, m0;      // first image, first pixel (BIG), first channel of the tile
    int mv, ncols;       // valid channels / columns of the tile
    int accumulate;      // bit 0: y1 += result, bit 1: y2 += result
    int64_t ys1, ys2;    // elements from one image to the next in y1 / y2 (dense: m1 * HW, (M - m1) * HW; larger when the slab is
                         // a channel range of a wider tensor: one group of a grouped convolution)
    float* stats;        // NULL, or [N][ptiles][M][2]: per (image, pixel tile, channel) the sum and the sum of squares of the
    const bf16_t* acc_src;       // NULL, or a tensor laid out as y1 whose elements -- where the sign-mask bit is set -- are ADDED to the result
    const uint8_t* acc_mask;     // (acc_mask[e >> 3] bit e & 7 for element e of y1): the residual's gradient gout * [block output > 0] of a
                                 // Bottleneck (models/cotnet.py:259-262) folded into conv1's data gradient -- bn3's backward then need not
                                 // write it (cot_conv1x1_backward_data_relu_res).  One output slab, planes of a multiple of 8 pixels.
    int ablate;          // DIAGNOSTIC (cot_set_tuning key 24): bit 5 (32) = no epilogue at all (nothing stored); bit 6 (64) = BIG tiles without a bias /
                         // second slab / accumulate / statistics: the accumulators stored directly (8 bytes per lane), no LDS round trip
    int ptiles;          // tile's outputs AS STORED (rounded to bf16) -- the statistics of the consumer's normalisation come out
                         // of the producing GEMM's epilogue (GroupNorm of the attention logits, cot_conv1x1_forward_gn9;
                         // SURVEY 7.6).  BIG tiles only.
};

template <int CB, int MB, int FLAT, int WAVES>
__device__ __forceinline__ void tile_epilogue(const f32x4_t (&acc)[CB][MB], const EpiArgs& a) {
    constexpr int NT = 64 * WAVES, BPX = 16 * WAVES * CB;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int HW = a.HW, M = a.M, m0 = a.m0, n0 = a.n0, p0 = a.p0, ncols = a.ncols, mv = a.mv;
    if (a.ablate & 32) return;
    if (!FLAT && (a.ablate & 64) && !a.bias && !a.y2 && !a.accumulate && !a.stats) {
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) {
            const int ml = mbk * 16 + i16;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int col = (wave * CB + cb) * 16 + 4 * g;
                bf16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[cb][mbk][r];
                if (ml < mv && col < ncols)
                    __builtin_memcpy(__builtin_assume_aligned(a.y1 + (int64_t)n0 * a.ys1 + (int64_t)(m0 + ml) * HW + p0 + col, 8), o, 8);
            }
        }
        return;
    }
    COT_LDS_BARRIER();  // every wave is done with the last stage: the stage memory is free
    bf16_t* const ot = reinterpret_cast<bf16_t*>(cot_smem);
    constexpr int OS = BPX + 8;                      // BIG: padded row stride of the tile image (bank spread)
    const int nimg = FLAT ? min(a.ni, a.N - n0) : 1;
    const int per = mv * HW, pers = (per + 7) & ~7;  // FLAT: elements of one image's channel block, its (16-byte) LDS stride
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        const int ml = mbk * 16 + i16;
        const float bs = (a.bias && ml < mv) ? (float)a.bias[m0 + ml] : 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int col = (wave * CB + cb) * 16 + 4 * g;
            bf16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[cb][mbk][r] + bs);
            if (!FLAT) {
                __builtin_memcpy(__builtin_assume_aligned(ot + ml * OS + col, 8), o, 8);
            } else if (HW % 4 == 0) {
                if (col < ncols && ml < mv) {  // 4 consecutive columns stay inside one image
                    const int img = col / HW, p = col - img * HW;
                    __builtin_memcpy(__builtin_assume_aligned(ot + img * pers + ml * HW + p, 8), o, 8);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = col + r;
                    if (c < ncols && ml < mv) {
                        const int img = c / HW, p = c - img * HW;
                        ot[img * pers + ml * HW + p] = o[r];
                    }
                }
            }
        }
    }
    COT_LDS_BARRIER();
    if (!FLAT && a.stats) {
        // per (tile, channel) the sum and the sum of squares of the ROUNDED values, from the tile image in LDS.  TPR consecutive lanes
        // share a channel row and take interleaved column pairs (pair q + TPR * j: one 4-byte read each; a wave's 64 lanes then touch 64
        // different banks), add up in a fixed order and meet in log2(TPR) exchanges.  (Round 4's form -- one wave per row, a 6-step
        // exchange per row, 16 rows per wave one after the other -- cost the BIG launches ~20 us each: profiles/r05_bn_epilogue_stats_ab.log)
        constexpr int BM = 16 * MB;
        constexpr int TPR = NT >= 16 * BM ? 16 : (NT >= 8 * BM ? 8 : (NT >= 4 * BM ? 4 : (NT >= 2 * BM ? 2 : 1)));  // lanes per row: 4 / 8 / 16 (1x1 tiles)
        constexpr int RPP = NT / TPR;                                                                             // rows per pass
        const int q = tid % TPR;
        float* const srow = a.stats + (((int64_t)n0 * a.ptiles + p0 / BPX) * M + m0) * 2;
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += RPP) {
            const int r = r0 + tid / TPR;
            float s1 = 0.f, s2 = 0.f;
            if (r < mv) {
#pragma unroll
                for (int j = 0; j < (BPX / 2 + TPR - 1) / TPR; ++j) {
                    const int c = 2 * (q + TPR * j);
                    if (c < ncols) {  // (ncols % 8 == 0: a pair is valid as a whole)
                        const uint32_t u = *reinterpret_cast<const uint32_t*>(ot + r * OS + c);
                        const float v0 = __builtin_bit_cast(float, u << 16), v1 = __builtin_bit_cast(float, u & 0xffff0000u);
                        s1 += v0 + v1;
                        s2 += v0 * v0 + v1 * v1;
                    }
                }
            }
#pragma unroll
            for (int o = TPR >> 1; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o);
                s2 += __shfl_xor(s2, o);
            }
            if (q == 0 && r < mv) {
                srow[2 * r] = s1;
                srow[2 * r + 1] = s2;
            }
        }
    }
    // copy out: 16 bytes per lane.  Output slabs (y1 | y2 at channel m1): a block lies in one slab or, when it straddles m1,
    // rows are routed one by one (m1 % 8 == 0 is checked on the host, so flat 16-byte pieces never straddle the slabs).
    if (!FLAT) {
        constexpr int cpr = BPX / 8;
        for (int q = tid; q < mv * cpr; q += NT) {
            const int row = q / cpr, c = q - row * cpr;
            if (c * 8 >= ncols) continue;
            const int m = m0 + row;
            const bool second = m >= a.m1;
            // (written as one offset behind a selected base: the two-armed pointer expression crashed the compiler's simplifycfg)
            const int64_t ioff = (int64_t)n0 * (second ? a.ys2 : a.ys1) + (int64_t)(second ? m - a.m1 : m) * HW + p0 + c * 8;
            bf16_t* dst = (second ? a.y2 : a.y1) + ioff;
            Vec<bf16_t, 8> v = *reinterpret_cast<const Vec<bf16_t, 8>*>(ot + row * OS + c * 8);
            if ((a.accumulate >> (second ? 1 : 0)) & 1) {
                const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(dst);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (float)pv.v[e]);
            } else if (a.acc_src) {  // (one slab: ioff is the element's offset in y1 and in acc_src alike; a multiple of 8)
                const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(a.acc_src + ioff);
                const unsigned mb = a.acc_mask[ioff >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (((mb >> e) & 1u) ? (float)pv.v[e] : 0.f));
            }
            stv<bf16_t, 8>(dst, v);
        }
    } else {
        const int cpi_o = pers / 8;  // (one image's channel block is contiguous in y -- inside one slab -- and in LDS)
        for (int q = tid; q < nimg * cpi_o; q += NT) {
            const int img = q / cpi_o, c = q - img * cpi_o;
            const int e0 = c * 8;                 // first element of the piece inside the block
            const int m = m0 + e0 / HW;           // its channel decides the slab (pieces do not straddle m1)
            const bool second = m >= a.m1;
            const int64_t boff = (int64_t)(n0 + img) * (second ? a.ys2 : a.ys1) + (int64_t)(second ? m0 - a.m1 : m0) * HW;
            bf16_t* blk = (second ? a.y2 : a.y1) + boff;
            const bool accu = (a.accumulate >> (second ? 1 : 0)) & 1;
            const bf16_t* src = ot + img * pers + e0;
            if (e0 + 8 <= per) {
                Vec<bf16_t, 8> v = *reinterpret_cast<const Vec<bf16_t, 8>*>(src);
                if (accu) {
                    const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(blk + e0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (float)pv.v[e]);
                } else if (a.acc_src) {  // (H*W % 8 == 0 with a mask: boff + e0 is a multiple of 8)
                    const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(a.acc_src + boff + e0);
                    const unsigned mb = a.acc_mask[(boff + e0) >> 3];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (((mb >> e) & 1u) ? (float)pv.v[e] : 0.f));
                }
                stv<bf16_t, 8>(blk + e0, v);
            } else {  // the block's last, partial piece (mv * HW % 8 != 0: only when mv is not a multiple of 8)
                for (int e = e0; e < per; ++e) blk[e] = (bf16_t)(accu ? (float)src[e - e0] + (float)blk[e] : (float)src[e - e0]);
            }
        }
    }
}

struct C1LdsArgs {
    const bf16_t* x1;
    const bf16_t* x2;  // second channel slab of the input (NULL: k1 == K)
    const bf16_t* w;   // wpacked 0: [M][K] row-major; 1: K-step-major [K/32][M][32]; 2: the TRANSPOSE [K][M] row-major (WT kernels)
    const bf16_t* bias;
    bf16_t* y1;
    bf16_t* y2;        // second channel slab of the output (NULL: m1 == M)
    int k1, m1, N, K, M, HW;
    int64_t xs1, xs2;  // elements from one image to the next in x1 / x2 (dense: k1 * HW, (K - k1) * HW), and in
    int64_t ys1, ys2;  // y1 / y2 (dense: m1 * HW, (M - m1) * HW): a slab may be a channel range of a wider tensor
    float* stats;      // NULL or the epilogue statistics workspace (EpiArgs::stats; BIG tiles only)
    const bf16_t* acc_src;      // EpiArgs::acc_src / acc_mask (third generation only)
    const uint8_t* acc_mask;
    int accumulate;    // bit 0: y1 += result, bit 1: y2 += result
    int wpacked;
    int mblocks;       // output-channel blocks of BM
    int ptiles;        // pixel tiles per image (BIG) / image groups (FLAT)
    int ni;            // FLAT: images per workgroup
    int xcd_remap;
    int xswz;          // conv_lds2.hip: bit 0 = BIG tiles: the X stage's 16-byte chunks XOR-permuted per k row; bit 1 = the W tile's
                       // chunk permutation in its conflict-free form (see conv1x1_lds_fwd2)
    int ablate;        // DIAGNOSTIC (cot_set_tuning key 24; results become wrong): bit 0 no copies after the prologue, bit 1 no
                       // fragment reads in the loop, bit 2 one MFMA per step, bit 3 no barriers, bit 4 no vmcnt waits
};

}  // namespace cot
