// conv_lds.hip -- second generation of the 1x1-convolution kernels: LDS-tiled MFMA GEMMs fed
// by asynchronous global->LDS copies, NCHW in and out, no layout change of the activations.
//
// Why a second generation: the first (conv1x1.hip: no LDS, every wave streams its own operand fragments from global
// memory through a register ring) measured 27-48 % of the HBM roofline on the 56x56 layers but only 3-11 % on the deep-K
// 14x14 / 7x7 layers (profiles/r02_*): X is re-read once per block of output channels, with 8-byte or 2-byte accesses,
// by one or two waves per SIMD.  Here a workgroup owns (a tile of pixels) x (up to 256 output channels):
//   * X[n][k0..k0+31][tile] and W[m-block][k0..k0+31] are copied with global_load_lds_dwordx4 (16 B per lane, no VGPR
//     round trip) into a double-buffered LDS stage: step s+1 is in flight while step s is multiplied, one barrier per
//     step; X is read from HBM ONCE per pixel tile (once per 256 output channels), in full 16-byte chunks whatever H*W is;
//   * small images (H*W = 196, 49: rows are not 16-byte multiples) are staged FLAT -- the K-step's 32 channel rows of an
//     image are one contiguous range of memory, copied as is -- and several images share a workgroup, so the MFMA column
//     blocks are dense (4 images x 49 pixels = 196 columns = 12.25 blocks instead of 4 x 4 blocks 3/4 empty);
//   * MFMA roles: pixels are the ROWS of the product (A = X^T, gathered from LDS with eight 2-byte reads per fragment --
//     the K index of X is the strided one in NCHW), channels the columns (B = W rows, one ds_read_b128 per fragment
//     from an XOR-swizzled image).  In the C/D map a lane then holds 4 CONSECUTIVE pixels of one channel -> 8-byte stores.
//   * the data gradient is the same kernel on dY with W^T, which a small transposition kernel writes into the call's
//     workspace first (weights are <= 2 MB; reading W^T in place needs 2-byte gathers on both operands).
// v_mfma_f32_16x16x32_bf16 operand maps as in mfma_common.h.
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

// dst[c][r] = src[r][c]   (R x C row-major -> C x R row-major), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                            int R, int C) {
    __shared__ bf16_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < R && c < C) tile[ty + 8 * i][tx] = src[(int64_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) dst[(int64_t)c * R + r] = tile[tx][ty + 8 * i];
    }
}

int transpose_bf16(const void* src, void* dst, int R, int C, hipStream_t stream) {
    COT_LAUNCH(transpose_bf16_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, stream, (const bf16_t*)src,
               (bf16_t*)dst, R, C);
    return check_launch("transpose_bf16_kernel");
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
#define COT_GLDS16(gptr, lds_wave_base) cot::glds16((gptr), (lds_wave_base))
#define COT_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define COT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
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

struct C1LdsArgs {
    const bf16_t* x1;
    const bf16_t* x2;  // second channel slab of the input (NULL: k1 == K)
    const bf16_t* w;   // [M][K] row-major
    const bf16_t* bias;
    bf16_t* y1;
    bf16_t* y2;        // second channel slab of the output (NULL: m1 == M)
    int k1, m1, N, K, M, HW;
    int accumulate;    // bit 0: y1 += result, bit 1: y2 += result
    int mblocks;       // output-channel blocks of BM
    int ptiles;        // pixel tiles per image (BIG) / image groups (FLAT)
    int ni;            // FLAT: images per workgroup
    int xcd_remap;
};

// One wave = CB x 16 columns (pixels) x MB x 16 channels; the WAVES waves side by side along the pixels (BPX = 16*WAVES*CB
// columns per workgroup), every wave computes all BM = 16*MB channels of its columns: the 2-byte gathers of the X fragments are done
// once per workgroup, the cheap 16-byte W fragment reads four times.
// FLAT = 0: a tile is BPX consecutive pixels of one image (H*W % 8 == 0: rows are 16-byte multiples)
// FLAT = 1: a tile is `ni` whole images (ni * H*W <= BPX columns); the K step's 32 rows of an image are one flat range
// NS = LDS stages: stages ks+1 .. ks+NS-2 are in flight while stage ks is multiplied, one barrier per K step.
// WAVES = 4 or 8 waves per workgroup.  With 8 (two per SIMD, from the same workgroup) one wave's LDS round trips -- the
// fragment gathers of a K step, ~10 dependent batches -- hide behind the other's MFMAs; with 4 and one workgroup per CU
// (the deep-K layers) they are fully exposed: measured 1.2 us per K step for 0.25 us of MFMA work.
template <int CB, int MB, int FLAT, int NS, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void conv1x1_lds_fwd(const C1LdsArgs a) {
    constexpr int NT = 64 * WAVES;
    constexpr int BPX = 16 * WAVES * CB, BM = 16 * MB, BK = 32;
    constexpr int XP = (BK * BPX / 8 + NT - 1) / NT;     // X copies per thread and stage (full passes of NT x 16 B)
    constexpr int WPASS = (BM * 4 + NT - 1) / NT;        // W copies per thread and stage
    constexpr int XST = XP * NT * 8, WST = WPASS * NT * 8;  // stage sizes in elements (padded to whole passes)
    constexpr int G = XP + WPASS;
    static_assert(XP >= 1 && (NS - 2) * G <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int HW = a.HW, K = a.K, M = a.M;
    bf16_t* const wsm = reinterpret_cast<bf16_t*>(cot_smem);  // [NS][WST] then [NS][XST]
    bf16_t* const xsm = wsm + NS * WST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;

    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const int mb = b % a.mblocks;  // consecutive (same-XCD) workgroups share the X tile
    const int t = b / a.mblocks;
    int n0, p0, ncols;  // first image, first pixel, valid columns of this tile
    if (FLAT) {
        n0 = t * a.ni;
        p0 = 0;
        ncols = min(a.ni, a.N - n0) * HW;
    } else {
        n0 = t / a.ptiles;
        p0 = (t - n0 * a.ptiles) * BPX;
        ncols = min(BPX, HW - p0);
    }
    const int m0 = mb * BM;

    // ---- staging: every wave issues exactly G copies per stage (lanes past a stage's data copy in-bounds bytes into the
    // stage's padding), so one vmcnt arithmetic holds for all waves
    // Source addresses are resolved once: a K step moves every X copy BK rows down its slab (BK*HW elements) and every W
    // copy BK elements along its row; the step's slab (first / second input tensor) is a scalar choice.
    const int cpi = BK * HW / 8, xtotal = FLAT ? a.ni * cpi : BK * BPX / 8;
    int xn[XP], xin[XP];  // this thread's copies: image, element offset inside the image's block of 32 rows
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int q = min(ps * NT + tid, xtotal - 1);
        int n, inner;  // image, element offset inside the image's 32-row block
        if (FLAT) {
            const int img = q / cpi, c = q - img * cpi;
            n = min(n0 + img, a.N - 1);  // images past the batch: in-bounds bytes, never stored
            inner = c * 8;
        } else {
            constexpr int cpr = BPX / 8;  // chunks per row
            const int row = q / cpr, c = q - row * cpr;
            int pc = p0 + c * 8;
            if (pc + 8 > HW) pc = 0;  // partial last tile: columns never stored; any in-bounds bytes will do
            n = n0;
            inner = row * HW + pc;
        }
        xn[ps] = n;
        xin[ps] = inner;
    }
    const bf16_t* wsrc[WPASS];
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
        const int q = min(ps * NT + tid, BM * 4 - 1);
        const int row = q >> 2, pos = q & 3;
        const int c = pos ^ ((row >> 2) & 3);  // XOR swizzle: position `pos` of a row holds its k-chunk c
        const int m = min(m0 + row, M - 1);    // rows past M: a copy of row M-1, never stored
        wsrc[ps] = a.w + (int64_t)m * K + c * 8;
    }
    auto stage = [&](int ks) __attribute__((always_inline)) {
        const int k0 = ks * BK, buf = ks % NS;
        bf16_t* xd = xsm + buf * XST;
        const bool first = k0 < a.k1;  // the K step's rows come from one slab (k1 % 32 == 0 is checked on the host)
        const bf16_t* xbase = first ? a.x1 + (int64_t)k0 * HW : a.x2 + (int64_t)(k0 - a.k1) * HW;
        const int64_t istride = (int64_t)(first ? a.k1 : K - a.k1) * HW;  // elements per image of that slab
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) COT_GLDS16(xbase + xn[ps] * istride + xin[ps], xd + (ps * NT + wave * 64) * 8);
        bf16_t* wd = wsm + buf * WST;
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) COT_GLDS16(wsrc[ps] + k0, wd + (ps * NT + wave * 64) * 8);
    };

    // ---- per-lane LDS offsets of the A (= X^T) gathers: column -> element offset of (k = 0, column) inside a stage
    int aoff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col = (wave * CB + cb) * 16 + i16;
        if (FLAT) {
            const int cc = min(col, a.ni * HW - 1);  // columns past the tile: any staged element
            const int img = cc / HW;
            aoff[cb] = img * BK * HW + (cc - img * HW) + 8 * g * HW;
        } else {
            aoff[cb] = col + 8 * g * BPX;
        }
    }
    const int rs = FLAT ? HW : BPX;  // row (= k) stride of the X stage
    int boff[MB];                    // B (= W) fragments: row i16 of channel block mbk, k-chunk g (swizzled position)
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        const int row = mbk * 16 + i16;
        boff[mbk] = row * BK + (g ^ ((row >> 2) & 3)) * 8;
    }

    f32x4_t acc[CB][MB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) acc[cb][mbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nk) stage(s0);
    for (int ks = 0; ks < nk; ++ks) {
        WaitBehind<G, NS - 2>::go(min(NS - 2, nk - 1 - ks));  // this wave's copies of stage ks have landed
        COT_LDS_BARRIER();                                // everybody's have; nobody still reads stage ks-1's buffer
        if (ks + NS - 1 < nk) stage(ks + NS - 1);
        const uint16_t* xb = reinterpret_cast<const uint16_t*>(xsm + (ks % NS) * XST);
        const bf16_t* wb = wsm + (ks % NS) * WST;
        typedef __attribute__((ext_vector_type(2))) uint16_t u16x2_t;
        bf16x8_t af[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const uint16_t* p = xb + aoff[cb];
            u16x2_t q4[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                q4[h][0] = p[(2 * h) * rs];
                q4[h][1] = p[(2 * h + 1) * rs];
            }
            __builtin_memcpy(&af[cb], q4, 16);
        }
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) {
            bf16x8_t bf;
            __builtin_memcpy(&bf, __builtin_assume_aligned(wb + boff[mbk], 16), 16);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb][mbk] = COT_MFMA_16X16X32_BF16(af[cb], bf, acc[cb][mbk]);
        }
    }

    // ---- epilogue through LDS.  In the C/D map a lane holds 4 consecutive pixels of ONE channel and the 16 lanes of a group
    // 16 different channels: stored directly that is 32 contiguous bytes per channel row and instruction.  Instead the tile
    // goes to LDS in its memory order (rounded to bf16, bias added) and is copied out in full 16-byte pieces, 256
    // contiguous bytes per 16 lanes: BIG = BM rows of BPX pixels; FLAT = per image one contiguous [channels][H*W] block
    // (whatever H*W is -- the channel block of an image IS one flat range of y).
    COT_LDS_BARRIER();  // every wave is done with the last stage: the stage memory is free
    bf16_t* const ot = reinterpret_cast<bf16_t*>(cot_smem);
    constexpr int OS = BPX + 8;                      // BIG: padded row stride of the tile image (bank spread)
    const int mv = min(BM, M - m0);                 // valid channels of this block (a multiple of 8 unless it is the last)
    const int nimg = FLAT ? min(a.ni, a.N - n0) : 1;
    const int per = mv * HW, pers = (per + 7) & ~7;  // FLAT: elements of one image's channel block, its (16-byte) LDS stride
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        const int ml = mbk * 16 + i16;
        const float bs = (a.bias && m0 + ml < M) ? (float)a.bias[m0 + ml] : 0.f;
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
    // copy out: 16 bytes per lane.  Output slabs (y1 | y2 at channel m1): a block lies in one slab or, when it straddles m1,
    // rows are routed one by one (m1 % 8 == 0 is checked on the host, so flat 16-byte pieces never straddle the slabs).
    if (!FLAT) {
        constexpr int cpr = BPX / 8;
        for (int q = tid; q < mv * cpr; q += NT) {
            const int row = q / cpr, c = q - row * cpr;
            if (c * 8 >= ncols) continue;
            const int m = m0 + row;
            const bool second = m >= a.m1;
            bf16_t* dst = (second ? a.y2 + ((int64_t)n0 * (M - a.m1) + (m - a.m1)) * HW : a.y1 + ((int64_t)n0 * a.m1 + m) * HW) + p0 + c * 8;
            Vec<bf16_t, 8> v = *reinterpret_cast<const Vec<bf16_t, 8>*>(ot + row * OS + c * 8);
            if ((a.accumulate >> (second ? 1 : 0)) & 1) {
                const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(dst);
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (float)pv.v[e]);
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
            bf16_t* blk = second ? a.y2 + ((int64_t)(n0 + img) * (M - a.m1) + (m0 - a.m1)) * HW
                                 : a.y1 + ((int64_t)(n0 + img) * a.m1 + m0) * HW;
            const bool accu = (a.accumulate >> (second ? 1 : 0)) & 1;
            const bf16_t* src = ot + img * pers + e0;
            if (e0 + 8 <= per) {
                Vec<bf16_t, 8> v = *reinterpret_cast<const Vec<bf16_t, 8>*>(src);
                if (accu) {
                    const Vec<bf16_t, 8> pv = ldv<bf16_t, 8>(blk + e0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)((float)v.v[e] + (float)pv.v[e]);
                }
                stv<bf16_t, 8>(blk + e0, v);
            } else {  // the block's last, partial piece (mv * HW % 8 != 0: only when mv is not a multiple of 8)
                for (int e = e0; e < per; ++e) blk[e] = (bf16_t)(accu ? (float)src[e - e0] + (float)blk[e] : (float)src[e - e0]);
            }
        }
    }
}

// tuning (cot_set_tuning keys 15..17): [0] 0 = first-generation kernels, 1 = LDS kernels where eligible (default),
// [1] images per workgroup in FLAT mode (0 = auto), [2] reserved
int g_conv_lds_tune[3] = {1, 0, 0};

template <int CB, int MB, int FLAT, int NS, int WAVES>
static int launch_c1(const C1LdsArgs& a, int tiles, hipStream_t stream) {
    constexpr int NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB;
    constexpr int XST = ((32 * BPX / 8 + NT - 1) / NT) * NT * 8, WST = ((BM * 4 + NT - 1) / NT) * NT * 8;
    size_t lds = (size_t)NS * (XST + WST) * sizeof(bf16_t);
    const size_t otile = (FLAT ? (size_t)a.ni * (((size_t)BM * a.HW + 7) & ~(size_t)7) : (size_t)BM * (BPX + 8)) * sizeof(bf16_t);
    if (otile > lds) lds = otile;
    const int64_t blocks = (int64_t)tiles * a.mblocks;
    C1LdsArgs b = a;
    b.xcd_remap = (blocks % 8 == 0) ? 1 : 0;
    if (lds > 64 * 1024) {  // more than the default dynamic-LDS window: opt in once per instantiation (160 KB per CU on gfx950)
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_lds_fwd<CB, MB, FLAT, NS, WAVES>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipGetLastError();
            raised = true;
        }
    }
    COT_LAUNCH((conv1x1_lds_fwd<CB, MB, FLAT, NS, WAVES>), dim3((unsigned)blocks), dim3(NT), lds, stream, b);
    return check_launch("conv1x1_lds_fwd");
}

bool conv1x1_lds_covers(int K, int k1, bool two_slabs, int HW) {
    if (!g_conv_lds_tune[0]) return false;
    if (K % 32 != 0 || K < 32 || (two_slabs && k1 % 32 != 0)) return false;
    return (HW % 8 == 0 && HW >= 256) || HW <= 256;
}

// returns COT_OK, an error, or -1 when the geometry is not covered (the caller then takes the first-generation kernel)
int conv1x1_lds_gemm(const void* x1, const void* x2, int k1, const void* w, const void* bias, void* y1, void* y2, int m1,
                     int N, int K, int M, int HW, int accumulate, hipStream_t stream) {
    if (!conv1x1_lds_covers(K, k1, x2 != nullptr, HW)) return -1;
    if (y2 && m1 % 8 != 0) return -1;  // (16-byte pieces of the output must not straddle the two output slabs)
    if (HW % 8 != 0 && ((y2 ? m1 : M) % 8 != 0 || (y2 && (M - m1) % 8 != 0))) return -1;  // image blocks of y 16-byte aligned
    C1LdsArgs a;
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.w = (const bf16_t*)w; a.bias = (const bf16_t*)bias;
    a.y1 = (bf16_t*)y1; a.y2 = (bf16_t*)y2;
    a.k1 = x2 ? k1 : K; a.m1 = y2 ? m1 : M; a.N = N; a.K = K; a.M = M; a.HW = HW; a.accumulate = accumulate;
    a.ni = 1; a.xcd_remap = 0;
    const int w8 = g_conv_lds_tune[2];  // tuning key 17: 0 = default (8-wave workgroups), 1 = 4-wave workgroups (A/B)
    if (HW > 256) {  // BIG: 128-pixel tiles of one image; three stages, several workgroups per CU
        a.ptiles = ceil_div(HW, 128);
        const int tiles = N * a.ptiles;
        if (M <= 32) { a.mblocks = 1; return w8 ? launch_c1<2, 2, 0, 3, 4>(a, tiles, stream) : launch_c1<1, 2, 0, 3, 8>(a, tiles, stream); }
        if (M <= 64) { a.mblocks = 1; return w8 ? launch_c1<2, 4, 0, 3, 4>(a, tiles, stream) : launch_c1<1, 4, 0, 3, 8>(a, tiles, stream); }
        a.mblocks = ceil_div(M, 128);
        return w8 ? launch_c1<2, 8, 0, 3, 4>(a, tiles, stream) : launch_c1<1, 8, 0, 3, 8>(a, tiles, stream);
    }
    // FLAT: whole images, up to 256 columns per workgroup, up to 128 channels.  These are the deep-K layers with few
    // workgroups (one per CU at best): six stages, five of them in flight per workgroup
    int ni = g_conv_lds_tune[1] > 0 ? g_conv_lds_tune[1] : 256 / HW;
    if (ni > N) ni = N;
    if (ni < 1 || ni * HW > 256) return -1;
    a.ni = ni;
    a.ptiles = ceil_div(N, ni);
    if (M <= 32) { a.mblocks = 1; return w8 ? launch_c1<4, 2, 1, 6, 4>(a, a.ptiles, stream) : launch_c1<2, 2, 1, 6, 8>(a, a.ptiles, stream); }
    if (M <= 64) { a.mblocks = 1; return w8 ? launch_c1<4, 4, 1, 6, 4>(a, a.ptiles, stream) : launch_c1<2, 4, 1, 6, 8>(a, a.ptiles, stream); }
    a.mblocks = ceil_div(M, 128);
    return w8 ? launch_c1<4, 8, 1, 6, 4>(a, a.ptiles, stream) : launch_c1<2, 8, 1, 6, 8>(a, a.ptiles, stream);
}

}  // namespace cot
