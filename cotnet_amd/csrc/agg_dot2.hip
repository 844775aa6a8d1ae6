// agg_dot2.hip -- fused aggregation_zeropad backward (gX and gW in one pass), bf16, 3x3 / stride 1 / pad 1, gfx950:
// the packed-bf16 dot-product form of the LDS-staged kernel of agg_nchw.hip (agg_bwd_nchw_k3_lds).
//
// Semantics: cupy_layers/aggregation_zeropad.py:48-79 (input backward), :81-110 (weight backward); same tile / slab
// staging as agg_bwd_nchw_k3_lds (a workgroup owns TR consecutive rows rho = wc*H + h of one image; per channel group
// j the rows rho0-1 .. rho0+TR of gO and x are ONE contiguous range, copied by asynchronous global->LDS DMA).
//
// Why a second kernel.  The fp32-unpacked loop of agg_bwd_nchw_k3_lds spends ~100 VALU instructions per (channel, 2
// pixels) for 36 multiply-adds: every bf16 value is unpacked to fp32, halo columns are moved and masked per channel, and
// the per-lane accesses are 4 bytes wide.  Measured: 0.55 of the HBM roofline at N80 x C64 x 56 x 56 with 1.13x the
// algorithmic traffic -- issue-bound, not bandwidth-bound.  Here the operands stay PACKED (two bf16 per
// 32-bit word, as they lie in memory) and every multiply-add is half of a v_dot2c_f32_bf16 (D += A.lo*B.lo + A.hi*B.hi,
// fp32 accumulate, products of bf16 values are exact in fp32):
//   * gX[c][p] = sum over 3 rows of a 3-tap correlation along W.  Two dot2 per output pixel and row: the word holding
//     columns (p-1, p) or (p, p+1) against a pre-packed weight pair, plus the neighbouring word against a pair with one
//     zero half.  The 6*P weight pairs of an item are built ONCE (they are shared by the C/wC = 8 channels).
//   * gW[tap][p] = sum over the 8 channels of x[c][p + off(tap)] * gO[c][p]: a dot product over CHANNELS.  Two
//     channels' words are re-packed per column (v_perm_b32: (ch a, ch b) of one pixel in one word), then one dot2 per
//     (tap, pixel) does two channels at once.
//   * halo columns come from the neighbouring lanes by DPP row shifts on PACKED words (row_shr:1 / row_shl:1 inside a
//     16-lane row, bound_ctrl zero fill); an image row occupies GS = 16 (or 8) lanes of which the last one or two are
//     idle lanes that hold zeros, so the lane left of a row's first lane and right of its last lane always supplies
//     ZERO: no per-channel row-end selection.
//   * rows outside the image (and everything an idle / out-of-range lane reads) are redirected ONCE, at address
//     computation, to a 16-byte zero chunk that follows every slab in LDS; all LDS offsets of the channel loop are
//     immediates (W, P, tile shape and slab pitch are template parameters).
// ~16-18 VALU instructions per (channel, pixel) instead of ~50, 8-byte accesses per lane instead of 4.
//
// Exactness: integer-valued data is bit-exact (exact products, exact fp32 sums); padded taps of gW are cleared by
// SELECTION once per item (exact zeros whatever gO holds there, as the reference writes 0 without multiplying).
#include "conv_lds_common.h"

namespace cot {

// XCD-aware tile order (see logical_block in agg_nchw.hip): placement only, never results
__device__ __forceinline__ unsigned logical_block_dot2(int xcd_remap) {
    const unsigned b = blockIdx.x, nblk = gridDim.x;
    if (!xcd_remap || (nblk & 7u) != 0) return b;
    return (b & 7u) * (nblk >> 3) + (b >> 3);
}

#ifndef COT_DOT2_BF16  // (tests/emul pre-defines the three primitives for its host build)
typedef __attribute__((ext_vector_type(2))) __bf16 cot_bf16x2;
// acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
#define COT_DOT2_BF16(a, b, acc) \
    __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cot_bf16x2, (uint32_t)(a)), __builtin_bit_cast(cot_bf16x2, (uint32_t)(b)), (acc), false)
// lane l <- lane l-1 / l+1 inside its 16-lane DPP row; the row's first / last lane <- 0 (bound_ctrl)
#define COT_ROW_PREV(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x111, 0xf, 0xf, true))
#define COT_ROW_NEXT(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x101, 0xf, 0xf, true))
// v_perm_b32: (lo half of a, lo half of b) / (hi half of a, hi half of b)
#define COT_PACK_LO(a, b) __builtin_amdgcn_perm((uint32_t)(b), (uint32_t)(a), 0x05040100u)
#define COT_PACK_HI(a, b) __builtin_amdgcn_perm((uint32_t)(b), (uint32_t)(a), 0x07060302u)
#endif

__device__ __forceinline__ uint32_t pack_lo(uint32_t a, uint32_t b) { return COT_PACK_LO(a, b); }
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return COT_PACK_HI(a, b); }
// (lo half of a, hi half of b)
__device__ __forceinline__ uint32_t pack_lo_hi(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b & 0xffff0000u); }

template <int W, int P, int GS, int NW>
struct Dot2Shape {
    static constexpr int SEGS = W / P;                 // active lanes of an image row's lane group
    static constexpr int RW = 64 / GS;                 // image rows per wave
    static constexpr int TR = NW * RW;                 // rows per workgroup
    static constexpr int PW = P / 2;                   // 32-bit words per lane and row
    static constexpr int CH = ((TR + 2) * W * 2 + 14 + 15) / 16;  // 16-byte data chunks per slab (rows + alignment slack)
    static constexpr int PC = CH + 1;                  // slab pitch in chunks: data + one chunk of zeros
    static constexpr int SLAB_B = PC * 16;
    static constexpr int ZOFF = CH * 16;               // byte offset of the zero chunk inside a slab
    static_assert(W % P == 0 && P % 2 == 0 && SEGS < GS && (GS == 16 || GS == 8), "lane group shape");
};

// gO and x slabs of JP channel groups (j0 ..), interleaved: slab 2*jj = gO, 2*jj + 1 = x.  The 64 chunks a wave copies per
// round are consecutive in LDS whatever slab they belong to (destination = wave-uniform base + lane*16); the zero chunk
// at each slab's end is skipped (its lane transfers nothing).
// Dead rows: when the tile starts (ends) at a plane boundary its first (last) slab row lies outside the image -- no lane ever
// reads it (the row is redirected to the zero chunk) -- so the chunks that lie entirely inside it are not copied: chunks
// [0, dead_lo) and [dead_hi, CH).  With plane-aligned tiles (TR a divisor or a multiple of H) this removes the halo
// re-reads of gO and x altogether (TR < H: one live halo row per tile is left).
// ASM = 1: the copies are issued from an asm statement the compiler does not count (conv_lds_common.h), so that they may
// stay in flight across the LDS reads of the phase being computed (the double-buffered form below tracks them by hand).
template <typename S, int JP, int ASM>
__device__ __forceinline__ void dot2_stage(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ x, int64_t src0,
                                           int64_t cstride, int64_t elems, char* smem, int dead_lo, int dead_hi) {
    constexpr int TOTAL = 2 * JP * S::PC;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int nthr = blockDim.x;
    for (int q0 = 0; q0 < TOTAL; q0 += nthr) {
        const int q = q0 + tid;
        const int s = q / S::PC, ch = q - s * S::PC;
        if (q < TOTAL && ch >= dead_lo && ch < dead_hi) {
            const bf16_t* base = (s & 1) ? x : gout;
            int64_t e = src0 + (int64_t)(s >> 1) * cstride + (int64_t)ch * 8;
            if (e + 8 > elems) e = elems - 8;  // a chunk beyond the tensor: in-bounds bytes, never used as data
            if (ASM)
                COT_GLDS16(base + e, smem + (int64_t)(q0 + wave * 64) * 16);
            else
                COT_ASYNC_COPY16(base + e, smem + (int64_t)(q0 + wave * 64) * 16);
        }
    }
}

// SAFE = 1: the half of a packed gO word that belongs to a column OUTSIDE an output's 3-tap window is cleared (one v_and per
// such operand, 4 per row) instead of merely meeting a zero weight: a non-finite gO value then reaches exactly the gX
// elements the reference's kernel puts it in (0 * NaN never happens).  SAFE = 0 saves those 12 instructions per channel;
// finite data gives identical bits either way.
// ROLL = 1 (used with JP = J = 8: EVERY channel group of the tile staged at once, one DMA latency per workgroup instead of
// one per phase): the channel-pair loop is not unrolled, so the compiler keeps one pair's words in registers (~100 VGPRs)
// instead of hoisting the LDS reads of the whole phase (187 at JP = 8 unrolled: two waves per SIMD).
// DB = 1: two LDS buffers; phase p+1's slabs are copied while phase p is computed.  A workgroup's life was a chain of
// (DMA latency -> compute) links, one per phase, with only the other three or four co-resident workgroups to fill the
// gaps: ~30 KB in flight per CU, i.e. latency-bound at ~4.4 TB/s on cold inputs whatever the instruction count (the
// fp32-unpacked kernel and this one measured the same 38 us cold).  Per phase: vmcnt(0) (the prefetched slabs, issued a
// whole compute phase ago, and the previous phase's gX stores -- vmcnt counts both, and stores retire out of order with
// loads, so a counted wait cannot single the copies out), ONE barrier (everybody's chunks visible AND everybody done
// with the buffer about to be refilled), issue the next copies, compute.
// (Holding the allocation to 96 VGPRs -- five waves per SIMD instead of four -- spills 11 values into the channel loop: 56.7 us
// against 34.7 at 56 x 56, profiles/r04_agg_dot2_in_model_tuning.log.  Not built.)
// GroupNorm-9 prologue (gn.mean != NULL): `w` holds the RAW logits; the weights are normalised on the fly with the formula and
// the rounding point of csrc/group_norm9.hip (see Gn9Dot2 / agg_fwd_nchw_k3_lds<SM = 2>), `gw` is the gradient w.r.t. the
// NORMALISED weights (what cot_group_norm9_backward takes as dy).
struct Gn9Dot2 {
    const float* mean;
    const float* rstd;
    const bf16_t* gamma;
    const bf16_t* beta;
    int gimg;
};
__device__ __forceinline__ uint32_t gn9_word(uint32_t v, float ga, float be) {
    const float lo = __builtin_bit_cast(float, v << 16) * ga + be, hi = __builtin_bit_cast(float, v & 0xffff0000u) * ga + be;
    Vec<bf16_t, 2> o;
    o.v[0] = (bf16_t)lo;
    o.v[1] = (bf16_t)hi;
    return __builtin_bit_cast(uint32_t, o);
}

template <int W, int P, int GS, int JP, int NW, int SAFE, int ROLL, int DB>
__global__ __launch_bounds__(NW * 64) void agg_bwd_nchw_k3_dot2(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w, bf16_t* __restrict__ gx,
                                                               bf16_t* __restrict__ gw, int C, int wC, int H, int tiles_per_n,
                                                               int64_t elems, int xcd_remap, Gn9Dot2 gn) {
    typedef Dot2Shape<W, P, GS, NW> S;
    constexpr int PW = S::PW;
    static_assert(JP % 2 == 0, "channels are processed in pairs");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    char* smem = cot_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = C / wC, rows_n = wC * H;
    const int64_t HW = (int64_t)H * W;
    const unsigned lb = logical_block_dot2(xcd_remap);
    const int tile = lb % tiles_per_n;
    const int n = lb / tiles_per_n;
    const int rho0 = tile * S::TR;
    int64_t gs = (int64_t)(rho0 - 1) * W;  // first slab element (row rho0-1), rounded down to a 16-byte chunk
    if (gs < 0) gs = 0;
    gs &= ~(int64_t)7;
    const int64_t cstride = (int64_t)wC * HW;
    const int64_t img = (int64_t)n * C * HW;

    // chunks of the slab's first / last row that need no copy (see dot2_stage)
    const int64_t top_end = ((rho0 % H) == 0) ? (int64_t)rho0 * W - gs : 0;                                   // elements
    const int rlast = rho0 + S::TR;                                                                            // row behind the tile
    const int64_t bot_beg = (rlast >= rows_n || (rlast % H) == 0) ? (int64_t)rlast * W - gs : (int64_t)S::CH * 8;
    const int dead_lo = (int)(top_end / 8);
    int dead_hi = (int)((bot_beg + 7) / 8);
    if (dead_hi > S::CH) dead_hi = S::CH;
    constexpr int BUF_B = 2 * JP * S::SLAB_B;
    dot2_stage<S, JP, DB>(gout, x, img + gs, cstride, elems, smem, dead_lo, dead_hi);  // phase 0 is on its way while the weights are fetched
    if (tid < 2 * JP * (DB ? 2 : 1)) {  // the zero chunk behind every slab (never written by the DMA)
        Vec<uint32_t, 4> z;
        z.v[0] = z.v[1] = z.v[2] = z.v[3] = 0u;
        *reinterpret_cast<Vec<uint32_t, 4>*>(smem + tid * S::SLAB_B + S::ZOFF) = z;
    }

    // this lane's item
    const int grp = lane / GS, seg = lane - grp * GS;
    int rho = rho0 + wave * S::RW + grp;
    const bool valid = seg < S::SEGS && rho < rows_n;
    if (!valid) rho = rows_n - 1;  // (addresses stay inside the tensor; the lane reads zeros and stores nothing)
    const int wc = rho / H, h = rho - wc * H;
    const int w0 = valid ? seg * P : 0;
    const int64_t plane = (int64_t)n * wC + wc;
    const int lidx = (int)((int64_t)rho * W + w0 - gs);  // centre-row vector inside a slab, in elements
    int base[3];  // byte offsets of rows h-1, h, h+1 inside a slab; rows outside the image -> the zero chunk
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const int hr = h - 1 + rr;
        base[rr] = (valid && hr >= 0 && hr < H) ? (lidx + (rr - 1) * W) * 2 : S::ZOFF;
    }

    // ---- gX weight pairs: A_rr[d](c) = Wt[kh = 2 - rr][kw = 1 - d] at (row h-1+rr, column w0 + c)
    uint32_t wpa[3][P], wpb[3][P];
    {
        const bf16_t* wp = w + plane * 9 * HW + w0;
        // every load of the prologue issued before the first use: (GroupNorm prologue) the 20 statistics / affine values, then
        // the 27 weight vectors -- one memory latency.  (Loading gamma / beta per tap inside the wave-uniform `if` made nine dependent
        // round trips of it: 43.6 us against 35 us per launch at 56 x 56 in the model, gpurun_out/r4u.)
        float gmu = 0.f, grs = 1.f;
        bf16_t gam[9], bet[9];  // (kept as loaded: a conversion here would wait for the loads before the weights' are issued)
#pragma unroll
        for (int t = 0; t < 9; ++t) gam[t] = bet[t] = __builtin_bit_cast(bf16_t, (uint16_t)0);
        if (gn.mean) {  // (wave-uniform; issued first, consumed after the weight loads below are on their way too)
            gmu = gn.mean[plane];
            grs = gn.rstd[plane];
            const int gq = (int)((unsigned)plane % (unsigned)gn.gimg);  // (planes < 2^31: dot2_run; a 64-bit modulo is a ~130-instruction routine)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                gam[t] = gn.gamma[gq * 9 + t];
                bet[t] = gn.beta[gq * 9 + t];
            }
        }
        uint32_t tw[3][3][PW];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int kh = 2 - rr, hr = h - 1 + rr;
            const int hc = hr < 0 ? 0 : (hr >= H ? H - 1 : hr);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const Vec<uint32_t, PW> v = *reinterpret_cast<const Vec<uint32_t, PW>*>(wp + (int64_t)(kh * 3 + kw) * HW + (int64_t)hc * W);
#pragma unroll
                for (int k = 0; k < PW; ++k) tw[rr][kw][k] = v.v[k];
            }
        }
        if (gn.mean) {  // GroupNorm of the logits, as the forward pass computed it
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = (2 - rr) * 3 + kw;
                    const float ga = (float)gam[t] * grs, be = (float)bet[t] - gmu * ga;
#pragma unroll
                    for (int k = 0; k < PW; ++k) tw[rr][kw][k] = gn9_word(tw[rr][kw][k], ga, be);
                }
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int hr = h - 1 + rr;
            const bool rok = valid && hr >= 0 && hr < H;
            uint32_t tv[3][PW];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int k = 0; k < PW; ++k) tv[kw][k] = rok ? tw[rr][kw][k] : 0u;  // (selection: a clamped row's weights never count)
            const uint32_t tvL2 = COT_ROW_PREV(tv[2][PW - 1]);  // tap kw=2 at column w0-1 (hi half)
            const uint32_t tvR0 = COT_ROW_NEXT(tv[0][0]);       // tap kw=0 at column w0+P (lo half)
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if ((i & 1) == 0) {
                    const uint32_t t2 = i == 0 ? tvL2 : tv[2][(i - 2) / 2 < 0 ? 0 : (i - 2) / 2];
                    wpa[rr][i] = t2 & 0xffff0000u;                        // (0, A-1(i-1))
                    wpb[rr][i] = pack_lo_hi(tv[1][i / 2], tv[0][i / 2]);  // (A0(i), A+1(i+1))
                } else {
                    const uint32_t t0 = i == P - 1 ? tvR0 : tv[0][(i + 1) / 2 >= PW ? PW - 1 : (i + 1) / 2];
                    wpa[rr][i] = pack_lo_hi(tv[2][(i - 1) / 2], tv[1][(i - 1) / 2]);  // (A-1(i-1), A0(i))
                    wpb[rr][i] = t0 & 0xffffu;                                        // (A+1(i+1), 0)
                }
            }
        }
    }
    float gwacc[9][P];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < P; ++i) gwacc[t][i] = 0.f;

    bf16_t* gxp = gx + img + (int64_t)rho * W + w0;
    for (int j0 = 0; j0 < J; j0 += JP) {
        const char* sb = smem;
        if (DB) {
            COT_WAIT_VM(0);     // this phase's slabs (this wave's chunks) have landed; the previous phase's stores are out
            COT_LDS_BARRIER();  // ... everybody's have, and everybody is done reading the other buffer
            const int pb = (j0 / JP) & 1;
            sb = smem + pb * BUF_B;
            if (j0 + JP < J)
                dot2_stage<S, JP, 1>(gout, x, img + (int64_t)(j0 + JP) * cstride + gs, cstride, elems, smem + (pb ^ 1) * BUF_B,
                                     dead_lo, dead_hi);
        } else {
            if (j0 > 0) {
                __syncthreads();  // everyone finished reading the previous phase's slabs
                dot2_stage<S, JP, 0>(gout, x, img + (int64_t)j0 * cstride + gs, cstride, elems, smem, dead_lo, dead_hi);
            }
            __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes every wave's chunks (and the zero chunks) visible
        }
#pragma unroll(ROLL ? 1 : JP / 2)
        for (int jp = 0; jp < JP; jp += 2) {
            uint32_t g[2][3][PW], xs[2][3][PW];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const Vec<uint32_t, PW> gv = *reinterpret_cast<const Vec<uint32_t, PW>*>(sb + (2 * (jp + c)) * S::SLAB_B + base[rr]);
                    const Vec<uint32_t, PW> xv = *reinterpret_cast<const Vec<uint32_t, PW>*>(sb + (2 * (jp + c) + 1) * S::SLAB_B + base[rr]);
#pragma unroll
                    for (int k = 0; k < PW; ++k) {
                        g[c][rr][k] = gv.v[k];
                        xs[c][rr][k] = xv.v[k];
                    }
                }
            // ---- gX of the two channels
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float acc[P];
#pragma unroll
                for (int i = 0; i < P; ++i) acc[i] = 0.f;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const uint32_t gL = COT_ROW_PREV(g[c][rr][PW - 1]);
                    const uint32_t gR = COT_ROW_NEXT(g[c][rr][0]);
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        if ((i & 1) == 0) {
                            uint32_t gm = i == 0 ? gL : g[c][rr][(i - 2) / 2 < 0 ? 0 : (i - 2) / 2];
                            if (SAFE) gm &= 0xffff0000u;  // only column i-1 of this word is inside the window
                            acc[i] = COT_DOT2_BF16(gm, wpa[rr][i], acc[i]);
                            acc[i] = COT_DOT2_BF16(g[c][rr][i / 2], wpb[rr][i], acc[i]);
                        } else {
                            uint32_t gp = i == P - 1 ? gR : g[c][rr][(i + 1) / 2 >= PW ? PW - 1 : (i + 1) / 2];
                            if (SAFE) gp &= 0xffffu;  // only column i+1
                            acc[i] = COT_DOT2_BF16(g[c][rr][(i - 1) / 2], wpa[rr][i], acc[i]);
                            acc[i] = COT_DOT2_BF16(gp, wpb[rr][i], acc[i]);
                        }
                    }
                }
                Vec<bf16_t, P> o;
#pragma unroll
                for (int i = 0; i < P; ++i) o.v[i] = (bf16_t)acc[i];
                if (valid) stv<bf16_t, P>(gxp + (int64_t)(j0 + jp + c) * cstride, o);
            }
            // ---- gW: the two channels' words re-packed per column, one dot2 per (tap, pixel)
            uint32_t gp2[P];
#pragma unroll
            for (int i = 0; i < P; ++i)
                gp2[i] = (i & 1) ? pack_hi(g[0][1][i / 2], g[1][1][i / 2]) : pack_lo(g[0][1][i / 2], g[1][1][i / 2]);
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                uint32_t xp[P + 2];
#pragma unroll
                for (int i = 0; i < P; ++i)
                    xp[i + 1] = (i & 1) ? pack_hi(xs[0][rr][i / 2], xs[1][rr][i / 2]) : pack_lo(xs[0][rr][i / 2], xs[1][rr][i / 2]);
                xp[0] = COT_ROW_PREV(xp[P]);
                xp[P + 1] = COT_ROW_NEXT(xp[1]);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int i = 0; i < P; ++i) gwacc[rr * 3 + kw][i] = COT_DOT2_BF16(xp[i + kw], gp2[i], gwacc[rr * 3 + kw][i]);
            }
        }
    }
    // taps that reach outside the image: exact zeros by selection, once per item (aggregation_zeropad.py:97-105)
    {
        const bool top = h > 0, bottom = h < H - 1, has_left = seg > 0, has_right = seg < S::SEGS - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int i = 0; i < P; ++i) {
                gwacc[0 + kw][i] = top ? gwacc[0 + kw][i] : 0.f;
                gwacc[6 + kw][i] = bottom ? gwacc[6 + kw][i] : 0.f;
            }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            gwacc[kh * 3 + 0][0] = has_left ? gwacc[kh * 3 + 0][0] : 0.f;
            gwacc[kh * 3 + 2][P - 1] = has_right ? gwacc[kh * 3 + 2][P - 1] : 0.f;
        }
    }
    if (valid) {
        bf16_t* gp = gw + plane * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Vec<bf16_t, P> o;
#pragma unroll
            for (int i = 0; i < P; ++i) o.v[i] = (bf16_t)gwacc[t][i];
            stv<bf16_t, P>(gp + t * HW, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch.  Tuning (cot_set_tuning keys 29 / 30 / 31 / 32): on (1, default) / off; channel groups per LDS phase (0 = by width: 2 at 56, else 4);
// XCD-aware tile order (-1 automatic as the fp32 kernel: planes up to 28 x 28, 0 off, 1 on); waves per workgroup (0 = by width: 7 / 5 = plane-aligned tiles; 2 | 4 | 5 | 7);
// key 33: SAFE operand masking (1 default; 0 = a non-finite gO may reach the next-nearest column of gX as well); key 34: double-
// buffered slabs (1 default: phase p+1 is copied while phase p is computed; 0 = copy, wait, compute)
// ------------------------------------------------------------------------------------------------
static int g_dot2[6] = {1, 0, -1, 0, 1, 1};
int set_tuning_dot2(int key, int value) {
    if (key < 0 || key > 5) return -1;
    g_dot2[key] = value;
    return 0;
}

template <int W, int P, int GS, int JP, int NW>
static int launch_dot2_nw(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                          hipStream_t s, const Gn9Dot2& gn) {
    typedef Dot2Shape<W, P, GS, NW> S;
    const bool safe = g_dot2[4] != 0;
    const int tiles = (g.wC * g.H + S::TR - 1) / S::TR;
    const bool db = g_dot2[5] != 0 && JP < 8;  // (JP = 8 stages everything at once: nothing to prefetch)
    const size_t lds = (size_t)2 * JP * S::SLAB_B * (db ? 2 : 1);
    const int xcd = g_dot2[2] < 0 ? (g.H * g.W <= 28 * 28 ? 1 : 0) : g_dot2[2];
    const int64_t ne = (int64_t)g.N * g.C * g.H * g.W;
    constexpr int ROLL = JP == 8 ? 1 : 0;
    const dim3 grid((unsigned)((int64_t)tiles * g.N)), block(NW * 64);
#define COT_DOT2_GO(SAFE_, DB_)                                                                                                \
    COT_LAUNCH((agg_bwd_nchw_k3_dot2<W, P, GS, JP, NW, SAFE_, ROLL, DB_>), grid, block, lds, s, gout, x, w, gx, gw, g.C, g.wC, g.H, \
               tiles, ne, xcd, gn)
    constexpr int DBV = JP < 8 ? 1 : 0;
    if (safe && db) COT_DOT2_GO(1, DBV);
    else if (safe) COT_DOT2_GO(1, 0);
    else if (db) COT_DOT2_GO(0, DBV);
    else COT_DOT2_GO(0, 0);
#undef COT_DOT2_GO
    return check_launch("agg_bwd_nchw_k3_dot2");
}

// defaults from the on-device A/Bs (profiles/r04_agg_dot2_variants.log, r04_agg_dot2_double_buffer.log; B = 80, us per launch
// cold | cache-warm, SAFE form): 56 x 56: JP 2 x 4 waves double-buffered 36.6 | 30.6, single-buffered 39.0 | 31.7, JP 4 38.0 | 34.5,
// JP 8 (everything staged at once, rolled pair loop) 42.2 | 35.5, 7 waves (plane-aligned tiles: no halo re-reads) 37.2 | 31.1,
// 8 waves 47.9 | 39.6;  28 x 28: JP 4 22.3 | 18.5, JP 2 23.2 | 18.4;  14 x 14: JP 4 12.6 | 10.5, JP 2 13.0 | 10.7
// (the fp32-unpacked LDS kernel: 38.9 | 36.7, 22.7 | 19.4, 14.8 | 13.2)
static inline int default_jp(int W) { return W >= 40 ? 2 : 4; }
static inline int default_nw(int W) { return 4; }

template <int W, int P, int GS, int JP>
static int launch_dot2(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                       hipStream_t s, const Gn9Dot2& gn) {
    // waves per workgroup: 7 (5) make TR = 28 or 56 (20 or 40) rows, i.e. tiles that start and end at plane boundaries of the
    // 56 / 28 / 14 (40 / 20 / 10) row planes -- no halo rows left to re-read; anything else 4.  (8 waves measured 20-25 %
    // slower than 4: not built.)
    int nw = g_dot2[3] > 0 ? g_dot2[3] : default_nw(W);
    if (nw == 5) nw = 4;
    if (nw == 7) {
        const int TRn = nw * (64 / GS);
        if (!(TRn % g.H == 0 || g.H % TRn == 0)) nw = 4;
    }
    switch (nw) {
        case 2: return launch_dot2_nw<W, P, GS, JP, 2>(gout, x, w, gx, gw, g, s, gn);
        case 7: return launch_dot2_nw<W, P, GS, JP, 7>(gout, x, w, gx, gw, g, s, gn);
        default: return launch_dot2_nw<W, P, GS, JP, 4>(gout, x, w, gx, gw, g, s, gn);
    }
}

template <int W, int P, int GS>
static int launch_dot2_jp(int JP, const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                          hipStream_t s, const Gn9Dot2& gn) {
    switch (JP) {  // (8 = everything staged at once, channel-pair loop rolled)
        case 8: return launch_dot2<W, P, GS, 8>(gout, x, w, gx, gw, g, s, gn);
        case 4: return launch_dot2<W, P, GS, 4>(gout, x, w, gx, gw, g, s, gn);
        default: return launch_dot2<W, P, GS, 2>(gout, x, w, gx, gw, g, s, gn);
    }
}

// -> -1: geometry not covered (caller keeps agg_bwd_nchw_k3_lds); else the launch status
static int dot2_run(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g, hipStream_t s,
                    const Gn9Dot2& gn);
int agg_backward_nchw_dot2(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                           hipStream_t s) {
    return dot2_run(gout, x, w, gx, gw, g, s, Gn9Dot2{nullptr, nullptr, nullptr, nullptr, 1});
}
// the same with the GroupNorm-9 prologue (w = raw logits); -1: geometry not covered
int agg_gn9_backward_nchw_dot2(const bf16_t* gout, const bf16_t* x, const bf16_t* logits, const float* mean, const float* rstd,
                               const bf16_t* gamma, const bf16_t* beta, int gimg, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                               hipStream_t s) {
    if (!mean || !rstd || !gamma || !beta || gimg <= 0) return -1;
    return dot2_run(gout, x, logits, gx, gw, g, s, Gn9Dot2{mean, rstd, gamma, beta, gimg});
}
static int dot2_run(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g, hipStream_t s,
                    const Gn9Dot2& gn) {
    if (!g_dot2[0] || !gx || !gw || g.heads != 1 || g.wC <= 0 || g.C % g.wC != 0) return -1;
    const int J = g.C / g.wC;
    const int64_t HW = (int64_t)g.H * g.W;
    if (J % 2 != 0 || ((int64_t)g.wC * HW) % 8 != 0 || (int64_t)g.N * g.C * HW < 8) return -1;
    if ((int64_t)g.wC * HW * J * g.N >= ((int64_t)1 << 31)) return -1;  // (32-bit element offsets inside an image are fine; total < 2^31 keeps lidx arithmetic simple)
    int JP = g_dot2[1] > 0 ? g_dot2[1] : default_jp(g.W);
    while (JP > 2 && J % JP != 0) JP >>= 1;
    if (J % JP != 0) return -1;
    switch (g.W) {
        case 56: return launch_dot2_jp<56, 4, 16>(JP, gout, x, w, gx, gw, g, s, gn);
        case 28: return launch_dot2_jp<28, 4, 8>(JP, gout, x, w, gx, gw, g, s, gn);
        case 14: return launch_dot2_jp<14, 2, 8>(JP, gout, x, w, gx, gw, g, s, gn);
        // SE-CoTNetD at 320 x 320 (models/cotnet_hybrid.py: CoT layers at 40 x 40, 20 x 20 and 10 x 10)
        case 40: return launch_dot2_jp<40, 4, 16>(JP, gout, x, w, gx, gw, g, s, gn);
        case 20: return launch_dot2_jp<20, 4, 8>(JP, gout, x, w, gx, gw, g, s, gn);
        case 10: return launch_dot2_jp<10, 2, 8>(JP, gout, x, w, gx, gw, g, s, gn);
        default: return -1;
    }
}

}  // namespace cot
