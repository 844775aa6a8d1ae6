// conv_wgrad2.hip -- 1x1 weight gradient, third generation (the weight gradients of
// models/cotnet.py:51-62 embed / conv1x1 and :206-224 Bottleneck conv1 / conv3):
//     dW[m][j] = sum over (n, p) of dY[n][m][p] * X[n][j][p]      (+ db[m] = sum dY)
// Both operands are contiguous along the reduction index (pixels), so both MFMA fragments are 16-byte LDS reads.  What the
// second generation (conv_lds.hip conv1x1_wgrad_lds<GEN>) spent its time on was not the product: every K step re-derived
// each thread's (image, pixel) position with vector arithmetic, ran divergent tensor-end checks per copy, and its one tile
// shape (128 x 128) left half of the rows of a 64-channel layer empty (profiles/r02_conv_abi.log: 3-10 % of the HBM
// roofline on the 14 x 14 / 7 x 7 layers, 20-40 % at 56 x 56).  Here
//   * the reduction runs per image in steps of 32 pixels; a copy is `global_load_lds_dwordx4 voff, s[base]`
//     (conv_lds_common.h glds16_s): per-lane offsets are loop constants -- one set for ordinary steps, one for the last step
//     of an image -- and the K step moves scalar pointers only;
//   * a plane that is not a multiple of 8 / 32 pixels needs no tensor-end special case: the chunk that would run over the
//     row's end is read 8 pixels ENDING at the row's end instead (always inside the row), the pixels it then repeats -- and
//     the chunks entirely past the row -- are removed by SELECTION from the dY fragment (the same row's own data, so a
//     non-finite value there is one the true sum contains anyway); both operands shift alike, so the k pairing holds;
//   * tile shapes follow the layer: 128 x 128, 64 x 256, 256 x 64, 64 x 64, 32 x 128 (dY rows x X rows), 8 waves each;
//   * the 16-byte chunks of a row are stored permuted by pos = chunk ^ ((-(row >> 2)) & 3): the four 16-lane groups a
//     ds_read_b128 is served in ({0-3,12-15,20-27}, ... -- MI355X_MICROARCH.md) then touch 16 different slots (the old
//     permutation chunk ^ ((row >> 2) & 3) was 2-way conflicting: measured 0.5 conflict cycles per LDS cycle);
//   * MFMA roles: A = X rows, B = dY rows, so a lane ends up with FOUR CONSECUTIVE j of one dW row: 16-byte stores of the
//     fp32 partial sums (8-byte stores of bf16 when the launch is not split).
// Slices of the reduction are summed by the deterministic reduce kernel of conv1x1.hip (no atomics).
//
// What bounds the K loop (round 3, profiles/r03_wgrad_loop_study.md): NOT its instruction stream.  Per step and CU the loop
// moves (TM + TJ) x 64 bytes from L2 into LDS, and the time per step is that byte count at ~27 GB/s per CU whatever the loop
// looks like -- scalar-light addressing, fragment prefetch, copies on loader waves of their own, one barrier per four
// steps all measured within 3 % of each other, while the same loop with the copies removed runs 2.5x faster.  A CU
// fetching 16 rows x 64 B pieces from rows that start on no 64-byte boundary (392 / 98 bytes per row) gets 23 B/clk from L2
// even alone with warm lines (scripts/ubench/ldpath.hip: flat 1 KB pieces 60 B/clk, 8 rows x 128 B 35 B/clk); with the
// first touch of every line coming from HBM at the same time it is 13-14 B/clk.  Hence the K64 form below.
#include <algorithm>

#include "cot_common.h"
#include "mfma_common.h"
#include "conv_lds_common.h"

namespace cot {

// cot_set_tuning key 25: bit 0 = off (second-generation kernels instead); bit 1 = the plain form (no fragment prefetch, 32-pixel
// stages, every wave copies: the reference point of the A/Bs and the form that carries the phase stamps); bit 2 = the old
// chunk permutation of that form (A/B of the bank-conflict fix); bit 6 = 32-pixel stages for every plane; bit 7 = 64-pixel
// stages + loader waves for every plane (default: planes of more than 64 pixels); bits 8..15 = partial-sum cap in percent
// of the input bytes (0 = 100); bits 16..23 = target workgroups per CU x 4 (0 = 4, i.e. one); bits 24..30 = forced slice
// count (tests)
int g_wgrad2_tune = 0;
extern int g_conv_ablate;
extern unsigned long long* g_debug_stamps;

struct Wg2Args {
    const bf16_t* gy;
    const bf16_t* x1;
    const bf16_t* x2;
    float* part;   // [S][M][Jp] partial sums (S > 1)
    bf16_t* gw;    // [M][J]   (S == 1: written directly)
    bf16_t* gb;    // [M] or NULL
    int k1, N, M, J, HW, has_bias, S, jtiles, mtiles, T;
    int spi;       // K steps per image = ceil(cpi / 4)
    int cpi;       // 8-pixel chunks per image = ceil(HW / 8)
    int oldswz;
    int xcd_remap;
    int ablate;    // DIAGNOSTIC (cot_set_tuning key 24; results become wrong): see C1LdsArgs::ablate
    // grouped 3x3 weight gradient (TAPS kernels): per group, M = Cout / G rows of dY against J = 9 Kc "virtual rows" of X, row
    // (tap, ci) = X's channel ci read at pixel + off(tap); gw is [Cout][Kc][3][3]
    int G, Kc, cy, cx, W;          // groups, input channels per group, channels per image of dY / X, image width
    int sy, sx;                    // 1x1 form: channels per image of dY / x1 when they are channel ranges of wider tensors (0: dense)
    const uint16_t* masks;         // conv3x3g_masks table: bit t of masks[p] = pixel p has an in-image neighbour for tap t
    unsigned long long* stamps;  // DIAGNOSTIC (cot_debug_stamps): per workgroup 6 x s_memtime (start, set up, first data, loop done,
                                 // stores issued, end) + the XCC id; NULL in production
};

// Three forms of the K loop over one set of tile shapes (WAVES = 8 consumer waves, WM x (8 / WM) of them, AM x AJ MFMA tiles each):
//   PF = 0          every wave copies; per step: wait, barrier, issue the refill, read the fragments, multiply
//   PF = 1, LW = 0  ... the fragments of step k+1 are read into a second register set before step k multiplies
//   LW > 0 (K64)    LW extra LOADER waves issue every copy, the consumer waves only read fragments and issue MFMAs; a stage
//                   holds 64 pixels (two MFMA k blocks) of every row and is copied in pieces of 8 rows x 128 bytes instead of
//                   16 rows x 64 bytes: the same bytes in half as many, twice as long runs per row -- the vector-memory path
//                   works on 128-byte lines and the deep layers' rows start on no particular boundary.  One meeting per stage:
//                   before it the loaders have seen the stage land, after it they refill the slot the consumers have left.
//
// TAPS (K64 form, nine consumer waves): the weight gradient of the grouped 3 x 3 convolution (models/cotnet.py:43-47 key_embed)
//     dW[m][ci][tap] = sum over (n, p) of dY[n][m][p] * X[n][ci][p + off(tap)]   (0 where the neighbour is outside the image)
// is the same product with nine shifted views of X as extra rows: a 16-row block of the X part is 16 channels of ONE tap,
// copied from `x + off(tap)` (the caller guarantees W + 1 readable elements either side of the tensor), and what the shift
// pulls in from the neighbouring row / channel / image is cleared in the fragment with the tap's bit of the validity table
// (by AND: the garbage may be Inf / NaN).  conv3x3g_wgrad_mfma (conv3x3g.hip) gives every WAVE its own 64 columns and loads
// each shifted row straight from L2: 4.7 % of the HBM roofline (profiles/r02_conv_abi.log).
template <int WAVES, int WM, int AM, int AJ, int NS, int PF, int LW = 0, int K64 = 0, int TAPS = 0>
__global__ __launch_bounds__(64 * (WAVES + LW), PF ? (LW ? (WAVES > 8 ? 4 : 3) : 2) : 4) void conv1x1_wgrad_lds2(const Wg2Args a) {  // (the prefetching form holds two fragment sets: more than 128 registers)
    constexpr int WJ = WAVES / WM, TM = WM * AM * 16, TJ = WJ * AJ * 16;
    constexpr int RBM = TM / 16, RBJ = TJ / 16, RB = RBM + RBJ;  // 16-row blocks of the dY / X parts of a stage
    constexpr int KB = K64 ? 2 : 1, KC = 4 * KB;                  // MFMA k blocks (32 pixels) / 8-pixel chunks per row and stage
    constexpr int SB = KC * 16;                                   // bytes per row and stage
    constexpr int CW = LW ? LW : WAVES;                           // waves that copy
    constexpr int G = (RB * KB + CW - 1) / CW;                    // copies (1 KB pieces) per copying wave and stage
    static_assert(LW == 0 || PF, "loader waves: prefetching form only");
    static_assert(K64 == 0 || LW > 0, "64-pixel stages: loader form only");
    static_assert(TAPS == 0 || K64, "3 x 3 taps: 64-pixel loader form only");
    constexpr int STG = RB * 512 * KB;                            // elements per stage ([RB * 16 rows][32 KB pixels])
    static_assert((NS - 1) * G <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* const sm = reinterpret_cast<bf16_t*>(cot_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6), i16 = lane & 15, g = lane >> 4;
    COT_STAMP(a.stamps, 0);
    const bool loader = LW && wave >= WAVES;        // (wave-uniform)
    const int cwv = LW ? wave - WAVES : wave;       // index among the copying waves
    const int wm = (loader ? 0 : wave) / WJ, wj = (loader ? 0 : wave) % WJ;
    const int HW = a.HW, M = a.M, J = a.J, Jp = J + (a.has_bias ? 1 : 0);
    const int spi = a.spi, cpi = a.cpi, rem = HW & 7;
    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    // (integer divisions are expanded into VECTOR code; without the explicit readfirstlanes everything derived from them --
    // the copies' base pointers, the loop counters, the branch conditions -- ends up in VGPRs / under exec masks)
    const int jt = uniform((int)(b % (unsigned)a.jtiles));
    const int rest = uniform((int)(b / (unsigned)a.jtiles));
    const int mt = uniform(rest % a.mtiles), r2 = uniform(rest / a.mtiles);
    const int sl = TAPS ? uniform(r2 % a.S) : r2, grp = TAPS ? uniform(r2 / a.S) : 0;
    const int m0 = mt * TM, j0 = jt * TJ;
    const int t0 = uniform((int)((unsigned)a.T * (unsigned)sl / (unsigned)a.S));  // (T * S < 2^31: checked on the host)
    const int t1 = uniform((int)((unsigned)a.T * (unsigned)(sl + 1) / (unsigned)a.S));
    const int nst = t1 - t0;
    // chunk permutation inside a row (applied to the SOURCE address; the image in LDS is lane-linear): bank-conflict-free
    // ds_read_b128 fragments (SQ_LDS_BANK_CONFLICT = 0, profiles/r03_conv_sq_counters.txt)
    auto f = [&](int row) { return K64 ? ((row >> 1) & 7) : (a.oldswz ? ((row >> 2) & 3) : ((-(row >> 2)) & 3)); };

    // ---- this wave's copies: 16-row block rb = wave + WAVES i (blocks past the stage: block RB-1 again, same bytes to the same place).
    // Scalar state per copy: ONE running byte pointer `sp` (the step's base, minus a bias that keeps every per-lane offset
    // non-negative) that advances 64 bytes per ordinary step and `wrapb` bytes at an image's last step; per-lane state: two
    // constant offsets (ordinary steps / the last step of an image).  The K step's scalar work is what the loop is made of
    // (SQ counters, profiles/r03_conv_sq_counters.txt: 8 scalar instructions and 1.6 branches per MFMA before this form).
    unsigned voff[G] = {}, voffl[G] = {};
    const char* sp[G] = {};
    int wrapb[G] = {};
    unsigned ldst[G] = {};        // scalar: byte offset of the block inside a stage
    const int n_first = uniform(t0 / spi);
    int s_s = t0 - n_first * spi;  // step inside the image of the NEXT stage to issue
    if (!LW || loader) {
        const int bias = SB * spi;
        const int pos = lane & (KC - 1);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int unit = min(cwv + CW * i, RB * KB - 1);  // (pieces past the stage: the last one again, same bytes to the same place)
            const int rb = unit / KB;
            const int rowl = K64 ? (unit % KB) * 8 + (lane >> 3) : (lane >> 2);  // row inside the 16-row block
            const int ch = pos ^ f(rb * 16 + rowl);  // the 8-pixel chunk of the step this lane fetches
            int r0, r;
            const bf16_t* base;
            int64_t stride;
            if (rb < RBM) {
                const int rr = m0 + rb * 16;
                r0 = min(rr, M - 1);
                r = min(rr + rowl, M - 1);  // rows past the matrix: copies of the last row, never stored
                base = a.gy + (int64_t)(TAPS ? grp * M + r0 : r0) * HW;
                stride = (int64_t)(TAPS ? a.cy : (a.sy ? a.sy : M)) * HW;
            } else if (TAPS) {
                const int rr = j0 + (rb - RBM) * 16;
                r0 = min(rr, J - 1) & ~15;            // (Kc % 16 == 0: a block is 16 channels of one tap)
                r = min(rr + rowl, J - 1);
                const int tap = r0 / a.Kc, ci0 = r0 - tap * a.Kc;
                base = a.x1 + (int64_t)(grp * a.Kc + ci0) * HW + ((tap / 3 - 1) * a.W + (tap % 3 - 1));
                stride = (int64_t)a.cx * HW;
            } else {
                const int rr = j0 + (rb - RBM) * 16;
                r0 = min(rr, J - 1);
                r = min(rr + rowl, J - 1);  // (the bias column's "row" J is made of ones at fragment level)
                const bool second = a.x2 && r0 >= a.k1;  // k1 % 16 == 0 (host): a block lies in one slab
                base = second ? a.x2 + (int64_t)(r0 - a.k1) * HW : a.x1 + (int64_t)r0 * HW;
                stride = (int64_t)(second ? J - a.k1 : (a.x2 ? a.k1 : (a.sx ? a.sx : J))) * HW;
            }
            sp[i] = reinterpret_cast<const char*>(base + n_first * stride) + (s_s * SB - bias);
            wrapb[i] = (int)(stride * 2) - (spi - 1) * SB;  // (2 * stride < 2^31: checked on the host)
            ldst[i] = (unsigned)(unit * 1024);
            voff[i] = (unsigned)(((r - r0) * HW + ch * 8) * 2 + bias);
            const int c = KC * (spi - 1) + ch;  // chunk index inside the image in its last step
            const int pl = (c < cpi - 1 || (c == cpi - 1 && rem == 0)) ? c * 8 : HW - 8;
            voffl[i] = (unsigned)(((r - r0) * HW + pl) * 2 - (spi - 1) * SB + bias);
        }
    }
    const unsigned lds0 = COT_LDS_ADDR(sm);
    // issue the next stage's copies into the slot at LDS byte address `slot_addr` (wave-uniform)
    auto stage = [&](unsigned slot_addr) __attribute__((always_inline)) {
        const bool last = s_s == spi - 1;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            COT_GLDS16S(sp[i], last ? voffl[i] : voff[i], slot_addr + ldst[i]);
            sp[i] += last ? wrapb[i] : SB;
        }
        s_s = last ? 0 : s_s + 1;
    };

    // ---- fragments: row i16 of a 16-row block, k chunk g (at its permuted position)
    int yoff[KB][AM], xoff[KB][AJ];
    bool ones[AJ];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int q = 0; q < AM; ++q) {
            const int row = (wm * AM + q) * 16 + i16;
            yoff[kb][q] = row * (KC * 8) + (((4 * kb + g) ^ f(row)) & (KC - 1)) * 8;
        }
#pragma unroll
        for (int u = 0; u < AJ; ++u) {
            const int row = RBM * 16 + (wj * AJ + u) * 16 + i16;
            xoff[kb][u] = row * (KC * 8) + (((4 * kb + g) ^ f(row)) & (KC - 1)) * 8;
        }
    }
#pragma unroll
    for (int u = 0; u < AJ; ++u) ones[u] = a.has_bias && j0 + (wj * AJ + u) * 16 + i16 == J;  // the bias gradient rides along as an X row of ones
    // the last step of an image: lane group g holds chunk c = 4 (spi-1) + g; elements below `lo` repeat earlier pixels (the
    // chunk was read ending at the row's end) or lie past the row altogether: cleared in the X fragments (the operand with
    // fewer fragments; its row of ones for the bias gradient is cleared the same way, which is what that sum needs) --
    // by an AND with a mask that is all ones in every other step: no branch
    uint32_t mk[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int c = KC * (spi - 1) + 4 * kb + g;
        const int lo = (c < cpi - 1 || (c == cpi - 1 && rem == 0)) ? 0 : (c == cpi - 1 ? 8 - rem : 8);
#pragma unroll
        for (int d = 0; d < 4; ++d) mk[kb][d] = (2 * d >= lo ? 0x0000ffffu : 0u) | (2 * d + 1 >= lo ? 0xffff0000u : 0u);
    }
    const bool tails = (HW & (KC * 8 - 1)) != 0;  // (else every chunk of the last step is whole: nothing to clear)
    int s_c = t0 - n_first * spi;       // step inside the image of the stage being READ next

    f32x4_t acc[AJ][AM];
#pragma unroll
    for (int u = 0; u < AJ; ++u)
#pragma unroll
        for (int q = 0; q < AM; ++q) acc[u][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragments of the stage at `sb` (slot base; a compile-time slot folds into the reads' immediate offsets): loads only -- the
    // clean-up below is applied when the fragments are USED, one step later, so that nothing waits for the reads here.
    // Returns whether the stage was the last step of an image.
    auto read_kblock = [&](auto kbc, const bf16_t* sb, uint32_t (&yq)[AM][4], uint32_t (&xq)[AJ][4]) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value;
#pragma unroll
        for (int q = 0; q < AM; ++q) __builtin_memcpy(yq[q], __builtin_assume_aligned(sb + yoff[kb][q], 16), 16);
#pragma unroll
        for (int u = 0; u < AJ; ++u) __builtin_memcpy(xq[u], __builtin_assume_aligned(sb + xoff[kb][u], 16), 16);
    };
    auto next_is_last = [&]() __attribute__((always_inline)) {  // the stage about to be read: last step of its image?
        const bool lastc = s_c == spi - 1;
        s_c = lastc ? 0 : s_c + 1;
        return lastc;
    };
    auto read_frags = [&](const bf16_t* sb, uint32_t (&yq)[AM][4], uint32_t (&xq)[AJ][4]) __attribute__((always_inline)) {
        const bool lastc = next_is_last();
        read_kblock(std::integral_constant<int, 0>{}, sb, yq, xq);
        return lastc;
    };
    // the X fragments' clean-up: the bias row becomes ones; in the last step of an image the repeated / out-of-row elements are
    // cleared (mask OR all-ones in every other step: data flow, no branch)
    auto fix_kblock = [&](auto kbc, uint32_t (&xq)[AJ][4], bool lastc) __attribute__((always_inline)) {
        constexpr int kb = decltype(kbc)::value;
        if (a.has_bias) {  // (wave-uniform: layers without a bias skip the selects)
#pragma unroll
            for (int u = 0; u < AJ; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) xq[u][e] = ones[u] ? 0x3f803f80u : xq[u][e];  // bf16 1.0 twice
        }
        if (tails) {  // (wave-uniform, loop-invariant)
            const uint32_t pass = lastc ? 0u : 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t m = mk[kb][e] | pass;
#pragma unroll
                for (int u = 0; u < AJ; ++u) xq[u][e] &= m;
            }
        }
    };
    auto fix_frags = [&](uint32_t (&xq)[AJ][4], bool lastc) __attribute__((always_inline)) {
        fix_kblock(std::integral_constant<int, 0>{}, xq, lastc);
    };
    const int abl = a.ablate;
    auto mma = [&](const uint32_t (&yq)[AM][4], const uint32_t (&xq)[AJ][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < AJ; ++u)
#pragma unroll
            for (int q = 0; q < AM; ++q) acc[u][q] = COT_MFMA_16X16X32_BF16(packed_as_frag(xq[u]), packed_as_frag(yq[q]), acc[u][q]);
    };
    auto multiply = [&](const uint32_t (&yq)[AM][4], const uint32_t (&xq)[AJ][4]) __attribute__((always_inline)) {
        if (abl & 4) {
            acc[0][0] = COT_MFMA_16X16X32_BF16(packed_as_frag(xq[0]), packed_as_frag(yq[0]), acc[0][0]);
            return;
        }
#pragma unroll
        for (int u = 0; u < AJ; ++u)
#pragma unroll
            for (int q = 0; q < AM; ++q) acc[u][q] = COT_MFMA_16X16X32_BF16(packed_as_frag(xq[u]), packed_as_frag(yq[q]), acc[u][q]);
    };

    COT_STAMP(a.stamps, 1);
    auto slot_addr = [&](int slot) { return lds0 + (unsigned)(slot * STG * 2); };
    if (PF) {
        // Ring of NS slots, stage s in slot s % NS; iteration ks multiplies the fragments of step ks, read one iteration earlier
        // (structure and vmcnt arithmetic: conv_lds2.hip conv1x1_lds_fwd2).  The steady state is unrolled NS times so that
        // every slot is a compile-time constant: LDS addresses become immediates of the reads, M0 one scalar add.
        static_assert(LW > 0 || NS % 2 == 0, "the register double buffer alternates per step");
        if constexpr (LW > 0) {
            if (loader) {
                int kiss = 0, slot = 0;  // next stage to issue, its slot
                auto issue = [&]() __attribute__((always_inline)) {
                    if (kiss < nst) {
                        if (!(a.ablate & 1) || kiss < NS) stage(slot_addr(slot));  // (DIAGNOSTIC bit 0: only the first ring-full is copied)
                        slot = slot + 1 == NS ? 0 : slot + 1;
                        ++kiss;
                    }
                };
#pragma unroll
                for (int b = 0; b < NS - 1; ++b) issue();
                for (int need = 1; need <= nst; ++need) {     // one meeting per stage, as the consumers below
                    WaitBehind<G, NS - 2>::go(kiss - need);  // stage need-1 has landed; the younger ones stay in flight
                    COT_LDS_BARRIER();
                    issue();                                 // into the slot the consumers have just left
                }
                return;
            }
            // TAPS: validity table in the chunk domain of the K loop, behind the ring: entry [c][e] belongs to element e of chunk c
            // of an image (chunk cpi-1 of a plane that is no multiple of 8 was read ending at the row's end: its first 8 - rem
            // elements repeat pixels and are cleared like the chunks past the row)
            uint16_t* const tbl = reinterpret_cast<uint16_t*>(sm + NS * STG);
            int xtap[AJ] = {};
            if (TAPS) {
                for (int i = tid; i < spi * KC * 8; i += 64 * WAVES) {
                    const int c = i >> 3, e = i & 7;
                    const bool whole = c < cpi - 1 || (c == cpi - 1 && rem == 0);
                    const int px = whole ? i : HW - 8 + e;
                    tbl[i] = (c >= cpi || (!whole && e < 8 - rem)) ? (uint16_t)0 : a.masks[px];
                }
#pragma unroll
                for (int u = 0; u < AJ; ++u) xtap[u] = uniform(min(j0 + (wj * AJ + u) * 16, J - 1) / a.Kc);
            }
            COT_LDS_BARRIER();
            // k blocks kk = 0 .. KB nst - 1 (stage kk / KB), fragments of block kk+1 on their way while block kk multiplies
            uint32_t y0[AM][4], x0[AJ][4], y1[AM][4], x1[AJ][4], v0[4] = {}, v1[4] = {};
            int scur = s_c;              // step inside its image of the stage being read
            bool lcur = next_is_last();  // ... is it the image's last?  (of the stage the block in registers belongs to)
            auto read_valid = [&](auto kbc, int st, uint32_t (&vq)[4]) __attribute__((always_inline)) {
                constexpr int kb = decltype(kbc)::value;
                if (TAPS) __builtin_memcpy(vq, __builtin_assume_aligned(tbl + ((KC * st + 4 * kb + g) << 3), 16), 16);
            };
            auto fix_taps = [&](uint32_t (&xq)[AJ][4], const uint32_t (&vq)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < AJ; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xq[u][e] &= ((vq[e] >> xtap[u]) & 0x00010001u) * 0xffffu;
            };
            read_kblock(std::integral_constant<int, 0>{}, sm, y0, x0);
            read_valid(std::integral_constant<int, 0>{}, scur, v0);
            int slot = 0;                // slot of the stage being read
            const int nkk = KB * nst;
            auto sub = [&](auto kbc, int kk, const uint32_t (&yc)[AM][4], uint32_t (&xc)[AJ][4], const uint32_t (&vc)[4],
                           uint32_t (&yn)[AM][4], uint32_t (&xn)[AJ][4], uint32_t (&vn)[4]) __attribute__((always_inline)) {
                constexpr int kb = decltype(kbc)::value, nkb = (kb + 1) % KB;
                const bool lc = lcur;
                if (kk + 1 < nkk) {
                    if (nkb == 0) {  // the next block opens a stage: meet (everything read so far is in registers: the slot behind is free)
                        slot = slot + 1 == NS ? 0 : slot + 1;
                        COT_LDS_BARRIER();
                        scur = s_c;
                        lcur = next_is_last();
                    }
                    read_kblock(std::integral_constant<int, nkb>{}, sm + slot * STG, yn, xn);
                    read_valid(std::integral_constant<int, nkb>{}, scur, vn);
                    COT_SCHED_FENCE();
                }
                if (TAPS) fix_taps(xc, vc);
                else fix_kblock(kbc, xc, lc);
                mma(yc, xc);
                COT_SCHED_FENCE();
            };
            for (int kk = 0; kk < nkk; kk += 2) {
                sub(std::integral_constant<int, 0>{}, kk, y0, x0, v0, y1, x1, v1);
                if (kk + 1 < nkk) sub(std::integral_constant<int, KB - 1>{}, kk + 1, y1, x1, v1, y0, x0, v0);
            }
        } else {
#pragma unroll
        for (int s0 = 0; s0 < NS; ++s0)
            if (s0 < nst) stage(slot_addr(s0));
        WaitBehind<G, NS - 1>::go(min(NS - 1, nst - 1));
        COT_LDS_BARRIER();
        uint32_t y0[AM][4], x0[AJ][4], y1[AM][4], x1[AJ][4];
        bool l0 = read_frags(sm, y0, x0), l1 = false;
        auto steady = [&](auto slot_c, int ks0, const uint32_t (&yc)[AM][4], uint32_t (&xc)[AJ][4], bool lc, uint32_t (&yn)[AM][4],
                          uint32_t (&xn)[AJ][4], bool& ln) __attribute__((always_inline)) {
            constexpr int U = decltype(slot_c)::value, NXT = (U + 1) % NS;
            // DIAGNOSTIC (as in the other form below): 0 loop top, 1 own copies landed, 2 barrier passed, 3 next fragments'
            // reads issued, 4 copies issued, 5 MFMAs issued
            const bool tr = a.stamps && blockIdx.x == 0 && ks0 + U >= 8 && ks0 + U < 16;
            unsigned long long* const tp = a.stamps + (size_t)(2048 + (tr ? ks0 + U - 8 : 0)) * 8 - (size_t)blockIdx.x * 8;
            if (tr) COT_STAMP(tp, 0);
            COT_WAIT_VM((NS - 2) * G);  // stage ks+1 has landed (this wave's copies); the NS-2 younger stages stay in flight
            if (tr) COT_STAMP(tp, 1);
            COT_LDS_BARRIER();          // ... everybody's; and nobody reads slot U any more (its fragments are in registers)
            if (tr) COT_STAMP(tp, 2);
            ln = read_frags(sm + NXT * STG, yn, xn);
            COT_SCHED_FENCE();          // the reads are in flight BEFORE copies and MFMAs are issued
            if (tr) COT_STAMP(tp, 3);
            stage(lds0 + (unsigned)(U * STG * 2));
            COT_SCHED_FENCE();
            if (tr) COT_STAMP(tp, 4);
            fix_frags(xc, lc);
            mma(yc, xc);
            COT_SCHED_FENCE();
            if (tr) COT_STAMP(tp, 5);
        };
        int ks = 0;
        if (NS == 4) {
            for (; ks + 2 * NS - 1 < nst; ks += NS) {  // (steps ks .. ks+NS-1 all re-fill: ks + NS-1 + NS < nst)
                steady(std::integral_constant<int, 0>{}, ks, y0, x0, l0, y1, x1, l1);
                steady(std::integral_constant<int, 1>{}, ks, y1, x1, l1, y0, x0, l0);
                steady(std::integral_constant<int, 2 % NS>{}, ks, y0, x0, l0, y1, x1, l1);
                steady(std::integral_constant<int, 3 % NS>{}, ks, y1, x1, l1, y0, x0, l0);
            }
        }
        int slot = 0;  // (ks is a multiple of NS here)
        auto step = [&](int k, const uint32_t (&yc)[AM][4], uint32_t (&xc)[AJ][4], bool lc, uint32_t (&yn)[AM][4],
                        uint32_t (&xn)[AJ][4], bool& ln) __attribute__((always_inline)) {
            if (k + 1 < nst) {
                WaitBehind<G, NS - 2>::go(min(NS - 2, nst - 2 - k));
                COT_LDS_BARRIER();
                const int freed = slot;
                slot = slot + 1 == NS ? 0 : slot + 1;
                ln = read_frags(sm + slot * STG, yn, xn);
                COT_SCHED_FENCE();
                if (k + NS < nst) stage(slot_addr(freed));
                COT_SCHED_FENCE();
            }
            fix_frags(xc, lc);
            mma(yc, xc);
            COT_SCHED_FENCE();
        };
        for (; ks < nst; ks += 2) {
            step(ks, y0, x0, l0, y1, x1, l1);
            if (ks + 1 < nst) step(ks + 1, y1, x1, l1, y0, x0, l0);
        }
        }
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nst) stage(slot_addr(s0));
        int slot = 0, fill = NS - 1;
        auto step = [&](auto steady, int ks) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady)::value;
            // DIAGNOSTIC: phase stamps of steps 8..15 of workgroup 0 (records 2048.. of the stamps buffer): 0 loop top, 1 copies
            // of the stage landed, 2 barrier passed, 3 copies issued, 4 fragments in registers, 5 MFMAs issued
            const bool tr = a.stamps && blockIdx.x == 0 && ks >= 8 && ks < 16;
            unsigned long long* const tp = a.stamps + (size_t)(2048 + (tr ? ks - 8 : 0)) * 8 - (size_t)blockIdx.x * 8;
            if (tr) COT_STAMP(tp, 0);
            if (!(abl & 16)) {
                if (STEADY) COT_WAIT_VM((NS - 2) * G);
                else WaitBehind<G, NS - 2>::go(min(NS - 2, nst - 1 - ks));  // this wave's copies of stage ks have landed
            }
            if (tr) COT_STAMP(tp, 1);
            if (!(abl & 8)) COT_LDS_BARRIER();                     // everybody's have; nobody still reads stage ks-1's slot
            if (tr) COT_STAMP(tp, 2);
            if ((STEADY || ks + NS - 1 < nst) && !(abl & 1)) stage(slot_addr(fill));
            fill = fill + 1 == NS ? 0 : fill + 1;
            if (tr) COT_STAMP(tp, 3);
            uint32_t yq[AM][4], xq[AJ][4];
            const bool lq = read_frags(sm + ((abl & 2) ? 0 : slot) * STG, yq, xq);
            fix_frags(xq, lq);
            slot = slot + 1 == NS ? 0 : slot + 1;
            if (tr) {
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the fragments have arrived
                COT_STAMP(tp, 4);
            }
            multiply(yq, xq);
            if (tr) COT_STAMP(tp, 5);
        };
        int ks = 0;
        if (nst > NS) {  // (the first step apart, so that the arrival of the first data can be stamped)
            step(std::true_type{}, 0);
            ks = 1;
            COT_STAMP(a.stamps, 2);
        }
        for (; ks + NS - 1 < nst; ++ks) step(std::true_type{}, ks);
        for (; ks < nst; ++ks) step(std::false_type{}, ks);
    }
    COT_STAMP(a.stamps, 3);

    // ---- D[i = X row j][col = dY row m]: lane holds j = jb + 4g .. 4g+3 of dW row m = mb + i16
    float* const ps = a.part + (int64_t)sl * (TAPS ? a.cy : M) * Jp;
    const bool vec = (Jp & 3) == 0;  // rows of the output start on 16-byte (fp32) / 8-byte (bf16) boundaries
#pragma unroll
    for (int q = 0; q < AM; ++q) {
        const int m = m0 + (wm * AM + q) * 16 + i16;
        if (m >= M) continue;
#pragma unroll
        for (int u = 0; u < AJ; ++u) {
            const int jj = j0 + (wj * AJ + u) * 16 + 4 * g;
            if (jj >= Jp) continue;
            if (TAPS && a.S == 1) {  // single slice: straight into [Cout][Kc][3][3] (column jj = tap Kc + ci -> ci 9 + tap)
                const int tap = jj / a.Kc, ci = jj - tap * a.Kc;
#pragma unroll
                for (int r = 0; r < 4; ++r) a.gw[((int64_t)(grp * M + m) * a.Kc + ci + r) * 9 + tap] = (bf16_t)acc[u][q][r];
                continue;
            }
            if (a.S > 1) {
                float* p = ps + (int64_t)(TAPS ? grp * M + m : m) * Jp + jj;
                if (vec) {
                    *reinterpret_cast<f32x4_t*>(__builtin_assume_aligned(p, 16)) = acc[u][q];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (jj + r < Jp) p[r] = acc[u][q][r];
                }
            } else if (vec) {  // (no bias: Jp == J)
                Vec<bf16_t, 4> o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o.v[r] = (bf16_t)acc[u][q][r];
                stv<bf16_t, 4>(a.gw + (int64_t)m * J + jj, o);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (jj + r < J) a.gw[(int64_t)m * J + jj + r] = (bf16_t)acc[u][q][r];
                    else if (jj + r == J && a.has_bias) a.gb[m] = (bf16_t)acc[u][q][r];
                }
            }
        }
    }
    COT_STAMP(a.stamps, 4);
    if (a.stamps) {
        COT_WAIT_VM(0);
        COT_STAMP(a.stamps, 5);
    }
}

// tile shape for a layer: 0 = 128 x 128, 1 = 64 x 256, 2 = 256 x 64, 3 = 64 x 64, 4 = 32 x 128   (dY rows x X rows)
static inline int wgrad2_shape(int M, int Jp, int* TM, int* TJ) {
    int s;
    if (M <= 32 && Jp > 64) s = 4;
    else if (M <= 64) s = Jp <= 64 ? 3 : 1;
    else if (Jp <= 64) s = 2;
    else s = 0;
    static const int tm[5] = {128, 64, 256, 64, 32}, tj[5] = {128, 256, 64, 64, 128};
    *TM = tm[s];
    *TJ = tj[s];
    return s;
}

bool conv1x1_wgrad2_covers(int N, int HW, int M, int J, int k1, bool two_slabs) {
    if (g_wgrad2_tune & 1) return false;
    if (HW < 8 || (two_slabs && k1 % 16 != 0)) return false;
    if ((int64_t)N * ceil_div(ceil_div(HW, 8), 4) * 1025 >= ((int64_t)1 << 31)) return false;  // (32-bit slice arithmetic in the kernel)
    if ((int64_t)std::max(M, J) * HW * 2 >= ((int64_t)1 << 31)) return false;                    // (image strides in bytes as int)
    // per-lane offsets are 32-bit and relative to an image row: 16 rows of a plane must lie within 2 GB (they always do)
    return (int64_t)HW * 16 * 2 < ((int64_t)1 << 30) && N > 0 && M > 0 && J > 0;
}

// 64-pixel stages + loader waves (see the kernel): planes of more than 64 pixels.  Measured over CoTNet-50's layers against the
// prefetching 32-pixel form (profiles/r03_wgrad_k64_ab.log): 56 x 56 +-1 %, 28 x 28 -7 %, 14 x 14 -10 %; 7 x 7 (one 64-pixel
// stage per image, 15 of its 64 pixels padding either way) +3 %, so those stay on the 32-pixel form.
static inline bool wgrad2_k64(int HW) {
    if ((g_wgrad2_tune >> 1) & 1) return false;  // the plain form
    if ((g_wgrad2_tune >> 6) & 1) return false;
    if ((g_wgrad2_tune >> 7) & 1) return true;
    return HW > 64;
}

// number of slices of the reduction (also sizes the workspace)
int conv1x1_wgrad2_splits(int N, int M, int J, int HW, int has_bias) {
    const int Jp = J + (has_bias ? 1 : 0);
    int TM, TJ;
    wgrad2_shape(M, Jp, &TM, &TJ);
    const int64_t tiles = (int64_t)ceil_div(M, TM) * ceil_div(Jp, TJ);
    const int cpi = ceil_div(HW, 8), spi = ceil_div(cpi, wgrad2_k64(HW) ? 8 : 4);
    const int64_t T = (int64_t)N * spi;
    const int force = (g_wgrad2_tune >> 24) & 127;  // (tests: bits 24..30 force the slice count)
    if (force) return (int)std::min<int64_t>(force, T);
    const int per_cu4 = (g_wgrad2_tune >> 16) & 255;
    // ONE workgroup per CU: measured on the MI355X over all CoTNet-50 layers (profiles/r03_wgrad_policy_ab.log, us per step):
    // two per CU / partial sums <= 50 % of the inputs 2475, one per CU 2250, one per CU and <= 100 % 2121 (second
    // generation: 2747) -- every extra slice is another M x J fp32 matrix written and read again
    int64_t S = ceil_div64((int64_t)64 * (per_cu4 > 0 ? per_cu4 : 4), tiles);
    const int64_t in_bytes = (int64_t)N * HW * (M + J) * 2, out_bytes = (int64_t)M * Jp * 4;
    const int pct = (g_wgrad2_tune >> 8) & 255;
    const int64_t cap = in_bytes * (pct > 0 ? pct : 100) / 100 / out_bytes;  // partial sums are written once and read once
    if (S > cap) S = cap;
    if (S > T / 8) S = T / 8;
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    return (int)S;
}

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb, hipStream_t stream);

// Sum of a FEW slices of a LARGE matrix (the deep layers: 0.26 .. 1 M outputs, 2 .. 16 slices).  The reduce kernel of
// conv1x1.hip gives a workgroup 32 outputs and spreads the slices over its 8 lane groups -- right for the small matrices of
// the first stages with their hundreds of slices, but here it is 32768 workgroups moving 128 bytes per slice each: 42 of the
// 57 us of the 2048 -> 512 @7x7 weight gradient were spent outside the GEMM (profiles/r03_wgrad_ablate.log).  This one gives
// every lane four consecutive outputs: 16-byte loads per slice, one 8-byte bf16 store.  (no bias column: [M][J] is flat)
__global__ __launch_bounds__(256) void wgrad_reduce_wide(const float* __restrict__ part, int S, int64_t tot4,
                                                        bf16_t* __restrict__ gw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= tot4) return;
    const f32x4_t* p = reinterpret_cast<const f32x4_t*>(part) + e;
    f32x4_t s0 = p[0], s1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int sl = 1;
    for (; sl + 1 < S; sl += 2) {
        s0 += p[(int64_t)sl * tot4];
        s1 += p[(int64_t)(sl + 1) * tot4];
    }
    if (sl < S) s0 += p[(int64_t)sl * tot4];
    s0 += s1;
    Vec<bf16_t, 4> o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o.v[r] = (bf16_t)s0[r];
    stv<bf16_t, 4>(gw + e * 4, o);
}

static int wgrad2_reduce(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb, hipStream_t stream) {
    const int64_t tot = (int64_t)M * J;
    if (!has_bias && (J & 3) == 0 && S <= 64 && tot >= 32768 && !((g_wgrad2_tune >> 4) & 1)) {  // (bit 4: always the general kernel, A/B)
        COT_LAUNCH(wgrad_reduce_wide, dim3((unsigned)ceil_div64(tot / 4, 256)), dim3(256), 0, stream, part, S, tot / 4, (bf16_t*)gw);
        return check_launch("wgrad_reduce_wide");
    }
    return conv1x1_wgrad_reduce_launch(part, S, M, J, has_bias, gw, gb, stream);
}

template <int WM, int AM, int AJ, int PF, int NS, int WAVES = 8, int LW = 0, int K64 = 0>
static int launch_wg2(const Wg2Args& a, int64_t blocks, hipStream_t stream) {
    constexpr int WJ = WAVES / WM, RB = WM * AM + WJ * AJ;
    const size_t lds = (size_t)NS * RB * 1024 * (K64 ? 2 : 1);
    static std::atomic<uint32_t> raised{0};
    if (lds > 64 * 1024 && !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv1x1_wgrad_lds2<WAVES, WM, AM, AJ, NS, PF, LW, K64>)))
        return -1;
    COT_LAUNCH((conv1x1_wgrad_lds2<WAVES, WM, AM, AJ, NS, PF, LW, K64>), dim3((unsigned)blocks), dim3(64 * (WAVES + LW)), lds, stream, a);
    return check_launch("conv1x1_wgrad_lds2");
}

// returns COT_OK, an error, or -1 when not covered (the caller then takes the second-generation kernels)
// sy / sx (channels; 0 = dense): dY / x1 are channel ranges of tensors with that many channels per image (one group of a grouped
// convolution; single-slab calls only)
int conv1x1_wgrad2_run(const void* gy, const void* x1, const void* x2, int k1, void* gw, void* gb, float* workspace, int N,
                       int J, int M, int HW, hipStream_t stream, int sy, int sx) {
    if (!conv1x1_wgrad2_covers(N, HW, M, J, k1, x2 != nullptr)) return -1;
    if ((sy || sx) && (x2 || (int64_t)std::max(sy, sx) * HW * 2 >= ((int64_t)1 << 31))) return -1;
    Wg2Args a;
    a.sy = sy; a.sx = sx;
    a.gy = (const bf16_t*)gy; a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.part = workspace;
    a.gw = (bf16_t*)gw; a.gb = (bf16_t*)gb; a.k1 = x2 ? k1 : J; a.N = N; a.M = M; a.J = J; a.HW = HW;
    a.has_bias = gb ? 1 : 0;
    const int Jp = J + a.has_bias;
    int TM, TJ;
    const int shape = wgrad2_shape(M, Jp, &TM, &TJ);
    a.S = conv1x1_wgrad2_splits(N, M, J, HW, a.has_bias);
    a.mtiles = ceil_div(M, TM);
    a.jtiles = ceil_div(Jp, TJ);
    a.cpi = ceil_div(HW, 8);
    const bool k64 = wgrad2_k64(HW);
    a.spi = ceil_div(a.cpi, k64 ? 8 : 4);
    a.T = N * a.spi;
    a.oldswz = (g_wgrad2_tune >> 2) & 1;
    a.ablate = g_conv_ablate;
    a.stamps = g_debug_stamps;
    const int64_t blocks = (int64_t)a.jtiles * a.mtiles * a.S;
    a.xcd_remap = blocks % 8 == 0;
    const bool pf = !((g_wgrad2_tune >> 1) & 1);  // fragment prefetch: default on (2272 -> 2089 us per step over CoTNet-50's layers, profiles/r03_wgrad_pf_ab.log)
    int rc;
    // ring: four stages; three 40 KB stages for the 320-row tiles in the 64-pixel form (160 KB of LDS)
#define COT_WG2(WM_, AM_, AJ_)                                                                                         \
    (k64 ? (WM_ * AM_ + (8 / WM_) * AJ_ > 16 ? launch_wg2<WM_, AM_, AJ_, 1, 3, 8, 4, 1>(a, blocks, stream)            \
                                             : launch_wg2<WM_, AM_, AJ_, 1, 4, 8, 4, 1>(a, blocks, stream))           \
         : (pf ? launch_wg2<WM_, AM_, AJ_, 1, 4>(a, blocks, stream) : launch_wg2<WM_, AM_, AJ_, 0, 4>(a, blocks, stream)))
    switch (shape) {
        case 0: rc = COT_WG2(2, 4, 2); break;
        case 1: rc = COT_WG2(1, 4, 2); break;
        case 2: rc = COT_WG2(4, 4, 2); break;
        case 3: rc = COT_WG2(2, 2, 1); break;
        default: rc = COT_WG2(1, 2, 1); break;
    }
#undef COT_WG2
    if (rc || a.S == 1) return rc;
    return wgrad2_reduce(workspace, a.S, M, J, a.has_bias, gw, gb, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// grouped 3 x 3 weight gradient on the TAPS kernels

// gw[row][ci][tap] = sum over slices of part[s][row][tap Kc + ci]  (fixed order: deterministic).  A lane owns four consecutive
// ci of one (row, tap) -- 16-byte loads along the partial sums' own layout -- and scatters four 2-byte results (the first form
// walked the OUTPUT order, i.e. read 4 bytes every Kc floats: 19 us per call for the 14 x 14 layers, as long as the GEMM itself).
__global__ __launch_bounds__(256) void wgrad_reduce_taps(const float* __restrict__ part, int S, int rows, int Kc,
                                                         bf16_t* __restrict__ gw) {
    // 64 columns (of four floats) x 4 slice lanes per workgroup: lane group q sums the slices q, q + 4, ..; the four partial sums
    // meet in LDS in a fixed order.  (The 56 x 56 layers have 9216 outputs and 64 slices: one lane per four outputs walked 64
    // dependent loads on nine workgroups -- 32 us, at the very end of the backward pass.)
    __shared__ f32x4_t red[3][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + col, tot = (int64_t)rows * Kc * 9, tot4 = tot >> 2;  // (Kc % 16 == 0)
    f32x4_t s0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, s1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (e < tot4) {
        const f32x4_t* p = reinterpret_cast<const f32x4_t*>(part) + e;
        int sl = q;
        for (; sl + 4 < S; sl += 8) {
            s0 += p[(int64_t)sl * tot4];
            s1 += p[(int64_t)(sl + 4) * tot4];
        }
        if (sl < S) s0 += p[(int64_t)sl * tot4];
        s0 += s1;
    }
    if (q > 0) red[q - 1][col] = s0;
    __syncthreads();
    if (q != 0 || e >= tot4) return;
    s0 = (s0 + red[0][col]) + (red[1][col] + red[2][col]);
    const int64_t o = e * 4;
    const int64_t row = o / (9 * Kc);
    const int rem = (int)(o - row * (9 * Kc)), tap = rem / Kc, ci = rem - tap * Kc;
    bf16_t* dst = gw + (row * Kc + ci) * 9 + tap;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r * 9] = (bf16_t)s0[r];
}

// tile of a layer: rows of dY per workgroup (all of the group's, at most 64) x 144 or 288 virtual rows of X
static inline void taps_tile(int Mg, int Kc, int* AM, int* AJ) {
    *AM = Mg >= 64 ? 4 : Mg / 16;
    *AJ = (Kc == 32 || Kc == 64) ? 2 : 1;
}

bool conv3x3g_wgrad2_covers(int N, int Cin, int Cout, int G, int H, int W, int x_guard) {
    if (g_wgrad2_tune & 1) return false;
    if (G <= 0 || Cin % G || Cout % G) return false;
    const int Kc = Cin / G, Mg = Cout / G, HW = H * W;
    if (Kc % 16 || (Mg != 16 && Mg != 32 && Mg % 64) || HW < 8 || W + 1 > x_guard) return false;
    if ((int64_t)N * ceil_div(ceil_div(HW, 8), 8) * 1025 >= ((int64_t)1 << 31)) return false;
    if ((int64_t)std::max(Cin, Cout) * HW * 2 >= ((int64_t)1 << 31)) return false;
    return (int64_t)ceil_div(HW, 64) * 64 * 2 <= 16 * 1024 && N > 0;  // (the validity table sits behind the ring in LDS)
}

int conv3x3g_wgrad2_splits(int N, int Cin, int Cout, int G, int HW) {
    const int Kc = Cin / G, Mg = Cout / G;
    int AM, AJ;
    taps_tile(Mg, Kc, &AM, &AJ);
    const int64_t tiles = (int64_t)G * ceil_div(Mg, AM * 16) * ceil_div(9 * Kc, 144 * AJ);
    const int64_t T = (int64_t)N * ceil_div(ceil_div(HW, 8), 8);
    const int force = (g_wgrad2_tune >> 24) & 127;
    if (force) return (int)std::min<int64_t>(force, T);
    int64_t S = ceil_div64(256, tiles);
    const int64_t in_bytes = (int64_t)N * HW * (Cin + Cout) * 2, out_bytes = (int64_t)Cout * 9 * Kc * 4;
    const int64_t cap = std::max<int64_t>(in_bytes / out_bytes, T >= 64 ? 4 : 1);  // (partial sums <= the inputs, but not one long chain)
    if (S > cap) S = cap;
    if (S > T / 4) S = T / 4;
    if (S < 1) S = 1;
    return (int)S;
}

template <int AM, int AJ, int NS>
static int launch_taps(const Wg2Args& a, int64_t blocks, int tbl_bytes, hipStream_t stream) {
    constexpr int RB = AM + 9 * AJ;
    const size_t lds = (size_t)NS * RB * 2048 + (size_t)tbl_bytes;
    static std::atomic<uint32_t> raised{0};
    if (lds > 64 * 1024 && !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv1x1_wgrad_lds2<9, 1, AM, AJ, NS, 1, 4, 1, 1>)))
        return -1;
    COT_LAUNCH((conv1x1_wgrad_lds2<9, 1, AM, AJ, NS, 1, 4, 1, 1>), dim3((unsigned)blocks), dim3(64 * 13), lds, stream, a);
    return check_launch("conv3x3g wgrad (taps)");
}

// returns COT_OK, an error, or -1 when not covered (the caller then takes conv3x3g_wgrad_mfma)
int conv3x3g_wgrad2_run(const void* gy, const void* x, void* gw, const void* masks, float* workspace, int N, int Cin, int Cout,
                        int G, int H, int W, int x_guard, hipStream_t stream) {
    if (!conv3x3g_wgrad2_covers(N, Cin, Cout, G, H, W, x_guard)) return -1;
    const int Kc = Cin / G, Mg = Cout / G, HW = H * W;
    Wg2Args a = {};
    a.gy = (const bf16_t*)gy; a.x1 = (const bf16_t*)x; a.x2 = nullptr; a.part = workspace; a.gw = (bf16_t*)gw; a.gb = nullptr;
    a.N = N; a.M = Mg; a.J = 9 * Kc; a.k1 = a.J; a.HW = HW; a.has_bias = 0;
    a.G = G; a.Kc = Kc; a.cy = Cout; a.cx = Cin; a.W = W; a.masks = (const uint16_t*)masks;
    int AM, AJ;
    taps_tile(Mg, Kc, &AM, &AJ);
    a.S = conv3x3g_wgrad2_splits(N, Cin, Cout, G, HW);
    a.mtiles = ceil_div(Mg, AM * 16);
    a.jtiles = ceil_div(a.J, 144 * AJ);
    a.cpi = ceil_div(HW, 8);
    a.spi = ceil_div(a.cpi, 8);
    a.T = N * a.spi;
    a.ablate = g_conv_ablate;
    a.stamps = nullptr;
    const int64_t blocks = (int64_t)a.jtiles * a.mtiles * a.S * G;
    a.xcd_remap = blocks % 8 == 0;
    const int tbl = a.spi * 64 * 2;
    int rc;
    if (AM == 1 && AJ == 1) rc = launch_taps<1, 1, 4>(a, blocks, tbl, stream);
    else if (AM == 2 && AJ == 1) rc = launch_taps<2, 1, 4>(a, blocks, tbl, stream);
    else if (AM == 2 && AJ == 2) rc = launch_taps<2, 2, 3>(a, blocks, tbl, stream);
    else if (AM == 4 && AJ == 2) rc = launch_taps<4, 2, 3>(a, blocks, tbl, stream);
    else if (AM == 4 && AJ == 1) rc = launch_taps<4, 1, 4>(a, blocks, tbl, stream);
    else if (AM == 1 && AJ == 2) rc = launch_taps<1, 2, 4>(a, blocks, tbl, stream);
    else return -1;
    if (rc || a.S == 1) return rc;
    const int64_t tot = (int64_t)Cout * Kc * 9;
    COT_LAUNCH(wgrad_reduce_taps, dim3((unsigned)ceil_div64(tot / 4, 64)), dim3(256), 0, stream, workspace, a.S, Cout, Kc, (bf16_t*)gw);
    return check_launch("wgrad_reduce_taps");
}

}  // namespace cot
