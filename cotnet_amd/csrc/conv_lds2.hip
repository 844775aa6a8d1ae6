// conv_lds2.hip -- third generation of the 1x1-convolution forward / data-gradient kernel (SURVEY 8a rows a7/a8/a11;
// models/cotnet.py:51-62 embed / conv1x1, :206-224 Bottleneck conv1 / conv3): the LDS-DMA pipelined MFMA GEMM of conv_lds.hip
// with the K step's instruction stream cut down to what the step needs.
//
// What round 3's first hardware session showed (gpurun_out/r3s1_ni_sweep.log): on the 7x7 layers the time of a launch is
// proportional to (workgroups x K steps) / CUs and does NOT depend on how many images a workgroup carries -- 1 image per
// workgroup (5x the workgroups, 1/5 of the data per step) is 2.3x SLOWER than 5.  A K step costs ~0.8 us = ~1900 cycles
// whatever it moves: the step is its own instruction stream.  The ISA of conv1x1_lds_fwd has ~110 instructions per step and
// wave for 16 MFMAs, most of them the three copies' addresses: 64-bit vector multiplies and adds, five readfirstlanes, an
// M0 save / restore per copy, a modulo for the ring slot -- two waves per SIMD issue all of that between two barriers, then
// everybody waits for LDS round trips, then everybody multiplies.
// Here:
//   * a copy is `global_load_lds_dwordx4 voff, s[base:base+1]` (conv_lds_common.h glds16_s): the per-lane byte offsets are
//     LOOP CONSTANTS, a K step moves one scalar base per operand; M0 and the ring slot are scalar adds;
//   * PF = 1: the fragments of step k+1 are read from LDS into a second register set BEFORE the MFMAs of step k are issued,
//     so LDS latency hides behind the matrix pipe instead of in front of it (one barrier per step as before; the barrier of
//     step k now also certifies that stage k+1 has landed);
//   * workgroups walk K from step 0 (the staggered start of the second generation measured nothing).
// Tile shapes, fragment maps, the epilogue and the dispatch rules are those of conv_lds.hip (conv_lds_common.h).
#include "cot_common.h"
#include "mfma_common.h"
#include <algorithm>

#include "conv_lds_common.h"

namespace cot {

extern int g_conv_lds_tune[3];
// cot_set_tuning key 23: bit 0 = second-generation forward kernel (conv_lds.hip) instead of this one; bit 1 = fragment prefetch
// off in the FLAT kernels (default on); bit 2 = fragment prefetch on in the BIG kernels (default off: it costs registers, i.e.
// workgroups per CU, where the layers are bandwidth-bound)
int g_conv_lds2_tune = 0;
int g_conv_big_fill = 200;  // cot_set_tuning key 46: BIG tiles -- output-channel blocks of 64 / 32 instead of 128 while the launch has fewer workgroups than this (0 = off; 200 measured best of 0 / 200 / 400 / 800, profiles/r05_probe_cnhw_fill.log)
int g_conv_big_xswz = 7;  // cot_set_tuning key 48: bit 0 = BIG tiles: bank-conflict-free (XOR-permuted) X stage (0: rows stored as they lie in memory); bit 1 = W tile: the permutation that is conflict-free under the hardware's ds_read_b128 lane groups (0: rounds 2-4's); bit 2 = the transposed W tile's (data gradient), see wt_perm
int g_conv_flat_ns3 = 1;  // cot_set_tuning key 43: FLAT 128-row tiles take three stages instead of six when the launch exceeds one workgroup per CU
int g_conv_ablate = 0;  // cot_set_tuning key 24 (diagnostic: see C1LdsArgs::ablate)
int g_conv_k_tail = 1;  // cot_set_tuning key 54: reduction depths that are multiples of 8 but not of 32 on these kernels (KT instantiations); 0 = the first-generation / general kernels as before round 6

// Chunk permutation of the TRANSPOSED weight tile (WT kernels: k rows of CPR 16-byte channel chunks): position p of k row `row`
// holds chunk p ^ wt_perm(row).  A half-wave's transposing read touches the eight rows 8g + q (g = 0, 1 or 2, 3; q = 0..3), 32
// bytes of each; rows alias in the banks every 256 B / (16 B * CPR) rows, so the rows that alias must take different 32-byte windows:
// window = ((g & 1) << 2 | q) >> (3 - log2(CPR / 2)).  fixed = 0: rounds 2-4's form ((g << 2 | q) & (CPR - 1)), which is that for
// CPR = 8 but lets rows q and q ^ 1 share a window at CPR = 16 and rows of g = 0 / 1 at CPR = 4 (SQ_LDS_BANK_CONFLICT 0.47 / 0.41 of
// SQ_LDS_IDX_ACTIVE in the data-gradient instances, profiles/r05_step_lds_conflicts_pmc.csv).  (row + 4: the same value.)
template <int CPR> __device__ __forceinline__ int wt_perm(int row, int fixed) {
    const int g1 = (row >> 3) & 1, q = row & 3;
    if (!fixed) return ((((row >> 3) & 3) << 2) | q) & (CPR - 1);
    constexpr int SH = CPR >= 16 ? 0 : (CPR == 8 ? 1 : 2);
    return ((((g1 << 2) | q) >> SH) << 1) & (CPR - 1);
}

// template parameters as conv1x1_lds_fwd (conv_lds.hip); PF = fragment prefetch (register double buffer)
// KT = 1: the reduction depth K is a multiple of 8 but not of 32 (CoXtLayer's grouped 1x1s: 48 / 24 / 216 / 432 channels per group,
// models/cotnet.py:118-135).  The last K step then has kt = K % 32 valid rows: its copies take their X rows / W rows (WT) or W k-chunks from
// INSIDE the operands (row min(r, kt - 1), chunk 0 -- nothing past a tensor's end is touched, whatever lies behind it) and the lanes
// holding k >= kt (lane group g with 8 g >= kt) clear their A fragment, so those products are exact zeros.  One input slab, weights in
// place.  KT = 0 instantiations carry none of this.
template <int CB, int MB, int FLAT, int NS, int WAVES, int TRD, int WT, int PF, int KT = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void conv1x1_lds_fwd2(const C1LdsArgs a) {
    constexpr int NT = 64 * WAVES;
    constexpr int BPX = 16 * WAVES * CB, BM = 16 * MB, BK = 32;
    constexpr int XP = (BK * BPX / 8 + NT - 1) / NT;     // X copies per thread and stage (full passes of NT x 16 B)
    constexpr int WPASS = (BM * 4 + NT - 1) / NT;        // W copies per thread and stage
    constexpr int XST = XP * NT * 8, WST = WPASS * NT * 8;  // stage sizes in elements (padded to whole passes)
    constexpr int G = XP + WPASS;
    static_assert(XP >= 1 && (NS - 1) * G <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int HW = a.HW, K = a.K, M = a.M;
    bf16_t* const wsm = reinterpret_cast<bf16_t*>(cot_smem);  // [NS][WST] then [NS][XST]
    bf16_t* const xsm = wsm + NS * WST;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;

    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    // (integer divisions are expanded into vector code: the explicit readfirstlanes keep what derives from them scalar)
    const int mb = uniform((int)(b % (unsigned)a.mblocks));  // consecutive (same-XCD) workgroups share the X tile
    const int t = uniform((int)(b / (unsigned)a.mblocks));
    int n0, p0, ncols;  // first image, first pixel, valid columns of this tile
    if (FLAT) {
        n0 = t * a.ni;
        p0 = 0;
        ncols = min(a.ni, a.N - n0) * HW;
    } else {
        n0 = uniform(t / a.ptiles);
        p0 = (t - n0 * a.ptiles) * BPX;
        ncols = min(BPX, HW - p0);
    }
    const int m0 = mb * BM;
    const int kt = KT ? (K & (BK - 1)) : 0;  // valid k rows of the last step (KT: never 0 -- the host picks the instantiation)

    // ---- staging.  Per-lane byte offsets of this thread's copies relative to the step's scalar base (image n0, channel k0 of
    // the slab the step reads): resolved once.  Every wave issues exactly G copies per stage (lanes past a stage's data copy
    // in-bounds bytes into the stage's padding), so one vmcnt arithmetic holds for all waves.
    const int cpi = BK * HW / 8, xtotal = FLAT ? a.ni * cpi : BK * BPX / 8;
    unsigned xvA[XP], xvB[XP];
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int q = min(ps * NT + tid, xtotal - 1);
        if (FLAT) {
            const int img = q / cpi, c = q - img * cpi;
            const int nrel = min(n0 + img, a.N - 1) - n0;  // images past the batch: in-bounds bytes, never stored
            xvA[ps] = (unsigned)((int64_t)nrel * a.xs1 + c * 8) * 2u;
            if (KT) {  // (one slab: the second slab's offsets serve the last K step -- chunks past its kt rows re-read chunks inside them)
                const int vc = kt * HW / 8;  // (kt % 8 == 0: whole chunks)
                xvB[ps] = (unsigned)((int64_t)nrel * a.xs1 + (c < vc ? c : c % vc) * 8) * 2u;
            } else {
                xvB[ps] = (unsigned)((int64_t)nrel * a.xs2 + c * 8) * 2u;
            }
        } else {
            constexpr int cpr = BPX / 8;  // chunks per row
            const int row = q / cpr, pos = q - row * cpr;
            // xswz: position `pos` of k row `row` holds pixel chunk pos ^ 2*((row & 3) | ((row >> 3) & 1) << 2).  A stage row is
            // 256 B = one sweep of the 64 banks, so the eight k rows a half-wave's transposing read touches (8g + 0..3, g = 0, 1
            // or 2, 3; the same 32-byte column window in each) fell on the SAME eight banks: eight LDS cycles instead of one
            // (SQ_LDS_BANK_CONFLICT 0.6 of SQ_LDS_IDX_ACTIVE, profiles/r03_conv_sq_counters.txt).  Permuted, they take eight
            // different 32-byte windows; a row's chunks still come from the same 256 contiguous bytes of global memory.
            const int c = ((a.xswz & 1) && TRD) ? pos ^ (2 * ((row & 3) | (((row >> 3) & 1) << 2))) : pos;
            int pc = p0 + c * 8;
            if (pc + 8 > HW) pc = 0;  // partial last tile: columns never stored; any in-bounds bytes will do
            xvA[ps] = (unsigned)(row * HW + pc) * 2u;
            xvB[ps] = KT ? (unsigned)(min(row, kt - 1) * HW + pc) * 2u : xvA[ps];
        }
    }
    const bf16_t* const xbaseA = a.x1 + (int64_t)n0 * a.xs1;
    const bf16_t* const xbaseB = a.x2 ? a.x2 + (int64_t)n0 * a.xs2 : a.x1;
    unsigned wv[WPASS], wvT[KT ? WPASS : 1];  // wvT: the last K step's (KT)
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
        const int q = min(ps * NT + tid, BM * 4 - 1);
        if (WT) {
            constexpr int CPR = BM / 8;  // 16-byte chunks per k row
            const int row = q / CPR, pos = q - row * CPR;
            const int c = pos ^ wt_perm<CPR>(row, a.xswz & 4);  // position `pos` holds channel chunk c
            int mcol = m0 + c * 8;
            if (mcol + 8 > M) mcol = M - 8;  // channels past M (M % 8 == 0): in-bounds bytes, never stored
            wv[ps] = (unsigned)(row * M + mcol) * 2u;
            if (KT) wvT[ps] = (unsigned)(min(row, kt - 1) * M + mcol) * 2u;
        } else {
            const int row = q >> 2, pos = q & 3;
            // XOR permutation: position `pos` of a row holds its k-chunk c.  (-(row >> 2)) & 3 is the form under which the four lane
            // groups a ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md) touch 16 different
            // 16-byte slots; (row >> 2) & 3 (rounds 2-4; key 48 bit 1 = 0) makes rows 0-3 / 4-7 and 8-11 / 12-15 of a group collide:
            // SQ_LDS_BANK_CONFLICT 0.45 of SQ_LDS_IDX_ACTIVE in the forward kernels (profiles/r05_conv1x1_lds_conflicts_pmc.log)
            const int c = pos ^ ((a.xswz & 2) ? (-(row >> 2)) & 3 : (row >> 2) & 3);
            const int m = min(m0 + row, M - 1);    // rows past M: a copy of row M-1, never stored
            wv[ps] = (unsigned)(a.wpacked ? m * 32 + c * 8 : m * K + c * 8) * 2u;
            if (KT) wvT[ps] = (unsigned)(m * K + (8 * c < kt ? c : 0) * 8) * 2u;
        }
    }
    const int64_t wstep = WT ? (int64_t)M * 32 : (a.wpacked ? (int64_t)M * 32 : 32);  // elements from one K step's W tile to the next
    const int nk = KT ? (K + BK - 1) / BK : K / BK;
    const unsigned lds_w = COT_LDS_ADDR(wsm) + (unsigned)(wave * 64 * 16);
    const unsigned lds_x = COT_LDS_ADDR(xsm) + (unsigned)(wave * 64 * 16);
    // stage `s` (K step s) into ring slot `buf`; both wave-uniform
    auto stage = [&](int s, int buf) __attribute__((always_inline)) {
        const int k0 = s * BK;
        const bool first = k0 < a.k1;  // the K step's rows come from one slab (k1 % 32 == 0 is checked on the host)
        const bool tl = KT && s == nk - 1;
        const bf16_t* xb = first ? xbaseA + (int64_t)k0 * HW : xbaseB + (int64_t)(k0 - a.k1) * HW;
        const unsigned xd = lds_x + (unsigned)(buf * XST * 2);
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) COT_GLDS16S(xb, (first && !tl) ? xvA[ps] : xvB[ps], xd + (unsigned)(ps * NT * 16));
        const bf16_t* wb = a.w + s * wstep;
        const unsigned wd = lds_w + (unsigned)(buf * WST * 2);
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) COT_GLDS16S(wb, (KT && tl) ? wvT[KT ? ps : 0] : wv[ps], wd + (unsigned)(ps * NT * 16));
    };

    // ---- per-lane LDS offsets of the A (= X^T) gathers: column -> element offset of (k = 0, column) inside a stage
    int aoff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        // TRD: this lane points at row 8g + (i16 >> 2) (+4 for the second read), columns 4*(i16 & 3) .. +3 of the block
        const int col = ((tid >> 6) * CB + cb) * 16 + (TRD ? 4 * (i16 & 3) : i16);
        const int row = TRD ? 8 * g + (i16 >> 2) : 8 * g;
        if (FLAT) {
            const int cc = min(col, a.ni * HW - (TRD ? 4 : 1));  // columns past the tile: any staged element
            const int img = cc / HW;
            aoff[cb] = img * BK * HW + (cc - img * HW) + row * HW;
        } else if (TRD && (a.xswz & 1)) {
            const int sw = 2 * ((row & 3) | (((row >> 3) & 1) << 2));  // (row + 4, the second read: the same permutation)
            aoff[cb] = (((col >> 3) ^ sw) << 3) + (col & 7) + row * BPX;
        } else {
            aoff[cb] = col + row * BPX;
        }
    }
    const int rs = FLAT ? HW : BPX;  // row (= k) stride of the X stage
    int boff[MB];                    // B (= W) fragments: row i16 of channel block mbk, k-chunk g (swizzled position)
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        if (WT) {  // k row 8g + (i16 >> 2) (+4 for the second read), channels mbk*16 + 4*(i16 & 3) .. +3
            constexpr int CPR = BM / 8;
            const int row = 8 * g + (i16 >> 2), c = 2 * mbk + ((i16 & 3) >> 1);
            const int pos = c ^ wt_perm<CPR>(row, a.xswz & 4);
            boff[mbk] = row * BM + pos * 8 + (i16 & 1) * 4;
        } else {
            const int row = mbk * 16 + i16;
            boff[mbk] = row * BK + (g ^ ((a.xswz & 2) ? (-(row >> 2)) & 3 : (row >> 2) & 3)) * 8;
        }
    }

    f32x4_t acc[CB][MB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) acc[cb][mbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((ext_vector_type(2))) uint16_t u16x2_t;
    auto read_frags = [&](int buf, bf16x8_t (&af)[CB], bf16x8_t (&bfr)[MB], bool tl) __attribute__((always_inline)) {
        const uint16_t* xb = reinterpret_cast<const uint16_t*>(xsm + buf * XST);
        const bf16_t* wb = wsm + buf * WST;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const uint16_t* p = xb + aoff[cb];
            if (TRD) {
                s16x4_t lo = COT_LDS_READ_TR16(p), hi = COT_LDS_READ_TR16(p + 4 * rs);
                __builtin_memcpy(&af[cb], &lo, 8);
                __builtin_memcpy(reinterpret_cast<char*>(&af[cb]) + 8, &hi, 8);
            } else {
                u16x2_t q4[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    q4[h][0] = p[(2 * h) * rs];
                    q4[h][1] = p[(2 * h + 1) * rs];
                }
                __builtin_memcpy(&af[cb], q4, 16);
            }
            if (KT && tl && 8 * g >= kt) __builtin_memset(&af[cb], 0, 16);  // (this lane's eight k of the last step lie past K)
        }
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) {
            if (WT) {
                const uint16_t* p = reinterpret_cast<const uint16_t*>(wb) + boff[mbk];
                s16x4_t lo = COT_LDS_READ_TR16(p), hi = COT_LDS_READ_TR16(p + 4 * BM);
                __builtin_memcpy(&bfr[mbk], &lo, 8);
                __builtin_memcpy(reinterpret_cast<char*>(&bfr[mbk]) + 8, &hi, 8);
            } else {
                __builtin_memcpy(&bfr[mbk], __builtin_assume_aligned(wb + boff[mbk], 16), 16);
            }
        }
    };
    const int abl = a.ablate;  // (diagnostic switches, all zero in production: scalar branches)
    auto multiply = [&](const bf16x8_t (&af)[CB], const bf16x8_t (&bfr)[MB]) __attribute__((always_inline)) {
        if (abl & 4) {
            acc[0][0] = COT_MFMA_16X16X32_BF16(af[0], bfr[0], acc[0][0]);
            return;
        }
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb][mbk] = COT_MFMA_16X16X32_BF16(af[cb], bfr[mbk], acc[cb][mbk]);
    };

    if (PF) {
        // Ring of NS slots, stage s in slot s % NS.  Iteration ks multiplies the fragments of step ks, which were read into
        // registers one iteration earlier.  Its barrier therefore says two things: every wave's copies of stage ks+1 have
        // landed (each waited for its own before arriving), and nobody reads slot ks % NS any more (each wave's fragment
        // reads of step ks completed -- lgkmcnt(0) -- before it arrived).  So behind the barrier slot ks % NS is re-filled
        // with stage ks+NS and the fragments of step ks+1 are fetched while the MFMAs of step ks run.
        // vmcnt: before iteration ks the stages 0 .. min(nk, ks+NS) - 1 have been issued; those younger than ks+1 number
        // min(NS-2, nk-2-ks) and stay in flight.
#pragma unroll
        for (int s0 = 0; s0 < NS; ++s0)
            if (s0 < nk) stage(s0, s0);
        WaitBehind<G, NS - 1>::go(min(NS - 1, nk - 1));
        COT_LDS_BARRIER();
        bf16x8_t a0[CB], b0[MB], a1[CB], b1[MB];
        read_frags(0, a0, b0, KT && nk == 1);
        if (abl & 2) read_frags(0, a1, b1, false);
        int slot = 0;  // ring slot of step ks
        // STEADY: ks + NS < nk is known (full wait count, the re-fill always happens): the loop body carries no conditions
        auto step = [&](auto steady, int ks, const bf16x8_t (&ac)[CB], const bf16x8_t (&bc)[MB], bf16x8_t (&an)[CB],
                        bf16x8_t (&bn)[MB]) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady)::value;
            if (STEADY || ks + 1 < nk) {
                if (!(abl & 16)) {
                    if (STEADY) COT_WAIT_VM((NS - 2) * G);
                    else WaitBehind<G, NS - 2>::go(min(NS - 2, nk - 2 - ks));
                }
                if (!(abl & 8)) COT_LDS_BARRIER();
                if ((STEADY || ks + NS < nk) && !(abl & 1)) stage(ks + NS, slot);
                slot = slot + 1 == NS ? 0 : slot + 1;
                if (!(abl & 2)) read_frags(slot, an, bn, KT && ks + 2 == nk);
                COT_SCHED_FENCE();  // the reads are in flight BEFORE the multiplies start (they hide the LDS round trip)
            }
            multiply(ac, bc);
            COT_SCHED_FENCE();
        };
        int ks = 0;
        for (; ks + 1 + NS < nk; ks += 2) {
            step(std::true_type{}, ks, a0, b0, a1, b1);
            step(std::true_type{}, ks + 1, a1, b1, a0, b0);
        }
        for (; ks < nk; ks += 2) {
            step(std::false_type{}, ks, a0, b0, a1, b1);
            if (ks + 1 < nk) step(std::false_type{}, ks + 1, a1, b1, a0, b0);
        }
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (s0 < nk) stage(s0, s0);
        int slot = 0, fill = NS - 1;  // slot of step ks; slot the next stage goes to
        for (int ks = 0; ks < nk; ++ks) {
            if (!(abl & 16)) WaitBehind<G, NS - 2>::go(min(NS - 2, nk - 1 - ks));  // this wave's copies of stage ks have landed
            if (!(abl & 8)) COT_LDS_BARRIER();                // everybody's have; nobody still reads stage ks-1's slot
            if (ks + NS - 1 < nk && !(abl & 1)) stage(ks + NS - 1, fill);
            fill = fill + 1 == NS ? 0 : fill + 1;
            bf16x8_t af[CB], bfr[MB];
            read_frags((abl & 2) ? 0 : slot, af, bfr, KT && ks + 1 == nk);
            slot = slot + 1 == NS ? 0 : slot + 1;
            multiply(af, bfr);
        }
    }

    EpiArgs e;
    e.y1 = a.y1; e.y2 = a.y2; e.bias = a.bias; e.m1 = a.m1; e.M = M; e.HW = HW; e.N = a.N; e.ni = a.ni;
    e.ys1 = a.ys1; e.ys2 = a.ys2; e.stats = a.stats; e.ptiles = a.ptiles;
    e.n0 = n0; e.p0 = p0; e.m0 = m0; e.mv = min(BM, M - m0); e.ncols = ncols; e.accumulate = a.accumulate;
    e.ablate = a.ablate; e.acc_src = a.acc_src; e.acc_mask = a.acc_mask;
    tile_epilogue<CB, MB, FLAT, WAVES>(acc, e);
}

template <int CB, int MB, int FLAT, int NS, int WAVES, int TRD, int WT, int PF, int KT>
static int launch_c1v2(const C1LdsArgs& a, int tiles, hipStream_t stream) {
    constexpr int NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB;
    constexpr int XST = ((32 * BPX / 8 + NT - 1) / NT) * NT * 8, WST = ((BM * 4 + NT - 1) / NT) * NT * 8;
    size_t lds = (size_t)NS * (XST + WST) * sizeof(bf16_t);
    const size_t otile = (FLAT ? (size_t)a.ni * (((size_t)BM * a.HW + 7) & ~(size_t)7) : (size_t)BM * (BPX + 8)) * sizeof(bf16_t);
    if (otile > lds) lds = otile;
    const int64_t blocks = (int64_t)tiles * a.mblocks;
    C1LdsArgs b = a;
    b.xcd_remap = (blocks % 8 == 0) ? 1 : 0;
    static std::atomic<uint32_t> raised{0};
    if (lds > 64 * 1024 &&
        !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv1x1_lds_fwd2<CB, MB, FLAT, NS, WAVES, TRD, WT, PF, KT>)))
        return -1;  // (the caller falls back to a kernel that needs no large LDS window)
    COT_LAUNCH((conv1x1_lds_fwd2<CB, MB, FLAT, NS, WAVES, TRD, WT, PF, KT>), dim3((unsigned)blocks), dim3(NT), lds, stream, b);
    return check_launch("conv1x1_lds_fwd2");
}

// same contract and dispatch rules as conv1x1_lds_gemm (conv_lds.hip), which calls this unless tuning key 23 bit 0 is set
int conv1x1_lds_gemm2(const C1LdsArgs& a0, hipStream_t stream) {
    C1LdsArgs a = a0;
    a.ablate = g_conv_ablate;
    a.xswz = g_conv_big_xswz;
    const int N = a.N, M = a.M, HW = a.HW;
    // per-lane offsets are 32-bit: a workgroup's images / the weight rows must lie within 2 GB of the scalar bases
    const int64_t slab = std::max(a.xs1, a.xs2) * 2;  // bytes from one image to the next
    if ((HW <= 256 ? slab * (256 / HW + 1) : slab) >= ((int64_t)1 << 31) || (int64_t)M * a.K * 2 >= ((int64_t)1 << 31)) return -1;
    const bool wt = a.wpacked == 2;
    const bool ktl = a.K % 32 != 0;  // K tail (KT instantiations): one slab, weights read in place
    if (ktl && (a.K % 8 != 0 || a.x2 || a.wpacked == 1 || a.k1 != a.K)) return -1;
    const bool pf_flat = !((g_conv_lds2_tune >> 1) & 1), pf_big = (g_conv_lds2_tune >> 2) & 1;
#define COT_C2W(CB_, MB_, FLAT_, NS_, TR_, PF_)                                                            \
    return ktl ? (wt ? launch_c1v2<CB_, MB_, FLAT_, NS_, 8, TR_, 1, PF_, 1>(a, tiles, stream)              \
                     : launch_c1v2<CB_, MB_, FLAT_, NS_, 8, TR_, 0, PF_, 1>(a, tiles, stream))             \
               : (wt ? launch_c1v2<CB_, MB_, FLAT_, NS_, 8, TR_, 1, PF_, 0>(a, tiles, stream)              \
                     : launch_c1v2<CB_, MB_, FLAT_, NS_, 8, TR_, 0, PF_, 0>(a, tiles, stream))
#define COT_C2(CB_, MB_, FLAT_, NS_)                                                                       \
    do {                                                                                                   \
        if (pf) {                                                                                          \
            if (tr) COT_C2W(CB_, MB_, FLAT_, NS_, 1, 1);                                                   \
            COT_C2W(CB_, MB_, FLAT_, NS_, 0, 1);                                                           \
        }                                                                                                  \
        if (tr) COT_C2W(CB_, MB_, FLAT_, NS_, 1, 0);                                                       \
        COT_C2W(CB_, MB_, FLAT_, NS_, 0, 0);                                                               \
    } while (0)
    if (a.stats && HW <= 256) return -1;  // (epilogue statistics: 128-pixel tiles only)
    if (HW > 256) {  // BIG: 128-pixel tiles of one image; three stages, several workgroups per CU
        a.ptiles = ceil_div(HW, 128);
        const int tiles = N * a.ptiles;
        const bool tr = true, pf = pf_big;
        if (M <= 32) { a.mblocks = 1; COT_C2(1, 2, 0, 3); }
        if (M <= 64) { a.mblocks = 1; COT_C2(1, 4, 0, 3); }
        // few tiles (round 5: the channel-major deep layers are ONE image of N*HW pixels -- 31 tiles at 7 x 7, 123 at 14 x 14, B = 80):
        // narrower output-channel blocks multiply the workgroups; these launches are bound by one workgroup's chain of K steps, not
        // by the bytes the narrower blocks re-read (from L2)
        if (g_conv_big_fill > 0 && (int64_t)tiles * ceil_div(M, 128) < g_conv_big_fill) {
            if ((int64_t)tiles * ceil_div(M, 64) < g_conv_big_fill && M % 32 == 0) { a.mblocks = ceil_div(M, 32); COT_C2(1, 2, 0, 3); }
            a.mblocks = ceil_div(M, 64);
            COT_C2(1, 4, 0, 3);
        }
        a.mblocks = ceil_div(M, 128);
        COT_C2(1, 8, 0, 3);
    }
    // FLAT: whole images, up to 256 columns per workgroup, up to 128 channels; six stages
    int ni = g_conv_lds_tune[1] > 0 ? g_conv_lds_tune[1] : 256 / HW;
    if (ni > N) ni = N;
    if (ni < 1 || ni * HW > 256) return -1;
    a.ni = ni;
    a.ptiles = ceil_div(N, ni);
    const int tiles = a.ptiles;
    const bool tr = !((g_conv_lds_tune[2] >> 1) & 1) && HW % 4 == 0, pf = pf_flat;  // (key 17 bit 1: 2-byte gathers, A/B and tests)
    if (M <= 32) { a.mblocks = 1; COT_C2(2, 2, 1, 6); }
    if (M <= 64) { a.mblocks = 1; COT_C2(2, 4, 1, 6); }
    // few tiles (the 7 x 7 stage at B = 80): 64-channel blocks double the workgroups (conv_lds.hip, profiles/r02_conv_fewtiles_ab.log)
    const int fw = g_conv_lds_tune[2] >> 8;  // (cot_set_tuning key 17 bits 8..: as in conv_lds.hip)
    if (fw != 1 && tiles <= 32 && (int64_t)tiles * ceil_div(M, 128) < (fw > 1 ? fw : 200)) {
        a.mblocks = ceil_div(M, 64);
        COT_C2(2, 4, 1, 6);
    }
    a.mblocks = ceil_div(M, 128);
    // more than one round of workgroups at one per CU (six stages = 144 KB of LDS): three stages (72 KB) let two workgroups share a
    // CU -- 256 -> 1024 @14x14, B = 80: 640 workgroups, three rounds
    if (g_conv_flat_ns3 == 2 || (g_conv_flat_ns3 && (int64_t)tiles * a.mblocks > 256)) COT_C2(2, 8, 1, 3);  // (2: always -- tests)
    COT_C2(2, 8, 1, 6);
#undef COT_C2
#undef COT_C2W
}

}  // namespace cot
