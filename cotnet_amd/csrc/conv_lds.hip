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
#include "conv_lds_common.h"

namespace cot {
extern int g_conv_ablate;  // (conv_lds2.hip; cot_set_tuning key 24)

extern int g_conv_lds_tune[3];
extern int g_conv_lds2_tune;
extern int g_conv_k_tail;  // (conv_lds2.hip; cot_set_tuning key 54)
int conv1x1_lds_gemm2(const C1LdsArgs& a0, hipStream_t stream);  // conv_lds2.hip

// dst[c][r] = src[r][c]   (R x C row-major -> C x R row-major), 32x32 tiles through LDS.
// pack = 1: the K-step-major form the LDS kernels read for the data gradient: dst[((r/32)*C + c)*32 + r%32] = src[r][c],
// i.e. for every block of 32 reduction rows r (= output channels of the convolution) a contiguous [C][32] image -- a K
// step's W tile is then ONE contiguous range instead of C segments of 64 bytes that are R*2 bytes apart (with R*2 = 4096
// all of them sit on one L2 channel: measured 2x slower than the same GEMM with a 1 KB row stride).  R % 32 == 0.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                            int R, int C, int pack) {
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
        if (r < R && c < C) dst[pack ? ((int64_t)blockIdx.y * C + c) * 32 + tx : (int64_t)c * R + r] = tile[tx][ty + 8 * i];
    }
}

int transpose_bf16(const void* src, void* dst, int R, int C, int pack, hipStream_t stream) {
    COT_LAUNCH(transpose_bf16_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, stream, (const bf16_t*)src,
               (bf16_t*)dst, R, C, pack);
    return check_launch("transpose_bf16_kernel");
}

// One wave = CB x 16 columns (pixels) x MB x 16 channels; the WAVES waves side by side along the pixels (BPX = 16*WAVES*CB
// columns per workgroup), every wave computes all BM = 16*MB channels of its columns: the 2-byte gathers of the X fragments are done
// once per workgroup, the cheap 16-byte W fragment reads four times.
// FLAT = 0: a tile is BPX consecutive pixels of one image (H*W % 8 == 0: rows are 16-byte multiples)
// FLAT = 1: a tile is `ni` whole images (ni * H*W <= BPX columns); the K step's 32 rows of an image are one flat range
// NS = LDS stages: stages ks+1 .. ks+NS-2 are in flight while stage ks is multiplied, one barrier per K step.
// WAVES = 4 or 8 waves per workgroup.  With 8 (two per SIMD, from the same workgroup) one wave's LDS round trips -- the
// fragment gathers of a K step, ~10 dependent batches -- hide behind the other's MFMAs; with 4 and one workgroup per CU
// (the deep-K layers) they are fully exposed: measured 1.2 us per K step for 0.25 us of MFMA work.
// TRD = 1: the X fragments come from two transposing reads (needs 4-column groups that are 8-byte aligned and inside one
// image: always true for BIG, H*W % 4 == 0 for FLAT); TRD = 0: eight 2-byte reads (any H*W, e.g. 7 x 7).
// WT = 1: the weight operand is given TRANSPOSED, [K][M] row-major -- the data gradient reads the forward's [Co][Ci] weight
// tensor in place (K = Co, M = Ci) instead of a transposed copy made by a launch of its own (50 launches / 0.5 ms per
// CoTNet-50 step).  A K step's tile is then 32 rows of BM consecutive channels, staged as it lies in memory, and the W
// fragments (8 consecutive k of one channel) come from the same transposing reads as the X fragments.  The 16 rows one such
// read touches are BM*2 bytes apart -- the same banks -- so the 16-byte chunks of a row are stored XOR-permuted by
// s(row) = 4*((row >> 3) & 3) + (row & 3), which sends those 16 rows to 16 different chunk positions (BM = 128).
template <int CB, int MB, int FLAT, int NS, int WAVES, int TRD, int WT>
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
        if (WT) {
            constexpr int CPR = BM / 8;  // 16-byte chunks per k row
            const int row = q / CPR, pos = q - row * CPR;
            const int c = pos ^ (((((row >> 3) & 3) << 2) | (row & 3)) & (CPR - 1));  // position `pos` holds channel chunk c
            int mcol = m0 + c * 8;
            if (mcol + 8 > M) mcol = M - 8;  // channels past M (M % 8 == 0): in-bounds bytes, never stored
            wsrc[ps] = a.w + (int64_t)row * M + mcol;
        } else {
            const int row = q >> 2, pos = q & 3;
            const int c = pos ^ ((row >> 2) & 3);  // XOR swizzle: position `pos` of a row holds its k-chunk c
            const int m = min(m0 + row, M - 1);    // rows past M: a copy of row M-1, never stored
            wsrc[ps] = a.wpacked ? a.w + (int64_t)m * 32 + c * 8 : a.w + (int64_t)m * K + c * 8;
        }
    }
    const int64_t wstep = WT ? (int64_t)M * 32 : (a.wpacked ? (int64_t)M * 32 : 32);  // elements from one K step's W tile to the next
    // Workgroups walk K from different starting steps (cyclically): otherwise all of them read the same 64-byte column
    // of W -- 128 segments that are K*2 bytes apart, i.e. (K >= 2048) ONE L2 channel -- at about the same time.
    const int nk = K / BK;
    const int ks0 = (int)(((unsigned)t * 5u + (unsigned)mb * 3u) % (unsigned)nk);
    auto stage = [&](int ksl) __attribute__((always_inline)) {
        int ks = ksl + ks0;
        if (ks >= nk) ks -= nk;
        const int k0 = ks * BK, buf = ksl % NS;
        bf16_t* xd = xsm + buf * XST;
        const bool first = k0 < a.k1;  // the K step's rows come from one slab (k1 % 32 == 0 is checked on the host)
        const bf16_t* xbase = first ? a.x1 + (int64_t)k0 * HW : a.x2 + (int64_t)(k0 - a.k1) * HW;
        const int64_t istride = (int64_t)(first ? a.k1 : K - a.k1) * HW;  // elements per image of that slab
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) COT_GLDS16(xbase + xn[ps] * istride + xin[ps], xd + (ps * NT + wave * 64) * 8);
        bf16_t* wd = wsm + buf * WST;
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) COT_GLDS16(wsrc[ps] + ks * wstep, wd + (ps * NT + wave * 64) * 8);
    };

    // ---- per-lane LDS offsets of the A (= X^T) gathers: column -> element offset of (k = 0, column) inside a stage
    int aoff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        // TRD: this lane points at row 8g + (i16 >> 2) (+4 for the second read), columns 4*(i16 & 3) .. +3 of the block
        const int col = (wave * CB + cb) * 16 + (TRD ? 4 * (i16 & 3) : i16);
        const int row = TRD ? 8 * g + (i16 >> 2) : 8 * g;
        if (FLAT) {
            const int cc = min(col, a.ni * HW - (TRD ? 4 : 1));  // columns past the tile: any staged element
            const int img = cc / HW;
            aoff[cb] = img * BK * HW + (cc - img * HW) + row * HW;
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
            const int pos = c ^ ((((g & 3) << 2) | (i16 >> 2)) & (CPR - 1));
            boff[mbk] = row * BM + pos * 8 + (i16 & 1) * 4;
        } else {
            const int row = mbk * 16 + i16;
            boff[mbk] = row * BK + (g ^ ((row >> 2) & 3)) * 8;
        }
    }

    f32x4_t acc[CB][MB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) acc[cb][mbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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
        }
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) {
            bf16x8_t bf;
            if (WT) {
                const uint16_t* p = reinterpret_cast<const uint16_t*>(wb) + boff[mbk];
                s16x4_t lo = COT_LDS_READ_TR16(p), hi = COT_LDS_READ_TR16(p + 4 * BM);
                __builtin_memcpy(&bf, &lo, 8);
                __builtin_memcpy(reinterpret_cast<char*>(&bf) + 8, &hi, 8);
            } else {
                __builtin_memcpy(&bf, __builtin_assume_aligned(wb + boff[mbk], 16), 16);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[cb][mbk] = COT_MFMA_16X16X32_BF16(af[cb], bf, acc[cb][mbk]);
        }
    }

    EpiArgs e;
    e.y1 = a.y1; e.y2 = a.y2; e.bias = a.bias; e.m1 = a.m1; e.M = M; e.HW = HW; e.N = a.N; e.ni = a.ni;
    e.ys1 = a.ys1; e.ys2 = a.ys2; e.stats = nullptr; e.ptiles = 0;
    e.n0 = n0; e.p0 = p0; e.m0 = m0; e.mv = min(BM, M - m0); e.ncols = ncols; e.accumulate = a.accumulate;
    e.ablate = 0; e.acc_src = nullptr; e.acc_mask = nullptr;
    tile_epilogue<CB, MB, FLAT, WAVES>(acc, e);
}

// cot_conv3x3g_pack / cot_conv3x3g_*_packed (round 5): the weights of a layer change once per optimizer step, its two packings (forward,
// data gradient) are each used once per step -- packing ahead of time saves no work but takes 32 small launches off the compute
// stream's critical path (the host side packs on a side stream right after the optimizer step).  0 = pack and run (the ordinary entry
// points), 1 = pack only, 2 = run on a packing made by an identical call in mode 1.
thread_local int t_c3_pack = 0;

// ====================================================================================================================
// Grouped 3x3 convolution (stride 1, padding 1) -- CotLayer.key_embed[0] (models/cotnet.py:43-47; groups 4; CoXtLayer 8) --
// on the same machinery: per group an implicit GEMM  Y (MM x HW) = sum over (tap, ci) Wr[tap][co][ci] * X[ci][p + off(tap)].
//   * the group's input channels are staged ONCE per 32-channel chunk (all 9 taps read the same staged rows at shifted
//     positions): BIG = a tile of TR image rows plus one halo row above and below, each channel's rows one contiguous range
//     (like the aggregation kernels' slabs); FLAT = whole small images, flat;
//   * a K step = (channel chunk, tap): 32 channels of one tap (KK >= 32), or two taps x 16 channels (KK == 16; the tenth
//     "tap" of the fifth step is a block of zeros in the repacked weights).  The A fragment (pixels x 8 channels) is the
//     same 2-byte gather as in the 1x1 kernel at `+ dy*W + dx`; positions whose neighbour lies outside the image read a
//     neighbouring row / channel / image and are zeroed by SELECTION with a per-lane 9-bit validity mask;
//   * weights are repacked per call into [group][tap][co][ci] (ci contiguous; data gradient: [tap][ci][co] with the taps
//     flipped) by a small kernel into the call's workspace (<= 1.2 MB), so W fragments are 16-byte rows as in the 1x1 case.
// dst[((g*NT + tap)*Mo + mo)*Kk + kk] = forward:  w[((g*Mo + mo)*Kk + kk)*9 + tap]
//                                        dgrad:    w[((g*Kk + kk)*Mo + mo)*9 + (8 - tap)]     (tap == 9: zeros)
// Kp >= Kk: the K dimension of the repacked tiles, padded with ZEROS to the kernel's 32-channel chunks (groups of 24 / 48 channels:
// CoXtLayer's key embedding, round 4) -- the staged chunk then also holds channels of the NEXT group, which meet zero weights.
// MBLK > 1: a group's output channels are cut into MBLK blocks of Mo rows ("virtual groups" g' = g*MBLK + blk that share the
// real group's input): dense 3x3 convolutions with more than 128 output channels per group (SE-CoTNetD's SplitAttn convs).
__global__ void conv3x3g_repack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ dst, int G, int Mo, int Kk, int Kp, int NTAP,
                                       int dgrad, int MBLK, int perm) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (int64_t)G * NTAP * Mo * Kp;
    if (o >= total) return;
    int kk = (int)(o % Kp);
    if (perm) {  // position 8 g + j of a 32-channel chunk holds channel 4 j + g (C3LdsArgs::perm)
        const int r = kk & 31;
        kk = (kk & ~31) + 4 * (r & 7) + (r >> 3);
    }
    int64_t r = o / Kp;
    const int mo = (int)(r % Mo);
    r /= Mo;
    const int tap = (int)(r % NTAP), g = (int)(r / NTAP);
    bf16_t v = (bf16_t)0.0f;
    if (tap < 9 && kk < Kk)  // (g counts virtual groups: real group g / MBLK, block g % MBLK of its output rows)
        v = dgrad ? w[(((int64_t)(g / MBLK) * Kk + kk) * ((int64_t)Mo * MBLK) + (g % MBLK) * Mo + mo) * 9 + (8 - tap)]
                  : w[(((int64_t)g * Mo + mo) * Kk + kk) * 9 + tap];
    dst[o] = v;
}

struct C3LdsArgs {
    const bf16_t* x;   // [N][G*KX][H*W]
    const bf16_t* wr;  // repacked weights [G][NTAP][MM][KK]
    bf16_t* y;         // [N][G*MM][H*W]
    int N, G, KK, MM, H, W;
    int KX;            // channels per group in x (KK = KX rounded up to the 32-channel chunks; the padding meets zero weights)
    int MBLK, CX;      // output-row blocks per real group (G counts virtual groups = real groups x MBLK); channels per image of x
    int wsingle;       // chunk-resident form: ONE weight buffer (re-filled behind a barrier after each chunk) -- two workgroups per CU
    int perm;          // chunk-resident form: K position 8 g + j of a 32-channel chunk holds channel 4 j + g (else 8 g + j), in x's gathers
                       // and in the repacked weights alike: the four lane groups of a 2-byte gather then read CONSECUTIVE channel rows
    int accumulate;
    int tiles;         // BIG: row tiles per image; FLAT: image groups
    int ni;            // FLAT: images per workgroup
    int TR;            // BIG: image rows per tile
    int SL;            // BIG: staged elements per channel ((TR+2)*W + 8, rounded up to 8)
    int xcd_remap;
    int ablate;        // DIAGNOSTIC (cot_set_tuning key 24 bit 1; results become wrong): the K loop skips its LDS operand gathers
};

// K16: KK == 16 (two taps per K step, 5 steps); else KK % 32 == 0 (9 steps per 32-channel chunk).  XP = X copies per thread
// and chunk (compile-time so that the vmcnt arithmetic is uniform; the host picks TR / ni to fit).
// NSW: weight-tile ring (NSW - 1 steps' copies in flight ahead of the one being read).  The loop is bound by the round trip of
// those copies, not by its arithmetic: with 3 buffers a step took 1.2 - 1.5 us whatever its MFMA count (CoTNet-50's 14 x 14 and
// 7 x 7 key embeddings: 18 / 36 steps, 27 / 42 us per launch, profiles/r04_rocprofv3_kernel_trace_new_per_shape.csv).
template <int CB, int MB, int FLAT, int K16, int XP, int WAVES, int NSW>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3g_lds_fwd(const C3LdsArgs a) {
    constexpr int NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB;
    constexpr int CH = K16 ? 16 : 32;                 // channels per staged chunk
    constexpr int WPASS = (BM * 4 + NT - 1) / NT;
    constexpr int XST = XP * NT * 8, WST = WPASS * NT * 8, D = NSW - 1;
    constexpr int GW = WPASS, GX = XP;
    static_assert((D - 1) * GW + GX <= 63, "vmcnt range");
    static_assert(D >= 2 && D - 1 <= (K16 ? 5 : 9), "at most one X stage among the copies in flight");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* const wsm = reinterpret_cast<bf16_t*>(cot_smem);  // [NSW][WST] then [2][XST]
    bf16_t* const xsm = wsm + NSW * WST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W, KK = a.KK, MM = a.MM, G = a.G;

    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const int grp = b % G;
    const int t = b / G;
    int n0, r0 = 0, ncols, gs = 0;
    if (FLAT) {
        n0 = t * a.ni;
        ncols = min(a.ni, a.N - n0) * HW;
    } else {
        n0 = t / a.tiles;
        r0 = (t - n0 * a.tiles) * a.TR;
        ncols = min(a.TR, H - r0) * W;
        gs = max(0, (r0 - 1) * W) & ~7;  // first staged element of every channel (16-byte aligned)
    }
    const int SLc = FLAT ? HW : a.SL;                          // channel stride inside a staged chunk
    const int xelems = FLAT ? a.ni * CH * HW : CH * a.SL;      // elements of a staged chunk
    const int KX = a.KX;
    const int64_t x_total = (int64_t)a.N * a.CX * HW;
    const int xg = grp / a.MBLK;  // the real group: whose input channels this (virtual) group reads

    // ---- X copies of this thread (resolved once; a chunk step moves them CH channels = CH*HW elements)
    int64_t xoff[XP];
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int q = min(ps * NT + tid, xelems / 8 - 1);
        int64_t e;
        if (FLAT) {
            const int cpi = CH * HW / 8, img = q / cpi, c = q - img * cpi;
            e = ((int64_t)min(n0 + img, a.N - 1) * a.CX + (int64_t)xg * KX) * HW + (int64_t)c * 8;
        } else {
            const int cpc = a.SL / 8, ch = q / cpc, c = q - ch * cpc;
            e = ((int64_t)n0 * a.CX + (int64_t)xg * KX + ch) * HW + gs + c * 8;
        }
        xoff[ps] = e;
    }
    // ---- W copies: row = output channel of the group, 4 chunks of 8 k per row (swizzled position)
    int64_t woff[WPASS];
#pragma unroll
    for (int ps = 0; ps < WPASS; ++ps) {
        const int q = min(ps * NT + tid, BM * 4 - 1);
        const int row = q >> 2, pos = q & 3, c = pos ^ ((-(row >> 2)) & 3);
        const int rr = min(row, MM - 1);
        if (K16) woff[ps] = ((int64_t)(c >> 1) * MM + rr) * 16 + (c & 1) * 8;  // chunk c: tap 2t + (c>>1), channels 8*(c&1)..
        else woff[ps] = (int64_t)rr * KK + c * 8;
    }
    const int ncc = K16 ? 1 : KK / 32, spc = K16 ? 5 : 9, nsteps = ncc * spc;  // chunks, steps per chunk
    const bf16_t* wgrp = a.wr + (int64_t)grp * (K16 ? 10 : 9) * MM * KK;
    auto stage_x = [&](int cc) __attribute__((always_inline)) {
        bf16_t* xd = xsm + (cc & 1) * XST;
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            int64_t e = xoff[ps] + (int64_t)cc * CH * HW;
            if (e + 8 > x_total) e = x_total - 8;  // (BIG: the last tile's halo row past the tensor: in-bounds bytes, masked)
            COT_GLDS16(a.x + e, xd + (ps * NT + wave * 64) * 8);
        }
    };
    auto stage_w = [&](int s) __attribute__((always_inline)) {
        const int cc = s / spc, tp = s - cc * spc;
        const bf16_t* src = K16 ? wgrp + (int64_t)(2 * tp) * MM * 16 : wgrp + (int64_t)tp * MM * KK + cc * 32;
        bf16_t* wd = wsm + (s % NSW) * WST;
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) COT_GLDS16(src + woff[ps], wd + (ps * NT + wave * 64) * 8);
    };
    auto stage = [&](int s) __attribute__((always_inline)) {  // a step's copies: its W tile, and the next chunk's X with a chunk's first step
        if (s % spc == 0) stage_x(s / spc);
        stage_w(s);
    };

    // ---- per-lane gather bases and tap-validity masks of the lane's column in every column block
    int abase[CB];
    unsigned amask[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col = (wave * CB + cb) * 16 + i16;
        int h, w, base;
        bool ok;
        if (FLAT) {
            const int cc = min(col, a.ni * HW - 1), img = cc / HW, p = cc - img * HW;
            h = p / W; w = p - h * W;
            base = img * CH * HW + p;
            ok = col < ncols;
        } else {
            const int cc = min(col, a.TR * W - 1);
            h = r0 + cc / W; w = cc - (cc / W) * W;
            base = r0 * W + cc - gs;
            ok = col < ncols;
        }
        unsigned m = 0;
        if (ok) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int hh = h + tp / 3 - 1, ww = w + tp % 3 - 1;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W) m |= 1u << tp;
            }
        }
        amask[cb] = m;
        abase[cb] = base + (K16 ? 8 * (g & 1) : 8 * g) * SLc;  // this lane group's 8 channels
    }
    int boff[MB];
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        const int row = mbk * 16 + i16;
        boff[mbk] = row * 32 + (g ^ ((-(row >> 2)) & 3)) * 8;
    }
    f32x4_t acc[CB][MB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) acc[cb][mbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nsteps) stage(d);
    for (int s = 0; s < nsteps; ++s) {
        // leave exactly the copies of steps s+1 .. s+D-1 in flight; when one of them opens a chunk its X copies (issued with it,
        // ahead of its W copies) stay in flight too
        const int ahead = min(D - 1, nsteps - 1 - s);
        const int nxt = (s / spc + 1) * spc;  // first step of the next chunk
        if (nxt <= s + ahead) WaitBehindX<GW, GX, D - 1>::go(ahead);
        else WaitBehind<GW, D - 1>::go(ahead);
        COT_LDS_BARRIER();
        if (s + D < nsteps) stage(s + D);
        const int cc = s / spc, tp = s - cc * spc;
        const uint16_t* xb = reinterpret_cast<const uint16_t*>(xsm + (cc & 1) * XST);
        const bf16_t* wb = wsm + (s % NSW) * WST;
        const int tapraw = K16 ? 2 * tp + (g >> 1) : tp;  // (K16: lane groups 2,3 take the step's second tap; "tap 9" = zero weights)
        const int tap = min(tapraw, 8);
        const int shift = (tap / 3 - 1) * W + (tap % 3 - 1);
        typedef __attribute__((ext_vector_type(2))) uint16_t u16x2_t;
        bf16x8_t af[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int idx = min(max(abase[cb] + shift, 0), xelems - 7 * SLc - 1);  // masked positions: any staged element
            const uint16_t* p = xb + idx;
            u16x2_t q4[4];
            if (a.ablate & 2) {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) q4[hh][0] = q4[hh][1] = (uint16_t)(0x3c00 + lane);
            } else {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    q4[hh][0] = p[(2 * hh) * SLc];
                    q4[hh][1] = p[(2 * hh + 1) * SLc];
                }
            }
            // (a padded chunk's channels past the group's end belong to the next group: cleared by selection like the padded taps --
            // their weights are zeros, but 0 * Inf / NaN must not reach the sum; KX % 8 == 0, so a lane's 8 channels go together)
            const bool valid = tapraw < 9 && ((amask[cb] >> tap) & 1) && (K16 || cc * 32 + 8 * g < KX);
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                q4[hh][0] = valid ? q4[hh][0] : (uint16_t)0;
                q4[hh][1] = valid ? q4[hh][1] : (uint16_t)0;
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
    EpiArgs e;
    e.y1 = a.y; e.y2 = nullptr; e.bias = nullptr; e.m1 = G * MM; e.M = G * MM; e.HW = HW; e.N = a.N; e.ni = a.ni;
    e.ys1 = (int64_t)G * MM * HW; e.ys2 = 0; e.stats = nullptr; e.ptiles = 0;
    e.n0 = n0; e.p0 = r0 * W; e.m0 = grp * MM; e.mv = MM; e.ncols = ncols; e.accumulate = a.accumulate;
    e.ablate = 0; e.acc_src = nullptr; e.acc_mask = nullptr;
    tile_epilogue<CB, MB, FLAT, WAVES>(acc, e);
}

int g_conv3x3_ring = 3;  // (cot_set_tuning key 38: kept for A/B builds; see launch_c3)
template <int CB, int MB, int FLAT, int K16, int XP, int NSW>
static int launch_c3n(const C3LdsArgs& a, int64_t blocks, hipStream_t stream) {
    constexpr int WAVES = 8, NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB;
    constexpr int XST = XP * NT * 8, WST = ((BM * 4 + NT - 1) / NT) * NT * 8;
    const int ncc = K16 ? 1 : a.KK / 32;
    size_t lds = (size_t)(NSW * WST + (ncc > 1 ? 2 : 1) * XST) * sizeof(bf16_t);
    const size_t otile = (FLAT ? (size_t)a.ni * (((size_t)BM * a.H * a.W + 7) & ~(size_t)7) : (size_t)BM * (BPX + 8)) * sizeof(bf16_t);
    if (otile > lds) lds = otile;
    C3LdsArgs b = a;
    b.xcd_remap = (blocks % 8 == 0) ? 1 : 0;
    b.ablate = g_conv_ablate;
    static std::atomic<uint32_t> raised{0};
    if (lds > 64 * 1024 &&
        !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv3x3g_lds_fwd<CB, MB, FLAT, K16, XP, WAVES, NSW>)))
        return -1;  // (not covered: the caller takes the first-generation kernel)
    COT_LAUNCH((conv3x3g_lds_fwd<CB, MB, FLAT, K16, XP, WAVES, NSW>), dim3((unsigned)blocks), dim3(NT), lds, stream, b);
    return check_launch("conv3x3g_lds_fwd");
}
template <int CB, int MB, int FLAT, int K16, int XP>
static int launch_c3(const C3LdsArgs& a, int64_t blocks, hipStream_t stream) {
    // (a ring of 5 -- four steps' copies in flight -- measured no faster anywhere and slower at 56 x 56: gpurun_out/r4x_ring.log;
    // only the ring of 3 is built)
    return launch_c3n<CB, MB, FLAT, K16, XP, 3>(a, blocks, stream);
}

// ====================================================================================================================
// Third form of the 3x3 kernel ("chunk-resident", round 4): ALL taps of a 32-channel chunk's weights sit in LDS beside the
// chunk's input rows -- one wait + one barrier per CHUNK instead of per (chunk, tap) step, the nine taps of a chunk run without
// any synchronisation and the next chunk's copies (input rows + weights, into the other pair of buffers) overlap them.
// Why: the per-step ring is bound by the serial chain inside a step (barrier -> address arithmetic -> 2-byte gathers -> LDS
// latency -> pack / select -> MFMA) at two waves per SIMD, not by memory or by the copies' round trip -- CoTNet-50's 14 x 14 /
// 7 x 7 key embeddings took 27 / 42 us for 18 / 36 steps (0.9 us per step + 12 us), a deeper weight ring changed nothing
// (gpurun_out/r4x_ring.log), and the loop issued ~250 instructions per step and wave for 8 MFMAs.  Here, per (column block, tap):
// eight 2-byte reads whose addresses are (a per-lane offset, fixed for the kernel) + (a per-row scalar) + (the tap's column as an
// immediate), four v_perm, one bit-field extract of the tap's validity bit as a mask, four v_and -- and no barrier, so the
// scheduler overlaps one tap's reads with the previous tap's MFMAs.
// Masked positions are not clamped: their addresses stay inside the workgroup's LDS (the input buffers lie behind the weight
// buffers, 1 KB of slack follows them) and whatever they read is cleared by the AND.
// Output rows come in blocks of at most 64 per workgroup ("virtual groups" sharing the real group's input, as MBLK above).
template <int CB, int MB, int FLAT, int K16, int XP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void conv3x3g_lds_res(const C3LdsArgs a) {
    constexpr int NT = 64 * WAVES, BM = 16 * MB;
    constexpr int CH = K16 ? 16 : 32, SPC = K16 ? 5 : 9;
    constexpr int WPIECES = SPC * BM * 4, WP = (WPIECES + NT - 1) / NT;
    constexpr int WBUF = WP * NT * 8, XST = XP * NT * 8;  // elements per weight / input buffer
    static_assert(WP + XP <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, HW = H * W, KK = a.KK, MM = a.MM, G = a.G;
    const int ncc = K16 ? 1 : KK / 32, nb = ncc > 1 ? 2 : 1;
    const int nbw = a.wsingle ? 1 : nb;                       // weight buffers
    bf16_t* const wsm = reinterpret_cast<bf16_t*>(cot_smem);  // [nbw][WBUF] then [nb][XST] then slack
    bf16_t* const xsm = wsm + nbw * WBUF;

    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const int grp = b % G;
    const int t = b / G;
    int n0, r0 = 0, ncols, gs = 0;
    if (FLAT) {
        n0 = t * a.ni;
        ncols = min(a.ni, a.N - n0) * HW;
    } else {
        n0 = t / a.tiles;
        r0 = (t - n0 * a.tiles) * a.TR;
        ncols = min(a.TR, H - r0) * W;
        gs = max(0, (r0 - 1) * W) & ~7;
    }
    const int SLc = FLAT ? HW : a.SL;
    const int xelems = FLAT ? a.ni * CH * HW : CH * a.SL;
    const int KX = a.KX;
    const int64_t x_total = (int64_t)a.N * a.CX * HW;
    const int xg = grp / a.MBLK;

    // ---- copies of this thread: input rows (as in conv3x3g_lds_fwd), and the chunk's weights: piece q = (tap step, row, swizzled
    // 8-channel position) lands at element 8 q of the buffer
    int64_t xoff[XP];
#pragma unroll
    for (int ps = 0; ps < XP; ++ps) {
        const int q = min(ps * NT + tid, xelems / 8 - 1);
        int64_t e;
        if (FLAT) {
            const int cpi = CH * HW / 8, img = q / cpi, c = q - img * cpi;
            e = ((int64_t)min(n0 + img, a.N - 1) * a.CX + (int64_t)xg * KX) * HW + (int64_t)c * 8;
        } else {
            const int cpc = a.SL / 8, ch = q / cpc, c = q - ch * cpc;
            e = ((int64_t)n0 * a.CX + (int64_t)xg * KX + ch) * HW + gs + c * 8;
        }
        xoff[ps] = e;
    }
    int woff[WP];
#pragma unroll
    for (int ps = 0; ps < WP; ++ps) {
        const int q = min(ps * NT + tid, WPIECES - 1);
        const int tp = q / (BM * 4), within = q - tp * (BM * 4);
        const int row = within >> 2, pos = within & 3, c = pos ^ ((-(row >> 2)) & 3);
        const int rr = min(row, MM - 1);
        if (K16) woff[ps] = ((2 * tp + (c >> 1)) * MM + rr) * 16 + (c & 1) * 8;  // tap 2 tp + (c >> 1) ("tap 9": zeros), channels 8 (c & 1)..
        else woff[ps] = (tp * MM + rr) * KK + c * 8;
    }
    const bf16_t* wgrp = a.wr + (int64_t)grp * (K16 ? 10 : 9) * MM * KK;
    auto stage_x = [&](int cc) __attribute__((always_inline)) {
        bf16_t* xd = xsm + (cc & 1) * XST;
#pragma unroll
        for (int ps = 0; ps < XP; ++ps) {
            int64_t e = xoff[ps] + (int64_t)cc * CH * HW;
            if (e + 8 > x_total) e = x_total - 8;  // (the last tile's halo row past the tensor: in-bounds bytes, masked)
            COT_GLDS16(a.x + e, xd + (ps * NT + wave * 64) * 8);
        }
    };
    auto stage_w = [&](int cc) __attribute__((always_inline)) {
        bf16_t* wd = wsm + (nbw == 2 ? (cc & 1) * WBUF : 0);
#pragma unroll
        for (int ps = 0; ps < WP; ++ps) COT_GLDS16(wgrp + woff[ps] + cc * 32, wd + (ps * NT + wave * 64) * 8);
    };
    stage_x(0);
    stage_w(0);

    // ---- per-lane byte offsets (inside an input buffer) of the lane's column in every column block, per channel of its 8;
    // validity bits of the 9 taps
    int aoff[CB][8];
    unsigned amask[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col = (wave * CB + cb) * 16 + i16;
        int h, w, base;
        bool ok;
        if (FLAT) {
            const int cc = min(col, a.ni * HW - 1), img = cc / HW, p = cc - img * HW;
            h = p / W; w = p - h * W;
            base = img * CH * HW + p;
            ok = col < ncols;
        } else {
            const int cc = min(col, a.TR * W - 1);
            h = r0 + cc / W; w = cc - (cc / W) * W;
            base = r0 * W + cc - gs;
            ok = col < ncols;
        }
        unsigned m = 0;
        if (ok) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int hh = h + tp / 3 - 1, ww = w + tp % 3 - 1;
                if (hh >= 0 && hh < H && ww >= 0 && ww < W) m |= 1u << tp;
            }
        }
        if (K16 && (g >> 1)) m >>= 1;  // (lane groups 2, 3 take the step's second tap: bit 2t of m is tap 2t + 1; "tap 9" = bit 8 = 0)
        amask[cb] = m;
        const int b0 = base + (K16 ? 8 * (g & 1) : (a.perm ? g : 8 * g)) * SLc;
        const int kst = (!K16 && a.perm) ? 4 * SLc : SLc;  // channel stride between the lane's 8 K positions
#pragma unroll
        for (int k = 0; k < 8; ++k) aoff[cb][k] = (b0 + k * kst) * 2;
    }
    int boff[MB];
#pragma unroll
    for (int mbk = 0; mbk < MB; ++mbk) {
        const int row = mbk * 16 + i16;
        boff[mbk] = (row * 32 + (g ^ ((-(row >> 2)) & 3)) * 8) * 2;
    }
    f32x4_t acc[CB][MB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int mbk = 0; mbk < MB; ++mbk) acc[cb][mbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const char* const xs0 = reinterpret_cast<const char*>(xsm);
    const char* const ws0 = reinterpret_cast<const char*>(wsm);
    for (int cc = 0; cc < ncc; ++cc) {
        COT_WAIT_VM(0);     // this chunk's rows and weights (this wave's pieces) have landed ...
        COT_LDS_BARRIER();  // ... everybody's have, and everybody is done with the other pair of buffers
        if (cc + 1 < ncc) {
            stage_x(cc + 1);
            if (nbw == 2) stage_w(cc + 1);
        }
        const char* xb = xs0 + (cc & 1) * (XST * 2);
        const char* wb = ws0 + (nbw == 2 ? (cc & 1) * (WBUF * 2) : 0);
        // (a padded chunk's channels past the group's end belong to the next group: cleared like the padded taps; KX % 8 == 0)
        // interleaved K order: the lane's position j is channel 4 j + g, valid while 4 j + g < KX - 32 cc, i.e. (KX % 8 == 0) j < kwv * 2
        const bool chan_ok = K16 || a.perm || cc * 32 + 8 * g < KX;
        const int kwv = (!K16 && a.perm) ? min(4, (KX - cc * 32) / 8) : 4;  // valid 2-channel words of the lane's fragment
        unsigned am[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) am[cb] = chan_ok ? amask[cb] : 0u;
        if (!K16) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const char* xr = xb + ((dy - 1) * W - 1) * 2;  // (scalar) row dy, column -1: the taps' columns are immediates 0, 2, 4
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tap = dy * 3 + dx;
                    bf16x8_t af[CB];
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        uint32_t q[4];
                        if (a.ablate & 2) {
#pragma unroll
                            for (int hh = 0; hh < 4; ++hh) q[hh] = 0x3c003c00u + lane;
                        } else {
#pragma unroll
                            for (int hh = 0; hh < 4; ++hh) {
                                const uint32_t lo = *reinterpret_cast<const uint16_t*>(xr + aoff[cb][2 * hh] + 2 * dx);
                                const uint32_t hi = *reinterpret_cast<const uint16_t*>(xr + aoff[cb][2 * hh + 1] + 2 * dx);
                                q[hh] = lo | (hi << 16);
                            }
                        }
                        const uint32_t msk = (uint32_t)(((int32_t)(am[cb] << (31 - tap))) >> 31);  // all ones / zero
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh) q[hh] = hh < kwv ? (q[hh] & msk) : 0u;
                        __builtin_memcpy(&af[cb], q, 16);
                    }
#pragma unroll
                    for (int mbk = 0; mbk < MB; ++mbk) {
                        bf16x8_t bf;
                        __builtin_memcpy(&bf, __builtin_assume_aligned(wb + tap * (BM * 64) + boff[mbk], 16), 16);
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) acc[cb][mbk] = COT_MFMA_16X16X32_BF16(af[cb], bf, acc[cb][mbk]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int tp = 0; tp < 5; ++tp) {
                // two taps per step: lane groups 0, 1 take tap 2 tp, groups 2, 3 tap 2 tp + 1 (step 4: "tap 9", masked, any address)
                const int ta = 2 * tp, tb = tp < 4 ? 2 * tp + 1 : 8;
                const int sa = ((ta / 3 - 1) * W + ta % 3 - 1) * 2, sb = ((tb / 3 - 1) * W + tb % 3 - 1) * 2;
                const int sh = (g >> 1) ? sb : sa;
                bf16x8_t af[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    uint32_t q[4];
                    if (a.ablate & 2) {
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh) q[hh] = 0x3c003c00u + lane;
                    } else {
#pragma unroll
                        for (int hh = 0; hh < 4; ++hh) {
                            const uint32_t lo = *reinterpret_cast<const uint16_t*>(xb + aoff[cb][2 * hh] + sh);
                            const uint32_t hi = *reinterpret_cast<const uint16_t*>(xb + aoff[cb][2 * hh + 1] + sh);
                            q[hh] = lo | (hi << 16);
                        }
                    }
                    const uint32_t msk = (uint32_t)(((int32_t)(am[cb] << (31 - 2 * tp))) >> 31);
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh) q[hh] &= msk;
                    __builtin_memcpy(&af[cb], q, 16);
                }
#pragma unroll
                for (int mbk = 0; mbk < MB; ++mbk) {
                    bf16x8_t bf;
                    __builtin_memcpy(&bf, __builtin_assume_aligned(wb + tp * (BM * 64) + boff[mbk], 16), 16);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) acc[cb][mbk] = COT_MFMA_16X16X32_BF16(af[cb], bf, acc[cb][mbk]);
                }
            }
        }
        if (nbw == 1 && cc + 1 < ncc) {  // one weight buffer: everybody is done with this chunk's weights, then the next chunk's go in
            COT_LDS_BARRIER();
            stage_w(cc + 1);
        }
    }
    EpiArgs e;
    e.y1 = a.y; e.y2 = nullptr; e.bias = nullptr; e.m1 = G * MM; e.M = G * MM; e.HW = HW; e.N = a.N; e.ni = a.ni;
    e.ys1 = (int64_t)G * MM * HW; e.ys2 = 0; e.stats = nullptr; e.ptiles = 0;
    e.n0 = n0; e.p0 = r0 * W; e.m0 = grp * MM; e.mv = MM; e.ncols = ncols; e.accumulate = a.accumulate;
    e.ablate = 0; e.acc_src = nullptr; e.acc_mask = nullptr;
    tile_epilogue<CB, MB, FLAT, WAVES>(acc, e);
}

int g_conv3x3_perm = 1;  // cot_set_tuning key 45: K order inside a chunk by LDS banks (0 = always blocked, 2 = always interleaved)
int g_conv3x3_cols = 1;  // cot_set_tuning key 44: one-chunk groups pick 512- or 256-column tiles by rounds of workgroups (0 = always 512 where the chip is filled)
int g_conv3x3_wsingle = 1;  // cot_set_tuning key 42: chunk-resident 3x3 with one weight buffer where it buys a second workgroup per CU
int g_conv3x3_res = 1;  // cot_set_tuning key 39: 1 (default) = the chunk-resident form for groups of >= 24 channels, 2 = also for 16-channel groups, 0 = the per-step ring
template <int CB, int MB, int FLAT, int K16, int XP>
static int launch_c3res(const C3LdsArgs& a, int64_t blocks, hipStream_t stream) {
    constexpr int WAVES = 8, NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB, SPC = K16 ? 5 : 9;
    constexpr int WBUF = ((SPC * BM * 4 + NT - 1) / NT) * NT * 8, XST = XP * NT * 8;
    const int nb = (K16 ? 1 : a.KK / 32) > 1 ? 2 : 1;
    size_t lds = (size_t)nb * (WBUF + XST) * sizeof(bf16_t) + 1024;
    const size_t otile = (FLAT ? (size_t)a.ni * (((size_t)BM * a.H * a.W + 7) & ~(size_t)7) : (size_t)BM * (BPX + 8)) * sizeof(bf16_t);
    C3LdsArgs b = a;
    b.wsingle = 0;
    // ONE weight buffer where that makes room for a second workgroup per CU (<= 80 KB) and the launch is more than one round of
    // workgroups at one per CU: the copy of the next chunk's weights is then exposed once per chunk, and hidden by the neighbour
    // (14 x 14 key embedding: 320 workgroups = two rounds at one per CU)
    const size_t lds1 = (size_t)(WBUF + nb * XST) * sizeof(bf16_t) + 1024;
    if (nb == 2 && (g_conv3x3_wsingle == 2 ||  // (2: wherever there are two chunks -- tests)
                    (g_conv3x3_wsingle && lds > 80 * 1024 && lds1 <= 80 * 1024 && otile <= 80 * 1024 && blocks > 256))) {
        b.wsingle = 1;
        lds = lds1;
    }
    if (otile > lds) lds = otile;
    if (lds > 160 * 1024) return -1;
    b.xcd_remap = (blocks % 8 == 0) ? 1 : 0;
    b.ablate = g_conv_ablate;
    static std::atomic<uint32_t> raised{0};
    if (lds > 64 * 1024 &&
        !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv3x3g_lds_res<CB, MB, FLAT, K16, XP, WAVES>)))
        return -1;
    COT_LAUNCH((conv3x3g_lds_res<CB, MB, FLAT, K16, XP, WAVES>), dim3((unsigned)blocks), dim3(NT), lds, stream, b);
    return check_launch("conv3x3g_lds_res");
}

bool conv3x3g_lds_covers(int KK, int MM, int H, int W);
// rows per output block of the chunk-resident form (0: not covered): at most 64, a multiple of 8, the group's rows in equal blocks
static int c3res_rows(int KX, int Mreal) {
    // 16-channel groups (two taps per K step, per-lane tap offsets): the per-step ring measured faster at 56 x 56 (42.8 us against
    // 46.3 us per call, B = 80: gpurun_out/r4y_res3.log) -- the chunk-resident K16 instantiations stay built for tuning key 39 = 2
    if (KX == 16 && Mreal == 16) return g_conv3x3_res == 2 ? 16 : 0;
    if (KX == 16) return 0;
    const int nblk = (Mreal + 63) / 64;
    if (Mreal % nblk != 0 || (Mreal / nblk) % 8 != 0) return 0;
    return Mreal / nblk;
}
// -> -1: not covered (the caller takes the per-step ring); `ws`: the repacked weights' workspace as for conv3x3g_lds_gemm
static int conv3x3g_res_gemm(const void* x, const void* w, void* y, void* ws, int N, int Cin, int Cout, int G, int H, int W, int mode,
                             int accumulate, hipStream_t stream) {
    const int KX = (mode == 0 ? Cin : Cout) / G, Mreal = (mode == 0 ? Cout : Cin) / G, HW = H * W;
    if (!g_conv3x3_res || !conv3x3g_lds_covers(KX, Mreal, H, W)) return -1;
    const int MM = c3res_rows(KX, Mreal);
    if (!MM) return -1;
    const int MBLK = Mreal / MM, Greal = G;
    G *= MBLK;
    const int K16 = KX == 16, NTAP = K16 ? 10 : 9;
    const int KK = K16 ? 16 : (KX + 31) / 32 * 32, ncc = K16 ? 1 : KK / 32;
    const int MB = K16 ? 1 : (MM <= 32 ? 2 : 4);
    C3LdsArgs a;
    a.x = (const bf16_t*)x; a.wr = (const bf16_t*)ws; a.y = (bf16_t*)y;
    a.N = N; a.G = G; a.KK = KK; a.KX = KX; a.MM = MM; a.H = H; a.W = W; a.accumulate = accumulate;
    a.MBLK = MBLK; a.CX = Greal * KX;
    a.ni = 1; a.TR = 0; a.SL = 0; a.tiles = 1; a.xcd_remap = 0; a.wsingle = 0; a.perm = 0;
    int64_t blocks;
    const bool flat = HW <= 256;
    int cols = 256, XPsel = 0;
    if (flat) {
        int ni = 256 / HW;
        const int want = (N * G + 127) / 128;
        if (ni > want) ni = want;
        if (ni > N) ni = N;
        if (ni < 1) ni = 1;
        // multi-chunk groups on small planes: the 256-column tile above, or a 128-column one (CB = 1: half the gathers per wave and
        // chunk, one row pass) -- by rounds of workgroups.  7 x 7, 128 -> 128 per group, B = 80: 16 tiles x 8 row blocks = 128
        // workgroups (half the chip) of 245 columns, or 40 x 8 = 320 of 98 columns, two per CU with one weight buffer
        if (!K16 && ncc > 1 && g_conv3x3_cols && HW <= 128) {
            int ni1 = 128 / HW;
            if (ni1 > N) ni1 = N;
            const int wbuf = (MB == 4 ? 40 : 24) * 1024;
            auto cost = [&](int nimg, int cb, int xp) {
                const int64_t bl = (int64_t)ceil_div(N, nimg) * G;
                const int64_t l2 = 2 * (wbuf + xp * 8192) + 1024, l1 = wbuf + 2 * xp * 8192 + 1024;
                const int64_t l = (l2 > 80 * 1024 && l1 <= 80 * 1024 && bl > 256) ? l1 : l2;   // (launch_c3res' one-weight-buffer rule)
                const int wpc = (int)std::min<int64_t>(4, (160 * 1024) / l);
                return (double)ceil_div64(bl, (int64_t)256 * wpc) * (0.3 + 0.7 * nimg * HW / 256.0) * (l == l1 && l != l2 ? 1.15 : 1.0);
            };
            const int xp2 = (int)ceil_div64((int64_t)ni * 32 * HW / 8, 512), xp1 = (int)ceil_div64((int64_t)ni1 * 32 * HW / 8, 512);
            // MEASURED NO BETTER (7 x 7, B = 80: 28.2 us with 256 columns, 29.3 with 128 -- the chain of four chunk copies per workgroup
            // is what takes the time, not the gathers; profiles/r04_conv3x3_tile_rounds_ab.log): only taken when forced (key 44 = 2,
            // which keeps the one-column-block instantiations under test)
            (void)cost;
            if (xp1 == 1 && xp2 <= 2 && g_conv3x3_cols == 2) {
                ni = ni1;
                cols = 128;
            }
        }
        a.ni = ni;
        a.tiles = ceil_div(N, ni);
        blocks = (int64_t)a.tiles * G;
    } else {
        // BIG: 512-column tiles (CB = 4) for up to 32 output rows, 256-column ones (CB = 2) for 64 rows (register budget) and
        // whenever 512 columns would leave the chip under-filled; the staged chunk within XPmax passes of 16-byte copies
        const int CH = K16 ? 16 : 32;
        const int XPmax = K16 ? 3 : ((ncc > 1 && MB == 4) ? 3 : 5);  // (two buffer pairs of 40 KB weights + XP x 8 KB rows each)
        cols = MB == 4 ? 256 : 512;
        int TR = cols / W;
        if (TR > H) TR = H;
        if (cols == 512 && (int64_t)N * G * ceil_div(H, TR > 0 ? TR : 1) < 256 && 256 / W >= 2) {
            cols = 256;
            TR = 256 / W;
            if (TR > H) TR = H;
        }
        // one chunk, up to 32 rows: rounds of workgroups decide (SQ counters).  512-column tiles take 65 KB (two per CU),
        // 256-column ones with three row passes 49 KB (three per CU) and ~0.65 of the time each: 128 -> 128 g4 @28x28, B = 80 is 640
        // workgroups = two rounds of 512 slots, or 1280 = two rounds of 768 slots of the smaller kind
        int TRpick = 0;  // rows per tile chosen by the rule below (0: the balanced rule that follows)
        if (cols == 512 && ncc == 1 && !K16 && g_conv3x3_cols && 256 / W >= 2 && H > 0) {
            // fewest tiles of at most c columns whose rows are 16-byte multiples and whose chunk fits `xp` row passes (0: none)
            auto rows_for = [&](int c, int xp) {
                const int tmax = std::min(c / W, H);
                for (int n = ceil_div(H, tmax); n <= H; ++n)
                    for (int t = ceil_div(H, n); t <= tmax && ceil_div(H, t) == n; ++t)
                        if ((t * W) % 8 == 0 && (int64_t)CH * (((t + 2) * W + 8 + 7) / 8) <= (int64_t)xp * 512) return t;
                return 0;
            };
            const int TR5 = rows_for(512, 5), TR2 = rows_for(256, 3);
            if (TR5 > 0 && TR2 > 0) {
                const int64_t b5 = (int64_t)N * G * ceil_div(H, TR5), b2 = (int64_t)N * G * ceil_div(H, TR2);
                const double c5 = (double)ceil_div64(b5, 512) * (0.3 + 0.7 * TR5 * W / 512.0);
                const double c2 = (double)ceil_div64(b2, 768) * (0.3 + 0.7 * TR2 * W / 512.0);
                if (c2 < c5) {
                    cols = 256;
                    TRpick = TR2;
                }
            }
        }
        if (TRpick) TR = TRpick;
        if (TR < 1) return -1;
        if (!TRpick) {
            const int nt = ceil_div(H, TR);
            TR = ceil_div(H, nt);
        }
        while (TR > 1 && ((TR * W) % 8 != 0 || (int64_t)CH * (((TR + 2) * W + 8 + 7) / 8) > (int64_t)XPmax * 512)) --TR;
        if ((TR * W) % 8 != 0 || (int64_t)CH * (((TR + 2) * W + 8 + 7) / 8) > (int64_t)XPmax * 512) return -1;
        a.TR = TR;
        a.SL = ((TR + 2) * W + 8 + 7) / 8 * 8;
        a.tiles = ceil_div(H, TR);
        blocks = (int64_t)N * a.tiles * G;
        XPsel = (int)ceil_div64((int64_t)CH * (a.SL / 8), 512);
    }
    // K order inside a chunk by LDS banks: a 2-byte gather reads, per lane group g, 16 consecutive pixels (32 bytes = 8 banks) of
    // one channel row; the four groups' rows are 8 SLc elements apart in the blocked order (channels 8 g + j), SLc in the interleaved
    // one (4 j + g).  Take the interleaved order when its four windows share fewer banks than the blocked ones
    // (228-dword rows of the 28 x 28 tiles: blocked = two windows on the same banks; 144-dword rows: all four)
    a.perm = 0;
    if (!K16 && g_conv3x3_perm) {
        const int slc = flat ? HW : a.SL;
        auto overlap = [&](int step_elems) {  // banks shared by the four groups' windows (9 banks each: the +-1 column taps start on an odd element)
            int lo[4], sum = 0;
            for (int q = 0; q < 4; ++q) lo[q] = (int)(((int64_t)q * step_elems * 2 / 4) % 64);
            for (int p1 = 0; p1 < 4; ++p1)
                for (int p2 = p1 + 1; p2 < 4; ++p2) {
                    int d = (lo[p1] - lo[p2] + 64) % 64;
                    if (d > 32) d = 64 - d;
                    if (d < 9) sum += 9 - d;
                }
            return sum;
        };
        if (g_conv3x3_perm == 2 || overlap(slc) < overlap(8 * slc)) a.perm = 1;
    }
    if (t_c3_pack != 2) {   // repack the weights: [G][NTAP][MM][KK]  (2 = `ws` already holds this call's packing: cot_conv3x3g_*_packed)
        const int64_t total = (int64_t)G * NTAP * MM * KK;
        COT_LAUNCH(conv3x3g_repack_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, (const bf16_t*)w,
                   (bf16_t*)ws, G, MM, KX, KK, NTAP, mode, MBLK, a.perm);
        int rc = check_launch("conv3x3g_repack_kernel");
        if (rc || t_c3_pack == 1) return rc;  // (1 = pack only: cot_conv3x3g_pack)
    }
    if (flat) {
        if (K16) return launch_c3res<2, 1, 1, 1, 1>(a, blocks, stream);
        if (cols == 128) return MB == 2 ? launch_c3res<1, 2, 1, 0, 1>(a, blocks, stream) : launch_c3res<1, 4, 1, 0, 1>(a, blocks, stream);
        if (MB == 2) return launch_c3res<2, 2, 1, 0, 2>(a, blocks, stream);
        return launch_c3res<2, 4, 1, 0, 2>(a, blocks, stream);
    }
    if (K16) return cols == 512 ? launch_c3res<4, 1, 0, 1, 3>(a, blocks, stream) : launch_c3res<2, 1, 0, 1, 3>(a, blocks, stream);
    if (MB == 2 && cols == 256 && XPsel <= 3) return launch_c3res<2, 2, 0, 0, 3>(a, blocks, stream);
    if (MB == 2) return cols == 512 ? launch_c3res<4, 2, 0, 0, 5>(a, blocks, stream) : launch_c3res<2, 2, 0, 0, 5>(a, blocks, stream);
    if (XPsel <= 3) return launch_c3res<2, 4, 0, 0, 3>(a, blocks, stream);
    return launch_c3res<2, 4, 0, 0, 5>(a, blocks, stream);
}

// KK / MM: reduction / output channels per group.  MM: any multiple of 8 up to 128 (the tile's rows past MM are clamped copies,
// never stored); KK: 16 (with MM == 16: two taps per K step) or any multiple of 8 from 24 on, rounded up to 32-channel chunks whose
// padding is zeros in the repacked weights (round 4: groups of 24 / 48 / 96 channels, CoXtLayer.key_embed)
bool conv3x3g_lds_covers(int KK, int MM, int H, int W) {
    if (!g_conv_lds_tune[0]) return false;
    if (MM > 128 && MM % 128 == 0 && MM <= 1024) MM = 128;  // (blocks of 128 rows that share the group's input: MBLK)
    if (MM % 8 != 0 || MM < 16 || MM > 128) return false;
    if (!((KK == 16 && MM == 16) || (KK % 8 == 0 && KK >= 24 && MM >= 24))) return false;
    const int HW = H * W;
    if (HW <= 256) return HW % 8 == 0 || (MM % 8 == 0 && KK % 8 == 0);
    return HW % 8 == 0 && W % 4 == 0 && W <= 256;  // (wide rows -- 160 at SE-CoTNetD's 320 x 320 -- take fewer rows per tile)
}

// mode 0: y = conv(x, w);  mode 1: data gradient (x := dY, y := dX, weights transposed and flipped).  `ws`: the call's
// workspace (>= G*10*MM*KK bf16).  Returns COT_OK, an error, or -1 when the geometry is not covered.
int conv3x3g_lds_gemm(const void* x, const void* w, void* y, void* ws, int N, int Cin, int Cout, int G, int H, int W,
                      int mode, int accumulate, hipStream_t stream) {
    const int KX = (mode == 0 ? Cin : Cout) / G, Mreal = (mode == 0 ? Cout : Cin) / G, HW = H * W;
    if (!conv3x3g_lds_covers(KX, Mreal, H, W)) return -1;
    {
        const int rc = conv3x3g_res_gemm(x, w, y, ws, N, Cin, Cout, G, H, W, mode, accumulate, stream);
        if (rc != -1) return rc;
    }
    const int MBLK = Mreal > 128 ? Mreal / 128 : 1, MM = Mreal / MBLK;
    const int Greal = G;
    G *= MBLK;  // virtual groups from here on
    const int K16 = KX == 16, NTAP = K16 ? 10 : 9;
    const int KK = K16 ? 16 : (KX + 31) / 32 * 32;  // the repacked tiles' K: whole 32-channel chunks
    C3LdsArgs a;
    a.x = (const bf16_t*)x; a.wr = (const bf16_t*)ws; a.y = (bf16_t*)y;
    a.N = N; a.G = G; a.KK = KK; a.KX = KX; a.MM = MM; a.H = H; a.W = W; a.accumulate = accumulate;
    a.MBLK = MBLK; a.CX = Greal * KX;
    a.ni = 1; a.TR = 0; a.SL = 0; a.tiles = 1; a.xcd_remap = 0; a.wsingle = 0; a.perm = 0;
    int64_t blocks;
    int big_cols = 512;
    const bool flat = HW <= 256;
    if (flat) {
        int ni = 256 / HW;                      // CB = 2: 256 columns
        const int want = (N * G + 127) / 128;   // ... but keep >= ~128 workgroups when the batch allows
        if (ni > want) ni = want;
        if (ni > N) ni = N;
        if (ni < 1) ni = 1;
        a.ni = ni;
        a.tiles = ceil_div(N, ni);
        blocks = (int64_t)a.tiles * G;
    } else {
        // BIG: TR image rows per tile (CB = 4: 512 columns; CB = 2: 256 columns when 512-column tiles would leave the chip
        // under-filled -- 256 -> 256 at 20 x 20, B = 64: 128 workgroups of 400 columns vs 256 of 200), TR*W a multiple of 8, the
        // staged chunk within XP passes
        const int CH = K16 ? 16 : 32, XPmax = K16 ? 3 : 5;
        int TR = 512 / W;
        if (TR > H) TR = H;
        if ((int64_t)N * G * ceil_div(H, TR) < 256 && 256 / W >= 2) {
            big_cols = 256;
            TR = 256 / W;
            if (TR > H) TR = H;
        }
        const int nt = ceil_div(H, TR);
        TR = ceil_div(H, nt);  // balanced tiles
        while (TR > 1 && ((TR * W) % 8 != 0 || (int64_t)CH * (((TR + 2) * W + 8 + 7) / 8) > (int64_t)XPmax * 512)) --TR;
        if ((TR * W) % 8 != 0 || (int64_t)CH * (((TR + 2) * W + 8 + 7) / 8) > (int64_t)XPmax * 512) return -1;
        a.TR = TR;
        a.SL = ((TR + 2) * W + 8 + 7) / 8 * 8;
        a.tiles = ceil_div(H, TR);
        blocks = (int64_t)N * a.tiles * G;
    }
    if (t_c3_pack != 2) {   // repack the weights: [G][NTAP][MM][KK]
        const int64_t total = (int64_t)G * NTAP * MM * KK;
        COT_LAUNCH(conv3x3g_repack_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, (const bf16_t*)w,
                   (bf16_t*)ws, G, MM, KX, KK, NTAP, mode, MBLK, 0);
        int rc = check_launch("conv3x3g_repack_kernel");
        if (rc || t_c3_pack == 1) return rc;
    }
    if (flat) {
        if (K16) return launch_c3<2, 1, 1, 1, 1>(a, blocks, stream);
        if (MM <= 32) return launch_c3<2, 2, 1, 0, 2>(a, blocks, stream);
        if (MM <= 64) return launch_c3<2, 4, 1, 0, 2>(a, blocks, stream);
        return launch_c3<2, 8, 1, 0, 2>(a, blocks, stream);
    }
    if (big_cols == 256) {
        if (K16) return launch_c3<2, 1, 0, 1, 3>(a, blocks, stream);
        if (MM <= 32) return launch_c3<2, 2, 0, 0, 5>(a, blocks, stream);
        if (MM <= 64) return launch_c3<2, 4, 0, 0, 5>(a, blocks, stream);
        return launch_c3<2, 8, 0, 0, 5>(a, blocks, stream);
    }
    if (K16) return launch_c3<4, 1, 0, 1, 3>(a, blocks, stream);
    if (MM <= 32) return launch_c3<4, 2, 0, 0, 5>(a, blocks, stream);
    if (MM <= 64) return launch_c3<4, 4, 0, 0, 5>(a, blocks, stream);
    return launch_c3<4, 8, 0, 0, 5>(a, blocks, stream);
}

// ====================================================================================================================
// 1x1 weight gradient on the same machinery:  dW[m][j] = sum over (n, p) of dY[n][m][p] * X[n][j][p]  (+ db[m] = sum dY).
// Both operands are contiguous along the reduction index (pixels), so both fragments are 16-byte LDS reads; what the
// first-generation kernel (conv1x1.hip) lacks is sharing: every wave fetched its own fragment-shaped pieces (16 rows x 64 B
// per instruction) from global memory, and the counters show it stalled on instruction issue behind the vector-memory
// pipe for half its cycles.  Here a workgroup (8 waves, 2 x 4) owns a 128 x 128 tile of dW and a slice of the reduction;
// each K step copies 128 rows x 32 pixels of dY and of X into LDS with full 16-byte pieces (4-stage pipeline), every
// piece is fetched once per workgroup.  The reduction index runs over the images as one sequence r = n*H*W + p in steps of
// 32 (H*W % 8 == 0: an 8-pixel piece never straddles two images), slices are summed by the deterministic reduce kernel.
struct WgLdsArgs {
    const bf16_t* gy;
    const bf16_t* x1;
    const bf16_t* x2;
    float* part;   // [S][M][Jp] partial sums (S > 1)
    bf16_t* gw;    // [M][J]   (S == 1: written directly)
    bf16_t* gb;    // [M] or NULL
    int k1, N, M, J, HW, has_bias, S, jtiles, T;
    int spi;       // GEN: K steps per image = ceil(H*W / 32)
    int xcd_remap;
};

// GEN = 0: H*W % 8 == 0 -- the reduction index runs over all images as one sequence r = n*H*W + p in steps of 32 (an
//          8-pixel piece never straddles two images), every piece 16-byte aligned.
// GEN = 1: any H*W >= 32 (14 x 14, 7 x 7, ..) -- ceil(H*W/32) steps per image; pieces start at 2-byte-aligned addresses (the
//          LDS-DMA takes them: scripts/ubench_misaligned.py, exact, ~80 % of the aligned rate) and the last step of an
//          image is partial: a piece past the row's end reads the following row(s) -- finite or not -- and is removed from
//          BOTH operands by selection at fragment level.  The one piece of a launch that would run past the END of a
//          tensor is fetched element by element instead.
template <int GEN>
__global__ __launch_bounds__(512, 2) void conv1x1_wgrad_lds(const WgLdsArgs a) {
    constexpr int NS = 4, STG = 128 * 32;  // stages; elements per operand stage
    constexpr int G = 2;                    // copies per thread and stage
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* const asm_ = reinterpret_cast<bf16_t*>(cot_smem);  // [NS][STG] dY tiles, then [NS][STG] X tiles
    bf16_t* const bsm = asm_ + NS * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const int wm = wave >> 2, wj = wave & 3;
    const int HW = a.HW, M = a.M, J = a.J, Jp = J + (a.has_bias ? 1 : 0);
    unsigned b = blockIdx.x;
    if (a.xcd_remap && (gridDim.x & 7u) == 0) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const int jt = b % a.jtiles;
    const int rest = b / a.jtiles;
    const int mtiles = (M + 127) / 128;
    const int mt = rest % mtiles, sl = rest / mtiles;
    const int m0 = mt * 128, j0 = jt * 128;
    const int t0 = (int)((int64_t)a.T * sl / a.S), t1 = (int)((int64_t)a.T * (sl + 1) / a.S);

    // this thread's piece of each stage: row tid/4, 8-pixel chunk tid%4 (swizzled LDS position as in the forward's W tile)
    const int row = tid >> 2, pos = tid & 3, chunk = pos ^ ((-(row >> 2)) & 3);
    const int mrow = min(m0 + row, M - 1), jrow = min(j0 + row, J - 1);  // rows past the matrix: copies, never stored
    const bool second = a.x2 && jrow >= a.k1;
    const int xch = second ? J - a.k1 : (a.x2 ? a.k1 : J);               // channels per image of this row's X slab
    const bf16_t* ybase = a.gy + (int64_t)mrow * HW;                       // + n * M * HW + p
    const bf16_t* xbase = second ? a.x2 + (int64_t)(jrow - a.k1) * HW : a.x1 + (int64_t)jrow * HW;
    const bf16_t* xten = second ? a.x2 : a.x1;
    const int64_t ystr = (int64_t)M * HW, xstr = (int64_t)xch * HW;
    const int64_t yend = (int64_t)a.N * M * HW, xend = (int64_t)a.N * xch * HW;  // elements of the two tensors
    // position of this thread's chunk at step t: GEN 0: r = 32 t + 8 chunk -> (image, pixel); GEN 1: image t / spi, pixel
    // 32 (t % spi) + 8 chunk
    int n_, p_;
    if (GEN) {
        n_ = t0 / a.spi;
        p_ = (t0 - n_ * a.spi) * 32 + chunk * 8;
    } else {
        const int64_t r = (int64_t)t0 * 32 + chunk * 8;
        n_ = (int)(r / HW);
        p_ = (int)(r - (int64_t)n_ * HW);
    }
    const int prow_end = GEN ? a.spi * 32 : HW;  // p_ wraps here
    auto stage = [&](int tl) __attribute__((always_inline)) {  // stages are issued in order: (n_, p_) walks along
        const int buf = tl % NS;
        bf16_t* ad = asm_ + buf * STG + (wave * 64) * 8;
        bf16_t* bd = bsm + buf * STG + (wave * 64) * 8;
        const int pp = GEN ? min(p_, HW - 1) : p_;  // (GEN: a piece entirely past the row's end: any in-bounds bytes, masked)
        const bf16_t* ys = ybase + n_ * ystr + pp;
        const bf16_t* xs = xbase + n_ * xstr + pp;
        if (GEN) {
            // the piece that would run past the end of its tensor (the last row of the last image only): element-wise
            const bool yover = (ys - a.gy) + 8 > yend, xover = (xs - xten) + 8 > xend;
            COT_GLDS16(yover ? a.gy : ys, ad);
            COT_GLDS16(xover ? xten : xs, bd);
            if (yover || xover) {
                COT_WAIT_VM(0);  // (rare: at most a few lanes of one workgroup per launch) the DMA above must land first
                if (yover)
                    for (int e = 0; e < 8; ++e) ad[lane * 8 + e] = (ys - a.gy) + e < yend ? ys[e] : (bf16_t)0.0f;
                if (xover)
                    for (int e = 0; e < 8; ++e) bd[lane * 8 + e] = (xs - xten) + e < xend ? xs[e] : (bf16_t)0.0f;
            }
        } else {
            COT_GLDS16(ys, ad);
            COT_GLDS16(xs, bd);
        }
        p_ += 32;
        if (p_ >= prow_end) {  // (H*W >= 32)
            p_ -= prow_end;
            ++n_;
        }
    };
    int aoff[4], boff[2];
    bool ones[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rw = wm * 64 + q * 16 + i16;
        aoff[q] = rw * 32 + (g ^ ((-(rw >> 2)) & 3)) * 8;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rw = wj * 32 + q * 16 + i16;
        boff[q] = rw * 32 + (g ^ ((-(rw >> 2)) & 3)) * 8;
        ones[q] = a.has_bias && j0 + rw == J;  // the bias gradient rides along as a column of ones
    }
    f32x4_t acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[q][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nst = t1 - t0;
    int st_ = GEN ? t0 % a.spi : 0;  // GEN: step inside the image of the stage being multiplied
    const int tailv = HW - (a.spi - 1) * 32 - 8 * g;  // GEN: valid pixels of this lane group's 8 in an image's last step
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nst) stage(s0);
    for (int tl = 0; tl < nst; ++tl) {
        WaitBehind<G, NS - 2>::go(min(NS - 2, nst - 1 - tl));
        COT_LDS_BARRIER();
        if (tl + NS - 1 < nst) stage(tl + NS - 1);
        const bf16_t* ab = asm_ + (tl % NS) * STG;
        const bf16_t* bb = bsm + (tl % NS) * STG;
        const bool tail = GEN && st_ == a.spi - 1 && (HW & 31) != 0;  // wave-uniform
        uint32_t bq[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            __builtin_memcpy(bq[u], __builtin_assume_aligned(bb + boff[u], 16), 16);
            if (tail) mask_packed<8>(bq[u], tailv);
            if (ones[u]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[u][e] = 0x3f803f80u;  // bf16 1.0 twice (dY's own tail mask keeps the sum right)
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t aq[4];
            __builtin_memcpy(aq, __builtin_assume_aligned(ab + aoff[q], 16), 16);
            if (tail) mask_packed<8>(aq, tailv);
            const bf16x8_t af = packed_as_frag(aq);
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[q][u] = COT_MFMA_16X16X32_BF16(af, packed_as_frag(bq[u]), acc[q][u]);
        }
        if (GEN && ++st_ == a.spi) st_ = 0;
    }
    // D[i = dY row][j = X row]: lane holds rows 4g .. 4g+3 of block q, column i16 of block u
    float* ps = a.part + (int64_t)sl * M * Jp;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + q * 16 + g * 4 + i;
            if (m >= M) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int jj = j0 + wj * 32 + u * 16 + i16;
                if (a.S > 1) {
                    if (jj < Jp) ps[(int64_t)m * Jp + jj] = acc[q][u][i];
                } else if (jj < J) {
                    a.gw[(int64_t)m * J + jj] = (bf16_t)acc[q][u][i];
                } else if (jj == J && a.has_bias) {
                    a.gb[m] = (bf16_t)acc[q][u][i];
                }
            }
        }
}

int g_wgrad_lds_cap_pct = 0;  // cot_set_tuning key 20 (0 = default 25 %)
static inline bool wgrad_lds_aligned(int N, int HW) { return HW % 8 == 0 && HW >= 64 && ((int64_t)N * HW) % 32 == 0; }

// The aligned form wherever it applies; the general form (GEN = 1, any plane) for the big weight matrices of the deep layers
// (14 x 14 / 7 x 7, Co*Ci >= 256 K: Bottleneck.conv1 / conv3 of stages 3-4, embed[0] / conv1x1 of stage 4), where its deeper
// pipeline beats the register kernel once it is given enough slices -- measured cold on the MI355X (scripts/
// ubench_wgrad_deep.py, profiles/r02_wgrad_deep.log): 2048->512 @7x7 50 us vs 71, 1024->256 @14x14 50 vs 59, the 512->512 /
// 1024->256 @7x7 pair 28 vs 30.  Smaller matrices stay on the register kernel (22 vs 25 us).  cot_set_tuning key 17 bit 3
// forces the general form everywhere (tests), bit 4 switches this rule off (A/B).
static inline bool wgrad_lds_general(int HW, int M, int J) {
    return HW >= 32 && (int64_t)M * J >= 262144 && !((g_conv_lds_tune[2] >> 4) & 1);
}
bool conv1x1_wgrad_lds_covers(int N, int HW, int M, int J) {
    if (!g_conv_lds_tune[0]) return false;
    return wgrad_lds_aligned(N, HW) || (((g_conv_lds_tune[2] >> 3) & 1) && HW >= 32) || wgrad_lds_general(HW, M, J);
}

// number of slices of the LDS weight-gradient kernel (also sizes the workspace)
int conv1x1_wgrad_lds_splits(int N, int M, int J, int HW, int has_bias) {
    const int Jp = J + (has_bias ? 1 : 0);
    const int64_t tiles = (int64_t)ceil_div(M, 128) * ceil_div(Jp, 128);
    const int64_t T = wgrad_lds_aligned(N, HW) ? (int64_t)N * HW / 32 : (int64_t)N * ceil_div(HW, 32);
    int64_t S = ceil_div64(1024, tiles);  // ~4 workgroups per CU
    const int64_t in_bytes = (int64_t)N * HW * (M + J) * 2, out_bytes = (int64_t)M * Jp * 4;
    // partial sums (written + read once) below a quarter of the inputs -- the same amount as the inputs for the general form
    // (few, large tiles: 25 % would leave most CUs idle)
    const int dflt = wgrad_lds_aligned(N, HW) ? 25 : 100;
    const int64_t cap = in_bytes * (g_wgrad_lds_cap_pct > 0 ? g_wgrad_lds_cap_pct : dflt) / 100 / out_bytes;
    if (S > cap) S = cap;
    if (S > T / 8) S = T / 8;
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    return (int)S;
}

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb, hipStream_t stream);

int conv1x1_wgrad_lds_run(const void* gy, const void* x1, const void* x2, int k1, void* gw, void* gb, float* workspace, int N,
                          int J, int M, int HW, hipStream_t stream) {
    WgLdsArgs a;
    a.gy = (const bf16_t*)gy; a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.part = workspace;
    a.gw = (bf16_t*)gw; a.gb = (bf16_t*)gb; a.k1 = x2 ? k1 : J; a.N = N; a.M = M; a.J = J; a.HW = HW;
    a.has_bias = gb ? 1 : 0;
    a.S = conv1x1_wgrad_lds_splits(N, M, J, HW, a.has_bias);
    a.jtiles = ceil_div(J + a.has_bias, 128);
    const bool aligned = wgrad_lds_aligned(N, HW);
    a.spi = ceil_div(HW, 32);
    a.T = aligned ? (int)((int64_t)N * HW / 32) : N * a.spi;
    const int64_t blocks = (int64_t)a.jtiles * ceil_div(M, 128) * a.S;
    a.xcd_remap = blocks % 8 == 0;
    const size_t lds = (size_t)2 * 4 * 128 * 32 * sizeof(bf16_t);  // 64 KB
    if (aligned) COT_LAUNCH((conv1x1_wgrad_lds<0>), dim3((unsigned)blocks), dim3(512), lds, stream, a);
    else COT_LAUNCH((conv1x1_wgrad_lds<1>), dim3((unsigned)blocks), dim3(512), lds, stream, a);
    int rc = check_launch("conv1x1_wgrad_lds");
    if (rc || a.S == 1) return rc;
    return conv1x1_wgrad_reduce_launch(workspace, a.S, M, J, a.has_bias, gw, gb, stream);
}

// tuning (cot_set_tuning keys 15..17): [0] 0 = first-generation kernels, 1 = LDS kernels where eligible (default),
// [1] images per workgroup in FLAT mode (0 = auto), [2] reserved
int g_conv_lds_tune[3] = {1, 0, 0};

template <int CB, int MB, int FLAT, int NS, int WAVES, int TRD, int WT>
static int launch_c1(const C1LdsArgs& a, int tiles, hipStream_t stream) {
    constexpr int NT = 64 * WAVES, BPX = 16 * WAVES * CB, BM = 16 * MB;
    constexpr int XST = ((32 * BPX / 8 + NT - 1) / NT) * NT * 8, WST = ((BM * 4 + NT - 1) / NT) * NT * 8;
    size_t lds = (size_t)NS * (XST + WST) * sizeof(bf16_t);
    const size_t otile = (FLAT ? (size_t)a.ni * (((size_t)BM * a.HW + 7) & ~(size_t)7) : (size_t)BM * (BPX + 8)) * sizeof(bf16_t);
    if (otile > lds) lds = otile;
    const int64_t blocks = (int64_t)tiles * a.mblocks;
    C1LdsArgs b = a;
    b.xcd_remap = (blocks % 8 == 0) ? 1 : 0;
    b.ablate = g_conv_ablate;
    static std::atomic<uint32_t> raised{0};  // more than the default dynamic-LDS window: opt in once per device and instantiation
    if (lds > 64 * 1024 &&
        !raise_dynamic_lds_once(raised, reinterpret_cast<const void*>(&conv1x1_lds_fwd<CB, MB, FLAT, NS, WAVES, TRD, WT>)))
        return -1;
    COT_LAUNCH((conv1x1_lds_fwd<CB, MB, FLAT, NS, WAVES, TRD, WT>), dim3((unsigned)blocks), dim3(NT), lds, stream, b);
    return check_launch("conv1x1_lds_fwd");
}

bool conv1x1_lds_covers(int K, int k1, bool two_slabs, int HW) {
    if (!g_conv_lds_tune[0]) return false;
    // (a reduction depth off the 32-row K step: the third-generation kernel's KT instantiations -- one slab, multiples of 8)
    const bool ktail = K % 32 != 0 && K % 8 == 0 && K >= 8 && !two_slabs && g_conv_k_tail && !(g_conv_lds2_tune & 1);
    if (!ktail && (K % 32 != 0 || K < 32 || (two_slabs && k1 % 32 != 0))) return false;
    return (HW % 8 == 0 && HW >= 256) || HW <= 256;
}

// returns COT_OK, an error, or -1 when the geometry is not covered (the caller then takes the first-generation kernel)
// xs / ys (elements; 0 = dense): image strides of the input / output when they are channel ranges of wider tensors -- one
// group of a grouped convolution (cot_conv1x1g_*): single-slab calls only, third-generation kernels only.
// stats: NULL or [N][ceil(HW/128)][M][2] floats -- per (image, 128-pixel tile, channel) sum and sum of squares of the stored
// outputs, written by the epilogue (planes of more than 256 pixels only)
int conv1x1_lds_gemm(const void* x1, const void* x2, int k1, const void* w, int wpacked, const void* bias, void* y1, void* y2,
                     int m1, int N, int K, int M, int HW, int accumulate, hipStream_t stream, int64_t xs, int64_t ys, float* stats,
                     const void* acc_src, const void* acc_mask) {
    if (!conv1x1_lds_covers(K, k1, x2 != nullptr, HW)) return -1;
    if (acc_src && (y2 || xs || ys || stats || (accumulate & 1) || HW % 8 != 0 || !acc_mask)) return -1;
    if ((xs || ys) && (x2 || y2)) return -1;
    if (stats && (y2 || accumulate || HW <= 256 || HW % 8 != 0)) return -1;
    if (y2 && m1 % 8 != 0) return -1;  // (16-byte pieces of the output must not straddle the two output slabs)
    if (HW % 8 != 0 && ((y2 ? m1 : M) % 8 != 0 || (y2 && (M - m1) % 8 != 0))) return -1;  // image blocks of y 16-byte aligned
    C1LdsArgs a;
    a.x1 = (const bf16_t*)x1; a.x2 = (const bf16_t*)x2; a.w = (const bf16_t*)w; a.bias = (const bf16_t*)bias;
    a.y1 = (bf16_t*)y1; a.y2 = (bf16_t*)y2;
    a.k1 = x2 ? k1 : K; a.m1 = y2 ? m1 : M; a.N = N; a.K = K; a.M = M; a.HW = HW; a.accumulate = accumulate;
    a.xs1 = xs ? xs : (int64_t)a.k1 * HW; a.xs2 = (int64_t)(K - a.k1) * HW;
    a.ys1 = ys ? ys : (int64_t)a.m1 * HW; a.ys2 = (int64_t)(M - a.m1) * HW;
    a.ni = 1; a.xcd_remap = 0; a.xswz = 0; a.wpacked = wpacked; a.ablate = 0; a.stats = stats;
    a.acc_src = (const bf16_t*)acc_src; a.acc_mask = (const uint8_t*)acc_mask;
    const int u16 = (g_conv_lds_tune[2] >> 1) & 1;  // tuning key 17 bit 1: 2-byte gathers everywhere (A/B; default: transposing reads)
    const bool wt = wpacked == 2;
    if (wt && (M % 8 != 0 || M < 8)) return -1;
    const bool ktail = K % 32 != 0;
    if (ktail && wpacked == 1) return -1;
    if (!(g_conv_lds2_tune & 1) || xs || ys || stats || acc_src || ktail) {  // third generation (conv_lds2.hip) unless tuning key 23 bit 0 asks for this one (A/B)
        const int rc2 = conv1x1_lds_gemm2(a, stream);
        if (rc2 != -1 || xs || ys || stats || acc_src || ktail) return rc2;  // (strided slabs / epilogue statistics / masked residual / K tail: third generation only)
    }
#define COT_C1W(CB_, MB_, FLAT_, NS_, WV_, TR_)                                                             \
    return wt ? launch_c1<CB_, MB_, FLAT_, NS_, WV_, TR_, 1>(a, tiles, stream)                             \
              : launch_c1<CB_, MB_, FLAT_, NS_, WV_, TR_, 0>(a, tiles, stream)
#define COT_C1(CB4, CB8, MB_, FLAT_, NS_)  /* (the 4-wave variants of round 2's A/B are gone: 8 waves won everywhere) */ \
    do {                                                                                                   \
        if (tr) COT_C1W(CB8, MB_, FLAT_, NS_, 8, 1);                                                       \
        COT_C1W(CB8, MB_, FLAT_, NS_, 8, 0);                                                               \
    } while (0)
    if (HW > 256) {  // BIG: 128-pixel tiles of one image; three stages, several workgroups per CU
        a.ptiles = ceil_div(HW, 128);
        const int tiles = N * a.ptiles;
        const bool tr = !u16;
        if (M <= 32) { a.mblocks = 1; COT_C1(2, 1, 2, 0, 3); }
        if (M <= 64) { a.mblocks = 1; COT_C1(2, 1, 4, 0, 3); }
        a.mblocks = ceil_div(M, 128);
        COT_C1(2, 1, 8, 0, 3);
    }
    // FLAT: whole images, up to 256 columns per workgroup, up to 128 channels.  These are the deep-K layers with few
    // workgroups (one per CU at best): six stages, five of them in flight per workgroup
    int ni = g_conv_lds_tune[1] > 0 ? g_conv_lds_tune[1] : 256 / HW;
    if (ni > N) ni = N;
    if (ni < 1 || ni * HW > 256) return -1;
    a.ni = ni;
    a.ptiles = ceil_div(N, ni);
    const int tiles = a.ptiles;
    const bool tr = !u16 && HW % 4 == 0;
    if (M <= 32) { a.mblocks = 1; COT_C1(4, 2, 2, 1, 6); }
    if (M <= 64) { a.mblocks = 1; COT_C1(4, 2, 4, 1, 6); }
    // few tiles (the 7 x 7 stage at B = 80: 16 image groups x M/128 channel blocks = 64 workgroups for 512 channels, each a
    // chain of K/32 = 64 steps): 64-channel blocks double the workgroups and halve a step's MFMA and W traffic.  Measured
    // cold, forward | data gradient us (profiles/r02_conv_fewtiles_ab.log): 2048->512 68 -> 54 | =, 1024->256 39 -> 32 | 21 ->
    // 15, 512->512 27 -> 20 | 27 -> 19, 512->2048 = | 67 -> 52.  Not for 14 x 14 (80 image groups: 31 -> 45 us).
    // cot_set_tuning key 17 bits 8..: 0 = this rule, 1 = off, n > 1 = threshold on the workgroup count instead of 200.
    const int fw = g_conv_lds_tune[2] >> 8;
    if (fw != 1 && tiles <= 32 && (int64_t)tiles * ceil_div(M, 128) < (fw > 1 ? fw : 200)) {
        a.mblocks = ceil_div(M, 64);
        COT_C1(4, 2, 4, 1, 6);
    }
    a.mblocks = ceil_div(M, 128);
    COT_C1(4, 2, 8, 1, 6);
#undef COT_C1
#undef COT_C1W
}

}  // namespace cot
