// gemm_kc.hip -- STUDY kernel for the layout decision of the 14x14 / 7x7 stages: the 1x1
// convolution of a channels-last activation is the K-contiguous GEMM
//
//        Y[M][Nn] = X[M][K] * Wt[Nn][K]^T        M = images * H * W pixels,  K = input channels,  Nn = output channels
//
// (models/cotnet.py:51-62,:206-224 hold the layers; a data gradient is the same product with the transposed weight as Wt).  Both
// operands have K innermost, so every LDS-DMA piece is a full 16 bytes of one row and every MFMA fragment one ds_read_b128 --
// none of the short-row pieces and transposing reads the NCHW kernels (conv_lds2.hip) need on 196- / 49-pixel rows.  It is NOT on
// any model's path: `cot_study_gemm_kc` (exported, not part of include/cotnet_amd.h's contract) exists so that
// scripts/bench_layout_study.py can time "our own K-contiguous kernel" beside the NCHW kernels and the vendor GEMM.
//
// Tile: TM x 128 outputs per workgroup of four waves (2 x 2; wave tile TM/2 x 64 = MI x 4 MFMA 16x16x32 tiles), K step 32,
// four-stage LDS ring filled by global_load_lds_dwordx4 (hand-counted vmcnt, one barrier per step).  A stage holds the X tile
// and the W tile as rows of 64 bytes; one copy instruction moves 16 rows x 4 chunks and lane (row r, slot c) fetches chunk
// c ^ ((r >> 2) & 3) of its row, so that the fragment reads of 16 consecutive rows at one K chunk touch 64 different banks.
// W is the MFMA's A operand and X its B operand: a lane's four accumulator registers are then four consecutive output
// channels of one pixel -- an 8-byte store into Y's row.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "conv_lds_common.h"

namespace cot {

// X1 | X2: the K range may be split over two tensors of the same M rows ([x, k] of CotLayer.embed[0], models/cotnet.py:81-82,
// without the cat): channels [0, K1) from X1 (row length K1), [K1, K) from X2 (row length K - K1); K1 a multiple of 32.
// bias: per output channel or NULL.  accumulate: Y += product (fp32 sum, one rounding) -- a data gradient that joins another.
// Nn: any multiple of 4 (the last column tile clamps its weight rows and skips the stores past the end).
template <int TM>
__global__ __launch_bounds__(256) void gemm_kc_tn(const bf16_t* __restrict__ X, const bf16_t* __restrict__ X2, int K1,
                                                  const bf16_t* __restrict__ Wt, const bf16_t* __restrict__ bias,
                                                  bf16_t* __restrict__ Y, int accumulate, int M, int Nn, int K, int ntn) {
    constexpr int TN = 128, NS = 4, MI = TM / 32, NJ = TN / 32;
    constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, ST = A_BYTES + B_BYTES;
    constexpr int CA = TM / 64, CB = TN / 64, G = CA + CB;  // copy instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * TM, n0 = nt * TN;

    // copy sources of this lane: row r of a 16-row block, chunk c ^ swizzle(r)
    const int r = lane >> 2, kc = (lane & 3) ^ ((r >> 2) & 3);
    const bf16_t* asrc[CA];
    const bf16_t* asrc2[CA];
    const bf16_t* bsrc[CB];
    const int KT1 = K1 / 32, K2 = K - K1;
#pragma unroll
    for (int q = 0; q < CA; ++q) {
        int row = m0 + (wave * CA + q) * 16 + r;
        row = row < M ? row : M - 1;  // (rows past the end: any valid row, their outputs are not stored)
        asrc[q] = X + (int64_t)row * K1 + kc * 8;
        asrc2[q] = X2 ? X2 + (int64_t)row * K2 + kc * 8 - K1 : asrc[q];  // (indexed by the global K offset)
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
        int row = n0 + (wave * CB + q) * 16 + r;
        row = row < Nn ? row : Nn - 1;
        bsrc[q] = Wt + (int64_t)row * K + kc * 8;
    }

    auto issue = [&](int stage, int kt) {
        char* base = cot_smem + stage * ST;
        const bool second = kt >= KT1;  // (wave-uniform)
#pragma unroll
        for (int q = 0; q < CA; ++q) COT_GLDS16((second ? asrc2[q] : asrc[q]) + kt * 32, base + (wave * CA + q) * 1024);
#pragma unroll
        for (int q = 0; q < CB; ++q) COT_GLDS16(bsrc[q] + kt * 32, base + A_BYTES + (wave * CB + q) * 1024);
    };

    // fragment addresses inside a stage: row (lane & 15) of a 16-row block, K chunk (lane >> 4) at its swizzled slot
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fslot = ((lane >> 4) ^ ((fr >> 2) & 3)) * 16;
    const int xoff = (wm * (TM / 2) + fr) * 64 + fslot;
    const int woff = A_BYTES + (wn * (TN / 2) + fr) * 64 + fslot;

    f32x4_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int KT = K / 32;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) issue(s, s);
    for (int kt = 0; kt < KT; ++kt) {
        const int left = KT - 1 - kt;
        WaitBehind<G, NS - 2>::go(left < NS - 2 ? left : NS - 2);  // this step's stage has landed (this wave's copies) ...
        COT_LDS_BARRIER();                                         // ... and everybody's; the stage read last step is free
        if (kt + NS - 1 < KT) issue((kt + NS - 1) % NS, kt + NS - 1);
        const char* st = cot_smem + (kt % NS) * ST;
        bf16x8_t xf[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(st + xoff + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(st + woff + j * 16 * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = COT_MFMA_16X16X32_BF16(wf[j], xf[i], acc[j][i]);
    }

    // D[i = 4 * (lane >> 4) + e][j = lane & 15]: i = output channel inside the 16-block, j = pixel
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (TM / 2) + i * 16 + (lane & 15);
        if (m < M) {
            const int nb = n0 + wn * (TN / 2) + 4 * (lane >> 4);
            bf16_t* yp = Y + (int64_t)m * Nn + nb;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (nb + j * 16 < Nn) {  // (Nn % 4 == 0: the four channels of a lane are inside or outside together)
                    Vec<bf16_t, 4> o, bv, old;
                    if (bias) bv = ldv<bf16_t, 4>(bias + nb + j * 16);
                    if (accumulate) old = ldv<bf16_t, 4>(yp + j * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[j][i][e];
                        if (bias) v += (float)bv.v[e];
                        if (accumulate) v += (float)old.v[e];
                        o.v[e] = (bf16_t)v;
                    }
                    stv<bf16_t, 4>(yp + j * 16, o);
                }
            }
        }
    }
}

// tm: 0 = choose (128-row tiles when they fill the chip, 64-row tiles otherwise), 64 / 128 = forced
int gemm_kc_forward(const void* x, const void* x2, int k1, const void* wt, const void* bias, void* y, int accumulate, int M, int Nn,
                    int K, int tm, hipStream_t s) {
    if (!x || !wt || !y || M <= 0 || Nn <= 0 || K <= 0) return -1;
    if (!x2) k1 = K;
    if (k1 <= 0 || k1 > K || (x2 && k1 == K)) return -1;
    // rows of 16-byte chunks: every row length a multiple of 8 channels follows from the 32-channel steps; Y rows of Nn * 2 bytes
    // take 8-byte stores (Nn % 4); bias 8-byte loads
    if (K % 32 || k1 % 32 || Nn % 4 || ((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)wt | (uintptr_t)y) % 16 || (uintptr_t)bias % 8) return -2;
    const int ntn = ceil_div(Nn, 128);
    if (tm == 0) tm = (int64_t)ceil_div(M, 128) * ntn >= 256 ? 128 : 64;
    if (tm != 64 && tm != 128) return -1;
    const int64_t blocks = (int64_t)ceil_div(M, tm) * ntn;
    if (blocks >= ((int64_t)1 << 31)) return -2;
    const dim3 grid((unsigned)blocks), block(256);
    if (tm == 128)
        COT_LAUNCH((gemm_kc_tn<128>), grid, block, 4 * (128 * 64 + 128 * 64), s, (const bf16_t*)x, (const bf16_t*)x2, k1, (const bf16_t*)wt,
                   (const bf16_t*)bias, (bf16_t*)y, accumulate, M, Nn, K, ntn);
    else
        COT_LAUNCH((gemm_kc_tn<64>), grid, block, 4 * (64 * 64 + 128 * 64), s, (const bf16_t*)x, (const bf16_t*)x2, k1, (const bf16_t*)wt,
                   (const bf16_t*)bias, (bf16_t*)y, accumulate, M, Nn, K, ntn);
    return check_launch("gemm_kc_tn");
}

}  // namespace cot

extern "C" int cot_study_gemm_kc(const void* x, const void* wt, void* y, int M, int Nn, int K, int tm, void* stream) {
    if (Nn % 128) return -2;  // (the plain study form: whole column tiles)
    return cot::gemm_kc_forward(x, nullptr, K, wt, nullptr, y, 0, M, Nn, K, tm, (hipStream_t)stream);
}
// the same kernel with what a 1x1 convolution of the CoT block needs: x as one or two channel slabs ([x, k] without the cat), bias,
// accumulation into y, any output width that is a multiple of 4 (embed[3]: 9 * C / 8)
extern "C" int cot_study_conv1x1_nhwc(const void* x1, const void* x2, int k1, const void* wt, const void* bias, void* y, int accumulate,
                                      int M, int Nn, int K, int tm, void* stream) {
    return cot::gemm_kc_forward(x1, x2, k1, wt, bias, y, accumulate, M, Nn, K, tm, (hipStream_t)stream);
}

namespace cot {

// ---- weight gradient of the same convolution, channels-last:  dW[Co][Ci] = sum over m of dY[m][co] * X[m][ci] ---------------------
// Here the reduction index (the pixel m) is the OUTER index of both operands, so the fragments (8 consecutive m of one channel) are
// columns of the staged tiles: two ds_read_b64_tr_b16 each.  A stage holds 32 rows x 128 channels of X and of dY (rows of 256
// bytes = 16 chunks; a copy instruction moves 4 rows); lane (row, slot c) fetches chunk c ^ s(row), s(row) = 2 * ((row & 3) |
// ((row >> 3) & 1) << 2): the sixteen 32-byte windows a transposing read of one wave touches then cover the 64 banks exactly twice.
// X is the MFMA's A operand: a lane's four accumulators are four consecutive ci of one co -- a 16-byte store into the fp32 partial
// sums part[slice][Co][Ci]; the reduction over pixels is split into `slices` ranges of 32-row steps, one workgroup each, and
// gemm_kc_wgrad_reduce adds the slices in order and rounds once (deterministic, no atomics).  Channel tiles past Ci / Co clamp
// their chunks (those products are never stored); rows past M in the very last step are cleared in LDS by the lanes that copied them.
__device__ __forceinline__ int kc_swz(int row) { return (((row & 3) | (((row >> 3) & 1) << 2)) << 1); }

// TAPS (the grouped 3x3 convolution's weight gradient, dW[g * Mg + j][t][i] = sum_m dY[m][g * Mg + j] * X[m + shift(t)][g * Kc + i]):
// blockIdx.y = group * 9 + tap; X and dY are then [M][ldx] / [M][ldy] with the group's channel window at xc / yc (Ci / Co = the
// window's widths, at most 128: one tile), the X rows of a step come from the shifted pixels (the zero block where the tap leaves
// the image) and the partial sums go to part[slice][(g * Mg + j) * 9 + t][i].
struct KcTaps {
    const bf16_t* zeros;
    int H, W, ldx, ldy, Kc, Mg;
};
template <bool TAPS>
__global__ __launch_bounds__(256) void gemm_kc_wgrad(const bf16_t* __restrict__ X, const bf16_t* __restrict__ dY,
                                                    float* __restrict__ part, int M, int Ci, int Co, int tci, int tco, int slices,
                                                    KcTaps tp) {
    constexpr int NS = 4, TB = 32 * 256, ST = 2 * TB, G = 4;  // tile bytes, stage bytes, copies per wave and stage
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = blockIdx.x % slices, tile = blockIdx.x / slices;
    const int ti = tile % tci, to = tile / tci;
    const int ci0 = ti * 128, co0 = to * 128;
    const int grp = TAPS ? (int)blockIdx.y / 9 : 0, tap = TAPS ? (int)blockIdx.y - grp * 9 : 0;
    const int tdy = tap / 3 - 1, tdx = tap - (tap / 3) * 3 - 1;
    const int ldx = TAPS ? tp.ldx : Ci, ldy = TAPS ? tp.ldy : Co;
    if (TAPS) {
        X += grp * tp.Kc;
        dY += grp * tp.Mg;
    }
    const int KT = ceil_div(M, 32);
    const int k0 = (int)((int64_t)KT * sl / slices), k1 = (int)((int64_t)KT * (sl + 1) / slices);

    // copy sources: instruction q (0, 1) of this wave = rows 8 * wave + 4 * q .. + 3 of the stage, slot (lane & 15)
    const bf16_t* xs[2];
    const bf16_t* ys[2];
    int crow[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        crow[q] = 8 * wave + 4 * q + (lane >> 4);
        const int gc = (lane & 15) ^ kc_swz(crow[q]);
        int cx = ci0 + gc * 8, cy = co0 + gc * 8;
        cx = cx < Ci ? cx : Ci - 8;
        cy = cy < Co ? cy : Co - 8;
        xs[q] = X + cx;
        ys[q] = dY + cy;
    }
    auto issue = [&](int stage, int kt) {
        char* base = cot_smem + stage * ST;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int m = kt * 32 + crow[q];
            m = m < M ? m : M - 1;  // (cleared after it has landed: see below)
            const bf16_t* xsrc = xs[q] + (int64_t)m * ldx;
            if (TAPS) {
                const int pix = m % (tp.H * tp.W), h = pix / tp.W, w = pix - h * tp.W;
                const bool in = (unsigned)(h + tdy) < (unsigned)tp.H && (unsigned)(w + tdx) < (unsigned)tp.W;
                xsrc = in ? xsrc + (int64_t)(tdy * tp.W + tdx) * ldx : tp.zeros;
            }
            COT_GLDS16(xsrc, base + (8 * wave + 4 * q) * 256);
            COT_GLDS16(ys[q] + (int64_t)m * ldy, base + TB + (8 * wave + 4 * q) * 256);
        }
    };

    // fragment addresses: 16-lane group g reads rows 8g .. 8g+7 (two transposing reads of 4 rows) of a 16-channel block
    const int wm = wave >> 1, wn = wave & 1;  // ci half, co half of the 128 x 128 tile
    const int g = lane >> 4, L = lane & 15;
    const int frow = 8 * g + (L >> 2);
    const int fsw = kc_swz(frow);  // (the same for frow + 4)
    int xo[4], yo[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int cxk = wm * 8 + b * 2 + ((L & 3) >> 1), cyk = wn * 8 + b * 2 + ((L & 3) >> 1);
        xo[b] = frow * 256 + ((cxk ^ fsw) << 4) + (L & 1) * 8;
        yo[b] = TB + frow * 256 + ((cyk ^ fsw) << 4) + (L & 1) * 8;
    }

    f32x4_t acc[4][4];  // [co block][ci block]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int steps = k1 - k0;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < steps) issue(s, k0 + s);
    for (int t = 0; t < steps; ++t) {
        const int left = steps - 1 - t;
        WaitBehind<G, NS - 2>::go(left < NS - 2 ? left : NS - 2);
        char* st = cot_smem + (t % NS) * ST;
        if ((k0 + t) * 32 + 32 > M) {  // the last step of the tensor: rows past M hold a copy of row M - 1 -- clear this lane's pieces
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if ((k0 + t) * 32 + crow[q] >= M) {
                    *reinterpret_cast<f32x4_t*>(st + (8 * wave + 4 * q) * 256 + lane * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4_t*>(st + TB + (8 * wave + 4 * q) * 256 + lane * 16) = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
        }
        COT_LDS_BARRIER();
        if (t + NS - 1 < steps) issue((t + NS - 1) % NS, k0 + t + NS - 1);
        bf16x8_t xf[4], yf[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s16x4_t lo = COT_LDS_READ_TR16(st + xo[b]), hi = COT_LDS_READ_TR16(st + xo[b] + 4 * 256);
            __builtin_memcpy(&xf[b], &lo, 8);
            __builtin_memcpy(reinterpret_cast<char*>(&xf[b]) + 8, &hi, 8);
            lo = COT_LDS_READ_TR16(st + yo[b]);
            hi = COT_LDS_READ_TR16(st + yo[b] + 4 * 256);
            __builtin_memcpy(&yf[b], &lo, 8);
            __builtin_memcpy(reinterpret_cast<char*>(&yf[b]) + 8, &hi, 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = COT_MFMA_16X16X32_BF16(xf[i], yf[j], acc[j][i]);
    }

    // D[i = 4 * (lane >> 4) + e][j = lane & 15]: i = ci inside its 16-block, j = co
    // plain: part[slice][Co][Ci]; TAPS: part[slice][groups * Mg][9][Kc], this workgroup's rows (g * Mg + co) * 9 + tap
    const int64_t slab = TAPS ? (int64_t)(gridDim.y / 9) * tp.Mg * 9 * tp.Kc : (int64_t)Co * Ci;
    float* pp = part + sl * slab;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + wn * 64 + j * 16 + L;
        if (co < Co) {
            float* row = TAPS ? pp + ((int64_t)(grp * tp.Mg + co) * 9 + tap) * tp.Kc : pp + (int64_t)co * Ci;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ci = ci0 + wm * 64 + i * 16 + 4 * g;
                if (ci < Ci)  // (Ci % 8 == 0: four channels inside or outside together)
                    *reinterpret_cast<f32x4_t*>(row + ci) = acc[j][i];
            }
        }
    }
}

// ci4 / ldw: a row of the result has ci4 groups of four channels and goes to dW + row * ldw (ldw = 4 * ci4: dense; larger: a column
// window of a wider gradient -- one slab of embed[0]'s [x | k] weight)
__global__ __launch_bounds__(256) void gemm_kc_wgrad_reduce(const float* __restrict__ part, bf16_t* __restrict__ dW, int64_t n4, int slices,
                                                           int ci4, int ldw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4_t s = *reinterpret_cast<const f32x4_t*>(part + 4 * i);
    for (int k = 1; k < slices; ++k) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(part + (int64_t)k * 4 * n4 + 4 * i);
        s += v;
    }
    Vec<bf16_t, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)s[e];
    const int64_t row = i / ci4;
    stv<bf16_t, 4>(dW + row * ldw + 4 * (i - row * ci4), o);
}

// slices of the reduction: enough workgroups for two per CU, at least four 32-row steps each (0 = choose; > 0 = forced, tests)
static int gemm_kc_wgrad_slices(int M, int Ci, int Co, int forced) {
    const int KT = ceil_div(M, 32), tiles = ceil_div(Ci, 128) * ceil_div(Co, 128);
    int s = forced > 0 ? forced : ceil_div(512, tiles);
    if (forced <= 0 && s > KT / 4) s = KT / 4;
    if (s > KT) s = KT;
    return s < 1 ? 1 : s;
}
size_t gemm_kc_wgrad_workspace(int M, int Ci, int Co, int forced) {
    return (size_t)gemm_kc_wgrad_slices(M, Ci, Co, forced) * Co * Ci * sizeof(float);
}
int gemm_kc_wgrad_run(const void* x, const void* dy, void* dw, void* workspace, int M, int Ci, int Co, int forced, hipStream_t s, int ldw = 0) {
    if (!x || !dy || !dw || !workspace || M <= 0 || Ci <= 0 || Co <= 0) return -1;
    if (ldw == 0) ldw = Ci;
    if (ldw < Ci || ldw % 4) return -2;
    if (Ci % 8 || Co % 8 || ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) % 16) return -2;
    const int slices = gemm_kc_wgrad_slices(M, Ci, Co, forced), tci = ceil_div(Ci, 128), tco = ceil_div(Co, 128);
    const int64_t blocks = (int64_t)tci * tco * slices;
    if (blocks >= ((int64_t)1 << 31)) return -2;
    COT_LAUNCH((gemm_kc_wgrad<false>), dim3((unsigned)blocks), dim3(256), 4 * 2 * 32 * 256, s, (const bf16_t*)x, (const bf16_t*)dy,
               (float*)workspace, M, Ci, Co, tci, tco, slices, KcTaps{});
    const int64_t n4 = (int64_t)Co * Ci / 4;
    COT_LAUNCH(gemm_kc_wgrad_reduce, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, s, (const float*)workspace, (bf16_t*)dw, n4, slices, Ci / 4,
               ldw);
    return check_launch("gemm_kc_wgrad");
}

// grouped 3x3 weight gradient, channels-last: x [N*H*W][C], dy [N*H*W][Co] -> dwr [Co][9][Kc] (the forward's repacked layout)
static int conv3x3g_kc_wgrad_slices(int M, int groups, int forced) {
    const int KT = ceil_div(M, 32);
    int s = forced > 0 ? forced : ceil_div(512, groups * 9);
    if (forced <= 0 && s > KT / 4) s = KT / 4;
    if (s > KT) s = KT;
    return s < 1 ? 1 : s;
}
size_t conv3x3g_kc_wgrad_workspace(int N, int H, int W, int C, int Co, int groups, int forced) {
    return (size_t)conv3x3g_kc_wgrad_slices(N * H * W, groups, forced) * Co * 9 * (C / groups) * sizeof(float);
}
int conv3x3g_kc_wgrad_run(const void* x, const void* dy, const void* zeros, void* dwr, void* workspace, int N, int H, int W, int C, int Co,
                          int groups, int forced, hipStream_t s) {
    if (!x || !dy || !zeros || !dwr || !workspace || N <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups || Co % groups) return -1;
    const int Kc = C / groups, Mg = Co / groups;
    if (Kc % 8 || Mg % 8 || Kc > 128 || Mg > 128 || ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)zeros | (uintptr_t)dwr | (uintptr_t)workspace) % 16)
        return -2;
    const int64_t M64 = (int64_t)N * H * W;
    if (M64 * (C > Co ? C : Co) >= ((int64_t)1 << 31)) return -2;
    const int M = (int)M64, slices = conv3x3g_kc_wgrad_slices(M, groups, forced);
    const KcTaps tp{(const bf16_t*)zeros, H, W, C, Co, Kc, Mg};
    COT_LAUNCH((gemm_kc_wgrad<true>), dim3((unsigned)slices, (unsigned)(groups * 9)), dim3(256), 4 * 2 * 32 * 256, s, (const bf16_t*)x,
               (const bf16_t*)dy, (float*)workspace, M, Kc, Mg, 1, 1, slices, tp);
    const int64_t n4 = (int64_t)Co * 9 * Kc / 4;
    COT_LAUNCH(gemm_kc_wgrad_reduce, dim3((unsigned)ceil_div64(n4, 256)), dim3(256), 0, s, (const float*)workspace, (bf16_t*)dwr, n4, slices,
               9 * Kc / 4, 9 * Kc);
    return check_launch("conv3x3g_kc_wgrad");
}

}  // namespace cot

extern "C" size_t cot_study_conv3x3g_nhwc_wgrad_workspace(int N, int H, int W, int C, int Co, int groups, int slices) {
    return cot::conv3x3g_kc_wgrad_workspace(N, H, W, C, Co, groups, slices);
}
extern "C" int cot_study_conv3x3g_nhwc_wgrad(const void* x, const void* dy, const void* zeros, void* dwr, void* workspace, int N, int H, int W,
                                             int C, int Co, int groups, int slices, void* stream) {
    return cot::conv3x3g_kc_wgrad_run(x, dy, zeros, dwr, workspace, N, H, W, C, Co, groups, slices, (hipStream_t)stream);
}
extern "C" size_t cot_study_conv1x1_nhwc_wgrad_workspace(int M, int Ci, int Co, int slices) {
    return cot::gemm_kc_wgrad_workspace(M, Ci, Co, slices);
}
extern "C" int cot_study_conv1x1_nhwc_wgrad(const void* x, const void* dy, void* dw, void* workspace, int M, int Ci, int Co, int slices,
                                            void* stream) {
    return cot::gemm_kc_wgrad_run(x, dy, dw, workspace, M, Ci, Co, slices, (hipStream_t)stream);
}
// the same into a column window of a wider gradient: rows of dw are ldw elements apart (dw already points at the window's first column)
extern "C" int cot_study_conv1x1_nhwc_wgrad_window(const void* x, const void* dy, void* dw, int ldw, void* workspace, int M, int Ci, int Co,
                                                   int slices, void* stream) {
    return cot::gemm_kc_wgrad_run(x, dy, dw, workspace, M, Ci, Co, slices, (hipStream_t)stream, ldw);
}

namespace cot {

// ---- grouped 3x3 convolution (stride 1, padding 1), channels-last: key_embed[0] of the CoT layer (models/cotnet.py:43-47) ------------
//        Y[m][g * Mg + j] = sum over taps t, channels i of  X[m + shift(t)][g * Kc + i] * Wr[g * Mg + j][t][i]
// With channels innermost this IS the kernel above: nine times the K steps, the weight rows repacked [Co][9][Kc] (so that the weight
// side is a plain K-contiguous row of 9 * Kc), and on the activation side a lane's copy source moves to the shifted pixel's row for
// each tap -- or, where the tap leaves the image, to a 16-byte block of zeros (`zeros`: any device buffer of >= 16 zero bytes), with
// no stride along K.  A data gradient is the same call with the taps reversed and the per-group weights transposed by the caller's
// repack.  TN = 64 when a group has 64 output channels (a column tile must not straddle groups).
template <int TM, int TN>
__global__ __launch_bounds__(256) void conv3x3g_kc(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wr, const bf16_t* __restrict__ zeros,
                                                  bf16_t* __restrict__ Y, int M, int H, int W, int C, int Kc, int Mg, int tiles_g,
                                                  int accumulate) {
    constexpr int NS = 4, MI = TM / 32, NJ = TN / 32;
    constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, ST = A_BYTES + B_BYTES;
    constexpr int CA = TM / 64, CB = TN / 64, G = CA + CB;  // copy instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = blockIdx.x, nt = blockIdx.y;
    const int grp = nt / tiles_g, n0 = grp * Mg + (nt - grp * tiles_g) * TN;
    const int m0 = mt * TM, Co = (int)gridDim.y / tiles_g * Mg, KS = Kc / 32, K9 = 9 * Kc, KT = 9 * KS;

    const int r = lane >> 2, kc = (lane & 3) ^ ((r >> 2) & 3);
    int am[CA], ah[CA], aw[CA];
    const bf16_t* bsrc[CB];
#pragma unroll
    for (int q = 0; q < CA; ++q) {
        int row = m0 + (wave * CA + q) * 16 + r;
        row = row < M ? row : M - 1;
        am[q] = row;
        const int pix = row % (H * W);
        ah[q] = pix / W;
        aw[q] = pix - ah[q] * W;
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {
        // TN = 64: the four waves copy 16 rows each (one instruction per wave); TN = 128: two instructions per wave
        const int row = n0 + (wave * CB + q) * 16 + r;
        bsrc[q] = Wr + (int64_t)row * K9 + kc * 8;
    }
    auto issue = [&](int stage, int kt) {
        char* base = cot_smem + stage * ST;
        const int tap = kt / KS, ks = kt - tap * KS;  // (wave-uniform)
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int q = 0; q < CA; ++q) {
            const bool in = (unsigned)(ah[q] + dy) < (unsigned)H && (unsigned)(aw[q] + dx) < (unsigned)W;
            const bf16_t* src = in ? X + (int64_t)(am[q] + dy * W + dx) * C + grp * Kc + ks * 32 + kc * 8 : zeros;
            COT_GLDS16(src, base + (wave * CA + q) * 1024);
        }
#pragma unroll
        for (int q = 0; q < CB; ++q) COT_GLDS16(bsrc[q] + kt * 32, base + A_BYTES + (wave * CB + q) * 1024);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fslot = ((lane >> 4) ^ ((fr >> 2) & 3)) * 16;
    const int xoff = (wm * (TM / 2) + fr) * 64 + fslot;
    const int woff = A_BYTES + (wn * (TN / 2) + fr) * 64 + fslot;
    f32x4_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) issue(s, s);
    for (int kt = 0; kt < KT; ++kt) {
        const int left = KT - 1 - kt;
        WaitBehind<G, NS - 2>::go(left < NS - 2 ? left : NS - 2);
        COT_LDS_BARRIER();
        if (kt + NS - 1 < KT) issue((kt + NS - 1) % NS, kt + NS - 1);
        const char* st = cot_smem + (kt % NS) * ST;
        bf16x8_t xf[MI], wf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(st + xoff + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(st + woff + j * 16 * 64);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = COT_MFMA_16X16X32_BF16(wf[j], xf[i], acc[j][i]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (TM / 2) + i * 16 + (lane & 15);
        if (m < M) {
            bf16_t* yp = Y + (int64_t)m * Co + n0 + wn * (TN / 2) + 4 * (lane >> 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                Vec<bf16_t, 4> o, old;
                if (accumulate) old = ldv<bf16_t, 4>(yp + j * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(acc[j][i][e] + (accumulate ? (float)old.v[e] : 0.f));
                stv<bf16_t, 4>(yp + j * 16, o);
            }
        }
    }
}

// x [N*H*W][C], wr [C_out][9][Kc] (C_out = groups * Mg; Kc = C / groups), y [N*H*W][C_out]
int conv3x3g_kc_forward(const void* x, const void* wr, const void* zeros, void* y, int accumulate, int N, int H, int W, int C, int Co,
                        int groups, hipStream_t s) {
    if (!x || !wr || !zeros || !y || N <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups || Co % groups) return -1;
    const int Kc = C / groups, Mg = Co / groups;
    if (Kc % 32 || (Mg != 64 && Mg % 128) || ((uintptr_t)x | (uintptr_t)wr | (uintptr_t)zeros | (uintptr_t)y) % 16) return -2;
    const int64_t M64 = (int64_t)N * H * W;
    if (M64 * (C > Co ? C : Co) >= ((int64_t)1 << 31)) return -2;  // (32-bit row indices in the copy sources)
    const int M = (int)M64, TN = Mg == 64 ? 64 : 128, tiles_g = Mg / TN;
    const int TM = (int64_t)ceil_div(M, 128) * groups * tiles_g >= 256 ? 128 : 64;
    const dim3 grid(ceil_div(M, TM), groups * tiles_g), block(256);
#define C3KC(TM_, TN_)                                                                                                              \
    COT_LAUNCH((conv3x3g_kc<TM_, TN_>), grid, block, 4 * (TM_ * 64 + TN_ * 64), s, (const bf16_t*)x, (const bf16_t*)wr, (const bf16_t*)zeros, \
               (bf16_t*)y, M, H, W, C, Kc, Mg, tiles_g, accumulate)
    if (TM == 128 && TN == 128) C3KC(128, 128);
    else if (TM == 128) C3KC(128, 64);
    else if (TN == 128) C3KC(64, 128);
    else C3KC(64, 64);
#undef C3KC
    return check_launch("conv3x3g_kc");
}

}  // namespace cot

extern "C" int cot_study_conv3x3g_nhwc(const void* x, const void* wr, const void* zeros, void* y, int accumulate, int N, int H, int W, int C,
                                       int Co, int groups, void* stream) {
    return cot::conv3x3g_kc_forward(x, wr, zeros, y, accumulate, N, H, W, C, Co, groups, (hipStream_t)stream);
}

namespace cot {

// ---- data gradient without a transposed weight copy:  Y[M][N] (+)= X[M][K] * B[K][N]   (B = the convolution's own weight [Co][Ci], K = Co,
// N = Ci; row stride ldb so that a column range of a wider weight -- one slab of embed[0]'s [x | k] -- can be addressed) ----------------------
// X side as in gemm_kc_tn (rows of 64 bytes, ds_read_b128 fragments).  B side as in gemm_kc_wgrad: a stage holds 32 k-rows x 128 columns
// (rows of 256 bytes, four rows per copy instruction, chunk slots swizzled by kc_swz), the fragment of 16 columns x 8 k is two
// transposing reads.  B is the MFMA's A operand again: four consecutive output columns per lane, 8-byte stores.
template <int TM>
__global__ __launch_bounds__(256) void gemm_kc_nn(const bf16_t* __restrict__ X, const bf16_t* __restrict__ B, bf16_t* __restrict__ Y, int accumulate,
                                                  int M, int N, int K, int ldb, int ldy, int ntn) {
    constexpr int NS = 4, MI = TM / 32, A_BYTES = TM * 64, B_BYTES = 32 * 256, ST = A_BYTES + B_BYTES;
    constexpr int CA = TM / 64, CB = 2, G = CA + CB;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
    const int m0 = mt * TM, n0 = nt * 128;

    const int r = lane >> 2, kc = (lane & 3) ^ ((r >> 2) & 3);
    const bf16_t* asrc[CA];
    const bf16_t* bsrc[CB];
#pragma unroll
    for (int q = 0; q < CA; ++q) {
        int row = m0 + (wave * CA + q) * 16 + r;
        row = row < M ? row : M - 1;
        asrc[q] = X + (int64_t)row * K + kc * 8;
    }
#pragma unroll
    for (int q = 0; q < CB; ++q) {  // rows 8 * wave + 4 * q + (lane >> 4) of the k-step, slot (lane & 15)
        const int brow = 8 * wave + 4 * q + (lane >> 4);
        int col = n0 + (((lane & 15) ^ kc_swz(brow)) << 3);
        col = col < N ? col : N - 8;
        bsrc[q] = B + (int64_t)brow * ldb + col;
    }
    auto issue = [&](int stage, int kt) {
        char* base = cot_smem + stage * ST;
#pragma unroll
        for (int q = 0; q < CA; ++q) COT_GLDS16(asrc[q] + kt * 32, base + (wave * CA + q) * 1024);
#pragma unroll
        for (int q = 0; q < CB; ++q) COT_GLDS16(bsrc[q] + (int64_t)kt * 32 * ldb, base + A_BYTES + (8 * wave + 4 * q) * 256);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fslot = ((lane >> 4) ^ ((fr >> 2) & 3)) * 16;
    const int xoff = (wm * (TM / 2) + fr) * 64 + fslot;
    const int g = lane >> 4, L = lane & 15, frow = 8 * g + (L >> 2), fsw = kc_swz(frow);
    int bo[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bo[b] = A_BYTES + frow * 256 + (((wn * 8 + b * 2 + ((L & 3) >> 1)) ^ fsw) << 4) + (L & 1) * 8;

    f32x4_t acc[4][MI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int KT = K / 32;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) issue(s, s);
    for (int kt = 0; kt < KT; ++kt) {
        const int left = KT - 1 - kt;
        WaitBehind<G, NS - 2>::go(left < NS - 2 ? left : NS - 2);
        COT_LDS_BARRIER();
        if (kt + NS - 1 < KT) issue((kt + NS - 1) % NS, kt + NS - 1);
        const char* st = cot_smem + (kt % NS) * ST;
        bf16x8_t xf[MI], wf[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(st + xoff + i * 16 * 64);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s16x4_t lo = COT_LDS_READ_TR16(st + bo[b]), hi = COT_LDS_READ_TR16(st + bo[b] + 4 * 256);
            __builtin_memcpy(&wf[b], &lo, 8);
            __builtin_memcpy(reinterpret_cast<char*>(&wf[b]) + 8, &hi, 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = COT_MFMA_16X16X32_BF16(wf[j], xf[i], acc[j][i]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (TM / 2) + i * 16 + (lane & 15);
        if (m < M) {
            const int nb = n0 + wn * 64 + 4 * (lane >> 4);
            bf16_t* yp = Y + (int64_t)m * ldy + nb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (nb + j * 16 < N) {
                    Vec<bf16_t, 4> o, old;
                    if (accumulate) old = ldv<bf16_t, 4>(yp + j * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(acc[j][i][e] + (accumulate ? (float)old.v[e] : 0.f));
                    stv<bf16_t, 4>(yp + j * 16, o);
                }
            }
        }
    }
}

int gemm_kc_nn_run(const void* x, const void* b, void* y, int accumulate, int M, int N, int K, int ldb, int ldy, int tm, hipStream_t s) {
    if (!x || !b || !y || M <= 0 || N <= 0 || K <= 0 || ldb < N || ldy < N) return -1;
    if (K % 32 || N % 8 || ldb % 8 || ldy % 4 || ((uintptr_t)x | (uintptr_t)b) % 16 || (uintptr_t)y % 8) return -2;
    const int ntn = ceil_div(N, 128);
    if (tm == 0) tm = (int64_t)ceil_div(M, 128) * ntn >= 256 ? 128 : 64;
    if (tm != 64 && tm != 128) return -1;
    const int64_t blocks = (int64_t)ceil_div(M, tm) * ntn;
    if (blocks >= ((int64_t)1 << 31)) return -2;
    const dim3 grid((unsigned)blocks), block(256);
    if (tm == 128)
        COT_LAUNCH((gemm_kc_nn<128>), grid, block, 4 * (128 * 64 + 32 * 256), s, (const bf16_t*)x, (const bf16_t*)b, (bf16_t*)y, accumulate, M, N, K,
                   ldb, ldy, ntn);
    else
        COT_LAUNCH((gemm_kc_nn<64>), grid, block, 4 * (64 * 64 + 32 * 256), s, (const bf16_t*)x, (const bf16_t*)b, (bf16_t*)y, accumulate, M, N, K,
                   ldb, ldy, ntn);
    return check_launch("gemm_kc_nn");
}

}  // namespace cot

// dx [M][ldy] (+)= dy [M][K] * w [K][ldb] on the columns [0, N) of w and dx: the data gradient of a channels-last 1x1 convolution straight
// from the convolution's weight (K = its output channels, N = its input channels or one slab of them)
extern "C" int cot_study_conv1x1_nhwc_dgrad(const void* dy, const void* w, void* dx, int accumulate, int M, int N, int K, int ldb, int ldy, int tm,
                                            void* stream) {
    return cot::gemm_kc_nn_run(dy, w, dx, accumulate, M, N, K, ldb, ldy, tm, (hipStream_t)stream);
}
