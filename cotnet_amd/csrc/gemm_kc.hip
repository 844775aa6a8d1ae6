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
