// conv_tiny.hip -- 1x1 convolutions on ONE image of at most 256 "pixels": the `se` branch of the CoT layer (models/cotnet.py:
// 71-77, :98-101) in the single-node layers, where the pooled descriptor is stored channel-major [C][N] and the two 1x1
// convolutions are calls on one image whose pixels are the batch.  At B = 80 these are GEMMs of 10-40 MFLOP;
// the tiled kernels (conv_lds.hip / conv1x1.hip) run them as 2-16 workgroups walking K in 16+ barrier-separated steps:
// 8-29 us per call, 1.4 ms per CoTNet-50 step for 16 layers x 6 calls.  Here every 16 x 16 output tile is one wave that
// issues ALL its loads back to back (no staging, nothing shared) and multiplies as they land: latency = one memory round trip
// plus a few MFMAs (forward / data gradient: the four waves of a workgroup split the reduction of one tile and add their
// partial tiles through 3 KB of LDS).
//     forward        y[m][n]  = sum_k w[m][k] * x[k][n] + b[m]       A = w rows (16-byte loads), B = x columns (2-byte gathers)
//     data gradient  gx[k][n] (+)= sum_m w[m][k] * gy[m][n]          both operands strided along the reduction: 2-byte gathers
//     weight grad.   gw[m][k] = sum_n gy[m][n] * x[k][n], gb[m] = sum_n gy[m][n]     both operands contiguous along n
// MFMA maps as in mfma_common.h (A: lane l holds row l&15, k = 8*(l>>4)..+7; B: column l&15; D: rows 4*(l>>4)+r, column l&15).
// The data is a few hundred KB and L2-resident; the 2-byte gathers cost issue slots, not bandwidth.
#include "mfma_common.h"

namespace cot {

int g_conv_tiny = 1;  // cot_set_tuning key 22: 1 = use these kernels where eligible (default), 0 = off (A/B)

namespace tiny {

__device__ __forceinline__ uint16_t ldu(const bf16_t* p) {
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}
// 8 elements `stride` apart starting at p (element e valid when e < cnt), packed as an MFMA operand
__device__ __forceinline__ bf16x8_t gather8(const bf16_t* p, int64_t stride, int cnt) {
    uint16_t q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = e < cnt ? ldu(p + e * stride) : (uint16_t)0;
    bf16x8_t f;
    __builtin_memcpy(&f, q, 16);
    return f;
}
__device__ __forceinline__ bf16x8_t vec8(const bf16_t* p, bool ok) {
    bf16x8_t f;
    if (ok) {
        __builtin_memcpy(&f, __builtin_assume_aligned(p, 16), 16);
    } else {
        uint32_t z[4] = {0u, 0u, 0u, 0u};
        __builtin_memcpy(&f, z, 16);
    }
    return f;
}

// the four waves of a workgroup share one 16 x 16 output tile and split the reduction (wave v takes the 32-steps v, v+4, ..):
// a 1024-long reduction is 8 dependent MFMAs deep instead of 32; the partial tiles are added through LDS in a fixed order
__device__ __forceinline__ f32x4_t sum_waves(f32x4_t acc, float (*red)[64][4]) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv - 1][lane][r] = acc[r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[v][lane][r];
    }
    return acc;
}

// one workgroup per (16 output channels) x (16 pixels) tile; K % 8 == 0
__global__ __launch_bounds__(256) void tiny_fwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                               const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int K, int M, int N,
                                               int ntiles, int total) {
    __shared__ float red[3][64][4];
    const int wid = blockIdx.x, wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int m0 = (wid / ntiles) * 16, n0 = (wid % ntiles) * 16;
    const int mr = min(m0 + i, M - 1), nc = min(n0 + i, N - 1);  // clamped rows / columns are computed and not stored
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 32 * wv; k0 < K; k0 += 128) {
        const int kb = k0 + 8 * g;
        const bf16x8_t a = vec8(w + (int64_t)mr * K + min(kb, K - 8), kb < K);
        const bf16x8_t b = gather8(x + (int64_t)min(kb, K - 8) * N + nc, N, kb < K ? 8 : 0);
        acc = COT_MFMA_16X16X32_BF16(a, b, acc);
    }
    acc = sum_waves(acc, red);
    const int n = n0 + i;
    if (wv != 0 || n >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (m < M) y[(int64_t)m * N + n] = (bf16_t)(acc[r] + (bias ? (float)bias[m] : 0.f));
    }
}

// one workgroup per (16 input channels k) x (16 pixels) tile of gx; reduction over the M output channels, split over the waves
__global__ __launch_bounds__(256) void tiny_dgrad(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ w,
                                                 bf16_t* __restrict__ gx, int K, int M, int N, int ntiles, int total,
                                                 int accumulate) {
    __shared__ float red[3][64][4];
    const int wid = blockIdx.x, wv = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int k0 = (wid / ntiles) * 16, n0 = (wid % ntiles) * 16;
    const int kr = min(k0 + i, K - 1), nc = min(n0 + i, N - 1);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int mb = 32 * wv; mb < M; mb += 128) {
        const int m = mb + 8 * g, cnt = M - m;  // (cnt <= 0: this lane group's chunk is past M)
        const int ms = min(m, M - 1);
        const bf16x8_t a = gather8(w + (int64_t)ms * K + kr, K, cnt);
        const bf16x8_t b = gather8(gy + (int64_t)ms * N + nc, N, cnt);
        acc = COT_MFMA_16X16X32_BF16(a, b, acc);
    }
    acc = sum_waves(acc, red);
    const int n = n0 + i;
    if (wv != 0 || n >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * g + r;
        if (k < K) {
            float v = acc[r];
            if (accumulate) v += (float)gx[(int64_t)k * N + n];
            gx[(int64_t)k * N + n] = (bf16_t)v;
        }
    }
}

// one wave per (16 output channels) x (16 input channels) tile of gw; reduction over the N pixels (N % 8 == 0)
__global__ __launch_bounds__(256) void tiny_wgrad(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x,
                                                 bf16_t* __restrict__ gw, bf16_t* __restrict__ gb, int K, int M, int N,
                                                 int ktiles, int total) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total) return;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int m0 = (wid / ktiles) * 16, k0 = (wid % ktiles) * 16;
    const int mr = min(m0 + i, M - 1), kr = min(k0 + i, K - 1);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int nb = 0; nb < N; nb += 32) {
        const int n = nb + 8 * g;
        const bf16x8_t a = vec8(gy + (int64_t)mr * N + min(n, N - 8), n < N);
        const bf16x8_t b = vec8(x + (int64_t)kr * N + min(n, N - 8), n < N);
        acc = COT_MFMA_16X16X32_BF16(a, b, acc);
        if (gb && k0 == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += (float)a[e];
        }
    }
    if (gb && k0 == 0) {  // the four lane groups hold the four 8-pixel chunks of a 32-pixel step
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (g == 0 && m0 + i < M) gb[m0 + i] = (bf16_t)bsum;
    }
    const int k = k0 + i;
    if (k >= K) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (m < M) gw[(int64_t)m * K + k] = (bf16_t)acc[r];
    }
}

}  // namespace tiny

// one image, few pixels, one tensor each side, bf16 (checked by the caller), reduction lengths on 8-element chunks
bool conv_tiny_covers(int N, int Ci, int Co, int HW) {
    return g_conv_tiny && N == 1 && HW <= 256 && Ci % 8 == 0 && Ci >= 8 && Co >= 1;
}

int conv_tiny_forward(const void* x, const void* w, const void* bias, void* y, int Ci, int Co, int HW, hipStream_t stream) {
    const int ntiles = ceil_div(HW, 16), total = ceil_div(Co, 16) * ntiles;
    COT_LAUNCH(tiny::tiny_fwd, dim3(total), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
               (const bf16_t*)bias, (bf16_t*)y, Ci, Co, HW, ntiles, total);
    return check_launch("tiny_fwd");
}

int conv_tiny_backward_data(const void* gy, const void* w, void* gx, int Ci, int Co, int HW, int accumulate,
                            hipStream_t stream) {
    const int ntiles = ceil_div(HW, 16), total = ceil_div(Ci, 16) * ntiles;
    COT_LAUNCH(tiny::tiny_dgrad, dim3(total), dim3(256), 0, stream, (const bf16_t*)gy, (const bf16_t*)w,
               (bf16_t*)gx, Ci, Co, HW, ntiles, total, accumulate);
    return check_launch("tiny_dgrad");
}

int conv_tiny_backward_weight(const void* gy, const void* x, void* gw, void* gb, int Ci, int Co, int HW, hipStream_t stream) {
    const int ktiles = ceil_div(Ci, 16), total = ceil_div(Co, 16) * ktiles;
    COT_LAUNCH(tiny::tiny_wgrad, dim3(ceil_div(total, 4)), dim3(256), 0, stream, (const bf16_t*)gy, (const bf16_t*)x,
               (bf16_t*)gw, (bf16_t*)gb, Ci, Co, HW, ktiles, total);
    return check_launch("tiny_wgrad");
}

}  // namespace cot
