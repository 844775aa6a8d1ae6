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

// ---- the se branch's first half as ONE launch each way (round 6): fc1 -> BatchNorm over the batch -> ReLU (models/cotnet.py:71-77: se[0],
// se[1], se[2]) and, backward, fc2's data gradient -> that BatchNorm's backward.  The five tiny launches between the pooled descriptor and
// the radix mix are each a dependent launch on the step's critical path -- 4.2-5.3 us for ~1 us of work -- and the BatchNorm (statistics
// over the N "pixels" = the batch) is the one with a cross-tile dependency: a workgroup that owns 16 channels for ALL N pixels (one wave
// per 16-pixel tile, N <= 128) has every sample of its channels and needs nobody else.  The convolution part is tiny_fwd / tiny_dgrad with
// the reduction kept inside a wave; the BatchNorm part repeats bn_small_fwd / bn_small_bwd (bn_act.hip: fp64, lane l takes samples
// l, l + 64, same butterfly) on the tile's bf16-rounded values, so both halves give the separate kernels' bits.
__device__ __forceinline__ double wave_allsum_dd(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// hpre[m][n] = bf16(sum_k w[m][k] x[k][n] + b[m]);  h = relu(bn(hpre)) over n;  one workgroup per 16 channels m, wave v = pixels 16v..
__global__ __launch_bounds__(512) void tiny_fwd_bn(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                  bf16_t* __restrict__ hpre, bf16_t* __restrict__ h, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                                                  float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt, int K,
                                                  int M, int N, float eps, float mom) {
    __shared__ float tile[16][128];
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16, n0 = wv * 16;
    const int mr = min(m0 + i, M - 1), nc = min(n0 + i, N - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int kb = k0 + 8 * g;
        const bf16x8_t a = vec8(w + (int64_t)mr * K + min(kb, K - 8), kb < K);
        const bf16x8_t b = gather8(x + (int64_t)min(kb, K - 8) * N + nc, N, kb < K ? 8 : 0);
        acc = COT_MFMA_16X16X32_BF16(a, b, acc);
    }
    const int n = n0 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ml = 4 * g + r, m = m0 + ml;
        const bf16_t v = (bf16_t)(acc[r] + ((bias && m < M) ? (float)bias[m] : 0.f));
        if (n < N) {
            tile[ml][n] = (float)v;
            if (m < M) hpre[(int64_t)m * N + n] = v;
        }
    }
    __syncthreads();
    for (int ml = wv; ml < 16; ml += nw) {  // a wave per channel, as bn_small_fwd
        const int m = m0 + ml;
        if (m >= M) break;
        double sm = 0.0;
        for (int e = lane; e < N; e += 64) sm += (double)tile[ml][e];
        const double mu = wave_allsum_dd(sm) / N;
        double m2 = 0.0;
        for (int e = lane; e < N; e += 64) {
            const double d = (double)tile[ml][e] - mu;
            m2 += d * d;
        }
        m2 = wave_allsum_dd(m2);
        const double var = m2 / N, r = 1.0 / sqrt(var + (double)eps);
        if (lane == 0) {
            mean[m] = (float)mu;
            rstd[m] = (float)r;
            if (rmean) {
                const double unbiased = N > 1 ? m2 / (N - 1) : var;
                rmean[m] = (float)((1.0 - mom) * rmean[m] + mom * mu);
                rvar[m] = (float)((1.0 - mom) * rvar[m] + mom * unbiased);
            }
        }
        const double ga = gamma[m], be = beta[m];
        for (int e = lane; e < N; e += 64) {
            double z = ((double)tile[ml][e] - mu) * r * ga + be;
            z = z > 0.0 ? z : 0.0;
            h[(int64_t)m * N + e] = (bf16_t)(float)z;
        }
    }
}

// gh[k][n] = bf16(sum_m w[m][k] gy[m][n]) (fc2's data gradient, not stored); ghpre = the BatchNorm + ReLU backward of gh over n, with the
// statistics recomputed from hpre in fp64 as bn_small_bwd does.  One workgroup per 16 channels k, wave v = pixels 16v..
__global__ __launch_bounds__(512) void tiny_dgrad_bn(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ w,
                                                    const bf16_t* __restrict__ hpre, bf16_t* __restrict__ ghpre,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    int K, int M, int N) {
    __shared__ float tile[16][128];
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 16, n0 = wv * 16;
    const int kr = min(k0 + i, K - 1), nc = min(n0 + i, N - 1);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int mb = 0; mb < M; mb += 32) {
        const int m = mb + 8 * g, cnt = M - m;
        const int ms = min(m, M - 1);
        const bf16x8_t a = gather8(w + (int64_t)ms * K + kr, K, cnt);
        const bf16x8_t b = gather8(gy + (int64_t)ms * N + nc, N, cnt);
        acc = COT_MFMA_16X16X32_BF16(a, b, acc);
    }
    const int n = n0 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (n < N) tile[4 * g + r][n] = (float)(bf16_t)acc[r];
    __syncthreads();
    for (int kl = wv; kl < 16; kl += nw) {
        const int k = k0 + kl;
        if (k >= K) break;
        const bf16_t* xp = hpre + (int64_t)k * N;
        double sm = 0.0;
        for (int e = lane; e < N; e += 64) sm += (double)(float)xp[e];
        const double mu = wave_allsum_dd(sm) / N, r = (double)rstd[k], ga = gamma[k], be = beta[k];
        double sg = 0.0, sgx = 0.0;
        for (int e = lane; e < N; e += 64) {
            const double xh = ((double)(float)xp[e] - mu) * r, z = xh * ga + be;
            const double gg = z > 0.0 ? (double)tile[kl][e] : 0.0;
            sg += gg;
            sgx += gg * xh;
        }
        sg = wave_allsum_dd(sg);
        sgx = wave_allsum_dd(sgx);
        if (lane == 0) {
            dbeta[k] = (float)sg;
            dgamma[k] = (float)sgx;
        }
        const double k1 = sg / N, k2 = sgx / N;
        for (int e = lane; e < N; e += 64) {
            const double xh = ((double)(float)xp[e] - mu) * r, z = xh * ga + be;
            const double gg = z > 0.0 ? (double)tile[kl][e] : 0.0;
            ghpre[(int64_t)k * N + e] = (bf16_t)(float)(ga * r * (gg - k1 - xh * k2));
        }
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

// the fused se halves: bf16, one image whose N <= 128 pixels are the batch, reduction lengths on 8-element chunks
bool conv_tiny_bn_covers(int Ci, int Co, int HW) { return g_conv_tiny && HW >= 1 && HW <= 128 && Ci % 8 == 0 && Ci >= 8 && Co >= 1; }

int conv_tiny_forward_bn(const void* x, const void* w, const void* bias, void* hpre, void* h, const float* gamma, const float* beta,
                         float* mean, float* rstd, float* rmean, float* rvar, long long* nbt, int Ci, int Co, int HW, float eps, float mom,
                         hipStream_t stream) {
    COT_LAUNCH(tiny::tiny_fwd_bn, dim3(ceil_div(Co, 16)), dim3(64 * ceil_div(HW, 16)), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
               (const bf16_t*)bias, (bf16_t*)hpre, (bf16_t*)h, gamma, beta, mean, rstd, rmean, rvar, nbt, Ci, Co, HW, eps, mom);
    return check_launch("tiny_fwd_bn");
}

int conv_tiny_backward_data_bn(const void* gy, const void* w, const void* hpre, void* ghpre, const float* gamma, const float* beta,
                               const float* rstd, float* dgamma, float* dbeta, int Ci, int Co, int HW, hipStream_t stream) {
    COT_LAUNCH(tiny::tiny_dgrad_bn, dim3(ceil_div(Ci, 16)), dim3(64 * ceil_div(HW, 16)), 0, stream, (const bf16_t*)gy, (const bf16_t*)w,
               (const bf16_t*)hpre, (bf16_t*)ghpre, gamma, beta, rstd, dgamma, dbeta, Ci, Co, HW);
    return check_launch("tiny_dgrad_bn");
}

}  // namespace cot
