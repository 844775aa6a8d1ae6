// stem3x3.hip -- a deep stem's first convolution: 3x3, stride 2, padding 1, 3 -> Co channels (Co = 32 or 64), NCHW bf16
// (reference: models/cotnet_hybrid.py:359 / models/resnet.py deep stems, `nn.Conv2d(in_chans, stem_chs_1, 3, stride=2, padding=1,
// bias=False)`), forward and weight gradient (the network input needs no data gradient).  With it -- and the stem's two stride-1
// 3x3 convolutions on conv_lds.hip -- no vendor convolution is left in an SE-CoTNetD training step.
//
// Both kernels are implicit GEMMs on v_mfma_f32_16x16x32_bf16 (operand maps: mfma_common.h) with K = (ci, kh, kw) = 27 taps in the
// weight tensor's own memory order, padded to 32 -- ONE K step:
//   forward   Y (Co x pixels) = Wt (Co x 32) * B (32 x pixels); Wt staged once per workgroup in LDS, B gathered from x (a 600 KB
//             image at 320 x 320: L1 / L2 hits) under bounds predicates.  MFMA column j of column set cs is output pixel 4j + cs,
//             so a lane ends up with 4 consecutive pixels of 4 channels per channel block -> 8-byte stores.  The op is bound by its
//             OUTPUT (210 MB for 64 x 64 x 160 x 160 against 39 MB of input).
//   wgrad     dW (Co x 32) = sum over pixels dY (Co x pixels) * B^T; dY fragments are 16-byte loads (pixels are contiguous), B
//             fragments are 8 stride-2 taps of one (ci, kh, kw) each; deterministic slices + the shared reduce kernel.
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb,
                                hipStream_t stream);  // conv1x1.hip

constexpr int kS3K = 27, kS3Kp = 32;

// tap index k -> (ci, kh, kw)
__device__ __forceinline__ void s3_tap(int k, int& ci, int& kh, int& kw) {
    ci = k / 9;
    const int r = k - 9 * ci;
    kh = r / 3;
    kw = r - 3 * kh;
}

// CB = Co / 16 channel blocks; one wave = 64 consecutive output pixels of one image
template <int CB>
__global__ void __launch_bounds__(256)
stem3x3s2_fwd_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int H, int W, int Ho, int Wo,
                   int tiles_per_image, int64_t total_waves) {
    constexpr int Co = 16 * CB;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(cot_smem);  // [Co][32], zero beyond tap 26
    for (int i = threadIdx.x; i < Co * kS3Kp; i += blockDim.x) {
        const int m = i / kS3Kp, k = i - m * kS3Kp;
        As[i] = k < kS3K ? w[m * kS3K + k] : (bf16_t)0.0f;
    }
    __syncthreads();
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int n = uniform((int)(wid / tiles_per_image)), tile = uniform((int)(wid % tiles_per_image));
    const int HWo = Ho * Wo;
    const int p0 = tile * 64 + 4 * j;  // this lane's 4 consecutive output pixels (one row: Wo % 4 == 0)
    const int oh = min(p0, HWo - 1) / Wo, ow0 = min(p0, HWo - 1) % Wo;
    const int ih0 = 2 * oh - 1, iw0 = 2 * ow0 - 1;
    const bf16_t* xn = x + (int64_t)n * 3 * H * W;

    bf16x8_t bfrag[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        int ci, kh, kw;
        s3_tap(min(k, kS3K - 1), ci, kh, kw);
        const int ih = ih0 + kh;
        const bool rok = k < kS3K && ih >= 0 && ih < H && p0 < HWo;
        const bf16_t* row = xn + ((int64_t)ci * H + (rok ? ih : 0)) * W;
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) {
            const int iw = iw0 + 2 * cs + kw;
            bfrag[cs][e] = (rok && iw >= 0 && iw < W) ? row[iw] : (bf16_t)0.0f;  // (padded taps too: 0 * Inf is not 0)
        }
    }
    // (every lane feeds an A row AND a B column of the wave's products: lanes whose pixels lie behind the image stay until the end)
#pragma unroll
    for (int a = 0; a < CB; ++a) {
        bf16x8_t af;
        __builtin_memcpy(&af, __builtin_assume_aligned(As + (16 * a + j) * kS3Kp + 8 * g, 16), 16);
        f32x4_t acc[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) acc[cs] = COT_MFMA_16X16X32_BF16(af, bfrag[cs], (f32x4_t{0.f, 0.f, 0.f, 0.f}));
        if (p0 >= HWo) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * a + 4 * g + i;
            bf16_t o[4];
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) o[cs] = (bf16_t)acc[cs][i];
            store_piece<4, 8>(y + ((int64_t)n * Co + co) * HWo + p0, o, HWo - p0);
        }
    }
}

// part[s][co][k] = sum over slice s of (n, pixel) of dY[n][co][p] * x[n][ci][2*oh - 1 + kh][2*ow - 1 + kw]
// one wave = slice s (all 32 tap columns as two 16-column MFMA operands); 4 waves per workgroup
template <int CB>
__global__ void __launch_bounds__(256)
stem3x3s2_wgrad_mfma(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, float* __restrict__ part, int N, int H, int W, int Ho,
                     int Wo, int S, int spi) {
    constexpr int Co = 16 * CB;
    const int s = uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (s >= S) return;
    const int lane = threadIdx.x & 63, i16 = lane & 15, lg = lane >> 4;
    const int HWo = Ho * Wo;
    const int T = N * spi, t0 = (int)((int64_t)T * s / S), t1 = (int)((int64_t)T * (s + 1) / S);
    int tci[2], tkh[2], tkw[2];
    bool tok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = q * 16 + i16;
        tok[q] = k < kS3K;
        s3_tap(min(k, kS3K - 1), tci[q], tkh[q], tkw[q]);
    }
    f32x4_t acc[CB][2];
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[a][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int t = t0; t < t1; ++t) {
        const int n = t / spi, st = t - n * spi;
        const int p = st * 32 + 8 * lg;  // 8 consecutive output pixels of one row (Wo % 8 == 0, HWo % 32 == 0)
        const int oh = p / Wo, ow = p - oh * Wo;
        bf16x8_t af[CB], bfr[2];
#pragma unroll
        for (int a = 0; a < CB; ++a)
            __builtin_memcpy(&af[a], __builtin_assume_aligned(gy + ((int64_t)n * Co + 16 * a + i16) * HWo + p, 16), 16);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ih = 2 * oh - 1 + tkh[q];
            const bool rok = tok[q] && ih >= 0 && ih < H;
            const bf16_t* row = x + (((int64_t)n * 3 + tci[q]) * H + (rok ? ih : 0)) * W;
            const int iw0 = 2 * ow - 1 + tkw[q];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iw = iw0 + 2 * e;
                bfr[q][e] = (rok && iw >= 0 && iw < W) ? row[iw] : (bf16_t)0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < CB; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[a][q] = COT_MFMA_16X16X32_BF16(af[a], bfr[q], acc[a][q]);
    }
    float* ps = part + (int64_t)s * Co * kS3K;
#pragma unroll
    for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * a + 4 * lg + i;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = q * 16 + i16;
                if (k < kS3K) ps[co * kS3K + k] = acc[a][q][i];
            }
        }
}

// host side
static bool s3_geometry(int H, int W, int Co, int* Ho, int* Wo) {
    *Ho = (H - 1) / 2 + 1;
    *Wo = (W - 1) / 2 + 1;
    return H > 0 && W > 0 && (Co == 32 || Co == 64) && (*Wo % 8) == 0 && ((*Ho * *Wo) % 32) == 0;
}

int stem3x3s2_splits(int N, int H, int W, int Co) {
    int Ho, Wo;
    if (!s3_geometry(H, W, Co, &Ho, &Wo)) return 0;
    const int64_t T = (int64_t)N * (Ho * Wo / 32);
    int64_t S = 2048;  // two waves per SIMD: the slices are independent chains of strided gathers
    if (S > T / 16) S = T / 16;
    return (int)(S < 1 ? 1 : S);
}

int stem3x3s2_forward(const void* x, const void* w, void* y, int N, int H, int W, int Co, hipStream_t stream) {
    int Ho, Wo;
    if (!s3_geometry(H, W, Co, &Ho, &Wo)) return COT_ERR_UNSUPPORTED;
    const int tpi = ceil_div(Ho * Wo, 64);
    const int64_t waves = (int64_t)N * tpi;
    const dim3 grid((unsigned)ceil_div64(waves, 4));
    if (Co == 64)
        COT_LAUNCH((stem3x3s2_fwd_mfma<4>), grid, dim3(256), Co * kS3Kp * 2, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, H, W,
                   Ho, Wo, tpi, waves);
    else
        COT_LAUNCH((stem3x3s2_fwd_mfma<2>), grid, dim3(256), Co * kS3Kp * 2, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, H, W,
                   Ho, Wo, tpi, waves);
    return check_launch("stem3x3s2_fwd_mfma");
}

int stem3x3s2_wgrad(const void* gy, const void* x, void* gw, float* workspace, int N, int H, int W, int Co, hipStream_t stream) {
    int Ho, Wo;
    if (!s3_geometry(H, W, Co, &Ho, &Wo)) return COT_ERR_UNSUPPORTED;
    const int S = stem3x3s2_splits(N, H, W, Co), spi = Ho * Wo / 32;
    const dim3 grid((unsigned)ceil_div(S, 4));
    if (Co == 64)
        COT_LAUNCH((stem3x3s2_wgrad_mfma<4>), grid, dim3(256), 0, stream, (const bf16_t*)gy, (const bf16_t*)x, workspace, N, H, W, Ho, Wo, S,
                   spi);
    else
        COT_LAUNCH((stem3x3s2_wgrad_mfma<2>), grid, dim3(256), 0, stream, (const bf16_t*)gy, (const bf16_t*)x, workspace, N, H, W, Ho, Wo, S,
                   spi);
    int rc = check_launch("stem3x3s2_wgrad_mfma");
    if (rc) return rc;
    return conv1x1_wgrad_reduce_launch(workspace, S, Co, kS3K, 0, gw, nullptr, stream);
}

}  // namespace cot
