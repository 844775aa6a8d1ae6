// stem7x7_f32.hip -- the backbone's first convolution (7x7, stride 2, padding 3, 3 -> 64 channels, NCHW; reference
// models/resnet.py:539-555) at the REFERENCE'S OWN PRECISION, fp32 (cot_experiments/CoTNet-50-350epoch/config.yaml:2 `amp: False`):
// forward and a deterministic weight gradient (the network input takes no gradient).  The fp32 step is a correctness path on the
// general kernels (conv_gen.hip); this file takes its last vendor convolution away, it is not a tuned kernel: plain fp32 FMAs, the
// reference's accumulation type, 147 taps in the weight tensor's memory order k = (ci, kh, kw).
//   forward   a workgroup = one image x a tile of 8 x 32 output pixels (one lane each) x all 64 channels: the 3 x 21 x 69 input
//             patch (zero-filled outside the image) and the weights transposed to [k][co] live in LDS; per tap a lane reads its
//             patch value and 16 broadcast float4 of weights, 64 FMAs.
//   wgrad     S workgroups walk the (image, 4 x 32-pixel tile) pairs s, s + S, ...; a lane = (output channel co, tap set kq, kq + 4, ...)
//             keeps its 37 partial sums in registers across the tiles it sees (dY tile [px][co] and the patch in LDS: the tap reads are
//             wave-wide broadcasts), writes them to workspace[s][co][k] once; a second kernel adds the S slices in order.
#include "cot_common.h"

namespace cot {

// (the kernels sit in namespace cot itself: `extern __shared__ cot_smem` must name the one array of the host emulation)
constexpr int CO = 64, KT = 147, TH = 8, TW = 32, PH = 2 * TH + 5, PW = 2 * TW + 5;  // patch rows / columns of a tile

__device__ __forceinline__ void stem32_load_patch(const float* __restrict__ xn, float* __restrict__ patch, int H, int W, int oh0, int ow0) {
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    for (int e = threadIdx.x; e < 3 * PH * PW; e += blockDim.x) {
        const int ci = e / (PH * PW), r = (e - ci * PH * PW) / PW, c = e - ci * PH * PW - r * PW;
        const int ih = ih0 + r, iw = iw0 + c;
        patch[e] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xn[((int64_t)ci * H + ih) * W + iw] : 0.f;
    }
}

__global__ __launch_bounds__(256) void stem7x7_fwd_f32(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                      int H, int W, int Ho, int Wo, int tiles_h, int tiles_w) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* wt = reinterpret_cast<float*>(cot_smem);  // [KT][CO]
    float* patch = wt + KT * CO;                     // [3][PH][PW]
    const int t = blockIdx.x % (tiles_h * tiles_w), n = blockIdx.x / (tiles_h * tiles_w);
    const int oh0 = (t / tiles_w) * TH, ow0 = (t % tiles_w) * TW;
    for (int e = threadIdx.x; e < KT * CO; e += blockDim.x) {
        const int co = e / KT, k = e - co * KT;  // (coalesced read of w [co][k], transposed write)
        wt[k * CO + co] = w[e];
    }
    stem32_load_patch(x + (int64_t)n * 3 * H * W, patch, H, W, oh0, ow0);
    __syncthreads();
    const int r = threadIdx.x / TW, c = threadIdx.x % TW;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < 3; ++ci)
        for (int kh = 0; kh < 7; ++kh) {
            const float* prow = patch + (ci * PH + 2 * r + kh) * PW + 2 * c;
            const float* wrow = wt + ((ci * 7 + kh) * 7) * CO;
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float xv = prow[kw];
#pragma unroll
                for (int j4 = 0; j4 < CO / 4; ++j4) {
                    const Vec<float, 4> wv = *reinterpret_cast<const Vec<float, 4>*>(wrow + kw * CO + 4 * j4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * j4 + q] += wv.v[q] * xv;
                }
            }
        }
    const int oh = oh0 + r, ow = ow0 + c;
    if (oh < Ho && ow < Wo) {
        float* yp = y + ((int64_t)n * CO * Ho + oh) * Wo + ow;
#pragma unroll
        for (int j = 0; j < CO; ++j) yp[(int64_t)j * Ho * Wo] = acc[j];
    }
}

constexpr int KQ = 4, NACC = (KT + KQ - 1) / KQ;  // tap sets per channel, partial sums per lane
constexpr int WTH = 4, WPH = 2 * WTH + 5, GS = CO + 1;  // weight gradient: tiles of 4 x 32 pixels (43 KB of LDS: three workgroups per CU);
                                                         // dY tile rows padded to 65 words (pixel-major writes and channel-major reads both spread over the banks)

// four independent loads in flight per lane and pass (a one-load loop was one memory round trip per element: 5.0 ms per call)
__device__ __forceinline__ void stem32_load_patch_w(const float* __restrict__ xn, float* __restrict__ patch, int H, int W, int oh0, int ow0) {
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3, NT = blockDim.x, total = 3 * WPH * PW;
    for (int e0 = threadIdx.x; e0 < total; e0 += 4 * NT) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * NT;
            const int ee = e < total ? e : 0;
            const int ci = ee / (WPH * PW), r = (ee - ci * WPH * PW) / PW, c = ee - ci * WPH * PW - r * PW;
            const int ih = ih0 + r, iw = iw0 + c;
            v[u] = (e < total && ih >= 0 && ih < H && iw >= 0 && iw < W) ? xn[((int64_t)ci * H + ih) * W + iw] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + u * NT < total) patch[e0 + u * NT] = v[u];
    }
}

__global__ __launch_bounds__(256) void stem7x7_wgrad_f32(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ part,
                                                        int N, int H, int W, int Ho, int Wo, int tiles_h, int tiles_w, int S) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* gt = reinterpret_cast<float*>(cot_smem);  // [WTH*TW][GS]: dY of the tile, pixel-major
    float* patch = gt + WTH * TW * GS;
    const int co = threadIdx.x & 63, kq = threadIdx.x >> 6;
    float acc[NACC], live[NACC];
    int off[NACC];  // this lane's taps inside a pixel's 3 x 7 x 7 window of the patch (resolved once: no divisions in the pixel loop)
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        acc[a] = 0.f;
        const int k = kq + KQ * a, kk = k < KT ? k : 0;
        const int ci = kk / 49, rr = (kk - 49 * ci) / 7, cc = kk - 49 * ci - 7 * rr;
        off[a] = (ci * WPH + rr) * PW + cc;
        live[a] = k < KT ? 1.f : 0.f;
    }
    const int tpi = tiles_h * tiles_w;
    const int64_t total = (int64_t)N * tpi;
    for (int64_t it = blockIdx.x; it < total; it += S) {
        const int n = (int)(it / tpi), t = (int)(it - (int64_t)n * tpi);
        const int oh0 = (t / tiles_w) * WTH, ow0 = (t % tiles_w) * TW;
        __syncthreads();  // everybody is done with the previous tile's LDS
        // dY tile: 64 channels x 4 rows x 8 quads of 4 pixels = 2048 16-byte loads, 8 per lane, all issued before the first LDS write
        // (Wo % 4 == 0: a quad is inside the plane or outside as a whole)
        {
            Vec<float, 4> q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = threadIdx.x + u * 256, c2 = e >> 5, rq = e & 31, r = rq >> 3, c4 = (rq & 7) * 4;
                const int oh = oh0 + r, ow = ow0 + c4;
                if (oh < Ho && ow < Wo) q[u] = ldv<float, 4>(gy + (((int64_t)n * CO + c2) * Ho + oh) * Wo + ow);
                else q[u].v[0] = q[u].v[1] = q[u].v[2] = q[u].v[3] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = threadIdx.x + u * 256, c2 = e >> 5, rq = e & 31, px = (rq >> 3) * TW + (rq & 7) * 4;
#pragma unroll
                for (int v = 0; v < 4; ++v) gt[(px + v) * GS + c2] = q[u].v[v];
            }
        }
        stem32_load_patch_w(x + (int64_t)n * 3 * H * W, patch, H, W, oh0, ow0);
        __syncthreads();
        for (int px = 0; px < WTH * TW; ++px) {
            const float g = gt[px * GS + co];
            const int r = px / TW, c = px - r * TW;
            const float* pb = patch + (2 * r) * PW + 2 * c;
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] += g * pb[off[a]];
        }
    }
    float* pp = part + ((int64_t)blockIdx.x * CO + co) * KT;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int k = kq + KQ * a;
        if (live[a] != 0.f) pp[k] = acc[a];
    }
}

__global__ __launch_bounds__(256) void stem7x7_wgrad_reduce_f32(const float* __restrict__ part, int S, float* __restrict__ gw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= CO * KT) return;
    float s = 0.f;
    for (int q = 0; q < S; ++q) s += part[(int64_t)q * CO * KT + e];
    gw[e] = s;
}

// S: the slices of the weight gradient's workspace (the caller sized it for the bf16 kernels' split count: S * 64 * 147 floats)
int stem7x7_f32_forward(const void* x, const void* w, void* y, int N, int H, int W, hipStream_t stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int th = ceil_div(Ho, TH), tw = ceil_div(Wo, TW);
    const size_t lds = ((size_t)KT * CO + 3 * PH * PW) * sizeof(float);
    COT_LAUNCH(stem7x7_fwd_f32, dim3((unsigned)((int64_t)N * th * tw)), dim3(256), lds, stream, (const float*)x, (const float*)w, (float*)y,
               H, W, Ho, Wo, th, tw);
    return check_launch("stem7x7_fwd_f32");
}

int stem7x7_f32_backward_weight(const void* gy, const void* x, void* gw, float* workspace, int S, int N, int H, int W, hipStream_t stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (Wo % 4 != 0) return COT_ERR_UNSUPPORTED;  // (16-byte dY loads; the entry point admits the bf16 kernels' geometry: Wo % 8 == 0)
    const int th = ceil_div(Ho, WTH), tw = ceil_div(Wo, TW);
    if (S < 1) S = 1;
    if ((int64_t)S > (int64_t)N * th * tw) S = (int)((int64_t)N * th * tw);
    const size_t lds = ((size_t)WTH * TW * GS + 3 * WPH * PW) * sizeof(float);
    COT_LAUNCH(stem7x7_wgrad_f32, dim3((unsigned)S), dim3(256), lds, stream, (const float*)gy, (const float*)x, workspace, N, H, W, Ho, Wo,
               th, tw, S);
    int rc = check_launch("stem7x7_wgrad_f32");
    if (rc) return rc;
    COT_LAUNCH(stem7x7_wgrad_reduce_f32, dim3(ceil_div(CO * KT, 256)), dim3(256), 0, stream, (const float*)workspace, S, (float*)gw);
    return check_launch("stem7x7_wgrad_reduce_f32");
}

}  // namespace cot
