// gn9_nhwc.hip -- STUDY (the channels-last route): GroupNorm with 9 channels per group on channels-last attention
// logits x[N][HW][9 * G] (CotLayer.embed[4] = nn.GroupNorm(dim/8, 9*dim/8), models/cotnet.py:56; csrc/group_norm9.hip is the NCHW
// implementation).  A group is 9 consecutive channels of every pixel of one image: a few workgroups per image (each a range of groups), a thread owns one
// group of a row (18 contiguous bytes; a wave reads 64 such blocks back to back) and walks down the rows; statistics with the
// group's first element as shift, two passes over the image inside the launch (the second one hits L2: an image's logits are
// 9 * G * HW * 2 bytes = 113 KB at 14 x 14).  Exported as cot_study_group_norm9_nhwc_*; host-emulated tests only.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cot_common.h"

namespace cot {

// sums over the RP row lanes of a group, in row-lane order, through LDS; returns the totals to every thread of the group
template <int NV>
__device__ __forceinline__ void gn9_group_sum(float (&v)[NV], float* sm, int G, int RP, int wc, int rl) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sm[(rl * G + wc) * NV + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float a = 0.f;
        for (int j = 0; j < RP; ++j) a += sm[(j * G + wc) * NV + k];
        v[k] = a;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void gn9_nhwc_fwd(const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                   T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int HW, int G,
                                                   float eps) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    // gridDim.y workgroups share an image's groups: GL groups each, 256 / GL row lanes
    const int GL = G / (int)gridDim.y, n = blockIdx.x, RP = 256 / GL, wl = threadIdx.x % GL, rl = threadIdx.x / GL, C = 9 * G;
    const int wc = blockIdx.y * GL + wl;
    const T* xi = x + (int64_t)n * HW * C + wc * 9;
    T* yi = y + (int64_t)n * HW * C + wc * 9;
    const float shift = (float)xi[0];
    float s[2] = {0.f, 0.f};
    for (int r = rl; r < HW; r += RP) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float d = (float)xi[(int64_t)r * C + t] - shift;
            s[0] += d;
            s[1] += d * d;
        }
    }
    gn9_group_sum<2>(s, sm, GL, RP, wl, rl);
    const float cnt = 9.f * (float)HW, md = s[0] / cnt, mu = shift + md;
    const float var = fmaxf(s[1] / cnt - md * md, 0.f), rs = 1.0f / sqrtf(var + eps);
    if (rl == 0) {
        mean[n * G + wc] = mu;
        rstd[n * G + wc] = rs;
    }
    float a[9], b[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        a[t] = (float)gamma[wc * 9 + t] * rs;
        b[t] = (float)beta[wc * 9 + t] - mu * a[t];
    }
    for (int r = rl; r < HW; r += RP) {
#pragma unroll
        for (int t = 0; t < 9; ++t) yi[(int64_t)r * C + t] = (T)((float)xi[(int64_t)r * C + t] * a[t] + b[t]);
    }
}

// dx of one image; per-image partial sums of dgamma / dbeta -> part[n][C][2]
template <typename T>
__global__ __launch_bounds__(256) void gn9_nhwc_bwd(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const T* __restrict__ gamma, T* __restrict__ dx,
                                                   float* __restrict__ part, int HW, int G) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    const int GL = G / (int)gridDim.y, n = blockIdx.x, RP = 256 / GL, wl = threadIdx.x % GL, rl = threadIdx.x / GL, C = 9 * G;
    const int wc = blockIdx.y * GL + wl;
    const int64_t base = (int64_t)n * HW * C + wc * 9;
    const float mu = mean[n * G + wc], rs = rstd[n * G + wc];
    float ga[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) ga[t] = (float)gamma[wc * 9 + t];
    float v[20];  // [0] sum g', [1] sum g' * xhat, [2 + t] sum dy (dbeta_t), [11 + t] sum dy * xhat (dgamma_t)
#pragma unroll
    for (int k = 0; k < 20; ++k) v[k] = 0.f;
    for (int r = rl; r < HW; r += RP) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float d = (float)dy[base + (int64_t)r * C + t], xh = ((float)x[base + (int64_t)r * C + t] - mu) * rs;
            v[0] += d * ga[t];
            v[1] += d * ga[t] * xh;
            v[2 + t] += d;
            v[11 + t] += d * xh;
        }
    }
    gn9_group_sum<20>(v, sm, GL, RP, wl, rl);
    if (rl == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float* p = part + ((int64_t)n * C + wc * 9 + t) * 2;
            p[0] = v[11 + t];
            p[1] = v[2 + t];
        }
    }
    const float inv = 1.0f / (9.f * (float)HW), m1 = v[0] * inv, m2 = v[1] * inv;
    for (int r = rl; r < HW; r += RP) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float d = (float)dy[base + (int64_t)r * C + t], xh = ((float)x[base + (int64_t)r * C + t] - mu) * rs;
            dx[base + (int64_t)r * C + t] = (T)(rs * (d * ga[t] - m1 - xh * m2));
        }
    }
}

// dgamma[c] = sum over images of part[n][c][0], dbeta[c] likewise (in image order)
template <typename T>
__global__ __launch_bounds__(256) void gn9_nhwc_param_reduce(const float* __restrict__ part, T* __restrict__ dgamma, T* __restrict__ dbeta, int N,
                                                            int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) {
        g += part[((int64_t)n * C + c) * 2];
        b += part[((int64_t)n * C + c) * 2 + 1];
    }
    dgamma[c] = (T)g;
    dbeta[c] = (T)b;
}

// workgroups per image: enough for ~512 in the launch, at least 4 groups each (G a power of two)
static int gn9_nhwc_split(int N, int G) {
    int s = 1;
    while (s * 2 * N <= 512 && G / (s * 2) >= 4) s *= 2;
    return s;
}

template <typename T>
static int gn9_nhwc_run_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N, int C, int HW,
                            float eps, hipStream_t s) {
    const int G = C / 9;
    COT_LAUNCH((gn9_nhwc_fwd<T>), dim3(N, gn9_nhwc_split(N, G)), dim3(256), 256 * 2 * sizeof(float), s, (const T*)x, (const T*)gamma, (const T*)beta, (T*)y, mean, rstd,
               HW, G, eps);
    return check_launch("gn9_nhwc_forward");
}
template <typename T>
static int gn9_nhwc_run_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, void* dx, void* dgamma,
                            void* dbeta, float* ws, int N, int C, int HW, hipStream_t s) {
    const int G = C / 9;
    COT_LAUNCH((gn9_nhwc_bwd<T>), dim3(N, gn9_nhwc_split(N, G)), dim3(256), 256 * 20 * sizeof(float), s, (const T*)dy, (const T*)x, mean, rstd, (const T*)gamma, (T*)dx,
               ws, HW, G);
    COT_LAUNCH((gn9_nhwc_param_reduce<T>), dim3(ceil_div(C, 256)), dim3(256), 0, s, (const float*)ws, (T*)dgamma, (T*)dbeta, N, C);
    return check_launch("gn9_nhwc_backward");
}

}  // namespace cot

// the arguments of cot_group_norm9_forward / _backward on x[N][HW][C]; C = 9 * G with G a power of two up to 256 (the CoT layers:
// 8 .. 64); workspace (backward): N * C * 2 floats
static int gn9_nhwc_covers(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return -1;
    const int G = C / 9;
    if (C % 9 || G > 256 || (G & (G - 1))) return -2;
    return 0;
}
extern "C" int cot_study_group_norm9_nhwc_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N,
                                                  int C, int HW, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return -1;
    const int rc = gn9_nhwc_covers(N, C, HW);
    if (rc) return rc;
    if (dtype == 2) return cot::gn9_nhwc_run_fwd<cot::bf16_t>(x, gamma, beta, y, mean, rstd, N, C, HW, eps, (hipStream_t)stream);
    if (dtype == 0) return cot::gn9_nhwc_run_fwd<float>(x, gamma, beta, y, mean, rstd, N, C, HW, eps, (hipStream_t)stream);
    return -2;
}
extern "C" int cot_study_group_norm9_nhwc_backward(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma,
                                                   void* dx, void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int dtype,
                                                   void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || !workspace) return -1;
    const int rc = gn9_nhwc_covers(N, C, HW);
    if (rc) return rc;
    if (dtype == 2)
        return cot::gn9_nhwc_run_bwd<cot::bf16_t>(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, N, C, HW, (hipStream_t)stream);
    if (dtype == 0) return cot::gn9_nhwc_run_bwd<float>(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, N, C, HW, (hipStream_t)stream);
    return -2;
}
