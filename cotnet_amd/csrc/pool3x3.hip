// pool3x3.hip -- the two 3x3 / stride-2 / padding-1 poolings of the CoTNet backbone, NCHW, forward and backward:
//   max pooling after the stem  (reference: models/resnet.py:556-561, nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
//   average pooling "avd" in front of the CoT layer of every stride-2 block (models/cotnet.py:216, nn.AvgPool2d(3, 2,
//   padding=1): count_include_pad=True, i.e. every window is divided by 9)
// Why: torch's backward kernels for these two are far off the memory roofline on gfx950 (round-1 profile:
// max_pool_backward_nchw 436 us for 80x64x112x112, avg_pool2d_backward 163 us per call; both are ~30 us of traffic), and
// the max-pool backward reads an int64 index tensor twice the size of its gradient.  Here the backward of the max pooling
// recomputes the arg-max from x with torch's tie rule (first maximum in row-major window order, `>` comparison, NaN
// wins -- after a ReLU ties at zero are the common case), so no index tensor exists.
// One thread per output pixel (forward) / input pixel (backward, gather form: no atomics), threads along W.
#include "cot_common.h"

namespace cot {

template <typename T>
__global__ __launch_bounds__(256) void avgpool3x3s2_fwd(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H,
                                                       int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho);
    const int64_t pl = i / ((int64_t)Wo * Ho);
    const T* xp = x + pl * H * W;
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * ow - 1 + kw;
            if (w >= 0 && w < W) s += (float)xp[h * W + w];
        }
    }
    y[i] = (T)(s * (1.f / 9.f));
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool3x3s2_bwd(const T* __restrict__ gy, T* __restrict__ gx, int64_t planes,
                                                       int H, int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)W * H);
    const T* gp = gy + pl * Ho * Wo;
    // windows (oh, ow) containing (h, w): 2*oh - 1 <= h <= 2*oh + 1
    float s = 0.f;
    const int oh0 = h / 2, oh1 = (h + 1) / 2, ow0 = w / 2, ow1 = (w + 1) / 2;  // oh in {oh0, oh1} (equal when h is even)
    for (int oh = oh0; oh <= oh1; ++oh) {
        if (oh >= Ho) continue;
        for (int ow = ow0; ow <= ow1; ++ow)
            if (ow < Wo) s += (float)gp[oh * Wo + ow];
    }
    gx[i] = (T)(s * (1.f / 9.f));
}

// arg-max of window (oh, ow) in torch's order: rows then columns, strictly-greater wins, NaN wins; -> h*W + w
template <typename T> __device__ __forceinline__ int window_argmax(const T* xp, int oh, int ow, int H, int W) {
    float best = -INFINITY;
    int idx = -1;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * ow - 1 + kw;
            if (w < 0 || w >= W) continue;
            const float v = (float)xp[h * W + w];
            if (v > best || v != v || idx < 0) {
                best = v;
                idx = h * W + w;
            }
        }
    }
    return idx;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H,
                                                       int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho);
    const int64_t pl = i / ((int64_t)Wo * Ho);
    const T* xp = x + pl * H * W;
    y[i] = xp[window_argmax<T>(xp, oh, ow, H, W)];
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd(const T* __restrict__ gy, const T* __restrict__ x,
                                                       T* __restrict__ gx, int64_t planes, int H, int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)W * H);
    const T* xp = x + pl * H * W;
    const T* gp = gy + pl * Ho * Wo;
    float s = 0.f;
    const int oh0 = h / 2, oh1 = (h + 1) / 2, ow0 = w / 2, ow1 = (w + 1) / 2;
    for (int oh = oh0; oh <= oh1; ++oh) {
        if (oh >= Ho) continue;
        for (int ow = ow0; ow <= ow1; ++ow)
            if (ow < Wo && window_argmax<T>(xp, oh, ow, H, W) == h * W + w) s += (float)gp[oh * Wo + ow];
    }
    gx[i] = (T)s;
}

template <typename T>
int pool3x3s2(int op, const void* a, const void* b, void* out, int64_t planes, int H, int W, hipStream_t stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
    const int64_t n_out = planes * Ho * Wo, n_in = planes * H * W;
    const dim3 block(256);
    switch (op) {
        case 0: COT_LAUNCH((avgpool3x3s2_fwd<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 1: COT_LAUNCH((avgpool3x3s2_bwd<T>), dim3((unsigned)ceil_div64(n_in, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 2: COT_LAUNCH((maxpool3x3s2_fwd<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        default: COT_LAUNCH((maxpool3x3s2_bwd<T>), dim3((unsigned)ceil_div64(n_in, 256)), block, 0, stream, (const T*)a, (const T*)b, (T*)out, planes, H, W, Ho, Wo); break;
    }
    return check_launch("pool3x3s2");
}
template int pool3x3s2<float>(int, const void*, const void*, void*, int64_t, int, int, hipStream_t);
template int pool3x3s2<bf16_t>(int, const void*, const void*, void*, int64_t, int, int, hipStream_t);

}  // namespace cot
