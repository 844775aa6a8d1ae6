// agg_mix.hip -- aggregation_zeropad_mix (3x3 + 5x5 taps over the same input), NCHW, gfx950.
//
// Reference: cupy_layers/aggregation_zeropad_mix.py:20-74 (forward), :76-140 (input backward),
// :142-207 (weight backward).  The op is not on any model path in the reference (LocalConvolutionMix is
// never instantiated), so these are straightforward one-thread-per-element kernels; the 3x3 and 5x5 tap
// loops are compile-time unrolled like the reference's hard-coded loops (:35-36, :53-54).
//
// Reference quirk kept on purpose: input backward sums head 0 only (:87-88).  `all_heads` != 0 gives
// the complete gradient instead.
#include "cot_common.h"

namespace cot {

template <typename T, int K>
__device__ __forceinline__ typename AccOf<T>::type mix_fwd_taps(const T* __restrict__ xp, const T* __restrict__ wp,
                                                               int ho, int wo, int ph, int pw,
                                                               const cot_agg_geom& g, int64_t HoWo) {
    typedef typename AccOf<T>::type A;
    A value = 0;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int h_in = -ph + ho * g.sh + kh * g.dh;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int w_in = -pw + wo * g.sw + kw * g.dw;
            if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W)
                value += ld(wp + (int64_t)(kh * K + kw) * HoWo) * ld(xp + (int64_t)h_in * g.W + w_in);
        }
    }
    return value;
}

template <typename T>
__global__ __launch_bounds__(256) void aggmix_fwd(const T* __restrict__ x, const T* __restrict__ w1,
                                                 const T* __restrict__ w2, T* __restrict__ out, cot_agg_geom g,
                                                 int p2h, int p2w, int Ho, int Wo, int64_t total) {
    const int64_t HoWo = (int64_t)Ho * Wo;
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wo = (int)(index % Wo);
        const int ho = (int)((index / Wo) % Ho);
        int64_t r = index / HoWo;  // ((n*2 + kidx)*heads + head)*C + c
        const int c = (int)(r % g.C);
        r /= g.C;
        const int head = (int)(r % g.heads);
        r /= g.heads;
        const int kidx = (int)(r % 2);
        const int n = (int)(r / 2);
        const T* xp = x + ((int64_t)n * g.C + c) * g.H * g.W;
        const int64_t wplane = ((int64_t)n * g.heads + head) * g.wC + c % g.wC;
        const int64_t o = (int64_t)ho * Wo + wo;
        typename AccOf<T>::type v;
        if (kidx == 0)
            v = mix_fwd_taps<T, 3>(xp, w1 + wplane * 9 * HoWo + o, ho, wo, g.ph, g.pw, g, HoWo);
        else
            v = mix_fwd_taps<T, 5>(xp, w2 + wplane * 25 * HoWo + o, ho, wo, p2h, p2w, g, HoWo);
        st(out + index, v);
    }
}

template <typename T, int K>
__device__ __forceinline__ void mix_gin_taps(typename AccOf<T>::type& value, const T* __restrict__ gp,
                                             const T* __restrict__ wp, int hi, int wi, int ph, int pw,
                                             const cot_agg_geom& g, int Ho, int Wo, int64_t HoWo) {
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
        const int h_out_s = hi + ph - kh * g.dh;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            const int w_out_s = wi + pw - kw * g.dw;
            if ((h_out_s % g.sh) == 0 && (w_out_s % g.sw) == 0) {
                const int h_out = h_out_s / g.sh, w_out = w_out_s / g.sw;
                if (h_out >= 0 && h_out < Ho && w_out >= 0 && w_out < Wo) {
                    const int64_t o = (int64_t)h_out * Wo + w_out;
                    value += ld(wp + (int64_t)(kh * K + kw) * HoWo + o) * ld(gp + o);
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void aggmix_bwd_input(const T* __restrict__ gout, const T* __restrict__ w1,
                                                       const T* __restrict__ w2, T* __restrict__ gx,
                                                       cot_agg_geom g, int p2h, int p2w, int nheads_used, int Ho,
                                                       int Wo, int64_t total) {
    typedef typename AccOf<T>::type A;
    const int64_t HoWo = (int64_t)Ho * Wo;
    const int64_t HW = (int64_t)g.H * g.W;
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wi = (int)(index % g.W);
        const int hi = (int)((index / g.W) % g.H);
        const int64_t plane = index / HW;
        const int c = (int)(plane % g.C);
        const int n = (int)(plane / g.C);
        A value = 0;
        for (int head = 0; head < nheads_used; ++head) {
            const int64_t wplane = ((int64_t)n * g.heads + head) * g.wC + c % g.wC;
            const T* g1 = gout + ((((int64_t)n * 2 + 0) * g.heads + head) * g.C + c) * HoWo;
            const T* g2 = gout + ((((int64_t)n * 2 + 1) * g.heads + head) * g.C + c) * HoWo;
            mix_gin_taps<T, 3>(value, g1, w1 + wplane * 9 * HoWo, hi, wi, g.ph, g.pw, g, Ho, Wo, HoWo);
            mix_gin_taps<T, 5>(value, g2, w2 + wplane * 25 * HoWo, hi, wi, p2h, p2w, g, Ho, Wo, HoWo);
        }
        st(gx + index, value);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void aggmix_bwd_weight(const T* __restrict__ gout, const T* __restrict__ x,
                                                        T* __restrict__ gw1, T* __restrict__ gw2, cot_agg_geom g,
                                                        int p2h, int p2w, int Ho, int Wo, int64_t total) {
    typedef typename AccOf<T>::type A;
    const int64_t HoWo = (int64_t)Ho * Wo;
    const int64_t HW = (int64_t)g.H * g.W;
    // one thread per (n, head, wc, tap of the 9+25=34 taps, ho, wo)
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wo = (int)(index % Wo);
        const int ho = (int)((index / Wo) % Ho);
        int64_t r = index / HoWo;
        const int tap34 = (int)(r % 34);
        const int64_t plane = r / 34;  // (n*heads + head)*wC + wc
        const int wc = (int)(plane % g.wC);
        const int64_t nh = plane / g.wC;
        const int head = (int)(nh % g.heads);
        const int n = (int)(nh / g.heads);
        const int kidx = tap34 < 9 ? 0 : 1;
        const int K = kidx ? 5 : 3;
        const int tap = kidx ? tap34 - 9 : tap34;
        const int kh = tap / K, kw = tap % K;
        const int h_in = -(kidx ? p2h : g.ph) + ho * g.sh + kh * g.dh;
        const int w_in = -(kidx ? p2w : g.pw) + wo * g.sw + kw * g.dw;
        A value = 0;
        if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W) {
            const T* gp = gout + (((int64_t)n * 2 + kidx) * g.heads + head) * g.C * HoWo + (int64_t)ho * Wo + wo;
            const T* xp = x + (int64_t)n * g.C * HW + (int64_t)h_in * g.W + w_in;
            for (int cc = wc; cc < g.C; cc += g.wC) value += ld(xp + (int64_t)cc * HW) * ld(gp + (int64_t)cc * HoWo);
        }
        T* dst = kidx ? gw2 : gw1;
        st(dst + (plane * (K * K) + tap) * HoWo + (int64_t)ho * Wo + wo, value);
    }
}

static inline int grid1d(int64_t total) {
    int64_t b = ceil_div64(total, 256);
    if (b > ((int64_t)1 << 20)) b = (int64_t)1 << 20;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
int aggmix_forward(const T* x, const T* w1, const T* w2, T* out, const cot_agg_geom& g, int p2h, int p2w, int Ho,
                   int Wo, hipStream_t s) {
    const int64_t total = (int64_t)g.N * 2 * g.heads * g.C * Ho * Wo;
    COT_LAUNCH((aggmix_fwd<T>), dim3(grid1d(total)), dim3(256), 0, s, x, w1, w2, out, g, p2h, p2w, Ho, Wo,
                       total);
    return check_launch("aggmix_fwd");
}
template <typename T>
int aggmix_backward_input(const T* gout, const T* w1, const T* w2, T* gx, const cot_agg_geom& g, int p2h, int p2w,
                          int all_heads, int Ho, int Wo, hipStream_t s) {
    const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
    COT_LAUNCH((aggmix_bwd_input<T>), dim3(grid1d(total)), dim3(256), 0, s, gout, w1, w2, gx, g, p2h, p2w,
                       all_heads ? g.heads : 1, Ho, Wo, total);
    return check_launch("aggmix_bwd_input");
}
template <typename T>
int aggmix_backward_weight(const T* gout, const T* x, T* gw1, T* gw2, const cot_agg_geom& g, int p2h, int p2w,
                           int Ho, int Wo, hipStream_t s) {
    const int64_t total = (int64_t)g.N * g.heads * g.wC * 34 * Ho * Wo;
    COT_LAUNCH((aggmix_bwd_weight<T>), dim3(grid1d(total)), dim3(256), 0, s, gout, x, gw1, gw2, g, p2h, p2w,
                       Ho, Wo, total);
    return check_launch("aggmix_bwd_weight");
}

#define INSTANTIATE(T)                                                                                          \
    template int aggmix_forward<T>(const T*, const T*, const T*, T*, const cot_agg_geom&, int, int, int, int,    \
                                   hipStream_t);                                                                \
    template int aggmix_backward_input<T>(const T*, const T*, const T*, T*, const cot_agg_geom&, int, int, int, \
                                          int, int, hipStream_t);                                               \
    template int aggmix_backward_weight<T>(const T*, const T*, T*, T*, const cot_agg_geom&, int, int, int, int, \
                                           hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
INSTANTIATE(bf16_t)
INSTANTIATE(f16_t)

}  // namespace cot
