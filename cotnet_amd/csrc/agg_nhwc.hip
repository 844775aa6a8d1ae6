// agg_nhwc.hip -- aggregation_zeropad for channels-last storage, gfx950.
//
// Same arithmetic as agg_nchw.hip (reference: cupy_layers/aggregation_zeropad.py:20-110), different
// addressing:  x[N,H,W,C]   w[N,Ho,Wo,heads,wC,taps]   out[N,Ho,Wo,heads*C]
// which is what torch's channels_last memory format gives for the reference's logical shapes (the 6-D
// weight is the `view(b,1,-1,k*k,H,W)` of a channels_last conv output, models/cotnet.py:85).
//
// Lanes run along channels: a thread owns V consecutive channels (V*sizeof(T) = 16 B when wC % V == 0)
// of one output pixel; the V weight channels it needs are consecutive too (c % wC), so its weights are
// the contiguous block w[pixel][wc0*taps .. (wc0+V)*taps).  Threads with the same wc0 (the C/wC channels
// sharing a weight) sit in the same wave for C <= 512 and their weight loads coalesce to one request.
#include "cot_common.h"

namespace cot {

template <typename T, int V>
__global__ __launch_bounds__(256) void agg_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ w,
                                                   T* __restrict__ out, cot_agg_geom g, int Ho, int Wo,
                                                   int64_t items) {
    typedef typename AccOf<T>::type A;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const int chunks = g.C / V;
    const int q = (int)(item % chunks);
    int64_t r = item / chunks;
    const int head = (int)(r % g.heads);
    r /= g.heads;
    const int wo = (int)(r % Wo);
    const int ho = (int)((r / Wo) % Ho);
    const int n = (int)(r / ((int64_t)Wo * Ho));
    const int c0 = q * V;
    const int wc0 = c0 % g.wC;
    const int taps = g.kh * g.kw;
    const int64_t pix = ((int64_t)n * Ho + ho) * Wo + wo;
    const T* wp = w + ((pix * g.heads + head) * g.wC + wc0) * taps;  // [V][taps] block
    A acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = (A)0;
    for (int kh = 0; kh < g.kh; ++kh) {
        const int h_in = -g.ph + ho * g.sh + kh * g.dh;
        for (int kw = 0; kw < g.kw; ++kw) {
            const int w_in = -g.pw + wo * g.sw + kw * g.dw;
            if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W) {
                const Vec<T, V> xv = ldv<T, V>(x + (((int64_t)n * g.H + h_in) * g.W + w_in) * g.C + c0);
                const int t = kh * g.kw + kw;
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += ld(wp + i * taps + t) * (A)xv.v[i];
            }
        }
    }
    Vec<T, V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = (T)acc[i];
    stv<T, V>(out + (pix * g.heads + head) * g.C + c0, o);
}

// 3x3/s1/d1/p1 specialisation: the thread's [V][9] weight block is fetched with 9*V*sizeof(T)/16 wide loads
// and the tap loop is fully unrolled.
template <typename T, int V>
__global__ __launch_bounds__(256) void agg_fwd_nhwc_k3(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ out, int heads, int C, int wC, int H, int W,
                                                      int64_t items) {
    typedef typename AccOf<T>::type A;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const int chunks = C / V;
    const int q = (int)(item % chunks);
    int64_t r = item / chunks;
    const int head = (int)(r % heads);
    r /= heads;  // pixel index n*H*W + h*W + w
    const int wi = (int)(r % W);
    const int hi = (int)((r / W) % H);
    const int c0 = q * V;
    const int wc0 = c0 % wC;
    const T* wp = w + ((r * heads + head) * wC + wc0) * 9;
    A wr[V * 9];
    {
        // 9*V elements, V-aligned start (wc0 % V == 0 and V*sizeof(T) == 16): 9 vector loads
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            Vec<T, V> v = ldv<T, V>(wp + k * V);
#pragma unroll
            for (int i = 0; i < V; ++i) wr[k * V + i] = (A)v.v[i];
        }
    }
    A acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = (A)0;
    const T* xc = x + r * C + c0;  // centre pixel
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h_in = hi - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w_in = wi - 1 + kw;
            if (h_in >= 0 && h_in < H && w_in >= 0 && w_in < W) {
                const Vec<T, V> xv = ldv<T, V>(xc + ((int64_t)(kh - 1) * W + (kw - 1)) * C);
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += wr[i * 9 + kh * 3 + kw] * (A)xv.v[i];
            }
        }
    }
    Vec<T, V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = (T)acc[i];
    stv<T, V>(out + (r * heads + head) * C + c0, o);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void agg_bwd_input_nhwc(const T* __restrict__ gout, const T* __restrict__ w,
                                                         T* __restrict__ gx, cot_agg_geom g, int Ho, int Wo,
                                                         int64_t items) {
    typedef typename AccOf<T>::type A;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const int chunks = g.C / V;
    const int q = (int)(item % chunks);
    const int64_t r = item / chunks;  // n*H*W + h*W + w
    const int wi = (int)(r % g.W);
    const int hi = (int)((r / g.W) % g.H);
    const int n = (int)(r / ((int64_t)g.W * g.H));
    const int c0 = q * V;
    const int wc0 = c0 % g.wC;
    const int taps = g.kh * g.kw;
    A acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = (A)0;
    for (int head = 0; head < g.heads; ++head) {
        for (int kh = 0; kh < g.kh; ++kh) {
            const int h_out_s = hi + g.ph - kh * g.dh;
            for (int kw = 0; kw < g.kw; ++kw) {
                const int w_out_s = wi + g.pw - kw * g.dw;
                if ((h_out_s % g.sh) == 0 && (w_out_s % g.sw) == 0) {
                    const int h_out = h_out_s / g.sh, w_out = w_out_s / g.sw;
                    if (h_out >= 0 && h_out < Ho && w_out >= 0 && w_out < Wo) {
                        const int64_t pix = ((int64_t)n * Ho + h_out) * Wo + w_out;
                        const Vec<T, V> gv = ldv<T, V>(gout + (pix * g.heads + head) * g.C + c0);
                        const T* wp = w + ((pix * g.heads + head) * g.wC + wc0) * taps + kh * g.kw + kw;
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[i] += ld(wp + i * taps) * (A)gv.v[i];
                    }
                }
            }
        }
    }
    Vec<T, V> o;
#pragma unroll
    for (int i = 0; i < V; ++i) o.v[i] = (T)acc[i];
    stv<T, V>(gx + r * g.C + c0, o);
}

// one thread per (pixel, head, wc, tap); lanes along (wc, tap) = contiguous in gw.
template <typename T>
__global__ __launch_bounds__(256) void agg_bwd_weight_nhwc(const T* __restrict__ gout, const T* __restrict__ x,
                                                          T* __restrict__ gw, cot_agg_geom g, int Ho, int Wo,
                                                          int64_t total) {
    typedef typename AccOf<T>::type A;
    const int taps = g.kh * g.kw;
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(index % taps);
        int64_t r = index / taps;
        const int wc = (int)(r % g.wC);
        r /= g.wC;
        const int head = (int)(r % g.heads);
        const int64_t pix = r / g.heads;
        const int wo = (int)(pix % Wo);
        const int ho = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((int64_t)Wo * Ho));
        const int kh = tap / g.kw, kw = tap % g.kw;
        const int h_in = -g.ph + ho * g.sh + kh * g.dh;
        const int w_in = -g.pw + wo * g.sw + kw * g.dw;
        A value = 0;
        if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W) {
            const T* xp = x + (((int64_t)n * g.H + h_in) * g.W + w_in) * g.C;
            const T* gp = gout + (pix * g.heads + head) * g.C;
            for (int cc = wc; cc < g.C; cc += g.wC) value += ld(xp + cc) * ld(gp + cc);
        }
        st(gw + index, value);
    }
}

static const char* g_last_kernel_nhwc = "";
const char* last_kernel_nhwc() { return g_last_kernel_nhwc; }

static inline int grid1d(int64_t total, int block, int64_t cap) {
    int64_t b = ceil_div64(total, block);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T> static inline int pick_V(const cot_agg_geom& g, int max_by_align) {
    int lim = (int)(16 / sizeof(T));
    if (lim > max_by_align) lim = max_by_align;
    for (int V = 8; V >= 1; V >>= 1)
        if (V <= lim && g.wC % V == 0 && g.C % V == 0) return V;
    return 1;
}

template <typename T, int V>
static int launch_fwd(const T* x, const T* w, T* out, const cot_agg_geom& g, int Ho, int Wo, hipStream_t s) {
    const int64_t items = (int64_t)g.N * Ho * Wo * g.heads * (g.C / V);
    const bool k3 = g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.dh == 1 &&
                    g.dw == 1 && V * sizeof(T) == 16;
    if (k3) {
        COT_LAUNCH((agg_fwd_nhwc_k3<T, V>), dim3(grid1d(items, 256, INT32_MAX)), dim3(256), 0, s, x, w, out,
                           g.heads, g.C, g.wC, g.H, g.W, items);
        g_last_kernel_nhwc = "agg_fwd_nhwc_k3";
    } else {
        COT_LAUNCH((agg_fwd_nhwc<T, V>), dim3(grid1d(items, 256, INT32_MAX)), dim3(256), 0, s, x, w, out, g,
                           Ho, Wo, items);
        g_last_kernel_nhwc = "agg_fwd_nhwc";
    }
    return check_launch("agg_fwd_nhwc");
}

template <typename T>
int agg_forward_nhwc(const T* x, const T* w, T* out, const cot_agg_geom& g, int Ho, int Wo, int max_vec,
                     hipStream_t s) {
    switch (pick_V<T>(g, max_vec)) {
        case 8: return launch_fwd<T, (sizeof(T) <= 2 ? 8 : 1)>(x, w, out, g, Ho, Wo, s);
        case 4: return launch_fwd<T, (sizeof(T) <= 4 ? 4 : 1)>(x, w, out, g, Ho, Wo, s);
        case 2: return launch_fwd<T, 2>(x, w, out, g, Ho, Wo, s);
        default: return launch_fwd<T, 1>(x, w, out, g, Ho, Wo, s);
    }
}

template <typename T, int V>
static int launch_bwd_in(const T* gout, const T* w, T* gx, const cot_agg_geom& g, int Ho, int Wo, hipStream_t s) {
    const int64_t items = (int64_t)g.N * g.H * g.W * (g.C / V);
    COT_LAUNCH((agg_bwd_input_nhwc<T, V>), dim3(grid1d(items, 256, INT32_MAX)), dim3(256), 0, s, gout, w, gx, g,
                       Ho, Wo, items);
    g_last_kernel_nhwc = "agg_bwd_input_nhwc";
    return check_launch("agg_bwd_input_nhwc");
}

template <typename T>
int agg_backward_nhwc(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g, int Ho, int Wo,
                      int max_vec, hipStream_t s) {
    int rc = COT_OK;
    if (gx) {
        switch (pick_V<T>(g, max_vec)) {
            case 8: rc = launch_bwd_in<T, (sizeof(T) <= 2 ? 8 : 1)>(gout, w, gx, g, Ho, Wo, s); break;
            case 4: rc = launch_bwd_in<T, (sizeof(T) <= 4 ? 4 : 1)>(gout, w, gx, g, Ho, Wo, s); break;
            case 2: rc = launch_bwd_in<T, 2>(gout, w, gx, g, Ho, Wo, s); break;
            default: rc = launch_bwd_in<T, 1>(gout, w, gx, g, Ho, Wo, s); break;
        }
        if (rc) return rc;
    }
    if (gw) {
        const int64_t total = (int64_t)g.N * Ho * Wo * g.heads * g.wC * g.kh * g.kw;
        COT_LAUNCH((agg_bwd_weight_nhwc<T>), dim3(grid1d(total, 256, (int64_t)1 << 20)), dim3(256), 0, s, gout,
                           x, gw, g, Ho, Wo, total);
        g_last_kernel_nhwc = "agg_bwd_weight_nhwc";
        rc = check_launch("agg_bwd_weight_nhwc");
    }
    return rc;
}

#define INSTANTIATE(T)                                                                                         \
    template int agg_forward_nhwc<T>(const T*, const T*, T*, const cot_agg_geom&, int, int, int, hipStream_t); \
    template int agg_backward_nhwc<T>(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, int, int, int,  \
                                      hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
INSTANTIATE(bf16_t)
INSTANTIATE(f16_t)

}  // namespace cot
