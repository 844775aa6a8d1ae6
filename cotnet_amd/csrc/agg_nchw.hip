// agg_nchw.hip -- aggregation_zeropad for the reference (NCHW-contiguous) layout, gfx950.
//
// Semantics: cupy_layers/aggregation_zeropad.py:20-46 (forward), :48-79 (input backward),
// :81-110 (weight backward).  Two families:
//
//  * generic kernels (any kernel size / stride / padding / dilation / heads): one thread per output
//    element, tap loops in the reference's order (kh outer, kw inner) so fp32/fp64 sums associate
//    identically; 64-bit flat indices.
//
//  * the 3x3 / stride 1 / dilation 1 / pad 1 fast path every model call site uses
//    (models/cotnet.py:64, models/cotnet_hybrid.py:76).  Work item = (n*heads, wc, row h, P consecutive
//    pixels): the 9*P weights of the item stay in registers and are reused by all C/wC channels that
//    share them (c = wc + j*wC) -- the reference re-reads them once per channel through L2.
//    Rows are contiguous in NCHW, so every load is a P-wide vector along W (16 B for P*sizeof(T)=16)
//    plus two scalar halo columns that hit the neighbouring lane's cache line.
//    The fused backward produces gX and gW from one pass over gO / x / w (5.25 elements of HBM traffic
//    per output element instead of 6.25 for the reference's two kernels).
#include "cot_common.h"

namespace cot {

// ------------------------------------------------------------------------------------------------
// generic kernels
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void agg_fwd_nchw_generic(const T* __restrict__ x, const T* __restrict__ w,
                                                           T* __restrict__ out, cot_agg_geom g, int Ho, int Wo,
                                                           int64_t total) {
    typedef typename AccOf<T>::type A;
    const int64_t HoWo = (int64_t)Ho * Wo;
    const int taps = g.kh * g.kw;
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wo = (int)(index % Wo);
        const int ho = (int)((index / Wo) % Ho);
        const int64_t plane = index / HoWo;  // (n*heads + head)*C + c
        const int c = (int)(plane % g.C);
        const int64_t nh = plane / g.C;
        const int n = (int)(nh / g.heads);
        const T* xp = x + ((int64_t)n * g.C + c) * g.H * g.W;
        const T* wp = w + ((nh * g.wC + c % g.wC) * taps) * HoWo + (int64_t)ho * Wo + wo;
        A value = 0;
        for (int kh = 0; kh < g.kh; ++kh) {
            const int h_in = -g.ph + ho * g.sh + kh * g.dh;
            for (int kw = 0; kw < g.kw; ++kw) {
                const int w_in = -g.pw + wo * g.sw + kw * g.dw;
                if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W) {
                    value += ld(wp + (int64_t)(kh * g.kw + kw) * HoWo) * ld(xp + (int64_t)h_in * g.W + w_in);
                }
            }
        }
        st(out + index, value);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void agg_bwd_input_nchw_generic(const T* __restrict__ gout,
                                                                 const T* __restrict__ w, T* __restrict__ gx,
                                                                 cot_agg_geom g, int Ho, int Wo, int64_t total) {
    typedef typename AccOf<T>::type A;
    const int64_t HoWo = (int64_t)Ho * Wo;
    const int64_t HW = (int64_t)g.H * g.W;
    const int taps = g.kh * g.kw;
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wi = (int)(index % g.W);
        const int hi = (int)((index / g.W) % g.H);
        const int64_t plane = index / HW;  // n*C + c
        const int c = (int)(plane % g.C);
        const int n = (int)(plane / g.C);
        A value = 0;
        for (int head = 0; head < g.heads; ++head) {
            const int64_t nh = (int64_t)n * g.heads + head;
            const T* gp = gout + (nh * g.C + c) * HoWo;
            const T* wp = w + ((nh * g.wC + c % g.wC) * taps) * HoWo;
            for (int kh = 0; kh < g.kh; ++kh) {
                const int h_out_s = hi + g.ph - kh * g.dh;
                for (int kw = 0; kw < g.kw; ++kw) {
                    const int w_out_s = wi + g.pw - kw * g.dw;
                    // truncating % and / on possibly negative ints, exactly as the reference (:62-66);
                    // negative multiples are rejected by the bounds test below.
                    if ((h_out_s % g.sh) == 0 && (w_out_s % g.sw) == 0) {
                        const int h_out = h_out_s / g.sh;
                        const int w_out = w_out_s / g.sw;
                        if (h_out >= 0 && h_out < Ho && w_out >= 0 && w_out < Wo) {
                            const int64_t o = (int64_t)h_out * Wo + w_out;
                            value += ld(wp + (int64_t)(kh * g.kw + kw) * HoWo + o) * ld(gp + o);
                        }
                    }
                }
            }
        }
        st(gx + index, value);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void agg_bwd_weight_nchw_generic(const T* __restrict__ gout,
                                                                  const T* __restrict__ x, T* __restrict__ gw,
                                                                  cot_agg_geom g, int Ho, int Wo, int64_t total) {
    typedef typename AccOf<T>::type A;
    const int64_t HoWo = (int64_t)Ho * Wo;
    const int64_t HW = (int64_t)g.H * g.W;
    const int taps = g.kh * g.kw;
    // one thread per (n, head, wc, tap, ho, wo): tap-parallel (the reference loops taps inside a thread)
    for (int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; index < total;
         index += (int64_t)gridDim.x * blockDim.x) {
        const int wo = (int)(index % Wo);
        const int ho = (int)((index / Wo) % Ho);
        const int64_t r = index / HoWo;  // ((n*heads+head)*wC + wc)*taps + tap
        const int tap = (int)(r % taps);
        const int64_t plane = r / taps;
        const int wc = (int)(plane % g.wC);
        const int64_t nh = plane / g.wC;
        const int n = (int)(nh / g.heads);
        const int kh = tap / g.kw, kw = tap % g.kw;
        const int h_in = -g.ph + ho * g.sh + kh * g.dh;
        const int w_in = -g.pw + wo * g.sw + kw * g.dw;
        A value = 0;
        if (h_in >= 0 && h_in < g.H && w_in >= 0 && w_in < g.W) {
            for (int cc = wc; cc < g.C; cc += g.wC) {
                value += ld(x + ((int64_t)n * g.C + cc) * HW + (int64_t)h_in * g.W + w_in) *
                         ld(gout + (nh * g.C + cc) * HoWo + (int64_t)ho * Wo + wo);
            }
        }
        st(gw + index, value);  // explicit 0 on padded taps (:97-105)
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 / s1 / d1 / p1 fast path
// ------------------------------------------------------------------------------------------------

// loads row (h+r) of one plane: columns w0-1 .. w0+P into dst[0..P+1]; zero outside the image.
template <typename T, int P, typename A>
__device__ __forceinline__ void load_row_halo(const T* __restrict__ plane, int hr, int H, int W, int w0,
                                              A (&dst)[P + 2]) {
    if (hr >= 0 && hr < H) {
        const T* p = plane + (int64_t)hr * W + w0;
        Vec<T, P> v = ldv<T, P>(p);
#pragma unroll
        for (int i = 0; i < P; ++i) dst[i + 1] = (A)v.v[i];
        dst[0] = (w0 > 0) ? (A)p[-1] : (A)0;
        dst[P + 1] = (w0 + P < W) ? (A)p[P] : (A)0;
    } else {
#pragma unroll
        for (int i = 0; i < P + 2; ++i) dst[i] = (A)0;
    }
}

template <typename T, int P>
__global__ __launch_bounds__(256) void agg_fwd_nchw_k3(const T* __restrict__ x, const T* __restrict__ w,
                                                      T* __restrict__ out, int heads, int C, int wC, int H, int W,
                                                      int64_t items) {
    typedef typename AccOf<T>::type A;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const int segs = W / P;
    const int seg = (int)(item % segs);
    const int h = (int)((item / segs) % H);
    const int64_t plane = item / ((int64_t)segs * H);  // (n*heads + head)*wC + wc
    const int wc = (int)(plane % wC);
    const int64_t nh = plane / wC;
    const int n = (int)(nh / heads);
    const int w0 = seg * P;
    const int64_t HW = (int64_t)H * W;

    // 9*P weights of this item, kept in registers across the channel loop
    A wr[9][P];
    {
        const T* wp = w + plane * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Vec<T, P> v = ldv<T, P>(wp + t * HW);
#pragma unroll
            for (int i = 0; i < P; ++i) wr[t][i] = (A)v.v[i];
        }
    }
    const int J = C / wC;
    for (int j = 0; j < J; ++j) {
        const int c = wc + j * wC;
        const T* xp = x + ((int64_t)n * C + c) * HW;
        A xr[3][P + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) load_row_halo<T, P, A>(xp, h - 1 + r, H, W, w0, xr[r]);
        Vec<T, P> o;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            A acc = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += wr[kh * 3 + kw][i] * xr[kh][i + kw];
            o.v[i] = (T)acc;
        }
        stv<T, P>(out + (nh * C + c) * HW + (int64_t)h * W + w0, o);
    }
}

// fused backward, heads == 1.  DO_GX / DO_GW select which gradients are produced.
template <typename T, int P, bool DO_GX, bool DO_GW>
__global__ __launch_bounds__(256) void agg_bwd_nchw_k3(const T* __restrict__ gout, const T* __restrict__ x,
                                                      const T* __restrict__ w, T* __restrict__ gx,
                                                      T* __restrict__ gw, int C, int wC, int H, int W,
                                                      int64_t items) {
    typedef typename AccOf<T>::type A;
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= items) return;
    const int segs = W / P;
    const int seg = (int)(item % segs);
    const int h = (int)((item / segs) % H);
    const int64_t plane = item / ((int64_t)segs * H);  // n*wC + wc
    const int wc = (int)(plane % wC);
    const int n = (int)(plane / wC);
    const int w0 = seg * P;
    const int64_t HW = (int64_t)H * W;

    // gX[h,w] = sum_{kh,kw} w_t[h+1-kh, w+1-kw] * gO[h+1-kh, w+1-kw],  t = kh*3+kw  (:59-75 with s=1,d=1,p=1)
    // ws[t][i] = w_t at the neighbour pixel that tap t of output pixel (h, w0+i) gathers from.
    A ws[9][P];
    if (DO_GX) {
        const T* wp = w + plane * 9 * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hr = h + 1 - kh;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kh * 3 + kw;
                if (hr >= 0 && hr < H) {
                    const T* p = wp + t * HW + (int64_t)hr * W + w0;
                    Vec<T, P> v = ldv<T, P>(p);
                    if (kw == 1) {
#pragma unroll
                        for (int i = 0; i < P; ++i) ws[t][i] = (A)v.v[i];
                    } else if (kw == 0) {  // columns w0+1 .. w0+P
#pragma unroll
                        for (int i = 0; i < P - 1; ++i) ws[t][i] = (A)v.v[i + 1];
                        ws[t][P - 1] = (w0 + P < W) ? (A)p[P] : (A)0;
                    } else {  // kw == 2: columns w0-1 .. w0+P-2
#pragma unroll
                        for (int i = 1; i < P; ++i) ws[t][i] = (A)v.v[i - 1];
                        ws[t][0] = (w0 > 0) ? (A)p[-1] : (A)0;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < P; ++i) ws[t][i] = (A)0;
                }
            }
        }
    }
    A gwacc[9][P];
    if (DO_GW) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < P; ++i) gwacc[t][i] = (A)0;
    }
    const int J = C / wC;
    for (int j = 0; j < J; ++j) {
        const int c = wc + j * wC;
        const int64_t pl = ((int64_t)n * C + c) * HW;
        A gr[3][P + 2];
        if (DO_GX) {
#pragma unroll
            for (int r = 0; r < 3; ++r) load_row_halo<T, P, A>(gout + pl, h - 1 + r, H, W, w0, gr[r]);
        } else {
            load_row_halo<T, P, A>(gout + pl, h, H, W, w0, gr[1]);
        }
        if (DO_GX) {
            Vec<T, P> o;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                A acc = 0;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc += ws[kh * 3 + kw][i] * gr[2 - kh][i + 2 - kw];
                o.v[i] = (T)acc;
            }
            stv<T, P>(gx + pl + (int64_t)h * W + w0, o);
        }
        if (DO_GW) {
            A xr[3][P + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) load_row_halo<T, P, A>(x + pl, h - 1 + r, H, W, w0, xr[r]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int i = 0; i < P; ++i) gwacc[kh * 3 + kw][i] += xr[kh][i + kw] * gr[1][i + 1];
        }
    }
    if (DO_GW) {
        T* gp = gw + plane * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Vec<T, P> o;
#pragma unroll
            for (int i = 0; i < P; ++i) o.v[i] = (T)gwacc[t][i];
            stv<T, P>(gp + t * HW, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
static thread_local const char* g_last_kernel = "";
const char* last_kernel_nchw() { return g_last_kernel; }

static inline int grid_for(int64_t total, int block, int64_t cap = (int64_t)1 << 20) {
    int64_t b = ceil_div64(total, block);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

static inline bool is_k3_fast(const cot_agg_geom& g) {
    return g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.ph == 1 && g.pw == 1 && g.dh == 1 && g.dw == 1;
}

// largest P in {8,4,2,1} with W % P == 0 and P*sizeof(T) <= 16 (and <= maxP)
template <typename T> static inline int pick_P(int W, int maxP) {
    int lim = (int)(16 / sizeof(T));
    if (lim > maxP) lim = maxP;
    for (int P = 8; P >= 1; P >>= 1)
        if (P <= lim && W % P == 0) return P;
    return 1;
}

template <typename T, int P>
static int launch_fwd_k3(const T* x, const T* w, T* out, const cot_agg_geom& g, hipStream_t s, const char* name) {
    const int64_t items = (int64_t)g.N * g.heads * g.wC * g.H * (g.W / P);
    hipLaunchKernelGGL((agg_fwd_nchw_k3<T, P>), dim3(grid_for(items, 256, INT32_MAX)), dim3(256), 0, s, x, w, out,
                       g.heads, g.C, g.wC, g.H, g.W, items);
    g_last_kernel = name;
    return check_launch(name);
}

template <typename T>
int agg_forward_nchw(const T* x, const T* w, T* out, const cot_agg_geom& g, int Ho, int Wo, hipStream_t s,
                     const char* tname) {
    (void)tname;
    if (is_k3_fast(g)) {
        switch (pick_P<T>(g.W, 8)) {
            case 8: return launch_fwd_k3<T, (sizeof(T) <= 2 ? 8 : 1)>(x, w, out, g, s, "agg_fwd_nchw_k3<P8>");
            case 4: return launch_fwd_k3<T, (sizeof(T) <= 4 ? 4 : 1)>(x, w, out, g, s, "agg_fwd_nchw_k3<P4>");
            case 2: return launch_fwd_k3<T, 2>(x, w, out, g, s, "agg_fwd_nchw_k3<P2>");
            default: return launch_fwd_k3<T, 1>(x, w, out, g, s, "agg_fwd_nchw_k3<P1>");
        }
    }
    const int64_t total = (int64_t)g.N * g.heads * g.C * Ho * Wo;
    hipLaunchKernelGGL((agg_fwd_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, x, w, out, g, Ho, Wo,
                       total);
    g_last_kernel = "agg_fwd_nchw_generic";
    return check_launch("agg_fwd_nchw_generic");
}

template <typename T, int P>
static int launch_bwd_k3(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g, hipStream_t s) {
    const int64_t items = (int64_t)g.N * g.wC * g.H * (g.W / P);
    const dim3 grid(grid_for(items, 256, INT32_MAX)), block(256);
    if (gx && gw) {
        hipLaunchKernelGGL((agg_bwd_nchw_k3<T, P, true, true>), grid, block, 0, s, gout, x, w, gx, gw, g.C, g.wC, g.H,
                           g.W, items);
        g_last_kernel = "agg_bwd_nchw_k3<gx,gw>";
    } else if (gx) {
        hipLaunchKernelGGL((agg_bwd_nchw_k3<T, P, true, false>), grid, block, 0, s, gout, x, w, gx, gw, g.C, g.wC, g.H,
                           g.W, items);
        g_last_kernel = "agg_bwd_nchw_k3<gx>";
    } else {
        hipLaunchKernelGGL((agg_bwd_nchw_k3<T, P, false, true>), grid, block, 0, s, gout, x, w, gx, gw, g.C, g.wC, g.H,
                           g.W, items);
        g_last_kernel = "agg_bwd_nchw_k3<gw>";
    }
    return check_launch("agg_bwd_nchw_k3");
}

template <typename T>
int agg_backward_nchw(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g, int Ho, int Wo,
                      hipStream_t s) {
    if (is_k3_fast(g) && g.heads == 1) {
        // P capped at 4: the fused kernel keeps 18*P fp32 of weights / weight-gradients in registers
        switch (pick_P<T>(g.W, 4)) {
            case 4: return launch_bwd_k3<T, (sizeof(T) <= 4 ? 4 : 1)>(gout, x, w, gx, gw, g, s);
            case 2: return launch_bwd_k3<T, 2>(gout, x, w, gx, gw, g, s);
            default: return launch_bwd_k3<T, 1>(gout, x, w, gx, gw, g, s);
        }
    }
    int rc = COT_OK;
    if (gx) {
        const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
        hipLaunchKernelGGL((agg_bwd_input_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, gout, w, gx, g,
                           Ho, Wo, total);
        g_last_kernel = "agg_bwd_input_nchw_generic";
        rc = check_launch("agg_bwd_input_nchw_generic");
        if (rc) return rc;
    }
    if (gw) {
        const int64_t total = (int64_t)g.N * g.heads * g.wC * g.kh * g.kw * Ho * Wo;
        hipLaunchKernelGGL((agg_bwd_weight_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, gout, x, gw,
                           g, Ho, Wo, total);
        g_last_kernel = "agg_bwd_weight_nchw_generic";
        rc = check_launch("agg_bwd_weight_nchw_generic");
    }
    return rc;
}

#define INSTANTIATE(T)                                                                                          \
    template int agg_forward_nchw<T>(const T*, const T*, T*, const cot_agg_geom&, int, int, hipStream_t,        \
                                     const char*);                                                              \
    template int agg_backward_nchw<T>(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, int, int,      \
                                      hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
INSTANTIATE(bf16_t)
INSTANTIATE(f16_t)

}  // namespace cot
