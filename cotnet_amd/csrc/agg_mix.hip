// agg_mix.hip -- aggregation_zeropad_mix (3x3 + 5x5 taps over the same input), NCHW, gfx950.
//
// Reference: cupy_layers/aggregation_zeropad_mix.py:20-74 (forward), :76-140 (input backward),
// :142-207 (weight backward).
//
// Two families.  (1) `aggmix_*_tile` (below the generic kernels): the geometry LocalConvolutionMix is built for -- stride 1,
// dilation 1, padding 1 / 2, so Ho = H, Wo = W -- as LDS-tiled plane kernels in the design of the 3x3 aggregation
// (agg_nchw.hip): a work item is (image, head, weight channel wc, row h, P consecutive pixels); its 9 + 25 weight vectors
// live in registers and are reused by the C / wC channels that share them; the channel planes are staged once per workgroup
// in LDS by asynchronous 16-byte LDS-DMA as contiguous slabs (channel wc + 1 follows wc in memory, so G consecutive weight
// channels of one channel group are ONE contiguous range), and the 3x3 and the 5x5 taps read the SAME staged 5-row window
// (pad 2) -- x / gout are read from HBM once, the weights once.  Sums run in the reference's order (kh outer, kw inner; 3x3
// before 5x5; channels ascending), padded taps contribute an exact 0.
// (2) the generic kernels: any stride / dilation / padding, one lane per output element, the reference's loop nest.
//
// Reference quirk kept on purpose: input backward sums head 0 only (:87-88).  `all_heads` != 0 gives
// the complete gradient instead.
#include <initializer_list>

#include "conv_lds_common.h"

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


// ---------------------------------------------------------------------------------------------------------------------
// (1) LDS-tiled plane kernels: stride 1, dilation 1, padding 1 (3x3) / 2 (5x5)
// ---------------------------------------------------------------------------------------------------------------------
// NS slabs of `slab_elems` contiguous elements (slab s starts at src + s * src_stride) back to back into LDS: 16-byte LDS-DMA
// chunks (no VGPR round trip, every byte of the workgroup's input in flight at once) when `wide` -- the slabs are 16-byte
// multiples on 16-byte boundaries -- element by element otherwise (odd planes).  Completed by the next __syncthreads().
template <typename T>
__device__ __forceinline__ void mix_stage(const T* __restrict__ src, int64_t src_stride, int NS, int slab_elems, T* lds,
                                          bool wide) {
    constexpr int VE = 16 / sizeof(T);
    if (wide) {
        const int cps = slab_elems / VE, total = NS * cps;  // 16-byte chunks per slab / in all
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        for (int c0 = wave * 64; c0 < total; c0 += nw * 64) {
            const int c = c0 + lane;
            if (c < total) {
                const int sl = c / cps, o = c - sl * cps;
                COT_ASYNC_COPY16(src + sl * src_stride + (int64_t)o * VE, lds + (int64_t)c0 * VE);
            }
        }
    } else {
        const int total = NS * slab_elems;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int sl = e / slab_elems, o = e - sl * slab_elems;
            lds[e] = src[sl * src_stride + o];
        }
    }
}

// P + 2R values of one staged row around columns w0 .. w0+P-1 as aligned P-wide LDS reads (W % P == 0: a vector lies inside
// the row or outside it as a whole); positions outside the row -- or a row outside the plane, `rv` false -- give 0 by
// selection.  Values the caller does not use cost nothing (their reads are dropped at compile time).
template <typename T, int P, int R>
__device__ __forceinline__ void mix_row(const T* row, bool rv, int W, int w0, typename AccOf<T>::type (&seg)[P + 2 * R]) {
    typedef typename AccOf<T>::type A;
    constexpr int NV = (R + P - 1) / P;
#pragma unroll
    for (int v = -NV; v <= NV; ++v) {
        const int ww = w0 + v * P;
        const bool ok = rv && ww >= 0 && ww < W;
        const Vec<T, P> t = ldv<T, P>(row + (ok ? ww : w0));
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int d = v * P + i;
            if (d >= -R && d < P + R) seg[d + R] = ok ? (A)t.v[i] : (A)0;
        }
    }
}
// the (2R+1) x (P+2R) window of a staged plane around the item (row h, columns w0 ..)
template <typename T, int P, int R>
__device__ __forceinline__ void mix_window(const T* plane, int H, int W, int h, int w0,
                                           typename AccOf<T>::type (&win)[2 * R + 1][P + 2 * R]) {
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
        const int hh = h + dy;
        const bool rv = hh >= 0 && hh < H;
        mix_row<T, P, R>(plane + (rv ? hh : h) * W, rv, W, w0, win[dy + R]);
    }
}

// unit u of a workgroup -> (weight channel inside the group, row, first column)
struct MixUnit { int gi, h, w0; };
__device__ __forceinline__ MixUnit mix_unit(int u, int H, int segs, int P) {
    MixUnit m;
    const int seg = u % segs, r = u / segs;
    m.h = r % H;
    m.gi = r / H;
    m.w0 = seg * P;
    return m;
}

struct MixArgs {
    int heads, C, wC, H, W, G, wide;
};

template <typename T, int P>
__device__ __forceinline__ void mix_load_weights(const T* __restrict__ w1, const T* __restrict__ w2, int64_t q, int64_t HW,
                                                 int64_t o, Vec<T, P> (&wa)[9], Vec<T, P> (&wb)[25]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wa[t] = ldv<T, P>(w1 + (q * 9 + t) * HW + o);
#pragma unroll
    for (int t = 0; t < 25; ++t) wb[t] = ldv<T, P>(w2 + (q * 25 + t) * HW + o);
}

// forward (mix.py:20-74): out[n][0][head][c] = sum_9 w1 * x, out[n][1][head][c] = sum_25 w2 * x for the J = C / wC channels
// c = wc + j * wC that share the weight planes of wc.  Workgroup = G consecutive weight channels of one (image, head):
// their J slabs of G channel planes in LDS (plane j * G + gi), the weights straight from HBM into registers -- issued right
// behind the slab copies, so everything the workgroup reads is in flight at once.
template <typename T, int P>
__global__ __launch_bounds__(256) void aggmix_fwd_tile(const T* __restrict__ x, const T* __restrict__ w1,
                                                      const T* __restrict__ w2, T* __restrict__ out, MixArgs a) {
    typedef typename AccOf<T>::type A;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    T* xs = reinterpret_cast<T*>(cot_smem);
    const int H = a.H, W = a.W, G = a.G, wC = a.wC, C = a.C;
    const int J = C / wC, gpw = wC / G, segs = W / P, units = G * H * segs;
    const int64_t HW = (int64_t)H * W;
    const int wc0 = (int)(blockIdx.x % (unsigned)gpw) * G;
    const int64_t nh = blockIdx.x / (unsigned)gpw;
    const int n = (int)(nh / a.heads), head = (int)(nh % a.heads);
    mix_stage<T>(x + ((int64_t)n * C + wc0) * HW, (int64_t)wC * HW, J, G * (int)HW, xs, a.wide);

    int u = threadIdx.x;
    Vec<T, P> wa[9], wb[25];
    MixUnit m = mix_unit(u < units ? u : 0, H, segs, P);
    if (u < units) mix_load_weights<T, P>(w1, w2, nh * wC + wc0 + m.gi, HW, (int64_t)m.h * W + m.w0, wa, wb);
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes every wave's chunks visible
    while (u < units) {
        const int64_t o = (int64_t)m.h * W + m.w0;
        T* o1 = out + ((((int64_t)n * 2 + 0) * a.heads + head) * C + wc0 + m.gi) * HW + o;
        T* o2 = out + ((((int64_t)n * 2 + 1) * a.heads + head) * C + wc0 + m.gi) * HW + o;
        for (int j = 0; j < J; ++j) {
            A win[5][P + 4];
            mix_window<T, P, 2>(xs + ((int64_t)j * G + m.gi) * HW, H, W, m.h, m.w0, win);
            Vec<T, P> r1, r2;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                A v1 = 0, v2 = 0;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) v1 += (A)wa[kh * 3 + kw].v[i] * win[kh + 1][kw + 1 + i];
#pragma unroll
                for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) v2 += (A)wb[kh * 5 + kw].v[i] * win[kh][kw + i];
                r1.v[i] = (T)v1;
                r2.v[i] = (T)v2;
            }
            stv<T, P>(o1 + (int64_t)j * wC * HW, r1);
            stv<T, P>(o2 + (int64_t)j * wC * HW, r2);
        }
        u += blockDim.x;
        if (u < units) {
            m = mix_unit(u, H, segs, P);
            mix_load_weights<T, P>(w1, w2, nh * wC + wc0 + m.gi, HW, (int64_t)m.h * W + m.w0, wa, wb);
        }
    }
}

// weight backward (mix.py:142-207): gw1[q][tap][h][w] = sum_j x[c_j][h - 1 + kh][w - 1 + kw] * gout[n][0][head][c_j][h][w]
// (gw2: 5x5, pad 2, gout[n][1]); the x planes staged in LDS, gout at the item's own pixels straight from HBM, the 34 x P
// sums in registers; channels ascending as in the reference's loop
template <typename T, int P>
__global__ __launch_bounds__(256) void aggmix_bwd_weight_tile(const T* __restrict__ gout, const T* __restrict__ x,
                                                             T* __restrict__ gw1, T* __restrict__ gw2, MixArgs a) {
    typedef typename AccOf<T>::type A;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    T* xs = reinterpret_cast<T*>(cot_smem);
    const int H = a.H, W = a.W, G = a.G, wC = a.wC, C = a.C;
    const int J = C / wC, gpw = wC / G, segs = W / P, units = G * H * segs;
    const int64_t HW = (int64_t)H * W;
    const int wc0 = (int)(blockIdx.x % (unsigned)gpw) * G;
    const int64_t nh = blockIdx.x / (unsigned)gpw;
    const int n = (int)(nh / a.heads), head = (int)(nh % a.heads);
    mix_stage<T>(x + ((int64_t)n * C + wc0) * HW, (int64_t)wC * HW, J, G * (int)HW, xs, a.wide);
    __syncthreads();
    for (int u = threadIdx.x; u < units; u += blockDim.x) {
        const MixUnit m = mix_unit(u, H, segs, P);
        const int64_t o = (int64_t)m.h * W + m.w0;
        const T* g1 = gout + ((((int64_t)n * 2 + 0) * a.heads + head) * C + wc0 + m.gi) * HW + o;
        const T* g2 = gout + ((((int64_t)n * 2 + 1) * a.heads + head) * C + wc0 + m.gi) * HW + o;
        A a1[9][P], a2[25][P];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < P; ++i) a1[t][i] = 0;
#pragma unroll
        for (int t = 0; t < 25; ++t)
#pragma unroll
            for (int i = 0; i < P; ++i) a2[t][i] = 0;
        for (int j = 0; j < J; ++j) {
            const Vec<T, P> v1 = ldv<T, P>(g1 + (int64_t)j * wC * HW), v2 = ldv<T, P>(g2 + (int64_t)j * wC * HW);
            A win[5][P + 4];
            mix_window<T, P, 2>(xs + ((int64_t)j * G + m.gi) * HW, H, W, m.h, m.w0, win);
#pragma unroll
            for (int i = 0; i < P; ++i) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) a1[kh * 3 + kw][i] += win[kh + 1][kw + 1 + i] * (A)v1.v[i];
#pragma unroll
                for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) a2[kh * 5 + kw][i] += win[kh][kw + i] * (A)v2.v[i];
            }
        }
        const int64_t q = nh * wC + wc0 + m.gi;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            Vec<T, P> r;
#pragma unroll
            for (int i = 0; i < P; ++i) r.v[i] = (T)a1[t][i];
            stv<T, P>(gw1 + (q * 9 + t) * HW + o, r);
        }
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            Vec<T, P> r;
#pragma unroll
            for (int i = 0; i < P; ++i) r.v[i] = (T)a2[t][i];
            stv<T, P>(gw2 + (q * 25 + t) * HW + o, r);
        }
    }
}

// input backward (mix.py:76-140), head 0 only as in the reference: gx[n][c][h][w] = sum_9 w1[.. at (h + 1 - kh, w + 1 - kw)]
// * gout[n][0][0][c][same pixel] + sum_25 w2[..] * gout[n][1][0][c][..], one running sum, 3x3 first.  The weights are
// needed AT THE OUTPUT PIXEL of each tap, i.e. shifted by (p - kh, p - kw) -- unaligned for a P-wide global access -- so
// the weight planes are staged in LDS as well (contiguous: G x 9 and G x 25 planes) and the shifted values come from the
// same aligned row reads as the windows.  LDS = [gout half 0: J slabs][half 1: J slabs][w1][w2].
template <typename T, int P>
__global__ __launch_bounds__(256) void aggmix_bwd_input_tile(const T* __restrict__ gout, const T* __restrict__ w1,
                                                            const T* __restrict__ w2, T* __restrict__ gx, MixArgs a) {
    typedef typename AccOf<T>::type A;
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    T* g1s = reinterpret_cast<T*>(cot_smem);
    const int H = a.H, W = a.W, G = a.G, wC = a.wC, C = a.C;
    const int J = C / wC, gpw = wC / G, segs = W / P, units = G * H * segs;
    const int64_t HW = (int64_t)H * W, S = (int64_t)G * HW;
    const int wc0 = (int)(blockIdx.x % (unsigned)gpw) * G;
    const int n = (int)(blockIdx.x / (unsigned)gpw);
    T* g2s = g1s + J * S;
    T* w1s = g1s + 2 * J * S;
    T* w2s = g1s + (2 * J + 9) * S;
    mix_stage<T>(gout + ((((int64_t)n * 2 + 0) * a.heads + 0) * C + wc0) * HW, (int64_t)wC * HW, J, (int)S, g1s, a.wide);
    mix_stage<T>(gout + ((((int64_t)n * 2 + 1) * a.heads + 0) * C + wc0) * HW, (int64_t)wC * HW, J, (int)S, g2s, a.wide);
    mix_stage<T>(w1 + (((int64_t)n * a.heads + 0) * wC + wc0) * 9 * HW, 0, 1, 9 * (int)S, w1s, a.wide);
    mix_stage<T>(w2 + (((int64_t)n * a.heads + 0) * wC + wc0) * 25 * HW, 0, 1, 25 * (int)S, w2s, a.wide);
    __syncthreads();
    for (int u = threadIdx.x; u < units; u += blockDim.x) {
        const MixUnit m = mix_unit(u, H, segs, P);
        A wa[9][P], wb[25][P];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = m.h + 1 - kh;
            const bool rv = hh >= 0 && hh < H;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                A seg[P + 2];
                mix_row<T, P, 1>(w1s + ((int64_t)m.gi * 9 + kh * 3 + kw) * HW + (rv ? hh : m.h) * W, rv, W, m.w0, seg);
#pragma unroll
                for (int i = 0; i < P; ++i) wa[kh * 3 + kw][i] = seg[i + 2 - kw];
            }
        }
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            const int hh = m.h + 2 - kh;
            const bool rv = hh >= 0 && hh < H;
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                A seg[P + 4];
                mix_row<T, P, 2>(w2s + ((int64_t)m.gi * 25 + kh * 5 + kw) * HW + (rv ? hh : m.h) * W, rv, W, m.w0, seg);
#pragma unroll
                for (int i = 0; i < P; ++i) wb[kh * 5 + kw][i] = seg[i + 4 - kw];
            }
        }
        T* gp = gx + ((int64_t)n * C + wc0 + m.gi) * HW + (int64_t)m.h * W + m.w0;
        for (int j = 0; j < J; ++j) {
            A win1[3][P + 2], win2[5][P + 4];
            mix_window<T, P, 1>(g1s + ((int64_t)j * G + m.gi) * HW, H, W, m.h, m.w0, win1);
            mix_window<T, P, 2>(g2s + ((int64_t)j * G + m.gi) * HW, H, W, m.h, m.w0, win2);
            Vec<T, P> r;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                A v = 0;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) v += wa[kh * 3 + kw][i] * win1[2 - kh][i + 2 - kw];
#pragma unroll
                for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) v += wb[kh * 5 + kw][i] * win2[4 - kh][i + 4 - kw];
                r.v[i] = (T)v;
            }
            stv<T, P>(gp + (int64_t)j * wC * HW, r);
        }
    }
}

// cot_set_tuning keys 51 / 52 / 53: 51 = 1 keeps every call on the generic kernels (A/B); 52 = lanes a workgroup aims for;
// 53 = pixels per lane (0 = widest the row width and the storage type allow)
int g_mix_tune[3] = {0, 256, 0};

struct MixPlan {
    bool ok;
    int P, block, grid;
    size_t lds;
    MixArgs a;
};
// `planes` = staged planes per weight channel (forward / weight backward: J; input backward: 2 J + 34);
// `nh` = (image, head) pairs the launch covers; `ptrs` = the tensors that are staged
template <typename T>
static MixPlan mix_plan(const cot_agg_geom& g, int p2h, int p2w, int64_t planes, int64_t nh, int prefer_p,
                        std::initializer_list<const void*> ptrs) {
    MixPlan p = {};
    if (g_mix_tune[0] == 1) return p;
    if (g.sh != 1 || g.sw != 1 || g.dh != 1 || g.dw != 1 || g.ph != 1 || g.pw != 1 || p2h != 2 || p2w != 2) return p;
    if (g.wC <= 0 || g.C % g.wC || g.heads <= 0) return p;
    const int64_t HW = (int64_t)g.H * g.W;
    const int64_t item_bytes = planes * HW * (int64_t)sizeof(T);
    const int64_t cap = 64 * 1024;  // (several weight channels per workgroup only inside the default 64 KB)
    if (item_bytes > 160 * 1024 || HW >= ((int64_t)1 << 22) || nh * g.wC >= ((int64_t)1 << 30)) return p;
    const int maxP = 16 / (int)sizeof(T);
    int P = (g.W % 4 == 0 && maxP >= 4) ? 4 : ((g.W % 2 == 0 && maxP >= 2) ? 2 : 1);
    // Narrower items = more lanes per plane and half the registers per lane (127 instead of 237: four waves per SIMD instead
    // of two).  Measured at (B = 64, C = 256, 20 x 20, wC = 32), profiles/r06_aggmix_variants.log: forward 26.5 vs 29.2 us bf16,
    // 35.9 vs 38.2 fp32; weight backward 28.0 vs 31.5 bf16 but 33.1 vs 31.5 fp32; input backward indifferent.
    if (P > prefer_p) P = prefer_p;
    if (g_mix_tune[2] > 0 && g_mix_tune[2] <= maxP && g.W % g_mix_tune[2] == 0) P = g_mix_tune[2];
    p.P = P;
    const int64_t per_item = (int64_t)g.H * (g.W / P);  // lanes one weight channel's plane takes
    int G = 1;
    for (int c = 1; c <= g.wC; ++c)
        if (g.wC % c == 0 && c * per_item <= (int64_t)g_mix_tune[1] && c * item_bytes <= cap) G = c;
    const int64_t lanes = G * per_item;
    p.block = (int)(lanes >= 256 ? 256 : ((lanes + 63) / 64) * 64);
    p.lds = (size_t)((G * item_bytes + 15) / 16 * 16);
    p.grid = (int)(nh * (g.wC / G));
    bool wide = (HW * (int64_t)sizeof(T)) % 16 == 0;
    for (const void* q : ptrs) wide = wide && ((uintptr_t)q % 16) == 0;
    p.a = MixArgs{g.heads, g.C, g.wC, g.H, g.W, G, wide ? 1 : 0};
    p.ok = true;
    return p;
}

static inline int grid1d(int64_t total) {
    int64_t b = ceil_div64(total, 256);
    if (b > ((int64_t)1 << 20)) b = (int64_t)1 << 20;
    if (b < 1) b = 1;
    return (int)b;
}

#define MIX_BY_P(PVAL, CALL)                 \
    switch (PVAL) {                          \
        case 4: { constexpr int P = 4; CALL; } break; \
        case 2: { constexpr int P = 2; CALL; } break; \
        default: { constexpr int P = 1; CALL; } break; \
    }

// (P * sizeof(T) <= 16, mix_plan: only the vector widths a storage type can take are instantiated)
static const char* g_mix_kernel = "";
const char* last_kernel_mix() { return g_mix_kernel; }

template <typename T, int P>
static int mix_launch_fwd(const MixPlan& p, const T* x, const T* w1, const T* w2, T* out, hipStream_t s) {
    if constexpr (P * sizeof(T) <= 16) {
        static std::atomic<uint32_t> raised{0};
        if (p.lds > 64 * 1024 && !raise_dynamic_lds_once(raised, (const void*)aggmix_fwd_tile<T, P>)) return -1;
        COT_LAUNCH((aggmix_fwd_tile<T, P>), dim3(p.grid), dim3(p.block), p.lds, s, x, w1, w2, out, p.a);
        g_mix_kernel = "aggmix_fwd_tile";
        return check_launch("aggmix_fwd_tile");
    }
    return -1;
}
template <typename T, int P>
static int mix_launch_bwi(const MixPlan& p, const T* gout, const T* w1, const T* w2, T* gx, hipStream_t s) {
    if constexpr (P * sizeof(T) <= 16) {
        static std::atomic<uint32_t> raised{0};
        if (p.lds > 64 * 1024 && !raise_dynamic_lds_once(raised, (const void*)aggmix_bwd_input_tile<T, P>)) return -1;
        COT_LAUNCH((aggmix_bwd_input_tile<T, P>), dim3(p.grid), dim3(p.block), p.lds, s, gout, w1, w2, gx, p.a);
        g_mix_kernel = "aggmix_bwd_input_tile";
        return check_launch("aggmix_bwd_input_tile");
    }
    return -1;
}
template <typename T, int P>
static int mix_launch_bww(const MixPlan& p, const T* gout, const T* x, T* gw1, T* gw2, hipStream_t s) {
    if constexpr (P * sizeof(T) <= 16) {
        static std::atomic<uint32_t> raised{0};
        if (p.lds > 64 * 1024 && !raise_dynamic_lds_once(raised, (const void*)aggmix_bwd_weight_tile<T, P>)) return -1;
        COT_LAUNCH((aggmix_bwd_weight_tile<T, P>), dim3(p.grid), dim3(p.block), p.lds, s, gout, x, gw1, gw2, p.a);
        g_mix_kernel = "aggmix_bwd_weight_tile";
        return check_launch("aggmix_bwd_weight_tile");
    }
    return -1;
}

template <typename T>
int aggmix_forward(const T* x, const T* w1, const T* w2, T* out, const cot_agg_geom& g, int p2h, int p2w, int Ho,
                   int Wo, hipStream_t s) {
    const MixPlan p = mix_plan<T>(g, p2h, p2w, g.C / (g.wC > 0 ? g.wC : 1), (int64_t)g.N * g.heads, 2, {x});
    if (p.ok) {
        int rc = COT_OK;
        MIX_BY_P(p.P, (rc = mix_launch_fwd<T, P>(p, x, w1, w2, out, s)));
        if (rc >= 0) return rc;
    }
    const int64_t total = (int64_t)g.N * 2 * g.heads * g.C * Ho * Wo;
    COT_LAUNCH((aggmix_fwd<T>), dim3(grid1d(total)), dim3(256), 0, s, x, w1, w2, out, g, p2h, p2w, Ho, Wo,
                       total);
    g_mix_kernel = "aggmix_fwd";
    return check_launch("aggmix_fwd");
}
template <typename T>
int aggmix_backward_input(const T* gout, const T* w1, const T* w2, T* gx, const cot_agg_geom& g, int p2h, int p2w,
                          int all_heads, int Ho, int Wo, hipStream_t s) {
    const int nheads_used = all_heads ? g.heads : 1;
    const MixPlan p = mix_plan<T>(g, p2h, p2w, 2 * (g.C / (g.wC > 0 ? g.wC : 1)) + 34, (int64_t)g.N, 4, {gout, w1, w2});
    if (p.ok && nheads_used == 1) {  // (the reference's head-0 sum; the complete gradient over several heads: generic kernel)
        int rc = COT_OK;
        MIX_BY_P(p.P, (rc = mix_launch_bwi<T, P>(p, gout, w1, w2, gx, s)));
        if (rc >= 0) return rc;
    }
    const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
    COT_LAUNCH((aggmix_bwd_input<T>), dim3(grid1d(total)), dim3(256), 0, s, gout, w1, w2, gx, g, p2h, p2w,
                       nheads_used, Ho, Wo, total);
    g_mix_kernel = "aggmix_bwd_input";
    return check_launch("aggmix_bwd_input");
}
template <typename T>
int aggmix_backward_weight(const T* gout, const T* x, T* gw1, T* gw2, const cot_agg_geom& g, int p2h, int p2w,
                           int Ho, int Wo, hipStream_t s) {
    const MixPlan p = mix_plan<T>(g, p2h, p2w, g.C / (g.wC > 0 ? g.wC : 1), (int64_t)g.N * g.heads, sizeof(T) <= 2 ? 2 : 4, {x});
    if (p.ok) {
        int rc = COT_OK;
        MIX_BY_P(p.P, (rc = mix_launch_bww<T, P>(p, gout, x, gw1, gw2, s)));
        if (rc >= 0) return rc;
    }
    const int64_t total = (int64_t)g.N * g.heads * g.wC * 34 * Ho * Wo;
    COT_LAUNCH((aggmix_bwd_weight<T>), dim3(grid1d(total)), dim3(256), 0, s, gout, x, gw1, gw2, g, p2h, p2w,
                       Ho, Wo, total);
    g_mix_kernel = "aggmix_bwd_weight";
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
