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
// 3x3 fast path, version 2: wave-aligned rows + cross-lane halos + prefetched channel loop
// ------------------------------------------------------------------------------------------------
// v1 above is latency-bound on MI355X (bf16 stage-1 forward: 21 % of the HBM roofline, the same time as
// fp32): 9 dependent load rounds per lane, 6 of the 9 loads per channel being 2-byte halo scalars.  v2:
//   * a wave owns L = floor(64/segs)*segs consecutive work items (segs = W/P row segments), i.e. WHOLE rows, so the
//     left/right halo of a lane is always the edge element of lane-1 / lane+1 (or zero at a row end).  Halos move
//     with one DPP wave-shift (or ds_bpermute) per row instead of two scalar loads: 3 loads per channel, all 16 B.
//   * the channel loop prefetches channel j+1's raw row vectors before it computes channel j, doubling the
//     bytes each lane keeps in flight; storage-type rows stay packed until use to keep VGPRs (occupancy) low.
// XCHG selects the lane-exchange primitive: 0 = v_mov_b32_dpp wave_shr/wave_shl, 1 = ds_bpermute (__shfl);
// the library probes the DPP direction once on the device and falls back to 1 if it is not what we expect.

// (bound_ctrl: a lane without a source -- lane 0 of wave_shr, lane 63 of wave_shl -- reads 0; with it and full masks the
// "old" operand is dead, so no register has to be zeroed per exchange and the move can fold into its consumer)
template <int XCHG> __device__ __forceinline__ float lane_prev(float v) {
    if (XCHG == 0) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
    return __shfl_up(v, 1);
}
template <int XCHG> __device__ __forceinline__ float lane_next(float v) {
    if (XCHG == 0) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
    return __shfl_down(v, 1);
}
template <int XCHG> __device__ __forceinline__ double lane_prev(double v) { return __shfl_up(v, 1); }
template <int XCHG> __device__ __forceinline__ double lane_next(double v) { return __shfl_down(v, 1); }

__global__ void dpp_probe_kernel(int* out) {
    const int l = threadIdx.x;
    // exactly the exchanges lane_prev / lane_next use (old = -1 here so that a lane that is NOT zero-filled shows)
    out[l] = __builtin_amdgcn_update_dpp(-1, l + 1, 0x138, 0xf, 0xf, true);       // expect l     (lane 0: 0)
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, l + 1, 0x130, 0xf, 0xf, true);  // expect l + 2 (lane 63: 0)
}

// raw (storage-type) row vector of plane row hr, zero when the row is outside the image
// (the load is unconditional, from the nearest row inside the plane, and cleared by selection afterwards: a predicated
// load is an exec-masked region of its own, and a lane's nine of them were issued and waited for one at a time)
template <typename T, int P>
__device__ __forceinline__ Vec<T, P> load_row_raw(const T* __restrict__ plane, int hr, int H, int W, int w0) {
    const int hc = hr < 0 ? 0 : (hr >= H ? H - 1 : hr);
    Vec<T, P> v = ldv<T, P>(plane + (int64_t)hc * W + w0);
    if (hc != hr) {
#pragma unroll
        for (int i = 0; i < P; ++i) v.v[i] = (T)0;
    }
    return v;
}

// dst[0..P+1] = columns w0-1 .. w0+P of the row: converted vector + halos taken from the neighbouring lanes
template <typename T, int P, int XCHG, typename A>
__device__ __forceinline__ void expand_row(const Vec<T, P>& raw, bool has_left, bool has_right, A (&dst)[P + 2]) {
#pragma unroll
    for (int i = 0; i < P; ++i) dst[i + 1] = (A)raw.v[i];
    const A l = lane_prev<XCHG>(dst[P]);
    const A r = lane_next<XCHG>(dst[1]);
    dst[0] = has_left ? l : (A)0;
    dst[P + 1] = has_right ? r : (A)0;
}

// the same without the row-end test: the halos are whatever the neighbouring lanes hold (the previous / next row's edge
// pixels at a row's end).  For accumulators that are cleared at row ends afterwards (weight gradient, below).
template <typename T, int P, int XCHG, typename A>
__device__ __forceinline__ void expand_row_unmasked(const Vec<T, P>& raw, A (&dst)[P + 2]) {
#pragma unroll
    for (int i = 0; i < P; ++i) dst[i + 1] = (A)raw.v[i];
    dst[0] = lane_prev<XCHG>(dst[P]);
    dst[P + 1] = lane_next<XCHG>(dst[1]);
}

struct ItemV2 {
    bool valid;
    int seg, h, wc;
    int64_t plane, nh;
};
// item = wave*L + lane for lane < L; lanes >= L and items >= total are clamped to a valid item and do not store
__device__ __forceinline__ ItemV2 decode_item_v2(int segs, int L, int H, int wC, int64_t items) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int64_t item = wave * L + lane;
    ItemV2 it;
    it.valid = lane < L && item < items;
    if (!it.valid) item = items - 1;
    it.seg = (int)(item % segs);
    it.h = (int)((item / segs) % H);
    it.plane = item / ((int64_t)segs * H);
    it.wc = (int)(it.plane % wC);
    it.nh = it.plane / wC;
    return it;
}

template <typename T, int P, int XCHG>
__global__ __launch_bounds__(256) void agg_fwd_nchw_k3_v2(const T* __restrict__ x, const T* __restrict__ w,
                                                         T* __restrict__ out, int heads, int C, int wC, int H, int W,
                                                         int L, int64_t items) {
    typedef typename AccOf<T>::type A;
    const int segs = W / P;
    const ItemV2 it = decode_item_v2(segs, L, H, wC, items);
    const int n = (int)(it.nh / heads);
    const int w0 = it.seg * P, h = it.h;
    const int64_t HW = (int64_t)H * W;
    const bool has_left = it.seg > 0, has_right = it.seg < segs - 1;

    // the item's 9*P weights stay packed in the storage type (bf16: 36 VGPRs at P=8) and are converted at use
    Vec<T, P> wr[9];
    {
        const T* wp = w + it.plane * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t] = ldv<T, P>(wp + t * HW);
    }
    const int J = C / wC;
    const int64_t cstride = (int64_t)wC * HW;
    const T* xp = x + ((int64_t)n * C + it.wc) * HW;
    T* op = out + (it.nh * C + it.wc) * HW + (int64_t)h * W + w0;
    Vec<T, P> cur[3], nxt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) cur[r] = load_row_raw<T, P>(xp, h - 1 + r, H, W, w0);
    for (int j = 0; j < J; ++j) {
        if (j + 1 < J) {
#pragma unroll
            for (int r = 0; r < 3; ++r) nxt[r] = load_row_raw<T, P>(xp + (j + 1) * cstride, h - 1 + r, H, W, w0);
        }
        A xr[3][P + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r) expand_row<T, P, XCHG, A>(cur[r], has_left, has_right, xr[r]);
        Vec<T, P> o;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            A acc = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += (A)wr[kh * 3 + kw].v[i] * xr[kh][i + kw];
            o.v[i] = (T)acc;
        }
        if (it.valid) stv<T, P>(op + j * cstride, o);
#pragma unroll
        for (int r = 0; r < 3; ++r) cur[r] = nxt[r];
    }
}

// fused backward v2 (heads == 1): same mapping; shifted weights via lane exchange; gO / x rows prefetched.
template <typename T, int P, int XCHG, bool DO_GX, bool DO_GW, int MINW>
__global__ __launch_bounds__(256, MINW) void agg_bwd_nchw_k3_v2(const T* __restrict__ gout, const T* __restrict__ x,
                                                         const T* __restrict__ w, T* __restrict__ gx,
                                                         T* __restrict__ gw, int C, int wC, int H, int W, int L,
                                                         int64_t items) {
    typedef typename AccOf<T>::type A;
    const int segs = W / P;
    const ItemV2 it = decode_item_v2(segs, L, H, wC, items);
    const int n = (int)it.nh;  // heads == 1
    const int w0 = it.seg * P, h = it.h;
    const int64_t HW = (int64_t)H * W;
    const bool has_left = it.seg > 0, has_right = it.seg < segs - 1;

    // ws[t][i] = w_t[h+1-kh, w0+i+1-kw]  (0 outside): the weight each gathered neighbour was multiplied with
    A ws[9][P];
    if (DO_GX) {
        const T* wp = w + it.plane * 9 * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kh * 3 + kw;
                A row[P + 2];
                expand_row<T, P, XCHG, A>(load_row_raw<T, P>(wp + t * HW, h + 1 - kh, H, W, w0), has_left, has_right,
                                          row);
#pragma unroll
                for (int i = 0; i < P; ++i) ws[t][i] = row[i + 2 - kw];
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
    const int64_t cstride = (int64_t)wC * HW;
    const int64_t pl0 = ((int64_t)n * C + it.wc) * HW;
    Vec<T, P> gcur[3], gnxt[3], xcur[3], xnxt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (DO_GX || r == 1) gcur[r] = load_row_raw<T, P>(gout + pl0, h - 1 + r, H, W, w0);
        if (DO_GW) xcur[r] = load_row_raw<T, P>(x + pl0, h - 1 + r, H, W, w0);
    }
    for (int j = 0; j < J; ++j) {
        const int64_t pl = pl0 + j * cstride;
        if (j + 1 < J) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (DO_GX || r == 1) gnxt[r] = load_row_raw<T, P>(gout + pl + cstride, h - 1 + r, H, W, w0);
                if (DO_GW) xnxt[r] = load_row_raw<T, P>(x + pl + cstride, h - 1 + r, H, W, w0);
            }
        }
        A gr[3][P + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (DO_GX || r == 1) expand_row<T, P, XCHG, A>(gcur[r], has_left, has_right, gr[r]);
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
            if (it.valid) stv<T, P>(gx + pl + (int64_t)h * W + w0, o);
        }
        if (DO_GW) {
            A xr[3][P + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) expand_row<T, P, XCHG, A>(xcur[r], has_left, has_right, xr[r]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int i = 0; i < P; ++i) gwacc[kh * 3 + kw][i] += xr[kh][i + kw] * gr[1][i + 1];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            gcur[r] = gnxt[r];
            xcur[r] = xnxt[r];
        }
    }
    if (DO_GW && it.valid) {
        T* gp = gw + it.plane * 9 * HW + (int64_t)h * W + w0;
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
// 3x3 fast path, version 3: LDS-staged channel slabs (asynchronous global->LDS DMA)
// ------------------------------------------------------------------------------------------------
// v2 still reads every x row three times through L1/L2 (rows h-1, h, h+1 are fetched by three lanes) and keeps only
// two channels' rows in flight per lane.  v3 stages, per 256-lane workgroup, the x rows of ALL channels that share
// the workgroup's weights in LDS:
//   * per image-head the (wc, h) rows are flattened to rho = wc*H + h; a workgroup owns TR = 4*R consecutive rho
//     (R = floor(64/segs) whole rows per wave, as in v2).  For channel group j (c = wc + j*wC) those rows plus one
//     halo row above and below are ONE contiguous range of x, because channel wc+1 follows channel wc in memory;
//   * the J ranges ("slabs") are copied with global_load_lds_dwordx4: 16 B per lane, no VGPR round trip, every
//     byte of the tile in flight at once (34 KB per workgroup for bf16 stage 1, x4-5 workgroups per CU);
//   * after one barrier each lane reads its three P-wide row vectors per channel with aligned ds_read_b128 (lanes
//     read consecutive 16 B -> conflict-free), takes the two halo columns from its neighbour lanes (DPP) and
//     streams the P-wide results to HBM.  x is read from HBM once (+ (TR+2)/TR halo rows), the weights once.
// Rows outside the image are never dereferenced as data: a lane substitutes zeros when h-1 < 0 or h+1 >= H, so the
// slab may contain the neighbouring plane's rows (or clamped garbage at the tensor ends) harmlessly.

// XCD-aware work-item order (guide T1): the dispatcher places block b on XCD b % 8, each XCD has its own L2.  Tiles that
// are adjacent in memory share halo rows, so when `xcd_remap` is set a block takes the tile whose index keeps
// consecutive tiles on the SAME XCD: logical = (b % 8) * (nblk / 8) + b / 8 (bijective when nblk % 8 == 0).
// Placement only affects speed (which L2 serves the halo re-reads), never results.
__device__ __forceinline__ unsigned logical_block(int xcd_remap) {
    const unsigned b = blockIdx.x, nblk = gridDim.x;
    if (!xcd_remap || (nblk & 7u) != 0) return b;
    return (b & 7u) * (nblk >> 3) + (b >> 3);
}

// The read is unconditional (a predicated LDS read is its own exec-masked region, and the six reads of a channel were
// issued and waited for one by one); rows outside the image are cleared afterwards by selection.  Every index a lane can
// form lies inside its slab -- the halo rows are staged too, whatever they hold -- except row -1 of the tensor's first
// tile, which is clamped to the slab's start (and cleared: that row is outside the image).
template <typename T, int P>
__device__ __forceinline__ Vec<T, P> lds_row(const T* __restrict__ slab, int64_t idx, bool row_ok) {
    Vec<T, P> v = *reinterpret_cast<const Vec<T, P>*>(slab + (idx > 0 ? idx : 0));
    if (!row_ok) {
#pragma unroll
        for (int i = 0; i < P; ++i) v.v[i] = (T)0;
    }
    return v;
}

// cooperative async copy of `nslab` slabs of `sle` elements: slab s starts at element src0 + s*sstride of `base`
template <typename T>
__device__ __forceinline__ void stage_slabs(const T* __restrict__ base, int64_t src0, int64_t sstride, int nslab,
                                            int sle, int64_t total_elems, T* __restrict__ lds) {
    constexpr int VE = 16 / sizeof(T);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int chunks = sle / VE, total = nslab * chunks, nthr = blockDim.x;
    for (int q0 = 0; q0 < total; q0 += nthr) {
        const int q = q0 + tid;
        if (q < total) {  // lanes past the end are masked off (EXEC): they transfer nothing
            const int sidx = q / chunks, ch = q - sidx * chunks;
            int64_t e = src0 + (int64_t)sidx * sstride + (int64_t)ch * VE;
            if (e + VE > total_elems) e = total_elems - VE;  // beyond the tensor: in-bounds bytes, never used as data
            COT_ASYNC_COPY16(base + e, lds + (int64_t)(q0 + wave * 64) * VE);
        }
    }
}

// SM = 1: window-softmax producer fused in front of the aggregation (LR-Net's
// F.softmax(w, dim=3) + LocalConvolution, models/lr_net.py:94-96): `w` holds LOGITS, the lane normalises its own
// 9 taps per pixel in registers (the taps of a pixel all live in one lane, so no cross-lane reduction is needed),
// writes the probabilities to `probs` (what backward needs) and aggregates with them.
// SM = 2: GroupNorm-9 applied in the prologue (CotLayer's own normalisation of its logits, models/cotnet.py:55-56,:84-85): `w`
// holds the RAW logits of embed[3], `gn` the statistics (from that convolution's epilogue, cot_conv1x1_forward_gn9 +
// cot_gn9_stats_finalize) and the affine parameters; the lane normalises its own 9 x P values exactly as csrc/group_norm9.hip
// does (ga = gamma * rstd, be = beta - mean * ga, x * ga + be, rounded once to the storage type) -- the normalised tensor is
// never written.  heads == 1.
template <typename T> struct Gn9Args {
    const float* mean;   // [N * wC]  (plane = n * wC + wc = image * groups + group)
    const float* rstd;
    const T* gamma;      // [groups_per_image * 9]
    const T* beta;
    int gimg;            // GroupNorm groups per image (= wC, or 2 wC for CoXtLayer's group -> batch fold)
};

// ST = 1 (round 6; VERDICT r5 next #2b): the BatchNorm that follows the aggregation (models/cotnet.py:88-89) gets its statistics out of
// this kernel's epilogue -- per output ROW (n, c, h) the sum and the sum of squares of the values AS STORED (rounded to T) go to
// rowstats[((n*C + c)*H + h)*2 + {0, 1}].  A lane's P outputs of channel group j are summed into an LDS scratch slot behind the slabs;
// after the channel loop one thread per (row, j) adds the row's `segs` slots in lane order: fixed order, deterministic.  8 bytes per row
// of W outputs written and read once by bn_rowstats_finalize instead of a statistics pass that reads the whole tensor.  heads == 1.
template <typename T, int P, int XCHG, int SM, int ST = 0>
__global__ __launch_bounds__(512) void agg_fwd_nchw_k3_lds(const T* __restrict__ x, const T* __restrict__ w,
                                                          T* __restrict__ out, int heads, int C, int wC, int H, int W,
                                                          int R, int tiles_per_nh, int sle, int64_t x_elems,
                                                          T* __restrict__ probs, int xcd_remap, Gn9Args<T> gn,
                                                          float* __restrict__ rowstats) {
    typedef typename AccOf<T>::type A;
    constexpr int VE = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    T* slab = reinterpret_cast<T*>(cot_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int segs = W / P, J = C / wC, TR = (blockDim.x >> 6) * R, rows_nh = wC * H;
    const int64_t HW = (int64_t)H * W;
    const unsigned lb = logical_block(xcd_remap);
    const int tile = lb % tiles_per_nh;
    const int64_t nh = lb / tiles_per_nh;
    const int n = (int)(nh / heads);
    const int rho0 = tile * TR;
    int64_t gs = (int64_t)(rho0 - 1) * W;  // first slab element (row rho0-1), rounded down to a 16-byte chunk
    if (gs < 0) gs = 0;
    gs &= ~(int64_t)(VE - 1);
    stage_slabs<T>(x, (int64_t)n * C * HW + gs, (int64_t)wC * HW, J, sle, x_elems, slab);

    // this lane's item: row r of the wave, segment seg of the row
    int r = lane / segs, seg = lane - r * segs;
    int rho = rho0 + wave * R + r;
    const bool valid = r < R && rho < rows_nh;
    if (!valid) {  // park on the last valid row's last segment: keeps neighbours' halo logic intact, never stores
        rho = rows_nh - 1;
        seg = segs - 1;
    }
    const int wc = rho / H, h = rho - wc * H, w0 = seg * P;
    const bool has_left = seg > 0, has_right = seg < segs - 1;

    Vec<T, P> wr[9];
    {
        const T* wp = w + (nh * wC + wc) * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t] = ldv<T, P>(wp + t * HW);
        if (SM == 2) {
            const int64_t pl = nh * wC + wc;
            const int gq = (int)((unsigned)pl % (unsigned)gn.gimg);  // (planes < 2^31: launch_gn9_fwd)
            const A mu = (A)gn.mean[pl], rs = (A)gn.rstd[pl];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const A ga = (A)gn.gamma[gq * 9 + t] * rs, be = (A)gn.beta[gq * 9 + t] - mu * ga;
#pragma unroll
                for (int i = 0; i < P; ++i) wr[t].v[i] = (T)((A)wr[t].v[i] * ga + be);
            }
        }
        if (SM == 1) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                A m = (A)wr[0].v[i];
#pragma unroll
                for (int t = 1; t < 9; ++t) m = (A)wr[t].v[i] > m ? (A)wr[t].v[i] : m;
                A e[9], sum = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    e[t] = (A)expf((float)((A)wr[t].v[i] - m));
                    sum += e[t];
                }
                const A inv = (A)1 / sum;
#pragma unroll
                for (int t = 0; t < 9; ++t) wr[t].v[i] = (T)(e[t] * inv);  // rounded once: forward and backward agree
            }
            if (valid && probs) {
                T* pp = probs + (nh * wC + wc) * 9 * HW + (int64_t)h * W + w0;
#pragma unroll
                for (int t = 0; t < 9; ++t) stv<T, P>(pp + t * HW, wr[t]);
            }
        }
    }
    __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and makes every wave's slab chunks visible

    const int64_t lidx = (int64_t)rho * W + w0 - gs;  // centre-row vector of this item inside a slab
    T* op = out + nh * C * HW + (int64_t)rho * W + w0;
    const int64_t cstride = (int64_t)wC * HW;
    for (int j = 0; j < J; ++j) {
        const T* sj = slab + (int64_t)j * sle;
        A xr[3][P + 2];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int hr = h - 1 + rr;
            expand_row<T, P, XCHG, A>(lds_row<T, P>(sj, lidx + (int64_t)(rr - 1) * W, hr >= 0 && hr < H), has_left,
                                      has_right, xr[rr]);
        }
        Vec<T, P> o;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            A acc = 0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += (A)wr[kh * 3 + kw].v[i] * xr[kh][i + kw];
            o.v[i] = (T)acc;
        }
        if (valid) stv<T, P>(op + j * cstride, o);
        if (ST) {  // this lane's share of row (rho, j): sum and sum of squares of the rounded outputs
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float v = (float)o.v[i];
                a1 += v;
                a2 += v * v;
            }
            float* red = reinterpret_cast<float*>(cot_smem + (((size_t)J * sle * sizeof(T) + 15) & ~(size_t)15));
            red[(2 * j) * blockDim.x + threadIdx.x] = valid ? a1 : 0.f;
            red[(2 * j + 1) * blockDim.x + threadIdx.x] = valid ? a2 : 0.f;
        }
    }
    if (ST) {
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(cot_smem + (((size_t)J * sle * sizeof(T) + 15) & ~(size_t)15));
        const int NT = blockDim.x;
        for (int idx = threadIdx.x; idx < TR * J; idx += NT) {
            const int jj = idx / TR, rl = idx - jj * TR;
            const int rr = rho0 + rl;
            if (rr >= rows_nh) continue;
            const int base = (rl / R) * 64 + (rl % R) * segs;  // first lane of the row inside the workgroup
            float t1 = 0.f, t2 = 0.f;
            for (int sg = 0; sg < segs; ++sg) {
                t1 += red[(2 * jj) * NT + base + sg];
                t2 += red[(2 * jj + 1) * NT + base + sg];
            }
            const int wcr = rr / H, hr = rr - wcr * H;
            float* rp = rowstats + ((nh * C + wcr + (int64_t)jj * wC) * H + hr) * 2;
            rp[0] = t1;
            rp[1] = t2;
        }
    }
}

// fused backward v3 (heads == 1): gO and x slabs staged JP channels at a time (JP*2 slabs resident)
// SM = 1: `w` holds the saved probabilities and `gw` receives the gradient w.r.t. the LOGITS:
// d_logit_t = p_t * (g_t - sum_u p_u g_u) per pixel, again entirely in-lane.
template <typename T, int P, int XCHG, bool DO_GX, bool DO_GW, int SM = 0>
__global__ __launch_bounds__(512) void agg_bwd_nchw_k3_lds(const T* __restrict__ gout, const T* __restrict__ x,
                                                          const T* __restrict__ w, T* __restrict__ gx,
                                                          T* __restrict__ gw, int C, int wC, int H, int W, int R,
                                                          int tiles_per_n, int sle, int JP, int64_t elems,
                                                          int xcd_remap) {
    typedef typename AccOf<T>::type A;
    constexpr int VE = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    T* gslab = reinterpret_cast<T*>(cot_smem);
    T* xslab = gslab + (int64_t)JP * sle;  // (only used when DO_GW)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int segs = W / P, J = C / wC, TR = (blockDim.x >> 6) * R, rows_n = wC * H;
    const int64_t HW = (int64_t)H * W;
    const unsigned lb = logical_block(xcd_remap);
    const int tile = lb % tiles_per_n;
    const int n = lb / tiles_per_n;
    const int rho0 = tile * TR;
    int64_t gs = (int64_t)(rho0 - 1) * W;
    if (gs < 0) gs = 0;
    gs &= ~(int64_t)(VE - 1);

    int r = lane / segs, seg = lane - r * segs;
    int rho = rho0 + wave * R + r;
    const bool valid = r < R && rho < rows_n;
    if (!valid) {
        rho = rows_n - 1;
        seg = segs - 1;
    }
    const int wc = rho / H, h = rho - wc * H, w0 = seg * P;
    const bool has_left = seg > 0, has_right = seg < segs - 1;
    const int64_t plane = (int64_t)n * wC + wc;

    const int64_t cstride = (int64_t)wC * HW;
    const int64_t img = (int64_t)n * C * HW;
    auto stage_phase = [&](int j0, int jn) __attribute__((always_inline)) {
        stage_slabs<T>(gout, img + (int64_t)j0 * cstride + gs, cstride, jn, sle, elems, gslab);
        if (DO_GW) stage_slabs<T>(x, img + (int64_t)j0 * cstride + gs, cstride, jn, sle, elems, xslab);
    };
    stage_phase(0, J < JP ? J : JP);  // the first phase's slabs are on their way while the weights are fetched

    A ws[9][P];
    if (DO_GX) {
        const T* wp = w + plane * 9 * HW;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int t = kh * 3 + kw;
                A row[P + 2];
                expand_row<T, P, XCHG, A>(load_row_raw<T, P>(wp + t * HW, h + 1 - kh, H, W, w0), has_left, has_right,
                                          row);
#pragma unroll
                for (int i = 0; i < P; ++i) ws[t][i] = row[i + 2 - kw];
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
    const int64_t lidx = (int64_t)rho * W + w0 - gs;
    for (int j0 = 0; j0 < J; j0 += JP) {
        const int jn = (J - j0 < JP) ? (J - j0) : JP;
        if (j0 > 0) {
            __syncthreads();  // everyone finished reading the previous phase's slabs
            stage_phase(j0, jn);
        }
        __syncthreads();
        for (int jj = 0; jj < jn; ++jj) {
            const T* gj = gslab + (int64_t)jj * sle;
            A gr[3][P + 2];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int hr = h - 1 + rr;
                if (DO_GX || rr == 1)
                    expand_row<T, P, XCHG, A>(lds_row<T, P>(gj, lidx + (int64_t)(rr - 1) * W, hr >= 0 && hr < H),
                                              has_left, has_right, gr[rr]);
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
                if (valid) stv<T, P>(gx + img + (int64_t)(j0 + jj) * cstride + (int64_t)rho * W + w0, o);
            }
            if (DO_GW) {
                const T* xj = xslab + (int64_t)jj * sle;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hr = h - 1 + kh;
                    // (rows outside the image and halo columns past a row's end are read as they are -- staged
                    // neighbours, in-bounds -- because every accumulator they feed is cleared after the loop)
                    A xr[P + 2];
                    expand_row_unmasked<T, P, XCHG, A>(lds_row<T, P>(xj, lidx + (int64_t)(kh - 1) * W, true), xr);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int i = 0; i < P; ++i) gwacc[kh * 3 + kw][i] += xr[i + kw] * gr[1][i + 1];
                }
            }
        }
    }
    if (DO_GW) {  // taps that reach outside the image saw neighbouring rows / planes: their sums are discarded here, once
                  // per item, instead of masking x once per channel.  Selection, so a padded tap's gradient is an exact
                  // zero whatever gO holds there (Inf/NaN included) -- as in the reference's kernel, which writes 0 for
                  // padded taps without multiplying (aggregation_zeropad.py:81-110)
        const bool top = h > 0, bottom = h < H - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int i = 0; i < P; ++i) {
                gwacc[0 + kw][i] = top ? gwacc[0 + kw][i] : (A)0;
                gwacc[6 + kw][i] = bottom ? gwacc[6 + kw][i] : (A)0;
            }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            gwacc[kh * 3 + 0][0] = has_left ? gwacc[kh * 3 + 0][0] : (A)0;
            gwacc[kh * 3 + 2][P - 1] = has_right ? gwacc[kh * 3 + 2][P - 1] : (A)0;
        }
    }
    if (DO_GW && SM) {
        const T* pp = w + plane * 9 * HW + (int64_t)h * W + w0;
        Vec<T, P> pr[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) pr[t] = ldv<T, P>(pp + t * HW);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            A dot = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) dot += (A)pr[t].v[i] * gwacc[t][i];
#pragma unroll
            for (int t = 0; t < 9; ++t) gwacc[t][i] = (A)pr[t].v[i] * (gwacc[t][i] - dot);
        }
    }
    if (DO_GW && valid) {
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
static const char* g_last_kernel = "";  // diagnostic only (written by whichever thread launched last)
const char* last_kernel_nchw() { return g_last_kernel; }

// run-time tuning knobs (cot_set_tuning): 0 = kernel version (0 auto, 1 force v1, 2 force v2),
// 1 = max P forward, 2 = max P backward, 3 = lane-exchange primitive (-1 auto/probe, 0 DPP, 1 ds_bpermute),
// 4 = v3 fused backward: channel groups staged in LDS per phase (0 = default 4),
// 5 = v3 waves per workgroup (4 or 8), 6 = v3 extra dynamic LDS per workgroup in KiB (occupancy shaping: fewer
//     co-resident workgroups => their load / compute / store phases interleave instead of running in lock-step),
// 7 = v3 XCD-aware tile order (-1 automatic, 0 off, 1 on; see xcd_order below), 8 = split the fused backward into a gX launch
//     and a gW launch (0 fused, 1 split; A/B on the MI355X: the split form is slower, fused stays the default)
// defaults from the on-device A/B (profiles/r01_agg_variants.log, N80xC64x56x56 bf16): forward P=4 (v3 22.0 us vs 26.0 at P=8),
// fused backward P=2 with 4 channel groups per LDS phase (v3 45.4 us; 27.3 vs 31.8 us at 28x28)
static int g_tune[9] = {0, 4, 2, -1, 0, 4, 0, -1, 0};
// XCD-aware tile order (key 7): -1 = automatic -- on for planes up to 28 x 28 (round-2 A/B on the MI355X: 28 x 28 forward
// 15.7 -> 14.7 us, backward 23.6 -> 22.5 us; 56 x 56 is 2 % slower with it), 0 off, 1 on
static inline int xcd_order(int H, int W) { return g_tune[7] < 0 ? (H * W <= 28 * 28 ? 1 : 0) : g_tune[7]; }
int set_tuning_nchw(int key, int value) {
    if (key < 0 || key > 8) return -1;
    g_tune[key] = value;
    return 0;
}

// One-time device probe: does v_mov_b32_dpp wave_shr:1 / wave_shl:1 move data the way the v2 kernels assume?
int xchg_mode() {
    if (g_tune[3] >= 0) return g_tune[3];
    if (g_dry_run) return 0;  // (no device to probe: report the path every MI355X takes)
    static const int probed = []() {
        int* d = nullptr;
        if (hipMalloc((void**)&d, 128 * sizeof(int)) != hipSuccess) return 1;
        hipLaunchKernelGGL(dpp_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, d);
        int h[128];
        bool ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
        hipFree(d);
        for (int l = 0; ok && l < 64; ++l) {
            if (h[l] != (l == 0 ? 0 : l)) ok = false;
            if (h[64 + l] != (l == 63 ? 0 : l + 2)) ok = false;
        }
        return ok ? 0 : 1;
    }();
    return probed;
}

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

static inline bool use_v2(int W, int P) { return g_tune[0] != 1 && (W / P) <= 64; }  // also the fallback of v3

// v3 (LDS slabs) needs 16-byte aligned channel-group / image strides and a tile that fits LDS comfortably
struct LdsPlan {
    bool ok;
    int R, TR, tiles, sle, nthreads;
    size_t lds_bytes;
};
template <typename T> static inline LdsPlan plan_lds(const cot_agg_geom& g, int P, int nslab) {
    LdsPlan p{};
    constexpr int VE = (int)(16 / sizeof(T));
    const int segs = g.W / P;
    const int64_t HW = (int64_t)g.H * g.W;
    if (segs > 64 || ((int64_t)g.wC * HW) % VE != 0 || ((int64_t)g.C * HW) % VE != 0) return p;
    if ((int64_t)g.N * g.C * HW < VE) return p;
    const int nw = g_tune[5] == 8 ? 8 : 4;
    p.nthreads = nw * 64;
    p.R = 64 / segs;
    p.TR = nw * p.R;
    p.tiles = (g.wC * g.H + p.TR - 1) / p.TR;
    p.sle = (((p.TR + 2) * g.W + VE) + VE - 1) / VE * VE;  // rows + alignment slack, multiple of a 16-byte chunk
    const int64_t chunks = (int64_t)nslab * (p.sle / VE);
    p.lds_bytes = (size_t)chunks * 16;
    p.ok = p.lds_bytes <= 64 * 1024;
    if (g_tune[6] > 0) p.lds_bytes += (size_t)g_tune[6] * 1024;
    if (p.lds_bytes > 160 * 1024) p.lds_bytes = 160 * 1024;
    return p;
}

template <typename T, int P>
static int launch_fwd_k3(const T* x, const T* w, T* out, const cot_agg_geom& g, hipStream_t s) {
    static_assert(P * sizeof(T) <= 16, "row vector wider than 16 bytes");
    if ((g_tune[0] == 0 || g_tune[0] == 3) && sizeof(T) <= 4) {
        const LdsPlan p = plan_lds<T>(g, P, g.C / g.wC);
        if (p.ok) {
            const dim3 grid((unsigned)((int64_t)p.tiles * g.N * g.heads)), block(p.nthreads);
            const int64_t xe = (int64_t)g.N * g.C * g.H * g.W;
            if (xchg_mode() == 0)
                COT_LAUNCH((agg_fwd_nchw_k3_lds<T, P, 0, 0>), grid, block, p.lds_bytes, s, x, w, out, g.heads, g.C,
                                   g.wC, g.H, g.W, p.R, p.tiles, p.sle, xe, (T*)nullptr, xcd_order(g.H, g.W), Gn9Args<T>{}, (float*)nullptr);
            else
                COT_LAUNCH((agg_fwd_nchw_k3_lds<T, P, 1, 0>), grid, block, p.lds_bytes, s, x, w, out, g.heads, g.C,
                                   g.wC, g.H, g.W, p.R, p.tiles, p.sle, xe, (T*)nullptr, xcd_order(g.H, g.W), Gn9Args<T>{}, (float*)nullptr);
            g_last_kernel = "agg_fwd_nchw_k3_lds";
            return check_launch(g_last_kernel);
        }
    }
    if (use_v2(g.W, P)) {
        const int segs = g.W / P, L = (64 / segs) * segs;
        const int64_t items = (int64_t)g.N * g.heads * g.wC * g.H * segs;
        const int64_t waves = ceil_div64(items, L);
        const dim3 grid((unsigned)ceil_div64(waves, 4)), block(256);
        if (xchg_mode() == 0)
            COT_LAUNCH((agg_fwd_nchw_k3_v2<T, P, 0>), grid, block, 0, s, x, w, out, g.heads, g.C, g.wC, g.H,
                               g.W, L, items);
        else
            COT_LAUNCH((agg_fwd_nchw_k3_v2<T, P, 1>), grid, block, 0, s, x, w, out, g.heads, g.C, g.wC, g.H,
                               g.W, L, items);
        g_last_kernel = P == 8 ? "agg_fwd_nchw_k3_v2<P8>" : P == 4 ? "agg_fwd_nchw_k3_v2<P4>"
                        : P == 2 ? "agg_fwd_nchw_k3_v2<P2>" : "agg_fwd_nchw_k3_v2<P1>";
        return check_launch(g_last_kernel);
    }
    const int64_t items = (int64_t)g.N * g.heads * g.wC * g.H * (g.W / P);
    COT_LAUNCH((agg_fwd_nchw_k3<T, P>), dim3(grid_for(items, 256, INT32_MAX)), dim3(256), 0, s, x, w, out,
                       g.heads, g.C, g.wC, g.H, g.W, items);
    g_last_kernel = "agg_fwd_nchw_k3<v1>";
    return check_launch(g_last_kernel);
}

template <typename T>
int agg_forward_nchw(const T* x, const T* w, T* out, const cot_agg_geom& g, int Ho, int Wo, hipStream_t s,
                     const char* tname) {
    (void)tname;
    if (is_k3_fast(g)) {
        constexpr int LIM = (int)(16 / sizeof(T));
        switch (pick_P<T>(g.W, g_tune[1])) {
            case 8: return launch_fwd_k3<T, (LIM >= 8 ? 8 : 1)>(x, w, out, g, s);
            case 4: return launch_fwd_k3<T, (LIM >= 4 ? 4 : 1)>(x, w, out, g, s);
            case 2: return launch_fwd_k3<T, 2>(x, w, out, g, s);
            default: return launch_fwd_k3<T, 1>(x, w, out, g, s);
        }
    }
    const int64_t total = (int64_t)g.N * g.heads * g.C * Ho * Wo;
    COT_LAUNCH((agg_fwd_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, x, w, out, g, Ho, Wo,
                       total);
    g_last_kernel = "agg_fwd_nchw_generic";
    return check_launch("agg_fwd_nchw_generic");
}

template <typename T, int P, bool GX, bool GW>
static int launch_bwd_k3_sel(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g,
                             hipStream_t s) {
    if ((g_tune[0] == 0 || g_tune[0] == 3) && sizeof(T) <= 4) {
        const int J = g.C / g.wC;
        int JP = g_tune[4] > 0 && g_tune[4] <= J ? g_tune[4] : (J >= 4 ? 4 : J);
        const LdsPlan p = plan_lds<T>(g, P, (GW ? 2 : 1) * JP);
        if (p.ok) {
            const dim3 grid((unsigned)((int64_t)p.tiles * g.N)), block(p.nthreads);
            const int64_t ne = (int64_t)g.N * g.C * g.H * g.W;
            if (xchg_mode() == 0)
                COT_LAUNCH((agg_bwd_nchw_k3_lds<T, P, 0, GX, GW>), grid, block, p.lds_bytes, s, gout, x, w, gx, gw,
                                   g.C, g.wC, g.H, g.W, p.R, p.tiles, p.sle, JP, ne, xcd_order(g.H, g.W));
            else
                COT_LAUNCH((agg_bwd_nchw_k3_lds<T, P, 1, GX, GW>), grid, block, p.lds_bytes, s, gout, x, w, gx, gw,
                                   g.C, g.wC, g.H, g.W, p.R, p.tiles, p.sle, JP, ne, xcd_order(g.H, g.W));
            g_last_kernel = GX && GW ? "agg_bwd_nchw_k3_lds<gx,gw>" : GX ? "agg_bwd_nchw_k3_lds<gx>" : "agg_bwd_nchw_k3_lds<gw>";
            return check_launch(g_last_kernel);
        }
    }
    if (use_v2(g.W, P)) {
        const int segs = g.W / P, L = (64 / segs) * segs;
        const int64_t items = (int64_t)g.N * g.wC * g.H * segs;
        const int64_t waves = ceil_div64(items, L);
        const dim3 grid((unsigned)ceil_div64(waves, 4)), block(256);
        if (xchg_mode() != 0)
            COT_LAUNCH((agg_bwd_nchw_k3_v2<T, P, 1, GX, GW, 1>), grid, block, 0, s, gout, x, w, gx, gw, g.C,
                               g.wC, g.H, g.W, L, items);
        else
            COT_LAUNCH((agg_bwd_nchw_k3_v2<T, P, 0, GX, GW, 1>), grid, block, 0, s, gout, x, w, gx, gw, g.C,
                               g.wC, g.H, g.W, L, items);
        g_last_kernel = GX && GW ? "agg_bwd_nchw_k3_v2<gx,gw>" : GX ? "agg_bwd_nchw_k3_v2<gx>" : "agg_bwd_nchw_k3_v2<gw>";
        return check_launch(g_last_kernel);
    }
    const int64_t items = (int64_t)g.N * g.wC * g.H * (g.W / P);
    COT_LAUNCH((agg_bwd_nchw_k3<T, P, GX, GW>), dim3(grid_for(items, 256, INT32_MAX)), dim3(256), 0, s, gout, x,
                       w, gx, gw, g.C, g.wC, g.H, g.W, items);
    g_last_kernel = GX && GW ? "agg_bwd_nchw_k3<gx,gw>" : GX ? "agg_bwd_nchw_k3<gx>" : "agg_bwd_nchw_k3<gw>";
    return check_launch(g_last_kernel);
}

template <typename T, int P>
static int launch_bwd_k3(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g, hipStream_t s) {
    if (gx && gw && g_tune[8] == 1) {
        const int rc = launch_bwd_k3_sel<T, P, true, false>(gout, x, w, gx, (T*)nullptr, g, s);
        return rc ? rc : launch_bwd_k3_sel<T, P, false, true>(gout, x, w, (T*)nullptr, gw, g, s);
    }
    if (gx && gw) return launch_bwd_k3_sel<T, P, true, true>(gout, x, w, gx, gw, g, s);
    if (gx) return launch_bwd_k3_sel<T, P, true, false>(gout, x, w, gx, gw, g, s);
    return launch_bwd_k3_sel<T, P, false, true>(gout, x, w, gx, gw, g, s);
}

// packed-bf16 dot-product form of the fused backward (agg_dot2.hip): -1 = geometry not covered
int agg_backward_nchw_dot2(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw, const cot_agg_geom& g,
                           hipStream_t s);
template <typename T> static inline int try_dot2(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, hipStream_t) { return -1; }
template <> inline int try_dot2<bf16_t>(const bf16_t* gout, const bf16_t* x, const bf16_t* w, bf16_t* gx, bf16_t* gw,
                                        const cot_agg_geom& g, hipStream_t s) {
    return agg_backward_nchw_dot2(gout, x, w, gx, gw, g, s);
}

template <typename T>
int agg_backward_nchw(const T* gout, const T* x, const T* w, T* gx, T* gw, const cot_agg_geom& g, int Ho, int Wo,
                      hipStream_t s) {
    if (is_k3_fast(g) && g.heads == 1 && gx && gw && g_tune[0] == 0 && g_tune[8] != 1) {  // (key 0 = 3 forces the LDS kernel below)
        const int rc = try_dot2<T>(gout, x, w, gx, gw, g, s);
        if (rc >= 0) {
            g_last_kernel = "agg_bwd_nchw_k3_dot2<gx,gw>";
            return rc;
        }
    }
    if (is_k3_fast(g) && g.heads == 1) {
        // P capped (default 4): the fused kernel keeps 18*P accumulate-type values of weights / weight-gradients live
        constexpr int LIM = (int)(16 / sizeof(T));
        switch (pick_P<T>(g.W, g_tune[2] > 4 ? 4 : g_tune[2])) {
            case 4: return launch_bwd_k3<T, (LIM >= 4 ? 4 : 1)>(gout, x, w, gx, gw, g, s);
            case 2: return launch_bwd_k3<T, 2>(gout, x, w, gx, gw, g, s);
            default: return launch_bwd_k3<T, 1>(gout, x, w, gx, gw, g, s);
        }
    }
    int rc = COT_OK;
    if (gx) {
        const int64_t total = (int64_t)g.N * g.C * g.H * g.W;
        COT_LAUNCH((agg_bwd_input_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, gout, w, gx, g,
                           Ho, Wo, total);
        g_last_kernel = "agg_bwd_input_nchw_generic";
        rc = check_launch("agg_bwd_input_nchw_generic");
        if (rc) return rc;
    }
    if (gw) {
        const int64_t total = (int64_t)g.N * g.heads * g.wC * g.kh * g.kw * Ho * Wo;
        COT_LAUNCH((agg_bwd_weight_nchw_generic<T>), dim3(grid_for(total, 256)), dim3(256), 0, s, gout, x, gw,
                           g, Ho, Wo, total);
        g_last_kernel = "agg_bwd_weight_nchw_generic";
        rc = check_launch("agg_bwd_weight_nchw_generic");
    }
    return rc;
}

// window-softmax mode: only the LDS-staged 3x3 kernels implement it; anything else reports "unsupported" and the
// Python layer composes softmax + aggregation instead.
template <typename T, int P>
static int launch_softmax_fwd(const T* x, const T* logits, T* out, T* probs, const cot_agg_geom& g, hipStream_t s) {
    const LdsPlan p = plan_lds<T>(g, P, g.C / g.wC);
    if (!p.ok) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)p.tiles * g.N * g.heads)), block(p.nthreads);
    const int64_t xe = (int64_t)g.N * g.C * g.H * g.W;
    if (xchg_mode() == 0)
        COT_LAUNCH((agg_fwd_nchw_k3_lds<T, P, 0, 1>), grid, block, p.lds_bytes, s, x, logits, out, g.heads, g.C, g.wC,
                   g.H, g.W, p.R, p.tiles, p.sle, xe, probs, xcd_order(g.H, g.W), Gn9Args<T>{}, (float*)nullptr);
    else
        COT_LAUNCH((agg_fwd_nchw_k3_lds<T, P, 1, 1>), grid, block, p.lds_bytes, s, x, logits, out, g.heads, g.C, g.wC,
                   g.H, g.W, p.R, p.tiles, p.sle, xe, probs, xcd_order(g.H, g.W), Gn9Args<T>{}, (float*)nullptr);
    g_last_kernel = "agg_fwd_nchw_k3_lds<softmax>";
    return check_launch(g_last_kernel);
}
template <typename T, int P>
static int launch_softmax_bwd(const T* gout, const T* x, const T* probs, T* gx, T* glogits, const cot_agg_geom& g,
                              hipStream_t s) {
    const int J = g.C / g.wC;
    const int JP = J >= 4 ? 4 : J;
    const LdsPlan p = plan_lds<T>(g, P, 2 * JP);
    if (!p.ok) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)p.tiles * g.N)), block(p.nthreads);
    const int64_t ne = (int64_t)g.N * g.C * g.H * g.W;
    if (xchg_mode() == 0)
        COT_LAUNCH((agg_bwd_nchw_k3_lds<T, P, 0, true, true, 1>), grid, block, p.lds_bytes, s, gout, x, probs, gx,
                   glogits, g.C, g.wC, g.H, g.W, p.R, p.tiles, p.sle, JP, ne, xcd_order(g.H, g.W));
    else
        COT_LAUNCH((agg_bwd_nchw_k3_lds<T, P, 1, true, true, 1>), grid, block, p.lds_bytes, s, gout, x, probs, gx,
                   glogits, g.C, g.wC, g.H, g.W, p.R, p.tiles, p.sle, JP, ne, xcd_order(g.H, g.W));
    g_last_kernel = "agg_bwd_nchw_k3_lds<softmax>";
    return check_launch(g_last_kernel);
}

// GroupNorm-9 prologue (SM = 2): bf16, the model geometries of the LDS kernel; COT_ERR_UNSUPPORTED otherwise
template <int P>
static int launch_gn9_fwd(const bf16_t* x, const bf16_t* logits, bf16_t* out, const cot_agg_geom& g, const Gn9Args<bf16_t>& gn,
                          hipStream_t s) {
    const LdsPlan p = plan_lds<bf16_t>(g, P, g.C / g.wC);
    if (!p.ok) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)p.tiles * g.N)), block(p.nthreads);
    const int64_t xe = (int64_t)g.N * g.C * g.H * g.W;
    if (xchg_mode() == 0)
        COT_LAUNCH((agg_fwd_nchw_k3_lds<bf16_t, P, 0, 2>), grid, block, p.lds_bytes, s, x, logits, out, 1, g.C, g.wC, g.H, g.W, p.R,
                   p.tiles, p.sle, xe, (bf16_t*)nullptr, xcd_order(g.H, g.W), gn, (float*)nullptr);
    else
        COT_LAUNCH((agg_fwd_nchw_k3_lds<bf16_t, P, 1, 2>), grid, block, p.lds_bytes, s, x, logits, out, 1, g.C, g.wC, g.H, g.W, p.R,
                   p.tiles, p.sle, xe, (bf16_t*)nullptr, xcd_order(g.H, g.W), gn, (float*)nullptr);
    g_last_kernel = "agg_fwd_nchw_k3_lds<gn9>";
    return check_launch(g_last_kernel);
}
int agg_gn9_forward_nchw(const bf16_t* x, const bf16_t* logits, const float* mean, const float* rstd, const bf16_t* gamma,
                         const bf16_t* beta, int gimg, bf16_t* out, const cot_agg_geom& g, hipStream_t s) {
    if (!is_k3_fast(g) || g.heads != 1 || (int64_t)g.N * g.wC >= ((int64_t)1 << 31)) return COT_ERR_UNSUPPORTED;
    const Gn9Args<bf16_t> gn{mean, rstd, gamma, beta, gimg};
    switch (pick_P<bf16_t>(g.W, g_tune[1])) {
        case 8: return launch_gn9_fwd<8>(x, logits, out, g, gn, s);
        case 4: return launch_gn9_fwd<4>(x, logits, out, g, gn, s);
        case 2: return launch_gn9_fwd<2>(x, logits, out, g, gn, s);
        default: return COT_ERR_UNSUPPORTED;
    }
}

// ---- forward with the following BatchNorm's row statistics out of the epilogue (ST = 1): bf16, the LDS kernel's geometries, one head, at
// most 8 channels per weight plane group; gn.mean != NULL: GroupNorm-9 prologue (SM = 2).  COT_ERR_UNSUPPORTED otherwise (the caller
// runs the plain forward and a statistics pass).
template <int P>
static int launch_rowstats_fwd(const bf16_t* x, const bf16_t* w, bf16_t* out, float* rowstats, const cot_agg_geom& g,
                               const Gn9Args<bf16_t>& gn, hipStream_t s) {
    const int J = g.C / g.wC;
    LdsPlan p = plan_lds<bf16_t>(g, P, J);
    if (!p.ok) return COT_ERR_UNSUPPORTED;
    const size_t slab = ((size_t)J * p.sle * sizeof(bf16_t) + 15) & ~(size_t)15;
    const size_t lds = slab + (size_t)2 * J * p.nthreads * sizeof(float);  // the slabs, then [2 J][threads] sums
    if (lds > 64 * 1024) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)p.tiles * g.N)), block(p.nthreads);
    const int64_t xe = (int64_t)g.N * g.C * g.H * g.W;
    const int xo = xcd_order(g.H, g.W);
#define COT_RS(XC_, SM_) COT_LAUNCH((agg_fwd_nchw_k3_lds<bf16_t, P, XC_, SM_, 1>), grid, block, lds, s, x, w, out, 1, g.C, g.wC, g.H, g.W, p.R, p.tiles, p.sle, xe, (bf16_t*)nullptr, xo, gn, rowstats)
    if (gn.mean) {
        if (xchg_mode() == 0) COT_RS(0, 2);
        else COT_RS(1, 2);
    } else {
        if (xchg_mode() == 0) COT_RS(0, 0);
        else COT_RS(1, 0);
    }
#undef COT_RS
    g_last_kernel = gn.mean ? "agg_fwd_nchw_k3_lds<gn9,rowstats>" : "agg_fwd_nchw_k3_lds<rowstats>";
    return check_launch(g_last_kernel);
}
int agg_forward_rowstats_nchw(const bf16_t* x, const bf16_t* w, bf16_t* out, float* rowstats, const float* mean, const float* rstd,
                              const bf16_t* gamma, const bf16_t* beta, int gimg, const cot_agg_geom& g, hipStream_t s) {
    if (!is_k3_fast(g) || g.heads != 1 || (int64_t)g.N * g.wC >= ((int64_t)1 << 31) || g.C % g.wC != 0 || g.C / g.wC > 8 ||
        !(g_tune[0] == 0 || g_tune[0] == 3))
        return COT_ERR_UNSUPPORTED;
    const Gn9Args<bf16_t> gn{mean, rstd, gamma, beta, gimg};
    switch (pick_P<bf16_t>(g.W, g_tune[1])) {
        case 8: return launch_rowstats_fwd<8>(x, w, out, rowstats, g, gn, s);
        case 4: return launch_rowstats_fwd<4>(x, w, out, rowstats, g, gn, s);
        case 2: return launch_rowstats_fwd<2>(x, w, out, rowstats, g, gn, s);
        default: return mean ? COT_ERR_UNSUPPORTED : launch_rowstats_fwd<1>(x, w, out, rowstats, g, gn, s);  // (odd rows with the GroupNorm prologue: as agg_gn9_forward_nchw)
    }
}

template <typename T>
int agg_softmax_forward_nchw(const T* x, const T* logits, T* out, T* probs, const cot_agg_geom& g, hipStream_t s) {
    if (!is_k3_fast(g) || sizeof(T) > 4) return COT_ERR_UNSUPPORTED;
    switch (pick_P<T>(g.W, 4)) {
        case 4: return launch_softmax_fwd<T, (sizeof(T) <= 4 ? 4 : 1)>(x, logits, out, probs, g, s);
        case 2: return launch_softmax_fwd<T, 2>(x, logits, out, probs, g, s);
        default: return launch_softmax_fwd<T, 1>(x, logits, out, probs, g, s);
    }
}
template <typename T>
int agg_softmax_backward_nchw(const T* gout, const T* x, const T* probs, T* gx, T* glogits, const cot_agg_geom& g,
                              hipStream_t s) {
    if (!is_k3_fast(g) || g.heads != 1 || sizeof(T) > 4) return COT_ERR_UNSUPPORTED;
    switch (pick_P<T>(g.W, 2)) {
        case 2: return launch_softmax_bwd<T, 2>(gout, x, probs, gx, glogits, g, s);
        default: return launch_softmax_bwd<T, 1>(gout, x, probs, gx, glogits, g, s);
    }
}

#define INSTANTIATE(T)                                                                                          \
    template int agg_softmax_forward_nchw<T>(const T*, const T*, T*, T*, const cot_agg_geom&, hipStream_t);     \
    template int agg_softmax_backward_nchw<T>(const T*, const T*, const T*, T*, T*, const cot_agg_geom&,        \
                                              hipStream_t);                                                     \
    template int agg_forward_nchw<T>(const T*, const T*, T*, const cot_agg_geom&, int, int, hipStream_t,        \
                                     const char*);                                                              \
    template int agg_backward_nchw<T>(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, int, int,      \
                                      hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
INSTANTIATE(bf16_t)
INSTANTIATE(f16_t)

}  // namespace cot
