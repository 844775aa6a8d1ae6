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
#include <algorithm>

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

// One thread per 2 x 2 block of input pixels (rows 2a, 2a+1; columns 2b, 2b+1).  The windows that contain them are
// (oh, ow) in {a, a+1} x {b, b+1} -- pixel (2a, 2b) belongs to window (a, b) only, (2a+1, 2b) to (a, b) and (a+1, b),
// (2a, 2b+1) to (a, b) and (a, b+1), (2a+1, 2b+1) to all four -- so the thread evaluates four arg-maxima (a 5 x 5 patch of x)
// for four pixels: 6 loads per pixel instead of the 25 of the one-thread-per-pixel form (measured 1.47 ms per step for the
// 80 x 64 x 112 x 112 stem output, ~60 us of traffic).  Same tie rule, same results.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd(const T* __restrict__ gy, const T* __restrict__ x,
                                                       T* __restrict__ gx, int64_t planes, int H, int W, int Ho, int Wo) {
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;  // 2 x 2 blocks per plane
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Hb * Wb) return;
    const int b = (int)(i % Wb), a = (int)((i / Wb) % Hb);
    const int64_t pl = i / ((int64_t)Wb * Hb);
    const T* xp = x + pl * H * W;
    const T* gp = gy + pl * Ho * Wo;
    int am[2][2];
    float g[2][2];
#pragma unroll
    for (int da = 0; da < 2; ++da)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int oh = a + da, ow = b + db;
            const bool ok = oh < Ho && ow < Wo;
            am[da][db] = ok ? window_argmax<T>(xp, oh, ow, H, W) : -1;
            g[da][db] = ok ? (float)gp[oh * Wo + ow] : 0.f;
        }
    const int h0 = 2 * a, w0 = 2 * b;
    const int p00 = h0 * W + w0, p01 = p00 + 1, p10 = p00 + W, p11 = p10 + 1;
    const float s00 = am[0][0] == p00 ? g[0][0] : 0.f;
    gx[pl * H * W + p00] = (T)s00;
    if (w0 + 1 < W) {
        float s = 0.f;  // windows in torch's accumulation order: (oh, ow) ascending
        if (am[0][0] == p01) s += g[0][0];
        if (am[0][1] == p01) s += g[0][1];
        gx[pl * H * W + p01] = (T)s;
    }
    if (h0 + 1 < H) {
        float s = 0.f;
        if (am[0][0] == p10) s += g[0][0];
        if (am[1][0] == p10) s += g[1][0];
        gx[pl * H * W + p10] = (T)s;
        if (w0 + 1 < W) {
            float s2 = 0.f;
            if (am[0][0] == p11) s2 += g[0][0];
            if (am[0][1] == p11) s2 += g[0][1];
            if (am[1][0] == p11) s2 += g[1][0];
            if (am[1][1] == p11) s2 += g[1][1];
            gx[pl * H * W + p11] = (T)s2;
        }
    }
}

// ---- the same pooling with the arg-max kept as one byte per window (its tap kh*3 + kw): the backward then reads dY and the
// taps (2 + 1 bytes per window) instead of a 5 x 5 patch of x per 2 x 2 block -- 176 MB instead of 290 MB of traffic and a
// tenth of the instructions for the 80 x 64 x 112 x 112 stem output.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_tap(const T* __restrict__ x, T* __restrict__ y,
                                                           uint8_t* __restrict__ tap, int64_t planes, int H, int W, int Ho,
                                                           int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho);
    const int64_t pl = i / ((int64_t)Wo * Ho);
    const T* xp = x + pl * H * W;
    const int p = window_argmax<T>(xp, oh, ow, H, W);
    y[i] = xp[p];
    const int h = p / W, w = p - h * W;
    tap[i] = (uint8_t)((h - (2 * oh - 1)) * 3 + (w - (2 * ow - 1)));
}

// One thread per 2 x 2 block of input pixels, as above; pixel (2a, 2b) is tap 4 of window (a, b); (2a, 2b+1) is tap 5 of
// (a, b) and tap 3 of (a, b+1); (2a+1, 2b) is tap 7 of (a, b) and tap 1 of (a+1, b); (2a+1, 2b+1) is tap 8 / 6 / 2 / 0 of
// (a, b) / (a, b+1) / (a+1, b) / (a+1, b+1).  Sums in torch's accumulation order (windows ascending).
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_tap(const T* __restrict__ gy, const uint8_t* __restrict__ tap,
                                                           T* __restrict__ gx, int64_t planes, int H, int W, int Ho,
                                                           int Wo) {
    const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Hb * Wb) return;
    const int b = (int)(i % Wb), a = (int)((i / Wb) % Hb);
    const int64_t pl = i / ((int64_t)Wb * Hb);
    const T* gp = gy + pl * Ho * Wo;
    const uint8_t* tp = tap + pl * Ho * Wo;
    int t[2][2];
    float g[2][2];
#pragma unroll
    for (int da = 0; da < 2; ++da)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int oh = a + da, ow = b + db;
            const bool ok = oh < Ho && ow < Wo;
            t[da][db] = ok ? (int)tp[oh * Wo + ow] : -1;
            g[da][db] = ok ? (float)gp[oh * Wo + ow] : 0.f;
        }
    T* o = gx + pl * H * W + (int64_t)(2 * a) * W + 2 * b;
    const bool right = 2 * b + 1 < W, below = 2 * a + 1 < H;
    const float s00 = t[0][0] == 4 ? g[0][0] : 0.f;
    float s01 = 0.f, s10 = 0.f, s11 = 0.f;
    if (t[0][0] == 5) s01 += g[0][0];
    if (t[0][1] == 3) s01 += g[0][1];
    if (t[0][0] == 7) s10 += g[0][0];
    if (t[1][0] == 1) s10 += g[1][0];
    if (t[0][0] == 8) s11 += g[0][0];
    if (t[0][1] == 6) s11 += g[0][1];
    if (t[1][0] == 2) s11 += g[1][0];
    if (t[1][1] == 0) s11 += g[1][1];
    if (right && (W & 1) == 0) {  // even rows: the pair is one aligned store
        Vec<T, 2> v;
        v.v[0] = (T)s00; v.v[1] = (T)s01;
        stv<T, 2>(o, v);
        if (below) {
            v.v[0] = (T)s10; v.v[1] = (T)s11;
            stv<T, 2>(o + W, v);
        }
    } else {
        o[0] = (T)s00;
        if (right) o[1] = (T)s01;
        if (below) {
            o[W] = (T)s10;
            if (right) o[W + 1] = (T)s11;
        }
    }
}

// ---- row-block form of the four kernels a CoTNet step runs (average pooling forward / backward, max pooling with byte taps
// forward / backward), for even W.  The one-lane-per-pixel kernels above are bound by their instruction count, not by
// memory: two 64-bit divisions, nine (or four) separately addressed and bounds-checked 2-byte loads and one 2-byte store per
// pixel -- 119 us for the gradient of the 56 x 56 -> 28 x 28 average pooling of a stride-2 block (80 MB of traffic, 13 us at
// the roofline), 155 us for the stem's max pooling; staging whole planes through LDS with the same per-pixel arithmetic
// measured no faster (gpurun_out/r3s29_pool.log).  Here a lane owns BG consecutive windows of one output row: it loads
// the 2 BG + 1 input columns of each of the three rows once (one wide load + the left neighbour) and produces BG outputs
// (forward), or loads BG + 1 columns of two gradient rows and produces a 2 x 2 BG block of the input gradient (backward) --
// the index arithmetic is paid once per lane, accesses are 4 .. 16 bytes wide.  Same sums in the same order: identical results.
template <typename T, int BG, bool MAXP>  // forward: y (and the arg-max taps of the max pooling)
__global__ __launch_bounds__(256) void pool3x3s2_fwd_blk(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ tap,
                                                         int64_t planes, int H, int W, int Ho, int Wo) {
    const int NB = Wo / BG;  // (Wo % BG == 0: host)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const int ow0 = bg * BG;
    const T* xp = x + pl * H * W;
    float acc[BG];
    int am[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        acc[j] = MAXP ? -INFINITY : 0.f;
        am[j] = -1;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
        if (h < 0 || h >= H) continue;
        float L[2 * BG + 1];
        const T* rp = xp + h * W + 2 * ow0;
        L[0] = ow0 > 0 ? (float)rp[-1] : 0.f;
        const Vec<T, 2 * BG> v = ldv<T, 2 * BG>(rp);
#pragma unroll
        for (int c = 0; c < 2 * BG; ++c) L[1 + c] = (float)v.v[c];
#pragma unroll
        for (int j = 0; j < BG; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (kw == 0 && ow0 + j == 0) continue;  // column -1
                const float val = L[2 * j + kw];
                if (MAXP) {  // torch's rule: first maximum in row-major window order, NaN wins
                    if (val > acc[j] || val != val || am[j] < 0) {
                        acc[j] = val;
                        am[j] = kh * 3 + kw;
                    }
                } else {
                    acc[j] += val;
                }
            }
    }
    Vec<T, BG> o;
    Vec<uint8_t, BG> ot;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        o.v[j] = (T)(MAXP ? acc[j] : acc[j] * (1.f / 9.f));
        ot.v[j] = (uint8_t)am[j];
    }
    const int64_t oi = pl * Ho * Wo + (int64_t)oh * Wo + ow0;
    stv<T, BG>(y + oi, o);
    if (MAXP && tap) stv<uint8_t, BG>(tap + oi, ot);
}

template <typename T, int BG, bool MAXP>  // backward: a lane = rows 2a, 2a+1, columns 2 b0 .. 2 (b0 + BG) - 1 of the input gradient
__global__ __launch_bounds__(256) void pool3x3s2_bwd_blk(const T* __restrict__ gy, const uint8_t* __restrict__ tap, T* __restrict__ gx,
                                                         int64_t planes, int H, int W, int Ho, int Wo) {
    const int Ha = (H + 1) / 2, NB = Wo / BG;  // (W even: Wo = W / 2 column pairs)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ha * NB) return;
    const int bg = (int)(i % NB), a = (int)((i / NB) % Ha);
    const int64_t pl = i / ((int64_t)NB * Ha);
    const int b0 = bg * BG;
    const T* gp = gy + pl * Ho * Wo;
    const uint8_t* tp = MAXP ? tap + pl * Ho * Wo : nullptr;
    float g[2][BG + 1];
    int t[2][BG + 1];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oh = a + r;
        const bool okr = oh < Ho;
        if (okr) {
            const Vec<T, BG> v = ldv<T, BG>(gp + oh * Wo + b0);
#pragma unroll
            for (int c = 0; c < BG; ++c) g[r][c] = (float)v.v[c];
            g[r][BG] = b0 + BG < Wo ? (float)gp[oh * Wo + b0 + BG] : 0.f;
            if (MAXP) {
                const Vec<uint8_t, BG> tv = ldv<uint8_t, BG>(tp + oh * Wo + b0);
#pragma unroll
                for (int c = 0; c < BG; ++c) t[r][c] = (int)tv.v[c];
                t[r][BG] = b0 + BG < Wo ? (int)tp[oh * Wo + b0 + BG] : -1;
            }
        } else {
#pragma unroll
            for (int c = 0; c <= BG; ++c) {
                g[r][c] = 0.f;
                t[r][c] = -1;
            }
        }
    }
    const bool right_ok = b0 + BG < Wo;  // (column b0 + BG exists)
    Vec<T, 2 * BG> o0, o1;
#pragma unroll
    for (int c = 0; c < BG; ++c) {
        const bool rc = c + 1 < BG || right_ok;  // window column b + 1 exists
        float s00, s01, s10, s11;
        if (MAXP) {  // pixel (2a, 2b) is tap 4 of window (a, b); (2a, 2b+1): tap 5 of (a, b), 3 of (a, b+1); (2a+1, 2b): 7 / 1;
                     // (2a+1, 2b+1): 8 / 6 / 2 / 0 of (a, b) / (a, b+1) / (a+1, b) / (a+1, b+1); windows ascending
            s00 = t[0][c] == 4 ? g[0][c] : 0.f;
            s01 = 0.f; s10 = 0.f; s11 = 0.f;
            if (t[0][c] == 5) s01 += g[0][c];
            if (t[0][c + 1] == 3) s01 += g[0][c + 1];
            if (t[0][c] == 7) s10 += g[0][c];
            if (t[1][c] == 1) s10 += g[1][c];
            if (t[0][c] == 8) s11 += g[0][c];
            if (t[0][c + 1] == 6) s11 += g[0][c + 1];
            if (t[1][c] == 2) s11 += g[1][c];
            if (t[1][c + 1] == 0) s11 += g[1][c + 1];
        } else {  // windows (oh, ow) ascending, as the per-pixel kernel adds them
            s00 = g[0][c];
            s01 = g[0][c];
            if (rc) s01 += g[0][c + 1];
            s10 = g[0][c];
            if (a + 1 < Ho) s10 += g[1][c];
            s11 = g[0][c];
            if (rc) s11 += g[0][c + 1];
            if (a + 1 < Ho) {
                s11 += g[1][c];
                if (rc) s11 += g[1][c + 1];
            }
            s00 *= 1.f / 9.f; s01 *= 1.f / 9.f; s10 *= 1.f / 9.f; s11 *= 1.f / 9.f;
        }
        o0.v[2 * c] = (T)s00; o0.v[2 * c + 1] = (T)s01;
        o1.v[2 * c] = (T)s10; o1.v[2 * c + 1] = (T)s11;
    }
    T* o = gx + pl * H * W + (int64_t)(2 * a) * W + 2 * b0;
    stv<T, 2 * BG>(o, o0);
    if (2 * a + 1 < H) stv<T, 2 * BG>(o + W, o1);
}

// ---- every second pixel of every second row (the input of a stride-2 1x1 projection shortcut, models/resnet.py downsample:
// nn.Conv2d(kernel_size=1, stride=2) reads exactly x[:, :, ::2, ::2]) and its gradient (the values back in place, zeros
// elsewhere) -- as torch ops: a strided copy forward, a fill plus a strided copy backward (35 + 18 + 35 us for 80 x 256 x 56 x 56).
// Even H and W; a lane owns BG output pixels (forward) / a 2 x 2 BG block of the input gradient (backward).
template <typename T, int BG>
__global__ __launch_bounds__(256) void subsample2_fwd_blk(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, NB = Wo / BG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const Vec<T, 2 * BG> v = ldv<T, 2 * BG>(x + pl * H * W + (int64_t)(2 * oh) * W + 2 * bg * BG);
    Vec<T, BG> o;
#pragma unroll
    for (int j = 0; j < BG; ++j) o.v[j] = v.v[2 * j];
    stv<T, BG>(y + pl * Ho * Wo + (int64_t)oh * Wo + bg * BG, o);
}

template <typename T, int BG>
__global__ __launch_bounds__(256) void subsample2_bwd_blk(const T* __restrict__ gy, T* __restrict__ gx, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, NB = Wo / BG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const Vec<T, BG> g = ldv<T, BG>(gy + pl * Ho * Wo + (int64_t)oh * Wo + bg * BG);
    Vec<T, 2 * BG> o0, o1;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        o0.v[2 * j] = g.v[j];
        o0.v[2 * j + 1] = (T)0.f;
        o1.v[2 * j] = (T)0.f;
        o1.v[2 * j + 1] = (T)0.f;
    }
    T* o = gx + pl * H * W + (int64_t)(2 * oh) * W + 2 * bg * BG;
    stv<T, 2 * BG>(o, o0);
    stv<T, 2 * BG>(o + W, o1);
}

template <typename T>
int subsample2(int bwd, const void* a, void* out, int64_t planes, int H, int W, hipStream_t stream) {
    if ((H & 1) || (W & 1)) return COT_ERR_UNSUPPORTED;
    const int Ho = H / 2, Wo = W / 2;
    const bool odd16 = ((H * W) & 7) || ((Ho * Wo) & 7), odd8 = ((H * W) & 3) || ((Ho * Wo) & 3);
    const int BG = (Wo % 4 == 0 && !odd16) ? 4 : ((Wo % 2 == 0 && !odd8) ? 2 : 1);
    const dim3 grid((unsigned)ceil_div64(planes * Ho * (Wo / BG), 256)), block(256);
#define COT_SUB2(BG_)                                                                                                  \
    if (bwd) COT_LAUNCH((subsample2_bwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W);     \
    else COT_LAUNCH((subsample2_fwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W)
    if (BG == 4) { COT_SUB2(4); } else if (BG == 2) { COT_SUB2(2); } else { COT_SUB2(1); }
#undef COT_SUB2
    return check_launch("subsample2");
}
template int subsample2<float>(int, const void*, void*, int64_t, int, int, hipStream_t);
template int subsample2<bf16_t>(int, const void*, void*, int64_t, int, int, hipStream_t);

// ---- nn.AvgPool2d(2, 2) on even planes (the pooling in front of the 1x1 projection of an `avg_down` shortcut, models/resnet.py:
// 377-394 downsample_avg: AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False) -- on even H and W every window is a full
// 2 x 2 block, so neither flag matters).  torch's avg_pool2d_backward took 585 us per launch in SE-CoTNetD-152's step (gpurun_out/
// r4v_secot_per_shape.csv: 4 launches, 2.3 ms) for a broadcast of g / 4.  A lane owns BG outputs (forward: two rows of 2 BG inputs)
// or the 2 x 2 BG input-gradient block under BG outputs (backward).  Sum in torch's order (row-major window), divided by 4.
template <typename T, int BG>
__global__ __launch_bounds__(256) void avgpool2x2s2_fwd_blk(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, NB = Wo / BG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const T* rp = x + pl * H * W + (int64_t)(2 * oh) * W + 2 * bg * BG;
    const Vec<T, 2 * BG> r0 = ldv<T, 2 * BG>(rp), r1 = ldv<T, 2 * BG>(rp + W);
    Vec<T, BG> o;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
        float sum = (float)r0.v[2 * j];
        sum += (float)r0.v[2 * j + 1];
        sum += (float)r1.v[2 * j];
        sum += (float)r1.v[2 * j + 1];
        o.v[j] = (T)(sum * 0.25f);
    }
    stv<T, BG>(y + pl * Ho * Wo + (int64_t)oh * Wo + bg * BG, o);
}

template <typename T, int BG>
__global__ __launch_bounds__(256) void avgpool2x2s2_bwd_blk(const T* __restrict__ gy, T* __restrict__ gx, int64_t planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2, NB = Wo / BG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const Vec<T, BG> g = ldv<T, BG>(gy + pl * Ho * Wo + (int64_t)oh * Wo + bg * BG);
    Vec<T, 2 * BG> o;
#pragma unroll
    for (int j = 0; j < BG; ++j) o.v[2 * j] = o.v[2 * j + 1] = (T)((float)g.v[j] * 0.25f);
    T* op = gx + pl * H * W + (int64_t)(2 * oh) * W + 2 * bg * BG;
    stv<T, 2 * BG>(op, o);
    stv<T, 2 * BG>(op + W, o);
}

template <typename T>
int avgpool2x2s2(int bwd, const void* a, void* out, int64_t planes, int H, int W, hipStream_t stream) {
    if ((H & 1) || (W & 1)) return COT_ERR_UNSUPPORTED;
    const int Ho = H / 2, Wo = W / 2;
    const bool odd16 = ((H * W) & 7) || ((Ho * Wo) & 7) || (W & 7), odd8 = ((H * W) & 3) || ((Ho * Wo) & 3) || (W & 3);
    const int BG = (Wo % 4 == 0 && !odd16) ? 4 : ((Wo % 2 == 0 && !odd8) ? 2 : 1);
    const dim3 grid((unsigned)ceil_div64(planes * Ho * (Wo / BG), 256)), block(256);
#define COT_AVG2(BG_)                                                                                                    \
    if (bwd) COT_LAUNCH((avgpool2x2s2_bwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W);     \
    else COT_LAUNCH((avgpool2x2s2_fwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W)
    if (BG == 4) { COT_AVG2(4); } else if (BG == 2) { COT_AVG2(2); } else { COT_AVG2(1); }
#undef COT_AVG2
    return check_launch("avgpool2x2s2");
}
template int avgpool2x2s2<float>(int, const void*, void*, int64_t, int, int, hipStream_t);
template int avgpool2x2s2<bf16_t>(int, const void*, void*, int64_t, int, int, hipStream_t);

int g_pool_tile = 1;  // cot_set_tuning key 27: 0 = one lane per pixel only
// windows per lane of the row-block form (0: not eligible): W even, every wide access naturally aligned
static int pool_blk_group(int op, int H, int W, int Ho, int Wo) {
    if (!g_pool_tile || !(op == 0 || op == 1 || op == 4 || op == 5 || op == 6 || op == 7) || (W & 1) || W < 2) return 0;
    if ((op == 6 || op == 7) && H < 2) return 0;
    const bool odd_planes = ((H * W) & 7) || ((Ho * Wo) & 7);  // (planes do not all start on 16-byte boundaries: narrower groups)
    if (Wo % 4 == 0 && !odd_planes) return 4;
    if (Wo % 2 == 0 && !(((H * W) & 3) || ((Ho * Wo) & 3))) return 2;
    return (((H * W) & 1) || ((Ho * Wo) & 1)) ? 0 : 1;
}

// ---- BlurPool2d(filt_size = 3, stride = 2): reflection padding by one pixel, depthwise binomial filter [1 2 1] x [1 2 1] / 16,
// stride 2 (the anti-aliased down-sampling of SE-CoTNetD, reference models/layers/blur_pool.py:53-58 -- there a
// ReflectionPad2d + a grouped F.conv2d whose weight is the filter repeated per channel: MIOpen runs a depthwise convolution
// with layout changes for what is a 9-tap stencil).  Output Ho = (H - 1)/2 + 1.  H, W >= 2 (reflection needs a neighbour).
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float binom3(int k) { return k == 1 ? 0.5f : ((k == 0 || k == 2) ? 0.25f : 0.f); }

template <typename T>
__global__ __launch_bounds__(256) void blurpool3x3s2_fwd(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H,
                                                        int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho);
    const int64_t pl = i / ((int64_t)Wo * Ho);
    const T* xp = x + pl * H * W;
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = reflect1(2 * oh - 1 + kh, H);
        float r = 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) r += binom3(kw) * (float)xp[h * W + reflect1(2 * ow - 1 + kw, W)];
        s += binom3(kh) * r;
    }
    y[i] = (T)s;
}

// weight with which output row `o` (of `no`) reads input row `h` (of `n`): the padded positions that reflect onto h are h
// itself, -1 when h == 1 and n when h == n - 2; position q is tap q - (2*o - 1) of window o
__device__ __forceinline__ float blur_w(int h, int o, int n) {
    float w = binom3(h - (2 * o - 1));
    if (h == 1) w += binom3(-1 - (2 * o - 1));
    if (h == n - 2) w += binom3(n - (2 * o - 1));
    return w;
}

// gather form (no atomics): one thread per input pixel sums the windows that read it
template <typename T>
__global__ __launch_bounds__(256) void blurpool3x3s2_bwd(const T* __restrict__ gy, T* __restrict__ gx, int64_t planes,
                                                        int H, int W, int Ho, int Wo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const int64_t pl = i / ((int64_t)W * H);
    const T* gp = gy + pl * Ho * Wo;
    // windows that can touch row h directly are (h-1)/2 .. (h+1)/2; the reflected positions add window 0 (h == 1) and the
    // last window (h == H-2): scan the (at most) four candidates per axis
    float s = 0.f;
    const int o0 = max(0, (h - 1) / 2 - 1), o1 = min(Ho - 1, (h + 1) / 2 + 1);
    const int p0 = max(0, (w - 1) / 2 - 1), p1 = min(Wo - 1, (w + 1) / 2 + 1);
    for (int oh = o0; oh <= o1; ++oh) {
        const float wh = blur_w(h, oh, H);
        if (wh == 0.f) continue;
        float r = 0.f;
        for (int ow = p0; ow <= p1; ++ow) {
            const float ww = blur_w(w, ow, W);
            if (ww != 0.f) r += ww * (float)gp[oh * Wo + ow];
        }
        s += wh * r;
    }
    gx[i] = (T)s;
}

// row-block forms (even W, wide naturally aligned accesses -- pool_blk_group): the per-pixel kernels above pay two 64-bit
// divisions and nine (or up to sixteen) separately addressed 2-byte loads per pixel: 407 us per launch averaged over the four
// BlurPool gradients of SE-CoTNetD-152 at 320 x 320, B = 64 (gpurun_out/r4q_secotnetd_kernels.json; the stem's 64 x 64 x 160 x 160
// plane set alone moves 262 MB = 33 us at the roofline).  Forward: a lane owns BG consecutive outputs of one output row, loads
// 2 BG + 1 columns of each of the three (reflected) input rows.  Backward: a lane owns a 2 x 2 BG block of the input gradient
// (rows 2a, 2a+1), loads BG + 1 columns of gradient rows a, a+1.  W even means no reflection at the right edge (the last
// window ends at column W - 1); the left one (column -1 -> 1) and both vertical ones are handled.  Sums in the order of the
// per-pixel kernels (rows outer, columns inner, ascending): identical results.
template <typename T, int BG>
__global__ __launch_bounds__(256) void blurpool3x3s2_fwd_blk(const T* __restrict__ x, T* __restrict__ y, int64_t planes, int H,
                                                            int W, int Ho, int Wo) {
    const int NB = Wo / BG;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * NB) return;
    const int bg = (int)(i % NB), oh = (int)((i / NB) % Ho);
    const int64_t pl = i / ((int64_t)NB * Ho);
    const int ow0 = bg * BG;
    const T* xp = x + pl * H * W;
    float acc[BG];
#pragma unroll
    for (int j = 0; j < BG; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = reflect1(2 * oh - 1 + kh, H);
        const T* rp = xp + (int64_t)h * W + 2 * ow0;
        float L[2 * BG + 1];
        L[0] = (float)(ow0 > 0 ? rp[-1] : rp[1]);  // column -1 reflects onto column 1
        const Vec<T, 2 * BG> v = ldv<T, 2 * BG>(rp);
#pragma unroll
        for (int c = 0; c < 2 * BG; ++c) L[1 + c] = (float)v.v[c];
#pragma unroll
        for (int j = 0; j < BG; ++j) {
            float r = 0.f;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) r += binom3(kw) * L[2 * j + kw];
            acc[j] += binom3(kh) * r;
        }
    }
    Vec<T, BG> o;
#pragma unroll
    for (int j = 0; j < BG; ++j) o.v[j] = (T)acc[j];
    stv<T, BG>(y + pl * Ho * Wo + (int64_t)oh * Wo + ow0, o);
}

template <typename T, int BG>
__global__ __launch_bounds__(256) void blurpool3x3s2_bwd_blk(const T* __restrict__ gy, T* __restrict__ gx, int64_t planes, int H,
                                                            int W, int Ho, int Wo) {
    const int NB = Wo / BG, HB = (H + 1) / 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * HB * NB) return;
    const int bg = (int)(i % NB), a = (int)((i / NB) % HB);
    const int64_t pl = i / ((int64_t)NB * HB);
    const int b0 = bg * BG;
    const T* gp = gy + pl * Ho * Wo;
    // rows 2a and 2a+1 read windows a and a+1 only (blur_w is zero elsewhere: direct taps reach (h-1)/2 .. (h+1)/2, the top
    // reflection adds window 0 to row 1, the bottom one the last window to row H-2 -- both inside {a, a+1})
    float g[2][BG + 1];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oh = a + r;
        if (oh < Ho) {
            const Vec<T, BG> v = ldv<T, BG>(gp + (int64_t)oh * Wo + b0);
#pragma unroll
            for (int c = 0; c < BG; ++c) g[r][c] = (float)v.v[c];
            g[r][BG] = b0 + BG < Wo ? (float)gp[(int64_t)oh * Wo + b0 + BG] : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c <= BG; ++c) g[r][c] = 0.f;
        }
    }
    Vec<T, 2 * BG> o[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int h = 2 * a + rr;
        float wh[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) wh[r] = (h < H && a + r < Ho) ? blur_w(h, a + r, H) : 0.f;
#pragma unroll
        for (int c = 0; c < BG; ++c) {
            // column 2(b0+c): window b0+c (weight 1/2); column 2(b0+c)+1: windows b0+c (1/4, + 1/4 reflected when it is column 1)
            // and b0+c+1 (1/4, when it exists)
            const float we = 0.5f, wo0 = (b0 + c == 0) ? 0.5f : 0.25f, wo1 = (b0 + c + 1 < Wo) ? 0.25f : 0.f;
            float se = 0.f, so = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (wh[r] == 0.f) continue;
                float re = 0.f, ro = 0.f;
                re += we * g[r][c];
                ro += wo0 * g[r][c];
                if (wo1 != 0.f) ro += wo1 * g[r][c + 1];
                se += wh[r] * re;
                so += wh[r] * ro;
            }
            o[rr].v[2 * c] = (T)se;
            o[rr].v[2 * c + 1] = (T)so;
        }
    }
    T* op = gx + pl * H * W + (int64_t)(2 * a) * W + 2 * b0;
    stv<T, 2 * BG>(op, o[0]);
    if (2 * a + 1 < H) stv<T, 2 * BG>(op + W, o[1]);
}

template <typename T>
int pool3x3s2(int op, const void* a, const void* b, void* out, int64_t planes, int H, int W, hipStream_t stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;  // floor((H + 2 - 3) / 2) + 1
    const int64_t n_out = planes * Ho * Wo, n_in = planes * H * W;
    const dim3 block(256);
    if (const int BG = pool_blk_group(op, H, W, Ho, Wo)) {
        const bool bwd = op == 1 || op == 5 || op == 7;
        const int64_t lanes = planes * (bwd ? (H + 1) / 2 : Ho) * (Wo / BG);
        const dim3 grid((unsigned)ceil_div64(lanes, 256));
#define COT_POOL_BLK(BG_)                                                                                                       \
    switch (op) {                                                                                                              \
        case 0: COT_LAUNCH((pool3x3s2_fwd_blk<T, BG_, false>), grid, block, 0, stream, (const T*)a, (T*)out, nullptr, planes, H, W, Ho, Wo); break; \
        case 4: COT_LAUNCH((pool3x3s2_fwd_blk<T, BG_, true>), grid, block, 0, stream, (const T*)a, (T*)out, (uint8_t*)const_cast<void*>(b), planes, H, W, Ho, Wo); break; \
        case 1: COT_LAUNCH((pool3x3s2_bwd_blk<T, BG_, false>), grid, block, 0, stream, (const T*)a, nullptr, (T*)out, planes, H, W, Ho, Wo); break; \
        case 6: COT_LAUNCH((blurpool3x3s2_fwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break; \
        case 7: COT_LAUNCH((blurpool3x3s2_bwd_blk<T, BG_>), grid, block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break; \
        default: COT_LAUNCH((pool3x3s2_bwd_blk<T, BG_, true>), grid, block, 0, stream, (const T*)a, (const uint8_t*)b, (T*)out, planes, H, W, Ho, Wo); break; \
    }
        if (BG == 4) { COT_POOL_BLK(4) } else if (BG == 2) { COT_POOL_BLK(2) } else { COT_POOL_BLK(1) }
#undef COT_POOL_BLK
        return check_launch("pool3x3s2 (row blocks)");
    }
    switch (op) {
        case 0: COT_LAUNCH((avgpool3x3s2_fwd<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 1: COT_LAUNCH((avgpool3x3s2_bwd<T>), dim3((unsigned)ceil_div64(n_in, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 2: COT_LAUNCH((maxpool3x3s2_fwd<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 6: COT_LAUNCH((blurpool3x3s2_fwd<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 7: COT_LAUNCH((blurpool3x3s2_bwd<T>), dim3((unsigned)ceil_div64(n_in, 256)), block, 0, stream, (const T*)a, (T*)out, planes, H, W, Ho, Wo); break;
        case 4: COT_LAUNCH((maxpool3x3s2_fwd_tap<T>), dim3((unsigned)ceil_div64(n_out, 256)), block, 0, stream, (const T*)a, (T*)out, (uint8_t*)const_cast<void*>(b), planes, H, W, Ho, Wo); break;
        case 5: COT_LAUNCH((maxpool3x3s2_bwd_tap<T>), dim3((unsigned)ceil_div64(planes * ((H + 1) / 2) * ((W + 1) / 2), 256)), block, 0, stream, (const T*)a, (const uint8_t*)b, (T*)out, planes, H, W, Ho, Wo); break;
        default: COT_LAUNCH((maxpool3x3s2_bwd<T>), dim3((unsigned)ceil_div64(planes * ((H + 1) / 2) * ((W + 1) / 2), 256)), block, 0, stream, (const T*)a, (const T*)b, (T*)out, planes, H, W, Ho, Wo); break;
    }
    return check_launch("pool3x3s2");
}
template int pool3x3s2<float>(int, const void*, const void*, void*, int64_t, int, int, hipStream_t);
template int pool3x3s2<bf16_t>(int, const void*, const void*, void*, int64_t, int, int, hipStream_t);

}  // namespace cot
