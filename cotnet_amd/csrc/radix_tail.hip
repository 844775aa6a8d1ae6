// radix_tail.hip -- the radix-2 split-attention tail of the CoT layer, NCHW, gfx950.
//
// Reference (models/cotnet.py:92-104): x,k -> view/cat to [B,C,2,H,W] -> sum(dim=2) -> mean(H,W) -> se -> softmax over the
// radix pair -> (x * attn).sum(dim=2): ~12 C*H*W-sized tensor passes for ~4 flop per element.  Here:
//   radix_gap      gap[b,c]  = mean_hw(y + k)                                   reads y,k once
//   radix_mix      out       = y*a0[b,c] + k*a1[b,c]                            reads y,k once, writes out
//   radix_mix_bwd  gy = g*a0, gk = g*a1, ga0[b,c] = sum_hw g*y, ga1 = sum_hw g*k one pass over g,y,k
// (the tiny se MLP + softmax between gap and mix stay in torch).  One wavefront per (b,c) plane: the per-plane scalars
// are wave-uniform, plane rows are contiguous, loads are V-wide vectors when H*W allows.
#include "cot_common.h"

namespace cot {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_gap_kernel(const T* __restrict__ y, const T* __restrict__ k,
                                                       T* __restrict__ gap, int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;  // whole wave exits together (plane is wave-uniform)
    const T* yp = y + plane * HW;
    const T* kp = k + plane * HW;
    float acc = 0.f;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> a = ldv<T, V>(yp + i), b = ldv<T, V>(kp + i);
#pragma unroll
        for (int j = 0; j < V; ++j) acc += (float)a.v[j] + (float)b.v[j];
    }
    acc = wave_sum_f(acc);
    if (lane == 0) gap[plane] = (T)(acc / (float)HW);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_mix_kernel(const T* __restrict__ y, const T* __restrict__ k,
                                                       const T* __restrict__ attn, T* __restrict__ out,
                                                       int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;
    const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
    const T* yp = y + plane * HW;
    const T* kp = k + plane * HW;
    T* op = out + plane * HW;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> a = ldv<T, V>(yp + i), b = ldv<T, V>(kp + i);
        Vec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = (T)((float)a.v[j] * a0 + (float)b.v[j] * a1);
        stv<T, V>(op + i, o);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_mix_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                           const T* __restrict__ k, const T* __restrict__ attn,
                                                           T* __restrict__ gy, T* __restrict__ gk,
                                                           T* __restrict__ gattn, int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;
    const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
    const int64_t base = plane * HW;
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> gv = ldv<T, V>(g + base + i), a = ldv<T, V>(y + base + i), b = ldv<T, V>(k + base + i);
        Vec<T, V> oy, ok;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gv.v[j];
            oy.v[j] = (T)(gg * a0);
            ok.v[j] = (T)(gg * a1);
            s0 += gg * (float)a.v[j];
            s1 += gg * (float)b.v[j];
        }
        stv<T, V>(gy + base + i, oy);
        stv<T, V>(gk + base + i, ok);
    }
    s0 = wave_sum_f(s0);
    s1 = wave_sum_f(s1);
    if (lane == 0) {
        gattn[plane * 2] = (T)s0;
        gattn[plane * 2 + 1] = (T)s1;
    }
}


// ---- channel-major ("T") variants: the se branch between gap and mix is a two-layer MLP on [N, C]; with the
// descriptors stored [C][N] it is two 1x1 convolutions over ONE image of N "pixels" (cot_conv1x1_*) with a BatchNorm
// between them (cot_bn_act_*, statistics over the N pixels = over the batch, as nn.BatchNorm2d on [N, A, 1, 1] does).
//   radix_gap_t          gapT[c][n] = mean_hw(y + k)
//   radix_mix_logits     a0 = softmax over the pair (logitsT[2c][n], logitsT[2c+1][n]); out = y*a0 + k*a1; attn saved
//   radix_mix_bwd_reduce s0 = sum g*y, s1 = sum g*k;  glogitsT[2c][n] = a0*a1*(s0 - s1) = -glogitsT[2c+1][n]
//   radix_mix_bwd_apply  gy = g*a0 + ggapT[c][n]/HW,  gk = g*a1 + ggapT[c][n]/HW
// per-tensor layout bit (round 5): 0 = plane (n, c) at (n*C + c)*HW (NCHW), 1 = at (c*N + n)*HW (channel-major)
__device__ __forceinline__ int64_t tail_base(int cm, int n, int c, int N, int C, int HW) {
    return ((int64_t)(cm ? c * N + n : n * C + c)) * HW;
}

// SEG < 64 (round 5): SEG lanes per plane, 64 / SEG planes per wave -- 7 x 7 planes as 7 lanes x 7 elements (V = 7, SEG = 8) instead of
// 49 lanes x one 2-byte element and a wave per plane (40960 waves for 80 x 512 planes).  Sums: xor butterfly over the segment (every lane
// ends with the total).  No early return in the kernels that shuffle: lanes of planes past the end follow along and store nothing.
// y*a0 + k*a1 with the contraction spelled out (one rounding of k*a1, one of the sum): the radix mix and its BatchNorm-folded twin
// must agree bit for bit, and left to the compiler the two kernels contracted differently (263 of 16 M outputs one bf16 ulp apart)
__device__ __forceinline__ float mix2(float y, float a0, float k, float a1) { return __builtin_fmaf(y, a0, k * a1); }

template <int SEG> __device__ __forceinline__ float seg_sum_f(float v) {
    if (SEG == 64) return wave_sum_f(v);
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_gap_t_kernel(const T* __restrict__ y, const T* __restrict__ k,
                                                         T* __restrict__ gapT, int N, int C, int HW, int lay) {
    const int lane = threadIdx.x & (SEG - 1);
    int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const bool live = plane < (int64_t)N * C;
    if (SEG == 64 && !live) return;
    if (!live) plane = (int64_t)N * C - 1;
    const int n = (int)(plane / C), c = (int)(plane % C);
    const T* yp = y + tail_base(lay & 1, n, c, N, C, HW);
    float acc = 0.f;
    if (k) {
        const T* kp = k + tail_base(lay & 2, n, c, N, C, HW);
        for (int i = lane * V; i < HW; i += SEG * V) {
            const Vec<T, V> a = ldv<T, V>(yp + i), b = ldv<T, V>(kp + i);
#pragma unroll
            for (int j = 0; j < V; ++j) acc += (float)a.v[j] + (float)b.v[j];
        }
    } else {  // plain global average pooling (the classifier head)
        for (int i = lane * V; i < HW; i += SEG * V) {
            const Vec<T, V> a = ldv<T, V>(yp + i);
#pragma unroll
            for (int j = 0; j < V; ++j) acc += (float)a.v[j];
        }
    }
    acc = seg_sum_f<SEG>(acc);
    if (lane == 0 && live) gapT[(int64_t)c * N + n] = (T)(acc / (float)HW);
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_logits_kernel(const T* __restrict__ y, const T* __restrict__ k,
                                                              const T* __restrict__ logitsT, T* __restrict__ out,
                                                              T* __restrict__ attn, int N, int C, int HW, int lay) {
    const int lane = threadIdx.x & (SEG - 1);
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    if (plane >= (int64_t)N * C) return;  // (no shuffles below)
    const int n = (int)(plane / C), c = (int)(plane % C);
    const float l0 = (float)logitsT[(int64_t)(2 * c) * N + n], l1 = (float)logitsT[(int64_t)(2 * c + 1) * N + n];
    const float a0 = 1.f / (1.f + __expf(l1 - l0)), a1 = 1.f - a0;  // softmax over the radix pair
    if (lane == 0) {
        attn[plane * 2] = (T)a0;
        attn[plane * 2 + 1] = (T)a1;
    }
    const T* yp = y + tail_base(lay & 1, n, c, N, C, HW);
    const T* kp = k + tail_base(lay & 2, n, c, N, C, HW);
    T* op = out + tail_base(lay & 4, n, c, N, C, HW);
    for (int i = lane * V; i < HW; i += SEG * V) {
        const Vec<T, V> a = ldv<T, V>(yp + i), b = ldv<T, V>(kp + i);
        Vec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = (T)mix2((float)a.v[j], a0, (float)b.v[j], a1);
        stv<T, V>(op + i, o);
    }
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                                  const T* __restrict__ k, const T* __restrict__ attn,
                                                                  T* __restrict__ glogitsT, int N, int C, int HW, int lay) {
    const int lane = threadIdx.x & (SEG - 1);
    int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const bool live = plane < (int64_t)N * C;
    if (SEG == 64 && !live) return;
    if (!live) plane = (int64_t)N * C - 1;
    const int n = (int)(plane / C), c = (int)(plane % C);
    const T* gp = g + tail_base(lay & 1, n, c, N, C, HW);
    const T* yp = y + tail_base(lay & 2, n, c, N, C, HW);
    const T* kp = k + tail_base(lay & 4, n, c, N, C, HW);
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane * V; i < HW; i += SEG * V) {
        const Vec<T, V> gv = ldv<T, V>(gp + i), a = ldv<T, V>(yp + i), b = ldv<T, V>(kp + i);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gv.v[j];
            s0 += gg * (float)a.v[j];
            s1 += gg * (float)b.v[j];
        }
    }
    s0 = seg_sum_f<SEG>(s0);
    s1 = seg_sum_f<SEG>(s1);
    if (lane == 0 && live) {
        const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
        const float gl = a0 * a1 * (s0 - s1);  // softmax backward for a pair: gl0 = a0*(s0 - (a0*s0 + a1*s1))
        glogitsT[(int64_t)(2 * c) * N + n] = (T)gl;
        glogitsT[(int64_t)(2 * c + 1) * N + n] = (T)(-gl);
    }
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ attn,
                                                                 const T* __restrict__ ggapT, T* __restrict__ gy,
                                                                 T* __restrict__ gk, int N, int C, int HW, int lay) {
    const int lane = threadIdx.x & (SEG - 1);
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    if (plane >= (int64_t)N * C) return;  // (no shuffles below)
    const int n = (int)(plane / C), c = (int)(plane % C);
    const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
    const float add = (float)ggapT[(int64_t)c * N + n] / (float)HW;  // d mean_hw(y + k)
    const T* gp = g + tail_base(lay & 1, n, c, N, C, HW);
    T* gyp = gy + tail_base(lay & 2, n, c, N, C, HW);
    T* gkp = gk + tail_base(lay & 4, n, c, N, C, HW);
    for (int i = lane * V; i < HW; i += SEG * V) {
        const Vec<T, V> gv = ldv<T, V>(gp + i);
        Vec<T, V> oy, ok;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gv.v[j];
            oy.v[j] = (T)(gg * a0 + add);
            ok.v[j] = (T)(gg * a1 + add);
        }
        stv<T, V>(gyp + i, oy);
        stv<T, V>(gkp + i, ok);
    }
}

// ---- BatchNorm + SiLU of the aggregation's output folded into the tail (round 6; VERDICT r5 J1 / next #2b; models/cotnet.py:88-104:
// x = local_conv(x, w); x = bn(x); x = act(x); then the radix tail).  The normalised, activated tensor y = silu(bn(a)) has exactly two
// readers, both in this file, and both read it plane by plane with the channel wave-uniform: they take the RAW aggregation output `a`
// and the BatchNorm's statistics and form y as they load -- y is never written, the BatchNorm's apply pass (1 read + 1 write) and the
// tail's two reads of y become two reads of a.  Backward: what BatchNorm needs per channel, sum g_z and sum g_z * xhat with
// g_z = (g*a0 + add) * silu'(z), splits into four plane sums that do not depend on `add` (the gradient of the pooled descriptor, known
// only after the se branch's backward): t0 = sum g*s', t1 = sum s', t2 = sum g*s'*xhat, t3 = sum s'*xhat.  The reduce kernel -- which
// reads g and a anyway -- emits them; the apply kernel's prologue combines them over the batch (fixed order: deterministic) and writes
// the gradient w.r.t. a directly.  4 launches and 11 tensor passes (mix apply, BatchNorm reduce, BatchNorm apply, + the reduce) become 2
// launches and 7 passes.
//   radix_gap_t_bn        gapT[c][n] = mean_hw(silu(bn(a)) + k); given bn_stats_sums' chunk sums its prologue also finalizes the statistics
//   radix_mix_logits_bn   out = silu(bn(a))*a0 + k*a1
//   radix_mix_bwd_reduce_bn  glogitsT as radix_mix_bwd_reduce + tsum[n][c][0..3]
//   radix_mix_bwd_apply_bn   ga = gamma*rstd*(g_z - dbeta/M - xhat*dgamma/M), gk = g*a1 + add; dgamma / dbeta written by plane n = 0
// z = a*sc + sh with sc = gamma*rstd, sh = beta - mean*sc, y rounded to the storage type: bit-identical to bn_apply_fwd{,_fold}.
struct BnTail {
    const float* gamma;
    const float* beta;
    const float* mean;
    const float* rstd;
};

// all-lanes sum over a segment of SEG lanes (xor butterfly; SEG = 64: the whole wave)
template <int SEG> __device__ __forceinline__ float seg_allsum_f(float v) {
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Loads run two vectors ahead of the arithmetic (exp + rcp per element: ~80 instructions per 8 elements -- with one vector in flight per
// lane and tensor the kernels added their VALU time to their memory time: gap_t_bn 24.4 us against gap_t's 11.2 at 56 x 56,
// profiles/r06_bn_tail_kernels.log), and the first two are issued before the per-plane scalars are fetched.
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ldv_if(const T* p, bool ok) {
    Vec<T, V> v;
    if (ok) v = ldv<T, V>(p);
    else {
#pragma unroll
        for (int j = 0; j < V; ++j) v.v[j] = (T)0.f;
    }
    return v;
}
template <typename T, int V, int STEP> struct Ahead2 {  // vectors i, i + STEP, i + 2 STEP of one plane tensor
    const T* p;
    int HW;
    Vec<T, V> v0, v1;
    __device__ __forceinline__ Ahead2(const T* p_, int i, int HW_) : p(p_), HW(HW_) {
        v0 = ldv_if<T, V>(p + i, i < HW);
        v1 = ldv_if<T, V>(p + i + STEP, i + STEP < HW);
    }
    __device__ __forceinline__ Vec<T, V> next(int i) {  // -> vector i; vector i + 2 STEP goes in flight
        const Vec<T, V> cur = v0;
        v0 = v1;
        v1 = ldv_if<T, V>(p + i + 2 * STEP, i + 2 * STEP < HW);
        return cur;
    }
};

// part != NULL: the prologue FINALIZES the statistics -- it adds the channel's chunk sums (bn_stats_sums, bn_act.hip: three additions per
// chunk), and the wave of plane n = 0 writes mean / rstd / the running statistics for everybody behind this launch.  NULL: bn.mean /
// bn.rstd are read.
struct BnFin {
    const float* part;  // [C][split][4] = (sum (x - shift), sum (x - shift)^2, count, shift)
    int split;
    float eps, momentum;
    float* mean;
    float* rstd;
    float* running_mean;
    float* running_var;
    long long* nbt;
};

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_gap_t_bn_kernel(const T* __restrict__ a, const T* __restrict__ k, T* __restrict__ gapT,
                                                            BnTail bn, BnFin fin, int N, int C, int HW, int lay) {
    constexpr int STEP = SEG * V;
    const int lane = threadIdx.x & (SEG - 1);
    int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const bool live = plane < (int64_t)N * C;
    if (SEG == 64 && !live) return;
    if (!live) plane = (int64_t)N * C - 1;
    const int n = (int)(plane / C), c = (int)(plane % C);
    const int i0 = lane * V;
    Ahead2<T, V, STEP> av(a + tail_base(lay & 1, n, c, N, C, HW), i0, HW), kv(k + tail_base(lay & 2, n, c, N, C, HW), i0, HW);
    float m, r;
    if (fin.part) {
        float s1 = 0.f, s2 = 0.f, cnt = 0.f;
        const float* p = fin.part + (int64_t)c * fin.split * 4;
        for (int q = 0; q < fin.split; ++q) {
            const Vec<float, 4> t = ldv<float, 4>(p + q * 4);
            s1 += t.v[0];
            s2 += t.v[1];
            cnt += t.v[2];
        }
        const float inv = 1.f / cnt, d = s1 * inv;
        float var = s2 * inv - d * d;
        var = var > 0.f ? var : 0.f;
        m = p[3] + d;
        r = 1.0f / sqrtf(var + fin.eps);
        if (n == 0 && lane == 0 && live) {
            fin.mean[c] = m;
            fin.rstd[c] = r;
            if (fin.running_mean) {
                const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
                fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * m;
                fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * unbiased;
            }
            if (c == 0 && fin.nbt) *fin.nbt += 1;
        }
    } else {
        m = bn.mean[c];
        r = bn.rstd[c];
    }
    const float sc = bn.gamma[c] * r, sh = bn.beta[c] - m * sc;
    float acc = 0.f;
    for (int i = i0; i < HW; i += STEP) {
        const Vec<T, V> x = av.next(i), b = kv.next(i);
#pragma unroll
        for (int j = 0; j < V; ++j) acc += (float)(T)silu_fwd((float)x.v[j] * sc + sh) + (float)b.v[j];
    }
    acc = seg_sum_f<SEG>(acc);
    if (lane == 0 && live) gapT[(int64_t)c * N + n] = (T)(acc / (float)HW);
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_logits_bn_kernel(const T* __restrict__ a, const T* __restrict__ k,
                                                                 const T* __restrict__ logitsT, T* __restrict__ out,
                                                                 T* __restrict__ attn, BnTail bn, int N, int C, int HW, int lay) {
    constexpr int STEP = SEG * V;
    const int lane = threadIdx.x & (SEG - 1);
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    if (plane >= (int64_t)N * C) return;  // (no shuffles below)
    const int n = (int)(plane / C), c = (int)(plane % C);
    const int i0 = lane * V;
    Ahead2<T, V, STEP> av(a + tail_base(lay & 1, n, c, N, C, HW), i0, HW), kv(k + tail_base(lay & 2, n, c, N, C, HW), i0, HW);
    const float l0 = (float)logitsT[(int64_t)(2 * c) * N + n], l1 = (float)logitsT[(int64_t)(2 * c + 1) * N + n];
    const float a0 = 1.f / (1.f + __expf(l1 - l0)), a1 = 1.f - a0;
    if (lane == 0) {
        attn[plane * 2] = (T)a0;
        attn[plane * 2 + 1] = (T)a1;
    }
    const float sc = bn.gamma[c] * bn.rstd[c], sh = bn.beta[c] - bn.mean[c] * sc;
    T* op = out + tail_base(lay & 4, n, c, N, C, HW);
    for (int i = i0; i < HW; i += STEP) {
        const Vec<T, V> x = av.next(i), b = kv.next(i);
        Vec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = (T)mix2((float)(T)silu_fwd((float)x.v[j] * sc + sh), a0, (float)b.v[j], a1);
        stv<T, V>(op + i, o);
    }
}

// tsum: [C][N][4] = (a0*t0, t1, a0*t2, t3) -- a channel's N records are one contiguous run (the apply kernel's prologue reads all of
// them in every plane of the channel: stored [N][C] and with a0 fetched from attn that was 128 scattered sectors per wave, 105 MB of L2
// traffic at 14 x 14 against 32 MB of tensors)
template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_bwd_reduce_bn_kernel(const T* __restrict__ g, const T* __restrict__ a,
                                                                     const T* __restrict__ k, const T* __restrict__ attn,
                                                                     T* __restrict__ glogitsT, float* __restrict__ tsum, BnTail bn,
                                                                     int N, int C, int HW, int lay) {
    constexpr int STEP = SEG * V;
    const int lane = threadIdx.x & (SEG - 1);
    int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const bool live = plane < (int64_t)N * C;
    if (SEG == 64 && !live) return;
    if (!live) plane = (int64_t)N * C - 1;
    const int n = (int)(plane / C), c = (int)(plane % C);
    const int i0 = lane * V;
    Ahead2<T, V, STEP> gv(g + tail_base(lay & 1, n, c, N, C, HW), i0, HW), av(a + tail_base(lay & 2, n, c, N, C, HW), i0, HW),
        kv(k + tail_base(lay & 4, n, c, N, C, HW), i0, HW);
    const float m = bn.mean[c], r = bn.rstd[c];
    const float sc = bn.gamma[c] * r, sh = bn.beta[c] - m * sc;
    float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int i = i0; i < HW; i += STEP) {
        const Vec<T, V> gq = gv.next(i), xq = av.next(i), b = kv.next(i);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gq.v[j], x = (float)xq.v[j];
            const float z = x * sc + sh, sg = COT_RCP(1.f + __expf(-z));
            const float y = (float)(T)(z * sg);            // the forward's y, as it was formed and rounded there (silu_fwd)
            const float sp = sg * (1.f + z * (1.f - sg));  // silu'(z)
            const float xh = (x - m) * r;
            const float gs = gg * sp;
            s0 += gg * y;
            s1 += gg * (float)b.v[j];
            t0 += gs;
            t1 += sp;
            t2 += gs * xh;
            t3 += sp * xh;
        }
    }
    s0 = seg_sum_f<SEG>(s0);
    s1 = seg_sum_f<SEG>(s1);
    t0 = seg_sum_f<SEG>(t0);
    t1 = seg_sum_f<SEG>(t1);
    t2 = seg_sum_f<SEG>(t2);
    t3 = seg_sum_f<SEG>(t3);
    if (lane == 0 && live) {
        const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
        const float gl = a0 * a1 * (s0 - s1);
        glogitsT[(int64_t)(2 * c) * N + n] = (T)gl;
        glogitsT[(int64_t)(2 * c + 1) * N + n] = (T)(-gl);
        Vec<float, 4> tv;
        tv.v[0] = a0 * t0;
        tv.v[1] = t1;
        tv.v[2] = a0 * t2;
        tv.v[3] = t3;
        stv<float, 4>(tsum + ((int64_t)c * N + n) * 4, tv);
    }
}

template <typename T, int V, int SEG = 64>
__global__ __launch_bounds__(256) void radix_mix_bwd_apply_bn_kernel(const T* __restrict__ g, const T* __restrict__ a,
                                                                    const T* __restrict__ attn, const T* __restrict__ ggapT,
                                                                    const float* __restrict__ tsum, T* __restrict__ ga,
                                                                    T* __restrict__ gk, BnTail bn, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int N, int C, int HW, int lay) {
    constexpr int STEP = SEG * V;
    const int lane = threadIdx.x & (SEG - 1);
    int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const bool live = plane < (int64_t)N * C;
    if (SEG == 64 && !live) return;
    if (!live) plane = (int64_t)N * C - 1;
    const int n = (int)(plane / C), c = (int)(plane % C);
    const int i0 = lane * V;
    Ahead2<T, V, STEP> gv(g + tail_base(lay & 1, n, c, N, C, HW), i0, HW), av(a + tail_base(lay & 2, n, c, N, C, HW), i0, HW);
    T* gap_ = ga + tail_base(lay & 2, n, c, N, C, HW);
    T* gkp = gk + tail_base(lay & 4, n, c, N, C, HW);
    const float inv_hw = COT_RCP((float)HW);  // (1 ulp: the pooled descriptor's gradient per pixel)
    // the channel's BatchNorm sums over the batch: sum_n (a0*t0 + add*t1), sum_n (a0*t2 + add*t3); every plane of a channel forms the
    // same two numbers in the same order
    float sb = 0.f, sgm = 0.f;
    for (int q = lane; q < N; q += SEG) {
        const Vec<float, 4> t = ldv<float, 4>(tsum + ((int64_t)c * N + q) * 4);
        const float qadd = (float)ggapT[(int64_t)c * N + q] * inv_hw;
        sb += t.v[0] + qadd * t.v[1];
        sgm += t.v[2] + qadd * t.v[3];
    }
    sb = seg_allsum_f<SEG>(sb);
    sgm = seg_allsum_f<SEG>(sgm);
    if (n == 0 && lane == 0 && live) {
        dbeta[c] = sb;
        dgamma[c] = sgm;
    }
    const float inv_m = COT_RCP((float)N * (float)HW);
    const float m = bn.mean[c], r = bn.rstd[c], gam = bn.gamma[c];
    const float sc = gam * r, sh = bn.beta[c] - m * sc;
    const float k1 = sb * inv_m, k2 = sgm * inv_m;
    const float a0 = (float)attn[plane * 2], a1 = (float)attn[plane * 2 + 1];
    const float add = (float)ggapT[(int64_t)c * N + n] * inv_hw;
    for (int i = i0; i < HW; i += STEP) {
        const Vec<T, V> gq = gv.next(i), xq = av.next(i);
        Vec<T, V> oa, ok;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gq.v[j], x = (float)xq.v[j];
            const float z = x * sc + sh, sg = COT_RCP(1.f + __expf(-z));
            const float sp = sg * (1.f + z * (1.f - sg));
            const float xh = (x - m) * r;
            const float gz = (gg * a0 + add) * sp;
            oa.v[j] = (T)(sc * (gz - k1 - xh * k2));
            ok.v[j] = (T)(gg * a1 + add);
        }
        if (live) {
            stv<T, V>(gap_ + i, oa);
            stv<T, V>(gkp + i, ok);
        }
    }
}

// ---- SE-style sigmoid gate of SplitAttnConv2d(radix = 1) (reference models/layers/split_attn.py:62-88 as
// SE-CoTNetD uses it, models/cotnet_hybrid.py:143-146): out = x * sigmoid(logit[n][c]) with logit = fc2(relu(bn1(fc1(
// mean_hw(x))))).  Three HBM-bound kernels around the tiny MLP, one wave per (image, channel) plane:
//   se_gap        gap[plane] = mean_hw(x)
//   se_gate       out = x * sigmoid(logit[plane])
//   se_gate_bwd   gx = g * sigmoid(logit);  glogit[plane] = sigmoid' * sum_hw g * x     (one pass over g and x)
// (the reference runs adaptive_avg_pool2d, sigmoid, a broadcast multiply and their three backward kernels)
template <typename T, int V>
__global__ __launch_bounds__(256) void se_gap_kernel(const T* __restrict__ x, T* __restrict__ gap, int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;
    const T* xp = x + plane * HW;
    float acc = 0.f;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> a = ldv<T, V>(xp + i);
#pragma unroll
        for (int j = 0; j < V; ++j) acc += (float)a.v[j];
    }
    acc = wave_sum_f(acc);
    if (lane == 0) gap[plane] = (T)(acc / (float)HW);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void se_gate_kernel(const T* __restrict__ x, const T* __restrict__ logit,
                                                     T* __restrict__ out, int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;
    const float a = 1.f / (1.f + __expf(-(float)logit[plane]));
    const T* xp = x + plane * HW;
    T* op = out + plane * HW;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> v = ldv<T, V>(xp + i);
        Vec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = (T)((float)v.v[j] * a);
        stv<T, V>(op + i, o);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                         const T* __restrict__ logit, T* __restrict__ gx,
                                                         T* __restrict__ glogit, int64_t planes, int HW) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (plane >= planes) return;
    const float a = 1.f / (1.f + __expf(-(float)logit[plane]));
    const int64_t base = plane * HW;
    float s = 0.f;
    for (int i = lane * V; i < HW; i += 64 * V) {
        const Vec<T, V> gv = ldv<T, V>(g + base + i), xv = ldv<T, V>(x + base + i);
        Vec<T, V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (float)gv.v[j];
            o.v[j] = (T)(gg * a);
            s += gg * (float)xv.v[j];
        }
        stv<T, V>(gx + base + i, o);
    }
    s = wave_sum_f(s);
    if (lane == 0) glogit[plane] = (T)(s * a * (1.f - a));
}

static inline int tail_vec(size_t esize, int HW) {
    int lim = (int)(16 / esize);
    for (int V = 8; V >= 1; V >>= 1)
        if (V <= lim && HW % V == 0) return V;
    return 1;
}

#define TAIL_DISPATCH(KERNEL, ...)                                                                                 \
    do {                                                                                                           \
        const dim3 grid((unsigned)ceil_div64(planes, 4)), block(256);                                              \
        const int v = tail_vec(sizeof(T), HW);                                                                     \
        if (v == 8) COT_LAUNCH((KERNEL<T, (sizeof(T) <= 2 ? 8 : 1)>), grid, block, 0, s, __VA_ARGS__);     \
        else if (v == 4) COT_LAUNCH((KERNEL<T, (sizeof(T) <= 4 ? 4 : 1)>), grid, block, 0, s, __VA_ARGS__);\
        else if (v == 2) COT_LAUNCH((KERNEL<T, 2>), grid, block, 0, s, __VA_ARGS__);                       \
        else COT_LAUNCH((KERNEL<T, 1>), grid, block, 0, s, __VA_ARGS__);                                   \
    } while (0)

int g_radix_pack7 = 1;  // cot_set_tuning key 50: 7 x 7 planes of bf16 -- 8 planes per wave, 7 lanes x 7 elements each (1 default), 0 = one wave per plane
// the channel-major-descriptor kernels (radix_gap_t / mix_logits / mix_bwd_reduce / mix_bwd_apply): 7 x 7 planes take the packed form
#define TAIL_DISPATCH7(KERNEL, ...)                                                                                \
    do {                                                                                                           \
        if (HW == 49 && sizeof(T) == 2 && g_radix_pack7) {                                                         \
            const dim3 grid7((unsigned)ceil_div64(planes, 32)), block7(256);                                       \
            COT_LAUNCH((KERNEL<T, 7, 8>), grid7, block7, 0, s, __VA_ARGS__);                                       \
        } else {                                                                                                   \
            TAIL_DISPATCH(KERNEL, __VA_ARGS__);                                                                    \
        }                                                                                                          \
    } while (0)

template <typename T> int radix_gap(const void* y, const void* k, void* gap, int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(radix_gap_kernel, (const T*)y, (const T*)k, (T*)gap, planes, HW);
    return check_launch("radix_gap");
}
template <typename T>
int radix_mix(const void* y, const void* k, const void* attn, void* out, int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(radix_mix_kernel, (const T*)y, (const T*)k, (const T*)attn, (T*)out, planes, HW);
    return check_launch("radix_mix");
}
template <typename T>
int radix_mix_bwd(const void* g, const void* y, const void* k, const void* attn, void* gy, void* gk, void* gattn,
                  int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(radix_mix_bwd_kernel, (const T*)g, (const T*)y, (const T*)k, (const T*)attn, (T*)gy, (T*)gk,
                  (T*)gattn, planes, HW);
    return check_launch("radix_mix_bwd");
}

template <typename T> int se_gap(const void* x, void* gap, int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(se_gap_kernel, (const T*)x, (T*)gap, planes, HW);
    return check_launch("se_gap");
}
template <typename T> int se_gate(const void* x, const void* logit, void* out, int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(se_gate_kernel, (const T*)x, (const T*)logit, (T*)out, planes, HW);
    return check_launch("se_gate");
}
template <typename T>
int se_gate_bwd(const void* g, const void* x, const void* logit, void* gx, void* glogit, int64_t planes, int HW, hipStream_t s) {
    TAIL_DISPATCH(se_gate_bwd_kernel, (const T*)g, (const T*)x, (const T*)logit, (T*)gx, (T*)glogit, planes, HW);
    return check_launch("se_gate_bwd");
}

template <typename T> int radix_gap_t(const void* y, const void* k, void* gapT, int N, int C, int HW, int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    TAIL_DISPATCH7(radix_gap_t_kernel, (const T*)y, (const T*)k, (T*)gapT, N, C, HW, lay);
    return check_launch("radix_gap_t");
}
template <typename T>
int radix_mix_logits(const void* y, const void* k, const void* logitsT, void* out, void* attn, int N, int C, int HW,
                     int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    TAIL_DISPATCH7(radix_mix_logits_kernel, (const T*)y, (const T*)k, (const T*)logitsT, (T*)out, (T*)attn, N, C, HW, lay);
    return check_launch("radix_mix_logits");
}
template <typename T>
int radix_mix_bwd_reduce(const void* g, const void* y, const void* k, const void* attn, void* glogitsT, int N, int C,
                         int HW, int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    TAIL_DISPATCH7(radix_mix_bwd_reduce_kernel, (const T*)g, (const T*)y, (const T*)k, (const T*)attn, (T*)glogitsT, N,
                  C, HW, lay);
    return check_launch("radix_mix_bwd_reduce");
}
template <typename T>
int radix_mix_bwd_apply(const void* g, const void* attn, const void* ggapT, void* gy, void* gk, int N, int C, int HW,
                        int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    TAIL_DISPATCH7(radix_mix_bwd_apply_kernel, (const T*)g, (const T*)attn, (const T*)ggapT, (T*)gy, (T*)gk, N, C, HW, lay);
    return check_launch("radix_mix_bwd_apply");
}

// ---- BatchNorm + SiLU folded into the tail (kernels above)
template <typename T>
int radix_gap_t_bn(const void* a, const void* k, void* gapT, const float* gamma, const float* beta, float* mean, float* rstd,
                   float* rmean, float* rvar, long long* nbt, const float* part, int split, float eps, float mom, int N, int C, int HW,
                   int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    const BnTail bn{gamma, beta, mean, rstd};
    const BnFin fin{part, split, eps, mom, mean, rstd, rmean, rvar, nbt};
    TAIL_DISPATCH7(radix_gap_t_bn_kernel, (const T*)a, (const T*)k, (T*)gapT, bn, fin, N, C, HW, lay);
    return check_launch("radix_gap_t_bn");
}
template <typename T>
int radix_mix_logits_bn(const void* a, const void* k, const void* logitsT, void* out, void* attn, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, int N, int C, int HW, int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    const BnTail bn{gamma, beta, mean, rstd};
    TAIL_DISPATCH7(radix_mix_logits_bn_kernel, (const T*)a, (const T*)k, (const T*)logitsT, (T*)out, (T*)attn, bn, N, C, HW, lay);
    return check_launch("radix_mix_logits_bn");
}
template <typename T>
int radix_mix_bwd_reduce_bn(const void* g, const void* a, const void* k, const void* attn, void* glogitsT, float* tsum,
                            const float* gamma, const float* beta, const float* mean, const float* rstd, int N, int C, int HW, int lay,
                            hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    const BnTail bn{gamma, beta, mean, rstd};
    TAIL_DISPATCH7(radix_mix_bwd_reduce_bn_kernel, (const T*)g, (const T*)a, (const T*)k, (const T*)attn, (T*)glogitsT, tsum, bn, N, C,
                   HW, lay);
    return check_launch("radix_mix_bwd_reduce_bn");
}
template <typename T>
int radix_mix_bwd_apply_bn(const void* g, const void* a, const void* attn, const void* ggapT, const float* tsum, void* ga, void* gk,
                           const float* gamma, const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                           int N, int C, int HW, int lay, hipStream_t s) {
    const int64_t planes = (int64_t)N * C;
    const BnTail bn{gamma, beta, mean, rstd};
    TAIL_DISPATCH7(radix_mix_bwd_apply_bn_kernel, (const T*)g, (const T*)a, (const T*)attn, (const T*)ggapT, tsum, (T*)ga, (T*)gk, bn,
                   dgamma, dbeta, N, C, HW, lay);
    return check_launch("radix_mix_bwd_apply_bn");
}

#define INST(T)                                                                                                    \
    template int radix_gap<T>(const void*, const void*, void*, int64_t, int, hipStream_t);                         \
    template int radix_mix<T>(const void*, const void*, const void*, void*, int64_t, int, hipStream_t);            \
    template int radix_mix_bwd<T>(const void*, const void*, const void*, const void*, void*, void*, void*, int64_t, \
                                  int, hipStream_t);                                                               \
    template int radix_gap_t<T>(const void*, const void*, void*, int, int, int, int, hipStream_t);                      \
    template int se_gap<T>(const void*, void*, int64_t, int, hipStream_t);                                         \
    template int se_gate<T>(const void*, const void*, void*, int64_t, int, hipStream_t);                           \
    template int se_gate_bwd<T>(const void*, const void*, const void*, void*, void*, int64_t, int, hipStream_t);   \
    template int radix_mix_logits<T>(const void*, const void*, const void*, void*, void*, int, int, int, int,      \
                                     hipStream_t);                                                                 \
    template int radix_mix_bwd_reduce<T>(const void*, const void*, const void*, const void*, void*, int, int, int, \
                                         int, hipStream_t);                                                        \
    template int radix_mix_bwd_apply<T>(const void*, const void*, const void*, void*, void*, int, int, int, int,   \
                                        hipStream_t);                                                              \
    template int radix_gap_t_bn<T>(const void*, const void*, void*, const float*, const float*, float*, float*, float*, float*,    \
                                   long long*, const float*, int, float, float, int, int, int, int, hipStream_t);  \
    template int radix_mix_logits_bn<T>(const void*, const void*, const void*, void*, void*, const float*, const float*,           \
                                        const float*, const float*, int, int, int, int, hipStream_t);              \
    template int radix_mix_bwd_reduce_bn<T>(const void*, const void*, const void*, const void*, void*, float*, const float*,       \
                                            const float*, const float*, const float*, int, int, int, int, hipStream_t); \
    template int radix_mix_bwd_apply_bn<T>(const void*, const void*, const void*, const void*, const float*, void*, void*,         \
                                           const float*, const float*, const float*, const float*, float*, float*, int, int, int,  \
                                           int, hipStream_t);
INST(float)
INST(bf16_t)

}  // namespace cot
