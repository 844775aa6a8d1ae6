// radix_nhwc.hip -- STUDY (the channels-last route): the radix-2 tail of the CoT layer (models/cotnet.py:92-104) on
// channels-last tensors y, k [N][HW][C]; csrc/radix_tail.hip is the NCHW implementation, the arithmetic is the same:
//     gap[n][c]    = mean over pixels of (y + k)                                   (the `se` branch's input)
//     out          = y * attn[n][c][0] + k * attn[n][c][1]
//     gy = gout * a0,  gk = gout * a1,  gattn[n][c][0] = sum_p gout * y,  gattn[n][c][1] = sum_p gout * k
// A thread owns V consecutive channels (16 bytes) and walks down an image's rows: whole rows are contiguous, the per-(image,
// channel) sums are column sums inside one workgroup (one workgroup per image).  Exported as cot_study_radix_nhwc_*.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cot_common.h"

namespace cot {

template <int NV>
__device__ __forceinline__ void radix_col_sum(float (&v)[NV], float* sm, int TPR, int RP, int cg, int rl) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sm[(rl * TPR + cg) * NV + k] = v[k];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float a = 0.f;
            for (int j = 0; j < RP; ++j) a += sm[(j * TPR + cg) * NV + k];
            v[k] = a;
        }
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_nhwc_gap(const T* __restrict__ y, const T* __restrict__ k, T* __restrict__ gap, int HW, int C) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    // gridDim.y workgroups share an image's channels: TPR threads per row each, 256 / TPR row lanes
    const int n = blockIdx.x, TPR = C / V / (int)gridDim.y, RP = 256 / TPR, cl = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int cg = blockIdx.y * TPR + cl;
    const int64_t base = (int64_t)n * HW * C + cg * V;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    for (int r = rl; r < HW; r += RP) {
        const Vec<T, V> a = ldv<T, V>(y + base + (int64_t)r * C), b = ldv<T, V>(k + base + (int64_t)r * C);
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += (float)a.v[e] + (float)b.v[e];
    }
    radix_col_sum<V>(s, sm, TPR, RP, cl, rl);
    if (rl == 0) {
        Vec<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) o.v[e] = (T)(s[e] / (float)HW);
        stv<T, V>(gap + (int64_t)n * C + cg * V, o);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_nhwc_mix(const T* __restrict__ y, const T* __restrict__ k, const T* __restrict__ attn,
                                                     T* __restrict__ out, int HW, int C, int64_t nvec) {
    const int vpr = C / V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / vpr;
        const int c0 = (int)(i - row * vpr) * V, n = (int)(row / HW);
        const Vec<T, V> a = ldv<T, V>(y + i * V), b = ldv<T, V>(k + i * V);
        const Vec<T, V> w0 = ldv<T, V>(attn + ((int64_t)n * C + c0) * 2), w1 = ldv<T, V>(attn + ((int64_t)n * C + c0) * 2 + V);
        Vec<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            // attn[n][c0 + e][0 / 1] = element 2e / 2e + 1 of the 2V consecutive values (w0 | w1)
            const float a0 = (float)(2 * e < V ? w0.v[(2 * e) % V] : w1.v[(2 * e) % V]);
            const float a1 = (float)(2 * e + 1 < V ? w0.v[(2 * e + 1) % V] : w1.v[(2 * e + 1) % V]);
            o.v[e] = (T)((float)a.v[e] * a0 + (float)b.v[e] * a1);
        }
        stv<T, V>(out + i * V, o);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_nhwc_mix_bwd(const T* __restrict__ gout, const T* __restrict__ y, const T* __restrict__ k,
                                                         const T* __restrict__ attn, T* __restrict__ gy, T* __restrict__ gk,
                                                         T* __restrict__ gattn, int HW, int C) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    // gridDim.y workgroups share an image's channels: TPR threads per row each, 256 / TPR row lanes
    const int n = blockIdx.x, TPR = C / V / (int)gridDim.y, RP = 256 / TPR, cl = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int cg = blockIdx.y * TPR + cl;
    const int64_t base = (int64_t)n * HW * C + cg * V;
    float a0[V], a1[V], s[2 * V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        a0[e] = (float)attn[((int64_t)n * C + cg * V + e) * 2];
        a1[e] = (float)attn[((int64_t)n * C + cg * V + e) * 2 + 1];
        s[2 * e] = s[2 * e + 1] = 0.f;
    }
    for (int r = rl; r < HW; r += RP) {
        const int64_t off = base + (int64_t)r * C;
        const Vec<T, V> g = ldv<T, V>(gout + off), a = ldv<T, V>(y + off), b = ldv<T, V>(k + off);
        Vec<T, V> oy, ok;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float gv = (float)g.v[e];
            oy.v[e] = (T)(gv * a0[e]);
            ok.v[e] = (T)(gv * a1[e]);
            s[2 * e] += gv * (float)a.v[e];
            s[2 * e + 1] += gv * (float)b.v[e];
        }
        stv<T, V>(gy + off, oy);
        stv<T, V>(gk + off, ok);
    }
    radix_col_sum<2 * V>(s, sm, TPR, RP, cl, rl);
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 2 * V; ++e) gattn[((int64_t)n * C + cg * V) * 2 + e] = (T)s[e];
    }
}

// the backward in two launches, as the NCHW path has it (cot_radix_mix_backward_reduce / _apply): the column sums first -- gattn feeds
// the `se` branch's backward, whose result ggap [N][C] (gradient of the pooled descriptor) is an input of the element-wise half:
//     gy = gout * a0 + ggap / HW,   gk = gout * a1 + ggap / HW
template <typename T, int V>
__global__ __launch_bounds__(256) void radix_nhwc_mix_bwd_reduce(const T* __restrict__ gout, const T* __restrict__ y, const T* __restrict__ k,
                                                                T* __restrict__ gattn, int HW, int C) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    // gridDim.y workgroups share an image's channels: TPR threads per row each, 256 / TPR row lanes
    const int n = blockIdx.x, TPR = C / V / (int)gridDim.y, RP = 256 / TPR, cl = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int cg = blockIdx.y * TPR + cl;
    const int64_t base = (int64_t)n * HW * C + cg * V;
    float s[2 * V];
#pragma unroll
    for (int e = 0; e < 2 * V; ++e) s[e] = 0.f;
    for (int r = rl; r < HW; r += RP) {
        const int64_t off = base + (int64_t)r * C;
        const Vec<T, V> g = ldv<T, V>(gout + off), a = ldv<T, V>(y + off), b = ldv<T, V>(k + off);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            s[2 * e] += (float)g.v[e] * (float)a.v[e];
            s[2 * e + 1] += (float)g.v[e] * (float)b.v[e];
        }
    }
    radix_col_sum<2 * V>(s, sm, TPR, RP, cl, rl);
    if (rl == 0) {
#pragma unroll
        for (int e = 0; e < 2 * V; ++e) gattn[((int64_t)n * C + cg * V) * 2 + e] = (T)s[e];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void radix_nhwc_mix_bwd_apply(const T* __restrict__ gout, const T* __restrict__ attn, const T* __restrict__ ggap,
                                                               T* __restrict__ gy, T* __restrict__ gk, int HW, int C, int64_t nvec) {
    const int vpr = C / V;
    const float inv = 1.0f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / vpr;
        const int c0 = (int)(i - row * vpr) * V, n = (int)(row / HW);
        const Vec<T, V> g = ldv<T, V>(gout + i * V);
        const Vec<T, V> w0 = ldv<T, V>(attn + ((int64_t)n * C + c0) * 2), w1 = ldv<T, V>(attn + ((int64_t)n * C + c0) * 2 + V);
        Vec<T, V> gg, oy, ok;
        if (ggap) gg = ldv<T, V>(ggap + (int64_t)n * C + c0);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float a0 = (float)(2 * e < V ? w0.v[(2 * e) % V] : w1.v[(2 * e) % V]);
            const float a1 = (float)(2 * e + 1 < V ? w0.v[(2 * e + 1) % V] : w1.v[(2 * e + 1) % V]);
            const float add = ggap ? (float)gg.v[e] * inv : 0.f;
            oy.v[e] = (T)((float)g.v[e] * a0 + add);
            ok.v[e] = (T)((float)g.v[e] * a1 + add);
        }
        stv<T, V>(gy + i * V, oy);
        stv<T, V>(gk + i * V, ok);
    }
}

// out[c] = sum over the M rows of x[m][c] (the bias gradient of a channels-last 1x1 convolution): one workgroup per 64 * V channels
// ... of a column block, rows in RP lanes; fp32 sums, rounded once
// grid (column blocks of 32 * V channels, row slabs): fp32 partial sums part[slab][C]; nhwc_col_sum_finish adds the slabs in order
template <typename T, int V>
__global__ __launch_bounds__(256) void nhwc_col_sum(const T* __restrict__ x, float* __restrict__ part, int M, int C, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    const int TPR = 32, RP = 8, cg = threadIdx.x % TPR, rl = threadIdx.x / TPR;  // 32 x V channels per workgroup, 8 row lanes
    const int c0 = (blockIdx.x * TPR + cg) * V;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (c0 < C)
        for (int r = r0 + rl; r < r1; r += RP) {
            const Vec<T, V> a = ldv<T, V>(x + (int64_t)r * C + c0);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += (float)a.v[e];
        }
    radix_col_sum<V>(s, sm, TPR, RP, cg, rl);
    if (rl == 0 && c0 < C) {
#pragma unroll
        for (int e = 0; e < V; ++e) part[(int64_t)blockIdx.y * C + c0 + e] = s[e];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_col_sum_finish(const float* __restrict__ part, T* __restrict__ out, int C, int slabs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += part[(int64_t)k * C + c];
    out[c] = (T)s;
}
static int nhwc_col_sum_slabs(int M) {
    int s = M / 64;  // at least 64 rows per slab, at most 128 slabs
    return s < 1 ? 1 : (s > 128 ? 128 : s);
}

// attn[i][0..1] = softmax(logits[i][0..1]) over the radix pair (models/cotnet.py:99-101: view(B, C, radix), softmax(dim=2)); its backward
// glogits = attn * (gattn - sum_r attn * gattn).  n2 = number of pairs (N * C); fp32 arithmetic, one rounding
template <typename T>
__global__ __launch_bounds__(256) void radix_softmax2(const T* __restrict__ logits, T* __restrict__ attn, int64_t n2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const float a = (float)logits[2 * i], b = (float)logits[2 * i + 1], m = fmaxf(a, b);
    const float ea = __expf(a - m), eb = __expf(b - m), inv = 1.0f / (ea + eb);
    attn[2 * i] = (T)(ea * inv);
    attn[2 * i + 1] = (T)(eb * inv);
}
template <typename T>
__global__ __launch_bounds__(256) void radix_softmax2_bwd(const T* __restrict__ attn, const T* __restrict__ gattn, T* __restrict__ glogits, int64_t n2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    const float a0 = (float)attn[2 * i], a1 = (float)attn[2 * i + 1], g0 = (float)gattn[2 * i], g1 = (float)gattn[2 * i + 1];
    const float dot = a0 * g0 + a1 * g1;
    glogits[2 * i] = (T)(a0 * (g0 - dot));
    glogits[2 * i + 1] = (T)(a1 * (g1 - dot));
}

template <typename T, int V> static bool radix_nhwc_covers(int C) {
    return C > 0 && C % V == 0 && C / V <= 256 && 256 % (C / V) == 0;
}
template <typename T, int V>
static int radix_nhwc_run(int what, const void* gout, const void* y, const void* k, const void* attn, void* o1, void* o2, void* o3, int N,
                          int HW, int C, hipStream_t s) {
    if (!radix_nhwc_covers<T, V>(C)) return -2;
    int split = 1;  // workgroups per image: ~512 in the launch, at least 4 threads per row each
    while (split * 2 * N <= 512 && C / V / (split * 2) >= 4) split *= 2;
    if (what == 0) {
        COT_LAUNCH((radix_nhwc_gap<T, V>), dim3(N, split), dim3(256), 256 * V * sizeof(float), s, (const T*)y, (const T*)k, (T*)o1, HW, C);
    } else if (what == 1) {
        const int64_t nvec = (int64_t)N * HW * C / V;
        int64_t b = ceil_div64(nvec, 256 * 2);
        b = b < 1 ? 1 : (b > 2048 ? 2048 : b);
        COT_LAUNCH((radix_nhwc_mix<T, V>), dim3((unsigned)b), dim3(256), 0, s, (const T*)y, (const T*)k, (const T*)attn, (T*)o1, HW, C, nvec);
    } else if (what == 2) {
        COT_LAUNCH((radix_nhwc_mix_bwd<T, V>), dim3(N, split), dim3(256), 256 * 2 * V * sizeof(float), s, (const T*)gout, (const T*)y, (const T*)k,
                   (const T*)attn, (T*)o1, (T*)o2, (T*)o3, HW, C);
    } else if (what == 3) {
        COT_LAUNCH((radix_nhwc_mix_bwd_reduce<T, V>), dim3(N, split), dim3(256), 256 * 2 * V * sizeof(float), s, (const T*)gout, (const T*)y,
                   (const T*)k, (T*)o1, HW, C);
    } else {  // 4: apply; `y` carries ggap (may be NULL)
        const int64_t nvec = (int64_t)N * HW * C / V;
        int64_t b = ceil_div64(nvec, 256 * 2);
        b = b < 1 ? 1 : (b > 2048 ? 2048 : b);
        COT_LAUNCH((radix_nhwc_mix_bwd_apply<T, V>), dim3((unsigned)b), dim3(256), 0, s, (const T*)gout, (const T*)attn, (const T*)y, (T*)o1,
                   (T*)o2, HW, C, nvec);
    }
    return check_launch("radix_nhwc");
}

}  // namespace cot

// y, k, out, gout, gy, gk: [N][HW][C]; gap: [N][C]; attn, gattn: [N][C][2].  dtype COT_F32 (0) / COT_BF16 (2); C / (16 bytes' worth of
// channels) a power of two up to 256
#define RADIX_NHWC_DISPATCH(...)                                                                      \
    do {                                                                                              \
        if (N <= 0 || HW <= 0 || C <= 0) return -1;                                                   \
        if (dtype == 2) return cot::radix_nhwc_run<cot::bf16_t, 8>(__VA_ARGS__, N, HW, C, (hipStream_t)stream); \
        if (dtype == 0) return cot::radix_nhwc_run<float, 4>(__VA_ARGS__, N, HW, C, (hipStream_t)stream);      \
        return -2;                                                                                    \
    } while (0)
extern "C" int cot_study_radix_nhwc_gap(const void* y, const void* k, void* gap, int N, int HW, int C, int dtype, void* stream) {
    if (!y || !k || !gap) return -1;
    RADIX_NHWC_DISPATCH(0, nullptr, y, k, nullptr, gap, nullptr, nullptr);
}
extern "C" int cot_study_radix_nhwc_mix(const void* y, const void* k, const void* attn, void* out, int N, int HW, int C, int dtype, void* stream) {
    if (!y || !k || !attn || !out) return -1;
    RADIX_NHWC_DISPATCH(1, nullptr, y, k, attn, out, nullptr, nullptr);
}
extern "C" int cot_study_radix_nhwc_mix_backward(const void* gout, const void* y, const void* k, const void* attn, void* gy, void* gk,
                                                 void* gattn, int N, int HW, int C, int dtype, void* stream) {
    if (!gout || !y || !k || !attn || !gy || !gk || !gattn) return -1;
    RADIX_NHWC_DISPATCH(2, gout, y, k, attn, gy, gk, gattn);
}
extern "C" int cot_study_radix_nhwc_mix_backward_reduce(const void* gout, const void* y, const void* k, void* gattn, int N, int HW, int C,
                                                        int dtype, void* stream) {
    if (!gout || !y || !k || !gattn) return -1;
    RADIX_NHWC_DISPATCH(3, gout, y, k, nullptr, gattn, nullptr, nullptr);
}
extern "C" int cot_study_radix_nhwc_mix_backward_apply(const void* gout, const void* attn, const void* ggap, void* gy, void* gk, int N, int HW,
                                                       int C, int dtype, void* stream) {
    if (!gout || !attn || !gy || !gk) return -1;
    RADIX_NHWC_DISPATCH(4, gout, ggap, nullptr, attn, gy, gk, nullptr);
}
// out[C] = column sums of x[M][C] (bias gradient of a channels-last convolution); C a multiple of 8 (bf16) / 4 (fp32); workspace:
// cot_study_nhwc_col_sum_workspace(M, C) floats
extern "C" int cot_study_nhwc_col_sum_workspace(int M, int C) { return M > 0 && C > 0 ? cot::nhwc_col_sum_slabs(M) * C : 0; }
extern "C" int cot_study_nhwc_col_sum(const void* x, void* out, float* workspace, int M, int C, int dtype, void* stream) {
    if (!x || !out || !workspace || M <= 0 || C <= 0) return -1;
    const int slabs = cot::nhwc_col_sum_slabs(M), rows_per = cot::ceil_div(M, slabs);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 2) {
        if (C % 8) return -2;
        COT_LAUNCH((cot::nhwc_col_sum<cot::bf16_t, 8>), dim3(cot::ceil_div(C, 256), slabs), dim3(256), 256 * 8 * sizeof(float), s, (const cot::bf16_t*)x,
                   workspace, M, C, rows_per);
        COT_LAUNCH((cot::nhwc_col_sum_finish<cot::bf16_t>), dim3(cot::ceil_div(C, 256)), dim3(256), 0, s, (const float*)workspace, (cot::bf16_t*)out, C, slabs);
    } else if (dtype == 0) {
        if (C % 4) return -2;
        COT_LAUNCH((cot::nhwc_col_sum<float, 4>), dim3(cot::ceil_div(C, 128), slabs), dim3(256), 256 * 4 * sizeof(float), s, (const float*)x, workspace, M,
                   C, rows_per);
        COT_LAUNCH((cot::nhwc_col_sum_finish<float>), dim3(cot::ceil_div(C, 256)), dim3(256), 0, s, (const float*)workspace, (float*)out, C, slabs);
    } else {
        return -2;
    }
    return cot::check_launch("nhwc_col_sum");
}
extern "C" int cot_study_radix_softmax2(const void* logits, void* attn, int64_t pairs, int dtype, void* stream) {
    if (!logits || !attn || pairs <= 0) return -1;
    const dim3 grid((unsigned)cot::ceil_div64(pairs, 256));
    if (dtype == 2) COT_LAUNCH((cot::radix_softmax2<cot::bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const cot::bf16_t*)logits, (cot::bf16_t*)attn, pairs);
    else if (dtype == 0) COT_LAUNCH((cot::radix_softmax2<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)logits, (float*)attn, pairs);
    else return -2;
    return cot::check_launch("radix_softmax2");
}
extern "C" int cot_study_radix_softmax2_backward(const void* attn, const void* gattn, void* glogits, int64_t pairs, int dtype, void* stream) {
    if (!attn || !gattn || !glogits || pairs <= 0) return -1;
    const dim3 grid((unsigned)cot::ceil_div64(pairs, 256));
    if (dtype == 2)
        COT_LAUNCH((cot::radix_softmax2_bwd<cot::bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const cot::bf16_t*)attn, (const cot::bf16_t*)gattn,
                   (cot::bf16_t*)glogits, pairs);
    else if (dtype == 0)
        COT_LAUNCH((cot::radix_softmax2_bwd<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)attn, (const float*)gattn, (float*)glogits, pairs);
    else return -2;
    return cot::check_launch("radix_softmax2_backward");
}
