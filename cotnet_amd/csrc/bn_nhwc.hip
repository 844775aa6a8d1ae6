// bn_nhwc.hip -- STUDY (the channels-last route for the 14x14 / 7x7 stages): training-mode BatchNorm2d + activation
// (+ residual) on a channels-last activation x[M][C], M = images * H * W (models/cotnet.py:231-235,:248-262 are the module
// sequences it replaces; csrc/bn_act.hip is the NCHW implementation whose arithmetic, partial-sum layouts and finalize kernels it
// shares).  Not on any model's path yet: exported as cot_study_bn_nhwc_*; host-emulated tests only.
//
// A channel is a COLUMN here: a thread owns 8 consecutive channels (one 16-byte access; 4 in fp32) and walks down the rows, so a
// workgroup reads whole rows -- contiguous memory -- whatever the plane size.  Statistics: per (row slab, channel) the shifted sums
// (count, mean, M2) of bn_act.hip's chunks, merged by its finalize kernel (Chan); the apply kernels are flat over 16-byte vectors
// with the thread's channels -- and with them its scale / shift -- fixed for the whole grid-stride loop (256 threads and every grid
// stride are multiples of the threads per row).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cot_common.h"

namespace cot {

void bn_launch_stats_finalize(const float*, int, int, float, float, float*, float*, float*, float*, long long*, hipStream_t);
void bn_launch_bwd_finalize(const float*, int, int, float*, float*, hipStream_t);

enum { N_NONE = 0, N_RELU = 1, N_SILU = 2, N_RELU_Y = 3 };
template <int ACT> __device__ __forceinline__ float nact_fwd(float z) {
    if (ACT == N_RELU || ACT == N_RELU_Y) return z > 0.f ? z : 0.f;
    if (ACT == N_SILU) return z / (1.f + __expf(-z));
    return z;
}
template <int ACT> __device__ __forceinline__ float nact_bwd(float dy, float z_or_y) {
    if (ACT == N_RELU || ACT == N_RELU_Y) return z_or_y > 0.f ? dy : 0.f;
    if (ACT == N_SILU) {
        const float sg = 1.f / (1.f + __expf(-z_or_y));
        return dy * sg * (1.f + z_or_y * (1.f - sg));
    }
    return dy;
}

// geometry of a column reduction: TPR threads cover one row of a column block of CB channels, RP rows at a time
struct ColGeom {
    int CB, TPR, RP, ncb;
};
template <int V> static bool col_geom(int C, ColGeom* g) {
    const int cbmax = 256 * V;
    if (C > cbmax || C % V) return false;  // (one column block: the apply kernels keep a thread's channels fixed, 256 % (C / V) == 0)
    g->CB = C;
    g->TPR = g->CB / V;
    if (256 % g->TPR) return false;  // (channel counts 8 * 2^k up to 2048 in bf16, 4 * 2^k up to 1024 in fp32)
    g->RP = 256 / g->TPR;
    g->ncb = C / g->CB;
    return true;
}

// per (slab, channel): count, mean, M2 of the slab's rows (shift = the slab's first row) -> part[(c * split + s) * 4 ..]
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_nhwc_stats(const T* __restrict__ x, float* __restrict__ part, int M, int C, int TPR, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    const int split = gridDim.y, s = blockIdx.y, RP = 256 / TPR;
    const int cg = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int c0 = blockIdx.x * TPR * V + cg * V;
    const int r0 = s * rows_per, r1 = min(M, r0 + rows_per);
    float sum[V], sq[V], shift[V];
#pragma unroll
    for (int k = 0; k < V; ++k) sum[k] = sq[k] = shift[k] = 0.f;
    if (r0 < r1) {
        const Vec<T, V> f = ldv<T, V>(x + (int64_t)r0 * C + c0);
#pragma unroll
        for (int k = 0; k < V; ++k) shift[k] = (float)f.v[k];
    }
#pragma unroll 2
    for (int r = r0 + rl; r < r1; r += RP) {
        const Vec<T, V> xv = ldv<T, V>(x + (int64_t)r * C + c0);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = (float)xv.v[k] - shift[k];
            sum[k] += d;
            sq[k] += d * d;
        }
    }
    // the RP threads of a channel group: through LDS, added in row-lane order
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sm[(threadIdx.x * V + k) * 2] = sum[k];
        sm[(threadIdx.x * V + k) * 2 + 1] = sq[k];
    }
    __syncthreads();
    if (rl == 0) {
        const float cnt = (float)(r1 > r0 ? r1 - r0 : 0);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float a = 0.f, b = 0.f;
            for (int j = 0; j < RP; ++j) {
                a += sm[((j * TPR + cg) * V + k) * 2];
                b += sm[((j * TPR + cg) * V + k) * 2 + 1];
            }
            float* p = part + ((int64_t)(c0 + k) * split + s) * 4;
            p[0] = cnt;
            p[1] = cnt > 0 ? shift[k] + a / cnt : 0.f;
            p[2] = cnt > 0 ? b - a * a / cnt : 0.f;
            p[3] = 0.f;
        }
    }
}

template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_nhwc_apply_fwd(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                        int64_t nvec) {
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c0 = (int)(t0 % (C / V)) * V;  // (the same for every vector of this thread: stride is a multiple of C / V)
    float a[V], b[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        a[k] = gamma[c0 + k] * rstd[c0 + k];
        b[k] = beta[c0 + k] - mean[c0 + k] * a[k];
    }
#pragma unroll 2
    for (int64_t i = t0; i < nvec; i += stride) {
        const Vec<T, V> xv = ldv<T, V>(x + i * V);
        Vec<T, V> rv, o;
        if (res) rv = ldv<T, V>(res + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = (float)xv.v[k] * a[k] + b[k];
            if (res) z += (float)rv.v[k];
            o.v[k] = (T)nact_fwd<ACT>(z);
        }
        stv<T, V>(y + i * V, o);
    }
}

// per (slab, channel): sum g, sum g * xhat, g = dy * act'(.) -> part[(c * split + s) * 2 ..]
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_nhwc_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ part, int M, int C, int TPR, int rows_per) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* sm = reinterpret_cast<float*>(cot_smem);
    const int split = gridDim.y, s = blockIdx.y, RP = 256 / TPR;
    const int cg = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int c0 = blockIdx.x * TPR * V + cg * V;
    const int r0 = s * rows_per, r1 = min(M, r0 + rows_per);
    float sg[V], sgx[V], m[V], r[V], ga[V], be[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sg[k] = sgx[k] = 0.f;
        m[k] = mean[c0 + k];
        r[k] = rstd[c0 + k];
        ga[k] = gamma[c0 + k];
        be[k] = beta[c0 + k];
    }
#pragma unroll 2
    for (int row = r0 + rl; row < r1; row += RP) {
        const int64_t off = (int64_t)row * C + c0;
        const Vec<T, V> dv = ldv<T, V>(dy + off), xv = ldv<T, V>(x + off);
        Vec<T, V> yv;
        if (ACT == N_RELU_Y) yv = ldv<T, V>(y + off);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = ((float)xv.v[k] - m[k]) * r[k];
            const float g = nact_bwd<ACT>((float)dv.v[k], ACT == N_RELU_Y ? (float)yv.v[k] : xh * ga[k] + be[k]);
            sg[k] += g;
            sgx[k] += g * xh;
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        sm[(threadIdx.x * V + k) * 2] = sg[k];
        sm[(threadIdx.x * V + k) * 2 + 1] = sgx[k];
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float a = 0.f, b = 0.f;
            for (int j = 0; j < RP; ++j) {
                a += sm[((j * TPR + cg) * V + k) * 2];
                b += sm[((j * TPR + cg) * V + k) * 2 + 1];
            }
            float* p = part + ((int64_t)(c0 + k) * split + s) * 2;
            p[0] = a;
            p[1] = b;
        }
    }
}

template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_nhwc_apply_bwd(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                        T* __restrict__ dx, T* __restrict__ dres, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                        const float* __restrict__ dbeta, int C, int64_t nvec, float inv_m) {
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c0 = (int)(t0 % (C / V)) * V;
    float m[V], r[V], ga[V], be[V], mg[V], mgx[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        m[k] = mean[c0 + k];
        r[k] = rstd[c0 + k];
        ga[k] = gamma[c0 + k];
        be[k] = beta[c0 + k];
        mg[k] = dbeta[c0 + k] * inv_m;
        mgx[k] = dgamma[c0 + k] * inv_m;
    }
#pragma unroll 2
    for (int64_t i = t0; i < nvec; i += stride) {
        const Vec<T, V> dv = ldv<T, V>(dy + i * V), xv = ldv<T, V>(x + i * V);
        Vec<T, V> yv, o, og;
        if (ACT == N_RELU_Y) yv = ldv<T, V>(y + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = ((float)xv.v[k] - m[k]) * r[k];
            const float g = nact_bwd<ACT>((float)dv.v[k], ACT == N_RELU_Y ? (float)yv.v[k] : xh * ga[k] + be[k]);
            og.v[k] = (T)g;
            o.v[k] = (T)(ga[k] * r[k] * (g - mg[k] - xh * mgx[k]));
        }
        stv<T, V>(dx + i * V, o);
        if (dres) stv<T, V>(dres + i * V, og);
    }
}

static void pick_slabs(int M, const ColGeom& g, int* split, int* rows_per) {
    // enough workgroups for two per CU; a slab is a whole number of row groups and at least four of them
    int s = 512 / g.ncb;
    const int groups = ceil_div(M, g.RP);
    if (s > groups / 4) s = groups / 4;
    if (s < 1) s = 1;
    *rows_per = ceil_div(groups, s) * g.RP;
    *split = ceil_div(M, *rows_per);
}
static unsigned flat_blocks(int64_t nvec) {
    int64_t b = ceil_div64(nvec, 256 * 4);
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

template <int V> static int nhwc_workspace_floats(int M, int C) {
    ColGeom g;
    if (!col_geom<V>(C, &g)) return 0;
    int split, rows_per;
    pick_slabs(M, g, &split, &rows_per);
    return C * split * 4;
}
int bn_nhwc_workspace_floats(int M, int C, int esize) {
    return esize == 2 ? nhwc_workspace_floats<8>(M, C) : nhwc_workspace_floats<4>(M, C);
}

template <typename T, int V>
static int nhwc_forward(const T* x, const T* res, T* y, const float* gamma, const float* beta, float* mean, float* rstd, float* rmean,
                        float* rvar, long long* nbt, float* ws, int M, int C, float eps, float mom, int act, hipStream_t s) {
    ColGeom g;
    if (!col_geom<V>(C, &g) || (int64_t)M * C >= ((int64_t)1 << 40)) return -2;
    int split, rows_per;
    pick_slabs(M, g, &split, &rows_per);
    COT_LAUNCH((bn_nhwc_stats<T, V>), dim3(g.ncb, split), dim3(256), 256 * V * 2 * sizeof(float), s, x, ws, M, C, g.TPR, rows_per);
    bn_launch_stats_finalize(ws, C, split, eps, mom, mean, rstd, rmean, rvar, nbt, s);
    const int64_t nvec = (int64_t)M * C / V;
    const dim3 grid(flat_blocks(nvec));
#define BN_NF(A_) COT_LAUNCH((bn_nhwc_apply_fwd<T, V, A_>), grid, dim3(256), 0, s, x, res, y, (const float*)mean, (const float*)rstd, gamma, beta, C, nvec)
    if (act == N_RELU) BN_NF(N_RELU);
    else if (act == N_SILU) BN_NF(N_SILU);
    else BN_NF(N_NONE);
#undef BN_NF
    return check_launch("bn_nhwc_forward");
}

template <typename T, int V>
static int nhwc_backward(const T* dy, const T* x, const T* y, T* dx, T* dres, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, float* dgamma, float* dbeta, float* ws, int M, int C, int act, hipStream_t s) {
    ColGeom g;
    if (!col_geom<V>(C, &g) || (int64_t)M * C >= ((int64_t)1 << 40)) return -2;
    int split, rows_per;
    pick_slabs(M, g, &split, &rows_per);
    const int64_t nvec = (int64_t)M * C / V;
    const dim3 grid(flat_blocks(nvec));
    const float inv_m = 1.0f / (float)M;
#define BN_NB(A_)                                                                                                                    \
    do {                                                                                                                             \
        COT_LAUNCH((bn_nhwc_bwd_reduce<T, V, A_>), dim3(g.ncb, split), dim3(256), 256 * V * 2 * sizeof(float), s, dy, x, y, mean, rstd,  \
                   gamma, beta, ws, M, C, g.TPR, rows_per);                                                                          \
        bn_launch_bwd_finalize(ws, C, split, dgamma, dbeta, s);                                                                      \
        COT_LAUNCH((bn_nhwc_apply_bwd<T, V, A_>), grid, dim3(256), 0, s, dy, x, y, dx, dres, mean, rstd, gamma, beta,                \
                   (const float*)dgamma, (const float*)dbeta, C, nvec, inv_m);                                                       \
    } while (0)
    if (act == N_RELU && y) BN_NB(N_RELU_Y);
    else if (act == N_RELU) BN_NB(N_RELU);
    else if (act == N_SILU) BN_NB(N_SILU);
    else BN_NB(N_NONE);
#undef BN_NB
    return check_launch("bn_nhwc_backward");
}

}  // namespace cot

// y = act(gamma * (x - mean_c) * rstd_c + beta [+ residual]) on x[M][C]; the arguments of cot_bn_act_forward with the tensor's two
// extents in place of (N, C, HW).  dtype: COT_F32 (0) / COT_BF16 (2).  workspace: cot_study_bn_nhwc_workspace floats.
extern "C" int cot_study_bn_nhwc_workspace(int M, int C, int dtype) { return cot::bn_nhwc_workspace_floats(M, C, dtype == 2 ? 2 : 4); }
extern "C" int cot_study_bn_nhwc_forward(const void* x, const void* residual, void* y, const float* gamma, const float* beta, float* save_mean,
                                         float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                         float* workspace, int M, int C, float eps, float momentum, int act, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace || M <= 0 || C <= 0 || act < 0 || act > 2) return -1;
    if (((uintptr_t)x | (uintptr_t)residual | (uintptr_t)y) % 16) return -1;
    if (dtype == 2)
        return cot::nhwc_forward<cot::bf16_t, 8>((const cot::bf16_t*)x, (const cot::bf16_t*)residual, (cot::bf16_t*)y, gamma, beta, save_mean,
                                                 save_rstd, running_mean, running_var, (long long*)num_batches_tracked, workspace, M, C, eps,
                                                 momentum, act, (hipStream_t)stream);
    if (dtype == 0)
        return cot::nhwc_forward<float, 4>((const float*)x, (const float*)residual, (float*)y, gamma, beta, save_mean, save_rstd, running_mean,
                                           running_var, (long long*)num_batches_tracked, workspace, M, C, eps, momentum, act,
                                           (hipStream_t)stream);
    return -2;
}
extern "C" int cot_study_bn_nhwc_backward(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                                          const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                          float* workspace, int M, int C, int act, int dtype, void* stream) {
    if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || act < 0 ||
        act > 2)
        return -1;
    if (act == 1 && !y && dresidual) return -1;  // (ReLU after a residual add: the sign of the saved output decides)
    if (act == 2 && dresidual) return -2;        // (SiLU after a residual add: no backward, as in cot_bn_act_backward)
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)dresidual) % 16) return -1;
    if (dtype == 2)
        return cot::nhwc_backward<cot::bf16_t, 8>((const cot::bf16_t*)dy, (const cot::bf16_t*)x, (const cot::bf16_t*)y, (cot::bf16_t*)dx,
                                                  (cot::bf16_t*)dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta, workspace, M, C, act,
                                                  (hipStream_t)stream);
    if (dtype == 0)
        return cot::nhwc_backward<float, 4>((const float*)dy, (const float*)x, (const float*)y, (float*)dx, (float*)dresidual, gamma, beta,
                                            save_mean, save_rstd, dgamma, dbeta, workspace, M, C, act, (hipStream_t)stream);
    return -2;
}
