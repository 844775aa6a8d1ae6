// bn_act.hip -- training-mode BatchNorm2d fused with its activation and residual add, NCHW, gfx950.
//
// SURVEY.md 8f rank 1 ("the steps either side of the CoT layer"): in the reference every BatchNorm is followed by a
// separate in-place ReLU / SiLU pass and, at the end of a Bottleneck, a separate residual add
// (models/cotnet.py:231-235,:248-262,:89-90).  On MI355X these are all HBM-bound passes over the same tensor, so they
// are folded into the normalisation:
//   forward   y = act(gamma * (x - mean_c) * rstd_c + beta [+ residual])           act in {identity, ReLU, SiLU}
//   backward  g = dL/dy * act'(.)  ;  dbeta_c = sum g ; dgamma_c = sum g * xhat
//             dx = gamma * rstd * (g - dbeta/M - xhat * dgamma/M) ;  dresidual = g
// Batch statistics use per-block shifted sums merged with Chan's parallel-variance formula (no E[x^2]-E[x]^2
// cancellation), biased variance for normalisation, unbiased for the running estimate -- torch.nn.BatchNorm2d's
// convention (the reference relies on it through nn.BatchNorm2d).
//
// All kernels are HBM-bound.  Algorithmic traffic per element (e = storage bytes): forward 3e (+e with residual)
// [stats read, apply read, write], backward 5e (+e) [reduce reads dy,x(,y); apply reads dy,x(,y), writes dx(,dres)].
#include "cot_common.h"

namespace cot {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
// kernel-side activation selector (template parameter: the element loops carry no branches on it); ACT_RELU_Y = ReLU
// whose backward reads the saved OUTPUT's sign instead of recomputing z
enum { ACT_RELU_Y = 3 };

// Index walkers: the kernels step through a tensor with a fixed stride (blockDim, or gridDim*blockDim); what they need
// per step is (image, vector-in-plane) or (channel, ...) of the flat index.  Dividing per step costs more instructions than
// the eight elements of a vector (the first version did: two integer divisions per 16 bytes); stepping the quotient and
// remainder costs three.
struct PlaneWalk {  // i = start, start + step, ... -> (n, v) = divmod(i, vpp)
    int n, v, qn, qv, vpp;
    __device__ __forceinline__ PlaneWalk(int start, int step, int vpp_) : vpp(vpp_) {
        n = start / vpp_;
        v = start - n * vpp_;
        qn = step / vpp_;
        qv = step - qn * vpp_;
    }
    __device__ __forceinline__ void next() {
        n += qn;
        v += qv;
        if (v >= vpp) {
            v -= vpp;
            ++n;
        }
    }
};
struct ChannelWalk {  // i = start, start + step, ... -> c = (i / vpp) % C  (and i itself)
    int64_t i, step;
    int c, v, qc, qv, vpp, C;
    __device__ __forceinline__ ChannelWalk(int64_t start, int64_t step_, int vpp_, int C_)
        : i(start), step(step_), vpp(vpp_), C(C_) {
        const int64_t plane = start / vpp_, qp = step_ / vpp_;
        c = (int)(plane % C_);
        v = (int)(start - plane * vpp_);
        qc = (int)(qp % C_);
        qv = (int)(step_ - qp * vpp_);
    }
    __device__ __forceinline__ void next() {
        i += step;
        v += qv;
        c += qc;
        if (v >= vpp) {
            v -= vpp;
            ++c;
        }
        if (c >= C) c -= C;
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// block-wide sum of up to 3 values; result valid in thread 0
template <int NV> __device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* >= NV*4 floats */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) smem[k * 16 + wave] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += smem[k * 16 + w];
            v[k] = s;
        }
    }
}

// ---- forward statistics: grid (C, SPLIT); block handles images [s*nper, (s+1)*nper) of channel c -----------------
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_stats_partial(const T* __restrict__ x, float* __restrict__ part, int N, int C,
                                                       int HW, int nper) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* smem = reinterpret_cast<float*>(cot_smem);
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const int n0 = s * nper, n1 = min(N, n0 + nper);
    const int vpp = HW / V;  // vectors per plane
    float shift = 0.f;
    if (n0 < n1) shift = (float)x[((int64_t)n0 * C + c) * HW];
    float acc[2] = {0.f, 0.f};
    // Two interleaved streams per thread (vectors t, t+2B, .. and t+B, t+3B, ..; B = blockDim): both loads are issued
    // before either is used -- twice the bytes in flight of this one-read kernel -- and the sums still run in the order
    // t, t+B, t+2B, .. of a one-stream loop.  Stream b is always one stride ahead of a, so "b valid" implies "a valid"
    // and at most one vector of stream a is left over.
    const int cnt = n1 - n0;
    auto accumulate = [&](const Vec<T, V>& xv) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = (float)xv.v[k] - shift;
            acc[0] += d;
            acc[1] += d * d;
        }
    };
    PlaneWalk a(threadIdx.x, 2 * blockDim.x, vpp), b(threadIdx.x + blockDim.x, 2 * blockDim.x, vpp);
    for (; b.n < cnt; a.next(), b.next()) {
        const Vec<T, V> xa = ldv<T, V>(x + ((int64_t)(n0 + a.n) * C + c) * HW + (int64_t)a.v * V);
        const Vec<T, V> xb = ldv<T, V>(x + ((int64_t)(n0 + b.n) * C + c) * HW + (int64_t)b.v * V);
        accumulate(xa);
        accumulate(xb);
    }
    if (a.n < cnt) accumulate(ldv<T, V>(x + ((int64_t)(n0 + a.n) * C + c) * HW + (int64_t)a.v * V));
    block_sum<2>(acc, smem);
    if (threadIdx.x == 0) {
        float* p = part + ((int64_t)c * split + s) * 4;
        const float cnt = (float)((int64_t)(n1 - n0) * HW);
        p[0] = cnt;
        p[1] = cnt > 0 ? shift + acc[0] / cnt : 0.f;                // mean of the chunk
        p[2] = cnt > 0 ? acc[1] - acc[0] * acc[0] / cnt : 0.f;       // M2 of the chunk
        p[3] = 0.f;
    }
}

// Chunk SUMS about a shift common to the channel's chunks (round 6): part[c][s] = (sum (x - shift), sum (x - shift)^2, count, shift),
// shift = the channel's first element.  Merging such chunks is three additions each -- cheap enough to live in the prologue of a consumer
// that has one wave per (image, channel) plane (radix_gap_t_bn: bn_stats_partial's Chan merge, two divisions per chunk, cost those
// waves more instructions than their plane at 14 x 14) -- so the statistics need no finalize launch.  One sample as the shift keeps
// sum (x - shift)^2 / n - (sum (x - shift) / n)^2 at a few times the variance: no cancellation to speak of in fp32.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_stats_sums(const T* __restrict__ x, float* __restrict__ part, int N, int C, int HW, int nper) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* smem = reinterpret_cast<float*>(cot_smem);
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const int n0 = s * nper, n1 = min(N, n0 + nper);
    const int vpp = HW / V;
    const float shift = (float)x[(int64_t)c * HW];
    float acc[2] = {0.f, 0.f};
    const int cnt = n1 - n0;
    auto accumulate = [&](const Vec<T, V>& xv) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = (float)xv.v[k] - shift;
            acc[0] += d;
            acc[1] += d * d;
        }
    };
    PlaneWalk a(threadIdx.x, 2 * blockDim.x, vpp), b(threadIdx.x + blockDim.x, 2 * blockDim.x, vpp);
    for (; b.n < cnt; a.next(), b.next()) {
        const Vec<T, V> xa = ldv<T, V>(x + ((int64_t)(n0 + a.n) * C + c) * HW + (int64_t)a.v * V);
        const Vec<T, V> xb = ldv<T, V>(x + ((int64_t)(n0 + b.n) * C + c) * HW + (int64_t)b.v * V);
        accumulate(xa);
        accumulate(xb);
    }
    if (a.n < cnt) accumulate(ldv<T, V>(x + ((int64_t)(n0 + a.n) * C + c) * HW + (int64_t)a.v * V));
    block_sum<2>(acc, smem);
    if (threadIdx.x == 0) {
        float* p = part + ((int64_t)c * split + s) * 4;
        p[0] = acc[0];
        p[1] = acc[1];
        p[2] = (float)((int64_t)(n1 > n0 ? n1 - n0 : 0) * HW);
        p[3] = shift;
    }
}

// one thread per channel: merge the SPLIT chunks (Chan), produce mean / rstd, update running statistics
__global__ void bn_stats_finalize(const float* __restrict__ part, int C, int split, float eps, float momentum,
                                  float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                  float* __restrict__ running_var, long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;  // nn.BatchNorm2d's bookkeeping, without a launch
    if (c >= C) return;
    float n = 0.f, m = 0.f, M2 = 0.f;
    for (int s = 0; s < split; ++s) {
        const float* p = part + ((int64_t)c * split + s) * 4;
        const float nb = p[0];
        if (nb <= 0.f) continue;
        const float delta = p[1] - m, nn = n + nb;
        m += delta * nb / nn;
        M2 += p[2] + delta * delta * n * nb / nn;
        n = nn;
    }
    const float var = n > 0 ? M2 / n : 0.f;
    mean[c] = m;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) {
        const float unbiased = n > 1 ? M2 / (n - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}


// ---- ReLU sign mask (round 4): bn3 + residual + ReLU is the one BatchNorm whose backward needs the sign of its OUTPUT (z + residual
// decides it); reading the whole output tensor back for one bit per element is 1/5 of that backward's traffic on the step's largest
// tensors.  With 8 elements per access the forward kernels also write one BYTE per vector (bit k = output element k > 0, taken from
// the ROUNDED output so that it is exactly the test the backward made on y), the backward kernels read that byte instead of y.
template <typename T, int V> __device__ __forceinline__ uint8_t sign_bits(const Vec<T, V>& o) {
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < V && k < 8; ++k) b |= ((float)o.v[k] > 0.f ? 1u : 0u) << k;
    return (uint8_t)b;
}

template <int ACT> __device__ __forceinline__ float act_fwd(float z) {
    if (ACT == ACT_RELU || ACT == ACT_RELU_Y) return z > 0.f ? z : 0.f;
    if (ACT == ACT_SILU) return silu_fwd(z);
    return z;
}

// ---- forward apply: flat over (plane, vector) -------------------------------------------------------------------
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_apply_fwd(const T* __restrict__ x, const T* __restrict__ res,
                                                   T* __restrict__ y, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, int C, int HW, int64_t nvec,
                                                   uint8_t* __restrict__ mask) {
    // two interleaved streams per thread, one grid stride apart: the loads of both are issued before either is used
    // (see bn_stats_partial); elements are independent, so the order does not matter here
    auto apply = [&](int c, int64_t i, const Vec<T, V>& xv, const Vec<T, V>& rv) __attribute__((always_inline)) {
        const float a = gamma[c] * rstd[c], b = beta[c] - mean[c] * a;
        Vec<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = (float)xv.v[k] * a + b;
            if (res) z += (float)rv.v[k];
            o.v[k] = (T)act_fwd<ACT>(z);
        }
        stv<T, V>(y + i * V, o);
        if (V == 8 && mask) mask[i] = sign_bits<T, V>(o);
    };
    const int64_t start = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    ChannelWalk wa(start, 2 * stride, HW / V, C), wb(start + stride, 2 * stride, HW / V, C);
    for (; wb.i < nvec; wa.next(), wb.next()) {
        const Vec<T, V> xa = ldv<T, V>(x + wa.i * V), xb = ldv<T, V>(x + wb.i * V);
        Vec<T, V> ra, rb;
        if (res) {
            ra = ldv<T, V>(res + wa.i * V);
            rb = ldv<T, V>(res + wb.i * V);
        }
        apply(wa.c, wa.i, xa, ra);
        apply(wb.c, wb.i, xb, rb);
    }
    if (wa.i < nvec) {
        Vec<T, V> ra;
        if (res) ra = ldv<T, V>(res + wa.i * V);
        apply(wa.c, wa.i, ldv<T, V>(x + wa.i * V), ra);
    }
}

// g = dy * act'(z): SiLU recomputes z from x; ReLU uses the saved output (y > 0) when it is given -- it must be when
// there was a residual (z + residual decides the sign) -- and otherwise recomputes z as well (one tensor read less)
template <int ACT> __device__ __forceinline__ float act_bwd(float dy, float z_or_y) {
    if (ACT == ACT_RELU || ACT == ACT_RELU_Y) return z_or_y > 0.f ? dy : 0.f;
    if (ACT == ACT_SILU) {
        const float sg = COT_RCP(1.f + __expf(-z_or_y));
        return dy * sg * (1.f + z_or_y * (1.f - sg));
    }
    return dy;
}

// ---- backward reductions: grid (C, SPLIT): sum g and sum g*xhat per channel chunk -------------------------------
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const T* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ part, int N,
                                                    int C, int HW, int nper, const float* __restrict__ ps,
                                                    const uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* smem = reinterpret_cast<float*>(cot_smem);
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const int n0 = s * nper, n1 = min(N, n0 + nper);
    const int vpp = HW / V;
    const float m = mean[c], r = rstd[c], ga = gamma[c], be = beta[c];
    const bool use_mask = V == 8 && ACT == ACT_RELU_Y && mask;
    float acc[2] = {0.f, 0.f};
#pragma unroll 2
    for (PlaneWalk w(threadIdx.x, blockDim.x, vpp); w.n < n1 - n0; w.next()) {
        const int64_t off = ((int64_t)(n0 + w.n) * C + c) * HW + (int64_t)w.v * V;
        const Vec<T, V> dv = ldv<T, V>(dy + off), xv = ldv<T, V>(x + off);
        Vec<T, V> yv;
        unsigned mb = 0;
        if (use_mask) mb = mask[off >> 3];
        else if (ACT == ACT_RELU_Y) yv = ldv<T, V>(y + off);
        const float sc = ps ? ps[n0 + w.n] : 1.f;  // stochastic depth: the normalised branch was scaled per sample
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = ((float)xv.v[k] - m) * r;
            const float yk = ACT == ACT_RELU_Y ? (use_mask ? (float)((mb >> (k & 7)) & 1u) : (float)yv.v[k]) : xh * ga + be;
            const float g = sc * act_bwd<ACT>((float)dv.v[k], yk);
            acc[0] += g;
            acc[1] += g * xh;
        }
    }
    block_sum<2>(acc, smem);
    if (threadIdx.x == 0) {
        float* p = part + ((int64_t)c * split + s) * 2;
        p[0] = acc[0];
        p[1] = acc[1];
    }
}

__global__ void bn_bwd_finalize(const float* __restrict__ part, int C, int split, float* __restrict__ dgamma,
                                float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sb = 0.f, sg = 0.f;
    for (int s = 0; s < split; ++s) {
        sb += part[((int64_t)c * split + s) * 2];
        sg += part[((int64_t)c * split + s) * 2 + 1];
    }
    dbeta[c] = sb;
    dgamma[c] = sg;
}


// ---- backward apply: dx (and dresidual = g) ---------------------------------------------------------------------
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_apply_bwd(const T* __restrict__ dy, const T* __restrict__ x,
                                                   const T* __restrict__ y, T* __restrict__ dx, T* __restrict__ dres,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                   int C, int HW, int64_t nvec, float inv_m,
                                                   const uint8_t* __restrict__ mask) {
    const bool use_mask = V == 8 && ACT == ACT_RELU_Y && mask;
#pragma unroll 2
    for (ChannelWalk w((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, HW / V, C);
         w.i < nvec; w.next()) {
        const int c = w.c;
        const int64_t i = w.i;
        const float m = mean[c], r = rstd[c], ga = gamma[c], be = beta[c];
        const float k1 = dbeta[c] * inv_m, k2 = dgamma[c] * inv_m, gr = ga * r;
        const Vec<T, V> dv = ldv<T, V>(dy + i * V), xv = ldv<T, V>(x + i * V);
        Vec<T, V> yv, o, og;
        unsigned mb = 0;
        if (use_mask) mb = mask[i];
        else if (ACT == ACT_RELU_Y) yv = ldv<T, V>(y + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = ((float)xv.v[k] - m) * r;
            const float yk = ACT == ACT_RELU_Y ? (use_mask ? (float)((mb >> (k & 7)) & 1u) : (float)yv.v[k]) : xh * ga + be;
            const float g = act_bwd<ACT>((float)dv.v[k], yk);
            o.v[k] = (T)(gr * (g - k1 - xh * k2));
            og.v[k] = (T)g;
        }
        stv<T, V>(dx + i * V, o);
        if (dres) stv<T, V>(dres + i * V, og);
    }
}


// ---- inference mode (nn.BatchNorm2d.eval(): running statistics): y = act(gamma*(x - running_mean)/sqrt(running_var + eps)
// + beta [+ residual]) in one pass -- the forward-only configuration (BASELINE config 2) otherwise runs torch's batch_norm,
// the activation and the residual add as separate passes
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_infer_fwd(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                   const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                   int C, int HW, int64_t nvec) {
    for (ChannelWalk w((int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x, HW / V, C);
         w.i < nvec; w.next()) {
        const int c = w.c;
        const float a = gamma[c] / sqrtf(rvar[c] + eps), mu = rmean[c], b = beta[c];
        const Vec<T, V> xv = ldv<T, V>(x + w.i * V);
        Vec<T, V> rv, o;
        if (res) rv = ldv<T, V>(res + w.i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = ((float)xv.v[k] - mu) * a + b;  // (centred first, as torch does: no cancellation against a large mean)
            if (res) z += (float)rv.v[k];
            o.v[k] = (T)act_fwd<ACT>(z);
        }
        stv<T, V>(y + w.i * V, o);
    }
}

// ---- "folded" variants (cot_set_tuning key 12): the per-channel finalize step lives in the apply kernel's prologue.
// Grid (C, SPLIT) like the partial kernels, so the channel is block-uniform: every thread merges the SPLIT chunk
// statistics of its channel (a few dozen flops), block (c, 0) writes mean / rstd / running statistics.  One launch less
// per BatchNorm forward and backward (~230 launches of a CoTNet-50 step).
template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_apply_fwd_fold(const T* __restrict__ x, const T* __restrict__ res,
                                                        T* __restrict__ y, const float* __restrict__ part,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        long long* __restrict__ num_batches_tracked, int N, int C, int HW,
                                                        int nper, float eps, float momentum, const float* __restrict__ ps,
                                                        uint8_t* __restrict__ mask) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    float n = 0.f, m = 0.f, M2 = 0.f;
    for (int q = 0; q < split; ++q) {
        const float* p = part + ((int64_t)c * split + q) * 4;
        const float nb = p[0];
        if (nb <= 0.f) continue;
        const float delta = p[1] - m, nn = n + nb;
        m += delta * nb / nn;
        M2 += p[2] + delta * delta * n * nb / nn;
        n = nn;
    }
    const float var = n > 0 ? M2 / n : 0.f;
    const float r = 1.0f / sqrtf(var + eps);
    if (s == 0 && threadIdx.x == 0) {
        mean[c] = m;
        rstd[c] = r;
        if (running_mean) {
            const float unbiased = n > 1 ? M2 / (n - 1.f) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    }
    const float a = gamma[c] * r, b = beta[c] - m * a;
    const int n0 = s * nper, n1 = min(N, n0 + nper);
    const int vpp = HW / V;
#pragma unroll 2
    for (PlaneWalk w(threadIdx.x, blockDim.x, vpp); w.n < n1 - n0; w.next()) {
        const int64_t off = ((int64_t)(n0 + w.n) * C + c) * HW + (int64_t)w.v * V;
        const Vec<T, V> xv = ldv<T, V>(x + off);
        Vec<T, V> rv, o;
        if (res) rv = ldv<T, V>(res + off);
        const float sc = ps ? ps[n0 + w.n] : 1.f;  // stochastic depth (models/cotnet.py:256-257): 0 or 1 / keep per sample
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float z = ((float)xv.v[k] * a + b) * sc;
            if (res) z += (float)rv.v[k];
            o.v[k] = (T)act_fwd<ACT>(z);
        }
        stv<T, V>(y + off, o);
        if (V == 8 && mask) mask[off >> 3] = sign_bits<T, V>(o);
    }
}

template <typename T, int V, int ACT>
__global__ __launch_bounds__(256) void bn_apply_bwd_fold(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const T* __restrict__ y, T* __restrict__ dx, T* __restrict__ dres,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ part, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int N, int C, int HW, int nper,
                                                        float inv_m, const float* __restrict__ ps,
                                                        const uint8_t* __restrict__ mask) {
    const bool use_mask = V == 8 && ACT == ACT_RELU_Y && mask;
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    float sb = 0.f, sg = 0.f;
    for (int q = 0; q < split; ++q) {
        sb += part[((int64_t)c * split + q) * 2];
        sg += part[((int64_t)c * split + q) * 2 + 1];
    }
    if (s == 0 && threadIdx.x == 0) {
        dbeta[c] = sb;
        dgamma[c] = sg;
    }
    const float m = mean[c], r = rstd[c], ga = gamma[c], be = beta[c];
    const float k1 = sb * inv_m, k2 = sg * inv_m, gr = ga * r;
    const int n0 = s * nper, n1 = min(N, n0 + nper);
    const int vpp = HW / V;
#pragma unroll 2
    for (PlaneWalk w(threadIdx.x, blockDim.x, vpp); w.n < n1 - n0; w.next()) {
        const int64_t off = ((int64_t)(n0 + w.n) * C + c) * HW + (int64_t)w.v * V;
        const Vec<T, V> dv = ldv<T, V>(dy + off), xv = ldv<T, V>(x + off);
        Vec<T, V> yv, o, og;
        unsigned mb = 0;
        if (use_mask) mb = mask[off >> 3];
        else if (ACT == ACT_RELU_Y) yv = ldv<T, V>(y + off);
        const float sc = ps ? ps[n0 + w.n] : 1.f;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = ((float)xv.v[k] - m) * r;
            const float yk = ACT == ACT_RELU_Y ? (use_mask ? (float)((mb >> (k & 7)) & 1u) : (float)yv.v[k]) : xh * ga + be;
            const float g = act_bwd<ACT>((float)dv.v[k], yk);
            o.v[k] = (T)(gr * (sc * g - k1 - xh * k2));
            og.v[k] = (T)g;  // (the residual's gradient is not scaled)
        }
        stv<T, V>(dx + off, o);
        if (dres) stv<T, V>(dres + off, og);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the sign mask of the current call (set by cot_bn_act_{forward,backward}_mask around the call; NULL otherwise)
thread_local uint8_t* t_bn_mask = nullptr;
int g_bn_fold = 1;          // cot_set_tuning key 12 (default on since round 3: 202 fewer launches per CoTNet-50 step, profiles/r02_bn_fold_ab.txt)
int g_bn_grid_cap = 4096;  // cot_set_tuning key 13: most workgroups of a flat (grid-stride) BatchNorm apply kernel
static inline int pick_vec(size_t esize, int HW) {
    int lim = (int)(16 / esize);
    for (int V = 8; V >= 1; V >>= 1)
        if (V <= lim && HW % V == 0) return V;
    return 1;
}
// cot_set_tuning key 28: workgroups a streaming BatchNorm kernel aims for (C x SPLIT).  1024 since round 3 (was 2048): fewer, longer
// workgroups and half as many partial statistics to merge in the apply kernel's prologue -- forward 64 ch x 56^2 29.7 -> 23.3 us,
// 32 ch x 56^2 25.6 -> 20.6, 256 ch x 56^2 75.9 -> 73.3, backward equal or 1-3 % better (profiles/r03_bn_split_target_sweep.log)
int g_bn_split_target = 1024;
static inline void pick_split(int N, int C, int* split, int* nper) {
    int s = 1;
    while (C * s < g_bn_split_target && s * 2 <= N) s *= 2;   // >= 8 blocks per CU of reduction work when the batch allows
    *nper = (N + s - 1) / s;
    *split = (N + *nper - 1) / *nper;
}
static inline unsigned flat_grid(int64_t nvec) {
    int64_t b = ceil_div64(nvec, 256);
    if (b > g_bn_grid_cap) b = g_bn_grid_cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// bytes of the ReLU sign mask for this tensor, 0 = not supported (the backward then reads the saved output): 8 elements per access
// on every kernel family that can serve the tensor, i.e. 2-byte elements, planes that are multiples of 8, and not the one-wave fp64
// path of tiny batches
extern int g_bn_small_m;
int64_t bn_relu_mask_bytes(int N, int C, int HW, int esize) {
    if (esize != 2 || HW % 8 != 0 || (int64_t)N * HW <= g_bn_small_m) return 0;
    return (int64_t)N * C * HW / 8;
}

int bn_workspace_floats(int N, int C) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    return C * split * 4;
}

template <typename T, int V, int ACT>
static int bn_fwd_launch_act(const T* x, const T* res, T* y, const float* gamma, const float* beta, float* mean,
                             float* rstd, float* rmean, float* rvar, long long* nbt, float* ws, int N, int C, int HW,
                             float eps, float mom, const float* ps, hipStream_t s) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    uint8_t* const mk = V == 8 ? t_bn_mask : nullptr;  // (read HERE: a launch macro may evaluate its arguments on another thread)
    COT_LAUNCH((bn_stats_partial<T, V>), dim3(C, split), dim3(256), 48 * sizeof(float), s, x, ws, N, C, HW, nper);
    if (g_bn_fold || ps) {  // (the per-sample scale lives in the kernels that walk a channel image by image)
        COT_LAUNCH((bn_apply_fwd_fold<T, V, ACT>), dim3(C, split), dim3(256), 0, s, x, res, y, (const float*)ws, gamma,
                   beta, mean, rstd, rmean, rvar, nbt, N, C, HW, nper, eps, mom, ps, mk);
        return check_launch("bn_act_forward");
    }
    COT_LAUNCH(bn_stats_finalize, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)ws, C, split, eps, mom, mean,
               rstd, rmean, rvar, nbt);
    const int64_t nvec = (int64_t)N * C * HW / V;
    COT_LAUNCH((bn_apply_fwd<T, V, ACT>), dim3(flat_grid(nvec)), dim3(256), 0, s, x, res, y, (const float*)mean,
               (const float*)rstd, gamma, beta, C, HW, nvec, mk);
    return check_launch("bn_act_forward");
}

template <typename T, int V>
static int bn_fwd_launch(const T* x, const T* res, T* y, const float* gamma, const float* beta, float* mean,
                         float* rstd, float* rmean, float* rvar, long long* nbt, float* ws, int N, int C, int HW,
                         float eps, float mom, int act, const float* ps, hipStream_t s) {
#define BN_FA(A_) return bn_fwd_launch_act<T, V, A_>(x, res, y, gamma, beta, mean, rstd, rmean, rvar, nbt, ws, N, C, HW, eps, mom, ps, s)
    if (act == ACT_RELU) BN_FA(ACT_RELU);
    if (act == ACT_SILU) BN_FA(ACT_SILU);
    BN_FA(ACT_NONE);
#undef BN_FA
}

template <typename T, int V, int ACT>
static int bn_bwd_launch_act(const T* dy, const T* x, const T* y, T* dx, T* dres, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, float* dgamma, float* dbeta, float* ws, int N, int C,
                             int HW, const float* ps, hipStream_t s) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    const uint8_t* const mk = V == 8 ? t_bn_mask : nullptr;  // (read here, see bn_fwd_launch_act)
    COT_LAUNCH((bn_bwd_reduce<T, V, ACT>), dim3(C, split), dim3(256), 48 * sizeof(float), s, dy, x, y, mean, rstd, gamma,
               beta, ws, N, C, HW, nper, ps, mk);
    if (g_bn_fold || ps) {
        COT_LAUNCH((bn_apply_bwd_fold<T, V, ACT>), dim3(C, split), dim3(256), 0, s, dy, x, y, dx, dres, mean, rstd, gamma,
                   beta, (const float*)ws, dgamma, dbeta, N, C, HW, nper, 1.0f / (float)((int64_t)N * HW), ps, mk);
        return check_launch("bn_act_backward");
    }
    COT_LAUNCH(bn_bwd_finalize, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)ws, C, split, dgamma, dbeta);
    const int64_t nvec = (int64_t)N * C * HW / V;
    COT_LAUNCH((bn_apply_bwd<T, V, ACT>), dim3(flat_grid(nvec)), dim3(256), 0, s, dy, x, y, dx, dres, mean, rstd, gamma,
               beta, (const float*)dgamma, (const float*)dbeta, C, HW, nvec, 1.0f / (float)((int64_t)N * HW), mk);
    return check_launch("bn_act_backward");
}

template <typename T, int V>
static int bn_bwd_launch(const T* dy, const T* x, const T* y, T* dx, T* dres, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, float* dgamma, float* dbeta, float* ws, int N, int C,
                         int HW, int act, const float* ps, hipStream_t s) {
#define BN_BA(A_) return bn_bwd_launch_act<T, V, A_>(dy, x, y, dx, dres, gamma, beta, mean, rstd, dgamma, dbeta, ws, N, C, HW, ps, s)
    if (act == ACT_RELU && y) BN_BA(ACT_RELU_Y);  // sign of the saved output (required when there was a residual)
    if (act == ACT_RELU) BN_BA(ACT_RELU);
    if (act == ACT_SILU) BN_BA(ACT_SILU);
    BN_BA(ACT_NONE);
#undef BN_BA
}

// ---- small batches: N*H*W <= g_bn_small_m (256) samples per channel (the `se` branch of a CoT layer normalises over the batch
// alone: [B, A, 1, 1], models/cotnet.py:71-77).  With a handful of samples the backward formula
//     dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))
// cancels catastrophically (for 2 samples the two terms agree to eps/(var+eps)); measured on the MI355X: MIOpen's fp32
// kernel is 300x further from an fp64 evaluation than the CPU's (3.8e-1 on a gradient of scale 2.8e2) and is what kept
// the 7x7 CotLayer off the 1e-3 parity bar.  Here one wave owns a channel and does the WHOLE computation in fp64
// (statistics recomputed from x in the backward, not taken from the fp32 saves): one launch each way, a few KB of traffic.
int g_bn_small_m = 256;  // cot_set_tuning(18, M): per-channel sample count up to which the fp64 path is taken (0 = never); one thread walks a whole channel, so this must stay small

// sum over the 64 lanes in fp64, result in every lane (butterfly with a fixed order: deterministic)
__device__ __forceinline__ double wave_allsum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one WAVE per channel: lane l takes the samples l, l+64, .. of the channel's N*H*W
template <typename T, int ACT>
__global__ __launch_bounds__(64) void bn_small_fwd(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ mean, float* __restrict__ rstd,
                                                  float* __restrict__ rmean, float* __restrict__ rvar,
                                                  long long* __restrict__ nbt, int N, int C, int HW, float eps, float mom) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0 && nbt) *nbt += 1;
    const int M = N * HW;
    double sm = 0.0;
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        sm += (double)(float)x[((int64_t)n * C + c) * HW + p];
    }
    const double mu = wave_allsum_d(sm) / M;
    double m2 = 0.0;
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        const double d = (double)(float)x[((int64_t)n * C + c) * HW + p] - mu;
        m2 += d * d;
    }
    m2 = wave_allsum_d(m2);
    const double var = m2 / M, r = 1.0 / sqrt(var + (double)eps);
    if (lane == 0) {
        mean[c] = (float)mu;
        rstd[c] = (float)r;
        if (rmean) {
            const double unbiased = M > 1 ? m2 / (M - 1) : var;
            rmean[c] = (float)((1.0 - mom) * rmean[c] + mom * mu);
            rvar[c] = (float)((1.0 - mom) * rvar[c] + mom * unbiased);
        }
    }
    const double ga = gamma[c], be = beta[c];
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        const int64_t i = ((int64_t)n * C + c) * HW + p;
        double z = ((double)(float)x[i] - mu) * r * ga + be;
        if (res) z += (double)(float)res[i];
        if (ACT == ACT_RELU) z = z > 0.0 ? z : 0.0;
        if (ACT == ACT_SILU) z = z / (1.0 + exp(-z));
        y[i] = (T)(float)z;
    }
}

template <typename T, int ACT>
__global__ __launch_bounds__(64) void bn_small_bwd(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                  T* __restrict__ dx, T* __restrict__ dres, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ rstd,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int HW) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int M = N * HW;
    double sm = 0.0;
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        sm += (double)(float)x[((int64_t)n * C + c) * HW + p];
    }
    // mean recomputed in fp64 (so that sum(xhat) == 0 to fp64 accuracy); rstd is the forward's (its own rounding only
    // scales the result by 1 + 6e-8, it does not enter the cancellation)
    const double mu = wave_allsum_d(sm) / M, r = (double)rstd[c], ga = gamma[c], be = beta[c];
    auto gof = [&](int64_t i, double xh) {
        const double d = (double)(float)dy[i];
        if (ACT == ACT_RELU_Y) return (float)y[i] > 0.f ? d : 0.0;
        const double z = xh * ga + be;
        if (ACT == ACT_RELU) return z > 0.0 ? d : 0.0;
        if (ACT == ACT_SILU) {
            const double sg = 1.0 / (1.0 + exp(-z));
            return d * sg * (1.0 + z * (1.0 - sg));
        }
        return d;
    };
    double sg = 0.0, sgx = 0.0;
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        const int64_t i = ((int64_t)n * C + c) * HW + p;
        const double xh = ((double)(float)x[i] - mu) * r, g = gof(i, xh);
        sg += g;
        sgx += g * xh;
    }
    sg = wave_allsum_d(sg);
    sgx = wave_allsum_d(sgx);
    if (lane == 0) {
        dbeta[c] = (float)sg;
        dgamma[c] = (float)sgx;
    }
    const double k1 = sg / M, k2 = sgx / M;
    for (int e = lane; e < M; e += 64) {
        const int n = e / HW, p = e - n * HW;
        const int64_t i = ((int64_t)n * C + c) * HW + p;
        const double xh = ((double)(float)x[i] - mu) * r, g = gof(i, xh);
        dx[i] = (T)(float)(ga * r * (g - k1 - xh * k2));
        if (dres) dres[i] = (T)(float)g;
    }
}

template <typename T>
int bn_act_inference(const void* x, const void* res, void* y, const float* gamma, const float* beta, const float* rmean,
                     const float* rvar, int N, int C, int HW, float eps, int act, hipStream_t s) {
    const int v = pick_vec(sizeof(T), HW);
#define BN_I(V_, A_)                                                                                                  \
    COT_LAUNCH((bn_infer_fwd<T, V_, A_>), dim3(flat_grid((int64_t)N * C * HW / V_)), dim3(256), 0, s, (const T*)x,     \
               (const T*)res, (T*)y, rmean, rvar, gamma, beta, eps, C, HW, (int64_t)N * C * HW / V_)
#define BN_IV(V_)                                 \
    do {                                          \
        if (act == ACT_RELU) BN_I(V_, ACT_RELU);  \
        else if (act == ACT_SILU) BN_I(V_, ACT_SILU); \
        else BN_I(V_, ACT_NONE);                  \
    } while (0)
    if (v == 8 && sizeof(T) <= 2) BN_IV(8);
    else if (v >= 4) BN_IV(4);
    else if (v == 2) BN_IV(2);
    else BN_IV(1);
#undef BN_IV
#undef BN_I
    return check_launch("bn_infer_fwd");
}
template int bn_act_inference<float>(const void*, const void*, void*, const float*, const float*, const float*, const float*,
                                     int, int, int, float, int, hipStream_t);
template int bn_act_inference<bf16_t>(const void*, const void*, void*, const float*, const float*, const float*, const float*,
                                      int, int, int, float, int, hipStream_t);

// ---- channel-resident kernels: N*H*W small enough for one workgroup to hold a channel in registers -------------------------
// The 14 x 14 and 7 x 7 stages (and 28 x 28 in bf16) have 4-63 K samples per channel: the streaming path's four launches
// (statistics, apply, backward reduce, backward apply) each move a few MB and cost 5-10 us of launch ramp apiece --
// 54 of CoTNet-50's 101 BatchNorms, ~1.7 ms of a 21 ms step.  Here one workgroup of up to 1024 lanes owns a channel: it loads
// the channel once (the N row segments of H*W elements, vector width V), keeps it packed in registers, reduces through LDS
// and writes the result: forward = 1 read + 1 write and ONE launch (streaming: 2 reads + 1 write, 2 launches), backward =
// 2 reads + 1 write and one launch (streaming: 4 reads + 1 write, 2 launches).  Statistics: mean, then the centred sum of
// squares from the registers (two-pass, no cancellation), per-lane fp32 partials added in fp64 across the workgroup.
int g_bn_chan = 1;  // cot_set_tuning key 21: 1 = on where eligible (default), 0 = streaming kernels only
int g_bn_chan7 = 1;  // cot_set_tuning key 40: 7-element accesses on odd planes that are multiples of 7 (1 default, 0 = element-wise)

template <typename T, int V, bool BWD> struct ChanRounds {  // most rounds (vectors per lane) of a channel-resident kernel:
    // 64 elements per lane forward; the backward holds two tensors and takes 32 in bf16 (128 registers per lane is all a
    // 1024-lane workgroup gets)
    static constexpr int value = V == 7 ? (BWD ? 4 : 8)
                                 : (sizeof(T) <= 2 ? (V == 8 ? (BWD ? 4 : 8) : (V == 4 ? (BWD ? 8 : 16) : 16)) : (V == 4 ? 8 : 16));
};

// sums of NV doubles over the workgroup, result in every lane; fixed order (deterministic)
template <int NV> __device__ __forceinline__ void block_allsum_d(double (&v)[NV], double* smem /* NV * 16 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_allsum_d(v[k]);
    __syncthreads();  // the previous use of smem is over
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) smem[k * 16 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += smem[k * 16 + w];
        v[k] = t;
    }
}

// RR: rounds (vectors per lane) this instance is unrolled for -- 0 = the most the registers hold (ChanRounds), else 2 / 4: a
// 14 x 14 channel of 80 images is 3920 four-element vectors = FOUR rounds of a 1024-lane workgroup, and the sixteen-round instance
// spent 128 registers (some of them in scratch) and twelve predicated-off rounds on it (round 5, profiles/r05_bn_chan_rounds_ab.log)
template <typename T, int V, int ACT, int RR>
__global__ __launch_bounds__(1024) void bn_chan_fwd(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ mean, float* __restrict__ rstd,
                                                   float* __restrict__ rmean, float* __restrict__ rvar,
                                                   long long* __restrict__ nbt, int N, int C, int HW, float eps, float mom,
                                                   const float* __restrict__ ps, uint8_t* __restrict__ mask) {
    constexpr int R = RR ? RR : ChanRounds<T, V, false>::value;
    __shared__ double red[16];
    const int c = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    const int vpp = HW / V, MV = N * vpp;
    const double M = (double)N * HW;
    if (c == 0 && t == 0 && nbt) *nbt += 1;
    Vec<T, V> xv[R];
    int off[R];  // element offset of round r's vector (host: N*C*HW < 2^31), -1 = past the channel's end
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * NT + t;
        off[r] = -1;
        if (e < MV) {
            const int n = e / vpp, v = e - n * vpp;
            off[r] = (n * C + c) * HW + v * V;
            xv[r] = ldv<T, V>(x + off[r]);
#pragma unroll
            for (int k = 0; k < V; ++k) s += (float)xv[r].v[k];
        }
    }
    double acc[1] = {(double)s};
    block_allsum_d<1>(acc, red);
    const float mu = (float)(acc[0] / M);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (off[r] >= 0) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float d = (float)xv[r].v[k] - mu;
                q += d * d;
            }
        }
    acc[0] = (double)q;
    block_allsum_d<1>(acc, red);
    const float var = (float)(acc[0] / M), rs = 1.0f / sqrtf(var + eps);
    if (t == 0) {
        mean[c] = mu;
        rstd[c] = rs;
        if (rmean) {
            const float unbiased = M > 1.0 ? (float)(acc[0] / (M - 1.0)) : var;
            rmean[c] = (1.f - mom) * rmean[c] + mom * mu;
            rvar[c] = (1.f - mom) * rvar[c] + mom * unbiased;
        }
    }
    const float a = gamma[c] * rs, b = beta[c] - mu * a;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (off[r] >= 0) {
            Vec<T, V> rv, o;
            if (res) rv = ldv<T, V>(res + off[r]);
            const float sc = ps ? ps[(r * NT + t) / vpp] : 1.f;  // (stochastic depth: per-sample scale of the normalised branch)
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float z = ((float)xv[r].v[k] * a + b) * sc;
                if (res) z += (float)rv.v[k];
                o.v[k] = (T)act_fwd<ACT>(z);
            }
            stv<T, V>(y + off[r], o);
            if (V == 8 && mask) mask[off[r] >> 3] = sign_bits<T, V>(o);
        }
}

template <typename T, int V, int ACT, int RR>
__global__ __launch_bounds__(1024) void bn_chan_bwd(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                   T* __restrict__ dx, T* __restrict__ dres, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int N, int C, int HW, const float* __restrict__ ps,
                                                   const uint8_t* __restrict__ mask) {
    constexpr int R = RR ? RR : ChanRounds<T, V, true>::value;
    __shared__ double red[32];
    const int c = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    const int vpp = HW / V, MV = N * vpp;
    const float inv_m = 1.0f / ((float)N * (float)HW);
    const float m = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    Vec<T, V> xv[R], dv[R];
    int off[R];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * NT + t;
        off[r] = -1;
        if (e < MV) {
            const int n = e / vpp, v = e - n * vpp;
            off[r] = (n * C + c) * HW + v * V;
            xv[r] = ldv<T, V>(x + off[r]);
            dv[r] = ldv<T, V>(dy + off[r]);
            if (ACT == ACT_RELU_Y) {  // the saved output's sign decides: fold it into dy now (exact), y is not kept
                if (V == 8 && mask) {
                    const unsigned mb = mask[off[r] >> 3];
#pragma unroll
                    for (int k = 0; k < V; ++k) dv[r].v[k] = ((mb >> (k & 7)) & 1u) ? dv[r].v[k] : (T)0.f;
                } else {
                    const Vec<T, V> yv = ldv<T, V>(y + off[r]);
#pragma unroll
                    for (int k = 0; k < V; ++k) dv[r].v[k] = (float)yv.v[k] > 0.f ? dv[r].v[k] : (T)0.f;
                }
            }
            const float sc = ps ? ps[n] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = ((float)xv[r].v[k] - m) * rs;
                const float g = sc * (ACT == ACT_RELU_Y ? (float)dv[r].v[k] : act_bwd<ACT>((float)dv[r].v[k], xh * ga + be));
                sg += g;
                sgx += g * xh;
            }
        }
    }
    double acc[2] = {(double)sg, (double)sgx};
    block_allsum_d<2>(acc, red);
    const float sb = (float)acc[0], sx = (float)acc[1];
    if (t == 0) {
        dbeta[c] = sb;
        dgamma[c] = sx;
    }
    const float k1 = sb * inv_m, k2 = sx * inv_m, gr = ga * rs;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (off[r] >= 0) {
            Vec<T, V> o, og;
            const float sc = ps ? ps[(r * NT + t) / vpp] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = ((float)xv[r].v[k] - m) * rs;
                const float g = ACT == ACT_RELU_Y ? (float)dv[r].v[k] : act_bwd<ACT>((float)dv[r].v[k], xh * ga + be);
                o.v[k] = (T)(gr * (sc * g - k1 - xh * k2));
                og.v[k] = (T)g;
            }
            stv<T, V>(dx + off[r], o);
            if (dres) stv<T, V>(dres + off[r], og);
        }
}

template <typename T, bool BWD> static int bn_chan_rounds(int V) {
    return V == 8 ? ChanRounds<T, 8, BWD>::value
                  : (V == 7 ? ChanRounds<T, 7, BWD>::value : (V == 4 ? ChanRounds<T, 4, BWD>::value : ChanRounds<T, 1, BWD>::value));
}
// vector width of the channel-resident kernels for this tensor, 0 = not eligible (streaming kernels)
template <typename T, bool BWD> static int bn_chan_vec(int N, int C, int HW) {
    if (!g_bn_chan || (int64_t)N * C * HW >= ((int64_t)1 << 31)) return 0;
    int V = pick_vec(sizeof(T), HW);
    if (V == 2) V = 1;  // (V = 2 has no instantiation: H*W = 2 * odd does not occur in the networks)
    if (sizeof(T) > 2 && V == 8) V = 4;
    // odd planes that are multiples of 7 (7 x 7: every BatchNorm of the last stage): 7 elements per lane and access instead of
    // one 2-byte element (0.13 of the HBM roofline on 2048 x 49 planes that way)
    if (V == 1 && HW % 7 == 0 && g_bn_chan7) V = 7;
    const int64_t MV = (int64_t)N * HW / V;
    const int rounds = bn_chan_rounds<T, BWD>(V);
    return MV <= (int64_t)1024 * rounds ? V : 0;
}
// lanes per workgroup (256 / 512 / 1024): at least what `rounds` vectors per lane need to cover the channel, and beyond
// that as many as it takes for the C workgroups to fill the chip (~1024 lanes per CU): few channels -> big workgroups (more
// loads in flight per CU), many channels -> small ones (more workgroups resident per CU, their load / reduce / store phases
// interleave).  Measured on the MI355X (scripts/ubench_bn.py, profiles/r02_ubench_bn.log), forward us at 256 / 512 / 1024
// lanes: 128 ch 14x14 11.2 / 7.9 / 7.5, 1024 ch 14x14 17.4 / 21.0 / 27.5, 2048 ch 7x7 23.8 / 24.6 / 44.2.
static inline int bn_chan_threads(int N, int C, int HW, int V, int rounds) {
    const int64_t MV = (int64_t)N * HW / V;
    if (g_bn_chan > 1 && g_bn_chan <= 1024 && MV <= (int64_t)g_bn_chan * rounds) return g_bn_chan;  // (A/B: forced size)
    const int need = MV <= (int64_t)256 * rounds ? 256 : (MV <= (int64_t)512 * rounds ? 512 : 1024);
    const int fill = C <= 256 ? 1024 : (C <= 512 ? 512 : 256);
    return need > fill ? need : fill;
}

// the instance's round count for a channel of MV vectors on `threads` lanes: 2, 4 or 0 (= the full ChanRounds instance)
int g_bn_chan_rr = 1;  // cot_set_tuning key 47: 1 (default) = short instances where they cover the channel, 0 = always the full one
static inline int bn_chan_rr(int64_t MV, int threads, int full) {
    if (!g_bn_chan_rr) return 0;
    const int64_t need = (MV + threads - 1) / threads;
    if (need <= 2 && full > 2) return 2;
    if (need <= 4 && full > 4) return 4;
    return 0;
}

template <typename T>
int bn_act_forward(const void* x, const void* res, void* y, const float* gamma, const float* beta, float* mean,
                   float* rstd, float* rmean, float* rvar, long long* nbt, float* ws, int N, int C, int HW, float eps,
                   float mom, int act, const float* ps, hipStream_t s) {
#define BN_F(VV) return bn_fwd_launch<T, VV>((const T*)x, (const T*)res, (T*)y, gamma, beta, mean, rstd, rmean, rvar, nbt, ws, N, C, HW, eps, mom, act, ps, s)
    // a sign mask is one byte per EIGHT elements: only the 8-element instances write / read it.  bn_relu_mask_bytes() hands out a
    // mask size exactly for the geometries that take them; a call that reaches another instance with a mask set must not run
    // (forward: the backward would read a mask nobody wrote; backward: the mask pointer stands in for y and a kernel that reads y
    // would walk 16 times the mask's size -- ADVICE r4)
    if (t_bn_mask && (sizeof(T) != 2 || pick_vec(sizeof(T), HW) != 8 || (int64_t)N * HW <= g_bn_small_m)) return -3;
    if ((int64_t)N * HW <= g_bn_small_m && !ps) {
        const dim3 grid(C), block(64);  // one wave per channel
#define BN_SF(A_) COT_LAUNCH((bn_small_fwd<T, A_>), grid, block, 0, s, (const T*)x, (const T*)res, (T*)y, gamma, beta, mean, rstd, rmean, rvar, nbt, N, C, HW, eps, mom)
        if (act == ACT_RELU) BN_SF(ACT_RELU);
        else if (act == ACT_SILU) BN_SF(ACT_SILU);
        else BN_SF(ACT_NONE);
#undef BN_SF
        return check_launch("bn_small_fwd");
    }
    if (const int cv = bn_chan_vec<T, false>(N, C, HW)) {
        const dim3 grid(C), block(bn_chan_threads(N, C, HW, cv, bn_chan_rounds<T, false>(cv)));
        uint8_t* const mk = cv == 8 ? t_bn_mask : nullptr;
        const int rr = bn_chan_rr((int64_t)N * HW / cv, (int)block.x, bn_chan_rounds<T, false>(cv));
#define BN_CFR(V_, A_, R_) COT_LAUNCH((bn_chan_fwd<T, V_, A_, R_>), grid, block, 0, s, (const T*)x, (const T*)res, (T*)y, gamma, beta, mean, rstd, rmean, rvar, nbt, N, C, HW, eps, mom, ps, (V_) == 8 ? mk : nullptr)
#define BN_CF(V_, A_)                      \
    do {                                   \
        if (rr == 2) BN_CFR(V_, A_, 2);    \
        else if (rr == 4) BN_CFR(V_, A_, 4); \
        else BN_CFR(V_, A_, 0);            \
    } while (0)
#define BN_CFV(V_)                                \
    do {                                          \
        if (act == ACT_RELU) BN_CF(V_, ACT_RELU); \
        else if (act == ACT_SILU) BN_CF(V_, ACT_SILU); \
        else BN_CF(V_, ACT_NONE);                 \
    } while (0)
        if (cv == 8) BN_CFV((sizeof(T) <= 2 ? 8 : 4));
        else if (cv == 7) BN_CFV(7);
        else if (cv == 4) BN_CFV(4);
        else BN_CFV(1);
#undef BN_CFV
#undef BN_CF
#undef BN_CFR
        return check_launch("bn_chan_fwd");
    }
    const int v = pick_vec(sizeof(T), HW);
    if (v == 8) BN_F((sizeof(T) <= 2 ? 8 : 1));
    if (v == 4) BN_F((sizeof(T) <= 4 ? 4 : 1));
    if (v == 2) BN_F(2);
    BN_F(1);
#undef BN_F
}

template <typename T>
int bn_act_backward(const void* dy, const void* x, const void* y, void* dx, void* dres, const float* gamma,
                    const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* ws,
                    int N, int C, int HW, int act, const float* ps, hipStream_t s) {
#define BN_B(VV) return bn_bwd_launch<T, VV>((const T*)dy, (const T*)x, (const T*)y, (T*)dx, (T*)dres, gamma, beta, mean, rstd, dgamma, dbeta, ws, N, C, HW, act, ps, s)
    if (t_bn_mask && (sizeof(T) != 2 || pick_vec(sizeof(T), HW) != 8 || (int64_t)N * HW <= g_bn_small_m)) return -3;  // (see bn_act_forward)
    if ((int64_t)N * HW <= g_bn_small_m && !ps) {
        const dim3 grid(C), block(64);  // one wave per channel
#define BN_SB(A_) COT_LAUNCH((bn_small_bwd<T, A_>), grid, block, 0, s, (const T*)dy, (const T*)x, (const T*)y, (T*)dx, (T*)dres, gamma, beta, rstd, dgamma, dbeta, N, C, HW)
        if (act == ACT_RELU && y) BN_SB(ACT_RELU_Y);
        else if (act == ACT_RELU) BN_SB(ACT_RELU);
        else if (act == ACT_SILU) BN_SB(ACT_SILU);
        else BN_SB(ACT_NONE);
#undef BN_SB
        return check_launch("bn_small_bwd");
    }
    if (const int cv = bn_chan_vec<T, true>(N, C, HW)) {
        const dim3 grid(C), block(bn_chan_threads(N, C, HW, cv, bn_chan_rounds<T, true>(cv)));
        const uint8_t* const mk = cv == 8 ? t_bn_mask : nullptr;
        const int rr = bn_chan_rr((int64_t)N * HW / cv, (int)block.x, bn_chan_rounds<T, true>(cv));
#define BN_CBR(V_, A_, R_) COT_LAUNCH((bn_chan_bwd<T, V_, A_, R_>), grid, block, 0, s, (const T*)dy, (const T*)x, (const T*)y, (T*)dx, (T*)dres, gamma, beta, mean, rstd, dgamma, dbeta, N, C, HW, ps, (V_) == 8 ? mk : nullptr)
#define BN_CB(V_, A_)                      \
    do {                                   \
        if (rr == 2) BN_CBR(V_, A_, 2);    \
        else if (rr == 4 && bn_chan_rounds<T, true>(V_) > 4) BN_CBR(V_, A_, 4); \
        else BN_CBR(V_, A_, 0);            \
    } while (0)
#define BN_CBV(V_)                                         \
    do {                                                   \
        if (act == ACT_RELU && y) BN_CB(V_, ACT_RELU_Y);   \
        else if (act == ACT_RELU) BN_CB(V_, ACT_RELU);     \
        else if (act == ACT_SILU) BN_CB(V_, ACT_SILU);     \
        else BN_CB(V_, ACT_NONE);                          \
    } while (0)
        if (cv == 8) BN_CBV((sizeof(T) <= 2 ? 8 : 4));
        else if (cv == 7) BN_CBV(7);
        else if (cv == 4) BN_CBV(4);
        else BN_CBV(1);
#undef BN_CBV
#undef BN_CB
#undef BN_CBR
        return check_launch("bn_chan_bwd");
    }
    const int v = pick_vec(sizeof(T), HW);
    if (v == 8) BN_B((sizeof(T) <= 2 ? 8 : 1));
    if (v == 4) BN_B((sizeof(T) <= 4 ? 4 : 1));
    if (v == 2) BN_B(2);
    BN_B(1);
#undef BN_B
}


// ---- the statistics of a BatchNorm alone (round 6): mean / rstd / running statistics of x [N, C, HW] -- bn_stats_partial, then
// bn_stats_finalize (the same chunk statistics and the same merge as the folded apply kernel's prologue: bit-identical mean / rstd).
// For consumers that normalise while they load -- the radix tail's cot_radix_*_bn kernels (radix_tail.hip) -- instead of reading a
// normalised tensor somebody wrote for them.  ws: bn_workspace_floats(N, C) floats.
// (Tried and dropped: ONE launch, the channel's last chunk to finish -- an atomic ticket per channel -- doing the merge.  The
// __threadfence() pair around the ticket is a device-scope release / acquire, which on this eight-XCD part writes back and invalidates
// the workgroup's L2: 33.7 us against 8.0 us for the plain pass at 64 ch x 56 x 56, B = 80 -- profiles/r06_bn_tail_kernels.log.  A
// dependent 256-thread launch costs ~3 us.)
template <typename T>
int bn_batch_stats(const void* x, float* mean, float* rstd, float* rmean, float* rvar, long long* nbt, float* ws, int N, int C, int HW,
                   float eps, float mom, hipStream_t s) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    int v = pick_vec(sizeof(T), HW);
    if (v == 1 && HW % 7 == 0 && sizeof(T) == 2) v = 7;  // 7 x 7 planes: 7 elements per lane and access (Vec<T, 7>)
    if (sizeof(T) > 2 && v == 8) v = 4;
#define BN_SO(V_) COT_LAUNCH((bn_stats_partial<T, V_>), dim3(C, split), dim3(256), 48 * sizeof(float), s, (const T*)x, ws, N, C, HW, nper)
    if (v == 8) BN_SO((sizeof(T) <= 2 ? 8 : 4));
    else if (v == 7) BN_SO(7);
    else if (v == 4) BN_SO(4);
    else if (v == 2) BN_SO(2);
    else BN_SO(1);
#undef BN_SO
    COT_LAUNCH(bn_stats_finalize, dim3((C + 255) / 256), dim3(256), 0, s, (const float*)ws, C, split, eps, mom, mean, rstd, rmean, rvar, nbt);
    return check_launch("bn_batch_stats");
}
template int bn_batch_stats<float>(const void*, float*, float*, float*, float*, long long*, float*, int, int, int, float, float, hipStream_t);
template int bn_batch_stats<bf16_t>(const void*, float*, float*, float*, float*, long long*, float*, int, int, int, float, float, hipStream_t);

int bn_stats_split(int N, int C) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    return split;
}
template <typename T>
int bn_stats_sums_launch(const void* x, float* ws, int N, int C, int HW, hipStream_t s) {
    int split, nper;
    pick_split(N, C, &split, &nper);
    int v = pick_vec(sizeof(T), HW);
    if (v == 1 && HW % 7 == 0 && sizeof(T) == 2) v = 7;
    if (sizeof(T) > 2 && v == 8) v = 4;
#define BN_SS(V_) COT_LAUNCH((bn_stats_sums<T, V_>), dim3(C, split), dim3(256), 48 * sizeof(float), s, (const T*)x, ws, N, C, HW, nper)
    if (v == 8) BN_SS((sizeof(T) <= 2 ? 8 : 4));
    else if (v == 7) BN_SS(7);
    else if (v == 4) BN_SS(4);
    else if (v == 2) BN_SS(2);
    else BN_SS(1);
#undef BN_SS
    return check_launch("bn_stats_sums");
}
template int bn_stats_sums_launch<float>(const void*, float*, int, int, int, hipStream_t);
template int bn_stats_sums_launch<bf16_t>(const void*, float*, int, int, int, hipStream_t);

// ---- statistics from the PRODUCER's epilogue (round 5): the 1x1 convolution that writes a BatchNorm's
// input also writes, per (image, 128-pixel tile, channel), the sum and the sum of squares of the bf16 values it stores
// (conv_lds_common.h tile_epilogue, BIG tiles: planes of more than 256 pixels).  This kernel turns them into the channel's batch
// statistics -- one wave per channel, N * PT entries added in fp64 in a fixed order -- and updates the running statistics exactly as
// bn_stats_finalize does; bn_apply_fwd then normalises.  It replaces bn_stats_partial, i.e. one full read of the tensor.
__global__ __launch_bounds__(64) void bn_tile_stats_finalize(const float* __restrict__ part, int N, int C, int PT, int HW, float eps,
                                                            float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            long long* __restrict__ num_batches_tracked) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double s1 = 0.0, s2 = 0.0;
    for (int e = lane; e < N * PT; e += 64) {
        const float* p = part + ((int64_t)e * C + c) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) {
        const double cnt = (double)N * HW, m = s1 / cnt;
        double var = s2 / cnt - m * m;  // (fp64 on sums of bf16-exact squares: no visible cancellation at these counts)
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}
int bn_tile_stats(const float* part, int N, int C, int HW, float eps, float mom, float* mean, float* rstd, float* rmean, float* rvar,
                  long long* nbt, hipStream_t s) {
    COT_LAUNCH(bn_tile_stats_finalize, dim3(C), dim3(64), 0, s, part, N, C, ceil_div(HW, 128), HW, eps, mom, mean, rstd, rmean, rvar, nbt);
    return check_launch("bn_tile_stats_finalize");
}
// ---- statistics from the AGGREGATION's epilogue (round 6): agg_fwd_nchw_k3_lds<ST = 1> writes per output row (n, c, h) the sum and the
// sum of squares of what it stores; one workgroup per channel adds the channel's N*H rows -- per-thread fp64 partials in a fixed
// interleaving, then a fixed-order workgroup sum: deterministic -- and finishes exactly as bn_tile_stats_finalize does.
__global__ __launch_bounds__(256) void bn_rowstats_finalize(const float* __restrict__ rows, int N, int C, int H, int W, float eps,
                                                           float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           long long* __restrict__ num_batches_tracked) {
    __shared__ double red[32];
    const int c = blockIdx.x, t = threadIdx.x;
    if (c == 0 && t == 0 && num_batches_tracked) *num_batches_tracked += 1;
    double s1 = 0.0, s2 = 0.0;
    const int NR = N * H;  // rows of this channel: image n, row h at ((n*C + c)*H + h)*2
    for (int e = t; e < NR; e += 256) {
        const int n = e / H, h = e - n * H;
        const Vec<float, 2> v = ldv<float, 2>(rows + (((int64_t)n * C + c) * H + h) * 2);
        s1 += (double)v.v[0];
        s2 += (double)v.v[1];
    }
    double acc[2] = {s1, s2};
    block_allsum_d<2>(acc, red);
    if (t == 0) {
        const double cnt = (double)N * H * W, m = acc[0] / cnt;
        double var = acc[1] / cnt - m * m;  // (fp64 on sums of bf16-exact squares: no visible cancellation at these counts)
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}
int bn_rowstats(const float* rows, int N, int C, int H, int W, float eps, float mom, float* mean, float* rstd, float* rmean, float* rvar,
                long long* nbt, hipStream_t s) {
    COT_LAUNCH(bn_rowstats_finalize, dim3(C), dim3(256), 0, s, rows, N, C, H, W, eps, mom, mean, rstd, rmean, rvar, nbt);
    return check_launch("bn_rowstats_finalize");
}
// y = act(gamma * (x - mean_c) * rstd_c + beta [+ residual]) from GIVEN batch statistics (bf16; the flat apply kernel)
int bn_apply_forward(const void* x, const void* res, void* y, const float* gamma, const float* beta, const float* mean, const float* rstd,
                     int N, int C, int HW, int act, hipStream_t s) {
    typedef bf16_t T;
    if (HW % 8 != 0) return -2;
    const int64_t nvec = (int64_t)N * C * HW / 8;
    uint8_t* const mk = t_bn_mask;
#define BN_AF(A_) COT_LAUNCH((bn_apply_fwd<T, 8, A_>), dim3(flat_grid(nvec)), dim3(256), 0, s, (const T*)x, (const T*)res, (T*)y, mean, rstd, gamma, beta, C, HW, nvec, mk)
    if (act == ACT_RELU) BN_AF(ACT_RELU);
    else if (act == ACT_SILU) BN_AF(ACT_SILU);
    else BN_AF(ACT_NONE);
#undef BN_AF
    return check_launch("bn_apply_fwd");
}

// ---- per-tensor layouts (round 5): the deep stages' Bottlenecks keep the operands of their 1x1 convolutions
// "channel-major" -- [C][N][HW]: a channel's N planes are one contiguous row, the GEMM / BatchNorm calls see N = 1, HW' = N*HW --
// while the plane kernels between them (grouped 3x3, aggregation) stay NCHW.  The layout changes where a BatchNorm already moves
// every element: these are the channel-resident kernels above with one (image, channel) stride pair PER TENSOR, a second output
// (bn1 feeds the 3x3 in NCHW and the 1x1 convolutions channel-major) and a second upstream gradient (the mirror image).  A lane's
// vector stays inside one plane (V divides HW), so element (n, c, p) has the same (n, p) in every tensor; only the plane's base moves.
// lay bits: 0 = NCHW ((n*C + c)*HW), 1 = channel-major ((c*N + n)*HW).
__device__ __forceinline__ int lay_off(int cm, int n, int c, int N, int C, int HW) { return cm ? (c * N + n) * HW : (n * C + c) * HW; }

template <typename T, int V, int ACT, int RR>
__global__ __launch_bounds__(1024) void bn_chan_fwd_lay(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                       T* __restrict__ y2, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ mean,
                                                       float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                                       long long* __restrict__ nbt, int N, int C, int HW, float eps, float mom,
                                                       const float* __restrict__ ps, int lay) {
    constexpr int R = RR ? RR : ChanRounds<T, V, false>::value;
    __shared__ double red[16];
    const int c = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    const int vpp = HW / V, MV = N * vpp;
    const double M = (double)N * HW;
    if (c == 0 && t == 0 && nbt) *nbt += 1;
    Vec<T, V> xv[R];
    int pn[R], pv[R];  // image and in-plane element offset of round r's vector, pn = -1: past the channel's end
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * NT + t;
        pn[r] = -1;
        pv[r] = 0;
        if (e < MV) {
            const int n = e / vpp;
            pn[r] = n;
            pv[r] = (e - n * vpp) * V;
            xv[r] = ldv<T, V>(x + lay_off(lay & 1, n, c, N, C, HW) + pv[r]);
#pragma unroll
            for (int k = 0; k < V; ++k) s += (float)xv[r].v[k];
        }
    }
    double acc[1] = {(double)s};
    block_allsum_d<1>(acc, red);
    const float mu = (float)(acc[0] / M);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (pn[r] >= 0) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float d = (float)xv[r].v[k] - mu;
                q += d * d;
            }
        }
    acc[0] = (double)q;
    block_allsum_d<1>(acc, red);
    const float var = (float)(acc[0] / M), rs = 1.0f / sqrtf(var + eps);
    if (t == 0) {
        mean[c] = mu;
        rstd[c] = rs;
        if (rmean) {
            const float unbiased = M > 1.0 ? (float)(acc[0] / (M - 1.0)) : var;
            rmean[c] = (1.f - mom) * rmean[c] + mom * mu;
            rvar[c] = (1.f - mom) * rvar[c] + mom * unbiased;
        }
    }
    const float a = gamma[c] * rs, b = beta[c] - mu * a;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (pn[r] >= 0) {
            Vec<T, V> rv, o;
            if (res) rv = ldv<T, V>(res + lay_off(lay & 2, pn[r], c, N, C, HW) + pv[r]);
            const float sc = ps ? ps[pn[r]] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float z = ((float)xv[r].v[k] * a + b) * sc;
                if (res) z += (float)rv.v[k];
                o.v[k] = (T)act_fwd<ACT>(z);
            }
            stv<T, V>(y + lay_off(lay & 4, pn[r], c, N, C, HW) + pv[r], o);
            if (y2) stv<T, V>(y2 + lay_off(lay & 8, pn[r], c, N, C, HW) + pv[r], o);
        }
}

// backward bits: 1 dy, 2 dy2, 4 x, 8 y, 16 dx, 32 dres.  dy2 != NULL: the upstream gradient is dy + dy2 (fp32 sum, one rounding --
// what the data-gradient kernels' `accumulate` does when both contributions share a layout)
template <typename T, int V, int ACT, int RR>
__global__ __launch_bounds__(1024) void bn_chan_bwd_lay(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ x,
                                                       const T* __restrict__ y, T* __restrict__ dx, T* __restrict__ dres,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int HW,
                                                       const float* __restrict__ ps, int lay) {
    constexpr int R = RR ? RR : ChanRounds<T, V, true>::value;
    __shared__ double red[32];
    const int c = blockIdx.x, t = threadIdx.x, NT = blockDim.x;
    const int vpp = HW / V, MV = N * vpp;
    const float inv_m = 1.0f / ((float)N * (float)HW);
    const float m = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    Vec<T, V> xv[R], dv[R];
    int pn[R], pv[R];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * NT + t;
        pn[r] = -1;
        pv[r] = 0;
        if (e < MV) {
            const int n = e / vpp;
            pn[r] = n;
            pv[r] = (e - n * vpp) * V;
            xv[r] = ldv<T, V>(x + lay_off(lay & 4, n, c, N, C, HW) + pv[r]);
            dv[r] = ldv<T, V>(dy + lay_off(lay & 1, n, c, N, C, HW) + pv[r]);
            if (dy2) {
                const Vec<T, V> d2 = ldv<T, V>(dy2 + lay_off(lay & 2, n, c, N, C, HW) + pv[r]);
#pragma unroll
                for (int k = 0; k < V; ++k) dv[r].v[k] = (T)((float)dv[r].v[k] + (float)d2.v[k]);
            }
            if (ACT == ACT_RELU_Y) {  // the saved output's sign decides: fold it into dy now (exact), y is not kept
                const Vec<T, V> yv = ldv<T, V>(y + lay_off(lay & 8, n, c, N, C, HW) + pv[r]);
#pragma unroll
                for (int k = 0; k < V; ++k) dv[r].v[k] = (float)yv.v[k] > 0.f ? dv[r].v[k] : (T)0.f;
            }
            const float sc = ps ? ps[n] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = ((float)xv[r].v[k] - m) * rs;
                const float g = sc * (ACT == ACT_RELU_Y ? (float)dv[r].v[k] : act_bwd<ACT>((float)dv[r].v[k], xh * ga + be));
                sg += g;
                sgx += g * xh;
            }
        }
    }
    double acc[2] = {(double)sg, (double)sgx};
    block_allsum_d<2>(acc, red);
    const float sb = (float)acc[0], sx = (float)acc[1];
    if (t == 0) {
        dbeta[c] = sb;
        dgamma[c] = sx;
    }
    const float k1 = sb * inv_m, k2 = sx * inv_m, gr = ga * rs;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (pn[r] >= 0) {
            Vec<T, V> o, og;
            const float sc = ps ? ps[pn[r]] : 1.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = ((float)xv[r].v[k] - m) * rs;
                const float g = ACT == ACT_RELU_Y ? (float)dv[r].v[k] : act_bwd<ACT>((float)dv[r].v[k], xh * ga + be);
                o.v[k] = (T)(gr * (sc * g - k1 - xh * k2));
                og.v[k] = (T)g;
            }
            stv<T, V>(dx + lay_off(lay & 16, pn[r], c, N, C, HW) + pv[r], o);
            if (dres) stv<T, V>(dres + lay_off(lay & 32, pn[r], c, N, C, HW) + pv[r], og);
        }
}

// host side: bf16 tensors the channel-resident kernels hold (every BatchNorm of the 14 x 14 / 7 x 7 stages at the benchmark batch);
// -2 = not covered (the caller keeps one layout and the ordinary entry points)
int bn_act_lay_covers(int N, int C, int HW) {
    return bn_chan_vec<bf16_t, false>(N, C, HW) > 1 && bn_chan_vec<bf16_t, true>(N, C, HW) > 1 && (int64_t)N * HW > g_bn_small_m;
}
int bn_act_forward_lay(const void* x, const void* res, void* y, void* y2, const float* gamma, const float* beta, float* mean, float* rstd,
                       float* rmean, float* rvar, long long* nbt, int N, int C, int HW, float eps, float mom, int act, const float* ps,
                       int lay, hipStream_t s) {
    typedef bf16_t T;
    const int cv = bn_chan_vec<T, false>(N, C, HW);
    if (cv <= 1 || (int64_t)N * HW <= g_bn_small_m) return -2;
    const dim3 grid(C), block(bn_chan_threads(N, C, HW, cv, bn_chan_rounds<T, false>(cv)));
    const int rr = bn_chan_rr((int64_t)N * HW / cv, (int)block.x, bn_chan_rounds<T, false>(cv));
#define BN_LFR(V_, A_, R_) COT_LAUNCH((bn_chan_fwd_lay<T, V_, A_, R_>), grid, block, 0, s, (const T*)x, (const T*)res, (T*)y, (T*)y2, gamma, beta, mean, rstd, rmean, rvar, nbt, N, C, HW, eps, mom, ps, lay)
#define BN_LF(V_, A_)                      \
    do {                                   \
        if (rr == 2) BN_LFR(V_, A_, 2);    \
        else if (rr == 4) BN_LFR(V_, A_, 4); \
        else BN_LFR(V_, A_, 0);            \
    } while (0)
#define BN_LFV(V_)                                \
    do {                                          \
        if (act == ACT_RELU) BN_LF(V_, ACT_RELU); \
        else if (act == ACT_SILU) BN_LF(V_, ACT_SILU); \
        else BN_LF(V_, ACT_NONE);                 \
    } while (0)
    if (cv == 8) BN_LFV(8);
    else if (cv == 7) BN_LFV(7);
    else BN_LFV(4);
#undef BN_LFV
#undef BN_LF
#undef BN_LFR
    return check_launch("bn_chan_fwd_lay");
}
int bn_act_backward_lay(const void* dy, const void* dy2, const void* x, const void* y, void* dx, void* dres, const float* gamma,
                        const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta, int N, int C, int HW, int act,
                        const float* ps, int lay, hipStream_t s) {
    typedef bf16_t T;
    const int cv = bn_chan_vec<T, true>(N, C, HW);
    if (cv <= 1 || (int64_t)N * HW <= g_bn_small_m) return -2;
    const dim3 grid(C), block(bn_chan_threads(N, C, HW, cv, bn_chan_rounds<T, true>(cv)));
    const int rr = bn_chan_rr((int64_t)N * HW / cv, (int)block.x, bn_chan_rounds<T, true>(cv));
#define BN_LBR(V_, A_, R_) COT_LAUNCH((bn_chan_bwd_lay<T, V_, A_, R_>), grid, block, 0, s, (const T*)dy, (const T*)dy2, (const T*)x, (const T*)y, (T*)dx, (T*)dres, gamma, beta, mean, rstd, dgamma, dbeta, N, C, HW, ps, lay)
#define BN_LB(V_, A_)                      \
    do {                                   \
        if (rr == 2) BN_LBR(V_, A_, 2);    \
        else if (rr == 4 && bn_chan_rounds<T, true>(V_) > 4) BN_LBR(V_, A_, 4); \
        else BN_LBR(V_, A_, 0);            \
    } while (0)
#define BN_LBV(V_)                                         \
    do {                                                   \
        if (act == ACT_RELU && y) BN_LB(V_, ACT_RELU_Y);   \
        else if (act == ACT_RELU) BN_LB(V_, ACT_RELU);     \
        else if (act == ACT_SILU) BN_LB(V_, ACT_SILU);     \
        else BN_LB(V_, ACT_NONE);                          \
    } while (0)
    if (cv == 8) BN_LBV(8);
    else if (cv == 7) BN_LBV(7);
    else BN_LBV(4);
#undef BN_LBV
#undef BN_LB
#undef BN_LBR
    return check_launch("bn_chan_bwd_lay");
}

template int bn_act_forward<float>(const void*, const void*, void*, const float*, const float*, float*, float*, float*,
                                   float*, long long*, float*, int, int, int, float, float, int, const float*, hipStream_t);
template int bn_act_forward<bf16_t>(const void*, const void*, void*, const float*, const float*, float*, float*, float*,
                                    float*, long long*, float*, int, int, int, float, float, int, const float*, hipStream_t);
template int bn_act_backward<float>(const void*, const void*, const void*, void*, void*, const float*, const float*,
                                    const float*, const float*, float*, float*, float*, int, int, int, int, const float*,
                                    hipStream_t);
template int bn_act_backward<bf16_t>(const void*, const void*, const void*, void*, void*, const float*, const float*,
                                     const float*, const float*, float*, float*, float*, int, int, int, int, const float*,
                                     hipStream_t);

}  // namespace cot
