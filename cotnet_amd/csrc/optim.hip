// optim.hip -- fused SGD (momentum / nesterov / weight decay) over FLAT parameter buffers, gfx950.
//
// Replaces the reference's per-tensor optimizer step (optim/optim_factory.py:54-56 -> torch.optim.SGD(nesterov=True),
// one or more elementwise launches per parameter tensor) with one launch per flat bucket:
//     g   = grad * grad_scale + wd * p
//     buf = momentum * buf + g
//     p  -= lr * (nesterov ? g + momentum * buf : buf)
// Low-precision training keeps an fp32 master copy: the kernel reads the bf16 gradient bucket, updates master and
// momentum in fp32 and writes the bf16 working copy the convolutions read -- 2+4+4 B read, 4+4+2 B written per
// parameter, HBM-bound, 16-byte accesses per lane.  Same arithmetic (in fp32) as torch.optim.SGD with dampening 0.
#include "cot_common.h"

namespace cot {

template <typename PT, typename GT, bool HAS_MASTER, int V>
__global__ __launch_bounds__(256) void sgd_flat_kernel(PT* __restrict__ param, float* __restrict__ master,
                                                      float* __restrict__ mom, const GT* __restrict__ grad,
                                                      int64_t n, float lr, float momentum, float wd, float gscale,
                                                      int nesterov) {
    const int64_t nvec = n / V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const Vec<GT, V> gv = ldv<GT, V>(grad + i * V);
        Vec<float, V> mv = ldv<float, V>(mom + i * V);
        Vec<float, V> pv;
        if (HAS_MASTER) {
            pv = ldv<float, V>(master + i * V);
        } else {
            const Vec<PT, V> t = ldv<PT, V>(param + i * V);
#pragma unroll
            for (int k = 0; k < V; ++k) pv.v[k] = (float)t.v[k];
        }
        Vec<PT, V> out;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float g = (float)gv.v[k] * gscale + wd * pv.v[k];
            const float b = momentum * mv.v[k] + g;
            mv.v[k] = b;
            pv.v[k] -= lr * (nesterov ? g + momentum * b : b);
            out.v[k] = (PT)pv.v[k];
        }
        stv<float, V>(mom + i * V, mv);
        if (HAS_MASTER) stv<float, V>(master + i * V, pv);
        stv<PT, V>(param + i * V, out);
    }
    // tail (n % V elements): first lanes of block 0
    const int64_t tail0 = nvec * V;
    if (blockIdx.x == 0 && threadIdx.x < n - tail0) {
        const int64_t i = tail0 + threadIdx.x;
        float p = HAS_MASTER ? master[i] : (float)param[i];
        const float g = (float)grad[i] * gscale + wd * p;
        const float b = momentum * mom[i] + g;
        mom[i] = b;
        p -= lr * (nesterov ? g + momentum * b : b);
        if (HAS_MASTER) master[i] = p;
        param[i] = (PT)p;
    }
}

// ema = decay * ema + (1 - decay) * src over a flat buffer (reference: utils/model_ema.py:45-53, ModelEmaV2._update, one
// elementwise op per state_dict tensor); src = the fp32 master copy (or the fp32 parameters), ema fp32
template <typename ST, int V>
__global__ __launch_bounds__(256) void ema_flat_kernel(float* __restrict__ ema, const ST* __restrict__ src, int64_t n,
                                                      float decay) {
    const int64_t nvec = n / V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        Vec<float, V> e = ldv<float, V>(ema + i * V);
        const Vec<ST, V> sv = ldv<ST, V>(src + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) e.v[k] = decay * e.v[k] + (1.f - decay) * (float)sv.v[k];
        stv<float, V>(ema + i * V, e);
    }
    const int64_t tail0 = nvec * V;
    if (blockIdx.x == 0 && threadIdx.x < n - tail0) {
        const int64_t i = tail0 + threadIdx.x;
        ema[i] = decay * ema[i] + (1.f - decay) * (float)src[i];
    }
}

int ema_flat(void* ema, const void* src, int64_t n, float decay, int src_dtype, hipStream_t s) {
    constexpr int V = 4;
    int64_t blocks = ceil_div64(n / V > 0 ? n / V : 1, 256);
    if (blocks > 2048) blocks = 2048;
    if (src_dtype == COT_F32)
        COT_LAUNCH((ema_flat_kernel<float, V>), dim3((unsigned)blocks), dim3(256), 0, s, (float*)ema, (const float*)src, n,
                   decay);
    else if (src_dtype == COT_BF16)
        COT_LAUNCH((ema_flat_kernel<bf16_t, V>), dim3((unsigned)blocks), dim3(256), 0, s, (float*)ema, (const bf16_t*)src, n,
                   decay);
    else
        return set_error(COT_ERR_UNSUPPORTED, "ema: source dtype %d not supported (float32 / bfloat16)", src_dtype);
    return check_launch("ema_flat_kernel");
}

template <typename PT, typename GT, bool HAS_MASTER>
static int launch_sgd(void* param, void* master, void* mom, const void* grad, int64_t n, float lr, float momentum,
                      float wd, float gscale, int nesterov, hipStream_t s) {
    constexpr int V = 4;  // 16 B of fp32 state per lane
    int64_t blocks = ceil_div64(n / V > 0 ? n / V : 1, 256);
    if (blocks > 2048) blocks = 2048;  // grid-stride: 8 blocks per CU
    COT_LAUNCH((sgd_flat_kernel<PT, GT, HAS_MASTER, V>), dim3((unsigned)blocks), dim3(256), 0, s, (PT*)param,
               (float*)master, (float*)mom, (const GT*)grad, n, lr, momentum, wd, gscale, nesterov);
    return check_launch("sgd_flat_kernel");
}

int sgd_flat(void* param, void* master, void* mom, const void* grad, int64_t n, float lr, float momentum, float wd,
             float gscale, int nesterov, int param_dtype, int grad_dtype, hipStream_t s) {
    if (param_dtype == COT_BF16 && grad_dtype == COT_BF16 && master)
        return launch_sgd<bf16_t, bf16_t, true>(param, master, mom, grad, n, lr, momentum, wd, gscale, nesterov, s);
    if (param_dtype == COT_BF16 && grad_dtype == COT_F32 && master)
        return launch_sgd<bf16_t, float, true>(param, master, mom, grad, n, lr, momentum, wd, gscale, nesterov, s);
    if (param_dtype == COT_F32 && grad_dtype == COT_F32 && !master)
        return launch_sgd<float, float, false>(param, master, mom, grad, n, lr, momentum, wd, gscale, nesterov, s);
    if (param_dtype == COT_F32 && grad_dtype == COT_BF16 && !master)
        return launch_sgd<float, bf16_t, false>(param, master, mom, grad, n, lr, momentum, wd, gscale, nesterov, s);
    return set_error(COT_ERR_UNSUPPORTED, "sgd: param dtype %d / grad dtype %d / master %s not supported", param_dtype,
                     grad_dtype, master ? "given" : "NULL");
}

}  // namespace cot
