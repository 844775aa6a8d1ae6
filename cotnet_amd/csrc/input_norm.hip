// input_norm.hip -- the device half of the reference's PrefetchLoader (datasets/loader.py:54-102): uint8 NCHW images
// -> (float(x) - 255*mean[c]) / (255*std[c]) in the network's input dtype, ONE pass (the reference runs three
// elementwise kernels on the prefetch stream: .float() / .half(), .sub_(mean), .div_(std)).
//   1 B read + sizeof(T) B written per element, HBM-bound: a lane takes 16 consecutive pixels (one 16-byte load) and
//   writes them with 16-byte stores.
// Arithmetic = the reference's, operation by operation: fp32: IEEE subtract, then IEEE divide (no reciprocal, no FMA
// contraction) -> bit-identical to torch; fp16 (reference `fp16=True`): every intermediate rounded to half as torch's
// half kernels do (fp32 holds a half product / quotient exactly enough that the double rounding is innocuous);
// bf16 (extension for the bf16 model): computed in fp32, rounded once.
#include "cot_common.h"

namespace cot {

template <typename T> __device__ __forceinline__ T norm_one(uint8_t u, float m, float s) {
#pragma clang fp contract(off)
    const float d = (float)u - m;
    return (T)(d / s);
}
template <> __device__ __forceinline__ f16_t norm_one<f16_t>(uint8_t u, float m, float s) {
#pragma clang fp contract(off)
    const f16_t d = (f16_t)((float)u - m);  // m, s are already half-representable (the host rounds them as the reference does)
    return (f16_t)((float)d / s);
}

template <typename T>
__global__ __launch_bounds__(256) void input_normalize_kernel(const uint8_t* __restrict__ x, T* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             int64_t planes, int C, int HW) {
    if (HW % 16 == 0) {
        const int vpp = HW / 16;
        const int64_t nvec = planes * vpp;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)((i / vpp) % C);
            const float m = mean[c], s = stdv[c];
            const Vec<uint8_t, 16> v = ldv<uint8_t, 16>(x + i * 16);
            constexpr int OV = 16 / sizeof(T);  // elements per 16-byte store
#pragma unroll
            for (int q = 0; q < 16 / OV; ++q) {
                Vec<T, OV> o;
#pragma unroll
                for (int k = 0; k < OV; ++k) o.v[k] = norm_one<T>(v.v[q * OV + k], m, s);
                stv<T, OV>(y + i * 16 + q * OV, o);
            }
        }
    } else {  // odd image sizes: element-wise
        const int64_t n = planes * HW;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int c = (int)((i / HW) % C);
            y[i] = norm_one<T>(x[i], mean[c], stdv[c]);
        }
    }
}

int input_normalize(const void* x, void* y, const float* mean, const float* stdv, int64_t planes, int C, int HW, int dtype,
                    hipStream_t s) {
    const int64_t work = HW % 16 == 0 ? planes * (HW / 16) : planes * (int64_t)HW;
    int64_t blocks = ceil_div64(work, 256);
    if (blocks > 4096) blocks = 4096;  // grid-stride, 16 blocks per CU
    const dim3 grid((unsigned)blocks), block(256);
    const uint8_t* xs = (const uint8_t*)x;
    switch (dtype) {
        case COT_F32: COT_LAUNCH((input_normalize_kernel<float>), grid, block, 0, s, xs, (float*)y, mean, stdv, planes, C, HW); break;
        case COT_BF16: COT_LAUNCH((input_normalize_kernel<bf16_t>), grid, block, 0, s, xs, (bf16_t*)y, mean, stdv, planes, C, HW); break;
        case COT_F16: COT_LAUNCH((input_normalize_kernel<f16_t>), grid, block, 0, s, xs, (f16_t*)y, mean, stdv, planes, C, HW); break;
        default: return set_error(COT_ERR_UNSUPPORTED, "cot_input_normalize: output dtype %d (float32 / bfloat16 / float16)", dtype);
    }
    return check_launch("input_normalize_kernel");
}

}  // namespace cot
