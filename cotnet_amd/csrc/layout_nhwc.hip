// layout_nhwc.hip -- STUDY (the channels-last route): the layout change at the boundary between the NCHW stages and the
// channels-last stages, per image  [C][HW] <-> [HW][C], any C and HW (14 x 14 and 7 x 7 planes are neither multiples of 8 nor of 16
// bytes, so the plane side moves 2-byte elements; the channel side moves 16-byte vectors when C % 8 == 0).  64 (pixels) x 64
// (channels) tiles through LDS, one workgroup per tile.  Exported as cot_study_nchw_to_nhwc / cot_study_nhwc_to_nchw (2-byte types).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "cot_common.h"

namespace cot {

// TO_CL: src [N][C][HW] -> dst [N][HW][C];  otherwise src [N][HW][C] -> dst [N][C][HW]
template <bool TO_CL>
__global__ __launch_bounds__(256) void layout_swap_2b(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int C, int HW, int ctiles) {
    __shared__ uint16_t tile[64][66];  // [pixel][channel], rows padded by two elements (odd dword stride: conflict-free columns)
    const int n = blockIdx.y / ctiles, p0 = blockIdx.x * 64, c0 = (blockIdx.y - n * ctiles) * 64;
    const uint16_t* s = src + (int64_t)n * C * HW;
    uint16_t* d = dst + (int64_t)n * C * HW;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    if (TO_CL) {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {  // rows of the plane side: channel c, 64 consecutive pixels
            const int c = c0 + ty + 4 * i, p = p0 + tx;
            if (c < C && p < HW) tile[tx][ty + 4 * i] = s[(int64_t)c * HW + p];
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {  // rows of the channel side: pixel p, 64 consecutive channels
            const int p = p0 + ty + 4 * i, c = c0 + tx;
            if (p < HW && c < C) d[(int64_t)p * C + c] = tile[ty + 4 * i][tx];
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int p = p0 + ty + 4 * i, c = c0 + tx;
            if (p < HW && c < C) tile[ty + 4 * i][tx] = s[(int64_t)p * C + c];
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty + 4 * i, p = p0 + tx;
            if (c < C && p < HW) d[(int64_t)c * HW + p] = tile[tx][ty + 4 * i];
        }
    }
}

int layout_swap(const void* src, void* dst, int N, int C, int HW, bool to_cl, hipStream_t s) {
    if (!src || !dst || N <= 0 || C <= 0 || HW <= 0) return -1;
    const int ctiles = ceil_div(C, 64);
    if ((int64_t)ctiles * N > 65535) return -2;
    const dim3 grid(ceil_div(HW, 64), ctiles * N);
    if (to_cl) COT_LAUNCH((layout_swap_2b<true>), grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, C, HW, ctiles);
    else COT_LAUNCH((layout_swap_2b<false>), grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, C, HW, ctiles);
    return check_launch("layout_swap");
}

}  // namespace cot

extern "C" int cot_study_nchw_to_nhwc(const void* src, void* dst, int N, int C, int HW, void* stream) {
    return cot::layout_swap(src, dst, N, C, HW, true, (hipStream_t)stream);
}
extern "C" int cot_study_nhwc_to_nchw(const void* src, void* dst, int N, int C, int HW, void* stream) {
    return cot::layout_swap(src, dst, N, C, HW, false, (hipStream_t)stream);
}
