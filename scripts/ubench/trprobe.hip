// Probe: what does ds_read_b64_tr_b16 return?  LDS holds element value = its element index (uint16); every lane passes the
// address of 4 consecutive elements; the 4 values each lane receives are written out.  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void trprobe(const int* __restrict__ lane_elem, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int e = lane_elem[threadIdx.x];  // element index this lane points at (multiple of 4)
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
    for (int k = 0; k < 4; ++k) out[threadIdx.x * 4 + k] = (uint16_t)v[k];
}
extern "C" int trprobe_run(const int* lane_elem, void* out, void* stream) {
    hipLaunchKernelGGL(trprobe, dim3(1), dim3(64), 0, (hipStream_t)stream, lane_elem, (uint16_t*)out);
    return (int)hipGetLastError();
}
