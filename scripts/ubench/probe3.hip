// Round-3 hardware probes (diagnostic, not part of the library):
//  (1) do ds_read_b64 / ds_read_b128 accept addresses that are only 2-byte aligned?  (7x7 planes: rows are 98 bytes)
//  (2) does the SGPR-base + VGPR-offset form of global_load_lds_dwordx4 work as the inline-asm statement the library wants
//      to use (one scalar base per K step, per-thread offsets constant), with 2-byte-aligned sources?
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void lds_misaligned(const int* __restrict__ lane_byte, uint32_t* __restrict__ out64, uint32_t* __restrict__ out128) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (unsigned)lane_byte[threadIdx.x];
    uint32_t r0, r1, q0, q1, q2, q3;
    uint64_t v64;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v64) : "v"(a) : "memory");
    r0 = (uint32_t)v64; r1 = (uint32_t)(v64 >> 32);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 v128;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v128) : "v"(a) : "memory");
    q0 = v128[0]; q1 = v128[1]; q2 = v128[2]; q3 = v128[3];
    out64[threadIdx.x * 2 + 0] = r0; out64[threadIdx.x * 2 + 1] = r1;
    out128[threadIdx.x * 4 + 0] = q0; out128[threadIdx.x * 4 + 1] = q1; out128[threadIdx.x * 4 + 2] = q2; out128[threadIdx.x * 4 + 3] = q3;
}

// out[lane*8 + e] = what lane `lane`'s 16-byte LDS slot holds after the copy
__global__ void glds_saddr(const uint16_t* __restrict__ src, const int* __restrict__ lane_byte_off, int base_elems,
                           uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdead;
    __syncthreads();
    const uint16_t* sbase = src + base_elems;  // wave-uniform
    const unsigned voff = (unsigned)lane_byte_off[threadIdx.x];
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds) + 256u;
    unsigned keep;
    const uint64_t sb = (uint64_t)(uintptr_t)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const uint64_t sbu = ((uint64_t)hi << 32) | lo;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbu), "s"(dst)
                 : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = 0; e < 8; ++e) out[threadIdx.x * 8 + e] = lds[128 + threadIdx.x * 8 + e];
}

extern "C" int probe_lds_misaligned(const int* lane_byte, void* out64, void* out128, void* stream) {
    hipLaunchKernelGGL(lds_misaligned, dim3(1), dim3(64), 0, (hipStream_t)stream, lane_byte, (uint32_t*)out64, (uint32_t*)out128);
    return (int)hipGetLastError();
}
extern "C" int probe_glds_saddr(const void* src, const int* lane_byte_off, int base_elems, void* out, void* stream) {
    hipLaunchKernelGGL(glds_saddr, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)src, lane_byte_off, base_elems,
                       (uint16_t*)out);
    return (int)hipGetLastError();
}
