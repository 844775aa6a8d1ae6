// Micro-benchmark (diagnostic, not part of the library): how many bytes per second can ONE CU pull in, by
// (0) LDS-DMA (global_load_lds_dwordx4) and (1) plain global_load_dwordx4 to VGPRs, from HBM-cold and from L2-hot data,
// as a function of the number of workgroups?   Built by scripts/ubench_ingest.py with hipcc; extern "C" launcher.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(lds_wave_base));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// every workgroup reads `iters` x 16 KB: region = base + (wg % nregions) * span, walking through it cyclically
template <int MODE>
__global__ __launch_bounds__(256) void ingest(const char* __restrict__ base, int64_t span, int nregions, int iters,
                                             unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // MODE 0: 8 slots of 16 KB
    const int tid = threadIdx.x, wave = tid >> 6;
    const char* reg = base + (int64_t)(blockIdx.x % nregions) * span;
    uint4 acc = {0, 0, 0, 0};
    int64_t off = ((int64_t)blockIdx.x * 16384 * 7) % span;  // different workgroups start at different places of a shared region
    if (MODE == 2) reg += 2;  // every 16-byte piece 2-byte aligned only
    if (MODE == 0 || MODE == 2) {
        for (int it = 0; it < iters; ++it) {
            char* slot = smem + (it & 7) * 16384;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) glds16(reg + off + (ps * 256 + tid) * 16, slot + (ps * 256 + wave * 64) * 16);
            off += 16384;
            if (off + 16384 + 16 > span) off = 0;
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // 6 slots (24 copies) stay in flight
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc.x = *(const unsigned*)(smem + tid * 4);
    } else {
        for (int it = 0; it < iters; ++it) {
            uint4 v[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) v[ps] = *(const uint4*)(reg + off + (ps * 256 + tid) * 16);
            off += 16384;
            if (off + 16384 > span) off = 0;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) { acc.x ^= v[ps].x; acc.y ^= v[ps].y; acc.z ^= v[ps].z; acc.w ^= v[ps].w; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

// misaligned LDS-DMA probe: copy 64 x 16 bytes from `src + off` (off = 0, 2, 4, 8 bytes) and write the LDS image out
__global__ void glds_misaligned(const char* __restrict__ src, int off, unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char lds[1024];
    glds16(src + off + threadIdx.x * 16, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = ((const unsigned*)lds)[i];
}
extern "C" int ubench_glds_misaligned(const void* src, int off, void* out, void* stream) {
    hipLaunchKernelGGL(glds_misaligned, dim3(1), dim3(64), 0, (hipStream_t)stream, (const char*)src, off, (unsigned*)out);
    return (int)hipGetLastError();
}
extern "C" int ubench_ingest(int mode, const void* base, int64_t span, int nregions, int iters, int nwg, void* sink, void* stream) {
    if (mode == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(ingest<0>, dim3(nwg), dim3(256), 8 * 16384, (hipStream_t)stream, (const char*)base, span, nregions, iters, (unsigned*)sink);
    } else if (mode == 2) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(ingest<2>, dim3(nwg), dim3(256), 8 * 16384, (hipStream_t)stream, (const char*)base, span, nregions, iters, (unsigned*)sink);
    } else {
        hipLaunchKernelGGL(ingest<1>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (const char*)base, span, nregions, iters, (unsigned*)sink);
    }
    return (int)hipGetLastError();
}
