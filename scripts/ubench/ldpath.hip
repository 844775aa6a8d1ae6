// Round-3 hardware probe (diagnostic, not part of the library): how many bytes per clock does ONE CU get from L2 when every CU
// streams at once -- through LDS-DMA (global_load_lds_dwordx4), through plain 16-byte loads to registers followed by
// ds_write_b128, and through plain loads alone -- for the two piece shapes the 1x1 kernels use (16 rows x 64 B, 8 rows x 128 B,
// rows 392 B apart as in a 14 x 14 plane) and for flat 1 KB pieces.  One workgroup of 8 waves per CU, `depth` pieces in flight
// per wave.  The source region is 2 MB shared by every workgroup (L2 hits after the first touch) or private per workgroup
// (streamed from HBM).      hipcc --offload-arch=gfx950 -O3 -o ldpath scripts/ubench/ldpath.hip && ./ldpath
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// per-lane byte offset of a piece's element for shape s: 0 flat 1 KB, 1 = 16 rows x 64 B (stride rs), 2 = 8 rows x 128 B
__device__ __forceinline__ unsigned lane_off(int shape, int lane, int rs) {
    if (shape == 1) return (unsigned)((lane >> 2) * rs + (lane & 3) * 16);
    if (shape == 2) return (unsigned)((lane >> 3) * rs + (lane & 7) * 16);
    return (unsigned)(lane * 16);
}

template <int MODE, int DEPTH>  // MODE 0 LDS-DMA, 1 load + ds_write, 2 load only
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, size_t region, size_t wg_stride, int iters, int shape, int rs,
                                              unsigned step, uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    const unsigned voff = lane_off(shape, lane, rs);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem) + (unsigned)wave * (DEPTH * 1024);
    unsigned pos = (unsigned)wave * step;  // byte position of this wave's next piece inside the region
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH];
    if (MODE == 3 && wave >= 4) {  // the other half of the workgroup keeps the LDS busy with 16-byte reads for as long as the copies run
        volatile int* stop = reinterpret_cast<volatile int*>(smem + 159 * 1024);
        u32x4 a = {0, 0, 0, 0};
        const u32x4* base4 = reinterpret_cast<const u32x4*>(smem + 64 * 1024);
        for (int it = 0; it < iters * DEPTH; ++it) {
#pragma unroll
            for (int r = 0; r < 6; ++r) a ^= base4[(lane + 64 * ((it + r) & 15)) & 1023];
        }
        if (a[0] == 0x12345678u && a[1] == 77u) sink[threadIdx.x] = a[2] ^ a[3];
        (void)stop;
        __syncthreads();
        return;
    }
    if (MODE == 0 || MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char* sb = base + pos;
                pos += 8u * step;
                if (pos >= region) pos -= (unsigned)region;
                const unsigned dst = lds0 + d * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(dst) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char* sb = base + pos;
                pos += 8u * step;
                if (pos >= region) pos -= (unsigned)region;
                r[d] = *reinterpret_cast<const u32x4*>(sb + voff);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (MODE == 1) *reinterpret_cast<u32x4*>(smem + wave * (DEPTH * 1024) + d * 1024 + lane * 16) = r[d];
                else acc ^= r[d];
            }
        }
    }
    __syncthreads();
    if (MODE != 2) acc = *reinterpret_cast<u32x4*>(smem + (threadIdx.x * 16) % (4 * DEPTH * 1024));
    if (acc[0] == 0x12345678u && acc[1] == 77u) sink[threadIdx.x] = acc[2] ^ acc[3];
}

static int g_blocks = 256;  // workgroups (= CUs used): argv[1]

template <int MODE, int DEPTH>
static void run(const char* name, const char* src, size_t region, size_t wg_stride, int shape, int rs, unsigned step, uint32_t* sink,
                double clock_ghz) {
    const int iters = 400, blocks = g_blocks;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = MODE == 3 ? (size_t)160 * 1024 : (size_t)8 * DEPTH * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int w = 0; w < 2; ++w) stream<MODE, DEPTH><<<blocks, 512, lds>>>(src, region, wg_stride, iters, shape, rs, step, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) stream<MODE, DEPTH><<<blocks, 512, lds>>>(src, region, wg_stride, iters, shape, rs, step, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)blocks * (MODE == 3 ? 4 : 8) * iters * DEPTH * 1024;
    printf("%-44s depth %2d  %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU  %5.1f B/clk/CU @%.1f GHz\n", name, DEPTH, us, bytes / us * 1e-6,
           bytes / us * 1e-3 / blocks, bytes / us * 1e-3 / blocks / clock_ghz, clock_ghz);
}

int main(int argc, char** argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    printf("workgroups: %d\n", g_blocks);
    const size_t total = (size_t)1 << 30;
    char* src;
    uint32_t* sink;
    CK(hipMalloc(&src, total + (1 << 20)));
    CK(hipMemset(src, 1, total + (1 << 20)));
    CK(hipMalloc(&sink, 4096));
    const double ghz = 2.0;
    const size_t shared_region = 2 << 20, private_region = 4 << 20;  // 256 x 4 MB = 1 GB: streamed from HBM
    // step between consecutive pieces of a wave: flat 1 KB; row shapes advance 64 / 128 B along the rows (rows stay 16 / 8 x rs apart)
    struct { const char* nm; int shape, rs; unsigned step; } pat[] = {
        {"flat 1 KB pieces", 0, 0, 1024}, {"16 rows x 64 B, rows 392 B apart", 1, 392, 6272}, {"8 rows x 128 B, rows 392 B apart", 2, 392, 3136},
        {"16 rows x 64 B, rows 6272 B apart (aligned)", 1, 6272, 64}, {"8 rows x 128 B, rows 6272 B apart (aligned)", 2, 6272, 128},
        // weight tiles read in place: row = output channel, K * 2 bytes apart (K = 256 .. 2048: powers of two)
        {"16 rows x 64 B, rows 512 B apart", 1, 512, 64}, {"16 rows x 64 B, rows 1024 B apart", 1, 1024, 64},
        {"16 rows x 64 B, rows 2048 B apart", 1, 2048, 64}, {"16 rows x 64 B, rows 4096 B apart", 1, 4096, 64},
        {"8 rows x 128 B, rows 2048 B apart", 2, 2048, 128}, {"8 rows x 128 B, rows 4096 B apart", 2, 4096, 128},
        {"16 rows x 64 B, rows 2112 B apart (2048 + 64)", 1, 2112, 64}};
    for (int loc = 0; loc < 2; ++loc) {
        const size_t region = loc ? private_region : shared_region, stride = loc ? private_region : 0;
        printf("== source %s\n", loc ? "private 4 MB per workgroup (HBM stream)" : "2 MB shared by all workgroups (L2 hits)");
        for (auto& p : pat) {
            char nm[128];
            snprintf(nm, sizeof nm, "LDS-DMA      | %s", p.nm);
            run<0, 4>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            run<0, 8>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            run<0, 16>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            snprintf(nm, sizeof nm, "LDS-DMA (4 waves) + 4 waves of ds_read_b128 | %s", p.nm);
            run<3, 8>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            run<3, 16>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            snprintf(nm, sizeof nm, "load only    | %s", p.nm);
            run<2, 8>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
            run<2, 16>(nm, src, region, stride, p.shape, p.rs, p.step, sink, ghz);
        }
    }
    return 0;
}
