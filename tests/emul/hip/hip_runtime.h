// tests/emul/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the product's .hip sources be compiled as plain host C++ (amdclang++ -x c++) so that the kernels'
// index arithmetic can be exercised against the oracle in the GPU-less build container before spending
// MI355X time.  A launch runs on 64 host threads = the 64 lanes of a wavefront; the threads walk the grid's
// waves in the same order, so cross-lane primitives (DPP wave shifts, __shfl_up/down) can be emulated with a
// shared slot array + two barriers.  Never shipped, never loaded by cotnet_amd/.
#pragma once
#include <pthread.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 blockIdx, threadIdx;
inline dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }

namespace emul {
// One host thread per lane of a WORKGROUP (<= 1024).  All threads walk the grid's blocks in the same order, so
// __syncthreads() is a barrier over the block's threads and the wave-level exchanges are barriers over 64 of them.
struct WaveCtx {
    pthread_barrier_t bar;
    uint64_t slot[64];
    uint64_t wide[64][4];  // MFMA operands: [lane][0..1] = A fragment (8 bf16), [lane][2..3] = B fragment
};
struct BlockCtx {
    pthread_barrier_t bar;
    WaveCtx wave[16];
};
inline BlockCtx* g_blk = nullptr;
inline thread_local int t_lane = 0, t_wave = 0;
alignas(16) inline char g_lds[160 * 1024];

template <typename V> inline V exchange(V v, int delta, V oob) {  // returns lane (l + delta)'s v, oob outside 0..63
    WaveCtx& c = g_blk->wave[t_wave];
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(V));
    c.slot[t_lane] = bits;
    pthread_barrier_wait(&c.bar);
    const int src = t_lane + delta;
    V r = oob;
    if (src >= 0 && src < 64) std::memcpy(&r, &c.slot[src], sizeof(V));
    pthread_barrier_wait(&c.bar);
    return r;
}

inline float bf16_bits_to_float(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// v_mfma_f32_16x16x32_bf16, D = A*B + C over the 64 lanes of the calling wave.  Operand maps (gfx950):
//   A: lane l holds A[i = l&15][k = 8*(l>>4) .. +7];  B: lane l holds B[k = 8*(l>>4) .. +7][j = l&15];
//   C/D: lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3.   Every lane of the wave must make the call.
template <typename AB, typename C> inline C mfma_16x16x32_bf16(const AB& a, const AB& b, const C& c) {
    static_assert(sizeof(AB) == 16, "8 x bf16 operand fragments");
    WaveCtx& w = g_blk->wave[t_wave];
    std::memcpy(&w.wide[t_lane][0], &a, 16);
    std::memcpy(&w.wide[t_lane][2], &b, 16);
    pthread_barrier_wait(&w.bar);
    C d = c;
    const int col = t_lane & 15, rg = t_lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        float sum = 0.f;
        for (int k = 0; k < 32; ++k) {
            uint16_t av, bv;
            std::memcpy(&av, (const char*)&w.wide[row + 16 * (k >> 3)][0] + 2 * (k & 7), 2);
            std::memcpy(&bv, (const char*)&w.wide[col + 16 * (k >> 3)][2] + 2 * (k & 7), 2);
            sum += bf16_bits_to_float(av) * bf16_bits_to_float(bv);
        }
        d[r] = c[r] + sum;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    gridDim = grid;
    blockDim = block;
    const unsigned nthreads = (block.x + 63) / 64 * 64, waves = nthreads / 64;
    BlockCtx ctx;
    pthread_barrier_init(&ctx.bar, nullptr, nthreads);
    for (unsigned w = 0; w < waves; ++w) pthread_barrier_init(&ctx.wave[w].bar, nullptr, 64);
    g_blk = &ctx;
    std::vector<std::thread> threads;
    for (unsigned t = 0; t < nthreads; ++t) {
        threads.emplace_back([=, &body]() {
            t_lane = t & 63;
            t_wave = t >> 6;
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned b = 0; b < grid.x; ++b) {
                    blockIdx = dim3(b, by, 0);
                    threadIdx = dim3(t, 0, 0);
                    if (t < block.x) body();
                    pthread_barrier_wait(&g_blk->bar);  // block boundary: LDS is reused by the next block
                }
        });
    }
    for (auto& th : threads) th.join();
    pthread_barrier_destroy(&ctx.bar);
    for (unsigned w = 0; w < waves; ++w) pthread_barrier_destroy(&ctx.wave[w].bar);
    g_blk = nullptr;
}
}  // namespace emul

inline void __syncthreads() { pthread_barrier_wait(&emul::g_blk->bar); }
// dynamic LDS: `extern __shared__ ... char cot_smem[]` in a kernel refers to this array
#define __shared__
namespace cot { alignas(16) inline char cot_smem[160 * 1024]; }  // the kernels live in namespace cot
// async 16-byte global->LDS copy: destination = wave-uniform base + lane*16
#define COT_ASYNC_COPY16(gptr, lds_wave_base) \
    std::memcpy((char*)(lds_wave_base) + emul::t_lane * 16, (const void*)(gptr), 16)

#define COT_MFMA_16X16X32_BF16(a, b, c) emul::mfma_16x16x32_bf16((a), (b), (c))
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only ever applied to wave-uniform values
inline float __expf(float x) { return std::exp(x); }
using std::min;
using std::max;
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
// only the two DPP controls the kernels use: 0x138 = wave_shr:1 (lane l <- l-1), 0x130 = wave_shl:1 (l <- l+1)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) {
    return emul::exchange<int>(src, ctrl == 0x138 ? -1 : +1, old);
}
template <typename V> inline V __shfl_up(V v, int d) { return emul::exchange<V>(v, -d, v); }
template <typename V> inline V __shfl_down(V v, int d) { return emul::exchange<V>(v, +d, v); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emul::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0, e1, flags, ...) \
    emul::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// host stand-ins for the few runtime calls the library makes outside launches
inline hipError_t hipMalloc(void** p, size_t n) { *p = ::operator new(n); return 0; }
inline hipError_t hipFree(void* p) { ::operator delete(p); return 0; }
#define hipMemcpyDeviceToHost 2
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
