// tests/emul/hip/hip_runtime.h -- TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets the product's .hip sources be compiled as plain host C++ (amdclang++ -x c++) so that the kernels'
// index arithmetic can be exercised against the oracle in the GPU-less build container before spending
// MI355X time.  Every lane of a workgroup runs as a user-space coroutine (see namespace emul), so cross-lane
// primitives (DPP wave shifts, __shfl_up/down, MFMA, __syncthreads) keep their lock-step meaning.
// Never shipped, never loaded by cotnet_amd/.
#pragma once
#include <atomic>
#include <algorithm>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
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
// Execution model: every lane of a workgroup is a COROUTINE (own stack, switched in user space by a 10-instruction
// x86-64 routine -- no syscalls); the lanes of one workgroup are scheduled round-robin by one OS thread, and a pool of OS
// threads works through the grid's blocks.  A lane runs until it reaches a cross-lane operation (wave exchange, MFMA,
// __syncthreads), parks there, and continues once all live lanes of its wave / block have arrived -- the same
// lock-step guarantee the hardware gives, at ~100 ns per switch instead of a futex barrier per operation.
extern "C" void emul_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .weak emul_switch
    .type emul_switch,@function
emul_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emul_switch, .-emul_switch
)");

constexpr int kStack = 64 * 1024;
struct Sched {
    void* main_sp = nullptr;
    std::vector<void*> lane_sp;
    std::vector<char> stacks;
    std::vector<char> done;
    int nthreads = 0, cur = 0, live_block = 0;
    int live_wave[16] = {0};
    int wave_arrived[16] = {0}, block_arrived = 0;
    uint64_t wave_gen[16] = {0}, block_gen = 0;
    uint64_t slot[16][64];
    uint64_t wide[16][64][4];  // MFMA operands: [lane][0..1] = A fragment (8 bf16), [lane][2..3] = B fragment
    const std::function<void()>* body = nullptr;
    // LDS-DMA copies issued and not yet waited for, per lane (g_dma_mode 1: a copy lands only when a counted wait retires it)
    struct Copy { void* dst; const void* src; };
    std::vector<std::deque<Copy>> pend;
};
inline int g_sched_order = 0;  // see run_block
// 0: an LDS-DMA copy lands the moment it is issued (the EARLIEST legal time: exposes write-after-read hazards -- a stage
//    re-filled while somebody may still read it);  1: it lands when the issuing lane's `s_waitcnt vmcnt(N)` retires it (the
//    LATEST legal time: exposes read-after-write hazards -- a fragment read that the vmcnt arithmetic does not cover reads
//    stale bytes).  Results must be identical under both.
inline int g_dma_mode = 0;
inline thread_local Sched* t_sched = nullptr;
inline thread_local int t_lane = 0, t_wave = 0;

inline void yield_lane() {
    Sched& S = *t_sched;
    emul_switch(&S.lane_sp[S.cur], S.main_sp);
}
inline void wave_barrier() {
    Sched& S = *t_sched;
    const int w = t_wave;
    const uint64_t gen = S.wave_gen[w];
    if (++S.wave_arrived[w] >= S.live_wave[w]) {
        S.wave_arrived[w] = 0;
        ++S.wave_gen[w];
    } else {
        while (S.wave_gen[w] == gen) yield_lane();
    }
}
inline void block_barrier() {
    Sched& S = *t_sched;
    const uint64_t gen = S.block_gen;
    if (++S.block_arrived >= S.live_block) {
        S.block_arrived = 0;
        ++S.block_gen;
    } else {
        while (S.block_gen == gen) yield_lane();
    }
}
inline void lane_entry() {  // first frame of every coroutine
    Sched& S = *t_sched;
    (*S.body)();
    const int me = S.cur, w = me >> 6;
    for (auto& c : S.pend[me]) std::memcpy(c.dst, c.src, 16);  // (s_endpgm: outstanding copies still land)
    S.pend[me].clear();
    S.done[me] = 1;
    // a finished lane no longer takes part in barriers; release anybody who was only waiting for it
    if (--S.live_wave[w] > 0 && S.wave_arrived[w] >= S.live_wave[w]) { S.wave_arrived[w] = 0; ++S.wave_gen[w]; }
    if (--S.live_block > 0 && S.block_arrived >= S.live_block) { S.block_arrived = 0; ++S.block_gen; }
    for (;;) yield_lane();  // never resumed again
}

template <typename V> inline V exchange(V v, int delta, V oob) {  // returns lane (l + delta)'s v, oob outside 0..63
    Sched& S = *t_sched;
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(V));
    S.slot[t_wave][t_lane] = bits;
    wave_barrier();
    const int src = t_lane + delta;
    V r = oob;
    if (src >= 0 && src < 64) std::memcpy(&r, &S.slot[t_wave][src], sizeof(V));
    wave_barrier();
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
    Sched& S = *t_sched;
    auto& w = S.wide[t_wave];
    std::memcpy(&w[t_lane][0], &a, 16);
    std::memcpy(&w[t_lane][2], &b, 16);
    wave_barrier();
    C d = c;
    const int col = t_lane & 15, rg = t_lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        float sum = 0.f;
        for (int k = 0; k < 32; ++k) {
            uint16_t av, bv;
            std::memcpy(&av, (const char*)&w[row + 16 * (k >> 3)][0] + 2 * (k & 7), 2);
            std::memcpy(&bv, (const char*)&w[col + 16 * (k >> 3)][2] + 2 * (k & 7), 2);
            sum += bf16_bits_to_float(av) * bf16_bits_to_float(bv);
        }
        d[r] = c[r] + sum;
    }
    wave_barrier();
    return d;
}

// v_mfma_f32_16x16x4_f32 (A: lane l holds A[i = l&15][k = l>>4], B: B[k = l>>4][j = l&15]) and
// v_mfma_f32_16x16x16_bf16 (A[i = l&15][k = 4*(l>>4) .. +3], B[k = 4*(l>>4) .. +3][j = l&15]); C/D as above.
template <typename C> inline C mfma_16x16x4_f32(float a, float b, const C& c) {
    Sched& S = *t_sched;
    auto& w = S.wide[t_wave];
    std::memcpy(&w[t_lane][0], &a, 4);
    std::memcpy(&w[t_lane][2], &b, 4);
    wave_barrier();
    C d = c;
    const int col = t_lane & 15, rg = t_lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        float sum = 0.f;
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, &w[row + 16 * k][0], 4);
            std::memcpy(&bv, &w[col + 16 * k][2], 4);
            sum += av * bv;
        }
        d[r] = c[r] + sum;
    }
    wave_barrier();
    return d;
}
template <typename AB, typename C> inline C mfma_16x16x16_bf16(const AB& a, const AB& b, const C& c) {
    static_assert(sizeof(AB) == 8, "4 x bf16 operand fragments");
    Sched& S = *t_sched;
    auto& w = S.wide[t_wave];
    std::memcpy(&w[t_lane][0], &a, 8);
    std::memcpy(&w[t_lane][2], &b, 8);
    wave_barrier();
    C d = c;
    const int col = t_lane & 15, rg = t_lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        float sum = 0.f;
        for (int k = 0; k < 16; ++k) {
            uint16_t av, bv;
            std::memcpy(&av, (const char*)&w[row + 16 * (k >> 2)][0] + 2 * (k & 3), 2);
            std::memcpy(&bv, (const char*)&w[col + 16 * (k >> 2)][2] + 2 * (k & 3), 2);
            sum += bf16_bits_to_float(av) * bf16_bits_to_float(bv);
        }
        d[r] = c[r] + sum;
    }
    wave_barrier();
    return d;
}

inline void run_block(Sched& S, unsigned bx, unsigned by, unsigned block_threads) {
    const int n = (int)block_threads;
    S.nthreads = n;
    S.lane_sp.assign(n, nullptr);
    S.done.assign(n, 0);
    S.pend.assign(n, {});
    if (S.stacks.size() < (size_t)n * kStack) S.stacks.resize((size_t)n * kStack);
    S.live_block = n;
    S.block_arrived = 0;
    for (int w = 0; w < 16; ++w) {
        S.live_wave[w] = std::max(0, std::min(64, n - 64 * w));
        S.wave_arrived[w] = 0;
    }
    for (int t = 0; t < n; ++t) {  // initial frame: six callee-saved registers + return address = lane_entry
        uintptr_t top = ((uintptr_t)(S.stacks.data() + (size_t)(t + 1) * kStack)) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;               // keeps (rsp + 8) % 16 == 0 at lane_entry's first instruction
        *--sp = (void*)&lane_entry;    // `ret` target
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        S.lane_sp[t] = sp;
    }
    blockIdx = dim3(bx, by, 0);
    int remaining = n;
    unsigned pass = 0;
    while (remaining > 0) {
        remaining = 0;
        ++pass;
        for (int u = 0; u < n; ++u) {
            // lane scheduling order: 0 = ascending, 1 = descending, 2 = a different rotation + stride every pass.  Results
            // must not depend on it -- a kernel that needs a barrier it does not have shows up under 1 / 2.
            int t = u;
            if (g_sched_order == 1) t = n - 1 - u;
            else if (g_sched_order == 2) t = (int)(((unsigned)u * 37u + pass * 101u) % (unsigned)n);  // 37 coprime to 64*k
            if (S.done[t]) continue;
            S.cur = t;
            t_lane = t & 63;
            t_wave = t >> 6;
            threadIdx = dim3(t, 0, 0);
            emul_switch(&S.main_sp, S.lane_sp[t]);
            if (!S.done[t]) ++remaining;
        }
    }
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    gridDim = grid;
    blockDim = block;
    const unsigned nblocks = grid.x * grid.y;
    const unsigned nworkers = std::max(1u, std::min(std::min(nblocks, std::thread::hardware_concurrency()), 16u));
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        Sched S;
        S.body = &body;
        t_sched = &S;
        for (;;) {
            const unsigned b = next.fetch_add(1);
            if (b >= nblocks) break;
            run_block(S, b % grid.x, b / grid.x, block.x);
        }
        t_sched = nullptr;
    };
    if (nworkers == 1) {
        worker();
    } else {
        std::vector<std::thread> threads;
        for (unsigned i = 0; i < nworkers; ++i) threads.emplace_back(worker);
        for (auto& th : threads) th.join();
    }
}
}  // namespace emul

extern "C" __attribute__((weak)) void emul_set_order(int order) { emul::g_sched_order = order; }
extern "C" __attribute__((weak)) void emul_set_dma_mode(int mode) { emul::g_dma_mode = mode; }
inline void __syncthreads() { emul::block_barrier(); }
// dynamic LDS: `extern __shared__ ... char cot_smem[]` in a kernel refers to this array
#define __shared__ thread_local  // one workgroup at a time per OS thread
namespace cot { alignas(16) inline thread_local char cot_smem[160 * 1024]; }  // the kernels live in namespace cot
// async 16-byte global->LDS copy: destination = wave-uniform base + lane*16
#define COT_ASYNC_COPY16(gptr, lds_wave_base) \
    std::memcpy((char*)(lds_wave_base) + emul::t_lane * 16, (const void*)(gptr), 16)

// the hand-counted LDS-DMA pipeline primitives of conv_lds*.hip: copies land at once (g_dma_mode 0) or when a counted wait of
// the issuing lane retires them (g_dma_mode 1, which checks the vmcnt arithmetic); the barrier is the block barrier
namespace emul {
inline void dma16(void* dst, const void* src) {
    if (g_dma_mode == 0) { std::memcpy(dst, src, 16); return; }
    Sched& S = *t_sched;
    S.pend[S.cur].push_back({dst, src});
}
inline void wait_vm(int n) {  // s_waitcnt vmcnt(n): the oldest copies complete until n are left
    if (g_dma_mode == 0) return;
    Sched& S = *t_sched;
    auto& q = S.pend[S.cur];
    while ((int)q.size() > n) {
        std::memcpy(q.front().dst, q.front().src, 16);
        q.pop_front();
    }
}
}  // namespace emul
#define COT_GLDS16(gptr, lds_wave_base) emul::dma16((char*)(lds_wave_base) + emul::t_lane * 16, (const void*)(gptr))
// scalar-base form: LDS "addresses" are byte offsets into cot_smem
#define COT_LDS_ADDR(p) ((unsigned)((const char*)(p) - (const char*)cot::cot_smem))
#define COT_GLDS16S(sbase, voff, lds_addr) \
    emul::dma16((char*)cot::cot_smem + (lds_addr) + emul::t_lane * 16, (const char*)(sbase) + (voff))
// ds_read_b64_tr_b16 (mapping measured on the MI355X, scripts/ubench_trprobe.py): within each group of 16 lanes, lane Li
// receives element (Li & 3) of the four lanes 4e + (Li >> 2), e = 0..3
typedef __attribute__((ext_vector_type(4))) short emul_s16x4;
inline emul_s16x4 emul_read_tr16(const void* p) {
    emul::Sched& S = *emul::t_sched;
    uint64_t mine;
    std::memcpy(&mine, p, 8);
    S.slot[emul::t_wave][emul::t_lane] = mine;
    emul::wave_barrier();
    const int base = emul::t_lane & ~15, Li = emul::t_lane & 15;
    emul_s16x4 r;
    for (int e = 0; e < 4; ++e) {
        const uint64_t v = S.slot[emul::t_wave][base + 4 * e + (Li >> 2)];
        r[e] = (short)((v >> (16 * (Li & 3))) & 0xffff);
    }
    emul::wave_barrier();
    return r;
}
#define COT_LDS_READ_TR16(p) emul_read_tr16((p))
#define COT_WAIT_VM(N) emul::wait_vm((N))
#define COT_LDS_BARRIER() emul::block_barrier()
#define COT_SCHED_FENCE() ((void)0)
#define COT_STAMP(p, i) ((void)0)
inline void __builtin_amdgcn_s_waitcnt(int) {}

#define COT_MFMA_16X16X32_BF16(a, b, c) emul::mfma_16x16x32_bf16((a), (b), (c))
#define COT_MFMA_16X16X4_F32(a, b, c) emul::mfma_16x16x4_f32((a), (b), (c))
#define COT_MFMA_16X16X16_BF16(a, b, c) emul::mfma_16x16x16_bf16((a), (b), (c))
#define COT_KEEP_PACKED(u) ((void)(u))
#define COT_WAIT_LOADS() ((void)0)
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only ever applied to wave-uniform values
inline float __expf(float x) { return std::exp(x); }
#define COT_RCP(x) (1.0f / (x))
using std::min;
using std::max;
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
// only the two DPP controls the kernels use: 0x138 = wave_shr:1 (lane l <- l-1), 0x130 = wave_shl:1 (l <- l+1)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
    return emul::exchange<int>(src, ctrl == 0x138 ? -1 : +1, bound_ctrl ? 0 : old);  // bound_ctrl: no source lane -> 0
}
// agg_dot2.hip's primitives: DPP row shifts inside a 16-lane row with zero fill at the row's ends (row_shr:1 / row_shl:1,
// bound_ctrl) and the packed-bf16 dot product v_dot2c_f32_bf16 (fp32 products of bf16 values are exact; the two products
// are added to the accumulator one after the other here -- the hardware's internal order is checked on the GPU at the
// tests' bf16 tolerance, integer-valued data is exact either way)
namespace emul {
inline uint32_t exchange_row16(uint32_t v, int delta) {
    const int src = (t_lane & 15) + delta;
    const uint32_t r = exchange<uint32_t>(v, delta, 0u);
    return (src >= 0 && src < 16) ? r : 0u;
}
inline float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    acc += bf16_bits_to_float((uint16_t)(a & 0xffffu)) * bf16_bits_to_float((uint16_t)(b & 0xffffu));
    acc += bf16_bits_to_float((uint16_t)(a >> 16)) * bf16_bits_to_float((uint16_t)(b >> 16));
    return acc;
}
}  // namespace emul
#define COT_DOT2_BF16(a, b, acc) emul::dot2_bf16((uint32_t)(a), (uint32_t)(b), (acc))
#define COT_ROW_PREV(v) emul::exchange_row16((uint32_t)(v), -1)
#define COT_ROW_NEXT(v) emul::exchange_row16((uint32_t)(v), +1)
#define COT_PACK_LO(a, b) (((uint32_t)(a) & 0xffffu) | ((uint32_t)(b) << 16))
#define COT_PACK_HI(a, b) (((uint32_t)(a) >> 16) | ((uint32_t)(b) & 0xffff0000u))
template <typename V> inline V __shfl_up(V v, int d) { return emul::exchange<V>(v, -d, v); }
template <typename V> inline V __shfl_down(V v, int d) { return emul::exchange<V>(v, +d, v); }
template <typename V> inline V __shfl_xor(V v, int m) { return emul::exchange<V>(v, (emul::t_lane ^ m) - emul::t_lane, v); }

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
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
