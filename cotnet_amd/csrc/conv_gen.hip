// conv_gen.hip -- general grouped convolutions on NCHW tensors (1x1, and 3x3 / stride 1 / padding 1) for what the
// specialised bf16 kernels (conv1x1.hip, conv3x3g.hip, conv_lds.hip) do not take:
//   * fp32 tensors -- the reference trains in fp32 (cot_experiments/*/config.yaml `amp: False`), so these are the kernels
//     that run rows a6-a8 at the reference's own precision: v_mfma_f32_16x16x4_f32, fp32 operands, fp32 accumulation;
//   * grouped 1x1 convolutions (CoXtLayer.embed[0] / embed[3] / conv1x1[0], groups = 2, models/cotnet.py:123-131) in bf16
//     or fp32, with any number of channels per group (12 / 24 / 48 / 54 ... in CoTNeXt-50 2x48d);
//   * grouped 3x3 convolutions whose channels per group are not a multiple of 8 (CoXtLayer.key_embed, groups = 8, 12 / 24
//     channels per group in the first two stages of CoTNeXt, models/cotnet.py:112-116).
//
// One implicit-GEMM form covers forward and backward-data:
//     out[n][g*Og + o][p] (+)= sum_tap sum_{i < Ig} Wsel(g, o, i, tap) * in[n][g*Ig + i][p + shift(tap)]  (+ bias)
// forward reads the weight tensor [Cout][Cin/G][taps] as it is, backward-data reads the same tensor with the roles of o and
// i swapped and the tap index mirrored (strides wso / wsi and `wflip`) -- no transposed copy.  The batch and the plane are
// flattened to one pixel index q = n*HW + p, so 7x7 planes fill 64-pixel tiles as well as 56x56 ones.
//
// Tiling: a 256-lane workgroup owns 64 output channels of one group x 64 pixels; per step it stages a [64 o][16 i] weight
// tile and a [64 px][16 i] input tile in LDS, reduction index contiguous (the input tile is transposed while it is written:
// lanes load along pixels -- coalesced -- and each writes 4 consecutive i of its pixel).  Channels past the group's end are
// staged as zeros, so any Ig / Og is handled by the same code.  Global loads of step s+1 are in flight while step s is
// multiplied.  Each wave multiplies 16 output channels x 64 pixels: one A fragment, four B fragments per step.
//     fp32:  lane l reads 4 consecutive i (16 B) of row l&15, quad l>>4; MFMA step s (of 4) multiplies component s of every
//            lane, i.e. reduction index 4*(l>>4) + s -- the same bijection on both operands, so the order of the sum over i
//            is permuted but the product is the same GEMM;
//     bf16:  the same 4 consecutive i (8 B) are exactly v_mfma_f32_16x16x16_bf16's operand (k = 4*(l>>4) .. +3).
// Weight gradient: the reduction runs over pixels instead; tiles [64 m][16 q] of dY and [64 i][16 q] of (shifted) x are
// already reduction-contiguous in memory.  Workgroups split the batch; partial sums go to the fp32 workspace in a fixed
// order and one small kernel adds them (deterministic, no atomics) and rounds once.
// These kernels are MFMA-bound in fp32 (157 TFLOP/s dense fp32 matrix peak, 1/16 of bf16) and staging-bound in bf16; they
// are the general path, not the tuned one -- DESIGN.md 4.16.
#include <algorithm>

#include "mfma_common.h"

namespace cot {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

#ifndef COT_MFMA_16X16X4_F32  // (tests/emul pre-defines both primitives for its host build)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#define COT_MFMA_16X16X4_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define COT_MFMA_16X16X16_BF16(a, b, c)                                                                                 \
    __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(cot::s16x4_t, (a)), __builtin_bit_cast(cot::s16x4_t, (b)), \
                                              (c), 0, 0, 0)
#endif

namespace gen {

constexpr int TILE = 64;  // output channels (rows) and pixels / input channels (columns) per workgroup
constexpr int BK = 16;    // reduction elements per step
constexpr int LD = 20;    // LDS row stride in elements (80 B fp32 / 40 B bf16: fragment reads stay 16- / 8-byte aligned)

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4_t type; };
template <> struct Frag<bf16_t> { typedef bf16x4_t type; };

template <typename T> struct alignas(sizeof(T) * 4) Quad { T v[4]; };

__device__ __forceinline__ f32x4_t mma(const f32x4_t& a, const f32x4_t& b, f32x4_t c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = COT_MFMA_16X16X4_F32(a[s], b[s], c);
    return c;
}
__device__ __forceinline__ f32x4_t mma(const bf16x4_t& a, const bf16x4_t& b, f32x4_t c) {
    return COT_MFMA_16X16X16_BF16(a, b, c);
}

// consecutive logical ids on the same XCD (the dispatcher deals blocks round-robin over 8 XCDs): the workgroups that share
// an input tile run next to each other on one L2
__device__ __forceinline__ unsigned logical_id() {
    const unsigned b = blockIdx.x, nblk = gridDim.x;
    return (nblk & 7u) ? b : (b & 7u) * (nblk >> 3) + (b >> 3);
}

struct Args {
    const void* x;
    const void* w;
    const void* bias;
    void* y;
    int N, Ci, Co;        // channels of the whole input / output tensor
    int G, Ig, Og;        // groups; input / output channels per group of THIS product
    int H, W, taps;       // taps = 1 or 9
    int64_t wso, wsi, wgs;  // weight strides in elements: output channel, input channel, group (taps are contiguous)
    int wflip;            // backward data: tap index mirrored
    int accumulate;       // y += result
    int otiles;           // ceil(Og / 64)
};

template <typename T> __global__ __launch_bounds__(256) void convg_fwd_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) T As[TILE * LD];
    __shared__ __attribute__((aligned(16))) T Bs[TILE * LD];
    typedef typename Frag<T>::type F;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = a.H * a.W;
    const int64_t Q = (int64_t)a.N * HW;
    const unsigned lb = logical_id();
    const int per_q = a.G * a.otiles;
    const int64_t q0 = (int64_t)(lb / per_q) * TILE;
    const int go = lb % per_q, g = go / a.otiles, ot = go - g * a.otiles;

    // staging roles: input tile -- pixel bpx, channels 4*bkq .. +3 of the step; weight tile -- row ao, channels 4*aiq .. +3
    const int bpx = t & 63, bkq = t >> 6, ao = t >> 2, aiq = t & 3;
    const int64_t q = q0 + bpx;
    const bool qok = q < Q;
    int n = 0, p = 0, h = 0, wc = 0;
    if (qok) {
        n = (int)(q / HW);
        p = (int)(q - (int64_t)n * HW);
        h = p / a.W;
        wc = p - h * a.W;
    }
    const T* xb = (const T*)a.x + ((int64_t)n * a.Ci + (int64_t)g * a.Ig) * HW;
    const int orow = ot * TILE + ao;
    const bool ook = orow < a.Og;
    const T* wb = (const T*)a.w + (int64_t)g * a.wgs + (int64_t)orow * a.wso;
    const int ksteps = (a.Ig + BK - 1) / BK, nsteps = a.taps * ksteps;

    Quad<T> xr, wr;
    auto load_step = [&](int s) __attribute__((always_inline)) {
        const int tap = s / ksteps, kb = (s - tap * ksteps) * BK;
        int dh = 0, dw = 0;
        if (a.taps == 9) {
            dh = tap / 3 - 1;
            dw = tap - (tap / 3) * 3 - 1;
        }
        const bool pok = qok && (unsigned)(h + dh) < (unsigned)a.H && (unsigned)(wc + dw) < (unsigned)a.W;
        const int64_t po = (int64_t)p + dh * a.W + dw;
        const int tapw = a.wflip ? a.taps - 1 - tap : tap;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kb + 4 * bkq + e, i = kb + 4 * aiq + e;
            xr.v[e] = (pok && k < a.Ig) ? xb[(int64_t)k * HW + po] : (T)0.f;
            wr.v[e] = (ook && i < a.Ig) ? wb[(int64_t)i * a.wsi + tapw] : (T)0.f;
        }
    };
    f32x4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    load_step(0);
    const int fr = lane & 15, fq = lane >> 4;
    for (int s = 0; s < nsteps; ++s) {
        *reinterpret_cast<Quad<T>*>(&Bs[bpx * LD + 4 * bkq]) = xr;
        *reinterpret_cast<Quad<T>*>(&As[ao * LD + 4 * aiq]) = wr;
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
        const F fa = *reinterpret_cast<const F*>(&As[(wave * 16 + fr) * LD + 4 * fq]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const F fb = *reinterpret_cast<const F*>(&Bs[(b * 16 + fr) * LD + 4 * fq]);
            acc[b] = mma(fa, fb, acc[b]);
        }
        __syncthreads();
    }
    // D[i = 4*(lane>>4) + r][j = lane&15]: rows = output channels, columns = pixels
    T* y = (T*)a.y;
    const T* bias = (const T*)a.bias;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t qq = q0 + b * 16 + fr;
        if (qq >= Q) continue;
        const int n2 = (int)(qq / HW);
        const int p2 = (int)(qq - (int64_t)n2 * HW);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = ot * TILE + wave * 16 + 4 * fq + r;
            if (o >= a.Og) continue;
            const int64_t idx = ((int64_t)n2 * a.Co + (int64_t)g * a.Og + o) * HW + p2;
            float v = acc[b][r];
            if (bias) v += (float)bias[g * a.Og + o];
            if (a.accumulate) v += (float)y[idx];
            y[idx] = (T)v;
        }
    }
}

struct WgArgs {
    const void* gy;
    const void* x;
    float* part;          // [splits][Co*Ig*taps] then [splits][Co]
    int N, Ci, Co, G, Ig, Og, H, W, taps, splits;
    int mtiles, ktiles;   // ceil(Og / 64), ceil(Ig / 64)
    int want_bias;
};

template <typename T> __global__ __launch_bounds__(256) void convg_wgrad_kernel(WgArgs a) {
    __shared__ __attribute__((aligned(16))) T As[TILE * LD];
    __shared__ __attribute__((aligned(16))) T Bs[TILE * LD];
    typedef typename Frag<T>::type F;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = a.H * a.W;
    int id = blockIdx.x;
    const int tap = id % a.taps;
    id /= a.taps;
    const int kt = id % a.ktiles;
    id /= a.ktiles;
    const int mt = id % a.mtiles, g = id / a.mtiles;
    const int split = blockIdx.y;
    int dh = 0, dw = 0;
    if (a.taps == 9) {
        dh = tap / 3 - 1;
        dw = tap - (tap / 3) * 3 - 1;
    }
    const int row = t >> 2, qq = t & 3;
    const int m = mt * TILE + row, k = kt * TILE + row;
    const bool mok = m < a.Og, kok = k < a.Ig;
    // the reduction runs over (image, 16-pixel block) pairs; a split owns a contiguous range of them, so even a 56 x 56 plane
    // of a 64-channel layer (4 output tiles) spreads over the whole chip
    const int psteps = (HW + BK - 1) / BK;
    const int64_t all_steps = (int64_t)a.N * psteps;
    const int64_t s0 = all_steps * split / a.splits;
    const int nsteps = (int)(all_steps * (split + 1) / a.splits - s0);
    const bool do_bias = a.want_bias && kt == 0 && tap == 0;
    float bsum = 0.f;

    Quad<T> ar, br;
    auto load_step = [&](int s) __attribute__((always_inline)) {
        const int64_t sg = s0 + s;
        const int n = (int)(sg / psteps), pb = (int)(sg - (int64_t)n * psteps) * BK + 4 * qq;
        const T* gp = (const T*)a.gy + ((int64_t)n * a.Co + (int64_t)g * a.Og + m) * HW;
        const T* xp = (const T*)a.x + ((int64_t)n * a.Ci + (int64_t)g * a.Ig + k) * HW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = pb + e;
            ar.v[e] = (mok && p < HW) ? gp[p] : (T)0.f;
            bool ok = kok && p < HW;
            int ps = p;
            if (a.taps == 9) {
                const int h = p / a.W, wc = p - h * a.W;
                ok = ok && (unsigned)(h + dh) < (unsigned)a.H && (unsigned)(wc + dw) < (unsigned)a.W;
                ps = p + dh * a.W + dw;
            }
            br.v[e] = ok ? xp[ps] : (T)0.f;
        }
    };
    f32x4_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) load_step(0);
    const int fr = lane & 15, fq = lane >> 4;
    for (int s = 0; s < nsteps; ++s) {
        *reinterpret_cast<Quad<T>*>(&As[row * LD + 4 * qq]) = ar;
        *reinterpret_cast<Quad<T>*>(&Bs[row * LD + 4 * qq]) = br;
        if (do_bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum += (float)ar.v[e];
        }
        __syncthreads();
        if (s + 1 < nsteps) load_step(s + 1);
        const F fa = *reinterpret_cast<const F*>(&As[(wave * 16 + fr) * LD + 4 * fq]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const F fb = *reinterpret_cast<const F*>(&Bs[(b * 16 + fr) * LD + 4 * fq]);
            acc[b] = mma(fa, fb, acc[b]);
        }
        __syncthreads();
    }
    const int64_t PS = (int64_t)a.Co * a.Ig * a.taps;
    float* part = a.part + (int64_t)split * PS;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int kk = kt * TILE + b * 16 + fr;
        if (kk >= a.Ig) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = mt * TILE + wave * 16 + 4 * fq + r;
            if (mm < a.Og) part[(((int64_t)g * a.Og + mm) * a.Ig + kk) * a.taps + tap] = acc[b][r];
        }
    }
    if (do_bias) {  // the four lanes of a row sit next to each other in the wave
        bsum += __shfl_xor(bsum, 1);
        bsum += __shfl_xor(bsum, 2);
        if (qq == 0 && mok) a.part[(int64_t)a.splits * PS + (int64_t)split * a.Co + g * a.Og + m] = bsum;
    }
}

// gw[i] = sum over splits of part[k][i] (bias entries appended after the weight entries).  16 elements x 16 split lanes
// per workgroup: a lane adds every 16th split, the 16 lane sums are added in a fixed order through LDS -- deterministic, and
// a 256-split reduction is 16 dependent loads deep instead of 256.
template <typename T>
__global__ __launch_bounds__(256) void convg_wgrad_reduce(const float* __restrict__ part, T* __restrict__ gw,
                                                         T* __restrict__ gbias, int64_t PS, int Co, int splits) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + e, total = PS + (gbias ? Co : 0);
    float s = 0.f;
    if (i < total) {
        const float* src = i < PS ? part + i : part + (int64_t)splits * PS + (i - PS);
        const int64_t stride = i < PS ? PS : Co;
        for (int k = sl; k < splits; k += 16) s += src[(int64_t)k * stride];
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e];
        if (i < PS) gw[i] = (T)t;
        else gbias[i - PS] = (T)t;
    }
}

}  // namespace gen

static bool convg_shape_ok(int N, int Cin, int Cout, int G, int H, int W, int ksize) {
    return N > 0 && Cin > 0 && Cout > 0 && G > 0 && H > 0 && W > 0 && Cin % G == 0 && Cout % G == 0 &&
           (ksize == 1 || ksize == 3) && (int64_t)N * H * W < (int64_t)1 << 31;
}

static int convg_launch(const gen::Args& a, int dtype, hipStream_t stream) {
    const int64_t Q = (int64_t)a.N * a.H * a.W;
    const int64_t blocks = ceil_div64(Q, gen::TILE) * a.G * a.otiles;
    if (blocks >= ((int64_t)1 << 31)) return set_error(COT_ERR_UNSUPPORTED, "convg: %lld workgroups", (long long)blocks);
    if (dtype == COT_F32)
        COT_LAUNCH((gen::convg_fwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else
        COT_LAUNCH((gen::convg_fwd_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return check_launch("convg_fwd_kernel");
}

int convg_forward(const void* x, const void* w, const void* bias, void* y, int N, int Cin, int Cout, int G, int H, int W,
                  int ksize, int accumulate, int dtype, hipStream_t stream) {
    if (!convg_shape_ok(N, Cin, Cout, G, H, W, ksize) || (dtype != COT_F32 && dtype != COT_BF16)) return COT_ERR_UNSUPPORTED;
    gen::Args a;
    const int taps = ksize * ksize;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    a.N = N; a.Ci = Cin; a.Co = Cout; a.G = G; a.Ig = Cin / G; a.Og = Cout / G;
    a.H = H; a.W = W; a.taps = taps;
    a.wso = (int64_t)a.Ig * taps; a.wsi = taps; a.wgs = (int64_t)a.Og * a.Ig * taps;
    a.wflip = 0; a.accumulate = accumulate; a.otiles = ceil_div(a.Og, gen::TILE);
    return convg_launch(a, dtype, stream);
}

// gx[n][g*Kg + k][p] (+)= sum_tap sum_m w[g*Mg + m][k][tap] * gy[n][g*Mg + m][p - shift(tap)]
int convg_backward_data(const void* gy, const void* w, void* gx, int N, int Cin, int Cout, int G, int H, int W, int ksize,
                        int accumulate, int dtype, hipStream_t stream) {
    if (!convg_shape_ok(N, Cin, Cout, G, H, W, ksize) || (dtype != COT_F32 && dtype != COT_BF16)) return COT_ERR_UNSUPPORTED;
    gen::Args a;
    const int taps = ksize * ksize, Kg = Cin / G, Mg = Cout / G;
    a.x = gy; a.w = w; a.bias = nullptr; a.y = gx;
    a.N = N; a.Ci = Cout; a.Co = Cin; a.G = G; a.Ig = Mg; a.Og = Kg;
    a.H = H; a.W = W; a.taps = taps;
    a.wso = taps; a.wsi = (int64_t)Kg * taps; a.wgs = (int64_t)Mg * Kg * taps;
    a.wflip = 1; a.accumulate = accumulate; a.otiles = ceil_div(a.Og, gen::TILE);
    return convg_launch(a, dtype, stream);
}

// splits of the pixel reduction: enough workgroups for ~4 per CU, at least 8 steps (128 pixels) per workgroup, and at most
// 32 MB of partial sums
static int convg_splits(int N, int Cin, int Cout, int G, int HW, int ksize) {
    const int64_t tiles = (int64_t)G * ceil_div(Cout / G, gen::TILE) * ceil_div(Cin / G, gen::TILE) * ksize * ksize;
    const int64_t steps = (int64_t)N * ceil_div(HW, gen::BK);
    const int64_t PS = (int64_t)Cout * (Cin / G) * ksize * ksize;
    int64_t s = (1024 + tiles - 1) / tiles;
    s = std::min(s, std::max<int64_t>(1, steps / 8));
    s = std::min(s, std::max<int64_t>(1, ((int64_t)32 << 20) / ((PS + Cout) * 4)));
    s = std::min<int64_t>(s, 512);
    return s < 1 ? 1 : (int)s;
}

int64_t convg_workspace(int N, int Cin, int Cout, int G, int H, int W, int ksize) {
    if (!convg_shape_ok(N, Cin, Cout, G, H, W, ksize)) return 0;
    const int64_t PS = (int64_t)Cout * (Cin / G) * ksize * ksize;
    return ((int64_t)convg_splits(N, Cin, Cout, G, H * W, ksize) * (PS + Cout) * 4 + 255) / 256 * 256;
}

int convg_backward_weight(const void* gy, const void* x, void* gw, void* gbias, float* workspace, int N, int Cin, int Cout,
                          int G, int H, int W, int ksize, int dtype, hipStream_t stream) {
    if (!convg_shape_ok(N, Cin, Cout, G, H, W, ksize) || (dtype != COT_F32 && dtype != COT_BF16)) return COT_ERR_UNSUPPORTED;
    gen::WgArgs a;
    a.gy = gy; a.x = x; a.part = workspace;
    a.N = N; a.Ci = Cin; a.Co = Cout; a.G = G; a.Ig = Cin / G; a.Og = Cout / G; a.H = H; a.W = W;
    a.taps = ksize * ksize; a.splits = convg_splits(N, Cin, Cout, G, H * W, ksize);
    a.mtiles = ceil_div(a.Og, gen::TILE); a.ktiles = ceil_div(a.Ig, gen::TILE);
    a.want_bias = gbias != nullptr;
    const dim3 grid((unsigned)(G * a.mtiles * a.ktiles * a.taps), (unsigned)a.splits);
    const int64_t PS = (int64_t)Cout * a.Ig * a.taps;
    const dim3 rgrid((unsigned)ceil_div64(PS + (gbias ? Cout : 0), 16));
    if (dtype == COT_F32) {
        COT_LAUNCH((gen::convg_wgrad_kernel<float>), grid, dim3(256), 0, stream, a);
        COT_LAUNCH((gen::convg_wgrad_reduce<float>), rgrid, dim3(256), 0, stream, (const float*)workspace, (float*)gw,
                   (float*)gbias, PS, Cout, a.splits);
    } else {
        COT_LAUNCH((gen::convg_wgrad_kernel<bf16_t>), grid, dim3(256), 0, stream, a);
        COT_LAUNCH((gen::convg_wgrad_reduce<bf16_t>), rgrid, dim3(256), 0, stream, (const float*)workspace, (bf16_t*)gw,
                   (bf16_t*)gbias, PS, Cout, a.splits);
    }
    return check_launch("convg_wgrad_kernel");
}

}  // namespace cot
