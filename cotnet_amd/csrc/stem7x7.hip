// stem7x7.hip -- the backbone's first convolution: 7x7, stride 2, padding 3, 3 -> 64 channels, NCHW bf16
// (reference: models/resnet.py:539-555, `self.conv1 = nn.Conv2d(in_chans, inplanes, kernel_size=7, stride=2, padding=3,
// bias=False)`), forward and weight gradient (the network input needs no data gradient).
//
// Why: with the CoT blocks, poolings and the head on this library's kernels it is the last MIOpen call of a CoTNet-50
// training step (an NHWC implicit GEMM bracketed by layout transposes; its weight gradient accumulates with atomics into a
// zeroed buffer -- the kind of kernel HIP-graph replay tripped over in round 1).
//
// Both kernels are implicit GEMMs on v_mfma_f32_16x16x32_bf16 (operand maps: mfma_common.h) with K = (ci, kh, kw) =
// 147 taps in the weight tensor's own memory order, padded to 160:
//   forward   Y (64 x pixels) = Wt (64 x 160) * B (160 x pixels); Wt staged once per workgroup in LDS (20 KB), B gathered
//             from x (a 300 KB image: L1/L2 hits) with bounds predicates.  MFMA column j of column set cs is output pixel
//             4j + cs, so a lane ends up with 4 consecutive pixels of 4 channels -> 8-byte stores.
//   wgrad     dW (64 x 160) = sum over pixels dY (64 x pixels) * B^T; dY fragments are 16-byte loads (pixels are
//             contiguous), B fragments are 8 strided taps of one (ci, kh, kw) each; deterministic slices + reduce kernel.
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb,
                                hipStream_t stream);  // conv1x1.hip

constexpr int kStemK = 147, kStemKp = 160, kStemCo = 64;

// tap index k -> (ci, kh, kw)
__device__ __forceinline__ void stem_tap(int k, int& ci, int& kh, int& kw) {
    ci = k / 49;
    const int r = k - 49 * ci;
    kh = r / 7;
    kw = r - 7 * kh;
}

__global__ void __launch_bounds__(256)
stem7x7_fwd_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int H, int W, int Ho,
                 int Wo, int tiles_per_image, int64_t total_waves) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(cot_smem);  // [64][160], zero beyond tap 146
    for (int i = threadIdx.x; i < kStemCo * kStemKp; i += blockDim.x) {
        const int m = i / kStemKp, k = i - m * kStemKp;
        As[i] = k < kStemK ? w[m * kStemK + k] : (bf16_t)0.0f;
    }
    __syncthreads();
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int n = uniform((int)(wid / tiles_per_image)), tile = uniform((int)(wid % tiles_per_image));
    const int HWo = Ho * Wo;
    const int p0 = tile * 64 + 4 * j;  // this lane's 4 consecutive output pixels (one row: Wo % 4 == 0)
    const int oh = min(p0, HWo - 1) / Wo, ow0 = min(p0, HWo - 1) % Wo;
    const int ih0 = 2 * oh - 3, iw0 = 2 * ow0 - 3;
    const bf16_t* xn = x + (int64_t)n * 3 * H * W;

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int step = 0; step < kStemKp / 32; ++step) {
        const int kb = 32 * step + 8 * g;
        bf16x8_t bfrag[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kb + e;
            int ci, kh, kw;
            stem_tap(min(k, kStemK - 1), ci, kh, kw);
            const int ih = ih0 + kh;
            const bool rok = k < kStemK && ih >= 0 && ih < H && p0 < HWo;
            const bf16_t* row = xn + ((int64_t)ci * H + (rok ? ih : 0)) * W;
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const int iw = iw0 + 2 * cs + kw;
                bfrag[cs][e] = (rok && iw >= 0 && iw < W) ? row[iw] : (bf16_t)0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8_t af;
            __builtin_memcpy(&af, __builtin_assume_aligned(As + (16 * a + j) * kStemKp + kb, 16), 16);
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) acc[a][cs] = COT_MFMA_16X16X32_BF16(af, bfrag[cs], acc[a][cs]);
        }
    }
    if (p0 >= HWo) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * a + 4 * g + i;
            bf16_t o[4];
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) o[cs] = (bf16_t)acc[a][cs][i];
            store_piece<4, 8>(y + ((int64_t)n * kStemCo + co) * HWo + p0, o, HWo - p0);
        }
}

// ---- forward with the input patch staged in LDS (round 4).  The kernel above gathers its B operand from global memory: 160
// bounds-predicated 2-byte loads per lane for 80 MFMAs (226 us for 80 x 3 x 224 x 224: 152 MB of traffic, 19 us at the roofline).
// Here a workgroup owns 256 consecutive output pixels of one image (4 waves x 64); the input rows those pixels read -- NR rows x 3
// channels, each row with 8 zero elements in front and behind (the horizontal padding; rows outside the image are zeros too) --
// are copied to LDS once with 16-byte accesses, and every B element is a 2-byte LDS read at (lane's patch position) + (the tap's
// offset, from a 160-entry table in LDS: 8 entries per lane group and K step come with one ds_read_b128) + (the column set as an
// immediate).  Weights: [64][168] in LDS (row stride 336 B: the 16 rows of a fragment read fall on distinct banks).  Same sums in
// the same order as the kernel above: identical results.
__global__ void __launch_bounds__(256)
stem7x7_fwd_lds(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int H, int W, int Ho, int Wo,
                int tiles_per_image, int NR, int RS) {
    constexpr int WS = 168;  // weight row stride (elements)
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* As = reinterpret_cast<bf16_t*>(cot_smem);                      // [64][WS], zero beyond tap 146
    uint16_t* toff = reinterpret_cast<uint16_t*>(As + kStemCo * WS);       // [160] tap -> element offset inside the patch
    bf16_t* patch = reinterpret_cast<bf16_t*>(toff + kStemKp);             // [3][NR][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    const int n = blockIdx.x / tiles_per_image, tile = blockIdx.x - n * tiles_per_image;
    const int HWo = Ho * Wo;
    const int pfirst = tile * 256;
    const int ih_base = 2 * (pfirst / Wo) - 3;  // first input row of the patch
    for (int i = tid; i < kStemCo * WS; i += 256) {
        const int m = i / WS, k = i - m * WS;
        As[i] = k < kStemK ? w[m * kStemK + k] : (bf16_t)0.0f;
    }
    if (tid < kStemKp) {
        int ci, kh, kw;
        stem_tap(min(tid, kStemK - 1), ci, kh, kw);
        toff[tid] = (uint16_t)((ci * NR + kh) * RS + kw);
    }
    {
        const int cpr = RS / 8, chunks = 3 * NR * cpr;  // 16-byte pieces per patch row (first and last: the zero margins)
        const bf16_t* xn = x + (int64_t)n * 3 * H * W;
        for (int q = tid; q < chunks; q += 256) {
            const int row = q / cpr, c = q - row * cpr;
            const int ci = row / NR, ih = ih_base + (row - ci * NR);
            Vec<bf16_t, 8> v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v.v[e] = (bf16_t)0.0f;
            if (c >= 1 && 8 * (c - 1) < W && ih >= 0 && ih < H) v = ldv<bf16_t, 8>(xn + ((int64_t)ci * H + ih) * W + 8 * (c - 1));
            stv<bf16_t, 8>(patch + (int64_t)row * RS + 8 * c, v);
        }
    }
    __syncthreads();
    const int p0 = pfirst + wave * 64 + 4 * j;  // this lane's 4 consecutive output pixels (one row: Wo % 4 == 0)
    const int pc = min(p0, HWo - 4);
    const int oh = pc / Wo, ow0 = pc - oh * Wo;
    const uint16_t* pl = reinterpret_cast<const uint16_t*>(patch) + (2 * oh - 3 - ih_base) * RS + 8 + 2 * ow0 - 3;

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int step = 0; step < kStemKp / 32; ++step) {
        const int kb = 32 * step + 8 * g;
        uint16_t to[8];
        __builtin_memcpy(to, __builtin_assume_aligned(toff + kb, 16), 16);
        uint16_t q[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint16_t* pe = pl + to[e];
            const bool kok = step < 4 || kb + e < kStemK;  // (the padded taps: zeros by selection -- their weights are zeros, but 0 * Inf is not)
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
                const uint16_t v = pe[2 * cs];
                q[cs][e] = kok ? v : (uint16_t)0;
            }
        }
        bf16x8_t bfrag[4];
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) __builtin_memcpy(&bfrag[cs], q[cs], 16);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8_t af;
            __builtin_memcpy(&af, __builtin_assume_aligned(As + (16 * a + j) * WS + kb, 16), 16);
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) acc[a][cs] = COT_MFMA_16X16X32_BF16(af, bfrag[cs], acc[a][cs]);
        }
    }
    if (p0 >= HWo) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * a + 4 * g + i;
            bf16_t o[4];
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) o[cs] = (bf16_t)acc[a][cs][i];
            store_piece<4, 8>(y + ((int64_t)n * kStemCo + co) * HWo + p0, o, HWo - p0);
        }
}

// part[s][co][k] = sum over slice s of (n, pixel) of dY[n][co][p] * x[n][ci][2*oh - 3 + kh][2*ow - 3 + kw]
// one wave = (80 of the 160 tap columns, slice s); 4 waves per workgroup
__global__ void __launch_bounds__(256)
stem7x7_wgrad_mfma(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, float* __restrict__ part, int N, int H,
                   int W, int Ho, int Wo, int S, int spi, int64_t total_waves) {
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, i16 = lane & 15, lg = lane >> 4;
    const int cb = uniform((int)(wid & 1)), s = uniform((int)(wid >> 1));
    const int HWo = Ho * Wo;
    const int T = N * spi, t0 = (int)((int64_t)T * s / S), t1 = (int)((int64_t)T * (s + 1) / S);
    int tci[5], tkh[5], tkw[5];
    bool tok[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int k = cb * 80 + q * 16 + i16;
        tok[q] = k < kStemK;
        stem_tap(min(k, kStemK - 1), tci[q], tkh[q], tkw[q]);
    }
    f32x4_t acc[4][5];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[a][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int t = t0; t < t1; ++t) {
        const int n = t / spi, st = t - n * spi;
        const int p = st * 32 + 8 * lg;  // 8 consecutive output pixels of one row (Wo % 8 == 0, HWo % 32 == 0)
        const int oh = p / Wo, ow = p - oh * Wo;
        bf16x8_t af[4], bfr[5];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            __builtin_memcpy(&af[a], __builtin_assume_aligned(gy + ((int64_t)n * kStemCo + 16 * a + i16) * HWo + p, 16), 16);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int ih = 2 * oh - 3 + tkh[q];
            const bool rok = tok[q] && ih >= 0 && ih < H;
            const bf16_t* row = x + (((int64_t)n * 3 + tci[q]) * H + (rok ? ih : 0)) * W;
            const int iw0 = 2 * ow - 3 + tkw[q];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iw = iw0 + 2 * e;
                bfr[q][e] = (rok && iw >= 0 && iw < W) ? row[iw] : (bf16_t)0.0f;
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[a][q] = COT_MFMA_16X16X32_BF16(af[a], bfr[q], acc[a][q]);
    }
    float* ps = part + (int64_t)s * kStemCo * kStemK;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 16 * a + 4 * lg + i;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int k = cb * 80 + q * 16 + i16;
                if (k < kStemK) ps[co * kStemK + k] = acc[a][q][i];
            }
        }
}

// ---- weight gradient with the input rows staged in LDS (round 4).  The kernel above gathers its B operand -- 8 stride-2 taps per
// fragment -- from global memory: 40 two-byte loads per lane and 32-pixel step, 165 us per launch at the very end of the backward
// pass.  Here a workgroup (4 waves) owns a slice of the (image, output-row PAIR) sequence; per pair it stages the nine input rows x
// three channels the pair's 2 Wo pixels read (whole rows with zero margins, 16-byte copies, the next pair's rows fetched into
// registers while this pair is multiplied: one barrier per pair) and takes every B element from LDS at (tap offset) + (pixel
// position) + (8 immediates).  The four waves split dW: wave w owns tap columns 80 (w & 1) .. +79 and output channels 32 (w >> 1) ..
// +31.  Partial sums in the layout of the kernel above (same reduce).  2 Wo % 32 == 0, Ho even, W % 8 == 0.
__global__ void __launch_bounds__(256)
stem7x7_wgrad_lds(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, float* __restrict__ part, int N, int H, int W, int Ho,
                  int Wo, int S, int RS) {
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    bf16_t* const patch = reinterpret_cast<bf16_t*>(cot_smem);  // [2][27][RS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, lg = lane >> 4;
    const int cb = wave & 1, mh = wave >> 1, s = blockIdx.x;
    const int HWo = Ho * Wo, ppi = Ho / 2;  // row pairs per image
    const int T = N * ppi, t0 = (int)((int64_t)T * s / S), t1 = (int)((int64_t)T * (s + 1) / S);
    const int cpr = RS / 8, chunks = 27 * cpr;
    constexpr int NCH = 4;  // 16-byte pieces per thread and pair (host: chunks <= 4 * 256)
    int toff[5];
    bool tok[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int k = cb * 80 + q * 16 + i16;
        int ci, kh, kw;
        stem_tap(min(k, kStemK - 1), ci, kh, kw);
        tok[q] = k < kStemK;
        toff[q] = (ci * 9 + kh) * RS + 8 - 3 + kw;
    }
    f32x4_t acc[2][5];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[a][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    Vec<bf16_t, 8> hold[NCH];
    auto fetch = [&](int t) __attribute__((always_inline)) {  // rows of pair t -> registers (zeros outside the image / in the margins)
        const int n = t / ppi, pr = t - n * ppi, ih0 = 4 * pr - 3;
        const bf16_t* xn = x + (int64_t)n * 3 * H * W;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int qd = j * 256 + tid;
#pragma unroll
            for (int e = 0; e < 8; ++e) hold[j].v[e] = (bf16_t)0.0f;
            if (qd < chunks) {
                const int row = qd / cpr, c = qd - row * cpr;
                const int ci = row / 9, ih = ih0 + (row - ci * 9);
                if (c >= 1 && 8 * (c - 1) < W && ih >= 0 && ih < H) hold[j] = ldv<bf16_t, 8>(xn + ((int64_t)ci * H + ih) * W + 8 * (c - 1));
            }
        }
    };
    auto put = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int qd = j * 256 + tid;
            if (qd < chunks) stv<bf16_t, 8>(patch + (int64_t)buf * 27 * RS + (int64_t)qd * 8, hold[j]);
        }
    };
    if (t0 < t1) {
        fetch(t0);
        put(0);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        if (t + 1 < t1) fetch(t + 1);
        const int n = t / ppi, pr = t - n * ppi;
        const uint16_t* pb = reinterpret_cast<const uint16_t*>(patch) + buf * 27 * RS;
        const bf16_t* gyn = gy + (int64_t)n * kStemCo * HWo + (int64_t)(2 * pr) * Wo;  // the pair's 2 Wo pixels are contiguous in dY
        for (int ks = 0; ks < 2 * Wo; ks += 32) {
            const int p = ks + 8 * lg;            // 8 consecutive pixels of one output row (Wo % 8 == 0)
            const int r2 = p >= Wo ? 1 : 0, ow = p - r2 * Wo;
            const int pbase = 2 * r2 * RS + 2 * ow;  // (input row 2 (2 pr + r2) - 3 = patch row 2 r2; column 2 ow - 3 = index 8 + 2 ow - 3)
            bf16x8_t af[2], bfr[5];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                __builtin_memcpy(&af[a], __builtin_assume_aligned(gyn + (int64_t)(32 * mh + 16 * a + i16) * HWo + p, 16), 16);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const uint16_t* pe = pb + pbase + toff[q];
                uint16_t v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tok[q] ? pe[2 * e] : (uint16_t)0;
                __builtin_memcpy(&bfr[q], v, 16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 5; ++q) acc[a][q] = COT_MFMA_16X16X32_BF16(af[a], bfr[q], acc[a][q]);
        }
        if (t + 1 < t1) put(buf ^ 1);
        __syncthreads();
    }
    float* ps = part + (int64_t)s * kStemCo * kStemK;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 32 * mh + 16 * a + 4 * lg + i;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int k = cb * 80 + q * 16 + i16;
                if (k < kStemK) ps[co * kStemK + k] = acc[a][q][i];
            }
        }
}

// host side
static bool stem_geometry(int H, int W, int* Ho, int* Wo) {
    *Ho = (H + 6 - 7) / 2 + 1;
    *Wo = (W + 6 - 7) / 2 + 1;
    return H > 0 && W > 0 && (*Wo % 8) == 0 && ((*Ho * *Wo) % 32) == 0;
}

int g_stem_lds = 1;  // cot_set_tuning key 41: 1 (default) = forward / weight gradient with the input rows staged in LDS, 0 = gathers from global memory
// geometry of the LDS-staged weight gradient (0: not covered): slices of ~8 output-row pairs
static int stem_wgrad_lds_slices(int N, int H, int W, int Ho, int Wo) {
    if (!g_stem_lds || W % 8 != 0 || (2 * Wo) % 32 != 0 || (Ho & 1) || 27 * ((W + 16) / 8) > 4 * 256) return 0;
    const int64_t T = (int64_t)N * (Ho / 2);
    int64_t S = T / 8;
    if (S > 1024) S = 1024;
    return (int)(S < 1 ? 1 : S);
}
int stem7x7_splits(int N, int H, int W) {
    int Ho, Wo;
    if (!stem_geometry(H, W, &Ho, &Wo)) return 0;
    if (const int sl = stem_wgrad_lds_slices(N, H, W, Ho, Wo)) return sl;
    const int64_t T = (int64_t)N * (Ho * Wo / 32);
    int64_t S = 512;
    if (S > T / 16) S = T / 16;
    return (int)(S < 1 ? 1 : S);
}

int stem7x7_forward(const void* x, const void* w, void* y, int N, int H, int W, hipStream_t stream) {
    int Ho, Wo;
    if (!stem_geometry(H, W, &Ho, &Wo)) return COT_ERR_UNSUPPORTED;
    if (g_stem_lds && W % 8 == 0 && Wo % 4 == 0 && (Ho * Wo) % 4 == 0) {
        // patch rows: a tile of 256 pixels touches at most 255 / Wo + 2 output rows
        const int rows = 255 / Wo + 2, NR = 2 * (rows - 1) + 7, RS = W + 16;
        const size_t lds = ((size_t)kStemCo * 168 + kStemKp + (size_t)3 * NR * RS) * 2;
        if (lds <= 64 * 1024 && (3 * NR + 6) * RS + 8 < 65536) {
            const int tiles = ceil_div(Ho * Wo, 256);
            COT_LAUNCH(stem7x7_fwd_lds, dim3((unsigned)((int64_t)N * tiles)), dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)w,
                       (bf16_t*)y, H, W, Ho, Wo, tiles, NR, RS);
            return check_launch("stem7x7_fwd_lds");
        }
    }
    const int tpi = ceil_div(Ho * Wo, 64);
    const int64_t waves = (int64_t)N * tpi;
    COT_LAUNCH(stem7x7_fwd_mfma, dim3((unsigned)ceil_div64(waves, 4)), dim3(256), kStemCo * kStemKp * 2, stream,
               (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, H, W, Ho, Wo, tpi, waves);
    return check_launch("stem7x7_fwd_mfma");
}

int stem7x7_wgrad(const void* gy, const void* x, void* gw, float* workspace, int N, int H, int W, hipStream_t stream) {
    int Ho, Wo;
    if (!stem_geometry(H, W, &Ho, &Wo)) return COT_ERR_UNSUPPORTED;
    if (const int SL = stem_wgrad_lds_slices(N, H, W, Ho, Wo)) {
        const int RS = W + 16;
        COT_LAUNCH(stem7x7_wgrad_lds, dim3((unsigned)SL), dim3(256), (size_t)2 * 27 * RS * 2, stream, (const bf16_t*)gy, (const bf16_t*)x,
                   workspace, N, H, W, Ho, Wo, SL, RS);
        int rc = check_launch("stem7x7_wgrad_lds");
        if (rc) return rc;
        return conv1x1_wgrad_reduce_launch(workspace, SL, kStemCo, kStemK, 0, gw, nullptr, stream);
    }
    const int S = stem7x7_splits(N, H, W), spi = Ho * Wo / 32;
    const int64_t waves = (int64_t)S * 2;
    COT_LAUNCH(stem7x7_wgrad_mfma, dim3((unsigned)ceil_div64(waves, 4)), dim3(256), 0, stream, (const bf16_t*)gy,
               (const bf16_t*)x, workspace, N, H, W, Ho, Wo, S, spi, waves);
    int rc = check_launch("stem7x7_wgrad_mfma");
    if (rc) return rc;
    return conv1x1_wgrad_reduce_launch(workspace, S, kStemCo, kStemK, 0, gw, nullptr, stream);
}

}  // namespace cot
