// conv1x1.hip -- 1x1 convolutions of the CoT block on NCHW bf16 tensors, as MFMA GEMMs WITHOUT layout changes.
//
// Replaces nn.Conv2d(kernel_size=1) forward / data-gradient / weight-gradient for the reference's
//   CotLayer.embed[0], embed[3] (models/cotnet.py:51-57), CotLayer.conv1x1[0] (:59-62),
//   Bottleneck.conv1 / conv3 / downsample conv (models/cotnet.py:206-224, models/resnet.py:383-401).
// Round-1 profile: MIOpen serves these through NHWC implicit-GEMM kernels bracketed by NCHW<->NHWC transposes and
// cast/zero kernels (batched_transpose_* + SubTensorOp* = 8 ms of a 36 ms step).  In NCHW the op is, per image n,
//       Y[n] (M x HW) = A (M x K) * X[n] (K x HW)            A = weight [Co][Ci]  (forward)
//                                                            A = weight^T         (data gradient, X = dY)
//       dW (M x J)   = sum_n dY[n] (M x HW) * X[n]^T (HW x J)                     (weight gradient)
// All three are HBM-bound on MI355X (arithmetic intensity <= Ci/2 flop/byte, far left of the MFMA ridge), so the
// kernels are organised around streaming X / Y exactly once with 16-byte accesses; MFMA does the arithmetic because
// it is free at this intensity, not because the op is compute-bound.
//
// MFMA operand maps used (v_mfma_f32_16x16x32_bf16, D = A*B + C, one wave):
//       A: lane l holds A[i = l&15][k = 8*(l>>4) .. +7]          (8 bf16, K-contiguous)
//       B: lane l holds B[k = 8*(l>>4) .. +7][j = l&15]
//     C/D: lane l holds D[i = 4*(l>>4) + r][j = l&15], r = 0..3
// Forward / data gradient: the K index of B = X is the STRIDED one in NCHW (channels are HW apart).  Each lane loads,
// for its 8 channels, PXV consecutive pixels (one 16- or 8-byte access per channel), and transposes that 8 x PXV block
// inside its own registers: MFMA number c takes pixel c of every lane's piece as column j.  Column j of MFMA c is
// therefore pixel 16*PXV*tile + PXV*j + c, so in the C/D map a lane ends up with PXV CONSECUTIVE pixels of 4 output
// channels -> one wide store per channel row.  No LDS, no cross-lane traffic, no barriers.
// Weight gradient: the reduction index is the pixel index, contiguous for both operands -> fragments are plain
// 16-byte loads.  Split over the (n, pixel) range into S deterministic partial sums (fp32 workspace) + one reduce
// kernel (no atomics: bit-reproducible, and capturable in a HIP graph without a zero-fill node).
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

// channel row `r` of image n in a tensor given as two channel slabs (concatenation along C): rows [0,c1) live in
// t1 (c1 channels per image), rows [c1,C) in t2 (C-c1 channels per image)
template <typename P> __device__ __forceinline__ P* row_ptr(P* t1, P* t2, int c1, int C, int n, int r, int HW) {
    return r < c1 ? t1 + ((int64_t)n * c1 + r) * HW : t2 + ((int64_t)n * (C - c1) + (r - c1)) * HW;
}

// ------------------------------------------------------------------------------------------------------------------
// Y[n][m][p] = sum_k A[m][k] * X[n][k][p] (+ bias[m]);  A row-major [M][K], K % 8 == 0.
// One wave = (image n, tile of 16*PXV pixels, block of 16*MT output channels); 4 waves per workgroup.
// TA: A is given TRANSPOSED (element (m, k) at A[k*M + m]) -- the data gradient reads the weight tensor as it is stored,
// with eight 2-byte loads per fragment (weights are L2-resident) instead of a transposed copy made by an extra launch.
template <int PXV, int MT, int AL, bool TA, int D>
__global__ void __launch_bounds__(256, 2)
conv1x1_fwd_mfma(const bf16_t* __restrict__ x1, const bf16_t* __restrict__ x2, int k1, const bf16_t* __restrict__ A,
                 const bf16_t* __restrict__ bias, bf16_t* __restrict__ y1, bf16_t* __restrict__ y2, int m1, int N, int K,
                 int M, int HW, int mblocks, int ptiles, int64_t total_waves, int xcd_remap, int accumulate) {
    const int64_t wid = wave_work_id(xcd_remap);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int mb = uniform((int)(wid % mblocks));
    const int64_t t = wid / mblocks;
    const int pt = uniform((int)(t % ptiles)), n = uniform((int)(t / ptiles));
    const int P0 = pt * (16 * PXV), p0 = P0 + j * PXV;  // this lane's first pixel
    const int cnt = HW - p0;                            // valid pixels from p0 on (<= 0: nothing to store)
    const int mbase = mb * (16 * MT);

    f32x4_t acc[MT][PXV];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < PXV; ++c) acc[mt][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const bf16_t* arow[MT];  // A rows of this lane (clamped: rows >= M are computed on a copy of row M-1, never stored)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mrow = min(mbase + mt * 16 + j, M - 1);
        arow[mt] = TA ? A + mrow : A + (int64_t)mrow * K;
    }

    // K loop with a D-stage register ring: the loads of steps k+1 .. k+D-1 are in flight while step k is multiplied.
    // Deep-K / small-image layers (K up to 2048 at 7x7, one or two waves per CU) are chains of dependent load latencies
    // otherwise; the big-image layers have enough waves per SIMD and use D = 1 to keep their 128 accumulators resident.
    // (pieces stay packed from the load to the multiply: mfma_common.h "pieces")
    uint32_t raw[D][8][PXV / 2];
    bf16x8_t af[D][MT];
    // Pixels past a row's end are loaded like any others (they are the next channel's data and only ever reach output
    // columns that are not stored); what must hold is that the wide loads stay inside their slab -- false only for the
    // last image's last channels in a partial pixel tile.  The K steps before the first such channel (all of them, for
    // every wave but a handful per launch) run the loads-only loop, the rest a second copy of the loop that tests every
    // stage (see ring_loop on why that matters).
    const int nsteps = ceil_div(K, 32);
    const bool wave_safe = ((int64_t)n * k1 + k1 - 1) * HW + P0 + 16 * PXV <= (int64_t)N * k1 * HW &&
                           (K == k1 || ((int64_t)n * (K - k1) + (K - k1) - 1) * HW + P0 + 16 * PXV <= (int64_t)N * (K - k1) * HW);
    int steps_safe = nsteps;
    if (!wave_safe) {
        steps_safe = 0;
        const int64_t over = (int64_t)P0 + 16 * PXV - HW;  // pixels this wave's tile reaches past a row's end
        if (n == N - 1 && over > 0) {  // the last ceil(over / HW) rows of a slab are the ones that can run out; slab 1 is first in K
            const int64_t bad = (over + HW - 1) / HW;
            steps_safe = (int)((k1 > bad ? k1 - bad : 0) / 32);
        }
    }
    int sbase = 0;
    auto load_stage = [&](auto safe, int d, int step) __attribute__((always_inline)) {
        const int k0 = 32 * (sbase + step), kb = k0 + 8 * g;
        const bool kok = kb < K;           // K % 8 == 0: a lane group's 8 channels are all inside or all outside
        const bool full_k = k0 + 32 <= K;  // wave-uniform: all four lane groups inside
        bool wide = true;
        if (!decltype(safe)::value) {  // wave-uniform test of this stage's rows
            const int rend = min(k0 + 32, K);
            if (k0 < k1) wide = ((int64_t)n * k1 + min(rend, k1) - 1) * HW + P0 + 16 * PXV <= (int64_t)N * k1 * HW;
            if (rend > k1)
                wide = wide && ((int64_t)n * (K - k1) + (rend - k1) - 1) * HW + P0 + 16 * PXV <= (int64_t)N * (K - k1) * HW;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            load_packed<PXV, AL>(raw[d][r], kok ? row_ptr(x1, x2, k1, K, n, kb + r, HW) + p0 : x1, kok ? cnt : 0, wide);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (TA) {
#pragma unroll
                for (int e = 0; e < 8; ++e) af[d][mt][e] = arow[mt][((kok ? kb : 0) + e) * M];  // (weights: < 2^31 elements)
            } else {
                __builtin_memcpy(&af[d][mt], __builtin_assume_aligned(arow[mt] + (kok ? kb : 0), 16), 16);
            }
        }
        if (!full_k && !kok) {  // partial last K step (scalar branch): lane groups past K contribute exact zeros
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < PXV / 2; ++c) raw[d][r][c] = 0u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 8; ++e) af[d][mt][e] = (bf16_t)0.0f;
        }
    };
    auto multiply_stage = [&](int d) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < PXV; ++c) {
            uint32_t bq[4];  // in-register transposition: pixel c of each of this lane's 8 channels
#pragma unroll
            for (int h = 0; h < 4; ++h) bq[h] = packed_pair(raw[d][2 * h], raw[d][2 * h + 1], c);
            const bf16x8_t bfrag = packed_as_frag(bq);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][c] = COT_MFMA_16X16X32_BF16(af[d][mt], bfrag, acc[mt][c]);
        }
    };
    ring_loop<D>(steps_safe, [&](int d, int step) __attribute__((always_inline)) { load_stage(std::true_type{}, d, step); },
                 multiply_stage);
    sbase = steps_safe;
    ring_loop<D>(nsteps - steps_safe, [&](int d, int step) __attribute__((always_inline)) { load_stage(std::false_type{}, d, step); },
                 multiply_stage);

    if (cnt <= 0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // y += result (accumulate bit 0: first slab, bit 1: second slab): the four rows' previous values are fetched
        // together, not one round trip per row
        uint32_t prev[4][PXV / 2];
        if (accumulate) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mbase + mt * 16 + g * 4 + i;
                if (m < M && ((accumulate >> (m < m1 ? 0 : 1)) & 1))
                    load_packed_lane<PXV, AL>(prev[i], row_ptr(y1, y2, m1, M, n, m, HW) + p0, cnt);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + mt * 16 + g * 4 + i;
            if (m < M) {
                const float b = bias ? (float)bias[m] : 0.f;
                bf16_t* dst = row_ptr(y1, y2, m1, M, n, m, HW) + p0;
                bf16_t o[PXV];
                if ((accumulate >> (m < m1 ? 0 : 1)) & 1) {
#pragma unroll
                    for (int c = 0; c < PXV; ++c) o[c] = (bf16_t)(acc[mt][c][i] + b + packed_get(prev[i], c));
                } else {
#pragma unroll
                    for (int c = 0; c < PXV; ++c) o[c] = (bf16_t)(acc[mt][c][i] + b);
                }
                store_piece<PXV, AL>(dst, o, cnt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// part[s][m][jj] = sum over the s-th slice of (n, pixel) of dY[n][m][p] * X[n][jj][p];  column jj == J (when has_bias)
// multiplies by 1 -> the bias gradient.  One wave = (64 x 64 output tile, slice s); 4 waves per workgroup.
// S == 1 (small problems): there is nothing to reduce -- the single slice is written straight to gw / gb (bf16).
template <int AL>
__global__ void __launch_bounds__(256, 2)
conv1x1_wgrad_mfma(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x1, const bf16_t* __restrict__ x2, int k1,
                   float* __restrict__ part, bf16_t* __restrict__ gw, bf16_t* __restrict__ gb, int N, int M, int J, int HW,
                   int has_bias, int mblocks, int jblocks, int S, int spi, int64_t total_waves, int xcd_remap) {
    // one WORKGROUP per (64 x 64 tile, slice): its four waves take a quarter of the slice's steps each and are summed through
    // LDS before the slice's partial sum is written -- four times fewer partial sums in memory than one wave per slice
    unsigned wgb = blockIdx.x;
    if (xcd_remap && (gridDim.x & 7u) == 0) wgb = (wgb & 7u) * (gridDim.x >> 3) + (wgb >> 3);
    const int64_t wid = wgb;
    if (wid >= total_waves) return;  // (whole workgroups: `total_waves` counts workgroups here)
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    const int jb = uniform((int)(wid % jblocks));
    const int64_t u = wid / jblocks;
    const int mb = uniform((int)(u % mblocks)), s = uniform((int)(u / mblocks));
    const int Jp = J + (has_bias ? 1 : 0);
    const int T = N * spi;  // (reduction steps: far below 2^31)
    const int ts0 = (int)((int64_t)T * s / S), ts1 = (int)((int64_t)T * (s + 1) / S);
    const int t0 = ts0 + (int)((int64_t)(ts1 - ts0) * wave / 4), t1 = ts0 + (int)((int64_t)(ts1 - ts0) * (wave + 1) / 4);

    int mrow[4], jrow[4];
    bool ones[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mrow[q] = min(mb * 64 + q * 16 + i16, M - 1);
        const int jj = jb * 64 + q * 16 + i16;
        ones[q] = has_bias && jj == J;
        jrow[q] = min(jj, J - 1);
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // reduction loop with a 3-stage register ring: two steps' loads are in flight while one is multiplied (a slice is a
    // chain of up to a few hundred dependent steps; without the ring every step pays a full memory latency)
    // Fragments stay packed and untouched from the load to the multiply (mfma_common.h "pieces"): the tail of a row is
    // cleared, and the column of ones for the bias gradient put in, at multiply time.
    constexpr int DW = 3;
    uint32_t aq[DW][4][4], bq[DW][4][4];
    int pleft[DW];   // valid pixels of this lane's pieces in the stage
    bool tails[DW];  // wave-uniform: the stage runs over the row's end
    const bool any_ones = has_bias && jb * 64 <= J && J < jb * 64 + 64;  // wave-uniform: the bias column is in this tile
    // every lane reads 8 elements at (row, st*32 + 8g): inside the tensor for ALL rows of image nn?  (the rows of later
    // images follow in memory; only the last image can run out) -- per tensor, by its channel count.  The steps of all
    // images but the last are covered by one test and run the loads-only loop; the steps of the last image, if the
    // slice has any, test each step in a second copy of the loop (see ring_loop).
    const int jmin = x2 ? min(k1, J - k1) : J;
    const int64_t over_max = (int64_t)spi * 32 - HW;  // pixels the last step of a row reads past the row's end
    int t_split = t1;                                  // steps [t0, t_split) are safe, [t_split, t1) are tested
    if (over_max > 0) t_split = (over_max <= (int64_t)HW * M && over_max <= (int64_t)HW * jmin) ? (N - 1) * spi : 0;
    t_split = max(t0, min(t1, t_split));
    int tbase = t0;
    auto load_stage = [&](auto safe, int d, int step) __attribute__((always_inline)) {
        const int t = tbase + step, nn = t / spi, st = t - nn * spi;
        const int p = st * 32 + g * 8;
        const int cnt = HW - p;
        const bool tail = (st + 1) * 32 > HW;  // wave-uniform: this step runs over the row's end
        bool wide = true;
        if (!decltype(safe)::value) {
            const int64_t over = (int64_t)st * 32 + 32 - HW, left = (int64_t)(N - 1 - nn) * HW;
            wide = over <= left * M && over <= left * jmin;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            load_packed<8, AL>(aq[d][q], gy + ((int64_t)nn * M + mrow[q]) * HW + p, cnt, wide);
            load_packed<8, AL>(bq[d][q], row_ptr(x1, x2, k1, J, nn, jrow[q], HW) + p, cnt, wide);
        }
        pleft[d] = cnt;
        tails[d] = tail;
    };
    auto multiply_stage = [&](int d) __attribute__((always_inline)) {
        if (any_ones) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) bq[d][q][i] = ones[q] ? 0x3f803f80u : bq[d][q][i];  // bf16 1.0 twice
        }
        if (tails[d]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mask_packed<8>(aq[d][q], pleft[d]);
                mask_packed<8>(bq[d][q], pleft[d]);
            }
        }
        bf16x8_t bfr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bfr[b] = packed_as_frag(bq[d][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bf16x8_t af = packed_as_frag(aq[d][a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = COT_MFMA_16X16X32_BF16(af, bfr[b], acc[a][b]);
        }
    };
    ring_loop<DW>(t_split - t0, [&](int d, int step) __attribute__((always_inline)) { load_stage(std::true_type{}, d, step); },
                  multiply_stage);
    tbase = t_split;
    ring_loop<DW>(t1 - t_split, [&](int d, int step) __attribute__((always_inline)) { load_stage(std::false_type{}, d, step); },
                  multiply_stage);

    // sum the four waves' accumulators: waves 2,3 -> LDS -> waves 0,1; wave 1 -> LDS -> wave 0 (lane-linear images: no conflicts)
    {
        extern __shared__ __attribute__((aligned(16))) char cot_smem[];
        float* red = reinterpret_cast<float*>(cot_smem);  // 2 x 4096 floats
        if (wave >= 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[(wave - 2) * 4096 + ((a * 4 + b) * 4 + i) * 64 + lane] = acc[a][b][i];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[a][b][i] += red[wave * 4096 + ((a * 4 + b) * 4 + i) * 64 + lane];
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[((a * 4 + b) * 4 + i) * 64 + lane] = acc[a][b][i];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[a][b][i] += red[((a * 4 + b) * 4 + i) * 64 + lane];
    }
    float* ps = part + (int64_t)s * M * Jp;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb * 64 + a * 16 + g * 4 + i;
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int jj = jb * 64 + b * 16 + i16;
                if (S > 1) {
                    if (jj < Jp) ps[(int64_t)m * Jp + jj] = acc[a][b][i];
                } else if (jj < J) {
                    gw[(int64_t)m * J + jj] = (bf16_t)acc[a][b][i];
                } else if (jj == J && has_bias) {
                    gb[m] = (bf16_t)acc[a][b][i];
                }
            }
        }
}

// gw[m][j] = bf16(sum_s part[s][m][j]);  gb[m] = bf16(sum_s part[s][m][J]).  A workgroup owns 32 consecutive outputs; its 8
// thread groups of 32 take the slices s = q, q+8, .. (coalesced 128-byte reads, four loads in flight per thread) and are
// summed through LDS in a fixed order (deterministic).  (The first version walked all S slices in one thread: 128 us for the
// 64 x 64 layers, whose S is in the hundreds.)
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce(const float* __restrict__ part, int S, int M, int J, int has_bias,
                                                           bf16_t* __restrict__ gw, bf16_t* __restrict__ gb) {
    __shared__ float red[8][32];
    const int Jp = J + (has_bias ? 1 : 0);
    const int64_t tot = (int64_t)M * Jp;
    const int el = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < tot) {
        int sl = q;
        for (; sl + 24 < S; sl += 32) {
            s0 += part[(int64_t)sl * tot + e];
            s1 += part[(int64_t)(sl + 8) * tot + e];
            s2 += part[(int64_t)(sl + 16) * tot + e];
            s3 += part[(int64_t)(sl + 24) * tot + e];
        }
        for (; sl < S; sl += 8) s0 += part[(int64_t)sl * tot + e];
    }
    red[q][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q != 0 || e >= tot) return;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += red[k][el];
    const int m = (int)(e / Jp), jj = (int)(e % Jp);
    if (jj < J) {
        if (gw) gw[(int64_t)m * J + jj] = (bf16_t)sum;
    } else if (gb) {
        gb[m] = (bf16_t)sum;
    }
}

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb,
                                hipStream_t stream) {
    const int64_t tot = (int64_t)M * (J + (has_bias ? 1 : 0));
    COT_LAUNCH(conv1x1_wgrad_reduce, dim3((unsigned)ceil_div64(tot, 32)), dim3(256), 0, stream, part, S, M, J, has_bias,
               (bf16_t*)gw, (bf16_t*)gb);
    return check_launch("conv1x1_wgrad_reduce");
}

// ------------------------------------------------------------------------------------------------------------------
// host side
extern int g_conv1x1_tune[4];  // [0] xcd remap (default 1), [1] MT override (0 = auto), [2] wgrad target waves, [3] launches of at least this many waves use ring depth 1 (0 = 8192)
int g_conv1x1_tune[4] = {1, 0, 2048, 0};
int g_wgrad_cap_pct = 0;  // cot_set_tuning key 19: partial-sum bytes of a weight gradient as a percentage of its input bytes (0 = the kernels' defaults)

template <int PXV, int AL, bool TA>
static int launch_fwd_mt(const bf16_t* x1, const bf16_t* x2, int k1, const bf16_t* A, const bf16_t* bias, bf16_t* y1,
                         bf16_t* y2, int m1, int N, int K, int M, int HW, int accumulate, hipStream_t stream) {
    const int ptiles = ceil_div(HW, 16 * PXV);
    // rows per wave: 64 when that still gives the chip several waves per SIMD (the X tile is then re-read by fewer
    // waves), 32 for small outputs and small problems -- more waves, and room for a deeper prefetch ring
    int MT = g_conv1x1_tune[1];
    if (MT != 2 && MT != 4) MT = (M <= 32 || (int64_t)N * ptiles * ceil_div(M, 64) < 2048) ? 2 : 4;
    if (TA && PXV == 8) MT = 2;  // (the strided weight gather of the data gradient does not fit next to 128 accumulators)
    const int mblocks = ceil_div(M, 16 * MT);
    const int64_t waves = (int64_t)N * ptiles * mblocks;
    const dim3 grid(wave_grid_blocks(waves)), block(256);
    const int xcd = g_conv1x1_tune[0];
    // ring depth: whatever the register file allows next to the accumulators (PXV*MT*4 registers)
    constexpr int D2 = PXV == 8 ? 2 : 4, D4 = PXV == 8 ? 1 : 3;
    const int64_t big = g_conv1x1_tune[3] > 0 ? g_conv1x1_tune[3] : 8192;
    if (MT == 2 && waves >= big)  // big launch: occupancy hides the latency, registers stay free for more waves
        COT_LAUNCH((conv1x1_fwd_mfma<PXV, 2, AL, TA, 1>), grid, block, 0, stream, x1, x2, k1, A, bias, y1, y2, m1, N, K, M,
                   HW, mblocks, ptiles, waves, xcd, accumulate);
    else if (MT == 2)
        COT_LAUNCH((conv1x1_fwd_mfma<PXV, 2, AL, TA, D2>), grid, block, 0, stream, x1, x2, k1, A, bias, y1, y2, m1, N, K, M,
                   HW, mblocks, ptiles, waves, xcd, accumulate);
    else if constexpr (!(TA && PXV == 8))
        COT_LAUNCH((conv1x1_fwd_mfma<PXV, 4, AL, TA, D4>), grid, block, 0, stream, x1, x2, k1, A, bias, y1, y2, m1, N, K, M,
                   HW, mblocks, ptiles, waves, xcd, accumulate);
    return check_launch("conv1x1_fwd_mfma");
}

// x = [x1 | x2] along channels (x2 may be NULL, then k1 == K); y = [y1 | y2] likewise (m1 == M when y2 == NULL);
// transposed_a: A holds the [K][M] matrix (the weight tensor itself, for the data gradient)
int conv1x1_gemm(const void* x1, const void* x2, int k1, const void* A, const void* bias, void* y1, void* y2, int m1,
                 int N, int K, int M, int HW, int accumulate, int transposed_a, hipStream_t stream) {
    const bf16_t *X1 = (const bf16_t*)x1, *X2 = (const bf16_t*)x2, *a = (const bf16_t*)A, *b = (const bf16_t*)bias;
    bf16_t *Y1 = (bf16_t*)y1, *Y2 = (bf16_t*)y2;
#define COT_C1_DISPATCH(TA_)                                                                                          \
    do {                                                                                                              \
        if (HW % 8 == 0) return launch_fwd_mt<8, 16, TA_>(X1, X2, k1, a, b, Y1, Y2, m1, N, K, M, HW, accumulate, stream); \
        if (HW % 4 == 0) return launch_fwd_mt<4, 8, TA_>(X1, X2, k1, a, b, Y1, Y2, m1, N, K, M, HW, accumulate, stream);  \
        return launch_fwd_mt<4, 2, TA_>(X1, X2, k1, a, b, Y1, Y2, m1, N, K, M, HW, accumulate, stream);                   \
    } while (0)
    if (transposed_a) COT_C1_DISPATCH(true);
    COT_C1_DISPATCH(false);
#undef COT_C1_DISPATCH
}

// number of deterministic partial sums (slices, one workgroup each per 64 x 64 tile) the weight gradient is split into;
// also sizes the workspace
int conv1x1_wgrad_splits(int N, int M, int J, int HW, int has_bias) {
    const int Jp = J + (has_bias ? 1 : 0);
    const int64_t units = (int64_t)ceil_div(M, 64) * ceil_div(Jp, 64);
    const int64_t T = (int64_t)N * ceil_div(HW, 32);
    if (g_conv1x1_tune[2] < 0) return (int)(-g_conv1x1_tune[2] < T ? -g_conv1x1_tune[2] : T);  // forced split (tests)
    // ~4 workgroups per CU ...
    int64_t S = ceil_div64(g_conv1x1_tune[2] > 0 ? g_conv1x1_tune[2] / 2 : 1024, units);
    // ... as long as the partial sums (written once, read once) stay below half of the input bytes (A/B on the MI355X,
    // whole step: 25 % 21.72 ms, 50 % 21.07, 100 % 21.15, 200 % 21.15 -- profiles/r02_wgrad_split_ab.txt) ...
    const int64_t in_bytes = (int64_t)N * HW * (M + J) * 2, out_bytes = (int64_t)M * Jp * 4;
    const int64_t cap = in_bytes * (g_wgrad_cap_pct > 0 ? g_wgrad_cap_pct : 50) / 100 / out_bytes;
    if (S > cap) S = cap;
    if (S > T / 8) S = T / 8;  // ... and every wave of a slice has at least two reduction steps
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    return (int)S;
}

int conv1x1_wgrad(const void* gy, const void* x1, const void* x2, int k1, void* gw, void* gb, float* workspace, int N,
                  int J, int M, int HW, hipStream_t stream) {
    const int has_bias = gb ? 1 : 0, Jp = J + has_bias;
    const int S = conv1x1_wgrad_splits(N, M, J, HW, has_bias);
    const int mblocks = ceil_div(M, 64), jblocks = ceil_div(Jp, 64), spi = ceil_div(HW, 32);
    const int64_t waves = (int64_t)S * mblocks * jblocks;  // workgroups: one per (tile, slice)
    const dim3 grid((unsigned)((waves + 7) / 8 * 8)), block(256);
    const int xcd = g_conv1x1_tune[0];
    const bf16_t *GY = (const bf16_t*)gy, *X1 = (const bf16_t*)x1, *X2 = (const bf16_t*)x2;
    constexpr size_t red_lds = 2 * 4096 * sizeof(float);
    if (HW % 8 == 0)
        COT_LAUNCH((conv1x1_wgrad_mfma<16>), grid, block, red_lds, stream, GY, X1, X2, k1, workspace, (bf16_t*)gw, (bf16_t*)gb, N,
                   M, J, HW, has_bias, mblocks, jblocks, S, spi, waves, xcd);
    else
        COT_LAUNCH((conv1x1_wgrad_mfma<2>), grid, block, red_lds, stream, GY, X1, X2, k1, workspace, (bf16_t*)gw, (bf16_t*)gb, N,
                   M, J, HW, has_bias, mblocks, jblocks, S, spi, waves, xcd);
    int rc = check_launch("conv1x1_wgrad_mfma");
    if (rc || S == 1) return rc;
    return conv1x1_wgrad_reduce_launch(workspace, S, M, J, has_bias, gw, gb, stream);
}

}  // namespace cot
