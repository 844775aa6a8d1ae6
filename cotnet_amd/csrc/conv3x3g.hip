// conv3x3g.hip -- grouped 3x3 convolution (stride 1, padding 1) on NCHW bf16 tensors as MFMA implicit GEMMs.
//
// Replaces the reference's CotLayer.key_embed[0] = nn.Conv2d(dim, dim, 3, padding=1, groups=4, bias=False)
// (models/cotnet.py:43-47; groups=8 in CoXtLayer, :112-116): forward, data gradient and weight gradient.
// Per group (Kc input channels -> Mg output channels), per image:
//       Y (Mg x HW) = A (Mg x 9*Kc) * B (9*Kc x HW),     B[(tap, ci)][p] = X[ci][p + dy*W + dx] or 0 outside the image
// B is never materialised.  Same in-register transposition as conv1x1.hip (mfma_common.h has the operand maps): a lane
// loads, for its 8 channels, PXV consecutive pixels STARTING AT THE SHIFTED ADDRESS p0 + dy*W + dx -- one wide, in
// general unaligned, load per channel; pixels whose shifted neighbour falls outside the image come in as the neighbouring
// row's / channel's data and are zeroed afterwards by select (never by multiplication: the garbage may be Inf/NaN) with
// a 9-bit per-pixel tap-validity mask (`masks`, one uint16 per pixel, built once per H x W by conv3x3g_masks).
// The K order is (tap, ci): every lane group's 8 channels share one tap, so the mask is applied per MFMA column.  The
// weight fragments are gathered from the tensor as torch stores it (2-byte loads; <= 1.2 MB, L2-resident); for the
// data gradient the same kernel runs on dY with the gather transposing (co, ci) and flipping the taps.
// Weight gradient: reduction over pixels, both operands pixel-contiguous; the B fragment of column (ci, tap) is X at the
// shifted address with the tap's mask bits ANDed in; deterministic split-K partial sums + the reduce kernel of
// conv1x1.hip.  Output columns are in the weight tensor's own [ci][3][3] order, so no re-ordering on the way out.
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

int conv1x1_wgrad_reduce_launch(const float* part, int S, int M, int J, int has_bias, void* gw, void* gb,
                                hipStream_t stream);  // conv1x1.hip
extern int g_conv1x1_tune[4];
extern int g_wgrad_cap_pct;

__host__ __device__ inline int masks_padded(int HW) { return (HW + 127) / 128 * 128 + 128; }

// masks[p] bit t (t = 3*(dy+1) + (dx+1)) = 1 iff pixel p = h*W + w has an in-image neighbour (h+dy, w+dx); 0 for p >= HW
__global__ void conv3x3g_masks_kernel(uint16_t* __restrict__ masks, int H, int W, int padded) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= padded) return;
    unsigned bits = 0;
    if (p < H * W) {
        const int h = p / W, w = p - h * W;
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) bits |= 1u << t;
        }
    }
    masks[p] = (uint16_t)bits;
}

// ------------------------------------------------------------------------------------------------------------------
// Y[n][grp*Mg + m][p] = sum_{tap, ci} Wt(m, tap, ci) * X[n][grp*Kc + ci][p + off(tap)]   (zero outside the image)
// One wave = (image, pixel tile of 16*PXV, group, block of 16*MT output channels of the group).
template <int PXV, int MT, int AL, int D>
__global__ void __launch_bounds__(256, 2)
conv3x3g_fwd_mfma(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wt, bf16_t* __restrict__ y,
                  const uint16_t* __restrict__ masks, int Cin, int Cout, int G, int H, int W, int mblocks, int ptiles,
                  int64_t total_waves, int64_t x_elems, int xcd_remap, int accumulate, int dgrad) {
    const int64_t wid = wave_work_id(xcd_remap);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, j = lane & 15, lg = lane >> 4;
    const int mb = uniform((int)(wid % mblocks));
    int64_t t = wid / mblocks;
    const int grp = uniform((int)(t % G));
    t /= G;
    const int pt = uniform((int)(t % ptiles)), n = uniform((int)(t / ptiles));
    const int HW = H * W, Kc = Cin / G, Mg = Cout / G, Kg = 9 * Kc;
    const int P0 = pt * (16 * PXV), p0 = P0 + j * PXV;
    const int cnt = HW - p0;
    const int mbase = mb * (16 * MT);
    const int64_t base = ((int64_t)n * Cin + (int64_t)grp * Kc) * HW;  // element offset of the group's first channel
    // every wide load of this wave stays inside the tensor (false only for the first / last waves of the launch)
    const bool wave_safe = base + P0 - W - 1 >= 0 && base + (int64_t)(Kc - 1) * HW + P0 + 16 * PXV + W + 1 <= x_elems;

    unsigned vm[PXV];  // tap-validity bits of this lane's pixels (masks is zero beyond HW and padded past every tile)
#pragma unroll
    for (int c = 0; c < PXV; ++c) vm[c] = masks[p0 + c];

    f32x4_t acc[MT][PXV];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < PXV; ++c) acc[mt][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // The A operand is read from the weight tensor [co][ci][3][3] AS STORED (eight 2-byte loads per fragment, L2 hits):
    //   forward        row m = co, K index (tap, ci):   wt[((grp*Mg + m)*Kc + ci)*9 + tap]
    //   data gradient  row m = ci, K index (tap, co):   wt[((grp*Kc + co)*Mg + m)*9 + (8 - tap)]     (Kc, Mg as THIS
    //                  launch sees them: K side = the convolution's output channels) -- transposed, taps flipped
    int mrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) mrow[mt] = min(mbase + mt * 16 + j, Mg - 1);

    // K loop (9*Kc/32 = 4.5 .. 36 steps) with a D-stage register ring, as in conv1x1.hip: pieces stay packed from the load
    // to the multiply, and the first / last waves of the launch (whose shifted loads could leave the tensor) run their
    // own copy of the loop with bounds-checked element loads
    uint32_t raw[D][8][PXV / 2];
    bf16x8_t af[D][MT];
    int tapd[D];
    auto load_stage = [&](auto safe, int d, int step) __attribute__((always_inline)) {
        const int k0 = 32 * step, kb = k0 + 8 * lg;
        const bool kok = kb < Kg;  // Kc % 8 == 0: the 8 channels of a lane group share one tap and are all in or all out
        const int kbc = kok ? kb : 0;
        const int tap = kbc / Kc, ci0 = kbc - tap * Kc;
        const int shift = (tap / 3 - 1) * W + (tap % 3 - 1);
        tapd[d] = kok ? tap : 9;  // bit 9 of a validity mask is never set: lane groups past K contribute zeros
        bool wide = true;
        if (!decltype(safe)::value && D == 1) {
            wide = false;  // big launches (D = 1): a few slow waves among thousands; keep this copy of the loop small
        } else if (!decltype(safe)::value) {
            // one of the launch's first / last waves: does THIS step stay inside the tensor?  (wave-uniform bound over the
            // step's K range: taps are ordered by their shift, channels within a tap by address.)  Small launches are one
            // round of waves: their duration is that of the slowest wave, so these must not check every element.
            const int kend = min(k0 + 32, Kg) - 1, tlo = k0 / Kc, thi = kend / Kc;
            const int cmin = tlo == thi ? k0 - tlo * Kc : 0, cmax = tlo == thi ? kend - tlo * Kc : Kc - 1;
            const int64_t lo = base + (int64_t)cmin * HW + P0 + ((tlo / 3 - 1) * W + (tlo % 3 - 1));
            const int64_t hi = base + (int64_t)cmax * HW + P0 + 16 * PXV + ((thi / 3 - 1) * W + (thi % 3 - 1));
            wide = lo >= 0 && hi <= x_elems;
            // (lane groups past K in a partial last step read at (tap 0, channel 0), the most negative shift)
            if (k0 + 32 > Kg) wide = wide && base + P0 - W - 1 >= 0;
        }
        if (wide) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                load_packed<PXV, 2>(raw[d][r], x + (base + (int64_t)(ci0 + r) * HW + p0 + shift), PXV, true);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                load_packed_checked<PXV>(raw[d][r], x, base + (int64_t)(ci0 + r) * HW + p0 + shift, x_elems);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int wi = dgrad ? ((grp * Kc + ci0 + e) * Mg + mrow[mt]) * 9 + (8 - tap)  // (< 2^31 elements)
                                     : ((grp * Mg + mrow[mt]) * Kc + ci0 + e) * 9 + tap;
                af[d][mt][e] = wt[wi];  // (kbc keeps the index inside the tensor for lane groups past K)
            }
        }
        if (k0 + 32 > Kg && !kok) {  // partial last K step (scalar branch first): lane groups past K contribute exact zeros
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 8; ++e) af[d][mt][e] = (bf16_t)0.0f;
        }
    };
    auto multiply_stage = [&](int d) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < PXV; ++c) {
            const uint32_t keep = ((vm[c] >> tapd[d]) & 1u) ? 0xffffffffu : 0u;  // tap outside the image: exact zeros
            uint32_t bq[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) bq[h] = packed_pair(raw[d][2 * h], raw[d][2 * h + 1], c) & keep;
            const bf16x8_t bfrag = packed_as_frag(bq);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][c] = COT_MFMA_16X16X32_BF16(af[d][mt], bfrag, acc[mt][c]);
        }
    };
    if (wave_safe)
        ring_loop<D>(ceil_div(Kg, 32), [&](int d, int step) __attribute__((always_inline)) { load_stage(std::true_type{}, d, step); }, multiply_stage);
    else
        ring_loop<D>(ceil_div(Kg, 32), [&](int d, int step) __attribute__((always_inline)) { load_stage(std::false_type{}, d, step); }, multiply_stage);

    if (cnt <= 0) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        uint32_t prev[4][PXV / 2];  // y += result: the four rows' previous values are fetched together
        if (accumulate) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mbase + mt * 16 + lg * 4 + i;
                if (m < Mg) load_packed_lane<PXV, AL>(prev[i], y + ((int64_t)n * Cout + (int64_t)grp * Mg + m) * HW + p0, cnt);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + mt * 16 + lg * 4 + i;
            if (m < Mg) {
                bf16_t* dst = y + ((int64_t)n * Cout + (int64_t)grp * Mg + m) * HW + p0;
                bf16_t o[PXV];
                if (accumulate) {
#pragma unroll
                    for (int c = 0; c < PXV; ++c) o[c] = (bf16_t)(acc[mt][c][i] + packed_get(prev[i], c));
                } else {
#pragma unroll
                    for (int c = 0; c < PXV; ++c) o[c] = (bf16_t)acc[mt][c][i];
                }
                store_piece<PXV, AL>(dst, o, cnt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// part[s][grp*Mg + m][ci*9 + tap] = sum over slice s of (n, p) of dY[n][grp*Mg + m][p] * X[n][grp*Kc + ci][p + off(tap)]
// One wave = (group, 16*MTW output channels, 64 of the group's 9*Kc weight columns, slice s).
template <int MTW, int AL>
__global__ void __launch_bounds__(256, 2)
conv3x3g_wgrad_mfma(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x, float* __restrict__ part,
                    bf16_t* __restrict__ gw, const uint16_t* __restrict__ masks, int N, int Cin, int Cout, int G, int H,
                    int W, int mblocks,
                    int jblocks, int S, int spi, int64_t total_waves, int64_t x_elems, int xcd_remap) {
    const int64_t wid = wave_work_id(xcd_remap);
    if (wid >= total_waves) return;
    const int lane = threadIdx.x & 63, i16 = lane & 15, lg = lane >> 4;
    const int jb = uniform((int)(wid % jblocks));
    int64_t u = wid / jblocks;
    const int mb = uniform((int)(u % mblocks));
    u /= mblocks;
    const int grp = uniform((int)(u % G)), s = uniform((int)(u / G));
    const int HW = H * W, Kc = Cin / G, Mg = Cout / G, Jg = 9 * Kc;
    const int T = N * spi;  // (reduction steps: far below 2^31)
    const int t0 = (int)((int64_t)T * s / S), t1 = (int)((int64_t)T * (s + 1) / S);

    int mrow[MTW], jch[4], jtap[4], jshift[4];
#pragma unroll
    for (int q = 0; q < MTW; ++q) mrow[q] = grp * Mg + min(mb * (16 * MTW) + q * 16 + i16, Mg - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int jj = min(jb * 64 + q * 16 + i16, Jg - 1);  // columns >= Jg are computed on a copy, never stored
        jch[q] = grp * Kc + jj / 9;
        jtap[q] = jj % 9;
        jshift[q] = (jtap[q] / 3 - 1) * W + (jtap[q] % 3 - 1);
    }
    f32x4_t acc[MTW][4];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // reduction loop with a 3-stage register ring (see conv1x1_wgrad_mfma).  Fragments stay packed and untouched from the
    // load to the multiply: the tap-validity masks (which also clear pixels past the row's end: they are zero there) and
    // the clearing of dY's tail are applied at multiply time.  Waves whose loads could leave a tensor -- the first and
    // last of the launch -- run their own copy of the loop with checked loads.
    constexpr int DW = MTW == 4 ? 2 : 3;  // (what fits next to 16*MTW accumulators)
    uint32_t aq[DW][MTW][4], bq[DW][4][4], vmd[DW][4];  // vmd: validity bits of the lane's 8 pixels, two per register
    int pleft[DW];    // valid pixels of this lane's pieces in the stage
    bool tails[DW];   // wave-uniform: the stage runs over the row's end
    // The slice is cut at image boundaries into [t0, t_lo) (first image), [t_lo, t_hi) and [t_hi, t1) (last image): one
    // test covers the middle part, which runs the loads-only loop; the outer parts, if the slice has any, test each step.
    int t_lo = max(t0, min(t1, spi)), t_hi = max(t_lo, min(t1, (N - 1) * spi));
    if (t_hi > t_lo) {
        const int n_first = t_lo / spi, n_last = (t_hi - 1) / spi;
        const bool mid_safe =
            ((int64_t)n_first * Cin + (int64_t)grp * Kc) * HW - W - 1 >= 0 &&
            ((int64_t)n_last * Cin + (int64_t)grp * Kc + Kc - 1) * HW + (int64_t)spi * 32 + W + 1 <= x_elems &&
            (int64_t)spi * 32 - HW <= (int64_t)(N - 1 - n_last) * Cout * HW;
        if (!mid_safe) t_lo = t_hi = t1;  // (tiny tensors: everything is tested)
    }
    int tbase = t0;
    auto load_stage = [&](auto safe, int d, int step) __attribute__((always_inline)) {
        const int t = tbase + step, n = t / spi, st = t - n * spi;
        const int P = st * 32, p = P + lg * 8;
        const int cnt = HW - p;
        __builtin_memcpy(vmd[d], __builtin_assume_aligned(masks + p, 16), 16);
        if (decltype(safe)::value) {
#pragma unroll
            for (int q = 0; q < MTW; ++q) load_packed<8, AL>(aq[d][q], gy + ((int64_t)n * Cout + mrow[q]) * HW + p, cnt, true);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                load_packed<8, 2>(bq[d][q], x + (((int64_t)n * Cin + jch[q]) * HW + p + jshift[q]), 8, true);
        } else {
            // every lane reads 8 elements of dY at (row, P + 8*lg): inside the tensor for all rows of image n?
            const bool wide_a = (int64_t)P + 32 - HW <= (int64_t)(N - 1 - n) * Cout * HW;
#pragma unroll
            for (int q = 0; q < MTW; ++q)
                load_packed<8, AL>(aq[d][q], gy + ((int64_t)n * Cout + mrow[q]) * HW + p, cnt, wide_a);
            // x: only the steps at the very start of the first image / end of the last one need checked loads (a slice
            // of this wave merely CONTAINS such a step; the others must not pay eight element loads per piece)
            const int64_t base = ((int64_t)n * Cin + (int64_t)grp * Kc) * HW;
            const bool step_safe = base + P - W - 1 >= 0 && base + (int64_t)(Kc - 1) * HW + P + 32 + W + 1 <= x_elems;
            if (step_safe) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    load_packed<8, 2>(bq[d][q], x + (((int64_t)n * Cin + jch[q]) * HW + p + jshift[q]), 8, true);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    load_packed_checked<8>(bq[d][q], x, ((int64_t)n * Cin + jch[q]) * HW + p + jshift[q], x_elems);
            }
        }
        pleft[d] = cnt;
        tails[d] = P + 32 > HW;
    };
    auto multiply_stage = [&](int d) __attribute__((always_inline)) {
        if (tails[d]) {
#pragma unroll
            for (int q = 0; q < MTW; ++q) mask_packed<8>(aq[d][q], pleft[d]);
        }
        bf16x8_t bfr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bq[d][b][i] &= ((vmd[d][i] >> jtap[b]) & 0x00010001u) * 0xffffu;
            bfr[b] = packed_as_frag(bq[d][b]);
        }
#pragma unroll
        for (int a = 0; a < MTW; ++a) {
            const bf16x8_t af = packed_as_frag(aq[d][a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = COT_MFMA_16X16X32_BF16(af, bfr[b], acc[a][b]);
        }
    };
    auto tested = [&](int d, int step) __attribute__((always_inline)) { load_stage(std::false_type{}, d, step); };
    ring_loop<DW>(t_lo - t0, tested, multiply_stage);
    tbase = t_lo;
    ring_loop<DW>(t_hi - t_lo, [&](int d, int step) __attribute__((always_inline)) { load_stage(std::true_type{}, d, step); },
                  multiply_stage);
    tbase = t_hi;
    ring_loop<DW>(t1 - t_hi, tested, multiply_stage);

    float* ps = part + (int64_t)s * Cout * Jg;
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb * (16 * MTW) + a * 16 + lg * 4 + i;
            if (m >= Mg) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int jj = jb * 64 + b * 16 + i16;
                if (jj < Jg) {
                    if (S > 1) ps[((int64_t)grp * Mg + m) * Jg + jj] = acc[a][b][i];
                    else gw[((int64_t)grp * Mg + m) * Jg + jj] = (bf16_t)acc[a][b][i];  // single slice: nothing to reduce
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
int64_t conv3x3g_masks_bytes(int H, int W) { return ((int64_t)masks_padded(H * W) * 2 + 255) / 256 * 256; }

int conv3x3g_masks(void* masks, int H, int W, hipStream_t stream) {
    const int padded = masks_padded(H * W);
    COT_LAUNCH(conv3x3g_masks_kernel, dim3(ceil_div(padded, 256)), dim3(256), 0, stream, (uint16_t*)masks, H, W,
               padded);
    return check_launch("conv3x3g_masks_kernel");
}

template <int PXV, int AL>
static int launch_fwd(const bf16_t* x, const bf16_t* A, bf16_t* y, const uint16_t* masks, int N, int Cin, int Cout,
                      int G, int H, int W, int accumulate, int dgrad, hipStream_t stream) {
    const int Mg = Cout / G, HW = H * W;
    const int MT = Mg <= 16 ? 1 : (Mg <= 32 ? 2 : 4);
    const int mblocks = ceil_div(Mg, 16 * MT), ptiles = ceil_div(HW, 16 * PXV);
    const int64_t waves = (int64_t)N * ptiles * G * mblocks, x_elems = (int64_t)N * Cin * HW;
    const dim3 grid(wave_grid_blocks(waves)), block(256);
    const int xcd = g_conv1x1_tune[0];
#define COT_C3_LAUNCH(MT_, D_)                                                                                         \
    COT_LAUNCH((conv3x3g_fwd_mfma<PXV, MT_, AL, D_>), grid, block, 0, stream, x, A, y, masks, Cin, Cout, G, H, W, mblocks, \
               ptiles, waves, x_elems, xcd, accumulate, dgrad)
    // ring depth: big launches (many waves per SIMD resident) hide latency by occupancy and keep their registers for
    // that; small ones get as deep a ring as fits next to the PXV*MT*4 accumulators
    const bool big = waves >= (g_conv1x1_tune[3] > 0 ? g_conv1x1_tune[3] : 8192);  // (tuning key 14)
    if (MT == 1) {
        if (big) COT_C3_LAUNCH(1, 1);
        else COT_C3_LAUNCH(1, (PXV == 8 ? 2 : 3));
    } else if (MT == 2) {
        if (big) COT_C3_LAUNCH(2, 1);
        else COT_C3_LAUNCH(2, (PXV == 8 ? 2 : 3));
    } else {
        COT_C3_LAUNCH(4, (PXV == 8 ? 1 : 2));
    }
#undef COT_C3_LAUNCH
    return check_launch("conv3x3g_fwd_mfma");
}

// mode 0: y = conv(x, w);  mode 1: x-gradient (x := dY with Cout channels, y := dX with Cin channels)
int conv3x3g_gemm(const void* x, const void* w, void* y, const void* masks, int N, int Cin, int Cout, int G, int H, int W,
                  int mode, int accumulate, hipStream_t stream) {
    const int HW = H * W;
    const bf16_t* X = (const bf16_t*)x;
    const bf16_t* A = (const bf16_t*)w;
    bf16_t* Y = (bf16_t*)y;
    const uint16_t* mk = (const uint16_t*)masks;
    // K side / M side channel counts of the GEMM that is actually run
    const int ck = mode == 0 ? Cin : Cout, cm = mode == 0 ? Cout : Cin;
    if (HW % 8 == 0) return launch_fwd<8, 16>(X, A, Y, mk, N, ck, cm, G, H, W, accumulate, mode, stream);
    if (HW % 4 == 0) return launch_fwd<4, 8>(X, A, Y, mk, N, ck, cm, G, H, W, accumulate, mode, stream);
    return launch_fwd<4, 2>(X, A, Y, mk, N, ck, cm, G, H, W, accumulate, mode, stream);
}

int conv3x3g_wgrad_splits(int N, int Cin, int Cout, int G, int HW) {
    const int Kc = Cin / G, Mg = Cout / G, Jg = 9 * Kc;
    const int MTW = Mg <= 16 ? 1 : (Mg <= 32 ? 2 : 4);
    const int64_t units = (int64_t)G * ceil_div(Mg, 16 * MTW) * ceil_div(Jg, 64);
    const int64_t T = (int64_t)N * ceil_div(HW, 32);
    if (g_conv1x1_tune[2] < 0) return (int)(-g_conv1x1_tune[2] < T ? -g_conv1x1_tune[2] : T);  // forced split (tests)
    int64_t S = ceil_div64(g_conv1x1_tune[2] > 0 ? g_conv1x1_tune[2] : 2048, units);
    const int64_t in_bytes = (int64_t)N * HW * (Cin + Cout) * 2, out_bytes = (int64_t)Cout * Jg * 4;
    int64_t cap = in_bytes * (g_wgrad_cap_pct > 0 ? g_wgrad_cap_pct : 50) / 100 / out_bytes;  // (see conv1x1_wgrad_splits)
    if (cap < 4 && T >= 64) cap = 4;  // (a slice should not be a chain of hundreds of dependent steps)
    if (S > cap) S = cap;
    if (S > T) S = T;
    if (S < 1) S = 1;
    return (int)S;
}

template <int AL>
static int launch_wgrad(const bf16_t* gy, const bf16_t* x, float* part, bf16_t* gw, const uint16_t* masks, int N, int Cin,
                        int Cout, int G, int H, int W, int S, hipStream_t stream) {
    const int Kc = Cin / G, Mg = Cout / G, Jg = 9 * Kc, HW = H * W;
    const int MTW = Mg <= 16 ? 1 : (Mg <= 32 ? 2 : 4);
    const int mblocks = ceil_div(Mg, 16 * MTW), jblocks = ceil_div(Jg, 64), spi = ceil_div(HW, 32);
    const int64_t waves = (int64_t)S * G * mblocks * jblocks, x_elems = (int64_t)N * Cin * HW;
    const dim3 grid(wave_grid_blocks(waves)), block(256);
    const int xcd = g_conv1x1_tune[0];
#define COT_C3_LAUNCH(MTW_)                                                                                        \
    COT_LAUNCH((conv3x3g_wgrad_mfma<MTW_, AL>), grid, block, 0, stream, gy, x, part, gw, masks, N, Cin, Cout, G, H, W, \
               mblocks, jblocks, S, spi, waves, x_elems, xcd)
    if (MTW == 1) COT_C3_LAUNCH(1);
    else if (MTW == 2) COT_C3_LAUNCH(2);
    else COT_C3_LAUNCH(4);
#undef COT_C3_LAUNCH
    return check_launch("conv3x3g_wgrad_mfma");
}

int conv3x3g_wgrad(const void* gy, const void* x, void* gw, const void* masks, float* ws, int N, int Cin, int Cout, int G,
                   int H, int W, hipStream_t stream) {
    const int HW = H * W, S = conv3x3g_wgrad_splits(N, Cin, Cout, G, HW);
    int rc;
    if (HW % 8 == 0)
        rc = launch_wgrad<16>((const bf16_t*)gy, (const bf16_t*)x, ws, (bf16_t*)gw, (const uint16_t*)masks, N, Cin, Cout, G,
                              H, W, S, stream);
    else
        rc = launch_wgrad<2>((const bf16_t*)gy, (const bf16_t*)x, ws, (bf16_t*)gw, (const uint16_t*)masks, N, Cin, Cout, G,
                             H, W, S, stream);
    if (rc || S == 1) return rc;
    return conv1x1_wgrad_reduce_launch(ws, S, Cout, 9 * (Cin / G), 0, gw, nullptr, stream);
}

}  // namespace cot
