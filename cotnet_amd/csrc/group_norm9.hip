// group_norm9.hip -- GroupNorm with 9 channels per group on NCHW bf16 tensors: the normalisation of the k*k = 9 attention
// logits that share a weight channel (reference: CotLayer.embed[4] = nn.GroupNorm(dim/8, 9*dim/8), models/cotnet.py:56;
// its output, viewed [B,1,dim/8,9,H,W] (:85), is the aggregation's weight tensor).
//
// One (image, group) = 9 consecutive channels = ONE contiguous run of 9*H*W elements (56 KB at 56x56), small enough to
// sit in the registers of one workgroup.  So:
//   forward   1 read + 1 write:  load the group, mean / variance in two in-register passes (no E[x^2]-E[x]^2
//             cancellation), normalise from registers.              (torch: moments pass + apply pass = 2R + 1W)
//   backward  2 reads + 1 write: load dy and x once, per-channel sums (sum dy, sum dy*xhat) by a block reduction,
//             dx from registers; the per-(image, channel) sums go to a small workspace and a second, tiny kernel
//             reduces them over the batch into dgamma / dbeta (deterministic).       (torch: 6R + 1W, 4-5 launches)
// Thread t of a workgroup owns, for each of the 9 channels and each round r < R, the 8 consecutive pixels starting at
// 8*(r*NT + t): all channel indices are compile-time, so per-channel accumulators stay in registers.
#include "cot_common.h"
#include "mfma_common.h"

namespace cot {

__device__ __forceinline__ float gn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// block-wide sums of NV values, result broadcast to every thread; smem >= NV * 16 floats, blockDim <= 1024
template <int NV> __device__ __forceinline__ void gn_block_sum_all(float (&v)[NV], float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = gn_wave_sum(v[k]);
    __syncthreads();  // previous use of smem is over
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) smem[k * 16 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += smem[k * 16 + w];
        v[k] = s;
    }
}

// SEG < 64 (round 5): SEG lanes per (image, group), 64 / SEG groups per wave -- small planes leave most of a wave idle otherwise
// (8 pixels per lane: 7 lanes of 64 at 7 x 7, one wave per group: 5120 single-wave workgroups at B = 80; 11.1 -> 6.9 us forward,
// 24.3 -> 13.2 us backward there, profiles/r05_gn9_packed_small_planes_ab.log).
// The sums run over the SEG-lane segment (xor butterfly: every lane of the segment ends with the total).
template <int NV, int SEG> __device__ __forceinline__ void gn_seg_sum_all(float (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int o = SEG / 2; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
}

// The group sits in registers PACKED (two bf16 per register; mfma_common.h "pieces"): all loads of a thread are issued
// back to back and nothing touches them until the last one is on its way.  Rounds whose pieces run over a channel's end
// (wave-uniform `tail`) are loaded wide all the same -- what follows is the next channel; lanes wholly past the end all
// read the 16 bytes at the end -- unless that could leave the tensor (the end of the last (image, group)), and have
// their excess elements cleared afterwards.
template <int NT, int R, int AL, int SEG = 64>
__global__ void __launch_bounds__(NT)
gn9_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
               bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int G, int HW,
               float eps, int lay, int NG) {
    static_assert(SEG == 64 || (NT == 64 && R == 1 && (SEG == 32 || SEG == 16 || SEG == 8)), "packed form: one wave, one round");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];  // 16 floats per reduced value
    float* smem = reinterpret_cast<float*>(cot_smem);
    // NG = (image, group) pairs of the tensor.  SEG == 64: one workgroup each; else 64 / SEG per wave (`live`: the last wave's spare segments)
    const int t = SEG < 64 ? (int)threadIdx.x % SEG : (int)threadIdx.x;
    int gid = SEG < 64 ? (int)blockIdx.x * (64 / SEG) + (int)threadIdx.x / SEG : (int)blockIdx.x;
    const bool live = gid < NG;
    if (!live) gid = NG - 1;
    const int g = gid % G;
    const int wend = SEG < 64 ? SEG : 64 * (uniform((int)threadIdx.x >> 6) + 1);  // one past the last lane index of this wave / segment
    const int64_t total = (int64_t)NG * 9 * HW;  // elements in the tensor
    // per-tensor layout (round 5): bit 0 = x, bit 1 = y channel-major ([C][N][HW]: the group's channels are N*HW apart)
    const int nimg = NG / G, img = gid / G;
    const int64_t nchw = (int64_t)gid * 9 * HW, cmaj = ((int64_t)g * 9 * nimg + img) * HW;  // (n*G + g) * 9 channels
    const int64_t base = (lay & 1) ? cmaj : nchw, xcs = (lay & 1) ? (int64_t)nimg * HW : HW;
    const int64_t ybase = (lay & 2) ? cmaj : nchw, ycs = (lay & 2) ? (int64_t)nimg * HW : HW;
    uint32_t v[9][R][4];
    bool tail[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tail[r] = 8 * (r * NT + wend) > HW;
    if (base + 8 * xcs + HW + 8 <= total) {  // every workgroup but the tensor's last: loads only, in a straight line
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int cl = 0; cl < 9; ++cl)
                load_packed<8, AL>(v[cl][r], x + base + cl * xcs + min(8 * (r * NT + t), HW), 8, true);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = 8 * (r * NT + t);
#pragma unroll
            for (int cl = 0; cl < 9; ++cl)
                load_packed<8, AL>(v[cl][r], x + base + cl * xcs + min(p, HW), HW - p,
                                   !tail[r] || base + cl * xcs + HW + 8 <= total);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tail[r]) {
#pragma unroll
            for (int cl = 0; cl < 9; ++cl) mask_packed<8>(v[cl][r], HW - 8 * (r * NT + t));
        }
    const float inv = 1.f / (9.f * (float)HW);
    float s[1] = {0.f};
#pragma unroll
    for (int cl = 0; cl < 9; ++cl)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[0] += packed_get(v[cl][r], e);
    if (SEG < 64) gn_seg_sum_all<1, SEG>(s);
    else gn_block_sum_all<1>(s, smem);
    const float mean = s[0] * inv;
    float q[1] = {0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int left = HW - 8 * (r * NT + t);
#pragma unroll
        for (int cl = 0; cl < 9; ++cl)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = packed_get(v[cl][r], e) - mean;
                if (!tail[r] || e < left) q[0] += d * d;
            }
    }
    if (SEG < 64) gn_seg_sum_all<1, SEG>(q);
    else gn_block_sum_all<1>(q, smem);
    const float rstd = 1.f / sqrtf(q[0] * inv + eps);
    if (t == 0 && live) {
        mean_out[gid] = mean;
        rstd_out[gid] = rstd;
    }
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = (float)gamma[g * 9 + cl] * rstd, be = (float)beta[g * 9 + cl] - mean * ga;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = 8 * (r * NT + t);
            bf16_t o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(packed_get(v[cl][r], e) * ga + be);
            if (p < HW && live) store_piece<8, AL>(y + ybase + cl * ycs + p, o, HW - p);
        }
    }
}

template <int NT, int R, int AL, int SEG = 64>
__global__ void __launch_bounds__(NT)
gn9_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ mean_in,
               const float* __restrict__ rstd_in, const bf16_t* __restrict__ gamma, bf16_t* __restrict__ dx,
               float* __restrict__ part, int G, int HW, int lay, int NG) {
    static_assert(SEG == 64 || (NT == 64 && R == 1 && (SEG == 32 || SEG == 16 || SEG == 8)), "packed form: one wave, one round");
    extern __shared__ __attribute__((aligned(16))) char cot_smem[];
    float* smem = reinterpret_cast<float*>(cot_smem);
    const int t = SEG < 64 ? (int)threadIdx.x % SEG : (int)threadIdx.x;  // (see gn9_fwd_kernel)
    int gid = SEG < 64 ? (int)blockIdx.x * (64 / SEG) + (int)threadIdx.x / SEG : (int)blockIdx.x;
    const bool live = gid < NG;
    if (!live) gid = NG - 1;
    const int g = gid % G;
    const int wend = SEG < 64 ? SEG : 64 * (uniform((int)threadIdx.x >> 6) + 1);
    const int64_t total = (int64_t)NG * 9 * HW;  // elements in the tensor
    // per-tensor layout: bit 0 = dy, bit 1 = x, bit 2 = dx channel-major
    const int nimg = NG / G, img = gid / G;
    const int64_t nchw = (int64_t)gid * 9 * HW, cmaj = ((int64_t)g * 9 * nimg + img) * HW, cms = (int64_t)nimg * HW;
    const int64_t gbase = (lay & 1) ? cmaj : nchw, gcs = (lay & 1) ? cms : HW;
    const int64_t base = (lay & 2) ? cmaj : nchw, xcs = (lay & 2) ? cms : HW;
    const int64_t dbase = (lay & 4) ? cmaj : nchw, dcs = (lay & 4) ? cms : HW;
    uint32_t xv[9][R][4], gv[9][R][4];
    bool tail[R];
#pragma unroll
    for (int r = 0; r < R; ++r) tail[r] = 8 * (r * NT + wend) > HW;
    if (base + 8 * xcs + HW + 8 <= total && gbase + 8 * gcs + HW + 8 <= total) {  // every workgroup but the tensor's last: loads only
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int cl = 0; cl < 9; ++cl) {
                const int o = min(8 * (r * NT + t), HW);
                load_packed<8, AL>(xv[cl][r], x + base + cl * xcs + o, 8, true);
                load_packed<8, AL>(gv[cl][r], dy + gbase + cl * gcs + o, 8, true);
            }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = 8 * (r * NT + t);
#pragma unroll
            for (int cl = 0; cl < 9; ++cl) {
                load_packed<8, AL>(xv[cl][r], x + base + cl * xcs + min(p, HW), HW - p, !tail[r] || base + cl * xcs + HW + 8 <= total);
                load_packed<8, AL>(gv[cl][r], dy + gbase + cl * gcs + min(p, HW), HW - p, !tail[r] || gbase + cl * gcs + HW + 8 <= total);
            }
        }
    }
    const float mean = mean_in[gid], rstd = rstd_in[gid];
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (tail[r]) {  // (dy alone decides: every use of x below is multiplied into a dy term or not stored)
#pragma unroll
            for (int cl = 0; cl < 9; ++cl) {
                mask_packed<8>(xv[cl][r], HW - 8 * (r * NT + t));
                mask_packed<8>(gv[cl][r], HW - 8 * (r * NT + t));
            }
        }
    float s[18];  // per channel: sum dy, sum dy * xhat
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = packed_get(gv[cl][r], e);
                a += d;
                b += d * ((packed_get(xv[cl][r], e) - mean) * rstd);
            }
        s[2 * cl] = a;
        s[2 * cl + 1] = b;
    }
    // the fp32 forms of this phase must not be kept for the next one (they would triple the registers held across the
    // reduction): as far as the optimiser can tell, these are new values
#pragma unroll
    for (int cl = 0; cl < 9; ++cl)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                COT_KEEP_PACKED(xv[cl][r][i]);
                COT_KEEP_PACKED(gv[cl][r][i]);
            }
    if (SEG < 64) gn_seg_sum_all<18, SEG>(s);
    else gn_block_sum_all<18>(s, smem);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = (float)gamma[g * 9 + cl];
        c1 += ga * s[2 * cl];
        c2 += ga * s[2 * cl + 1];
    }
    const float inv = 1.f / (9.f * (float)HW);
    c1 *= inv;
    c2 *= inv;
    if (SEG < 64) {
#pragma unroll
        for (int k = 0; k < 18; ++k)
            if (t == k % SEG && live) part[(int64_t)gid * 18 + k] = s[k];
    } else if (t < 18) {
        part[(int64_t)gid * 18 + t] = s[t];
    }
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = (float)gamma[g * 9 + cl];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = 8 * (r * NT + t);
            bf16_t o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (packed_get(xv[cl][r], e) - mean) * rstd;
                o[e] = (bf16_t)(rstd * (ga * packed_get(gv[cl][r], e) - c1 - xh * c2));
            }
            if (p < HW && live) store_piece<8, AL>(dx + dbase + cl * dcs + p, o, HW - p);
        }
    }
}

// dgamma[c] = sum_n part[n][c].dy_xhat, dbeta[c] = sum_n part[n][c].dy      part is [N][G*9][2]
// A workgroup owns 32 channels; its 8 thread groups take the images n = q, q+8, .. (the one-thread-per-channel loop over the
// whole batch was a chain of N strided loads: 21 us for 80 images) and are summed through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void gn9_bwd_params_kernel(const float* __restrict__ part, T* __restrict__ dgamma,
                                                            T* __restrict__ dbeta, int N, int Ctot) {
    __shared__ float red[8][32][2];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
    if (c < Ctot) {
        int n = q;
        for (; n + 8 < N; n += 16) {
            const float* u = part + ((int64_t)n * Ctot + c) * 2;
            const float* v = part + ((int64_t)(n + 8) * Ctot + c) * 2;
            const float ux = u[0], uy = u[1], vx = v[0], vy = v[1];
            a0 += ux; b0 += uy; a1 += vx; b1 += vy;
        }
        for (; n < N; n += 8) {
            const float* u = part + ((int64_t)n * Ctot + c) * 2;
            a0 += u[0]; b0 += u[1];
        }
    }
    red[q][cl][0] = a0 + a1;
    red[q][cl][1] = b0 + b1;
    __syncthreads();
    if (q != 0 || c >= Ctot) return;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a += red[k][cl][0];
        b += red[k][cl][1];
    }
    dbeta[c] = (T)a;
    dgamma[c] = (T)b;
}

// ---- fp32 tensors (the reference's own precision, config.yaml `amp: False`): one workgroup per (image, group) sweeps the
// group's 9*HW contiguous floats -- 113 KB at 56x56, more than a workgroup's registers hold comfortably -- three times in
// forward (mean, centred variance, normalise) and twice in backward (sums, dx); every sweep after the first is served by
// the L2 of the XCD the workgroup runs on, so HBM sees 1 read + 1 write (forward) and 2 reads + 1 write (backward) as in
// the bf16 kernels.  Same statistics (two-pass variance), same workspace layout, same batch reduction for the parameters.
__global__ __launch_bounds__(256) void gn9f_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, int G,
                                                      int HW, float eps) {
    __shared__ float smem[16];
    const int t = threadIdx.x, g = blockIdx.x % G, n9 = 9 * HW;
    const float* xb = x + (int64_t)blockIdx.x * n9;
    float s[1] = {0.f};
    for (int i = t; i < n9; i += 256) s[0] += xb[i];
    gn_block_sum_all<1>(s, smem);
    const float mean = s[0] / (float)n9;
    float v[1] = {0.f};
    for (int i = t; i < n9; i += 256) {
        const float d = xb[i] - mean;
        v[0] += d * d;
    }
    gn_block_sum_all<1>(v, smem);
    const float rstd = 1.f / sqrtf(v[0] / (float)n9 + eps);
    if (t == 0) {
        mean_out[blockIdx.x] = mean;
        rstd_out[blockIdx.x] = rstd;
    }
    float* yb = y + (int64_t)blockIdx.x * n9;
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = gamma[g * 9 + cl], be = beta[g * 9 + cl];
        for (int p = t; p < HW; p += 256) yb[cl * HW + p] = (xb[cl * HW + p] - mean) * rstd * ga + be;
    }
}

__global__ __launch_bounds__(256) void gn9f_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                      const float* __restrict__ gamma, float* __restrict__ dx,
                                                      float* __restrict__ part, int G, int HW) {
    __shared__ float smem[18 * 16];
    const int t = threadIdx.x, g = blockIdx.x % G, n9 = 9 * HW;
    const float* xb = x + (int64_t)blockIdx.x * n9;
    const float* gb = dy + (int64_t)blockIdx.x * n9;
    const float mean = mean_in[blockIdx.x], rstd = rstd_in[blockIdx.x];
    float s[18];  // per channel: sum dy, sum dy * xhat
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        float a = 0.f, b = 0.f;
        for (int p = t; p < HW; p += 256) {
            const float d = gb[cl * HW + p];
            a += d;
            b += d * ((xb[cl * HW + p] - mean) * rstd);
        }
        s[2 * cl] = a;
        s[2 * cl + 1] = b;
    }
    gn_block_sum_all<18>(s, smem);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = gamma[g * 9 + cl];
        c1 += ga * s[2 * cl];
        c2 += ga * s[2 * cl + 1];
    }
    const float inv = 1.f / (float)n9;
    c1 *= inv;
    c2 *= inv;
#pragma unroll
    for (int k = 0; k < 18; ++k)
        if (t == k) part[(int64_t)blockIdx.x * 18 + k] = s[k];
    float* db = dx + (int64_t)blockIdx.x * n9;
#pragma unroll
    for (int cl = 0; cl < 9; ++cl) {
        const float ga = gamma[g * 9 + cl];
        for (int p = t; p < HW; p += 256) {
            const float xh = (xb[cl * HW + p] - mean) * rstd;
            db[cl * HW + p] = rstd * (ga * gb[cl * HW + p] - c1 - xh * c2);
        }
    }
}

int gn9f_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N, int C,
                 int HW, float eps, hipStream_t stream) {
    const int G = C / 9;
    COT_LAUNCH(gn9f_fwd_kernel, dim3((unsigned)((int64_t)N * G)), dim3(256), 0, stream, (const float*)x,
               (const float*)gamma, (const float*)beta, (float*)y, mean, rstd, G, HW, eps);
    return check_launch("gn9f_fwd_kernel");
}

// smallest (threads, rounds) configuration whose 8*NT*R pixels cover one channel plane; 0 = not covered
static int gn9_config(int HW) {
    const int need = (HW + 7) / 8;  // pieces per channel
    if (need <= 64) return 1;
    if (need <= 128) return 2;
    if (need <= 256) return 3;
    if (need <= 512) return 4;
    if (need <= 1024) return 5;
    return 0;  // (> 8192 pixels per plane: caller keeps torch's GroupNorm)
}

#define GN9_SWITCH(CFG, CALL)                 \
    switch (CFG) {                            \
        case 1: { CALL(64, 1); } break;       \
        case 2: { CALL(64, 2); } break;       \
        case 3: { CALL(256, 1); } break;      \
        case 4: { CALL(256, 2); } break;      \
        default: { CALL(512, 2); } break;     \
    }

// (mean, rstd) of the GroupNorm-9 groups from the producing convolution's EPILOGUE statistics (conv_lds_common.h tile_epilogue:
// part [N][PT][C][2] = per image, 128-pixel tile and channel the sum and the sum of squares of the stored outputs): one wave per
// (image, group), PT x 9 entries added in fp64 in a fixed order.  Replaces the statistics half of gn9_fwd_kernel; the
// normalisation itself happens in the consumer's prologue (agg_fwd_nchw_k3_lds<SM = 2>, agg_bwd_nchw_k3_dot2 with Gn9Dot2).
__global__ __launch_bounds__(64) void gn9_stats_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int G, int C, int PT, int HW, float eps) {
    const int ng = blockIdx.x, n = ng / G, g = ng - n * G, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int e = lane; e < PT * 9; e += 64) {
        const int pt = e / 9, cl = e - pt * 9;
        const float* p = part + (((int64_t)n * PT + pt) * C + g * 9 + cl) * 2;
        s1 += (double)p[0];
        s2 += (double)p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) {
        const double cnt = 9.0 * (double)HW, m = s1 / cnt;
        double var = s2 / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean[ng] = (float)m;
        rstd[ng] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
int gn9_stats_finalize(const float* part, float* mean, float* rstd, int N, int C, int HW, float eps, hipStream_t stream) {
    const int G = C / 9, PT = ceil_div(HW, 128);
    COT_LAUNCH(gn9_stats_finalize_kernel, dim3((unsigned)(N * G)), dim3(64), 0, stream, part, mean, rstd, G, C, PT, HW, eps);
    return check_launch("gn9_stats_finalize");
}

int g_gn9_pack = 1;  // cot_set_tuning key 49: planes of at most 128 pixels (7 x 7, 10 x 10) -- several (image, group) pairs per wave (1 default), 0 = one each
// lanes per (image, group) of the packed form: the smallest of 8 / 16 / 32 that covers the plane's 8-pixel pieces, 64 = not packed
static int gn9_seg(int HW) {
    const int need = (HW + 7) / 8;
    if (!g_gn9_pack || need > 16) return 64;  // (32 lanes per group at 14 x 14 -- 25 of them busy -- measured the same as one wave each)
    return need <= 8 ? 8 : 16;
}
#define GN9_AL(KERNEL, SEG_, ...)                                                                            \
    do {                                                                                                     \
        if (HW % 8 == 0) COT_LAUNCH((KERNEL<64, 1, 16, SEG_>), pgrid, dim3(64), 0, stream, __VA_ARGS__);      \
        else if (HW % 4 == 0) COT_LAUNCH((KERNEL<64, 1, 8, SEG_>), pgrid, dim3(64), 0, stream, __VA_ARGS__);  \
        else COT_LAUNCH((KERNEL<64, 1, 2, SEG_>), pgrid, dim3(64), 0, stream, __VA_ARGS__);                   \
    } while (0)
#define GN9_PACKED(KERNEL, ...)                                          \
    do {                                                                 \
        const dim3 pgrid((unsigned)ceil_div(NG, 64 / seg));              \
        if (seg == 8) GN9_AL(KERNEL, 8, __VA_ARGS__);                    \
        else GN9_AL(KERNEL, 16, __VA_ARGS__);                            \
    } while (0)

int gn9_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N, int C,
                int HW, float eps, int lay, hipStream_t stream) {
    const int cfg = gn9_config(HW), G = C / 9;
    if (!cfg) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)N * G));
    const int seg = gn9_seg(HW), NG = N * G;
    if (seg < 64) {
        GN9_PACKED(gn9_fwd_kernel, (const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, mean, rstd, G, HW, eps, lay, NG);
        return check_launch("gn9_fwd_kernel");
    }
#define GN9_FWD(NT_, R_)                                                                                             \
    if (HW % 8 == 0)                                                                                                 \
        COT_LAUNCH((gn9_fwd_kernel<NT_, R_, 16>), grid, dim3(NT_), 16 * 4, stream, (const bf16_t*)x, (const bf16_t*)gamma, \
                   (const bf16_t*)beta, (bf16_t*)y, mean, rstd, G, HW, eps, lay, (int)grid.x);                                    \
    else if (HW % 4 == 0) /* 14 x 14: rows of 392 bytes start on 8-byte boundaries -- two 8-byte accesses per piece, not eight 2-byte ones */ \
        COT_LAUNCH((gn9_fwd_kernel<NT_, R_, 8>), grid, dim3(NT_), 16 * 4, stream, (const bf16_t*)x, (const bf16_t*)gamma, \
                   (const bf16_t*)beta, (bf16_t*)y, mean, rstd, G, HW, eps, lay, (int)grid.x);                                    \
    else                                                                                                             \
        COT_LAUNCH((gn9_fwd_kernel<NT_, R_, 2>), grid, dim3(NT_), 16 * 4, stream, (const bf16_t*)x, (const bf16_t*)gamma,  \
                   (const bf16_t*)beta, (bf16_t*)y, mean, rstd, G, HW, eps, lay, (int)grid.x)
    GN9_SWITCH(cfg, GN9_FWD)
#undef GN9_FWD
    return check_launch("gn9_fwd_kernel");
}

// dgamma / dbeta from the per-(image, channel) sums gn9_bwd_kernel left in `workspace` ([N][C][2] floats): the second, tiny launch of
// the backward -- its own entry point so that a caller may put it on the stream its other parameter gradients run on
int gn9_backward_params(const float* workspace, void* dgamma, void* dbeta, int N, int C, hipStream_t stream) {
    COT_LAUNCH((gn9_bwd_params_kernel<bf16_t>), dim3(ceil_div(C, 32)), dim3(256), 0, stream, workspace, (bf16_t*)dgamma, (bf16_t*)dbeta, N, C);
    return check_launch("gn9_bwd_params_kernel");
}

int gn9_backward(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, void* dx,
                 void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int lay, hipStream_t stream) {
    const int cfg = gn9_config(HW), G = C / 9;
    if (!cfg) return COT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((int64_t)N * G));
    const int seg = gn9_seg(HW), NG = N * G;
    if (seg < 64) {
        GN9_PACKED(gn9_bwd_kernel, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, (const bf16_t*)gamma, (bf16_t*)dx, workspace, G, HW, lay, NG);
        const int rc = check_launch("gn9_bwd_kernel");
        if (rc || !dgamma) return rc;  // (dgamma == NULL: the caller launches gn9_backward_params itself, e.g. on another stream)
        return gn9_backward_params(workspace, dgamma, dbeta, N, C, stream);
    }
#define GN9_BWD(NT_, R_)                                                                                             \
    if (HW % 8 == 0)                                                                                                 \
        COT_LAUNCH((gn9_bwd_kernel<NT_, R_, 16>), grid, dim3(NT_), 18 * 16 * 4, stream, (const bf16_t*)dy, (const bf16_t*)x,    \
                   mean, rstd, (const bf16_t*)gamma, (bf16_t*)dx, workspace, G, HW, lay, (int)grid.x);                            \
    else if (HW % 4 == 0)                                                                                            \
        COT_LAUNCH((gn9_bwd_kernel<NT_, R_, 8>), grid, dim3(NT_), 18 * 16 * 4, stream, (const bf16_t*)dy, (const bf16_t*)x,     \
                   mean, rstd, (const bf16_t*)gamma, (bf16_t*)dx, workspace, G, HW, lay, (int)grid.x);                            \
    else                                                                                                             \
        COT_LAUNCH((gn9_bwd_kernel<NT_, R_, 2>), grid, dim3(NT_), 18 * 16 * 4, stream, (const bf16_t*)dy, (const bf16_t*)x,     \
                   mean, rstd, (const bf16_t*)gamma, (bf16_t*)dx, workspace, G, HW, lay, (int)grid.x)
    GN9_SWITCH(cfg, GN9_BWD)
#undef GN9_BWD
    int rc = check_launch("gn9_bwd_kernel");
    if (rc || !dgamma) return rc;
    return gn9_backward_params(workspace, dgamma, dbeta, N, C, stream);
}

int gn9f_backward(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma, void* dx,
                  void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, hipStream_t stream) {
    const int G = C / 9;
    COT_LAUNCH(gn9f_bwd_kernel, dim3((unsigned)((int64_t)N * G)), dim3(256), 0, stream, (const float*)dy, (const float*)x,
               mean, rstd, (const float*)gamma, (float*)dx, workspace, G, HW);
    int rc = check_launch("gn9f_bwd_kernel");
    if (rc) return rc;
    COT_LAUNCH((gn9_bwd_params_kernel<float>), dim3(ceil_div(C, 32)), dim3(256), 0, stream, (const float*)workspace,
               (float*)dgamma, (float*)dbeta, N, C);
    return check_launch("gn9_bwd_params_kernel");
}

}  // namespace cot
