/*
 * cotnet_amd.h -- C ABI of libcotnet_hip.so, the MI355X (gfx950) drop-in for the CoT-block hot path
 * of JDAI-CV/CoTNet.
 *
 * What it replaces.  The reference reaches its device code through CuPy:
 *     kernel = cupy.cuda.compile_with_cache(src).get_function(name)      cupy_layers/utils.py:14-18
 *     kernel(block=(1024,1,1), grid=(G,1,1), args=[raw device ptrs],
 *            stream=Stream(ptr=torch.cuda.current_stream().cuda_stream)) cupy_layers/aggregation_zeropad.py:140-143
 * i.e. "raw device pointers + a stream handle", with every dimension baked into the JIT-compiled source.
 * This header is that same boundary, ahead-of-time compiled, with the dimensions as run-time arguments:
 * plain pointers, ints and an opaque hipStream_t -- no torch types.  A reference maintainer binds it
 * with ctypes (INTEGRATION.md shows the stub that replaces load_kernel/f(...) in AggregationZeropad).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) owned by the caller; kernels write EVERY element of their
 *     outputs (the reference allocates outputs uninitialised, aggregation_zeropad.py:123,:169,:178);
 *   - calls are asynchronous on `stream` (a hipStream_t cast to void*; NULL = the null stream),
 *     and re-entrant.  Process-global state is limited to (a) the thread-local last-error / last-kernel strings, (b) the
 *     DEVELOPER knobs of cot_set_tuning (kernel-variant selectors for A/B measurements and tests; every default is the measured
 *     best and a production caller never touches them -- they are read at launch time, so changing one while another thread
 *     launches is a race the caller must avoid) and (c) the cot_profile_* recorder;
 *   - return value: COT_OK (0) or a negative cot_status; cot_last_error() describes the failure.
 *     The reference reports errors through Python asserts (aggregation_zeropad.py:117,:122,:189).
 */
#ifndef COTNET_AMD_H
#define COTNET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COTNET_AMD_ABI_VERSION 1

typedef enum {
    COT_OK = 0,
    COT_ERR_INVALID_ARG = -1,   /* bad geometry / null pointer / unsupported combination */
    COT_ERR_UNSUPPORTED = -2,   /* dtype or layout not implemented for this entry point */
    COT_ERR_LAUNCH = -3         /* hipGetLastError() != hipSuccess after launch */
} cot_status;

typedef enum {
    COT_F32 = 0,   /* 'float'  in the reference (cupy_layers/utils.py:9-10) */
    COT_F64 = 1,   /* 'double' in the reference (cupy_layers/utils.py:11-12) */
    COT_BF16 = 2,  /* extension: bf16 storage, fp32 accumulation */
    COT_F16 = 3    /* extension: fp16 storage, fp32 accumulation */
} cot_dtype;

typedef enum {
    COT_NCHW = 0,  /* reference layout: x[N,C,H,W], w[N,heads,wC,kh*kw,Ho,Wo], out[N,heads*C,Ho,Wo] */
    COT_NHWC = 1   /* channels-last:    x[N,H,W,C], w[N,Ho,Wo,heads,wC,kh*kw], out[N,Ho,Wo,heads*C]
                      (what torch's channels_last gives for the same logical shapes / the 6-D view) */
} cot_layout;

/* Geometry of one aggregation_zeropad call; mirrors the ${...} substitutions of
 * cupy_layers/aggregation_zeropad.py:131-139.  Ho/Wo are derived with cot_agg_out_size. */
typedef struct {
    int32_t N, C, H, W;          /* input  [N, C, H, W]                                        */
    int32_t heads, wC;           /* weight [N, heads, wC, kh*kw, Ho, Wo];  C % wC == 0          */
    int32_t kh, kw;              /* kernel_size                                                */
    int32_t sh, sw;              /* stride                                                     */
    int32_t ph, pw;              /* zero padding                                               */
    int32_t dh, dw;              /* dilation                                                   */
} cot_agg_geom;

/* int((in + 2p - (d(k-1)+1)) / s + 1), aggregation_zeropad.py:120-121 */
int cot_agg_out_size(int in, int k, int s, int p, int d);

/* out[n,head,c,ho,wo] = sum_{kh,kw} w[n,head,c%wC,kh*KW+kw,ho,wo] * x[n,c,ho*s-p+kh*d, wo*s-p+kw*d]
 * replaces aggregation_zeropad_forward_kernel (aggregation_zeropad.py:20-46). */
int cot_agg_forward(const void* x, const void* w, void* out,
                    const cot_agg_geom* g, int dtype, int layout, void* stream);

/* gx[n,c,h,w] = sum_head sum_taps w[...]*gout[...] -- replaces
 * aggregation_zeropad_input_backward_kernel (aggregation_zeropad.py:48-79). */
int cot_agg_backward_input(const void* gout, const void* w, void* gx,
                           const cot_agg_geom* g, int dtype, int layout, void* stream);

/* gw[n,head,wc,tap,ho,wo] = sum_{cc = wc (mod wC)} x[n,cc,h_in,w_in]*gout[n,head,cc,ho,wo], 0 on padded taps
 * -- replaces aggregation_zeropad_weight_backward_kernel (aggregation_zeropad.py:81-110). */
int cot_agg_backward_weight(const void* gout, const void* x, void* gw,
                            const cot_agg_geom* g, int dtype, int layout, void* stream);

/* Both gradients from ONE pass over gout/x/w (21 B/elem instead of 25 B/elem in fp32).
 * Either of gx / gw may be NULL (then only the other is produced).  Same results as the two calls above. */
int cot_agg_backward(const void* gout, const void* x, const void* w, void* gx, void* gw,
                     const cot_agg_geom* g, int dtype, int layout, void* stream);

/* ---- window softmax fused in front of the aggregation: what LR-Net's SelfAttLayer does with
 * F.softmax(w, dim=3) followed by LocalConvolution (models/lr_net.py:94-96), in one kernel each way.
 *   forward : probs = softmax over the kh*kw taps of `logits` (same shape as a weight tensor), written to `probs`
 *             (may be NULL for inference); out = aggregation(x, probs)
 *   backward: gx as cot_agg_backward_input with w = probs; glogits_t = p_t*(g_t - sum_u p_u g_u), g = d(out)/d(probs)
 * NCHW, 3x3 / stride 1 / pad 1 / dilation 1 (backward: heads == 1) only: other geometries return
 * COT_ERR_UNSUPPORTED and the caller composes softmax + cot_agg_forward. */
int cot_agg_softmax_forward(const void* x, const void* logits, void* out, void* probs, const cot_agg_geom* g,
                            int dtype, void* stream);
int cot_agg_softmax_backward(const void* gout, const void* x, const void* probs, void* gx, void* glogits,
                             const cot_agg_geom* g, int dtype, void* stream);

/* ---- aggregation_zeropad_mix: 3x3 (w1) and 5x5 (w2) aggregation of the same x; NCHW only.
 * out[N, 2*heads*C, Ho, Wo] ordered [kernel_idx][head][c] (aggregation_zeropad_mix.py:20-74).
 * geometry: kh/kw/ph/pw of `g` describe the 3x3 set (kh=kw=3, pad1); p2h/p2w pad the 5x5 set.
 * backward_input reproduces the reference (only head 0 contributes, mix.py:87-88) unless
 * all_heads != 0. */
int cot_aggmix_forward(const void* x, const void* w1, const void* w2, void* out,
                       const cot_agg_geom* g, int p2h, int p2w, int dtype, void* stream);
int cot_aggmix_backward_input(const void* gout, const void* w1, const void* w2, void* gx,
                              const cot_agg_geom* g, int p2h, int p2w, int all_heads, int dtype, void* stream);
int cot_aggmix_backward_weight(const void* gout, const void* x, void* gw1, void* gw2,
                               const cot_agg_geom* g, int p2h, int p2w, int dtype, void* stream);

/* ---- library info / errors ---- */
int cot_abi_version(void);
const char* cot_last_error(void);          /* thread-local, valid until the next failing call */
const char* cot_status_string(int status);
/* name of the kernel variant the last successful aggregation call on this thread dispatched to
 * (e.g. "agg_fwd_nchw_k3<bf16,P8>"); used by tests to prove the fast path ran. */
const char* cot_last_kernel(void);

/* Developer knobs for A/B benchmarking (process-global, not part of the drop-in contract):
 *   key 0: 3x3 fast-path kernel version (0 auto, 1 = v1 scalar-halo, 2 = v2 wave-aligned, 3 = v3 LDS-staged kernels)
 *   key 1: max pixels per lane, forward   key 2: max pixels per lane, fused backward
 *   key 3: lane-exchange primitive (-1 probe on first use, 0 = DPP wave shift, 1 = ds_bpermute)
 *   key 4: v3 backward channel groups per LDS phase   key 5: v3 waves per workgroup (4|8)
 *   key 6: v3 extra LDS KiB per workgroup             key 7: v3 XCD-aware tile order (0|1)
 *   key 8: issue the fused backward as separate gX and gW launches (0|1)
 *   key 9: 1x1-convolution kernels XCD-aware wave order (0|1)   key 10: 16-row tiles per wave, forward (0 auto|2|4)
 *   key 11: weight-gradient target wave count (sizes the split of the reduction AND therefore cot_*_workspace: query the
 *           workspace after setting it); negative = force -value splits
 *   key 12: BatchNorm: fold the per-channel finalize step into the apply kernels (1 default | 0; one launch less each way)
 *   key 13: BatchNorm: most workgroups of the flat (grid-stride) apply kernels (default 4096; <= 0 restores it)
 *   key 14: convolutions: launches of at least this many waves use the shallow register ring (default 8192;
 *           a huge value = deep rings everywhere, 1 = shallow everywhere; <= 0 restores the default)
 *   key 15: LDS-pipelined convolution kernels (conv_lds.hip) on (1, default) / off (0: first-generation kernels)
 *   key 16: conv_lds FLAT mode: images per workgroup (0 = auto)
 *   key 17: conv_lds bit field -- bit 0 (retired: four-wave workgroups), bit 1 2-byte gathers instead of transposing LDS reads, bit 2 LDS
 *           weight gradient off, bit 3 general LDS weight gradient everywhere, bit 4 its deep-layer rule off, bit 5 data
 *           gradient on a transposed weight copy instead of reading the weight in place, bits 8..: 7 x 7 stage channel-block
 *           rule (0 default, 1 off, n > 1 workgroup-count threshold)
 *   key 18: BatchNorm: per-channel sample count up to which the fp64 one-wave-per-channel path is taken (default 256)
 *   key 19: weight gradients (register kernels): partial-sum bytes allowed as a percentage of the input bytes (default 50)
 *   key 20: the same for the LDS weight gradient (default 25 aligned planes / 100 general form)
 *   key 21: BatchNorm channel-resident kernels (1 default, 0 off, 256 / 512 / 1024 = forced workgroup size)
 *   key 22: one-image 1x1 convolutions on the staging-free kernels of conv_tiny.hip (1 default, 0 off)
 *   key 23: third-generation 1x1 forward / data-gradient kernel (conv_lds2.hip) bit field -- bit 0 off (second generation
 *           instead), bit 1 no fragment prefetch in the small-image (FLAT) kernels, bit 2 fragment prefetch in the BIG kernels
 *   key 24: DIAGNOSTIC timing ablations of the third-generation 1x1 kernel (results become wrong; 0 = off)
 *   key 25: third-generation 1x1 weight gradient (conv_wgrad2.hip) bit field -- bit 0 off, bit 1 the plain K loop (no fragment
 *           prefetch, 32-pixel stages), bit 2 old LDS chunk permutation of the 32-pixel forms, bit 4 general reduce kernel only,
 *           bit 6 32-pixel stages for every plane, bit 7 64-pixel stages + loader waves for every plane (default: planes of more
 *           than 64 pixels), bits 8..15 partial-sum cap in % of the input bytes (0 = 100), bits 16..23 target workgroups per
 *           CU x 4 (0 = 4), bits 24..30 forced slice count (tests)
 *   key 26: dry run (1): no kernel is launched, no HIP call is made; launches are recorded for cot_launch_log()
 *   key 27: 3x3 / stride-2 poolings: 1 (default) = row blocks (a lane owns a group of windows), 0 = one lane per pixel
 *   key 28: streaming BatchNorm kernels: workgroups aimed for (channels x batch chunks; default 1024).  Changes
 *           cot_bn_act_workspace: query it after setting the key
 *   key 29: bf16 fused aggregation backward on the packed dot-product kernel (agg_dot2.hip; 1 default, 0 = the fp32-unpacked
 *           LDS kernel)   key 30: its channel groups per LDS phase (0 = by width; 2 | 4 | 8)   key 31: its XCD-aware tile order
 *           (-1 automatic, 0 off, 1 on)   key 32: its waves per workgroup (0 = by width; 2 | 4 | 5 | 7)
 *   key 33: 1 (default) = operand halves outside an output's window are cleared, so Inf / NaN in gO reach exactly the gX
 *           elements the reference puts them in; 0 = they may also reach the next-nearest column (12 instructions per channel less)
 *   key 34: 1 (default) = its slabs are double-buffered (phase p+1 copied while phase p is computed), 0 = copy, wait, compute
 *   key 36: grouped 1x1 convolutions (cot_conv1x1g_*) group by group on the tuned LDS-pipelined kernels where a group's depth is a
 *           multiple of 32 (1 default), 0 = general kernels everywhere
 *   key 37: grouped 3x3 weight gradient with group widths off the 8-channel grid (12 per group): neighbouring groups merged, the
 *           wider convolution's gradient taken on the tuned kernels and its diagonal blocks copied out (1 default), 0 = general kernel
 *   key 38: (A/B builds only) weight-tile ring of the per-step form of the LDS-staged 3x3 kernel
 *   key 46: 1x1 forward / data gradient on 128-pixel tiles (planes of more than 256 pixels, i.e. also the channel-major rows of the
 *           deep stages): output-channel blocks of 64 / 32 instead of 128 while the launch has fewer workgroups than `value` (200 default, 0 = off)
 *   key 47: channel-resident BatchNorm: instances unrolled for 2 / 4 rounds where they cover the channel (1 default), 0 = always the
 *           full-capacity instance
 *   key 48: LDS layouts of the 1x1 forward / data-gradient kernel against bank conflicts (7 default; results identical either way):
 *           bit 0 = the X stage of 128-pixel tiles with its 16-byte chunks XOR-permuted per channel row, bit 1 / bit 2 = the weight
 *           tile's / transposed weight tile's chunk permutation in the form that is conflict-free under the hardware's lane groups
 *   key 49: GroupNorm of the attention logits on planes of at most 128 pixels (7 x 7, 10 x 10): several (image, group) pairs per
 *           wave (1 default), 0 = one workgroup each
 *   key 50: radix-2 tail kernels with channel-major descriptors on 7 x 7 bf16 planes: eight planes per wave, 7 lanes x 7 elements
 *           each (1 default), 0 = one wave per plane
 *   key 51: aggregation_zeropad_mix: 1 = the one-lane-per-element kernels for every call (A/B; default 0 = LDS-tiled kernels where
 *           the geometry is stride 1, padding 1 / 2);  key 52: lanes a tiled workgroup aims for (default 256);  key 53: pixels per lane
 *           (0 = the planner's choice per direction, else 1 / 2 / 4)
 *   key 54: 1x1 convolutions whose reduction depth is a multiple of 8 but not of 32 (CoXtLayer's groups of 24 / 48 / 216 / 432 channels)
 *           on the LDS-pipelined kernels with a partial last K step (1 default), 0 = first-generation / general kernels
 *   key 39: LDS-staged 3x3 forward / data gradient: 1 (default) = the chunk-resident form (all nine taps' weights of a 32-channel
 *           chunk in LDS, one barrier per chunk) for groups of >= 24 channels, 2 = also for 16-channel groups, 0 = the per-step
 *           ring everywhere
 *   key 40: channel-resident BatchNorm on odd planes that are multiples of 7 (7 x 7): 1 (default) = 7 elements per access,
 *           0 = element by element
 *   key 41: stem 7x7 forward: 1 (default) = input patch staged in LDS, 0 = operand gathered from global memory
 *   key 42: chunk-resident 3x3 form: 1 (default) = one weight buffer where that makes room for a second workgroup per CU, 0 = always two
 *   key 43: 1x1 forward / data gradient on whole small images (H*W <= 256), 128-row tiles: 1 (default) = three LDS stages instead of
 *           six when the launch has more than one workgroup per CU (two then share a CU), 0 = always six
 *   key 44: chunk-resident 3x3 form, groups of <= 32 channels on planes > 256 pixels: 1 (default) = 512- or 256-column tiles by
 *           rounds of workgroups, 0 = 512-column tiles wherever they fill the chip
 *   key 45: chunk-resident 3x3 form, order of the 32 channels of a chunk along K: 1 (default) = blocked or interleaved by LDS bank
 *           windows of the staged rows, 0 = always blocked, 2 = always interleaved
 * Keys 11, 15, 17 (bits 2-4), 19, 20, 25 change split counts / kernel choice: query cot_*_workspace after setting them. */
int cot_set_tuning(int key, int value);
/* Dry-run log of the calling thread (cot_set_tuning(26, 1)): one line per launch the library WOULD have issued --
 * "kernel | launcher instantiation | grid | block | lds".  Copies up to cap-1 bytes into buf (NUL-terminated), clears the log
 * and returns the bytes copied; buf == NULL returns the pending size.  Used to pin the dispatch table without a GPU. */
int cot_launch_log(char* buf, int cap);
/* 0 if the device probe confirmed the DPP wave_shr/wave_shl semantics the v2 kernels rely on, 1 if the
 * library fell back to ds_bpermute.  Launches a 64-thread probe kernel on the null stream on first call. */
int cot_xchg_mode(void);

/* ---- radix-2 split-attention tail of the CoT layer (models/cotnet.py:92-104), NCHW, `planes` = N*C, HW = H*W:
 *   cot_radix_gap           gap[plane] = mean_hw(y + k)                    (replaces cat / sum(dim=2) / mean, :95-98)
 *   cot_radix_mix           out = y*attn[plane][0] + k*attn[plane][1]      (replaces (x*attn).sum(dim=2), :102)
 *   cot_radix_mix_backward  gy = g*a0, gk = g*a1, gattn[plane] = (sum g*y, sum g*k)
 * attn / gattn are [planes][2] in the storage dtype (the softmax over the radix pair, :100-101, stays in torch). */
int cot_radix_gap(const void* y, const void* k, void* gap, int64_t planes, int HW, int dtype, void* stream);
int cot_radix_mix(const void* y, const void* k, const void* attn, void* out, int64_t planes, int HW, int dtype,
                  void* stream);
int cot_radix_mix_backward(const void* gout, const void* y, const void* k, const void* attn, void* gy, void* gk,
                           void* gattn, int64_t planes, int HW, int dtype, void* stream);

/* ---- 1x1 convolution on NCHW tensors WITHOUT layout changes (CotLayer.embed[0], embed[3],
 * conv1x1[0] -- models/cotnet.py:51-62 -- and Bottleneck.conv1/conv3/downsample, models/cotnet.py:206-224).  Replaces
 * nn.Conv2d(kernel_size=1, stride=1, groups=1) forward and both gradients.  COT_BF16 (fp32 accumulation; channel counts
 * that are not multiples of 8 return COT_ERR_UNSUPPORTED and the caller keeps nn.Conv2d -- or uses cot_conv1x1g_* with
 * groups = 1) and COT_F32 (the reference's own precision, `amp: False`: fp32 MFMA, any channel counts, one input tensor:
 * x2 / gx2 must be NULL and the bias is fp32).  ONE image of at most 256 pixels (N == 1: the `se` branch of the single-node
 * layers, whose "pixels" are the batch) runs on staging-free kernels (csrc/conv_tiny.hip).
 *     y[n][co][p] = sum_ci weight[co][ci] * x[n][ci][p] + bias[co]         weight [Co][Ci] row-major, bias NULL or [Co]
 * x may be given as TWO channel slabs that the reference concatenates first (`torch.cat([x, k], dim=1)`,
 * models/cotnet.py:81): x1 = [N][c1][HW], x2 = [N][Ci-c1][HW]; x2 == NULL means one tensor and c1 must equal Ci.
 * backward_data writes the gradient of that concatenation to gx1 / gx2 the same way; `accumulate` bit 0 / bit 1 make it
 * ADD to the existing contents of gx1 / gx2 instead (a tensor with several consumers collects its gradient without
 * separate add kernels; the sum is formed in fp32 and rounded once).
 * backward_weight: gweight [Co][Ci], gbias [Co] or NULL; deterministic (partial sums in `workspace`, no atomics).
 * workspace: cot_conv1x1_workspace(...) BYTES (256-byte multiple): partial sums of backward_weight (backward_data reads
 * the weight tensor in place, transposed, and only checks the pointer). */
int64_t cot_conv1x1_workspace(int N, int Ci, int Co, int HW, int has_bias);
int cot_conv1x1_forward(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, int N,
                        int Ci, int Co, int HW, int dtype, void* stream);
int cot_conv1x1_backward_data(const void* gy, const void* weight, void* gx1, void* gx2, int c1, int accumulate,
                              void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream);
int cot_conv1x1_backward_weight(const void* gy, const void* x1, const void* x2, int c1, void* gweight, void* gbias,
                                void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream);

/* ---- grouped 3x3 convolution, stride 1, padding 1, NCHW, no layout changes (CotLayer.key_embed[0] =
 * nn.Conv2d(dim, dim, 3, padding=1, groups=4, bias=False), models/cotnet.py:43-47; groups=8 in CoXtLayer, :112-116).
 * weight [Cout][Cin/groups][3][3] as torch stores it.  COT_BF16 (fp32 accumulation) or COT_F32; any channels per group
 * (multiples of 8 in bf16 take the tuned kernels, everything else the general ones of csrc/conv_gen.hip).
 *   masks:     per-pixel tap-validity table for an H x W image: cot_conv3x3g_masks_bytes(H, W) bytes, filled once by
 *              cot_conv3x3g_masks and reusable by every call with the same H, W (read-only afterwards)
 *   workspace: cot_conv3x3g_workspace(...) bytes (partial sums of the weight gradient; forward and backward_data read
 *              the weight tensor in place and only check the pointer).  backward_weight is deterministic (no atomics).
 *   accumulate (backward_data): nonzero = gx += result instead of gx = result. */
int64_t cot_conv3x3g_masks_bytes(int H, int W);
int cot_conv3x3g_masks(void* masks, int H, int W, void* stream);
int64_t cot_conv3x3g_workspace(int N, int Cin, int Cout, int groups, int H, int W);
int cot_conv3x3g_forward(const void* x, const void* weight, void* y, const void* masks, void* workspace, int N, int Cin,
                         int Cout, int groups, int H, int W, int dtype, void* stream);
int cot_conv3x3g_backward_data(const void* gy, const void* weight, void* gx, int accumulate, const void* masks,
                               void* workspace, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                               void* stream);
int cot_conv3x3g_backward_weight(const void* gy, const void* x, void* gweight, const void* masks, void* workspace, int N,
                                 int Cin, int Cout, int groups, int H, int W, int dtype, void* stream);
/* The same with a promise about the memory around x: `x_guard_elems` elements before x[0] and behind x[N*Cin*H*W - 1] belong
 * to the caller's allocation and may be READ (their contents do not matter).  With x_guard_elems >= W + 1 the COT_BF16 weight
 * gradient of layers with 16 | Cin/groups and Cout/groups in {16, 32, 64 k} runs on the LDS-staged kernel (conv_wgrad2.hip,
 * TAPS form: nine shifted views of x as extra rows of the 1x1 weight gradient's GEMM); otherwise, and always with 0, on
 * the per-wave kernel.  Same results up to the order of the fp32 sums; deterministic either way. */
int cot_conv3x3g_backward_weight_guarded(const void* gy, const void* x, void* gweight, const void* masks, void* workspace, int N,
                                         int Cin, int Cout, int groups, int H, int W, int dtype, int x_guard_elems, void* stream);

/* Packings made ahead of time (round 5).  The LDS kernels run on a re-ordered copy of the weights that the forward / data-gradient
 * entry points above make per call in their workspace; a layer's weights change once per optimizer step, so a trainer can pack right
 * after the step, off the critical path (cotnet_amd/cot_layer_fused.py does, on its side stream):
 *   cot_conv3x3g_packed_bytes  size of one packing
 *   cot_conv3x3g_pack          mode 0 = the packing cot_conv3x3g_forward would make for this geometry, 1 = cot_conv3x3g_backward_data's
 *   cot_conv3x3g_*_packed      the same convolutions on such a packing (same geometry arguments as the pack call)
 * COT_BF16, geometries the LDS kernels cover; otherwise COT_ERR_UNSUPPORTED (use the ordinary entry points). */
int64_t cot_conv3x3g_packed_bytes(int Cin, int Cout, int groups);
int cot_conv3x3g_pack(const void* weight, void* packed, int mode, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                      void* stream);
int cot_conv3x3g_forward_packed(const void* x, const void* packed, void* y, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                                void* stream);
int cot_conv3x3g_backward_data_packed(const void* gy, const void* packed, void* gx, int accumulate, int N, int Cin, int Cout, int groups,
                                      int H, int W, int dtype, void* stream);

/* ---- grouped 1x1 convolution, NCHW (CoXtLayer.embed[0] = Conv2d(2*dim, dim/2, 1, groups=2),
 * embed[3] = Conv2d(dim/2, 9*dim/8, 1, groups=2) with bias, conv1x1[0] = Conv2d(dim, dim, 1, groups=2),
 * models/cotnet.py:123-131).  weight [Co][Ci/groups] as torch stores it, bias NULL or [Co]; COT_BF16 or COT_F32, any channel
 * counts (12 / 24 / 54 per group occur).  backward_weight is deterministic (partial sums in `workspace`, fixed order);
 * accumulate (backward_data): nonzero = gx += result.
 * Workspace of the general kernels: cot_convg_workspace(N, Cin, Cout, groups, H, W, ksize) bytes, ksize 1 (H*W = the plane,
 * pass H = HW, W = 1) or 3.  It is what backward_weight needs for every cot_conv1x1g_* call AND for the cot_conv1x1_* /
 * cot_conv3x3g_* calls that the general kernels serve: COT_F32 tensors, and COT_BF16 3x3 convolutions with channels per
 * group that are not multiples of 8 (cot_conv1x1_workspace / cot_conv3x3g_workspace size the tuned bf16 kernels only). */
int64_t cot_convg_workspace(int N, int Cin, int Cout, int groups, int H, int W, int ksize);
int cot_conv1x1g_forward(const void* x, const void* weight, const void* bias, void* y, int N, int Ci, int Co, int groups,
                         int HW, int dtype, void* stream);
int cot_conv1x1g_backward_data(const void* gy, const void* weight, void* gx, int accumulate, int N, int Ci, int Co,
                               int groups, int HW, int dtype, void* stream);
int cot_conv1x1g_backward_weight(const void* gy, const void* x, void* gweight, void* gbias, void* workspace, int N, int Ci,
                                 int Co, int groups, int HW, int dtype, void* stream);

/* Channel-major variants for running the `se` branch (models/cotnet.py:71-77,:98-101) on the library's own kernels: with
 * the pooled descriptor stored [C][N] the two 1x1 convolutions of `se` are cot_conv1x1_* calls on ONE image of N
 * "pixels" and its BatchNorm is cot_bn_act_* over those N pixels (= over the batch, as nn.BatchNorm2d on [N,A,1,1]).
 *   cot_radix_gap_t               gapT[c][n] = mean_hw(y + k)   (k == NULL: mean_hw(y), the classifier head's global pooling)
 *   cot_radix_mix_logits          attn[n][c][0..1] = softmax over the pair (logitsT[2c][n], logitsT[2c+1][n])  (saved for
 *                                 backward), out = y*attn0 + k*attn1
 *   cot_radix_mix_backward_reduce glogitsT[2c][n] = a0*a1*(sum g*y - sum g*k) = -glogitsT[2c+1][n]  (pair-softmax backward)
 *   cot_radix_mix_backward_apply  gy = g*a0 + ggapT[c][n]/HW,  gk = g*a1 + ggapT[c][n]/HW */
int cot_radix_gap_t(const void* y, const void* k, void* gapT, int N, int C, int HW, int dtype, void* stream);
int cot_radix_mix_logits(const void* y, const void* k, const void* logitsT, void* out, void* attn, int N, int C, int HW,
                         int dtype, void* stream);
int cot_radix_mix_backward_reduce(const void* gout, const void* y, const void* k, const void* attn, void* glogitsT, int N,
                                  int C, int HW, int dtype, void* stream);
int cot_radix_mix_backward_apply(const void* gout, const void* attn, const void* ggapT, void* gy, void* gk, int N, int C,
                                 int HW, int dtype, void* stream);

/* ---- SE-style sigmoid gate of SplitAttnConv2d(radix = 1) as SE-CoTNetD uses it (reference
 * models/layers/split_attn.py:62-88, models/cotnet_hybrid.py:143-146):  out = x * sigmoid(fc2(relu(bn1(fc1(mean_hw(x))))));
 * the two fc layers and bn1 act on [N, C] descriptors (cot_conv1x1_* / cot_bn_act_* or any GEMM), these three kernels are the
 * passes over x: `planes` = N*C planes of HW elements, gap / logit / glogit [planes] in the storage dtype, COT_F32 / COT_BF16.
 *   cot_se_gap            gap[i] = mean_hw(x[i])
 *   cot_se_gate           out[i] = x[i] * sigmoid(logit[i])
 *   cot_se_gate_backward  gx[i] = g[i] * sigmoid(logit[i]);  glogit[i] = sigmoid'(logit[i]) * sum_hw g[i] * x[i]
 *                         (the gradient that arrives through the pooled descriptor is added by the caller: d gap / HW) */
int cot_se_gap(const void* x, void* gap, int64_t planes, int HW, int dtype, void* stream);
int cot_se_gate(const void* x, const void* logit, void* out, int64_t planes, int HW, int dtype, void* stream);
int cot_se_gate_backward(const void* g, const void* x, const void* logit, void* gx, void* glogit, int64_t planes, int HW,
                         int dtype, void* stream);

/* ---- GroupNorm with 9 channels per group, NCHW, COT_BF16 or COT_F32 (CotLayer.embed[4] =
 * nn.GroupNorm(dim/8, 9*dim/8), models/cotnet.py:56 -- the normalisation of the 3x3 attention logits; its output is the
 * aggregation's weight tensor).  One (image, group) is a contiguous run of 9*HW elements that one workgroup keeps in
 * registers: forward = 1 read + 1 write, backward = 2 reads + 1 write + a tiny batch reduction.
 *   y = (x - mean_g) * rstd_g * gamma[c] + beta[c];  mean / rstd: fp32 [N * C/9], written by forward, read by backward
 *   gamma / beta / dgamma / dbeta: [C] in the storage dtype;  workspace (backward): N*C*2 floats
 * COT_BF16 with HW > 8192 returns COT_ERR_UNSUPPORTED (caller keeps torch's GroupNorm); COT_F32 sweeps the group through
 * L2 instead of holding it in registers and takes any HW. */
int cot_group_norm9_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N,
                            int C, int HW, float eps, int dtype, void* stream);
int cot_group_norm9_backward(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma,
                             void* dx, void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int dtype,
                             void* stream);

/* ---- GroupNorm-9 fused into its producer and its consumer (reference: models/cotnet.py:55-56 embed[3] -> embed[4],
 * :84-88 w.view(...) -> local_conv).  The statistics come out of the EPILOGUE of the 1x1 convolution that writes the logits, the
 * normalisation is applied in the PROLOGUE of the aggregation kernels; the normalised tensor is never written or read:
 *   cot_conv1x1_forward_gn9   = cot_conv1x1_forward that also fills `stats` (cot_gn9_stats_floats(N, Co, HW) floats: per image,
 *                               128-pixel tile and channel the sum and sum of squares of the stored bf16 outputs)
 *   cot_gn9_stats_finalize    -> mean / rstd [N * Co/9] (fp64 sums, one wave per (image, group))
 *   cot_agg_gn9_forward       = cot_agg_forward whose weight tensor is GroupNorm-9(logits): logits [N, wC*9, H, W] raw, mean / rstd
 *                               indexed by plane n*wC + wc, gamma / beta [groups_per_image*9] (group of a plane = plane %
 *                               groups_per_image: wC, or 2*wC when two convolution groups are folded into the batch)
 *   cot_agg_gn9_backward      = cot_agg_backward on the same operands; gw = gradient w.r.t. the NORMALISED weights = the `dy` of
 *                               cot_group_norm9_backward(dy, x = logits, mean, rstd, ...), which stays the backward of the norm
 * COT_BF16, 3x3 / stride 1 / pad 1 / one head, planes of more than 256 pixels with a width the packed dot-product backward covers:
 * cot_gn9_fused_covers(Ci, c1, two_slabs, HW, W) says whether a layer qualifies (else COT_ERR_UNSUPPORTED: compose the three ops). */
int64_t cot_gn9_stats_floats(int N, int C, int HW);
int cot_gn9_fused_covers(int Ci, int c1, int two_slabs, int HW, int W);
int cot_conv1x1_forward_gn9(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, float* stats,
                            int N, int Ci, int Co, int HW, int dtype, void* stream);
int cot_gn9_stats_finalize(const float* stats, float* mean, float* rstd, int N, int C, int HW, float eps, void* stream);
int cot_agg_gn9_forward(const void* x, const void* logits, const float* mean, const float* rstd, const void* gamma, const void* beta,
                        int groups_per_image, void* out, const cot_agg_geom* g, int dtype, void* stream);
int cot_agg_gn9_backward(const void* gout, const void* x, const void* logits, const float* mean, const float* rstd, const void* gamma,
                         const void* beta, int groups_per_image, void* gx, void* gw, const cot_agg_geom* g, int dtype, void* stream);

/* ---- conv1's data gradient of an identity-shortcut Bottleneck with the residual's gradient folded in (models/cotnet.py:228-264: the
 * block returns relu(bn3(...) + x), so d loss / d x = conv1's data gradient + gout * [block output > 0]):
 *   gx = weight^T [Ci][Co] . gy  +  gout * relu_mask      gy [N, Co, HW], gx / gout [N, Ci, HW], relu_mask = bn3's sign mask
 * (cot_bn_act_forward_mask: one bit per element of the [N, Ci, HW] block output).  bn3's backward is then called with dresidual = NULL.
 * COT_BF16, HW % 8 == 0, the LDS kernels' geometries: cot_conv1x1_backward_data_relu_res_covers(...) == 1, else COT_ERR_UNSUPPORTED
 * (keep cot_bn_act_backward_mask's dresidual + cot_conv1x1_backward_data(accumulate = 1)). */
int cot_conv1x1_backward_data_relu_res_covers(int N, int Ci, int Co, int HW, int dtype);
int cot_conv1x1_backward_data_relu_res(const void* gy, const void* weight, void* gx, const void* gout, const void* relu_mask, int N, int Ci,
                                       int Co, int HW, int dtype, void* stream);

/* ---- the backbone's first convolution: 7x7, stride 2, padding 3, 3 -> 64 channels, NCHW, COT_BF16 (MFMA implicit GEMM) or COT_F32
 * (plain fp32 kernels: the reference's own precision) (reference: models/resnet.py:539-555, conv1 of the default stem), forward and
 * weight gradient (the network input takes no gradient).
 * weight [64][3][7][7] as torch stores it; x [N][3][H][W]; y / gy [N][64][Ho][Wo], Ho = (H - 1)/2 + 1.  Covered when Wo is a
 * multiple of 8 and Ho*Wo of 32 (224, 256, 288, 320 inputs); otherwise COT_ERR_UNSUPPORTED / workspace 0 (caller keeps
 * nn.Conv2d).  backward_weight is deterministic; workspace: cot_stem7x7s2_workspace(...) bytes. */
int64_t cot_stem7x7s2_workspace(int N, int H, int W);
int cot_stem7x7s2_forward(const void* x, const void* weight, void* y, int N, int H, int W, int dtype, void* stream);
int cot_stem7x7s2_backward_weight(const void* gy, const void* x, void* gweight, void* workspace, int N, int H, int W,
                                  int dtype, void* stream);

/* ---- a deep stem's first convolution: 3x3, stride 2, padding 1, 3 -> Cout channels (Cout = 32 or 64), no bias, NCHW, COT_BF16
 * (replaces nn.Conv2d(in_chans, stem_chs_1, 3, stride=2, padding=1, bias=False), models/cotnet_hybrid.py:359 -- the SE-CoTNetD
 * stems -- and the 'deep' stems of models/resnet.py), forward and weight gradient (the network input takes no gradient).
 * weight [Cout][3][3][3] as torch stores it; x [N][3][H][W]; y / gy [N][Cout][Ho][Wo], Ho = (H - 1)/2 + 1.  Covered when Wo is a
 * multiple of 8 and Ho*Wo of 32; otherwise COT_ERR_UNSUPPORTED / workspace 0 (caller keeps nn.Conv2d).  backward_weight is
 * deterministic; workspace: cot_stem3x3s2_workspace(...) bytes.  (The stem's two stride-1 3x3 convolutions are cot_conv3x3g_*
 * calls with groups = 1.) */
int64_t cot_stem3x3s2_workspace(int N, int H, int W, int Cout);
int cot_stem3x3s2_forward(const void* x, const void* weight, void* y, int N, int H, int W, int Cout, int dtype, void* stream);
int cot_stem3x3s2_backward_weight(const void* gy, const void* x, void* gweight, void* workspace, int N, int H, int W, int Cout,
                                  int dtype, void* stream);

/* ---- the backbone's two 3x3 / stride-2 / padding-1 poolings, NCHW, `planes` = N*C images of H x W -> Ho = (H-1)/2 + 1:
 *   cot_avgpool3x3s2_*  nn.AvgPool2d(3, 2, padding=1) (count_include_pad: every window / 9) -- the "avd" pooling of
 *                       stride-2 bottlenecks, models/cotnet.py:216
 *   cot_maxpool3x3s2_*  nn.MaxPool2d(kernel_size=3, stride=2, padding=1) after the stem, models/resnet.py:556-561; the
 *                       backward recomputes the arg-max from x with torch's tie rule (first maximum in row-major window
 *                       order) instead of reading an index tensor.   COT_F32 / COT_BF16.
 *   cot_maxpool3x3s2_forward_taps / _backward_taps: the same pooling with the arg-max kept as ONE BYTE per output window
 *                       (taps [planes][Ho][Wo], value kh*3 + kw of the winning element): the backward reads dY and the taps
 *                       and never touches x (torch keeps an int64 index per window). */
/*   cot_subsample2_*    x[:, :, ::2, ::2] -- what a stride-2 1x1 projection shortcut reads (models/resnet.py downsample: nn.Conv2d(
 *                       kernel_size=1, stride=2)) -- as its own contiguous tensor y [planes][H/2][W/2], and the gradient of that
 *                       selection (gx [planes][H][W]: gy's values in place, zeros elsewhere, every element written).  Even H, W. */
int cot_subsample2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream);
int cot_subsample2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream);
/*   cot_avgpool2x2s2_*  nn.AvgPool2d(2, 2) on even H, W: the pooling in front of the 1x1 projection of an `avg_down` shortcut
 *                       (models/resnet.py:380-394 downsample_avg, used by cotnet_hybrid.py's SE-CoTNetD; with even planes its
 *                       ceil_mode / count_include_pad flags change nothing).  y [planes][H/2][W/2]; backward writes every element
 *                       of gx [planes][H][W] (= gy / 4 under its window).  COT_F32 / COT_BF16; odd H or W: COT_ERR_UNSUPPORTED. */
int cot_avgpool2x2s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream);
int cot_avgpool2x2s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream);
int cot_avgpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream);
int cot_avgpool3x3s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream);
int cot_maxpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream);
int cot_maxpool3x3s2_backward(const void* gy, const void* x, void* gx, int64_t planes, int H, int W, int dtype,
                              void* stream);
int cot_maxpool3x3s2_forward_taps(const void* x, void* y, void* taps, int64_t planes, int H, int W, int dtype, void* stream);
/*   cot_blurpool3x3s2_* BlurPool2d(channels, filt_size=3, stride=2) of SE-CoTNetD (reference
 *                       models/layers/blur_pool.py:53-58: ReflectionPad2d(1) + depthwise binomial [1 2 1]x[1 2 1]/16, stride 2)
 *                       as a 9-tap stencil, forward and gather-form backward; H, W >= 2. */
int cot_blurpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream);
int cot_blurpool3x3s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream);
int cot_maxpool3x3s2_backward_taps(const void* gy, const void* taps, void* gx, int64_t planes, int H, int W, int dtype,
                                   void* stream);

/* ---- fused SGD over a flat parameter bucket (replaces torch.optim.SGD(nesterov=True),
 * optim/optim_factory.py:54-56, which launches per parameter tensor):
 *     g = grad*grad_scale + weight_decay*p;  buf = momentum*buf + g;  p -= lr*(nesterov ? g + momentum*buf : buf)
 * param_dtype COT_BF16 needs `master` (fp32 copy, updated in place; `param` receives the rounded working copy);
 * param_dtype COT_F32 takes master == NULL.  grad_dtype COT_BF16 or COT_F32.  momentum_buf is fp32, n elements. */
int cot_sgd_step(void* param, void* master, void* momentum_buf, const void* grad, int64_t n, float lr,
                 float momentum, float weight_decay, float grad_scale, int nesterov, int param_dtype, int grad_dtype,
                 void* stream);

/* ---- exponential moving average of the weights over a flat buffer (replaces the per-tensor
 * `ema = decay*ema + (1-decay)*model` of ModelEmaV2._update, utils/model_ema.py:45-53):  ema fp32 [n], src the fp32
 * master copy / fp32 parameters (COT_F32) or bf16 parameters (COT_BF16). */
int cot_ema_step(void* ema, const void* src, int64_t n, float decay, int src_dtype, void* stream);

/* 1 when cot_conv1x1_forward / _backward_data run the LDS-tiled second-generation kernel (csrc/conv_lds.hip) for this
 * reduction length K (first slab k1, two_slabs 0/1) and plane size HW, 0 when they take the first-generation kernel. */
int cot_conv1x1_lds_covers(int K, int k1, int two_slabs, int HW);

/* ---- on-device input pipeline (replaces the three elementwise kernels of the reference's
 * PrefetchLoader, datasets/loader.py:85-90: `next_input.float().sub_(mean).div_(std)` / the `.half()` form):
 *     y[n,c,h,w] = (x[n,c,h,w] - mean[c]) / std[c],   x uint8 NCHW, mean/std fp32 device arrays of C entries
 *     (the caller passes 255*mean and 255*std as the reference does, loader.py:66-67)
 * `planes` = N*C images of HW pixels.  dtype of y: COT_F32 (bit-identical to torch: IEEE subtract then divide),
 * COT_F16 (the reference's fp16=True: each step rounded to half), COT_BF16 (fp32 arithmetic, one rounding). */
int cot_input_normalize(const void* x_u8, void* y, const float* mean, const float* std, int64_t planes, int C, int HW,
                        int dtype, void* stream);

/* ---- training-mode BatchNorm2d fused with activation and residual add, NCHW.
 * Replaces nn.BatchNorm2d + in-place ReLU/SiLU (+ `x += residual`) sequences of the reference's blocks
 * (models/cotnet.py:231-235, :248-262, :89-90):
 *     y = act(gamma*(x-mean_c)*rstd_c + beta [+ residual]),   act: 0 identity, 1 ReLU, 2 SiLU
 * Batch statistics over (N, H*W) per channel (biased variance), saved in save_mean / save_rstd [C] for backward;
 * running_mean/var (may both be NULL) are updated with `momentum` and the unbiased variance, as torch does;
 * num_batches_tracked (int64 scalar on the device, may be NULL) is incremented by one, as nn.BatchNorm2d does.
 * gamma/beta/statistics are fp32; x/residual/y are `dtype` (COT_F32 or COT_BF16).  workspace: cot_bn_act_workspace
 * floats.  backward: dx, dgamma, dbeta (and dresidual = dy*act' when non-NULL); ReLU with a residual needs the saved
 * output y (its sign is the mask); without a residual y may be NULL and the mask is recomputed from x.  SiLU after a residual add
 * (or, in the _ps form, a per-sample scale) has a forward only: its backward returns COT_ERR_UNSUPPORTED (the reference has no
 * such block). */
int cot_bn_act_workspace(int N, int C);
int cot_bn_act_forward(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                       float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* workspace, int N, int C, int HW, float eps, float momentum,
                       int act, int dtype, void* stream);
int cot_bn_act_backward(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                        float* workspace, int N, int C, int HW, int act, int dtype, void* stream);
/* The same with stochastic depth folded in (models/cotnet.py:250-262 bn3 -> drop_path -> += residual -> act3; models/layers/
 * drop.py:140-168): `sample_scale` [N] fp32 on the device holds, per sample, 0 (path dropped) or 1 / keep_prob:
 *     y = act(sample_scale[n] * (gamma*(x-mean_c)*rstd_c + beta) + residual)
 * backward: dresidual = dy*act', the BatchNorm backward runs on sample_scale[n] * dy*act'.  NULL = no scaling. */
int cot_bn_act_forward_ps(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                          float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* workspace, const float* sample_scale, int N, int C, int HW,
                          float eps, float momentum, int act, int dtype, void* stream);
int cot_bn_act_backward_ps(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                           const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                           float* workspace, const float* sample_scale, int N, int C, int HW, int act, int dtype, void* stream);
/* ReLU sign mask for bn3 + residual + ReLU (models/cotnet.py:250-262): its backward needs the sign of its OUTPUT; instead of reading
 * the saved output back (1/5 of that backward's traffic) the forward can also write one byte per 8 output elements (bit k = element
 * k > 0, from the rounded output) and the backward read those.  cot_bn_relu_mask_bytes: size of `relu_mask` for this tensor, 0 = not
 * supported (use the _ps entry points with the saved output).  act must be 1 (ReLU); results are identical to the _ps pair's. */
int64_t cot_bn_relu_mask_bytes(int N, int C, int HW, int dtype);
int cot_bn_act_forward_mask(const void* x, const void* residual, void* y, void* relu_mask, const float* gamma, const float* beta,
                            float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* workspace, const float* sample_scale, int N, int C, int HW,
                            float eps, float momentum, int act, int dtype, void* stream);
int cot_bn_act_backward_mask(const void* dy, const void* x, const void* relu_mask, void* dx, void* dresidual, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                             float* workspace, const float* sample_scale, int N, int C, int HW, int act, int dtype, void* stream);
/* ---- per-tensor layouts for the deep stages (round 5).  The 1x1 convolutions and BatchNorms of a Bottleneck
 * (models/cotnet.py:228-264, :43-62) take any layout in which a channel's samples are contiguous; stored "channel-major" --
 * [C][N][HW], element (n, c, p) at (c*N + n)*HW + p -- a channel is ONE row of N*HW elements and the ordinary entry points above
 * (cot_conv1x1_*, cot_bn_act_*) serve it with N = 1, HW' = N*HW (measured 1.2-1.5x faster than [N][C][HW] on the 14 x 14 / 7 x 7
 * layers, profiles/r05_probe_cnhw.log).  The plane kernels between them (grouped 3x3, aggregation) stay NCHW; the layout changes
 * inside kernels that move every element anyway.  `lay` is a bit mask with one bit per tensor argument in the order given below:
 * 0 = NCHW, 1 = channel-major.  COT_BF16 only.
 *   cot_bn_act_forward_lay    cot_bn_act_forward_ps with a second output y2 (NULL: none) of the same values; lay bits: 1 x, 2 residual,
 *                             4 y, 8 y2.  cot_bn_act_lay_covers(N, C, HW, dtype) == 1 where the channel-resident kernels hold a
 *                             channel (else COT_ERR_UNSUPPORTED: keep one layout)
 *   cot_bn_act_backward_lay   cot_bn_act_backward_ps whose upstream gradient is dy + dy2 (dy2 NULL: dy alone; fp32 sum, one
 *                             rounding); lay bits: 1 dy, 2 dy2, 4 x, 8 y, 16 dx, 32 dresidual
 *   cot_radix_*_lay           lay bits in argument order of the plane tensors: gap_t: 1 y, 2 k;  mix_logits: 1 y, 2 k, 4 out;
 *                             backward_reduce: 1 gout, 2 y, 4 k;  backward_apply: 1 gout, 2 gy, 4 gk.  (attn, gapT, logitsT keep
 *                             their index order)
 *   cot_group_norm9_*_lay     forward: 1 x, 2 y;  backward: 1 dy, 2 x, 4 dx */
int cot_bn_act_lay_covers(int N, int C, int HW, int dtype);
int cot_bn_act_forward_lay(const void* x, const void* residual, void* y, void* y2, const float* gamma, const float* beta, float* save_mean,
                           float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           const float* sample_scale, int N, int C, int HW, float eps, float momentum, int act, int lay, int dtype,
                           void* stream);
int cot_bn_act_backward_lay(const void* dy, const void* dy2, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                            const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                            const float* sample_scale, int N, int C, int HW, int act, int lay, int dtype, void* stream);
int cot_radix_gap_t_lay(const void* y, const void* k, void* gapT, int N, int C, int HW, int lay, int dtype, void* stream);
int cot_radix_mix_logits_lay(const void* y, const void* k, const void* logitsT, void* out, void* attn, int N, int C, int HW, int lay,
                             int dtype, void* stream);
int cot_radix_mix_backward_reduce_lay(const void* gout, const void* y, const void* k, const void* attn, void* glogitsT, int N,
                                      int C, int HW, int lay, int dtype, void* stream);
int cot_radix_mix_backward_apply_lay(const void* gout, const void* attn, const void* ggapT, void* gy, void* gk, int N, int C,
                                     int HW, int lay, int dtype, void* stream);
/* ---- BatchNorm + SiLU of the aggregation's output folded into the radix tail (round 6; replaces `x = self.bn(x); x = self.act(x)` +
 * the tail, models/cotnet.py:89-104, as ONE read of the raw aggregation output per consumer).  y = silu(bn(a)) has two readers, the
 * pooled descriptor and the radix mix; both take `a` and the BatchNorm's statistics and form y as they load (rounded to the storage
 * type, bit-identical to cot_bn_act_forward's output), so y is never written.  `lay` as for the cot_radix_*_lay calls (a in y's place;
 * backward_apply_bn: 1 gout, 2 a and ga, 4 gk).  COT_F32 / COT_BF16.
 *   cot_bn_batch_stats               the statistics of cot_bn_act_forward alone: save_mean / save_rstd [C], running statistics and
 *                                    num_batches_tracked updated as there (bit-identical to its streaming kernels' values).
 *                                    workspace: cot_bn_act_workspace(N, C) floats
 *   cot_bn_stats_sums                chunk sums of x about a per-channel shift into `workspace` (cot_bn_act_workspace(N, C) floats, 16-byte
 *                                    aligned): what cot_radix_gap_t_bn's prologue finalizes -- statistics without a finalize launch
 *   cot_radix_gap_t_bn               gapT[c][n] = mean_hw(silu(bn(a)) + k).  workspace != NULL (cot_bn_stats_sums' output): the prologue
 *                                    finalizes the statistics and save_mean / save_rstd / the running statistics / num_batches_tracked
 *                                    are WRITTEN as by cot_bn_act_forward; NULL: save_mean / save_rstd are read (cot_bn_batch_stats,
 *                                    cot_bn_rowstats_finalize)
 *   cot_radix_mix_logits_bn          cot_radix_mix_logits on silu(bn(a))
 *   cot_radix_mix_backward_reduce_bn cot_radix_mix_backward_reduce + tsum[c][n][0..3] (fp32, N*C*4 floats, 16-byte aligned) = a0 * sum_hw
 *                                    g*s', sum_hw s', a0 * sum_hw g*s'*xhat, sum_hw s'*xhat with s' = silu'(z), xhat = (a - mean)*rstd: the
 *                                    plane sums the BatchNorm's backward needs and that do not depend on the pooled descriptor's gradient
 *   cot_radix_mix_backward_apply_bn  ga = d loss / d a (through the mix, the pooling, SiLU and the BatchNorm), gk = g*a1 + ggapT/HW;
 *                                    dgamma / dbeta [C] fp32 written (sums over the batch in a fixed order: deterministic) */
/* The statistics pass disappears where the aggregation that writes `a` emits them (models/cotnet.py:88-89: x = self.local_conv(x, w);
 * x = self.bn(x)): cot_agg_forward_rowstats = cot_agg_forward (gn_mean == NULL: w holds the weights) or cot_agg_gn9_forward (gn_mean
 * given: w holds the raw logits) that ALSO writes, per output row (n, c, h), the sum and the sum of squares of the values it stores to
 * rowstats[((n*C + c)*H + h)*2 + {0, 1}] (cot_agg_rowstats_floats(N, C, H) floats, 16-byte aligned); cot_bn_rowstats_finalize turns them
 * into save_mean / save_rstd / the running statistics (fp64 sums in a fixed order: deterministic).  COT_BF16, the 3x3 / stride 1 / pad 1
 * one-head geometry with at most 8 channels per weight channel; COT_ERR_UNSUPPORTED otherwise (then: cot_agg_forward + cot_bn_batch_stats). */
int64_t cot_agg_rowstats_floats(int N, int C, int H);
int cot_agg_forward_rowstats(const void* x, const void* w, void* out, float* rowstats, const float* gn_mean, const float* gn_rstd,
                             const void* gn_gamma, const void* gn_beta, int groups_per_image, const cot_agg_geom* g, int dtype,
                             void* stream);
int cot_bn_rowstats_finalize(const float* rowstats, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, int N, int C, int H, int W, float eps, float momentum, void* stream);
int cot_bn_batch_stats(const void* x, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* workspace, int N, int C, int HW, float eps, float momentum, int dtype,
                       void* stream);
int cot_bn_stats_sums(const void* x, float* workspace, int N, int C, int HW, int dtype, void* stream);
int cot_radix_gap_t_bn(const void* a, const void* k, void* gapT, const float* gamma, const float* beta, float* save_mean,
                       float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* workspace,
                       int N, int C, int HW, float eps, float momentum, int lay, int dtype, void* stream);
int cot_radix_mix_logits_bn(const void* a, const void* k, const void* logitsT, void* out, void* attn, const float* gamma,
                            const float* beta, const float* save_mean, const float* save_rstd, int N, int C, int HW, int lay, int dtype,
                            void* stream);
int cot_radix_mix_backward_reduce_bn(const void* gout, const void* a, const void* k, const void* attn, void* glogitsT, float* tsum,
                                     const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int N, int C,
                                     int HW, int lay, int dtype, void* stream);
int cot_radix_mix_backward_apply_bn(const void* gout, const void* a, const void* attn, const void* ggapT, const float* tsum, void* ga,
                                    void* gk, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                                    float* dgamma, float* dbeta, int N, int C, int HW, int lay, int dtype, void* stream);
int cot_group_norm9_forward_lay(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N,
                                int C, int HW, float eps, int lay, int dtype, void* stream);
int cot_group_norm9_backward_lay(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma,
                                 void* dx, void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int lay, int dtype,
                                 void* stream);
/* The GroupNorm backward is two launches: dx + per-(image, channel) sums into `workspace`, then dgamma / dbeta out of those sums.
 * cot_group_norm9_backward[_lay] with dgamma == dbeta == NULL (COT_BF16) issues the first only; this entry point is the second --
 * for callers that put parameter gradients on another stream (`workspace` must stay untouched until it has run). */
int cot_group_norm9_backward_params(const float* workspace, void* dgamma, void* dbeta, int N, int C, int dtype, void* stream);
/* ---- BatchNorm statistics out of the producing convolution's epilogue (models/cotnet.py:51-62, :228-264: every
 * conv1x1 -> BatchNorm pair of a Bottleneck).  On planes of more than 256 pixels (the 56 x 56 / 28 x 28 stages) the 1x1 kernels
 * can write, per (image, 128-pixel tile, channel), the sum and the sum of squares of the bf16 values they store:
 *   cot_conv1x1_forward_stats   = cot_conv1x1_forward that also fills `stats` (cot_gn9_stats_floats(N, Co, HW) floats); covered
 *                                 where cot_conv1x1_stats_covers(Ci, c1, two_slabs, HW) == 1 (cot_conv1x1_forward_gn9 is this call)
 *   cot_bn_tile_stats_finalize  -> the channels' batch mean / rstd (fp64 sums in a fixed order) + the running-statistics update of
 *                                 nn.BatchNorm2d; replaces the statistics pass over the tensor
 *   cot_bn_act_apply_forward    y = act(gamma * (x - mean_c) * rstd_c + beta [+ residual]) from those statistics, optionally with
 *                                 the ReLU sign mask of cot_bn_act_forward_mask.  The backward is cot_bn_act_backward[_mask]. */
int cot_conv1x1_stats_covers(int Ci, int c1, int two_slabs, int HW);
int cot_conv1x1_forward_stats(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, float* stats,
                              int N, int Ci, int Co, int HW, int dtype, void* stream);
int cot_bn_tile_stats_finalize(const float* stats, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int N, int C, int HW, float eps, float momentum, void* stream);
int cot_bn_act_apply_forward(const void* x, const void* residual, void* y, void* relu_mask, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, int N, int C, int HW, int act, int dtype, void* stream);
/* inference mode (nn.BatchNorm2d.eval()): y = act(gamma*(x - running_mean)/sqrt(running_var + eps) + beta [+ residual]) in one
 * pass; nothing is updated. */
int cot_bn_act_inference(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, int N, int C, int HW, float eps, int act, int dtype,
                         void* stream);

/* Per-launch device timing for bench.py's roofline object.  Between cot_profile_begin() and cot_profile_end()
 * every aggregation kernel (every kernel of the library when the environment has COT_PROFILE_ALL=1) is launched with start/stop events attached to its dispatch (hipExtLaunchKernelGGL),
 * so `ms` is the kernel's execution time on the device -- what rocprofv3 --kernel-trace reports -- free of host
 * launch gaps.  cot_profile_end() synchronises, fills at most max_records records in launch order and returns the
 * number of launches recorded.  Process-global; not meant to be left on in production. */
typedef struct {
    char kernel[48];       /* kernel function name without template arguments */
    int32_t kind;          /* 0 = forward, 1 = backward */
    int32_t flags;         /* backward: bit 0 = gx produced, bit 1 = gw produced */
    int32_t dtype, layout; /* cot_dtype, cot_layout */
    cot_agg_geom geom;
    float ms;
} cot_profile_rec;
int cot_profile_begin(void);
int cot_profile_end(cot_profile_rec* out, int max_records);

#ifdef __cplusplus
}
#endif
#endif /* COTNET_AMD_H */
