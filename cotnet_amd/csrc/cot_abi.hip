// cot_abi.hip -- extern "C" entry points of libcotnet_hip.so (declared in include/cotnet_amd.h):
// argument validation, dtype/layout dispatch, error strings.  No torch types anywhere.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include <string>

#include "cot_common.h"

namespace cot {

static thread_local char g_err[512] = "";
static const char* g_kernel = "";  // diagnostic; backward runs on autograd worker threads, so not thread_local

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int g_dry_run = 0;
static thread_local std::string t_dry_log;
void dry_note(const char* where, const char* kernel, dim3 grid, dim3 block, size_t shmem) {
    char buf[768];
    snprintf(buf, sizeof(buf), "%s | %s | grid %u x %u | block %u | lds %zu\n", kernel, where, grid.x, grid.y, block.x, shmem);
    t_dry_log += buf;
}

int check_launch(const char* what) {
    if (g_dry_run) return COT_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(COT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return COT_OK;
}

namespace prof {
struct Rec {
    cot_profile_rec pub;
    hipEvent_t e0, e1;
};
static std::atomic<bool> g_on{false};
static std::mutex g_mu;
static std::vector<Rec> g_recs;
static thread_local size_t t_mark = 0;  // records [t_mark, size) were created by the current ABI call on this thread

static std::atomic<bool> g_all{false};  // COT_PROFILE_ALL=1: time every kernel of the library, not only the aggregation
bool enabled() { return g_on.load(std::memory_order_relaxed); }
bool enabled_for(const char* kernel) {
    if (!g_on.load(std::memory_order_relaxed)) return false;
    return g_all.load(std::memory_order_relaxed) || strstr(kernel, "agg_") != nullptr;
}
void begin_launch(hipEvent_t* e0, hipEvent_t* e1) {
    (void)hipEventCreate(e0);
    (void)hipEventCreate(e1);
}
void end_launch(const char* name, hipEvent_t e0, hipEvent_t e1) {
    Rec r;
    memset(&r.pub, 0, sizeof(r.pub));
    // "(agg_fwd_nchw_k3_lds<T, P, 0>)" -> "agg_fwd_nchw_k3_lds"
    const char* b = name;
    while (*b == '(' || *b == ' ') ++b;
    size_t n = strcspn(b, "<)");
    if (n >= sizeof(r.pub.kernel)) n = sizeof(r.pub.kernel) - 1;
    memcpy(r.pub.kernel, b, n);
    r.e0 = e0;
    r.e1 = e1;
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(r);
}
static void mark() {
    std::lock_guard<std::mutex> lk(g_mu);
    t_mark = g_recs.size();
}
static void annotate(const cot_agg_geom& g, int dtype, int layout, int kind, int flags) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = t_mark; i < g_recs.size(); ++i) {
        g_recs[i].pub.geom = g;
        g_recs[i].pub.dtype = dtype;
        g_recs[i].pub.layout = layout;
        g_recs[i].pub.kind = kind;
        g_recs[i].pub.flags = flags;
    }
}
// non-aggregation ops (kh == 0 marks them): kind 10/11/12 = 1x1 convolution forward / data gradient / weight gradient,
// 13/14/15 = grouped 3x3, 20/21 = BatchNorm(+act) forward / backward; geom.N = N, .C = input channels, .W = output channels,
// .H = pixels per image, .heads = groups, flags = op-specific (BatchNorm: bit 0 residual, bit 1 saved output read)
static void annotate_op(int kind, int N, int Ci, int Co, int HW, int groups, int dtype, int flags) {
    cot_agg_geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.C = Ci; g.W = Co; g.H = HW; g.heads = groups;
    annotate(g, dtype, 0, kind, flags);
}
}  // namespace prof

// implemented in agg_nchw.hip / agg_nhwc.hip / agg_mix.hip
template <typename T>
int agg_forward_nchw(const T*, const T*, T*, const cot_agg_geom&, int, int, hipStream_t, const char*);
template <typename T>
int agg_backward_nchw(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, int, int, hipStream_t);
template <typename T> int agg_forward_nhwc(const T*, const T*, T*, const cot_agg_geom&, int, int, int, hipStream_t);
template <typename T>
int agg_backward_nhwc(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, int, int, int, hipStream_t);
template <typename T>
int aggmix_forward(const T*, const T*, const T*, T*, const cot_agg_geom&, int, int, int, int, hipStream_t);
template <typename T>
int aggmix_backward_input(const T*, const T*, const T*, T*, const cot_agg_geom&, int, int, int, int, int,
                          hipStream_t);
template <typename T>
int aggmix_backward_weight(const T*, const T*, T*, T*, const cot_agg_geom&, int, int, int, int, hipStream_t);
const char* last_kernel_mix();
extern int g_mix_tune[3];  // agg_mix.hip: [0] = 1 generic kernels only, [1] = lanes a tiled workgroup aims for, [2] = pixels per lane
int sgd_flat(void*, void*, void*, const void*, int64_t, float, float, float, float, int, int, int, hipStream_t);
int ema_flat(void*, const void*, int64_t, float, int, hipStream_t);
int bn_workspace_floats(int N, int C);
int bn_act_lay_covers(int N, int C, int HW);
int bn_tile_stats(const float*, int, int, int, float, float, float*, float*, float*, float*, long long*, hipStream_t);
int bn_apply_forward(const void*, const void*, void*, const float*, const float*, const float*, const float*, int, int, int, int, hipStream_t);
int bn_act_forward_lay(const void*, const void*, void*, void*, const float*, const float*, float*, float*, float*, float*, long long*, int,
                       int, int, float, float, int, const float*, int, hipStream_t);
int bn_act_backward_lay(const void*, const void*, const void*, const void*, void*, void*, const float*, const float*, const float*,
                        const float*, float*, float*, int, int, int, int, const float*, int, hipStream_t);
int64_t bn_relu_mask_bytes(int N, int C, int HW, int esize);
extern thread_local uint8_t* t_bn_mask;
extern thread_local int t_c3_pack;
extern int g_bn_fold, g_bn_grid_cap, g_bn_small_m, g_bn_split_target;
template <typename T>
int bn_act_forward(const void*, const void*, void*, const float*, const float*, float*, float*, float*, float*, long long*,
                   float*, int, int, int, float, float, int, const float*, hipStream_t);
template <typename T>
int bn_act_inference(const void*, const void*, void*, const float*, const float*, const float*, const float*, int, int, int,
                     float, int, hipStream_t);
template <typename T>
int bn_act_backward(const void*, const void*, const void*, void*, void*, const float*, const float*, const float*,
                    const float*, float*, float*, float*, int, int, int, int, const float*, hipStream_t);
template <typename T>
int agg_softmax_forward_nchw(const T*, const T*, T*, T*, const cot_agg_geom&, hipStream_t);
template <typename T>
int agg_softmax_backward_nchw(const T*, const T*, const T*, T*, T*, const cot_agg_geom&, hipStream_t);
template <typename T> int radix_gap(const void*, const void*, void*, int64_t, int, hipStream_t);
template <typename T> int radix_mix(const void*, const void*, const void*, void*, int64_t, int, hipStream_t);
template <typename T>
int radix_mix_bwd(const void*, const void*, const void*, const void*, void*, void*, void*, int64_t, int, hipStream_t);
// implemented in conv1x1.hip
int conv1x1_gemm(const void*, const void*, int, const void*, const void*, void*, void*, int, int, int, int, int, int, int,
                 hipStream_t);
int conv1x1_wgrad_splits(int N, int M, int J, int HW, int has_bias);
int conv1x1_wgrad(const void*, const void*, const void*, int, void*, void*, float*, int, int, int, int, hipStream_t);
extern int g_conv1x1_tune[4];
extern int g_wgrad_cap_pct;
extern int g_wgrad_lds_cap_pct;
extern int g_bn_chan;
extern int g_bn_chan7;
extern int g_stem_lds;
// implemented in conv_lds.hip
extern int g_conv_lds_tune[3];
extern int g_conv3x3_ring;
extern int g_conv3x3_res;
extern int g_conv3x3_wsingle;
extern int g_conv3x3_cols;
extern int g_conv3x3_perm;
extern int g_conv_flat_ns3;
extern int g_conv_big_fill;
extern int g_bn_chan_rr;
extern int g_conv_big_xswz;
extern int g_gn9_pack;
extern int g_radix_pack7;
extern int g_conv_lds2_tune;
extern int g_conv_k_tail;
extern int g_conv_ablate;
unsigned long long* g_debug_stamps = nullptr;  // DIAGNOSTIC: see cot_debug_stamps
static int g_grouped_tuned = 1;  // tuning key 36: grouped 1x1 convolutions group by group on the tuned kernels
static int g_conv3x3_merge = 1;  // tuning key 37: merged-groups 3x3 weight gradient for group widths off the 8-channel grid
static inline bool g_conv_lds_tune_wgrad_off() { return (g_conv_lds_tune[2] >> 2) & 1; }  // tuning key 17 bit 2 (A/B)
bool conv1x1_lds_covers(int K, int k1, bool two_slabs, int HW);
int conv1x1_lds_gemm(const void*, const void*, int, const void*, int, const void*, void*, void*, int, int, int, int, int, int,
                     hipStream_t, int64_t xs = 0, int64_t ys = 0, float* stats = nullptr, const void* acc_src = nullptr,
                     const void* acc_mask = nullptr);
int gn9_stats_finalize(const float*, float*, float*, int, int, int, float, hipStream_t);
int agg_gn9_forward_nchw(const bf16_t*, const bf16_t*, const float*, const float*, const bf16_t*, const bf16_t*, int, bf16_t*,
                         const cot_agg_geom&, hipStream_t);
int agg_gn9_backward_nchw_dot2(const bf16_t*, const bf16_t*, const bf16_t*, const float*, const float*, const bf16_t*, const bf16_t*, int,
                               bf16_t*, bf16_t*, const cot_agg_geom&, hipStream_t);
int transpose_bf16(const void* src, void* dst, int R, int C, int pack, hipStream_t stream);
bool conv1x1_wgrad_lds_covers(int N, int HW, int M, int J);
int conv1x1_wgrad_lds_splits(int N, int M, int J, int HW, int has_bias);
int conv1x1_wgrad_lds_run(const void*, const void*, const void*, int, void*, void*, float*, int, int, int, int, hipStream_t);
extern int g_wgrad2_tune;  // conv_wgrad2.hip (third-generation weight gradient)
extern int g_pool_tile;    // pool3x3.hip (row-block pooling kernels)
template <typename T> int subsample2(int bwd, const void* a, void* out, int64_t planes, int H, int W, hipStream_t stream);
template <typename T> int avgpool2x2s2(int bwd, const void* a, void* out, int64_t planes, int H, int W, hipStream_t stream);
bool conv1x1_wgrad2_covers(int N, int HW, int M, int J, int k1, bool two_slabs);
int conv1x1_wgrad2_splits(int N, int M, int J, int HW, int has_bias);
int conv1x1_wgrad2_run(const void*, const void*, const void*, int, void*, void*, float*, int, int, int, int, hipStream_t, int sy = 0,
                       int sx = 0);
bool conv3x3g_wgrad2_covers(int N, int Cin, int Cout, int G, int H, int W, int x_guard);
int conv3x3g_wgrad2_splits(int N, int Cin, int Cout, int G, int HW);
int conv3x3g_wgrad2_run(const void*, const void*, void*, const void*, float*, int, int, int, int, int, int, int, hipStream_t);
int conv3x3g_lds_gemm(const void*, const void*, void*, void*, int, int, int, int, int, int, int, int, hipStream_t);
// implemented in stem7x7.hip
int stem7x7_splits(int N, int H, int W);
int stem7x7_forward(const void*, const void*, void*, int, int, int, hipStream_t);
int stem7x7_wgrad(const void*, const void*, void*, float*, int, int, int, hipStream_t);
int stem3x3s2_splits(int N, int H, int W, int Co);
int stem3x3s2_forward(const void*, const void*, void*, int, int, int, int, hipStream_t);
int stem3x3s2_wgrad(const void*, const void*, void*, float*, int, int, int, int, hipStream_t);
// implemented in pool3x3.hip
template <typename T> int pool3x3s2(int, const void*, const void*, void*, int64_t, int, int, hipStream_t);
// implemented in group_norm9.hip
int gn9_forward(const void*, const void*, const void*, void*, float*, float*, int, int, int, float, int, hipStream_t);
int gn9_backward(const void*, const void*, const float*, const float*, const void*, void*, void*, void*, float*, int, int,
                 int, int, hipStream_t);
int gn9_backward_params(const float* workspace, void* dgamma, void* dbeta, int N, int C, hipStream_t stream);
int gn9f_forward(const void*, const void*, const void*, void*, float*, float*, int, int, int, float, hipStream_t);
int gn9f_backward(const void*, const void*, const float*, const float*, const void*, void*, void*, void*, float*, int, int,
                  int, hipStream_t);
// implemented in conv_tiny.hip (one image of <= 256 pixels: the se branch's convolutions over the batch axis)
extern int g_conv_tiny;
int stem7x7_f32_forward(const void*, const void*, void*, int, int, int, hipStream_t);                       // stem7x7_f32.hip
int stem7x7_f32_backward_weight(const void*, const void*, void*, float*, int, int, int, int, hipStream_t);
bool conv_tiny_covers(int N, int Ci, int Co, int HW);
int conv_tiny_forward(const void*, const void*, const void*, void*, int, int, int, hipStream_t);
int conv_tiny_backward_data(const void*, const void*, void*, int, int, int, int, hipStream_t);
int conv_tiny_backward_weight(const void*, const void*, void*, void*, int, int, int, hipStream_t);
// implemented in conv_gen.hip (general grouped 1x1 / 3x3 convolutions, fp32 or bf16, any channel counts)
int convg_forward(const void*, const void*, const void*, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int convg_backward_data(const void*, const void*, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int64_t convg_workspace(int, int, int, int, int, int, int);
int convg_backward_weight(const void*, const void*, void*, void*, float*, int, int, int, int, int, int, int, int,
                          hipStream_t);
// implemented in conv3x3g.hip
int64_t conv3x3g_masks_bytes(int H, int W);
int conv3x3g_masks(void*, int, int, hipStream_t);
int conv3x3g_gemm(const void*, const void*, void*, const void*, int, int, int, int, int, int, int, int, hipStream_t);
int conv3x3g_wgrad_splits(int N, int Cin, int Cout, int G, int HW);
int conv3x3g_wgrad(const void*, const void*, void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
template <typename T> int radix_gap_t(const void*, const void*, void*, int, int, int, int, hipStream_t);
template <typename T> int se_gap(const void*, void*, int64_t, int, hipStream_t);
template <typename T> int se_gate(const void*, const void*, void*, int64_t, int, hipStream_t);
template <typename T> int se_gate_bwd(const void*, const void*, const void*, void*, void*, int64_t, int, hipStream_t);
template <typename T>
int radix_mix_logits(const void*, const void*, const void*, void*, void*, int, int, int, int, hipStream_t);
template <typename T>
int radix_mix_bwd_reduce(const void*, const void*, const void*, const void*, void*, int, int, int, int, hipStream_t);
template <typename T>
int radix_mix_bwd_apply(const void*, const void*, const void*, void*, void*, int, int, int, int, hipStream_t);
template <typename T>
int radix_gap_t_bn(const void*, const void*, void*, const float*, const float*, float*, float*, float*, float*, long long*, const float*,
                   int, float, float, int, int, int, int, hipStream_t);
int bn_stats_split(int N, int C);
template <typename T> int bn_stats_sums_launch(const void*, float*, int, int, int, hipStream_t);
template <typename T>
int radix_mix_logits_bn(const void*, const void*, const void*, void*, void*, const float*, const float*, const float*, const float*, int,
                        int, int, int, hipStream_t);
template <typename T>
int radix_mix_bwd_reduce_bn(const void*, const void*, const void*, const void*, void*, float*, const float*, const float*, const float*,
                            const float*, int, int, int, int, hipStream_t);
template <typename T>
int radix_mix_bwd_apply_bn(const void*, const void*, const void*, const void*, const float*, void*, void*, const float*, const float*,
                           const float*, const float*, float*, float*, int, int, int, int, hipStream_t);
int agg_forward_rowstats_nchw(const bf16_t*, const bf16_t*, bf16_t*, float*, const float*, const float*, const bf16_t*, const bf16_t*, int,
                              const cot_agg_geom&, hipStream_t);
int bn_rowstats(const float*, int, int, int, int, float, float, float*, float*, float*, float*, long long*, hipStream_t);
template <typename T>
int bn_batch_stats(const void*, float*, float*, float*, float*, long long*, float*, int, int, int, float, float, hipStream_t);
int input_normalize(const void*, void*, const float*, const float*, int64_t, int, int, int, hipStream_t);  // input_norm.hip
const char* last_kernel_nchw();
const char* last_kernel_nhwc();
int set_tuning_nchw(int key, int value);
int set_tuning_dot2(int key, int value);
int xchg_mode();

static int out_size(int in, int k, int s, int p, int d) {
    // python: int((in + 2p - (d(k-1)+1)) / s + 1) -- float division, truncation toward zero
    double v = (double)(in + 2 * p - (d * (k - 1) + 1)) / (double)s + 1.0;
    return (int)v;
}

static int validate(const cot_agg_geom* g, int* Ho, int* Wo) {
    if (!g) return set_error(COT_ERR_INVALID_ARG, "geometry pointer is NULL");
    if (g->N <= 0 || g->C <= 0 || g->H <= 0 || g->W <= 0 || g->heads <= 0 || g->wC <= 0)
        return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d C=%d H=%d W=%d heads=%d wC=%d", g->N, g->C,
                         g->H, g->W, g->heads, g->wC);
    if (g->C % g->wC != 0)  // aggregation_zeropad.py:189
        return set_error(COT_ERR_INVALID_ARG, "input channels %d not divisible by weight channels %d", g->C, g->wC);
    if (g->kh <= 0 || g->kw <= 0 || g->sh <= 0 || g->sw <= 0 || g->dh <= 0 || g->dw <= 0 || g->ph < 0 || g->pw < 0)
        return set_error(COT_ERR_INVALID_ARG, "bad kernel/stride/dilation/padding");
    *Ho = out_size(g->H, g->kh, g->sh, g->ph, g->dh);
    *Wo = out_size(g->W, g->kw, g->sw, g->pw, g->dw);
    if (*Ho <= 0 || *Wo <= 0) return set_error(COT_ERR_INVALID_ARG, "empty output %dx%d", *Ho, *Wo);
    return COT_OK;
}

// largest power-of-two element count (<= 8) such that every pointer is aligned to that many elements
static int align_elems(size_t esize, std::initializer_list<const void*> ptrs) {
    int best = 8;
    for (const void* p : ptrs) {
        if (!p) continue;
        uintptr_t a = (uintptr_t)p;
        int v = 8;
        while (v > 1 && (a % (v * esize)) != 0) v >>= 1;
        if (v < best) best = v;
    }
    return best;
}

template <typename T>
static int fwd_t(const void* x, const void* w, void* out, const cot_agg_geom& g, int Ho, int Wo, int layout,
                 hipStream_t s) {
    const int av = align_elems(sizeof(T), {x, w, out});
    if (layout == COT_NCHW) {
        // entry points require 16-byte aligned base pointers, so every P-wide row vector is aligned
        int rc = agg_forward_nchw<T>((const T*)x, (const T*)w, (T*)out, g, Ho, Wo, s, "");
        g_kernel = last_kernel_nchw();
        return rc;
    }
    int rc = agg_forward_nhwc<T>((const T*)x, (const T*)w, (T*)out, g, Ho, Wo, av, s);
    g_kernel = last_kernel_nhwc();
    return rc;
}

template <typename T>
static int bwd_t(const void* gout, const void* x, const void* w, void* gx, void* gw, const cot_agg_geom& g, int Ho,
                 int Wo, int layout, hipStream_t s) {
    const int av = align_elems(sizeof(T), {gout, x, w, gx, gw});
    int rc;
    if (layout == COT_NCHW) {
        rc = agg_backward_nchw<T>((const T*)gout, (const T*)x, (const T*)w, (T*)gx, (T*)gw, g, Ho, Wo, s);
        g_kernel = last_kernel_nchw();
    } else {
        rc = agg_backward_nhwc<T>((const T*)gout, (const T*)x, (const T*)w, (T*)gx, (T*)gw, g, Ho, Wo, av, s);
        g_kernel = last_kernel_nhwc();
    }
    return rc;
}

}  // namespace cot

using namespace cot;

#define DISPATCH_DTYPE(dtype, CALL)                                                   \
    switch (dtype) {                                                                  \
        case COT_F32: { typedef float T; return CALL; }                               \
        case COT_F64: { typedef double T; return CALL; }                              \
        case COT_BF16: { typedef bf16_t T; return CALL; }                             \
        case COT_F16: { typedef f16_t T; return CALL; }                               \
        default: return set_error(COT_ERR_UNSUPPORTED, "unknown dtype %d", dtype);    \
    }

static int check_align16(std::initializer_list<const void*> ptrs) {
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p % 16) != 0)
            return set_error(COT_ERR_INVALID_ARG, "device pointer %p is not 16-byte aligned", p);
    return COT_OK;
}

extern "C" {

int cot_abi_version(void) { return COTNET_AMD_ABI_VERSION; }
const char* cot_last_error(void) { return g_err; }
const char* cot_last_kernel(void) { return g_kernel; }
const char* cot_status_string(int status) {
    switch (status) {
        case COT_OK: return "ok";
        case COT_ERR_INVALID_ARG: return "invalid argument";
        case COT_ERR_UNSUPPORTED: return "unsupported dtype/layout";
        case COT_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

int cot_agg_out_size(int in, int k, int s, int p, int d) { return out_size(in, k, s, p, d); }

int cot_set_tuning(int key, int value) {
    if (key >= 9 && key <= 11) {
        g_conv1x1_tune[key - 9] = value;
        return COT_OK;
    }
    if (key == 12) {
        g_bn_fold = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 14) {
        g_conv1x1_tune[3] = value;
        return COT_OK;
    }
    if (key >= 15 && key <= 17) {
        g_conv_lds_tune[key - 15] = value;
        return COT_OK;
    }
    if (key == 38) {
        g_conv3x3_ring = value >= 5 ? 5 : 3;
        return COT_OK;
    }
    if (key == 39) {
        g_conv3x3_res = value == 2 ? 2 : (value ? 1 : 0);
        return COT_OK;
    }
    if (key == 18) {
        g_bn_small_m = value > 0 ? value : 0;
        return COT_OK;
    }
    if (key == 13) {
        g_bn_grid_cap = value > 0 ? value : 4096;
        return COT_OK;
    }
    if (key == 22) {
        g_conv_tiny = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 23) {
        g_conv_lds2_tune = value;
        return COT_OK;
    }
    if (key == 26) {  // dry run: record launches instead of issuing them (see cot_launch_log)
        g_dry_run = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 25) {
        g_wgrad2_tune = value;
        return COT_OK;
    }
    if (key == 27) {
        g_pool_tile = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 28) {
        g_bn_split_target = value > 0 ? value : 1024;
        return COT_OK;
    }
    if (key == 24) {  // DIAGNOSTIC: timing ablations of the third-generation 1x1 kernel (results become wrong)
        g_conv_ablate = value;
        return COT_OK;
    }
    if (key == 45) {
        g_conv3x3_perm = value == 2 ? 2 : (value ? 1 : 0);
        return COT_OK;
    }
    if (key == 44) {
        g_conv3x3_cols = value == 2 ? 2 : (value ? 1 : 0);
        return COT_OK;
    }
    if (key == 43) {
        g_conv_flat_ns3 = value == 2 ? 2 : (value ? 1 : 0);
        return COT_OK;
    }
    if (key == 46) {
        g_conv_big_fill = value > 0 ? value : 0;
        return COT_OK;
    }
    if (key == 47) {
        g_bn_chan_rr = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 48) {
        g_conv_big_xswz = value & 7;
        return COT_OK;
    }
    if (key == 49) {
        g_gn9_pack = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 50) {
        g_radix_pack7 = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 51) {
        g_mix_tune[0] = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 52) {
        g_mix_tune[1] = value >= 64 ? value : 256;
        return COT_OK;
    }
    if (key == 53) {
        g_mix_tune[2] = value == 1 || value == 2 || value == 4 ? value : 0;
        return COT_OK;
    }
    if (key == 42) {
        g_conv3x3_wsingle = value == 2 ? 2 : (value ? 1 : 0);
        return COT_OK;
    }
    if (key == 41) {
        g_stem_lds = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 40) {
        g_bn_chan7 = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 21) {
        g_bn_chan = value > 0 ? value : 0;  // 0 off, 1 on, 256 / 512 / 1024: on with that workgroup size (A/B)
        return COT_OK;
    }
    if (key == 20) {
        g_wgrad_lds_cap_pct = value > 0 ? value : 0;
        return COT_OK;
    }
    if (key == 19) {
        g_wgrad_cap_pct = value > 0 ? value : 0;
        return COT_OK;
    }
    if (key == 37) {
        g_conv3x3_merge = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 36) {
        g_grouped_tuned = value ? 1 : 0;
        return COT_OK;
    }
    if (key == 54) {
        g_conv_k_tail = value ? 1 : 0;
        return COT_OK;
    }
    if (key >= 29 && key <= 34) {
        set_tuning_dot2(key - 29, value);
        return COT_OK;
    }
    if (set_tuning_nchw(key, value) != 0) return set_error(COT_ERR_INVALID_ARG, "unknown tuning key %d", key);
    return COT_OK;
}
int cot_xchg_mode(void) { return g_dry_run ? 0 : xchg_mode(); }
int cot_launch_log(char* buf, int cap) {
    if (!buf || cap <= 0) return (int)t_dry_log.size();
    const int n = (int)std::min<size_t>(t_dry_log.size(), (size_t)cap - 1);
    memcpy(buf, t_dry_log.data(), n);
    buf[n] = 0;
    t_dry_log.clear();
    return n;
}
/* DIAGNOSTIC (not in the header's contract): device buffer of 8 x uint64 per workgroup the instrumented kernels (third-generation
 * weight gradient) write s_memtime stamps into; NULL switches it off */
int cot_debug_stamps(void* device_buffer) {
    g_debug_stamps = (unsigned long long*)device_buffer;
    return COT_OK;
}

int cot_sgd_step(void* param, void* master, void* momentum_buf, const void* grad, int64_t n, float lr, float momentum,
                 float weight_decay, float grad_scale, int nesterov, int param_dtype, int grad_dtype, void* stream) {
    if (!param || !momentum_buf || !grad) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (n <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive element count %lld", (long long)n);
    int rc = check_align16({param, master, momentum_buf, grad});
    if (rc) return rc;
    return sgd_flat(param, master, momentum_buf, grad, n, lr, momentum, weight_decay, grad_scale, nesterov, param_dtype,
                    grad_dtype, (hipStream_t)stream);
}

int cot_ema_step(void* ema, const void* src, int64_t n, float decay, int src_dtype, void* stream) {
    if (!ema || !src) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (n <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive element count %lld", (long long)n);
    if (!(decay >= 0.f && decay <= 1.f)) return set_error(COT_ERR_INVALID_ARG, "decay %g outside [0, 1]", (double)decay);
    int rc = check_align16({ema, src});
    if (rc) return rc;
    return ema_flat(ema, src, n, decay, src_dtype, (hipStream_t)stream);
}

static int conv1x1_validate(int N, int Ci, int Co, int HW, int c1, bool split, int dtype, int kdim) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || HW <= 0)
        return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d Ci=%d Co=%d HW=%d", N, Ci, Co, HW);
    if (split ? (c1 <= 0 || c1 >= Ci) : (c1 != Ci))
        return set_error(COT_ERR_INVALID_ARG, "channel split c1=%d does not fit Ci=%d (second slab %s)", c1, Ci,
                         split ? "given" : "NULL");
    if (dtype == COT_F32) {  // general kernels (conv_gen.hip): one tensor, any channel counts
        if (split)
            return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_*: COT_F32 takes one input tensor (concatenate the slabs first)");
        return COT_OK;
    }
    if (dtype != COT_BF16)
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_*: COT_BF16 or COT_F32 tensors (dtype %d given)", dtype);
    if (kdim % 8 != 0)
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_*: reduction channel count %d is not a multiple of 8", kdim);
    return COT_OK;
}

int64_t cot_conv1x1_workspace(int N, int Ci, int Co, int HW, int has_bias) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || HW <= 0) return 0;
    const int64_t wt = (int64_t)Ci * Co * 2;
    int splits = conv1x1_wgrad_splits(N, Co, Ci, HW, has_bias);
    if (conv1x1_wgrad_lds_covers(N, HW, Co, Ci)) splits = std::max(splits, conv1x1_wgrad_lds_splits(N, Co, Ci, HW, has_bias));
    splits = std::max(splits, conv1x1_wgrad2_splits(N, Co, Ci, HW, has_bias));  // (whichever kernel ends up running)
    const int64_t part = (int64_t)splits * Co * (Ci + (has_bias ? 1 : 0)) * 4;
    return ((wt > part ? wt : part) + 255) / 256 * 256;
}

static int cot_conv1x1_forward_impl(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, int N,
                        int Ci, int Co, int HW, int dtype, void* stream) {
    int rc = conv1x1_validate(N, Ci, Co, HW, c1, x2 != nullptr, dtype, Ci);
    if (rc) return rc;
    if (!x1 || !weight || !y) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x1, x2, weight, y}))) return rc;
    if (dtype == COT_F32) return convg_forward(x1, weight, bias, y, N, Ci, Co, 1, HW, 1, 1, 0, dtype, (hipStream_t)stream);
    if (!x2 && conv_tiny_covers(N, Ci, Co, HW)) return conv_tiny_forward(x1, weight, bias, y, Ci, Co, HW, (hipStream_t)stream);
    if (conv1x1_lds_covers(Ci, c1, x2 != nullptr, HW)) {
        rc = conv1x1_lds_gemm(x1, x2, c1, weight, 0, bias, y, nullptr, Co, N, Ci, Co, HW, 0, (hipStream_t)stream);
        if (rc != -1) return rc;
    }
    return conv1x1_gemm(x1, x2, c1, weight, bias, y, nullptr, Co, N, Ci, Co, HW, 0, 0, (hipStream_t)stream);
}
int cot_conv1x1_forward(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, int N,
                        int Ci, int Co, int HW, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv1x1_forward_impl(x1, x2, c1, weight, bias, y, N, Ci, Co, HW, dtype, stream);
    if (p) prof::annotate_op(10, N, Ci, Co, HW, 1, dtype, 0);
    return rc;
}

/* ---- GroupNorm-9 fused into its neighbours (SURVEY 7.6 / VERDICT r3 J1): statistics out of the producing convolution's
 * epilogue, normalisation in the aggregation's prologue.  See include/cotnet_amd.h. */
int64_t cot_gn9_stats_floats(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return (int64_t)N * ceil_div(HW, 128) * C * 2;
}
int cot_gn9_fused_covers(int Ci, int c1, int two_slabs, int HW, int W) {
    const bool w_ok = W == 56 || W == 28 || W == 14 || W == 40 || W == 20 || W == 10;  // (the packed dot-product backward's widths)
    return (HW > 256 && HW % 8 == 0 && w_ok && HW % W == 0 && conv1x1_lds_covers(Ci, c1, two_slabs != 0, HW)) ? 1 : 0;
}
int cot_conv1x1_forward_stats(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, float* stats,
                              int N, int Ci, int Co, int HW, int dtype, void* stream) {
    int rc = conv1x1_validate(N, Ci, Co, HW, c1, x2 != nullptr, dtype, Ci);
    if (rc) return rc;
    if (!x1 || !weight || !y || !stats) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x1, x2, weight, y, stats}))) return rc;
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_forward_stats: COT_BF16 only");
    if (!cot_conv1x1_stats_covers(Ci, c1, x2 != nullptr, HW))
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_forward_stats: geometry not covered (cot_conv1x1_stats_covers)");
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = conv1x1_lds_gemm(x1, x2, c1, weight, 0, bias, y, nullptr, Co, N, Ci, Co, HW, 0, (hipStream_t)stream, 0, 0, stats);
    if (p) prof::annotate_op(10, N, Ci, Co, HW, 1, dtype, 0);
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_forward_stats: geometry not covered (cot_conv1x1_stats_covers)");
    return rc;
}
int cot_conv1x1_stats_covers(int Ci, int c1, int two_slabs, int HW) {
    return (HW > 256 && HW % 8 == 0 && conv1x1_lds_covers(Ci, c1, two_slabs != 0, HW)) ? 1 : 0;
}
int cot_conv1x1_forward_gn9(const void* x1, const void* x2, int c1, const void* weight, const void* bias, void* y, float* stats,
                            int N, int Ci, int Co, int HW, int dtype, void* stream) {
    if (Co % 9 != 0) return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_forward_gn9: output channels a multiple of 9 (Co = %d)", Co);
    return cot_conv1x1_forward_stats(x1, x2, c1, weight, bias, y, stats, N, Ci, Co, HW, dtype, stream);
}
int cot_bn_tile_stats_finalize(const float* stats, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int N, int C, int HW, float eps, float momentum, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0) return set_error(COT_ERR_INVALID_ARG, "bad geometry N=%d C=%d HW=%d", N, C, HW);
    if (!stats || !save_mean || !save_rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    return bn_tile_stats(stats, N, C, HW, eps, momentum, save_mean, save_rstd, running_mean, running_var, (long long*)num_batches_tracked,
                         (hipStream_t)stream);
}
int cot_gn9_stats_finalize(const float* stats, float* mean, float* rstd, int N, int C, int HW, float eps, void* stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || C % 9 != 0) return set_error(COT_ERR_INVALID_ARG, "bad geometry N=%d C=%d HW=%d", N, C, HW);
    if (!stats || !mean || !rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    return gn9_stats_finalize(stats, mean, rstd, N, C, HW, eps, (hipStream_t)stream);
}
int cot_agg_gn9_forward(const void* x, const void* logits, const float* mean, const float* rstd, const void* gamma, const void* beta,
                        int groups_per_image, void* out, const cot_agg_geom* g, int dtype, void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !logits || !mean || !rstd || !gamma || !beta || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, logits, out}))) return rc;
    if (dtype != COT_BF16 || groups_per_image <= 0) return set_error(COT_ERR_UNSUPPORTED, "cot_agg_gn9_*: COT_BF16 only");
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = agg_gn9_forward_nchw((const bf16_t*)x, (const bf16_t*)logits, mean, rstd, (const bf16_t*)gamma, (const bf16_t*)beta,
                              groups_per_image, (bf16_t*)out, *g, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_agg_gn9_forward: geometry not covered (3x3, stride 1, pad 1, one head)");
    else g_kernel = last_kernel_nchw();
    if (p) prof::annotate(*g, dtype, COT_NCHW, 0, 0);  // (same algorithmic bytes as the plain aggregation: x, logits read, out written)
    return rc;
}
// ---- aggregation forward that also emits the following BatchNorm's row statistics (agg_fwd_nchw_k3_lds<ST = 1>), and their finalize
int64_t cot_agg_rowstats_floats(int N, int C, int H) { return (N > 0 && C > 0 && H > 0) ? (int64_t)N * C * H * 2 : 0; }
int cot_agg_forward_rowstats(const void* x, const void* w, void* out, float* rowstats, const float* gn_mean, const float* gn_rstd,
                             const void* gn_gamma, const void* gn_beta, int groups_per_image, const cot_agg_geom* g, int dtype,
                             void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w || !out || !rowstats) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (gn_mean && (!gn_rstd || !gn_gamma || !gn_beta || groups_per_image <= 0))
        return set_error(COT_ERR_INVALID_ARG, "GroupNorm prologue: rstd / gamma / beta / groups_per_image missing");
    if ((rc = check_align16({x, w, out, rowstats}))) return rc;
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_agg_forward_rowstats: COT_BF16 only");
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = agg_forward_rowstats_nchw((const bf16_t*)x, (const bf16_t*)w, (bf16_t*)out, rowstats, gn_mean, gn_rstd, (const bf16_t*)gn_gamma,
                                   (const bf16_t*)gn_beta, groups_per_image, *g, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED)
        return set_error(rc, "cot_agg_forward_rowstats: geometry not covered (3x3, stride 1, pad 1, one head, C / wC <= 8, the LDS kernel's planes)");
    g_kernel = last_kernel_nchw();
    if (p) prof::annotate(*g, dtype, COT_NCHW, 0, 0);
    return rc;
}
int cot_bn_rowstats_finalize(const float* rowstats, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, int N, int C, int H, int W, float eps, float momentum, void* stream) {
    if (!rowstats || !save_mean || !save_rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "bad N/C/H/W");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    return bn_rowstats(rowstats, N, C, H, W, eps, momentum, save_mean, save_rstd, running_mean, running_var, (long long*)num_batches_tracked,
                       (hipStream_t)stream);
}
int cot_agg_gn9_backward(const void* gout, const void* x, const void* logits, const float* mean, const float* rstd, const void* gamma,
                         const void* beta, int groups_per_image, void* gx, void* gw, const cot_agg_geom* g, int dtype, void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!gout || !x || !logits || !mean || !rstd || !gamma || !beta || !gx || !gw) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, x, logits, gx, gw}))) return rc;
    if (dtype != COT_BF16 || groups_per_image <= 0) return set_error(COT_ERR_UNSUPPORTED, "cot_agg_gn9_*: COT_BF16 only");
    const bool k3 = g->kh == 3 && g->kw == 3 && g->sh == 1 && g->sw == 1 && g->ph == 1 && g->pw == 1 && g->dh == 1 && g->dw == 1;
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = k3 ? agg_gn9_backward_nchw_dot2((const bf16_t*)gout, (const bf16_t*)x, (const bf16_t*)logits, mean, rstd, (const bf16_t*)gamma,
                                         (const bf16_t*)beta, groups_per_image, (bf16_t*)gx, (bf16_t*)gw, *g, (hipStream_t)stream)
            : -1;
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_agg_gn9_backward: geometry not covered (cot_gn9_fused_covers)");
    g_kernel = "agg_bwd_nchw_k3_dot2<gx,gw,gn9>";
    if (p) prof::annotate(*g, dtype, COT_NCHW, 1, 3);
    return rc;
}

static int cot_conv1x1_backward_data_impl(const void* gy, const void* weight, void* gx1, void* gx2, int c1, int accumulate,
                              void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream) {
    int rc = conv1x1_validate(N, Ci, Co, HW, c1, gx2 != nullptr, dtype, Co);
    if (rc) return rc;
    if (!gy || !weight || !gx1 || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, weight, gx1, gx2, workspace}))) return rc;
    if (dtype == COT_F32)
        return convg_backward_data(gy, weight, gx1, N, Ci, Co, 1, HW, 1, 1, accumulate & 1, dtype, (hipStream_t)stream);
    if (!gx2 && conv_tiny_covers(N, Ci, Co, HW))
        return conv_tiny_backward_data(gy, weight, gx1, Ci, Co, HW, accumulate & 1, (hipStream_t)stream);
    if (conv1x1_lds_covers(Co, Co, false, HW)) {
        // the LDS forward kernel on dY; its weight operand is W^T, which IS the [Co][Ci] weight tensor read as [K][M] (WT
        // kernels: transposing LDS reads) -- tuning key 17 bit 5 brings back the round-2-mid form (a transposed, K-step-major
        // copy [Co/32][Ci][32] written into the workspace by a launch of its own) for A/B
        if ((g_conv_lds_tune[2] >> 5) & 1) {
            if ((rc = transpose_bf16(weight, workspace, Co, Ci, 1, (hipStream_t)stream))) return rc;
            rc = conv1x1_lds_gemm(gy, nullptr, Co, workspace, 1, nullptr, gx1, gx2, c1, N, Co, Ci, HW, accumulate & 3,
                                  (hipStream_t)stream);
        } else {
            rc = conv1x1_lds_gemm(gy, nullptr, Co, weight, 2, nullptr, gx1, gx2, c1, N, Co, Ci, HW, accumulate & 3,
                                  (hipStream_t)stream);
        }
        if (rc != -1) return rc;
    }
    // the forward kernel on dY with A = weight^T, read in place from the [Co][Ci] weight tensor
    return conv1x1_gemm(gy, nullptr, Co, weight, nullptr, gx1, gx2, c1, N, Co, Ci, HW, accumulate & 3, 1,
                        (hipStream_t)stream);
}
// ---- conv1's data gradient of an identity-shortcut Bottleneck with the residual's gradient folded in (models/cotnet.py:228-264: out =
// relu(bn3(...) + x)): gx = W^T gy + gout * [block output > 0], the second term from the block's upstream gradient and bn3's sign mask
// (cot_bn_act_forward_mask) -- bn3's backward (cot_bn_act_backward_mask with dresidual = NULL) then does not write the residual's gradient
// and this call does not read it back: one write and most of one read of a [N, Cin, HW] tensor per block less.
int cot_conv1x1_backward_data_relu_res_covers(int N, int Ci, int Co, int HW, int dtype) {
    if (dtype != COT_BF16 || N <= 0 || Ci <= 0 || Co <= 0 || HW <= 0 || HW % 8 != 0 || Ci % 8 != 0 || Co % 8 != 0) return 0;
    if (conv_tiny_covers(N, Ci, Co, HW) || !conv1x1_lds_covers(Co, Co, false, HW)) return 0;
    const int64_t slab = (int64_t)std::max(Ci, Co) * HW * 2;  // (conv1x1_lds_gemm2's 32-bit lane offsets)
    if ((HW <= 256 ? slab * (256 / HW + 1) : slab) >= ((int64_t)1 << 31) || (int64_t)Ci * Co * 2 >= ((int64_t)1 << 31)) return 0;
    if (cot_bn_relu_mask_bytes(N, Ci, HW, COT_BF16) <= 0) return 0;
    return 1;
}
int cot_conv1x1_backward_data_relu_res(const void* gy, const void* weight, void* gx, const void* gout, const void* relu_mask, int N, int Ci,
                                       int Co, int HW, int dtype, void* stream) {
    if (!gy || !weight || !gx || !gout || !relu_mask) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (!cot_conv1x1_backward_data_relu_res_covers(N, Ci, Co, HW, dtype))
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_backward_data_relu_res: N=%d %d->%d HW=%d is off the LDS kernels (_covers)", N, Ci, Co, HW);
    int rc = check_align16({gy, weight, gx, gout});
    if (rc) return rc;
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = conv1x1_lds_gemm(gy, nullptr, Co, weight, 2, nullptr, gx, nullptr, Ci, N, Co, Ci, HW, 0, (hipStream_t)stream, 0, 0, nullptr, gout,
                          relu_mask);
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1_backward_data_relu_res: the kernel refused N=%d %d->%d HW=%d", N, Ci, Co, HW);
    if (p) prof::annotate_op(11, N, Ci, Co, HW, 1, dtype, 0);
    return rc;
}

int cot_conv1x1_backward_data(const void* gy, const void* weight, void* gx1, void* gx2, int c1, int accumulate,
                              void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv1x1_backward_data_impl(gy, weight, gx1, gx2, c1, accumulate, workspace, N, Ci, Co, HW, dtype, stream);
    if (p) prof::annotate_op(11, N, Ci, Co, HW, 1, dtype, 0);
    return rc;
}

static int cot_conv1x1_backward_weight_impl(const void* gy, const void* x1, const void* x2, int c1, void* gweight, void* gbias,
                                void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream) {
    int rc = conv1x1_validate(N, Ci, Co, HW, c1, x2 != nullptr, dtype, 8);
    if (rc) return rc;
    if (!gy || !x1 || !gweight || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, x1, x2, gweight, workspace}))) return rc;
    if (dtype == COT_F32)
        return convg_backward_weight(gy, x1, gweight, gbias, (float*)workspace, N, Ci, Co, 1, HW, 1, 1, dtype,
                                     (hipStream_t)stream);
    if (!x2 && HW % 8 == 0 && conv_tiny_covers(N, Ci, Co, HW))
        return conv_tiny_backward_weight(gy, x1, gweight, gbias, Ci, Co, HW, (hipStream_t)stream);
    if (conv1x1_wgrad2_covers(N, HW, Co, Ci, c1, x2 != nullptr)) {
        rc = conv1x1_wgrad2_run(gy, x1, x2, c1, gweight, gbias, (float*)workspace, N, Ci, Co, HW, (hipStream_t)stream);
        if (rc != -1) return rc;
    }
    if (conv1x1_wgrad_lds_covers(N, HW, Co, Ci) && !g_conv_lds_tune_wgrad_off())
        return conv1x1_wgrad_lds_run(gy, x1, x2, c1, gweight, gbias, (float*)workspace, N, Ci, Co, HW, (hipStream_t)stream);
    return conv1x1_wgrad(gy, x1, x2, c1, gweight, gbias, (float*)workspace, N, Ci, Co, HW, (hipStream_t)stream);
}
int cot_conv1x1_backward_weight(const void* gy, const void* x1, const void* x2, int c1, void* gweight, void* gbias,
                                void* workspace, int N, int Ci, int Co, int HW, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv1x1_backward_weight_impl(gy, x1, x2, c1, gweight, gbias, workspace, N, Ci, Co, HW, dtype, stream);
    if (p) prof::annotate_op(12, N, Ci, Co, HW, 1, dtype, 0);
    return rc;
}

static int conv3x3g_validate(int N, int Cin, int Cout, int G, int H, int W, int dtype, int kdim_per_group) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || G <= 0 || H <= 0 || W <= 0)
        return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d Cin=%d Cout=%d groups=%d H=%d W=%d", N, Cin,
                         Cout, G, H, W);
    if (Cin % G != 0 || Cout % G != 0)
        return set_error(COT_ERR_INVALID_ARG, "channels %d -> %d not divisible by groups %d", Cin, Cout, G);
    if (dtype != COT_BF16 && dtype != COT_F32)
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_*: COT_BF16 or COT_F32 tensors (dtype %d given)", dtype);
    (void)kdim_per_group;  // (channel counts the MFMA-32 kernels do not tile go to the general kernels of conv_gen.hip)
    return COT_OK;
}
// the specialised bf16 kernels take groups whose channel counts are multiples of 8; everything else (and fp32) is conv_gen's
static bool conv3x3g_general(int Cin, int Cout, int G, int dtype) {
    return dtype == COT_F32 || (Cin / G) % 8 != 0 || (Cout / G) % 8 != 0;
}

// Weight gradient of a grouped 3x3 convolution whose group width is off the MFMA kernels' 8-channel grid (CoXtLayer.key_embed at
// dim 96: 8 groups of 12, models/cotnet.py:113-117): f neighbouring groups are MERGED into one of f*Kc channels the tuned kernels
// tile, the gradient of that wider convolution is taken -- the wanted blocks are its diagonal, the f*(f-1) cross blocks are wasted
// work on a kernel that is still >10x faster than the general 64 x 64-tile kernel on 12 x 12 blocks (1137 -> ~90 us at N64 x 96 x
// 56 x 56) -- and the diagonal blocks are copied out.  f = the smallest of 2 / 4 / 8 that puts both widths on the grid; 0 = none.
static int conv3x3g_merge_factor(int Cin, int Cout, int G) {
    for (int f = 2; f <= 8; f *= 2)
        if (G % f == 0 && !conv3x3g_general(Cin, Cout, G / f, COT_BF16)) return f;
    return 0;
}
// gw[co][ci][t] = wide[co][(g % f) * Kc + ci][t], g = co / Mg  (wide: [Cout][f*Kc][9] of the merged convolution)
__global__ __launch_bounds__(256) void conv3x3g_diag_blocks(const bf16_t* __restrict__ wide, bf16_t* __restrict__ gw, int Kc, int Mg,
                                                            int f, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i % 9, ci = (i / 9) % Kc, co = i / (9 * Kc);
    gw[i] = wide[((int64_t)co * (f * Kc) + ((co / Mg) % f) * Kc + ci) * 9 + t];
}

int64_t cot_conv3x3g_masks_bytes(int H, int W) { return (H > 0 && W > 0) ? conv3x3g_masks_bytes(H, W) : 0; }

int cot_conv3x3g_masks(void* masks, int H, int W, void* stream) {
    if (H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive image size %dx%d", H, W);
    if (!masks) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    int rc = check_align16({masks});
    if (rc) return rc;
    return conv3x3g_masks(masks, H, W, (hipStream_t)stream);
}

int64_t cot_conv3x3g_workspace(int N, int Cin, int Cout, int groups, int H, int W) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || groups <= 0 || H <= 0 || W <= 0 || Cin % groups || Cout % groups) return 0;
    // repacked weights of the LDS kernels (10 taps: one of zeros; K padded to whole 32-channel chunks), forward or data gradient
    const int64_t kpf = ((Cin / groups) + 31) / 32 * 32, kpd = ((Cout / groups) + 31) / 32 * 32;
    const int64_t wb = std::max((int64_t)Cout * kpf, (int64_t)Cin * kpd) * 10 * 2;
    int splits = conv3x3g_wgrad_splits(N, Cin, Cout, groups, H * W);
    if (conv3x3g_wgrad2_covers(N, Cin, Cout, groups, H, W, 1 << 30))  // (whichever kernel ends up running)
        splits = std::max(splits, conv3x3g_wgrad2_splits(N, Cin, Cout, groups, H * W));
    const int64_t part = (int64_t)splits * Cout * (Cin / groups) * 9 * 4;
    int64_t ws = ((wb > part ? wb : part) + 255) / 256 * 256;
    if (conv3x3g_general(Cin, Cout, groups, COT_BF16)) {
        ws = std::max(ws, convg_workspace(N, Cin, Cout, groups, H, W, 3));
        const int f = conv3x3g_merge_factor(Cin, Cout, groups);
        if (f)  // merged-groups weight gradient: the wide gradient (bf16) followed by the merged convolution's own workspace
            ws = std::max(ws, ((int64_t)Cout * (Cin / groups) * f * 9 * 2 + 255) / 256 * 256 + cot_conv3x3g_workspace(N, Cin, Cout, groups / f, H, W));
    }
    return ws;
}

static int cot_conv3x3g_forward_impl(const void* x, const void* weight, void* y, const void* masks, void* workspace, int N, int Cin,
                         int Cout, int groups, int H, int W, int dtype, void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, Cin / (groups > 0 ? groups : 1));
    if (rc) return rc;
    if (!x || !weight || !y || !masks || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, weight, y, masks, workspace}))) return rc;
    if (conv3x3g_general(Cin, Cout, groups, dtype))
        return convg_forward(x, weight, nullptr, y, N, Cin, Cout, groups, H, W, 3, 0, dtype, (hipStream_t)stream);
    rc = conv3x3g_lds_gemm(x, weight, y, workspace, N, Cin, Cout, groups, H, W, 0, 0, (hipStream_t)stream);
    if (rc != -1) return rc;
    return conv3x3g_gemm(x, weight, y, masks, N, Cin, Cout, groups, H, W, 0, 0, (hipStream_t)stream);
}
int cot_conv3x3g_forward(const void* x, const void* weight, void* y, const void* masks, void* workspace, int N, int Cin,
                         int Cout, int groups, int H, int W, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv3x3g_forward_impl(x, weight, y, masks, workspace, N, Cin, Cout, groups, H, W, dtype, stream);
    if (p) prof::annotate_op(13, N, Cin, Cout, H * W, groups, dtype, 0);
    return rc;
}

static int cot_conv3x3g_backward_data_impl(const void* gy, const void* weight, void* gx, int accumulate, const void* masks,
                               void* workspace, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                               void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, Cout / (groups > 0 ? groups : 1));
    if (rc) return rc;
    if (!gy || !weight || !gx || !masks || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, weight, gx, masks, workspace}))) return rc;
    if (conv3x3g_general(Cin, Cout, groups, dtype))
        return convg_backward_data(gy, weight, gx, N, Cin, Cout, groups, H, W, 3, accumulate ? 1 : 0, dtype,
                                   (hipStream_t)stream);
    // Dense convolutions on rows wider than 128 pixels (SE-CoTNetD's deep stem at 160 x 160, models/cotnet_hybrid.py:359-368): an
    // LDS tile of the per-step ring holds ONE output row between two halo rows there, and the data gradient -- K = Cout up to 128
    // channels deep -- runs faster on the first-generation register-ring kernel: 64 -> 64 986 -> 560 us, 64 -> 128 1784 -> 1183 us
    // at B = 64 (the forward does not: 1012 vs 1140, 1237 vs 2058 us; profiles/r06_deep_stem_kernel_choice.log)
    const bool wide_dense = groups == 1 && W > 128;
    if (!wide_dense) {
        rc = conv3x3g_lds_gemm(gy, weight, gx, workspace, N, Cin, Cout, groups, H, W, 1, accumulate ? 1 : 0, (hipStream_t)stream);
        if (rc != -1) return rc;
    }
    return conv3x3g_gemm(gy, weight, gx, masks, N, Cin, Cout, groups, H, W, 1, accumulate ? 1 : 0, (hipStream_t)stream);
}
int cot_conv3x3g_backward_data(const void* gy, const void* weight, void* gx, int accumulate, const void* masks,
                               void* workspace, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                               void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv3x3g_backward_data_impl(gy, weight, gx, accumulate, masks, workspace, N, Cin, Cout, groups, H, W, dtype, stream);
    if (p) prof::annotate_op(14, N, Cin, Cout, H * W, groups, dtype, 0);
    return rc;
}

// ---- packings made ahead of time (round 5): cot_conv3x3g_pack fills `packed` with exactly what the forward (mode 0) / data-gradient
// (mode 1) call of the same geometry would pack into its workspace, the _packed entry points run on it without packing.
struct C3PackScope {
    explicit C3PackScope(int m) { t_c3_pack = m; }
    ~C3PackScope() { t_c3_pack = 0; }
};
int64_t cot_conv3x3g_packed_bytes(int Cin, int Cout, int groups) {
    if (Cin <= 0 || Cout <= 0 || groups <= 0 || Cin % groups || Cout % groups) return 0;
    const int64_t kpf = ((Cin / groups) + 31) / 32 * 32, kpd = ((Cout / groups) + 31) / 32 * 32;
    return (std::max((int64_t)Cout * kpf, (int64_t)Cin * kpd) * 10 * 2 + 255) / 256 * 256;
}
int cot_conv3x3g_pack(const void* weight, void* packed, int mode, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                      void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, (mode ? Cout : Cin) / (groups > 0 ? groups : 1));
    if (rc) return rc;
    if (!weight || !packed || (mode != 0 && mode != 1)) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer / mode not 0 | 1");
    if ((rc = check_align16({weight, packed}))) return rc;
    if (dtype != COT_BF16 || conv3x3g_general(Cin, Cout, groups, dtype))
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_pack: this convolution does not run on packed weights (general kernels)");
    C3PackScope scope(1);
    // (x / y are not touched in pack-only mode: the planner needs the geometry alone)
    rc = conv3x3g_lds_gemm(packed, weight, packed, packed, N, Cin, Cout, groups, H, W, mode, 0, (hipStream_t)stream);
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_pack: geometry off the LDS kernels (they gather the weights in place)");
    return rc;
}
int cot_conv3x3g_forward_packed(const void* x, const void* packed, void* y, int N, int Cin, int Cout, int groups, int H, int W, int dtype,
                                void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, Cin / (groups > 0 ? groups : 1));
    if (rc) return rc;
    if (!x || !packed || !y) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, packed, y}))) return rc;
    if (dtype != COT_BF16 || conv3x3g_general(Cin, Cout, groups, dtype)) return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_forward_packed: general kernels");
    const bool p = prof::enabled();
    if (p) prof::mark();
    {
        C3PackScope scope(2);
        rc = conv3x3g_lds_gemm(x, packed, y, const_cast<void*>(packed), N, Cin, Cout, groups, H, W, 0, 0, (hipStream_t)stream);
    }
    if (p) prof::annotate_op(13, N, Cin, Cout, H * W, groups, dtype, 0);
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_forward_packed: geometry off the LDS kernels");
    return rc;
}
int cot_conv3x3g_backward_data_packed(const void* gy, const void* packed, void* gx, int accumulate, int N, int Cin, int Cout, int groups,
                                      int H, int W, int dtype, void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, Cout / (groups > 0 ? groups : 1));
    if (rc) return rc;
    if (!gy || !packed || !gx) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, packed, gx}))) return rc;
    if (dtype != COT_BF16 || conv3x3g_general(Cin, Cout, groups, dtype)) return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_backward_data_packed: general kernels");
    const bool p = prof::enabled();
    if (p) prof::mark();
    {
        C3PackScope scope(2);
        rc = conv3x3g_lds_gemm(gy, packed, gx, const_cast<void*>(packed), N, Cin, Cout, groups, H, W, 1, accumulate ? 1 : 0, (hipStream_t)stream);
    }
    if (p) prof::annotate_op(14, N, Cin, Cout, H * W, groups, dtype, 0);
    if (rc == -1) return set_error(COT_ERR_UNSUPPORTED, "cot_conv3x3g_backward_data_packed: geometry off the LDS kernels");
    return rc;
}

static int cot_conv3x3g_backward_weight_impl(const void* gy, const void* x, void* gweight, const void* masks, void* workspace, int N,
                                 int Cin, int Cout, int groups, int H, int W, int dtype, int x_guard, void* stream) {
    int rc = conv3x3g_validate(N, Cin, Cout, groups, H, W, dtype, 8);
    if (rc) return rc;
    if (!gy || !x || !gweight || !masks || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, x, gweight, masks, workspace}))) return rc;
    if (conv3x3g_general(Cin, Cout, groups, dtype)) {
        const int f = (dtype == COT_BF16 && g_conv3x3_merge) ? conv3x3g_merge_factor(Cin, Cout, groups) : 0;
        if (f) {
            const int Kc = Cin / groups, Mg = Cout / groups;
            bf16_t* wide = (bf16_t*)workspace;
            char* rest = (char*)workspace + ((int64_t)Cout * Kc * f * 9 * 2 + 255) / 256 * 256;
            rc = conv3x3g_wgrad2_run(gy, x, wide, masks, (float*)rest, N, Cin, Cout, groups / f, H, W, x_guard, (hipStream_t)stream);
            if (rc < 0) rc = conv3x3g_wgrad(gy, x, wide, masks, (float*)rest, N, Cin, Cout, groups / f, H, W, (hipStream_t)stream);
            if (rc) return rc;
            const int total = Cout * Kc * 9;
            COT_LAUNCH(conv3x3g_diag_blocks, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wide,
                       (bf16_t*)gweight, Kc, Mg, f, total);
            return check_launch("conv3x3g_diag_blocks");
        }
        return convg_backward_weight(gy, x, gweight, nullptr, (float*)workspace, N, Cin, Cout, groups, H, W, 3, dtype,
                                     (hipStream_t)stream);
    }
    // the LDS-staged kernel reads X at pixel + tap offset through whole 16-byte pieces: it needs W + 1 readable elements
    // before and behind the tensor (x_guard: what the caller guarantees)
    rc = conv3x3g_wgrad2_run(gy, x, gweight, masks, (float*)workspace, N, Cin, Cout, groups, H, W, x_guard, (hipStream_t)stream);
    if (rc >= 0) return rc;
    return conv3x3g_wgrad(gy, x, gweight, masks, (float*)workspace, N, Cin, Cout, groups, H, W, (hipStream_t)stream);
}
int cot_conv3x3g_backward_weight_guarded(const void* gy, const void* x, void* gweight, const void* masks, void* workspace, int N,
                                         int Cin, int Cout, int groups, int H, int W, int dtype, int x_guard_elems, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_conv3x3g_backward_weight_impl(gy, x, gweight, masks, workspace, N, Cin, Cout, groups, H, W, dtype,
                                                     x_guard_elems < 0 ? 0 : x_guard_elems, stream);
    if (p) prof::annotate_op(15, N, Cin, Cout, H * W, groups, dtype, 0);
    return rc;
}
int cot_conv3x3g_backward_weight(const void* gy, const void* x, void* gweight, const void* masks, void* workspace, int N,
                                 int Cin, int Cout, int groups, int H, int W, int dtype, void* stream) {
    return cot_conv3x3g_backward_weight_guarded(gy, x, gweight, masks, workspace, N, Cin, Cout, groups, H, W, dtype, 0, stream);
}

int cot_agg_softmax_forward(const void* x, const void* logits, void* out, void* probs, const cot_agg_geom* g, int dtype,
                            void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !logits || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, logits, out, probs}))) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COT_F32: rc = agg_softmax_forward_nchw<float>((const float*)x, (const float*)logits, (float*)out, (float*)probs, *g, s); break;
        case COT_BF16: rc = agg_softmax_forward_nchw<bf16_t>((const bf16_t*)x, (const bf16_t*)logits, (bf16_t*)out, (bf16_t*)probs, *g, s); break;
        case COT_F16: rc = agg_softmax_forward_nchw<f16_t>((const f16_t*)x, (const f16_t*)logits, (f16_t*)out, (f16_t*)probs, *g, s); break;
        default: rc = COT_ERR_UNSUPPORTED;
    }
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "fused window-softmax aggregation: geometry/dtype not covered (compose softmax + cot_agg_forward)");
    else g_kernel = last_kernel_nchw();
    return rc;
}

int cot_agg_softmax_backward(const void* gout, const void* x, const void* probs, void* gx, void* glogits,
                             const cot_agg_geom* g, int dtype, void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!gout || !x || !probs || !gx || !glogits) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, x, probs, gx, glogits}))) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case COT_F32: rc = agg_softmax_backward_nchw<float>((const float*)gout, (const float*)x, (const float*)probs, (float*)gx, (float*)glogits, *g, s); break;
        case COT_BF16: rc = agg_softmax_backward_nchw<bf16_t>((const bf16_t*)gout, (const bf16_t*)x, (const bf16_t*)probs, (bf16_t*)gx, (bf16_t*)glogits, *g, s); break;
        case COT_F16: rc = agg_softmax_backward_nchw<f16_t>((const f16_t*)gout, (const f16_t*)x, (const f16_t*)probs, (f16_t*)gx, (f16_t*)glogits, *g, s); break;
        default: rc = COT_ERR_UNSUPPORTED;
    }
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "fused window-softmax aggregation backward: geometry/dtype not covered");
    else g_kernel = last_kernel_nchw();
    return rc;
}

static int tail_check(int64_t planes, int HW, int dtype) {
    if (planes <= 0 || HW <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive planes/HW");
    if (dtype != COT_F32 && dtype != COT_BF16)
        return set_error(COT_ERR_UNSUPPORTED, "radix tail: dtype %d (float32 / bfloat16 only)", dtype);
    return COT_OK;
}

int cot_radix_gap(const void* y, const void* k, void* gap, int64_t planes, int HW, int dtype, void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!y || !k || !gap) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({y, k}))) return rc;
    return dtype == COT_F32 ? radix_gap<float>(y, k, gap, planes, HW, (hipStream_t)stream)
                            : radix_gap<bf16_t>(y, k, gap, planes, HW, (hipStream_t)stream);
}

int cot_se_gap(const void* x, void* gap, int64_t planes, int HW, int dtype, void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!x || !gap) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x}))) return rc;
    return dtype == COT_F32 ? se_gap<float>(x, gap, planes, HW, (hipStream_t)stream)
                            : se_gap<bf16_t>(x, gap, planes, HW, (hipStream_t)stream);
}
int cot_se_gate(const void* x, const void* logit, void* out, int64_t planes, int HW, int dtype, void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!x || !logit || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, out}))) return rc;
    return dtype == COT_F32 ? se_gate<float>(x, logit, out, planes, HW, (hipStream_t)stream)
                            : se_gate<bf16_t>(x, logit, out, planes, HW, (hipStream_t)stream);
}
int cot_se_gate_backward(const void* g, const void* x, const void* logit, void* gx, void* glogit, int64_t planes, int HW,
                         int dtype, void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!g || !x || !logit || !gx || !glogit) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({g, x, gx}))) return rc;
    return dtype == COT_F32 ? se_gate_bwd<float>(g, x, logit, gx, glogit, planes, HW, (hipStream_t)stream)
                            : se_gate_bwd<bf16_t>(g, x, logit, gx, glogit, planes, HW, (hipStream_t)stream);
}

int cot_radix_mix(const void* y, const void* k, const void* attn, void* out, int64_t planes, int HW, int dtype,
                  void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!y || !k || !attn || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({y, k, out}))) return rc;
    return dtype == COT_F32 ? radix_mix<float>(y, k, attn, out, planes, HW, (hipStream_t)stream)
                            : radix_mix<bf16_t>(y, k, attn, out, planes, HW, (hipStream_t)stream);
}

int cot_radix_mix_backward(const void* gout, const void* y, const void* k, const void* attn, void* gy, void* gk,
                           void* gattn, int64_t planes, int HW, int dtype, void* stream) {
    int rc = tail_check(planes, HW, dtype);
    if (rc) return rc;
    if (!gout || !y || !k || !attn || !gy || !gk || !gattn) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, y, k, gy, gk}))) return rc;
    return dtype == COT_F32 ? radix_mix_bwd<float>(gout, y, k, attn, gy, gk, gattn, planes, HW, (hipStream_t)stream)
                            : radix_mix_bwd<bf16_t>(gout, y, k, attn, gy, gk, gattn, planes, HW, (hipStream_t)stream);
}

int cot_radix_gap_t(const void* y, const void* k, void* gapT, int N, int C, int HW, int dtype, void* stream) {
    return cot_radix_gap_t_lay(y, k, gapT, N, C, HW, 0, dtype, stream);
}
int cot_radix_gap_t_lay(const void* y, const void* k, void* gapT, int N, int C, int HW, int lay, int dtype, void* stream) {
    int rc = tail_check((int64_t)N * C, HW, dtype);
    if (rc) return rc;
    if (N <= 0 || C <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive N=%d C=%d", N, C);
    if (!y || !gapT) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");  // k == NULL: pool y alone
    if ((rc = check_align16({y, k}))) return rc;
    return dtype == COT_F32 ? radix_gap_t<float>(y, k, gapT, N, C, HW, lay, (hipStream_t)stream)
                            : radix_gap_t<bf16_t>(y, k, gapT, N, C, HW, lay, (hipStream_t)stream);
}

int cot_radix_mix_logits(const void* y, const void* k, const void* logitsT, void* out, void* attn, int N, int C, int HW,
                         int dtype, void* stream) {
    return cot_radix_mix_logits_lay(y, k, logitsT, out, attn, N, C, HW, 0, dtype, stream);
}
int cot_radix_mix_logits_lay(const void* y, const void* k, const void* logitsT, void* out, void* attn, int N, int C, int HW, int lay,
                             int dtype, void* stream) {
    int rc = tail_check((int64_t)N * C, HW, dtype);
    if (rc) return rc;
    if (N <= 0 || C <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive N=%d C=%d", N, C);
    if (!y || !k || !logitsT || !out || !attn) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({y, k, out}))) return rc;
    return dtype == COT_F32 ? radix_mix_logits<float>(y, k, logitsT, out, attn, N, C, HW, lay, (hipStream_t)stream)
                            : radix_mix_logits<bf16_t>(y, k, logitsT, out, attn, N, C, HW, lay, (hipStream_t)stream);
}

int cot_radix_mix_backward_reduce(const void* gout, const void* y, const void* k, const void* attn, void* glogitsT, int N,
                                  int C, int HW, int dtype, void* stream) {
    return cot_radix_mix_backward_reduce_lay(gout, y, k, attn, glogitsT, N, C, HW, 0, dtype, stream);
}
int cot_radix_mix_backward_reduce_lay(const void* gout, const void* y, const void* k, const void* attn, void* glogitsT, int N,
                                      int C, int HW, int lay, int dtype, void* stream) {
    int rc = tail_check((int64_t)N * C, HW, dtype);
    if (rc) return rc;
    if (N <= 0 || C <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive N=%d C=%d", N, C);
    if (!gout || !y || !k || !attn || !glogitsT) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, y, k}))) return rc;
    return dtype == COT_F32 ? radix_mix_bwd_reduce<float>(gout, y, k, attn, glogitsT, N, C, HW, lay, (hipStream_t)stream)
                            : radix_mix_bwd_reduce<bf16_t>(gout, y, k, attn, glogitsT, N, C, HW, lay, (hipStream_t)stream);
}

int cot_radix_mix_backward_apply(const void* gout, const void* attn, const void* ggapT, void* gy, void* gk, int N, int C,
                                 int HW, int dtype, void* stream) {
    return cot_radix_mix_backward_apply_lay(gout, attn, ggapT, gy, gk, N, C, HW, 0, dtype, stream);
}
int cot_radix_mix_backward_apply_lay(const void* gout, const void* attn, const void* ggapT, void* gy, void* gk, int N, int C,
                                     int HW, int lay, int dtype, void* stream) {
    int rc = tail_check((int64_t)N * C, HW, dtype);
    if (rc) return rc;
    if (N <= 0 || C <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive N=%d C=%d", N, C);
    if (!gout || !attn || !ggapT || !gy || !gk) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, gy, gk}))) return rc;
    return dtype == COT_F32 ? radix_mix_bwd_apply<float>(gout, attn, ggapT, gy, gk, N, C, HW, lay, (hipStream_t)stream)
                            : radix_mix_bwd_apply<bf16_t>(gout, attn, ggapT, gy, gk, N, C, HW, lay, (hipStream_t)stream);
}

// ---- BatchNorm + SiLU of the aggregation's output folded into the radix tail (radix_tail.hip; models/cotnet.py:88-104)
int cot_bn_batch_stats(const void* x, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* workspace, int N, int C, int HW, float eps, float momentum, int dtype,
                       void* stream) {
    if (!x || !save_mean || !save_rstd || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    int rc = check_align16({x});
    if (rc) return rc;
    if (dtype == COT_F32)
        return bn_batch_stats<float>(x, save_mean, save_rstd, running_mean, running_var, (long long*)num_batches_tracked, workspace,
                                     N, C, HW, eps, momentum, (hipStream_t)stream);
    if (dtype == COT_BF16)
        return bn_batch_stats<bf16_t>(x, save_mean, save_rstd, running_mean, running_var, (long long*)num_batches_tracked, workspace,
                                      N, C, HW, eps, momentum, (hipStream_t)stream);
    return set_error(COT_ERR_UNSUPPORTED, "cot_bn_batch_stats: dtype %d (float32 / bfloat16 only)", dtype);
}
static int tail_bn_check(int N, int C, int HW, int dtype, const char* what) {
    int rc = tail_check((int64_t)N * C, HW, dtype);
    if (rc) return rc;
    if (N <= 0 || C <= 0) return set_error(COT_ERR_INVALID_ARG, "%s: non-positive N=%d C=%d", what, N, C);
    return COT_OK;
}
int cot_bn_stats_sums(const void* x, float* workspace, int N, int C, int HW, int dtype, void* stream) {
    if (!x || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW");
    int rc = check_align16({x, workspace});
    if (rc) return rc;
    if (dtype == COT_F32) return bn_stats_sums_launch<float>(x, workspace, N, C, HW, (hipStream_t)stream);
    if (dtype == COT_BF16) return bn_stats_sums_launch<bf16_t>(x, workspace, N, C, HW, (hipStream_t)stream);
    return set_error(COT_ERR_UNSUPPORTED, "cot_bn_stats_sums: dtype %d (float32 / bfloat16 only)", dtype);
}
int cot_radix_gap_t_bn(const void* a, const void* k, void* gapT, const float* gamma, const float* beta, float* save_mean,
                       float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* workspace,
                       int N, int C, int HW, float eps, float momentum, int lay, int dtype, void* stream) {
    int rc = tail_bn_check(N, C, HW, dtype, "cot_radix_gap_t_bn");
    if (rc) return rc;
    if (!a || !k || !gapT || !gamma || !beta || !save_mean || !save_rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    if ((rc = check_align16({a, k, workspace}))) return rc;
    const int split = workspace ? bn_stats_split(N, C) : 0;
    return dtype == COT_F32 ? radix_gap_t_bn<float>(a, k, gapT, gamma, beta, save_mean, save_rstd, running_mean, running_var,
                                                    (long long*)num_batches_tracked, workspace, split, eps, momentum, N, C, HW, lay,
                                                    (hipStream_t)stream)
                            : radix_gap_t_bn<bf16_t>(a, k, gapT, gamma, beta, save_mean, save_rstd, running_mean, running_var,
                                                     (long long*)num_batches_tracked, workspace, split, eps, momentum, N, C, HW, lay,
                                                     (hipStream_t)stream);
}
int cot_radix_mix_logits_bn(const void* a, const void* k, const void* logitsT, void* out, void* attn, const float* gamma,
                            const float* beta, const float* save_mean, const float* save_rstd, int N, int C, int HW, int lay, int dtype,
                            void* stream) {
    int rc = tail_bn_check(N, C, HW, dtype, "cot_radix_mix_logits_bn");
    if (rc) return rc;
    if (!a || !k || !logitsT || !out || !attn || !gamma || !beta || !save_mean || !save_rstd)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({a, k, out}))) return rc;
    return dtype == COT_F32 ? radix_mix_logits_bn<float>(a, k, logitsT, out, attn, gamma, beta, save_mean, save_rstd, N, C, HW, lay,
                                                         (hipStream_t)stream)
                            : radix_mix_logits_bn<bf16_t>(a, k, logitsT, out, attn, gamma, beta, save_mean, save_rstd, N, C, HW, lay,
                                                          (hipStream_t)stream);
}
int cot_radix_mix_backward_reduce_bn(const void* gout, const void* a, const void* k, const void* attn, void* glogitsT, float* tsum,
                                     const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, int N, int C,
                                     int HW, int lay, int dtype, void* stream) {
    int rc = tail_bn_check(N, C, HW, dtype, "cot_radix_mix_backward_reduce_bn");
    if (rc) return rc;
    if (!gout || !a || !k || !attn || !glogitsT || !tsum || !gamma || !beta || !save_mean || !save_rstd)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, a, k}))) return rc;
    return dtype == COT_F32 ? radix_mix_bwd_reduce_bn<float>(gout, a, k, attn, glogitsT, tsum, gamma, beta, save_mean, save_rstd, N, C,
                                                             HW, lay, (hipStream_t)stream)
                            : radix_mix_bwd_reduce_bn<bf16_t>(gout, a, k, attn, glogitsT, tsum, gamma, beta, save_mean, save_rstd, N, C,
                                                              HW, lay, (hipStream_t)stream);
}
int cot_radix_mix_backward_apply_bn(const void* gout, const void* a, const void* attn, const void* ggapT, const float* tsum, void* ga,
                                    void* gk, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd,
                                    float* dgamma, float* dbeta, int N, int C, int HW, int lay, int dtype, void* stream) {
    int rc = tail_bn_check(N, C, HW, dtype, "cot_radix_mix_backward_apply_bn");
    if (rc) return rc;
    if (!gout || !a || !attn || !ggapT || !tsum || !ga || !gk || !gamma || !beta || !save_mean || !save_rstd || !dgamma || !dbeta)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gout, a, ga, gk}))) return rc;
    return dtype == COT_F32 ? radix_mix_bwd_apply_bn<float>(gout, a, attn, ggapT, tsum, ga, gk, gamma, beta, save_mean, save_rstd, dgamma,
                                                            dbeta, N, C, HW, lay, (hipStream_t)stream)
                            : radix_mix_bwd_apply_bn<bf16_t>(gout, a, attn, ggapT, tsum, ga, gk, gamma, beta, save_mean, save_rstd,
                                                             dgamma, dbeta, N, C, HW, lay, (hipStream_t)stream);
}

int64_t cot_stem7x7s2_workspace(int N, int H, int W) {
    if (N <= 0) return 0;
    const int S = stem7x7_splits(N, H, W);
    return S ? ((int64_t)S * 64 * 147 * 4 + 255) / 256 * 256 : 0;
}

int cot_stem7x7s2_forward(const void* x, const void* weight, void* y, int N, int H, int W, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d H=%d W=%d", N, H, W);
    if (!x || !weight || !y) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != COT_BF16 && dtype != COT_F32) return set_error(COT_ERR_UNSUPPORTED, "cot_stem7x7s2_*: COT_BF16 or COT_F32 (dtype %d given)", dtype);
    int rc = check_align16({x, weight, y});
    if (rc) return rc;
    if (dtype == COT_F32) {  // the reference's own precision: plain fp32 kernels (stem7x7_f32.hip), the bf16 kernels' geometry
        if (stem7x7_splits(N, H, W) <= 0) return set_error(COT_ERR_UNSUPPORTED, "cot_stem7x7s2_forward: a %dx%d input is not covered", H, W);
        return stem7x7_f32_forward(x, weight, y, N, H, W, (hipStream_t)stream);
    }
    rc = stem7x7_forward(x, weight, y, N, H, W, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_stem7x7s2_forward: output width of a %dx%d input is not a multiple of 8", H, W);
    return rc;
}

int cot_stem7x7s2_backward_weight(const void* gy, const void* x, void* gweight, void* workspace, int N, int H, int W,
                                  int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d H=%d W=%d", N, H, W);
    if (!gy || !x || !gweight || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != COT_BF16 && dtype != COT_F32) return set_error(COT_ERR_UNSUPPORTED, "cot_stem7x7s2_*: COT_BF16 or COT_F32 (dtype %d given)", dtype);
    int rc = check_align16({gy, x, gweight, workspace});
    if (rc) return rc;
    if (dtype == COT_F32) {
        const int S = stem7x7_splits(N, H, W);  // (the workspace is sized for S slices of 64 x 147 floats: cot_stem7x7s2_workspace)
        if (S <= 0) return set_error(COT_ERR_UNSUPPORTED, "cot_stem7x7s2_backward_weight: %dx%d input not covered", H, W);
        rc = stem7x7_f32_backward_weight(gy, x, gweight, (float*)workspace, S, N, H, W, (hipStream_t)stream);
        if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_stem7x7s2_backward_weight: the fp32 kernel's LDS window was refused");
        return rc;
    }
    rc = stem7x7_wgrad(gy, x, gweight, (float*)workspace, N, H, W, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_stem7x7s2_backward_weight: %dx%d input not covered", H, W);
    return rc;
}

int64_t cot_stem3x3s2_workspace(int N, int H, int W, int Cout) {
    if (N <= 0) return 0;
    const int S = stem3x3s2_splits(N, H, W, Cout);
    return S ? ((int64_t)S * Cout * 27 * 4 + 255) / 256 * 256 : 0;
}

int cot_stem3x3s2_forward(const void* x, const void* weight, void* y, int N, int H, int W, int Cout, int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d H=%d W=%d", N, H, W);
    if (!x || !weight || !y) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_stem3x3s2_*: only COT_BF16 (dtype %d given)", dtype);
    int rc = check_align16({x, weight, y});
    if (rc) return rc;
    rc = stem3x3s2_forward(x, weight, y, N, H, W, Cout, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED)
        set_error(rc, "cot_stem3x3s2_forward: %d output channels / the output of a %dx%d input not covered (32 or 64; Wo %% 8 == 0)", Cout, H, W);
    return rc;
}

int cot_stem3x3s2_backward_weight(const void* gy, const void* x, void* gweight, void* workspace, int N, int H, int W, int Cout,
                                  int dtype, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d H=%d W=%d", N, H, W);
    if (!gy || !x || !gweight || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_stem3x3s2_*: only COT_BF16 (dtype %d given)", dtype);
    int rc = check_align16({gy, x, gweight, workspace});
    if (rc) return rc;
    rc = stem3x3s2_wgrad(gy, x, gweight, (float*)workspace, N, H, W, Cout, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_stem3x3s2_backward_weight: %d channels / %dx%d input not covered", Cout, H, W);
    return rc;
}

static int pool_call(int op, const void* a, const void* b, void* out, int64_t planes, int H, int W, int dtype, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive planes/H/W");
    if (!a || !out || (op >= 3 && op <= 5 && !b)) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (op >= 6 && (H < 2 || W < 2)) return set_error(COT_ERR_UNSUPPORTED, "blur pooling reflects by one pixel: H, W >= 2");
    if (dtype == COT_F32) return pool3x3s2<float>(op, a, b, out, planes, H, W, (hipStream_t)stream);
    if (dtype == COT_BF16) return pool3x3s2<bf16_t>(op, a, b, out, planes, H, W, (hipStream_t)stream);
    return set_error(COT_ERR_UNSUPPORTED, "cot_*pool3x3s2_*: dtype %d (float32 / bfloat16 only)", dtype);
}
static int subsample_call(int bwd, const void* a, void* out, int64_t planes, int H, int W, int dtype, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive planes/H/W");
    if (!a || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    int rc = COT_ERR_UNSUPPORTED;
    if (dtype == COT_F32) rc = subsample2<float>(bwd, a, out, planes, H, W, (hipStream_t)stream);
    else if (dtype == COT_BF16) rc = subsample2<bf16_t>(bwd, a, out, planes, H, W, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_subsample2_*: even H and W, float32 / bfloat16 (H %d W %d dtype %d given)", H, W, dtype);
    return rc;
}
int cot_subsample2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream) {
    return subsample_call(0, x, y, planes, H, W, dtype, stream);
}
int cot_subsample2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream) {
    return subsample_call(1, gy, gx, planes, H, W, dtype, stream);
}
static int avgpool2_call(int bwd, const void* a, void* out, int64_t planes, int H, int W, int dtype, void* stream) {
    if (planes <= 0 || H <= 0 || W <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive planes/H/W");
    if (!a || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    int rc = COT_ERR_UNSUPPORTED;
    if (dtype == COT_F32) rc = avgpool2x2s2<float>(bwd, a, out, planes, H, W, (hipStream_t)stream);
    else if (dtype == COT_BF16) rc = avgpool2x2s2<bf16_t>(bwd, a, out, planes, H, W, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_avgpool2x2s2_*: even H and W, float32 / bfloat16 (H %d W %d dtype %d given)", H, W, dtype);
    return rc;
}
int cot_avgpool2x2s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream) {
    return avgpool2_call(0, x, y, planes, H, W, dtype, stream);
}
int cot_avgpool2x2s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream) {
    return avgpool2_call(1, gy, gx, planes, H, W, dtype, stream);
}
int cot_avgpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(0, x, nullptr, y, planes, H, W, dtype, stream);
}
int cot_avgpool3x3s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(1, gy, nullptr, gx, planes, H, W, dtype, stream);
}
int cot_maxpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(2, x, nullptr, y, planes, H, W, dtype, stream);
}
int cot_maxpool3x3s2_backward(const void* gy, const void* x, void* gx, int64_t planes, int H, int W, int dtype,
                              void* stream) {
    return pool_call(3, gy, x, gx, planes, H, W, dtype, stream);
}

int cot_blurpool3x3s2_forward(const void* x, void* y, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(6, x, nullptr, y, planes, H, W, dtype, stream);
}
int cot_blurpool3x3s2_backward(const void* gy, void* gx, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(7, gy, nullptr, gx, planes, H, W, dtype, stream);
}
int cot_maxpool3x3s2_forward_taps(const void* x, void* y, void* taps, int64_t planes, int H, int W, int dtype, void* stream) {
    return pool_call(4, x, taps, y, planes, H, W, dtype, stream);
}
int cot_maxpool3x3s2_backward_taps(const void* gy, const void* taps, void* gx, int64_t planes, int H, int W, int dtype,
                                   void* stream) {
    return pool_call(5, gy, taps, gx, planes, H, W, dtype, stream);
}

// ---- grouped 1x1 convolution (CoXtLayer, groups = 2): the general kernels of conv_gen.hip
static int conv1x1g_validate(int N, int Ci, int Co, int G, int HW, int dtype) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || G <= 0 || HW <= 0)
        return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d Ci=%d Co=%d groups=%d HW=%d", N, Ci, Co, G, HW);
    if (Ci % G != 0 || Co % G != 0)
        return set_error(COT_ERR_INVALID_ARG, "channels %d -> %d not divisible by groups %d", Ci, Co, G);
    if (dtype != COT_BF16 && dtype != COT_F32)
        return set_error(COT_ERR_UNSUPPORTED, "cot_conv1x1g_*: COT_BF16 or COT_F32 tensors (dtype %d given)", dtype);
    return COT_OK;
}

int64_t cot_convg_workspace(int N, int Cin, int Cout, int groups, int H, int W, int ksize) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || groups <= 0 || H <= 0 || W <= 0 || Cin % groups || Cout % groups) return 0;
    int64_t ws = convg_workspace(N, Cin, Cout, groups, H, W, ksize);
    if (ksize == 1 && groups > 1)  // a group's weight gradient may run on the tuned 1x1 kernel: its partial sums (one group at a time)
        ws = std::max(ws, cot_conv1x1_workspace(N, Cin / groups, Cout / groups, H * W, 1));
    return ws;
}

// Grouped 1x1 convolutions on the TUNED kernels (round 4): a group is an ordinary 1x1 convolution on a channel range of x and y,
// i.e. the LDS-pipelined kernels with the image strides of the full tensors (conv_lds2.hip / conv_wgrad2.hip: xs / ys, sy / sx),
// one launch per group.  Taken when the group's reduction depth is on the kernels' 32-channel grid and everything is 16-byte
// aligned; the general kernels of conv_gen.hip (64 x 64 tiles, 2-byte staging loads) keep the rest.  Tuning key 36 = 0: general
// kernels everywhere (A/B).
static inline bool grouped_on_tuned(int Kg, int Mg, int HW, int groups, int dtype) {
    return g_grouped_tuned && dtype == COT_BF16 && groups > 1 && groups <= 8 && Kg % 8 == 0 && Mg % 8 == 0 &&  // (Kg % 32 != 0: the kernels' K tail)
           conv1x1_lds_covers(Kg, Kg, false, HW);
}

int cot_conv1x1g_forward(const void* x, const void* weight, const void* bias, void* y, int N, int Ci, int Co, int groups,
                         int HW, int dtype, void* stream) {
    int rc = conv1x1g_validate(N, Ci, Co, groups, HW, dtype);
    if (rc) return rc;
    if (!x || !weight || !y) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, weight, y}))) return rc;
    const int Kg = Ci / groups, Mg = Co / groups;
    if (grouped_on_tuned(Kg, Mg, HW, groups, dtype)) {
        const bf16_t *xb = (const bf16_t*)x, *wb = (const bf16_t*)weight, *bb = (const bf16_t*)bias;
        bf16_t* yb = (bf16_t*)y;
        for (int g = 0; g < groups; ++g) {
            rc = conv1x1_lds_gemm(xb + (int64_t)g * Kg * HW, nullptr, Kg, wb + (int64_t)g * Mg * Kg, 0, bb ? bb + g * Mg : nullptr,
                                  yb + (int64_t)g * Mg * HW, nullptr, Mg, N, Kg, Mg, HW, 0, (hipStream_t)stream, (int64_t)Ci * HW,
                                  (int64_t)Co * HW);
            if (rc == -1 && g == 0) break;  // not covered after all: the general kernel does the whole convolution
            if (rc) return rc == -1 ? set_error(COT_ERR_LAUNCH, "grouped 1x1: group %d fell off the tuned kernel", g) : rc;
        }
        if (rc == COT_OK) return rc;
    }
    return convg_forward(x, weight, bias, y, N, Ci, Co, groups, HW, 1, 1, 0, dtype, (hipStream_t)stream);
}

int cot_conv1x1g_backward_data(const void* gy, const void* weight, void* gx, int accumulate, int N, int Ci, int Co,
                               int groups, int HW, int dtype, void* stream) {
    int rc = conv1x1g_validate(N, Ci, Co, groups, HW, dtype);
    if (rc) return rc;
    if (!gy || !weight || !gx) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, weight, gx}))) return rc;
    const int Kg = Ci / groups, Mg = Co / groups;
    if (grouped_on_tuned(Mg, Kg, HW, groups, dtype)) {  // (the data gradient reduces over the group's OUTPUT channels)
        const bf16_t *gb = (const bf16_t*)gy, *wb = (const bf16_t*)weight;
        bf16_t* xb = (bf16_t*)gx;
        for (int g = 0; g < groups; ++g) {  // the forward kernel on dY; W^T = the group's [Mg][Kg] block read in place (WT kernels)
            rc = conv1x1_lds_gemm(gb + (int64_t)g * Mg * HW, nullptr, Mg, wb + (int64_t)g * Mg * Kg, 2, nullptr, xb + (int64_t)g * Kg * HW,
                                  nullptr, Kg, N, Mg, Kg, HW, accumulate ? 1 : 0, (hipStream_t)stream, (int64_t)Co * HW, (int64_t)Ci * HW);
            if (rc == -1 && g == 0) break;
            if (rc) return rc == -1 ? set_error(COT_ERR_LAUNCH, "grouped 1x1 data gradient: group %d fell off the tuned kernel", g) : rc;
        }
        if (rc == COT_OK) return rc;
    }
    return convg_backward_data(gy, weight, gx, N, Ci, Co, groups, HW, 1, 1, accumulate ? 1 : 0, dtype, (hipStream_t)stream);
}

int cot_conv1x1g_backward_weight(const void* gy, const void* x, void* gweight, void* gbias, void* workspace, int N, int Ci,
                                 int Co, int groups, int HW, int dtype, void* stream) {
    int rc = conv1x1g_validate(N, Ci, Co, groups, HW, dtype);
    if (rc) return rc;
    if (!gy || !x || !gweight || !workspace) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({gy, x, gweight, workspace}))) return rc;
    const int Kg = Ci / groups, Mg = Co / groups;
    // (any channel counts per group whose slabs start on 16-byte boundaries: the kernel clamps rows past a matrix' end)
    if (g_grouped_tuned && dtype == COT_BF16 && groups > 1 && groups <= 8 && ((int64_t)Kg * HW) % 8 == 0 && ((int64_t)Mg * HW) % 8 == 0 &&
        (Mg * Kg) % 8 == 0 && Kg % 8 == 0 && conv1x1_wgrad2_covers(N, HW, Mg, Kg, Kg, false)) {
        // (the partial sums of group g are consumed by its own reduce launch before group g+1's kernel overwrites them: one stream)
        const bf16_t *gb = (const bf16_t*)gy, *xb = (const bf16_t*)x;
        bf16_t *wg = (bf16_t*)gweight, *bg = (bf16_t*)gbias;
        for (int g = 0; g < groups; ++g) {
            rc = conv1x1_wgrad2_run(gb + (int64_t)g * Mg * HW, xb + (int64_t)g * Kg * HW, nullptr, Kg, wg + (int64_t)g * Mg * Kg,
                                    bg ? bg + g * Mg : nullptr, (float*)workspace, N, Kg, Mg, HW, (hipStream_t)stream, Co, Ci);
            if (rc == -1 && g == 0) break;
            if (rc) return rc == -1 ? set_error(COT_ERR_LAUNCH, "grouped 1x1 weight gradient: group %d fell off the tuned kernel", g) : rc;
        }
        if (rc == COT_OK) return rc;
    }
    return convg_backward_weight(gy, x, gweight, gbias, (float*)workspace, N, Ci, Co, groups, HW, 1, 1, dtype,
                                 (hipStream_t)stream);
}

static int gn9_validate(int N, int C, int HW, int dtype) {
    if (N <= 0 || C <= 0 || HW <= 0) return set_error(COT_ERR_INVALID_ARG, "non-positive dimension N=%d C=%d HW=%d", N, C, HW);
    if (C % 9 != 0) return set_error(COT_ERR_INVALID_ARG, "channel count %d is not a multiple of 9", C);
    if (dtype != COT_BF16 && dtype != COT_F32)
        return set_error(COT_ERR_UNSUPPORTED, "cot_group_norm9_*: COT_BF16 or COT_F32 (dtype %d given)", dtype);
    return COT_OK;
}

int cot_group_norm9_forward(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N,
                            int C, int HW, float eps, int dtype, void* stream) {
    return cot_group_norm9_forward_lay(x, gamma, beta, y, mean, rstd, N, C, HW, eps, 0, dtype, stream);
}
int cot_group_norm9_forward_lay(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int N,
                                int C, int HW, float eps, int lay, int dtype, void* stream) {
    int rc = gn9_validate(N, C, HW, dtype);
    if (lay && dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_group_norm9_*_lay: COT_BF16 only");
    if (rc) return rc;
    if (!x || !gamma || !beta || !y || !mean || !rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({x, y}))) return rc;
    if (dtype == COT_F32) return gn9f_forward(x, gamma, beta, y, mean, rstd, N, C, HW, eps, (hipStream_t)stream);
    rc = gn9_forward(x, gamma, beta, y, mean, rstd, N, C, HW, eps, lay, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_group_norm9_forward: %d pixels per plane exceed one workgroup", HW);
    return rc;
}

int cot_group_norm9_backward(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma,
                             void* dx, void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int dtype,
                             void* stream) {
    return cot_group_norm9_backward_lay(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, N, C, HW, 0, dtype, stream);
}
int cot_group_norm9_backward_lay(const void* dy, const void* x, const float* mean, const float* rstd, const void* gamma,
                                 void* dx, void* dgamma, void* dbeta, float* workspace, int N, int C, int HW, int lay, int dtype,
                                 void* stream) {
    int rc = gn9_validate(N, C, HW, dtype);
    if (lay && dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_group_norm9_*_lay: COT_BF16 only");
    if (rc) return rc;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !workspace || (!dgamma) != (!dbeta) || (!dgamma && dtype != COT_BF16))
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if ((rc = check_align16({dy, x, dx}))) return rc;
    if (dtype == COT_F32)
        return gn9f_backward(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, N, C, HW, (hipStream_t)stream);
    rc = gn9_backward(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, workspace, N, C, HW, lay, (hipStream_t)stream);
    if (rc == COT_ERR_UNSUPPORTED) set_error(rc, "cot_group_norm9_backward: %d pixels per plane exceed one workgroup", HW);
    return rc;
}

int cot_group_norm9_backward_params(const float* workspace, void* dgamma, void* dbeta, int N, int C, int dtype, void* stream) {
    if (!workspace || !dgamma || !dbeta) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_group_norm9_backward_params: COT_BF16 only");
    if (N <= 0 || C <= 0 || C % 9 != 0) return set_error(COT_ERR_INVALID_ARG, "bad N=%d / C=%d", N, C);
    return gn9_backward_params(workspace, dgamma, dbeta, N, C, (hipStream_t)stream);
}

int cot_conv1x1_lds_covers(int K, int k1, int two_slabs, int HW) { return conv1x1_lds_covers(K, k1, two_slabs != 0, HW) ? 1 : 0; }

int cot_input_normalize(const void* x_u8, void* y, const float* mean, const float* stdv, int64_t planes, int C, int HW,
                        int dtype, void* stream) {
    if (!x_u8 || !y || !mean || !stdv) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (planes <= 0 || C <= 0 || HW <= 0 || planes % C != 0)
        return set_error(COT_ERR_INVALID_ARG, "bad planes=%lld / C=%d / HW=%d", (long long)planes, C, HW);
    int rc = check_align16({x_u8, y});
    if (rc) return rc;
    return input_normalize(x_u8, y, mean, stdv, planes, C, HW, dtype, (hipStream_t)stream);
}

int cot_bn_act_workspace(int N, int C) { return (N > 0 && C > 0) ? bn_workspace_floats(N, C) : 0; }

static int cot_bn_act_forward_ps_impl(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                          float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* workspace, const float* sample_scale, int N, int C, int HW,
                          float eps, float momentum, int act, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    int rc = check_align16({x, residual, y});
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == COT_F32)
        return bn_act_forward<float>(x, residual, y, gamma, beta, save_mean, save_rstd, running_mean, running_var,
                                     (long long*)num_batches_tracked, workspace, N, C, HW, eps, momentum, act, sample_scale, s);
    if (dtype == COT_BF16)
        return bn_act_forward<bf16_t>(x, residual, y, gamma, beta, save_mean, save_rstd, running_mean, running_var,
                                      (long long*)num_batches_tracked, workspace, N, C, HW, eps, momentum, act, sample_scale, s);
    return set_error(COT_ERR_UNSUPPORTED, "bn_act: dtype %d (float32 / bfloat16 only)", dtype);
}
int cot_bn_act_forward_ps(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                          float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* workspace, const float* sample_scale, int N, int C, int HW,
                          float eps, float momentum, int act, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_bn_act_forward_ps_impl(x, residual, y, gamma, beta, save_mean, save_rstd, running_mean, running_var, num_batches_tracked, workspace, sample_scale, N, C, HW, eps, momentum, act, dtype, stream);
    if (p) prof::annotate_op(20, N, C, C, HW, 1, dtype, (residual ? 1 : 0));
    return rc;
}
int cot_bn_act_forward(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                       float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float* workspace, int N, int C, int HW, float eps, float momentum,
                       int act, int dtype, void* stream) {
    return cot_bn_act_forward_ps(x, residual, y, gamma, beta, save_mean, save_rstd, running_mean, running_var,
                                 num_batches_tracked, workspace, NULL, N, C, HW, eps, momentum, act, dtype, stream);
}
static int cot_bn_act_backward_ps_impl(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                           const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                           float* workspace, const float* sample_scale, int N, int C, int HW, int act, int dtype, void* stream) {
    if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (act == 1 && !y && dresidual)
        return set_error(COT_ERR_INVALID_ARG, "ReLU backward after a residual add needs the saved output y");
    if (act == 2 && (dresidual || sample_scale))  // (SiLU' needs z = s_n * bn(x) + residual; the kernels recompute bn(x) alone.  The reference has no such block)
        return set_error(COT_ERR_UNSUPPORTED, "SiLU backward after a residual add / a per-sample scale is not covered");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act");
    int rc = check_align16({dy, x, y, dx, dresidual});
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == COT_F32)
        return bn_act_backward<float>(dy, x, y, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta,
                                      workspace, N, C, HW, act, sample_scale, s);
    if (dtype == COT_BF16)
        return bn_act_backward<bf16_t>(dy, x, y, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta,
                                       workspace, N, C, HW, act, sample_scale, s);
    return set_error(COT_ERR_UNSUPPORTED, "bn_act: dtype %d (float32 / bfloat16 only)", dtype);
}
int cot_bn_act_backward_ps(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                           const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                           float* workspace, const float* sample_scale, int N, int C, int HW, int act, int dtype, void* stream) {
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_bn_act_backward_ps_impl(dy, x, y, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta, workspace, sample_scale, N, C, HW, act, dtype, stream);
    if (p) prof::annotate_op(21, N, C, C, HW, 1, dtype, (dresidual ? 1 : 0) | (y ? 2 : 0));
    return rc;
}
int cot_bn_act_backward(const void* dy, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                        const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                        float* workspace, int N, int C, int HW, int act, int dtype, void* stream) {
    return cot_bn_act_backward_ps(dy, x, y, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta, workspace, NULL,
                                  N, C, HW, act, dtype, stream);
}

// ---- per-tensor layouts: the channel-resident bf16 kernels with one layout bit per tensor, a second output and a second
// upstream gradient
int cot_bn_act_lay_covers(int N, int C, int HW, int dtype) {
    return (dtype == COT_BF16 && N > 0 && C > 0 && HW > 0 && bn_act_lay_covers(N, C, HW)) ? 1 : 0;
}
int cot_bn_act_forward_lay(const void* x, const void* residual, void* y, void* y2, const float* gamma, const float* beta, float* save_mean,
                           float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           const float* sample_scale, int N, int C, int HW, float eps, float momentum, int act, int lay, int dtype,
                           void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2 || (lay & ~15)) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act/lay");
    if ((running_mean == NULL) != (running_var == NULL))
        return set_error(COT_ERR_INVALID_ARG, "running_mean and running_var must be given together");
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_*_lay: COT_BF16 only (dtype %d given)", dtype);
    int rc = check_align16({x, residual, y, y2});
    if (rc) return rc;
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = bn_act_forward_lay(x, residual, y, y2, gamma, beta, save_mean, save_rstd, running_mean, running_var, (long long*)num_batches_tracked,
                            N, C, HW, eps, momentum, act, sample_scale, lay, (hipStream_t)stream);
    if (rc == -2) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_forward_lay: N=%d C=%d HW=%d is off the channel-resident kernels (cot_bn_act_lay_covers)", N, C, HW);
    if (p) prof::annotate_op(20, N, C, C, HW, 1, dtype, (residual ? 1 : 0) | (y2 ? 4 : 0));
    return rc;
}
int cot_bn_act_backward_lay(const void* dy, const void* dy2, const void* x, const void* y, void* dx, void* dresidual, const float* gamma,
                            const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                            const float* sample_scale, int N, int C, int HW, int act, int lay, int dtype, void* stream) {
    if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_rstd || !dgamma || !dbeta)
        return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (act == 1 && !y && dresidual) return set_error(COT_ERR_INVALID_ARG, "ReLU backward after a residual add needs the saved output y");
    if (act == 2 && (dresidual || sample_scale))
        return set_error(COT_ERR_UNSUPPORTED, "SiLU backward after a residual add / a per-sample scale is not covered");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2 || (lay & ~63)) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act/lay");
    if (dtype != COT_BF16) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_*_lay: COT_BF16 only (dtype %d given)", dtype);
    int rc = check_align16({dy, dy2, x, y, dx, dresidual});
    if (rc) return rc;
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = bn_act_backward_lay(dy, dy2, x, y, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta, N, C, HW, act, sample_scale, lay,
                             (hipStream_t)stream);
    if (rc == -2) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_backward_lay: N=%d C=%d HW=%d is off the channel-resident kernels (cot_bn_act_lay_covers)", N, C, HW);
    if (p) prof::annotate_op(21, N, C, C, HW, 1, dtype, (dresidual ? 1 : 0) | (y ? 2 : 0) | (dy2 ? 4 : 0));
    return rc;
}

// ---- the same pair with the ReLU sign mask (bn3 + residual + ReLU): the forward also writes one byte per 8 output elements, the
// backward reads those instead of the saved output
int64_t cot_bn_relu_mask_bytes(int N, int C, int HW, int dtype) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return bn_relu_mask_bytes(N, C, HW, dtype == COT_BF16 ? 2 : (dtype == COT_F32 ? 4 : 0));
}
struct BnMaskScope {  // (the kernels' launch code reads the mask of the call in flight from a thread-local)
    explicit BnMaskScope(uint8_t* m) { t_bn_mask = m; }
    ~BnMaskScope() { t_bn_mask = nullptr; }
};
int cot_bn_act_forward_mask(const void* x, const void* residual, void* y, void* relu_mask, const float* gamma, const float* beta,
                            float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* workspace, const float* sample_scale, int N, int C, int HW,
                            float eps, float momentum, int act, int dtype, void* stream) {
    if (!relu_mask) return set_error(COT_ERR_INVALID_ARG, "relu_mask is NULL (cot_bn_act_forward_ps is the call without one)");
    if (act != 1 || cot_bn_relu_mask_bytes(N, C, HW, dtype) == 0)
        return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_forward_mask: ReLU, and a geometry cot_bn_relu_mask_bytes accepts");
    BnMaskScope scope((uint8_t*)relu_mask);
    const int rc = cot_bn_act_forward_ps(x, residual, y, gamma, beta, save_mean, save_rstd, running_mean, running_var, num_batches_tracked,
                                         workspace, sample_scale, N, C, HW, eps, momentum, act, dtype, stream);
    if (rc == -3) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_forward_mask: the kernel serving N=%d C=%d HW=%d cannot write a sign mask", N, C, HW);
    return rc;
}
int cot_bn_act_backward_mask(const void* dy, const void* x, const void* relu_mask, void* dx, void* dresidual, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                             float* workspace, const float* sample_scale, int N, int C, int HW, int act, int dtype, void* stream) {
    if (!relu_mask) return set_error(COT_ERR_INVALID_ARG, "relu_mask is NULL");
    if (act != 1 || cot_bn_relu_mask_bytes(N, C, HW, dtype) == 0)
        return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_backward_mask: ReLU, and a geometry cot_bn_relu_mask_bytes accepts");
    BnMaskScope scope((uint8_t*)const_cast<void*>(relu_mask));
    // (`y` of the ordinary entry point only selects the sign-from-output kernels; with a mask they never dereference it)
    const bool p = prof::enabled();
    if (p) prof::mark();
    const int rc = cot_bn_act_backward_ps_impl(dy, x, /*y=*/relu_mask, dx, dresidual, gamma, beta, save_mean, save_rstd, dgamma, dbeta,
                                               workspace, sample_scale, N, C, HW, act, dtype, stream);
    if (p) prof::annotate_op(21, N, C, C, HW, 1, dtype, (dresidual ? 1 : 0));  // (no saved-output read: the mask is 1/16 of it)
    if (rc == -3) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_backward_mask: the kernel serving N=%d C=%d HW=%d cannot read a sign mask", N, C, HW);
    return rc;
}

int cot_bn_act_apply_forward(const void* x, const void* residual, void* y, void* relu_mask, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, int N, int C, int HW, int act, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || !mean || !rstd) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act");
    if (dtype != COT_BF16 || HW % 8 != 0) return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_apply_forward: COT_BF16, planes of a multiple of 8 pixels");
    if (relu_mask && (act != 1 || cot_bn_relu_mask_bytes(N, C, HW, dtype) == 0))
        return set_error(COT_ERR_UNSUPPORTED, "cot_bn_act_apply_forward: a sign mask needs ReLU and a geometry cot_bn_relu_mask_bytes accepts");
    int rc = check_align16({x, residual, y});
    if (rc) return rc;
    BnMaskScope scope((uint8_t*)relu_mask);
    const bool p = prof::enabled();
    if (p) prof::mark();
    rc = bn_apply_forward(x, residual, y, gamma, beta, mean, rstd, N, C, HW, act, (hipStream_t)stream);
    if (p) prof::annotate_op(20, N, C, C, HW, 1, dtype, (residual ? 1 : 0) | 8);
    return rc;
}

int cot_bn_act_inference(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                         const float* running_mean, const float* running_var, int N, int C, int HW, float eps, int act, int dtype,
                         void* stream) {
    if (!x || !y || !gamma || !beta || !running_mean || !running_var) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (N <= 0 || C <= 0 || HW <= 0 || act < 0 || act > 2) return set_error(COT_ERR_INVALID_ARG, "bad N/C/HW/act");
    int rc = check_align16({x, residual, y});
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == COT_F32) return bn_act_inference<float>(x, residual, y, gamma, beta, running_mean, running_var, N, C, HW, eps, act, s);
    if (dtype == COT_BF16) return bn_act_inference<bf16_t>(x, residual, y, gamma, beta, running_mean, running_var, N, C, HW, eps, act, s);
    return set_error(COT_ERR_UNSUPPORTED, "bn_act: dtype %d (float32 / bfloat16 only)", dtype);
}

int cot_profile_begin(void) {
    std::lock_guard<std::mutex> lk(prof::g_mu);
    for (auto& r : prof::g_recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    prof::g_recs.clear();
    const char* all = getenv("COT_PROFILE_ALL");
    prof::g_all.store(all && all[0] == '1');
    prof::g_on.store(true);
    return COT_OK;
}

int cot_profile_end(cot_profile_rec* out, int max_records) {
    prof::g_on.store(false);
    std::lock_guard<std::mutex> lk(prof::g_mu);
    int n = 0;
    for (auto& r : prof::g_recs) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
        r.pub.ms = ms;
        if (out && n < max_records) out[n] = r.pub;
        ++n;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    prof::g_recs.clear();
    return n;  // total number of launches recorded (may exceed max_records)
}

int cot_agg_forward(const void* x, const void* w, void* out, const cot_agg_geom* g, int dtype, int layout,
                    void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    if (layout != COT_NCHW && layout != COT_NHWC) return set_error(COT_ERR_UNSUPPORTED, "unknown layout %d", layout);
    if ((rc = check_align16({x, w, out}))) return rc;
    const bool p = prof::enabled();
    if (p) prof::mark();
    switch (dtype) {
        case COT_F32: rc = fwd_t<float>(x, w, out, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_F64: rc = fwd_t<double>(x, w, out, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_BF16: rc = fwd_t<bf16_t>(x, w, out, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_F16: rc = fwd_t<f16_t>(x, w, out, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        default: return set_error(COT_ERR_UNSUPPORTED, "unknown dtype %d", dtype);
    }
    if (p) prof::annotate(*g, dtype, layout, 0, 0);
    return rc;
}

int cot_agg_backward(const void* gout, const void* x, const void* w, void* gx, void* gw, const cot_agg_geom* g,
                     int dtype, int layout, void* stream) {
    int Ho, Wo, rc = validate(g, &Ho, &Wo);
    if (rc) return rc;
    if (!gout) return set_error(COT_ERR_INVALID_ARG, "gout is NULL");
    if (!gx && !gw) return set_error(COT_ERR_INVALID_ARG, "both gx and gw are NULL: nothing to compute");
    if (gx && !w) return set_error(COT_ERR_INVALID_ARG, "gx requested but w is NULL");
    if (gw && !x) return set_error(COT_ERR_INVALID_ARG, "gw requested but x is NULL");
    if (layout != COT_NCHW && layout != COT_NHWC) return set_error(COT_ERR_UNSUPPORTED, "unknown layout %d", layout);
    if ((rc = check_align16({gout, x, w, gx, gw}))) return rc;
    const bool p = prof::enabled();
    if (p) prof::mark();
    switch (dtype) {
        case COT_F32: rc = bwd_t<float>(gout, x, w, gx, gw, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_F64: rc = bwd_t<double>(gout, x, w, gx, gw, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_BF16: rc = bwd_t<bf16_t>(gout, x, w, gx, gw, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        case COT_F16: rc = bwd_t<f16_t>(gout, x, w, gx, gw, *g, Ho, Wo, layout, (hipStream_t)stream); break;
        default: return set_error(COT_ERR_UNSUPPORTED, "unknown dtype %d", dtype);
    }
    if (p) prof::annotate(*g, dtype, layout, 1, (gx ? 1 : 0) | (gw ? 2 : 0));
    return rc;
}

int cot_agg_backward_input(const void* gout, const void* w, void* gx, const cot_agg_geom* g, int dtype, int layout,
                           void* stream) {
    if (!gx) return set_error(COT_ERR_INVALID_ARG, "gx is NULL");
    return cot_agg_backward(gout, NULL, w, gx, NULL, g, dtype, layout, stream);
}

int cot_agg_backward_weight(const void* gout, const void* x, void* gw, const cot_agg_geom* g, int dtype, int layout,
                            void* stream) {
    if (!gw) return set_error(COT_ERR_INVALID_ARG, "gw is NULL");
    return cot_agg_backward(gout, x, NULL, NULL, gw, g, dtype, layout, stream);
}

static int validate_mix(const cot_agg_geom* g, int p2h, int p2w, int* Ho, int* Wo) {
    int rc = validate(g, Ho, Wo);
    if (rc) return rc;
    if (g->kh != 3 || g->kw != 3)  // LocalConvolutionMix asserts kernel sizes 3 and 5 (mix.py:328-329)
        return set_error(COT_ERR_INVALID_ARG, "mix: first kernel must be 3x3, got %dx%d", g->kh, g->kw);
    if (p2h < 0 || p2w < 0) return set_error(COT_ERR_INVALID_ARG, "mix: negative padding2");
    return COT_OK;
}

static int mix_done(int rc) {
    g_kernel = last_kernel_mix();
    return rc;
}

int cot_aggmix_forward(const void* x, const void* w1, const void* w2, void* out, const cot_agg_geom* g, int p2h,
                       int p2w, int dtype, void* stream) {
    int Ho, Wo, rc = validate_mix(g, p2h, p2w, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w1 || !w2 || !out) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    DISPATCH_DTYPE(dtype, mix_done(aggmix_forward<T>((const T*)x, (const T*)w1, (const T*)w2, (T*)out, *g, p2h, p2w, Ho, Wo,
                                                     (hipStream_t)stream)));
}

int cot_aggmix_backward_input(const void* gout, const void* w1, const void* w2, void* gx, const cot_agg_geom* g,
                              int p2h, int p2w, int all_heads, int dtype, void* stream) {
    int Ho, Wo, rc = validate_mix(g, p2h, p2w, &Ho, &Wo);
    if (rc) return rc;
    if (!gout || !w1 || !w2 || !gx) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    DISPATCH_DTYPE(dtype, mix_done(aggmix_backward_input<T>((const T*)gout, (const T*)w1, (const T*)w2, (T*)gx, *g, p2h, p2w,
                                                            all_heads, Ho, Wo, (hipStream_t)stream)));
}

int cot_aggmix_backward_weight(const void* gout, const void* x, void* gw1, void* gw2, const cot_agg_geom* g, int p2h,
                               int p2w, int dtype, void* stream) {
    int Ho, Wo, rc = validate_mix(g, p2h, p2w, &Ho, &Wo);
    if (rc) return rc;
    if (!gout || !x || !gw1 || !gw2) return set_error(COT_ERR_INVALID_ARG, "NULL device pointer");
    DISPATCH_DTYPE(dtype, mix_done(aggmix_backward_weight<T>((const T*)gout, (const T*)x, (T*)gw1, (T*)gw2, *g, p2h, p2w, Ho,
                                                             Wo, (hipStream_t)stream)));
}

}  // extern "C"
