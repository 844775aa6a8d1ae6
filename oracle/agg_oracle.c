/*
 * oracle/agg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's `aggregation_zeropad` local weighted
 * aggregation (forward, input-backward, weight-backward) and of the
 * two-kernel-size `aggregation_zeropad_mix` variant.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product path (cotnet_amd/) never does.
 *
 * Reference semantics followed (paths relative to the reference checkout):
 *   forward           cupy_layers/aggregation_zeropad.py:20-46
 *   input backward    cupy_layers/aggregation_zeropad.py:48-79
 *   weight backward   cupy_layers/aggregation_zeropad.py:81-110
 *   output geometry   cupy_layers/aggregation_zeropad.py:120-123
 *   mix forward       cupy_layers/aggregation_zeropad_mix.py:20-74
 *   mix input bwd     cupy_layers/aggregation_zeropad_mix.py:76-140  (head 0 only, :87-88)
 *   mix weight bwd    cupy_layers/aggregation_zeropad_mix.py:142-207
 *
 * Pinning: tests/test_oracle.py checks this file against (i) the analytic
 * nn.Unfold oracle the reference's own self-tests use
 * (aggregation_zeropad.py:249-251) on the self-test shapes, (ii) the
 * reference's kernel source itself compiled for the CPU (oracle/_ref, built by
 * oracle/build_ref.py) and (iii) the committed fixtures in tests/golden/.
 *
 * Layout: everything NCHW-contiguous as in the reference.
 *   x   [N, C, H, W]
 *   w   [N, heads, wC, kh*kw, Ho, Wo]
 *   out [N, heads*C, Ho, Wo]          (head-major)
 * Loop structure is written output-element-major like the reference's
 * one-thread-per-element kernels, so the summation ORDER (kh outer, kw inner;
 * heads outermost in input-backward; channels cc ascending in weight-backward)
 * is the reference's: results are bit-identical to it in IEEE arithmetic.
 */
#include <stddef.h>
#include <stdint.h>

typedef struct {
    int N, C, H, W;        /* input */
    int heads, wC;         /* weight heads, weight channels (C % wC == 0) */
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int Ho, Wo;            /* output spatial size */
} agg_geom;

int agg_oracle_out_size(int in, int k, int s, int p, int d) {
    /* aggregation_zeropad.py:120  int((H + 2p - (d(k-1)+1)) / s + 1), python float division then int() */
    double v = (double)(in + 2 * p - (d * (k - 1) + 1)) / (double)s + 1.0;
    return (int)v;
}

#define DEFINE_AGG(T, SUF)                                                                     \
void agg_oracle_forward_##SUF(const T *x, const T *w, T *out, const agg_geom *g) {             \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t total = (int64_t)g->N * g->heads * g->C * HoWo;                              \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wo = (int)(index % g->Wo);                                                   \
        const int ho = (int)((index / g->Wo) % g->Ho);                                         \
        const int c = (int)((index / HoWo) % g->C);                                            \
        const int head = (int)((index / HoWo / g->C) % g->heads);                              \
        const int n = (int)(index / HoWo / g->C / g->heads);                                   \
        T value = 0;                                                                           \
        for (int kh = 0; kh < g->kh; ++kh) {                                                   \
            for (int kw = 0; kw < g->kw; ++kw) {                                               \
                const int h_in = -g->ph + ho * g->sh + kh * g->dh;                             \
                const int w_in = -g->pw + wo * g->sw + kw * g->dw;                             \
                if (h_in >= 0 && h_in < g->H && w_in >= 0 && w_in < g->W) {                    \
                    const int64_t ob = (((int64_t)n * g->C + c) * g->H + h_in) * g->W + w_in;  \
                    const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC + c % g->wC)   \
                                        * (g->kh * g->kw) + (kh * g->kw + kw)) * HoWo          \
                                       + (int64_t)ho * g->Wo + wo;                             \
                    value += w[ow] * x[ob];                                                    \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        out[index] = value;                                                                    \
    }                                                                                          \
}                                                                                              \
                                                                                               \
void agg_oracle_backward_input_##SUF(const T *gout, const T *w, T *gx, const agg_geom *g) {    \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t HW = (int64_t)g->H * g->W;                                                   \
    const int64_t total = (int64_t)g->N * g->C * HW;                                           \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wi = (int)(index % g->W);                                                    \
        const int hi = (int)((index / g->W) % g->H);                                           \
        const int c = (int)((index / HW) % g->C);                                              \
        const int n = (int)(index / HW / g->C);                                                \
        T value = 0;                                                                           \
        for (int head = 0; head < g->heads; ++head) {                                          \
            for (int kh = 0; kh < g->kh; ++kh) {                                               \
                for (int kw = 0; kw < g->kw; ++kw) {                                           \
                    const int h_out_s = hi + g->ph - kh * g->dh;                               \
                    const int w_out_s = wi + g->pw - kw * g->dw;                               \
                    /* C truncating % and / exactly as the reference (:62-66) */               \
                    if ((h_out_s % g->sh) == 0 && (w_out_s % g->sw) == 0) {                    \
                        const int h_out = h_out_s / g->sh;                                     \
                        const int w_out = w_out_s / g->sw;                                     \
                        if (h_out >= 0 && h_out < g->Ho && w_out >= 0 && w_out < g->Wo) {      \
                            const int64_t ot = ((((int64_t)n * g->heads + head) * g->C + c)    \
                                                * g->Ho + h_out) * g->Wo + w_out;              \
                            const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC        \
                                                 + c % g->wC) * (g->kh * g->kw)                \
                                                + (kh * g->kw + kw)) * HoWo                    \
                                               + (int64_t)h_out * g->Wo + w_out;               \
                            value += w[ow] * gout[ot];                                         \
                        }                                                                      \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        gx[index] = value;                                                                     \
    }                                                                                          \
}                                                                                              \
                                                                                               \
void agg_oracle_backward_weight_##SUF(const T *gout, const T *x, T *gw, const agg_geom *g) {   \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t total = (int64_t)g->N * g->heads * g->wC * HoWo;                             \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wo = (int)(index % g->Wo);                                                   \
        const int ho = (int)((index / g->Wo) % g->Ho);                                         \
        const int c = (int)((index / HoWo) % g->wC);                                           \
        const int head = (int)((index / HoWo / g->wC) % g->heads);                             \
        const int n = (int)(index / HoWo / g->wC / g->heads);                                  \
        for (int kh = 0; kh < g->kh; ++kh) {                                                   \
            for (int kw = 0; kw < g->kw; ++kw) {                                               \
                const int h_in = -g->ph + ho * g->sh + kh * g->dh;                             \
                const int w_in = -g->pw + wo * g->sw + kw * g->dw;                             \
                const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC + c)               \
                                    * (g->kh * g->kw) + (kh * g->kw + kw)) * HoWo              \
                                   + (int64_t)ho * g->Wo + wo;                                 \
                T value = 0;                                                                   \
                if (h_in >= 0 && h_in < g->H && w_in >= 0 && w_in < g->W) {                    \
                    for (int cc = c; cc < g->C; cc += g->wC) {                                 \
                        const int64_t ob = (((int64_t)n * g->C + cc) * g->H + h_in) * g->W     \
                                           + w_in;                                             \
                        const int64_t ot = ((((int64_t)n * g->heads + head) * g->C + cc)       \
                                            * g->Ho + ho) * g->Wo + wo;                        \
                        value += x[ob] * gout[ot];                                             \
                    }                                                                          \
                }                                                                              \
                gw[ow] = value; /* padded taps get an explicit 0 (:97-105) */                  \
            }                                                                                  \
        }                                                                                      \
    }                                                                                          \
}

DEFINE_AGG(float, f32)
DEFINE_AGG(double, f64)

/* ------------------------------------------------------------------------
 * aggregation_zeropad_mix: one input, two weight sets with kernel sizes 3 and
 * 5 (hard-coded tap loops, mix.py:35-36,:53-54), output
 * [N, 2*heads*C, Ho, Wo] ordered [kernel_idx][head][c] (mix.py:26-29).
 * Only stride/dilation/padding per the module call (stride s, pads p1/p2,
 * dilation d shared).  Input backward follows the reference quirk: ONLY head 0
 * contributes (mix.py:87-88) -- reproduced faithfully and flagged in DESIGN.md.
 * ---------------------------------------------------------------------- */
typedef struct {
    int N, C, H, W;
    int heads, wC;
    int sh, sw, dh, dw;
    int p1h, p1w, p2h, p2w; /* pads for the 3x3 and the 5x5 set */
    int Ho, Wo;
} aggmix_geom;

#define DEFINE_MIX(T, SUF)                                                                     \
void aggmix_oracle_forward_##SUF(const T *x, const T *w1, const T *w2, T *out,                 \
                                 const aggmix_geom *g) {                                       \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t total = (int64_t)g->N * 2 * g->heads * g->C * HoWo;                          \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wo = (int)(index % g->Wo);                                                   \
        const int ho = (int)((index / g->Wo) % g->Ho);                                         \
        const int c = (int)((index / HoWo) % g->C);                                            \
        const int head = (int)((index / HoWo / g->C) % g->heads);                              \
        const int kidx = (int)((index / HoWo / g->C / g->heads) % 2);                          \
        const int n = (int)(index / HoWo / g->C / g->heads / 2);                               \
        const int K = kidx == 0 ? 3 : 5;                                                       \
        const int ph = kidx == 0 ? g->p1h : g->p2h;                                            \
        const int pw = kidx == 0 ? g->p1w : g->p2w;                                            \
        const T *w = kidx == 0 ? w1 : w2;                                                      \
        T value = 0;                                                                           \
        for (int kh = 0; kh < K; ++kh) {                                                       \
            for (int kw = 0; kw < K; ++kw) {                                                   \
                const int h_in = -ph + ho * g->sh + kh * g->dh;                                \
                const int w_in = -pw + wo * g->sw + kw * g->dw;                                \
                if (h_in >= 0 && h_in < g->H && w_in >= 0 && w_in < g->W) {                    \
                    const int64_t ob = (((int64_t)n * g->C + c) * g->H + h_in) * g->W + w_in;  \
                    const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC + c % g->wC)   \
                                        * (K * K) + (kh * K + kw)) * HoWo                      \
                                       + (int64_t)ho * g->Wo + wo;                             \
                    value += w[ow] * x[ob];                                                    \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        out[index] = value;                                                                    \
    }                                                                                          \
}                                                                                              \
                                                                                               \
/* all_heads = 0 reproduces the reference (head 0 only); 1 is the mathematically */            \
/* complete gradient, used to document the difference.                          */            \
void aggmix_oracle_backward_input_##SUF(const T *gout, const T *w1, const T *w2, T *gx,        \
                                        const aggmix_geom *g, int all_heads) {                 \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t HW = (int64_t)g->H * g->W;                                                   \
    const int64_t total = (int64_t)g->N * g->C * HW;                                           \
    const int nh = all_heads ? g->heads : 1;                                                   \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wi = (int)(index % g->W);                                                    \
        const int hi = (int)((index / g->W) % g->H);                                           \
        const int c = (int)((index / HW) % g->C);                                              \
        const int n = (int)(index / HW / g->C);                                                \
        T value = 0;                                                                           \
        for (int head = 0; head < nh; ++head) {                                                \
            for (int kidx = 0; kidx < 2; ++kidx) {                                             \
                const int K = kidx == 0 ? 3 : 5;                                               \
                const int ph = kidx == 0 ? g->p1h : g->p2h;                                    \
                const int pw = kidx == 0 ? g->p1w : g->p2w;                                    \
                const T *w = kidx == 0 ? w1 : w2;                                              \
                for (int kh = 0; kh < K; ++kh) {                                               \
                    for (int kw = 0; kw < K; ++kw) {                                           \
                        const int h_out_s = hi + ph - kh * g->dh;                              \
                        const int w_out_s = wi + pw - kw * g->dw;                              \
                        if ((h_out_s % g->sh) == 0 && (w_out_s % g->sw) == 0) {                \
                            const int h_out = h_out_s / g->sh;                                 \
                            const int w_out = w_out_s / g->sw;                                 \
                            if (h_out >= 0 && h_out < g->Ho && w_out >= 0 && w_out < g->Wo) {  \
                                const int64_t ot = (((((int64_t)n * 2 + kidx) * g->heads       \
                                                      + head) * g->C + c) * g->Ho + h_out)     \
                                                   * g->Wo + w_out;                            \
                                const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC    \
                                                     + c % g->wC) * (K * K) + (kh * K + kw))   \
                                                   * HoWo + (int64_t)h_out * g->Wo + w_out;    \
                                value += w[ow] * gout[ot];                                     \
                            }                                                                  \
                        }                                                                      \
                    }                                                                          \
                }                                                                              \
            }                                                                                  \
        }                                                                                      \
        gx[index] = value;                                                                     \
    }                                                                                          \
}                                                                                              \
                                                                                               \
void aggmix_oracle_backward_weight_##SUF(const T *gout, const T *x, T *gw1, T *gw2,            \
                                         const aggmix_geom *g) {                               \
    const int64_t HoWo = (int64_t)g->Ho * g->Wo;                                               \
    const int64_t total = (int64_t)g->N * 2 * g->heads * g->wC * HoWo;                         \
    _Pragma("omp parallel for schedule(static)")                                               \
    for (int64_t index = 0; index < total; ++index) {                                          \
        const int wo = (int)(index % g->Wo);                                                   \
        const int ho = (int)((index / g->Wo) % g->Ho);                                         \
        const int c = (int)((index / HoWo) % g->wC);                                           \
        const int head = (int)((index / HoWo / g->wC) % g->heads);                             \
        const int kidx = (int)((index / HoWo / g->wC / g->heads) % 2);                         \
        const int n = (int)(index / HoWo / g->wC / g->heads / 2);                              \
        const int K = kidx == 0 ? 3 : 5;                                                       \
        const int ph = kidx == 0 ? g->p1h : g->p2h;                                            \
        const int pw = kidx == 0 ? g->p1w : g->p2w;                                            \
        T *gw = kidx == 0 ? gw1 : gw2;                                                         \
        for (int kh = 0; kh < K; ++kh) {                                                       \
            for (int kw = 0; kw < K; ++kw) {                                                   \
                const int h_in = -ph + ho * g->sh + kh * g->dh;                                \
                const int w_in = -pw + wo * g->sw + kw * g->dw;                                \
                const int64_t ow = ((((int64_t)n * g->heads + head) * g->wC + c) * (K * K)     \
                                    + (kh * K + kw)) * HoWo + (int64_t)ho * g->Wo + wo;        \
                T value = 0;                                                                   \
                if (h_in >= 0 && h_in < g->H && w_in >= 0 && w_in < g->W) {                    \
                    for (int cc = c; cc < g->C; cc += g->wC) {                                 \
                        const int64_t ob = (((int64_t)n * g->C + cc) * g->H + h_in) * g->W     \
                                           + w_in;                                             \
                        const int64_t ot = (((((int64_t)n * 2 + kidx) * g->heads + head)       \
                                             * g->C + cc) * g->Ho + ho) * g->Wo + wo;          \
                        value += x[ob] * gout[ot];                                             \
                    }                                                                          \
                }                                                                              \
                gw[ow] = value;                                                                \
            }                                                                                  \
        }                                                                                      \
    }                                                                                          \
}

DEFINE_MIX(float, f32)
DEFINE_MIX(double, f64)
