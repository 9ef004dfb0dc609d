/*
 * rn_oracle.c -- CPU restatement of the reference's seven forward ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker.
 *
 * Every function restates the arithmetic of one kernel of the reference
 * (olehskip/resnet.c, cuda/ops.cu) on NCHW fp32 buffers:
 *   - same accumulation order (ic -> kh -> kw for conv, ascending i for fc),
 *   - one fp32 accumulator per output element,
 *   - padded taps are skipped, not multiplied by zero,
 *   - `sum += a * b` is restated as fmaf(a, b, sum) because nvcc contracts
 *     that statement to an FMA by default,
 *   - batch-norm as cuda/ops.cu:149-150 types it: `inp - mean[c]` is float - float, an
 *     fp32 subtraction; `var[c] + 1e-5` promotes to double (the literal is one), so the
 *     square root, the divide, `* weight[c]` and `+ bias[c]` are double and the result is
 *     rounded to fp32 once at the store.  `q * weight + bias` is one double fma here
 *     because nvcc contracts it (-fmad=true is its default; same rule as the conv sums),
 *   - average pool divides twice by (float)kernel_size (cuda/ops.cu:107).
 *
 * The loops are arranged so that the innermost loop runs over output columns
 * (independent accumulators), which lets the compiler vectorise without
 * touching the per-element summation order.
 *
 * Pinning: see oracle/README.md.  The reference's CUDA sources cannot be
 * compiled in this image (no nvcc, no <cuda/std/limits>), so this file is
 * pinned by (a) the known-answer rows recorded in SURVEY.md section 4 for the
 * reference's own conv2dTest input pattern and (b) the golden logits produced
 * by the reference's nn.Module definitions (tests/golden/).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define RN_ORACLE_API __attribute__((visibility("default")))

/* cuda/ops.cuh:9-13 -- unsigned arithmetic, integer division. */
RN_ORACLE_API uint64_t rn_oracle_conv_output_size(uint64_t x, uint64_t kernel_size, uint64_t stride,
                                                  uint64_t padding)
{
    return (2 * padding + x - kernel_size) / stride + 1;
}

/* Valid output-column range [lo, hi) for one kernel tap: 0 <= ow*stride - pad + kw < W. */
static inline void tap_range(int64_t kw, int64_t stride, int64_t pad, int64_t W, int64_t w_out,
                             int64_t *lo, int64_t *hi)
{
    int64_t l = 0, h = w_out;
    /* ow*stride >= pad - kw */
    int64_t need = pad - kw;
    if (need > 0) {
        l = (need + stride - 1) / stride;
    }
    /* ow*stride <= W - 1 + pad - kw */
    int64_t top = W - 1 + pad - kw;
    if (top < 0) {
        h = 0;
    } else {
        int64_t last = top / stride;
        if (last + 1 < h) {
            h = last + 1;
        }
    }
    if (l > h) {
        l = h;
    }
    *lo = l;
    *hi = h;
}

__attribute__((target("fma,avx2"))) static void conv_row_fma(
    const float *inp_b, float *out_row, const float *w_oc, int64_t k, int64_t stride, int64_t pad,
    int64_t oh, int64_t w_out, int64_t Cin, int64_t H, int64_t W)
{
    for (int64_t ow = 0; ow < w_out; ++ow) {
        out_row[ow] = 0.0f;
    }
    for (int64_t ic = 0; ic < Cin; ++ic) {
        for (int64_t kh = 0; kh < k; ++kh) {
            const int64_t ih = oh * stride - pad + kh;
            if (ih < 0 || ih >= H) {
                continue;
            }
            const float *in_row = inp_b + (ic * H + ih) * W;
            for (int64_t kw = 0; kw < k; ++kw) {
                int64_t lo, hi;
                tap_range(kw, stride, pad, W, w_out, &lo, &hi);
                const float wv = w_oc[(ic * k + kh) * k + kw];
                const float *src = in_row - pad + kw;
                if (stride == 1) {
                    for (int64_t ow = lo; ow < hi; ++ow) {
                        out_row[ow] = __builtin_fmaf(src[ow], wv, out_row[ow]);
                    }
                } else {
                    for (int64_t ow = lo; ow < hi; ++ow) {
                        out_row[ow] = __builtin_fmaf(src[ow * stride], wv, out_row[ow]);
                    }
                }
            }
        }
    }
}

static void conv_row_generic(const float *inp_b, float *out_row, const float *w_oc, int64_t k,
                             int64_t stride, int64_t pad, int64_t oh, int64_t w_out, int64_t Cin,
                             int64_t H, int64_t W)
{
    for (int64_t ow = 0; ow < w_out; ++ow) {
        out_row[ow] = 0.0f;
    }
    for (int64_t ic = 0; ic < Cin; ++ic) {
        for (int64_t kh = 0; kh < k; ++kh) {
            const int64_t ih = oh * stride - pad + kh;
            if (ih < 0 || ih >= H) {
                continue;
            }
            const float *in_row = inp_b + (ic * H + ih) * W;
            for (int64_t kw = 0; kw < k; ++kw) {
                int64_t lo, hi;
                tap_range(kw, stride, pad, W, w_out, &lo, &hi);
                const float wv = w_oc[(ic * k + kh) * k + kw];
                const float *src = in_row - pad + kw;
                for (int64_t ow = lo; ow < hi; ++ow) {
                    out_row[ow] = fmaf(src[ow * stride], wv, out_row[ow]);
                }
            }
        }
    }
}

/* cuda/ops.cu:14-48 (kernel) + cuda/nn.cu:3-16 (grid = every output element). */
RN_ORACLE_API void rn_oracle_conv2d(const float *inp, float *out, const float *weight,
                                    uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                    uint64_t h_out, uint64_t w_out, uint64_t B,
                                    uint64_t in_channels, uint64_t out_channels, uint64_t H,
                                    uint64_t W)
{
    const int have_fma = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    const int64_t rows = (int64_t)(B * out_channels * h_out);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t oh = r % (int64_t)h_out;
        const int64_t oc = (r / (int64_t)h_out) % (int64_t)out_channels;
        const int64_t b = r / (int64_t)(h_out * out_channels);
        const float *inp_b = inp + (size_t)b * in_channels * H * W;
        const float *w_oc = weight + (size_t)oc * in_channels * kernel_size * kernel_size;
        float *out_row = out + (size_t)r * w_out;
        if (have_fma) {
            conv_row_fma(inp_b, out_row, w_oc, (int64_t)kernel_size, (int64_t)stride,
                         (int64_t)padding, oh, (int64_t)w_out, (int64_t)in_channels, (int64_t)H,
                         (int64_t)W);
        } else {
            conv_row_generic(inp_b, out_row, w_oc, (int64_t)kernel_size, (int64_t)stride,
                             (int64_t)padding, oh, (int64_t)w_out, (int64_t)in_channels,
                             (int64_t)H, (int64_t)W);
        }
    }
}

/* cuda/ops.cu:50-78; fmax() is NaN-suppressing like the device fmax. */
RN_ORACLE_API void rn_oracle_maxpool2d(const float *inp, float *out, uint64_t kernel_size,
                                       uint64_t stride, uint64_t padding, uint64_t h_out,
                                       uint64_t w_out, uint64_t B, uint64_t channels, uint64_t H,
                                       uint64_t W)
{
    const int64_t planes = (int64_t)(B * channels);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p) {
        const float *src = inp + (size_t)p * H * W;
        float *dst = out + (size_t)p * h_out * w_out;
        for (int64_t oh = 0; oh < (int64_t)h_out; ++oh) {
            for (int64_t ow = 0; ow < (int64_t)w_out; ++ow) {
                float mx = -INFINITY;
                for (int64_t kh = 0; kh < (int64_t)kernel_size; ++kh) {
                    for (int64_t kw = 0; kw < (int64_t)kernel_size; ++kw) {
                        const int64_t ih = oh * (int64_t)stride - (int64_t)padding + kh;
                        const int64_t iw = ow * (int64_t)stride - (int64_t)padding + kw;
                        if (ih < 0 || ih >= (int64_t)H || iw < 0 || iw >= (int64_t)W) {
                            continue;
                        }
                        mx = fmaxf(mx, src[ih * (int64_t)W + iw]);
                    }
                }
                dst[oh * (int64_t)w_out + ow] = mx;
            }
        }
    }
}

/* cuda/ops.cu:80-108; divisor is always k (twice), padded taps contribute nothing. */
RN_ORACLE_API void rn_oracle_avgpool2d(const float *inp, float *out, uint64_t kernel_size,
                                       uint64_t stride, uint64_t padding, uint64_t h_out,
                                       uint64_t w_out, uint64_t B, uint64_t channels, uint64_t H,
                                       uint64_t W)
{
    const int64_t planes = (int64_t)(B * channels);
    const float kf = (float)kernel_size;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p) {
        const float *src = inp + (size_t)p * H * W;
        float *dst = out + (size_t)p * h_out * w_out;
        for (int64_t oh = 0; oh < (int64_t)h_out; ++oh) {
            for (int64_t ow = 0; ow < (int64_t)w_out; ++ow) {
                float sum = 0.0f;
                for (int64_t kh = 0; kh < (int64_t)kernel_size; ++kh) {
                    for (int64_t kw = 0; kw < (int64_t)kernel_size; ++kw) {
                        const int64_t ih = oh * (int64_t)stride - (int64_t)padding + kh;
                        const int64_t iw = ow * (int64_t)stride - (int64_t)padding + kw;
                        if (ih < 0 || ih >= (int64_t)H || iw < 0 || iw >= (int64_t)W) {
                            continue;
                        }
                        sum += src[ih * (int64_t)W + iw];
                    }
                }
                dst[oh * (int64_t)w_out + ow] = sum / kf / kf;
            }
        }
    }
}

/* cuda/ops.cu:110-128; bias (nullable) is added after the whole chain. */
RN_ORACLE_API void rn_oracle_linear(const float *inp, float *out, const float *weight,
                                    const float *bias, uint64_t B, uint64_t in_features,
                                    uint64_t out_features)
{
    const int64_t total = (int64_t)(B * out_features);
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < total; ++t) {
        const int64_t b = t / (int64_t)out_features;
        const int64_t o = t % (int64_t)out_features;
        const float *x = inp + (size_t)b * in_features;
        const float *w = weight + (size_t)o * in_features;
        float curr = 0.0f;
        for (uint64_t i = 0; i < in_features; ++i) {
            curr = fmaf(x[i], w[i], curr);
        }
        if (bias) {
            curr += bias[o];
        }
        out[t] = curr;
    }
}

/* cuda/ops.cu:130-137; fmax(NaN, 0) == 0.  In-place allowed. */
RN_ORACLE_API void rn_oracle_relu(const float *inp, float *out, uint64_t N)
{
    for (uint64_t n = 0; n < N; ++n) {
        out[n] = fmaxf(inp[n], 0.0f);
    }
}

/* cuda/ops.cu:139-151.  The subtraction has two float operands: it is an fp32 subtraction.
 * The 1e-5 literal is a double, so from `var[c] + 1e-5` on (sqrt, divide, multiply, add) the
 * expression is double; the multiply-add is contracted to an fma as nvcc does by default. */
RN_ORACLE_API void rn_oracle_batchnorm2d(const float *inp, float *out, const float *weight,
                                         const float *bias, const float *mean, const float *var,
                                         uint64_t B, uint64_t C, uint64_t N)
{
    const int64_t planes = (int64_t)(B * C);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p) {
        const int64_t c = p % (int64_t)C;
        const double denom = sqrt((double)var[c] + 1e-5);
        const float m = mean[c];
        const double g = (double)weight[c], beta = (double)bias[c];
        const float *src = inp + (size_t)p * N;
        float *dst = out + (size_t)p * N;
        for (uint64_t n = 0; n < N; ++n) {
            const float a = src[n] - m; /* fp32, ops.cu:150 */
            dst[n] = (float)fma((double)a / denom, g, beta);
        }
    }
}

/* cuda/ops.cu:153-160.  out may alias inp1. */
RN_ORACLE_API void rn_oracle_add(const float *inp1, const float *inp2, float *out, uint64_t N)
{
    for (uint64_t n = 0; n < N; ++n) {
        out[n] = inp1[n] + inp2[n];
    }
}

/* cuda/inference/main.cu:243-251 -- strict '<', so the first maximum wins. */
RN_ORACLE_API void rn_oracle_argmax(const float *logits, uint64_t B, uint64_t classes,
                                    uint64_t *out_idx)
{
    for (uint64_t b = 0; b < B; ++b) {
        uint64_t mx = 0;
        for (uint64_t i = 1; i < classes; ++i) {
            if (logits[b * classes + mx] < logits[b * classes + i]) {
                mx = i;
            }
        }
        out_idx[b] = mx;
    }
}
