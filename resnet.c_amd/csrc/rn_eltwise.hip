// Bandwidth-bound ops: ReLU, residual add, inference batch-norm, argmax.
//
// All three elementwise kernels are grid-stride float4 streams (16 B per lane,
// 1 KiB per wave instruction) with a scalar tail, capped at 2048 blocks.  Their
// roofline is HBM: relu 8 B/element, add 12 B/element, batch-norm 8 B/element
// plus 16 B per channel of parameters.
//
// Batch-norm keeps the reference's arithmetic (cuda/ops.cu:149-150): because
// `1e-5` is a double literal the whole expression is evaluated in double and
// rounded to fp32 once at the store.  sqrt(var[c] + 1e-5) depends on the
// channel only, so a tiny prologue kernel evaluates it once per channel (same
// double value the reference recomputes per element) into context scratch.
#include "rn_internal.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void relu_kernel(const float *inp,
                                                      float *out, uint64_t n4,
                                                      uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float4 *in4 = reinterpret_cast<const float4 *>(inp);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 v = in4[i];
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
        out4[i] = v;
    }
    // tail (N % 4) handled by the first threads of block 0
    const uint64_t t = n4 * 4 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) out[t] = fmaxf(inp[t], 0.f);
}

__global__ __launch_bounds__(kBlock) void relu_scalar_kernel(const float *inp, float *out,
                                                             uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = fmaxf(inp[i], 0.f);
}

__global__ __launch_bounds__(kBlock) void add_kernel(const float *a,
                                                     const float *b,
                                                     float *out, uint64_t n4,
                                                     uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float4 *a4 = reinterpret_cast<const float4 *>(a);
    const float4 *b4 = reinterpret_cast<const float4 *>(b);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const float4 x = a4[i], y = b4[i];
        out4[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    const uint64_t t = n4 * 4 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) out[t] = a[t] + b[t];
}

__global__ __launch_bounds__(kBlock) void add_scalar_kernel(const float *a, const float *b,
                                                            float *out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = a[i] + b[i];
}

// per channel: {mean, sqrt(var + 1e-5), weight, bias} as doubles
__global__ void bn_prep_kernel(const float *weight, const float *bias, const float *mean,
                               const float *var, double *params, uint64_t C)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    params[4 * c + 0] = (double)mean[c];
    params[4 * c + 1] = sqrt((double)var[c] + 1e-5);
    params[4 * c + 2] = (double)weight[c];
    params[4 * c + 3] = (double)bias[c];
}

__device__ __forceinline__ float bn_apply(float x, const double *p)
{
    return (float)(((double)x - p[0]) / p[1] * p[2] + p[3]);
}

// NCHW, N % 4 == 0: one float4 never straddles a (b, c) plane.
__global__ __launch_bounds__(kBlock) void bn_nchw_vec_kernel(const float *inp,
                                                             float *out,
                                                             const double *__restrict__ params,
                                                             uint64_t total4, uint32_t n4,
                                                             uint32_t C)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float4 *in4 = reinterpret_cast<const float4 *>(inp);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total4; i += stride) {
        const uint32_t plane = (uint32_t)(i / n4);
        const double *p = params + 4 * (uint64_t)(plane % C);
        float4 v = in4[i];
        v.x = bn_apply(v.x, p);
        v.y = bn_apply(v.y, p);
        v.z = bn_apply(v.z, p);
        v.w = bn_apply(v.w, p);
        out4[i] = v;
    }
}

// NHWC, C % 4 == 0: a float4 covers channels c4*4 .. c4*4+3 of one pixel.
__global__ __launch_bounds__(kBlock) void bn_nhwc_vec_kernel(const float *inp,
                                                             float *out,
                                                             const double *__restrict__ params,
                                                             uint64_t total4, uint32_t c4n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const float4 *in4 = reinterpret_cast<const float4 *>(inp);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total4; i += stride) {
        const uint32_t c4 = (uint32_t)(i % c4n);
        const double *p = params + 16 * (uint64_t)c4;
        float4 v = in4[i];
        v.x = bn_apply(v.x, p);
        v.y = bn_apply(v.y, p + 4);
        v.z = bn_apply(v.z, p + 8);
        v.w = bn_apply(v.w, p + 12);
        out4[i] = v;
    }
}

// any shape, either layout
__global__ __launch_bounds__(kBlock) void bn_scalar_kernel(const float *inp, float *out,
                                                           const double *params, uint64_t total,
                                                           uint64_t N, uint64_t C, int nhwc)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const uint64_t c = nhwc ? (i % C) : ((i / N) % C);
        out[i] = bn_apply(inp[i], params + 4 * c);
    }
}

__global__ void bn_fold_kernel(const float *weight, const float *bias, const float *mean,
                               const float *var, float *scale, float *shift, uint64_t C)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s = (double)weight[c] / sqrt((double)var[c] + 1e-5);
    scale[c] = (float)s;
    shift[c] = (float)((double)bias[c] - (double)mean[c] * s);
}

// one wave per image; strict '<' of main.cu:245-248 == first maximum wins.  A NaN
// never wins a '<' comparison, except that a NaN at index 0 is never displaced.
__global__ __launch_bounds__(64) void argmax_kernel(const float *logits, uint64_t *idx,
                                                    uint64_t classes)
{
    const float *row = logits + (uint64_t)blockIdx.x * classes;
    const int lane = threadIdx.x;
    float best = -INFINITY;
    uint64_t best_i = ~(uint64_t)0;
    for (uint64_t i = lane; i < classes; i += 64) {
        float v = row[i];
        if (v != v) v = -INFINITY;
        if (v > best || (v == best && i < best_i)) {
            best = v;
            best_i = i;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const uint64_t oi = __shfl_down(best_i, off, 64);
        if (ov > best || (ov == best && oi < best_i)) {
            best = ov;
            best_i = oi;
        }
    }
    if (lane == 0) {
        const float first = row[0];
        idx[blockIdx.x] = (first != first) ? 0 : best_i;
    }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int rn_relu_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t N)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_REQUIRE(ctx, N == 0 || (inp && out), "null tensor");
    if (N == 0) return RN_OK;
    if (aligned16(inp) && aligned16(out)) {
        const uint64_t n4 = N / 4;
        relu_kernel<<<rn_stream_grid(n4, kBlock), kBlock, 0, ctx->stream>>>(inp, out, n4, N);
    } else {
        relu_scalar_kernel<<<rn_stream_grid(N, kBlock), kBlock, 0, ctx->stream>>>(inp, out, N);
    }
    return rn_after_launch(ctx, "rn_relu_forward");
}

int rn_add_forward(rn_ctx *ctx, const float *inp1, const float *inp2, float *out, uint64_t N)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_REQUIRE(ctx, N == 0 || (inp1 && inp2 && out), "null tensor");
    if (N == 0) return RN_OK;
    if (aligned16(inp1) && aligned16(inp2) && aligned16(out)) {
        const uint64_t n4 = N / 4;
        add_kernel<<<rn_stream_grid(n4, kBlock), kBlock, 0, ctx->stream>>>(inp1, inp2, out, n4,
                                                                           N);
    } else {
        add_scalar_kernel<<<rn_stream_grid(N, kBlock), kBlock, 0, ctx->stream>>>(inp1, inp2, out,
                                                                                 N);
    }
    return rn_after_launch(ctx, "rn_add_forward");
}

int rn_batchnorm2d_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                           const float *bias, const float *mean, const float *var, uint64_t B,
                           uint64_t C, uint64_t N)
{
    if (!ctx) return RN_ERR_INVALID;
    const uint64_t total = B * C * N;
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out && weight && bias && mean && var, "null tensor");
    RN_REQUIRE(ctx, C < (1ull << 31) && N < (1ull << 32), "dimension too large");
    void *scratch = nullptr;
    RN_TRY(rn_scratch(ctx, 0, C * 4 * sizeof(double), &scratch));
    double *params = static_cast<double *>(scratch);
    bn_prep_kernel<<<(unsigned)rn_ceil_div(C, 256), 256, 0, ctx->stream>>>(weight, bias, mean, var,
                                                                         params, C);
    const bool al = aligned16(inp) && aligned16(out);
    if (ctx->layout == RN_LAYOUT_NHWC && al && C % 4 == 0) {
        const uint64_t total4 = total / 4;
        bn_nhwc_vec_kernel<<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, params, total4, (uint32_t)(C / 4));
    } else if (ctx->layout == RN_LAYOUT_NCHW && al && N % 4 == 0 &&
               total / 4 / (N / 4) < (1ull << 32)) {
        const uint64_t total4 = total / 4;
        bn_nchw_vec_kernel<<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, params, total4, (uint32_t)(N / 4), (uint32_t)C);
    } else {
        bn_scalar_kernel<<<rn_stream_grid(total, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, params, total, N, C, ctx->layout == RN_LAYOUT_NHWC);
    }
    return rn_after_launch(ctx, "rn_batchnorm2d_forward");
}

int rn_batchnorm2d_fold(rn_ctx *ctx, const float *weight, const float *bias, const float *mean,
                        const float *var, float *scale, float *shift, uint64_t C)
{
    if (!ctx) return RN_ERR_INVALID;
    if (C == 0) return RN_OK;
    RN_REQUIRE(ctx, weight && bias && mean && var && scale && shift, "null tensor");
    bn_fold_kernel<<<(unsigned)rn_ceil_div(C, 256), 256, 0, ctx->stream>>>(weight, bias, mean, var,
                                                                         scale, shift, C);
    return rn_after_launch(ctx, "rn_batchnorm2d_fold");
}

int rn_argmax_forward(rn_ctx *ctx, const float *logits, uint64_t *idx, uint64_t B,
                      uint64_t classes)
{
    if (!ctx) return RN_ERR_INVALID;
    if (B == 0) return RN_OK;
    RN_REQUIRE(ctx, logits && idx && classes > 0, "null tensor or zero classes");
    RN_REQUIRE(ctx, B < (1ull << 31), "batch too large");
    argmax_kernel<<<(unsigned)B, 64, 0, ctx->stream>>>(logits, idx, classes);
    return rn_after_launch(ctx, "rn_argmax_forward");
}

}  // extern "C"
