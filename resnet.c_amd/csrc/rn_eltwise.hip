// Bandwidth-bound ops: ReLU, residual add, inference batch-norm, argmax.
//
// All three elementwise kernels are grid-stride float4 streams (16 B per lane,
// 1 KiB per wave instruction) with a scalar tail, capped at 2048 blocks.  Their
// roofline is HBM: relu 8 B/element, add 12 B/element, batch-norm 8 B/element
// plus 16 B per channel of parameters.
//
// Batch-norm keeps the reference's arithmetic (cuda/ops.cu:149-150) type by type:
// `inp - mean[c]` has two float operands and is an fp32 subtraction (v_sub_f32);
// `1e-5` is a double literal, so `var[c] + 1e-5`, the square root, the divide, the
// multiply by weight[c] and the add of bias[c] are double, rounded to fp32 once at the
// store.  `q * weight + bias` is ONE double fma: nvcc contracts it by default, and it is
// written as fma() here so that it does not hang on a compiler flag.
// d = sqrt(var[c] + 1e-5) depends on the channel only; d and 1/d are derived once per
// channel (same double values the reference recomputes per element).  The per-element
// quotient a / d is then formed as q0 = a * (1/d); r = fma(-q0, d, a);
// q = fma(r, 1/d, q0), which is the correctly rounded double quotient (Markstein) at
// 3 FMAs -- a full IEEE v_div sequence per element made this kernel compute-bound at
// 2.1 TB/s instead of HBM-bound.
#include "rn_internal.h"

namespace {

constexpr int kBlock = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void relu_kernel(const float *inp,
                                                      float *out, uint64_t n4,
                                                      uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(inp);
    f32x4 *out4 = reinterpret_cast<f32x4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        f32x4 v = __builtin_nontemporal_load(&in4[i]);  // streamed once: keep it out of L2
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
        __builtin_nontemporal_store(v, &out4[i]);
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
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(b);
    f32x4 *out4 = reinterpret_cast<f32x4 *>(out);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 x = __builtin_nontemporal_load(&a4[i]), y = __builtin_nontemporal_load(&b4[i]);
        __builtin_nontemporal_store(x + y, &out4[i]);
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

constexpr int kBnStride = 8;  // doubles per channel: {mean (an fp32 value), d, 1/d, weight, bias, pad x3}

__global__ void bn_prep_kernel(const float *weight, const float *bias, const float *mean,
                               const float *var, double *params, uint64_t C)
{
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double d = sqrt((double)var[c] + 1e-5);
    double *p = params + kBnStride * c;
    p[0] = (double)mean[c];
    p[1] = d;
    p[2] = 1.0 / d;
    p[3] = (double)weight[c];
    p[4] = (double)bias[c];
}

__device__ __forceinline__ float bn_apply(float x, const double *p)
{
    const double a = (double)(x - (float)p[0]);  // fp32 subtraction, ops.cu:150
    const double q0 = a * p[2];
    const double r = fma(-q0, p[1], a);
    const double q = (r == r) ? fma(r, p[2], q0) : q0;  // == a / d, correctly rounded; a = +-inf or d = 0: r is NaN, q0 already is the quotient
    return (float)fma(q, p[3], p[4]);
}

struct BnParams {
    float m;
    double d, rinv, g, beta;
};

__device__ __forceinline__ BnParams bn_load(const double *p)
{
    BnParams q;
    q.m = (float)p[0], q.d = p[1], q.rinv = p[2], q.g = p[3], q.beta = p[4];
    return q;
}

// the same five numbers bn_prep_kernel tabulates, straight from the layer's tensors
__device__ __forceinline__ BnParams bn_derive(const float *weight, const float *bias,
                                              const float *mean, const float *var, uint32_t c)
{
    BnParams q;
    q.d = sqrt((double)var[c] + 1e-5);
    q.rinv = 1.0 / q.d;
    q.m = mean[c];
    q.g = (double)weight[c];
    q.beta = (double)bias[c];
    return q;
}

__device__ __forceinline__ float bn_apply_reg(float x, const BnParams &p)
{
    const double a = (double)(x - p.m);  // fp32 subtraction, ops.cu:150
    const double q0 = a * p.rinv;
    const double r = fma(-q0, p.d, a);
    const double q = (r == r) ? fma(r, p.rinv, q0) : q0;  // infinite a or d = 0: r is NaN and q0 already is a / d
    return (float)fma(q, p.g, p.beta);
}

// NCHW, N % 4 == 0: one wave walks one (b, c) plane at a time, so the channel's five
// doubles are wave-uniform and the lanes stream the plane as float4.  The constants are derived
// from the layer's tensors at the top of each plane (one square root and one division per plane
// of hundreds of pixels) instead of by a separate launch per call.
__global__ __launch_bounds__(kBlock) void bn_nchw_vec_kernel(const float *inp, float *out, uint32_t planes,
                                                             uint32_t n4, uint32_t C, const float *weight,
                                                             const float *bias, const float *mean, const float *var)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    for (uint32_t plane = wave; plane < planes; plane += nwaves) {
        const BnParams p = bn_derive(weight, bias, mean, var, plane % C);
        const f32x4 *in4 = reinterpret_cast<const f32x4 *>(inp) + (uint64_t)plane * n4;
        f32x4 *out4 = reinterpret_cast<f32x4 *>(out) + (uint64_t)plane * n4;
        for (uint32_t i = lane; i < n4; i += 64) {
            f32x4 v = __builtin_nontemporal_load(&in4[i]);  // streamed once: keep it out of L2
            v.x = bn_apply_reg(v.x, p);
            v.y = bn_apply_reg(v.y, p);
            v.z = bn_apply_reg(v.z, p);
            v.w = bn_apply_reg(v.w, p);
            __builtin_nontemporal_store(v, &out4[i]);
        }
    }
}

// NCHW, any plane size (7x7 = 49 pixels is not a multiple of 4, and such planes are not 16-byte
// aligned): the same walk with one float per lane.  A plane is one contiguous run, so a wave's
// loads are whole cache lines; no per-element division to find the channel.
__global__ __launch_bounds__(kBlock) void bn_nchw_plane_kernel(const float *inp, float *out, uint32_t planes,
                                                               uint32_t N, uint32_t C, const float *weight,
                                                               const float *bias, const float *mean, const float *var)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t nwaves = gridDim.x * (kBlock / 64);
    for (uint32_t plane = wave; plane < planes; plane += nwaves) {
        const BnParams p = bn_derive(weight, bias, mean, var, plane % C);
        const float *in = inp + (uint64_t)plane * N;
        float *o = out + (uint64_t)plane * N;
        for (uint32_t i = lane; i < N; i += 64) o[i] = bn_apply_reg(in[i], p);
    }
}

// NCHW, small planes (14x14, 7x7): with one wave per plane the square root and the division of
// bn_derive stand in front of one or two loads per lane (7x7: 2.1-2.5 TB/s).  Here a lane keeps ONE
// position of the C x N image -- channel (position / N), the same in every image -- derives that
// channel's constants once and walks the batch: image b's element is b * C * N further, a wave's
// 64 lanes are 64 consecutive floats (VEC: float4s; a float4 never straddles a plane when N % 4 == 0)
// of every image.  blockIdx.y splits the batch so that the grid fills the chip.
template <bool VEC>
__global__ __launch_bounds__(kBlock) void bn_nchw_batchwalk_kernel(const float *inp, float *out, uint32_t per_image,
                                                                   uint32_t N, uint32_t B, uint32_t b_per_block,
                                                                   const float *weight, const float *bias,
                                                                   const float *mean, const float *var)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // position in the image, in floats or float4s
    if (i >= per_image) return;
    const uint32_t b0 = blockIdx.y * b_per_block, b1 = min(B, b0 + b_per_block);
    const BnParams p = bn_derive(weight, bias, mean, var, (VEC ? 4 * i : i) / N);
    if constexpr (VEC) {
        const f32x4 *in4 = reinterpret_cast<const f32x4 *>(inp) + i;
        f32x4 *out4 = reinterpret_cast<f32x4 *>(out) + i;
#pragma unroll 4
        for (uint32_t b = b0; b < b1; ++b) {
            f32x4 v = __builtin_nontemporal_load(&in4[(uint64_t)b * per_image]);
            v.x = bn_apply_reg(v.x, p);
            v.y = bn_apply_reg(v.y, p);
            v.z = bn_apply_reg(v.z, p);
            v.w = bn_apply_reg(v.w, p);
            __builtin_nontemporal_store(v, &out4[(uint64_t)b * per_image]);
        }
    } else {
#pragma unroll 4
        for (uint32_t b = b0; b < b1; ++b) {
            const uint64_t at = (uint64_t)b * per_image + i;
            out[at] = bn_apply_reg(inp[at], p);
        }
    }
}

// NHWC, C % 4 == 0: a float4 covers channels c4*4 .. c4*4+3 of one pixel.  When the grid
// stride is a multiple of C/4 every thread keeps the same four channels for its whole
// walk, so their parameters are loaded once into registers (kFixed); reloading 20
// doubles per float4 through the vector cache made the kernel load-issue bound.
template <bool kFixed>
__global__ __launch_bounds__(kBlock) void bn_nhwc_vec_kernel(const float *inp, float *out,
                                                             const double *__restrict__ params,
                                                             uint64_t total4, uint32_t c4n,
                                                             const float *weight, const float *bias,
                                                             const float *mean, const float *var)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(inp);
    f32x4 *out4 = reinterpret_cast<f32x4 *>(out);
    const uint64_t first = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    BnParams p0, p1, p2, p3;
    if (kFixed) {
        // every thread keeps its four channels: derive their constants here instead of in a
        // separate launch (53 tiny launches per ResNet-50 forward in the op-by-op mode)
        const uint32_t c0 = 4 * (uint32_t)(first % c4n);
        p0 = bn_derive(weight, bias, mean, var, c0), p1 = bn_derive(weight, bias, mean, var, c0 + 1),
        p2 = bn_derive(weight, bias, mean, var, c0 + 2), p3 = bn_derive(weight, bias, mean, var, c0 + 3);
    }
    for (uint64_t i = first; i < total4; i += stride) {
        if (!kFixed) {
            const double *p = params + 4 * kBnStride * (uint64_t)(uint32_t)(i % c4n);
            p0 = bn_load(p), p1 = bn_load(p + kBnStride), p2 = bn_load(p + 2 * kBnStride),
            p3 = bn_load(p + 3 * kBnStride);
        }
        f32x4 v = __builtin_nontemporal_load(&in4[i]);
        v.x = bn_apply_reg(v.x, p0);
        v.y = bn_apply_reg(v.y, p1);
        v.z = bn_apply_reg(v.z, p2);
        v.w = bn_apply_reg(v.w, p3);
        __builtin_nontemporal_store(v, &out4[i]);
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
        out[i] = bn_apply(inp[i], params + kBnStride * c);
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
    if (RN_DEFERS(ctx)) return rn_defer_eltwise(ctx, 0, inp, nullptr, out, N);
    RN_ENTER(ctx);
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
    if (RN_DEFERS(ctx)) return rn_defer_eltwise(ctx, 1, inp1, inp2, out, N);
    RN_ENTER(ctx);
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
    if (RN_DEFERS(ctx)) return rn_defer_bn(ctx, inp, out, weight, bias, mean, var, B, C, N);
    RN_ENTER(ctx);
    void *scratch = nullptr;
    RN_TRY(rn_scratch(ctx, 0, C * kBnStride * sizeof(double), &scratch));
    double *params = static_cast<double *>(scratch);
    auto prep = [&]() {
        bn_prep_kernel<<<(unsigned)rn_ceil_div(C, 256), 256, 0, ctx->stream>>>(weight, bias, mean, var,
                                                                             params, C);
    };
    const bool al = aligned16(inp) && aligned16(out);
    if (ctx->layout == RN_LAYOUT_NHWC && al && C % 4 == 0) {
        const uint64_t total4 = total / 4;
        const uint32_t c4n = (uint32_t)(C / 4);
        unsigned grid = rn_stream_grid(total4, kBlock);
        // make the grid stride a multiple of C/4 when a nearby grid size allows it
        bool fixed = false;
        for (unsigned g = grid; g >= 1 && g + 64 > grid; --g) {
            if (((uint64_t)g * kBlock) % c4n == 0) {
                grid = g;
                fixed = true;
                break;
            }
        }
        if (fixed) {
            bn_nhwc_vec_kernel<true><<<grid, kBlock, 0, ctx->stream>>>(inp, out, params, total4, c4n,
                                                                       weight, bias, mean, var);
        } else {
            prep();
            bn_nhwc_vec_kernel<false><<<grid, kBlock, 0, ctx->stream>>>(inp, out, params, total4, c4n,
                                                                        weight, bias, mean, var);
        }
    } else if (ctx->layout == RN_LAYOUT_NCHW && N <= 256 && B >= 8 && C * N < (1ull << 31)) {
        // small planes, a batch to walk: one position of the image per lane (see the kernel)
        const bool vec = al && N % 4 == 0;
        const uint32_t per_image = (uint32_t)(vec ? C * N / 4 : C * N);
        const unsigned gx = (unsigned)rn_ceil_div(per_image, kBlock);
        // about 2,048 blocks (8 per CU), at least 4 images per block
        uint64_t gy = rn_ceil_div(2048, gx);
        if (gy > B / 4) gy = B / 4;
        if (gy < 1) gy = 1;
        const uint32_t bpb = (uint32_t)rn_ceil_div(B, gy);
        const dim3 grid(gx, (unsigned)rn_ceil_div(B, bpb));
        if (vec)
            bn_nchw_batchwalk_kernel<true><<<grid, kBlock, 0, ctx->stream>>>(inp, out, per_image, (uint32_t)N, (uint32_t)B,
                                                                             bpb, weight, bias, mean, var);
        else
            bn_nchw_batchwalk_kernel<false><<<grid, kBlock, 0, ctx->stream>>>(inp, out, per_image, (uint32_t)N, (uint32_t)B,
                                                                              bpb, weight, bias, mean, var);
    } else if (ctx->layout == RN_LAYOUT_NCHW && al && N % 4 == 0 && B * C < (1ull << 32)) {
        const uint64_t planes = B * C;
        const unsigned grid = rn_stream_grid(planes * 64, kBlock);
        bn_nchw_vec_kernel<<<grid, kBlock, 0, ctx->stream>>>(inp, out, (uint32_t)planes, (uint32_t)(N / 4),
                                                             (uint32_t)C, weight, bias, mean, var);
    } else if (ctx->layout == RN_LAYOUT_NCHW && B * C < (1ull << 32)) {
        const uint64_t planes = B * C;
        bn_nchw_plane_kernel<<<rn_stream_grid(planes * 64, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (uint32_t)planes, (uint32_t)N, (uint32_t)C, weight, bias, mean, var);
    } else {
        prep();
        bn_scalar_kernel<<<rn_stream_grid(total, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, params, total, N, C, ctx->layout == RN_LAYOUT_NHWC);
    }
    return rn_after_launch(ctx, "rn_batchnorm2d_forward");
}

int rn_batchnorm2d_fold(rn_ctx *ctx, const float *weight, const float *bias, const float *mean,
                        const float *var, float *scale, float *shift, uint64_t C)
{
    RN_ENTER(ctx);
    if (C == 0) return RN_OK;
    RN_REQUIRE(ctx, weight && bias && mean && var && scale && shift, "null tensor");
    bn_fold_kernel<<<(unsigned)rn_ceil_div(C, 256), 256, 0, ctx->stream>>>(weight, bias, mean, var,
                                                                         scale, shift, C);
    return rn_after_launch(ctx, "rn_batchnorm2d_fold");
}

int rn_argmax_forward(rn_ctx *ctx, const float *logits, uint64_t *idx, uint64_t B,
                      uint64_t classes)
{
    RN_ENTER(ctx);
    if (B == 0) return RN_OK;
    RN_REQUIRE(ctx, logits && idx && classes > 0, "null tensor or zero classes");
    RN_REQUIRE(ctx, B < (1ull << 31), "batch too large");
    argmax_kernel<<<(unsigned)B, 64, 0, ctx->stream>>>(logits, idx, classes);
    return rn_after_launch(ctx, "rn_argmax_forward");
}

}  // extern "C"
