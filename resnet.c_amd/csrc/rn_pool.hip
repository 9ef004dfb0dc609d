// Max / average pooling (reference cuda/ops.cu:50-108), NHWC and NCHW.
//
// HBM-bound: algorithmic bytes = 4 * (input + output elements).  NHWC kernels give
// each lane a float4 of channels of one output pixel, so a wave reads whole
// 256 B..1 KiB pixel rows; the window taps re-read neighbouring pixels through
// L1/L2 (3x3 stride 2 touches each input pixel 2.25x on average, from cache).
// Padded taps are skipped (-inf init for max; the average always divides by k
// twice, ops.cu:107), and the tap order kh -> kw is the reference's, so the fp32
// sum of the average is bit-identical.
#include "rn_internal.h"

namespace {

constexpr int kBlock = 256;

template <bool kMax>
__global__ __launch_bounds__(kBlock) void pool_nhwc_vec_kernel(
    const float *__restrict__ inp, float *__restrict__ out, int k, int stride, int pad, int Ho,
    int Wo, int C4, int H, int W, uint64_t total4)
{
    const float4 *in4 = reinterpret_cast<const float4 *>(inp);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    const float kf = (float)k;
    // total4 < 2^32 (checked by the dispatcher): 32-bit index arithmetic
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < total4; i64 += gstride) {
        const uint32_t i = (uint32_t)i64;
        const int c4 = (int)(i % (uint32_t)C4);
        uint32_t pix = i / (uint32_t)C4;
        const int ow = (int)(pix % (uint32_t)Wo);
        pix /= (uint32_t)Wo;
        const int oh = (int)(pix % (uint32_t)Ho);
        const uint64_t b = pix / (uint32_t)Ho;
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        float4 acc = kMax ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        // window inside the image (always so for the global average pool): no bounds tests, so
        // the loads of a row of taps are independent and issue together; the adds keep the
        // reference's kh-major order
        const bool inside = ih0 >= 0 && iw0 >= 0 && ih0 + k <= H && iw0 + k <= W;
        if (inside && !kMax) {
            const float4 *row = in4 + ((b * H + ih0) * W + iw0) * C4 + c4;
            for (int kh = 0; kh < k; ++kh, row += (size_t)W * C4) {
#pragma unroll 8
                for (int kw = 0; kw < k; ++kw) {
                    const float4 v = row[(size_t)kw * C4];
                    acc.x += v.x;
                    acc.y += v.y;
                    acc.z += v.z;
                    acc.w += v.w;
                }
            }
        } else
        for (int kh = 0; kh < k; ++kh) {
            const int ih = ih0 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = iw0 + kw;
                if (iw < 0 || iw >= W) continue;
                const float4 v = in4[((b * H + ih) * W + iw) * C4 + c4];
                if (kMax) {
                    acc.x = fmaxf(acc.x, v.x);
                    acc.y = fmaxf(acc.y, v.y);
                    acc.z = fmaxf(acc.z, v.z);
                    acc.w = fmaxf(acc.w, v.w);
                } else {
                    acc.x += v.x;
                    acc.y += v.y;
                    acc.z += v.z;
                    acc.w += v.w;
                }
            }
        }
        if (!kMax) {
            acc.x = acc.x / kf / kf;
            acc.y = acc.y / kf / kf;
            acc.z = acc.z / kf / kf;
            acc.w = acc.w / kf / kf;
        }
        // written once, read by the next op from HBM: a streaming store keeps the input rows,
        // which neighbouring windows re-read, in the cache
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4 *>(&out4[i64]));
    }
}

// any channel count, either layout; lanes run along the contiguous dimension
// (channels for NHWC, output columns for NCHW)
template <bool kMax>
__global__ __launch_bounds__(kBlock) void pool_scalar_kernel(const float *__restrict__ inp,
                                                             float *__restrict__ out, int k,
                                                             int stride, int pad, int Ho, int Wo,
                                                             int C, int H, int W, uint64_t total,
                                                             int nhwc)
{
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    const float kf = (float)k;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += gstride) {
        int c, oh, ow;
        uint64_t b;
        if (nhwc) {
            c = (int)(i % (uint64_t)C);
            uint64_t p = i / (uint64_t)C;
            ow = (int)(p % (uint64_t)Wo);
            p /= (uint64_t)Wo;
            oh = (int)(p % (uint64_t)Ho);
            b = p / (uint64_t)Ho;
        } else {
            ow = (int)(i % (uint64_t)Wo);
            uint64_t p = i / (uint64_t)Wo;
            oh = (int)(p % (uint64_t)Ho);
            p /= (uint64_t)Ho;
            c = (int)(p % (uint64_t)C);
            b = p / (uint64_t)C;
        }
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        float acc = kMax ? -INFINITY : 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int ih = ih0 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = iw0 + kw;
                if (iw < 0 || iw >= W) continue;
                const uint64_t src = nhwc ? (((b * H + ih) * W + iw) * C + c)
                                          : (((b * C + c) * H + ih) * W + iw);
                const float v = inp[src];
                acc = kMax ? fmaxf(acc, v) : acc + v;
            }
        }
        out[i] = kMax ? acc : acc / kf / kf;
    }
}

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

// bf16 NHWC, C % 8 == 0: 8 channels (16 bytes) per lane, fp32 max / sum
template <bool kMax>
__global__ __launch_bounds__(kBlock) void pool_nhwc_bf16_kernel(
    const bf16_t *__restrict__ inp, bf16_t *__restrict__ out, int k, int stride, int pad, int Ho,
    int Wo, int C8, int H, int W, uint64_t total8)
{
    const bf16x8 *in8 = reinterpret_cast<const bf16x8 *>(inp);
    bf16x8 *out8 = reinterpret_cast<bf16x8 *>(out);
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    const float kf = (float)k;
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < total8; i64 += gstride) {
        const uint32_t i = (uint32_t)i64;
        const int c8 = (int)(i % (uint32_t)C8);
        uint32_t pix = i / (uint32_t)C8;
        const int ow = (int)(pix % (uint32_t)Wo);
        pix /= (uint32_t)Wo;
        const int oh = (int)(pix % (uint32_t)Ho);
        const uint64_t b = pix / (uint32_t)Ho;
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = kMax ? -INFINITY : 0.f;
        for (int kh = 0; kh < k; ++kh) {
            const int ih = ih0 + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = iw0 + kw;
                if (iw < 0 || iw >= W) continue;
                const bf16x8 v = in8[((b * H + ih) * W + iw) * C8 + c8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = kMax ? fmaxf(acc[j], (float)v[j]) : acc[j] + (float)v[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(kMax ? acc[j] : acc[j] / kf / kf);
        out8[i64] = o;
    }
}

template <bool kMax>
int pool_bf16_dispatch(rn_ctx *ctx, const void *inp, void *out, uint64_t k, uint64_t stride,
                       uint64_t pad, uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t C,
                       uint64_t H, uint64_t W, const char *what)
{
    const uint64_t total = B * C * h_out * w_out;
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out && inp != out, "null or aliased tensor");
    RN_REQUIRE(ctx, k >= 1 && stride >= 1 && k < (1u << 15) && pad < (1u << 15) &&
                        stride < (1u << 15) && H < (1u << 30) && W < (1u << 30),
               "dimension out of range");
    RN_REQUIRE(ctx, C % 8 == 0 && total / 8 < (1ull << 32), "bf16 pooling needs C % 8 == 0");
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 15) == 0,
               "bf16 tensors must be 16-byte aligned");
    const uint64_t total8 = total / 8;
    pool_nhwc_bf16_kernel<kMax><<<rn_stream_grid(total8, kBlock), kBlock, 0, ctx->stream>>>(
        (const bf16_t *)inp, (bf16_t *)out, (int)k, (int)stride, (int)pad, (int)h_out, (int)w_out,
        (int)(C / 8), (int)H, (int)W, total8);
    return rn_after_launch(ctx, what);
}

template <bool kMax>
int pool_dispatch(rn_ctx *ctx, const float *inp, float *out, uint64_t k, uint64_t stride,
                  uint64_t pad, uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t C, uint64_t H,
                  uint64_t W, const char *what)
{
    RN_ENTER(ctx);
    const uint64_t total = B * C * h_out * w_out;
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out, "null tensor");
    RN_REQUIRE(ctx, inp != out, "pooling cannot run in place");
    RN_REQUIRE(ctx, k >= 1 && stride >= 1, "kernel_size and stride must be >= 1");
    RN_REQUIRE(ctx, H < (1u << 30) && W < (1u << 30) && C < (1u << 30) && k < (1u << 15) &&
                        pad < (1u << 15) && stride < (1u << 15),
               "dimension too large");
    RN_REQUIRE(ctx, h_out < (1u << 30) && w_out < (1u << 30), "dimension too large");
    const bool al = ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 15) ==
                    0;
    if (ctx->layout == RN_LAYOUT_NHWC && al && C % 4 == 0 && total / 4 < (1ull << 32)) {
        const uint64_t total4 = total / 4;
        pool_nhwc_vec_kernel<kMax><<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)k, (int)stride, (int)pad, (int)h_out, (int)w_out, (int)(C / 4), (int)H,
            (int)W, total4);
    } else {
        pool_scalar_kernel<kMax><<<rn_stream_grid(total, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)k, (int)stride, (int)pad, (int)h_out, (int)w_out, (int)C, (int)H,
            (int)W, total, ctx->layout == RN_LAYOUT_NHWC);
    }
    return rn_after_launch(ctx, what);
}

}  // namespace

extern "C" {

uint64_t rn_conv_output_size(uint64_t x, uint64_t kernel_size, uint64_t stride, uint64_t padding)
{
    // cuda/ops.cuh:9-13: unsigned arithmetic, integer division
    return (2 * padding + x - kernel_size) / stride + 1;
}

int rn_maxpool2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, const void *inp, void *out,
                                 uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                 uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t channels,
                                 uint64_t H, uint64_t W)
{
    RN_ENTER(ctx);
    if (dtype == RN_DTYPE_BF16)
        return pool_bf16_dispatch<true>(ctx, inp, out, kernel_size, stride, padding, h_out, w_out, B,
                                        channels, H, W, "rn_maxpool2d_nhwc_forward_dt");
    RN_REQUIRE(ctx, dtype == RN_DTYPE_F32, "unknown dtype");
    const int saved = ctx->layout;
    ctx->layout = RN_LAYOUT_NHWC;
    const int st = rn_maxpool2d_forward(ctx, (const float *)inp, (float *)out, kernel_size, stride,
                                        padding, h_out, w_out, B, channels, H, W);
    ctx->layout = saved;
    return st;
}

int rn_avgpool2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, const void *inp, void *out,
                                 uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                 uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t channels,
                                 uint64_t H, uint64_t W)
{
    RN_ENTER(ctx);
    if (dtype == RN_DTYPE_BF16)
        return pool_bf16_dispatch<false>(ctx, inp, out, kernel_size, stride, padding, h_out, w_out,
                                         B, channels, H, W, "rn_avgpool2d_nhwc_forward_dt");
    RN_REQUIRE(ctx, dtype == RN_DTYPE_F32, "unknown dtype");
    const int saved = ctx->layout;
    ctx->layout = RN_LAYOUT_NHWC;
    const int st = rn_avgpool2d_forward(ctx, (const float *)inp, (float *)out, kernel_size, stride,
                                        padding, h_out, w_out, B, channels, H, W);
    ctx->layout = saved;
    return st;
}

int rn_maxpool2d_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t kernel_size,
                         uint64_t stride, uint64_t padding, uint64_t h_out, uint64_t w_out,
                         uint64_t B, uint64_t channels, uint64_t H, uint64_t W)
{
    return pool_dispatch<true>(ctx, inp, out, kernel_size, stride, padding, h_out, w_out, B,
                               channels, H, W, "rn_maxpool2d_forward");
}

int rn_avgpool2d_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t kernel_size,
                         uint64_t stride, uint64_t padding, uint64_t h_out, uint64_t w_out,
                         uint64_t B, uint64_t channels, uint64_t H, uint64_t W)
{
    return pool_dispatch<false>(ctx, inp, out, kernel_size, stride, padding, h_out, w_out, B,
                                channels, H, W, "rn_avgpool2d_forward");
}

}  // extern "C"
