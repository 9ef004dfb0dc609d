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

// Global k x k average (avgPool2dKernel ops.cu:80-108 as the network uses it, main.cu:120,213: 7x7 over
// a 7x7 image): one output pixel per image, every tap inside.  A lane owns V = 16 bytes of channels;
// all TAPS loads are issued before the first add -- 49 independent 16-byte loads in flight per lane
// instead of a row of seven at a time (the launch is 27 us for 103 MB: memory parallelism is all it
// has) -- then the taps are added one by one in the reference's kh-major order and divided by k twice.
template <typename E, int N, int TAPS>
__global__ __launch_bounds__(kBlock) void avgpool_global_kernel(const void *__restrict__ inp, void *__restrict__ outp,
                                                                int CV, float kf, uint64_t total)
{
    typedef E V __attribute__((ext_vector_type(N)));
    const V *in = static_cast<const V *>(inp);
    V *out = static_cast<V *>(outp);
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += gstride) {
        const uint64_t b = i / (uint64_t)CV, c = i - b * (uint64_t)CV;
        const V *src = in + b * (uint64_t)TAPS * (uint64_t)CV + c;
        V v[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) v[t] = __builtin_nontemporal_load(src + (size_t)t * CV);
        __builtin_amdgcn_sched_barrier(0);  // every load issued before the first add (else hipcc folds them into the chain)
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] += (float)v[t][j];
        V o;
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = (E)(acc[j] / kf / kf);
        out[i] = o;
    }
}

// 3x3 max-pool whose every window holds at least one real pixel (the network's pool after the
// stem, ops.cu:50-78 with k 3, stride 2, padding 1).  A padded tap is replaced by the nearest
// pixel inside the image, which is a tap of the same window already, so the maximum -- NaN
// rules of fmaxf included -- is the reference's; and then the nine 16-byte loads of a lane
// need no bounds branch and are all in flight together.  E x N = float x 4 or bf16 x 8.
template <typename E, int N>
__global__ __launch_bounds__(kBlock) void maxpool3_nhwc_kernel(const void *__restrict__ inp, void *__restrict__ outp,
                                                               int stride, int pad, int Ho, int Wo, int CV,
                                                               int H, int W, uint64_t total)
{
    typedef E V __attribute__((ext_vector_type(N)));
    const V *in = static_cast<const V *>(inp);
    V *out = static_cast<V *>(outp);
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += gstride) {
        const uint32_t i = (uint32_t)i64;
        const int cv = (int)(i % (uint32_t)CV);
        uint32_t pix = i / (uint32_t)CV;
        const int ow = (int)(pix % (uint32_t)Wo);
        pix /= (uint32_t)Wo;
        const int oh = (int)(pix % (uint32_t)Ho);
        const uint64_t b = pix / (uint32_t)Ho;
        const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
        int rows[3], cols[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            rows[t] = min(max(ih0 + t, 0), H - 1);
            cols[t] = min(max(iw0 + t, 0), W - 1);
        }
        V v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                v[3 * kh + kw] = in[((b * H + rows[kh]) * W + cols[kw]) * CV + cv];
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = -INFINITY;
#pragma unroll
        for (int t = 0; t < 9; ++t)  // tap order kh -> kw, as the reference
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] = fmaxf(acc[j], (float)v[t][j]);
        V o;
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = (E)acc[j];
        __builtin_nontemporal_store(o, &out[i64]);
    }
}

// The network's pool (3x3, stride 2, padding 1) as a walk DOWN the image: a lane owns one output column
// (b, ow, 16 bytes of channels) over a segment of output rows and carries the row maximum of the input row
// two windows share -- six 16-byte loads per output instead of nine, every input row met by a lane exactly
// once (the segment's first row twice), the loads of two output rows in flight together.  Padded taps are
// the nearest pixel inside the image (a tap of the same window), every row maximum starts from -inf
// (fmaxf drops NaNs, so a window of NaNs gives -inf as in the reference's acc = -inf; ops.cu:50-78), and
// v_max_f32 orders -0 < +0 whatever the operand order: the same value as the reference's kh -> kw walk.
template <typename E, int N>
__global__ __launch_bounds__(kBlock) void maxpool3s2_walk_kernel(const void *__restrict__ inp, void *__restrict__ outp,
                                                                 int Ho, int Wo, int CV, int H, int W, int segs,
                                                                 int seg_rows, uint32_t total)
{
    typedef E V __attribute__((ext_vector_type(N)));
    const V *in = static_cast<const V *>(inp);
    V *out = static_cast<V *>(outp);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total) return;
    const int cv = (int)(i % (uint32_t)CV);
    uint32_t q = i / (uint32_t)CV;
    const int ow = (int)(q % (uint32_t)Wo);
    q /= (uint32_t)Wo;
    const int seg = (int)(q % (uint32_t)segs);
    const uint64_t b = q / (uint32_t)segs;
    const int oh0 = seg * seg_rows, oh1 = min(Ho, oh0 + seg_rows);
    int cols[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) cols[t] = min(max(2 * ow - 1 + t, 0), W - 1);
    const V *img = in + b * (uint64_t)H * W * CV + cv;
    auto row_max = [&](int r, float (&m)[N]) {
        const V *row = img + (uint64_t)min(max(r, 0), H - 1) * W * CV;
        V v[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) v[t] = row[(uint64_t)cols[t] * CV];
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = fmaxf(fmaxf(fmaxf(-INFINITY, (float)v[0][j]), (float)v[1][j]), (float)v[2][j]);
    };
    float carry[N];
    row_max(2 * oh0 - 1, carry);
    V *o = out + ((b * Ho + oh0) * (uint64_t)Wo + ow) * CV + cv;
#pragma unroll 2  // (four output rows in flight measured slower: 5.1 against 5.4 TB/s)
    for (int oh = oh0; oh < oh1; ++oh) {
        float r1[N], r2[N];
        row_max(2 * oh, r1);
        row_max(2 * oh + 1, r2);
        V ov;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            ov[j] = (E)fmaxf(fmaxf(carry[j], r1[j]), r2[j]);
            carry[j] = r2[j];
        }
        __builtin_nontemporal_store(ov, o);
        o += (uint64_t)Wo * CV;
    }
}

// ---- NCHW forms of the network's two pools (the literal drop-in route; cuda/nn.cu:31-53 on NCHW) ----
//
// Global 7x7 average over NCHW planes of 49 contiguous floats: a block loads 256 planes as one
// contiguous run (coalesced dwords) into LDS, then lane l adds plane l's 49 taps in memory order =
// the reference's kh-major order (49 l + t: odd pitch, conflict-free).  pool_scalar_kernel put its
// lanes along a 1-pixel-wide output row: 0.37 ms for 103 MB.
template <int TAPS>
__global__ __launch_bounds__(kBlock) void avgpool_global_nchw_kernel(const float *__restrict__ inp,
                                                                     float *__restrict__ out, uint64_t planes, float kf)
{
    __shared__ float tile[kBlock * TAPS];
    const uint64_t p0 = (uint64_t)blockIdx.x * kBlock;
    const uint64_t n = min((uint64_t)kBlock, planes - p0) * TAPS;
    const float *src = inp + p0 * TAPS;
    for (uint64_t i = threadIdx.x; i < n; i += kBlock) tile[i] = src[i];
    __syncthreads();
    if (p0 + threadIdx.x < planes) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc += tile[threadIdx.x * TAPS + t];
        out[p0 + threadIdx.x] = acc / kf / kf;
    }
}

// 3x3 / stride 2 / padding 1 max-pool on NCHW planes whose width is a multiple of 8: a lane owns four
// adjacent outputs of one row and reads, per input row, two aligned float4 and the one column to
// their left -- 9 loads for 4 outputs where the scalar kernel issues 36 stride-2 dword loads.
// -inf start and fmaxf: a padded tap (skipped here, skipped by the reference: ops.cu:65-67) never wins.
__global__ __launch_bounds__(kBlock) void maxpool3s2_nchw_kernel(const float *__restrict__ inp, float *__restrict__ out,
                                                                 int H, int W, int Ho, int Q, uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += gstride) {
        const uint32_t i = (uint32_t)i64;
        const int j = (int)(i % (uint32_t)Q);
        const uint32_t r = i / (uint32_t)Q;
        const int oh = (int)(r % (uint32_t)Ho);
        const uint64_t plane = r / (uint32_t)Ho;
        const float *base = inp + plane * (uint64_t)H * W + 8 * j;
        float o0 = -INFINITY, o1 = -INFINITY, o2 = -INFINITY, o3 = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = 2 * oh - 1 + kh;
            if (ih < 0 || ih >= H) continue;
            const float *row = base + (size_t)ih * W;
            const float4 a = *reinterpret_cast<const float4 *>(row), b = *reinterpret_cast<const float4 *>(row + 4);
            const float e = j > 0 ? row[-1] : -INFINITY;
            // tap order kw = 0, 1, 2 within the window, rows outermost: the reference's
            o0 = fmaxf(fmaxf(fmaxf(o0, e), a.x), a.y);
            o1 = fmaxf(fmaxf(fmaxf(o1, a.y), a.z), a.w);
            o2 = fmaxf(fmaxf(fmaxf(o2, a.w), b.x), b.y);
            o3 = fmaxf(fmaxf(fmaxf(o3, b.y), b.z), b.w);
        }
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4{o0, o1, o2, o3}, reinterpret_cast<f32x4 *>(out + (plane * Ho + oh) * (uint64_t)(4 * Q) + 4 * j));
    }
}

// every window of the pooling geometry holds a real pixel, in both dimensions
static bool windows_never_empty(uint64_t k, uint64_t stride, uint64_t pad, uint64_t h_out, uint64_t w_out,
                                uint64_t H, uint64_t W)
{
    return pad < k && h_out >= 1 && w_out >= 1 && (h_out - 1) * stride < H + pad &&
           (w_out - 1) * stride < W + pad;
}

// the walk pays when every lane has a column of at least a few output rows and the grid still fills the chip
template <typename E, int N>
static bool maxpool_walk_launch(rn_ctx *ctx, const void *inp, void *out, uint64_t k, uint64_t stride, uint64_t pad,
                                uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t C, uint64_t H, uint64_t W)
{
    if (k != 3 || stride != 2 || pad != 1 || h_out != (H - 1) / 2 + 1 || w_out != (W - 1) / 2 + 1 || h_out < 8) return false;
    const uint64_t CV = C / N, cols = B * w_out * CV;
    // segments of at least 8 output rows, enough of them for ~512k lanes
    uint64_t segs = 1;
    while (segs * 2 * 8 <= h_out && cols * segs < (1ull << 19)) segs *= 2;
    const uint64_t seg_rows = (h_out + segs - 1) / segs;
    segs = (h_out + seg_rows - 1) / seg_rows;
    const uint64_t total = cols * segs;
    if (total >= (1ull << 31) || B * H * W * CV >= (1ull << 40)) return false;
    maxpool3s2_walk_kernel<E, N><<<(unsigned)rn_ceil_div(total, kBlock), kBlock, 0, ctx->stream>>>(
        inp, out, (int)h_out, (int)w_out, (int)CV, (int)H, (int)W, (int)segs, (int)seg_rows, (uint32_t)total);
    return true;
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
    if (kMax && maxpool_walk_launch<bf16_t, 8>(ctx, inp, out, k, stride, pad, h_out, w_out, B, C, H, W))
        return rn_after_launch(ctx, what);
    if (kMax && k == 3 && windows_never_empty(k, stride, pad, h_out, w_out, H, W)) {
        maxpool3_nhwc_kernel<bf16_t, 8><<<rn_stream_grid(total8, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)stride, (int)pad, (int)h_out, (int)w_out, (int)(C / 8), (int)H, (int)W, total8);
        return rn_after_launch(ctx, what);
    }
    if (!kMax && k == 7 && H == 7 && W == 7 && pad == 0 && h_out == 1 && w_out == 1) {
        avgpool_global_kernel<bf16_t, 8, 49><<<rn_stream_grid(total8, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)(C / 8), 7.f, total8);
        return rn_after_launch(ctx, what);
    }
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
    if (!ctx) return RN_ERR_INVALID;
    const uint64_t total = B * C * h_out * w_out;
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out, "null tensor");
    RN_REQUIRE(ctx, inp != out, "pooling cannot run in place");
    RN_REQUIRE(ctx, k >= 1 && stride >= 1, "kernel_size and stride must be >= 1");
    RN_REQUIRE(ctx, H < (1u << 30) && W < (1u << 30) && C < (1u << 30) && k < (1u << 15) &&
                        pad < (1u << 15) && stride < (1u << 15),
               "dimension too large");
    RN_REQUIRE(ctx, h_out < (1u << 30) && w_out < (1u << 30), "dimension too large");
    if (RN_DEFERS(ctx)) return rn_defer_pool(ctx, kMax ? 1 : 0, inp, out, k, stride, pad, h_out, w_out, B, C, H, W);
    RN_ENTER(ctx);
    const bool al = ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 15) ==
                    0;
    if (ctx->layout == RN_LAYOUT_NHWC && al && C % 4 == 0 && total / 4 < (1ull << 32)) {
        const uint64_t total4 = total / 4;
        if (kMax && maxpool_walk_launch<float, 4>(ctx, inp, out, k, stride, pad, h_out, w_out, B, C, H, W))
            return rn_after_launch(ctx, what);
        if (kMax && k == 3 && windows_never_empty(k, stride, pad, h_out, w_out, H, W)) {
            maxpool3_nhwc_kernel<float, 4><<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
                inp, out, (int)stride, (int)pad, (int)h_out, (int)w_out, (int)(C / 4), (int)H, (int)W, total4);
            return rn_after_launch(ctx, what);
        }
        if (!kMax && k == 7 && H == 7 && W == 7 && pad == 0 && h_out == 1 && w_out == 1) {
            avgpool_global_kernel<float, 4, 49><<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
                inp, out, (int)(C / 4), 7.f, total4);
            return rn_after_launch(ctx, what);
        }
        pool_nhwc_vec_kernel<kMax><<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)k, (int)stride, (int)pad, (int)h_out, (int)w_out, (int)(C / 4), (int)H,
            (int)W, total4);
    } else if (ctx->layout == RN_LAYOUT_NCHW && !kMax && k == 7 && H == 7 && W == 7 && pad == 0 && h_out == 1 &&
               w_out == 1) {
        const uint64_t planes = B * C;
        avgpool_global_nchw_kernel<49><<<(unsigned)rn_ceil_div(planes, kBlock), kBlock, 0, ctx->stream>>>(inp, out, planes,
                                                                                                          7.f);
    } else if (ctx->layout == RN_LAYOUT_NCHW && kMax && k == 3 && stride == 2 && pad == 1 && W % 8 == 0 && al &&
               w_out == W / 2 && h_out == (H - 1) / 2 + 1 && total / 4 < (1ull << 32)) {
        const uint64_t total4 = total / 4;  // one lane per four adjacent outputs
        maxpool3s2_nchw_kernel<<<rn_stream_grid(total4, kBlock), kBlock, 0, ctx->stream>>>(
            inp, out, (int)H, (int)W, (int)h_out, (int)(w_out / 4), total4);
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
