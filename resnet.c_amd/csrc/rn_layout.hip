// Layout converters between the reference's NCHW / OIHW world and the engine's
// NHWC activations and K-major weight panels.  None of this exists in the
// reference (it is NCHW end to end, cuda/ops.cu:3-7); it is the price of the
// MI355X-first data layout and is paid once per weight at load time and once per
// network input.
#include "rn_internal.h"

namespace {

constexpr int kTile = 32;

// src viewed as [batch][R][S] row-major -> dst [batch][S][R].  NCHW->NHWC is R=C,
// S=H*W; NHWC->NCHW is R=H*W, S=C.  32x32 tile through LDS (+1 pad: conflict-free
// column reads), 32x8 threads, both global sides coalesced along their inner dim.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ src,
                                                        float *__restrict__ dst, uint32_t R,
                                                        uint32_t S)
{
    __shared__ float tile[kTile][kTile + 1];
    const uint64_t img = (uint64_t)blockIdx.z * R * S;
    const uint32_t s0 = blockIdx.x * kTile, r0 = blockIdx.y * kTile;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t j = ty; j < kTile; j += 8) {
        const uint32_t r = r0 + j, s = s0 + tx;
        if (r < R && s < S) tile[j][tx] = src[img + (uint64_t)r * S + s];
    }
    __syncthreads();
    for (uint32_t j = ty; j < kTile; j += 8) {
        const uint32_t s = s0 + j, r = r0 + tx;
        if (r < R && s < S) dst[img + (uint64_t)s * R + r] = tile[tx][j];
    }
}

// The same with 16-byte accesses on both global sides: a 64 x 64 tile, every thread loads four
// float4s along S and stores four float4s along R (R % 4 == 0, S % 4 == 0, 16-byte aligned
// tensors).  LDS pitch 65 floats: the four dword writes of a loaded float4 and the four dword reads
// behind a stored one both spread over the banks (two lanes per bank = the wave's two passes).
// The 32 x 32 form moves one dword per lane and access: 3.6 TB/s on the 3x3 layers' inputs of
// the drop-in route.
__global__ __launch_bounds__(256) void transpose64_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          uint32_t R, uint32_t S)
{
    __shared__ float tile[64][65];
    const uint64_t img = (uint64_t)blockIdx.z * R * S;
    const uint32_t s0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const uint32_t q = threadIdx.x & 15, j0 = threadIdx.x >> 4;  // float4 column 0..15, row 0..15 (+16 i)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = r0 + j0 + 16 * i, s = s0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && s < S) v = *reinterpret_cast<const float4 *>(src + img + (uint64_t)r * S + s);
        float *t = &tile[j0 + 16 * i][4 * q];
        t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t sl = j0 + 16 * i, s = s0 + sl, r = r0 + 4 * q;
        if (r < R && s < S) {
            float4 v;
            v.x = tile[4 * q][sl], v.y = tile[4 * q + 1][sl], v.z = tile[4 * q + 2][sl], v.w = tile[4 * q + 3][sl];
            *reinterpret_cast<float4 *>(dst + img + (uint64_t)s * R + r) = v;
        }
    }
}

// NCHW [B,C,HW] -> NHWC [B,HW,4], channels >= C zero-filled.  One float4 per pixel.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float *__restrict__ src,
                                                            float4 *__restrict__ dst, uint32_t C,
                                                            uint32_t HW, uint64_t total_pix)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < total_pix; p += gstride) {
        const uint64_t b = p / HW;
        const uint32_t hw = (uint32_t)(p - b * HW);
        const float *s = src + b * C * HW + hw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = s[0];
        if (C > 1) v.y = s[HW];
        if (C > 2) v.z = s[2 * (uint64_t)HW];
        if (C > 3) v.w = s[3 * (uint64_t)HW];
        dst[p] = v;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float *__restrict__ src,
                                                               float *__restrict__ dst,
                                                               uint32_t C, uint32_t HW,
                                                               uint32_t Cpad, uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t c = (uint32_t)(i % Cpad);
        const uint64_t p = i / Cpad;
        const uint64_t b = p / HW;
        const uint32_t hw = (uint32_t)(p - b * HW);
        dst[i] = c < C ? src[(b * C + c) * HW + hw] : 0.f;
    }
}

// OIHW -> [Cout][kh][kw][Cin]
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ w,
                                                          float *__restrict__ packed, uint32_t Cin,
                                                          uint32_t k, uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t kk = k * k;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t ic = (uint32_t)(i % Cin);
        uint64_t r = i / Cin;
        const uint32_t tap = (uint32_t)(r % kk);
        const uint64_t oc = r / kk;
        packed[i] = w[(oc * Cin + ic) * kk + tap];
    }
}

// exact-K small-Cin panel: row oc = [kh][kw][Cin] packed without slots, zero-padded to Kpad
__global__ __launch_bounds__(256) void pack_weight_exact_kernel(const float *__restrict__ w,
                                                                float *__restrict__ packed,
                                                                uint32_t Cin, uint32_t k,
                                                                uint32_t Kpad, uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t kk = k * k, kreal = kk * Cin;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t q = (uint32_t)(i % Kpad);
        const uint64_t oc = i / Kpad;
        float v = 0.f;
        if (q < kreal) {
            const uint32_t ic = q % Cin, tap = q / Cin;
            v = w[(oc * Cin + ic) * kk + tap];
        }
        packed[i] = v;
    }
}

// fused pair: row oc = [kh][kw][Cin] of w1 * s1[oc], then [Cin2] of the 1x1 w2 * s2[oc]
template <typename TP>
__global__ __launch_bounds__(256) void pack_weight_pair_kernel(
    const float *__restrict__ w1, const float *__restrict__ s1, const float *__restrict__ w2,
    const float *__restrict__ s2, TP *__restrict__ packed, uint32_t Cin, uint32_t k, uint32_t Cin2,
    uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t kk = k * k, K1 = kk * Cin, Kt = K1 + Cin2;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t q = (uint32_t)(i % Kt);
        const uint64_t oc = i / Kt;
        float v;
        if (q < K1) {
            const uint32_t ic = q % Cin, tap = q / Cin;
            v = w1[(oc * Cin + ic) * kk + tap];
            if (s1) v *= s1[oc];
        } else {
            v = w2[oc * Cin2 + (q - K1)];
            if (s2) v *= s2[oc];
        }
        packed[i] = (TP)v;
    }
}

// small-Cin ("stem") panel: [Cout][kh][8 kw slots][4 channel slots], zero where kw >= k
// or ic >= Cin, so one 32-float K segment = 8 consecutive pixels of a 4-channel image
__global__ __launch_bounds__(256) void pack_weight_c4_kernel(const float *__restrict__ w,
                                                             float *__restrict__ packed,
                                                             uint32_t Cin, uint32_t k,
                                                             uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t ic = (uint32_t)(i & 3);
        const uint32_t kw = (uint32_t)((i >> 2) & 7);
        const uint64_t r = i >> 5;
        const uint32_t kh = (uint32_t)(r % k);
        const uint64_t oc = r / k;
        packed[i] = (ic < Cin && kw < k) ? w[((oc * Cin + ic) * k + kh) * k + kw] : 0.f;
    }
}

typedef __bf16 bf16_t;

// OIHW fp32 -> bf16 [Cout][kh][kw][Cin]
__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float *__restrict__ w,
                                                               bf16_t *__restrict__ packed,
                                                               uint32_t Cin, uint32_t k,
                                                               uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t kk = k * k;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t ic = (uint32_t)(i % Cin);
        uint64_t r = i / Cin;
        const uint32_t tap = (uint32_t)(r % kk);
        const uint64_t oc = r / kk;
        packed[i] = (bf16_t)w[(oc * Cin + ic) * kk + tap];
    }
}

// small-Cin bf16 panel: [Cout][ceil(k/2)][2 kernel rows][8 kw slots][4 channel slots]: one
// 128-byte K segment = 8 consecutive pixels of a 4-channel bf16 image in each of two
// consecutive image rows (kernel rows 2t and 2t+1); zero where kh >= k, kw >= k or ic >= Cin
__global__ __launch_bounds__(256) void pack_weight_c4_bf16_kernel(const float *__restrict__ w,
                                                                  bf16_t *__restrict__ packed,
                                                                  uint32_t Cin, uint32_t k,
                                                                  uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t kt = (k + 1) / 2;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t ic = (uint32_t)(i & 3);
        const uint32_t kw = (uint32_t)((i >> 2) & 7);
        const uint32_t hi = (uint32_t)((i >> 5) & 1);
        const uint64_t r = i >> 6;
        const uint32_t kh = 2 * (uint32_t)(r % kt) + hi;
        const uint64_t oc = r / kt;
        packed[i] = (bf16_t)((ic < Cin && kw < k && kh < k) ? w[((oc * Cin + ic) * k + kh) * k + kw] : 0.f);
    }
}

// fp32 NCHW -> T NHWC [B][H+2b][W+2b][Cpad] with a zero border of b pixels
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_border_kernel(const float *__restrict__ src,
                                                                  T *__restrict__ dst, uint32_t C,
                                                                  uint32_t H, uint32_t W,
                                                                  uint32_t Cpad, uint32_t border,
                                                                  uint64_t total)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    const uint32_t Hp = H + 2 * border, Wp = W + 2 * border;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gstride) {
        const uint32_t c = (uint32_t)(i % Cpad);
        uint64_t p = i / Cpad;
        const uint32_t wp = (uint32_t)(p % Wp);
        p /= Wp;
        const uint32_t hp = (uint32_t)(p % Hp);
        const uint64_t b = p / Hp;
        const int h = (int)hp - (int)border, w = (int)wp - (int)border;
        float v = 0.f;
        if (c < C && h >= 0 && h < (int)H && w >= 0 && w < (int)W)
            v = src[((b * C + c) * H + h) * W + w];
        dst[i] = (T)v;
    }
}

// same, fp32, one 16-byte store per thread (needs (H+2b)*(W+2b)*Cpad % 4 == 0); divisions by
// Cpad and W+2b as multiply-high (n / d == umulhi(n, mul) >> shr for n < 2^31)
__global__ __launch_bounds__(256) void nchw_to_nhwc_border_f4_kernel(
    const float *__restrict__ src, float4 *__restrict__ dst, uint32_t C, uint32_t H, uint32_t W,
    uint32_t Cpad, uint32_t border, uint32_t Wp, uint32_t n4, uint32_t mul_c, uint32_t shr_c,
    uint32_t mul_w, uint32_t shr_w)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint64_t b = blockIdx.y;
    const float *img = src + b * C * H * W;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t idx = 4 * i + e;
        const uint32_t pix = Cpad == 1 ? idx : __umulhi(idx, mul_c) >> shr_c;
        const uint32_t c = idx - pix * Cpad;
        const uint32_t hp = __umulhi(pix, mul_w) >> shr_w;
        const uint32_t h = hp - border, w = pix - hp * Wp - border;  // wrap = out of range
        v[e] = (c < C && h < H && w < W) ? img[(c * H + h) * W + w] : 0.f;
    }
    dst[b * n4 + i] = make_float4(v[0], v[1], v[2], v[3]);
}

// same, bf16, Cpad = 4: two pixels (16 bytes) per thread; needs an even W + 2b
__global__ __launch_bounds__(256) void nchw_to_nhwc4_border_bf16_kernel(
    const float *__restrict__ src, uint4 *__restrict__ dst, uint32_t C, uint32_t H, uint32_t W,
    uint32_t border, uint32_t Wp, uint32_t n2, uint32_t mul_w, uint32_t shr_w)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;  // pixel pair of one image
    if (i >= n2) return;
    const uint64_t b = blockIdx.y;
    const float *img = src + b * C * H * W;
    const uint32_t pix = 2 * i;
    const uint32_t hp = __umulhi(pix, mul_w) >> shr_w;
    const uint32_t h = hp - border, w0 = pix - hp * Wp - border;  // wrap = out of range
    typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t c = e & 3, w = w0 + (e >> 2);
        v[e] = (bf16_t)((c < C && h < H && w < W) ? img[(c * H + h) * W + w] : 0.f);
    }
    dst[b * n2 + i] = __builtin_bit_cast(uint4, v);
}

void magic_div(uint32_t d, uint32_t *mul, uint32_t *shr)
{
    uint32_t lg = 0;
    while ((1u << lg) < d) ++lg;
    const uint32_t p = 31 + lg;
    *mul = (uint32_t)(((1ull << p) + d - 1) / d);
    *shr = p - 32;
}

}  // namespace

bool rn_conv_is_c4(uint64_t Cin, uint64_t k) { return Cin <= 4 && k <= 8; }

extern "C" {

uint64_t rn_conv2d_input_channels(uint64_t in_channels)
{
    return in_channels < 4 ? 4 : in_channels;
}

uint64_t rn_conv2d_packed_weight_numel(uint64_t in_channels, uint64_t out_channels,
                                       uint64_t kernel_size)
{
    if (rn_conv_is_c4(in_channels, kernel_size)) return out_channels * kernel_size * 32;
    return out_channels * kernel_size * kernel_size * in_channels;
}

int rn_conv2d_pack_weight(rn_ctx *ctx, const float *weight_oihw, float *packed,
                          uint64_t in_channels, uint64_t out_channels, uint64_t kernel_size)
{
    RN_ENTER(ctx);
    const uint64_t total = rn_conv2d_packed_weight_numel(in_channels, out_channels, kernel_size);
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, weight_oihw && packed && weight_oihw != packed, "null or aliased tensor");
    RN_REQUIRE(ctx, in_channels < (1u << 30) && kernel_size < (1u << 15), "dimension too large");
    if (rn_conv_is_c4(in_channels, kernel_size)) {
        pack_weight_c4_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            weight_oihw, packed, (uint32_t)in_channels, (uint32_t)kernel_size, total);
    } else {
        pack_weight_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            weight_oihw, packed, (uint32_t)in_channels, (uint32_t)kernel_size, total);
    }
    return rn_after_launch(ctx, "rn_conv2d_pack_weight");
}

uint64_t rn_conv2d_packed_weight_numel_exact(uint64_t in_channels, uint64_t out_channels,
                                             uint64_t kernel_size)
{
    return out_channels * rn_ceil_div(kernel_size * kernel_size * in_channels, 32) * 32;
}

int rn_conv2d_pack_weight_exact(rn_ctx *ctx, const float *weight_oihw, float *packed,
                                uint64_t in_channels, uint64_t out_channels, uint64_t kernel_size)
{
    RN_ENTER(ctx);
    const uint64_t total =
        rn_conv2d_packed_weight_numel_exact(in_channels, out_channels, kernel_size);
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, weight_oihw && packed && weight_oihw != packed, "null or aliased tensor");
    RN_REQUIRE(ctx, in_channels >= 1 && in_channels <= 16 && kernel_size >= 1 && kernel_size <= 15,
               "in_channels / kernel_size out of range");
    pack_weight_exact_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
        weight_oihw, packed, (uint32_t)in_channels, (uint32_t)kernel_size,
        (uint32_t)(total / out_channels), total);
    return rn_after_launch(ctx, "rn_conv2d_pack_weight_exact");
}

uint64_t rn_conv2d_packed_pair_weight_numel(uint64_t in_channels, uint64_t out_channels,
                                            uint64_t kernel_size, uint64_t in_channels2)
{
    return out_channels * (kernel_size * kernel_size * in_channels + in_channels2);
}

int rn_conv2d_pack_weight_pair_dt(rn_ctx *ctx, int dtype, const float *w1_oihw, const float *scale1,
                                  const float *w2_oihw, const float *scale2, void *packed,
                                  uint64_t in_channels, uint64_t out_channels, uint64_t kernel_size,
                                  uint64_t in_channels2)
{
    RN_ENTER(ctx);
    RN_REQUIRE(ctx, dtype == RN_DTYPE_F32 || dtype == RN_DTYPE_BF16, "unknown dtype");
    const uint64_t total =
        rn_conv2d_packed_pair_weight_numel(in_channels, out_channels, kernel_size, in_channels2);
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, w1_oihw && w2_oihw && packed, "null tensor");
    RN_REQUIRE(ctx, in_channels >= 1 && in_channels2 >= 1 && kernel_size >= 1, "empty convolution");
    RN_REQUIRE(ctx, kernel_size < (1u << 12) &&
                        kernel_size * kernel_size * in_channels + in_channels2 < (1ull << 31),
               "dimension too large");
    if (dtype == RN_DTYPE_F32)
        pack_weight_pair_kernel<float><<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            w1_oihw, scale1, w2_oihw, scale2, (float *)packed, (uint32_t)in_channels,
            (uint32_t)kernel_size, (uint32_t)in_channels2, total);
    else
        pack_weight_pair_kernel<bf16_t><<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            w1_oihw, scale1, w2_oihw, scale2, (bf16_t *)packed, (uint32_t)in_channels,
            (uint32_t)kernel_size, (uint32_t)in_channels2, total);
    return rn_after_launch(ctx, "rn_conv2d_pack_weight_pair_dt");
}

uint64_t rn_conv2d_packed_weight_numel_dt(int dtype, uint64_t in_channels, uint64_t out_channels,
                                          uint64_t kernel_size)
{
    if (dtype == RN_DTYPE_F32)
        return rn_conv2d_packed_weight_numel(in_channels, out_channels, kernel_size);
    if (rn_conv_is_c4(in_channels, kernel_size)) return out_channels * rn_ceil_div(kernel_size, 2) * 64;
    return out_channels * kernel_size * kernel_size * in_channels;
}

int rn_conv2d_pack_weight_dt(rn_ctx *ctx, int dtype, const float *weight_oihw, void *packed,
                             uint64_t in_channels, uint64_t out_channels, uint64_t kernel_size)
{
    RN_ENTER(ctx);
    if (dtype == RN_DTYPE_F32)
        return rn_conv2d_pack_weight(ctx, weight_oihw, (float *)packed, in_channels, out_channels,
                                     kernel_size);
    RN_REQUIRE(ctx, dtype == RN_DTYPE_BF16, "unknown dtype");
    const uint64_t total =
        rn_conv2d_packed_weight_numel_dt(dtype, in_channels, out_channels, kernel_size);
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, weight_oihw && packed, "null tensor");
    RN_REQUIRE(ctx, in_channels < (1u << 30) && kernel_size < (1u << 15), "dimension too large");
    if (rn_conv_is_c4(in_channels, kernel_size)) {
        pack_weight_c4_bf16_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            weight_oihw, (bf16_t *)packed, (uint32_t)in_channels, (uint32_t)kernel_size, total);
    } else {
        pack_weight_bf16_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            weight_oihw, (bf16_t *)packed, (uint32_t)in_channels, (uint32_t)kernel_size, total);
    }
    return rn_after_launch(ctx, "rn_conv2d_pack_weight_dt");
}

int rn_nchw_to_nhwc_pad_dt(rn_ctx *ctx, int dtype, const float *src, void *dst, uint64_t B,
                           uint64_t C, uint64_t H, uint64_t W, uint64_t Cpad, uint64_t border)
{
    RN_ENTER(ctx);
    if (dtype == RN_DTYPE_F32 && border == 0)
        return rn_nchw_to_nhwc_pad(ctx, src, (float *)dst, B, C, H, W, Cpad);
    const uint64_t total = B * (H + 2 * border) * (W + 2 * border) * Cpad;
    if (total == 0) return RN_OK;
    RN_REQUIRE(ctx, src && dst && (const void *)src != dst, "null or aliased tensor");
    RN_REQUIRE(ctx, Cpad >= C && C >= 1, "Cpad must be >= C >= 1");
    RN_REQUIRE(ctx, H < (1u << 20) && W < (1u << 20) && Cpad < (1u << 20) && border < (1u << 10),
               "dimension too large");
    if (dtype == RN_DTYPE_BF16 && Cpad == 4 && (W + 2 * border) % 2 == 0 &&
        (H + 2 * border) * (W + 2 * border) < (1ull << 31) &&
        (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const uint32_t n2 = (uint32_t)((H + 2 * border) * (W + 2 * border) / 2);
        uint32_t mw = 0, sw = 0;
        magic_div((uint32_t)(W + 2 * border), &mw, &sw);
        for (uint64_t b0 = 0; b0 < B; b0 += 65535) {
            const uint64_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
            nchw_to_nhwc4_border_bf16_kernel<<<dim3((n2 + 255) / 256, (unsigned)nb), 256, 0,
                                               ctx->stream>>>(
                src + b0 * C * H * W, (uint4 *)dst + b0 * n2, (uint32_t)C, (uint32_t)H,
                (uint32_t)W, (uint32_t)border, (uint32_t)(W + 2 * border), n2, mw, sw);
        }
    } else if (dtype == RN_DTYPE_BF16) {
        nchw_to_nhwc_border_kernel<bf16_t><<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            src, (bf16_t *)dst, (uint32_t)C, (uint32_t)H, (uint32_t)W, (uint32_t)Cpad,
            (uint32_t)border, total);
    } else if (dtype == RN_DTYPE_F32 && ((H + 2 * border) * (W + 2 * border) * Cpad) % 4 == 0 &&
               (H + 2 * border) * (W + 2 * border) * Cpad < (1ull << 31) && Cpad >= 1 &&
               W + 2 * border >= 2 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const uint32_t n4 = (uint32_t)((H + 2 * border) * (W + 2 * border) * Cpad / 4);
        uint32_t mc = 0, sc = 0, mw = 0, sw = 0;
        if (Cpad > 1) magic_div((uint32_t)Cpad, &mc, &sc);
        magic_div((uint32_t)(W + 2 * border), &mw, &sw);
        for (uint64_t b0 = 0; b0 < B; b0 += 65535) {
            const uint64_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
            nchw_to_nhwc_border_f4_kernel<<<dim3((n4 + 255) / 256, (unsigned)nb), 256, 0,
                                            ctx->stream>>>(
                src + b0 * C * H * W, (float4 *)dst + b0 * n4, (uint32_t)C, (uint32_t)H,
                (uint32_t)W, (uint32_t)Cpad, (uint32_t)border, (uint32_t)(W + 2 * border), n4, mc,
                sc, mw, sw);
        }
    } else if (dtype == RN_DTYPE_F32) {
        nchw_to_nhwc_border_kernel<float><<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            src, (float *)dst, (uint32_t)C, (uint32_t)H, (uint32_t)W, (uint32_t)Cpad,
            (uint32_t)border, total);
    } else {
        return rn_set_error(ctx, RN_ERR_INVALID, "unknown dtype %d", dtype);
    }
    return rn_after_launch(ctx, "rn_nchw_to_nhwc_pad_dt");
}

static int transpose_launch(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t R,
                            uint64_t S, const char *what)
{
    if (B * R * S == 0) return RN_OK;
    RN_REQUIRE(ctx, src && dst && src != dst, "null or aliased tensor");
    RN_REQUIRE(ctx, R < (1ull << 31) && S < (1ull << 31), "dimension too large");
    if (R % 4 == 0 && S % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
        rn_ceil_div(R, 64) <= 65535) {
        for (uint64_t b0 = 0; b0 < B; b0 += 65535) {
            const uint64_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
            dim3 grid((unsigned)rn_ceil_div(S, 64), (unsigned)rn_ceil_div(R, 64), (unsigned)nb);
            transpose64_kernel<<<grid, 256, 0, ctx->stream>>>(src + b0 * R * S, dst + b0 * R * S, (uint32_t)R, (uint32_t)S);
        }
        return rn_after_launch(ctx, what);
    }
    const uint64_t gy = rn_ceil_div(R, kTile);
    RN_REQUIRE(ctx, gy <= 65535, "too many rows for one launch");
    // gridDim.z is limited to 65535: walk the batch in slabs
    for (uint64_t b0 = 0; b0 < B; b0 += 65535) {
        const uint64_t nb = (B - b0) < 65535 ? (B - b0) : 65535;
        dim3 grid((unsigned)rn_ceil_div(S, kTile), (unsigned)gy, (unsigned)nb);
        transpose_kernel<<<grid, 256, 0, ctx->stream>>>(src + b0 * R * S, dst + b0 * R * S,
                                                        (uint32_t)R, (uint32_t)S);
    }
    return rn_after_launch(ctx, what);
}

int rn_nchw_to_nhwc(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C, uint64_t H,
                    uint64_t W)
{
    RN_ENTER(ctx);
    return transpose_launch(ctx, src, dst, B, C, H * W, "rn_nchw_to_nhwc");
}

int rn_nhwc_to_nchw(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C, uint64_t H,
                    uint64_t W)
{
    RN_ENTER(ctx);
    return transpose_launch(ctx, src, dst, B, H * W, C, "rn_nhwc_to_nchw");
}

int rn_nchw_to_nhwc_pad(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C,
                        uint64_t H, uint64_t W, uint64_t Cpad)
{
    RN_ENTER(ctx);
    const uint64_t HW = H * W;
    if (B * HW * Cpad == 0) return RN_OK;
    RN_REQUIRE(ctx, src && dst && src != dst, "null or aliased tensor");
    RN_REQUIRE(ctx, Cpad >= C && C >= 1, "Cpad must be >= C >= 1");
    RN_REQUIRE(ctx, HW < (1ull << 31) && Cpad < (1ull << 31), "dimension too large");
    if (Cpad == 4 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const uint64_t pix = B * HW;
        nchw_to_nhwc4_kernel<<<rn_stream_grid(pix, 256), 256, 0, ctx->stream>>>(
            src, reinterpret_cast<float4 *>(dst), (uint32_t)C, (uint32_t)HW, pix);
    } else {
        const uint64_t total = B * HW * Cpad;
        nchw_to_nhwc_pad_kernel<<<rn_stream_grid(total, 256), 256, 0, ctx->stream>>>(
            src, dst, (uint32_t)C, (uint32_t)HW, (uint32_t)Cpad, total);
    }
    return rn_after_launch(ctx, "rn_nchw_to_nhwc_pad");
}

}  // extern "C"
