// conv3 of one bottleneck block chained with conv1 of the next, bf16 storage, 64 mid channels.
//
// Reference sequence (layerForward, cuda/inference/main.cu:131-164, twice): conv3 (1x1, 64 -> 256)
// + bn3 + residual add + ReLU ends block M, conv1 (1x1, 256 -> 64 | 128) + bn1 + ReLU opens block
// M+1.  As two launches the 256-channel tensor y is written, read back by conv1, and read a third
// time as block M+1's residual.  Rows are independent for 1x1 convolutions, so one launch can do
// both products on a tile of rows: y is still written once (the next block's residual needs it)
// but conv1 takes it from LDS -- at B = 256 that is 411 MB less to read per chain and one launch
// less.  Both halves do exactly what the separate launches do: the same k order per output
// element, the same affine / residual / ReLU expression in fp32, y rounded to bf16 before conv1
// multiplies it (the separate conv1 reads the stored bf16 tensor), so the bits are the same.
//
//  * persistent blocks of 8 waves, 64 rows per step; the block's weights are MFMA operands in
//    REGISTERS for its lifetime (conv3: 32 output channels per wave x 4 k-steps, conv1: one
//    32-channel fragment x 16 k-steps);
//  * the step's t2 rows (8 KB) and residual rows (32 KB) come by LDS-DMA, one step ahead, into
//    double buffers; t2 in the MFMA operand image ((row>>1)&7 chunk swizzle), the residual rows
//    with their 16-byte slots XORed by row&15 so that the epilogue's 8-byte reads spread over
//    the banks;
//  * operands swapped (weights = rows, pixels = columns): a lane ends with four consecutive
//    channels of one pixel, adds its residual from LDS, and v_permlane32_swap pairs the half-waves
//    into 16-byte runs -- written into the y tile in LDS, laid out as conv1's operand image.
//    After one barrier the tile leaves for HBM as whole 512-byte rows while conv1 multiplies it;
//  * conv1's results go straight from registers to HBM (16 bytes per lane).
//
// Bound: HBM (80 KB per 64 rows; the matrix work of a step is 24 MFMAs per wave).
#include "rn_conv_params.h"
#include "rn_lds_dma.h"

using namespace rn_gemm;
using namespace rn_dma;

namespace {

struct ChainParams {
    const void *t2;  // [M][64]   conv3 input
    const void *x;   // [M][256]  residual; DUAL: [M][64] the second contraction input (the block's input)
    void *y;         // [M][256]  conv3 output (the block's output)
    const void *w3;  // [256][64] packed K-major
    const float *sc3, *sh3;
    void *t1;        // [M][N1]   conv1 output of the next block
    const void *w1;  // [N1][256]
    const float *sc1, *sh1;
    int M, nsteps;
    int t2_bytes, x_bytes, y_bytes, t1_bytes;
};

constexpr int kRows = 64;                          // rows of a step
constexpr int kT2 = 0, kT2Bytes = kRows * 128;     // two t2 buffers
constexpr int kX = 2 * kT2Bytes, kXBytes = kRows * 512;  // two residual buffers
constexpr int kY = kX + 2 * kXBytes;               // y tile: 4 K tiles of [64][128 B]
constexpr int kS = kY + kRows * 512;               // scale3[256] shift3[256] scale1[128] shift1[128]
constexpr int kLds = kS + (512 + 256) * 4;

typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));

// four fp32 -> two dwords of bf16 (round to nearest even)
__device__ __forceinline__ void pack4(const float (&v)[4], unsigned (&d)[2])
{
    bf16x2 a, b;
    a[0] = (bf16_t)v[0], a[1] = (bf16_t)v[1], b[0] = (bf16_t)v[2], b[1] = (bf16_t)v[3];
    d[0] = __builtin_bit_cast(unsigned, a);
    d[1] = __builtin_bit_cast(unsigned, b);
}

// N1F: 32-channel fragments of conv1's output (2: 64 channels, 4: 128).  DUAL: the first product is
// the fused conv3 + downsample pair of a stage's first block -- K = 64 + 64 from two tensors,
// batch-norm scales folded into the weight panel, no residual (rn_conv2d_nhwc_pair_forward_dt).
template <int N1F, bool DUAL>
__global__ __launch_bounds__(512, 2) void chain_kernel(const ChainParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[kLds];
    float *const ssl = reinterpret_cast<float *>(lds + kS);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    constexpr int N1 = 32 * N1F;

    // this block's steps: a contiguous range, the remainder to the first blocks
    int nst, s0;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (v < rem ? 1u : 0u));
        s0 = (int)(v * base + min(v, rem));
    }

    // weights: conv3 -- wave w owns output channels 32w .. 32w+31; conv1 -- wave w owns rows
    // 32(w&1) .. +31 of the step and channel fragment w>>1 (waves past 2*N1F have none)
    const int pf2 = wave & 1, cf2 = wave >> 1;
    const bool has2 = cf2 < N1F;  // wave-uniform
    constexpr int K3S = DUAL ? 8 : 4;  // k-steps of the first product
    i32x4 w3r[K3S], w1r[16];
    {
        const char *r3 = static_cast<const char *>(p.w3) + (size_t)(32 * wave + li) * (K3S * 32) + lh * 16;
#pragma unroll
        for (int ks = 0; ks < K3S; ++ks) w3r[ks] = *reinterpret_cast<const i32x4 *>(r3 + ks * 32);
        const char *r1 = static_cast<const char *>(p.w1) + (size_t)(32 * (has2 ? cf2 : 0) + li) * 512 + lh * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) w1r[s] = *reinterpret_cast<const i32x4 *>(r1 + s * 32);
    }
    for (int i = t; i < 256; i += 512) {
        ssl[i] = p.sc3 ? p.sc3[i] : 1.f;
        ssl[256 + i] = p.sh3 ? p.sh3[i] : -0.f;  // -0.0 keeps a -0.0 sum
    }
    for (int i = t; i < N1; i += 512) {
        ssl[512 + i] = p.sc1 ? p.sc1[i] : 1.f;
        ssl[640 + i] = p.sh1 ? p.sh1[i] : -0.f;
    }

    const i32x4 srd_t2 = make_srd(p.t2, p.t2_bytes);
    const i32x4 srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, p.t1_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // the DMA pieces of step s into buffer `buf`: one of t2 (rows 8w .. 8w+7), four of the residual
    // (piece q = 8j + w holds rows 2q, 2q+1)
    auto fetch = [&](int s, int buf) {
        const int m0 = (s0 + s) * kRows;
        const bool live = s < nst;
        {
            const int r = 8 * wave + (lane >> 3), pc = lane & 7, m = m0 + r;
            dma16((live && m < p.M) ? m * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_t2, 0,
                  lds_base + (unsigned)(kT2 + buf * kT2Bytes + wave * 1024));
        }
        if constexpr (DUAL) {  // the second input's rows, in the same operand image as t2
            const int r = 8 * wave + (lane >> 3), pc = lane & 7, m = m0 + r;
            dma16((live && m < p.M) ? m * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_x, 0,
                  lds_base + (unsigned)(kX + buf * kXBytes + wave * 1024));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 8 * j + wave;
                const int r = 2 * q + (lane >> 5), sl = lane & 31, m = m0 + r;
                const int c = (sl & ~15) | ((sl & 15) ^ (r & 15));  // global chunk held by LDS slot sl
                dma16((live && m < p.M) ? m * 512 + (c << 4) : kOob, srd_x, 0,
                      lds_base + (unsigned)(kX + buf * kXBytes + q * 1024));
            }
        }
    };

    fetch(0, 0);
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        const int m0 = (s0 + s) * kRows;
        // this step's rows have landed (and the previous step's stores are out), all waves are past
        // the previous step: its buffers take the next step's rows
        wait_and_barrier<0>();
        fetch(s + 1, buf ^ 1);

        // ---- conv3: y[64][256] = t2[64][64] . w3^T, wave w the channels 32w .. 32w+31 ----
        f32x16 acc[2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pf][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K3S; ++ks) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int r = 32 * pf + li;
                const char *src = ks < 4 ? lds + kT2 + buf * kT2Bytes : lds + kX + buf * kXBytes;
                const i32x4 px = *reinterpret_cast<const i32x4 *>(
                    src + r * 128 + (((2 * (ks & 3) + lh) ^ ((r >> 1) & 7)) << 4));
                acc[pf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[ks]),
                                                                  __builtin_bit_cast(bf16x8, px), acc[pf], 0, 0, 0);
            }
        }
        // lane (li, lh): channels 32w + 8j + 4lh + {0..3}, j = 0..3, of row 32pf + li.  Affine,
        // residual (8 bytes from the swizzled LDS rows), ReLU, bf16; the half-waves trade groups so
        // that each lane owns 8 consecutive channels = one 16-byte chunk of the y tile.
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int r = 32 * pf + li;
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = 32 * wave + 8 * j + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + c4);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 256 + c4);
                float v[4] = {fmaf(acc[pf][4 * j], sc.x, sh.x), fmaf(acc[pf][4 * j + 1], sc.y, sh.y),
                              fmaf(acc[pf][4 * j + 2], sc.z, sh.z), fmaf(acc[pf][4 * j + 3], sc.w, sh.w)};
                if constexpr (!DUAL) {
                    const int chunk = 4 * wave + j;  // 16-byte chunk of the residual row
                    const bf16x4 rv = *reinterpret_cast<const bf16x4 *>(
                        lds + kX + buf * kXBytes + r * 512 + (((chunk & ~15) | ((chunk & 15) ^ (r & 15))) << 4) + 8 * lh);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = v[k] + (float)rv[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                pack4(v, d[j]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                // channels 32w + 16h + 8lh + {0..7}: chunk 4w + 2h + lh of the 256-channel row
                const int cy = 4 * wave + 2 * h + lh;
                *reinterpret_cast<i32x4 *>(lds + kY + (cy >> 3) * (kRows * 128) + r * 128 +
                                           (((cy & 7) ^ ((r >> 1) & 7)) << 4)) =
                    i32x4{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
            }
        }
        __syncthreads();

        // ---- y leaves as whole rows ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (t >> 5) + 16 * i, c = t & 31, m = m0 + r;
            const i32x4 v = *reinterpret_cast<const i32x4 *>(lds + kY + (c >> 3) * (kRows * 128) + r * 128 +
                                                             (((c & 7) ^ ((r >> 1) & 7)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_y,
                                                   m < p.M ? m * 512 + (c << 4) : kOob, 0, 0);
        }

        // ---- conv1: t1[64][N1] = y[64][256] . w1^T ----
        if (has2) {
            const int r = 32 * pf2 + li;
            f32x16 a2;
#pragma unroll
            for (int e = 0; e < 16; ++e) a2[e] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const i32x4 px = *reinterpret_cast<const i32x4 *>(
                    lds + kY + (s2 >> 2) * (kRows * 128) + r * 128 + (((2 * (s2 & 3) + lh) ^ ((r >> 1) & 7)) << 4));
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[s2]),
                                                             __builtin_bit_cast(bf16x8, px), a2, 0, 0, 0);
            }
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = 32 * cf2 + 8 * j + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + 512 + c4);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 640 + c4);
                float v[4] = {fmaxf(fmaf(a2[4 * j], sc.x, sh.x), 0.f), fmaxf(fmaf(a2[4 * j + 1], sc.y, sh.y), 0.f),
                              fmaxf(fmaf(a2[4 * j + 2], sc.z, sh.z), 0.f), fmaxf(fmaf(a2[4 * j + 3], sc.w, sh.w), 0.f)};
                pack4(v, d[j]);
            }
            const int m = m0 + r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                const u32x4 o = {x0[0], x1[0], x0[1], x1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(
                    o, rsrc_t1, m < p.M ? (m * N1 + 32 * cf2 + 16 * h + 8 * lh) * 2 : kOob, 0, 0);
            }
        }
    }
}

int chain_launch(rn_ctx *ctx, const char *what, int dtype, const void *t2, const void *x, bool dual, void *y,
                 const void *w3, const float *scale3, const float *shift3, void *t1, const void *w1,
                 const float *scale1, const float *shift1, uint64_t rows, uint64_t mid_channels,
                 uint64_t channels, uint64_t next_mid)
{
    if (rows == 0) return RN_OK;
    RN_REQUIRE(ctx, dtype == RN_DTYPE_BF16, "bf16 storage only");
    RN_REQUIRE(ctx, t2 && x && y && w3 && t1 && w1, "null tensor");
    RN_REQUIRE(ctx, mid_channels == 64 && channels == 256 && (next_mid == 64 || next_mid == 128),
               "shapes: 64 -> 256 -> 64 | 128 channels");
    RN_REQUIRE(ctx, rows * 512 < (1ull << 31), "tensor too large");
    ChainParams p;
    p.t2 = t2, p.x = x, p.y = y, p.w3 = w3, p.sc3 = scale3, p.sh3 = shift3;
    p.t1 = t1, p.w1 = w1, p.sc1 = scale1, p.sh1 = shift1;
    p.M = (int)rows;
    p.nsteps = (int)((rows + kRows - 1) / kRows);
    p.t2_bytes = (int)(rows * 128), p.x_bytes = (int)(rows * (dual ? 128 : 512));
    p.y_bytes = (int)(rows * 512), p.t1_bytes = (int)(rows * next_mid * 2);
    const dim3 grid(p.nsteps < 256 ? p.nsteps : 256), block(512);  // one block per CU
    if (dual) {
        if (next_mid == 64)
            chain_kernel<2, true><<<grid, block, 0, ctx->stream>>>(p);
        else
            chain_kernel<4, true><<<grid, block, 0, ctx->stream>>>(p);
    } else {
        if (next_mid == 64)
            chain_kernel<2, false><<<grid, block, 0, ctx->stream>>>(p);
        else
            chain_kernel<4, false><<<grid, block, 0, ctx->stream>>>(p);
    }
    return rn_after_launch(ctx, what);
}

}  // namespace

extern "C" {

// conv3 (+ bn + residual + ReLU) of a block and conv1 (+ bn + ReLU) of the next as one launch;
// see the head of this file.  mid channels 64, block channels 256, next_mid 64 or 128.
int rn_conv_chain_forward_dt(rn_ctx *ctx, int dtype, const void *t2, const void *residual, void *y,
                             const void *packed_w3, const float *scale3, const float *shift3,
                             void *t1, const void *packed_w1, const float *scale1, const float *shift1,
                             uint64_t rows, uint64_t mid_channels, uint64_t channels, uint64_t next_mid)
{
    RN_ENTER(ctx);
    return chain_launch(ctx, "rn_conv_chain_forward_dt", dtype, t2, residual, false, y, packed_w3, scale3, shift3,
                        t1, packed_w1, scale1, shift1, rows, mid_channels, channels, next_mid);
}

// the same with the fused conv3 + downsample pair (rn_conv2d_pack_weight_pair_dt panel: scales
// folded, K = mid + in2 channels) as the first product: y = relu(t2 . w3s + x2 . wds + shift)
int rn_conv_chain_pair_forward_dt(rn_ctx *ctx, int dtype, const void *t2, const void *x2, void *y,
                                  const void *packed_pair, const float *shift, void *t1,
                                  const void *packed_w1, const float *scale1, const float *shift1,
                                  uint64_t rows, uint64_t mid_channels, uint64_t in2_channels,
                                  uint64_t channels, uint64_t next_mid)
{
    RN_ENTER(ctx);
    RN_REQUIRE(ctx, in2_channels == 64, "second input: 64 channels");
    return chain_launch(ctx, "rn_conv_chain_pair_forward_dt", dtype, t2, x2, true, y, packed_pair, nullptr, shift,
                        t1, packed_w1, scale1, shift1, rows, mid_channels, channels, next_mid);
}

}  // extern "C"
