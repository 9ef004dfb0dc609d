// conv3 of one bottleneck block chained with conv1 of the next for the 256-channel blocks of stage 3
// (bf16 storage): 256 -> 1024 (+ bn3 + residual + ReLU) and 1024 -> 256 (+ bn1 + ReLU) as one launch.
//
// Reference sequence: layerForward, cuda/inference/main.cu:131-164, twice (see rn_chain.hip).  The
// chain kernels of stage 1-2 keep both weight panels in registers; here they are 2 x 512 KB, more
// than a CU's register file, so the panels STREAM through LDS while the block's rows stay put:
//
//  * a block owns 128 rows (pixels).  Their t2 tile (128 x 256 bf16 = 64 KB) is fetched once and
//    stays in LDS as an MFMA operand image for the whole block;
//  * the 1024 channels of y are walked in eight chunks of 128.  Per chunk: the first product
//    (t2 . W3[chunk]^T, K = 256) on one fragment pair per wave, its epilogue (affine, residual,
//    ReLU, bf16) writes the y chunk into LDS as the SECOND product's operand image, and that chunk
//    is 128 of the second product's 1024 k: acc2 += y[chunk] . W1[:, chunk]^T on four fragments
//    per wave that live in registers across all eight chunks.  y leaves for HBM from the LDS image
//    as whole 256-byte runs (the next block's residual needs it); conv1 never reads it back;
//  * the weight panels arrive by LDS-DMA in 32-KB slots of a two-slot ring, four fills per chunk
//    (W3 chunk k 0..127, k 128..255, W1 chunk k-tile 0, k-tile 1), each asked for one step ahead
//    behind a counted s_waitcnt (the residual loads and the y stores stay in flight across it);
//    five barriers per chunk;
//  * operands swapped as in the other chain kernels (weights = MFMA rows, pixels = columns): a lane
//    ends with four consecutive channels of one pixel, v_permlane32_swap pairs the half-waves into
//    16-byte runs.  The residual comes straight from global memory (8 bytes per lane, asked for a
//    step before the epilogue), the channel constants likewise (L1 / L2 hits): LDS is full to the
//    byte (t2 64 KB + ring 64 KB + y chunk 32 KB).
//
// Same bits as the two launches: conv3's K tiles in order with the 32x32x16 operand map of every
// other bf16 kernel here, the same fp32 epilogue expression and one rounding to bf16, conv1's k in
// increasing channel order (chunk c = K tiles 2c, 2c + 1), its epilogue unchanged.
//
// Bound: LDS bandwidth (1.25 fragment reads per MFMA on two-fragment wave tiles) and the barriers of
// its short steps; against the two launches it saves conv1's read of y (103 MB at B = 256), one
// launch ramp and conv3's drain.
#include "rn_conv_params.h"
#include "rn_lds_dma.h"

using namespace rn_gemm;
using namespace rn_dma;

namespace {

struct WChainParams {
    const void *t2, *x;
    void *y;
    const void *w3;
    const float *sc3, *sh3;
    void *t1;
    const void *w1;
    const float *sc1, *sh1;
    int M;
    int t2_bytes, x_bytes, y_bytes, t1_bytes, w3_bytes, w1_bytes;
};

typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pack4(const float (&v)[4], unsigned (&d)[2])
{
    bf16x2 a, b;
    a[0] = (bf16_t)v[0], a[1] = (bf16_t)v[1], b[0] = (bf16_t)v[2], b[1] = (bf16_t)v[3];  // round to nearest even
    d[0] = __builtin_bit_cast(unsigned, a);
    d[1] = __builtin_bit_cast(unsigned, b);
}

constexpr int kMid = 256, kC = 1024, kN1 = 256;  // conv3: kMid -> kC, conv1: kC -> kN1
constexpr int kBM = 128, kCN = 128, kChunks = kC / kCN;
constexpr int kImg = kBM * 128;                   // one K tile (64 channels) of a 128-row operand image
constexpr int oT2 = 0, kSlot = 32 * 1024, oRing = (kMid / 64) * kImg, oY = oRing + 2 * kSlot;
constexpr int kLds = oY + (kCN / 64) * kImg;
static_assert(kLds == 160 * 1024, "LDS: t2 tile + two ring slots + y chunk");

__global__ __launch_bounds__(512, 2) void chain_wide_kernel(const WChainParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[kLds];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = (int)blockIdx.x * kBM;

    const i32x4 srd_t2 = make_srd(p.t2, p.t2_bytes);
    const i32x4 srd_w3 = make_srd(p.w3, p.w3_bytes);
    const i32x4 srd_w1 = make_srd(p.w1, p.w1_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, p.t1_bytes, 0x00020000);
    // channel constants through descriptors: an absent vector is a zero-record descriptor, so the
    // loads are issued either way (the counted waits below count them)
    const __amdgpu_buffer_rsrc_t rsrc_sc3 = __builtin_amdgcn_make_buffer_rsrc(
        p.sc3 ? (void *)const_cast<float *>(p.sc3) : p.y, 0, p.sc3 ? kC * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sh3 = __builtin_amdgcn_make_buffer_rsrc(
        p.sh3 ? (void *)const_cast<float *>(p.sh3) : p.y, 0, p.sh3 ? kC * 4 : 0, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // first product: wave w owns channel fragment cf of the chunk and pixel fragments 2pg, 2pg + 1;
    // second: channel fragments 2cf, 2cf + 1 of conv1's 256 and the same two pixel fragments
    const int cf = wave & 3, pg = wave >> 2;
    const int prow = lane >> 3, pc = lane & 7;
    const int sw = (li >> 1) & 7;  // swizzle term of this lane's fragment rows (rows 32f + li)

    // ---- LDS-DMA fills: 1-KiB pieces (8 rows x 128 B), lane l = (row l / 8, physical chunk l % 8)
    // fetching logical chunk (l % 8) ^ ((row >> 1) & 7) ----
    auto fetch_t2 = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * j + wave, kt = q >> 4, r = 8 * (q & 15) + prow, m = m0 + r;
            dma16(m < p.M ? m * (kMid * 2) + kt * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_t2, 0,
                  lds_base + (unsigned)(oT2 + q * 1024));
        }
    };
    // W3 rows c*128 .. +127, K tiles 2h and 2h + 1, as two 128-row images
    auto fill_w3 = [&](int c, int h, int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 8 * j + wave, ktl = q >> 4, r = 8 * (q & 15) + prow;
            dma16(((c * kCN + r) * kMid + (2 * h + ktl) * 64) * 2 + ((pc ^ ((r >> 1) & 7)) << 4), srd_w3, 0,
                  lds_base + (unsigned)(oRing + slot * kSlot + q * 1024));
        }
    };
    // W1 rows 0 .. 255, K tile 2c + j2 (64 of the chunk's 128 channels), as one 256-row image
    auto fill_w1 = [&](int c, int j2, int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 8 * j + wave, r = 8 * q + prow;
            dma16((r * kC + (2 * c + j2) * 64) * 2 + ((pc ^ ((r >> 1) & 7)) << 4), srd_w1, 0,
                  lds_base + (unsigned)(oRing + slot * kSlot + q * 1024));
        }
    };

    f32x16 acc2[2][2];  // [channel fragment 2cf + i][pixel fragment 2pg + k]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][k][e] = 0.f;

    // first product, K tiles kt0, kt0 + 1 of t2 against the two images of ring slot `slot`
    f32x16 acc1[2];
    auto p1_half = [&](int slot, int kt0) {
#pragma unroll
        for (int ktl = 0; ktl < 2; ++ktl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int co = ((2 * ks + lh) ^ sw) << 4;
                const i32x4 a = *reinterpret_cast<const i32x4 *>(lds + oRing + slot * kSlot + ktl * kImg +
                                                                  (32 * cf + li) * 128 + co);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const i32x4 b = *reinterpret_cast<const i32x4 *>(lds + oT2 + (kt0 + ktl) * kImg +
                                                                      (32 * (2 * pg + k) + li) * 128 + co);
                    acc1[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                                      __builtin_bit_cast(bf16x8, b), acc1[k], 0, 0, 0);
                }
            }
    };
    // second product, one K tile: W1 image in ring slot `slot`, y image K tile j2
    auto p2_tile = [&](int slot, int j2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((2 * ks + lh) ^ sw) << 4;
            i32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const i32x4 *>(lds + oRing + slot * kSlot + (32 * (2 * cf + i) + li) * 128 + co);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                b[k] = *reinterpret_cast<const i32x4 *>(lds + oY + j2 * kImg + (32 * (2 * pg + k) + li) * 128 + co);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    acc2[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                         __builtin_bit_cast(bf16x8, b[k]), acc2[i][k], 0, 0, 0);
        }
    };

    fetch_t2();
    fill_w3(0, 0, 0);
    wait_and_barrier<0>();  // the t2 tile and the first half of W3's first chunk are in LDS

#pragma unroll 1
    for (int c = 0; c < kChunks; ++c) {
        // ---- step 0: first product, k 0..127.  Asked for now: W3's second half, and the residual
        // and channel constants of this chunk's epilogue (16 loads that stay in flight) ----
        fill_w3(c, 1, 1);
        u32x2 resv[2][4];
        u32x4 scv[4], shv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = c * kCN + 32 * cf + 8 * j + 4 * lh;  // four consecutive channels
            scv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc3, ch * 4, 0, 0);
            shv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh3, ch * 4, 0, 0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int m = m0 + 32 * (2 * pg + k) + li;
                resv[k][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, m < p.M ? (m * kC + ch) * 2 : kOob, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[k][e] = 0.f;
        p1_half(0, 0);
        wait_and_barrier<16>();  // W3's second half has landed; every wave is past slot 0

        // ---- step 1: first product, k 128..255, and its epilogue into the y image ----
        fill_w1(c, 0, 0);
        p1_half(1, 2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = 32 * (2 * pg + k) + li;
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x4 rv = __builtin_bit_cast(bf16x4, resv[k][j]);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sc = p.sc3 ? __uint_as_float(scv[j][q]) : 1.f;
                    const float sh = p.sh3 ? __uint_as_float(shv[j][q]) : -0.f;  // -0.0 keeps a -0.0 sum
                    v[q] = fmaxf(fmaf(acc1[k][4 * j + q], sc, sh) + (float)rv[q], 0.f);
                }
                pack4(v, d[j]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                // channels 32cf + 16h + 8lh + {0..7} of the chunk: 16-byte chunk 4cf + 2h + lh of its row
                const int cy = 4 * cf + 2 * h + lh;
                *reinterpret_cast<i32x4 *>(lds + oY + (cy >> 3) * kImg + r * 128 + (((cy & 7) ^ ((r >> 1) & 7)) << 4)) =
                    i32x4{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
            }
        }
        wait_and_barrier<0>();  // the y chunk is complete, W1's first K tile has landed

        // ---- step 2: second product, first K tile of the chunk ----
        fill_w1(c, 1, 1);
        p2_tile(0, 0);
        wait_and_barrier<0>();

        // ---- step 3: second product, second K tile; the y chunk leaves as 256-byte runs; the next
        // chunk's first W3 half is asked for (its four pieces are older than the four stores) ----
        if (c + 1 < kChunks) fill_w3(c + 1, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = t + 512 * i, r = g >> 4, cc = g & 15, m = m0 + r;
            const i32x4 v = *reinterpret_cast<const i32x4 *>(lds + oY + (cc >> 3) * kImg + r * 128 +
                                                             (((cc & 7) ^ ((r >> 1) & 7)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_y,
                                                   m < p.M ? (m * kC + c * kCN) * 2 + (cc << 4) : kOob, 0, 0);
        }
        p2_tile(1, 1);
        wait_and_barrier<4>();  // the next chunk's W3 half has landed (the stores may still fly)
    }

    // ---- conv1's epilogue: registers -> HBM, 16 bytes per lane ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nf = 2 * cf + i, r = 32 * (2 * pg + k) + li, m = m0 + r;
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n4 = 32 * nf + 8 * j + 4 * lh;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sc = p.sc1 ? p.sc1[n4 + q] : 1.f;
                    const float sh = p.sh1 ? p.sh1[n4 + q] : -0.f;
                    v[q] = fmaxf(fmaf(acc2[i][k][4 * j + q], sc, sh), 0.f);
                }
                pack4(v, d[j]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                const u32x4 o = {x0[0], x1[0], x0[1], x1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_t1, m < p.M ? (m * kN1 + 32 * nf + 16 * h + 8 * lh) * 2 : kOob,
                                                       0, 0);
            }
        }
}

}  // namespace

// library-internal (rn_chain.hip): arguments checked by the caller (alignment, aliasing, shapes
// 256 -> 1024 -> 256, bf16)
int rn_chain_wide_launch(rn_ctx *ctx, const void *t2, const void *x, void *y, const void *w3, const float *sc3,
                         const float *sh3, void *t1, const void *w1, const float *sc1, const float *sh1,
                         uint64_t rows)
{
    WChainParams p;
    p.t2 = t2, p.x = x, p.y = y, p.w3 = w3, p.sc3 = sc3, p.sh3 = sh3;
    p.t1 = t1, p.w1 = w1, p.sc1 = sc1, p.sh1 = sh1;
    p.M = (int)rows;
    p.t2_bytes = (int)(rows * kMid * 2), p.x_bytes = p.y_bytes = (int)(rows * kC * 2);
    p.t1_bytes = (int)(rows * kN1 * 2);
    p.w3_bytes = kC * kMid * 2, p.w1_bytes = kN1 * kC * 2;
    const unsigned blocks = (unsigned)((rows + kBM - 1) / kBM);
    chain_wide_kernel<<<dim3(blocks), dim3(512), 0, ctx->stream>>>(p);
    return RN_OK;
}
