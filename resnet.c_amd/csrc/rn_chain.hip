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

// Geometry per mid-channel count MID (block channels C = 4 MID):
//   MID  64: 8 waves, 64 rows per step; a wave owns 32 of the 256 channels of the first product
//   MID 128: 4 waves, 32 rows per step; a wave owns 128 of the 512 channels -- one wave per SIMD,
//            so that its 128 + 128 weight registers and 80 accumulator registers fit the SIMD's
//            512, and a residual double buffer of 2 x 32 KB instead of 2 x 64 fits LDS
template <int MID>
struct Geo {
    static constexpr int C = 4 * MID;
    static constexpr int WAVES = MID == 64 ? 8 : 4;
    static constexpr int ROWS = MID == 64 ? 64 : 32;
    static constexpr int PF = ROWS / 32;             // 32-row fragments of a step
    static constexpr int CF1 = C / (32 * WAVES);     // channel fragments of the first product per wave
    static constexpr int KT2 = MID / 64;             // 128-byte K tiles of a t2 row
    static constexpr int T2B = ROWS * MID * 2;       // bytes of a t2 buffer
    static constexpr int XB = ROWS * C * 2;          // bytes of a residual buffer (= of the y tile)
    static constexpr int oT2 = 0, oX = 2 * T2B, oY = oX + 2 * XB, oS = oY + XB;
    static constexpr int LDS = oS + (2 * C + 512) * 4;  // + scale/shift of both products
};

// N1: channels of the second product (conv1 of the next block).  DUAL (MID 64 only): the first
// product is the fused conv3 + downsample pair of a stage's first block -- K = 64 + 64 from two
// tensors, batch-norm scales folded into the weight panel, no residual
// (rn_conv2d_nhwc_pair_forward_dt).
template <int MID, int N1, bool DUAL>
__global__ __launch_bounds__(64 * Geo<MID>::WAVES) void chain_kernel(const ChainParams p)
{
    using G = Geo<MID>;
    constexpr int C = G::C, ROWS = G::ROWS, PF = G::PF, CF1 = G::CF1, THREADS = 64 * G::WAVES;
    constexpr int N1F = N1 / 32;
    static_assert(!DUAL || MID == 64, "pair chain: 64 mid channels");
    static_assert(PF * N1F <= G::WAVES, "one fragment of the second product per wave");
    __shared__ __attribute__((aligned(16))) char lds[G::LDS];
    float *const ssl = reinterpret_cast<float *>(lds + G::oS);  // sc3[C] sh3[C] sc1[256] sh1[256]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // this block's steps: a contiguous range, the remainder to the first blocks
    int nst, s0;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (v < rem ? 1u : 0u));
        s0 = (int)(v * base + min(v, rem));
    }

    // weights.  First product: wave w owns channel fragments CF1*w .. CF1*w + CF1-1.  Second: wave w
    // owns rows 32(w % PF) .. +31 of the step and channel fragment w / PF (waves past PF*N1F: none)
    const int pf2 = wave % PF, cf2 = wave / PF;
    const bool has2 = cf2 < N1F;  // wave-uniform
    constexpr int K3S = (DUAL ? 2 : 1) * MID / 16;  // k-steps of the first product
    constexpr int K1S = C / 16;                     // ... of the second
    i32x4 w3r[CF1][K3S], w1r[K1S];
#pragma unroll
    for (int f = 0; f < CF1; ++f) {
        const char *r3 = static_cast<const char *>(p.w3) + (size_t)(32 * (CF1 * wave + f) + li) * (K3S * 32) + lh * 16;
#pragma unroll
        for (int ks = 0; ks < K3S; ++ks) w3r[f][ks] = *reinterpret_cast<const i32x4 *>(r3 + ks * 32);
    }
    {
        const char *r1 = static_cast<const char *>(p.w1) + (size_t)(32 * (has2 ? cf2 : 0) + li) * (K1S * 32) + lh * 16;
#pragma unroll
        for (int s = 0; s < K1S; ++s) w1r[s] = *reinterpret_cast<const i32x4 *>(r1 + s * 32);
    }
    for (int i = t; i < C; i += THREADS) {
        ssl[i] = p.sc3 ? p.sc3[i] : 1.f;
        ssl[C + i] = p.sh3 ? p.sh3[i] : -0.f;  // -0.0 keeps a -0.0 sum
    }
    for (int i = t; i < N1; i += THREADS) {
        ssl[2 * C + i] = p.sc1 ? p.sc1[i] : 1.f;
        ssl[2 * C + 256 + i] = p.sh1 ? p.sh1[i] : -0.f;
    }

    const i32x4 srd_t2 = make_srd(p.t2, p.t2_bytes);
    const i32x4 srd_x = make_srd(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, p.t1_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // the DMA pieces (1 KiB each) of step s into buffer `buf`.  t2: K tile kt, rows 8i .. 8i+7 in
    // the MFMA operand image; residual: consecutive 16-byte slots of the [ROWS][2C-byte] rows, slot
    // sl of row r holding the row's chunk sl ^ (r & 15) (low four bits)
    auto fetch = [&](int s, int buf) {
        const int m0 = (s0 + s) * ROWS;
        const bool live = s < nst;
        constexpr int T2P = G::KT2 * ROWS / 8;  // pieces of a t2 buffer
#pragma unroll
        for (int j = 0; j < T2P / G::WAVES; ++j) {
            const int q = G::WAVES * j + wave, kt = q / (ROWS / 8), r = 8 * (q % (ROWS / 8)) + (lane >> 3);
            const int pc = lane & 7, m = m0 + r;
            dma16((live && m < p.M) ? m * (MID * 2) + kt * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_t2, 0,
                  lds_base + (unsigned)(G::oT2 + buf * G::T2B + q * 1024));
        }
        if constexpr (DUAL) {  // the second input's rows, in the same operand image as t2
#pragma unroll
            for (int j = 0; j < T2P / G::WAVES; ++j) {
                const int q = G::WAVES * j + wave, r = 8 * q + (lane >> 3), pc = lane & 7, m = m0 + r;
                dma16((live && m < p.M) ? m * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_x, 0,
                      lds_base + (unsigned)(G::oX + buf * G::XB + q * 1024));
            }
        } else {
            constexpr int XP = G::XB / 1024, SPR = C / 8;  // pieces; 16-byte slots per row
#pragma unroll
            for (int j = 0; j < XP / G::WAVES; ++j) {
                const int q = G::WAVES * j + wave, g = 64 * q + lane;
                const int r = g / SPR, sl = g % SPR, m = m0 + r;
                const int c = (sl & ~15) | ((sl & 15) ^ (r & 15));
                dma16((live && m < p.M) ? m * (C * 2) + (c << 4) : kOob, srd_x, 0,
                      lds_base + (unsigned)(G::oX + buf * G::XB + q * 1024));
            }
        }
    };

    fetch(0, 0);
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        const int m0 = (s0 + s) * ROWS;
        // this step's rows have landed (and the previous step's stores are out), all waves are past
        // the previous step: its buffers take the next step's rows
        wait_and_barrier<0>();
        // With one wave per SIMD (MID 128) the ten DMA issues would stand in front of the step's first
        // MFMA: they go out behind the first product instead (they still have the rest of the step)
        if constexpr (MID == 64) fetch(s + 1, buf ^ 1);

        // ---- first product: y[ROWS][C] = t2[ROWS][MID] . w3^T (+ second input . wd^T) ----
        f32x16 acc[PF][CF1];
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int f = 0; f < CF1; ++f)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pf][f][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < K3S; ++ks) {
#pragma unroll
            for (int pf = 0; pf < PF; ++pf) {
                const int r = 32 * pf + li;
                const bool second = DUAL && ks >= MID / 16;
                const int kk = second ? ks - MID / 16 : ks;
                const char *src = second ? lds + G::oX + buf * G::XB : lds + G::oT2 + buf * G::T2B;
                const i32x4 px = *reinterpret_cast<const i32x4 *>(
                    src + (kk >> 2) * (ROWS * 128) + r * 128 + (((2 * (kk & 3) + lh) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
                for (int f = 0; f < CF1; ++f)
                    acc[pf][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[f][ks]),
                                                                         __builtin_bit_cast(bf16x8, px), acc[pf][f], 0, 0, 0);
            }
        }
        if constexpr (MID != 64) fetch(s + 1, buf ^ 1);
        // lane (li, lh): channels 32cf + 8j + 4lh + {0..3}, j = 0..3, of row 32pf + li.  Affine,
        // residual (8 bytes from the swizzled LDS rows), ReLU, bf16; the half-waves trade groups so
        // that each lane owns 8 consecutive channels = one 16-byte chunk of the y tile.
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
#pragma unroll
            for (int f = 0; f < CF1; ++f) {
                const int r = 32 * pf + li, cf = CF1 * wave + f;
                unsigned d[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c4 = 32 * cf + 8 * j + 4 * lh;
                    const float4 sc = *reinterpret_cast<const float4 *>(ssl + c4);
                    const float4 sh = *reinterpret_cast<const float4 *>(ssl + C + c4);
                    float v[4] = {fmaf(acc[pf][f][4 * j], sc.x, sh.x), fmaf(acc[pf][f][4 * j + 1], sc.y, sh.y),
                                  fmaf(acc[pf][f][4 * j + 2], sc.z, sh.z), fmaf(acc[pf][f][4 * j + 3], sc.w, sh.w)};
                    if constexpr (!DUAL) {
                        const int chunk = 4 * cf + j;  // 16-byte chunk of the residual row
                        const bf16x4 rv = *reinterpret_cast<const bf16x4 *>(
                            lds + G::oX + buf * G::XB + r * (C * 2) +
                            (((chunk & ~15) | ((chunk & 15) ^ (r & 15))) << 4) + 8 * lh);
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
                    // channels 32cf + 16h + 8lh + {0..7}: chunk 4cf + 2h + lh of the C-channel row
                    const int cy = 4 * cf + 2 * h + lh;
                    *reinterpret_cast<i32x4 *>(lds + G::oY + (cy >> 3) * (ROWS * 128) + r * 128 +
                                               (((cy & 7) ^ ((r >> 1) & 7)) << 4)) =
                        i32x4{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
                }
            }
        __syncthreads();

        // ---- y leaves as whole rows ----
#pragma unroll
        for (int i = 0; i < ROWS * (C / 8) / THREADS; ++i) {
            const int g = t + i * THREADS, r = g / (C / 8), c = g % (C / 8), m = m0 + r;
            const i32x4 v = *reinterpret_cast<const i32x4 *>(lds + G::oY + (c >> 3) * (ROWS * 128) + r * 128 +
                                                             (((c & 7) ^ ((r >> 1) & 7)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_y,
                                                   m < p.M ? m * (C * 2) + (c << 4) : kOob, 0, 0);
        }

        // ---- second product: t1[ROWS][N1] = y[ROWS][C] . w1^T ----
        if (has2) {
            const int r = 32 * pf2 + li;
            f32x16 a2;
#pragma unroll
            for (int e = 0; e < 16; ++e) a2[e] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < K1S; ++s2) {
                const i32x4 px = *reinterpret_cast<const i32x4 *>(
                    lds + G::oY + (s2 >> 2) * (ROWS * 128) + r * 128 + (((2 * (s2 & 3) + lh) ^ ((r >> 1) & 7)) << 4));
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[s2]),
                                                             __builtin_bit_cast(bf16x8, px), a2, 0, 0, 0);
            }
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c4 = 32 * cf2 + 8 * j + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + 2 * C + c4);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 2 * C + 256 + c4);
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

template <int MID, int N1, bool DUAL>
void chain_go(rn_ctx *ctx, const ChainParams &p)
{
    const int blocks = p.nsteps < 256 ? p.nsteps : 256;  // one block per CU
    chain_kernel<MID, N1, DUAL><<<dim3(blocks), dim3(64 * Geo<MID>::WAVES), 0, ctx->stream>>>(p);
}

// ---- fp32 ---------------------------------------------------------------------------------------
// The same chain with fp32 storage (64 mid channels): 8 waves, 64 rows per step, weights as
// v_mfma_f32_32x32x2_f32 operands in registers (conv3 32, conv1 128 per lane).  In fp32 both products
// are matrix-bound (a step is 2 x 8,192 MFMA cycles per SIMD against 7.4 us of HBM time for its 160
// KB), so what the chain saves is the second launch's own read of y and its ramp.  Differences from
// the bf16 kernel: the D registers of a lane are four consecutive fp32 channels = 16 bytes already,
// no lane exchange; the residual tile is fetched INTO the y tile's place in LDS (same operand image)
// and the epilogue updates it in place, so two 64 KB buffers serve as residual double buffer and y
// tile; t2 has one buffer, refilled behind the mid-step barrier.  k pairs per MFMA exactly as
// conv_gemm_kernel<float>: lane half lh holds k = 8ks + 4lh + j for the j-th MFMA of k-step ks.
constexpr int k32XY = 0, k32XYB = 64 * 1024, k32T2 = 2 * k32XYB, k32T2B = 64 * 256;
constexpr int k32S = k32T2 + k32T2B, k32Lds = k32S + (512 + 256) * 4;

template <int N1>
__global__ __launch_bounds__(512) void chain32_kernel(const ChainParams p)
{
    constexpr int ROWS = 64, N1F = N1 / 32;
    __shared__ __attribute__((aligned(16))) char lds[k32Lds];
    float *const ssl = reinterpret_cast<float *>(lds + k32S);  // sc3[256] sh3[256] sc1[128] sh1[128]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int nst, s0;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (v < rem ? 1u : 0u));
        s0 = (int)(v * base + min(v, rem));
    }
    const int pf2 = wave & 1, cf2 = wave >> 1;
    const bool has2 = cf2 < N1F;  // wave-uniform
    u32x4 w3r[2][4], w1r[8][4];
    {
        const char *r3 = static_cast<const char *>(p.w3) + (size_t)(32 * wave + li) * 256 + lh * 16;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w3r[kt][ks] = *reinterpret_cast<const u32x4 *>(r3 + kt * 128 + ks * 32);
        const char *r1 = static_cast<const char *>(p.w1) + (size_t)(32 * (has2 ? cf2 : 0) + li) * 1024 + lh * 16;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w1r[kt][ks] = *reinterpret_cast<const u32x4 *>(r1 + kt * 128 + ks * 32);
    }
    for (int i = t; i < 256; i += 512) {
        ssl[i] = p.sc3 ? p.sc3[i] : 1.f;
        ssl[256 + i] = p.sh3 ? p.sh3[i] : -0.f;
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
    const int prow = lane >> 3, pc = lane & 7;

    // piece q of an operand image with KT 128-byte K tiles per row: K tile q / 8, rows 8(q % 8) ..
    auto fetch_t2 = [&](int s) {
        const int m0 = (s0 + s) * ROWS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 8 * j + wave, r = 8 * (q & 7) + prow, m = m0 + r;
            dma16((s < nst && m < p.M) ? m * 256 + (q >> 3) * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_t2, 0,
                  lds_base + (unsigned)(k32T2 + q * 1024));
        }
    };
    auto fetch_x = [&](int s, int buf) {
        const int m0 = (s0 + s) * ROWS;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 8 * j + wave, r = 8 * (q & 7) + prow, m = m0 + r;
            dma16((s < nst && m < p.M) ? m * 1024 + (q >> 3) * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob, srd_x, 0,
                  lds_base + (unsigned)(k32XY + buf * k32XYB + q * 1024));
        }
    };

    fetch_t2(0);
    fetch_x(0, 0);
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        const int m0 = (s0 + s) * ROWS;
        char *const xy = lds + k32XY + buf * k32XYB;
        wait_and_barrier<0>();   // t2 and the residual rows of this step are in LDS; everyone is past step s-1
        fetch_x(s + 1, buf ^ 1);  // (that buffer was step s-1's y tile)

        // ---- conv3: wave w the channels 32w .. 32w+31 of both 32-row fragments ----
        f32x16 acc[2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pf][e] = 0.f;
        // (fragment-major: the first fragment's accumulators are final while the second is still being
        // multiplied, so its epilogue below can be scheduled among those MFMAs)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = 32 * pf + li;
                    const u32x4 px = *reinterpret_cast<const u32x4 *>(
                        lds + k32T2 + kt * 8192 + r * 128 + (((2 * ks + lh) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[pf] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w3r[kt][ks][j]),
                                                                        __uint_as_float(px[j]), acc[pf], 0, 0, 0);
                }
        // lane (li, lh): channels 32w + 8g + 4lh + {0..3}, g = 0..3, of row 32pf + li: chunk 2g + lh of K
        // tile w of the y tile.  The residual sits there; the result replaces it.
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int r = 32 * pf + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = 32 * wave + 8 * g + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + c4);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 256 + c4);
                float4 *slot = reinterpret_cast<float4 *>(xy + wave * 8192 + r * 128 + (((2 * g + lh) ^ ((r >> 1) & 7)) << 4));
                const float4 rv = *slot;
                float4 o;
                o.x = fmaxf(fmaf(acc[pf][4 * g], sc.x, sh.x) + rv.x, 0.f);
                o.y = fmaxf(fmaf(acc[pf][4 * g + 1], sc.y, sh.y) + rv.y, 0.f);
                o.z = fmaxf(fmaf(acc[pf][4 * g + 2], sc.z, sh.z) + rv.z, 0.f);
                o.w = fmaxf(fmaf(acc[pf][4 * g + 3], sc.w, sh.w) + rv.w, 0.f);
                *slot = o;
            }
        }
        __syncthreads();
        fetch_t2(s + 1);  // every wave has read this step's t2

        // ---- y leaves as whole 1-KB rows (rounds of four chunks per thread: registers).  With 64
        // channels in conv1 only waves 0-3 have a fragment of it: the other four carry y out alone
        // and conv1 starts at once ----
        {
            constexpr bool kSplit = N1F == 2;
            const int tt = kSplit ? t - 256 : t;
            if (!kSplit || wave >= 4) {
#pragma unroll 1
                for (int i0 = 0; i0 < (kSplit ? 16 : 8); i0 += 4) {
#pragma unroll
                    for (int i = i0; i < i0 + 4; ++i) {
                        const int g = tt + i * (kSplit ? 256 : 512), r = g >> 6, c = g & 63, m = m0 + r;
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(xy + (c >> 3) * 8192 + r * 128 +
                                                                         (((c & 7) ^ ((r >> 1) & 7)) << 4));
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, m < p.M ? m * 1024 + (c << 4) : kOob, 0, 0);
                    }
                }
            }
        }

        // ---- conv1 ----
        if (has2) {
            const int r = 32 * pf2 + li;
            f32x16 a2;
#pragma unroll
            for (int e = 0; e < 16; ++e) a2[e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 px = *reinterpret_cast<const u32x4 *>(
                        xy + kt * 8192 + r * 128 + (((2 * ks + lh) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w1r[kt][ks][j]), __uint_as_float(px[j]),
                                                                  a2, 0, 0, 0);
                }
            const int m = m0 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = 32 * cf2 + 8 * g + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + 512 + c4);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 640 + c4);
                u32x4 o;
                o[0] = __float_as_uint(fmaxf(fmaf(a2[4 * g], sc.x, sh.x), 0.f));
                o[1] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 1], sc.y, sh.y), 0.f));
                o[2] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 2], sc.z, sh.z), 0.f));
                o[3] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 3], sc.w, sh.w), 0.f));
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_t1, m < p.M ? (m * N1 + c4) * 4 : kOob, 0, 0);
            }
        }
    }
}

// fp32 chain out of the fused conv3 + downsample pair (first block of stage 1): K = 64 + 64 from two
// tensors, scales folded into the panel, no residual.  With 64 registers for the pair's panel there is
// no room for conv1's 128, so conv1's weights live in LDS as an operand image (64 KB, fetched once);
// LDS is then full to the byte (y 64 + w1 64 + t2 16 + second input 16 KB), and the channel
// constants come from global memory (L1 hits) in the epilogues.
constexpr int kP32Y = 0, kP32W1 = 64 * 1024, kP32T2 = 128 * 1024, kP32X2 = 144 * 1024, kP32Lds = 160 * 1024;

__global__ __launch_bounds__(512) void chain32_pair_kernel(const ChainParams p)
{
    constexpr int ROWS = 64;
    __shared__ __attribute__((aligned(16))) char lds[kP32Lds];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int nst, s0;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (v < rem ? 1u : 0u));
        s0 = (int)(v * base + min(v, rem));
    }
    const int pf2 = wave & 1, cf2 = wave >> 1;
    const bool has2 = cf2 < 2;  // 64 channels of conv1: four fragments, waves 0-3
    u32x4 w3r[4][4];            // K tiles 0-1: conv3's rows of the panel, 2-3: the downsample's
    {
        const char *r3 = static_cast<const char *>(p.w3) + (size_t)(32 * wave + li) * 512 + lh * 16;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w3r[kt][ks] = *reinterpret_cast<const u32x4 *>(r3 + kt * 128 + ks * 32);
    }
    const i32x4 srd_t2 = make_srd(p.t2, p.t2_bytes);
    const i32x4 srd_x = make_srd(p.x, p.x_bytes);
    const i32x4 srd_w1 = make_srd(p.w1, 64 * 256 * 4);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_t1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, p.t1_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);
    const int prow = lane >> 3, pc = lane & 7;

    // conv1's panel [64][256] as an operand image: K tile q / 8, rows 8(q % 8) ..
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = 8 * j + wave, r = 8 * (q & 7) + prow;
        dma16(r * 1024 + (q >> 3) * 128 + ((pc ^ ((r >> 1) & 7)) << 4), srd_w1, 0,
              lds_base + (unsigned)(kP32W1 + q * 1024));
    }
    auto fetch_in = [&](int s) {  // both 64-channel inputs of step s (one buffer each)
        const int m0 = (s0 + s) * ROWS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = 8 * j + wave, r = 8 * (q & 7) + prow, m = m0 + r;
            const int off = (s < nst && m < p.M) ? m * 256 + (q >> 3) * 128 + ((pc ^ ((r >> 1) & 7)) << 4) : kOob;
            dma16(off, srd_t2, 0, lds_base + (unsigned)(kP32T2 + q * 1024));
            dma16(off, srd_x, 0, lds_base + (unsigned)(kP32X2 + q * 1024));
        }
    };

    fetch_in(0);
    for (int s = 0; s < nst; ++s) {
        const int m0 = (s0 + s) * ROWS;
        wait_and_barrier<0>();  // this step's inputs are in LDS; everyone is past step s-1 (its y tile is free)

        f32x16 acc[2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pf][e] = 0.f;
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)  // fragment-major, as in chain32_kernel
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = 32 * pf + li;
                    const u32x4 px = *reinterpret_cast<const u32x4 *>(
                        lds + (kt < 2 ? kP32T2 : kP32X2) + (kt & 1) * 8192 + r * 128 + (((2 * ks + lh) ^ ((r >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[pf] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w3r[kt][ks][j]),
                                                                        __uint_as_float(px[j]), acc[pf], 0, 0, 0);
                }
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) {
            const int r = 32 * pf + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = 32 * wave + 8 * g + 4 * lh;
                const float4 sh = p.sh3 ? *reinterpret_cast<const float4 *>(p.sh3 + c4) : make_float4(-0.f, -0.f, -0.f, -0.f);
                float4 o;  // the pair's epilogue: scale folded into the panel, shift, no residual, ReLU
                o.x = fmaxf(fmaf(acc[pf][4 * g], 1.f, sh.x), 0.f);
                o.y = fmaxf(fmaf(acc[pf][4 * g + 1], 1.f, sh.y), 0.f);
                o.z = fmaxf(fmaf(acc[pf][4 * g + 2], 1.f, sh.z), 0.f);
                o.w = fmaxf(fmaf(acc[pf][4 * g + 3], 1.f, sh.w), 0.f);
                *reinterpret_cast<float4 *>(lds + kP32Y + wave * 8192 + r * 128 + (((2 * g + lh) ^ ((r >> 1) & 7)) << 4)) = o;
            }
        }
        __syncthreads();
        fetch_in(s + 1);  // every wave has read this step's inputs

        if (wave >= 4) {  // waves 0-3 go straight to conv1; these four carry y out
#pragma unroll 1
            for (int i0 = 0; i0 < 16; i0 += 4) {
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) {
                    const int g = t - 256 + i * 256, r = g >> 6, c = g & 63, m = m0 + r;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(lds + kP32Y + (c >> 3) * 8192 + r * 128 +
                                                                     (((c & 7) ^ ((r >> 1) & 7)) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_y, m < p.M ? m * 1024 + (c << 4) : kOob, 0, 0);
                }
            }
        }

        if (has2) {
            const int r = 32 * pf2 + li, wr = 32 * cf2 + li;
            f32x16 a2;
#pragma unroll
            for (int e = 0; e < 16; ++e) a2[e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 px = *reinterpret_cast<const u32x4 *>(
                        lds + kP32Y + kt * 8192 + r * 128 + (((2 * ks + lh) ^ ((r >> 1) & 7)) << 4));
                    const u32x4 wv = *reinterpret_cast<const u32x4 *>(
                        lds + kP32W1 + kt * 8192 + wr * 128 + (((2 * ks + lh) ^ ((wr >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[j]), __uint_as_float(px[j]), a2, 0, 0, 0);
                }
            const int m = m0 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = 32 * cf2 + 8 * g + 4 * lh;
                const float4 sc = p.sc1 ? *reinterpret_cast<const float4 *>(p.sc1 + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float4 sh = p.sh1 ? *reinterpret_cast<const float4 *>(p.sh1 + c4) : make_float4(-0.f, -0.f, -0.f, -0.f);
                u32x4 o;
                o[0] = __float_as_uint(fmaxf(fmaf(a2[4 * g], sc.x, sh.x), 0.f));
                o[1] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 1], sc.y, sh.y), 0.f));
                o[2] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 2], sc.z, sh.z), 0.f));
                o[3] = __float_as_uint(fmaxf(fmaf(a2[4 * g + 3], sc.w, sh.w), 0.f));
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_t1, m < p.M ? (m * 64 + c4) * 4 : kOob, 0, 0);
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
    RN_REQUIRE(ctx, dtype == RN_DTYPE_BF16 || dtype == RN_DTYPE_F32, "unknown dtype");
    RN_REQUIRE(ctx, t2 && x && y && w3 && t1 && w1, "null tensor");
    // the kernels move 16-byte pieces (LDS-DMA, b128 buffer loads and stores, float4 loads of the
    // folded batch-norm constants): a misaligned pointer would fault or silently shift data
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(t2) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                      reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(w3) | reinterpret_cast<uintptr_t>(w1) |
                      reinterpret_cast<uintptr_t>(scale3) | reinterpret_cast<uintptr_t>(shift3) |
                      reinterpret_cast<uintptr_t>(scale1) | reinterpret_cast<uintptr_t>(shift1)) & 15) == 0,
               "tensors, panels and scale / shift vectors must be 16-byte aligned");
    // a block reads t2 / x rows one step ahead of the rows it writes: an output on top of an input
    // (or y on top of t1) is read back half-written
    RN_REQUIRE(ctx, y != t1 && y != t2 && y != x && t1 != t2 && t1 != x, "y and t1 must not alias each other or an input");
    const bool s1 = mid_channels == 64 && channels == 256 && (next_mid == 64 || next_mid == 128);
    const bool s2 = mid_channels == 128 && channels == 512 && next_mid == 128 && !dual && dtype == RN_DTYPE_BF16;
    RN_REQUIRE(ctx, s1 || s2, "shapes: 64 -> 256 -> 64 | 128 channels, or (bf16) 128 -> 512 -> 128");
    RN_REQUIRE(ctx, !(dual && dtype == RN_DTYPE_F32 && next_mid != 64), "fp32 pair chain: next_mid 64");
    const uint64_t es = dtype == RN_DTYPE_BF16 ? 2 : 4;
    RN_REQUIRE(ctx, rows * channels * es < (1ull << 31), "tensor too large");
    if (dtype == RN_DTYPE_F32) {
        ChainParams q;
        q.t2 = t2, q.x = x, q.y = y, q.w3 = w3, q.sc3 = scale3, q.sh3 = shift3;
        q.t1 = t1, q.w1 = w1, q.sc1 = scale1, q.sh1 = shift1;
        q.M = (int)rows;
        q.nsteps = (int)((rows + 63) / 64);
        q.t2_bytes = (int)(rows * 256), q.x_bytes = q.y_bytes = (int)(rows * 1024), q.t1_bytes = (int)(rows * next_mid * 4);
        const dim3 grid(q.nsteps < 256 ? q.nsteps : 256), block(512);
        if (dual) {
            q.x_bytes = (int)(rows * 256);
            chain32_pair_kernel<<<grid, block, 0, ctx->stream>>>(q);
        } else if (next_mid == 64)
            chain32_kernel<64><<<grid, block, 0, ctx->stream>>>(q);
        else
            chain32_kernel<128><<<grid, block, 0, ctx->stream>>>(q);
        return rn_after_launch(ctx, what);
    }
    ChainParams p;
    p.t2 = t2, p.x = x, p.y = y, p.w3 = w3, p.sc3 = scale3, p.sh3 = shift3;
    p.t1 = t1, p.w1 = w1, p.sc1 = scale1, p.sh1 = shift1;
    p.M = (int)rows;
    const uint64_t step = s1 ? 64 : 32;
    p.nsteps = (int)((rows + step - 1) / step);
    p.t2_bytes = (int)(rows * mid_channels * 2), p.x_bytes = (int)(rows * (dual ? 128 : channels * 2));
    p.y_bytes = (int)(rows * channels * 2), p.t1_bytes = (int)(rows * next_mid * 2);
    if (s2)
        chain_go<128, 128, false>(ctx, p);
    else if (dual)
        next_mid == 64 ? chain_go<64, 64, true>(ctx, p) : chain_go<64, 128, true>(ctx, p);
    else
        next_mid == 64 ? chain_go<64, 64, false>(ctx, p) : chain_go<64, 128, false>(ctx, p);
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
