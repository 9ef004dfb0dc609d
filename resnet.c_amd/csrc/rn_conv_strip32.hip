// fp32 strip kernel: 3x3 / stride 1 / padding 1 convolution with 64 input and 64 output channels
// -- conv2 of the first stage's bottleneck blocks (cuda/inference/main.cu:64-65 builds them,
// conv2dForwardKernel cuda/ops.cu:14-48 computes them), 3 of ResNet-50's 16 3x3 layers, 34.7 GFLOP
// per launch at B = 256.
//
// The construction of conv_strip128_kernel (rn_conv_wide.hip), whose bytes it shares: an fp32 pixel
// of 64 channels is the 256 bytes a bf16 pixel of 128 channels is, a weight row [3][3][64] fp32 the
// 2,304 bytes of [3][3][128] bf16.  What differs is the matrix instruction: v_mfma_f32_32x32x2_f32
// takes one float per lane where the bf16 form takes eight values, so a 16-byte fragment read feeds
// FOUR MFMAs of 64 cycles instead of one of 32 -- the loop is eight times more matrix-bound for
// the same bytes and a single wave per SIMD issues back to back.
//
//  * persistent, 4 waves per block, one per SIMD, up to 512 registers each.  The weights of a wave's
//    32 output channels are MFMA operands held in registers for the block's lifetime: 72 k-step
//    groups x 4 floats = 288 VGPRs (fetched once through LDS);
//  * a block walks a contiguous range of the flattened zero-PADDED image (pitch W + 2, one zero row
//    between images) in steps of 128 positions: there tap (kh, kw) is the constant shift
//    (kh-1)(W+2) + (kw-1), and border taps read zeros that really are in LDS -- no masks; the
//    5 % padding positions are multiplied and dropped;
//  * the input sits in a rolling LDS ring of 384 positions x two 128-byte channel segments filled
//    by LDS-DMA (range check = zeros for padding), every input row enters LDS once per block;
//    one barrier per step = per 576 MFMAs of a wave;
//  * waves 0, 1 multiply the step's positions 0..63 for channels 0..31 / 32..63, waves 2, 3
//    positions 64..127: two position fragments x one channel fragment each;
//  * operands swapped (weights = rows, pixels = columns): a lane ends with 16 channels of one pixel
//    in four groups of four consecutive channels -- four 16-byte stores straight from the
//    registers, issued among the next step's MFMAs; no LDS staging of the output.
//
// Same k order and products as conv_gemm_kernel<float> (K tiles (tap, segment) in order, inside a
// tile the k pairs (8s + j, 8s + 4 + j)), the same epilogue expression: the same bits; K = 576 is
// 18 K tiles, below the chunked-sum threshold.  Tuner candidate "strip".
//
// Measured at B = 256 (56 x 56): 491 us = 120.5 TFLOP/s against 500 us of the best tile -- no gain,
// and the reason is the finding: PMC shows this kernel and the tile kernel at the same matrix-pipe
// duty (3.32 of 4 SIMDs busy per busy-CU cycle, clock 2.25-2.3 GHz) although this one issues its
// MFMAs back to back (see the ISA: eight MFMAs between two pairs of ds_read_b128, nothing else).
// Lab builds of this loop (-DS32_LAB_*): without the fragment reads 476 us, without the 16
// vector-memory instructions of a step (8 LDS-DMA pieces, 8 stores) 447 us, without both 427 us =
// 138.6 TFLOP/s -- and with those 16 instructions ISSUED but given out-of-range offsets (they move
// nothing) 437-439 us.  So it is not their issue slots that cost the matrix stream 12 %: it is the
// 0.84 TB/s of operands and results they move.  A kernel that multiplies at this density AND
// streams from HBM is held at 120-125 TFLOP/s by the chip (the same 82-83 % matrix-pipe duty shows
// in the tile kernel and in the fused stem), whatever its instruction stream looks like.
#include "rn_conv_params.h"
#include "rn_lds_dma.h"

using namespace rn_gemm;
using namespace rn_dma;

namespace {

struct Strip32Params {
    const void *in, *w;
    void *out;
    const float *scale, *shift;
    int relu;
    int B, H, W;
    int Wp, Hq;  // W + 2, H + 1
    unsigned mul_wp, shr_wp, mul_hq, shr_hq;
    int U;       // padded positions: (B * Hq + 1) * Wp
    int nsteps;  // ceil(U / 128)
    int in_bytes, out_bytes;
};

constexpr int kMargin = 64;              // ring position of a block's first output position; >= W + 3
constexpr int kRingP = 256 + 2 * kMargin;  // positions in the ring: this step's with both margins + the next step's
constexpr int kSegB = kRingP * 128;      // bytes of one channel segment's image of the ring
constexpr int kPre = (128 + 2 * kMargin) / 8;  // pieces of 8 positions the first step needs, per segment

__global__ __launch_bounds__(256) void conv_strip32_kernel(const Strip32Params p)
{
    __shared__ __attribute__((aligned(16))) char lds[64 * 145 * 16];  // weight staging (148 KB) >= 2 * kSegB (96 KB)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int prow = lane >> 3, pc = lane & 7;
    const int cf = wave & 1, ph = wave >> 1;  // channel fragment, half of the step's positions
    int nst, ub;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned q = total >> 3, r = total & 7, xcd = v & 7;
        const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (logical < rem ? 1u : 0u));
        ub = (int)(logical * base + min(logical, rem)) * 128;
    }
    const i32x4 srd_in = make_srd(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // weights [64][576] fp32 = 64 rows of 144 chunks; LDS rows of 145 chunks (145 r mod 16 differs for
    // the 16 rows of a ds_read_b128 lane group)
    i32x4 wreg[72];
    {
        const i32x4 srd_w = make_srd(p.w, 64 * 576 * 4);
#pragma unroll 1
        for (int q = wave; q < 145; q += 4) {  // 64 rows x 145 chunks = 145 pieces of 64 chunks
            const int pos = q * 64 + lane, r = (pos * 3616) >> 19, c = pos - r * 145;  // pos / 145, pos < 9,280
            dma16(c < 144 ? (r * 144 + c) * 16 : kOob, srd_w, 0,
                  (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(q * 1024))));
        }
        wait_and_barrier<0>();
        const char *wrow = lds + (32 * cf + li) * (145 * 16) + lh * 16;
#pragma unroll
        for (int s = 0; s < 72; ++s) wreg[s] = *reinterpret_cast<const i32x4 *>(wrow + s * 32);
        __syncthreads();  // the staging area becomes the ring
    }
    // channel constants of this lane's 16 output channels (D map: 32 cf + 8j + 4lh + {0..3})
    float4 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = 32 * cf + 8 * j + 4 * lh;
        sc[j] = p.scale ? *reinterpret_cast<const float4 *>(p.scale + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[j] = p.shift ? *reinterpret_cast<const float4 *>(p.shift + c4) : make_float4(-0.f, -0.f, -0.f, -0.f);
    }

    auto pixel_of = [&](int u) -> int {
        if (u < 0 || u >= p.U) return -1;
        const unsigned R = __umulhi((unsigned)u, p.mul_wp) >> p.shr_wp;
        const int cc = u - (int)R * p.Wp;
        const unsigned b = __umulhi(R, p.mul_hq) >> p.shr_hq;
        const int rr = (int)R - (int)b * p.Hq;
        if (cc < 1 || cc > p.W || rr < 1) return -1;
        return ((int)b * p.H + rr - 1) * p.W + cc - 1;
    };
    // source of this lane's 16 bytes of a piece (8 ring positions from ring position slot, channel segment seg)
    auto src_off = [&](int u, int slot, int seg) -> int {
        const int g = pixel_of(u);
        return g < 0 ? kOob : g * 256 + seg * 128 + ((pc ^ ((slot >> 1) & 7)) << 4);
    };
    // ring positions 0 .. 255 = padded positions ub - 64 .. ub + 191: all of step 0's
#pragma unroll 1
    for (int q = wave; q < 2 * kPre; q += 4) {
        const int seg = q / kPre, g8 = q - seg * kPre, slot = 8 * g8 + prow;
        dma16(src_off(ub - kMargin + slot, slot, seg), srd_in, 0,
              (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(seg * kSegB + g8 * 1024))));
    }

    // results of the previous step wait in registers and leave among this step's MFMAs
    u32x4 pend[2][4];
    int pend_off[2] = {kOob, kOob};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) pend[i][j] = u32x4{0, 0, 0, 0};
    auto store_piece = [&](int q) {
        const int i = q >> 2, j = q & 3;
        __builtin_amdgcn_raw_buffer_store_b128(pend[i][j], rsrc_o, pend_off[i] == kOob ? kOob : pend_off[i] + 32 * j, 0, 0);
    };

    int relbase = 0;  // (128 * s) mod kRingP
    for (int s = 0; s < nst; ++s) {
        wait_and_barrier<0>();
        // next step's 128 positions: 32 pieces (2 segments x 16), eight per wave, issued among the MFMAs
        int nxt_off[8];
        unsigned nxt_dst[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 4 * j + wave, seg = q >> 4, g8 = q & 15;
            int slot0 = relbase + 128 + 2 * kMargin + 8 * g8;
            slot0 = slot0 >= kRingP ? slot0 - kRingP : slot0;
            const int u = ub - kMargin + 128 * s + 128 + 2 * kMargin + 8 * g8 + prow;
            nxt_off[j] = s + 1 < nst ? src_off(u, slot0 + prow, seg) : kOob;
#ifdef S32_LAB_OOB  // lab: the step's vector-memory instructions are issued but move nothing
            nxt_off[j] = kOob;
#endif
            nxt_dst[j] = lds_base + (unsigned)(seg * kSegB + slot0 * 128);
        }

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        // 144 fragment reads, each feeding four MFMAs: n = 2 * kstep + i, kstep = (tap * 2 + seg) * 4 + ks;
        // reads kDepth ahead through a ring of fragment registers, pinned where they are written
        constexpr int kDepth = 6;
        i32x4 px[kDepth];
        int y[9][2];
        auto read = [&](int n) -> i32x4 {
            const int i = n & 1, kstep = n >> 1, ks = kstep & 3, seg = (kstep >> 2) & 1, tap = kstep >> 3;
            if (ks == 0 && seg == 0) {
                const int shift = (tap / 3 - 1) * p.Wp + (tap % 3 - 1);
                int sb = relbase + kMargin + 64 * ph + 32 * i + shift;  // scalar, >= 0
                sb = sb >= kRingP ? sb - kRingP : sb;
                const unsigned r0 = (unsigned)(sb + li);
                const unsigned r = min(r0, r0 - (unsigned)kRingP);
                y[tap][i] = (int)((r << 7) | (((unsigned)lh ^ ((r >> 1) & 7u)) << 4));
            }
            return *reinterpret_cast<const i32x4 *>(lds + seg * kSegB + (y[tap][i] ^ (ks << 5)));
        };
#pragma unroll
        for (int n = 0; n < kDepth; ++n) px[n] = read(n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks_ = 0; ks_ < 72; ++ks_) {
            // the two position fragments share the k-step group's weights; their MFMAs alternate, so no
            // instruction waits for the accumulator of the one in front of it
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(wreg[ks_][j]),
                                                                 __int_as_float(px[(2 * ks_ + i) % kDepth][j]), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#ifndef S32_LAB_NOREAD  // lab builds only (tools: A/B of the loop with parts switched off; results are garbage)
                if (2 * ks_ + i + kDepth < 144) px[(2 * ks_ + i) % kDepth] = read(2 * ks_ + i + kDepth);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks_ % 9 == 2 || ks_ % 9 == 6) {  // 16 slots: eight DMA pieces, then the eight stores of the previous step
                const int k = (ks_ / 9) * 2 + (ks_ % 9 == 6 ? 1 : 0);
#ifndef S32_LAB_NOVMEM
                if (k < 8)
                    dma16(nxt_off[k], srd_in, 0, (unsigned)__builtin_amdgcn_readfirstlane((int)nxt_dst[k]));
                else
                    store_piece(k - 8);
#endif
            }
        }

#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {fmaf(acc[i][4 * j], sc[j].x, sh[j].x), fmaf(acc[i][4 * j + 1], sc[j].y, sh[j].y),
                              fmaf(acc[i][4 * j + 2], sc[j].z, sh[j].z), fmaf(acc[i][4 * j + 3], sc[j].w, sh[j].w)};
#pragma unroll
                for (int c = 0; c < 4; ++c) pend[i][j][c] = __float_as_uint(p.relu ? fmaxf(v[c], 0.f) : v[c]);
            }
            const int g = pixel_of(ub + 128 * s + 64 * ph + 32 * i + li);
            pend_off[i] = g < 0 ? kOob : g * 256 + (32 * cf + 4 * lh) * 4;
#ifdef S32_LAB_OOB
            pend_off[i] = kOob;
#endif
        }
        relbase = relbase + 128 >= kRingP ? relbase + 128 - kRingP : relbase + 128;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) store_piece(q);
}

}  // namespace

bool rn_conv_strip32_eligible(const GemmParams &p)
{
    return p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.chunk_dw == 0 && p.kreal == 0 &&
           p.tap_rows == 1 && p.residual == nullptr && p.Ho == p.H && p.Wo == p.W && !p.out_nchw &&
           (uint64_t)(p.M / (p.H * p.W) * (p.H + 1) + 1) * (uint64_t)(p.W + 2) < (1ull << 30) &&
           p.Cs == 64 && p.Cout == 64 && p.cseg == 2 && p.Ktot == 576 && p.W + 3 <= kMargin;
}

void rn_conv_strip32_launch(rn_ctx *ctx, const GemmParams &g)
{
    Strip32Params p;
    p.in = g.in, p.w = g.w, p.out = g.out, p.scale = g.scale, p.shift = g.shift, p.relu = g.relu;
    p.H = g.H, p.W = g.W, p.B = g.M / (g.H * g.W);
    p.Wp = p.W + 2, p.Hq = p.H + 1;
    rn_fast_div((unsigned)p.Wp, &p.mul_wp, &p.shr_wp);
    rn_fast_div((unsigned)p.Hq, &p.mul_hq, &p.shr_hq);
    p.U = (p.B * p.Hq + 1) * p.Wp;
    p.in_bytes = g.in_bytes, p.out_bytes = g.out_bytes;
    p.nsteps = (p.U + 127) / 128;
    conv_strip32_kernel<<<dim3(p.nsteps < ctx->cus ? p.nsteps : ctx->cus), dim3(256), 0, ctx->stream>>>(p);
}
