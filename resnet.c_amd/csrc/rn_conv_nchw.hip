// Convolution straight on the reference's layout: NCHW in, OIHW weight, NCHW out, fp32.
// Replaces conv2dForwardKernel (cuda/ops.cu:14-48) behind Conv2d::forward (cuda/nn.cu:3-16) on the
// literal drop-in route (rn_conv2d_forward with the context in RN_LAYOUT_NCHW, not deferred): every
// 1x1 convolution -- 36 of ResNet-50's 53, the stride-2 projection shortcuts included -- and the
// k x k ones where it pays (rn_ctx_set_nchw_taps; below).
//
// The engine is NHWC inside, so that route used to transpose every convolution's input
// (rn_nchw_to_nhwc into scratch) before the implicit-GEMM kernel, whose epilogue then wrote NCHW.
// NCHW *is* a natural MFMA layout once the operands are swapped:
//
//   D[cout][pixel] = sum_(kh,kw,c) W[cout][kh][kw][c] * X[c][pixel moved by tap (kh, kw)]
//
//  * A operand = the weight, rows = output channels, K contiguous: for a 1x1 convolution exactly the
//    reference's OIHW buffer [Cout][Cin] -- no packing, no packed-weight cache; for k x k the packed
//    panel [Cout][kh][kw][Cin] the NHWC contraction uses (taps outermost: the same K order);
//  * B operand = the image, one row per (tap, input channel), pixels contiguous: a channel plane of
//    the NCHW tensor, read at (s oh + kh - pad, s ow + kw - pad).  A lane of
//    v_mfma_f32_32x32x2_f32's B operand holds (k = lane half, n = lane & 31): 32 consecutive pixels
//    of one channel, a 128-byte run inside an image row;
//  * D: a lane ends with 16 output channels of one pixel, and for a fixed accumulator element the
//    32 lanes of a half-wave are 32 consecutive pixels of one output channel: coalesced NCHW stores
//    straight from the accumulators, no LDS staging of the output.
//
// "Pixels" are the flat index q = b * Ho*Wo + p over the batch, so a tile may straddle images
// (14x14 planes are 196 pixels); every access splits q into (b, p) itself.  1x1 / stride 1 / no
// padding on planes of a multiple of 4 pixels is staged by 16-byte loads (a quad of pixels then
// never straddles an image); everything else -- 7x7 planes, strides, taps, padding -- by one dword
// load per pixel and channel (GATHER).  in_channels % 32 != 0 (the stem) keeps the transposing
// route; so do the k x k layers on small planes by default: measured at B = 256
// (tools/nchw_bench.py), the gathering K loop is 15-25 % slower per tile than the NHWC one and the
// transpose it saves costs 70 us on a 56x56 tensor but 10-25 us on the 14x14 / 7x7 ones.
//
// Same bits as conv_gemm_kernel<float>: every output is the same fma chain -- K tiles of 32
// channels in order, taps outermost, inside a tile the k pairs (8s + j, 8s + 4 + j), s = 0..3,
// j = 0..3, one pair per MFMA; a tap in the padding multiplies by the same zero -- and layers with
// 32 or more K tiles (K >= 1024) add their eight chunk sums ((c0 + c1) + c2) + ... exactly as
// rn_conv.hip does (GemmParams::chunk_L).  Swapping the MFMA's A and B swaps the factors of each
// product, nothing else.
//
// Block: 256 threads = WM x WN waves, wave tile 64 output channels x 64 pixels (2 x 2 MFMA tiles),
// block tile 64 WM x 64 WN.  W and X tiles of one K tile (32 channels) go through LDS, register-
// staged double buffering, one barrier per K tile (the structure of conv_gemm_kernel): W as
// [rows][32] with the 16-byte chunks XOR-swizzled by (row >> 1) & 7 (one ds_read_b128 = this
// lane's four k of a k-step group), X as [32][pixels] read one dword per MFMA operand
// (consecutive lanes = consecutive banks).
#include "rn_conv_params.h"

using namespace rn_gemm;

namespace {

struct NchwParams {
    const float *in, *w;
    float *out;
    int Cin, Cout, HW;   // HW: pixels of an OUTPUT plane
    int HWin, Hin, Win, Wo, stride;  // GATHER: input plane geometry
    int KS, pad, ctiles;        // GATHER: kernel size, padding, K tiles per tap (Cin / 32)
    int Kw;                     // floats of a weight row: KS * KS * Cin
    unsigned mul_wo, shr_wo;
    unsigned Q;          // B * HW output pixels
    int nk;              // K tiles of 32 channels
    int chunk_L;         // K tiles per chunk sum; >= nk: one plain sum
    unsigned tiles_m, total_tiles;
    unsigned mul_hw, shr_hw;  // q / HW as a multiply-high
    int in_bytes, w_bytes, out_bytes;
};

// GATHER: every staged pixel is fetched by a dword load of its own -- planes whose size is not a
// multiple of 4 (7x7: a quad of consecutive pixels may straddle two images, and is not 16-byte
// aligned), stride-2 convolutions (the projection shortcuts: input pixel (2 oh, 2 ow)) and k x k
// convolutions (tap (kh, kw) of a K tile: input pixel (s oh + kh - pad, s ow + kw - pad), zero where
// that lies in the padding).  A thread keeps ONE pixel column of the tile and walks the channel rows:
// the 64 lanes of a load are 64 consecutive output pixels -- consecutive input addresses inside an
// image row -- and their LDS stores 64 consecutive banks.
template <int WM, int WN, bool GATHER>
__global__ __launch_bounds__(256, 2) void conv_nchw_kernel(const NchwParams p)
{
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int WT = BM * 32, XT = 32 * BN;  // floats of one staged tile
    constexpr int WP = BM / 32, XP = BN / 32;  // 16-byte pieces per thread and K tile
    __shared__ __attribute__((aligned(16))) float lds[2 * (WT + XT)];
    constexpr int kOob = (int)0x80000000;

    // XCD-aware tile order: each XCD takes a contiguous range of tiles, the output-channel tiles of
    // one pixel tile adjacent (they re-read the same X panel from that XCD's L2)
    unsigned m0, q0;
    {
        const unsigned v = blockIdx.x, tt = p.total_tiles;
        const unsigned q = tt >> 3, r = tt & 7, xcd = v & 7;
        const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        m0 = (logical % p.tiles_m) * BM;
        q0 = (logical / p.tiles_m) * BN;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsrc_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);

    // staging: W piece j = row (t >> 3) + 32 j, chunk t & 7; X piece j = channel (t / (BN/4)) + (1024/BN) j
    // ... of the K tile, pixels 4 (t % (BN/4)) ..
    const int wc = t & 7, wr0 = t >> 3;
    int w_off[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const unsigned row = m0 + (unsigned)(wr0 + 32 * j);
        w_off[j] = row < (unsigned)p.Cout ? (int)((row * (unsigned)p.Kw + (unsigned)(4 * wc)) * 4u) : kOob;
    }
    constexpr int XQ = BN / 4;         // quads per channel row
    constexpr int XR = 256 / XQ;       // channel rows per pass
    constexpr int GR = 256 / BN;       // GATHER: channel rows per pass (one pixel per thread)
    constexpr int GP = 32 / GR;        // GATHER: passes = dwords per thread and K tile (= 4 XP)
    const int xq = GATHER ? t % BN : t % XQ, xr0 = GATHER ? t / BN : t / XQ;
    int x_off = kOob;                  // quads: byte offset of this thread's quad in channel xr0 of K tile 0
    int g_base = 0, g_ih0 = 0, g_iw0 = 0;  // GATHER: byte offset of (image, channel xr0), input position of tap (0, 0)
    bool g_ok = false;
    {
        const unsigned q = q0 + (GATHER ? (unsigned)xq : 4u * (unsigned)xq);
        const unsigned b = p.HW == 1 ? q : __umulhi(q, p.mul_hw) >> p.shr_hw;  // (rn_fast_div leaves d = 1 to the kernel)
        const unsigned pp = q - b * (unsigned)p.HW;
        if constexpr (GATHER) {  // output pixel -> input pixel of tap (0, 0)
            const unsigned oh = p.Wo == 1 ? pp : __umulhi(pp, p.mul_wo) >> p.shr_wo, ow = pp - oh * (unsigned)p.Wo;
            g_ih0 = (int)oh * p.stride - p.pad;
            g_iw0 = (int)ow * p.stride - p.pad;
            g_ok = q < p.Q;
            g_base = g_ok ? (int)((b * (unsigned)p.Cin + (unsigned)xr0) * (unsigned)p.HWin * 4u) : 0;
        } else {
            x_off = q < p.Q ? (int)(((b * (unsigned)p.Cin + (unsigned)xr0) * (unsigned)p.HWin + pp) * 4u) : kOob;
        }
    }
    const int x_step = (GATHER ? GR : XR) * p.HWin * 4;  // bytes between this thread's pieces (that many channels further)
    int g_ct = 0, g_kh = 0, g_kw = 0;  // GATHER: channel tile and tap of the next K tile to load (uniform)

    u32x4 rw[WP], rx[XP];  // (GATHER: rx is GP = 4 XP dwords, one per channel row of this thread's pixel)
    auto load_tile = [&](int kt) {  // called for kt = 0, 1, 2, ... in order
#pragma unroll
        for (int j = 0; j < WP; ++j)
            rw[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[j], kt * 128, 0));
        if constexpr (GATHER) {
            const int ih = g_ih0 + g_kh, iw = g_iw0 + g_kw;
            const bool ok = g_ok && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
            // (kOob + j x_step stays above 2^31 as an unsigned offset: out of range of the descriptor)
            const int o = ok ? g_base + (ih * p.Win + iw) * 4 : kOob;
            const int so = g_ct * 32 * p.HWin * 4;
#pragma unroll
            for (int j = 0; j < GP; ++j)
                rx[j >> 2][j & 3] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, o + j * x_step, so, 0);
            if (++g_ct == p.ctiles) {
                g_ct = 0;
                if (++g_kw == p.KS) g_kw = 0, ++g_kh;
            }
        } else {
#pragma unroll
            for (int j = 0; j < XP; ++j)
                rx[j] = __builtin_bit_cast(
                    u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, x_off == kOob ? kOob : x_off + j * x_step,
                                                                 kt * 32 * p.HWin * 4, 0));
        }
    };
    auto store_tile = [&](int buf) {
        float *wl = lds + buf * (WT + XT), *xl = wl + WT;
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int row = wr0 + 32 * j;
            *reinterpret_cast<u32x4 *>(wl + row * 32 + ((wc ^ ((row >> 1) & 7)) << 2)) = rw[j];
        }
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < GP; ++j) xl[(xr0 + GR * j) * BN + xq] = __uint_as_float(rx[j >> 2][j & 3]);
        } else {
#pragma unroll
            for (int j = 0; j < XP; ++j) *reinterpret_cast<u32x4 *>(xl + (xr0 + XR * j) * BN + 4 * xq) = rx[j];
        }
    };

    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f, tot[mi][ni][e] = 0.f;
    const int sw = (li >> 1) & 7;
    auto compute_tile = [&](int buf) {
        const float *wl = lds + buf * (WT + XT), *xl = wl + WT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 a[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                a[mi] = *reinterpret_cast<const u32x4 *>(wl + (wm * 64 + mi * 32 + li) * 32 + (((2 * s + lh) ^ sw) << 2));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = xl[(8 * s + 4 * lh + j) * BN + wn * 64 + ni * 32 + li];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[mi][j]), b[ni],
                                                                            acc[mi][ni], 0, 0, 0);
            }
        }
    };
    // chunk sums as rn_conv.hip folds them: at the first K tile of chunk c >= 1 the finished chunk
    // goes into the running total (total = c0, then total + c), the accumulators restart from zero;
    // at the end the result is total + last chunk
    bool folded = false;
    auto fold = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    tot[mi][ni][e] = folded ? tot[mi][ni][e] + acc[mi][ni][e] : acc[mi][ni][e];
                    acc[mi][ni][e] = 0.f;
                }
        folded = true;
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int fold_at = p.chunk_L;
    for (int kt = 0; kt < p.nk; ++kt) {
        const int buf = kt & 1;
        if (kt == fold_at) {  // uniform
            fold();
            fold_at += p.chunk_L;
        }
        if (kt + 1 < p.nk) load_tile(kt + 1);
        compute_tile(buf);
        if (kt + 1 < p.nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (folded) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = tot[mi][ni][e] + acc[mi][ni][e];
    }

    // accumulator element e of lane (li, lh): output channel 8 (e >> 2) + 4 lh + (e & 3) of the MFMA
    // tile, pixel li: a half-wave stores 32 consecutive pixels of one channel
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const unsigned q = q0 + (unsigned)(wn * 64 + ni * 32 + li);
        const unsigned b = p.HW == 1 ? q : __umulhi(q, p.mul_hw) >> p.shr_hw, pp = q - b * (unsigned)p.HW;
        const unsigned obase = b * (unsigned)p.Cout * (unsigned)p.HW + pp;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned co = m0 + (unsigned)(wm * 64 + mi * 32 + 8 * (e >> 2) + 4 * lh + (e & 3));
                const int off = (q < p.Q && co < (unsigned)p.Cout) ? (int)((obase + co * (unsigned)p.HW) * 4u) : kOob;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mi][ni][e]), rsrc_o, off, 0, 0);
            }
    }
}

}  // namespace

// library-internal (rn_conv.hip)
bool rn_conv_nchw_eligible(uint64_t kernel_size, uint64_t stride, uint64_t padding, uint64_t B, uint64_t Cin,
                           uint64_t Cout, uint64_t H, uint64_t W)
{
    if (kernel_size < 1 || kernel_size > 7 || padding > 7 || stride < 1 || stride > 8 || H == 0 || W == 0) return false;
    if (H + 2 * padding < kernel_size || W + 2 * padding < kernel_size) return false;
    const uint64_t Ho = (H + 2 * padding - kernel_size) / stride + 1, Wo = (W + 2 * padding - kernel_size) / stride + 1;
    return Cin % 32 == 0 && Cin >= 32 && B * Cin * H * W < (1ull << 29) && B * Cout * Ho * Wo < (1ull << 29) &&
           kernel_size * kernel_size * Cin * Cout < (1ull << 29) && B * Ho * Wo + 1024 < (1ull << 31);
}

// weight: [Cout][kh][kw][Cin] -- for kernel_size 1 the reference's OIHW buffer itself, otherwise the packed panel
int rn_conv_nchw_launch(rn_ctx *ctx, const float *inp, float *out, const float *weight, uint64_t kernel_size,
                        uint64_t stride, uint64_t padding, uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W)
{
    const uint64_t Ho = (H + 2 * padding - kernel_size) / stride + 1, Wo = (W + 2 * padding - kernel_size) / stride + 1;
    NchwParams p;
    p.in = inp;
    p.w = weight;
    p.out = out;
    p.Cin = (int)Cin;
    p.Cout = (int)Cout;
    p.HW = (int)(Ho * Wo);
    p.HWin = (int)(H * W);
    p.Hin = (int)H;
    p.Win = (int)W;
    p.Wo = (int)Wo;
    p.stride = (int)stride;
    p.KS = (int)kernel_size;
    p.pad = (int)padding;
    p.ctiles = (int)(Cin / 32);
    p.Kw = (int)(kernel_size * kernel_size * Cin);
    p.Q = (unsigned)(B * Ho * Wo);
    p.nk = p.Kw / 32;
    // the chunked K sum is a property of the layer (rn_conv.hip: fp32, 32 or more K tiles, Cout % 4 == 0)
    p.chunk_L = (p.nk >= 32 && Cout % 4 == 0) ? 2 * (int)rn_ceil_div((uint64_t)p.nk, 16) : p.nk;
    rn_fast_div((unsigned)p.HW, &p.mul_hw, &p.shr_hw);
    rn_fast_div((unsigned)p.Wo, &p.mul_wo, &p.shr_wo);
    p.in_bytes = (int)(B * Cin * H * W * 4);
    p.w_bytes = (int)((uint64_t)p.Kw * Cout * 4);
    p.out_bytes = (int)(B * Cout * Ho * Wo * 4);
    // whole quads of consecutive pixels by one 16-byte load where that is possible
    const bool gather = kernel_size != 1 || padding != 0 || stride != 1 || (H * W) % 4 != 0 ||
                        (reinterpret_cast<uintptr_t>(inp) & 15) != 0;
    // 64 output channels: one wave row, 256 pixels per block; otherwise 128 x 128
    const bool narrow = Cout <= 64;
    const unsigned BM = narrow ? 64 : 128, BN = narrow ? 256 : 128;
    p.tiles_m = (unsigned)rn_ceil_div(Cout, BM);
    const uint64_t total = (uint64_t)p.tiles_m * rn_ceil_div((uint64_t)p.Q, BN);
    if (total >= (1ull << 31)) return rn_set_error(ctx, RN_ERR_INVALID, "rn_conv2d_forward: too many tiles");
    p.total_tiles = (unsigned)total;
    const dim3 grid(p.total_tiles), block(256);
    if (narrow && gather)
        conv_nchw_kernel<1, 4, true><<<grid, block, 0, ctx->stream>>>(p);
    else if (narrow)
        conv_nchw_kernel<1, 4, false><<<grid, block, 0, ctx->stream>>>(p);
    else if (gather)
        conv_nchw_kernel<2, 2, true><<<grid, block, 0, ctx->stream>>>(p);
    else
        conv_nchw_kernel<2, 2, false><<<grid, block, 0, ctx->stream>>>(p);
    return rn_after_launch(ctx, "rn_conv2d_forward(nchw)");
}
