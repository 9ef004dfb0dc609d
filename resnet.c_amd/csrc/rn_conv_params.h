// Parameters and small types shared by the contraction kernels (rn_conv.hip: the 4-wave
// kernels for fp32 and bf16; rn_conv_wide.hip: the 8-wave LDS-DMA kernels for bf16 -- 256-wide
// tiles and the strip kernel).
#ifndef RN_CONV_PARAMS_H
#define RN_CONV_PARAMS_H

#include "rn_internal.h"

namespace rn_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct GemmParams {
    const void *in;
    const void *w;
    void *out;
    const float *scale;
    const float *shift;
    const void *residual;  // element type of the output
    int relu;
    int H, W, Cs;  // input height, width, elements per input pixel
    int Ho, Wo, Cout;
    int KH, KW;  // taps walked by the K loop (small-Cin form: KW = 1)
    int stride, pad;
    int cseg;       // 128-byte segments per tap
    int chunk_dw;   // small-Cin form: pixels per 16-byte chunk (0 otherwise)
    int c4_chunks;  // small-Cin form: chunks of a segment that carry real taps
    // small-Cin form, bf16: a K tile holds tap_rows = 2 kernel rows (chunks 0-3: row 2t, chunks
    // 4-7: row 2t+1, two 4-channel pixels per chunk), KH = ceil(k / 2) K tiles; k_rows = k, the
    // kernel rows that exist.  Everywhere else tap_rows = 1 and k_rows = KH.
    int tap_rows, k_rows;
    int M;          // B * Ho * Wo
    int Ktot;       // packed weight row length in elements
    int nk;         // K tiles
    int tiles_n;
    unsigned total_tiles;  // the grid may be smaller: blocks then walk tiles grid-stride
    // tile order (rn_conv.hip, conv_gemm_kernel): the tiles the remap covers are dealt to the 8 XCDs as
    // contiguous ranges of an ORDER; xg = 0: the order is the logical one (M panel major, the N tiles of a
    // panel adjacent: an XCD reads its A panels once and streams the whole weight panel);  xg > 0: the order
    // is (N group of xg tiles, M panel, N tile in the group) over the first xrows whole rows of M panels, so
    // that an XCD keeps to one group of N tiles -- a slice of the weight panel small enough to stay in its
    // 4 MB L2 while M advances -- and the A panels are read by tiles_n / xg XCDs instead
    int xg;
    unsigned xrows;
    int HoWo;
    unsigned mul_hw, shr_hw, mul_w, shr_w;  // n / d == umulhi(n, mul) >> shr for n < 2^31
    unsigned mul_cs, shr_cs, mul_kw, shr_kw;  // K tile -> (tap, segment), tap -> (kh, kw)
    int in_bytes, w_bytes, out_bytes;
    // fused pair (DUAL kernels only): K tiles nk1.. come from a second NHWC tensor through a
    // 1x1 / padding-0 convolution of the same output geometry (the downsample branch)
    const void *in2;
    int in2_bytes, H2, W2, Cs2, stride2, nk1;
    // split K (latency mode): work item v = split * total_tiles + tile; split s sums K tiles
    // [s*kchunk, min((s+1)*kchunk, nk)) and writes its raw fp32 partial tile to
    // out + s*split_stride bytes; a second kernel adds the partials in order and finishes
    int ksplit, kchunk;
    unsigned total_work;  // total_tiles * ksplit
    unsigned grid_items;  // host only: blocks of a non-persistent launch
    long long split_stride;
    // chunked K sum (CHUNK kernels only): every output of the layer is ((c0 + c1) + c2) + ...
    // with c_i the sum over K tiles [i*chunk_L, (i+1)*chunk_L).  Tiles below full_tiles fold the
    // chunks in registers; the last tail_tiles logical tiles are cut into (tile, chunk) pieces
    // that write their raw chunk sum to ws + chunk*ws_stride (same [M][Cout] addressing as the
    // output) -- a second kernel adds them in the same order and runs the epilogue.
    int chunk_L;
    unsigned full_tiles, tail_tiles;
    void *ws;
    long long ws_stride;
    // exact-K small-Cin form (XK kernels only): K index q = (kh*KW + kw)*Cin + c over a
    // physically padded image; element q of an A row sits q + (q / kc) * kskip floats after
    // the row's first element (kc = KW*Cin, kskip = (W - KW)*Cin), zero weight past kreal
    int kc, kskip, kreal;
    unsigned mul_kc, shr_kc;
    // fp32 output written as NCHW instead of NHWC (the literal drop-in route, rn_conv2d_forward
    // on NCHW tensors): saves the transpose launch behind the contraction
    int out_nchw;
    // diagnostic only (tools/conv_stamps.py): 16 stamp slots per block, or null
    unsigned long long *stamps;
};

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

// K tile = one 128-byte row segment: 32 fp32 or 64 bf16 elements; 16-byte chunk = 4 / 8
template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int CH = 4;
};
template <>
struct Elem<bf16_t> {
    static constexpr int CH = 8;
};

}  // namespace rn_gemm

// n / d == umulhi(n, mul) >> shr for 0 <= n < 2^31, d >= 1 (round-up magic number)
inline void rn_fast_div(unsigned d, unsigned *mul, unsigned *shr)
{
    if (d <= 1) {  // the kernels special-case a divisor of 1 (fc, 1x1 outputs)
        *mul = 0;
        *shr = 0;
        return;
    }
    unsigned lg = 0;
    while ((1u << lg) < d) ++lg;
    const unsigned p = 31 + lg;
    const uint64_t m = ((1ull << p) + d - 1) / d;
    *mul = (unsigned)m;
    *shr = p - 32;
}

// bf16 contraction on 256-wide block tiles (rn_conv_wide.hip).  which: 0 = 256x256, 1 = 256x128,
// 2 = 128x256, 3 = 256x64, 4 = 224x256, 5 = 128x128 (two blocks per CU).  Caller has checked rn_conv_wide_eligible().
int rn_conv_wide_count(void);
bool rn_conv_wide_eligible(const rn_gemm::GemmParams &p, int which);
void rn_conv_wide_tile(int which, int *bm, int *bn);
void rn_conv_wide_launch(rn_ctx *ctx, rn_gemm::GemmParams &p, int which, bool dual);
// bf16 3x3 / stride 1 / 64 -> 64 channels with the weights in registers and the input in a rolling
// LDS ring (rn_conv_wide.hip, conv_strip_kernel): the tuner's candidate after the wide tiles.
bool rn_conv_strip_eligible(const rn_gemm::GemmParams &p);
void rn_conv_strip_launch(rn_ctx *ctx, const rn_gemm::GemmParams &p);

// fp32 convolution on NCHW tensors without the input transpose (rn_conv_nchw.hip): the literal drop-in route's
// rn_conv2d_forward.  weight = [Cout][kh][kw][Cin]: the OIHW buffer as it is for kernel_size 1, the packed panel
// otherwise.  Same bits as the NHWC contraction.
bool rn_conv_nchw_eligible(uint64_t kernel_size, uint64_t stride, uint64_t padding, uint64_t B, uint64_t Cin,
                           uint64_t Cout, uint64_t H, uint64_t W);
int rn_conv_nchw_launch(rn_ctx *ctx, const float *inp, float *out, const float *weight, uint64_t kernel_size,
                        uint64_t stride, uint64_t padding, uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W);

#endif
