// Stem in one launch: conv 7x7 stride 2 (3 -> 64 channels) + folded batch-norm + ReLU + max-pool
// 3x3 stride 2 padding 1.  Reference: the first four ops of resnet152Forward
// (cuda/inference/main.cu:179-192 -> conv2dForwardKernel ops.cu:14-48, batchNorm2dForwardKernel
// ops.cu:139-151, reluForwardKernel ops.cu:130-137, maxPool2dKernel ops.cu:50-78).
//
// As an implicit GEMM the stem is a bad fit: K = 147, N = 64, and every input pixel is gathered
// ~12 times from L2 in 64-byte pieces (the contraction kernel spends 0.72 ms fp32 / 0.30 ms bf16
// on 0.38 / 0.08 ms of matrix work), after which its 112x112x64 output -- the largest tensor of
// the network -- goes to HBM only to be read back by the pool.  Here:
//
//  * a block owns two pooled rows of one image: stem rows 2*ph0-1 .. 2*ph0+3 (five rows, one of
//    them a halo recomputed by the next block: +25 % matrix work) x all columns x 64 channels;
//  * the input patch those rows need -- 15 rows of the physically padded NHWC image, one
//    contiguous block of memory -- is copied to LDS once; the MFMA A operands are read
//    straight out of it: for fixed kernel row, the k index runs over (kw, c) = consecutive
//    floats / bf16 of the patch row, so a fragment is one ds_read_b32 (fp32, 32x32x2 MFMA) or
//    one ds_read_b128 (bf16, 32x32x16 MFMA) at base(position) + constant(k-step).  No im2col;
//  * the weights live in registers for the lifetime of the (persistent) block: 77 floats or
//    14 x 8 bf16 per lane, one 32-channel half per wave;
//  * K order: kernel row major; fp32 (kw, c) with one zero-weight slot per row (22 per row, 154);
//    bf16 (kw pair, kw parity, c of 4) with kw = 7 and c = 3 zero-weight (32 per row, 224);
//  * epilogue in registers: y = max(acc * scale + shift, 0) (bf16: rounded to bf16), then the
//    max-pool: the vertical part in registers (an M tile is 4 stem rows x 8 columns, so a lane
//    holds four rows of four adjacent columns of its channel), the rest as LDS integer maxima:
//    y >= 0, so its bit pattern orders like the value and ds_max_u32 into a zeroed [2][PW][64]
//    buffer gives exactly the reference's maximum over the window's real pixels (every window
//    has one; padded taps are skipped, ops.cu:65-67);
//  * the pooled rows leave as whole 128..256-byte pixels.  The stem tensor is never written.
//
// Bound: matrix pipe in fp32 (0.52 ms at peak for B=256 with the halo), HBM in bf16 (0.3 GB).
// Measured at B=256 (timing-only builds with parts switched off): fp32 0.68 ms = 0.57 matrix
// phase + 0.09 epilogue arithmetic (of which 0.01 the LDS maxima) + 0.02 the rest; bf16 0.22 ms =
// 0.11 + 0.09 + 0.02: the register epilogue (80 values per lane: affine, ReLU, rounding, maxima)
// is what a second resident block would hide, and 242 VGPRs allow only one.
#include <type_traits>

#include "rn_conv_params.h"

using namespace rn_gemm;

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kCout = 64, kK = 7;
constexpr int kPatchRows = 15;  // input rows behind five stem rows: 2*4 + 7
constexpr int kMaxTiles = 5;    // 32-position M tiles per wave (4 wave rows): Wo <= 128

struct StemParams {
    const void *in;      // [B][Hp][Wp][CS] physically padded NHWC image (CS = 3 fp32, 4 bf16)
    const void *w;       // packed panel, see rn_stem_pool_pack_weight_dt
    const float *scale;  // folded batch-norm, may be null
    const float *shift;
    void *out;           // [B][PH][PW][64]
    int B, Hp, Wp, Ho, Wo, PH, PW;
    int Cin;             // NCHW input only: channels of the image (1..3)
    int relu;
    unsigned items;      // B * ceil(PH / 2)
    int pairs;           // ceil(PH / 2)
};

template <typename T>
struct Cfg;
template <>
struct Cfg<float> {
    static constexpr int CS = 3;        // channels per pixel in memory
    static constexpr int KROW = 22;     // k per kernel row (21 + one zero-weight slot)
    static constexpr int KPS = 2;       // k per MFMA step
    static constexpr int STEPS_ROW = 11;
};
template <>
struct Cfg<bf16_t> {
    static constexpr int CS = 4;
    static constexpr int KROW = 32;
    static constexpr int KPS = 16;
    static constexpr int STEPS_ROW = 2;
};

// NCHW_IN: p.in is the reference's fp32 NCHW image [B][Cin][H][W] itself (H = Hp - 6, W = Wp - 6):
// the patch is assembled in LDS from its channel rows -- zero border, channel interleave and the
// conversion to T included -- and the separate layout kernel and its padded copy of the image
// disappear.
template <typename T, bool NCHW_IN>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const StemParams p)
{
    using C = Cfg<T>;
    constexpr int ES = (int)sizeof(T);
    constexpr int PIXB = C::CS * ES;            // bytes per pixel: 12 / 8
    constexpr int STEPS = kK * C::STEPS_ROW;    // MFMA k-steps: 77 / 14
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int rowb = p.Wp * PIXB;               // bytes per patch row
    // LDS: two input patches and two pooled-row buffers (item i is multiplied out of one patch
    // while the next item's lands in the other and the previous item's pooled rows leave)
    const int patch_bytes = 48 + ((kPatchRows * rowb + 15) & ~15);  // slack: shifted base in front, piece overhang behind
    const int pooled_n = 2 * p.PW * kCout;                           // floats of one [2][PW][64] buffer
    float *const pooled0 = reinterpret_cast<float *>(lds + 2 * patch_bytes);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n = wn * 32 + li;  // this lane's output channel
    const int n2d = p.Wo >> 3, ntiles = n2d + ((p.Wo + 31) >> 5);  // 4x8 tiles of rows 0..3, 1x32 tiles of row 4

    // weights of this lane's channel: B operand of every k-step, loaded once
    typename std::conditional<sizeof(T) == 4, float, i32x4>::type bw[STEPS];
    {
        const char *wrow = static_cast<const char *>(p.w) + (size_t)n * (kK * C::KROW) * ES;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if constexpr (sizeof(T) == 4)
                bw[s] = *reinterpret_cast<const float *>(wrow + (s * 2 + lh) * 4);
            else
                bw[s] = *reinterpret_cast<const i32x4 *>(wrow + (s * 16 + lh * 8) * 2);
        }
    }
    const float sc = p.scale ? p.scale[n] : 1.f;
    const float sh = p.shift ? p.shift[n] : 0.f;

    for (int i = t; i < 2 * pooled_n; i += 512) pooled0[i] = 0.f;
    if constexpr (NCHW_IN) {  // the patches' border pixels and pad channel are zero and stay zero
        for (int i = t * 16; i < 2 * patch_bytes; i += 512 * 16)
            *reinterpret_cast<i32x4 *>(lds + i) = i32x4{0, 0, 0, 0};
        __syncthreads();
    }

    // The input patch of an item: rows [row0, row0 + 15) of the padded image clipped to the image --
    // one contiguous block of memory -- fetched as 16-byte pieces from the 16-byte boundary below
    // its first byte (fp32 rows are only 8-byte multiples; the LDS image is shifted likewise, it
    // starts 16 or 24 bytes into the array).  All pieces of a thread are independent loads, issued for
    // item i+1 before the contraction of item i and written to LDS after it: the fetch hides
    // behind the matrix work instead of standing in front of it.
    // 512 threads x 16 B x 6 = 48 KB >= 15 rows of 266 fp32 pixels; NCHW: 6 x 8 waves >= 45 channel rows
    constexpr int kPieces = 6;
    i32x4 stage[kPieces];
    auto patch_src = [&](unsigned item, const char *&src, int &dst_off, int &nbytes, int &base) {
        const int b = (int)(item / (unsigned)p.pairs), pj = (int)(item % (unsigned)p.pairs);
        const int row0 = 2 * (4 * pj - 1);
        const int lo = max(row0, 0), hi = min(row0 + kPatchRows, p.Hp);
        const char *first = static_cast<const char *>(p.in) + ((size_t)b * p.Hp + lo) * rowb;
        const int mis = (int)(reinterpret_cast<uintptr_t>(first) & 15);
        src = first - mis;
        // LDS offset of patch row 0: 16 or 24, whichever puts the first fetched piece on a
        // 16-byte boundary
        base = 16 + ((mis - (lo - row0) * rowb) & 15);
        dst_off = base + (lo - row0) * rowb - mis;
        nbytes = (hi - lo) * rowb + mis;
    };
    // NCHW_IN: a wave fetches one channel row of the patch per piece -- lane q its pixels 4q..4q+3
    // (one 16-byte load; lanes past W/4 idle) -- piece k of wave w the row-channel rc = w + 8k =
    // (patch row r) * Cin + c, i.e. channel c of image row row0 + r - 3.  Stored as four elements
    // of T at pixel stride; rows outside the image are stored as zeros (the buffer held another
    // item's rows).
    const int W4 = (p.Wp - 6) >> 2;
    auto unit_of = [&](int k, int &r, int &c, int &q) -> bool {
        const int rc = (t >> 6) + 8 * k;
        q = t & 63;
        r = p.Cin == 3 ? (rc * 43) >> 7 : p.Cin == 2 ? rc >> 1 : rc;  // rc / Cin for rc < 48
        c = rc - r * p.Cin;
        return r < kPatchRows && q < W4;
    };
    auto patch_fetch = [&](unsigned item) {
        if constexpr (NCHW_IN) {
            const int b = (int)(item / (unsigned)p.pairs), pj = (int)(item % (unsigned)p.pairs);
            const int row0 = 2 * (4 * pj - 1), H = p.Hp - 6, W = p.Wp - 6;
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                int r, c, q;
                const bool ok = unit_of(k, r, c, q);
                const int ih = row0 + r - 3;
                const bool real = ok && ih >= 0 && ih < H;
                const float *src = static_cast<const float *>(p.in) +
                                   (((size_t)b * p.Cin + (real ? c : 0)) * H + (real ? ih : 0)) * W + (real ? 4 * q : 0);
                const i32x4 v = *reinterpret_cast<const i32x4 *>(src);
                stage[k] = real ? v : i32x4{0, 0, 0, 0};
            }
        } else {
            const char *src;
            int dst_off, nbytes, base;
            patch_src(item, src, dst_off, nbytes, base);
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                const int o = (k * 512 + t) * 16;
                stage[k] = *reinterpret_cast<const i32x4 *>(src + (o < nbytes ? o : 0));
            }
        }
    };
    auto patch_store = [&](unsigned item, int buf) -> int {
        if constexpr (NCHW_IN) {
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                int r, c, q;
                if (!unit_of(k, r, c, q)) continue;
                T *dst = reinterpret_cast<T *>(lds + buf * patch_bytes + 16 + r * rowb) + (3 + 4 * q) * C::CS + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[i * C::CS] = (T)__int_as_float(stage[k][i]);
            }
            return buf * patch_bytes + 16;
        } else {
            const char *src;
            int dst_off, nbytes, base;
            patch_src(item, src, dst_off, nbytes, base);
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                const int o = (k * 512 + t) * 16;
                if (o < nbytes) *reinterpret_cast<i32x4 *>(lds + buf * patch_bytes + dst_off + o) = stage[k];
            }
            return buf * patch_bytes + base;
        }
    };
    // pooled rows of a finished item -> global, then cleared for the item after next
    auto pooled_out = [&](unsigned item, float *pooled) {
        const int b = (int)(item / (unsigned)p.pairs), ph0 = 2 * (int)(item % (unsigned)p.pairs);
        const int rows = min(2, p.PH - ph0);
        const int nout = rows * p.PW * kCout;
        char *obase = static_cast<char *>(p.out) + ((size_t)b * p.PH + ph0) * p.PW * kCout * ES;
        for (int i = t * 4; i < nout; i += 512 * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(pooled + i);
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(obase + (size_t)i * 4) = v;
            } else {
                typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 o;
                o[0] = (bf16_t)v.x, o[1] = (bf16_t)v.y, o[2] = (bf16_t)v.z, o[3] = (bf16_t)v.w;
                *reinterpret_cast<bf16x4 *>(obase + (size_t)i * 2) = o;
            }
            *reinterpret_cast<float4 *>(pooled + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    // Software pipeline over the block's items, ONE barrier per item: while item i is multiplied
    // out of patch[i&1], the patch of item i+1 is written to the other buffer (fetched to
    // registers during item i-1), the fetch of item i+2 is issued, and the pooled rows of item
    // i-1 leave from pooled[(i-1)&1].  All of that sits in the middle of item i's k loop, between
    // MFMAs; only the epilogue of an item is not covered by matrix work.
    const unsigned step = gridDim.x;
    if (blockIdx.x >= p.items) return;
    patch_fetch(blockIdx.x);
    int patch_off = patch_store(blockIdx.x, 0);
    if (blockIdx.x + step < p.items) patch_fetch(blockIdx.x + step);
    __syncthreads();
    int cur = 0;
    for (unsigned item = blockIdx.x; item < p.items; item += step, cur ^= 1) {
        const int pj = (int)(item % (unsigned)p.pairs);
        const int ph0 = 2 * pj;
        const int oh_first = 2 * ph0 - 1;     // stem row of r = 0 (-1 for the first pair: no such row)
        const char *const patch = lds + patch_off;  // patch row 0 of this item
        float *const pooled = pooled0 + cur * pooled_n;
        int next_off = 0;

        // Per-item copies of the lane / wave coordinates that the compiler cannot see through:
        // everything below depends only on them and on the kernel arguments, and hoisted out of
        // the item loop (385 fragment addresses, the epilogue's index arithmetic) it costs more
        // than a thousand spilled registers.
        int li_ = li, lh_ = lh, wm_ = wm;
        asm volatile("" : "+v"(li_), "+v"(lh_), "+s"(wm_));

        // ---- contraction: wave (wm, wn) owns M tiles wm, wm+4, ... and channels 32*wn ..
        // M tiles: n2d tiles of 4 stem rows (r = 0..3) x 8 columns -- lane li is (row li>>3, column
        // li&7) -- then n1d tiles of 32 columns of stem row r = 4.  In the accumulator of a 4x8
        // tile a lane then holds, for its channel, rows 0..3 of four adjacent columns (element e:
        // row e>>2, column 4*lh + (e&3)): the vertical part of the pool happens in registers.
        f32x16 acc[kMaxTiles];
        int abase[kMaxTiles];
#pragma unroll
        for (int j = 0; j < kMaxTiles; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            const int tile = wm_ + 4 * j;
            const bool two_d = tile < n2d;
            const int r = two_d ? li_ >> 3 : 4;
            const int ox = two_d ? 8 * tile + (li_ & 7) : 32 * (tile - n2d) + li_;
            // patch row 2r + kh, pixel 2*ox + kw; the lane half takes the upper half of a k-step
            abase[j] = ox < p.Wo ? 2 * r * rowb + 2 * ox * PIXB + lh_ * (C::KPS / 2) * ES : 0;
        }
        // NT = tiles this wave multiplies (a compile-time count: a per-tile "does it exist" test
        // inside the k loop puts a branch around every MFMA and serialises read -> wait -> MFMA)
        auto contract = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;
#pragma unroll
            for (int kh = 0; kh < kK; ++kh) {
                if (kh == 3) {  // the other buffers' turn (see above); their last users are a barrier behind
                    if (item + step < p.items) {
                        next_off = patch_store(item + step, cur ^ 1);
                        if (item + 2 * step < p.items) patch_fetch(item + 2 * step);
                    }
                    if (item != blockIdx.x) pooled_out(item - step, pooled0 + (cur ^ 1) * pooled_n);
                }
                int arow[NT];  // this kernel row's fragment base; the k-steps are immediates
#pragma unroll
                for (int j = 0; j < NT; ++j) arow[j] = abase[j] + kh * rowb;
#pragma unroll
                for (int q = 0; q < C::STEPS_ROW; ++q) {
                    const int s = kh * C::STEPS_ROW + q;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (sizeof(T) == 4) {
                            const float a = *reinterpret_cast<const float *>(patch + arow[j] + q * C::KPS * ES);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[s], acc[j], 0, 0, 0);
                        } else {
                            const i32x4 a = *reinterpret_cast<const i32x4 *>(patch + arow[j] + q * C::KPS * ES);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bw[s]), acc[j], 0, 0, 0);
                        }
                    }
                }
            }
        };
        // tiles wm, wm+4, ... < ntiles: five for the widest images' first wave rows, else four or
        // fewer (a tile that does not exist is multiplied on patch garbage and never looked at)
        if (wm_ + 4 * (kMaxTiles - 1) < ntiles)
            contract(std::integral_constant<int, kMaxTiles>{});
        else
            contract(std::integral_constant<int, kMaxTiles - 1>{});

        // ---- epilogue: affine, ReLU, max into the pooled rows.  y >= 0 everywhere, so a stem row
        // that does not exist (above the image for the first pair, below it for the last) counts
        // as 0, the value the pooled rows start from.  Four adjacent columns 4c .. 4c+3 feed the
        // windows 2c (4c, 4c+1), 2c+1 (4c+1 .. 4c+3) and 2c+2 (4c+3): three LDS maxima.
        auto finish = [&](float a) -> unsigned {
            float v = fmaf(a, sc, sh);
            if (p.relu) v = v > 0.f ? v : 0.f;  // never -0.0: the bit pattern must order like the value
            if constexpr (sizeof(T) == 2) v = (float)(bf16_t)v;
            return __float_as_uint(v);
        };
        auto put = [&](int pl, int pw, const unsigned (&v)[4]) {
            if (ph0 + pl >= p.PH) return;
            unsigned *q0 = reinterpret_cast<unsigned *>(pooled) + ((pl * p.PW + pw) * kCout + n);
            atomicMax(q0, max(v[0], v[1]));
            atomicMax(q0 + kCout, max(max(v[1], v[2]), v[3]));
            if (pw + 2 < p.PW) atomicMax(q0 + 2 * kCout, v[3]);
        };
#pragma unroll
        for (int j = 0; j < kMaxTiles; ++j) {
            const int tile = wm_ + 4 * j;
            if (tile >= ntiles) continue;
            if (tile < n2d) {
                const int pw = 4 * tile + 2 * lh_;  // (8*tile + 4*lh) / 2
                unsigned y[4][4];
#pragma unroll
                for (int dr = 0; dr < 4; ++dr) {
                    const int oh = oh_first + dr;
                    const bool real = oh >= 0 && oh < p.Ho;  // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[dr][i] = real ? finish(acc[j][4 * dr + i]) : 0u;
                }
                unsigned v0[4], v1[4];  // windows of pooled row 0: stem rows 0..2; of row 1: 2..4
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v0[i] = max(max(y[0][i], y[1][i]), y[2][i]);
                    v1[i] = max(y[2][i], y[3][i]);
                }
                put(0, pw, v0);
                put(1, pw, v1);
            } else {
                const int oh = oh_first + 4;
                if (oh >= p.Ho) continue;  // wave-uniform
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ox = 32 * (tile - n2d) + 8 * g + 4 * lh_;
                    if (ox >= p.Wo) continue;
                    unsigned v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = finish(acc[j][4 * g + i]);
                    put(1, ox >> 1, v);
                }
            }
        }
        __syncthreads();
        patch_off = next_off;
    }
    // the last item's pooled rows (cur was flipped once more by the loop)
    {
        const unsigned n_mine = (p.items - 1 - blockIdx.x) / step;  // index of this block's last item
        pooled_out(blockIdx.x + n_mine * step, pooled0 + (cur ^ 1) * pooled_n);
    }
}

// OIHW fp32 [64][3][7][7] -> the panel the kernel keeps in registers
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float *__restrict__ w, T *__restrict__ packed,
                                                        int Cin)
{
    using C = Cfg<T>;
    const int total = kCout * kK * C::KROW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tpos = i % C::KROW, kh = (i / C::KROW) % kK, oc = i / (C::KROW * kK);
        int kw, c;
        if (sizeof(T) == 4) {
            kw = tpos / 3;
            c = tpos % 3;
            if (tpos >= 21) kw = kK;  // the pad slot
        } else {
            kw = 2 * (tpos >> 3) + ((tpos >> 2) & 1);
            c = tpos & 3;
        }
        const float v = (kw < kK && c < Cin) ? w[((oc * Cin + c) * kK + kh) * kK + kw] : 0.f;
        packed[i] = (T)v;
    }
}

}  // namespace

extern "C" {

uint64_t rn_stem_pool_packed_weight_numel(int dtype)
{
    return (uint64_t)kCout * kK * (dtype == RN_DTYPE_BF16 ? Cfg<bf16_t>::KROW : Cfg<float>::KROW);
}

int rn_stem_pool_pack_weight_dt(rn_ctx *ctx, int dtype, const float *weight_oihw, void *packed,
                                uint64_t in_channels)
{
    RN_ENTER(ctx);
    RN_REQUIRE(ctx, weight_oihw && packed, "null tensor");
    RN_REQUIRE(ctx, in_channels >= 1 && in_channels <= 3, "the fused stem takes 1..3 input channels");
    if (dtype == RN_DTYPE_BF16)
        stem_pack_kernel<bf16_t><<<16, 256, 0, ctx->stream>>>(weight_oihw, (bf16_t *)packed, (int)in_channels);
    else if (dtype == RN_DTYPE_F32)
        stem_pack_kernel<float><<<16, 256, 0, ctx->stream>>>(weight_oihw, (float *)packed, (int)in_channels);
    else
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_stem_pool_pack_weight_dt: unknown dtype");
    return rn_after_launch(ctx, "rn_stem_pool_pack_weight_dt");
}

// shared by the two entry points: nchw = the image is the reference's fp32 NCHW tensor, Cin channels
static int stem_pool_launch(rn_ctx *ctx, int dtype, const void *inp, void *out, const void *packed_weight,
                            const float *scale, const float *shift, int relu, uint64_t B, uint64_t Hp,
                            uint64_t Wp, bool nchw, uint64_t Cin, const char *what)
{
    if (B == 0) return RN_OK;
    RN_REQUIRE(ctx, dtype == RN_DTYPE_F32 || dtype == RN_DTYPE_BF16, "unknown dtype");
    RN_REQUIRE(ctx, inp && out && packed_weight && inp != out, "null or aliased tensor");
    // the pool is an unsigned-integer maximum of float bit patterns into rows that start at 0:
    // only right for values >= 0, i.e. behind the ReLU (a negative value would win as an integer)
    RN_REQUIRE(ctx, relu != 0, "the fused stem + max-pool needs the ReLU (relu = 1); without it use "
                               "rn_conv2d_nhwc_forward_dt + rn_maxpool2d_nhwc_forward_dt");
    RN_REQUIRE(ctx, Hp >= 7 && Wp >= 7 && Hp < (1u << 14) && Wp < (1u << 14), "image size out of range");
    const uint64_t Ho = rn_conv_output_size(Hp, 7, 2, 0), Wo = rn_conv_output_size(Wp, 7, 2, 0);
    const uint64_t PH = rn_conv_output_size(Ho, 3, 2, 1), PW = rn_conv_output_size(Wo, 3, 2, 1);
    RN_REQUIRE(ctx, Wo / 8 + (Wo + 31) / 32 <= 4 * kMaxTiles, "image too wide for the fused stem (conv output width <= 128)");
    RN_REQUIRE(ctx, Wo % 8 == 0, "the fused stem needs a conv output width that is a multiple of 8");
    const int es = dtype == RN_DTYPE_BF16 ? 2 : 4, cs = dtype == RN_DTYPE_BF16 ? 4 : 3;
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(packed_weight)) & 15) == 0,
               "tensors must be 16-byte aligned");
    if (nchw) {
        RN_REQUIRE(ctx, Cin >= 1 && Cin <= 3, "the fused stem takes 1..3 input channels");
        RN_REQUIRE(ctx, (Wp - 6) % 4 == 0, "NCHW input: the image width must be a multiple of 4");
        RN_REQUIRE(ctx, (Wp - 6) / 4 <= 64, "image too wide for the fused stem (NCHW fetch: W <= 256)");
    } else {
        // the patch copy moves 16-byte pieces from the 16-byte boundary below a row start
        RN_REQUIRE(ctx, (Wp * cs * es) % (dtype == RN_DTYPE_BF16 ? 16 : 8) == 0,
                   "padded image width must make rows a multiple of 16 bytes (bf16) / 8 bytes (fp32)");
    }
    RN_REQUIRE(ctx, B * Hp * Wp * cs < (1ull << 40) && B * PH * PW < (1ull << 31), "tensor too large");
    StemParams p;
    p.in = inp;
    p.w = packed_weight;
    p.scale = scale;
    p.shift = shift;
    p.out = out;
    p.B = (int)B;
    p.Hp = (int)Hp;
    p.Wp = (int)Wp;
    p.Ho = (int)Ho;
    p.Wo = (int)Wo;
    p.PH = (int)PH;
    p.PW = (int)PW;
    p.Cin = (int)Cin;
    p.relu = relu;
    p.pairs = (int)rn_ceil_div(PH, 2);
    p.items = (unsigned)(B * (uint64_t)p.pairs);
    const size_t patch = 48 + (((size_t)kPatchRows * Wp * cs * es + 15) & ~(size_t)15);
    RN_REQUIRE(ctx, patch <= 48 * 1024, "image too wide for the fused stem (patch)");
    const size_t lds_bytes = 2 * patch + (size_t)2 * 2 * PW * kCout * sizeof(float);
    RN_REQUIRE(ctx, lds_bytes <= 160 * 1024, "image too wide for the fused stem (LDS)");
    unsigned grid = 256u;  // one block per CU (registers): persistent, items grid-stride
    if (grid > p.items) grid = p.items;
    const bool bf = dtype == RN_DTYPE_BF16;
    const void *fn = bf ? (nchw ? (const void *)stem_pool_kernel<bf16_t, true> : (const void *)stem_pool_kernel<bf16_t, false>)
                        : (nchw ? (const void *)stem_pool_kernel<float, true> : (const void *)stem_pool_kernel<float, false>);
    // more than 64 KB of dynamic LDS has to be allowed once per kernel and device (not a stream
    // operation: done on the first call, which a capturing caller makes eagerly anyway)
    int *allowed = &ctx->occupancy[248 + (bf ? 1 : 0) + (nchw ? 2 : 0)];
    if (lds_bytes > 64 * 1024 && *allowed == 0) {
        RN_HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        *allowed = 1;
    }
    if (bf && nchw)
        stem_pool_kernel<bf16_t, true><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    else if (bf)
        stem_pool_kernel<bf16_t, false><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    else if (nchw)
        stem_pool_kernel<float, true><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    else
        stem_pool_kernel<float, false><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    return rn_after_launch(ctx, what);
}

int rn_stem_pool_forward_dt(rn_ctx *ctx, int dtype, const void *inp_padded, void *out,
                            const void *packed_weight, const float *scale, const float *shift,
                            int relu, uint64_t B, uint64_t Hp, uint64_t Wp)
{
    RN_ENTER(ctx);
    return stem_pool_launch(ctx, dtype, inp_padded, out, packed_weight, scale, shift, relu, B, Hp, Wp, false, 3,
                            "rn_stem_pool_forward_dt");
}

int rn_stem_pool_nchw_forward_dt(rn_ctx *ctx, int dtype, const float *inp_nchw, void *out,
                                 const void *packed_weight, const float *scale, const float *shift,
                                 int relu, uint64_t B, uint64_t in_channels, uint64_t H, uint64_t W)
{
    RN_ENTER(ctx);
    return stem_pool_launch(ctx, dtype, inp_nchw, out, packed_weight, scale, shift, relu, B, H + 6, W + 6, true,
                            in_channels, "rn_stem_pool_nchw_forward_dt");
}

}  // extern "C"
