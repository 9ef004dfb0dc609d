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
//  * a block walks DOWN one image (or a contiguous segment of it) in items of four stem rows =
//    two pooled rows: stem rows 4j .. 4j+3 x all columns x 64 channels.  The pool windows
//    overlap by one stem row (pooled row 2j+2 also needs stem row 4j+3): instead of computing
//    that row twice (the round-2 kernel owned five rows per item, +25 % matrix work) its maxima
//    go into the pooled row of the NEXT item, which lives in the same block's LDS -- a ring of
//    five pooled rows.  Only a segment that starts inside an image computes the one stem row
//    above it once more (the "halo tile", one row in 4 * seg_len);
//  * the input patch an item needs -- 13 rows of the physically padded NHWC image, one
//    contiguous block of memory -- is copied to LDS once; the MFMA A operands are read
//    straight out of it: for fixed kernel row, the k index runs over (kw, c) = consecutive
//    floats / bf16 of the patch row, so a fragment is one ds_read_b32 (fp32, 32x32x2 MFMA) or
//    one ds_read_b128 (bf16, 32x32x16 MFMA) at base(position) + constant(k-step).  No im2col;
//  * the weights live in registers for the lifetime of the block: 77 floats or 14 x 8 bf16 per
//    lane, one 32-channel half per wave;
//  * K order: kernel row major; fp32 (kw, c) with one zero-weight slot per row (22 per row, 154);
//    bf16 (kw pair, kw parity, c of 4) with kw = 7 and c = 3 zero-weight (32 per row, 224);
//  * a wave multiplies its M tiles (4 stem rows x 8 columns each) ONE AFTER THE OTHER, and the
//    register epilogue of tile j -- y = max(acc * scale + shift, 0) (bf16: rounded to bf16), the
//    vertical part of the pool (a lane holds four rows of four adjacent columns of its channel),
//    the rest as LDS integer maxima: y >= 0, so its bit pattern orders like the value and
//    ds_max_u32 into the zeroed ring gives exactly the reference's maximum over the window's real
//    pixels (every window has one; padded taps are skipped, ops.cu:65-67) -- sits in the
//    instruction stream of tile j+1's MFMA chain, whose matrix work covers it (the round-2
//    kernel ran all tiles' chains first and then 80 values per lane of epilogue with the matrix
//    pipe idle: 0.09 of its 0.68 / 0.22 ms);
//  * the pooled rows leave as whole 128..256-byte pixels.  The stem tensor is never written.
//
// Bound: matrix pipe in fp32 (0.41 ms at peak for B=256: 154/147 of the stem's 60.4 GFLOP), HBM
// in bf16 (0.3 GB).
#include <type_traits>

#include "rn_conv_params.h"
#include "rn_lds_dma.h"

using namespace rn_gemm;

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kCout = 64, kK = 7;
constexpr int kPatchRows = 13;  // input rows behind four stem rows: 2*3 + 7
constexpr int kHaloRows = 7;    // ... behind the one stem row above a segment
constexpr int kRing = 5;        // pooled rows in LDS: three being written, two leaving
constexpr int kMaxTiles = 4;    // 4x8 M tiles per wave (4 wave rows): Wo <= 128

struct StemParams {
    const void *in;      // [B][Hp][Wp][CS] physically padded NHWC image (CS = 3 fp32, 4 bf16)
    const void *w;       // packed panel, see rn_stem_pool_pack_weight_dt
    const float *scale;  // folded batch-norm, may be null
    const float *shift;
    void *out;           // [B][PH][PW][64]
    void *y;             // WRITE_Y: the stem tensor itself, [B][Ho][Wo][64] fp32 (relu(bn(conv))), else unused
    int B, Hp, Wp, Ho, Wo, PH, PW;
    int Cin;             // NCHW input only: channels of the image (1..3)
    int pairs;           // items per image: ceil(PH / 2)
    int seg_len, segs;   // items per block, blocks per image (segs * seg_len >= pairs)
    int lrow;            // bytes between patch rows in LDS (>= Wp * pixel bytes)
    unsigned ppr_mul, ppr_shr;  // bf16: piece / (16-byte pieces per image row) as a multiply-high
    // diagnostic only (tools/stem_stamps.py): 16 shader-clock stamps per wave of the block's
    // fourth item, or null
    unsigned long long *stamps;
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
// WRITE_Y (fp32): the launch also writes the stem tensor -- for callers that name it as an output of their own
// (the deferred route of the reference's op-by-op sequence, rn_defer.hip): every value leaves once, from the
// registers of the epilogue that feeds the pool, 32 consecutive channels of a pixel per half-wave and store.
template <typename T, bool NCHW_IN, bool WRITE_Y = false>
__global__ __launch_bounds__(512, 2) void stem_pool_kernel(const StemParams p)
{
    static_assert(!WRITE_Y || sizeof(T) == 4, "the stem tensor is written in fp32 only");
    using C = Cfg<T>;
    constexpr int ES = (int)sizeof(T);
    constexpr int PIXB = C::CS * ES;            // bytes per pixel: 12 / 8
    constexpr int STEPS = kK * C::STEPS_ROW;    // MFMA k-steps: 77 / 14
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int rowb = p.Wp * PIXB;               // bytes per image row in memory
    // Pitch of a patch row in LDS.  bf16: a stem row is two patch rows, and the four stem rows of
    // an M tile are read by one ds_read_b128 -- 16 lanes a cycle, 8 of them 128 consecutive bytes
    // of one row, the other 8 of another: conflict-free when two rows are 128 bytes apart
    // modulo 256, i.e. pitch = 64 (mod 128).  Image rows of 230 pixels are 48 (mod 128): PMC showed
    // 54 % of the kernel's LDS cycles as bank conflicts.  fp32 rows are copied as one flat block
    // (they are not 16-byte multiples) and keep the memory pitch.
    const int lrow = p.lrow;
    // LDS: two input patches (item i is multiplied out of one while the next item's lands in the
    // other) and the ring of pooled rows
    const int patch_bytes = 48 + ((kPatchRows * lrow + 15) & ~15);  // slack: shifted base in front, piece overhang behind
    const int ring_row = p.PW * kCout;                               // 32-bit words of one pooled row
    unsigned *const ring = reinterpret_cast<unsigned *>(lds + 2 * patch_bytes);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n = wn * 32 + li;  // this lane's output channel
    const int n2d = p.Wo >> 3, n1d = (p.Wo + 31) >> 5;  // 4x8 tiles of an item, 1x32 tiles of a halo row
    const int b = (int)(blockIdx.x / (unsigned)p.segs);
    const int pj0 = ((int)blockIdx.x - b * p.segs) * p.seg_len;
    const int pj1 = min(pj0 + p.seg_len, p.pairs);
    if (pj0 >= pj1) return;

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
    // WRITE_Y: this image of the stem tensor (Ho x Wo x 64 fp32: below 2^31 bytes, checked by the launcher)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        WRITE_Y ? static_cast<char *>(p.y) + (size_t)b * p.Ho * p.Wo * kCout * 4 : nullptr, 0,
        WRITE_Y ? p.Ho * p.Wo * kCout * 4 : 0, 0x00020000);

    for (int i = t; i < kRing * ring_row; i += 512) ring[i] = 0u;
    if constexpr (NCHW_IN) {  // the patches' border pixels and pad channel are zero and stay zero
        for (int i = t * 16; i < 2 * patch_bytes; i += 512 * 16)
            *reinterpret_cast<i32x4 *>(lds + i) = i32x4{0, 0, 0, 0};
        __syncthreads();
    }

    // An input patch: rows [row0, row0 + nrows) of the padded image (row0 >= 0; rows past the image
    // read as zeros), one contiguous block of memory.
    //
    // Padded NHWC image (the default): LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction)
    // straight into the patch buffer the NEXT item reads, issued at the top of an item and waited
    // for (vmcnt) in front of the barrier that ends it: no staging registers, no ds_write (the
    // round-2 kernel's register-staged copy cost a wave ~1,400 cycles an item in bf16, with the
    // matrix pipe idle: stamps).  fp32 rows are not 16-byte multiples: the block is copied flat,
    // from the 16-byte boundary below its first byte to LDS offset 16 (patch row 0 then starts
    // 16 + mis bytes in).  bf16 rows land at the padded pitch `lrow`: lane l of piece q fills LDS
    // bytes [16 (64 q + l), + 16) and fetches the image bytes that belong there (pad bytes: an
    // out-of-range offset, which lands as zeros).
    const i32x4 srd = rn_dma::make_srd(static_cast<const char *>(p.in) + (size_t)b * p.Hp * rowb, p.Hp * rowb);
    const unsigned lds_base = (unsigned)(uintptr_t)((rn_dma::lds_void *)lds);
    // piece k of the calling wave (wave-instruction q = wave + 8 k of the copy); k = 0 .. kDmaPieces-1
    // covers 13 rows of up to 266 pixels.  Returns the byte offset of patch row 0 in lds[].
    constexpr int kDmaPieces = 6;
    auto patch_dma_piece = [&](int row0, int nrows, int buf, int k) -> int {
        const int rel = buf * patch_bytes + 16;  // byte offset of the copy's first piece in lds[]
        const unsigned lds0 = lds_base + (unsigned)rel;
        const int q = wave + 8 * k;
        if constexpr (sizeof(T) == 4) {
            const int g0 = row0 * rowb;
            const int mis = (int)((reinterpret_cast<uintptr_t>(p.in) + (size_t)b * p.Hp * rowb + (size_t)g0) & 15);
            const int nbytes = nrows * rowb + mis;
            // lanes past the block's last byte stay out (EXEC): what lies behind the patch in LDS
            // -- the other patch, the ring -- is in use
            if (lane * 16 + q * 1024 < nbytes)
                rn_dma::dma16(lane * 16 + q * 1024 + g0 - mis, srd, 0,
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)q * 1024u)));
            return rel + mis;
        } else {
            const int nbytes = nrows * lrow, ppr = rowb >> 4;
            const unsigned pi = (unsigned)(q * 64 + lane);
            const int r = (int)(__umulhi(pi, p.ppr_mul) >> p.ppr_shr), c16 = (int)pi - r * (lrow >> 4);
            const int off = (r < nrows && c16 < ppr) ? (row0 + r) * rowb + c16 * 16 : rn_dma::kOob;
            if ((int)pi * 16 < nbytes)  // see above
                rn_dma::dma16(off, srd, 0, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)q * 1024u)));
            return rel;
        }
    };
    auto patch_dma = [&](int row0, int nrows, int buf) -> int {
        int off = 0;
#pragma unroll
        for (int k = 0; k < kDmaPieces; ++k) off = patch_dma_piece(row0, nrows, buf, k);
        return off;
    };
    auto dma_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // NCHW_IN: the patch is assembled through registers.  A wave fetches one channel row of the
    // patch per piece -- lane q its pixels 4q..4q+3 (one 16-byte load; lanes past W/4 idle) --
    // piece k of wave w the row-channel rc = w + 8k = (patch row r) * Cin + c, i.e. channel c of
    // image row row0 + r - 3; the loads are issued for item i+1 during item i-1 and stored, as
    // four elements of T at pixel stride, in the middle of item i.  Rows outside the image are
    // stored as zeros (the buffer held another item's rows).  6 x 8 waves >= 39 channel rows.
    constexpr int kPieces = NCHW_IN ? 6 : 1;
    i32x4 stage[kPieces];
    const int W4 = (p.Wp - 6) >> 2;
    auto unit_of = [&](int k, int nrows, int &r, int &c, int &q) -> bool {
        const int rc = (t >> 6) + 8 * k;
        q = t & 63;
        r = p.Cin == 3 ? (rc * 43) >> 7 : p.Cin == 2 ? rc >> 1 : rc;  // rc / Cin for rc < 48
        c = rc - r * p.Cin;
        return r < nrows && q < W4;
    };
    auto patch_fetch = [&](int row0, int nrows) {
        if constexpr (NCHW_IN) {
            const int H = p.Hp - 6, W = p.Wp - 6;
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                int r, c, q;
                const bool ok = unit_of(k, nrows, r, c, q);
                const int ih = row0 + r - 3;
                const bool real = ok && ih >= 0 && ih < H;
                const float *src = static_cast<const float *>(p.in) +
                                   (((size_t)b * p.Cin + (real ? c : 0)) * H + (real ? ih : 0)) * W + (real ? 4 * q : 0);
                const i32x4 v = *reinterpret_cast<const i32x4 *>(src);
                stage[k] = real ? v : i32x4{0, 0, 0, 0};
            }
        }
    };
    auto patch_store = [&](int nrows, int buf) -> int {
        if constexpr (NCHW_IN) {
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                int r, c, q;
                if (!unit_of(k, nrows, r, c, q)) continue;
                T *dst = reinterpret_cast<T *>(lds + buf * patch_bytes + 16 + r * lrow) + (3 + 4 * q) * C::CS + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[i * C::CS] = (T)__int_as_float(stage[k][i]);
            }
        }
        return buf * patch_bytes + 16;
    };
    // The two pooled rows an item completed -> global, their ring slots cleared for the rows that
    // come round to them (2 * item + 5, + 6: first touched two items later).  Waves 0-3 take the
    // first row, waves 4-7 the second, each at a point of its own choosing inside the next item:
    // the two waves of a SIMD (w and w + 4) then never do this at the same time, and the one keeps
    // the matrix pipe busy while the other moves bytes.
    auto ship_row = [&](int ph) {
        if (ph >= p.PH) return;
        unsigned *src = ring + (ph % kRing) * ring_row;
        char *obase = static_cast<char *>(p.out) + ((size_t)b * p.PH + ph) * ring_row * ES;
        for (int i = (t & 255) * 4; i < ring_row; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + i);
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(obase + (size_t)i * 4) = v;
            } else {
                typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 o;
                o[0] = (bf16_t)v.x, o[1] = (bf16_t)v.y, o[2] = (bf16_t)v.z, o[3] = (bf16_t)v.w;
                *reinterpret_cast<bf16x4 *>(obase + (size_t)i * 2) = o;
            }
            *reinterpret_cast<float4 *>(src + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // y = relu(acc * scale + shift) as the bit pattern the integer maxima order by.  bf16: the
    // unfused path rounds the stem output before the pool; rounding is monotone, so the maximum of
    // the rounded values is the rounded maximum -- the ring keeps fp32 and ship() rounds once.
    auto finish = [&](float a) -> unsigned {
        float v = fmaf(a, sc, sh);
        v = v > 0.f ? v : 0.f;  // never -0.0: the bit pattern must order like the value
        return __float_as_uint(v);
    };
    // Four adjacent columns 4c .. 4c+3 of one stem row (or the vertical maximum of several) feed
    // the windows 2c (4c, 4c+1), 2c+1 (4c+1 .. 4c+3) and 2c+2 (4c+3): three LDS maxima into the
    // ring row `row`.
    auto put = [&](unsigned *row, int pw, unsigned v0, unsigned v1, unsigned v2, unsigned v3) {
        unsigned *q0 = row + pw * kCout + n;
        atomicMax(q0, max(v0, v1));
        atomicMax(q0 + kCout, max(max(v1, v2), v3));
        // past the right edge there is no window 2c+2: v3 goes to window 2c+1 once more (it is
        // part of that maximum already) instead of a branch around the instruction
        atomicMax(q0 + (pw + 2 < p.PW ? 2 * kCout : kCout), v3);
    };
    // one kernel row of an MFMA chain: the 32 positions `arow` points at x this lane's channel.
    // The chain's first step starts from a constant zero (an inline operand of the instruction,
    // not sixteen registers to clear).
    auto mfma_row = [&](f32x16 &acc, const char *patch, int arow, int kh) {
#pragma unroll
        for (int q = 0; q < C::STEPS_ROW; ++q) {
            const int s = kh * C::STEPS_ROW + q;
            f32x16 c = acc;
            if (kh == 0 && q == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) c[e] = 0.f;
            }
            if constexpr (sizeof(T) == 4) {
                const float a = *reinterpret_cast<const float *>(patch + arow + q * C::KPS * ES);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[s], c, 0, 0, 0);
            } else {
                const i32x4 a = *reinterpret_cast<const i32x4 *>(patch + arow + q * C::KPS * ES);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                             __builtin_bit_cast(bf16x8, bw[s]), c, 0, 0, 0);
            }
        }
    };

    // diagnostic stamps of one steady-state item (the fourth of the block), every wave for itself
    auto stamp = [&](int pj, int slot) {
        if (p.stamps && pj == pj0 + 3 && lane == 0)
            p.stamps[((size_t)blockIdx.x * 8 + wave) * 16 + slot] = __builtin_amdgcn_s_memtime();
    };

    // ---- start-up: the first item's patch, the halo row's if the segment starts inside the image
    int patch_off, halo_off = 0;
    if constexpr (NCHW_IN) {
        patch_fetch(8 * pj0, kPatchRows);
        patch_off = patch_store(kPatchRows, 0);
        if (pj0 > 0) {
            patch_fetch(8 * pj0 - 2, kHaloRows);
            halo_off = patch_store(kHaloRows, 1);
        }
        if (pj0 + 1 < pj1) patch_fetch(8 * (pj0 + 1), kPatchRows);
    } else {
        patch_off = patch_dma(8 * pj0, kPatchRows, 0);
        if (pj0 > 0) halo_off = patch_dma(8 * pj0 - 2, kHaloRows, 1);
        dma_landed();
    }
    __syncthreads();
    if (pj0 > 0) {
        // stem row 4*pj0 - 1, the top row of pooled row 2*pj0's windows: 1x32 tiles, one per wave row
        if (wm < n1d) {
            f32x16 acc;
            const int ox = 32 * wm + li;
            const int abase = ox < p.Wo ? 2 * ox * PIXB + lh * (C::KPS / 2) * ES : 0;
#pragma unroll
            for (int kh = 0; kh < kK; ++kh) mfma_row(acc, lds + halo_off, abase + kh * lrow, kh);
            unsigned *row = ring + ((2 * pj0) % kRing) * ring_row;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oxg = 32 * wm + 8 * g + 4 * lh;
                if (oxg >= p.Wo) continue;
                put(row, oxg >> 1, finish(acc[4 * g]), finish(acc[4 * g + 1]), finish(acc[4 * g + 2]),
                    finish(acc[4 * g + 3]));
            }
        }
        __syncthreads();  // the second patch buffer is free for item pj0 + 1
    }

    // ---- the items, ONE barrier each: while item i is multiplied out of patch[i&1], the patch of
    // item i+1 lands in the other buffer and the pooled rows item i-1 completed leave.
    const int ntw = wm < n2d ? (n2d - wm + 3) >> 2 : 0;  // this wave's tiles: wm, wm+4, ...
    // the tile in whose chain this wave ships its pooled row: the two waves of a SIMD differ
    const int ship_tile = (wave >= 4 && ntw > 1) ? 1 : 0;
    int cur = 0;
    for (int pj = pj0; pj < pj1; ++pj, cur ^= 1) {
        const char *const patch = lds + patch_off;  // patch row 0 of this item = padded image row 8*pj
        int next_off = 0;
        // Per-item copies of the lane / wave coordinates that the compiler cannot see through:
        // everything below depends only on them and on the kernel arguments, and hoisted out of
        // the item loop (the fragment addresses, the epilogue's index arithmetic) it would cost
        // hundreds of registers.
        int li_ = li, lh_ = lh, wm_ = wm;
        asm volatile("" : "+v"(li_), "+v"(lh_), "+s"(wm_));
        [[maybe_unused]] int n_ = n;
        if constexpr (WRITE_Y) asm volatile("" : "+v"(n_));
        // ring rows of pooled rows 2pj, 2pj+1 (completed by this item) and 2pj+2 (its top row only)
        unsigned *const row0 = ring + ((2 * pj) % kRing) * ring_row;
        unsigned *const row1 = ring + ((2 * pj + 1) % kRing) * ring_row;
        unsigned *const row2 = ring + ((2 * pj + 2) % kRing) * ring_row;
        // (a pooled row at or past PH -- the last item of an image -- takes its maxima like any
        // other: its ring row is never shipped and the block ends with that item)
        const int oh0 = 4 * pj;  // stem row of tile row 0; rows at or below Ho do not exist and count as 0

        stamp(pj, 0);
        // the next item's patch (the other buffer's last readers are a barrier behind).  Measured:
        // the pieces spread over the kernel rows of one tile's chain instead of issued here, 107 ->
        // 110 us in bf16 -- the wave pays for their issue wherever they stand.
        if constexpr (!NCHW_IN) {
            if (pj + 1 < pj1) next_off = patch_dma(8 * (pj + 1), kPatchRows, cur ^ 1);
        }
        stamp(pj, 1);
        auto staged_patch = [&]() {  // NCHW_IN: registers -> the other buffer, next fetch issued
            if constexpr (NCHW_IN) {
                if (pj + 1 < pj1) {
                    next_off = patch_store(kPatchRows, cur ^ 1);
                    if (pj + 2 < pj1) patch_fetch(8 * (pj + 2), kPatchRows);
                }
            }
        };
        auto ship_mine = [&]() {
            stamp(pj, 3);
            if (pj > pj0) ship_row(2 * (pj - 1) + (wave >> 2));
            stamp(pj, 4);
        };
        // M tile `tile`: lane li is (stem row li>>3, column 8*tile + (li&7)); in the accumulator a
        // lane then holds, for its channel, rows 0..3 of four adjacent columns (element e: row e>>2,
        // column 4*lh + (e&3)): the vertical part of the pool happens in registers.
        auto abase_of = [&](int tile) {
            return 2 * (li_ >> 3) * lrow + 2 * (8 * tile + (li_ & 7)) * PIXB + lh_ * (C::KPS / 2) * ES;
        };
        auto epilogue = [&](const f32x16 &acc, int tile) {
            const int pw = 4 * tile + 2 * lh_;  // (8*tile + 4*lh) / 2
            unsigned y[4][4];
#pragma unroll
            for (int dr = 0; dr < 4; ++dr) {
                const bool real = oh0 + dr < p.Ho;  // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) y[dr][i] = real ? finish(acc[4 * dr + i]) : 0u;
            }
            if constexpr (WRITE_Y) {
                // element (dr, i): stem row oh0 + dr, column 8 tile + 4 lh + i, channel n; rows past Ho do not exist
                const int voff = ((8 * tile + 4 * lh_) * kCout + n_) * 4;
#pragma unroll
                for (int dr = 0; dr < 4; ++dr) {
                    if (oh0 + dr >= p.Ho) continue;  // wave-uniform
                    const int soff = (oh0 + dr) * p.Wo * kCout * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(y[dr][i], rsrc_y, voff + i * kCout * 4, soff, 0);
                }
            }
            // windows of pooled row 2pj: stem rows (4pj-1), 4pj, 4pj+1; of row 2pj+1: 4pj+1 .. 4pj+3;
            // of row 2pj+2: 4pj+3 (and the next item's first two)
            put(row0, pw, max(y[0][0], y[1][0]), max(y[0][1], y[1][1]), max(y[0][2], y[1][2]), max(y[0][3], y[1][3]));
            put(row1, pw, max(max(y[1][0], y[2][0]), y[3][0]), max(max(y[1][1], y[2][1]), y[3][1]),
                max(max(y[1][2], y[2][2]), y[3][2]), max(max(y[1][3], y[2][3]), y[3][3]));
            put(row2, pw, y[3][0], y[3][1], y[3][2], y[3][3]);
        };
        // NT tiles one after the other, two accumulators taking turns: the chain of tile j carries
        // the epilogue of tile j-1 in its instruction stream (and, at one kernel row, this wave's
        // share of the housekeeping); only the last tile's epilogue has no matrix work beside it.
        auto run_tiles = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int ab = abase_of(wm_ + 4 * j);
#pragma unroll
                for (int kh = 0; kh < kK; ++kh) {
                    if (kh == 3) {
                        if (j == 0) staged_patch();
                        if (j == ship_tile) ship_mine();
                    }
                    if (kh == 1 && j > 0) epilogue(acc[(j - 1) & 1], wm_ + 4 * (j - 1));
                    mfma_row(acc[j & 1], patch, ab + kh * lrow, kh);
                }
                if (j == 0) stamp(pj, 5);
            }
            stamp(pj, 6);
            epilogue(acc[(NT - 1) & 1], wm_ + 4 * (NT - 1));
        };
        switch (ntw) {
            case 0: staged_patch(); ship_mine(); break;
            case 1: run_tiles(std::integral_constant<int, 1>{}); break;
            case 2: run_tiles(std::integral_constant<int, 2>{}); break;
            case 3: run_tiles(std::integral_constant<int, 3>{}); break;
            default: run_tiles(std::integral_constant<int, 4>{}); break;
        }
        stamp(pj, 7);
        if constexpr (!NCHW_IN) dma_landed();  // this wave's pieces of the next patch are in LDS
        __syncthreads();
        stamp(pj, 8);
        patch_off = next_off;
    }
    // the last item's pooled rows
    ship_row(2 * (pj1 - 1) + (wave >> 2));
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
                            uint64_t Wp, bool nchw, uint64_t Cin, const char *what, void *y = nullptr)
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
    RN_REQUIRE(ctx, Wo / 8 <= 4 * kMaxTiles, "image too wide for the fused stem (conv output width <= 128)");
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
        // bf16 rows land in LDS piece by piece at a pitch of their own: whole 16-byte pieces per row.
        // (fp32 rows are copied as one flat block from the 16-byte boundary below it: any width.)
        RN_REQUIRE(ctx, dtype != RN_DTYPE_BF16 || (Wp * cs * es) % 16 == 0,
                   "bf16: the padded image width must be even (rows of whole 16-byte pieces)");
    }
    RN_REQUIRE(ctx, B * Hp * Wp * cs < (1ull << 40) && B * PH * PW < (1ull << 31), "tensor too large");
    if (y) {
        RN_REQUIRE(ctx, nchw && dtype == RN_DTYPE_F32, "the stem tensor is written by the fp32 NCHW-input form only");
        RN_REQUIRE(ctx, y != out && y != inp && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "aliased or misaligned stem tensor");
        RN_REQUIRE(ctx, Ho * Wo * kCout * 4 < (1ull << 31), "stem tensor of one image too large");
    }
    StemParams p;
    p.in = inp;
    p.y = y;
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
    p.pairs = (int)rn_ceil_div(PH, 2);
    p.stamps = (unsigned long long *)ctx->debug_stamps;
    // A block walks seg_len items of one image; a segment that starts inside an image computes one
    // stem row more (the halo tile).  One block per CU (LDS): the launch takes ceil(blocks / CUs)
    // rounds of the longest block, so pick the segment length that minimises rounds x rows per block.
    {
        const int cus = ctx->cus;
        uint64_t best = ~0ull;
        p.seg_len = 1;
        for (int len = p.pairs; len >= 1; --len) {
            const uint64_t segs = rn_ceil_div((uint64_t)p.pairs, (uint64_t)len);
            const uint64_t rounds = rn_ceil_div(B * segs, (uint64_t)cus);
            const uint64_t cost = rounds * (4ull * (uint64_t)len + (segs > 1 ? 1 : 0) + 1);  // + start-up
            if (cost < best) best = cost, p.seg_len = len;
        }
        if (ctx->stem_items > 0) p.seg_len = ctx->stem_items < p.pairs ? ctx->stem_items : p.pairs;  // rn_ctx_set_stem_items
        p.segs = (int)rn_ceil_div((uint64_t)p.pairs, (uint64_t)p.seg_len);
    }
    p.lrow = (int)(Wp * cs * es);
    p.ppr_mul = p.ppr_shr = 0;
    if (dtype == RN_DTYPE_BF16 && !nchw) {
        // pitch = 64 (mod 128); rows are whole 16-byte pieces (checked above)
        p.lrow += (int)((64 + 128 - (p.lrow & 127)) & 127);
        rn_fast_div((unsigned)(p.lrow / 16), &p.ppr_mul, &p.ppr_shr);
    } else if (dtype == RN_DTYPE_BF16) {
        p.lrow += (int)((64 + 128 - (p.lrow & 127)) & 127);  // the NCHW fetch writes pixel by pixel: any pitch
    }
    const size_t patch = 48 + (((size_t)kPatchRows * p.lrow + 15) & ~(size_t)15);
    RN_REQUIRE(ctx, patch <= 48 * 1024, "image too wide for the fused stem (patch)");
    const size_t lds_bytes = 2 * patch + (size_t)kRing * PW * kCout * sizeof(float);
    RN_REQUIRE(ctx, lds_bytes <= 160 * 1024, "image too wide for the fused stem (LDS)");
    RN_REQUIRE(ctx, B * (uint64_t)p.segs < (1ull << 31), "too many blocks");
    const unsigned grid = (unsigned)(B * (uint64_t)p.segs);
    const bool bf = dtype == RN_DTYPE_BF16;
    const void *fn = bf ? (nchw ? (const void *)stem_pool_kernel<bf16_t, true> : (const void *)stem_pool_kernel<bf16_t, false>)
                        : (y ? (const void *)stem_pool_kernel<float, true, true>
                             : nchw ? (const void *)stem_pool_kernel<float, true> : (const void *)stem_pool_kernel<float, false>);
    // more than 64 KB of dynamic LDS has to be allowed once per kernel and device (not a stream
    // operation: done on the first call, which a capturing caller makes eagerly anyway)
    int *allowed = &ctx->occupancy[y ? 252 : 248 + (bf ? 1 : 0) + (nchw ? 2 : 0)];
    if (lds_bytes > 64 * 1024 && *allowed == 0) {
        RN_HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        *allowed = 1;
    }
    if (bf && nchw)
        stem_pool_kernel<bf16_t, true><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    else if (bf)
        stem_pool_kernel<bf16_t, false><<<grid, 512, lds_bytes, ctx->stream>>>(p);
    else if (y)
        stem_pool_kernel<float, true, true><<<grid, 512, lds_bytes, ctx->stream>>>(p);
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

// the same launch writing the stem tensor as well: stem_out [B,Ho,Wo,64] = relu(bn(conv)) in NHWC, fp32
int rn_stem_conv_pool_nchw_forward(rn_ctx *ctx, const float *inp_nchw, float *stem_out, float *pool_out,
                                   const void *packed_weight, const float *scale, const float *shift, uint64_t B,
                                   uint64_t in_channels, uint64_t H, uint64_t W)
{
    RN_ENTER(ctx);
    RN_REQUIRE(ctx, stem_out != nullptr, "null tensor");
    return stem_pool_launch(ctx, RN_DTYPE_F32, inp_nchw, pool_out, packed_weight, scale, shift, 1, B, H + 6, W + 6, true,
                            in_channels, "rn_stem_conv_pool_nchw_forward", stem_out);
}

}  // extern "C"
