// conv2d and linear: the contraction kernels.
//
// Reference: conv2dForwardKernel (cuda/ops.cu:14-48) is a direct convolution, one
// thread per output element, K = Cin*k*k global-load pairs each;
// linearForwardKernel (ops.cu:110-128) the same for x.W^T + b.
//
// Here both are ONE implicit-GEMM kernel on the matrix cores of gfx950:
//
//     out[m][n] = sum_k A[m][k] * Wp[n][k]        m = (b, oh, ow), n = out channel
//
//  * activations are NHWC, so for a fixed kernel tap (kh, kw) the K slice of a row
//    of A is a contiguous run of input channels: the K loop walks (kh, kw, 128-byte
//    channel segment) and every A row of a K tile is one 128-byte read, or zeros
//    when the tap falls into the padding (the reference skips those taps,
//    ops.cu:35-37);
//  * weights are pre-packed K-major [Cout][kh][kw][Cin] so B rows are 128-byte reads;
//  * the 3-channel stem reads a zero-padded 4-channel image: one K segment is then 8
//    consecutive pixels of one input row (fp32: 7 K tiles for the 7x7 stem) or of each of
//    two consecutive rows (bf16: 4 K tiles); unused kw / kh slots and channel 3 carry zero
//    weights;
//  * fp32: v_mfma_f32_32x32x2_f32 -- exact fp32 products and sums (bitwise an fmaf
//    chain in k order), 256 FLOP/clk/CU = the 157 TFLOP/s fp32 roof of the chip;
//    bf16 storage: v_mfma_f32_32x32x16_bf16 on the same 128-byte LDS rows;
//  * 256 threads = 4 waves as 2x2, each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles;
//    operands staged through LDS as [rows][128 B] images whose 16-byte chunks are
//    XOR-swizzled with (row>>1)&7, which makes both the ds_write_b128 of the staging
//    pass and the ds_read_b128 of the fragment reads bank-conflict free;
//  * one ds_read_b128 feeds four fp32 MFMAs (one bf16 MFMA): lane (i, h) reads chunk
//    2*ks+h of row i; A and B use the same map, so every output element sums the same
//    k set in the same fixed order whatever the tile shape (all tile candidates are
//    bit-identical);
//  * register-staged double buffering: global loads of tile t+1 are issued before
//    the MFMAs of tile t and written to the other LDS buffer after them; one
//    barrier per K tile;
//  * no vector-ALU work between the MFMAs (it competes with fp32 MFMA issue): operand
//    offsets are per-tap constants, the K position is scalar arithmetic, K segments go
//    through the buffer load's scalar offset, LDS addresses are per-thread constants
//    plus immediates (fp32: two K tiles per loop trip);
//  * every load and store is issued unconditionally through range-checked buffer
//    descriptors (padding, ragged rows, absent residual = out-of-range offset): a
//    conditional memory operation between a prefetch and its use makes hipcc wait
//    vmcnt(0), which had serialised the row stores of the epilogue;
//  * XCD-aware tile order: each of the 8 XCDs gets a contiguous range of tiles with
//    the N tiles of one M panel adjacent, so the A panel is re-read from that XCD's
//    L2; a block either owns one tile or (resident grid) walks tiles and fetches the
//    next tile's first operands during its epilogue;
//  * fused epilogue through LDS -> row-contiguous 16-byte stores: per-channel
//    scale/shift (folded batch-norm or fc bias), residual add (tile prefetched), ReLU.
//
// Roofline: MFMA-bound for K*N/(K+N) above ~80 in fp32 (every ResNet shape except the
// 56x56 convs with 64 channels on one side, which are HBM-bound); HBM/latency-bound
// everywhere with bf16 storage.  Algorithmic bytes: es*(M*K_in + N*K + M*N) (+ es*M*N
// with a residual), flops 2*M*N*K.
//
// Shapes the GEMM cannot take (Cin not a multiple of 32 and not the small-Cin stem
// form, or tensors of 2 GiB and more) run a direct fp32 kernel that keeps the
// reference's exact summation order ic -> kh -> kw.
#include <type_traits>

#include "rn_conv_params.h"

bool rn_conv_is_c4(uint64_t Cin, uint64_t k);

using namespace rn_gemm;

namespace {

constexpr int ROW_FLOATS = 32;  // LDS row = 128 bytes, addressed as 32 dwords

// Diagnostic time stamps (100 MHz wall clock) of a block's phases; off unless a debug
// buffer was attached to the context.  Nothing else reads that buffer.
__device__ __forceinline__ void stamp(unsigned long long *buf, int slot)
{
    if (buf && threadIdx.x == 0) buf[(size_t)blockIdx.x * 16 + slot] = wall_clock64();
}
// ... and of the shader clock (s_memtime), to read the clock the chip holds inside the K loop
__device__ __forceinline__ void stamp_cycles(unsigned long long *buf, int slot)
{
    if (buf && threadIdx.x == 0) buf[(size_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memtime();
}

// EPT consecutive output elements of one row: load (residual) / store as one 16-byte access
template <typename TO>
struct OutVec;
template <>
struct OutVec<float> {
    static constexpr int EPT = 4;
    static __device__ __forceinline__ void unpack(const u32x4 &x, float (&v)[4])
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(x[j]);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[4])
    {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                     __float_as_uint(v[3])};
    }
    static __device__ __forceinline__ float load1(const void *base, size_t idx)
    {
        return static_cast<const float *>(base)[idx];
    }
    static __device__ __forceinline__ void store1(void *base, size_t idx, float v)
    {
        static_cast<float *>(base)[idx] = v;
    }
};
template <>
struct OutVec<bf16_t> {
    static constexpr int EPT = 8;
    static __device__ __forceinline__ void unpack(const u32x4 &r, float (&v)[8])
    {
        const bf16x8 x = __builtin_bit_cast(bf16x8, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[8])
    {
        bf16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (bf16_t)v[j];  // round to nearest even
        return __builtin_bit_cast(u32x4, x);
    }
    static __device__ __forceinline__ float load1(const void *base, size_t idx)
    {
        return (float)static_cast<const bf16_t *>(base)[idx];
    }
    static __device__ __forceinline__ void store1(void *base, size_t idx, float v)
    {
        static_cast<bf16_t *>(base)[idx] = (bf16_t)v;
    }
};

// T: element type of activations and weights; TO: element type of the output (and residual)
// DUAL: the K loop continues through a second (input, weight-row tail) pair, see GemmParams
// XK: exact-K small-Cin form (the 7x7x3 stem as K = 147 -> 160 instead of 224), fp32 only
// CHUNK: chunked K sum with the tail tiles cut into pieces, fp32 only (see GemmParams)
template <typename T, typename TO, int BM, int BN, bool DUAL = false, bool XK = false, bool CHUNK = false>
__global__ __launch_bounds__(256, (BM * BN <= 64 * 64 ? 4 : BM * BN <= 64 * 128 ? 3 : 2)) void conv_gemm_kernel(const GemmParams p)
{
    constexpr int CH = Elem<T>::CH, ES = (int)sizeof(T);
    constexpr int AP = BM / 32;  // A rows staged per thread
    constexpr int BP = BN / 32;
    constexpr int MI = BM / 64;  // 32x32 tiles per wave along M
    constexpr int NI = BN / 64;
    constexpr int STAGE = (BM + BN) * ROW_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    // XCD-aware tile order (bijective for any tile count): virtual block v of total_tiles
    // gets the v-th tile of its XCD's contiguous range, N tiles of one M panel adjacent.
    // A block walks v = blockIdx.x, + gridDim.x, ... (persistent when the grid is smaller
    // than the tile count: the next tile's operands are fetched during this tile's epilogue).
    __builtin_amdgcn_s_setprio(3);
    stamp(p.stamps, 0);
    const unsigned total_tiles = CHUNK ? p.full_tiles : p.total_tiles;  // tiles the remap covers
    auto logical_origin = [&](unsigned logical, int &m0_, int &n0_) {
        n0_ = (int)(logical % (unsigned)p.tiles_n) * BN;
        m0_ = (int)(logical / (unsigned)p.tiles_n) * BM;
    };
    auto tile_origin = [&](unsigned v, int &m0_, int &n0_) {
        const unsigned q = total_tiles >> 3, r = total_tiles & 7, xcd = v & 7;
        unsigned pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        if (p.xg > 0) {
            // position in the (N group, M panel, N tile of the group) order -> logical tile; the tiles past
            // the last whole row of M panels keep their place at the end (a bijection for any tile count)
            const unsigned xg = (unsigned)p.xg, per_group = p.xrows * xg;
            if (pos < p.xrows * (unsigned)p.tiles_n) {
                const unsigned grp = pos / per_group, rem = pos - grp * per_group;
                const unsigned mp = rem / xg, nn = rem - mp * xg;
                pos = mp * (unsigned)p.tiles_n + grp * xg + nn;
            }
        }
        logical_origin(pos, m0_, n0_);
    };
    // work item -> tile origin, K range and split index (one item per tile unless K is split)
    auto work_item = [&](unsigned v, int &m0_, int &n0_, int &kb_, int &ke_, int &s_) {
        if constexpr (CHUNK) {
            if (v >= p.full_tiles) {  // a piece: one chunk of one tail tile, raw sum to the workspace
                const unsigned qq = v - p.full_tiles, chunk = qq / p.tail_tiles;
                logical_origin(p.full_tiles + (qq - chunk * p.tail_tiles), m0_, n0_);
                kb_ = (int)chunk * p.chunk_L;
                ke_ = min(kb_ + p.chunk_L, p.nk);
                s_ = (int)chunk;
            } else {
                tile_origin(v, m0_, n0_);
                kb_ = 0;
                ke_ = p.nk;
                s_ = -1;
            }
            return;
        }
        unsigned s = 0;
        if (p.ksplit > 1) {
            s = v / total_tiles;
            v -= s * total_tiles;
        }
        tile_origin(v, m0_, n0_);
        kb_ = (int)s * p.kchunk;
        ke_ = min(kb_ + p.kchunk, p.nk);
        s_ = (int)s;
    };
    unsigned vtile = blockIdx.x;
    int m0, n0, kb, ke, ks_idx;
    work_item(vtile, m0, n0, kb, ke, ks_idx);

    const int t = threadIdx.x;
    const int c = t & 7;    // 16-byte chunk of the 128-byte row this thread stages
    const int r0 = t >> 3;  // first row; rows r0 + 32*j

    // Operands are fetched with buffer loads: the descriptor's range check turns an
    // out-of-range offset into zeros, so a padded tap (or a row past M / Cout) costs
    // one select on the offset instead of a branch around the load.
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(DUAL ? p.in2 : p.in), 0, DUAL ? p.in2_bytes : 0, 0x00020000);
    constexpr int kOob = (int)0x80000000;  // >= num_records for every tensor we accept

    // per staged A row: byte offset of tap (0,0) chunk c, and which taps are in bounds:
    // bit kh of the low half = row ih0+kh inside [0,H), bit kw of the high half = column;
    // per staged B row: constant per-thread offset (the K tile goes into the scalar offset)
    int a_off[AP], a_mask[AP], b_off[BP];
    int a_off2[DUAL ? AP : 1];  // second source: byte offset of chunk c of the row's pixel
    auto setup_rows = [&](int m0_, int n0_) {
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int m = m0_ + r0 + 32 * j;
            if (m < p.M) {
                const int b = p.HoWo == 1 ? m : (int)(__umulhi((unsigned)m, p.mul_hw) >> p.shr_hw);
                const int rem = m - b * p.HoWo;
                const int oh = p.Wo == 1 ? rem : (int)(__umulhi((unsigned)rem, p.mul_w) >> p.shr_w);
                const int ow = rem - oh * p.Wo;
                const int ih0 = oh * p.stride - p.pad;
                const int iw0 = ow * p.stride - p.pad;
                // small-Cin forms: this thread's chunk is pixel(s) cc*chunk_dw.. of kernel row
                // tap_rows*t + hi of K tile t
                const int hi = p.tap_rows == 2 ? c >> 2 : 0, cc = p.tap_rows == 2 ? c & 3 : c;
                const int iwc = iw0 + cc * p.chunk_dw;
                a_off[j] = (((b * p.H + ih0 + hi) * p.W + iw0) * p.Cs + (XK ? 0 : cc * CH)) * ES;
                int rm = 0;
                if (p.tap_rows == 1) {
                    const int rlo = max(0, -ih0), rhi = min(p.KH, p.H - ih0);
                    rm = rhi > rlo ? ((1 << rhi) - 1) & ~((1 << rlo) - 1) : 0;
                } else {
                    for (int tt = 0; tt < p.KH; ++tt) {
                        const int kh = p.tap_rows * tt + hi, row = ih0 + kh;
                        if (kh < p.k_rows && row >= 0 && row < p.H) rm |= 1 << tt;
                    }
                }
                const int clo = max(0, -iwc), chi = min(p.KW, p.W - iwc);
                int cm = chi > clo ? ((1 << chi) - 1) & ~((1 << clo) - 1) : 0;
                if (p.chunk_dw && cc >= p.c4_chunks) cm = 0;  // chunk holds only zero-weight slots
                a_mask[j] = rm | (cm << 16);
                if constexpr (DUAL)
                    a_off2[j] = (((b * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.Cs2 + c * CH) * ES;
            } else {
                a_off[j] = 0;
                a_mask[j] = 0;
                if constexpr (DUAL) a_off2[j] = kOob;
            }
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int n = n0_ + r0 + 32 * j;
            b_off[j] = n < p.Cout ? (n * p.Ktot + c * CH) * ES : kOob;
        }
    };
    setup_rows(m0, n0);

    u32x4 ra[AP], rb[BP];
    int a_cur[AP];  // byte offset of the current tap's first segment, or kOob

    // always AP + BP loads, never a branch around a load: a masked row just gets an
    // out-of-range offset.  The K position (kh, kw, segment) is derived from the tile index
    // with scalar arithmetic; the per-row offset only changes with the tap, and the 128-byte
    // segments of one tap are walked through the scalar offset, so a K tile inside a tap
    // costs no vector ALU work at all (vector ALU work competes with the fp32 MFMA stream).
    // first: the K range of a work item may start inside a tap (split K), so its first load
    // derives the tap offsets whatever the segment
    auto load_tile = [&](int kt, u32x4 (&xa)[AP], u32x4 (&xb)[BP], auto first_c) {
        constexpr bool first = decltype(first_c)::value;
        const unsigned s_kt = (unsigned)__builtin_amdgcn_readfirstlane(kt);
        if constexpr (XK) {
            // four dword gathers per chunk: the chunk's K indices may straddle a kernel row
            int eoff[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned q = s_kt * 32u + (unsigned)(c * 4 + e);
                const unsigned kh = __umulhi(q, p.mul_kc) >> p.shr_kc;
                eoff[e] = (int)q < p.kreal ? (int)(q + kh * (unsigned)p.kskip) * 4 : kOob;
            }
#pragma unroll
            for (int j = 0; j < AP; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int off = (a_mask[j] != 0 && eoff[e] != kOob) ? a_off[j] + eoff[e] : kOob;
                    xa[j][e] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_a, off, 0, 0);
                }
            const int soff_b = (int)s_kt * 128;
#pragma unroll
            for (int j = 0; j < BP; ++j)
                xb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_off[j], soff_b, 0);
            return;
        }
        int s_cs;
        __amdgpu_buffer_rsrc_t rs = rsrc_a;
        if (DUAL && s_kt >= (unsigned)p.nk1) {
            // second source: one tap, its segments follow the first source's K tiles
            s_cs = (int)s_kt - p.nk1;
            rs = rsrc_a2;
            if (first || s_cs == 0) {
#pragma unroll
                for (int j = 0; j < AP; ++j) a_cur[j] = a_off2[DUAL ? j : 0];
            }
        } else {
            const unsigned tap = p.cseg == 1 ? s_kt : (__umulhi(s_kt, p.mul_cs) >> p.shr_cs);
            s_cs = (int)(s_kt - tap * (unsigned)p.cseg);
            if (first || s_cs == 0) {
                const unsigned ukh = p.KW == 1 ? tap : (__umulhi(tap, p.mul_kw) >> p.shr_kw);
                const int s_kh = (int)ukh, s_kw = (int)(tap - ukh * (unsigned)p.KW);
                const int toff = (s_kh * p.tap_rows * p.W + s_kw) * p.Cs * ES;
#pragma unroll
                for (int j = 0; j < AP; ++j) {
                    const bool ok = ((a_mask[j] >> s_kh) & (a_mask[j] >> (16 + s_kw)) & 1) != 0;
                    a_cur[j] = ok ? a_off[j] + toff : kOob;
                }
            }
        }
        const int seg = s_cs * 128;
#pragma unroll
        for (int j = 0; j < AP; ++j)
            xa[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_cur[j], seg, 0);
        const int soff = (int)s_kt * 128;
#pragma unroll
        for (int j = 0; j < BP; ++j)
            xb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_off[j], soff, 0);
    };

    // LDS addresses are per-thread constants plus compile-time offsets (buffer, row block):
    // (row >> 1) & 7 of rows r0 + 32j is that of r0, so one base serves every staged row
    float *const stage_base = lds + r0 * ROW_FLOATS + (c ^ ((r0 >> 1) & 7)) * 4;
    auto store_tile = [&](auto buf_c, const u32x4 (&xa)[AP], const u32x4 (&xb)[BP]) {
        const int buf = buf_c;  // compile-time (Buf0 / Buf1) or a run-time 0 / 1
#pragma unroll
        for (int j = 0; j < AP; ++j)
            *reinterpret_cast<u32x4 *>(stage_base + buf * STAGE + 32 * j * ROW_FLOATS) = xa[j];
#pragma unroll
        for (int j = 0; j < BP; ++j)
            *reinterpret_cast<u32x4 *>(stage_base + buf * STAGE + (BM + 32 * j) * ROW_FLOATS) = xb[j];
    };
    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, 1>;

    const int lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;  // swizzle term of this lane's fragment rows

    f32x16 acc[MI][NI];
    // CHUNK: running total of the finished chunks; fold_acc moves the chunk sum into it (the
    // final call leaves total + last chunk in acc for the epilogue)
    [[maybe_unused]] f32x16 tot[CHUNK ? MI : 1][CHUNK ? NI : 1];
    [[maybe_unused]] auto fold_acc = [&](bool have_total) {
        if constexpr (CHUNK) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float t_ = have_total ? tot[mi][ni][e] + acc[mi][ni][e] : acc[mi][ni][e];
                        tot[mi][ni][e] = t_;
                        acc[mi][ni][e] = 0.f;
                    }
        }
    };
    [[maybe_unused]] auto finish_acc = [&]() {  // acc = total + last chunk
        if constexpr (CHUNK) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = tot[mi][ni][e] + acc[mi][ni][e];
        }
    };

    // fragment addresses: lane (li, lh) reads chunk 2*ks+lh of its rows: fp32 -> k =
    // 8ks+4lh+{0..3}, bf16 -> k = 16ks+8lh+{0..7} (exactly the operand map of the 32x32x16
    // MFMA).  Four per-thread bases (one per k-step), everything else is an immediate.
    const float *frag_a[4], *frag_b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int pc = ((2 * ks + lh) ^ sw) * 4;
        frag_a[ks] = lds + (wr * (BM / 2) + li) * ROW_FLOATS + pc;
        frag_b[ks] = lds + BM * ROW_FLOATS + (wc * (BN / 2) + li) * ROW_FLOATS + pc;
    }
    auto compute_tile = [&](auto buf_c) {
        const int buf = buf_c;  // compile-time (Buf0 / Buf1) or a run-time 0 / 1
        // Issue priority: LOW while this wave streams MFMAs, HIGH for everything else (a wave
        // in its prologue / staging / epilogue shares the SIMD with another block's MFMAs).
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = *reinterpret_cast<const u32x4 *>(frag_a[ks] + buf * STAGE + mi * 32 * ROW_FLOATS);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *reinterpret_cast<const u32x4 *>(frag_b[ks] + buf * STAGE + ni * 32 * ROW_FLOATS);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (sizeof(T) == 4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                __uint_as_float(a[mi][j]), __uint_as_float(b[ni][j]), acc[mi][ni], 0,
                                0, 0);
                    } else {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]),
                            acc[mi][ni], 0, 0, 0);
                    }
                }
        }
        __builtin_amdgcn_s_setprio(3);
    };

    // epilogue geometry: 16 bytes of one output row per thread and pass
    constexpr int EPT = OutVec<TO>::EPT;  // output elements per thread and pass (16 bytes)
    constexpr int CV = BN / EPT;          // threads per tile row
    constexpr int RPP = 256 / CV;         // tile rows per pass of the block
    constexpr int PASSES = BM / RPP;
    const int cv = t % CV, rr = t / CV;
    const bool vec = (p.Cout % EPT) == 0;  // then n + EPT <= Cout and rows are 16-B aligned
    // Everything the epilogue touches goes through range-checked buffer descriptors and is
    // issued UNCONDITIONALLY (a row past M, a column past Cout or an absent tensor gets an
    // out-of-range offset / a zero-record descriptor): with conditional loads or stores in
    // between, hipcc has to assume the fewest younger operations and put vmcnt(0) in front
    // of every use, which serialised the 8-16 row stores of a tile (stamps: 7-9 us).
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const bool has_res = p.residual != nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        has_res ? const_cast<void *>(p.residual) : p.out, 0, has_res ? p.out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sc = __builtin_amdgcn_make_buffer_rsrc(
        has_scale ? (void *)const_cast<float *>(p.scale) : p.out, 0, has_scale ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sh = __builtin_amdgcn_make_buffer_rsrc(
        has_shift ? (void *)const_cast<float *>(p.shift) : p.out, 0, has_shift ? p.Cout * 4 : 0, 0x00020000);
    u32x4 resv[PASSES], scv[EPT / 4], shv[EPT / 4];
    // Fetching the residual tile before the K loop hides its latency completely, but its
    // registers stay live across the loop; with 16 passes (fp32 128x128: 64 VGPRs) that
    // squeezed the fragment registers and exposed LDS latency inside the MFMA stream, so
    // large tiles fetch it at the start of the epilogue instead (the accumulators are dead
    // by then).
    constexpr bool EARLY_RES = !CHUNK && PASSES * (int)sizeof(T) <= 16;

    // residual rows and the channel constants of the CURRENT tile (m0, n0)
    auto prefetch_epilogue = [&]() {
        const int n = n0 + cv * EPT;
        const bool col_ok = vec && n < p.Cout;
#pragma unroll
        for (int j4 = 0; j4 < EPT / 4; ++j4) {
            scv[j4] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc, col_ok ? (n + 4 * j4) * 4 : kOob, 0, 0);
            shv[j4] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh, col_ok ? (n + 4 * j4) * 4 : kOob, 0, 0);
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int m = m0 + rr + ps * RPP;
            const int off = (col_ok && m < p.M) ? (m * p.Cout + n) * (int)sizeof(TO) : kOob;
            resv[ps] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, off, 0, 0);
        }
    };

    stamp(p.stamps, 7);  // block set up, first loads about to be issued
    load_tile(kb, ra, rb, std::true_type{});
    for (;;) {
        const int n = n0 + cv * EPT;
        // the residual tile and the channel constants travel while the K loop runs
        if constexpr (EARLY_RES) prefetch_epilogue();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        store_tile(Buf0{}, ra, rb);
        __syncthreads();
        if (vtile == blockIdx.x) {
            stamp(p.stamps, 1);  // first operands landed and staged
            stamp_cycles(p.stamps, 8);
        }

        const unsigned vnext = vtile + gridDim.x;
        const bool has_next = vnext < p.total_work;
        int m0n = 0, n0n = 0, kbn = 0, ken = 0, ksn = 0;
        if constexpr (sizeof(T) == 4) {
            // fp32 (MFMA-bound): two K tiles per trip so that the LDS buffer of every access is
            // a compile-time offset -- no vector ALU work at all between the MFMAs.  Inside the
            // loop both loads are unconditional (no phi copies of the staging registers); the
            // last one or two tiles are peeled.
            int kt = kb;
            [[maybe_unused]] int fold_at = kb + p.chunk_L;  // CHUNK: chunk_L is even
            [[maybe_unused]] bool folded = false;
            while (kt + 2 < ke) {
                if constexpr (CHUNK) {
                    if (kt == fold_at) {  // uniform; only whole tiles get here (a piece is one chunk)
                        fold_acc(folded);
                        folded = true;
                        fold_at += p.chunk_L;
                    }
                }
                load_tile(kt + 1, ra, rb, std::false_type{});
                compute_tile(Buf0{});
                store_tile(Buf1{}, ra, rb);
                __syncthreads();
                load_tile(kt + 2, ra, rb, std::false_type{});
                compute_tile(Buf1{});
                store_tile(Buf0{}, ra, rb);
                __syncthreads();
                kt += 2;
            }
            if constexpr (CHUNK) {
                if (kt == fold_at) {  // the last chunk is the one or two peeled tiles
                    fold_acc(folded);
                    folded = true;
                }
            }
            if (kt + 1 < ke) {  // two tiles left: kt in buffer 0, kt + 1 still to fetch
                load_tile(kt + 1, ra, rb, std::false_type{});
                compute_tile(Buf0{});
                store_tile(Buf1{}, ra, rb);
                __syncthreads();
                compute_tile(Buf1{});
            } else {  // one tile left, in buffer 0
                compute_tile(Buf0{});
            }
            if constexpr (CHUNK) {
                if (folded) finish_acc();  // total + last chunk: the order the pieces are added in
            }
            __syncthreads();
        } else {
            // bf16 (latency-bound): the compact loop keeps the register count, and with it the
            // number of resident blocks, where the unrolled form costs 10 % throughput
            [[maybe_unused]] int fold_at = kb + p.chunk_L;
            [[maybe_unused]] bool folded = false;
            for (int kt = kb; kt < ke; ++kt) {
                if constexpr (CHUNK) {
                    if (kt == fold_at) {  // uniform; only whole tiles get here (a piece is one chunk)
                        fold_acc(folded);
                        folded = true;
                        fold_at += p.chunk_L;
                    }
                }
                const bool more = kt + 1 < ke;
                if (more) load_tile(kt + 1, ra, rb, std::false_type{});
                compute_tile((kt - kb) & 1);
                if (more) store_tile((kt - kb + 1) & 1, ra, rb);
                __syncthreads();
            }
            if constexpr (CHUNK) {
                if (folded) finish_acc();
            }
        }
        // next tile of this block: its first K tile starts travelling before the epilogue
        if (has_next) {
            work_item(vnext, m0n, n0n, kbn, ken, ksn);
            setup_rows(m0n, n0n);
            load_tile(kbn, ra, rb, std::true_type{});
        }

        // epilogue.  The accumulators go through LDS (free after the K loop) so that global
        // traffic is row-contiguous 16-byte accesses: C/D map of the 32x32 MFMA is
        // col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5); a ds_write_b32 of one register
        // puts 32 consecutive columns of two rows, conflict-free.
        if (vtile == blockIdx.x) {
            stamp_cycles(p.stamps, 9);
            stamp(p.stamps, 2);  // K loop done
        }
        float *Cs = lds;  // [BM][BN] fp32
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float *dst = Cs + (wr * (BM / 2) + mi * 32 + 4 * lh) * BN + wc * (BN / 2) + ni * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[mi][ni][e];
        }
    if constexpr (!EARLY_RES) prefetch_epilogue();
    if (vtile == blockIdx.x) stamp(p.stamps, 5);  // accumulators written to LDS (issued)
    __syncthreads();
    if (vtile == blockIdx.x) stamp(p.stamps, 6);  // ... by every wave

    // output of this work item: the tensor itself, or partial-sum slice ks_idx of the workspace
    const bool raw = CHUNK && ks_idx >= 0;  // a piece: plain chunk sum, no epilogue
    void *const out_base = raw ? static_cast<char *>(p.ws) + (long long)ks_idx * p.ws_stride
                               : static_cast<char *>(p.out) + (long long)ks_idx * p.split_stride;
    const __amdgpu_buffer_rsrc_t rsrc_o =
        __builtin_amdgcn_make_buffer_rsrc(out_base, 0, p.out_bytes, 0x00020000);
    if (sizeof(TO) == 4 && p.out_nchw && !raw) {
        // NCHW output (drop-in route): lanes along the channels -- conflict-free column reads of
        // the staged tile -- and four consecutive pixels of one channel per 16-byte store.  The
        // stores of a wave scatter over 64 channel planes; consecutive steps of a lane continue
        // its plane, so L2 sees whole lines, and against a separate transpose launch (a read and
        // a write of the whole tensor) the address unit's time is cheap.
        const int c = t % BN, nn = n0 + c;
        const bool ch_ok = nn < p.Cout;
        const float scl = (has_scale && ch_ok) ? p.scale[nn] : 1.f;
        const float shf = (has_shift && ch_ok) ? p.shift[nn] : -0.f;
        for (int q = t / BN; q < BM / 4; q += 256 / BN) {
            const int m = m0 + 4 * q;
            if (!ch_ok || m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = fmaf(Cs[(4 * q + j) * BN + c], scl, shf);
                v[j] = p.relu ? fmaxf(y, 0.f) : y;
            }
            const int b = (int)(__umulhi((unsigned)m, p.mul_hw) >> p.shr_hw), hw = m - b * p.HoWo;
            if (m + 3 < p.M && hw + 3 < p.HoWo) {
                u32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(v[j]);
                __builtin_amdgcn_raw_buffer_store_b128(o, rsrc_o, ((b * p.Cout + nn) * p.HoWo + hw) * 4, 0, 0);
            } else {
                for (int j = 0; j < 4 && m + j < p.M; ++j) {
                    const int mj = m + j;
                    const int bj = (int)(__umulhi((unsigned)mj, p.mul_hw) >> p.shr_hw);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j]), rsrc_o,
                                                          ((bj * p.Cout + nn) * p.HoWo + mj - bj * p.HoWo) * 4, 0, 0);
                }
            }
        }
    } else if (vec) {
        // straight-line: 16-byte LDS read, channel affine, residual, ReLU, 16-byte store per pass.
        // The two launch-uniform switches (residual, ReLU) pick one of four copies of the loop:
        // as per-element selects they cost as many vector instructions as the arithmetic itself.
        float sc[EPT], sh[EPT];
#pragma unroll
        for (int j4 = 0; j4 < EPT / 4; ++j4)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc[4 * j4 + j] = (has_scale && !raw) ? __uint_as_float(scv[j4][j]) : 1.f;
                // -0.0 as the neutral addend keeps a -0.0 convolution result bit-exact
                sh[4 * j4 + j] = (has_shift && !raw) ? __uint_as_float(shv[j4][j]) : -0.f;
            }
        const bool col_ok = n < p.Cout;
        auto rows_out = [&](auto with_res, auto with_relu) {
            constexpr bool RES = decltype(with_res)::value, RELU = decltype(with_relu)::value;
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int row = rr + ps * RPP;
                const int m = m0 + row;
                float v[EPT], res[EPT];
                if constexpr (RES) OutVec<TO>::unpack(resv[ps], res);
#pragma unroll
                for (int j4 = 0; j4 < EPT / 4; ++j4) {
                    const float4 x = *reinterpret_cast<const float4 *>(Cs + row * BN + cv * EPT + 4 * j4);
                    v[4 * j4] = x.x, v[4 * j4 + 1] = x.y, v[4 * j4 + 2] = x.z, v[4 * j4 + 3] = x.w;
                }
                // two outputs per instruction: v_pk_fma_f32 / v_pk_add_f32 (the same IEEE results)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < EPT; j += 2) {
                    f32x2 y = __builtin_elementwise_fma(f32x2{v[j], v[j + 1]}, f32x2{sc[j], sc[j + 1]},
                                                        f32x2{sh[j], sh[j + 1]});
                    if constexpr (RES) y = y + f32x2{res[j], res[j + 1]};
                    v[j] = RELU ? fmaxf(y[0], 0.f) : y[0];
                    v[j + 1] = RELU ? fmaxf(y[1], 0.f) : y[1];
                }
                const int off = (col_ok && m < p.M) ? (m * p.Cout + n) * (int)sizeof(TO) : kOob;
                __builtin_amdgcn_raw_buffer_store_b128(OutVec<TO>::pack(v), rsrc_o, off, 0, 0);
            }
        };
        const bool do_res = has_res && !raw, do_relu = p.relu && !raw;
        if (do_res) {
            if (do_relu) rows_out(std::true_type{}, std::true_type{});
            else rows_out(std::true_type{}, std::false_type{});
        } else {
            if (do_relu) rows_out(std::false_type{}, std::true_type{});
            else rows_out(std::false_type{}, std::false_type{});
        }
    } else if (n < p.Cout) {
        // ragged channel count (Cout % EPT != 0): element-wise, rare
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = rr + ps * RPP;
            const int m = m0 + row;
            if (m >= p.M) continue;
            for (int j = 0; j < EPT; ++j) {
                if (n + j >= p.Cout) break;
                const size_t o = (size_t)m * p.Cout + n + j;
                float y = Cs[row * BN + cv * EPT + j];
                y = fmaf(y, has_scale ? p.scale[n + j] : 1.f, has_shift ? p.shift[n + j] : -0.f);
                if (has_res) y += OutVec<TO>::load1(p.residual, o);
                y = p.relu ? fmaxf(y, 0.f) : y;
                OutVec<TO>::store1(out_base, o, y);
            }
        }
    }
        if (p.stamps && vtile == blockIdx.x) {
            stamp(p.stamps, 3);  // epilogue stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(p.stamps, 4);  // ... and acknowledged
        }
        if (!has_next) break;
        __syncthreads();  // the C staging area becomes operand stage 0 again
        vtile = vnext;
        m0 = m0n;
        n0 = n0n;
        kb = kbn;
        ke = ken;
        ks_idx = ksn;
    }
}

// any shape: one thread per output element, the reference's loop order and one fp32
// fmaf chain (ops.cu:30-45), lanes along the contiguous output dimension.
struct DirectParams {
    const float *in;
    const float *w;
    float *out;
    const float *scale;
    const float *shift;
    const float *residual;
    int relu;
    int k, stride, pad, Ho, Wo, Cin, Cs, Cout, H, W;
    int nhwc;      // activation layout
    int w_packed;  // 0: OIHW, 1: [Cout][kh][kw][Cin], 2: small-Cin panel [Cout][kh][8][4]
    uint64_t total;
};

__global__ __launch_bounds__(256) void conv_direct_kernel(const DirectParams p)
{
    const uint64_t gstride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += gstride) {
        int oc, oh, ow;
        uint64_t b;
        if (p.nhwc) {
            oc = (int)(i % (uint64_t)p.Cout);
            uint64_t q = i / (uint64_t)p.Cout;
            ow = (int)(q % (uint64_t)p.Wo);
            q /= (uint64_t)p.Wo;
            oh = (int)(q % (uint64_t)p.Ho);
            b = q / (uint64_t)p.Ho;
        } else {
            ow = (int)(i % (uint64_t)p.Wo);
            uint64_t q = i / (uint64_t)p.Wo;
            oh = (int)(q % (uint64_t)p.Ho);
            q /= (uint64_t)p.Ho;
            oc = (int)(q % (uint64_t)p.Cout);
            b = q / (uint64_t)p.Cout;
        }
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
        float sum = 0.f;
        for (int ic = 0; ic < p.Cin; ++ic) {
            for (int kh = 0; kh < p.k; ++kh) {
                const int ih = ih0 + kh;
                if (ih < 0 || ih >= p.H) continue;
                for (int kw = 0; kw < p.k; ++kw) {
                    const int iw = iw0 + kw;
                    if (iw < 0 || iw >= p.W) continue;
                    const uint64_t ii = p.nhwc ? (((b * p.H + ih) * p.W + iw) * p.Cs + ic)
                                               : (((b * p.Cin + ic) * p.H + ih) * p.W + iw);
                    uint64_t wi;
                    if (p.w_packed == 0)
                        wi = (((uint64_t)oc * p.Cin + ic) * p.k + kh) * p.k + kw;
                    else if (p.w_packed == 1)
                        wi = (((uint64_t)oc * p.k + kh) * p.k + kw) * p.Cin + ic;
                    else
                        wi = (((uint64_t)oc * p.k + kh) * 8 + kw) * 4 + ic;
                    sum = fmaf(p.in[ii], p.w[wi], sum);
                }
            }
        }
        if (p.scale) {
            sum = fmaf(sum, p.scale[oc], p.shift ? p.shift[oc] : 0.f);
        } else if (p.shift) {
            sum += p.shift[oc];
        }
        if (p.residual) sum += p.residual[i];
        if (p.relu) sum = fmaxf(sum, 0.f);
        p.out[i] = sum;
    }
}

// split K, second step: out = epilogue(partial[0] + partial[1] + ... in that order); four
// consecutive channels per thread (Cout % 4 == 0), the epilogue of conv_gemm_kernel
struct FinishParams {
    const float *partial;
    void *out;
    const float *scale, *shift;
    const void *residual;
    int relu, splits, Cout;
    uint64_t slice;   // elements to finish (rows * Cout, starting at the pointers given)
    uint64_t stride;  // elements between consecutive partial slices
};

template <typename TO>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const FinishParams p)
{
    const uint64_t n4 = p.slice / 4, gstride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += gstride) {
        float4 v = *reinterpret_cast<const float4 *>(p.partial + 4 * i);
        for (int s = 1; s < p.splits; ++s) {
            const float4 x = *reinterpret_cast<const float4 *>(p.partial + (uint64_t)s * p.stride + 4 * i);
            v.x += x.x, v.y += x.y, v.z += x.z, v.w += x.w;
        }
        const int n = (int)((4 * i) % (uint64_t)p.Cout);
        float y[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sc = p.scale ? p.scale[n + j] : 1.f;
            const float sh = p.shift ? p.shift[n + j] : -0.f;
            float r = fmaf(y[j], sc, sh);
            if (p.residual) r += OutVec<TO>::load1(p.residual, 4 * i + j);
            y[j] = p.relu ? fmaxf(r, 0.f) : r;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) OutVec<TO>::store1(p.out, 4 * i + j, y[j]);
    }
}

bool fits_i32(uint64_t v) { return v < (1ull << 31); }

void fast_div(unsigned d, unsigned *mul, unsigned *shr) { rn_fast_div(d, mul, shr); }

// Blocks of one instantiation that fit a CU at once (registers and LDS) on the context's device,
// asked once per context and instantiation: the answer lives in the context, not in the process.
template <typename T, typename TO, int BM, int BN, bool DUAL, bool XK, bool CHUNK>
int resident_blocks_per_cu(rn_ctx *ctx)
{
    constexpr int tile = (BM == 128 ? 0 : 2) + (BN == 128 ? 0 : 1);
    constexpr int types = sizeof(T) == 4 ? 0 : sizeof(TO) == 4 ? 1 : 2;
    constexpr int id = tile + 4 * ((DUAL ? 1 : 0) + 2 * (XK ? 1 : 0) + 4 * (CHUNK ? 1 : 0)) + 32 * types;
    static_assert(id < (int)(sizeof(ctx->occupancy) / sizeof(ctx->occupancy[0])), "occupancy table");
    if (ctx->occupancy[id] == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_gemm_kernel<T, TO, BM, BN, DUAL, XK, CHUNK>, 256,
                                                         0) != hipSuccess ||
            nb < 1)
            nb = 1;
        ctx->occupancy[id] = nb;
    }
    return ctx->occupancy[id];
}

template <typename T, typename TO, int BM, int BN, bool DUAL, bool XK, bool CHUNK = false>
void launch_one(rn_ctx *ctx, GemmParams &p, bool persistent)
{
    unsigned grid = p.grid_items;
    if (persistent) {
        const unsigned slots = 256u * (unsigned)resident_blocks_per_cu<T, TO, BM, BN, DUAL, XK, CHUNK>(ctx);
        if (grid > slots) grid = slots;
    }
    conv_gemm_kernel<T, TO, BM, BN, DUAL, XK, CHUNK><<<dim3(grid), dim3(256), 0, ctx->stream>>>(p);
}

template <typename T, typename TO, bool DUAL = false, bool XK = false>
void launch_tiles(rn_ctx *ctx, GemmParams &p, int BMsel, int BNsel, bool persistent)
{
    if (BMsel == 128 && BNsel == 128)
        launch_one<T, TO, 128, 128, DUAL, XK>(ctx, p, persistent);
    else if (BMsel == 128 && BNsel == 64)
        launch_one<T, TO, 128, 64, DUAL, XK>(ctx, p, persistent);
    else if (BMsel == 64 && BNsel == 128)
        launch_one<T, TO, 64, 128, DUAL, XK>(ctx, p, persistent);
    else
        launch_one<T, TO, 64, 64, DUAL, XK>(ctx, p, persistent);
}

// chunked K sum (fp32, and bf16 operands with an fp32 result: the fc of a bf16 model; the 128x128 tile has
// no registers to spare for a second accumulator)
void launch_tiles_chunked(rn_ctx *ctx, GemmParams &p, int BMsel, int BNsel, bool persistent, bool bf16_in)
{
    if (bf16_in)
        launch_one<bf16_t, float, 64, 64, false, false, true>(ctx, p, persistent);
    else if (BMsel == 128)
        launch_one<float, float, 128, 64, false, false, true>(ctx, p, persistent);
    else if (BNsel == 128)
        launch_one<float, float, 64, 128, false, false, true>(ctx, p, persistent);
    else
        launch_one<float, float, 64, 64, false, false, true>(ctx, p, persistent);
}

// Which order the tiles are dealt to the XCDs in (GemmParams::xg).  Every XCD has its own 4 MB L2; what an
// L2 misses comes from the Infinity Cache or HBM and is what FETCH_SIZE counts.  The blocks resident on an XCD
// (R = 32 CUs x the tile's blocks per CU) work on nt N tiles x R / nt M panels at a time and need
// nt weight-row slabs (BN x K) and R / nt input slabs (BM x Cin) over and over while their K loops run:
// when that working set fits the L2, the launch fetches about G x the input + 8 / G x the weight panel
// (G groups of N tiles: an XCD keeps to one group, the input rows are read by G XCDs); when it does not, blocks
// that have drifted apart in K evict each other's rows and the launch fetches several times that.
// Measured on the stage 3-4 shapes at B = 256 (tools/xcd_order_ab.sh, profiles/round4/xcd_tile_order_ab.txt):
// layer4 conv3 (K = 512, N = 2048: 4.2 MB of weights) 430-630 MB in the logical order, 81-91 MB in two groups
// (algorithmic 30 MB + the output), 113 in four, 214 in eight; 2048 -> 512 at 7 x 7 253 MB logical, 281 in two
// groups (twice the input for a weight panel that almost fits: not worth it); 256 -> 1024 at 14 x 14 fits in the
// logical order (60 MB for 52); the 3 x 3 layers with 512 channels fit in NO grouping (a slab of weight rows is
// 1.2 MB, and 128 resident blocks need 17 MB of input slabs across their nine tap passes in the grouped orders:
// 532 MB logical, 454 / 839 / 1545 in 2 / 4 / 8 groups) and keep the logical order.
// The order changes which block computes a tile, never a bit of the result.
void choose_tile_order(rn_ctx *ctx, GemmParams &p, unsigned remap_tiles, int BM, int BN)
{
    p.xg = 0;
    p.xrows = 0;
    const unsigned tn = (unsigned)p.tiles_n;
    if (tn < 2 || remap_tiles < 8 * tn) return;  // too few rows of M panels for eight ranges per group
    int groups = ctx->xcd_groups;                // forced (RN_XCD_NGROUPS / rn_ctx_set_xcd_groups): A/B runs
    if (groups <= 0) {
        const double es = p.w_bytes / ((double)p.Cout * p.Ktot);  // bytes per element
        const double w_slab = (double)BN * p.Ktot * es;           // weight rows of one N tile
        const double a_slab = (double)BM * es * (p.Cs + (p.in2 ? p.Cs2 : 0));    // input rows of one M panel
        const double R = 32.0 * (BM * BN <= 64 * 64 ? 4 : BM * BN <= 64 * 128 ? 3 : 2);
        const double A = (double)p.in_bytes + (p.in2 ? (double)p.in2_bytes / (p.stride2 * p.stride2) : 0.0);
        const double Wb = (double)p.w_bytes, MB = 1024.0 * 1024.0;
        const double panels = (double)(remap_tiles / tn);  // M panels of the launch
        const bool taps = p.KH * p.KW > 1;
        // bytes a grouping fetches.  A 1x1 convolution streams its input rows once per (M panel, XCD that has
        // tiles of it) -- the N tiles of a panel are adjacent in the order and run together -- so what has to
        // stay in the L2 is the group's slice of the weight panel (2.5 of the 4 MB: the input, output and
        // residual streams pass through as well); a k x k convolution returns to its input rows once per tap,
        // so the input slabs of the resident M panels have to stay too.  A slice that does not stay is
        // streamed again by every generation of resident tiles, about three times per generation once the
        // blocks have drifted apart in K (measured: 4.5 x for 2048 -> 512 at 7 x 7, 12-18 x for 512 -> 2048).
        auto cost = [&](int g) {
            const double nt = (double)tn / g, mp = R / nt < 1 ? 1 : R / nt;
            const bool stays = taps ? nt * w_slab + mp * a_slab <= 3.8 * MB : nt * w_slab <= 2.5 * MB;
            const double generations = panels / (8.0 / g) / mp;
            const double again = stays ? 1.0 : 3.0 * (generations < 1 ? 1 : generations);
            if (taps && !stays && g > 1) return 1e300;  // (measured: grouping a thrashing 3x3 only adds input reads)
            return (g > 1 ? 1.3 : 1.0) * g * A + 8.0 * (Wb / g) * again;
        };
        groups = 1;
        double best = cost(1);
        for (int g = 2; g <= 8; g *= 2) {
            if (tn % (unsigned)g) break;
            const double c = cost(g);
            if (c < 0.8 * best) best = c, groups = g;  // (the model is good to about 25 %)
        }
    }
    if (groups > 8) groups = 8;
    while (groups > 1 && tn % (unsigned)groups) groups /= 2;
    if (groups <= 1) return;
    p.xg = (int)(tn / (unsigned)groups);
    p.xrows = remap_tiles / tn;
}

// GEMM launch on NHWC data with packed weights.  Caller has checked eligibility.
// dt_in: element type of activations and weights; dt_out: of the output and the residual.
int launch_gemm(rn_ctx *ctx, int dt_in, int dt_out, const void *inp, void *out, const void *packed,
                uint64_t k, uint64_t stride, uint64_t pad, uint64_t h_out, uint64_t w_out,
                uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W,
                const rn_epilogue *ep, const char *what, const rn_conv_second *second = nullptr,
                bool exact = false, bool out_nchw = false)
{
    const int es = dt_in == RN_DTYPE_BF16 ? 2 : 4;
    const int bke = 128 / es;
    GemmParams p;
    p.xg = 0;  // tile order: the logical one unless choose_tile_order says otherwise (tiled launches only)
    p.xrows = 0;
    // NCHW output: fp32, no residual (it would be NHWC); a 1x1 output image is the same in both
    p.out_nchw = out_nchw && dt_out == RN_DTYPE_F32 && !(ep && ep->residual) && h_out * w_out > 1;
    p.in = inp;
    p.w = packed;
    p.out = out;
    p.scale = ep ? ep->scale : nullptr;
    p.shift = ep ? ep->shift : nullptr;
    p.residual = ep ? ep->residual : nullptr;
    p.relu = ep ? ep->relu : 0;
    const bool c4 = !exact && rn_conv_is_c4(Cin, k);
    p.H = (int)H;
    p.W = (int)W;
    p.Cs = c4 ? 4 : (int)Cin;
    p.Ho = (int)h_out;
    p.Wo = (int)w_out;
    p.Cout = (int)Cout;
    const bool c4pair = c4 && dt_in == RN_DTYPE_BF16;  // two kernel rows per K tile
    p.tap_rows = c4pair ? 2 : 1;
    p.k_rows = (int)k;
    p.KH = c4pair ? (int)rn_ceil_div(k, 2) : (int)k;
    p.KW = c4 ? 1 : (int)k;
    p.stride = (int)stride;
    p.pad = (int)pad;
    p.cseg = c4 ? 1 : (int)(Cin / bke);
    p.chunk_dw = c4 ? 16 / (4 * es) : 0;  // pixels of a 4-channel image per 16-byte chunk
    p.c4_chunks = c4 ? (int)rn_ceil_div(k, p.chunk_dw) : 0;  // <= 4 in the two-row form (k <= 8)
    p.M = (int)(B * h_out * w_out);
    p.nk = p.KH * p.KW * p.cseg;
    p.nk1 = p.nk;
    p.in2 = nullptr;
    p.in2_bytes = p.H2 = p.W2 = p.Cs2 = p.stride2 = 0;
    if (second) {
        p.in2 = second->inp;
        p.H2 = (int)second->H;
        p.W2 = (int)second->W;
        p.Cs2 = (int)second->in_channels;
        p.stride2 = (int)second->stride;
        p.in2_bytes = (int)(B * second->H * second->W * second->in_channels * es);
        p.nk += (int)(second->in_channels / bke);
    }
    p.Ktot = p.nk * bke;
    p.kc = p.kskip = p.kreal = 0;
    p.mul_kc = p.shr_kc = 0;
    if (exact) {
        p.kreal = (int)(k * k * Cin);
        p.kc = (int)(k * Cin);
        p.kskip = (int)((W - k) * Cin);
        p.nk = p.nk1 = (int)rn_ceil_div((uint64_t)p.kreal, 32);
        p.Ktot = p.nk * 32;
        p.cseg = 1;
        fast_div((unsigned)p.kc, &p.mul_kc, &p.shr_kc);
    }
    p.HoWo = p.Ho * p.Wo;
    fast_div((unsigned)p.HoWo, &p.mul_hw, &p.shr_hw);
    fast_div((unsigned)p.Wo, &p.mul_w, &p.shr_w);
    fast_div((unsigned)p.cseg, &p.mul_cs, &p.shr_cs);
    fast_div((unsigned)p.KW, &p.mul_kw, &p.shr_kw);
    p.stamps = (unsigned long long *)ctx->debug_stamps;
    p.in_bytes = (int)(B * H * W * (uint64_t)p.Cs * es);
    p.w_bytes = (int)(Cout * (uint64_t)p.Ktot * es);
    p.out_bytes = (int)(B * h_out * w_out * Cout * (uint64_t)(dt_out == RN_DTYPE_BF16 ? 2 : 4));

    // bf16 on 256-wide block tiles (rn_conv_wide.hip): candidates 9.. of the tuner; without a
    // tuned choice, the K-heavy layers whose tiles fill at least half the chip.  Same k order
    // per output element as every other candidate, so this too only changes the speed.
    if (dt_in == RN_DTYPE_BF16 && dt_out == RN_DTYPE_BF16 && ctx->split_k <= 1) {
        const int nwide = rn_conv_wide_count();
        int which = -1;
        // 3x3 / 64 -> 64 channels: the strip kernel (weights in registers, input ring in LDS);
        // the candidate after the wide tiles, and the untuned choice where it applies
        if ((ctx->conv_tile == 9 + nwide || (ctx->conv_tile == 0 && p.M >= 256 * 256)) && second == nullptr &&
            rn_conv_strip_eligible(p)) {
            rn_conv_strip_launch(ctx, p);
            return rn_after_launch(ctx, what);
        }
        if (ctx->conv_tile > 8 && ctx->conv_tile <= 8 + nwide) {
            if (rn_conv_wide_eligible(p, ctx->conv_tile - 9)) which = ctx->conv_tile - 9;
        } else if (ctx->conv_tile == 0 && p.nk >= 8 && Cout >= 128) {
            // relative efficiency of a full tile (the 64x64 wave tiles of the 128-wide forms
            // read twice the LDS bytes per MFMA)
            static const double eff_wide[6] = {1.00, 0.80, 0.80, 0.50, 0.90, 0.55};
            double best = 1e300;
            for (int wi = 0; wi < nwide; ++wi) {
                int bm, bn;
                rn_conv_wide_tile(wi, &bm, &bn);
                if (!rn_conv_wide_eligible(p, wi) || (uint64_t)bn > 2 * Cout) continue;
                const uint64_t tiles = rn_ceil_div((uint64_t)p.M, bm) * rn_ceil_div(Cout, bn);
                if (tiles < 128) continue;
                const double cost = (double)rn_ceil_div(tiles, 256) * bm * bn / eff_wide[wi];
                if (cost < best * 0.999) {
                    best = cost;
                    which = wi;
                }
            }
        }
        if (which >= 0) {
            int bm, bn;
            rn_conv_wide_tile(which, &bm, &bn);
            p.tiles_n = (int)rn_ceil_div(Cout, bn);
            const uint64_t total = rn_ceil_div((uint64_t)p.M, bm) * (uint64_t)p.tiles_n;
            RN_REQUIRE(ctx, fits_i32(total), "too many tiles");
            p.total_tiles = (unsigned)total;
            p.ksplit = 1;
            p.kchunk = p.nk;
            p.total_work = p.total_tiles;
            p.grid_items = p.total_tiles;
            p.split_stride = 0;
            p.chunk_L = p.nk;
            p.full_tiles = p.total_tiles;
            p.tail_tiles = 0;
            p.ws = nullptr;
            p.ws_stride = 0;
            rn_conv_wide_launch(ctx, p, which, second != nullptr);
            return rn_after_launch(ctx, what);
        }
    }

    // tile choice: the contraction is matrix-core bound, so a launch takes about
    // ceil(tiles / 256 CUs) rounds of one tile's MFMA time; pick the candidate with the
    // least (rounds * tile area / relative tile efficiency), i.e. the least padded,
    // best balanced cover of the 256 CUs.  rn_model_tune measures instead of guessing.
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    // relative efficiency of a full tile, measured on 3x3 and 1x1 layers at B=256: fp32 is
    // bound by the matrix pipe and likes many small tiles (occupancy, tails); bf16 is bound by
    // operand traffic per MFMA and likes large ones
    static const double eff_f32[4] = {0.84, 0.95, 0.96, 1.00}, eff_bf16[4] = {1.00, 0.92, 0.95, 0.80};
    const double *cand_eff = dt_in == RN_DTYPE_BF16 ? eff_bf16 : eff_f32;
    int BMsel = 128, BNsel = 128;
    bool persistent;
    if (ctx->conv_tile >= 1 && ctx->conv_tile <= 8) {
        // candidates 1-4: one block per tile; 5-8: the same tiles walked by a resident grid
        BMsel = cand[(ctx->conv_tile - 1) & 3][0];
        BNsel = cand[(ctx->conv_tile - 1) & 3][1];
        persistent = ctx->conv_tile > 4;
    } else {
        double best = 1e300;
        for (int ci = 0; ci < 4; ++ci) {
            const uint64_t tm = rn_ceil_div((uint64_t)p.M, cand[ci][0]);
            const uint64_t tn = rn_ceil_div(Cout, cand[ci][1]);
            const double rounds = (double)rn_ceil_div(tm * tn, 256);
            const double cost = rounds * cand[ci][0] * cand[ci][1] / cand_eff[ci];
            if (cost < best * 0.999) {
                best = cost;
                BMsel = cand[ci][0];
                BNsel = cand[ci][1];
            }
        }
        persistent = true;
    }
    // Chunked K sum: a property of the LAYER (element type, kind, K), never of the batch size or
    // the tile, so that every launch of the layer adds the same products in the same order.
    // (bf16 operands with fp32 results: the fc of a bf16 model -- 64 tiles of 32 K steps at B = 256, 30 us as
    // one block per tile on a quarter of the CUs)
    const bool chunk_bf16 = dt_in == RN_DTYPE_BF16 && dt_out == RN_DTYPE_F32;
    const bool chunked = ((dt_in == RN_DTYPE_F32 && dt_out == RN_DTYPE_F32) || chunk_bf16) && !second && !exact &&
                         p.nk >= 32 && Cout % 4 == 0;  // (nk >= 16 measured: -0.3 % on the fp32 step)
    if (chunked && BMsel == 128 && BNsel == 128) BNsel = 64;
    if (chunked && chunk_bf16) BMsel = BNsel = 64;
    const uint64_t tiles_n = rn_ceil_div(Cout, BNsel);
    const uint64_t tiles_m = rn_ceil_div((uint64_t)p.M, BMsel);
    p.tiles_n = (int)tiles_n;
    const uint64_t total = tiles_m * tiles_n;
    RN_REQUIRE(ctx, fits_i32(total), "too many tiles");
    p.total_tiles = (unsigned)total;
    p.ksplit = 1;
    p.kchunk = p.nk;
    p.total_work = p.total_tiles;
    p.split_stride = 0;
    p.chunk_L = p.nk;
    p.full_tiles = p.total_tiles;
    p.tail_tiles = 0;
    p.ws = nullptr;
    p.ws_stride = 0;
    // Latency mode (rn_ctx_set_split_k): a launch whose tiles cannot fill the chip splits its
    // K loop over several blocks; each writes a raw fp32 partial tile to scratch and a second
    // kernel adds the partials in split order and applies the epilogue.  Deterministic, but
    // the summation order differs from the unsplit launch, so it is opt-in.
    if (ctx->split_k > 1 && total < 512 && p.nk >= 8 && Cout % 4 == 0 && !p.out_nchw &&
        !(second && dt_in == RN_DTYPE_BF16)) {
        int S = (int)rn_ceil_div(1024, total);
        if (S > ctx->split_k) S = ctx->split_k;
        if (S > p.nk / 4) S = p.nk / 4;  // at least four K tiles per block
        if (S > 1) {
            const int chunk = (int)rn_ceil_div((uint64_t)p.nk, (uint64_t)S);
            S = (int)rn_ceil_div((uint64_t)p.nk, (uint64_t)chunk);
            const uint64_t slice = (uint64_t)p.M * Cout;
            void *ws = nullptr;
            RN_TRY(rn_scratch(ctx, 4, (uint64_t)S * slice * sizeof(float), &ws));
            GemmParams q = p;
            q.out = ws;
            q.scale = q.shift = nullptr;
            q.residual = nullptr;
            q.relu = 0;
            q.ksplit = S;
            q.kchunk = chunk;
            q.total_work = p.total_tiles * (unsigned)S;
            q.split_stride = (long long)(slice * sizeof(float));
            q.out_bytes = (int)(slice * sizeof(float));
            q.grid_items = q.total_work;
            if (exact)
                launch_tiles<float, float, false, true>(ctx, q, BMsel, BNsel, persistent);
            else if (second)
                launch_tiles<float, float, true>(ctx, q, BMsel, BNsel, persistent);
            else if (dt_in == RN_DTYPE_F32)
                launch_tiles<float, float>(ctx, q, BMsel, BNsel, persistent);
            else
                launch_tiles<bf16_t, float>(ctx, q, BMsel, BNsel, persistent);
            RN_TRY(rn_after_launch(ctx, what));
            FinishParams f;
            f.partial = (const float *)ws;
            f.out = out;
            f.scale = p.scale;
            f.shift = p.shift;
            f.residual = p.residual;
            f.relu = p.relu;
            f.splits = S;
            f.Cout = (int)Cout;
            f.slice = slice;
            f.stride = slice;
            const unsigned fgrid = rn_stream_grid(slice / 4, 256);
            if (dt_out == RN_DTYPE_BF16)
                splitk_finish_kernel<bf16_t><<<fgrid, 256, 0, ctx->stream>>>(f);
            else
                splitk_finish_kernel<float><<<fgrid, 256, 0, ctx->stream>>>(f);
            return rn_after_launch(ctx, what);
        }
    }
    if (chunked) {
        // eight chunks (the last may be shorter; measured: as fast as four at B=256, 11 % less
        // B=1 latency), chunk length even for the two-tile loop trips
        p.chunk_L = 2 * (int)rn_ceil_div((uint64_t)p.nk, 16);  // (even: the fp32 loop takes two K tiles per trip)
        const int S = (int)rn_ceil_div((uint64_t)p.nk, (uint64_t)p.chunk_L);
        // tail = the tiles past the last full round of the 256 CUs, in whole rows of M tiles;
        // cutting them into S pieces pays when the pieces need fewer CU rounds than S
        // (a launch with at most one round of tiles is all tail: small batches get S times the
        // blocks, in the same summation order as any other batch size)
        unsigned tail = total > 256 ? (unsigned)(total % 256) : (unsigned)total;
        tail -= tail % (unsigned)tiles_n;
        // (NCHW output: the finishing kernel writes NHWC, so the tail tiles fold their chunks in
        // registers like the others -- the same sum)
        const bool cut = ctx->split_k <= 1 && tail > 0 && !p.out_nchw &&
                         rn_ceil_div((uint64_t)tail * S, 256) < (uint64_t)S;
        // the workspace holds only the tail rows [row0, M) of every chunk slice; the kernel
        // addresses it like the output, through a base moved back by row0 rows
        uint64_t row0 = 0, tail_elems = 0;
        void *ws_real = nullptr;
        if (cut) {
            row0 = (uint64_t)((p.total_tiles - tail) / (unsigned)tiles_n) * (uint64_t)BMsel;
            tail_elems = ((uint64_t)p.M - row0) * Cout;
            RN_TRY(rn_scratch(ctx, 4, (uint64_t)S * tail_elems * sizeof(float), &ws_real));
            // (integer arithmetic: the moved base lies before the allocation and is never dereferenced)
            p.ws = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(ws_real) - row0 * Cout * sizeof(float));
            p.ws_stride = (long long)(tail_elems * sizeof(float));
            p.tail_tiles = tail;
            p.full_tiles = p.total_tiles - tail;
            p.total_work = p.full_tiles + tail * (unsigned)S;
        }
        p.grid_items = p.total_work;
        choose_tile_order(ctx, p, p.full_tiles, BMsel, BNsel);
        launch_tiles_chunked(ctx, p, BMsel, BNsel, persistent, chunk_bf16);
        RN_TRY(rn_after_launch(ctx, what));
        if (cut) {
            FinishParams f;
            f.partial = (const float *)ws_real;
            f.out = (float *)out + row0 * Cout;
            f.scale = p.scale;
            f.shift = p.shift;
            f.residual = p.residual ? (const float *)p.residual + row0 * Cout : nullptr;
            f.relu = p.relu;
            f.splits = S;
            f.Cout = (int)Cout;
            f.slice = tail_elems;
            f.stride = tail_elems;
            splitk_finish_kernel<float><<<rn_stream_grid(f.slice / 4, 256), 256, 0, ctx->stream>>>(f);
            return rn_after_launch(ctx, what);
        }
        return RN_OK;
    }

    p.grid_items = p.total_tiles;
    choose_tile_order(ctx, p, p.total_tiles, BMsel, BNsel);
    if (exact)
        launch_tiles<float, float, false, true>(ctx, p, BMsel, BNsel, persistent);
    else if (second && dt_in == RN_DTYPE_F32 && dt_out == RN_DTYPE_F32)
        launch_tiles<float, float, true>(ctx, p, BMsel, BNsel, persistent);
    else if (second && dt_in == RN_DTYPE_BF16 && dt_out == RN_DTYPE_BF16)
        launch_tiles<bf16_t, bf16_t, true>(ctx, p, BMsel, BNsel, persistent);
    else if (second)
        return rn_set_error(ctx, RN_ERR_UNSUPPORTED, "%s: fused pair needs equal in/out dtype", what);
    else if (dt_in == RN_DTYPE_F32 && dt_out == RN_DTYPE_F32)
        launch_tiles<float, float>(ctx, p, BMsel, BNsel, persistent);
    else if (dt_in == RN_DTYPE_BF16 && dt_out == RN_DTYPE_BF16)
        launch_tiles<bf16_t, bf16_t>(ctx, p, BMsel, BNsel, persistent);
    else if (dt_in == RN_DTYPE_BF16 && dt_out == RN_DTYPE_F32)
        launch_tiles<bf16_t, float>(ctx, p, BMsel, BNsel, persistent);
    else
        return rn_set_error(ctx, RN_ERR_UNSUPPORTED, "%s: dtype combination %d -> %d", what, dt_in,
                            dt_out);
    return rn_after_launch(ctx, what);
}

int launch_direct(rn_ctx *ctx, const float *inp, float *out, const float *w, uint64_t k,
                  uint64_t stride, uint64_t pad, uint64_t h_out, uint64_t w_out, uint64_t B,
                  uint64_t Cin, uint64_t Cs, uint64_t Cout, uint64_t H, uint64_t W, int nhwc,
                  int w_packed, const rn_epilogue *ep, const char *what)
{
    DirectParams p;
    p.in = inp;
    p.w = w;
    p.out = out;
    p.scale = ep ? ep->scale : nullptr;
    p.shift = ep ? ep->shift : nullptr;
    p.residual = ep ? static_cast<const float *>(ep->residual) : nullptr;
    p.relu = ep ? ep->relu : 0;
    p.k = (int)k;
    p.stride = (int)stride;
    p.pad = (int)pad;
    p.Ho = (int)h_out;
    p.Wo = (int)w_out;
    p.Cin = (int)Cin;
    p.Cs = (int)Cs;
    p.Cout = (int)Cout;
    p.H = (int)H;
    p.W = (int)W;
    p.nhwc = nhwc;
    p.w_packed = w_packed;
    p.total = B * Cout * h_out * w_out;
    conv_direct_kernel<<<rn_stream_grid(p.total, 256), 256, 0, ctx->stream>>>(p);
    return rn_after_launch(ctx, what);
}

int check_conv_args(rn_ctx *ctx, const float *inp, const float *out, const float *weight,
                    uint64_t k, uint64_t stride, uint64_t pad, uint64_t h_out, uint64_t w_out,
                    uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W)
{
    RN_REQUIRE(ctx, inp && out && weight, "null tensor");
    RN_REQUIRE(ctx, inp != out, "conv2d cannot run in place");
    RN_REQUIRE(ctx, k >= 1 && stride >= 1, "kernel_size and stride must be >= 1");
    RN_REQUIRE(ctx, k < (1u << 12) && stride < (1u << 12) && pad < (1u << 12), "dimension too large");
    const uint64_t cs = Cin < 4 ? 4 : Cin;
    RN_REQUIRE(ctx, fits_i32(B * H * W * cs) && fits_i32(B * h_out * w_out * Cout) &&
                        fits_i32(Cout * k * k * cs + 64),
               "tensor has 2^31 or more elements");
    return RN_OK;
}

bool gemm_eligible(const void *inp, const void *out, const void *w, uint64_t Cin, uint64_t k,
                   uint64_t in_elems, uint64_t w_elems, uint64_t out_elems)
{
    // buffer descriptors carry byte counts below 2^31
    if (in_elems >= (1ull << 29) || w_elems >= (1ull << 29) || out_elems >= (1ull << 29))
        return false;
    const bool al = ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    return al && k <= 15 && (Cin % 32 == 0 || rn_conv_is_c4(Cin, k));
}

}  // namespace

extern "C" {

int rn_conv2d_nhwc_forward(rn_ctx *ctx, const float *inp, float *out, const float *packed_weight,
                           uint64_t kernel_size, uint64_t stride, uint64_t padding, uint64_t h_out,
                           uint64_t w_out, uint64_t B, uint64_t in_channels, uint64_t out_channels,
                           uint64_t H, uint64_t W, const rn_epilogue *epilogue)
{
    RN_ENTER(ctx);
    if (B * out_channels * h_out * w_out == 0) return RN_OK;
    RN_TRY(check_conv_args(ctx, inp, out, packed_weight, kernel_size, stride, padding, h_out, w_out,
                           B, in_channels, out_channels, H, W));
    RN_REQUIRE(ctx, in_channels >= 1, "in_channels must be >= 1");
    if (epilogue && epilogue->residual)
        RN_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(epilogue->residual) & 3) == 0,
                   "misaligned residual");
    if (gemm_eligible(inp, out, packed_weight, in_channels, kernel_size,
                      B * H * W * rn_conv2d_input_channels(in_channels),
                      rn_conv2d_packed_weight_numel(in_channels, out_channels, kernel_size),
                      B * h_out * w_out * out_channels)) {
        return launch_gemm(ctx, RN_DTYPE_F32, RN_DTYPE_F32, inp, out, packed_weight, kernel_size,
                           stride, padding, h_out, w_out, B, in_channels, out_channels, H, W,
                           epilogue, "rn_conv2d_nhwc_forward");
    }
    const bool c4 = rn_conv_is_c4(in_channels, kernel_size);
    return launch_direct(ctx, inp, out, packed_weight, kernel_size, stride, padding, h_out, w_out,
                         B, in_channels, rn_conv2d_input_channels(in_channels), out_channels, H, W,
                         1, c4 ? 2 : 1, epilogue, "rn_conv2d_nhwc_forward(direct)");
}

int rn_conv2d_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                      uint64_t kernel_size, uint64_t stride, uint64_t padding, uint64_t h_out,
                      uint64_t w_out, uint64_t B, uint64_t in_channels, uint64_t out_channels,
                      uint64_t H, uint64_t W)
{
    if (!ctx) return RN_ERR_INVALID;
    if (B * out_channels * h_out * w_out == 0) return RN_OK;
    RN_TRY(check_conv_args(ctx, inp, out, weight, kernel_size, stride, padding, h_out, w_out, B,
                           in_channels, out_channels, H, W));
    RN_REQUIRE(ctx, in_channels >= 1, "in_channels must be >= 1");
    if (RN_DEFERS(ctx))  // recorded; runs with the in-place batch-norm / add / ReLU behind it folded in (rn_defer.hip)
        return rn_defer_conv(ctx, inp, out, weight, kernel_size, stride, padding, h_out, w_out, B, in_channels,
                             out_channels, H, W);
    RN_ENTER(ctx);
    const bool c4 = rn_conv_is_c4(in_channels, kernel_size);
    const bool fast = (in_channels % 32 == 0 || c4) && kernel_size <= 15 &&
                      B * H * W * rn_conv2d_input_channels(in_channels) < (1ull << 29) &&
                      B * h_out * w_out * out_channels < (1ull << 29) &&
                      rn_conv2d_packed_weight_numel(in_channels, out_channels, kernel_size) <
                          (1ull << 29);
    if (!fast || (ctx->layout == RN_LAYOUT_NHWC && in_channels < 4)) {
        // exact reference order; OIHW weights as given
        return launch_direct(ctx, inp, out, weight, kernel_size, stride, padding, h_out, w_out, B,
                             in_channels, in_channels, out_channels, H, W,
                             ctx->layout == RN_LAYOUT_NHWC, 0, nullptr,
                             "rn_conv2d_forward(direct)");
    }
    // NCHW tensors: NCHW is the MFMA's own layout with the operands swapped (rn_conv_nchw.hip) -- no transpose of
    // the input, the same bits.  1x1: the OIHW weight as it is, no packing either
    const bool native = ctx->layout == RN_LAYOUT_NCHW && h_out == (H + 2 * padding - kernel_size) / stride + 1 &&
                        w_out == (W + 2 * padding - kernel_size) / stride + 1 &&
                        rn_conv_nchw_eligible(kernel_size, stride, padding, B, in_channels, out_channels, H, W);
    if (native && kernel_size == 1 && padding == 0 && (reinterpret_cast<uintptr_t>(weight) & 15) == 0)
        return rn_conv_nchw_launch(ctx, inp, out, weight, 1, stride, 0, B, in_channels, out_channels, H, W);
    // K-major panel of the OIHW weight: packed per call into scratch, or -- with the context's
    // weight cache on -- once per (weight buffer, shape) and kept until that buffer is freed or
    // written through the rn_* calls
    const uint64_t wn = rn_conv2d_packed_weight_numel(in_channels, out_channels, kernel_size);
    void *wp = ctx->wcache_on ? rn_wcache_find(ctx, weight, in_channels, out_channels, kernel_size) : nullptr;
    if (!wp) {
        if (ctx->wcache_on)
            RN_TRY(rn_wcache_add(ctx, weight, in_channels, out_channels, kernel_size, wn * sizeof(float), &wp));
        else
            RN_TRY(rn_scratch(ctx, 1, wn * sizeof(float), &wp));
        const int pst = rn_conv2d_pack_weight(ctx, weight, (float *)wp, in_channels, out_channels, kernel_size);
        if (pst != RN_OK) {
            // a cached entry whose panel was never filled must not answer the next lookup
            if (ctx->wcache_on) rn_wcache_remove(ctx, wp);
            return pst;
        }
    }
    if (ctx->layout == RN_LAYOUT_NHWC) {
        return launch_gemm(ctx, RN_DTYPE_F32, RN_DTYPE_F32, inp, out, wp, kernel_size, stride,
                           padding, h_out, w_out, B, in_channels, out_channels, H, W, nullptr,
                           "rn_conv2d_forward(nhwc)");
    }
    // k x k on NCHW tensors: the taps gathered from the channel planes, the packed panel as the MFMA rows.
    // Measured at B = 256 (tools/nchw_bench.py, RN_NCHW_TAPS=2 against 0): the gathering K loop is 15-25 % slower
    // per tile than the NHWC one, the transpose it saves costs 70 us on a 56x56 tensor and 10-25 us on the
    // 14x14 / 7x7 ones -- it pays on the large planes only (rn_ctx_set_nchw_taps: 1, the default)
    if (native && (ctx->nchw_taps >= 2 || (ctx->nchw_taps == 1 && H * W >= 2048)))
        return rn_conv_nchw_launch(ctx, inp, out, (const float *)wp, kernel_size, stride, padding, B, in_channels,
                                   out_channels, H, W);
    // anything else on NCHW tensors: transpose in, contract; the contraction's epilogue writes NCHW itself
    const uint64_t cs = rn_conv2d_input_channels(in_channels);
    void *xin = nullptr;
    RN_TRY(rn_scratch(ctx, 2, B * H * W * cs * sizeof(float), &xin));
    if (cs != in_channels) {
        RN_TRY(rn_nchw_to_nhwc_pad(ctx, inp, (float *)xin, B, in_channels, H, W, cs));
    } else {
        RN_TRY(rn_nchw_to_nhwc(ctx, inp, (float *)xin, B, in_channels, H, W));
    }
    return launch_gemm(ctx, RN_DTYPE_F32, RN_DTYPE_F32, xin, out, wp, kernel_size, stride, padding,
                       h_out, w_out, B, in_channels, out_channels, H, W, nullptr,
                       "rn_conv2d_forward(gemm)", nullptr, false, true);
}

int rn_conv2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, int out_dtype, const void *inp, void *out,
                              const void *packed_weight, uint64_t kernel_size, uint64_t stride,
                              uint64_t padding, uint64_t h_out, uint64_t w_out, uint64_t B,
                              uint64_t in_channels, uint64_t out_channels, uint64_t H, uint64_t W,
                              const rn_epilogue *epilogue)
{
    RN_ENTER(ctx);
    if (dtype == RN_DTYPE_F32) {
        RN_REQUIRE(ctx, out_dtype == RN_DTYPE_F32, "fp32 input implies fp32 output");
        return rn_conv2d_nhwc_forward(ctx, (const float *)inp, (float *)out,
                                      (const float *)packed_weight, kernel_size, stride, padding,
                                      h_out, w_out, B, in_channels, out_channels, H, W, epilogue);
    }
    RN_REQUIRE(ctx, dtype == RN_DTYPE_BF16, "unknown dtype");
    if (B * out_channels * h_out * w_out == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out && packed_weight && inp != out, "null or aliased tensor");
    RN_REQUIRE(ctx, kernel_size >= 1 && kernel_size <= 15 && stride >= 1 && stride < (1u << 12) &&
                        padding < (1u << 12),
               "kernel_size / stride / padding out of range");
    const bool c4 = rn_conv_is_c4(in_channels, kernel_size);
    const uint64_t cs = c4 ? 4 : in_channels;
    if (c4) {
        // two 4-channel pixels per 16-byte chunk: the image must carry its own zero border
        RN_REQUIRE(ctx, padding == 0 && stride % 2 == 0 && W % 2 == 0,
                   "bf16 small-Cin form needs a physically padded image, even stride and width");
    } else if (in_channels % 64 != 0) {
        return rn_set_error(ctx, RN_ERR_UNSUPPORTED,
                            "bf16 convolution needs in_channels %% 64 == 0 (got %llu)",
                            (unsigned long long)in_channels);
    }
    const uint64_t ktot = c4 ? rn_ceil_div(kernel_size, 2) * 64 : kernel_size * kernel_size * in_channels;
    RN_REQUIRE(ctx, B * H * W * cs < (1ull << 30) && out_channels * ktot < (1ull << 30) &&
                        B * h_out * w_out * out_channels < (1ull << 29),
               "tensor too large");
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(packed_weight)) & 15) == 0,
               "bf16 tensors must be 16-byte aligned");
    return launch_gemm(ctx, RN_DTYPE_BF16, out_dtype, inp, out, packed_weight, kernel_size, stride,
                       padding, h_out, w_out, B, in_channels, out_channels, H, W, epilogue,
                       "rn_conv2d_nhwc_forward_dt");
}

int rn_conv2d_nhwc_exact_forward(rn_ctx *ctx, const float *inp_padded, float *out,
                                 const float *packed_exact_weight, uint64_t kernel_size,
                                 uint64_t stride, uint64_t h_out, uint64_t w_out, uint64_t B,
                                 uint64_t in_channels, uint64_t out_channels, uint64_t Hp,
                                 uint64_t Wp, const rn_epilogue *epilogue)
{
    RN_ENTER(ctx);
    if (B * out_channels * h_out * w_out == 0) return RN_OK;
    RN_REQUIRE(ctx, inp_padded && out && packed_exact_weight && inp_padded != out,
               "null or aliased tensor");
    RN_REQUIRE(ctx, kernel_size >= 1 && kernel_size <= 15 && stride >= 1 && stride < (1u << 12) &&
                        in_channels >= 1 && in_channels <= 16,
               "kernel_size / stride / in_channels out of range");
    RN_REQUIRE(ctx, Hp >= kernel_size && Wp >= kernel_size &&
                        rn_conv_output_size(Hp, kernel_size, stride, 0) == h_out &&
                        rn_conv_output_size(Wp, kernel_size, stride, 0) == w_out,
               "h_out / w_out do not match the padded image");
    const uint64_t ktot = rn_ceil_div(kernel_size * kernel_size * in_channels, 32) * 32;
    RN_REQUIRE(ctx, B * Hp * Wp * in_channels < (1ull << 29) && out_channels * ktot < (1ull << 29) &&
                        B * h_out * w_out * out_channels < (1ull << 29),
               "tensor too large");
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(packed_exact_weight)) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(inp_padded) & 3) == 0,
               "misaligned tensor");
    if (epilogue && epilogue->residual)
        RN_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(epilogue->residual) & 15) == 0,
                   "misaligned residual");
    return launch_gemm(ctx, RN_DTYPE_F32, RN_DTYPE_F32, inp_padded, out, packed_exact_weight,
                       kernel_size, stride, 0, h_out, w_out, B, in_channels, out_channels, Hp, Wp,
                       epilogue, "rn_conv2d_nhwc_exact_forward", nullptr, true);
}

int rn_conv2d_nhwc_pair_forward_dt(rn_ctx *ctx, int dtype, int out_dtype, const void *inp, void *out,
                                   const void *packed_pair_weight, uint64_t kernel_size,
                                   uint64_t stride, uint64_t padding, uint64_t h_out,
                                   uint64_t w_out, uint64_t B, uint64_t in_channels,
                                   uint64_t out_channels, uint64_t H, uint64_t W,
                                   const rn_conv_second *second, const rn_epilogue *epilogue)
{
    RN_ENTER(ctx);
    RN_REQUIRE(ctx, second && second->inp, "second source missing");
    RN_REQUIRE(ctx, dtype == RN_DTYPE_F32 || dtype == RN_DTYPE_BF16, "unknown dtype");
    RN_REQUIRE(ctx, out_dtype == dtype, "the fused pair keeps one element type");
    if (B * out_channels * h_out * w_out == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out && packed_pair_weight && inp != out && second->inp != out,
               "null or aliased tensor");
    RN_REQUIRE(ctx, kernel_size >= 1 && kernel_size <= 15 && stride >= 1 && stride < (1u << 12) &&
                        padding < (1u << 12) && second->stride >= 1 && second->stride < (1u << 12),
               "kernel_size / stride / padding out of range");
    const uint64_t seg = dtype == RN_DTYPE_BF16 ? 64 : 32;  // elements per 128-byte K tile
    if (in_channels % seg != 0 || second->in_channels % seg != 0 || in_channels == 0 ||
        second->in_channels == 0)
        return rn_set_error(ctx, RN_ERR_UNSUPPORTED,
                            "fused pair needs both channel counts to be multiples of %llu",
                            (unsigned long long)seg);
    RN_REQUIRE(ctx, rn_conv_output_size(H, kernel_size, stride, padding) == h_out &&
                        rn_conv_output_size(W, kernel_size, stride, padding) == w_out,
               "h_out / w_out do not match the first convolution");
    RN_REQUIRE(ctx, rn_conv_output_size(second->H, 1, second->stride, 0) == h_out &&
                        rn_conv_output_size(second->W, 1, second->stride, 0) == w_out,
               "the second (1x1) convolution has a different output size");
    const uint64_t ktot = kernel_size * kernel_size * in_channels + second->in_channels;
    RN_REQUIRE(ctx, B * H * W * in_channels < (1ull << 29) &&
                        B * second->H * second->W * second->in_channels < (1ull << 29) &&
                        out_channels * ktot < (1ull << 29) &&
                        B * h_out * w_out * out_channels < (1ull << 29),
               "tensor too large");
    RN_REQUIRE(ctx, ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out) |
                      reinterpret_cast<uintptr_t>(packed_pair_weight) |
                      reinterpret_cast<uintptr_t>(second->inp)) & 15) == 0,
               "tensors must be 16-byte aligned");
    if (epilogue && epilogue->residual)
        RN_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(epilogue->residual) & 15) == 0,
                   "misaligned residual");
    return launch_gemm(ctx, dtype, out_dtype, inp, out, packed_pair_weight, kernel_size, stride,
                       padding, h_out, w_out, B, in_channels, out_channels, H, W, epilogue,
                       "rn_conv2d_nhwc_pair_forward_dt", second);
}

int rn_linear_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                      const float *bias, uint64_t B, uint64_t in_features, uint64_t out_features)
{
    if (!ctx) return RN_ERR_INVALID;
    if (B * out_features == 0) return RN_OK;
    RN_REQUIRE(ctx, inp && out && weight, "null tensor");
    RN_REQUIRE(ctx, inp != out, "linear cannot run in place");
    RN_REQUIRE(ctx, in_features >= 1, "in_features must be >= 1");
    RN_REQUIRE(ctx, fits_i32(B * in_features) && fits_i32(B * out_features) &&
                        fits_i32(out_features * in_features + 64),
               "tensor has 2^31 or more elements");
    if (RN_DEFERS(ctx)) return rn_defer_linear(ctx, inp, out, weight, bias, B, in_features, out_features);
    RN_ENTER(ctx);
    rn_epilogue ep = {nullptr, bias, nullptr, 0};
    // W is [out][in] row-major == the K-major panel of a 1x1 convolution on a 1x1 image
    if (in_features % 32 == 0 &&
        gemm_eligible(inp, out, weight, in_features, 1, B * in_features, out_features * in_features,
                      B * out_features)) {
        return launch_gemm(ctx, RN_DTYPE_F32, RN_DTYPE_F32, inp, out, weight, 1, 1, 0, 1, 1, B,
                           in_features, out_features, 1, 1, &ep, "rn_linear_forward");
    }
    return launch_direct(ctx, inp, out, weight, 1, 1, 0, 1, 1, B, in_features, in_features,
                         out_features, 1, 1, 1, 0, &ep, "rn_linear_forward(direct)");
}

}  // extern "C"
