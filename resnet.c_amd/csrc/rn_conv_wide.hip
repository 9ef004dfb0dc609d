// bf16 contraction on 256-wide block tiles: the K-heavy convolutions of the bf16 network.
//
// Same implicit GEMM as conv_gemm_kernel (rn_conv.hip; reference: conv2dForwardKernel,
// cuda/ops.cu:14-48, on bf16-rounded operands with fp32 sums), same MFMA
// (v_mfma_f32_32x32x16_bf16), same operand map and the same k order per output element, so
// every tile candidate of a layer produces the same bits.  What differs is how operands reach
// the matrix cores.  With 64..128-wide tiles a CU has to pull 64 B/clk of operands through its
// vector-memory path and push them through ds_write_b128 (79 B/clk) -- the 4-wave kernel stalls
// there at 600-700 TFLOP/s.  Here:
//
//  * block tile 256x256 (also 256x128, 128x256, 256x64): 512 threads = 8 waves, each wave
//    128x64 (64x64 / 32x64) of 32x32 MFMA tiles; one block per CU, two waves per SIMD.  A
//    256x256x64 K step is 2048 MFMA cycles per SIMD for 64 KB of operands: 32 B/clk/CU;
//  * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers,
//    no ds_write.  One wave instruction moves 8 rows x 128 B; the LDS image is lane-linear, so
//    the (row>>1)&7 chunk swizzle that makes the ds_read_b128 fragment reads conflict-free is
//    applied to the per-lane SOURCE offset.  The descriptor's range check writes zeros for a
//    padded tap or a row past M / Cout, exactly like the register path;
//  * rings in LDS: three K tiles of A (two in flight while one is multiplied) and two or three
//    of B; all 160 KB of the CU for the 256x256 tile.  One barrier per K step, with a counted
//    s_waitcnt vmcnt(N) in front of it: the wave's own pieces of the tile about to be read
//    have landed, the younger tile stays in flight across the barrier.  The DMA pieces of a
//    step are issued between its MFMA groups, not in a block;
//  * the DMA is issued from inline asm (hipcc would otherwise drain vmcnt(0) in front of every
//    LDS read); everything else -- fragment reads, MFMAs, the epilogue -- is plain HIP;
//  * epilogue as in the 4-wave kernel: accumulators -> LDS (fp32, in one or two passes of up to
//    128 KB) -> row-contiguous 16-byte stores with the channel affine, residual and ReLU.
//
// Roofline: MFMA for K >= ~512 and N >= 128 (the 3x3 convolutions, conv1 of layer3/4, the fused
// conv3 + downsample pairs); short-K layers stay on the 4-wave kernel (rn_model_tune decides).
//
// Second kernel of this file (same DMA helpers): conv_strip_kernel for the 3x3 / 64 -> 64 layers,
// described where it starts.
#include <type_traits>

#include "rn_conv_params.h"
#include "rn_lds_dma.h"

using namespace rn_gemm;

namespace {

using namespace rn_dma;

template <typename TO>
struct Out;
template <>
struct Out<bf16_t> {
    static constexpr int EPT = 8;
    static __device__ __forceinline__ void unpack(const i32x4 &r, float (&v)[8])
    {
        const bf16x8 x = __builtin_bit_cast(bf16x8, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    }
    static __device__ __forceinline__ i32x4 pack(const float (&v)[8])
    {
        bf16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (bf16_t)v[j];  // round to nearest even
        return __builtin_bit_cast(i32x4, x);
    }
};

// BM x BN block tile, WM x WN waves (WM * WN == 8), DUAL as in conv_gemm_kernel.
// LDS_CAP: the block's LDS budget.  All of the CU's 160 KB for the 256-wide tiles (one block per CU);
// 80 KB for the 128x128 tile, so that TWO blocks share a CU (four waves per SIMD, 128 registers each)
// and one's prologue and epilogue run beside the other's K loop -- the short-K and memory-paced 1x1
// layers of stage 3-4, where a lone block's fill and drain are a third of its life.
template <typename TO, int BM, int BN, int WM, int WN, bool DUAL, int LDS_CAP = 160 * 1024>
__global__ __launch_bounds__(512, LDS_CAP <= 80 * 1024 ? 4 : 2) void conv_wide_kernel(const GemmParams p)
{
    static_assert(WM * WN == 8, "eight waves");
    constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
    constexpr int MI = TM / 32, NI = TN / 32;
    // BM need not be a multiple of 64 (224 = 7 fragments: 50,176 rows are 224 tiles, one round of
    // 256 CUs at 7/8 of the 256-row tile's time); the A slot is, its last rows are never fetched
    constexpr int AM = (BM + 63) / 64 * 64;
    constexpr int PA = AM / 64, PB = BN / 64;  // 1-KiB DMA pieces per wave and K tile
    static_assert(PA >= 1 && PB >= 1 && MI >= 1 && NI >= 1, "tile too small for eight waves");
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile in whole fragments");
    constexpr int A_SLOT = AM * 128, B_SLOT = BN * 128;  // bytes of one K tile of A / of B
    constexpr int SA = 3;
    constexpr int SB = 3 * (A_SLOT + B_SLOT) <= LDS_CAP ? 3 : 2;
    constexpr int LDS_BYTES = SA * A_SLOT + SB * B_SLOT;
    static_assert(LDS_BYTES <= LDS_CAP, "LDS");
    // epilogue staging: the fp32 tile in one pass, or its two halves of BM/2 rows
    constexpr int CPASS = AM * BN * 4 <= LDS_BYTES ? 1 : 2;
    constexpr int EROWS = AM / CPASS;
    static_assert(EROWS * BN * 4 <= LDS_BYTES && EROWS % 32 == 0, "epilogue staging");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    stamp(p.stamps, 0);
    // XCD-aware tile order, as conv_gemm_kernel: each XCD walks a contiguous range of tiles with
    // the N tiles of one M panel adjacent
    int m0, n0;
    {
        const unsigned total = p.total_tiles, v = blockIdx.x;
        const unsigned q = total >> 3, r = total & 7, xcd = v & 7;
        const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        n0 = (int)(logical % (unsigned)p.tiles_n) * BN;
        m0 = (int)(logical / (unsigned)p.tiles_n) * BM;
    }
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const i32x4 srd_a = make_srd(p.in, p.in_bytes);
    const i32x4 srd_b = make_srd(p.w, p.w_bytes);
    const i32x4 srd_a2 = make_srd(DUAL ? p.in2 : p.in, DUAL ? p.in2_bytes : 0);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // DMA geometry: piece q = 8*j + wave of an operand tile covers tile rows 8q .. 8q+7; lane l
    // fills (row 8q + l/8, physical chunk l%8) and therefore fetches logical chunk
    // (l%8) ^ ((row>>1)&7).  Per piece and lane: byte offset of tap (0,0) of the row, and which
    // taps are inside the image (bit kh of the low half, bit kw of the high half).
    const int prow = lane >> 3, pc = lane & 7;
    int a_off[PA], a_mask[PA], a_cur[PA], b_off[PB];
    int a_off2[DUAL ? PA : 1];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int r = 8 * (8 * j + wave) + prow;
        const int chunk = (pc ^ ((r >> 1) & 7)) * 16;
        const int m = m0 + r;
        if (r < BM && m < p.M) {
            const int b = p.HoWo == 1 ? m : (int)(__umulhi((unsigned)m, p.mul_hw) >> p.shr_hw);
            const int rem = m - b * p.HoWo;
            const int oh = p.Wo == 1 ? rem : (int)(__umulhi((unsigned)rem, p.mul_w) >> p.shr_w);
            const int ow = rem - oh * p.Wo;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            a_off[j] = ((b * p.H + ih0) * p.W + iw0) * p.Cs * 2 + chunk;
            const int rlo = max(0, -ih0), rhi = min(p.KH, p.H - ih0);
            const int clo = max(0, -iw0), chi = min(p.KW, p.W - iw0);
            const int rm = rhi > rlo ? ((1 << rhi) - 1) & ~((1 << rlo) - 1) : 0;
            const int cm = chi > clo ? ((1 << chi) - 1) & ~((1 << clo) - 1) : 0;
            a_mask[j] = rm | (cm << 16);
            if constexpr (DUAL)
                a_off2[j] = ((b * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.Cs2 * 2 + chunk;
        } else {
            a_off[j] = 0;
            a_mask[j] = 0;
            if constexpr (DUAL) a_off2[j] = kOob;
        }
        a_cur[j] = kOob;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int r = 8 * (8 * j + wave) + prow;
        const int n = n0 + r;
        b_off[j] = n < p.Cout ? n * p.Ktot * 2 + (pc ^ ((r >> 1) & 7)) * 16 : kOob;
    }

    // K tile kt of A into ring slot `slot`.  The K position (tap, 128-byte segment) is scalar
    // arithmetic; the per-row offsets change only when the tap does, the segments of one tap go
    // through the scalar offset.  Tiles are issued in increasing kt from 0, so a tap's first
    // segment always passes here before its others.
    // prep: the K position of tile kt (tap offsets when a tap begins, descriptor, segment)
    i32x4 cur_srd = srd_a;
    int cur_seg = 0, cur_soff_b = 0;
    unsigned dst_a = 0, dst_b = 0;
    auto prep_a = [&](int kt, int slot) {
        const unsigned s_kt = (unsigned)__builtin_amdgcn_readfirstlane(kt);
        int s_cs;
        cur_srd = srd_a;
        if (DUAL && s_kt >= (unsigned)p.nk1) {
            s_cs = (int)s_kt - p.nk1;
            cur_srd = srd_a2;
            if (s_cs == 0) {
#pragma unroll
                for (int j = 0; j < PA; ++j) a_cur[j] = a_off2[DUAL ? j : 0];
            }
        } else {
            const unsigned tap = p.cseg == 1 ? s_kt : (__umulhi(s_kt, p.mul_cs) >> p.shr_cs);
            s_cs = (int)(s_kt - tap * (unsigned)p.cseg);
            if (s_cs == 0) {
                const unsigned ukh = p.KW == 1 ? tap : (__umulhi(tap, p.mul_kw) >> p.shr_kw);
                const int s_kh = (int)ukh, s_kw = (int)(tap - ukh * (unsigned)p.KW);
                const int toff = (s_kh * p.W + s_kw) * p.Cs * 2;  // tap_rows == 1 (eligibility)
#pragma unroll
                for (int j = 0; j < PA; ++j) {
                    const bool ok = ((a_mask[j] >> s_kh) & (a_mask[j] >> (16 + s_kw)) & 1) != 0;
                    a_cur[j] = ok ? a_off[j] + toff : kOob;
                }
            }
        }
        cur_seg = s_cs * 128;
        dst_a = lds_base + (unsigned)(slot * A_SLOT + wave * 1024);
    };
    auto prep_b = [&](int kt, int slot) {
        cur_soff_b = __builtin_amdgcn_readfirstlane(kt) * 128;
        dst_b = lds_base + (unsigned)(SA * A_SLOT + slot * B_SLOT + wave * 1024);
    };
    // A tile past the end of the K loop is still "issued", with out-of-range offsets: zeros land
    // in a slot nobody reads any more, no global traffic, and the loop body stays one straight
    // line with one constant vmcnt (a second, DMA-free copy of the step made hipcc keep the
    // accumulators in two register sets: +128 VGPRs and spills).
    bool live_a = true, live_b = true;
    auto dma_a = [&](int j) {
        dma16(live_a ? a_cur[j] : kOob, cur_srd, cur_seg, dst_a + (unsigned)(j * 8192));
    };
    auto dma_b = [&](int j) {
        dma16(live_b ? b_off[j] : kOob, srd_b, cur_soff_b, dst_b + (unsigned)(j * 8192));
    };
    const int nk = p.nk;
    auto issue_a = [&](int kt, int slot) {
        live_a = kt < nk;
        prep_a(live_a ? kt : nk - 1, slot);
#pragma unroll
        for (int j = 0; j < PA; ++j) dma_a(j);
    };
    auto issue_b = [&](int kt, int slot) {
        live_b = kt < nk;
        prep_b(kt, slot);
#pragma unroll
        for (int j = 0; j < PB; ++j) dma_b(j);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // fragment reads: lane (li, lh) reads chunk 2*ks+lh of its rows -> k = 16ks + 8lh + {0..7},
    // the operand map of the 32x32x16 MFMA; (row>>1)&7 of rows base+32i+li is that of li
    const int sw = (li >> 1) & 7;
    int frag_co[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) frag_co[ks] = ((2 * ks + lh) ^ sw) * 16;
    const int a_thr = (wm * TM + li) * 128, b_thr = SA * A_SLOT + (wn * TN + li) * 128;
    // One K step: fragment reads and MFMAs of tile (sa, sb), and between the MFMA groups of its
    // four k-steps the step's DMA pieces -- piece(i) issues the i-th.  Issued as one block after
    // the barrier, the pieces kept the matrix pipe idle (a wave issues in order, and its SIMD
    // partner does the same thing at the same time): spread out, 6-34 % faster per layer.
    // The fragment reads are left to hipcc's own interleave: an explicit one-k-step-ahead
    // prefetch (second fragment set, sched_group_barrier slots) measured 2-9 % slower, and
    // giving the two waves of a SIMD different k-steps to issue their pieces in 2-12 % slower.
    auto compute = [&](int sa, int sb, auto npieces, auto piece) {
        constexpr int NP = decltype(npieces)::value;
        const char *const abase = lds + a_thr + sa * A_SLOT;
        const char *const bbase = lds + b_thr + sb * B_SLOT;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            i32x4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = *reinterpret_cast<const i32x4 *>(abase + frag_co[ks] + mi * 4096);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = *reinterpret_cast<const i32x4 *>(bbase + frag_co[ks] + ni * 4096);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[mi]), __builtin_bit_cast(bf16x8, b[ni]), acc[mi][ni],
                        0, 0, 0);
#pragma unroll
            for (int i = ks * NP / 4; i < (ks + 1) * NP / 4; ++i) piece(i);
        }
    };
    using AllPieces = std::integral_constant<int, PA + PB>;

    constexpr int EPT = Out<TO>::EPT;
    constexpr int CV = BN / EPT, RPP = 512 / CV, STEPS = EROWS / RPP;
    static_assert(STEPS >= 1 && EROWS % RPP == 0, "epilogue geometry");
    const int cv = t % CV, rr = t / CV;
    const int n = n0 + cv * EPT;
    const bool col_ok = n < p.Cout;  // Cout % 8 == 0 (eligibility): then n + 8 <= Cout
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const bool has_res = p.residual != nullptr;
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
        has_res ? const_cast<void *>(p.residual) : p.out, 0, has_res ? p.out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sc = __builtin_amdgcn_make_buffer_rsrc(
        has_scale ? (void *)const_cast<float *>(p.scale) : p.out, 0, has_scale ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_sh = __builtin_amdgcn_make_buffer_rsrc(
        has_shift ? (void *)const_cast<float *>(p.shift) : p.out, 0, has_shift ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    // residual rows of a pass.  (Asking for the first pass's earlier -- during the last K tile, or
    // before the K loop -- was measured on the K = 256 layers: the wait only moves, into the K
    // loop's closing vmcnt(0) or into the first operand tiles' (+2.5 us there); those launches
    // run at the memory system's pace, 3.9 TB/s.)
    i32x4 resv[STEPS];
    auto load_res = [&](int row0) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int m = m0 + row0 + rr + s * RPP;
            const int off = (col_ok && m < p.M && row0 + rr + s * RPP < BM) ? (m * p.Cout + n) * (int)sizeof(TO) : kOob;
            resv[s] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, off, 0, 0));
        }
    };

    // ---- K loop ----------------------------------------------------------------------------
    // issue order is what the counted waits rely on:
    //   SB == 3:  prologue A0 B0 A1 B1; step t issues A(t+2) B(t+2);  wait leaves A(t+1) B(t+1)
    //   SB == 2:  prologue A0 B0 A1;    step t issues B(t+1) A(t+2);  wait leaves A(t+1)
    issue_a(0, 0);
    issue_b(0, 0);
    issue_a(1, 1);
    if constexpr (SB == 3) issue_b(1, 1);
    int sa = 0, sb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        wait_and_barrier<SB == 3 ? PA + PB : PA>();
        if (kt == 0) {
            stamp(p.stamps, 1);
            stamp_cycles(p.stamps, 8);
        }
        // every wave is past its reads of tile kt-1: its slots take the tiles after next
        const int sa2 = sa == 0 ? 2 : sa - 1;
        if constexpr (SB == 3) {  // pieces in issue order: A(kt+2), then B(kt+2)
            live_a = live_b = kt + 2 < nk;
            prep_a(live_a ? kt + 2 : nk - 1, sa2);
            prep_b(kt + 2, sb == 0 ? 2 : sb - 1);
            compute(sa, sb, AllPieces{}, [&](int i) {
                if (i < PA)
                    dma_a(i);
                else
                    dma_b(i - PA);
            });
        } else {  // pieces in issue order: B(kt+1), then A(kt+2)
            live_b = kt + 1 < nk;
            live_a = kt + 2 < nk;
            prep_b(kt + 1, sb ^ 1);
            prep_a(live_a ? kt + 2 : nk - 1, sa2);
            compute(sa, sb, AllPieces{}, [&](int i) {
                if (i < PB)
                    dma_b(i);
                else
                    dma_a(i - PB);
            });
        }
        sa = sa == SA - 1 ? 0 : sa + 1;
        sb = sb == SB - 1 ? 0 : sb + 1;
    }
    wait_and_barrier<0>();  // every wave is done with the rings: they become the C staging area
    stamp_cycles(p.stamps, 9);
    stamp(p.stamps, 2);

    // ---- epilogue ----------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5); one
    // ds_write_b32 per register puts 32 consecutive columns of two rows, conflict-free.  Then
    // 16 bytes of one output row per thread and step: channel affine, residual, ReLU, store.
    float sc[EPT], sh[EPT];
#pragma unroll
    for (int j4 = 0; j4 < EPT / 4; ++j4) {
        const i32x4 s4 = __builtin_bit_cast(
            i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc, col_ok ? (n + 4 * j4) * 4 : kOob, 0, 0));
        const i32x4 h4 = __builtin_bit_cast(
            i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh, col_ok ? (n + 4 * j4) * 4 : kOob, 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[4 * j4 + j] = has_scale ? __int_as_float(s4[j]) : 1.f;
            sh[4 * j4 + j] = has_shift ? __int_as_float(h4[j]) : -0.f;  // -0.0 keeps a -0.0 sum
        }
    }
    float *const Cs = reinterpret_cast<float *>(lds);  // [EROWS][BN] fp32
    // The read-back half, specialised on the two launch-uniform switches: without a residual there
    // are no residual loads, unpacks, adds and selects, and a ReLU that is known needs no select --
    // 20 instead of ~45 vector instructions per 8 outputs (two thirds of the network's launches
    // have no residual, all have the ReLU).
    auto read_back = [&](int row0, auto with_res, auto with_relu) {
        constexpr bool RES = decltype(with_res)::value, RELU = decltype(with_relu)::value;
        if constexpr (RES) load_res(row0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int fr = wm * TM + mi * 32 - row0;  // fragment's first row in this pass: wave-uniform
            if (fr < 0 || fr >= EROWS) continue;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float *dst = Cs + (fr + 4 * lh) * BN + wn * TN + ni * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[mi][ni][e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int row = rr + s * RPP;
            const int m = m0 + row0 + row;
            float v[EPT], res[EPT];
            if constexpr (RES) Out<TO>::unpack(resv[s], res);
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
            const int off = (col_ok && m < p.M && row0 + row < BM) ? (m * p.Cout + n) * (int)sizeof(TO) : kOob;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, Out<TO>::pack(v)), rsrc_o, off, 0, 0);
        }
    };
#pragma unroll
    for (int pass = 0; pass < CPASS; ++pass) {
        const int row0 = pass * EROWS;  // first tile row of this pass
        if (has_res) {
            if (p.relu) read_back(row0, std::true_type{}, std::true_type{});
            else read_back(row0, std::true_type{}, std::false_type{});
        } else {
            if (p.relu) read_back(row0, std::false_type{}, std::true_type{});
            else read_back(row0, std::false_type{}, std::false_type{});
        }
        if (pass + 1 < CPASS) __syncthreads();
    }
    if (p.stamps) {
        stamp(p.stamps, 3);  // epilogue stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(p.stamps, 4);  // ... and acknowledged
    }
}

struct WideTile {
    int bm, bn;
};
constexpr WideTile kTiles[] = {{256, 256}, {256, 128}, {128, 256}, {256, 64}, {224, 256}, {128, 128}};
constexpr int kNumTiles = (int)(sizeof(kTiles) / sizeof(kTiles[0]));

template <int BM, int BN, int WM, int WN, int LDS_CAP = 160 * 1024>
void launch(rn_ctx *ctx, const GemmParams &p, bool dual)
{
    if (dual)
        conv_wide_kernel<bf16_t, BM, BN, WM, WN, true, LDS_CAP><<<dim3(p.total_tiles), dim3(512), 0, ctx->stream>>>(p);
    else
        conv_wide_kernel<bf16_t, BM, BN, WM, WN, false, LDS_CAP><<<dim3(p.total_tiles), dim3(512), 0, ctx->stream>>>(p);
}


// ---- 3x3 / stride 1 / pad 1, 64 -> 64 channels (conv2 of the first stage): the strip kernel ----
//
// With N = 64 the tile kernels above spend their time on per-K-tile overhead: a 256x64 tile has
// 8 MFMAs per wave between two barriers, and the nine taps re-stage the same input rows nine
// times (measured: removing eight of the nine A fetches changes nothing -- the loop, not L2, is
// the limit).  This kernel has no barrier inside a tile and stages every input row once:
//
//  * the 9 x 64 x 64 weights live in REGISTERS: wave (nf, mq) owns output channels
//    32nf .. 32nf+31 and keeps their 36 k-steps as MFMA operands (144 VGPRs), loaded once;
//  * a block walks a contiguous range of the flattened, zero-PADDED image (pitch W+2, one zero
//    row between images: position u = (b(H+1) + oh + 1)(W+2) + ow + 1), 256 positions per step.
//    In that space tap (kh, kw) is the constant shift (kh-1)(W+2) + (kw-1), and every border
//    tap reads a zero that is really there -- no masks in the loop; the 5 % of positions that
//    are padding are multiplied and not stored;
//  * the input lives in a rolling ring of 640 positions x 128 B in LDS, filled by LDS-DMA
//    (zeros for padding positions through the descriptor's range check): each step first asks for the
//    next 256 positions, then multiplies the current ones, every input row enters LDS once
//    per block.  One barrier per step (= per 72 MFMAs of a wave);
//  * operands swapped: weights are the MFMA's rows, pixels its columns, so a lane ends up with
//    16 channels of ONE pixel; v_permlane32_swap pairs the two half-waves' 4-channel groups
//    into 16-byte runs and the lane stores them itself -- no LDS staging of the output.  The
//    stores of a step are issued after the next step's barrier, so its s_waitcnt vmcnt(0)
//    waits on nothing fresh.
//
// Same k order per output element as the tile kernels (tap-major, 16 channels per MFMA), same
// products (a*b commutes), so the bits are the same as every other candidate's.
struct StripParams {
    const void *in, *w;
    void *out;
    const float *scale, *shift;
    int relu;
    int B, H, W;
    int Wp, Hq;  // W + 2, H + 1
    unsigned mul_wp, shr_wp, mul_hq, shr_hq;
    int U;       // padded positions: (B * Hq + 1) * Wp
    int nsteps;  // ceil(U / 256)
    int in_bytes, out_bytes;
    unsigned long long *stamps;  // diagnostic only (tools/conv_stamps.py --raw)
};

constexpr int kRing = 640;       // positions in the LDS ring (5 x 128: 256-position steps wrap every 5)
constexpr int kStripMargin = 64;  // ring position of a block's first output position; >= W + 3

__global__ __launch_bounds__(512, 2) void conv_strip_kernel(const StripParams p)
{
    // ring | scale[64], shift[64] | the weights as they lie in memory (start-up only)
    __shared__ __attribute__((aligned(16))) char lds[kRing * 128 + 512 + 73 * 1024];
    stamp(p.stamps, 0);
    float *const ssl = reinterpret_cast<float *>(lds + kRing * 128);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nf = wave >> 2, mq = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int prow = lane >> 3, pc = lane & 7;

    // this block's steps: XCD-aware order (neighbouring ranges, which share their halo rows,
    // on one XCD's L2), remainder steps to the first blocks
    int nst, ub;
    {
        const unsigned total = gridDim.x, v = blockIdx.x;
        const unsigned q = total >> 3, r = total & 7, xcd = v & 7;
        const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        const unsigned base = (unsigned)p.nsteps / total, rem = (unsigned)p.nsteps % total;
        nst = (int)(base + (logical < rem ? 1u : 0u));
        ub = (int)(logical * base + min(logical, rem)) * 256;
    }

    if (t < 64) {
        ssl[t] = p.scale ? p.scale[t] : 1.f;
        ssl[64 + t] = p.shift ? p.shift[t] : -0.f;  // -0.0 keeps a -0.0 sum
    }

    const i32x4 srd_in = make_srd(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);

    // Weights: 72 KB, the same for every block.  Fetched coalesced (1-KiB DMA pieces) and picked
    // out of LDS -- a lane's 36 fragments are 16 bytes of 36 places in one 1,152-byte row, and 32
    // rows per wave instruction straight from global cost the start of every block 8 us of
    // address-coalescer time.  The LDS rows are 73 chunks long, not 72: with 72 the rows of a
    // ds_read_b128 lane group start on two bank offsets only (8-way conflict, PMC: 19 % of the
    // kernel's LDS cycles); 9r mod 16 is different for all 16 rows of a group.
    {
        const i32x4 srd_w = make_srd(p.w, 64 * 576 * 2);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (8 * j + wave >= 73) break;               // 64 rows x 73 chunks = 73 pieces (wave-uniform)
            const int pos = (8 * j + wave) * 64 + lane;  // 16-byte chunk of the LDS image
            const int r = (pos * 7183) >> 19;            // pos / 73 for pos < 4,736
            const int c = pos - r * 73;
            dma16(c < 72 ? (r * 72 + c) * 16 : kOob, srd_w, 0,
                  lds_base + (unsigned)(kRing * 128 + 512 + (8 * j + wave) * 1024));
        }
    }

    // padded position u -> pixel index of the NHWC tensor, or -1 for a padding position
    auto pixel_of = [&](int u) -> int {
        if (u < 0 || u >= p.U) return -1;
        const unsigned R = __umulhi((unsigned)u, p.mul_wp) >> p.shr_wp;
        const int cc = u - (int)R * p.Wp;
        const unsigned b = __umulhi(R, p.mul_hq) >> p.shr_hq;
        const int rr = (int)R - (int)b * p.Hq;
        if (cc < 1 || cc > p.W || rr < 1) return -1;
        return ((int)b * p.H + rr - 1) * p.W + cc - 1;
    };
    // source offset of this lane's 16 bytes of a DMA piece: ring position `slot` holds padded
    // position u, physical chunk pc holds logical chunk pc ^ ((slot>>1)&7)
    auto src_off = [&](int u, int slot) -> int {
        const int g = pixel_of(u);
        return g < 0 ? kOob : g * 128 + ((pc ^ ((slot >> 1) & 7)) << 4);
    };

    // ring positions 0 .. 383 = padded positions ub - 64 .. ub + 319: all of step 0's
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int slot = 8 * (8 * j + wave) + prow;
        dma16(src_off(ub - kStripMargin + slot, slot), srd_in, 0, lds_base + (unsigned)((8 * j + wave) * 1024));
    }

    i32x4 wreg[36];
    {
        wait_and_barrier<0>();
        const char *wrow = lds + kRing * 128 + 512 + (32 * nf + li) * (73 * 16) + lh * 16;
#pragma unroll
        for (int s = 0; s < 36; ++s) wreg[s] = *reinterpret_cast<const i32x4 *>(wrow + s * 32);
    }

    i32x4 pend[2][2];
    int pend_off[2] = {kOob, kOob};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) pend[i][h] = i32x4{0, 0, 0, 0};
    // the four 16-byte stores of a step's results: store q = 2i + h
    auto store_piece = [&](int q) {
        const int i = q >> 1, h = q & 1;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[i][h]), rsrc_o,
                                               pend_off[i] == kOob ? kOob : pend_off[i] + 32 * h, 0, 0);
    };

    // diagnostic stamps of step 4 (top, own memory landed, barrier passed, taps done, results packed)
    // and of step 5's end, kept in registers and written when the block is done: a stamp stored
    // inside the loop would be waited for by the next s_waitcnt vmcnt(0)
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    int relbase = 0;  // (256 * s) mod 640
    for (int s = 0; s < nst; ++s) {
        // this wave's pieces of step s have landed, its reads of step s-1 are back; then all meet
        if (s == 4) tk[0] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (s == 4) tk[1] = wall_clock64();
        wait_and_barrier<0>();
        if (s == 4) tk[2] = wall_clock64();
        if (s == 0) stamp(p.stamps, 1);

        // next step's 256 positions: four pieces per wave, issued between the taps below
        int nxt_off[4];
        unsigned nxt_dst[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int slot0 = relbase + 384 + 8 * (8 * j + wave);  // scalar; ring position of the piece
            slot0 = slot0 >= kRing ? slot0 - kRing : slot0;
            slot0 = slot0 >= kRing ? slot0 - kRing : slot0;
            const int u = ub - kStripMargin + 256 * s + 384 + 8 * (8 * j + wave) + prow;
            nxt_off[j] = s + 1 < nst ? src_off(u, slot0 + prow) : kOob;
            nxt_dst[j] = lds_base + (unsigned)(slot0 * 128);
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        // 72 (fragment read, MFMA) pairs in k order: n = (4 tap + ks) * 2 + i.  The reads run kDepth
        // pairs ahead of their MFMA through a ring of fragment registers -- left to itself hipcc
        // issues each read one MFMA before its use (the 144 weight registers leave it little
        // room) and the matrix pipe waits out the LDS latency 72 times a step.
        constexpr int kDepth = 6;
        i32x4 px[kDepth];
        int y[9][2];
        auto read = [&](int n) -> i32x4 {
            const int tap = n >> 3, ks = (n >> 1) & 3, i = n & 1;
            if (ks == 0) {
                const int shift = (tap / 3 - 1) * p.Wp + (tap % 3 - 1);
                int sb = relbase + kStripMargin + 64 * mq + 32 * i + shift;  // scalar, >= 0
                sb = sb >= kRing ? sb - kRing : sb;
                const unsigned r0 = (unsigned)(sb + li);
                const unsigned r = min(r0, r0 - (unsigned)kRing);  // wraps at most once
                y[tap][i] = (int)((r << 7) | (((unsigned)lh ^ ((r >> 1) & 7u)) << 4));
            }
            return *reinterpret_cast<const i32x4 *>(lds + (y[tap][i] ^ (ks << 5)));
        };
#pragma unroll
        for (int n = 0; n < kDepth; ++n) px[n] = read(n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 72; ++n) {
            acc[n & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[n >> 1]),
                                                                 __builtin_bit_cast(bf16x8, px[n % kDepth]),
                                                                 acc[n & 1], 0, 0, 0);
            if (n + kDepth < 72) px[n % kDepth] = read(n + kDepth);
            __builtin_amdgcn_sched_barrier(0);  // keep the read where it is written, kDepth pairs ahead
            // The step's eight vector-memory instructions -- four DMA pieces for the next step, four
            // stores of the previous step's results (nothing at s == 0) -- one every eight MFMAs.
            // Issued together at the top of the step, the 64 of a block queued in front of the
            // CU's one address unit and the waves started their MFMAs up to 1.5 us apart.
            if ((n & 7) == 1) {
                if (n < 32)
                    dma16(nxt_off[n >> 3], srd_in, 0, (unsigned)__builtin_amdgcn_readfirstlane((int)nxt_dst[n >> 3]));
                else if (n < 64)
                    store_piece((n >> 3) - 4);
            }
        }
        if (s == 4) tk[3] = wall_clock64();

        // results: lane (li, lh) holds channels 32nf + 8j + 4lh + {0..3}, j = 0..3, of pixel li of
        // each fragment.  Affine + ReLU, bf16, then the half-waves trade 4-channel groups so that
        // each lane owns channels 32nf + 16h + 8lh + {0..7}: two 16-byte stores per fragment.
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 sc = *reinterpret_cast<const float4 *>(ssl + 32 * nf + 8 * j + 4 * lh);
                const float4 sh = *reinterpret_cast<const float4 *>(ssl + 64 + 32 * nf + 8 * j + 4 * lh);
                float v[4] = {fmaf(acc[i][4 * j], sc.x, sh.x), fmaf(acc[i][4 * j + 1], sc.y, sh.y),
                              fmaf(acc[i][4 * j + 2], sc.z, sh.z), fmaf(acc[i][4 * j + 3], sc.w, sh.w)};
                typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x2 o;
                    o[0] = (bf16_t)(p.relu ? fmaxf(v[2 * c], 0.f) : v[2 * c]);
                    o[1] = (bf16_t)(p.relu ? fmaxf(v[2 * c + 1], 0.f) : v[2 * c + 1]);
                    d[j][c] = __builtin_bit_cast(unsigned, o);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // upper half of d[2h] <-> lower half of d[2h+1]
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                pend[i][h] = i32x4{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
            }
            const int g = pixel_of(ub + 256 * s + 64 * mq + 32 * i + li);
            pend_off[i] = g < 0 ? kOob : g * 128 + (32 * nf + 8 * lh) * 2;
        }
        relbase = relbase + 256 >= kRing ? relbase + 256 - kRing : relbase + 256;
        if (s == 4) tk[4] = wall_clock64();
        if (s == 5) tk[5] = wall_clock64();
    }
    stamp(p.stamps, 10);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_piece(q);
    if (p.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(p.stamps, 11);
        if (threadIdx.x == 0)
            for (int i = 0; i < 6; ++i) p.stamps[(size_t)blockIdx.x * 16 + 2 + i] = tk[i];
    }
}


// ---- the strip kernel for 128 -> 128 channels (conv2 of the second stage) ----
//
// The same construction with K = 1,152: a 32-channel fragment's weights are 72 k-steps x 4 = 288
// registers, which only a wave that has its SIMD to itself can hold -- so 4 waves per block, one per
// SIMD, 512 registers each; wave w owns output channels 32w .. 32w+31 and multiplies ALL of a step's
// 128 positions (four fragments, four independent accumulators).  The ring holds the two 128-byte
// channel segments of a position as two images of 320 positions.  The weights (295 KB) pass through
// LDS in two halves before the ring is in use.
constexpr int kRing2 = 320, kMargin2 = 32, kSeg2 = kRing2 * 128;  // W + 3 <= 32

__global__ __launch_bounds__(256) void conv_strip128_kernel(const StripParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[64 * 145 * 16];  // >= 2 * kSeg2: weight staging is the larger
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int prow = lane >> 3, pc = lane & 7;
    stamp(p.stamps, 0);
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

    // weights: two rounds of 64 output channels; LDS rows of 145 chunks (144 + 1: 145r mod 16 differs
    // for the 16 rows of a ds_read_b128 lane group)
    i32x4 wreg[72];
    {
        const i32x4 srd_w = make_srd(p.w, 128 * 1152 * 2);
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            if (round) __syncthreads();  // the first half has been picked up
#pragma unroll 1
            for (int q = wave; q < 145; q += 4) {  // 64 rows x 145 chunks = 145 pieces of 64 chunks
                const int pos = q * 64 + lane, r = (pos * 3616) >> 19, c = pos - r * 145;  // pos / 145, pos < 9,280
                dma16(c < 144 ? ((64 * round + r) * 144 + c) * 16 : kOob, srd_w, 0,
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(q * 1024))));
            }
            wait_and_barrier<0>();
            if ((wave >> 1) == round) {
                const char *wrow = lds + (32 * (wave & 1) + li) * (145 * 16) + lh * 16;
#pragma unroll
                for (int s = 0; s < 72; ++s) wreg[s] = *reinterpret_cast<const i32x4 *>(wrow + s * 32);
            }
        }
        __syncthreads();
    }
    stamp(p.stamps, 1);
    // channel constants of this lane's 16 output channels (D map: 32w + 8j + 4lh + {0..3})
    float4 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c4 = 32 * wave + 8 * j + 4 * lh;
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
    // piece of 8 ring positions starting at ring position slot0, channel segment seg
    auto src_off = [&](int u, int slot, int seg) -> int {
        const int g = pixel_of(u);
        return g < 0 ? kOob : g * 256 + seg * 128 + ((pc ^ ((slot >> 1) & 7)) << 4);
    };
    // ring positions 0 .. 191 = padded positions ub - 32 .. ub + 159: all of step 0's; 48 pieces
#pragma unroll 1
    for (int q = wave; q < 48; q += 4) {
        const int seg = q / 24, slot = 8 * (q % 24) + prow;
        dma16(src_off(ub - kMargin2 + slot, slot, seg), srd_in, 0,
              (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(seg * kSeg2 + (q % 24) * 1024))));
    }

    i32x4 pend[4][2];
    int pend_off[4] = {kOob, kOob, kOob, kOob};
#pragma unroll
    for (int i = 0; i < 4; ++i) pend[i][0] = pend[i][1] = i32x4{0, 0, 0, 0};
    auto store_piece = [&](int q) {
        const int i = q >> 1, h = q & 1;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend[i][h]), rsrc_o,
                                               pend_off[i] == kOob ? kOob : pend_off[i] + 32 * h, 0, 0);
    };

    int relbase = 0;  // (128 * s) mod 320
    for (int s = 0; s < nst; ++s) {
        wait_and_barrier<0>();
        if (s == 0) stamp(p.stamps, 2);
        if (s == 1) stamp(p.stamps, 5);
        // next step's 128 positions: 32 pieces (2 segments x 16), eight per wave, issued among the MFMAs
        int nxt_off[8];
        unsigned nxt_dst[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 4 * j + wave, seg = q >> 4, g8 = q & 15;
            int slot0 = relbase + 192 + 8 * g8;
            slot0 = slot0 >= kRing2 ? slot0 - kRing2 : slot0;
            const int u = ub - kMargin2 + 128 * s + 192 + 8 * g8 + prow;
            nxt_off[j] = s + 1 < nst ? src_off(u, slot0 + prow, seg) : kOob;
            nxt_dst[j] = lds_base + (unsigned)(seg * kSeg2 + slot0 * 128);
        }

        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        // 288 (fragment read, MFMA) pairs: n = 4 * kstep + i, kstep = (tap * 2 + seg) * 4 + ks; reads kDepth
        // pairs ahead through a ring of fragment registers, pinned where they are written
        constexpr int kDepth = 8;
        i32x4 px[kDepth];
        int y[9][4];
        auto read = [&](int n) -> i32x4 {
            const int i = n & 3, kstep = n >> 2, ks = kstep & 3, seg = (kstep >> 2) & 1, tap = kstep >> 3;
            if (ks == 0 && seg == 0) {
                const int shift = (tap / 3 - 1) * p.Wp + (tap % 3 - 1);
                int sb = relbase + kMargin2 + 32 * i + shift;  // scalar, >= 0
                sb = sb >= kRing2 ? sb - kRing2 : sb;
                const unsigned r0 = (unsigned)(sb + li);
                const unsigned r = min(r0, r0 - (unsigned)kRing2);
                y[tap][i] = (int)((r << 7) | (((unsigned)lh ^ ((r >> 1) & 7u)) << 4));
            }
            return *reinterpret_cast<const i32x4 *>(lds + seg * kSeg2 + (y[tap][i] ^ (ks << 5)));
        };
#pragma unroll
        for (int n = 0; n < kDepth; ++n) px[n] = read(n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 288; ++n) {
            acc[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[n >> 2]),
                                                                 __builtin_bit_cast(bf16x8, px[n % kDepth]),
                                                                 acc[n & 3], 0, 0, 0);
            if (n + kDepth < 288) px[n % kDepth] = read(n + kDepth);
            __builtin_amdgcn_sched_barrier(0);
            if ((n & 15) == 3) {  // 18 slots: eight DMA pieces, then the eight stores of the previous step
                const int k = n >> 4;
                if (k < 8)
                    dma16(nxt_off[k], srd_in, 0, (unsigned)__builtin_amdgcn_readfirstlane((int)nxt_dst[k]));
                else if (k < 16)
                    store_piece(k - 8);
            }
        }

        if (s == 0) stamp(p.stamps, 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {fmaf(acc[i][4 * j], sc[j].x, sh[j].x), fmaf(acc[i][4 * j + 1], sc[j].y, sh[j].y),
                              fmaf(acc[i][4 * j + 2], sc[j].z, sh[j].z), fmaf(acc[i][4 * j + 3], sc[j].w, sh[j].w)};
                typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bf16x2 o;
                    o[0] = (bf16_t)(p.relu ? fmaxf(v[2 * c], 0.f) : v[2 * c]);
                    o[1] = (bf16_t)(p.relu ? fmaxf(v[2 * c + 1], 0.f) : v[2 * c + 1]);
                    d[j][c] = __builtin_bit_cast(unsigned, o);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto x0 = __builtin_amdgcn_permlane32_swap(d[2 * h][0], d[2 * h + 1][0], false, false);
                const auto x1 = __builtin_amdgcn_permlane32_swap(d[2 * h][1], d[2 * h + 1][1], false, false);
                pend[i][h] = i32x4{(int)x0[0], (int)x1[0], (int)x0[1], (int)x1[1]};
            }
            const int g = pixel_of(ub + 128 * s + 32 * i + li);
            pend_off[i] = g < 0 ? kOob : g * 256 + (32 * wave + 8 * lh) * 2;
        }
        relbase = relbase + 128 >= kRing2 ? relbase + 128 - kRing2 : relbase + 128;
        if (s == 0) stamp(p.stamps, 4);
    }
    stamp(p.stamps, 10);
#pragma unroll
    for (int q = 0; q < 8; ++q) store_piece(q);
}

}  // namespace

int rn_conv_wide_count(void) { return kNumTiles; }

void rn_conv_wide_tile(int which, int *bm, int *bn)
{
    *bm = kTiles[which].bm;
    *bn = kTiles[which].bn;
}

// bf16 in and out, whole 128-byte channel segments (not the small-Cin stem forms), 16-byte
// output rows, K unsplit; the caller has filled the shape fields of p
bool rn_conv_wide_eligible(const GemmParams &p, int which)
{
    if (which < 0 || which >= kNumTiles) return false;
    return p.chunk_dw == 0 && p.kreal == 0 && p.Cout % 8 == 0 && p.nk >= 1;
}

void rn_conv_wide_launch(rn_ctx *ctx, GemmParams &p, int which, bool dual)
{
    switch (which) {
    case 0: launch<256, 256, 2, 4>(ctx, p, dual); break;
    case 1: launch<256, 128, 4, 2>(ctx, p, dual); break;
    case 2: launch<128, 256, 2, 4>(ctx, p, dual); break;
    case 3: launch<256, 64, 8, 1>(ctx, p, dual); break;
    case 4: launch<224, 256, 1, 8>(ctx, p, dual); break;
    default: launch<128, 128, 2, 4, 80 * 1024>(ctx, p, dual); break;
    }
}

// ---- strip kernel: host side ----

bool rn_conv_strip_eligible(const GemmParams &p)
{
    const bool common = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.chunk_dw == 0 && p.kreal == 0 &&
                        p.tap_rows == 1 && p.residual == nullptr && p.Ho == p.H && p.Wo == p.W &&
                        (uint64_t)(p.M / (p.H * p.W) * (p.H + 1) + 1) * (uint64_t)(p.W + 2) < (1ull << 30);
    if (!common) return false;
    if (p.Cs == 64 && p.Cout == 64 && p.cseg == 1 && p.Ktot == 576) return p.W + 3 <= kStripMargin;
    return p.Cs == 128 && p.Cout == 128 && p.cseg == 2 && p.Ktot == 1152 && p.W + 3 <= kMargin2;
}

void rn_conv_strip_launch(rn_ctx *ctx, const GemmParams &g)
{
    StripParams p;
    p.in = g.in, p.w = g.w, p.out = g.out, p.scale = g.scale, p.shift = g.shift, p.relu = g.relu;
    p.H = g.H, p.W = g.W, p.B = g.M / (g.H * g.W);
    p.Wp = p.W + 2, p.Hq = p.H + 1;
    rn_fast_div((unsigned)p.Wp, &p.mul_wp, &p.shr_wp);
    rn_fast_div((unsigned)p.Hq, &p.mul_hq, &p.shr_hq);
    p.U = (p.B * p.Hq + 1) * p.Wp;
    p.in_bytes = g.in_bytes, p.out_bytes = g.out_bytes;
    p.stamps = g.stamps;
    if (g.Cs == 128) {
        p.nsteps = (p.U + 127) / 128;
        conv_strip128_kernel<<<dim3(p.nsteps < 256 ? p.nsteps : 256), dim3(256), 0, ctx->stream>>>(p);
        return;
    }
    p.nsteps = (p.U + 255) / 256;
    const int blocks = p.nsteps < 256 ? p.nsteps : 256;  // one block per CU
    conv_strip_kernel<<<dim3(blocks), dim3(512), 0, ctx->stream>>>(p);
}
