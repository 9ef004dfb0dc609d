// gemm_lab: ablation bench for the contraction kernel's main loop (fp32 MFMA).
// Standalone: hipcc -O3 --offload-arch=gfx950 tools/gemm_lab.hip -o gemm_lab && ./gemm_lab
// C[M][N] = A[M][K] * B[N][K]^T with the same LDS image, swizzle, fragment map and
// register-staged double buffering as resnet.c_amd/csrc/rn_conv.hip; LAB bits switch
// parts of the loop off to see where the MFMA pipe goes idle (outputs are then wrong;
// only the timing matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;

// -DLAB_BF16: the same byte layout read as bf16 (a 128-byte row = 64 elements, one
// ds_read_b128 = one operand of v_mfma_f32_32x32x16_bf16): K below is then in units of two
// bf16, and every K tile is 512 MFMA cycles per wave instead of 4096.
typedef __bf16 lab_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void lab_mfma(f32x16 &acc, const float4 &a, const float4 &b)
{
#ifdef LAB_BF16
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(lab_bf16x8, a), __builtin_bit_cast(lab_bf16x8, b), acc, 0, 0, 0);
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
#endif
}
#ifdef LAB_BF16
constexpr double kFlopScale = 2.0;
#else
constexpr double kFlopScale = 1.0;
#endif

enum { NO_GLOBAL = 1, NO_STAGE = 2, NO_LDSREAD = 4, PRIO = 8, NO_BARRIER = 16, SAME_TILE = 32, DEEP = 64 };

template <int BM, int BN, int LAB, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_kernel(const float *A, const float *B, float *C,
                                                        int M, int N, int K, int tiles_n)
{
    constexpr int AP = BM / 32, BP = BN / 32, MI = BM / 64, NI = BN / 64;
    constexpr int STAGE = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = (int)(logical % (unsigned)tiles_n), tile_m = (int)(logical / (unsigned)tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, c = t & 7, r0 = t >> 3;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, N * K * 4, 0x00020000);
    int a_off[AP], b_off[BP];
#pragma unroll
    for (int j = 0; j < AP; ++j) a_off[j] = ((m0 + r0 + 32 * j) * K + c * 4) * 4;
#pragma unroll
    for (int j = 0; j < BP; ++j) b_off[j] = ((n0 + r0 + 32 * j) * K + c * 4) * 4;
    float4 ra[AP], rb[BP];
#pragma unroll
    for (int j = 0; j < AP; ++j) ra[j] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int j = 0; j < BP; ++j) rb[j] = make_float4(1.f, 2.f, 3.f, 4.f);
    auto load_tile = [&](int kt) {
        if (LAB & NO_GLOBAL) return;
        const int soff = (LAB & SAME_TILE) ? 0 : kt * BK * 4;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_, a_off[j], soff, 0);
            ra[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb_, b_off[j], soff, 0);
            rb[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto store_tile = [&](int buf) {
        if (LAB & NO_STAGE) return;
        float *As = lds + buf * STAGE, *Bs = As + BM * BK;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
            *reinterpret_cast<float4 *>(As + row * BK + pc * 4) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
            *reinterpret_cast<float4 *>(Bs + row * BK + pc * 4) = rb[j];
        }
    };
    const int lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    float4 fa[MI], fb[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fa[mi] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fb[ni] = make_float4(1.f, 2.f + lane, 3.f, 4.f);
    auto compute_tile = [&](int buf) {
        const float *As = lds + buf * STAGE + (wr * (BM / 2) + li) * BK;
        const float *Bs = lds + buf * STAGE + BM * BK + (wc * (BN / 2) + li) * BK;
        if (LAB & PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int pc = ((2 * ks + lh) ^ sw) * 4;
            float4 a[MI], b[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                a[mi] = (LAB & NO_LDSREAD) ? fa[mi] : *reinterpret_cast<const float4 *>(As + mi * 32 * BK + pc);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                b[ni] = (LAB & NO_LDSREAD) ? fb[ni] : *reinterpret_cast<const float4 *>(Bs + ni * 32 * BK + pc);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    lab_mfma(acc[mi][ni], a[mi], b[ni]);
                }
        }
        if (LAB & PRIO) __builtin_amdgcn_s_setprio(0);
    };
    const int nk = K / BK;
    if (LAB & DEEP) {
        // two register sets: loads run two K tiles ahead of the MFMAs that consume them
        float4 ra2[AP], rb2[BP];
        auto load2 = [&](int kt) {
            const int soff = kt * BK * 4;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_, a_off[j], soff, 0);
                ra2[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rb_, b_off[j], soff, 0);
                rb2[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        };
        auto store2 = [&](int buf) {
            float *As = lds + buf * STAGE, *Bs = As + BM * BK;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
                *reinterpret_cast<float4 *>(As + row * BK + pc * 4) = ra2[j];
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
                *reinterpret_cast<float4 *>(Bs + row * BK + pc * 4) = rb2[j];
            }
        };
        // nk is even in this lab
        load_tile(0);
        store_tile(0);
        load_tile(1);   // set 1 holds tile 1
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            // even step: compute tile kt from buf0; set1 holds kt+1; issue kt+2 into set2
            if (kt + 2 < nk) load2(kt + 2);
            compute_tile(0);
            if (kt + 1 < nk) store_tile(1);   // tile kt+1 (loaded one full step ago)
            __syncthreads();
            // odd step: compute tile kt+1 from buf1; set2 holds kt+2; issue kt+3 into set1
            if (kt + 3 < nk) load_tile(kt + 3);
            compute_tile(1);
            if (kt + 2 < nk) store2(0);
            __syncthreads();
        }
    } else {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        compute_tile(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        if (!(LAB & NO_BARRIER)) __syncthreads();
    }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wc * (BN / 2) + ni * 32 + li;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int mb = m0 + wr * (BM / 2) + mi * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) C[(size_t)(mb + (e & 3) + 8 * (e >> 2)) * N + n] = acc[mi][ni][e];
        }
    }
}


// ---------------------------------------------------------------------------------
// Wave-specialised variant: 8 waves per block.  Waves 0-3 (one per SIMD) only read
// fragments from LDS and issue MFMAs; waves 4-7 only fetch operands (buffer loads,
// D tiles deep in registers) and write them into a 3-stage LDS ring.  One barrier per
// K tile; B_t = "tile t published".  Consumers execute B_{i+1} at the start of
// iteration i, so a producer passing B_t knows tile t-2 has been consumed and stage
// (t+1)%3 is free.
template <int BM, int BN, int D, int SAME = 0>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(const float *A, const float *B, float *C,
                                                         int M, int N, int K, int tiles_n)
{
    constexpr int AP = BM / 32, BP = BN / 32, MI = BM / 64, NI = BN / 64;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int S = 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = (int)(logical % (unsigned)tiles_n), tile_m = (int)(logical / (unsigned)tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[MI][NI];
    const int lane = threadIdx.x & 63, cw = wave & 3, wr = cw >> 1, wc = cw & 1, li = lane & 31, lh = lane >> 5;

    if (wave >= 4) {
        // ---------------- producer ----------------
        const int t = threadIdx.x - 256, c = t & 7, r0 = t >> 3;
        const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, M * K * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, N * K * 4, 0x00020000);
        int a_off[AP], b_off[BP];
#pragma unroll
        for (int j = 0; j < AP; ++j) a_off[j] = ((m0 + r0 + 32 * j) * K + c * 4) * 4;
#pragma unroll
        for (int j = 0; j < BP; ++j) b_off[j] = ((n0 + r0 + 32 * j) * K + c * 4) * 4;
        u32x4 ra[D][AP], rb[D][BP];
        auto issue = [&](int kt, u32x4 (&xa)[AP], u32x4 (&xb)[BP]) {
            const int soff = SAME ? 0 : kt * BK * 4;
#pragma unroll
            for (int j = 0; j < AP; ++j) xa[j] = __builtin_amdgcn_raw_buffer_load_b128(ra_, a_off[j], soff, 0);
#pragma unroll
            for (int j = 0; j < BP; ++j) xb[j] = __builtin_amdgcn_raw_buffer_load_b128(rb_, b_off[j], soff, 0);
        };
        auto stage_write = [&](int st, u32x4 (&xa)[AP], u32x4 (&xb)[BP]) {
            float *As = lds + st * STAGE, *Bs = As + BM * BK;
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
                *reinterpret_cast<u32x4 *>(As + row * BK + pc * 4) = xa[j];
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
                *reinterpret_cast<u32x4 *>(Bs + row * BK + pc * 4) = xb[j];
            }
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nk) issue(d, ra[d], rb[d]);
        int st = 0;
        for (int t0 = 0; t0 < nk; t0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int kt = t0 + d;
                if (kt < nk) {
                    stage_write(st, ra[d], rb[d]);
                    if (kt + D < nk) issue(kt + D, ra[d], rb[d]);
                    st = st == S - 1 ? 0 : st + 1;
                    __syncthreads();  // B_kt
                }
            }
        }
        __syncthreads();  // E0
        __syncthreads();  // E1
    } else {
        // ---------------- consumer ----------------
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        const int sw = (li >> 1) & 7;
        const int a_base = (wr * (BM / 2) + li) * BK, b_base = BM * BK + (wc * (BN / 2) + li) * BK;
        float4 fa[2][MI], fb[2][NI];
        auto read_frags = [&](int st, int ks, float4 (&xa)[MI], float4 (&xb)[NI]) {
            const float *base = lds + st * STAGE;
            const int pc = ((2 * ks + lh) ^ sw) * 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<const float4 *>(base + a_base + mi * 32 * BK + pc);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) xb[ni] = *reinterpret_cast<const float4 *>(base + b_base + ni * 32 * BK + pc);
        };
        auto mfmas = [&](float4 (&xa)[MI], float4 (&xb)[NI]) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi].x, xb[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi].y, xb[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi].z, xb[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi].w, xb[ni].w, acc[mi][ni], 0, 0, 0);
                }
        };
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();  // B_0
        read_frags(0, 0, fa[0], fb[0]);
        int st = 0;
        for (int i = 0; i < nk; ++i) {
            const int nst = st == S - 1 ? 0 : st + 1;
            if (i + 1 < nk) __syncthreads();  // B_{i+1}
            read_frags(st, 1, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(st, 2, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(st, 3, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nk) read_frags(nst, 0, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            st = nst;
        }
        __syncthreads();  // E0: every wave is done with the ring
        float *Cs = lds;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float *dst = Cs + (wr * (BM / 2) + mi * 32 + 4 * lh) * BN + wc * (BN / 2) + ni * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[mi][ni][e];
            }
        __syncthreads();  // E1
    }
    // all 8 waves: row-contiguous float4 stores
    constexpr int C4 = BN / 4, RPP = 512 / C4, PASSES = BM / RPP;
    const int t = threadIdx.x, c4 = t % C4, rr = t / C4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = rr + ps * RPP;
        const float4 v = *reinterpret_cast<const float4 *>(lds + row * BN + c4 * 4);
        *reinterpret_cast<float4 *>(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = v;
    }
}

template <int BM, int BN, int D, int SAME = 0>
float run_ws(const float *A, const float *B, float *C, int M, int N, int K, int reps)
{
    const int tn = N / BN, tm = M / BM;
    const size_t lds_bytes = 3 * (BM + BN) * BK * 4;
    hipFuncSetAttribute((const void *)gemm_ws_kernel<BM, BN, D, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) gemm_ws_kernel<BM, BN, D, SAME><<<tm * tn, 512, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gemm_ws_kernel<BM, BN, D, SAME><<<tm * tn, 512, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  WS%s BM=%d BN=%d D=%d : %.3f ms  %.1f TF  (%s)\n", SAME ? " sametile" : "", BM, BN, D, ms, 2.0 * M * N * K / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
    return ms;
}


// ---------------------------------------------------------------------------------
// B-direct variant: only the A operand goes through LDS; every wave fetches its B operand
// (the weights: small, L2-resident, K-contiguous rows) from global memory straight into
// MFMA-layout registers -- lane (li, lh) of k-step ks needs the 16 bytes at row n = li,
// byte 128*kt + 16*(2*ks + lh), which is exactly one buffer_load_dwordx4.  Halves the LDS
// bytes written and read per MFMA; B fragments of tile kt+1 travel while tile kt is
// multiplied (two register sets, loop unrolled by two).
template <int BM, int BN, int WPS>
__global__ __launch_bounds__(256, WPS) void gemm_bdirect_kernel(const float *A, const float *B, float *C,
                                                                int M, int N, int K, int tiles_n)
{
    constexpr int AP = BM / 32, MI = BM / 64, NI = BN / 64;
    constexpr int STAGE = BM * BK;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // max(2*STAGE, BM*BN) floats
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = (int)(logical % (unsigned)tiles_n), tile_m = (int)(logical / (unsigned)tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, c = t & 7, r0 = t >> 3;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, N * K * 4, 0x00020000);
    const int lane = t & 63, wave = t >> 6, wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 1) & 7;
    int a_off[AP], b_off[NI];
#pragma unroll
    for (int j = 0; j < AP; ++j) a_off[j] = ((m0 + r0 + 32 * j) * K + c * 4) * 4;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#ifdef LAB_BFRAG
        // B re-packed in fragment order: [32-row group][K tile][k-step][lane] x 16 bytes, so a
        // wave's fragment load is one contiguous KiB
        b_off[ni] = (((n0 + wc * (BN / 2) + ni * 32) / 32) * (K / BK) * 4 * 64 + lane) * 16;
#else
        b_off[ni] = ((n0 + wc * (BN / 2) + ni * 32 + li) * K + lh * 4) * 4;
#endif
    }
    u32x4 ra[AP];
    u32x4 fb0[4][NI], fb1[4][NI];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int j = 0; j < AP; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(ra_, a_off[j], kt * 128, 0);
    };
    auto load_b = [&](int kt, u32x4 (&fb)[4][NI]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#ifdef LAB_BFRAG
                fb[ks][ni] = __builtin_amdgcn_raw_buffer_load_b128(rb_, b_off[ni], (kt * 4 + ks) * 1024, 0);
#else
                fb[ks][ni] = __builtin_amdgcn_raw_buffer_load_b128(rb_, b_off[ni], kt * 128 + ks * 32, 0);
#endif
    };
    auto store_a = [&](int buf) {
        float *As = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            const int row = r0 + 32 * j, pc = c ^ ((row >> 1) & 7);
            *reinterpret_cast<u32x4 *>(As + row * BK + pc * 4) = ra[j];
        }
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    auto compute = [&](int buf, const u32x4 (&fb)[4][NI]) {
        const float *As = lds + buf * STAGE + (wr * (BM / 2) + li) * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int pc = ((2 * ks + lh) ^ sw) * 4;
            float4 a[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const float4 *>(As + mi * 32 * BK + pc);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float4 b = make_float4(__uint_as_float(fb[ks][ni].x), __uint_as_float(fb[ks][ni].y),
                                                 __uint_as_float(fb[ks][ni].z), __uint_as_float(fb[ks][ni].w));
                    lab_mfma(acc[mi][ni], a[mi], b);
                }
        }
    };
    const int nk = K / BK;  // even in every shape the lab runs
    load_a(0);
    load_b(0, fb0);
    store_a(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        load_a(kt + 1);  // kt + 1 < nk because nk is even
        load_b(kt + 1, fb1);
        compute(0, fb0);
        store_a(1);
        __syncthreads();
        if (kt + 2 < nk) {
            load_a(kt + 2);
            load_b(kt + 2, fb0);
        }
        compute(1, fb1);
        if (kt + 2 < nk) store_a(0);
        __syncthreads();
    }
    float *Cs = lds;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float *dst = Cs + (wr * (BM / 2) + mi * 32 + 4 * lh) * BN + wc * (BN / 2) + ni * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[mi][ni][e];
        }
    __syncthreads();
    constexpr int C4 = BN / 4, RPP = 256 / C4, PASSES = BM / RPP;
    const int c4 = t % C4, rr = t / C4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = rr + ps * RPP;
        const float4 v = *reinterpret_cast<const float4 *>(lds + row * BN + c4 * 4);
        *reinterpret_cast<float4 *>(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = v;
    }
}

template <int BM, int BN, int WPS>
float run_bdirect(const float *A, const float *B, float *C, int M, int N, int K, int reps)
{
    const int tn = N / BN, tm = M / BM;
    size_t lds_bytes = (size_t)2 * BM * BK * 4;
    if (lds_bytes < (size_t)BM * BN * 4) lds_bytes = (size_t)BM * BN * 4;
    hipFuncSetAttribute((const void *)gemm_bdirect_kernel<BM, BN, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) gemm_bdirect_kernel<BM, BN, WPS><<<tm * tn, 256, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gemm_bdirect_kernel<BM, BN, WPS><<<tm * tn, 256, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  B-direct BM=%d BN=%d WPS=%d lds=%zuK : %.3f ms  %.1f TF  (%s)\n", BM, BN, WPS, lds_bytes / 1024, ms,
           kFlopScale * 2.0 * M * N * K / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return ms;
}

// ---------------------------------------------------------------------------------
// LDS-DMA variant: operands go global -> LDS with global_load_lds (16 B per lane,
// 1 KiB = 8 rows x 128 B per wave instruction, swizzle applied on the SOURCE address),
// no VGPR staging and no ds_write.  S-stage ring, tiles t+1 .. t+S-1 in flight while
// tile t is multiplied; one raw s_barrier per K tile with a counted vmcnt.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int BM, int BN, int S, int PIN, int SAME = 0>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const float *A, const float *B, float *C,
                                                       int M, int N, int K, int tiles_n)
{
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int STAGE = (BM + BN) * BK;
    constexpr int PA = BM / 32, PB = BN / 32;  // 1-KiB pieces per wave per tile (A, B)
    constexpr int NPIECE = PA + PB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = (int)(logical % (unsigned)tiles_n), tile_m = (int)(logical / (unsigned)tiles_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = K / BK;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    // DMA piece geometry: piece p covers rows 8p..8p+7; this lane fills (row 8p + lane/8,
    // physical chunk lane%8) and therefore fetches logical chunk pc ^ ((row>>1)&7).
    const int prow = lane >> 3, pc = lane & 7;
    const float *a_src[PA];
    const float *b_src[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = 8 * (wave + 4 * j) + prow;
        a_src[j] = A + (size_t)(m0 + row) * K + ((pc ^ ((row >> 1) & 7)) * 4);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = 8 * (wave + 4 * j) + prow;
        b_src[j] = B + (size_t)(n0 + row) * K + ((pc ^ ((row >> 1) & 7)) * 4);
    }
    // The DMA is issued from inline asm so that hipcc's waitcnt pass does not see a
    // pending LDS write (it would put vmcnt(0) in front of every ds_read); completion is
    // counted by hand below.  M0 carries the wave-uniform LDS byte address.
    const unsigned lds_base = (unsigned)(uintptr_t)((lds_void *)lds);
    auto glds16 = [&](const float *gsrc, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    };
    auto dma_tile = [&](int kt, int st) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(st * STAGE + wave * 256) * 4u);
        const unsigned sb = sa + BM * BK * 4;
        if (SAME) kt = 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) glds16(a_src[j] + kt * BK, sa + j * 4096);
#pragma unroll
        for (int j = 0; j < PB; ++j) glds16(b_src[j] + kt * BK, sb + j * 4096);
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    const int sw = (li >> 1) & 7;
    const int a_base = (wr * (BM / 2) + li) * BK, b_base = BM * BK + (wc * (BN / 2) + li) * BK;
    float4 fa[2][MI], fb[2][NI];
    auto read_frags = [&](int st, int ks, float4 (&xa)[MI], float4 (&xb)[NI]) {
        const float *base = lds + st * STAGE;
        const int pcs = ((2 * ks + lh) ^ sw) * 4;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xa[mi] = *reinterpret_cast<const float4 *>(base + a_base + mi * 32 * BK + pcs);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) xb[ni] = *reinterpret_cast<const float4 *>(base + b_base + ni * 32 * BK + pcs);
    };
    auto mfmas = [&](float4 (&xa)[MI], float4 (&xb)[NI]) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                lab_mfma(acc[mi][ni], xa[mi], xb[ni]);
            }
    };
    // prologue: S-1 tiles in flight
#pragma unroll
    for (int d = 0; d < S - 1; ++d)
        if (d < nk) dma_tile(d, d);
    int st = 0;
    for (int i = 0; i < nk; ++i) {
        // retire this wave's pieces of tile i (younger tiles may stay in flight)
        const int younger = min(S - 2, nk - 1 - i);
#define LAB_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n) * NPIECE > 63 ? 63 : (n) * NPIECE) : "memory")
        switch (younger) {
        case 0: LAB_WAIT(0); break;
        case 1: LAB_WAIT(1); break;
        case 2: LAB_WAIT(2); break;
        case 3: LAB_WAIT(3); break;
        case 4: LAB_WAIT(4); break;
        case 5: LAB_WAIT(5); break;
        default: LAB_WAIT(6); break;
        }
#undef LAB_WAIT
        asm volatile("s_barrier" ::: "memory");
        // stage (i-1)%S is free now (everyone finished tile i-1): refill it with tile i+S-1
        if (i + S - 1 < nk) dma_tile(i + S - 1, st == 0 ? S - 1 : st - 1);
        if (PIN) {
            read_frags(st, 0, fa[0], fb[0]);
            read_frags(st, 1, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(st, 2, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_frags(st, 3, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                read_frags(st, ks, fa[0], fb[0]);
                mfmas(fa[0], fb[0]);
            }
        }
        st = st == S - 1 ? 0 : st + 1;
    }
    asm volatile("s_barrier" ::: "memory");
    float *Cs = lds;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            float *dst = Cs + (wr * (BM / 2) + mi * 32 + 4 * lh) * BN + wc * (BN / 2) + ni * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[mi][ni][e];
        }
    __syncthreads();
    constexpr int C4 = BN / 4, RPP = 256 / C4, PASSES = BM / RPP;
    const int c4 = t % C4, rr = t / C4;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int row = rr + ps * RPP;
        const float4 v = *reinterpret_cast<const float4 *>(lds + row * BN + c4 * 4);
        *reinterpret_cast<float4 *>(C + (size_t)(m0 + row) * N + n0 + c4 * 4) = v;
    }
}

template <int BM, int BN, int S, int PIN, int SAME = 0>
float run_dma(const float *A, const float *B, float *C, int M, int N, int K, int reps)
{
    const int tn = N / BN, tm = M / BM;
    size_t lds_bytes = (size_t)S * (BM + BN) * BK * 4;
    if (lds_bytes < (size_t)BM * BN * 4) lds_bytes = (size_t)BM * BN * 4;
    hipFuncSetAttribute((const void *)gemm_dma_kernel<BM, BN, S, PIN, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) gemm_dma_kernel<BM, BN, S, PIN, SAME><<<tm * tn, 256, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gemm_dma_kernel<BM, BN, S, PIN, SAME><<<tm * tn, 256, lds_bytes>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  DMA%s BM=%d BN=%d S=%d pin=%d lds=%zuK : %.3f ms  %.1f TF  (%s)\n", SAME ? " sametile" : "", BM, BN, S, PIN, lds_bytes / 1024, ms,
           kFlopScale * 2.0 * M * N * K / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return ms;
}

static double check(const float *C, const float *Cref, size_t n)
{
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), C, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), Cref, n * 4, hipMemcpyDeviceToHost);
    double mx = 0;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, (double)fabsf(a[i] - b[i]));
    return mx;
}

template <int BM, int BN, int LAB, int WPS>
float run(const float *A, const float *B, float *C, int M, int N, int K, int reps, int dyn = 0)
{
    const int tn = N / BN, tm = M / BM;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    if (dyn) hipFuncSetAttribute((const void *)gemm_kernel<BM, BN, LAB, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    for (int i = 0; i < 2; ++i) gemm_kernel<BM, BN, LAB, WPS><<<tm * tn, 256, dyn>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gemm_kernel<BM, BN, LAB, WPS><<<tm * tn, 256, dyn>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  %sBM=%d BN=%d WPS=%d LAB=%2d%s%s%s%s%s%s : %.3f ms  %.1f TF\n", dyn ? "[1 block/CU] " : "", BM, BN, WPS, LAB, (LAB & DEEP) ? " deep" : (LAB & SAME_TILE) ? " sametile" : "",
           (LAB & NO_GLOBAL) ? " -global" : "", (LAB & NO_STAGE) ? " -stage" : "",
           (LAB & NO_LDSREAD) ? " -ldsread" : "", (LAB & PRIO) ? " +prio" : "",
           (LAB & NO_BARRIER) ? " -barrier" : "", ms, kFlopScale * 2.0 * M * N * K / ms / 1e9);
    return ms;
}

#ifdef LAB_BF16
int main()
{
    // layer3 3x3 conv (M = 50176, N = 256, K = 2304 bf16) as a plain GEMM; K in float units
    const int M = 50176, N = 256, K = 1152;
    float *A, *B, *C, *C2;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)N * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&C2, (size_t)M * N * 4);
    std::vector<unsigned short> h((size_t)M * K * 2);
    for (auto &v : h) {  // bf16 bit patterns of values in [-1, 1)
        const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("bf16: M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, 2 * K, 4.0 * M * N * K / 1e9);
    float *Bf = B;
#ifdef LAB_BFRAG
    {
        const float *hb = reinterpret_cast<const float *>(h.data());  // B = first N rows of h
        std::vector<float> f((size_t)N * K);
        const int nkt = K / BK;
        for (int g = 0; g < N / 32; ++g)
            for (int kt = 0; kt < nkt; ++kt)
                for (int ks = 0; ks < 4; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 31, lh = lane >> 5;
                        const float *src = hb + (size_t)(g * 32 + li) * K + kt * BK + (2 * ks + lh) * 4;
                        float *dst = f.data() + ((((size_t)g * nkt + kt) * 4 + ks) * 64 + lane) * 4;
                        for (int e = 0; e < 4; ++e) dst[e] = src[e];
                    }
        hipMalloc(&Bf, (size_t)N * K * 4);
        hipMemcpy(Bf, f.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    }
#endif
    const int R = 20;
    run<128, 128, 0, 2>(A, B, C, M, N, K, R);
    run<128, 64, 0, 3>(A, B, C2, M, N, K, R);
    run<64, 64, 0, 4>(A, B, C2, M, N, K, R);
    run<128, 128, NO_GLOBAL, 2>(A, B, C2, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE, 2>(A, B, C2, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 2>(A, B, C2, M, N, K, R);
    run_bdirect<128, 128, 2>(A, Bf, C2, M, N, K, R);
    printf("    max|bdirect - base| = %g\n", check(C2, C, (size_t)M * N));
    run_bdirect<128, 64, 2>(A, Bf, C2, M, N, K, R);
    run_bdirect<128, 64, 3>(A, Bf, C2, M, N, K, R);
    printf("    max|bdirect - base| = %g\n", check(C2, C, (size_t)M * N));
    run_bdirect<64, 64, 3>(A, Bf, C2, M, N, K, R);
    run_bdirect<64, 64, 4>(A, Bf, C2, M, N, K, R);
    run_bdirect<64, 128, 2>(A, Bf, C2, M, N, K, R);
    run_bdirect<64, 128, 3>(A, Bf, C2, M, N, K, R);
    run_dma<128, 128, 2, 0>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    run_dma<128, 128, 3, 0>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    run_dma<128, 128, 4, 0>(A, B, C2, M, N, K, R);
    run_dma<128, 128, 3, 1>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 3, 0>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 4, 0>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 6, 0>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 3, 0>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 4, 0>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 6, 0>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 8, 0>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    return 0;
}
#elif defined(LAB_F32_64)
int main()
{
    // the 64x64 tile (the tuner's usual pick) on the layer3 3x3 shape with M cut to an exact
    // 12 tiles per CU (no tail), fp32: which part of the loop keeps the matrix pipe idle?
    const int M = 49152, N = 256, K = 2304;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)N * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("fp32: M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
    const int R = 10;
    run<64, 64, 0, 4>(A, B, C, M, N, K, R);
    run<64, 64, PRIO, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_GLOBAL, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_GLOBAL | NO_STAGE, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_GLOBAL | NO_STAGE | NO_BARRIER, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_GLOBAL | NO_STAGE | NO_LDSREAD, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_BARRIER, 4>(A, B, C, M, N, K, R);
    run<64, 64, NO_LDSREAD, 4>(A, B, C, M, N, K, R);
    run<64, 64, SAME_TILE, 4>(A, B, C, M, N, K, R);
    run<128, 128, 0, 2>(A, B, C, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 2>(A, B, C, M, N, K, R);
    run<128, 64, 0, 3>(A, B, C, M, N, K, R);
    run<128, 64, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 3>(A, B, C, M, N, K, R);
    return 0;
}
#else
int main()
{
    const int M = 200704, N = 128, K = 1152;  // layer2 3x3 conv as a plain GEMM
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&B, (size_t)N * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
    const int R = 10;
    float *C2;
    hipMalloc(&C2, (size_t)M * N * 4);
    run<128, 128, 0, 2>(A, B, C, M, N, K, R);
    run_dma<128, 128, 3, 0>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    run_dma<128, 128, 3, 1>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    run_dma<128, 128, 3, 0, 1>(A, B, C2, M, N, K, R);
    run_dma<128, 128, 3, 1, 1>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 3, 0, 1>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 3, 1, 1>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 4, 1, 1>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 3, 0>(A, B, C2, M, N, K, R);
    run_dma<128, 64, 3, 1>(A, B, C2, M, N, K, R);
    printf("    max|dma - base| = %g\n", check(C2, C, (size_t)M * N));
    run_dma<128, 64, 4, 1>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 3, 1>(A, B, C2, M, N, K, R);
    run_dma<64, 64, 4, 1>(A, B, C2, M, N, K, R);
    return 0;
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 2>(A, B, C, M, N, K, R, 90 * 1024);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_BARRIER, 2>(A, B, C, M, N, K, R, 90 * 1024);
    run<128, 128, NO_GLOBAL | NO_STAGE, 2>(A, B, C, M, N, K, R, 90 * 1024);
    run<128, 128, NO_GLOBAL, 2>(A, B, C, M, N, K, R, 90 * 1024);
    run<128, 128, 0, 2>(A, B, C, M, N, K, R, 90 * 1024);
    run_ws<128, 128, 2>(A, B, C2, M, N, K, R);
    printf("    max|ws - base| = %g\n", check(C2, C, (size_t)M * N));
    run_ws<128, 128, 1>(A, B, C2, M, N, K, R);
    run_ws<128, 128, 1, 1>(A, B, C2, M, N, K, R);
    run_ws<128, 64, 1>(A, B, C2, M, N, K, R);
    run_ws<128, 64, 1, 1>(A, B, C2, M, N, K, R);
    run_ws<64, 64, 1>(A, B, C2, M, N, K, R);
    run_ws<128, 64, 2>(A, B, C2, M, N, K, R);
    printf("    max|ws - base| = %g\n", check(C2, C, (size_t)M * N));
    run_ws<128, 64, 3>(A, B, C2, M, N, K, R);
    run_ws<64, 64, 3>(A, B, C2, M, N, K, R);
    return 0;
    run<128, 128, NO_GLOBAL, 2>(A, B, C, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE, 2>(A, B, C, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_BARRIER, 2>(A, B, C, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_LDSREAD, 2>(A, B, C, M, N, K, R);
    run<128, 128, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 2>(A, B, C, M, N, K, R);
    run<128, 128, DEEP, 2>(A, B, C, M, N, K, R);
    run<128, 128, DEEP | PRIO, 2>(A, B, C, M, N, K, R);
    run<128, 64, DEEP, 3>(A, B, C, M, N, K, R);
    run<128, 64, DEEP | PRIO, 3>(A, B, C, M, N, K, R);
    run<128, 64, DEEP | PRIO, 2>(A, B, C, M, N, K, R);
    run<128, 128, SAME_TILE, 2>(A, B, C, M, N, K, R);
    run<128, 128, SAME_TILE | PRIO, 2>(A, B, C, M, N, K, R);
    run<128, 128, PRIO, 2>(A, B, C, M, N, K, R);
    run<128, 128, PRIO | NO_GLOBAL, 2>(A, B, C, M, N, K, R);
    run<128, 64, PRIO, 3>(A, B, C, M, N, K, R);
    run<128, 64, SAME_TILE, 3>(A, B, C, M, N, K, R);
    run<128, 64, SAME_TILE | PRIO, 3>(A, B, C, M, N, K, R);
    run<128, 128, NO_LDSREAD, 2>(A, B, C, M, N, K, R);
    run<128, 64, 0, 2>(A, B, C, M, N, K, R);
    run<128, 64, 0, 3>(A, B, C, M, N, K, R);
    run<128, 64, NO_GLOBAL | NO_STAGE | NO_LDSREAD | NO_BARRIER, 3>(A, B, C, M, N, K, R);
    run<64, 64, 0, 4>(A, B, C, M, N, K, R);
    return 0;
}
#endif
