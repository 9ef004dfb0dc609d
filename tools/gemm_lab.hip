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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32;

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
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
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

template <int BM, int BN, int LAB, int WPS>
float run(const float *A, const float *B, float *C, int M, int N, int K, int reps)
{
    const int tn = N / BN, tm = M / BM;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) gemm_kernel<BM, BN, LAB, WPS><<<tm * tn, 256>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) gemm_kernel<BM, BN, LAB, WPS><<<tm * tn, 256>>>(A, B, C, M, N, K, tn);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  BM=%d BN=%d WPS=%d LAB=%2d%s%s%s%s%s%s : %.3f ms  %.1f TF\n", BM, BN, WPS, LAB, (LAB & DEEP) ? " deep" : (LAB & SAME_TILE) ? " sametile" : "",
           (LAB & NO_GLOBAL) ? " -global" : "", (LAB & NO_STAGE) ? " -stage" : "",
           (LAB & NO_LDSREAD) ? " -ldsread" : "", (LAB & PRIO) ? " +prio" : "",
           (LAB & NO_BARRIER) ? " -barrier" : "", ms, 2.0 * M * N * K / ms / 1e9);
    return ms;
}

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
    run<128, 128, 0, 2>(A, B, C, M, N, K, R);
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
