// LDS-DMA helpers shared by the 8-wave bf16 kernels (rn_conv_wide.hip, rn_chain.hip): buffer
// descriptor, the buffer_load ... lds wave instruction, the counted wait + barrier, stamps.
#ifndef RN_LDS_DMA_H
#define RN_LDS_DMA_H

#include "rn_internal.h"

namespace rn_dma {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kOob = (int)0x80000000;  // >= num_records of every tensor we accept

// buffer descriptor in four SGPRs for the inline-asm DMA (raw buffer, no swizzle, 32-bit offsets)
__device__ __forceinline__ i32x4 make_srd(const void *ptr, int bytes)
{
    const unsigned long long a = (unsigned long long)ptr;
    i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    r[2] = bytes;
    r[3] = 0x00020000;
    return r;
}

// one LDS-DMA wave instruction: lane l fetches 16 bytes at (descriptor base + voff + soff) and
// the 64 x 16 bytes land at lds_dst + 16*l.  An out-of-range offset lands as zeros.  M0 carries
// the destination; it is compiler-reserved, so it is saved and restored inside the statement.
// s_nop 3: the scalar operands may come fresh from a v_readfirstlane (5 wait states to a VMEM
// read of an SGPR) and M0 needs one before the load.
__device__ __forceinline__ void dma16(int voff, i32x4 srd, int soff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 3\n\t"
                 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(srd), "s"(lds_dst), "s"(soff)
                 : "memory");
}

// the wave's DMA pieces of the tile about to be read have landed (N younger ones may stay in
// flight), its own LDS reads of the previous tile are back, then the block meets
template <int N>
__device__ __forceinline__ void wait_and_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void stamp(unsigned long long *buf, int slot)
{
    if (buf && threadIdx.x == 0) buf[(size_t)blockIdx.x * 16 + slot] = wall_clock64();
}

__device__ __forceinline__ void stamp_cycles(unsigned long long *buf, int slot)
{
    if (buf && threadIdx.x == 0) buf[(size_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memtime();
}

}  // namespace rn_dma

#endif
