// rn/ops.hpp -- the seven kernel entry points of the reference's cuda/ops.cuh as plain
// functions with the same names and argument order.  Where reference code wrote
//     conv2dForwardKernel<<<blocks, block_size>>>(x, out, w, k, s, p, h_out, w_out, B, Cin, Cout, H, W);
// it now writes
//     conv2dForwardKernel(x, out, w, k, s, p, h_out, w_out, B, Cin, Cout, H, W);
// (launch geometry is the library's business).  Each call forwards to the C-ABI entry
// point that replaces that kernel and aborts on error like gpuAssert.
#ifndef RN_OPS_HPP
#define RN_OPS_HPP

#include "tensor.hpp"

inline uint64_t convOutputSize(uint64_t x, uint64_t kernel_size, uint64_t stride, uint64_t padding)
{
    return rn_conv_output_size(x, kernel_size, stride, padding);  // ops.cuh:9-13
}

inline void conv2dForwardKernel(float *inp, float *out, float *weight, uint64_t kernel_size,
                                uint64_t stride, uint64_t padding, uint64_t h_out, uint64_t w_out,
                                uint64_t B, uint64_t in_channels, uint64_t out_channels, uint64_t H,
                                uint64_t W)
{
    gpuErrchk(rn_conv2d_forward(rn::context(), inp, out, weight, kernel_size, stride, padding, h_out,
                                w_out, B, in_channels, out_channels, H, W));
}

inline void maxPool2dKernel(float *inp, float *out, uint64_t kernel_size, uint64_t stride,
                            uint64_t padding, uint64_t h_out, uint64_t w_out, uint64_t B,
                            uint64_t channels, uint64_t H, uint64_t W)
{
    gpuErrchk(rn_maxpool2d_forward(rn::context(), inp, out, kernel_size, stride, padding, h_out,
                                   w_out, B, channels, H, W));
}

inline void avgPool2dKernel(float *inp, float *out, uint64_t kernel_size, uint64_t stride,
                            uint64_t padding, uint64_t h_out, uint64_t w_out, uint64_t B,
                            uint64_t channels, uint64_t H, uint64_t W)
{
    gpuErrchk(rn_avgpool2d_forward(rn::context(), inp, out, kernel_size, stride, padding, h_out,
                                   w_out, B, channels, H, W));
}

inline void linearForwardKernel(float *inp, float *out, float *weight, float *bias, uint64_t B,
                                uint64_t in_features, uint64_t out_features)
{
    gpuErrchk(rn_linear_forward(rn::context(), inp, out, weight, bias, B, in_features, out_features));
}

inline void reluForwardKernel(float *inp, float *out, uint64_t N)
{
    gpuErrchk(rn_relu_forward(rn::context(), inp, out, N));
}

inline void batchNorm2dForwardKernel(float *inp, float *out, float *weight, float *bias, float *mean,
                                     float *var, uint64_t B, uint64_t C, uint64_t N)
{
    gpuErrchk(rn_batchnorm2d_forward(rn::context(), inp, out, weight, bias, mean, var, B, C, N));
}

inline void addForwardKernel(float *inp1, float *inp2, float *out, uint64_t N)
{
    gpuErrchk(rn_add_forward(rn::context(), inp1, inp2, out, N));
}

#endif  // RN_OPS_HPP
