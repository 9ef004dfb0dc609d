// rn/nn.hpp -- layer wrappers with the reference's class and member names
// (cuda/nn.cuh:8-136): Conv2d, BatchNorm2d, Pool2d, Linear, reluForward, addForward.
// forward(x, out) never allocates and runs in the layout of x (NCHW unless the
// caller tagged the tensor NHWC); one C-ABI call per forward.  The veneer's context is
// deferred (rn_ctx_set_deferred): those calls are recorded and run when a result is
// observed, a convolution together with the in-place batch-norm / add / ReLU behind it.
#ifndef RN_NN_HPP
#define RN_NN_HPP

#include <string>

#include "ops.hpp"

namespace rn
{
inline const std::string &weights_dir()
{
    static const std::string dir = "weights_bin/";  // nn.cuh:21,58-61,113,117
    return dir;
}
inline void use_layout(const FloatTensor &x)
{
    gpuErrchk(rn_ctx_set_layout(context(), static_cast<int>(x.layout)));
}
inline void require(bool ok)
{
    if (!ok) std::abort();  // the reference asserts
}
}  // namespace rn

class Conv2d
{
public:
    Conv2d(FloatTensor w, uint64_t in_c, uint64_t out_c, uint64_t k, uint64_t s = 1, uint64_t p = 0)
        : weight(std::move(w)), in_channels(in_c), out_channels(out_c), kernel_size(k), stride(s),
          padding(p)
    {
    }

    static Conv2d loadWeightToCuda(std::string name, uint64_t in_c, uint64_t out_c, uint64_t k,
                                   uint64_t s = 1, uint64_t p = 0)
    {
        return Conv2d(FloatTensor::loadToCuda(rn::weights_dir() + name + ".weight")
                          .view(Shape({out_c, in_c, k, k})),
                      in_c, out_c, k, s, p);
    }

    Shape getOutShape(const Shape &x) const
    {
        rn::require(x.size() == 4 && x[1] == in_channels);
        return Shape({x[0], out_channels, convOutputSize(x[2], kernel_size, stride, padding),
                      convOutputSize(x[3], kernel_size, stride, padding)});
    }

    void forward(FloatTensor &x, FloatTensor &out)
    {
        const auto [B, C, H, W] = x.shape().as_tuple<4>();
        const auto [ob, oc, h_out, w_out] = out.shape().as_tuple<4>();
        (void)C, (void)ob, (void)oc;
        rn::use_layout(x);
        out.layout = x.layout;
        conv2dForwardKernel(x.raw(), out.raw(), weight.raw(), kernel_size, stride, padding, h_out,
                            w_out, B, in_channels, out_channels, H, W);
    }

    FloatTensor weight;
    const uint64_t in_channels, out_channels, kernel_size, stride, padding;
};

class BatchNorm2d
{
public:
    BatchNorm2d(FloatTensor &&w, FloatTensor &&b, FloatTensor &&m, FloatTensor &&v, uint64_t channels)
        : weight(std::move(w)), bias(std::move(b)), mean(std::move(m)), var(std::move(v)),
          channels_num(channels)
    {
        const Shape want({channels});
        rn::require(weight.shape() == want && bias.shape() == want && mean.shape() == want &&
                    var.shape() == want);
    }

    static BatchNorm2d loadWeightToCuda(std::string name, uint64_t channels)
    {
        const std::string base = rn::weights_dir() + name;
        return BatchNorm2d(FloatTensor::loadToCuda(base + ".weight"),
                           FloatTensor::loadToCuda(base + ".bias"),
                           FloatTensor::loadToCuda(base + ".running_mean"),
                           FloatTensor::loadToCuda(base + ".running_var"), channels);
    }

    void forward(FloatTensor &x, FloatTensor &out)
    {
        const auto [B, C, h, w] = x.shape().as_tuple<4>();
        (void)C;
        rn::use_layout(x);
        out.layout = x.layout;
        batchNorm2dForwardKernel(x.raw(), out.raw(), weight.raw(), bias.raw(), mean.raw(), var.raw(), B,
                                 channels_num, h * w);
    }

    FloatTensor weight, bias, mean, var;
    const uint64_t channels_num;
};

class Pool2d
{
public:
    Pool2d(uint64_t c, uint64_t k, uint64_t s = 1, uint64_t p = 0)
        : channels(c), kernel_size(k), stride(s), padding(p)
    {
    }

    uint64_t outSideSize(uint64_t side) const
    {
        return convOutputSize(side, kernel_size, stride, padding);
    }

    Shape getOutShape(const Shape &x) const
    {
        rn::require(x.size() == 4 && x[1] == channels);
        return Shape({x[0], channels, outSideSize(x[2]), outSideSize(x[3])});
    }

    void maxforward(FloatTensor &x, FloatTensor &out) { run(true, x, out); }
    void avgforward(FloatTensor &x, FloatTensor &out) { run(false, x, out); }

    const uint64_t channels, kernel_size, stride, padding;

private:
    void run(bool is_max, FloatTensor &x, FloatTensor &out)
    {
        const auto [B, C, H, W] = x.shape().as_tuple<4>();
        rn::use_layout(x);
        out.layout = x.layout;
        auto *fn = is_max ? maxPool2dKernel : avgPool2dKernel;
        fn(x.raw(), out.raw(), kernel_size, stride, padding, out.shape().at(2), out.shape().at(3), B,
           C, H, W);
    }
};

class Linear
{
public:
    Linear(FloatTensor w, FloatTensor b, uint64_t in_f, uint64_t out_f)
        : weight(std::move(w)), bias(std::move(b)), in_features(in_f), out_features(out_f)
    {
        rn::require(weight.shape() == Shape({out_f, in_f}) && bias.shape() == Shape({out_f}));
    }

    static Linear loadWeightToCuda(std::string name, uint64_t in_f, uint64_t out_f)
    {
        const std::string base = rn::weights_dir() + name;
        return Linear(FloatTensor::loadToCuda(base + ".weight").view(Shape({out_f, in_f})),
                      FloatTensor::loadToCuda(base + ".bias").view(Shape({out_f})), in_f, out_f);
    }

    Shape getOutShape(const Shape &x) const
    {
        rn::require(x.size() == 2 && x[1] == in_features);
        return Shape({x.at(0), out_features});
    }

    void forward(FloatTensor &x, FloatTensor &out)
    {
        linearForwardKernel(x.raw(), out.raw(), weight.raw(), bias.raw(), x.shape().at(0),
                            in_features, out_features);
    }

    FloatTensor weight, bias;
    const uint64_t in_features, out_features;
};

inline void reluForward(FloatTensor &x, FloatTensor &out)
{
    rn::require(x.shape() == out.shape());
    out.layout = x.layout;
    reluForwardKernel(x.raw(), out.raw(), x.numel());
}

inline void addForward(FloatTensor &a, FloatTensor &b, FloatTensor &out)
{
    rn::require(a.shape() == b.shape() && a.shape() == out.shape() && a.layout == b.layout);
    out.layout = a.layout;
    addForwardKernel(a.raw(), b.raw(), out.raw(), a.numel());
}

#endif  // RN_NN_HPP
