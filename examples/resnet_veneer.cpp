// A whole ResNet through the C++ veneer (include/rn/*.hpp), one veneer call per reference op:
// the route a user of olehskip/resnet.c takes when only the include lines and the build
// line change.  Weights come from ./weights_bin/<state_dict key>
// (save_weights.py's format), the input from a [1,3,224,224] fp32 file; prints the
// reference's "max index is N" (cuda/inference/main.cu:250).
//
//   g++ -std=c++17 -Iinclude examples/resnet_veneer.cpp -Lresnet.c_amd -lrn_hip
//       -Wl,-rpath,$PWD/resnet.c_amd -o resnet_veneer
//   (cd dir_with_weights_bin && resnet_veneer 50 input.bin [logits_out.bin])
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "rn/nn.hpp"

namespace
{
struct ConvBn {
    Conv2d conv;
    BatchNorm2d bn;
    // y = bn(conv(x)), optionally followed by ReLU; y is allocated here
    FloatTensor operator()(FloatTensor &x, bool relu)
    {
        FloatTensor y(conv.getOutShape(x.shape()), Device::GPU);
        conv.forward(x, y);
        bn.forward(y, y);
        if (relu) reluForward(y, y);
        return y;
    }
};

ConvBn load_conv_bn(const std::string &conv, const std::string &bn, uint64_t cin, uint64_t cout,
                    uint64_t k, uint64_t stride, uint64_t pad)
{
    return ConvBn{Conv2d::loadWeightToCuda(conv, cin, cout, k, stride, pad),
                  BatchNorm2d::loadWeightToCuda(bn, cout)};
}

struct Bottleneck {
    ConvBn reduce, spatial, expand;
    std::unique_ptr<ConvBn> project;  // 1x1 shortcut of a stage's first block

    FloatTensor operator()(FloatTensor &x)
    {
        FloatTensor a = reduce(x, true);
        FloatTensor b = spatial(a, true);
        FloatTensor y = expand(b, false);
        if (project) {
            FloatTensor s = (*project)(x, false);
            addForward(y, s, y);
        } else {
            addForward(y, x, y);
        }
        reluForward(y, y);
        return y;
    }
};

struct Net {
    ConvBn stem;
    Pool2d maxpool{64, 3, 2, 1}, avgpool{2048, 7};
    std::vector<Bottleneck> blocks;
    Linear fc;
};

Net load_net(int depth)
{
    const int counts50[4] = {3, 4, 6, 3}, counts101[4] = {3, 4, 23, 3}, counts152[4] = {3, 8, 36, 3};
    const int *counts = depth == 50 ? counts50 : depth == 101 ? counts101 : counts152;
    Net net{load_conv_bn("conv1", "bn1", 3, 64, 7, 2, 3), {64, 3, 2, 1}, {2048, 7}, {},
            Linear::loadWeightToCuda("fc", 2048, 1000)};
    uint64_t in = 64;
    for (int stage = 0; stage < 4; ++stage) {
        const uint64_t mid = 64ull << stage, out = 4 * mid;
        for (int b = 0; b < counts[stage]; ++b) {
            const std::string p = "layer" + std::to_string(stage + 1) + "." + std::to_string(b) + ".";
            const uint64_t stride = (b == 0 && stage > 0) ? 2 : 1;
            Bottleneck blk{load_conv_bn(p + "conv1", p + "bn1", in, mid, 1, 1, 0),
                           load_conv_bn(p + "conv2", p + "bn2", mid, mid, 3, stride, 1),
                           load_conv_bn(p + "conv3", p + "bn3", mid, out, 1, 1, 0), nullptr};
            if (b == 0)
                blk.project = std::make_unique<ConvBn>(
                    load_conv_bn(p + "downsample.0", p + "downsample.1", in, out, 1, stride, 0));
            net.blocks.push_back(std::move(blk));
            in = out;
        }
    }
    return net;
}

FloatTensor forward(Net &net, FloatTensor &image)
{
    FloatTensor x = net.stem(image, true);
    FloatTensor pooled(net.maxpool.getOutShape(x.shape()), Device::GPU);
    net.maxpool.maxforward(x, pooled);
    x = std::move(pooled);
    for (Bottleneck &blk : net.blocks) {
        FloatTensor y = blk(x);
        x = std::move(y);
    }
    FloatTensor feat(net.avgpool.getOutShape(x.shape()), Device::GPU);
    net.avgpool.avgforward(x, feat);
    FloatTensor flat = feat.view(Shape({feat.shape()[0], 2048}));
    FloatTensor logits(net.fc.getOutShape(flat.shape()), Device::GPU);
    net.fc.forward(flat, logits);
    return logits;
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s 50|101|152 input.bin [logits_out.bin]\n", argv[0]);
        return 2;
    }
    const int depth = std::atoi(argv[1]);
    if (depth != 50 && depth != 101 && depth != 152) return 2;
    Net net = load_net(depth);
    FloatTensor flat = FloatTensor::loadToCuda(argv[2]);
    const uint64_t B = flat.numel() / (3 * 224 * 224);
    FloatTensor image = flat.view(Shape({B, 3, 224, 224}));
    FloatTensor logits = forward(net, image).cpu();
    if (argc > 3) logits.save(argv[3]);
    for (uint64_t b = 0; b < B; ++b) {
        const float *row = logits.data() + b * 1000;
        uint64_t best = 0;
        for (uint64_t i = 1; i < 1000; ++i)
            if (row[i] > row[best]) best = i;  // first maximum wins (main.cu:243-251)
        std::cout << "max index is " << best << "\n";
    }
    return 0;
}
