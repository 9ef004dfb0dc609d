// Compile-and-run check of the C++ veneer (include/rn/*.hpp): one bottleneck-style
// chain conv -> bn -> relu -> add written exactly as reference code would write it.
//   g++ -std=c++17 -Iinclude examples/veneer_smoke.cpp -Lresnet.c_amd -lrn_hip -Wl,-rpath,$PWD/resnet.c_amd
#include <cmath>
#include <cstdio>

#include "rn/nn.hpp"

int main()
{
    const uint64_t B = 2, C = 32, H = 6, W = 6;
    FloatTensor w_cpu(Shape({C, C, 3, 3}));
    for (uint64_t i = 0; i < w_cpu.numel(); ++i) w_cpu.data()[i] = 0.01f * float(i % 7) - 0.02f;
    Conv2d conv(w_cpu.cuda(), C, C, 3, 1, 1);

    auto vec = [&](float v) {
        FloatTensor t(Shape({C}));
        for (uint64_t i = 0; i < C; ++i) t.data()[i] = v;
        return t.cuda();
    };
    BatchNorm2d bn(vec(1.f), vec(0.5f), vec(0.f), vec(1.f), C);

    FloatTensor x_cpu(Shape({B, C, H, W}));
    for (uint64_t i = 0; i < x_cpu.numel(); ++i) x_cpu.data()[i] = float(i % 5) - 2.f;
    FloatTensor x = x_cpu.cuda();
    FloatTensor y(conv.getOutShape(x.shape()), Device::GPU);
    conv.forward(x, y);
    bn.forward(y, y);
    reluForward(y, y);
    addForward(y, x, y);
    FloatTensor out = y.cpu();
    double sum = 0;
    for (uint64_t i = 0; i < out.numel(); ++i) sum += out.data()[i];
    std::cout << "out shape " << out.shape() << " checksum " << sum << "\n";
    return std::isfinite(sum) ? 0 : 1;
}
