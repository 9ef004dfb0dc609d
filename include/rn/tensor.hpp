// rn/tensor.hpp -- C++ veneer over the C-ABI (rn_hip.h) with the reference's names.
//
// Gives code written against olehskip/resnet.c's cuda/tensor.cuh + cuda/helpers.cuh
// the same vocabulary -- Device, Shape, Tensor<T>, FloatTensor, CEIL, gpuAssert -- on
// top of librn_hip.so.  Header-only, C++17, no HIP headers needed by the includer:
// everything device-side happens behind the C-ABI.
//
//   reference                         here
//   Shape::numel/as_tuple/<<          same                     (tensor.cuh:21-57)
//   Tensor(Device) empty tensor       same: shape {0}, !bool   (tensor.cuh:60-65)
//   Tensor(Shape, Device) alloc       malloc / rn_malloc       (tensor.cuh:67-94)
//   loadToCpu / loadToCuda / save     same file format         (tensor.cuh:126-163)
//   view (aliases storage)            same                     (tensor.cuh:165-170)
//   toDevice / cuda / cpu             rn_memcpy_h2d / d2h      (tensor.cuh:184-209)
//   data()                            same pointer; on a GPU tensor it first runs what the layer
//                                     classes recorded and gives the buffer its NCHW content back
//                                     (rn_observe): forward() calls are deferred, see rn_hip.h
//   gpuAssert(code, file, line, abort) same: message on stderr, then abort
//                                                              (helpers.cuh:13-22)
//   safeCudaMalloc(size)              rn_malloc, checked; -DDEBUG logs every allocation
//                                     and the running total    (helpers.cuh:24-35)
//   <cassert>, <iomanip>, <numeric>   included here as tensor.cuh:4-11 does, so a caller
//                                     that relied on them (main.cu:171,242) still compiles
//
// One process-wide context (device 0 unless rn::set_device() ran first) stands in
// for CUDA's implicit current device; multi-GPU hosts create one rn_ctx per thread
// through the C-ABI directly.
#ifndef RN_TENSOR_HPP
#define RN_TENSOR_HPP

#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../rn_hip.h"

#define CEIL(a, b) (((a) + (b)-1) / (b))

namespace rn
{

inline int &device_slot()
{
    static int device = 0;
    return device;
}
inline void set_device(int device) { device_slot() = device; }

inline rn_ctx *context()
{
    static rn_ctx *ctx = [] {
        rn_ctx *c = nullptr;
        const int st = rn_ctx_create(&c, device_slot(), nullptr);
        if (st != RN_OK) {
            std::cerr << "rn_ctx_create: " << rn_status_string(st) << "\n";
            std::abort();
        }
        rn_ctx_set_sync_each_op(c, 1);  // the reference synchronises after every op
        rn_ctx_set_weight_cache(c, 1);  // layers own their weights (nn.cuh:13): pack each once
        // forward() calls are recorded and run when a result is observed (cpu(), data(), a free), with
        // conv + in-place batch-norm / add / ReLU as one fused launch (rn_ctx_set_deferred, rn_hip.h);
        // RN_VENEER_LITERAL=1 in the environment: one launch per call, at once, as the reference does
        const char *literal = std::getenv("RN_VENEER_LITERAL");
        rn_ctx_set_deferred(c, (literal && literal[0] == '1') ? 0 : 1);
        return c;
    }();
    return ctx;
}

// the reference's error convention: message on stderr, then abort (helpers.cuh:13-22)
inline void check(int status, const char *file, int line, bool fatal = true)
{
    if (status == RN_OK) return;
    std::cerr << "GPUassert: " << rn_status_string(status) << " (" << rn_last_error(context())
              << ") " << file << " " << line << "\n";
    if (fatal) std::abort();
}

}  // namespace rn

// gpuErrchk / gpuAssert with the reference's signature (helpers.cuh:8-22); `code` is an
// rn_hip.h status instead of a cudaError_t
#define gpuErrchk(ans)                        \
    {                                         \
        gpuAssert((ans), __FILE__, __LINE__); \
    }
inline void gpuAssert(int code, const char *file, int line, bool abort = true)
{
    rn::check(code, file, line, abort);
}

// checked device allocation on the veneer's context (helpers.cuh:24-35)
inline void *safeCudaMalloc(uint64_t size)
{
    void *dest = nullptr;
    gpuErrchk(rn_malloc(rn::context(), &dest, size));
#ifdef DEBUG
    static uint64_t total = 0;
    total += size;
    std::cerr << "GPU allocate ptr: " << dest << ". Size: " << size << " bytes. Total: " << total
              << " bytes" << std::endl;
#endif
    return dest;
}

enum class Device { CPU, GPU };
enum class Layout { NCHW = RN_LAYOUT_NCHW, NHWC = RN_LAYOUT_NHWC };

class Shape : public std::vector<uint64_t>
{
public:
    using std::vector<uint64_t>::vector;

    uint64_t numel() const
    {
        if (empty()) std::abort();
        uint64_t n = 1;
        for (uint64_t d : *this) n *= d;
        return n;
    }

    template <std::size_t N>
    auto as_tuple() const
    {
        if (size() != N) std::abort();
        return unpack(std::make_index_sequence<N>{});
    }

    friend std::ostream &operator<<(std::ostream &os, const Shape &s)
    {
        os << "(";
        for (std::size_t i = 0; i < s.size(); ++i) os << (i ? ", " : "") << s[i];
        return os << ")";
    }

private:
    template <std::size_t... I>
    auto unpack(std::index_sequence<I...>) const
    {
        return std::make_tuple((*this)[I]...);
    }
};

template <class T>
struct Tensor {
    explicit Tensor(Device dev) : device(dev), shape_({0}) {}

    Tensor(Shape shape, Device dev = Device::CPU) : device(dev), shape_(std::move(shape))
    {
        if (shape_.empty()) std::abort();
        if (numel() == 0) return;
        if (device == Device::CPU) {
            storage_ = std::shared_ptr<T>(static_cast<T *>(std::malloc(size())), std::free);
        } else {
            void *p = safeCudaMalloc(size());
            storage_ = std::shared_ptr<T>(static_cast<T *>(p),
                                          [](T *q) { rn_free(rn::context(), q); });
        }
    }

    Tensor(Tensor &&other) noexcept
        : device(other.device), layout(other.layout), shape_(other.shape_), storage_(other.storage_)
    {
    }
    Tensor(const Tensor &) = delete;
    void operator=(const Tensor &) = delete;
    void operator=(Tensor &&other)
    {
        if (device != other.device) std::abort();
        storage_ = std::move(other.storage_);
        shape_ = std::move(other.shape_);
        layout = other.layout;
        other.storage_.reset();
        other.shape_ = Shape({0});
    }

    static Tensor loadToCpu(const std::string &file_name)
    {
        std::ifstream f(file_name, std::ios::binary | std::ios::ate);
        if (!f) {
            std::cerr << "Can't open " << file_name << std::endl;
            std::abort();
        }
        const std::streamoff bytes = f.tellg();
        const uint64_t n = static_cast<uint64_t>(bytes) / sizeof(T);
        if (n == 0) std::abort();
        Tensor out(Shape({n}), Device::CPU);
        f.seekg(0);
        f.read(reinterpret_cast<char *>(out.data()), static_cast<std::streamsize>(n * sizeof(T)));
        if (!f) std::abort();
        return out;
    }

    static Tensor loadToCuda(const std::string &file_name) { return loadToCpu(file_name).cuda(); }

    void save(const std::string &file_name) const
    {
        if (device != Device::CPU) std::abort();
        std::ofstream f(file_name, std::ios::binary);
        f.write(reinterpret_cast<const char *>(data()), static_cast<std::streamsize>(size()));
        if (!f) std::abort();
    }

    Tensor view(Shape new_shape) const
    {
        if (new_shape.empty() || new_shape.numel() != numel()) std::abort();
        Tensor v(device);
        v.shape_ = std::move(new_shape);
        v.storage_ = storage_;
        v.layout = layout;
        return v;
    }

    uint64_t numel() const { return shape_.numel(); }
    uint64_t size() const { return numel() * sizeof(T); }

    Tensor toDevice(Device to) const
    {
        Tensor ret(shape_, to);
        ret.layout = layout;
        if (device == Device::CPU && to == Device::GPU) {
            gpuErrchk(rn_memcpy_h2d(rn::context(), ret.raw(), raw(), size()));
        } else if (device == Device::GPU && to == Device::CPU) {
            gpuErrchk(rn_memcpy_d2h(rn::context(), ret.raw(), raw(), size()));
        } else {
            throw std::runtime_error("Unsupported device transfer combination");
        }
        return ret;
    }
    Tensor cuda() const { return toDevice(Device::GPU); }
    Tensor cpu() const { return toDevice(Device::CPU); }

    explicit operator bool() const { return static_cast<bool>(storage_); }
    const Shape &shape() const { return shape_; }
    // the pointer for the caller's own use: what was recorded runs and the buffer holds its NCHW content
    // (rn_observe); the layer classes hand raw() to the library, which knows what each buffer holds
    T *data() const
    {
        if (device == Device::GPU && storage_) gpuErrchk(rn_observe(rn::context(), storage_.get()));
        return storage_.get();
    }
    T *raw() const { return storage_.get(); }

    const Device device;
    Layout layout = Layout::NCHW;

private:
    Shape shape_;
    std::shared_ptr<T> storage_;
};

using FloatTensor = Tensor<float>;

#endif  // RN_TENSOR_HPP
