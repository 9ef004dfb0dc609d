/*
 * rn_hip.h -- C-ABI of the MI355X (gfx950) ResNet forward path.
 *
 * Plain C: pointers, sizes and opaque handles only.  Every entry point returns
 * an int status (RN_OK == 0) and never aborts; rn_last_error() gives the text.
 * Device pointers are ordinary HIP device addresses (hipMalloc / rn_malloc /
 * any framework's allocator on the same device).
 *
 * Each op entry point takes the arguments of the reference kernel it replaces,
 * in the same order, preceded by the context:
 *
 *   rn_conv_output_size      <- convOutputSize            cuda/ops.cuh:9-13
 *   rn_conv2d_forward        <- conv2dForwardKernel       cuda/ops.cuh:15-18  + Conv2d::forward      cuda/nn.cu:3-16
 *   rn_maxpool2d_forward     <- maxPool2dKernel           cuda/ops.cuh:19-21  + Pool2d::maxforward   cuda/nn.cu:43-53
 *   rn_avgpool2d_forward     <- avgPool2dKernel           cuda/ops.cuh:22-24  + Pool2d::avgforward   cuda/nn.cu:31-41
 *   rn_linear_forward        <- linearForwardKernel       cuda/ops.cuh:25-26  + Linear::forward      cuda/nn.cu:55-64
 *   rn_relu_forward          <- reluForwardKernel         cuda/ops.cuh:27     + reluForward          cuda/nn.cu:66-75
 *   rn_batchnorm2d_forward   <- batchNorm2dForwardKernel  cuda/ops.cuh:29-31  + BatchNorm2d::forward cuda/nn.cu:18-29
 *   rn_add_forward           <- addForwardKernel          cuda/ops.cuh:32     + addForward           cuda/nn.cu:77-87
 *
 * Buffers mirror Tensor<T> (cuda/tensor.cuh:59-245):
 *   rn_malloc / rn_free          <- safeCudaMalloc / cudaFree deleter   helpers.cuh:24-35, tensor.cuh:81-86
 *   rn_memcpy_h2d / rn_memcpy_d2h<- Tensor::toDevice                    tensor.cuh:184-199
 *   rn_load_f32_file             <- Tensor::loadToCuda                  tensor.cuh:126-152
 *   rn_save_f32_file             <- Tensor::save                        tensor.cuh:154-163
 *   rn_sync                      <- cudaDeviceSynchronize + gpuAssert   nn.cu:14-15
 *
 * The model entry points replace the reference driver
 * (cuda/inference/main.cu:53-226,243-251): createLayer/createResnet152,
 * layerForward/resnet152Forward and the host argmax.
 *
 * Layout.  The reference is NCHW everywhere.  With the context in
 * RN_LAYOUT_NCHW (default) every op reads and writes exactly what the
 * reference kernel does: 1x1 / padding-0 convolutions on an NCHW-native
 * contraction that takes the OIHW weight as it is, other convolutions through
 * a transpose of their input into context scratch, batch-norm and the
 * network's two pools on NCHW forms of their own.  The engine itself runs
 * NHWC: in RN_LAYOUT_NHWC the same entry points take [B,H,W,C] activations
 * (shape arguments unchanged), and rn_conv2d_nhwc_forward takes weights
 * pre-packed by rn_conv2d_pack_weight plus an optional fused epilogue.
 *
 * Aliasing: out == inp is allowed for relu, batchnorm2d and add (out may alias
 * inp1), as the reference driver uses them (main.cu:138,145-146,162-163).
 * conv, pool and linear must not alias.
 *
 * Threading: one context per host thread; a context = (device, stream,
 * scratch).  No global mutable state.  A host thread may own contexts on
 * several devices: every entry point makes its context's device the calling
 * thread's current HIP device before it touches the device.
 */
#ifndef RN_HIP_H
#define RN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define RN_API __attribute__((visibility("default")))
#else
#define RN_API
#endif

typedef struct rn_ctx rn_ctx;
typedef struct rn_model rn_model;
typedef struct rn_event rn_event;

enum {
    RN_OK = 0,
    RN_ERR_INVALID = 1,     /* bad argument / shape precondition (reference: assert) */
    RN_ERR_HIP = 2,         /* HIP runtime error (reference: gpuAssert -> abort)      */
    RN_ERR_IO = 3,          /* file could not be opened / short read                  */
    RN_ERR_NOMEM = 4,
    RN_ERR_UNSUPPORTED = 5
};

enum { RN_LAYOUT_NCHW = 0, RN_LAYOUT_NHWC = 1 };

/* element types of the engine-side ("_dt") entry points; accumulation is always fp32 */
enum { RN_DTYPE_F32 = 0, RN_DTYPE_BF16 = 1 };

/* forward modes of rn_model_forward */
enum {
    RN_FWD_REFERENCE_OPS = 0, /* one kernel per reference op, same sequence as main.cu:168-226 */
    RN_FWD_FUSED = 1          /* BN / ReLU / residual add folded into the conv epilogue        */
};

/* ---- context ----------------------------------------------------------- */
RN_API int rn_ctx_create(rn_ctx **out, int device, void *hip_stream /* NULL: own stream */);
RN_API int rn_ctx_destroy(rn_ctx *ctx);
RN_API int rn_ctx_set_layout(rn_ctx *ctx, int layout);
RN_API int rn_ctx_get_layout(const rn_ctx *ctx);
/* 1: synchronise and check after every op, like the reference (nn.cu:14-15). Default 0. */
RN_API int rn_ctx_set_sync_each_op(rn_ctx *ctx, int on);
/* Deferred execution of the reference's op-by-op call sequence (main.cu:127-166: conv.forward,
 * bn.forward in place, [addForward in place], reluForward in place -- separate launches, each a pass
 * over its tensor).  on = 1: with the context in its NCHW layout the seven reference entry points
 * (rn_conv2d_forward, rn_batchnorm2d_forward, rn_relu_forward, rn_add_forward, rn_maxpool2d_forward,
 * rn_avgpool2d_forward, rn_linear_forward) only RECORD their call and return.  The recorded list runs
 * when something is observed -- rn_memcpy_d2h / rn_save_f32_file, rn_sync, rn_flush, rn_observe,
 * rn_event_record, rn_free, a write into a buffer it names, or any other entry point -- and then a
 * convolution with the in-place batch-norm / add / ReLU that follow it on its output buffer is ONE
 * launch of the NHWC contraction with the fused epilogue (folded batch-norm, residual, ReLU), which
 * writes NHWC into the caller's own output buffer.  The context remembers which caller buffers hold
 * NHWC: ops that follow run on them as NHWC (no transposes between the first and the last op of a
 * network), and a buffer gets its NCHW content back only when it is observed (a whole-tensor
 * rn_memcpy_d2h transposes on its way out; rn_observe, partial reads, device-to-device copies and
 * entry points outside the seven rewrite the buffer in place first).  Every buffer named as an output
 * holds its value once the list has run: only in-place chains are folded, nothing is skipped.  Where
 * such a group is conv3 + bn + add + ReLU of a 64-channel bottleneck block and the next recorded group
 * is conv1 + bn + ReLU of the following block on that output, both run as one launch
 * (rn_conv_chain_forward_dt: the block output is written on its way to conv1; the same bits as the two
 * launches; RN_DEFER_CHAINS=0 in the environment keeps them apart).  The projection shortcut of a
 * stage's first block (a convolution of the same output with its in-place batch-norm, main.cu:131-137)
 * may stand between the two: it then runs right behind that launch, which nothing can observe as long as
 * neither group names a buffer the other writes (checked; otherwise the call order is kept).
 * Contract for the caller: device memory it handed to the seven ops is read and written by other
 * means (own kernels, hipMemcpy) only after rn_observe(ctx, ptr) -- the C++ veneer's Tensor::data()
 * does that; weights and batch-norm parameters are cached in packed / folded form per buffer and the
 * cache follows writes made through rn_* calls only (like rn_ctx_set_weight_cache).
 * Numerics: the folded batch-norm is one fp32 fmaf per element instead of ops.cu:150's double
 * expression (<= 2e-5 on ResNet logits; the same epilogue as RN_FWD_FUSED).  on = 0 (default): every
 * call launches at once on NCHW tensors, the parity baseline; switching off runs what is recorded and
 * gives every buffer its NCHW content back. */
RN_API int rn_ctx_set_deferred(rn_ctx *ctx, int on);
RN_API int rn_ctx_get_deferred(const rn_ctx *ctx);
/* run the recorded ops now (asynchronously, on the context's stream) */
RN_API int rn_flush(rn_ctx *ctx);
/* run the recorded ops and make the caller buffer at dev_ptr hold its NCHW content */
RN_API int rn_observe(rn_ctx *ctx, const void *dev_ptr);
/* counters of the deferred route: ops recorded and not yet run, caller buffers currently NHWC, launches
 * with a folded chain, literal launches, layout passes (transposes) so far; any pointer may be NULL */
RN_API int rn_ctx_deferred_stats(const rn_ctx *ctx, uint64_t *pending_ops, uint64_t *nhwc_buffers,
                                 uint64_t *fused_launches, uint64_t *literal_launches,
                                 uint64_t *transposes);
/* Tile shape of the contraction kernel: 0 = chosen per launch (default), 1..N = force
 * candidate i (all candidates give bit-identical results; used by rn_model_tune). */
/* rn_conv2d_forward (the reference's OIHW / NCHW signature) re-packs the weight into the
 * engine's K-major panel on every call (not for 1x1 convolutions on NCHW tensors: their kernel
 * reads the OIHW buffer itself).  With the cache on, the panel is packed once per
 * (weight buffer, shape) and reused; entries die when the buffer is rn_free'd or written by
 * rn_memcpy_h2d / rn_memcpy_d2d / rn_memset.  A caller that turns it on promises not to change
 * a weight buffer by any other means (the reference's layers own their weights and never
 * do: nn.cuh:13,47-48,104).  Off by default; the C++ veneer turns it on. */
RN_API int rn_ctx_set_weight_cache(rn_ctx *ctx, int on);
/* kernel launches issued through this context so far (a contraction whose tail tiles are cut
 * into K chunks is two launches: the pieces and the kernel that adds them) */
RN_API uint64_t rn_ctx_launch_count(const rn_ctx *ctx);
RN_API int rn_conv_tile_candidates(void);
RN_API int rn_ctx_set_conv_tile(rn_ctx *ctx, int candidate);
/* Latency mode: max_splits > 1 lets a contraction whose output tiles cannot fill the chip
 * (small batches: B = 1 has 8 tiles of 64x64 in layer4's 3x3 convolutions, on 256 CUs) split
 * its K loop over up to max_splits blocks per tile; partial sums meet in context scratch and
 * a second kernel adds them in split order and applies the epilogue.  Deterministic, but the
 * summation order (and so the last bits) differs from the unsplit launch: 0 (default) keeps
 * results independent of the batch size.  Range 0..64. */
RN_API int rn_ctx_set_split_k(rn_ctx *ctx, int max_splits);
/* Fused stem + max-pool (rn_stem_pool_forward_dt): a block walks `items` consecutive items (four
 * stem rows = two pooled rows each) of one image; 0 (default) = chosen per launch so that the
 * blocks fill the device's CUs in the fewest rounds.  The bits do not depend on it: a block that
 * starts inside an image computes the one stem row above its segment once more. */
RN_API int rn_ctx_set_stem_items(rn_ctx *ctx, int items);
/* Tile order of the fp32 / bf16 4-wave contraction over the 8 XCDs (each with an L2 of its own): the tiles
 * are dealt as contiguous ranges of an order that keeps an XCD to one of `groups` groups of N tiles -- its
 * slice of the weight panel stays in that L2 while the M panels go by, the input is fetched by `groups`
 * XCDs.  0 (default): chosen per launch from the sizes of the input and of the weight panel; 1: M panel
 * major (an XCD reads its input rows once and streams the whole weight panel); 2 / 4 / 8: forced (also
 * RN_XCD_NGROUPS in the environment at context creation).  Changes which block computes a tile, no bit. */
RN_API int rn_ctx_set_xcd_groups(rn_ctx *ctx, int groups);
/* rn_conv2d_forward on NCHW tensors (RN_LAYOUT_NCHW, not deferred), kernel_size > 1: 0 = transpose the input into
 * scratch and run the NHWC contraction (its epilogue writes NCHW); 2 = gather the taps from the channel planes
 * (rn_conv_nchw.hip) wherever the shape is eligible (in_channels % 32 == 0, kernel_size <= 7); 1 (default) = gather
 * on planes of 2048 pixels or more, where it measures faster (tools/nchw_bench.py).  Also RN_NCHW_TAPS in the
 * environment at context creation.  Changes the route, no bit. */
RN_API int rn_ctx_set_nchw_taps(rn_ctx *ctx, int mode);
/* Diagnostics: device buffer of 16 x uint64 per block that the contraction kernel fills with
 * wall-clock and shader-clock stamps of its phases (tools/conv_stamps.py); NULL (default) = off. */
RN_API int rn_ctx_set_debug_stamps(rn_ctx *ctx, void *dev_buffer);
RN_API void *rn_ctx_stream(rn_ctx *ctx);
RN_API int rn_ctx_device(const rn_ctx *ctx);
RN_API int rn_sync(rn_ctx *ctx);
RN_API const char *rn_last_error(const rn_ctx *ctx);
RN_API const char *rn_status_string(int status);
RN_API int rn_device_count(int *count);
/* Where a device sits in the host: PCI address ("0000:c1:00.0"), the NUMA node of its slot and the
 * CPUs local to it as Linux lists them ("0-47,96-143"); -1 / "" where sysfs does not say.  The
 * host threads of rn_shard_* run on those cores. */
RN_API int rn_device_locality(int device, char *pci_bus_id, uint64_t pci_cap, int *numa_node,
                              char *cpulist, uint64_t cpulist_cap);
RN_API const char *rn_version(void);

/* ---- buffers ------------------------------------------------------------ */
RN_API int rn_malloc(rn_ctx *ctx, void **dev_ptr, uint64_t bytes);
RN_API int rn_free(rn_ctx *ctx, void *dev_ptr);
RN_API int rn_memcpy_h2d(rn_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
RN_API int rn_memcpy_d2h(rn_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
RN_API int rn_memcpy_d2d(rn_ctx *ctx, void *dev_dst, const void *dev_src, uint64_t bytes);
RN_API int rn_memset(rn_ctx *ctx, void *dev_ptr, int byte_value, uint64_t bytes);
/* whole file -> new device buffer; numel = file_size / 4 (tensor.cuh:138) */
RN_API int rn_load_f32_file(rn_ctx *ctx, const char *path, float **dev_ptr, uint64_t *numel);
RN_API int rn_save_f32_file(rn_ctx *ctx, const char *path, const float *dev_ptr, uint64_t numel);

/* ---- timing (HIP events on the context's stream) ------------------------ */
RN_API int rn_event_create(rn_ctx *ctx, rn_event **out);
RN_API int rn_event_destroy(rn_event *ev);
RN_API int rn_event_record(rn_ctx *ctx, rn_event *ev);
RN_API int rn_event_elapsed_ms(rn_event *start, rn_event *stop, float *ms); /* syncs on stop */

/* ---- the seven reference ops -------------------------------------------- */
RN_API uint64_t rn_conv_output_size(uint64_t x, uint64_t kernel_size, uint64_t stride,
                                    uint64_t padding);
RN_API int rn_conv2d_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                             uint64_t kernel_size, uint64_t stride, uint64_t padding,
                             uint64_t h_out, uint64_t w_out, uint64_t B, uint64_t in_channels,
                             uint64_t out_channels, uint64_t H, uint64_t W);
RN_API int rn_maxpool2d_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t kernel_size,
                                uint64_t stride, uint64_t padding, uint64_t h_out, uint64_t w_out,
                                uint64_t B, uint64_t channels, uint64_t H, uint64_t W);
RN_API int rn_avgpool2d_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t kernel_size,
                                uint64_t stride, uint64_t padding, uint64_t h_out, uint64_t w_out,
                                uint64_t B, uint64_t channels, uint64_t H, uint64_t W);
RN_API int rn_linear_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                             const float *bias /* nullable */, uint64_t B, uint64_t in_features,
                             uint64_t out_features);
RN_API int rn_relu_forward(rn_ctx *ctx, const float *inp, float *out, uint64_t N);
RN_API int rn_batchnorm2d_forward(rn_ctx *ctx, const float *inp, float *out, const float *weight,
                                  const float *bias, const float *mean, const float *var,
                                  uint64_t B, uint64_t C, uint64_t N);
RN_API int rn_add_forward(rn_ctx *ctx, const float *inp1, const float *inp2, float *out,
                          uint64_t N);
/* host argmax of main.cu:243-251 as a device op: first maximum wins. idx is a device buffer. */
RN_API int rn_argmax_forward(rn_ctx *ctx, const float *logits, uint64_t *idx, uint64_t B,
                             uint64_t classes);

/* ---- layout converters and weight packing -------------------------------- */
RN_API int rn_nchw_to_nhwc(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C,
                           uint64_t H, uint64_t W);
RN_API int rn_nhwc_to_nchw(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C,
                           uint64_t H, uint64_t W);
/* channels of the engine-side input image for a convolution: in_channels, or 4 when
 * in_channels < 4 (the 3-channel stem reads a [B,H,W,4] zero-padded image). */
RN_API uint64_t rn_conv2d_input_channels(uint64_t in_channels);
/* NCHW [B,C,H,W] -> NHWC with the channel dimension zero-padded to Cpad */
RN_API int rn_nchw_to_nhwc_pad(rn_ctx *ctx, const float *src, float *dst, uint64_t B, uint64_t C,
                               uint64_t H, uint64_t W, uint64_t Cpad);
RN_API uint64_t rn_conv2d_packed_weight_numel(uint64_t in_channels, uint64_t out_channels,
                                              uint64_t kernel_size);
/* OIHW (the weights_bin file order) -> K-major panel [Cout][kh][kw][Cin] the GEMM reads */
RN_API int rn_conv2d_pack_weight(rn_ctx *ctx, const float *weight_oihw, float *packed,
                                 uint64_t in_channels, uint64_t out_channels,
                                 uint64_t kernel_size);
/* scale = w / sqrt(var + 1e-5), shift = b - mean * scale, evaluated in double */
RN_API int rn_batchnorm2d_fold(rn_ctx *ctx, const float *weight, const float *bias,
                               const float *mean, const float *var, float *scale, float *shift,
                               uint64_t C);

typedef struct rn_epilogue {
    const float *scale;    /* per out-channel multiplier, NULL = 1 */
    const float *shift;    /* per out-channel addend,     NULL = 0 */
    const void *residual;  /* NHWC tensor of the output's shape and element type, added after
                              scale/shift; NULL = none */
    int relu;              /* apply max(x, 0) last */
} rn_epilogue;

/* NHWC in / NHWC out, packed weights, optional fused epilogue.  The input must have
 * rn_conv2d_input_channels(in_channels) floats per pixel. */
RN_API int rn_conv2d_nhwc_forward(rn_ctx *ctx, const float *inp, float *out,
                                  const float *packed_weight, uint64_t kernel_size,
                                  uint64_t stride, uint64_t padding, uint64_t h_out,
                                  uint64_t w_out, uint64_t B, uint64_t in_channels,
                                  uint64_t out_channels, uint64_t H, uint64_t W,
                                  const rn_epilogue *epilogue /* nullable */);

/* ---- element-type tagged engine entry points (bf16 storage, fp32 accumulate) ---------
 * The fp32 functions above are the dtype == RN_DTYPE_F32 case of these.  bf16 tensors
 * are NHWC only, 16-byte aligned, convolution in_channels a multiple of 64 (or the
 * small-Cin stem form, in_channels <= 4 and kernel_size <= 8, reading a 4-channel image that
 * carries its own zero border: padding = 0, even stride and width; its K tile is 8 pixels of
 * each of two consecutive image rows, so the 7x7 stem is 4 K tiles).                 */
RN_API uint64_t rn_conv2d_packed_weight_numel_dt(int dtype, uint64_t in_channels,
                                                 uint64_t out_channels, uint64_t kernel_size);
/* fp32 OIHW weights (the weights_bin order) -> K-major panel of `dtype` */
RN_API int rn_conv2d_pack_weight_dt(rn_ctx *ctx, int dtype, const float *weight_oihw, void *packed,
                                    uint64_t in_channels, uint64_t out_channels,
                                    uint64_t kernel_size);
/* fp32 NCHW [B,C,H,W] -> `dtype` NHWC [B, H+2*border, W+2*border, Cpad], zeros in the border
 * and in channels >= C */
RN_API int rn_nchw_to_nhwc_pad_dt(rn_ctx *ctx, int dtype, const float *src, void *dst, uint64_t B,
                                  uint64_t C, uint64_t H, uint64_t W, uint64_t Cpad,
                                  uint64_t border);
/* inp/packed_weight of `dtype`, out and epilogue->residual of `out_dtype` */
RN_API int rn_conv2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, int out_dtype, const void *inp,
                                     void *out, const void *packed_weight, uint64_t kernel_size,
                                     uint64_t stride, uint64_t padding, uint64_t h_out,
                                     uint64_t w_out, uint64_t B, uint64_t in_channels,
                                     uint64_t out_channels, uint64_t H, uint64_t W,
                                     const rn_epilogue *epilogue /* nullable */);
RN_API int rn_maxpool2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, const void *inp, void *out,
                                        uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                        uint64_t h_out, uint64_t w_out, uint64_t B,
                                        uint64_t channels, uint64_t H, uint64_t W);
RN_API int rn_avgpool2d_nhwc_forward_dt(rn_ctx *ctx, int dtype, const void *inp, void *out,
                                        uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                        uint64_t h_out, uint64_t w_out, uint64_t B,
                                        uint64_t channels, uint64_t H, uint64_t W);

/* ---- model (main.cu driver) ---------------------------------------------- */
/* arch: 50, 101 or 152 (block counts 3/4/6/3, 3/4/23/3, 3/8/36/3). */
RN_API int rn_model_create(rn_ctx *ctx, rn_model **out, int arch);
RN_API int rn_model_destroy(rn_model *m);
/* state_dict key -> host data; numel must match the layer table. */
RN_API int rn_model_set_tensor(rn_model *m, const char *key, const float *host_data,
                               uint64_t numel);
/* read every tensor from dir/<key> (the reference's weights_bin/ directory) */
RN_API int rn_model_load_dir(rn_model *m, const char *weights_dir);
/* storage type of activations and packed weights inside the model: RN_DTYPE_F32 (default)
 * or RN_DTYPE_BF16 (fused mode only).  Input images and logits stay fp32.  Call before
 * rn_model_finalize. */
RN_API int rn_model_set_dtype(rn_model *m, int dtype);
/* upload-side work done once: pack conv weights, fold batch-norms */
RN_API int rn_model_finalize(rn_model *m);
/* names of the tensors the loader expects, one per call; returns NULL past the end */
RN_API const char *rn_model_tensor_key(const rn_model *m, uint64_t index, uint64_t *numel);
/* input: device NCHW [B,3,224,224]; logits: device [B,1000]. Asynchronous on the stream.
 *
 * Fixed geometry of the model driver (the op entry points above are general; the driver, like
 * the reference's -- main.cu:230 hard-codes {1, 3, 224, 224} -- is not):
 *   - images are 3 x 224 x 224 fp32, NCHW; there is no size argument, so a buffer of another
 *     geometry cannot be expressed: callers that read files check the element count first
 *     (rn_infer does and reports RN_ERR_UNSUPPORTED's text for anything but B*3*224*224 floats);
 *   - 1000 classes, bottleneck depths 50 / 101 / 152;
 *   - any B >= 1: the kernels address a tensor with 32-bit byte offsets (2^29 fp32 elements; the
 *     stem output of 669 images is the first to pass it), so a batch runs as sub-batches of at
 *     most 512 images through the same arenas, each as `streams` parts (rn_model_set_streams);
 *     every image's logits are independent of that split, bit for bit;
 *   - arenas: 13.6 MB (fp32) / 6.8 MB (bf16) of activations per image of the largest sub-batch
 *     seen, allocated on first use (RN_ERR_NOMEM when the device cannot hold them). */
RN_API int rn_model_forward(rn_model *m, const float *input_nchw, uint64_t B, float *logits,
                            int mode);
/* Run one forward, then time every tile candidate of every convolution at batch B on the
 * device (events on the context's stream, on the buffers that forward used) and remember the fastest per layer for
 * that batch size.  Results do not change (candidates are bit-identical), only speed. */
RN_API int rn_model_tune(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode);
/* The table that rn_model_tune filled, as 64-bit words (header: architecture, element type, fusion
 * settings, batch, mode, candidate count of the build; then the (tile, launch batch) slots of every
 * layer), and its way into another model of the same architecture / element type / settings on an
 * identical device: the shards of a node take over ONE shard's measurement (rn_shard_tune), a later
 * process a stored one.  export with words == NULL returns the size in *n_words; import refuses a
 * table measured for anything else (RN_ERR_INVALID, nothing changed).  Tiles change speed only. */
RN_API int rn_model_export_tuning(const rn_model *m, uint64_t *words, uint64_t cap, uint64_t *n_words);
RN_API int rn_model_import_tuning(rn_model *m, const uint64_t *words, uint64_t n_words);
/* per-op timing of the next forwards: 1 = bracket every op with events */
/* Fused mode only: run conv3 + downsample of the first block of each stage as one contraction
 * (rn_conv2d_nhwc_pair_forward_dt); on by default, off = downsample first, then conv3 with it
 * as the residual.  Changing it invalidates the tuned tiles.  NOT bit-neutral: the pair has one
 * epilogue, so both batch-norm scales are folded into the weight rows (rounded once more) instead of
 * multiplying the fp32 sums; both forms are within the fused mode's tolerance of the reference. */
RN_API int rn_model_set_pair_fusion(rn_model *m, int on);
/* fp32 models: stem through rn_conv2d_nhwc_exact_forward (K = 160; default) or through the
 * 4-channel / 8-slot form of rn_conv2d_nhwc_forward (K = 224).  Invalidates the tuned tiles. */
RN_API int rn_model_set_stem_exact(rn_model *m, int on);
/* streams = 1, 2 or 4: a (sub-)batch runs as that many contiguous parts (of at least 64 images
 * each) on streams of their own -- the launches of one part fill the tails of the others';
 * every image's logits are independent of what else is in its launch, so no bit changes.
 * Default: 2 (measured at B = 256: fp32 +0.8 %, bf16 storage +5..9 %: its 256-wide tiles leave
 * CUs idle in the late stages); as long as this function has not been called, an fp32 model splits
 * only into parts of at least 128 images (parts of 96 measured 3 % slower than one stream, parts
 * of 64 equal).  Profiled and tuning forwards always use one stream.  Tuned tiles
 * are kept per launch batch size (rn_model_tune times the parts' size and the whole batch), so
 * switching between the tuned part count and one stream keeps them.  streams = 0 returns to the
 * library default (as if this function had never been called). */
RN_API int rn_model_set_streams(rn_model *m, int streams);
RN_API int rn_model_get_streams(const rn_model *m);
/* the number of parts (streams) a forward of B images actually runs as under the rule above */
RN_API int rn_model_parts(const rn_model *m, uint64_t B);
/* parts = 1 (default), 2, 4, 8 or 16: the stem, the max-pool and the first stage -- the layers
 * with the largest tensors -- run in that many slices of each batch part, one after the other,
 * so that what one kernel writes is still in the 256 MB Infinity Cache when the next reads it;
 * the other stages then run on the whole part.  Same bits.  Invalidates the tuned tiles. */
RN_API int rn_model_set_front_parts(rn_model *m, int parts);
RN_API int rn_model_set_profiling(rn_model *m, int on);
/* after a profiled forward + rn_sync: number of ops, then one record per op */
RN_API uint64_t rn_model_profile_count(const rn_model *m);
RN_API int rn_model_profile_get(const rn_model *m, uint64_t index, const char **op_name,
                                const char **layer_name, float *ms, double *flops,
                                double *bytes);
RN_API uint64_t rn_model_activation_bytes(const rn_model *m);

/* ---- exact-K small-Cin form (fp32): the 7x7x3 stem as K = 147 -> 160 -----------------
 * rn_conv2d_nhwc_forward pads a 3-channel image to 4 channels and 8 kernel-column slots
 * (K = 7*32 = 224 per output for 147 real products).  This form reads a physically padded
 * [B,Hp,Wp,Cin] image (rn_nchw_to_nhwc_pad_dt with Cpad = Cin and border = the convolution's
 * padding; Hp = H + 2*pad) with dword gathers and packs K = k*k*Cin contiguously, rounded up
 * to a multiple of 32 — 5 K tiles instead of 7 for the ResNet stem (conv1, main.cu:111). */
RN_API uint64_t rn_conv2d_packed_weight_numel_exact(uint64_t in_channels, uint64_t out_channels,
                                                    uint64_t kernel_size);
RN_API int rn_conv2d_pack_weight_exact(rn_ctx *ctx, const float *weight_oihw, float *packed,
                                       uint64_t in_channels, uint64_t out_channels,
                                       uint64_t kernel_size);
RN_API int rn_conv2d_nhwc_exact_forward(rn_ctx *ctx, const float *inp_padded, float *out,
                                        const float *packed_exact_weight, uint64_t kernel_size,
                                        uint64_t stride, uint64_t h_out, uint64_t w_out,
                                        uint64_t B, uint64_t in_channels, uint64_t out_channels,
                                        uint64_t Hp, uint64_t Wp, const rn_epilogue *epilogue);

/* ---- fused stem: conv 7x7/2 + folded batch-norm + ReLU + max-pool 3x3/2/1 in one launch ----
 * The first four ops of the reference's forward (main.cu:179-192) without ever writing the
 * 112x112x64 stem tensor: a direct convolution out of an LDS-resident input patch, the pool as
 * LDS integer maxima of the non-negative ReLU outputs.  inp_padded: NHWC image that carries its
 * own 3-pixel zero border (rn_nchw_to_nhwc_pad_dt with border 3; Cpad = 3 for fp32, 4 for
 * bf16), [B,Hp,Wp,Cpad]; out: [B,PH,PW,64] of `dtype`.  Needs 64 output channels, a conv
 * output width that is a multiple of 8 and at most 128 (ResNet: 112), any height; bf16: an even
 * Wp.  A block walks down an image (or a segment of it, rn_ctx_set_stem_items) and carries the
 * pooled row two consecutive row groups share in LDS.  scale/shift: per channel, nullable.  Same products as conv + bn + relu + maxpool, summed in another order.
 * relu must be non-zero (RN_ERR_INVALID otherwise): the pool is taken as an integer maximum of
 * the non-negative ReLU outputs' bit patterns. */
RN_API uint64_t rn_stem_pool_packed_weight_numel(int dtype);
RN_API int rn_stem_pool_pack_weight_dt(rn_ctx *ctx, int dtype, const float *weight_oihw /* [64,Cin,7,7] */,
                                       void *packed, uint64_t in_channels);
RN_API int rn_stem_pool_forward_dt(rn_ctx *ctx, int dtype, const void *inp_padded, void *out,
                                   const void *packed_weight, const float *scale, const float *shift,
                                   int relu, uint64_t B, uint64_t Hp, uint64_t Wp);
/* The same launch reading the reference's fp32 NCHW image [B,in_channels,H,W] directly: the zero
 * border, the channel interleave and the conversion to `dtype` happen while the input patch is
 * assembled in LDS, so no layout kernel and no padded copy of the image are needed.  W % 4 == 0. */
RN_API int rn_stem_pool_nchw_forward_dt(rn_ctx *ctx, int dtype, const float *inp_nchw, void *out,
                                        const void *packed_weight, const float *scale,
                                        const float *shift, int relu, uint64_t B,
                                        uint64_t in_channels, uint64_t H, uint64_t W);
/* The same launch for a caller that names the stem tensor as an output of its own (the reference does:
 * main.cu:179-190 keeps conv -> bn -> relu in `act1`, then pools it): stem_out [B,Ho,Wo,64] = relu(bn(conv(x))),
 * NHWC fp32, written from the registers that feed the pool, beside pool_out [B,PH,PW,64].  fp32, NCHW image
 * [B,in_channels,H,W], W % 4 == 0; otherwise as rn_stem_pool_nchw_forward_dt.  What the deferred route
 * (rn_ctx_set_deferred) runs for conv 7x7/2 -> bn -> relu -> maxpool 3x3/2/1. */
RN_API int rn_stem_conv_pool_nchw_forward(rn_ctx *ctx, const float *inp_nchw, float *stem_out, float *pool_out,
                                          const void *packed_weight, const float *scale, const float *shift,
                                          uint64_t B, uint64_t in_channels, uint64_t H, uint64_t W);
/* conv3 + bn3 + residual add + ReLU of one bottleneck block and conv1 + bn1 + ReLU of the next
 * (layerForward, main.cu:131-164, end of one pass and start of the next) as ONE launch on bf16 or
 * fp32 NHWC tensors: t2 [rows][mid] -> y [rows][channels] (written: the next block's residual) ->
 * t1 [rows][next_mid]; y reaches conv1 through LDS instead of HBM.  The same bits as the two
 * separate rn_conv2d_nhwc_forward_dt calls.  (mid, channels, next_mid) = (64, 256, 64 | 128) or, bf16
 * only, (128, 512, 128); packed weights from rn_conv2d_pack_weight_dt; scale/shift may be NULL. */
RN_API int rn_conv_chain_forward_dt(rn_ctx *ctx, int dtype, const void *t2, const void *residual,
                                    void *y, const void *packed_w3, const float *scale3,
                                    const float *shift3, void *t1, const void *packed_w1,
                                    const float *scale1, const float *shift1, uint64_t rows,
                                    uint64_t mid_channels, uint64_t channels, uint64_t next_mid);
/* The same chain out of the fused conv3 + downsample pair of a stage's first block
 * (rn_conv2d_nhwc_pair_forward_dt: panel from rn_conv2d_pack_weight_pair_dt with the scales folded
 * in, K = mid + in2 channels, no residual): y = relu(t2 . w3s + x2 . wds + shift), then conv1.
 * in2_channels 64; fp32: next_mid 64. */
RN_API int rn_conv_chain_pair_forward_dt(rn_ctx *ctx, int dtype, const void *t2, const void *x2,
                                         void *y, const void *packed_pair, const float *shift,
                                         void *t1, const void *packed_w1, const float *scale1,
                                         const float *shift1, uint64_t rows, uint64_t mid_channels,
                                         uint64_t in2_channels, uint64_t channels, uint64_t next_mid);
/* Fused mode: conv3 of a block and conv1 of the block after it as one launch where a chain kernel
 * exists (rn_conv_chain_forward_dt; default on).  The same bits whatever the setting. */
RN_API int rn_model_set_chain(rn_model *m, int on);
/* Fused mode: use it for conv1 + bn1 + relu + maxpool (default on; fp32 needs the exact-K stem
 * image, rn_model_set_stem_exact).  on == 2: through rn_stem_pool_nchw_forward_dt, the input
 * layout launch disappears as well.  Invalidates the tuned tiles.  on = 1 and on = 2 give the same
 * bits; on = 0 (stem and max-pool as separate launches) sums the 147 products of an output in another
 * order (another K padding), so its results differ in the last bits -- which is why the model keeps
 * ONE setting for every batch size. */
RN_API int rn_model_set_stem_pool_fusion(rn_model *m, int on);

/* ---- fused pair: out = epilogue(conv(inp, W1) + conv1x1(inp2, W2)) ------------------
 * One contraction whose K loop runs through both convolutions: the bottleneck's conv3 and the
 * block's downsample convolution (main.cu:134-147) without writing and re-reading the
 * downsample tensor.  One accumulator, so each branch's BatchNorm scale is folded into its
 * weight rows when the pair is packed; the epilogue then carries only the summed shifts
 * (scale = NULL), an optional residual and ReLU.  The second convolution is 1x1 / padding 0
 * and must produce the same [B,h_out,w_out,Cout] shape.  Both channel counts must be
 * multiples of 32 (fp32) / 64 (bf16): RN_ERR_UNSUPPORTED otherwise. */
typedef struct rn_conv_second {
    const void *inp;      /* [B,H,W,in_channels] NHWC, element type = dtype */
    uint64_t in_channels, H, W, stride;
} rn_conv_second;
RN_API uint64_t rn_conv2d_packed_pair_weight_numel(uint64_t in_channels, uint64_t out_channels,
                                                   uint64_t kernel_size, uint64_t in_channels2);
/* rows [Cout][k*k*Cin + Cin2]; scale1 / scale2 (per out channel, nullable = 1) multiply the rows */
RN_API int rn_conv2d_pack_weight_pair_dt(rn_ctx *ctx, int dtype, const float *w1_oihw,
                                         const float *scale1, const float *w2_oihw,
                                         const float *scale2, void *packed, uint64_t in_channels,
                                         uint64_t out_channels, uint64_t kernel_size,
                                         uint64_t in_channels2);
RN_API int rn_conv2d_nhwc_pair_forward_dt(rn_ctx *ctx, int dtype, int out_dtype, const void *inp,
                                          void *out, const void *packed_pair_weight,
                                          uint64_t kernel_size, uint64_t stride, uint64_t padding,
                                          uint64_t h_out, uint64_t w_out, uint64_t B,
                                          uint64_t in_channels, uint64_t out_channels, uint64_t H,
                                          uint64_t W, const rn_conv_second *second,
                                          const rn_epilogue *epilogue);

/* ---- captured forward ---------------------------------------------------------------
 * The forward of one (input, B, logits, mode) as a hipGraph: ~57 launches (RN-50, fused) replayed
 * by one call.  Worth it at small B, where the launches' host cost is comparable to their GPU
 * time; at B=256 the GPU is the bound either way.  Capture runs one eager forward first (arenas
 * and scratch are allocated outside the capture) and bakes in the tile choice of that moment
 * (call rn_model_tune before).  Needs profiling and sync_each_op off (RN_ERR_INVALID otherwise).
 * The buffers must stay valid while the graph lives; while any graph of a context lives, a
 * call that would have to grow the context's scratch or the model's activation arenas (a
 * larger batch than any before) returns RN_ERR_INVALID instead of moving them.
 * Teardown order: rn_graph_destroy, then rn_model_destroy, then rn_ctx_destroy.  A graph points into
 * its model's arenas and into the scratch of the contexts the model owns for its extra streams:
 * rn_model_destroy returns RN_ERR_INVALID (and frees nothing) while a graph captured from the model
 * lives. */
typedef struct rn_graph rn_graph;
RN_API int rn_model_capture(rn_model *m, const float *input_nchw, uint64_t B, float *logits,
                            int mode, rn_graph **out);
RN_API int rn_graph_launch(rn_graph *g); /* asynchronous, on the context's stream */
RN_API int rn_graph_destroy(rn_graph *g);
RN_API uint64_t rn_graph_node_count(const rn_graph *g);

/* ---- host pipeline: overlapped upload / forward / download --------------------------
 * What main() of the reference does once (H2D of the image, forward, D2H of the logits:
 * main.cu:236-240) as a stream of batches: two slots; the upload of batch i+1 (pinned
 * staging -> device, on a copy stream) runs beside the forward of batch i.  submit() blocks
 * only when both slots are busy; collect() returns the logits of the oldest batch. */
typedef struct rn_pipeline rn_pipeline;
RN_API int rn_pipeline_create(rn_model *m, rn_pipeline **out, uint64_t B, int mode);
RN_API int rn_pipeline_destroy(rn_pipeline *p);
/* The pinned staging buffer (B*3*224*224 floats) the next submit will upload from: a decoder
 * that writes its output there saves the host-side copy.  RN_ERR_INVALID when both slots are
 * busy. */
RN_API int rn_pipeline_input_buffer(rn_pipeline *p, float **host_staging);
/* host_input_nchw: B*3*224*224 floats in any host memory (copied into the staging buffer), or
 * NULL / the pointer rn_pipeline_input_buffer returned when the staging buffer is already
 * filled. */
RN_API int rn_pipeline_submit(rn_pipeline *p, const float *host_input_nchw);
/* host_logits: B*1000 floats.  RN_ERR_INVALID when nothing is in flight. */
RN_API int rn_pipeline_collect(rn_pipeline *p, float *host_logits);
RN_API uint64_t rn_pipeline_in_flight(const rn_pipeline *p);
/* The same for a batch of n <= B images (a ragged last batch), and with the class indices
 * (first maximum wins, main.cu:243-249) next to the logits; host_logits ([n,1000]), host_top1
 * ([n]) and n may each be NULL. */
RN_API int rn_pipeline_submit_n(rn_pipeline *p, const float *host_input_nchw, uint64_t n);
RN_API int rn_pipeline_collect_n(rn_pipeline *p, float *host_logits, uint64_t *host_top1, uint64_t *n);

/* ---- one batch over several devices of a node ---------------------------------------
 * The multi-device form of the reference's main() (main.cu:228-254).  The forward has no
 * cross-image reduction, so the batch shards contiguously (shard g of G owns images
 * [lo, hi) of rn_shard_bounds), weights are replicated, no data moves between devices and the
 * logits are concatenated on the host.  One host thread + one context + one model per listed
 * device, created and driven by the group; a device may be listed more than once.  Calls on a
 * group are serialised by the caller.  host_* pointers are host memory; host_logits
 * ([B,1000]) and host_top1 ([B], first maximum wins as main.cu:243-249) may each be NULL. */
typedef struct rn_shard rn_shard;
RN_API void rn_shard_bounds(uint64_t B, int rank, int world, uint64_t *lo, uint64_t *hi);
RN_API int rn_shard_create(rn_shard **out, const int *devices, int n_devices, int arch);
RN_API int rn_shard_destroy(rn_shard *g);
RN_API int rn_shard_count(const rn_shard *g);
RN_API const char *rn_shard_last_error(const rn_shard *g);
RN_API int rn_shard_set_tensor(rn_shard *g, const char *key, const float *host_data, uint64_t numel);
RN_API int rn_shard_load_dir(rn_shard *g, const char *weights_dir);
RN_API int rn_shard_set_dtype(rn_shard *g, int dtype);
RN_API int rn_shard_finalize(rn_shard *g);
RN_API int rn_shard_forward(rn_shard *g, const float *host_input_nchw, uint64_t B,
                            float *host_logits, uint64_t *host_top1, int mode);
/* Tuned tiles for the launches the group will issue for batches of B: with a stream open
 * (rn_shard_stream_open) those of a whole shard per device, otherwise those of rn_shard_forward's
 * chunks (at most 128 images).  ONE shard measures (rn_model_tune on shard 0, the other devices
 * idle), the shards with the same share take its table over (rn_model_import_tuning: identical
 * devices, and every shard then runs the same tiles); a shard with another share (uneven split)
 * measures for itself.  Refused while submitted batches are in flight. */
RN_API int rn_shard_tune(rn_shard *g, const float *host_input_nchw, uint64_t B, int mode);
/* Where shard `rank` runs: its device, the NUMA node of that device and the CPUs its host thread was
 * bound to ("" = not bound).  A shard's thread is bound to the cores local to its device
 * (rn_device_locality, intersected with the cores the process may use) when it starts: it copies every
 * batch from the caller's pageable memory into pinned staging, and on a two-socket node the remote
 * socket's memory path halves that copy.  Best effort; RN_SHARD_AFFINITY=0 in the environment turns it off. */
RN_API int rn_shard_placement(const rn_shard *g, int rank, int *device, int *numa_node, char *cpulist,
                              uint64_t cpulist_cap);
/* Shard `rank`'s model, for settings and queries (rn_model_set_streams, rn_model_export_tuning ...)
 * between calls on the group, when its host thread is parked.  Owned by the group; running a forward
 * on it from the caller's thread is not supported (its context belongs to the shard's thread). */
RN_API rn_model *rn_shard_model(rn_shard *g, int rank);
/* Upload, forward and download overlap on every device (main.cu:236-240 and tensor.cuh:184-199
 * do them strictly in sequence, from pageable memory): each device owns an rn_pipeline -- pinned
 * staging, a copy stream, two slots.  rn_shard_forward sends a shard through it in chunks of at
 * most 128 images, two in flight.  The streaming form below keeps two whole BATCHES in flight:
 *   rn_shard_stream_open(g, B, mode);            shard r owns images rn_shard_bounds(B, r, G)
 *   rn_shard_submit(g, batch0); rn_shard_submit(g, batch1);
 *   rn_shard_collect(g, logits0, top1_0); rn_shard_submit(g, batch2); ...
 * submit returns once every device has queued its shard (the copy into pinned staging is done
 * by the device's own host thread, all devices in parallel); collect returns the oldest batch,
 * rows in image order.  rn_shard_stream_buffer gives shard `rank`'s pinned staging buffer of the
 * NEXT submit and the image range [lo, hi) it holds: a decoder that writes there and submits
 * NULL saves the host-side copy.  At most two batches in flight (RN_ERR_INVALID otherwise).
 * rn_shard_forward re-sizes the per-device pipelines for its own chunks: it is refused while
 * submitted batches are in flight, and a stream must be opened again after it. */
RN_API int rn_shard_stream_open(rn_shard *g, uint64_t B, int mode);
RN_API int rn_shard_stream_buffer(rn_shard *g, int rank, float **host_staging, uint64_t *lo,
                                  uint64_t *hi);
RN_API int rn_shard_submit(rn_shard *g, const float *host_input_nchw /* NULL: staging filled */);
RN_API int rn_shard_collect(rn_shard *g, float *host_logits, uint64_t *host_top1);
RN_API int rn_shard_in_flight(const rn_shard *g);
RN_API int rn_shard_stream_close(rn_shard *g);

#ifdef __cplusplus
}
#endif
#endif /* RN_HIP_H */
