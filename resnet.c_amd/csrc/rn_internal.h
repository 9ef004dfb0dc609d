// Internal definitions shared by the HIP translation units of librn_hip.so.
#ifndef RN_INTERNAL_H
#define RN_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "rn_hip.h"

struct rn_wcache_entry {
    const void *key;  // the caller's OIHW weight buffer
    uint64_t cin, cout, k;
    void *packed;
};

struct rn_ctx {
    int device;
    int cus;  // compute units of the device (256 on MI355X), asked once at creation
    hipStream_t stream;
    bool own_stream;
    int layout;
    int sync_each_op;
    int conv_tile;  // 0 = choose per launch, 1..N = force candidate (tuning)
    int graphs_live;  // captured forwards that hold pointers into the scratch and the arenas
    int split_k;    // 0 = never split the K loop; n = up to n partial sums per output (latency mode)
    int stem_items; // fused stem: items (four stem rows) a block walks; 0 = chosen per launch
    int nchw_taps;  // k x k convolutions on NCHW tensors without the input transpose: 0 never, 1 large planes, 2 always
    int xcd_groups; // N-tile groups the contraction's tiles are dealt to the XCDs in; 0 = chosen per launch
    void *debug_stamps;  // diagnostic phase stamps of the contraction kernel, normally null
    // scratch grown on demand (never inside a graph capture; callers that capture
    // warm up first so the sizes are already settled)
    void *scratch[5];  // 0 batch-norm constants, 1-3 NCHW convolution, 4 split-K partial sums
    uint64_t scratch_bytes[5];
    // resident blocks per CU of each contraction-kernel instantiation on THIS context's device,
    // asked once per context (0 = not asked yet); no process-wide mutable state
    int occupancy[256];
    uint64_t launches;  // kernel launches issued through this context (rn_after_launch)
    // packed-weight cache of the NCHW drop-in route (rn_conv2d_forward): OIHW weight pointer +
    // shape -> K-major panel, packed on first use; off unless rn_ctx_set_weight_cache(ctx, 1)
    int wcache_on, wcache_n, wcache_cap;
    struct rn_wcache_entry *wcache;
    // deferred execution of the reference's op-by-op calls (rn_defer.hip): recorded ops, the caller
    // buffers that currently hold NHWC, folded batch-norm constants
    int defer;          // rn_ctx_set_deferred
    int defer_running;  // the recorded list (or a layout rewrite) is executing: entry points run for real
    struct rn_defer_state *ds;
    char err[512];
};

struct rn_event {
    hipEvent_t ev;
};

int rn_set_error(rn_ctx *ctx, int status, const char *fmt, ...);
// make the context's device the calling thread's current device (a host thread may own
// contexts on several devices; a launch on the wrong current device would fail or, worse, run
// elsewhere).  Every entry point that touches the device starts with RN_ENTER.
int rn_bind_device(rn_ctx *ctx);
int rn_check_hip(rn_ctx *ctx, hipError_t e, const char *what);
// scratch slot `slot` with at least `bytes`; contents undefined
int rn_scratch(rn_ctx *ctx, int slot, uint64_t bytes, void **ptr);
// packed-weight cache: the panel for (weight, shape), or null; add = allocate an entry's panel
// (the caller packs into it; remove = take that entry back when the pack failed); drop = forget
// every entry whose weight buffer [key, key + cin*cout*k*k*4) overlaps [lo, lo + bytes)
void *rn_wcache_find(rn_ctx *ctx, const void *weight, uint64_t cin, uint64_t cout, uint64_t k);
int rn_wcache_add(rn_ctx *ctx, const void *weight, uint64_t cin, uint64_t cout, uint64_t k,
                  uint64_t bytes, void **packed);
void rn_wcache_remove(rn_ctx *ctx, const void *packed);
void rn_wcache_drop(rn_ctx *ctx, const void *lo, uint64_t bytes);
// launch epilogue shared by every op: launch-error check and optional per-op sync
int rn_after_launch(rn_ctx *ctx, const char *what);

#define RN_HIP_TRY(ctx, expr)                                  \
    do {                                                       \
        int rn_st_ = rn_check_hip((ctx), (expr), #expr);       \
        if (rn_st_ != RN_OK) return rn_st_;                    \
    } while (0)

// rn_defer.hip.  flush: run the recorded ops; barrier: flush + every caller buffer back to NCHW;
// before_write / before_read: an access to caller memory from outside the recorded list
int rn_defer_flush(rn_ctx *ctx);
int rn_defer_barrier(rn_ctx *ctx);
int rn_defer_before_write(rn_ctx *ctx, const void *ptr, uint64_t bytes, int whole_allocation);
int rn_defer_before_read(rn_ctx *ctx, const void *ptr, uint64_t bytes, const void **alt);
void rn_defer_destroy(rn_ctx *ctx);
int rn_defer_conv(rn_ctx *ctx, const float *inp, float *out, const float *w, uint64_t k, uint64_t s, uint64_t p,
                  uint64_t ho, uint64_t wo, uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W);
int rn_defer_bn(rn_ctx *ctx, const float *inp, float *out, const float *w, const float *b, const float *mean,
                const float *var, uint64_t B, uint64_t C, uint64_t N);
int rn_defer_eltwise(rn_ctx *ctx, int is_add, const float *a, const float *b, float *out, uint64_t N);
int rn_defer_pool(rn_ctx *ctx, int is_max, const float *inp, float *out, uint64_t k, uint64_t s, uint64_t p,
                  uint64_t ho, uint64_t wo, uint64_t B, uint64_t C, uint64_t H, uint64_t W);
int rn_defer_linear(rn_ctx *ctx, const float *inp, float *out, const float *w, const float *b, uint64_t B,
                    uint64_t in_f, uint64_t out_f);
// a reference entry point on a deferred context in its NCHW layout records its call instead of launching
#define RN_DEFERS(ctx) ((ctx)->defer && !(ctx)->defer_running && (ctx)->layout == RN_LAYOUT_NCHW)

// every entry point that touches the device: bind the device; on a context with deferred state, run what
// has been recorded and give every caller buffer its NCHW content back first (the seven reference ops
// record themselves before they get here; the copies and frees of rn_ctx.hip have finer rules)
#define RN_ENTER(ctx)                                  \
    do {                                               \
        if (!(ctx)) return RN_ERR_INVALID;             \
        int rn_en_ = rn_bind_device(ctx);              \
        if (rn_en_ != RN_OK) return rn_en_;            \
        if ((ctx)->ds && !(ctx)->defer_running) {      \
            rn_en_ = rn_defer_barrier(ctx);            \
            if (rn_en_ != RN_OK) return rn_en_;        \
        }                                              \
    } while (0)

#define RN_TRY(expr)                       \
    do {                                   \
        int rn_st_ = (expr);               \
        if (rn_st_ != RN_OK) return rn_st_; \
    } while (0)

#define RN_REQUIRE(ctx, cond, msg)                                   \
    do {                                                             \
        if (!(cond)) return rn_set_error((ctx), RN_ERR_INVALID, "%s: %s", __func__, (msg)); \
    } while (0)

static inline uint64_t rn_ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride bandwidth kernel: enough blocks to fill 256 CUs x 8, no more
static inline unsigned rn_stream_grid(uint64_t work_items, unsigned block)
{
    uint64_t g = rn_ceil_div(work_items, block);
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    return (unsigned)g;
}

// small-Cin ("stem") form of the contraction: Cin <= 4 and k <= 8
bool rn_conv_is_c4(uint64_t Cin, uint64_t k);


#endif
