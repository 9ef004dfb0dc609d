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

#define RN_ENTER(ctx)                                  \
    do {                                               \
        if (!(ctx)) return RN_ERR_INVALID;             \
        int rn_en_ = rn_bind_device(ctx);              \
        if (rn_en_ != RN_OK) return rn_en_;            \
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
