// Context, buffers, file I/O and events: the Tensor / helpers side of the boundary
// (reference cuda/tensor.cuh:59-245, cuda/helpers.cuh:6-35), as status-returning C.
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rn_conv_params.h"

int rn_set_error(rn_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

int rn_check_hip(rn_ctx *ctx, hipError_t e, const char *what)
{
    if (e == hipSuccess) return RN_OK;
    return rn_set_error(ctx, RN_ERR_HIP, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e),
                        what);
}

int rn_after_launch(rn_ctx *ctx, const char *what)
{
    ++ctx->launches;
    RN_TRY(rn_check_hip(ctx, hipGetLastError(), what));
    if (ctx->sync_each_op) {
        RN_TRY(rn_check_hip(ctx, hipStreamSynchronize(ctx->stream), what));
    }
    return RN_OK;
}

int rn_scratch(rn_ctx *ctx, int slot, uint64_t bytes, void **ptr)
{
    if (slot < 0 || slot >= 5) return rn_set_error(ctx, RN_ERR_INVALID, "bad scratch slot");
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->graphs_live > 0)
            return rn_set_error(ctx, RN_ERR_INVALID,
                                "scratch would have to grow while %d captured graph(s) point into it: "
                                "destroy them first", ctx->graphs_live);
        // the old block may still be in use by queued kernels
        RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch[slot]) RN_HIP_TRY(ctx, hipFree(ctx->scratch[slot]));
        ctx->scratch[slot] = nullptr;
        ctx->scratch_bytes[slot] = 0;
        uint64_t want = (bytes + 255) & ~(uint64_t)255;
        RN_HIP_TRY(ctx, hipMalloc(&ctx->scratch[slot], want));
        ctx->scratch_bytes[slot] = want;
    }
    *ptr = ctx->scratch[slot];
    return RN_OK;
}

void *rn_wcache_find(rn_ctx *ctx, const void *weight, uint64_t cin, uint64_t cout, uint64_t k)
{
    for (int i = 0; i < ctx->wcache_n; ++i) {
        const rn_wcache_entry &e = ctx->wcache[i];
        if (e.key == weight && e.cin == cin && e.cout == cout && e.k == k) return e.packed;
    }
    return nullptr;
}

int rn_wcache_add(rn_ctx *ctx, const void *weight, uint64_t cin, uint64_t cout, uint64_t k,
                  uint64_t bytes, void **packed)
{
    if (ctx->wcache_n == ctx->wcache_cap) {
        const int ncap = ctx->wcache_cap ? 2 * ctx->wcache_cap : 64;
        rn_wcache_entry *ne = (rn_wcache_entry *)realloc(ctx->wcache, (size_t)ncap * sizeof(rn_wcache_entry));
        if (!ne) return rn_set_error(ctx, RN_ERR_NOMEM, "weight cache table");
        ctx->wcache = ne;
        ctx->wcache_cap = ncap;
    }
    void *p = nullptr;
    RN_HIP_TRY(ctx, hipMalloc(&p, bytes));
    rn_wcache_entry &e = ctx->wcache[ctx->wcache_n++];
    e.key = weight;
    e.cin = cin;
    e.cout = cout;
    e.k = k;
    e.packed = p;
    *packed = p;
    return RN_OK;
}

void rn_wcache_remove(rn_ctx *ctx, const void *packed)
{
    for (int i = 0; i < ctx->wcache_n; ++i) {
        if (ctx->wcache[i].packed != packed) continue;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->wcache[i].packed);
        ctx->wcache[i] = ctx->wcache[--ctx->wcache_n];
        return;
    }
}

void rn_wcache_drop(rn_ctx *ctx, const void *lo, uint64_t bytes)
{
    // a write that starts INSIDE a cached weight buffer stales its panel too: compare intervals,
    // [key, key + extent) of the OIHW weight against [lo, lo + bytes) of the write
    const uintptr_t a = (uintptr_t)lo;
    const uintptr_t b = bytes >= ~(uintptr_t)0 - a ? ~(uintptr_t)0 : a + (bytes ? bytes : 1);
    int kept = 0, dropped = 0;
    for (int i = 0; i < ctx->wcache_n; ++i) {
        const rn_wcache_entry &e = ctx->wcache[i];
        const uintptr_t key = (uintptr_t)e.key;
        const uintptr_t end = key + (uintptr_t)(e.cin * e.cout * e.k * e.k * sizeof(float));
        if (key < b && end > a) {
            if (!dropped++) (void)hipStreamSynchronize(ctx->stream);  // a queued launch may read it
            (void)hipFree(ctx->wcache[i].packed);
        } else {
            ctx->wcache[kept++] = ctx->wcache[i];
        }
    }
    ctx->wcache_n = kept;
}

int rn_bind_device(rn_ctx *ctx)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == ctx->device) return RN_OK;
    return rn_check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
}

extern "C" {

const char *rn_version(void) { return "resnet.c_amd 0.1 (gfx950)"; }

const char *rn_status_string(int status)
{
    switch (status) {
        case RN_OK: return "ok";
        case RN_ERR_INVALID: return "invalid argument";
        case RN_ERR_HIP: return "HIP runtime error";
        case RN_ERR_IO: return "I/O error";
        case RN_ERR_NOMEM: return "out of memory";
        case RN_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

int rn_device_count(int *count)
{
    if (!count) return RN_ERR_INVALID;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return RN_ERR_HIP;
    }
    return RN_OK;
}

// Where a device sits in the host: its PCI address, the NUMA node of that slot and the CPUs local to
// it, as Linux reports them (/sys/bus/pci/devices/<address>/numa_node, local_cpulist).  A host thread
// that feeds a device (pageable -> pinned copies, launches) should run on those cores: on a two-socket
// 8-GPU node half of the devices hang off the other socket.  Missing sysfs entries are not an error:
// numa_node = -1, cpulist = "".
int rn_device_locality(int device, char *pci_bus_id, uint64_t pci_cap, int *numa_node, char *cpulist,
                       uint64_t cpulist_cap)
{
    char addr[32] = {0}, path[128];
    if (numa_node) *numa_node = -1;
    if (cpulist && cpulist_cap) cpulist[0] = 0;
    if (pci_bus_id && pci_cap) pci_bus_id[0] = 0;
    if (hipDeviceGetPCIBusId(addr, (int)sizeof(addr) - 1, device) != hipSuccess) return RN_ERR_HIP;
    for (char *c = addr; *c; ++c)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs spells the address in lower case
    if (pci_bus_id && pci_cap) snprintf(pci_bus_id, (size_t)pci_cap, "%s", addr);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", addr);
    if (FILE *f = fopen(path, "r")) {
        int node = -1;
        if (fscanf(f, "%d", &node) == 1 && numa_node) *numa_node = node;
        fclose(f);
    }
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", addr);
    if (FILE *f = fopen(path, "r")) {
        if (cpulist && cpulist_cap && fgets(cpulist, (int)cpulist_cap, f)) {
            for (char *c = cpulist; *c; ++c)
                if (*c == '\n') *c = 0;
        }
        fclose(f);
    }
    return RN_OK;
}

int rn_ctx_create(rn_ctx **out, int device, void *hip_stream)
{
    if (!out) return RN_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RN_ERR_HIP;
    if (device < 0 || device >= n) return RN_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return RN_ERR_HIP;
    rn_ctx *ctx = (rn_ctx *)calloc(1, sizeof(rn_ctx));
    if (!ctx) return RN_ERR_NOMEM;
    ctx->device = device;
    ctx->cus = 256;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) ctx->cus = v;
    }
    ctx->layout = RN_LAYOUT_NCHW;
    if (const char *xg = getenv("RN_XCD_NGROUPS")) {  // A/B runs of the tile order (tools/tile_fetch_ab.sh)
        const int g = atoi(xg);
        if (g == 1 || g == 2 || g == 4 || g == 8) ctx->xcd_groups = g;
    }
    ctx->nchw_taps = 1;
    if (const char *nt = getenv("RN_NCHW_TAPS")) {  // A/B runs of the literal route (tools/nchw_bench.py)
        const int v = atoi(nt);
        if (v >= 0 && v <= 2) ctx->nchw_taps = v;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            free(ctx);
            return RN_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return RN_OK;
}

int rn_ctx_destroy(rn_ctx *ctx)
{
    if (!ctx) return RN_OK;
    (void)hipSetDevice(ctx->device);
    (void)rn_defer_flush(ctx);  // recorded ops run (their outputs may be read through other contexts later)
    (void)hipStreamSynchronize(ctx->stream);
    rn_defer_destroy(ctx);
    for (int i = 0; i < 5; ++i) {
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    }
    for (int i = 0; i < ctx->wcache_n; ++i) (void)hipFree(ctx->wcache[i].packed);
    free(ctx->wcache);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    free(ctx);
    return RN_OK;
}

int rn_ctx_set_layout(rn_ctx *ctx, int layout)
{
    if (!ctx) return RN_ERR_INVALID;
    if (layout != RN_LAYOUT_NCHW && layout != RN_LAYOUT_NHWC)
        return rn_set_error(ctx, RN_ERR_INVALID, "unknown layout %d", layout);
    if (ctx->ds && !ctx->defer_running && layout != ctx->layout) {
        // recorded ops were called under the old layout; buffers tagged NHWC by the deferred route
        // get their NCHW content back before the caller starts to say what layout tensors have
        RN_TRY(rn_bind_device(ctx));
        RN_TRY(rn_defer_barrier(ctx));
    }
    ctx->layout = layout;
    return RN_OK;
}

int rn_ctx_get_layout(const rn_ctx *ctx) { return ctx ? ctx->layout : -1; }

int rn_ctx_set_sync_each_op(rn_ctx *ctx, int on)
{
    if (!ctx) return RN_ERR_INVALID;
    ctx->sync_each_op = on ? 1 : 0;
    return RN_OK;
}

int rn_ctx_set_conv_tile(rn_ctx *ctx, int candidate)
{
    if (!ctx) return RN_ERR_INVALID;
    if (candidate < 0 || candidate > rn_conv_tile_candidates())
        return rn_set_error(ctx, RN_ERR_INVALID, "conv tile candidate %d out of range", candidate);
    ctx->conv_tile = candidate;
    return RN_OK;
}

int rn_conv_tile_candidates(void) { return 8 + rn_conv_wide_count() + 1; }  // + the bf16 strip kernels

// library-internal (rn_model.c is plain C and sees the context only through functions)
int rn_ctx_graphs_live(const rn_ctx *ctx) { return ctx ? ctx->graphs_live : 0; }

int rn_ctx_set_split_k(rn_ctx *ctx, int max_splits)
{
    if (!ctx) return RN_ERR_INVALID;
    if (max_splits < 0 || max_splits > 64)
        return rn_set_error(ctx, RN_ERR_INVALID, "split_k %d out of range [0, 64]", max_splits);
    ctx->split_k = max_splits <= 1 ? 0 : max_splits;
    return RN_OK;
}

int rn_ctx_set_xcd_groups(rn_ctx *ctx, int groups)
{
    if (!ctx) return RN_ERR_INVALID;
    if (groups != 0 && groups != 1 && groups != 2 && groups != 4 && groups != 8)
        return rn_set_error(ctx, RN_ERR_INVALID, "xcd groups %d: 0 (chosen per launch), 1, 2, 4 or 8", groups);
    ctx->xcd_groups = groups;
    return RN_OK;
}

int rn_ctx_set_nchw_taps(rn_ctx *ctx, int mode)
{
    if (!ctx) return RN_ERR_INVALID;
    if (mode < 0 || mode > 2) return rn_set_error(ctx, RN_ERR_INVALID, "nchw taps mode %d: 0 (never), 1 (large planes) or 2 (always)", mode);
    ctx->nchw_taps = mode;
    return RN_OK;
}

int rn_ctx_set_stem_items(rn_ctx *ctx, int items)
{
    if (!ctx) return RN_ERR_INVALID;
    if (items < 0) return rn_set_error(ctx, RN_ERR_INVALID, "stem items %d: 0 (chosen per launch) or a positive count", items);
    ctx->stem_items = items;
    return RN_OK;
}

// library-internal: everything queued on ctx's stream after this call waits for the event
int rn_ctx_wait_event(rn_ctx *ctx, rn_event *ev)
{
    if (!ctx || !ev) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    RN_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev->ev, 0));
    return RN_OK;
}

uint64_t rn_ctx_launch_count(const rn_ctx *ctx) { return ctx ? ctx->launches : 0; }

int rn_ctx_set_weight_cache(rn_ctx *ctx, int on)
{
    if (!ctx) return RN_ERR_INVALID;
    ctx->wcache_on = on ? 1 : 0;
    if (!on && ctx->wcache_n) {
        RN_TRY(rn_bind_device(ctx));
        rn_wcache_drop(ctx, nullptr, ~(uint64_t)0);
    }
    return RN_OK;
}

int rn_ctx_set_debug_stamps(rn_ctx *ctx, void *dev_buffer)
{
    if (!ctx) return RN_ERR_INVALID;
    ctx->debug_stamps = dev_buffer;
    return RN_OK;
}

void *rn_ctx_stream(rn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int rn_ctx_device(const rn_ctx *ctx) { return ctx ? ctx->device : -1; }

int rn_sync(rn_ctx *ctx)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    RN_TRY(rn_defer_flush(ctx));  // "everything I called has run" includes the recorded ops
    RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    RN_HIP_TRY(ctx, hipGetLastError());
    return RN_OK;
}

const char *rn_last_error(const rn_ctx *ctx) { return ctx ? ctx->err : "no context"; }

int rn_malloc(rn_ctx *ctx, void **dev_ptr, uint64_t bytes)
{
    if (!ctx || !dev_ptr) return RN_ERR_INVALID;
    *dev_ptr = nullptr;
    if (bytes == 0) return RN_OK;  // empty tensor: null data (tensor.cuh:62-65)
    RN_TRY(rn_bind_device(ctx));
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        return rn_set_error(ctx, RN_ERR_NOMEM, "hipMalloc(%llu) out of memory",
                            (unsigned long long)bytes);
    }
    return rn_check_hip(ctx, e, "hipMalloc");
}

int rn_free(rn_ctx *ctx, void *dev_ptr)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!dev_ptr) return RN_OK;
    if (ctx->ds) {  // recorded ops may name this buffer: they run first; its tag and folded constants go
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, dev_ptr) == hipSuccess) {
            RN_TRY(rn_defer_before_write(ctx, base, size, 1));
        } else {
            (void)hipGetLastError();
            RN_TRY(rn_defer_before_write(ctx, dev_ptr, 1, 1));
        }
    }
    RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->wcache_n) {  // panels packed from weights inside this allocation are stale now
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, dev_ptr) == hipSuccess)
            rn_wcache_drop(ctx, base, size);
        else
            (void)hipGetLastError(), rn_wcache_drop(ctx, dev_ptr, 1);
    }
    RN_HIP_TRY(ctx, hipFree(dev_ptr));
    return RN_OK;
}

int rn_memcpy_h2d(rn_ctx *ctx, void *dev_dst, const void *host_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dev_dst || !host_src))) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!bytes) return RN_OK;
    if (ctx->ds) RN_TRY(rn_defer_before_write(ctx, dev_dst, bytes, 0));
    if (ctx->wcache_n) rn_wcache_drop(ctx, dev_dst, bytes);
    RN_HIP_TRY(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RN_OK;
}

int rn_memcpy_d2h(rn_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!host_dst || !dev_src))) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!bytes) return RN_OK;
    // a buffer the deferred route holds as NHWC goes out as NCHW: a whole tensor through scratch (the
    // buffer stays NHWC for the ops that follow), anything else after an in-place rewrite
    if (ctx->ds) RN_TRY(rn_defer_before_read(ctx, dev_src, bytes, &dev_src));
    RN_HIP_TRY(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RN_OK;
}

int rn_memcpy_d2d(rn_ctx *ctx, void *dev_dst, const void *dev_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dev_dst || !dev_src))) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!bytes) return RN_OK;
    if (ctx->ds) {
        RN_TRY(rn_defer_before_read(ctx, dev_src, bytes, nullptr));
        RN_TRY(rn_defer_before_write(ctx, dev_dst, bytes, 0));
    }
    if (ctx->wcache_n) rn_wcache_drop(ctx, dev_dst, bytes);
    RN_HIP_TRY(ctx,
               hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return RN_OK;
}

int rn_memset(rn_ctx *ctx, void *dev_ptr, int byte_value, uint64_t bytes)
{
    if (!ctx || (bytes && !dev_ptr)) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!bytes) return RN_OK;
    if (ctx->ds) RN_TRY(rn_defer_before_write(ctx, dev_ptr, bytes, 0));
    if (ctx->wcache_n) rn_wcache_drop(ctx, dev_ptr, bytes);
    RN_HIP_TRY(ctx, hipMemsetAsync(dev_ptr, byte_value, bytes, ctx->stream));
    return RN_OK;
}

int rn_load_f32_file(rn_ctx *ctx, const char *path, float **dev_ptr, uint64_t *numel)
{
    if (!ctx || !path || !dev_ptr || !numel) return RN_ERR_INVALID;
    *dev_ptr = nullptr;
    *numel = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return rn_set_error(ctx, RN_ERR_IO, "Can't open %s: %s", path, strerror(errno));
    if (fseek(f, 0, SEEK_END) != 0) {
        fclose(f);
        return rn_set_error(ctx, RN_ERR_IO, "seek failed on %s", path);
    }
    long sz = ftell(f);
    rewind(f);
    uint64_t n = sz > 0 ? (uint64_t)sz / sizeof(float) : 0;
    if (n == 0) {
        fclose(f);
        return rn_set_error(ctx, RN_ERR_IO, "%s holds no floats", path);
    }
    float *host = (float *)malloc(n * sizeof(float));
    if (!host) {
        fclose(f);
        return rn_set_error(ctx, RN_ERR_NOMEM, "host buffer for %s", path);
    }
    size_t got = fread(host, sizeof(float), n, f);
    fclose(f);
    if (got != n) {
        free(host);
        return rn_set_error(ctx, RN_ERR_IO, "short read on %s", path);
    }
    void *d = nullptr;
    int st = rn_malloc(ctx, &d, n * sizeof(float));
    if (st == RN_OK) st = rn_memcpy_h2d(ctx, d, host, n * sizeof(float));
    free(host);
    if (st != RN_OK) {
        if (d) (void)hipFree(d);
        return st;
    }
    *dev_ptr = (float *)d;
    *numel = n;
    return RN_OK;
}

int rn_save_f32_file(rn_ctx *ctx, const char *path, const float *dev_ptr, uint64_t numel)
{
    if (!ctx || !path || (numel && !dev_ptr)) return RN_ERR_INVALID;
    float *host = (float *)malloc(numel ? numel * sizeof(float) : 1);
    if (!host) return rn_set_error(ctx, RN_ERR_NOMEM, "host buffer for %s", path);
    int st = rn_memcpy_d2h(ctx, host, dev_ptr, numel * sizeof(float));
    if (st == RN_OK) {
        FILE *f = fopen(path, "wb");
        if (!f) {
            st = rn_set_error(ctx, RN_ERR_IO, "Can't open %s: %s", path, strerror(errno));
        } else {
            if (fwrite(host, sizeof(float), numel, f) != numel)
                st = rn_set_error(ctx, RN_ERR_IO, "short write on %s", path);
            fclose(f);
        }
    }
    free(host);
    return st;
}

int rn_event_create(rn_ctx *ctx, rn_event **out)
{
    if (!ctx || !out) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    rn_event *ev = (rn_event *)calloc(1, sizeof(rn_event));
    if (!ev) return RN_ERR_NOMEM;
    int st = rn_check_hip(ctx, hipEventCreate(&ev->ev), "hipEventCreate");
    if (st != RN_OK) {
        free(ev);
        return st;
    }
    *out = ev;
    return RN_OK;
}

int rn_event_destroy(rn_event *ev)
{
    if (!ev) return RN_OK;
    (void)hipEventDestroy(ev->ev);
    free(ev);
    return RN_OK;
}

int rn_event_record(rn_ctx *ctx, rn_event *ev)
{
    if (!ctx || !ev) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    RN_TRY(rn_defer_flush(ctx));  // an event marks "after everything called so far": recorded ops included
    RN_HIP_TRY(ctx, hipEventRecord(ev->ev, ctx->stream));
    return RN_OK;
}

int rn_event_elapsed_ms(rn_event *start, rn_event *stop, float *ms)
{
    if (!start || !stop || !ms) return RN_ERR_INVALID;
    if (hipEventSynchronize(stop->ev) != hipSuccess) return RN_ERR_HIP;
    if (hipEventElapsedTime(ms, start->ev, stop->ev) != hipSuccess) return RN_ERR_HIP;
    return RN_OK;
}

}  // extern "C"
