// Host pipeline: pinned staging, a copy stream, two slots.  The reference uploads one image,
// runs the forward and downloads the logits strictly in sequence with a device
// synchronisation between every step (main.cu:236-240, tensor.cuh:184-199); at 13k images/s a
// 154 MB batch upload (2.4 ms on PCIe Gen5 x16) would cost 13 % if it were not overlapped.
#include <sched.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "rn_internal.h"

struct rn_graph {
    rn_ctx *ctx;
    rn_model *model;  // counted in the model: rn_model_destroy refuses while a graph of it lives
    rn_ctx *pinned[8];  // every context the model queued batch parts on: pinned while the graph lives
    int n_pinned;
    hipGraph_t graph;
    hipGraphExec_t exec;
    size_t nodes;
};

struct rn_pipeline_slot {
    float *h_in, *h_out;  // pinned
    float *d_in, *d_out;
    uint64_t *h_idx, *d_idx;  // class indices (first maximum wins, main.cu:243-249)
    uint64_t n;               // images of the batch in this slot (<= B)
    hipEvent_t uploaded, done;
    int busy;
};

struct rn_pipeline {
    rn_model *model;
    rn_ctx *ctx;
    uint64_t B;
    int mode;
    hipStream_t copy_stream;
    int copy_threads;  // helper threads of the pageable -> pinned copy (RN_COPY_THREADS, default 3)
    rn_pipeline_slot slot[2];
    uint64_t head, tail;  // submitted / collected counts
};

extern "C" {

// the model keeps its context private; the pipeline needs it for the compute stream
rn_ctx *rn_model_context(rn_model *m);

int rn_model_profiling_enabled(const rn_model *m);
// the contexts the model has queued batch parts on so far (rn_model.c)
int rn_model_contexts(rn_model *m, rn_ctx **out, int cap);
void rn_model_graph_ref(rn_model *m, int delta);

int rn_graph_destroy(rn_graph *g)
{
    if (!g) return RN_OK;
    if (g->ctx) (void)hipStreamSynchronize(g->ctx->stream);
    if (g->exec) {
        (void)hipGraphExecDestroy(g->exec);
        // the pinned contexts are alive: those of the model's extra streams belong to the model, and
        // the model cannot be destroyed while this graph is counted in it
        for (int i = 0; i < g->n_pinned; ++i) --g->pinned[i]->graphs_live;
        if (g->n_pinned) rn_model_graph_ref(g->model, -1);
    }
    if (g->graph) (void)hipGraphDestroy(g->graph);
    free(g);
    return RN_OK;
}

int rn_model_capture(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode,
                     rn_graph **out)
{
    if (!m || !out) return RN_ERR_INVALID;
    *out = nullptr;
    rn_ctx *ctx = rn_model_context(m);
    RN_TRY(rn_bind_device(ctx));
    if (ctx->sync_each_op)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_model_capture: sync_each_op must be off");
    if (rn_model_profiling_enabled(m))
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_model_capture: profiling must be off");
    // eager pass: validates the arguments, sizes the arenas and the scratch outside the capture
    RN_TRY(rn_model_forward(m, input_nchw, B, logits, mode));
    RN_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rn_graph *g = (rn_graph *)calloc(1, sizeof(rn_graph));
    if (!g) return RN_ERR_NOMEM;
    g->ctx = ctx;
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        free(g);
        return rn_check_hip(ctx, e, "hipStreamBeginCapture");
    }
    const int st = rn_model_forward(m, input_nchw, B, logits, mode);
    e = hipStreamEndCapture(ctx->stream, &g->graph);  // always end the capture, even on error
    if (st != RN_OK || e != hipSuccess) {
        rn_graph_destroy(g);
        return st != RN_OK ? st : rn_check_hip(ctx, e, "hipStreamEndCapture");
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        g->exec = nullptr;
        rn_graph_destroy(g);
        return rn_check_hip(ctx, e, "hipGraphInstantiate");
    }
    // from here on the scratch and the arenas must stay where they are: the captured nodes point
    // into the scratch of the primary context AND of the contexts of the other batch parts
    g->n_pinned = rn_model_contexts(m, g->pinned, 8);
    for (int i = 0; i < g->n_pinned; ++i) ++g->pinned[i]->graphs_live;
    g->model = m;
    rn_model_graph_ref(m, +1);
    e = hipGraphGetNodes(g->graph, nullptr, &g->nodes);
    if (e != hipSuccess) {
        rn_graph_destroy(g);
        return rn_check_hip(ctx, e, "hipGraphGetNodes");
    }
    *out = g;
    return RN_OK;
}

int rn_graph_launch(rn_graph *g)
{
    if (!g) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(g->ctx));
    RN_HIP_TRY(g->ctx, hipGraphLaunch(g->exec, g->ctx->stream));
    return RN_OK;
}

uint64_t rn_graph_node_count(const rn_graph *g) { return g ? (uint64_t)g->nodes : 0; }

int rn_pipeline_destroy(rn_pipeline *p)
{
    if (!p) return RN_OK;
    if (p->ctx) (void)hipSetDevice(p->ctx->device);
    if (p->ctx) (void)hipStreamSynchronize(p->ctx->stream);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    for (int i = 0; i < 2; ++i) {
        rn_pipeline_slot *s = &p->slot[i];
        if (s->h_in) (void)hipHostFree(s->h_in);
        if (s->h_out) (void)hipHostFree(s->h_out);
        if (s->h_idx) (void)hipHostFree(s->h_idx);
        if (s->d_in) (void)hipFree(s->d_in);
        if (s->d_out) (void)hipFree(s->d_out);
        if (s->d_idx) (void)hipFree(s->d_idx);
        if (s->uploaded) (void)hipEventDestroy(s->uploaded);
        if (s->done) (void)hipEventDestroy(s->done);
    }
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    free(p);
    return RN_OK;
}

int rn_pipeline_create(rn_model *m, rn_pipeline **out, uint64_t B, int mode)
{
    if (!m || !out || B == 0) return RN_ERR_INVALID;
    *out = nullptr;
    rn_ctx *ctx = rn_model_context(m);
    rn_pipeline *p = (rn_pipeline *)calloc(1, sizeof(rn_pipeline));
    if (!p) return RN_ERR_NOMEM;
    p->model = m;
    p->ctx = ctx;
    p->B = B;
    p->mode = mode;
    {
        const char *ct = getenv("RN_COPY_THREADS");
        const int n = ct ? atoi(ct) : 3;
        p->copy_threads = n < 1 ? 1 : n > 16 ? 16 : n;
    }
    const size_t in_bytes = (size_t)B * 3 * 224 * 224 * sizeof(float);
    const size_t out_bytes = (size_t)B * 1000 * sizeof(float);
    const size_t idx_bytes = (size_t)B * sizeof(uint64_t);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        rn_pipeline_slot *s = &p->slot[i];
        e = hipHostMalloc((void **)&s->h_in, in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out, out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_idx, idx_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_in, in_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_out, out_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_idx, idx_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        const int st = rn_check_hip(ctx, e, "rn_pipeline_create");
        rn_pipeline_destroy(p);
        return st;
    }
    *out = p;
    return RN_OK;
}

uint64_t rn_pipeline_in_flight(const rn_pipeline *p) { return p ? p->head - p->tail : 0; }
uint64_t rn_pipeline_batch(const rn_pipeline *p) { return p ? p->B : 0; }

int rn_pipeline_input_buffer(rn_pipeline *p, float **host_staging)
{
    if (!p || !host_staging) return RN_ERR_INVALID;
    if (p->head - p->tail >= 2)
        return rn_set_error(p->ctx, RN_ERR_INVALID, "rn_pipeline_input_buffer: both slots busy, collect first");
    *host_staging = p->slot[p->head & 1].h_in;
    return RN_OK;
}

int rn_pipeline_submit_n(rn_pipeline *p, const float *host_input_nchw, uint64_t n)
{
    if (!p) return RN_ERR_INVALID;
    rn_ctx *ctx = p->ctx;
    RN_TRY(rn_bind_device(ctx));
    if (n == 0 || n > p->B)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_pipeline_submit_n: %llu images, the pipeline holds 1..%llu",
                            (unsigned long long)n, (unsigned long long)p->B);
    if (p->head - p->tail >= 2)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_pipeline_submit: both slots busy, collect first");
    rn_pipeline_slot *s = &p->slot[p->head & 1];
    const size_t in_bytes = (size_t)n * 3 * 224 * 224 * sizeof(float);
    const int st = [&]() -> int {
        if (host_input_nchw && host_input_nchw != s->h_in) {
            // pageable -> pinned -> device in pieces of 16 images (9.6 MB): the upload of piece i runs on the
            // copy stream while piece i+1 is being copied, so a batch costs max(copy, upload) instead of their
            // sum before its forward can start.  One core copies a 154 MB fp32 batch in 6-7 ms (23 GB/s), which
            // is longer than its upload (3.8 ms) and than a bf16 forward (3.3 ms): the pieces are copied by
            // `copiers` helper threads (they inherit this thread's CPU affinity: the cores local to the device)
            // while this thread queues each piece's upload as soon as it has landed in staging.
            const size_t piece = (size_t)16 * 3 * 224 * 224 * sizeof(float);
            const size_t npieces = (in_bytes + piece - 1) / piece;
            const int copiers = npieces >= 4 ? p->copy_threads : 1;
            auto copy_piece = [&](size_t i) {
                const size_t at = i * piece, nb = in_bytes - at < piece ? in_bytes - at : piece;
                memcpy((char *)s->h_in + at, (const char *)host_input_nchw + at, nb);
            };
            auto upload_piece = [&](size_t i) -> hipError_t {
                const size_t at = i * piece, nb = in_bytes - at < piece ? in_bytes - at : piece;
                return hipMemcpyAsync((char *)s->d_in + at, (char *)s->h_in + at, nb, hipMemcpyHostToDevice, p->copy_stream);
            };
            if (copiers <= 1) {
                for (size_t i = 0; i < npieces; ++i) {
                    copy_piece(i);
                    RN_HIP_TRY(ctx, upload_piece(i));
                }
            } else {
                std::vector<std::atomic<int>> landed(npieces);
                for (auto &f : landed) f.store(0, std::memory_order_relaxed);
                std::vector<std::thread> pool;
                int started = 0;  // a thread that cannot be created leaves its pieces to this thread
                try {
                    for (int t = 0; t < copiers; ++t) {
                        pool.emplace_back([&, t]() {
                            for (size_t i = (size_t)t; i < npieces; i += (size_t)copiers) {
                                copy_piece(i);
                                landed[i].store(1, std::memory_order_release);
                            }
                        });
                        ++started;
                    }
                } catch (...) {
                }
                hipError_t e = hipSuccess;
                for (size_t i = 0; i < npieces && e == hipSuccess; ++i) {
                    if ((int)(i % (size_t)copiers) >= started)
                        copy_piece(i);
                    else
                        while (!landed[i].load(std::memory_order_acquire)) sched_yield();
                    e = upload_piece(i);
                }
                for (auto &th : pool) th.join();  // also on an error: the threads write into the slot's staging
                RN_HIP_TRY(ctx, e);
            }
        } else {
            RN_HIP_TRY(ctx, hipMemcpyAsync(s->d_in, s->h_in, in_bytes, hipMemcpyHostToDevice, p->copy_stream));
        }
        RN_HIP_TRY(ctx, hipEventRecord(s->uploaded, p->copy_stream));
        // forward on the compute stream once the upload has landed; the other slot's forward may
        // still be running there, which is exactly the overlap
        RN_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->uploaded, 0));
        RN_TRY(rn_model_forward(p->model, s->d_in, n, s->d_out, p->mode));
        RN_TRY(rn_argmax_forward(ctx, s->d_out, s->d_idx, n, 1000));
        RN_HIP_TRY(ctx, hipMemcpyAsync(s->h_out, s->d_out, (size_t)n * 1000 * sizeof(float), hipMemcpyDeviceToHost,
                                       ctx->stream));
        RN_HIP_TRY(ctx, hipMemcpyAsync(s->h_idx, s->d_idx, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                       ctx->stream));
        RN_HIP_TRY(ctx, hipEventRecord(s->done, ctx->stream));
        return RN_OK;
    }();
    if (st != RN_OK) {
        // the slot stays free, so the next submit writes its staging buffer again: whatever part of
        // this batch was queued (the upload reads h_in) has to be over before that
        (void)hipStreamSynchronize(p->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        return st;
    }
    s->busy = 1;
    s->n = n;
    ++p->head;
    return RN_OK;
}

int rn_pipeline_submit(rn_pipeline *p, const float *host_input_nchw)
{
    return p ? rn_pipeline_submit_n(p, host_input_nchw, p->B) : RN_ERR_INVALID;
}

int rn_pipeline_collect_n(rn_pipeline *p, float *host_logits, uint64_t *host_top1, uint64_t *n)
{
    if (!p) return RN_ERR_INVALID;
    rn_ctx *ctx = p->ctx;
    if (p->head == p->tail)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_pipeline_collect: nothing in flight");
    rn_pipeline_slot *s = &p->slot[p->tail & 1];
    RN_TRY(rn_bind_device(ctx));
    RN_HIP_TRY(ctx, hipEventSynchronize(s->done));
    if (host_logits) memcpy(host_logits, s->h_out, (size_t)s->n * 1000 * sizeof(float));
    if (host_top1) memcpy(host_top1, s->h_idx, (size_t)s->n * sizeof(uint64_t));
    if (n) *n = s->n;
    s->busy = 0;
    ++p->tail;
    return RN_OK;
}

int rn_pipeline_collect(rn_pipeline *p, float *host_logits)
{
    if (!p || !host_logits) return RN_ERR_INVALID;
    return rn_pipeline_collect_n(p, host_logits, nullptr, nullptr);
}

}  // extern "C"
