// Host pipeline: pinned staging, a copy stream, two slots.  The reference uploads one image,
// runs the forward and downloads the logits strictly in sequence with a device
// synchronisation between every step (main.cu:236-240, tensor.cuh:184-199); at 13k images/s a
// 154 MB batch upload (2.4 ms on PCIe Gen5 x16) would cost 13 % if it were not overlapped.
#include <stdlib.h>
#include <string.h>

#include "rn_internal.h"

struct rn_graph {
    rn_ctx *ctx;
    hipGraph_t graph;
    hipGraphExec_t exec;
    size_t nodes;
};

struct rn_pipeline_slot {
    float *h_in, *h_out;  // pinned
    float *d_in, *d_out;
    hipEvent_t uploaded, done;
    int busy;
};

struct rn_pipeline {
    rn_model *model;
    rn_ctx *ctx;
    uint64_t B;
    int mode;
    hipStream_t copy_stream;
    rn_pipeline_slot slot[2];
    uint64_t head, tail;  // submitted / collected counts
};

extern "C" {

// the model keeps its context private; the pipeline needs it for the compute stream
rn_ctx *rn_model_context(rn_model *m);

int rn_model_profiling_enabled(const rn_model *m);

int rn_graph_destroy(rn_graph *g)
{
    if (!g) return RN_OK;
    if (g->ctx) (void)hipStreamSynchronize(g->ctx->stream);
    if (g->exec) {
        (void)hipGraphExecDestroy(g->exec);
        if (g->ctx) --g->ctx->graphs_live;
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
    ++ctx->graphs_live;  // from here on the scratch and the arenas must stay where they are
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
    if (p->ctx) (void)hipStreamSynchronize(p->ctx->stream);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    for (int i = 0; i < 2; ++i) {
        rn_pipeline_slot *s = &p->slot[i];
        if (s->h_in) (void)hipHostFree(s->h_in);
        if (s->h_out) (void)hipHostFree(s->h_out);
        if (s->d_in) (void)hipFree(s->d_in);
        if (s->d_out) (void)hipFree(s->d_out);
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
    const size_t in_bytes = (size_t)B * 3 * 224 * 224 * sizeof(float);
    const size_t out_bytes = (size_t)B * 1000 * sizeof(float);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        rn_pipeline_slot *s = &p->slot[i];
        e = hipHostMalloc((void **)&s->h_in, in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out, out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_in, in_bytes);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_out, out_bytes);
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

int rn_pipeline_input_buffer(rn_pipeline *p, float **host_staging)
{
    if (!p || !host_staging) return RN_ERR_INVALID;
    if (p->head - p->tail >= 2)
        return rn_set_error(p->ctx, RN_ERR_INVALID, "rn_pipeline_input_buffer: both slots busy, collect first");
    *host_staging = p->slot[p->head & 1].h_in;
    return RN_OK;
}

int rn_pipeline_submit(rn_pipeline *p, const float *host_input_nchw)
{
    if (!p) return RN_ERR_INVALID;
    rn_ctx *ctx = p->ctx;
    RN_TRY(rn_bind_device(ctx));
    if (p->head - p->tail >= 2)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_pipeline_submit: both slots busy, collect first");
    rn_pipeline_slot *s = &p->slot[p->head & 1];
    const size_t in_bytes = (size_t)p->B * 3 * 224 * 224 * sizeof(float);
    const size_t out_bytes = (size_t)p->B * 1000 * sizeof(float);
    if (host_input_nchw && host_input_nchw != s->h_in)
        memcpy(s->h_in, host_input_nchw, in_bytes);  // pageable -> pinned
    RN_HIP_TRY(ctx, hipMemcpyAsync(s->d_in, s->h_in, in_bytes, hipMemcpyHostToDevice, p->copy_stream));
    RN_HIP_TRY(ctx, hipEventRecord(s->uploaded, p->copy_stream));
    // forward on the compute stream once the upload has landed; the other slot's forward may
    // still be running there, which is exactly the overlap
    RN_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s->uploaded, 0));
    RN_TRY(rn_model_forward(p->model, s->d_in, p->B, s->d_out, p->mode));
    RN_HIP_TRY(ctx, hipMemcpyAsync(s->h_out, s->d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    RN_HIP_TRY(ctx, hipEventRecord(s->done, ctx->stream));
    s->busy = 1;
    ++p->head;
    return RN_OK;
}

int rn_pipeline_collect(rn_pipeline *p, float *host_logits)
{
    if (!p || !host_logits) return RN_ERR_INVALID;
    rn_ctx *ctx = p->ctx;
    if (p->head == p->tail)
        return rn_set_error(ctx, RN_ERR_INVALID, "rn_pipeline_collect: nothing in flight");
    rn_pipeline_slot *s = &p->slot[p->tail & 1];
    RN_HIP_TRY(ctx, hipEventSynchronize(s->done));
    memcpy(host_logits, s->h_out, (size_t)p->B * 1000 * sizeof(float));
    s->busy = 0;
    ++p->tail;
    return RN_OK;
}

}  // extern "C"
