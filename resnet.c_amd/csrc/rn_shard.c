/*
 * rn_shard -- one batch over several devices of one node.
 *
 * The forward has no cross-image reduction (inference batch-norm uses running statistics,
 * cuda/ops.cu:139-151; every kernel indexes the batch independently), so a batch shards
 * contiguously: device g of G takes images [g*B/G, (g+1)*B/G), weights are replicated, nothing is
 * exchanged between devices, and the logits are concatenated on the host.
 * This is the multi-device form of the reference's main() (cuda/inference/main.cu:228-254):
 * one host thread + one context (device, stream, scratch) + one model per device.  Each
 * context is only ever touched by its own thread; the caller's thread posts a job and waits.
 * Plain C over the C-ABI of rn_hip.h, pthreads, no HIP headers.
 *
 * The same device may be listed more than once (two shards on device 0): that is how the
 * sharding code is exercised on a one-GPU box.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rn_hip.h"

#define IMG_FLOATS ((uint64_t)3 * 224 * 224)
#define RN_CLASSES 1000

enum { JOB_NONE = 0, JOB_CREATE, JOB_SET_TENSOR, JOB_LOAD_DIR, JOB_SET_DTYPE, JOB_FINALIZE,
       JOB_FORWARD, JOB_TUNE, JOB_QUIT };

typedef struct rn_shard_worker {
    struct rn_shard *group;
    pthread_t thread;
    int started;
    int rank, device;
    rn_ctx *ctx;
    rn_model *model;
    float *d_in, *d_logits;
    uint64_t *d_idx;
    uint64_t cap; /* images the device buffers hold */
    uint64_t seen; /* last job sequence number this worker ran */
    int status;
    char err[512];
} rn_shard_worker;

struct rn_shard {
    int n, arch;
    rn_shard_worker *w;
    pthread_mutex_t mu;
    pthread_cond_t cv_job, cv_done;
    uint64_t seq; /* job sequence number */
    int pending;  /* workers still running the current job */
    /* the current job */
    int kind;
    const char *text;       /* key or directory */
    const float *tensor;    /* SET_TENSOR: host data; FORWARD: host input */
    uint64_t numel;         /* SET_TENSOR: element count; FORWARD: B */
    float *logits;
    uint64_t *top1;
    int ivalue; /* dtype or mode */
    char err[640];
};

void rn_shard_bounds(uint64_t B, int rank, int world, uint64_t *lo, uint64_t *hi)
{
    const uint64_t base = B / (uint64_t)world, rem = B % (uint64_t)world, r = (uint64_t)rank;
    const uint64_t l = r * base + (r < rem ? r : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < rem ? 1 : 0);
}

static int fail(rn_shard_worker *w, int st, const char *what)
{
    snprintf(w->err, sizeof(w->err), "shard %d (device %d): %s: %s (%s)", w->rank, w->device, what,
             rn_status_string(st), w->ctx ? rn_last_error(w->ctx) : "no context");
    return st;
}

#define WTRY(w, expr)                                   \
    do {                                                \
        int st_ = (expr);                               \
        if (st_ != RN_OK) return fail((w), st_, #expr); \
    } while (0)

static int ensure_buffers(rn_shard_worker *w, uint64_t nb)
{
    if (nb <= w->cap) return RN_OK;
    if (w->d_in) rn_free(w->ctx, w->d_in);
    if (w->d_logits) rn_free(w->ctx, w->d_logits);
    if (w->d_idx) rn_free(w->ctx, w->d_idx);
    w->d_in = w->d_logits = NULL;
    w->d_idx = NULL;
    w->cap = 0;
    WTRY(w, rn_malloc(w->ctx, (void **)&w->d_in, nb * IMG_FLOATS * sizeof(float)));
    WTRY(w, rn_malloc(w->ctx, (void **)&w->d_logits, nb * RN_CLASSES * sizeof(float)));
    WTRY(w, rn_malloc(w->ctx, (void **)&w->d_idx, nb * sizeof(uint64_t)));
    w->cap = nb;
    return RN_OK;
}

static int run_job(rn_shard_worker *w, const struct rn_shard *g)
{
    switch (g->kind) {
    case JOB_CREATE:
        WTRY(w, rn_ctx_create(&w->ctx, w->device, NULL));
        WTRY(w, rn_model_create(w->ctx, &w->model, g->arch));
        return RN_OK;
    case JOB_SET_TENSOR:
        WTRY(w, rn_model_set_tensor(w->model, g->text, g->tensor, g->numel));
        return RN_OK;
    case JOB_LOAD_DIR:
        WTRY(w, rn_model_load_dir(w->model, g->text));
        return RN_OK;
    case JOB_SET_DTYPE:
        WTRY(w, rn_model_set_dtype(w->model, g->ivalue));
        return RN_OK;
    case JOB_FINALIZE:
        WTRY(w, rn_model_finalize(w->model));
        return RN_OK;
    case JOB_FORWARD:
    case JOB_TUNE: {
        uint64_t lo, hi, nb;
        rn_shard_bounds(g->numel, w->rank, g->n, &lo, &hi);
        nb = hi - lo;
        if (nb == 0) return RN_OK;
        WTRY(w, ensure_buffers(w, nb));
        /* upload of this shard's images, forward, download of its logits / class indices:
         * all on this device's stream, beside the other devices' */
        WTRY(w, rn_memcpy_h2d(w->ctx, w->d_in, g->tensor + lo * IMG_FLOATS,
                              nb * IMG_FLOATS * sizeof(float)));
        if (g->kind == JOB_TUNE) {
            WTRY(w, rn_model_tune(w->model, w->d_in, nb, w->d_logits, g->ivalue));
            WTRY(w, rn_sync(w->ctx));
            return RN_OK;
        }
        WTRY(w, rn_model_forward(w->model, w->d_in, nb, w->d_logits, g->ivalue));
        if (g->top1) {
            /* first maximum wins, as the reference's host loop (main.cu:243-249) */
            WTRY(w, rn_argmax_forward(w->ctx, w->d_logits, w->d_idx, nb, RN_CLASSES));
            WTRY(w, rn_memcpy_d2h(w->ctx, g->top1 + lo, w->d_idx, nb * sizeof(uint64_t)));
        }
        if (g->logits)
            WTRY(w, rn_memcpy_d2h(w->ctx, g->logits + lo * RN_CLASSES, w->d_logits,
                                  nb * RN_CLASSES * sizeof(float)));
        WTRY(w, rn_sync(w->ctx));
        return RN_OK;
    }
    default:
        return RN_OK;
    }
}

static void release_device_state(rn_shard_worker *w)
{
    if (w->ctx) {
        if (w->d_in) rn_free(w->ctx, w->d_in);
        if (w->d_logits) rn_free(w->ctx, w->d_logits);
        if (w->d_idx) rn_free(w->ctx, w->d_idx);
    }
    if (w->model) rn_model_destroy(w->model);
    if (w->ctx) rn_ctx_destroy(w->ctx);
    w->model = NULL;
    w->ctx = NULL;
}

static void *worker_main(void *arg)
{
    rn_shard_worker *w = (rn_shard_worker *)arg;
    struct rn_shard *g = w->group;
    for (;;) {
        int kind, st;
        pthread_mutex_lock(&g->mu);
        while (g->seq == w->seen) pthread_cond_wait(&g->cv_job, &g->mu);
        w->seen = g->seq;
        kind = g->kind;
        pthread_mutex_unlock(&g->mu);
        /* the job fields are stable until every worker has reported back */
        if (kind == JOB_QUIT) {
            release_device_state(w);
            st = RN_OK;
        } else {
            st = run_job(w, g);
        }
        pthread_mutex_lock(&g->mu);
        w->status = st;
        if (--g->pending == 0) pthread_cond_signal(&g->cv_done);
        pthread_mutex_unlock(&g->mu);
        if (kind == JOB_QUIT) return NULL;
    }
}

/* post one job to every worker, wait for all, return the first failure */
static int post(struct rn_shard *g, int kind)
{
    int i, st = RN_OK;
    pthread_mutex_lock(&g->mu);
    g->kind = kind;
    g->pending = g->n;
    ++g->seq;
    pthread_cond_broadcast(&g->cv_job);
    while (g->pending > 0) pthread_cond_wait(&g->cv_done, &g->mu);
    pthread_mutex_unlock(&g->mu);
    for (i = 0; i < g->n; ++i) {
        if (g->w[i].status != RN_OK && st == RN_OK) {
            st = g->w[i].status;
            snprintf(g->err, sizeof(g->err), "%s", g->w[i].err);
        }
    }
    return st;
}

int rn_shard_destroy(rn_shard *g)
{
    int i;
    if (!g) return RN_OK;
    if (g->w) {
        int live = 0;
        for (i = 0; i < g->n; ++i) live += g->w[i].started;
        if (live == g->n) {
            post(g, JOB_QUIT);
            for (i = 0; i < g->n; ++i) pthread_join(g->w[i].thread, NULL);
        }
        free(g->w);
    }
    pthread_cond_destroy(&g->cv_done);
    pthread_cond_destroy(&g->cv_job);
    pthread_mutex_destroy(&g->mu);
    free(g);
    return RN_OK;
}

int rn_shard_create(rn_shard **out, const int *devices, int n_devices, int arch)
{
    struct rn_shard *g;
    int i, st;
    if (!out || !devices || n_devices < 1 || n_devices > 64) return RN_ERR_INVALID;
    *out = NULL;
    g = (struct rn_shard *)calloc(1, sizeof(*g));
    if (!g) return RN_ERR_NOMEM;
    g->n = n_devices;
    g->arch = arch;
    pthread_mutex_init(&g->mu, NULL);
    pthread_cond_init(&g->cv_job, NULL);
    pthread_cond_init(&g->cv_done, NULL);
    g->w = (rn_shard_worker *)calloc((size_t)n_devices, sizeof(rn_shard_worker));
    if (!g->w) {
        rn_shard_destroy(g);
        return RN_ERR_NOMEM;
    }
    for (i = 0; i < n_devices; ++i) {
        g->w[i].group = g;
        g->w[i].rank = i;
        g->w[i].device = devices[i];
    }
    for (i = 0; i < n_devices; ++i) {
        if (pthread_create(&g->w[i].thread, NULL, worker_main, &g->w[i]) != 0) {
            /* threads already started wait for a job that never comes: tell them to quit */
            int k;
            g->n = i;
            if (i > 0) {
                for (k = 0; k < i; ++k) g->w[k].started = 1;
            }
            rn_shard_destroy(g);
            return RN_ERR_NOMEM;
        }
        g->w[i].started = 1;
    }
    st = post(g, JOB_CREATE);
    if (st != RN_OK) {
        fprintf(stderr, "rn_shard_create: %s\n", g->err);
        rn_shard_destroy(g);
        return st;
    }
    *out = g;
    return RN_OK;
}

int rn_shard_count(const rn_shard *g) { return g ? g->n : 0; }
const char *rn_shard_last_error(const rn_shard *g) { return g ? g->err : "no shard group"; }

int rn_shard_set_tensor(rn_shard *g, const char *key, const float *host_data, uint64_t numel)
{
    if (!g || !key || !host_data) return RN_ERR_INVALID;
    g->text = key;
    g->tensor = host_data;
    g->numel = numel;
    return post(g, JOB_SET_TENSOR);
}

int rn_shard_load_dir(rn_shard *g, const char *weights_dir)
{
    if (!g || !weights_dir) return RN_ERR_INVALID;
    g->text = weights_dir;
    return post(g, JOB_LOAD_DIR);
}

int rn_shard_set_dtype(rn_shard *g, int dtype)
{
    if (!g) return RN_ERR_INVALID;
    g->ivalue = dtype;
    return post(g, JOB_SET_DTYPE);
}

int rn_shard_finalize(rn_shard *g)
{
    if (!g) return RN_ERR_INVALID;
    return post(g, JOB_FINALIZE);
}

static int forward_like(rn_shard *g, int kind, const float *host_input_nchw, uint64_t B,
                        float *host_logits, uint64_t *host_top1, int mode)
{
    if (!g || !host_input_nchw || B == 0) return RN_ERR_INVALID;
    g->tensor = host_input_nchw;
    g->numel = B;
    g->logits = host_logits;
    g->top1 = host_top1;
    g->ivalue = mode;
    return post(g, kind);
}

int rn_shard_forward(rn_shard *g, const float *host_input_nchw, uint64_t B, float *host_logits,
                     uint64_t *host_top1, int mode)
{
    return forward_like(g, JOB_FORWARD, host_input_nchw, B, host_logits, host_top1, mode);
}

int rn_shard_tune(rn_shard *g, const float *host_input_nchw, uint64_t B, int mode)
{
    return forward_like(g, JOB_TUNE, host_input_nchw, B, NULL, NULL, mode);
}
