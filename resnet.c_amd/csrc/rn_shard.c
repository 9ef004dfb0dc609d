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
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET */
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rn_hip.h"

#define IMG_FLOATS ((uint64_t)3 * 224 * 224)
#define RN_CLASSES 1000

enum { JOB_NONE = 0, JOB_CREATE, JOB_SET_TENSOR, JOB_LOAD_DIR, JOB_SET_DTYPE, JOB_FINALIZE,
       JOB_FORWARD, JOB_TUNE, JOB_TUNE_REST, JOB_STREAM_OPEN, JOB_SUBMIT, JOB_COLLECT, JOB_STREAM_CLOSE, JOB_QUIT };

/* A shard's images go through the device in chunks of at most this many: the upload of chunk
 * i+1 (pinned staging, copy stream) runs beside the forward of chunk i (rn_pipeline_*).  Measured
 * on one device, 256 images per call (tools/shard_rate.py): chunks of 256 / 128 / 64 give
 * 9.2k / 11.0k / 11.8k images/s in fp32 and 20k / 30.6k / 26.6k with bf16 storage (smaller chunks
 * overlap more of the upload and run the forward on less efficient launches). */
#define RN_SHARD_CHUNK 128

typedef struct rn_shard_worker {
    struct rn_shard *group;
    pthread_t thread;
    int started;
    int rank, device;
    rn_ctx *ctx;
    rn_model *model;
    float *d_in, *d_logits; /* tuning only: the forward itself runs out of the pipeline's slots */
    uint64_t cap; /* images the device buffers hold */
    /* pinned staging + copy stream + two slots on this device: upload, forward and download of
     * consecutive chunks (rn_shard_forward) or batches (rn_shard_submit / _collect) overlap */
    rn_pipeline *pipe;
    uint64_t pipe_B;
    int pipe_mode;
    uint64_t stream_lo, stream_hi; /* streaming form: this shard's image range of every batch */
    uint64_t seen; /* last job sequence number this worker ran */
    int numa_node;     /* of the device's PCI slot, -1 = unknown */
    char cpus[256];    /* the cores this thread was bound to, "" = not bound */
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
    uint64_t *tuning;     /* rn_shard_tune: shard 0's table on its way to the others */
    uint64_t tuning_words, tuning_nb; /* its size; the launch batch it was measured at */
    uint64_t stream_B;    /* streaming form: batch size of rn_shard_stream_open, 0 = closed */
    int stream_in_flight; /* batches submitted and not yet collected (0..2) */
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
    w->d_in = w->d_logits = NULL;
    w->cap = 0;
    WTRY(w, rn_malloc(w->ctx, (void **)&w->d_in, nb * IMG_FLOATS * sizeof(float)));
    WTRY(w, rn_malloc(w->ctx, (void **)&w->d_logits, nb * RN_CLASSES * sizeof(float)));
    w->cap = nb;
    return RN_OK;
}

/* forget whatever is still in flight on this device (after a failure, or when a stream is re-opened):
 * the batches finish, their results go nowhere */
static void drain(rn_shard_worker *w)
{
    while (w->pipe && rn_pipeline_in_flight(w->pipe) > 0) {
        if (rn_pipeline_collect_n(w->pipe, NULL, NULL, NULL) != RN_OK) break;
    }
}

/* the worker's pipeline for batches of up to B images in `mode` (rebuilt when either changes) */
static int ensure_pipeline(rn_shard_worker *w, uint64_t B, int mode)
{
    drain(w);
    if (w->pipe && w->pipe_B >= B && w->pipe_mode == mode) return RN_OK;
    if (w->pipe) {
        rn_pipeline_destroy(w->pipe);
        w->pipe = NULL;
    }
    WTRY(w, rn_pipeline_create(w->model, &w->pipe, B, mode));
    w->pipe_B = B;
    w->pipe_mode = mode;
    return RN_OK;
}

/* one collected chunk goes to rows [at, at + n) of the caller's arrays */
static int collect_chunk(rn_shard_worker *w, const struct rn_shard *g, uint64_t at, uint64_t *n)
{
    WTRY(w, rn_pipeline_collect_n(w->pipe, g->logits ? g->logits + at * RN_CLASSES : NULL,
                                  g->top1 ? g->top1 + at : NULL, n));
    return RN_OK;
}

/* the launch batch rn_shard_tune measures for on this shard: with a stream open a whole shard per
 * launch (rn_shard_submit), otherwise the chunks rn_shard_forward issues */
static uint64_t tune_batch(const rn_shard_worker *w, const struct rn_shard *g)
{
    uint64_t lo, hi, nb;
    rn_shard_bounds(g->numel, w->rank, g->n, &lo, &hi);
    nb = hi - lo;
    if (g->stream_B == 0 && nb > RN_SHARD_CHUNK) nb = RN_SHARD_CHUNK;
    return nb;
}

/* "0-3,8,10-11" -> cpu set; returns the number of CPUs named */
static int parse_cpulist(const char *text, cpu_set_t *set)
{
    int n = 0;
    const char *p = text;
    CPU_ZERO(set);
    while (*p) {
        char *end;
        long a = strtol(p, &end, 10), b;
        if (end == p) break;
        b = a;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
            if (end == p) break;
        }
        for (; a <= b && a < CPU_SETSIZE; ++a) {
            if (a >= 0) {
                CPU_SET((int)a, set);
                ++n;
            }
        }
        p = *end == ',' ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return n;
}

/* bind this worker's thread to the cores local to its device that the process may use (best effort) */
static void bind_near_device(rn_shard_worker *w)
{
    char local[256] = {0};
    cpu_set_t near_set, allowed, both;
    const char *off = getenv("RN_SHARD_AFFINITY");
    int cpu, n = 0, first = -1, last = -1;
    size_t at = 0;
    w->numa_node = -1;
    w->cpus[0] = 0;
    if (rn_device_locality(w->device, NULL, 0, &w->numa_node, local, sizeof(local)) != RN_OK) return;
    if ((off && off[0] == '0') || !local[0] || parse_cpulist(local, &near_set) == 0) return;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    CPU_AND(&both, &near_set, &allowed);
    if (CPU_COUNT(&both) == 0) return; /* the process was confined elsewhere: leave it there */
    if (pthread_setaffinity_np(pthread_self(), sizeof(both), &both) != 0) return;
    /* what it got, as a list of ranges */
    for (cpu = 0; cpu <= CPU_SETSIZE; ++cpu) {
        const int in = cpu < CPU_SETSIZE && CPU_ISSET(cpu, &both);
        if (in && first < 0) first = cpu;
        if (in) last = cpu;
        if (!in && first >= 0) {
            int k = snprintf(w->cpus + at, sizeof(w->cpus) - at, first == last ? "%s%d" : "%s%d-%d", n ? "," : "",
                             first, last);
            if (k < 0 || (size_t)k >= sizeof(w->cpus) - at) break;
            at += (size_t)k;
            ++n;
            first = -1;
        }
    }
}

static int run_job(rn_shard_worker *w, const struct rn_shard *g)
{
    switch (g->kind) {
    case JOB_CREATE:
        WTRY(w, rn_ctx_create(&w->ctx, w->device, NULL));
        bind_near_device(w);
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
    case JOB_TUNE:
    case JOB_TUNE_REST: {
        /* JOB_TUNE: shard 0 measures alone (a second shard on the same device, or on the same host
         * memory path, would be in its timings); JOB_TUNE_REST: the others take its table over when
         * their share is the same, and measure for themselves when it is not (uneven split) */
        const uint64_t nb = tune_batch(w, g);
        if ((g->kind == JOB_TUNE) != (w->rank == 0) || nb == 0) return RN_OK;
        if (g->kind == JOB_TUNE_REST && g->tuning && nb == g->tuning_nb &&
            rn_model_import_tuning(w->model, g->tuning, g->tuning_words) == RN_OK)
            return RN_OK;
        {
            uint64_t lo, hi;
            rn_shard_bounds(g->numel, w->rank, g->n, &lo, &hi);
            WTRY(w, ensure_buffers(w, nb));
            WTRY(w, rn_memcpy_h2d(w->ctx, w->d_in, g->tensor + lo * IMG_FLOATS, nb * IMG_FLOATS * sizeof(float)));
            WTRY(w, rn_model_tune(w->model, w->d_in, nb, w->d_logits, g->ivalue));
            WTRY(w, rn_sync(w->ctx));
        }
        return RN_OK;
    }
    case JOB_FORWARD: {
        /* this shard's images in chunks through the device's pipeline, two in flight: chunk i+1
         * is copied into pinned staging and uploaded on the copy stream while chunk i runs;
         * logits and class indices (first maximum wins, main.cu:243-249) come back per chunk */
        uint64_t lo, hi, sent, got, n;
        rn_shard_bounds(g->numel, w->rank, g->n, &lo, &hi);
        if (hi == lo) return RN_OK;
        WTRY(w, ensure_pipeline(w, hi - lo < RN_SHARD_CHUNK ? hi - lo : RN_SHARD_CHUNK, g->ivalue));
        for (sent = got = lo; got < hi;) {
            int st;
            if (sent < hi && rn_pipeline_in_flight(w->pipe) < 2) {
                n = hi - sent < RN_SHARD_CHUNK ? hi - sent : RN_SHARD_CHUNK;
                st = rn_pipeline_submit_n(w->pipe, g->tensor + sent * IMG_FLOATS, n);
                if (st != RN_OK) {
                    fail(w, st, "rn_pipeline_submit_n");
                    drain(w); /* the chunks already queued finish; nothing stays in flight behind an error */
                    return st;
                }
                sent += n;
                continue;
            }
            st = collect_chunk(w, g, got, &n);
            if (st != RN_OK) {
                drain(w);
                return st;
            }
            got += n;
        }
        return RN_OK;
    }
    case JOB_STREAM_OPEN: {
        rn_shard_bounds(g->numel, w->rank, g->n, &w->stream_lo, &w->stream_hi);
        if (w->stream_hi == w->stream_lo) return RN_OK;
        WTRY(w, ensure_pipeline(w, w->stream_hi - w->stream_lo, g->ivalue));
        return RN_OK;
    }
    case JOB_SUBMIT: {
        const uint64_t nb = w->stream_hi - w->stream_lo;
        if (nb == 0) return RN_OK;
        if (!w->pipe) return fail(w, RN_ERR_INVALID, "rn_shard_submit before rn_shard_stream_open");
        /* NULL: the caller filled the pinned staging buffers (rn_shard_stream_buffer) in place */
        WTRY(w, rn_pipeline_submit_n(w->pipe, g->tensor ? g->tensor + w->stream_lo * IMG_FLOATS : NULL, nb));
        return RN_OK;
    }
    case JOB_COLLECT: {
        uint64_t n;
        if (w->stream_hi == w->stream_lo) return RN_OK;
        if (!w->pipe) return fail(w, RN_ERR_INVALID, "rn_shard_collect before rn_shard_stream_open");
        return collect_chunk(w, g, w->stream_lo, &n);
    }
    case JOB_STREAM_CLOSE:
        if (w->pipe) rn_pipeline_destroy(w->pipe);
        w->pipe = NULL;
        w->stream_lo = w->stream_hi = 0;
        return RN_OK;
    default:
        return RN_OK;
    }
}

static void release_device_state(rn_shard_worker *w)
{
    if (w->ctx) {
        if (w->pipe) rn_pipeline_destroy(w->pipe);
        if (w->d_in) rn_free(w->ctx, w->d_in);
        if (w->d_logits) rn_free(w->ctx, w->d_logits);
    }
    w->pipe = NULL;
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
    free(g->tuning);
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
    if (g && g->stream_in_flight > 0) {
        snprintf(g->err, sizeof(g->err), "rn_shard_forward: %d submitted batch(es) in flight: collect them first",
                 g->stream_in_flight);
        return RN_ERR_INVALID;
    }
    if (g) g->stream_B = 0; /* the per-device pipelines are re-sized for this call's chunks */
    return forward_like(g, JOB_FORWARD, host_input_nchw, B, host_logits, host_top1, mode);
}

int rn_shard_tune(rn_shard *g, const float *host_input_nchw, uint64_t B, int mode)
{
    int st;
    if (g && g->stream_in_flight > 0) {
        snprintf(g->err, sizeof(g->err), "rn_shard_tune: %d submitted batch(es) in flight: collect them first",
                 g->stream_in_flight);
        return RN_ERR_INVALID;
    }
    if (g && g->stream_B != 0 && B != g->stream_B) {
        snprintf(g->err, sizeof(g->err), "rn_shard_tune: the open stream runs batches of %llu images, not %llu",
                 (unsigned long long)g->stream_B, (unsigned long long)B);
        return RN_ERR_INVALID;
    }
    st = forward_like(g, JOB_TUNE, host_input_nchw, B, NULL, NULL, mode); /* shard 0 measures */
    if (st != RN_OK) return st;
    /* its table, read from this thread while the workers are parked */
    free(g->tuning);
    g->tuning = NULL;
    g->tuning_words = 0;
    g->tuning_nb = tune_batch(&g->w[0], g);
    if (g->n > 1 && rn_model_export_tuning(g->w[0].model, NULL, 0, &g->tuning_words) == RN_OK) {
        g->tuning = (uint64_t *)malloc(g->tuning_words * sizeof(uint64_t));
        if (g->tuning &&
            rn_model_export_tuning(g->w[0].model, g->tuning, g->tuning_words, &g->tuning_words) != RN_OK) {
            free(g->tuning);
            g->tuning = NULL;
        }
    }
    st = g->n > 1 ? post(g, JOB_TUNE_REST) : RN_OK;
    free(g->tuning);
    g->tuning = NULL;
    return st;
}

/* between calls on the group the workers are parked: shard `rank`'s model for settings and queries */
rn_model *rn_shard_model(rn_shard *g, int rank) { return g && rank >= 0 && rank < g->n ? g->w[rank].model : NULL; }

int rn_shard_placement(const rn_shard *g, int rank, int *device, int *numa_node, char *cpulist, uint64_t cpulist_cap)
{
    if (!g || rank < 0 || rank >= g->n) return RN_ERR_INVALID;
    if (device) *device = g->w[rank].device;
    if (numa_node) *numa_node = g->w[rank].numa_node;
    if (cpulist && cpulist_cap) snprintf(cpulist, (size_t)cpulist_cap, "%s", g->w[rank].cpus);
    return RN_OK;
}

/* ---- streaming form: consecutive batches of B images, two in flight per device ---------- */
int rn_shard_stream_open(rn_shard *g, uint64_t B, int mode)
{
    int st;
    if (!g || B == 0) return RN_ERR_INVALID;
    g->numel = B;
    g->ivalue = mode;
    g->stream_B = B;
    g->stream_in_flight = 0;
    st = post(g, JOB_STREAM_OPEN);
    if (st != RN_OK) g->stream_B = 0; /* a device without its pipeline: the stream is not open */
    return st;
}

int rn_shard_stream_close(rn_shard *g)
{
    if (!g) return RN_ERR_INVALID;
    g->stream_B = 0;
    g->stream_in_flight = 0;
    return post(g, JOB_STREAM_CLOSE);
}

int rn_shard_stream_buffer(rn_shard *g, int rank, float **host_staging, uint64_t *lo, uint64_t *hi)
{
    rn_shard_worker *w;
    if (!g || rank < 0 || rank >= g->n || !host_staging) return RN_ERR_INVALID;
    w = &g->w[rank];
    *host_staging = NULL;
    if (lo) *lo = w->stream_lo;
    if (hi) *hi = w->stream_hi;
    if (g->stream_B == 0) return RN_ERR_INVALID;
    if (w->stream_hi == w->stream_lo) return RN_OK; /* an empty shard has no staging */
    /* between jobs the worker is parked: reading its pipeline from the caller's thread is safe */
    return rn_pipeline_input_buffer(w->pipe, host_staging);
}

/* A device that fails in the middle of a stream leaves the devices out of step (some hold a batch
 * the others do not): the stream is closed -- open it again (that drains every device) to go on. */
int rn_shard_submit(rn_shard *g, const float *host_input_nchw)
{
    int st;
    if (!g || g->stream_B == 0 || g->stream_in_flight >= 2) return RN_ERR_INVALID;
    g->tensor = host_input_nchw;
    st = post(g, JOB_SUBMIT);
    if (st == RN_OK) ++g->stream_in_flight;
    else g->stream_B = 0, g->stream_in_flight = 0;
    return st;
}

int rn_shard_collect(rn_shard *g, float *host_logits, uint64_t *host_top1)
{
    int st;
    if (!g || g->stream_B == 0 || g->stream_in_flight == 0) return RN_ERR_INVALID;
    g->logits = host_logits;
    g->top1 = host_top1;
    st = post(g, JOB_COLLECT);
    if (st == RN_OK) --g->stream_in_flight;
    else g->stream_B = 0, g->stream_in_flight = 0;
    return st;
}

int rn_shard_in_flight(const rn_shard *g) { return g ? g->stream_in_flight : 0; }
