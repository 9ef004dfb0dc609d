/*
 * rn_model.c -- the network driver, plain C over the C-ABI of rn_hip.h.
 *
 * Replaces the reference driver cuda/inference/main.cu:
 *   createLayer / createResnet152   main.cu:53-89,109-125  -> rn_model_create + set_tensor/load_dir
 *   layerForward                    main.cu:127-166        -> block_forward
 *   resnet152Forward                main.cu:168-226        -> rn_model_forward
 * generalised over the block counts (ResNet-50/101/152) and the batch size.
 *
 * Differences that are deliberate (MI355X-first):
 *   - activations are NHWC inside; the NCHW input image is converted once (to a
 *     4-channel zero-padded NHWC image the stem contraction reads);
 *   - no device synchronisation between ops: everything is queued on the
 *     context's stream (the reference syncs after every launch, nn.cu:14-85);
 *   - activation buffers are a fixed set of ping-pong arenas sized for the largest
 *     batch seen, instead of one cached tensor per block (main.cu:141-159): the
 *     same "second forward allocates nothing" behaviour with ~3x less memory;
 *   - the unused device-to-host copy of the layer4 activation (main.cu:207) is gone;
 *   - RN_FWD_FUSED folds batch-norm, ReLU and the residual add into the
 *     contraction's epilogue; RN_FWD_REFERENCE_OPS launches one kernel per reference
 *     op in the reference's order (conv, bn in place, relu in place, add into act3,
 *     relu), which is the parity baseline for the fused path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rn_hip.h"

#define RN_MAX_KEY 96
#define RN_CLASSES 1000

#define RN_MAX_STREAMS 4

typedef struct {
    char key[RN_MAX_KEY];
    uint64_t numel;
    float *dev;
    int is_set;
} rn_param;

typedef struct {
    char name[RN_MAX_KEY];
    uint64_t cin, cout, k, stride, pad;
    int w, bn_w, bn_b, bn_m, bn_v; /* indices into params */
    void *packed;                  /* K-major panel, model dtype */
    float *scale, *shift;          /* folded batch-norm */
    /* tuned contraction tile (0 = per-launch choice) and the launch batch it was tuned at: slot 0
     * for the parts a batch runs as on its streams, slot 1 for the whole (sub-)batch on one stream
     * (profiled forwards, rn_model_set_streams(m, 1)) */
    int tile[2];
    uint64_t tile_B[2];
} rn_conv;

typedef struct {
    char name[RN_MAX_KEY];
    int conv1, conv2, conv3, ds; /* indices into convs; ds = -1 when absent */
    /* blocks with a downsample branch: conv3 and downsample as one contraction
     * (rn_conv2d_nhwc_pair_forward_dt): rows [Cout][K3 + Kd] with both batch-norm scales
     * folded in, and the sum of the two shifts */
    void *pair_packed;
    float *pair_shift;
    int pair_tile[2];
    uint64_t pair_tile_B[2];
} rn_block;

typedef struct {
    const char *op;
    char layer[RN_MAX_KEY];
    double flops, bytes;
    rn_event *start, *stop;
    float ms;
} rn_prof;

typedef struct {
    int conv;
    int pair_block; /* >= 0: the fused conv3 + downsample call of that block (x2, H2, W2) */
    int exact;      /* the stem in its exact-K form (x = physically padded image) */
    const void *x, *x2;
    void *y;
    uint64_t B, H, W, pad, H2, W2;
    rn_epilogue ep;
    int has_ep;
} rn_conv_call;

struct rn_model {
    rn_ctx *ctx;
    int arch;
    int depths[4];
    rn_param *params;
    uint64_t n_params;
    rn_conv *convs;
    int n_convs;
    rn_block *blocks;
    int n_blocks;
    int fc_w, fc_b;
    int finalized;
    int dtype;        /* storage type of activations and packed weights */
    int pair_fusion;  /* fused mode: conv3 + downsample as one contraction (default on) */
    int stem_exact;   /* fp32: stem in the exact-K form, K = 160 instead of 224 (default on) */
    float *stem_packed_exact;
    int graphs_live;         /* graphs captured from this model that still live (rn_model_capture / rn_graph_destroy) */
    int chain;               /* fused bf16 mode: conv3 of a 64- / 128-channel block + conv1 of the next block as one launch (default on) */
    int t1_ready;            /* the previous block's chained launch has produced this block's conv1 output */
    int stem_pool;           /* fused mode: stem + batch-norm + ReLU + max-pool as one launch (default on) */
    void *stem_pool_packed;  /* its weight panel, model dtype */
    void *fc_packed;  /* fc.weight in the model dtype (bf16 models only) */
    /* activation arenas, sized for batch_cap images */
    uint64_t batch_cap;
    float *x4, *p0, *p1, *dsb, *t1, *t2, *pooled;
    uint64_t act_bytes;
    /* Two halves of a batch on two streams: the launches of one half fill the tails (last
     * round of tiles, prologues, epilogues) of the other's.  Every image's logits are
     * independent of what else is in its launch, so the split changes no bit.  The second
     * stream belongs to a second context on the same device (own scratch). */
    int streams;            /* parts a sub-batch is split into (each >= RN_STREAM_MIN_PART(m) images) */
    rn_ctx *ctxn[RN_MAX_STREAMS - 1];     /* contexts of parts 1.. (part 0 runs on ctx) */
    rn_event *ev_fork, *ev_join[RN_MAX_STREAMS - 1];
    int streams_set;        /* rn_model_set_streams was called: keep it whatever the dtype */
    int single_stream_only; /* tuning pass: its recorded calls are one half */
    /* depth-first front: the stem, the pool and the first stage (the largest tensors) run in
     * front_parts slices of the (sub-)batch, one after the other, so that what one kernel
     * writes is still in the 256 MB Infinity Cache when the next reads it; the rest of the
     * network then runs on the whole batch.  1 = off. */
    int front_parts;
    /* what the ops of the sub-batch being queued run on: context, and views into the arenas */
    rn_ctx *run;
    struct { float *x4, *p0, *p1, *dsb, *t1, *t2, *pooled; } v;
    /* tile tuning: calls of the last forward, and the batch size the tiles were tuned for */
    rn_conv_call *calls;
    int n_calls, cap_calls, recording;
    uint64_t tuned_B;
    int tuned_mode, cur_mode;
    /* profiling */
    int profiling;
    rn_prof *prof;
    uint64_t n_prof, cap_prof;
};

static const uint64_t kWidths[4][3] = {{64, 64, 256}, {256, 128, 512}, {512, 256, 1024},
                                       {1024, 512, 2048}};
static const uint64_t kStrides[4] = {1, 2, 2, 2};

static int add_param(rn_model *m, const char *key, uint64_t numel)
{
    rn_param *p = &m->params[m->n_params];
    snprintf(p->key, RN_MAX_KEY, "%s", key);
    p->numel = numel;
    p->dev = NULL;
    p->is_set = 0;
    return (int)m->n_params++;
}

static int add_conv(rn_model *m, const char *name, const char *bn_name, uint64_t cin, uint64_t cout,
                    uint64_t k, uint64_t stride, uint64_t pad)
{
    rn_conv *c = &m->convs[m->n_convs];
    char key[RN_MAX_KEY + 16];
    memset(c, 0, sizeof(*c));
    snprintf(c->name, RN_MAX_KEY, "%s", name);
    c->cin = cin;
    c->cout = cout;
    c->k = k;
    c->stride = stride;
    c->pad = pad;
    snprintf(key, sizeof(key), "%s.weight", name);
    c->w = add_param(m, key, cout * cin * k * k);
    snprintf(key, sizeof(key), "%s.weight", bn_name);
    c->bn_w = add_param(m, key, cout);
    snprintf(key, sizeof(key), "%s.bias", bn_name);
    c->bn_b = add_param(m, key, cout);
    snprintf(key, sizeof(key), "%s.running_mean", bn_name);
    c->bn_m = add_param(m, key, cout);
    snprintf(key, sizeof(key), "%s.running_var", bn_name);
    c->bn_v = add_param(m, key, cout);
    return m->n_convs++;
}

int rn_model_create(rn_ctx *ctx, rn_model **out, int arch)
{
    static const int d50[4] = {3, 4, 6, 3}, d101[4] = {3, 4, 23, 3}, d152[4] = {3, 8, 36, 3};
    const int *d;
    rn_model *m;
    int li, bi, total_blocks = 0, max_convs;
    if (!ctx || !out) return RN_ERR_INVALID;
    *out = NULL;
    if (arch == 50) d = d50;
    else if (arch == 101) d = d101;
    else if (arch == 152) d = d152;
    else return RN_ERR_UNSUPPORTED;
    m = (rn_model *)calloc(1, sizeof(rn_model));
    if (m) m->pair_fusion = m->stem_exact = 1;
    if (m) m->streams = 2;
    if (m) m->stem_pool = 1;
    if (m) m->chain = 1;
    if (m) m->front_parts = 1;
    if (!m) return RN_ERR_NOMEM;
    m->ctx = ctx;
    m->arch = arch;
    for (li = 0; li < 4; ++li) {
        m->depths[li] = d[li];
        total_blocks += d[li];
    }
    max_convs = 1 + 3 * total_blocks + 4;
    m->params = (rn_param *)calloc((size_t)max_convs * 5 + 2, sizeof(rn_param));
    m->convs = (rn_conv *)calloc((size_t)max_convs, sizeof(rn_conv));
    m->blocks = (rn_block *)calloc((size_t)total_blocks, sizeof(rn_block));
    m->cap_calls = max_convs;
    m->calls = (rn_conv_call *)calloc((size_t)m->cap_calls, sizeof(rn_conv_call));
    if (!m->params || !m->convs || !m->blocks || !m->calls) {
        rn_model_destroy(m);
        return RN_ERR_NOMEM;
    }
    /* stem: conv1 7x7 s2 p3 + bn1 (main.cu:111-112) */
    add_conv(m, "conv1", "bn1", 3, 64, 7, 2, 3);
    for (li = 0; li < 4; ++li) {
        for (bi = 0; bi < d[li]; ++bi) {
            rn_block *b = &m->blocks[m->n_blocks++];
            char pre[RN_MAX_KEY], name[RN_MAX_KEY + 16], bn[RN_MAX_KEY + 16];
            const uint64_t cin = bi == 0 ? kWidths[li][0] : kWidths[li][2];
            const uint64_t mid = kWidths[li][1], cout = kWidths[li][2];
            const uint64_t stride = bi == 0 ? kStrides[li] : 1;
            snprintf(pre, sizeof(pre), "layer%d.%d", li + 1, bi);
            snprintf(b->name, RN_MAX_KEY, "%s", pre);
            b->ds = -1;
            /* projection shortcut iff block 0 and (stride != 1 or cin != cout): main.cu:71 */
            if (bi == 0 && (stride != 1 || cin != cout)) {
                snprintf(name, sizeof(name), "%s.downsample.0", pre);
                snprintf(bn, sizeof(bn), "%s.downsample.1", pre);
                b->ds = add_conv(m, name, bn, cin, cout, 1, stride, 0);
            }
            snprintf(name, sizeof(name), "%s.conv1", pre);
            snprintf(bn, sizeof(bn), "%s.bn1", pre);
            b->conv1 = add_conv(m, name, bn, cin, mid, 1, 1, 0);
            snprintf(name, sizeof(name), "%s.conv2", pre);
            snprintf(bn, sizeof(bn), "%s.bn2", pre);
            b->conv2 = add_conv(m, name, bn, mid, mid, 3, stride, 1); /* stride on the 3x3 */
            snprintf(name, sizeof(name), "%s.conv3", pre);
            snprintf(bn, sizeof(bn), "%s.bn3", pre);
            b->conv3 = add_conv(m, name, bn, mid, cout, 1, 1, 0);
        }
    }
    m->fc_w = add_param(m, "fc.weight", (uint64_t)RN_CLASSES * 2048);
    m->fc_b = add_param(m, "fc.bias", RN_CLASSES);
    *out = m;
    return RN_OK;
}

static void free_acts(rn_model *m)
{
    float **bufs[7];
    int i;
    bufs[0] = &m->x4; bufs[1] = &m->p0; bufs[2] = &m->p1; bufs[3] = &m->dsb;
    bufs[4] = &m->t1; bufs[5] = &m->t2; bufs[6] = &m->pooled;
    for (i = 0; i < 7; ++i) {
        if (*bufs[i]) rn_free(m->ctx, *bufs[i]);
        *bufs[i] = NULL;
    }
    m->batch_cap = 0;
    m->act_bytes = 0;
}

static void free_prof(rn_model *m)
{
    uint64_t i;
    for (i = 0; i < m->cap_prof; ++i) {
        rn_event_destroy(m->prof[i].start);
        rn_event_destroy(m->prof[i].stop);
    }
    free(m->prof);
    m->prof = NULL;
    m->n_prof = m->cap_prof = 0;
}

int rn_model_destroy(rn_model *m)
{
    uint64_t i;
    int c;
    if (!m) return RN_OK;
    /* a captured graph points into this model's arenas, weights and the scratch of the contexts of
     * its extra streams, and rn_graph_destroy unpins those contexts: the graphs go first */
    if (m->graphs_live > 0) return RN_ERR_INVALID;
    if (m->ctx) rn_sync(m->ctx);
    if (m->params) {
        for (i = 0; i < m->n_params; ++i) rn_free(m->ctx, m->params[i].dev);
    }
    if (m->convs) {
        for (c = 0; c < m->n_convs; ++c) {
            rn_free(m->ctx, m->convs[c].packed);
            rn_free(m->ctx, m->convs[c].scale);
            rn_free(m->ctx, m->convs[c].shift);
        }
    }
    if (m->blocks) {
        for (c = 0; c < m->n_blocks; ++c) {
            rn_free(m->ctx, m->blocks[c].pair_packed);
            rn_free(m->ctx, m->blocks[c].pair_shift);
        }
    }
    rn_free(m->ctx, m->fc_packed);
    rn_free(m->ctx, m->stem_packed_exact);
    rn_free(m->ctx, m->stem_pool_packed);
    free_acts(m);
    free_prof(m);
    {
        int k;
        rn_event_destroy(m->ev_fork);
        for (k = 0; k < RN_MAX_STREAMS - 1; ++k) {
            if (!m->ctxn[k]) continue;
            rn_sync(m->ctxn[k]);
            rn_event_destroy(m->ev_join[k]);
            rn_ctx_destroy(m->ctxn[k]);
        }
    }
    free(m->params);
    free(m->convs);
    free(m->blocks);
    free(m->calls);
    free(m);
    return RN_OK;
}

const char *rn_model_tensor_key(const rn_model *m, uint64_t index, uint64_t *numel)
{
    if (!m || index >= m->n_params) return NULL;
    if (numel) *numel = m->params[index].numel;
    return m->params[index].key;
}

int rn_model_set_tensor(rn_model *m, const char *key, const float *host_data, uint64_t numel)
{
    uint64_t i;
    if (!m || !key || !host_data) return RN_ERR_INVALID;
    for (i = 0; i < m->n_params; ++i) {
        rn_param *p = &m->params[i];
        if (strcmp(p->key, key) != 0) continue;
        if (p->numel != numel) return RN_ERR_INVALID;
        if (!p->dev) {
            int st = rn_malloc(m->ctx, (void **)&p->dev, numel * sizeof(float));
            if (st != RN_OK) return st;
        }
        p->is_set = 1;
        m->finalized = 0;
        return rn_memcpy_h2d(m->ctx, p->dev, host_data, numel * sizeof(float));
    }
    return RN_ERR_INVALID; /* unknown key (e.g. *.num_batches_tracked): callers skip those */
}

int rn_model_load_dir(rn_model *m, const char *weights_dir)
{
    uint64_t i;
    if (!m || !weights_dir) return RN_ERR_INVALID;
    for (i = 0; i < m->n_params; ++i) {
        rn_param *p = &m->params[i];
        char path[1024];
        float *dev = NULL;
        uint64_t n = 0;
        int st;
        snprintf(path, sizeof(path), "%s/%s", weights_dir, p->key);
        st = rn_load_f32_file(m->ctx, path, &dev, &n);
        if (st != RN_OK) return st;
        if (n != p->numel) {
            rn_free(m->ctx, dev);
            return RN_ERR_INVALID;
        }
        if (p->dev) rn_free(m->ctx, p->dev);
        p->dev = dev;
        p->is_set = 1;
    }
    m->finalized = 0;
    return RN_OK;
}

static uint64_t elem_size(const rn_model *m) { return m->dtype == RN_DTYPE_BF16 ? 2 : 4; }

int rn_model_set_dtype(rn_model *m, int dtype)
{
    int c;
    if (!m || (dtype != RN_DTYPE_F32 && dtype != RN_DTYPE_BF16)) return RN_ERR_INVALID;
    if (!m->streams_set) m->streams = 2; /* measured default for both element types (fp32 +0.8 %, bf16 +5 %) */
    if (dtype == m->dtype) return RN_OK;
    /* packed panels and arenas depend on the element size: drop them */
    for (c = 0; c < m->n_convs; ++c) {
        rn_free(m->ctx, m->convs[c].packed);
        m->convs[c].packed = NULL;
        m->convs[c].tile[0] = m->convs[c].tile[1] = 0;
    }
    for (c = 0; c < m->n_blocks; ++c) {
        rn_free(m->ctx, m->blocks[c].pair_packed);
        m->blocks[c].pair_packed = NULL;
        m->blocks[c].pair_tile[0] = m->blocks[c].pair_tile[1] = 0;
    }
    rn_free(m->ctx, m->fc_packed);
    m->fc_packed = NULL;
    rn_free(m->ctx, m->stem_pool_packed);
    m->stem_pool_packed = NULL;
    free_acts(m);
    m->dtype = dtype;
    m->finalized = 0;
    m->tuned_B = 0;
    return RN_OK;
}

int rn_model_finalize(rn_model *m)
{
    uint64_t i;
    int c, st;
    if (!m) return RN_ERR_INVALID;
    for (i = 0; i < m->n_params; ++i) {
        if (!m->params[i].is_set) return RN_ERR_INVALID;
    }
    for (c = 0; c < m->n_convs; ++c) {
        rn_conv *cv = &m->convs[c];
        const uint64_t pn = rn_conv2d_packed_weight_numel_dt(m->dtype, cv->cin, cv->cout, cv->k);
        if (!cv->packed) {
            st = rn_malloc(m->ctx, &cv->packed, pn * elem_size(m));
            if (st != RN_OK) return st;
        }
        if (!cv->scale) {
            st = rn_malloc(m->ctx, (void **)&cv->scale, cv->cout * sizeof(float));
            if (st != RN_OK) return st;
            st = rn_malloc(m->ctx, (void **)&cv->shift, cv->cout * sizeof(float));
            if (st != RN_OK) return st;
        }
        st = rn_conv2d_pack_weight_dt(m->ctx, m->dtype, m->params[cv->w].dev, cv->packed, cv->cin,
                                      cv->cout, cv->k);
        if (st != RN_OK) return st;
        st = rn_batchnorm2d_fold(m->ctx, m->params[cv->bn_w].dev, m->params[cv->bn_b].dev,
                                 m->params[cv->bn_m].dev, m->params[cv->bn_v].dev, cv->scale,
                                 cv->shift, cv->cout);
        if (st != RN_OK) return st;
    }
    if (m->dtype == RN_DTYPE_F32) {
        const rn_conv *stem = &m->convs[0];
        if (!m->stem_packed_exact) {
            st = rn_malloc(m->ctx, (void **)&m->stem_packed_exact,
                           rn_conv2d_packed_weight_numel_exact(stem->cin, stem->cout, stem->k) *
                               sizeof(float));
            if (st != RN_OK) return st;
        }
        st = rn_conv2d_pack_weight_exact(m->ctx, m->params[stem->w].dev, m->stem_packed_exact,
                                         stem->cin, stem->cout, stem->k);
        if (st != RN_OK) return st;
    }
    {   /* panel of the fused stem + max-pool launch */
        const rn_conv *stem = &m->convs[0];
        if (!m->stem_pool_packed) {
            st = rn_malloc(m->ctx, &m->stem_pool_packed, rn_stem_pool_packed_weight_numel(m->dtype) * elem_size(m));
            if (st != RN_OK) return st;
        }
        st = rn_stem_pool_pack_weight_dt(m->ctx, m->dtype, m->params[stem->w].dev, m->stem_pool_packed, stem->cin);
        if (st != RN_OK) return st;
    }
    for (c = 0; c < m->n_blocks; ++c) {
        rn_block *b = &m->blocks[c];
        const rn_conv *c3, *cd;
        if (b->ds < 0) continue;
        c3 = &m->convs[b->conv3];
        cd = &m->convs[b->ds];
        if (!b->pair_packed) {
            st = rn_malloc(m->ctx, &b->pair_packed,
                           rn_conv2d_packed_pair_weight_numel(c3->cin, c3->cout, c3->k, cd->cin) *
                               elem_size(m));
            if (st != RN_OK) return st;
        }
        if (!b->pair_shift) {
            st = rn_malloc(m->ctx, (void **)&b->pair_shift, c3->cout * sizeof(float));
            if (st != RN_OK) return st;
        }
        st = rn_conv2d_pack_weight_pair_dt(m->ctx, m->dtype, m->params[c3->w].dev, c3->scale,
                                           m->params[cd->w].dev, cd->scale, b->pair_packed, c3->cin,
                                           c3->cout, c3->k, cd->cin);
        if (st != RN_OK) return st;
        st = rn_add_forward(m->ctx, c3->shift, cd->shift, b->pair_shift, c3->cout);
        if (st != RN_OK) return st;
    }
    if (m->dtype != RN_DTYPE_F32) {
        /* fc.weight [1000][2048] is a 1x1 convolution panel: same packer, k = 1 */
        if (!m->fc_packed) {
            st = rn_malloc(m->ctx, &m->fc_packed, (uint64_t)RN_CLASSES * 2048 * elem_size(m));
            if (st != RN_OK) return st;
        }
        st = rn_conv2d_pack_weight_dt(m->ctx, m->dtype, m->params[m->fc_w].dev, m->fc_packed, 2048,
                                      RN_CLASSES, 1);
        if (st != RN_OK) return st;
    }
    st = rn_sync(m->ctx);
    if (st != RN_OK) return st;
    m->finalized = 1;
    return RN_OK;
}

/* per-image element counts of the arenas (see the header comment) */
#define X4_PER_IMG ((uint64_t)230 * 230 * 4) /* bf16 models keep a 3-pixel zero border */
#define P_PER_IMG ((uint64_t)56 * 56 * 256) /* == 112*112*64, the stem output */
#define T_PER_IMG ((uint64_t)56 * 56 * 128) /* layer2.0 conv1 output, the largest mid tensor */

int rn_ctx_graphs_live(const rn_ctx *ctx); /* rn_ctx.hip */

static int ensure_acts(rn_model *m, uint64_t B)
{
    int st;
    if (B <= m->batch_cap) return RN_OK;
    if (m->batch_cap > 0 && rn_ctx_graphs_live(m->ctx) > 0)
        return RN_ERR_INVALID; /* captured graphs point into the arenas: destroy them first */
    free_acts(m);
    {
        const uint64_t es = elem_size(m);
        st = rn_malloc(m->ctx, (void **)&m->x4, B * X4_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->p0, B * P_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->p1, B * P_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->dsb, B * P_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->t1, B * T_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->t2, B * T_PER_IMG * es);
        if (st == RN_OK) st = rn_malloc(m->ctx, (void **)&m->pooled, B * 2048 * es);
    }
    if (st != RN_OK) {
        free_acts(m);
        return st;
    }
    m->batch_cap = B;
    m->act_bytes = B * (X4_PER_IMG + 3 * P_PER_IMG + 2 * T_PER_IMG + 2048) * elem_size(m);
    return RN_OK;
}

uint64_t rn_model_activation_bytes(const rn_model *m) { return m ? m->act_bytes : 0; }

/* library-internal: the pipeline (rn_pipeline.hip) queues on the model's stream */
rn_ctx *rn_model_context(rn_model *m) { return m ? m->ctx : NULL; }

/* library-internal: every context the model has queued batch parts on (m->ctx and the m->ctxn[]
 * of the other streams).  A captured forward holds pointers into the scratch of each of them:
 * rn_model_capture pins them all for the graph's lifetime (rn_scratch refuses to grow a pinned
 * context's slots). */
int rn_model_contexts(rn_model *m, rn_ctx **out, int cap)
{
    int k, n = 0;
    if (!m || !out) return 0;
    if (n < cap) out[n++] = m->ctx;
    for (k = 0; k < RN_MAX_STREAMS - 1; ++k)
        if (m->ctxn[k] && n < cap) out[n++] = m->ctxn[k];
    return n;
}
/* library-internal: rn_model_capture / rn_graph_destroy count the graphs that hold this model */
void rn_model_graph_ref(rn_model *m, int delta)
{
    if (m) m->graphs_live += delta;
}

int rn_model_set_pair_fusion(rn_model *m, int on)
{
    if (!m) return RN_ERR_INVALID;
    m->pair_fusion = on ? 1 : 0;
    m->tuned_B = 0; /* the set of launches changes */
    return RN_OK;
}

int rn_model_set_streams(rn_model *m, int streams)
{
    if (!m || (streams != 0 && streams != 1 && streams != 2 && streams != 4)) return RN_ERR_INVALID;
    if (streams == 0) { /* back to the library default, as if this function had never been called */
        m->streams = 2;
        m->streams_set = 0;
        return RN_OK;
    }
    m->streams = streams;
    m->streams_set = 1;
    /* the tuned tiles are looked up by launch batch size: those of other part sizes simply stop matching */
    return RN_OK;
}

int rn_model_get_streams(const rn_model *m) { return m ? m->streams : 0; }

static int parts_of(const rn_model *m, uint64_t B);
int rn_model_parts(const rn_model *m, uint64_t B) { return m && B ? parts_of(m, B) : 0; }

int rn_model_set_stem_pool_fusion(rn_model *m, int on)
{
    if (!m) return RN_ERR_INVALID;
    m->stem_pool = on < 0 ? 0 : on > 2 ? 2 : on; /* 2: fetch the patches from the NCHW input */
    m->tuned_B = 0;
    return RN_OK;
}

int rn_model_set_chain(rn_model *m, int on)
{
    if (!m) return RN_ERR_INVALID;
    m->chain = on ? 1 : 0;
    return RN_OK;
}

int rn_model_set_front_parts(rn_model *m, int parts)
{
    if (!m || parts < 1 || parts > 16 || (parts & (parts - 1))) return RN_ERR_INVALID;
    m->front_parts = parts;
    m->tuned_B = 0;
    return RN_OK;
}

int rn_model_set_stem_exact(rn_model *m, int on)
{
    if (!m) return RN_ERR_INVALID;
    m->stem_exact = on ? 1 : 0;
    m->tuned_B = 0;
    return RN_OK;
}

int rn_model_profiling_enabled(const rn_model *m) { return m ? m->profiling : 0; }

/* ---- profiling --------------------------------------------------------- */
static int prof_begin(rn_model *m, const char *op, const char *layer, double flops, double bytes)
{
    rn_prof *r;
    if (!m->profiling) return RN_OK;
    if (m->n_prof == m->cap_prof) {
        const uint64_t ncap = m->cap_prof ? m->cap_prof * 2 : 256;
        rn_prof *np = (rn_prof *)realloc(m->prof, ncap * sizeof(rn_prof));
        uint64_t i;
        if (!np) return RN_ERR_NOMEM;
        m->prof = np;
        for (i = m->cap_prof; i < ncap; ++i) {
            int st;
            memset(&np[i], 0, sizeof(rn_prof));
            st = rn_event_create(m->run, &np[i].start);
            if (st == RN_OK) st = rn_event_create(m->run, &np[i].stop);
            if (st != RN_OK) {
                m->cap_prof = i;
                return st;
            }
        }
        m->cap_prof = ncap;
    }
    r = &m->prof[m->n_prof];
    r->op = op;
    snprintf(r->layer, RN_MAX_KEY, "%s", layer);
    r->flops = flops;
    r->bytes = bytes;
    r->ms = -1.f;
    return rn_event_record(m->run, r->start);
}

static int prof_end(rn_model *m)
{
    if (!m->profiling) return RN_OK;
    return rn_event_record(m->run, m->prof[m->n_prof++].stop);
}

int rn_model_set_profiling(rn_model *m, int on)
{
    if (!m) return RN_ERR_INVALID;
    m->profiling = on ? 1 : 0;
    return RN_OK;
}

uint64_t rn_model_profile_count(const rn_model *m) { return m ? m->n_prof : 0; }

int rn_model_profile_get(const rn_model *m, uint64_t index, const char **op_name,
                         const char **layer_name, float *ms, double *flops, double *bytes)
{
    rn_prof *r;
    if (!m || index >= m->n_prof) return RN_ERR_INVALID;
    r = &m->prof[index];
    if (r->ms < 0.f) {
        int st = rn_event_elapsed_ms(r->start, r->stop, &r->ms);
        if (st != RN_OK) return st;
    }
    if (op_name) *op_name = r->op;
    if (layer_name) *layer_name = r->layer;
    if (ms) *ms = r->ms;
    if (flops) *flops = r->flops;
    if (bytes) *bytes = r->bytes;
    return RN_OK;
}

#define TRY(expr)                   \
    do {                            \
        int st_ = (expr);           \
        if (st_ != RN_OK) return st_; \
    } while (0)

/* ---- ops with profiling brackets --------------------------------------- */
/* pad_override >= 0 replaces the layer's padding (the bf16 stem reads an image that carries
 * its own zero border: H, W are then the padded sizes and the padding is 0); RN_PAD_EXACT: the
 * fp32 stem in its exact-K form, x = [B,H,W,cin] physically padded */
/* next record of the tuning pass's call list (the slices of a depth-first front repeat calls) */
static rn_conv_call *next_call(rn_model *m)
{
    if (m->n_calls == m->cap_calls) {
        const int ncap = 2 * m->cap_calls + 16;
        rn_conv_call *nc = (rn_conv_call *)realloc(m->calls, (size_t)ncap * sizeof(rn_conv_call));
        if (!nc) return NULL;
        m->calls = nc;
        m->cap_calls = ncap;
    }
    return &m->calls[m->n_calls++];
}

/* the tile tuned for a launch of B images in the current mode, or 0 = per-launch choice */
static int tuned_tile(const rn_model *m, const int tile[2], const uint64_t tile_B[2], uint64_t B)
{
    if (!m->tuned_B || m->tuned_mode != m->cur_mode) return 0;
    return tile_B[0] == B ? tile[0] : tile_B[1] == B ? tile[1] : 0;
}

#define RN_PAD_EXACT (-2)
static int op_conv(rn_model *m, const rn_conv *cv, const void *x, void *y, uint64_t B, uint64_t H,
                   uint64_t W, const rn_epilogue *ep, int64_t pad_override)
{
    const int exact = pad_override == RN_PAD_EXACT;
    const uint64_t pad = exact ? 0 : pad_override >= 0 ? (uint64_t)pad_override : cv->pad;
    const uint64_t ho = rn_conv_output_size(H, cv->k, cv->stride, pad);
    const uint64_t wo = rn_conv_output_size(W, cv->k, cv->stride, pad);
    const double M = (double)(B * ho * wo), K = (double)(cv->cin * cv->k * cv->k);
    const double es = (double)elem_size(m);
    double bytes = es * ((double)(B * H * W * cv->cin) + K * (double)cv->cout +
                         M * (double)cv->cout);
    if (ep && ep->residual) bytes += es * M * (double)cv->cout;
    if (m->recording) {
        rn_conv_call *c = next_call(m);
        if (!c) return RN_ERR_NOMEM;
        c->conv = (int)(cv - m->convs);
        c->pair_block = -1;
        c->exact = exact;
        c->x = x;
        c->y = y;
        c->B = B;
        c->H = H;
        c->W = W;
        c->pad = pad;
        c->has_ep = ep != NULL;
        if (ep) c->ep = *ep;
    }
    TRY(prof_begin(m, ep ? "conv2d+epilogue" : "conv2d", cv->name, 2.0 * M * (double)cv->cout * K,
                   bytes));
    rn_ctx_set_conv_tile(m->run, tuned_tile(m, cv->tile, cv->tile_B, B));
    {
        const int st =
            exact ? rn_conv2d_nhwc_exact_forward(m->run, (const float *)x, (float *)y,
                                                 m->stem_packed_exact, cv->k, cv->stride, ho, wo, B,
                                                 cv->cin, cv->cout, H, W, ep)
                  : rn_conv2d_nhwc_forward_dt(m->run, m->dtype, m->dtype, x, y, cv->packed, cv->k,
                                              cv->stride, pad, ho, wo, B, cv->cin, cv->cout, H, W, ep);
        rn_ctx_set_conv_tile(m->run, 0);
        if (st != RN_OK) return st;
    }
    return prof_end(m);
}

/* conv3 (input t, [B,H,W,c3->cin]) + downsample (input x, [B,H2,W2,cd->cin]) + shifts + ReLU */
static int op_pair(rn_model *m, rn_block *b, const void *t, const void *x, void *y, uint64_t B,
                   uint64_t H, uint64_t W, uint64_t H2, uint64_t W2)
{
    const rn_conv *c3 = &m->convs[b->conv3], *cd = &m->convs[b->ds];
    const double M = (double)(B * H * W), K = (double)(c3->cin + cd->cin);
    const double es = (double)elem_size(m);
    const double bytes = es * ((double)(B * H * W * c3->cin) + (double)(B * H2 * W2 * cd->cin) +
                               K * (double)c3->cout + M * (double)c3->cout);
    char name[RN_MAX_KEY];
    rn_conv_second second;
    rn_epilogue ep;
    int st;
    second.inp = x; second.in_channels = cd->cin; second.H = H2; second.W = W2;
    second.stride = cd->stride;
    ep.scale = NULL; ep.shift = b->pair_shift; ep.residual = NULL; ep.relu = 1;
    if (m->recording) {
        rn_conv_call *c = next_call(m);
        if (!c) return RN_ERR_NOMEM;
        c->conv = b->conv3;
        c->pair_block = (int)(b - m->blocks);
        c->exact = 0;
        c->x = t; c->x2 = x; c->y = y;
        c->B = B; c->H = H; c->W = W; c->pad = 0; c->H2 = H2; c->W2 = W2;
        c->has_ep = 1;
        c->ep = ep;
    }
    snprintf(name, sizeof(name), "%.*s+downsample", (int)(RN_MAX_KEY - 12), c3->name);
    TRY(prof_begin(m, "conv2d+epilogue", name, 2.0 * M * (double)c3->cout * K, bytes));
    rn_ctx_set_conv_tile(m->run, tuned_tile(m, b->pair_tile, b->pair_tile_B, B));
    st = rn_conv2d_nhwc_pair_forward_dt(m->run, m->dtype, m->dtype, t, y, b->pair_packed, c3->k,
                                        c3->stride, c3->pad, H, W, B, c3->cin, c3->cout, H, W,
                                        &second, &ep);
    rn_ctx_set_conv_tile(m->run, 0);
    if (st != RN_OK) return st;
    return prof_end(m);
}

/* conv3 + bn3 + residual + ReLU of block b and conv1 + bn1 + ReLU of the block after it as one
 * launch (rn_conv_chain_forward_dt): t2 -> y (written: the next block's residual) -> the next
 * block's t1, y reaching conv1 through LDS.  Algorithmic work: both contractions' FLOPs; bytes
 * t2 + residual + y + t1 + both weight panels (y is not read back). */
static int chain_applies(const rn_model *m, const rn_block *b, int mode)
{
    const int bi = (int)(b - m->blocks);
    const rn_conv *c3 = &m->convs[b->conv3], *n1;
    if (!m->chain || mode != RN_FWD_FUSED || m->recording) return 0;
    if (b->ds >= 0) { /* first block of a stage: only as the fused pair at equal resolution (stage 1) */
        const rn_conv *cd = &m->convs[b->ds];
        if (!m->pair_fusion || cd->stride != 1 || cd->cin != 64) return 0;
    }
    if (bi + 1 >= m->n_blocks) return 0;
    if (m->front_parts > 1 && bi + 1 == m->depths[0]) return 0; /* the next block runs in another slice */
    n1 = &m->convs[m->blocks[bi + 1].conv1];
    if (c3->k != 1 || c3->stride != 1 || n1->k != 1 || n1->stride != 1 || n1->cin != c3->cout) return 0;
    if (c3->cin == 64 && c3->cout == 256) return n1->cout == 64 || n1->cout == 128;
    if (m->dtype != RN_DTYPE_BF16 || b->ds >= 0) return 0;
    /* bf16: the 128-channel blocks of stage 2 (panels in registers).  A chain for the 256-channel
     * blocks of stage 3 (panels streamed through LDS) was built in round 3 and measured 112 us against
     * 91-95 for its two launches (profiles/round3/chain_against_two_launches_bf16.txt): removed */
    return c3->cin == 128 && c3->cout == 512 && n1->cout == 128;
}

static int op_chain(rn_model *m, const rn_block *b, const void *t2, const void *shortcut, void *y,
                    uint64_t B, uint64_t H, uint64_t W)
{
    const rn_conv *c3 = &m->convs[b->conv3];
    const rn_conv *n1 = &m->convs[m->blocks[(b - m->blocks) + 1].conv1];
    const double M = (double)(B * H * W), es = (double)elem_size(m);
    char name[RN_MAX_KEY];
    snprintf(name, sizeof(name), "%.*s%s+next.conv1", (int)(RN_MAX_KEY - 24), c3->name, b->ds >= 0 ? "+downsample" : "");
    {
        const double k1 = (double)c3->cin + (b->ds >= 0 ? (double)m->convs[b->ds].cin : 0.0);
        /* second operand of the first product: the residual (c3->cout channels) or the block input */
        const double op2 = b->ds >= 0 ? (double)m->convs[b->ds].cin : (double)c3->cout;
        TRY(prof_begin(m, "conv2d+epilogue+conv2d", name,
                       2.0 * M * ((double)c3->cout * k1 + (double)n1->cout * (double)n1->cin),
                       es * (M * ((double)c3->cin + op2 + (double)c3->cout + (double)n1->cout) +
                             (double)c3->cout * k1 + (double)(n1->cout * n1->cin))));
    }
    if (b->ds >= 0) /* shortcut = the block's input: the downsample branch rides in the first product */
        TRY(rn_conv_chain_pair_forward_dt(m->run, m->dtype, t2, shortcut, y, b->pair_packed, b->pair_shift,
                                          m->v.t1, n1->packed, n1->scale, n1->shift, B * H * W, c3->cin,
                                          m->convs[b->ds].cin, c3->cout, n1->cout));
    else
        TRY(rn_conv_chain_forward_dt(m->run, m->dtype, t2, shortcut, y, c3->packed, c3->scale, c3->shift,
                                     m->v.t1, n1->packed, n1->scale, n1->shift, B * H * W, c3->cin,
                                     c3->cout, n1->cout));
    m->t1_ready = 1;
    return prof_end(m);
}

static int op_bn(rn_model *m, const rn_conv *cv, float *y, uint64_t B, uint64_t HW)
{
    const double n = (double)(B * cv->cout * HW);
    TRY(prof_begin(m, "batchnorm2d", cv->name, 0.0, 8.0 * n + 16.0 * (double)cv->cout));
    TRY(rn_batchnorm2d_forward(m->run, y, y, m->params[cv->bn_w].dev, m->params[cv->bn_b].dev,
                               m->params[cv->bn_m].dev, m->params[cv->bn_v].dev, B, cv->cout, HW));
    return prof_end(m);
}

static int op_relu(rn_model *m, const char *layer, float *y, uint64_t n)
{
    TRY(prof_begin(m, "relu", layer, 0.0, 8.0 * (double)n));
    TRY(rn_relu_forward(m->run, y, y, n));
    return prof_end(m);
}

static int op_add(rn_model *m, const char *layer, float *y, const float *shortcut, uint64_t n)
{
    TRY(prof_begin(m, "add", layer, 0.0, 12.0 * (double)n));
    TRY(rn_add_forward(m->run, y, shortcut, y, n)); /* out aliases inp1: main.cu:162 */
    return prof_end(m);
}

/* conv1 + bn1 + ReLU + max-pool (main.cu:179-192) as one launch: x4 (physically padded image)
 * -> p0 (pooled, where the separate max-pool writes too).  Algorithmic work: the stem's FLOPs
 * (no halo), the image read once, the pooled tensor written once. */
static int op_stem_pool(rn_model *m, const rn_conv *stem, const float *input_nchw, uint64_t B,
                        uint64_t Hp, uint64_t Wp, uint64_t ho, uint64_t wo)
{
    const double es = (double)elem_size(m);
    const uint64_t ph = rn_conv_output_size(ho, 3, 2, 1), pw = rn_conv_output_size(wo, 3, 2, 1);
    const double cs = m->dtype == RN_DTYPE_BF16 ? 4.0 : 3.0;
    TRY(prof_begin(m, "conv2d+epilogue+maxpool", "conv1+maxpool",
                   2.0 * (double)(B * ho * wo) * (double)stem->cout * (double)(stem->cin * stem->k * stem->k),
                   (input_nchw ? 4.0 * (double)(B * (Hp - 6) * (Wp - 6) * stem->cin)
                               : es * (double)(B * Hp * Wp) * cs) +
                       es * ((double)(stem->cout * stem->cin * stem->k * stem->k) +
                             (double)(B * ph * pw * stem->cout))));
    if (input_nchw) /* stem_pool == 2: the patch fetch reads the caller's NCHW fp32 image itself */
        TRY(rn_stem_pool_nchw_forward_dt(m->run, m->dtype, input_nchw, m->v.p0, m->stem_pool_packed,
                                         stem->scale, stem->shift, 1, B, stem->cin, Hp - 6, Wp - 6));
    else
        TRY(rn_stem_pool_forward_dt(m->run, m->dtype, m->v.x4, m->v.p0, m->stem_pool_packed, stem->scale,
                                    stem->shift, 1, B, Hp, Wp));
    return prof_end(m);
}

/* one bottleneck block (layerForward body, main.cu:131-164).  x -> y, both NHWC. */
static int block_forward(rn_model *m, rn_block *b, const float *x, float *y, uint64_t B,
                         uint64_t *H, uint64_t *W, int mode)
{
    const rn_conv *c1 = &m->convs[b->conv1], *c2 = &m->convs[b->conv2], *c3 = &m->convs[b->conv3];
    const uint64_t h = *H, w = *W;
    const uint64_t ho = rn_conv_output_size(h, c2->k, c2->stride, c2->pad);
    const uint64_t wo = rn_conv_output_size(w, c2->k, c2->stride, c2->pad);
    const float *shortcut = x;
    if (mode == RN_FWD_FUSED) {
        rn_epilogue ep;
        const int pair = b->ds >= 0 && m->pair_fusion;
        if (b->ds >= 0 && !pair) {
            const rn_conv *cd = &m->convs[b->ds];
            ep.scale = cd->scale; ep.shift = cd->shift; ep.residual = NULL; ep.relu = 0;
            TRY(op_conv(m, cd, x, m->v.dsb, B, h, w, &ep, -1));
            shortcut = m->v.dsb;
        }
        ep.scale = c1->scale; ep.shift = c1->shift; ep.residual = NULL; ep.relu = 1;
        if (m->t1_ready)
            m->t1_ready = 0; /* the block before has left this conv1's output in t1 (op_chain) */
        else
            TRY(op_conv(m, c1, x, m->v.t1, B, h, w, &ep, -1));
        ep.scale = c2->scale; ep.shift = c2->shift;
        TRY(op_conv(m, c2, m->v.t1, m->v.t2, B, h, w, &ep, -1));
        if (chain_applies(m, b, mode)) {
            TRY(op_chain(m, b, m->v.t2, pair ? x : shortcut, y, B, ho, wo));
        } else if (pair) {
            /* the downsample tensor is never materialised: its K rows ride in conv3's loop */
            TRY(op_pair(m, b, m->v.t2, x, y, B, ho, wo, h, w));
        } else {
            ep.scale = c3->scale; ep.shift = c3->shift; ep.residual = shortcut;
            TRY(op_conv(m, c3, m->v.t2, y, B, ho, wo, &ep, -1));
        }
    } else {
        if (b->ds >= 0) {
            const rn_conv *cd = &m->convs[b->ds];
            TRY(op_conv(m, cd, x, m->v.dsb, B, h, w, NULL, -1));
            TRY(op_bn(m, cd, m->v.dsb, B, ho * wo));
            shortcut = m->v.dsb;
        }
        TRY(op_conv(m, c1, x, m->v.t1, B, h, w, NULL, -1));
        TRY(op_bn(m, c1, m->v.t1, B, h * w));
        TRY(op_relu(m, c1->name, m->v.t1, B * h * w * c1->cout));
        TRY(op_conv(m, c2, m->v.t1, m->v.t2, B, h, w, NULL, -1));
        TRY(op_bn(m, c2, m->v.t2, B, ho * wo));
        TRY(op_relu(m, c2->name, m->v.t2, B * ho * wo * c2->cout));
        TRY(op_conv(m, c3, m->v.t2, y, B, ho, wo, NULL, -1));
        TRY(op_bn(m, c3, y, B, ho * wo));
        TRY(op_add(m, b->name, y, shortcut, B * ho * wo * c3->cout));
        TRY(op_relu(m, b->name, y, B * ho * wo * c3->cout));
    }
    *H = ho;
    *W = wo;
    return RN_OK;
}

int rn_ctx_wait_event(rn_ctx *ctx, rn_event *ev); /* rn_ctx.hip: the stream waits, not the host */

/* B images whose activations live at image offset img_off of the arenas, queued on `run`. */
enum { RN_PHASE_ALL = 0, RN_PHASE_FRONT = 1, RN_PHASE_BACK = 2 };

static int forward_sub(rn_model *m, rn_ctx *run, uint64_t img_off, const float *input_nchw,
                       uint64_t B, float *logits, int mode, int phase)
{
    const int nfront = m->depths[0]; /* blocks of the front phase: the first stage */
    int fused_pool = 0;
    const rn_conv *stem;
    uint64_t H = 224, W = 224, ho, wo, ph, pw;
    float *x, *y, *tmp;
    int bi, saved_layout, st;
    {
        const uint64_t es = elem_size(m);
        m->run = run;
        m->v.x4 = (float *)((char *)m->x4 + img_off * X4_PER_IMG * es);
        m->v.p0 = (float *)((char *)m->p0 + img_off * P_PER_IMG * es);
        m->v.p1 = (float *)((char *)m->p1 + img_off * P_PER_IMG * es);
        m->v.dsb = (float *)((char *)m->dsb + img_off * P_PER_IMG * es);
        m->v.t1 = (float *)((char *)m->t1 + img_off * T_PER_IMG * es);
        m->v.t2 = (float *)((char *)m->t2 + img_off * T_PER_IMG * es);
        m->v.pooled = (float *)((char *)m->pooled + img_off * 2048 * es);
    }
    m->cur_mode = mode;
    m->t1_ready = 0;
    saved_layout = rn_ctx_get_layout(m->run);
    rn_ctx_set_layout(m->run, RN_LAYOUT_NHWC);
    st = RN_OK;
    do {
#define STEP(expr) if ((st = (expr)) != RN_OK) break
        const double es = (double)elem_size(m);
        const int bf16 = m->dtype == RN_DTYPE_BF16;
        stem = &m->convs[0];
        if (phase == RN_PHASE_BACK) {
            /* the front ran already (in slices): x holds the first stage's output */
            H = W = 56;
            x = (nfront & 1) ? m->v.p1 : m->v.p0;
            y = (nfront & 1) ? m->v.p0 : m->v.p1;
            for (bi = nfront; bi < m->n_blocks; ++bi) {
                STEP(block_forward(m, &m->blocks[bi], x, y, B, &H, &W, mode));
                tmp = x; x = y; y = tmp;
            }
            if (st != RN_OK) break;
            goto tail_ops;
        }
        if (bf16) {
            /* bf16 stem: [B,230,230,4] image with its own 3-pixel zero border, padding 0 */
            const uint64_t border = stem->pad;
            const int from_nchw = m->stem_pool == 2;
            if (!from_nchw) {
                STEP(prof_begin(m, "nchw_to_nhwc4", "input", 0.0,
                                (double)B * (4.0 * 3 * 224 * 224 + es * 230 * 230 * 4)));
                STEP(rn_nchw_to_nhwc_pad_dt(m->run, m->dtype, input_nchw, m->v.x4, B, 3, H, W, 4, border));
                STEP(prof_end(m));
            }
            ho = rn_conv_output_size(H + 2 * border, stem->k, stem->stride, 0);
            wo = rn_conv_output_size(W + 2 * border, stem->k, stem->stride, 0);
            if (m->stem_pool) {
                STEP(op_stem_pool(m, stem, from_nchw ? input_nchw : NULL, B, H + 2 * border, W + 2 * border, ho, wo));
                fused_pool = 1;
            } else {
                rn_epilogue ep;
                ep.scale = stem->scale; ep.shift = stem->shift; ep.residual = NULL; ep.relu = 1;
                STEP(op_conv(m, stem, m->v.x4, m->v.p1, B, H + 2 * border, W + 2 * border, &ep, 0));
            }
        } else {
            /* exact-K form: [B,230,230,3] with a physical border; else [B,224,224,4] */
            const uint64_t border = m->stem_exact ? stem->pad : 0;
            const uint64_t sh = H + 2 * border, sw = W + 2 * border;
            const int64_t form = m->stem_exact ? RN_PAD_EXACT : -1;
            const int from_nchw = mode == RN_FWD_FUSED && m->stem_pool == 2 && m->stem_exact;
            if (from_nchw) {
                /* no layout launch */
            } else if (m->stem_exact) {
                STEP(prof_begin(m, "nchw_to_nhwc3", "input", 0.0,
                                4.0 * (double)B * (3.0 * 224 * 224 + 3.0 * 230 * 230)));
                STEP(rn_nchw_to_nhwc_pad_dt(m->run, RN_DTYPE_F32, input_nchw, m->v.x4, B, 3, H, W, 3,
                                            border));
            } else {
                STEP(prof_begin(m, "nchw_to_nhwc4", "input", 0.0, 4.0 * (double)(B * 224 * 224 * 7)));
                STEP(rn_nchw_to_nhwc_pad(m->run, input_nchw, m->v.x4, B, 3, H, W, 4));
            }
            if (!from_nchw) STEP(prof_end(m));
            ho = rn_conv_output_size(H, stem->k, stem->stride, stem->pad);
            wo = rn_conv_output_size(W, stem->k, stem->stride, stem->pad);
            if (mode == RN_FWD_FUSED && m->stem_pool && m->stem_exact) {
                STEP(op_stem_pool(m, stem, from_nchw ? input_nchw : NULL, B, sh, sw, ho, wo));
                fused_pool = 1;
            } else if (mode == RN_FWD_FUSED) {
                rn_epilogue ep;
                ep.scale = stem->scale; ep.shift = stem->shift; ep.residual = NULL; ep.relu = 1;
                STEP(op_conv(m, stem, m->v.x4, m->v.p1, B, sh, sw, &ep, form));
            } else {
                STEP(op_conv(m, stem, m->v.x4, m->v.p1, B, sh, sw, NULL, form));
                STEP(op_bn(m, stem, m->v.p1, B, ho * wo));
                STEP(op_relu(m, "conv1", m->v.p1, B * ho * wo * 64));
            }
        }
        /* maxpool 3x3 s2 p1 (main.cu:114,192) */
        ph = rn_conv_output_size(ho, 3, 2, 1);
        pw = rn_conv_output_size(wo, 3, 2, 1);
        if (!fused_pool) {
            STEP(prof_begin(m, "maxpool2d", "maxpool", 0.0,
                            es * (double)(B * 64 * (ho * wo + ph * pw))));
            STEP(rn_maxpool2d_nhwc_forward_dt(m->run, m->dtype, m->v.p1, m->v.p0, 3, 2, 1, ph, pw, B, 64, ho,
                                              wo));
            STEP(prof_end(m));
        }
        H = ph;
        W = pw;
        x = m->v.p0;
        y = m->v.p1;
        for (bi = 0; bi < (phase == RN_PHASE_FRONT ? nfront : m->n_blocks); ++bi) {
            STEP(block_forward(m, &m->blocks[bi], x, y, B, &H, &W, mode));
            tmp = x; x = y; y = tmp;
        }
        if (st != RN_OK || phase == RN_PHASE_FRONT) break;
    tail_ops:
        /* global 7x7 average (main.cu:120,213) then fc (main.cu:122,224) */
        STEP(prof_begin(m, "avgpool2d", "avgpool", 0.0, es * (double)(B * 2048 * (H * W + 1))));
        STEP(rn_avgpool2d_nhwc_forward_dt(m->run, m->dtype, x, m->v.pooled, 7, 1, 0,
                                          rn_conv_output_size(H, 7, 1, 0),
                                          rn_conv_output_size(W, 7, 1, 0), B, 2048, H, W));
        STEP(prof_end(m));
        STEP(prof_begin(m, "linear", "fc", 2.0 * (double)B * 2048.0 * RN_CLASSES,
                        es * ((double)B * 2048.0 + 2048.0 * RN_CLASSES) +
                            4.0 * (RN_CLASSES + (double)B * RN_CLASSES)));
        if (bf16) {
            rn_epilogue ep;
            ep.scale = NULL; ep.shift = m->params[m->fc_b].dev; ep.residual = NULL; ep.relu = 0;
            STEP(rn_conv2d_nhwc_forward_dt(m->run, m->dtype, RN_DTYPE_F32, m->v.pooled, logits,
                                           m->fc_packed, 1, 1, 0, 1, 1, B, 2048, RN_CLASSES, 1, 1,
                                           &ep));
        } else {
            STEP(rn_linear_forward(m->run, m->v.pooled, logits, m->params[m->fc_w].dev,
                                   m->params[m->fc_b].dev, B, 2048, RN_CLASSES));
        }
        STEP(prof_end(m));
#undef STEP
    } while (0);
    rn_ctx_set_layout(m->run, saved_layout);
    return st;
}

/* smallest batch part worth a stream of its own.  bf16 launches are short (fill and drain are a
 * third of their life): parts of 64 still gain 6-8 %.  fp32 launches are matrix-bound rounds of
 * tiles: parts of 128 gain 1 %, parts of 96 LOSE 3 % (B = 192), parts of 64 nothing
 * (tools/streams_ab.sh) -- so the library's own default of two streams starts at parts of 128 there;
 * a count the caller set (rn_model_set_streams) is taken down to parts of 64 */
#define RN_STREAM_MIN_PART(m) ((m)->dtype == RN_DTYPE_F32 && !(m)->streams_set ? 128u : 64u)

#define RN_MAX_SUB_BATCH 512 /* images per launch batch: see rn_model_forward */

/* parts (streams) a launch batch of B images runs as */
static int parts_of(const rn_model *m, uint64_t B)
{
    int parts = m->streams;
    if (B > RN_MAX_SUB_BATCH) B = RN_MAX_SUB_BATCH;
    while (parts > 1 && B / (uint64_t)parts < RN_STREAM_MIN_PART(m)) parts /= 2;
    return parts;
}
#define RN_FRONT_MIN_SLICE 16

/* B images at image offset img_off on `run`: whole, or depth-first through the front */
static int forward_part(rn_model *m, rn_ctx *run, uint64_t img_off, const float *input_nchw,
                        uint64_t B, float *logits, int mode)
{
    int fp = m->front_parts, j;
    uint64_t lo = 0;
    while (fp > 1 && B / (uint64_t)fp < RN_FRONT_MIN_SLICE) fp /= 2;
    if (fp < 2) return forward_sub(m, run, img_off, input_nchw, B, logits, mode, RN_PHASE_ALL);
    for (j = 0; j < fp; ++j) {
        const uint64_t hi = B * (uint64_t)(j + 1) / (uint64_t)fp;
        TRY(forward_sub(m, run, img_off + lo, input_nchw + lo * 3 * 224 * 224, hi - lo, logits, mode,
                        RN_PHASE_FRONT));
        lo = hi;
    }
    return forward_sub(m, run, img_off, input_nchw, B, logits, mode, RN_PHASE_BACK);
}

/* One sub-batch: every tensor of it stays below the kernels' 2^29-element range.  Large enough,
 * it runs as `streams` contiguous parts on as many streams (see rn_model.streams). */
static int forward_chunk(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode)
{
    uint64_t lo = 0;
    int parts = parts_of(m, B), i;
    TRY(ensure_acts(m, B));
    m->n_prof = 0;
    if (parts < 2 || m->profiling || m->single_stream_only || m->recording)
        return forward_part(m, m->ctx, 0, input_nchw, B, logits, mode);
    if (!m->ev_fork) TRY(rn_event_create(m->ctx, &m->ev_fork));
    for (i = 0; i < parts - 1; ++i) {
        if (m->ctxn[i]) continue;
        TRY(rn_ctx_create(&m->ctxn[i], rn_ctx_device(m->ctx), NULL));
        TRY(rn_event_create(m->ctxn[i], &m->ev_join[i]));
    }
    /* fork: whatever the caller queued before this forward (the input upload) is done before
     * the other streams start; join: the first stream carries on after every part */
    TRY(rn_event_record(m->ctx, m->ev_fork));
    for (i = 0; i < parts; ++i) {
        const uint64_t hi = B * (uint64_t)(i + 1) / (uint64_t)parts;
        rn_ctx *run = i == 0 ? m->ctx : m->ctxn[i - 1];
        if (i > 0) TRY(rn_ctx_wait_event(run, m->ev_fork));
        TRY(forward_part(m, run, lo, input_nchw + lo * 3 * 224 * 224, hi - lo, logits + lo * RN_CLASSES,
                         mode));
        if (i > 0) TRY(rn_event_record(run, m->ev_join[i - 1]));
        lo = hi;
    }
    for (i = 0; i < parts - 1; ++i) TRY(rn_ctx_wait_event(m->ctx, m->ev_join[i]));
    return RN_OK;
}

/* The reference has no batch limit other than memory (main.cu:168-226).  Here the contraction
 * kernels address every tensor with 32-bit byte offsets (2^29 fp32 elements; the stem output
 * of 669 images is the first to pass it), so a larger batch runs as sub-batches of at most
 * RN_MAX_SUB_BATCH images through the same arenas.  Every image's logits are independent of
 * what else is in its launch (batch invariance, bit for bit), so the split changes nothing.
 * (RN_MAX_SUB_BATCH is defined above, next to the stream split.) */

int rn_model_forward(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode)
{
    uint64_t done = 0;
    if (!m || !input_nchw || !logits || B == 0) return RN_ERR_INVALID;
    if (mode != RN_FWD_REFERENCE_OPS && mode != RN_FWD_FUSED) return RN_ERR_INVALID;
    if (!m->finalized) return RN_ERR_INVALID;
    /* bf16 storage exists only with the fused epilogues (no standalone bf16 bn/relu/add) */
    if (m->dtype != RN_DTYPE_F32 && mode != RN_FWD_FUSED) return RN_ERR_UNSUPPORTED;
    while (done < B) {
        const uint64_t nb = B - done < RN_MAX_SUB_BATCH ? B - done : RN_MAX_SUB_BATCH;
        TRY(forward_chunk(m, input_nchw + done * 3 * 224 * 224, nb, logits + done * RN_CLASSES, mode));
        done += nb;
    }
    return RN_OK;
}

#define RN_TUNE_ROUNDS 6

/* one recorded contraction call again, on context `run` */
static int replay_call(rn_model *m, rn_ctx *run, const rn_conv_call *k)
{
    rn_conv *cv = &m->convs[k->conv];
    const uint64_t ho = rn_conv_output_size(k->H, cv->k, cv->stride, k->pad);
    const uint64_t wo = rn_conv_output_size(k->W, cv->k, cv->stride, k->pad);
    if (k->pair_block >= 0) {
        const rn_block *pb = &m->blocks[k->pair_block];
        const rn_conv *cd = &m->convs[pb->ds];
        rn_conv_second second;
        second.inp = k->x2; second.in_channels = cd->cin; second.H = k->H2;
        second.W = k->W2; second.stride = cd->stride;
        return rn_conv2d_nhwc_pair_forward_dt(run, m->dtype, m->dtype, k->x, k->y, pb->pair_packed, cv->k,
                                              cv->stride, k->pad, ho, wo, k->B, cv->cin, cv->cout, k->H,
                                              k->W, &second, &k->ep);
    }
    if (k->exact)
        return rn_conv2d_nhwc_exact_forward(run, (const float *)k->x, (float *)k->y, m->stem_packed_exact,
                                            cv->k, cv->stride, ho, wo, k->B, cv->cin, cv->cout, k->H, k->W,
                                            k->has_ep ? &k->ep : NULL);
    return rn_conv2d_nhwc_forward_dt(run, m->dtype, m->dtype, k->x, k->y, cv->packed, cv->k, cv->stride,
                                     k->pad, ho, wo, k->B, cv->cin, cv->cout, k->H, k->W,
                                     k->has_ep ? &k->ep : NULL);
}

/* time every tile candidate of every contraction of a forward of B images (one launch batch);
 * the winners go to slot `slot` of the layers' tile tables */
static int tune_at(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode, int slot)
{
    rn_event *e0 = NULL, *e1 = NULL;
    const int ncand = rn_conv_tile_candidates();
    float cand_ms[64];
    int i, c, r, st;
    if (ncand + 1 > 64) return RN_ERR_INVALID;
    m->single_stream_only = 1;
    /* one recorded forward with the per-launch choice: fills the buffers with real data */
    m->n_calls = 0;
    m->recording = 1;
    st = rn_model_forward(m, input_nchw, B, logits, mode);
    m->recording = 0;
    m->single_stream_only = 0;
    if (st != RN_OK) return st;
    st = rn_event_create(m->ctx, &e0);
    if (st == RN_OK) st = rn_event_create(m->ctx, &e1);
    for (i = 0; st == RN_OK && i < m->n_calls; ++i) {
        const rn_conv_call *k = &m->calls[i];
        rn_conv *cv = &m->convs[k->conv];
        float best = 1e30f;
        int best_c = 0, seen = 0, j;
        for (j = 0; j < i; ++j) /* the slices of a depth-first front repeat their calls */
            seen |= m->calls[j].conv == k->conv && m->calls[j].pair_block == k->pair_block &&
                    m->calls[j].B == k->B;
        if (seen) continue;
        /* repetitions outside, candidates inside: a clock or temperature drift during the
         * measurement meets every candidate alike (candidate by candidate, the later ones were
         * timed on a warmer chip); the first round warms the caches and is not counted */
        for (c = 0; c <= ncand; ++c) cand_ms[c] = 1e30f;
        for (r = 0; r < RN_TUNE_ROUNDS && st == RN_OK; ++r) {
            for (c = 0; c <= ncand && st == RN_OK; ++c) { /* 0 = the per-launch choice itself */
                float t = 0.f;
                rn_ctx_set_conv_tile(m->ctx, c);
                st = rn_event_record(m->ctx, e0);
                if (st == RN_OK) st = replay_call(m, m->ctx, k);
                if (st == RN_OK) st = rn_event_record(m->ctx, e1);
                if (st == RN_OK) st = rn_event_elapsed_ms(e0, e1, &t);
                if (r > 0 && t < cand_ms[c]) cand_ms[c] = t;
            }
        }
        for (c = 0; c <= ncand; ++c) {
            if (cand_ms[c] < best * 0.995f) { /* ties keep the earlier (default) candidate */
                best = cand_ms[c];
                best_c = c;
            }
        }
        if (k->pair_block >= 0) {
            m->blocks[k->pair_block].pair_tile[slot] = best_c;
            m->blocks[k->pair_block].pair_tile_B[slot] = k->B;
        } else {
            cv->tile[slot] = best_c;
            cv->tile_B[slot] = k->B;
        }
    }
    rn_ctx_set_conv_tile(m->ctx, 0);
    rn_event_destroy(e0);
    rn_event_destroy(e1);
    return st;
}

int rn_model_tune(rn_model *m, const float *input_nchw, uint64_t B, float *logits, int mode)
{
    const uint64_t B_all = B;
    uint64_t Bp;
    int c, st;
    if (!m) return RN_ERR_INVALID;
    if (B > RN_MAX_SUB_BATCH) B = RN_MAX_SUB_BATCH; /* the launches of a larger batch are sub-batches */
    Bp = B;
    {   /* ... and those run as `streams` parts */
        const int parts = parts_of(m, B);
        if (parts > 1) Bp = B / (uint64_t)parts;
    }
    m->tuned_B = 0;
    for (c = 0; c < m->n_convs; ++c) m->convs[c].tile_B[0] = m->convs[c].tile_B[1] = 0;
    for (c = 0; c < m->n_blocks; ++c) m->blocks[c].pair_tile_B[0] = m->blocks[c].pair_tile_B[1] = 0;
    st = tune_at(m, input_nchw, Bp, logits, mode, 0);
    /* the whole (sub-)batch on one stream is what a profiled forward launches: its own tiles */
    if (st == RN_OK && Bp != B) st = tune_at(m, input_nchw, B, logits, mode, 1);
    if (st != RN_OK) return st;
    m->tuned_B = B;
    m->tuned_mode = mode;
    /* leave the buffers and the logits as a normal forward would */
    return rn_model_forward(m, input_nchw, B_all, logits, mode);
}

/* ---- tuned tiles as data ----------------------------------------------------------------
 * The table rn_model_tune fills, as words: a header that names what it was measured for, then
 * per convolution and per fused pair the two (tile, launch batch) slots.  A model of the same
 * architecture, element type and fusion settings on an identical device takes it over instead of
 * timing every candidate again: the shards of a node (rn_shard_tune tunes ONE shard), or a later
 * process.  Tiles only change speed, never bits, so a stale table costs time, not parity. */
#define RN_TUNING_MAGIC 0x726e54554e453034ull /* "rnTUNE04" */
#define RN_TUNING_HEADER 10

static uint64_t tuning_settings(const rn_model *m)
{
    return (uint64_t)m->pair_fusion | (uint64_t)m->stem_exact << 1 | (uint64_t)m->stem_pool << 2 |
           (uint64_t)m->chain << 4 | (uint64_t)m->front_parts << 8;
}

int rn_model_export_tuning(const rn_model *m, uint64_t *words, uint64_t cap, uint64_t *n_words)
{
    uint64_t need, at = 0;
    int c, k;
    if (!m || !n_words) return RN_ERR_INVALID;
    need = RN_TUNING_HEADER + 4 * ((uint64_t)m->n_convs + (uint64_t)m->n_blocks);
    *n_words = need;
    if (!words) return RN_OK; /* size query */
    if (cap < need || !m->tuned_B) return RN_ERR_INVALID;
    words[at++] = RN_TUNING_MAGIC;
    words[at++] = (uint64_t)m->arch;
    words[at++] = (uint64_t)m->dtype;
    words[at++] = tuning_settings(m);
    words[at++] = m->tuned_B;
    words[at++] = (uint64_t)m->tuned_mode;
    words[at++] = (uint64_t)m->n_convs;
    words[at++] = (uint64_t)m->n_blocks;
    words[at++] = (uint64_t)rn_conv_tile_candidates();
    words[at++] = 0;
    for (c = 0; c < m->n_convs; ++c)
        for (k = 0; k < 2; ++k) {
            words[at++] = (uint64_t)m->convs[c].tile[k];
            words[at++] = m->convs[c].tile_B[k];
        }
    for (c = 0; c < m->n_blocks; ++c)
        for (k = 0; k < 2; ++k) {
            words[at++] = (uint64_t)m->blocks[c].pair_tile[k];
            words[at++] = m->blocks[c].pair_tile_B[k];
        }
    return RN_OK;
}

int rn_model_import_tuning(rn_model *m, const uint64_t *words, uint64_t n_words)
{
    uint64_t at = RN_TUNING_HEADER;
    int c, k;
    if (!m || !words || n_words < RN_TUNING_HEADER) return RN_ERR_INVALID;
    if (words[0] != RN_TUNING_MAGIC || words[1] != (uint64_t)m->arch || words[2] != (uint64_t)m->dtype ||
        words[3] != tuning_settings(m) || words[6] != (uint64_t)m->n_convs || words[7] != (uint64_t)m->n_blocks ||
        words[8] != (uint64_t)rn_conv_tile_candidates() ||
        n_words != RN_TUNING_HEADER + 4 * ((uint64_t)m->n_convs + (uint64_t)m->n_blocks))
        return RN_ERR_INVALID; /* measured for another model, setting or build */
    for (c = 0; c < m->n_convs + m->n_blocks; ++c) /* a candidate this build does not have */
        if (words[at + 4 * (uint64_t)c] > (uint64_t)rn_conv_tile_candidates() ||
            words[at + 4 * (uint64_t)c + 2] > (uint64_t)rn_conv_tile_candidates())
            return RN_ERR_INVALID;
    for (c = 0; c < m->n_convs; ++c)
        for (k = 0; k < 2; ++k) {
            m->convs[c].tile[k] = (int)words[at++];
            m->convs[c].tile_B[k] = words[at++];
        }
    for (c = 0; c < m->n_blocks; ++c)
        for (k = 0; k < 2; ++k) {
            m->blocks[c].pair_tile[k] = (int)words[at++];
            m->blocks[c].pair_tile_B[k] = words[at++];
        }
    m->tuned_B = words[4];
    m->tuned_mode = (int)words[5];
    return RN_OK;
}
