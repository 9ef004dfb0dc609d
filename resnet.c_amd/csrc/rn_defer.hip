// Deferred execution of the reference's op-by-op call sequence (rn_ctx_set_deferred).
//
// The reference's callers (cuda/inference/main.cu:127-166, cuda/nn.cu) run a bottleneck block as
//     conv.forward(x, t); bn.forward(t, t); reluForward(t, t);                      (conv1, conv2)
//     conv.forward(t2, y); bn.forward(y, y); addForward(y, shortcut, y); reluForward(y, y);   (conv3)
// seven separate launches on NCHW tensors, each a full pass over its tensor (12 of the literal
// route's 31 ms per 256 images are those batch-norm / ReLU / add passes).  With the context in
// deferred mode the seven reference entry points -- rn_conv2d_forward, rn_batchnorm2d_forward,
// rn_relu_forward, rn_add_forward, rn_maxpool2d_forward, rn_avgpool2d_forward, rn_linear_forward --
// only RECORD their call.  The list runs when something is observed (a copy to the host, rn_sync,
// rn_observe, a free, any other entry point), and then
//   * conv -> [batch-norm in place] -> [add in place] -> [ReLU in place] on one buffer becomes ONE
//     launch of the NHWC contraction with the fused epilogue (folded batch-norm scale / shift,
//     residual, ReLU: rn_conv2d_nhwc_forward; the 3-channel stem in its exact-K form).  Only in-place
//     chains are folded: after them nothing but the final content of the buffer can be observed,
//     and that is what the fused launch writes.  Nothing is skipped: every buffer the caller named
//     as an output holds its value when the list has run;
//   * that launch writes NHWC INTO THE CALLER'S BUFFER, and the context remembers it (a tag: base
//     pointer + B, C, H, W).  The next convolution, pool, batch-norm, ReLU or add that reads a
//     tagged buffer runs on it as NHWC, so a whole network stays NHWC between its first and last
//     op with no transposes;
//   * a tagged buffer is turned back into NCHW only when it is observed: rn_memcpy_d2h of a whole
//     tensor transposes through scratch on its way out (the buffer stays NHWC), anything else
//     (rn_observe, a device-to-device copy, a partial read, an op that has no NHWC form here)
//     rewrites the buffer in place as NCHW first.
// Everything else is executed literally, in call order.  Results: the fused epilogue applies the
// folded batch-norm as one fp32 fmaf (rn_batchnorm2d_fold) instead of the reference's double
// expression -- at most 2e-5 on ResNet logits, inside the 1e-4 bar, like RN_FWD_FUSED of the model
// driver; the literal route (deferred off) stays the parity baseline.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rn_internal.h"

namespace {

enum Kind { K_CONV, K_BN, K_RELU, K_ADD, K_MAXPOOL, K_AVGPOOL, K_LINEAR };

struct Op {
    Kind kind;
    const float *in, *in2;
    float *out;
    const float *w, *b, *mean, *var;
    uint64_t k, s, p, ho, wo, B, Cin, Cout, H, W, N;
};

struct Tag {  // a caller buffer that currently holds NHWC
    const void *ptr;
    uint64_t B, C, H, W;
    uint64_t numel() const { return B * C * H * W; }
};

struct Fold {  // batch-norm parameters folded to scale / shift (rn_batchnorm2d_fold), per parameter set
    const float *w, *b, *m, *v;
    uint64_t C;
    float *scale, *shift;
};

struct ExactPack {  // exact-K panel of a small-Cin (stem) weight, or (stem) the fused stem launch's register panel
    const float *w;
    uint64_t cin, cout, k;
    float *packed;
    bool stem;
};

inline bool overlaps(const void *p, uint64_t n, const void *q, uint64_t m)
{
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)q;
    return a < b + (m ? m : 1) && b < a + (n ? n : 1);
}

}  // namespace

struct rn_defer_state {
    std::vector<Op> ops;
    std::vector<Tag> tags;
    std::vector<Fold> folds;
    std::vector<ExactPack> exact;
    uint64_t fused_launches = 0, literal_launches = 0, transposes = 0;
    // conv3 of a 64-channel block + conv1 of the next as one launch (rn_chain.hip); RN_DEFER_CHAINS=0: A/B runs
    bool chains = !(getenv("RN_DEFER_CHAINS") && atoi(getenv("RN_DEFER_CHAINS")) == 0);
    // the stem convolution + bn + ReLU and the max-pool behind it as one launch that writes both tensors
    // (rn_stem_conv_pool_nchw_forward); RN_DEFER_STEM=0: A/B runs
    bool stem = !(getenv("RN_DEFER_STEM") && atoi(getenv("RN_DEFER_STEM")) == 0);
};

namespace {

rn_defer_state *state(rn_ctx *ctx)
{
    if (!ctx->ds) ctx->ds = new rn_defer_state();
    return ctx->ds;
}

Tag *find_tag(rn_defer_state *ds, const void *ptr)
{
    for (Tag &t : ds->tags)
        if (t.ptr == ptr) return &t;
    return nullptr;
}

void untag(rn_defer_state *ds, const void *ptr)
{
    for (size_t i = 0; i < ds->tags.size(); ++i)
        if (ds->tags[i].ptr == ptr) {
            ds->tags[i] = ds->tags.back();
            ds->tags.pop_back();
            return;
        }
}

// layouts coincide when a tensor has one channel or one pixel: never worth a tag
void set_tag(rn_defer_state *ds, const void *ptr, uint64_t B, uint64_t C, uint64_t H, uint64_t W)
{
    untag(ds, ptr);
    if (C == 1 || H * W == 1) return;
    ds->tags.push_back(Tag{ptr, B, C, H, W});
}

// the buffer at t.ptr back to NCHW, in place (through scratch slot 3)
int materialise(rn_ctx *ctx, const void *ptr)
{
    rn_defer_state *ds = ctx->ds;
    Tag *t = ds ? find_tag(ds, ptr) : nullptr;
    if (!t) return RN_OK;
    const Tag tag = *t;
    void *tmp = nullptr;
    RN_TRY(rn_scratch(ctx, 3, tag.numel() * sizeof(float), &tmp));
    RN_TRY(rn_nhwc_to_nchw(ctx, (const float *)tag.ptr, (float *)tmp, tag.B, tag.C, tag.H, tag.W));
    RN_HIP_TRY(ctx, hipMemcpyAsync(const_cast<void *>(tag.ptr), tmp, tag.numel() * sizeof(float),
                                   hipMemcpyDeviceToDevice, ctx->stream));
    ++ds->transposes;
    untag(ds, ptr);
    return RN_OK;
}

// An operand that lies INSIDE a tagged buffer without being that buffer -- a batch slice, a sub-view at an
// offset, an output that covers only part of it -- is NCHW arithmetic on the caller's side: the buffer gets its
// NCHW content back first.  (An operand that IS a tagged buffer, base and size, is the ops' own business.)
int settle(rn_ctx *ctx, const void *ptr, uint64_t bytes)
{
    rn_defer_state *ds = ctx->ds;
    if (!ptr || !bytes) return RN_OK;
    for (size_t i = 0; i < ds->tags.size();) {
        const Tag t = ds->tags[i];
        const uint64_t tb = t.numel() * sizeof(float);
        if (overlaps(ptr, bytes, t.ptr, tb) && !(t.ptr == ptr && tb == bytes)) {
            RN_TRY(materialise(ctx, t.ptr));
            i = 0;
        } else {
            ++i;
        }
    }
    return RN_OK;
}

int settle_op(rn_ctx *ctx, const Op &o)
{
    const uint64_t f = sizeof(float);
    switch (o.kind) {
    case K_CONV:
        RN_TRY(settle(ctx, o.in, o.B * o.Cin * o.H * o.W * f));
        return settle(ctx, o.out, o.B * o.Cout * o.ho * o.wo * f);
    case K_BN:
        RN_TRY(settle(ctx, o.in, o.B * o.Cout * o.N * f));
        return settle(ctx, o.out, o.B * o.Cout * o.N * f);
    case K_RELU:
        RN_TRY(settle(ctx, o.in, o.N * f));
        return settle(ctx, o.out, o.N * f);
    case K_ADD:
        RN_TRY(settle(ctx, o.in, o.N * f));
        RN_TRY(settle(ctx, o.in2, o.N * f));
        return settle(ctx, o.out, o.N * f);
    case K_MAXPOOL:
    case K_AVGPOOL:
        RN_TRY(settle(ctx, o.in, o.B * o.Cout * o.H * o.W * f));
        return settle(ctx, o.out, o.B * o.Cout * o.ho * o.wo * f);
    case K_LINEAR:
        RN_TRY(settle(ctx, o.in, o.B * o.Cin * f));
        return settle(ctx, o.out, o.B * o.Cout * f);
    }
    return RN_OK;
}

int fold_for(rn_ctx *ctx, const Op &bn, const float **scale, const float **shift)
{
    rn_defer_state *ds = ctx->ds;
    for (const Fold &f : ds->folds)
        if (f.w == bn.w && f.b == bn.b && f.m == bn.mean && f.v == bn.var && f.C == bn.Cout) {
            *scale = f.scale, *shift = f.shift;
            return RN_OK;
        }
    Fold f{bn.w, bn.b, bn.mean, bn.var, bn.Cout, nullptr, nullptr};
    RN_HIP_TRY(ctx, hipMalloc((void **)&f.scale, 2 * bn.Cout * sizeof(float)));
    f.shift = f.scale + bn.Cout;
    const int st = rn_batchnorm2d_fold(ctx, bn.w, bn.b, bn.mean, bn.var, f.scale, f.shift, bn.Cout);
    if (st != RN_OK) {
        (void)hipFree(f.scale);
        return st;
    }
    ds->folds.push_back(f);
    *scale = f.scale, *shift = f.shift;
    return RN_OK;
}

int packed_for(rn_ctx *ctx, const Op &c, const float **packed)
{
    void *wp = rn_wcache_find(ctx, c.w, c.Cin, c.Cout, c.k);
    if (!wp) {
        const uint64_t wn = rn_conv2d_packed_weight_numel(c.Cin, c.Cout, c.k);
        RN_TRY(rn_wcache_add(ctx, c.w, c.Cin, c.Cout, c.k, wn * sizeof(float), &wp));
        const int st = rn_conv2d_pack_weight(ctx, c.w, (float *)wp, c.Cin, c.Cout, c.k);
        if (st != RN_OK) {
            rn_wcache_remove(ctx, wp);
            return st;
        }
    }
    *packed = (const float *)wp;
    return RN_OK;
}

int exact_for(rn_ctx *ctx, const Op &c, const float **packed, bool stem = false)
{
    rn_defer_state *ds = ctx->ds;
    for (const ExactPack &e : ds->exact)
        if (e.w == c.w && e.cin == c.Cin && e.cout == c.Cout && e.k == c.k && e.stem == stem) {
            *packed = e.packed;
            return RN_OK;
        }
    ExactPack e{c.w, c.Cin, c.Cout, c.k, nullptr, stem};
    const uint64_t numel = stem ? rn_stem_pool_packed_weight_numel(RN_DTYPE_F32)
                                : rn_conv2d_packed_weight_numel_exact(c.Cin, c.Cout, c.k);
    RN_HIP_TRY(ctx, hipMalloc((void **)&e.packed, numel * sizeof(float)));
    const int st = stem ? rn_stem_pool_pack_weight_dt(ctx, RN_DTYPE_F32, c.w, e.packed, c.Cin)
                        : rn_conv2d_pack_weight_exact(ctx, c.w, e.packed, c.Cin, c.Cout, c.k);
    if (st != RN_OK) {
        (void)hipFree(e.packed);
        return st;
    }
    ds->exact.push_back(e);
    *packed = e.packed;
    return RN_OK;
}

bool is_tagged_as(rn_defer_state *ds, const void *ptr, uint64_t B, uint64_t C, uint64_t H, uint64_t W)
{
    const Tag *t = find_tag(ds, ptr);
    return t && t->B == B && t->C == C && t->H == H && t->W == W;
}

// the contraction forms the fused route can take: whole 128-byte channel segments, or the exact-K
// small-Cin form; tensors within the kernels' 32-bit byte offsets
bool fusable_conv(const Op &c)
{
    const uint64_t lim = 1ull << 29;
    if (c.B * c.H * c.W * c.Cin >= lim || c.B * c.ho * c.wo * c.Cout >= lim || c.k > 15 || c.k < 1) return false;
    if (((uintptr_t)c.in | (uintptr_t)c.out | (uintptr_t)c.w) & 15) return false;
    if (c.Cin % 32 == 0) return rn_conv2d_packed_weight_numel(c.Cin, c.Cout, c.k) < lim && c.Cout % 4 == 0;
    return c.Cin <= 4 && c.k <= 8 && c.Cout % 4 == 0 && c.s < 4096 &&
           c.B * (c.H + 2 * c.p) * (c.W + 2 * c.p) * c.Cin < lim;
}

// every buffer a recorded op names: {pointer, bytes, written}
struct Range {
    const void *ptr;
    uint64_t bytes;
    bool write;
};
int op_ranges(const Op &o, Range r[7])
{
    const uint64_t f = sizeof(float);
    int n = 0;
    switch (o.kind) {
    case K_CONV:
        r[n++] = {o.in, o.B * o.Cin * o.H * o.W * f, false};
        r[n++] = {o.w, o.Cout * o.Cin * o.k * o.k * f, false};
        r[n++] = {o.out, o.B * o.Cout * o.ho * o.wo * f, true};
        break;
    case K_BN:
        r[n++] = {o.in, o.B * o.Cout * o.N * f, false};
        r[n++] = {o.w, o.Cout * f, false};
        r[n++] = {o.b, o.Cout * f, false};
        r[n++] = {o.mean, o.Cout * f, false};
        r[n++] = {o.var, o.Cout * f, false};
        r[n++] = {o.out, o.B * o.Cout * o.N * f, true};
        break;
    case K_RELU:
        r[n++] = {o.in, o.N * f, false};
        r[n++] = {o.out, o.N * f, true};
        break;
    case K_ADD:
        r[n++] = {o.in, o.N * f, false};
        r[n++] = {o.in2, o.N * f, false};
        r[n++] = {o.out, o.N * f, true};
        break;
    case K_MAXPOOL:
    case K_AVGPOOL:
        r[n++] = {o.in, o.B * o.Cout * o.H * o.W * f, false};
        r[n++] = {o.out, o.B * o.Cout * o.ho * o.wo * f, true};
        break;
    case K_LINEAR:
        r[n++] = {o.in, o.B * o.Cin * f, false};
        r[n++] = {o.w, o.Cout * o.Cin * f, false};
        if (o.b) r[n++] = {o.b, o.Cout * f, false};
        r[n++] = {o.out, o.B * o.Cout * f, true};
        break;
    }
    return n;
}

// may ops [a0, a1) and ops [b0, b1) of the list run in either order?  (no buffer one group writes is named by the other)
bool independent(const rn_defer_state *ds, size_t a0, size_t a1, size_t b0, size_t b1)
{
    Range ra[7], rb[7];
    for (size_t a = a0; a < a1; ++a) {
        const int na = op_ranges(ds->ops[a], ra);
        for (size_t b = b0; b < b1; ++b) {
            const int nb = op_ranges(ds->ops[b], rb);
            for (int x = 0; x < na; ++x)
                for (int y = 0; y < nb; ++y)
                    if ((ra[x].write || rb[y].write) && overlaps(ra[x].ptr, ra[x].bytes, rb[y].ptr, rb[y].bytes)) return false;
        }
    }
    return true;
}

int run_conv(rn_ctx *ctx, size_t i, size_t *next);
int run_literal(rn_ctx *ctx, const Op &o);

// c = conv3 (1x1, 64 -> 256) with bn + add + ReLU folded, ops[j] = the op behind that group.  When conv1 of the next
// block follows on c's output (1x1, 256 -> 64 | 128, bn and ReLU in place behind it) and every tensor is already NHWC,
// both groups run as ONE launch (rn_conv_chain_forward_dt: the block output is written on its way through LDS, so
// the caller's tensor holds its value as ever; the same bits as the two launches).  One other convolution of c's
// output with its in-place batch-norm / ReLU may stand between the two (the projection shortcut of a stage's first
// block, main.cu:131-137, which the reference runs before conv1): it runs right after the launch instead of right
// before conv1, which no buffer can tell when neither group names a buffer the other writes.  *done = taken.
int try_chain(rn_ctx *ctx, const Op &c, const Op &bn, const float *r, size_t j, size_t *next, bool *done)
{
    rn_defer_state *ds = ctx->ds;
    const size_t n = ds->ops.size();
    *done = false;
    if (!ds->chains || j >= n || ds->ops[j].kind != K_CONV) return RN_OK;
    if (!(c.k == 1 && c.s == 1 && c.p == 0 && c.Cin == 64 && c.Cout == 256 && c.ho == c.H && c.wo == c.W)) return RN_OK;
    const uint64_t rows = c.B * c.H * c.W, f = sizeof(float);
    if (rows * 256 * f >= (1ull << 31)) return RN_OK;
    auto is_conv1 = [&](const Op &o) {
        return o.kind == K_CONV && o.in == c.out && o.k == 1 && o.s == 1 && o.p == 0 && o.Cin == 256 &&
               (o.Cout == 64 || o.Cout == 128) && o.B == c.B && o.H == c.H && o.W == c.W && o.ho == c.H && o.wo == c.W &&
               fusable_conv(o);
    };
    // the in-place batch-norm / ReLU tail of the convolution at ops[at]: index of the first op behind it
    auto tail_end = [&](size_t at, const Op **bn_, const Op **relu_) {
        const Op &o = ds->ops[at];
        size_t e = at + 1;
        if (e < n && ds->ops[e].kind == K_BN && ds->ops[e].in == o.out && ds->ops[e].out == o.out &&
            ds->ops[e].Cout == o.Cout && ds->ops[e].B == o.B && ds->ops[e].N == o.ho * o.wo)
            *bn_ = &ds->ops[e++];
        if (e < n && ds->ops[e].kind == K_RELU && ds->ops[e].in == o.out && ds->ops[e].out == o.out &&
            ds->ops[e].N == o.B * o.Cout * o.ho * o.wo)
            *relu_ = &ds->ops[e++];
        return e;
    };
    size_t jx = j;  // [j, jx): the group that stands between, if any
    if (!is_conv1(ds->ops[j])) {
        if (ds->ops[j].in != c.out || ds->ops[j].out == c.out) return RN_OK;
        const Op *b_ = nullptr, *r_ = nullptr;
        jx = tail_end(j, &b_, &r_);
        if (jx >= n || !is_conv1(ds->ops[jx])) return RN_OK;
    }
    const Op c2 = ds->ops[jx];
    const Op *bn2 = nullptr, *relu2 = nullptr;
    const size_t j2 = tail_end(jx, &bn2, &relu2);
    if (!relu2) return RN_OK;
    if (jx > j && !independent(ds, j, jx, jx, j2)) return RN_OK;
    // the kernel reads its inputs a step ahead of the rows it writes: no output on top of an input or of the other
    const void *bufs[4] = {c.in, r, c.out, c2.out};
    const uint64_t sizes[4] = {rows * 64 * f, rows * 256 * f, rows * 256 * f, rows * c2.Cout * f};
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b)
            if ((a >= 2 || b >= 2) && overlaps(bufs[a], sizes[a], bufs[b], sizes[b])) return RN_OK;
    RN_TRY(settle_op(ctx, c2));
    if (bn2) RN_TRY(settle_op(ctx, *bn2));
    RN_TRY(settle_op(ctx, *relu2));
    if (!is_tagged_as(ds, c.in, c.B, 64, c.H, c.W) || !is_tagged_as(ds, r, c.B, 256, c.H, c.W)) return RN_OK;
    const float *sc3 = nullptr, *sh3 = nullptr, *sc1 = nullptr, *sh1 = nullptr, *w3 = nullptr, *w1 = nullptr;
    RN_TRY(fold_for(ctx, bn, &sc3, &sh3));
    if (bn2) RN_TRY(fold_for(ctx, *bn2, &sc1, &sh1));
    RN_TRY(packed_for(ctx, c, &w3));
    RN_TRY(packed_for(ctx, c2, &w1));
    RN_TRY(rn_conv_chain_forward_dt(ctx, RN_DTYPE_F32, c.in, r, c.out, w3, sc3, sh3, c2.out, w1, sc1, sh1, rows, 64, 256,
                                    c2.Cout));
    set_tag(ds, c.out, c.B, 256, c.H, c.W);
    set_tag(ds, c2.out, c.B, c2.Cout, c.H, c.W);
    ++ds->fused_launches;
    // the group that stood between: now
    for (size_t i = j; i < jx;) {
        RN_TRY(settle_op(ctx, ds->ops[i]));
        if (ds->ops[i].kind == K_CONV) {
            size_t nx = i + 1;
            RN_TRY(run_conv(ctx, i, &nx));
            i = nx < jx ? nx : jx;
        } else {
            const Op o = ds->ops[i];
            RN_TRY(run_literal(ctx, o));
            ++i;
        }
    }
    *next = j2;
    *done = true;
    return RN_OK;
}

// ops[i] is a convolution: fold the in-place chain behind it into its epilogue; *next = first op not consumed
int run_conv(rn_ctx *ctx, size_t i, size_t *next)
{
    rn_defer_state *ds = ctx->ds;
    const Op c = ds->ops[i];
    const uint64_t numel = c.B * c.Cout * c.ho * c.wo;
    size_t j = i + 1;
    const size_t n = ds->ops.size();
    const Op *bn = nullptr, *add = nullptr, *relu = nullptr;
    *next = i + 1;
    if (!fusable_conv(c)) {
        // no NHWC contraction for this shape: the literal call on NCHW tensors
        RN_TRY(materialise(ctx, c.in));
        untag(ds, c.out);
        ++ds->literal_launches;
        return rn_conv2d_forward(ctx, c.in, c.out, c.w, c.k, c.s, c.p, c.ho, c.wo, c.B, c.Cin, c.Cout, c.H, c.W);
    }
    if (j < n && ds->ops[j].kind == K_BN && ds->ops[j].in == c.out && ds->ops[j].out == c.out &&
        ds->ops[j].Cout == c.Cout && ds->ops[j].B == c.B && ds->ops[j].N == c.ho * c.wo)
        bn = &ds->ops[j++];
    if (j < n && ds->ops[j].kind == K_ADD && ds->ops[j].out == c.out && ds->ops[j].N == numel &&
        ((ds->ops[j].in == c.out) != (ds->ops[j].in2 == c.out)))
        add = &ds->ops[j++];
    if (j < n && ds->ops[j].kind == K_RELU && ds->ops[j].in == c.out && ds->ops[j].out == c.out &&
        ds->ops[j].N == numel)
        relu = &ds->ops[j++];
    if (bn) RN_TRY(settle_op(ctx, *bn));      // (in place on c.out: nothing to do unless it overlaps another buffer)
    if (add) RN_TRY(settle_op(ctx, *add));    // the residual may be a slice of a tagged buffer
    if (relu) RN_TRY(settle_op(ctx, *relu));
    rn_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    if (bn) RN_TRY(fold_for(ctx, *bn, &ep.scale, &ep.shift));
    if (add && ((uintptr_t)(add->in == c.out ? add->in2 : add->in) & 15)) {
        add = nullptr, relu = nullptr, j = i + 1 + (bn ? 1 : 0);  // the epilogue reads the residual in 16-byte rows
    }
    // Every rewrite of a buffer to NCHW (an operand tagged under ANOTHER shape: a view) comes first: it goes
    // through scratch slot 3, which the transposed residual below lives in until the launch has read it.
    // (the small-Cin form builds its padded image from the NCHW tensor: its input always goes back to NCHW)
    if (c.Cin % 32 != 0 || !is_tagged_as(ds, c.in, c.B, c.Cin, c.H, c.W)) RN_TRY(materialise(ctx, c.in));
    if (add) {
        const float *r = add->in == c.out ? add->in2 : add->in;
        if (!is_tagged_as(ds, r, c.B, c.Cout, c.ho, c.wo)) RN_TRY(materialise(ctx, r));
    }
    if (bn && add && relu) {
        bool done = false;
        RN_TRY(try_chain(ctx, c, *bn, add->in == c.out ? add->in2 : add->in, j, next, &done));
        if (done) return RN_OK;
    }
    if (add) {
        const float *r = add->in == c.out ? add->in2 : add->in;
        if (is_tagged_as(ds, r, c.B, c.Cout, c.ho, c.wo) || c.Cout == 1 || c.ho * c.wo == 1) {
            ep.residual = r;
        } else {
            void *rt = nullptr;
            RN_TRY(rn_scratch(ctx, 3, numel * sizeof(float), &rt));
            RN_TRY(rn_nchw_to_nhwc(ctx, r, (float *)rt, c.B, c.Cout, c.ho, c.wo));
            ++ds->transposes;
            ep.residual = rt;
        }
    }
    ep.relu = relu ? 1 : 0;
    const rn_epilogue *epp = (bn || add || relu) ? &ep : nullptr;
    int st;
    if (c.Cin % 32 == 0) {
        const float *x = c.in;
        if (!is_tagged_as(ds, c.in, c.B, c.Cin, c.H, c.W) && c.H * c.W != 1) {
            void *xt = nullptr;
            RN_TRY(rn_scratch(ctx, 2, c.B * c.Cin * c.H * c.W * sizeof(float), &xt));
            RN_TRY(rn_nchw_to_nhwc(ctx, c.in, (float *)xt, c.B, c.Cin, c.H, c.W));
            ++ds->transposes;
            x = (const float *)xt;
        }
        const float *wp = nullptr;
        RN_TRY(packed_for(ctx, c, &wp));
        st = rn_conv2d_nhwc_forward(ctx, x, c.out, wp, c.k, c.s, c.p, c.ho, c.wo, c.B, c.Cin, c.Cout, c.H, c.W, epp);
    } else {
        // The reference's stem -- conv 7x7 / 2 / 3 to 64 channels, bn and ReLU in place, then max-pool 3x3 / 2 / 1 of
        // that tensor into another (main.cu:179-192) -- is one launch that writes BOTH tensors, the stem tensor from the
        // registers that feed the pool (rn_stem.hip, WRITE_Y), straight from the NCHW image: no layout pass, no
        // second read of the largest tensor of the network
        if (ds->stem && relu && !add && c.k == 7 && c.s == 2 && c.p == 3 && c.Cout == 64 && c.Cin <= 3 && j < n &&
            ds->ops[j].kind == K_MAXPOOL) {
            const Op pl = ds->ops[j];
            const uint64_t ph = (c.ho + 2 - 3) / 2 + 1, pw = (c.wo + 2 - 3) / 2 + 1, f = sizeof(float);
            const uint64_t patch = 48 + ((13 * (c.W + 6) * 3 * f + 15) & ~15ull);
            const bool shapes = pl.in == c.out && pl.k == 3 && pl.s == 2 && pl.p == 1 && pl.B == c.B && pl.Cout == 64 &&
                                pl.H == c.ho && pl.W == c.wo && pl.ho == ph && pl.wo == pw && c.ho >= 1 && c.wo >= 8 &&
                                c.W % 4 == 0 && c.W <= 256 && c.wo % 8 == 0 && c.wo <= 128 && patch <= 48 * 1024 &&
                                2 * patch + 5 * pw * 64 * f <= 160 * 1024 && c.ho * c.wo * 64 * f < (1ull << 31) &&
                                c.B * ph * pw < (1ull << 31) && c.H + 6 < (1u << 14) &&
                                (((uintptr_t)pl.out | (uintptr_t)c.in) & 15) == 0;
            const uint64_t xin = c.B * c.Cin * c.H * c.W * f, yb = numel * f, pb = c.B * 64 * ph * pw * f;
            if (shapes && !overlaps(pl.out, pb, c.out, yb) && !overlaps(pl.out, pb, c.in, xin) &&
                !overlaps(c.out, yb, c.in, xin)) {
                RN_TRY(settle_op(ctx, pl));
                const float *wp = nullptr;
                RN_TRY(exact_for(ctx, c, &wp, true));
                RN_TRY(rn_stem_conv_pool_nchw_forward(ctx, c.in, c.out, pl.out, wp, ep.scale, ep.shift, c.B, c.Cin, c.H,
                                                      c.W));
                set_tag(ds, c.out, c.B, 64, c.ho, c.wo);
                set_tag(ds, pl.out, c.B, 64, ph, pw);
                ++ds->fused_launches;
                *next = j + 1;
                return RN_OK;
            }
        }
        // small Cin (the stem): NCHW image -> [B, H + 2p, W + 2p, Cin] with its zero border, exact-K panel
        const uint64_t Hp = c.H + 2 * c.p, Wp = c.W + 2 * c.p;
        void *xt = nullptr;
        RN_TRY(rn_scratch(ctx, 2, c.B * Hp * Wp * c.Cin * sizeof(float), &xt));
        RN_TRY(rn_nchw_to_nhwc_pad_dt(ctx, RN_DTYPE_F32, c.in, xt, c.B, c.Cin, c.H, c.W, c.Cin, c.p));
        ++ds->transposes;
        const float *wp = nullptr;
        RN_TRY(exact_for(ctx, c, &wp));
        st = rn_conv2d_nhwc_exact_forward(ctx, (const float *)xt, c.out, wp, c.k, c.s, c.ho, c.wo, c.B, c.Cin, c.Cout,
                                          Hp, Wp, epp);
    }
    RN_TRY(st);
    set_tag(ds, c.out, c.B, c.Cout, c.ho, c.wo);
    ++ds->fused_launches;
    *next = j;
    return RN_OK;
}

// one recorded op as the literal call, in the layout its input holds
int run_literal(rn_ctx *ctx, const Op &o)
{
    rn_defer_state *ds = ctx->ds;
    int st = RN_OK;
    ++ds->literal_launches;
    switch (o.kind) {
    case K_BN: {
        const Tag *t = find_tag(ds, o.in);
        const bool nhwc = t && t->B == o.B && t->C == o.Cout && t->H * t->W == o.N;
        if (t && !nhwc) RN_TRY(materialise(ctx, o.in));
        ctx->layout = nhwc ? RN_LAYOUT_NHWC : RN_LAYOUT_NCHW;
        const Tag keep = nhwc ? *t : Tag{nullptr, 0, 0, 0, 0};
        st = rn_batchnorm2d_forward(ctx, o.in, o.out, o.w, o.b, o.mean, o.var, o.B, o.Cout, o.N);
        ctx->layout = RN_LAYOUT_NCHW;
        if (o.out != o.in) {
            if (nhwc) set_tag(ds, o.out, keep.B, keep.C, keep.H, keep.W);
            else untag(ds, o.out);
        }
        return st;
    }
    case K_RELU: {
        const Tag *t = find_tag(ds, o.in);
        if (t && t->numel() != o.N) {
            RN_TRY(materialise(ctx, o.in));
            t = nullptr;
        }
        const Tag keep = t ? *t : Tag{nullptr, 0, 0, 0, 0};
        st = rn_relu_forward(ctx, o.in, o.out, o.N);
        if (o.out != o.in) {
            if (t) set_tag(ds, o.out, keep.B, keep.C, keep.H, keep.W);
            else untag(ds, o.out);
        }
        return st;
    }
    case K_ADD: {
        const Tag *a = find_tag(ds, o.in), *b = find_tag(ds, o.in2);
        const bool same = a && b && a->numel() == o.N && a->B == b->B && a->C == b->C && a->H == b->H && a->W == b->W;
        Tag keep{nullptr, 0, 0, 0, 0};
        if (same) {
            keep = *a;
        } else {
            if (a) RN_TRY(materialise(ctx, o.in));
            if (b) RN_TRY(materialise(ctx, o.in2));
        }
        st = rn_add_forward(ctx, o.in, o.in2, o.out, o.N);
        if (same) set_tag(ds, o.out, keep.B, keep.C, keep.H, keep.W);
        else untag(ds, o.out);
        return st;
    }
    case K_MAXPOOL:
    case K_AVGPOOL: {
        const bool nhwc = is_tagged_as(ds, o.in, o.B, o.Cout, o.H, o.W);
        if (!nhwc) RN_TRY(materialise(ctx, o.in));
        ctx->layout = nhwc ? RN_LAYOUT_NHWC : RN_LAYOUT_NCHW;
        st = (o.kind == K_MAXPOOL ? rn_maxpool2d_forward : rn_avgpool2d_forward)(ctx, o.in, o.out, o.k, o.s, o.p, o.ho,
                                                                                  o.wo, o.B, o.Cout, o.H, o.W);
        ctx->layout = RN_LAYOUT_NCHW;
        if (nhwc) set_tag(ds, o.out, o.B, o.Cout, o.ho, o.wo);
        else untag(ds, o.out);
        return st;
    }
    case K_LINEAR:
        RN_TRY(materialise(ctx, o.in));  // (a [B,C,1,1] tensor never carries a tag)
        untag(ds, o.out);
        return rn_linear_forward(ctx, o.in, o.out, o.w, o.b, o.B, o.Cin, o.Cout);
    default:
        return RN_ERR_INVALID;
    }
}

int run_list(rn_ctx *ctx)
{
    rn_defer_state *ds = ctx->ds;
    int st = RN_OK;
    size_t i = 0;
    while (st == RN_OK && i < ds->ops.size()) {
        st = settle_op(ctx, ds->ops[i]);
        if (st != RN_OK) break;
        if (ds->ops[i].kind == K_CONV) {
            size_t next = i + 1;
            st = run_conv(ctx, i, &next);
            i = next;
        } else {
            const Op o = ds->ops[i];
            st = run_literal(ctx, o);
            ++i;
        }
    }
    ds->ops.clear();
    return st;
}

int push(rn_ctx *ctx, const Op &o)
{
    state(ctx)->ops.push_back(o);
    return RN_OK;
}

}  // namespace

// ---- library-internal interface (rn_internal.h) ---------------------------------------------

// run what has been recorded; the tags stay
int rn_defer_flush(rn_ctx *ctx)
{
    if (!ctx->ds || ctx->ds->ops.empty() || ctx->defer_running) return RN_OK;
    ctx->defer_running = 1;
    const int sync_each = ctx->sync_each_op;
    ctx->sync_each_op = 0;  // one check at the end of the list instead of one per launch
    int st = run_list(ctx);
    ctx->sync_each_op = sync_each;
    ctx->layout = RN_LAYOUT_NCHW;
    ctx->defer_running = 0;
    if (st == RN_OK && sync_each) st = rn_check_hip(ctx, hipStreamSynchronize(ctx->stream), "deferred ops");
    return st;
}

// flush, and every caller buffer back to NCHW: what an entry point outside the deferred set sees
int rn_defer_barrier(rn_ctx *ctx)
{
    if (!ctx->ds || ctx->defer_running) return RN_OK;
    RN_TRY(rn_defer_flush(ctx));
    ctx->defer_running = 1;
    int st = RN_OK;
    while (st == RN_OK && !ctx->ds->tags.empty()) st = materialise(ctx, ctx->ds->tags.back().ptr);
    ctx->defer_running = 0;
    return st;
}

// [ptr, ptr + bytes) is about to be overwritten or freed by something outside the list
int rn_defer_before_write(rn_ctx *ctx, const void *ptr, uint64_t bytes, int whole_allocation)
{
    rn_defer_state *ds = ctx->ds;
    if (!ds || ctx->defer_running) return RN_OK;
    RN_TRY(rn_defer_flush(ctx));
    for (size_t i = 0; i < ds->tags.size();) {
        const Tag t = ds->tags[i];
        if (!overlaps(ptr, bytes, t.ptr, t.numel() * sizeof(float))) {
            ++i;
            continue;
        }
        // a write that covers the tensor (or frees it) replaces its content; a partial write lands in an
        // NCHW image of it
        const bool covers = whole_allocation || ((uintptr_t)ptr <= (uintptr_t)t.ptr &&
                                                 (uintptr_t)ptr + bytes >= (uintptr_t)t.ptr + t.numel() * sizeof(float));
        if (covers) {
            untag(ds, t.ptr);
        } else {
            ctx->defer_running = 1;
            const int st = materialise(ctx, t.ptr);
            ctx->defer_running = 0;
            RN_TRY(st);
        }
        i = 0;
    }
    bool synced = false;
    for (size_t i = 0; i < ds->folds.size();) {
        const Fold &f = ds->folds[i];
        const uint64_t cb = f.C * sizeof(float);
        if (overlaps(ptr, bytes, f.w, cb) || overlaps(ptr, bytes, f.b, cb) || overlaps(ptr, bytes, f.m, cb) ||
            overlaps(ptr, bytes, f.v, cb)) {
            if (!synced) (void)hipStreamSynchronize(ctx->stream), synced = true;
            (void)hipFree(f.scale);
            ds->folds[i] = ds->folds.back();
            ds->folds.pop_back();
        } else {
            ++i;
        }
    }
    for (size_t i = 0; i < ds->exact.size();) {
        const ExactPack &e = ds->exact[i];
        if (overlaps(ptr, bytes, e.w, e.cin * e.cout * e.k * e.k * sizeof(float))) {
            if (!synced) (void)hipStreamSynchronize(ctx->stream), synced = true;
            (void)hipFree(e.packed);
            ds->exact[i] = ds->exact.back();
            ds->exact.pop_back();
        } else {
            ++i;
        }
    }
    return RN_OK;
}

// [ptr, ptr + bytes) is about to be read as NCHW.  A read of exactly one whole tagged tensor can take
// its NCHW image from scratch (*alt) and leave the buffer NHWC; anything else rewrites it in place.
int rn_defer_before_read(rn_ctx *ctx, const void *ptr, uint64_t bytes, const void **alt)
{
    rn_defer_state *ds = ctx->ds;
    if (alt) *alt = ptr;
    if (!ds || ctx->defer_running) return RN_OK;
    RN_TRY(rn_defer_flush(ctx));
    for (size_t i = 0; i < ds->tags.size();) {
        const Tag t = ds->tags[i];
        if (!overlaps(ptr, bytes, t.ptr, t.numel() * sizeof(float))) {
            ++i;
            continue;
        }
        ctx->defer_running = 1;
        int st = RN_OK;
        if (alt && t.ptr == ptr && bytes == t.numel() * sizeof(float)) {
            void *tmp = nullptr;
            st = rn_scratch(ctx, 3, bytes, &tmp);
            if (st == RN_OK) st = rn_nhwc_to_nchw(ctx, (const float *)t.ptr, (float *)tmp, t.B, t.C, t.H, t.W);
            ++ds->transposes;
            *alt = tmp;
            ctx->defer_running = 0;
            return st;
        }
        st = materialise(ctx, t.ptr);
        ctx->defer_running = 0;
        RN_TRY(st);
        i = 0;
    }
    return RN_OK;
}

void rn_defer_destroy(rn_ctx *ctx)
{
    if (!ctx->ds) return;
    for (const Fold &f : ctx->ds->folds) (void)hipFree(f.scale);
    for (const ExactPack &e : ctx->ds->exact) (void)hipFree(e.packed);
    delete ctx->ds;
    ctx->ds = nullptr;
}

// the seven reference entry points call these first when the context is deferred (and the list is not
// running): 1 = recorded, 0 = go on and execute
int rn_defer_conv(rn_ctx *ctx, const float *inp, float *out, const float *w, uint64_t k, uint64_t s, uint64_t p,
                  uint64_t ho, uint64_t wo, uint64_t B, uint64_t Cin, uint64_t Cout, uint64_t H, uint64_t W)
{
    Op o;
    memset(&o, 0, sizeof(o));
    o.kind = K_CONV, o.in = inp, o.out = out, o.w = w;
    o.k = k, o.s = s, o.p = p, o.ho = ho, o.wo = wo, o.B = B, o.Cin = Cin, o.Cout = Cout, o.H = H, o.W = W;
    return push(ctx, o);
}

int rn_defer_bn(rn_ctx *ctx, const float *inp, float *out, const float *w, const float *b, const float *mean,
                const float *var, uint64_t B, uint64_t C, uint64_t N)
{
    Op o;
    memset(&o, 0, sizeof(o));
    o.kind = K_BN, o.in = inp, o.out = out, o.w = w, o.b = b, o.mean = mean, o.var = var;
    o.B = B, o.Cout = C, o.N = N;
    return push(ctx, o);
}

int rn_defer_eltwise(rn_ctx *ctx, int is_add, const float *a, const float *b, float *out, uint64_t N)
{
    Op o;
    memset(&o, 0, sizeof(o));
    o.kind = is_add ? K_ADD : K_RELU, o.in = a, o.in2 = b, o.out = out, o.N = N;
    return push(ctx, o);
}

int rn_defer_pool(rn_ctx *ctx, int is_max, const float *inp, float *out, uint64_t k, uint64_t s, uint64_t p,
                  uint64_t ho, uint64_t wo, uint64_t B, uint64_t C, uint64_t H, uint64_t W)
{
    Op o;
    memset(&o, 0, sizeof(o));
    o.kind = is_max ? K_MAXPOOL : K_AVGPOOL, o.in = inp, o.out = out;
    o.k = k, o.s = s, o.p = p, o.ho = ho, o.wo = wo, o.B = B, o.Cout = C, o.H = H, o.W = W;
    return push(ctx, o);
}

int rn_defer_linear(rn_ctx *ctx, const float *inp, float *out, const float *w, const float *b, uint64_t B,
                    uint64_t in_f, uint64_t out_f)
{
    Op o;
    memset(&o, 0, sizeof(o));
    o.kind = K_LINEAR, o.in = inp, o.out = out, o.w = w, o.b = b, o.B = B, o.Cin = in_f, o.Cout = out_f;
    return push(ctx, o);
}

extern "C" {

int rn_ctx_set_deferred(rn_ctx *ctx, int on)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    if (!on && ctx->defer) RN_TRY(rn_defer_barrier(ctx));  // everything runs, every buffer back to NCHW
    ctx->defer = on ? 1 : 0;
    return RN_OK;
}

int rn_ctx_get_deferred(const rn_ctx *ctx) { return ctx ? ctx->defer : 0; }

int rn_flush(rn_ctx *ctx)
{
    if (!ctx) return RN_ERR_INVALID;
    RN_TRY(rn_bind_device(ctx));
    return rn_defer_flush(ctx);
}

int rn_observe(rn_ctx *ctx, const void *dev_ptr)
{
    if (!ctx) return RN_ERR_INVALID;
    if (!ctx->ds) return RN_OK;
    RN_TRY(rn_bind_device(ctx));
    return rn_defer_before_read(ctx, dev_ptr, 1, nullptr);
}

int rn_ctx_deferred_stats(const rn_ctx *ctx, uint64_t *pending_ops, uint64_t *nhwc_buffers, uint64_t *fused_launches,
                          uint64_t *literal_launches, uint64_t *transposes)
{
    if (!ctx) return RN_ERR_INVALID;
    const rn_defer_state *ds = ctx->ds;
    if (pending_ops) *pending_ops = ds ? ds->ops.size() : 0;
    if (nhwc_buffers) *nhwc_buffers = ds ? ds->tags.size() : 0;
    if (fused_launches) *fused_launches = ds ? ds->fused_launches : 0;
    if (literal_launches) *literal_launches = ds ? ds->literal_launches : 0;
    if (transposes) *transposes = ds ? ds->transposes : 0;
    return RN_OK;
}

}  // extern "C"
