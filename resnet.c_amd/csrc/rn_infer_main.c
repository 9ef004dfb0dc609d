/*
 * rn_infer -- plain-C driver, the main() of the reference (cuda/inference/main.cu:228-254):
 * build the model from weights_bin/, read test_bins/<image>.bin, run the forward pass,
 * print "max index is N" per image.  The reference hard-codes everything (ResNet-152,
 * B = 1, paths); here the same defaults can be overridden from the command line.
 *
 *   rn_infer [--arch 50|101|152] [--weights DIR] [--input FILE] [--batch B]
 *            [--mode fused|ops] [--device N | --devices a,b,c,...]
 *
 * --devices shards the batch contiguously over the listed devices (rn_shard_*: one host
 * thread + context + model per device, no data moves between devices) and prints the class
 * indices in image order, exactly as the single-device run does.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rn_hip.h"

#define CHECK(ctx, expr)                                                                  \
    do {                                                                                  \
        int st_ = (expr);                                                                 \
        if (st_ != RN_OK) {                                                               \
            fprintf(stderr, "rn_infer: %s failed: %s (%s)\n", #expr, rn_status_string(st_), \
                    (ctx) ? rn_last_error(ctx) : "");                                     \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static int run_sharded(const int *devices, int ndev, int arch, const char *weights,
                       const char *input, uint64_t B, int mode)
{
    rn_shard *g = NULL;
    float *host = NULL;
    uint64_t *idx = NULL, b;
    const uint64_t want = B * 3 * 224 * 224;
    FILE *f;
    int st;
    st = rn_shard_create(&g, devices, ndev, arch);
    if (st != RN_OK) { fprintf(stderr, "rn_infer: rn_shard_create: %s\n", rn_status_string(st)); return 1; }
#define SCHECK(expr)                                                                         \
    do {                                                                                     \
        int st_ = (expr);                                                                    \
        if (st_ != RN_OK) {                                                                  \
            fprintf(stderr, "rn_infer: %s failed: %s (%s)\n", #expr, rn_status_string(st_),  \
                    rn_shard_last_error(g));                                                 \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
    SCHECK(rn_shard_load_dir(g, weights));
    SCHECK(rn_shard_finalize(g));
    printf("created model\n");
    {   /* where every shard's host thread runs (stderr: stdout stays what the single-device run prints) */
        int r;
        for (r = 0; r < rn_shard_count(g); ++r) {
            int dev = -1, node = -1;
            char cpus[256];
            if (rn_shard_placement(g, r, &dev, &node, cpus, sizeof(cpus)) == RN_OK)
                fprintf(stderr, "rn_infer: shard %d on device %d, NUMA node %d, host thread on cpus [%s]\n", r, dev, node,
                        cpus[0] ? cpus : "not bound");
        }
    }
    host = (float *)malloc(want * sizeof(float));
    idx = (uint64_t *)malloc(B * sizeof(uint64_t));
    f = fopen(input, "rb");
    if (!host || !idx || !f || fread(host, sizeof(float), want, f) != want || fgetc(f) != EOF) {
        fprintf(stderr, "rn_infer: %s: %s: the model driver takes 3 x 224 x 224 fp32 images only; the file "
                        "does not hold exactly %llu floats (batch %llu)\n", rn_status_string(RN_ERR_UNSUPPORTED),
                input, (unsigned long long)want, (unsigned long long)B);
        return 1;
    }
    fclose(f);
    SCHECK(rn_shard_forward(g, host, B, NULL, idx, mode));
#undef SCHECK
    printf("Finished\n");
    for (b = 0; b < B; ++b) printf("max index is %llu\n", (unsigned long long)idx[b]);
    free(idx);
    free(host);
    rn_shard_destroy(g);
    return 0;
}

int main(int argc, char **argv)
{
    int arch = 152, device = 0, mode = RN_FWD_FUSED, i;
    int devices[64], ndev = 0;
    uint64_t B = 1, numel = 0, b;
    const char *weights = "weights_bin";
    const char *input = "test_bins/ILSVRC2012_val_00004749.bin";
    rn_ctx *ctx = NULL;
    rn_model *model = NULL;
    float *inp = NULL, *logits = NULL;
    uint64_t *idx_dev = NULL, *idx = NULL;

    for (i = 1; i < argc; ++i) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--arch") && v) { arch = atoi(v); ++i; }
        else if (!strcmp(a, "--weights") && v) { weights = v; ++i; }
        else if (!strcmp(a, "--input") && v) { input = v; ++i; }
        else if (!strcmp(a, "--batch") && v) { B = strtoull(v, NULL, 10); ++i; }
        else if (!strcmp(a, "--device") && v) { device = atoi(v); ++i; }
        else if (!strcmp(a, "--devices") && v) {
            const char *q = v;
            while (*q && ndev < 64) {
                devices[ndev++] = (int)strtol(q, (char **)&q, 10);
                if (*q == ',') ++q;
            }
            ++i;
        }
        else if (!strcmp(a, "--mode") && v) { mode = strcmp(v, "ops") ? RN_FWD_FUSED : RN_FWD_REFERENCE_OPS; ++i; }
        else {
            fprintf(stderr, "usage: %s [--arch 50|101|152] [--weights DIR] [--input FILE] "
                            "[--batch B] [--mode fused|ops] [--device N | --devices a,b,...]\n", argv[0]);
            return 2;
        }
    }
    printf("Started\n");
    if (ndev > 0) return run_sharded(devices, ndev, arch, weights, input, B, mode);
    CHECK(ctx, rn_ctx_create(&ctx, device, NULL));
    CHECK(ctx, rn_model_create(ctx, &model, arch));
    CHECK(ctx, rn_model_load_dir(model, weights));
    CHECK(ctx, rn_model_finalize(model));
    printf("created model\n");

    CHECK(ctx, rn_load_f32_file(ctx, input, &inp, &numel));
    if (numel != B * 3 * 224 * 224) {
        fprintf(stderr, "rn_infer: %s: the model driver takes 3 x 224 x 224 fp32 images only; %s holds "
                        "%llu floats, expected %llu for batch %llu\n", rn_status_string(RN_ERR_UNSUPPORTED), input,
                (unsigned long long)numel, (unsigned long long)(B * 3 * 224 * 224),
                (unsigned long long)B);
        return 1;
    }
    CHECK(ctx, rn_malloc(ctx, (void **)&logits, B * 1000 * sizeof(float)));
    CHECK(ctx, rn_malloc(ctx, (void **)&idx_dev, B * sizeof(uint64_t)));
    CHECK(ctx, rn_model_forward(model, inp, B, logits, mode));
    CHECK(ctx, rn_argmax_forward(ctx, logits, idx_dev, B, 1000));
    idx = (uint64_t *)malloc(B * sizeof(uint64_t));
    if (!idx) return 1;
    CHECK(ctx, rn_memcpy_d2h(ctx, idx, idx_dev, B * sizeof(uint64_t)));
    printf("Finished\n");
    for (b = 0; b < B; ++b) printf("max index is %llu\n", (unsigned long long)idx[b]);

    free(idx);
    rn_free(ctx, idx_dev);
    rn_free(ctx, logits);
    rn_free(ctx, inp);
    rn_model_destroy(model);
    rn_ctx_destroy(ctx);
    return 0;
}
