/*
 * rn_infer -- plain-C driver, the main() of the reference (cuda/inference/main.cu:228-254):
 * build the model from weights_bin/, read test_bins/<image>.bin, run the forward pass,
 * print "max index is N" per image.  The reference hard-codes everything (ResNet-152,
 * B = 1, paths); here the same defaults can be overridden from the command line.
 *
 *   rn_infer [--arch 50|101|152] [--weights DIR] [--input FILE] [--batch B]
 *            [--mode fused|ops] [--device N]
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

int main(int argc, char **argv)
{
    int arch = 152, device = 0, mode = RN_FWD_FUSED, i;
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
        else if (!strcmp(a, "--mode") && v) { mode = strcmp(v, "ops") ? RN_FWD_FUSED : RN_FWD_REFERENCE_OPS; ++i; }
        else {
            fprintf(stderr, "usage: %s [--arch 50|101|152] [--weights DIR] [--input FILE] "
                            "[--batch B] [--mode fused|ops] [--device N]\n", argv[0]);
            return 2;
        }
    }
    printf("Started\n");
    CHECK(ctx, rn_ctx_create(&ctx, device, NULL));
    CHECK(ctx, rn_model_create(ctx, &model, arch));
    CHECK(ctx, rn_model_load_dir(model, weights));
    CHECK(ctx, rn_model_finalize(model));
    printf("created model\n");

    CHECK(ctx, rn_load_f32_file(ctx, input, &inp, &numel));
    if (numel != B * 3 * 224 * 224) {
        fprintf(stderr, "rn_infer: %s holds %llu floats, expected %llu for batch %llu\n", input,
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
