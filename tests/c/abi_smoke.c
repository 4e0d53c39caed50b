/* include/nwe.h from plain C: the header must compile as C99, every declared entry point must link, and the host-only
 * paths (context, validation, packing) must behave.  Built and run by tests/test_abi.py with gcc; no GPU needed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nwe.h"

#define CHECK(cond)                                                    \
    do {                                                               \
        if (!(cond)) {                                                 \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                  \
        }                                                              \
    } while (0)

int main(void) {
    nwe_ctx *ctx = NULL;
    CHECK(nwe_create(&ctx, -1) == NWE_OK && ctx);           /* host-only context: packs, cannot render */
    enum { D = 4, W = 128, IN_XYZ = 63, IN_DIR = 27 };
    const int ins[D + 4] = {IN_XYZ, W, W, W, W + IN_DIR, W, W, W / 2};
    const int outs[D + 4] = {W, W, W, W, W / 2, W, 1, 3};
    float *w[D + 4], *b[D + 4];
    unsigned s = 12345u;
    for (int l = 0; l < D + 4; ++l) {
        w[l] = (float *)malloc(sizeof(float) * ins[l] * outs[l]);
        b[l] = (float *)calloc(outs[l], sizeof(float));
        for (int i = 0; i < ins[l] * outs[l]; ++i) {
            s = s * 1664525u + 1013904223u;
            w[l][i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.1f;
        }
    }
    CHECK(nwe_set_network(ctx, NWE_NET_COARSE, D, W, IN_XYZ, IN_DIR, -1, (const float *const *)w, (const float *const *)b) == NWE_OK);
    CHECK(nwe_flops_per_eval(ctx, NWE_NET_COARSE) == 167680);                       /* BASELINE.md section 2 */
    CHECK(nwe_packed_bytes(ctx, NWE_NET_COARSE) == 272 * 1024);                     /* 4x128, feature layer folded, no alpha tile */
    CHECK(nwe_packed_bias_count(ctx, NWE_NET_COARSE) == (19 + 5) * 32);             /* 19 chunks + the alpha layer's 4 + 1 dot rows */
    const float scale = nwe_packed_scale(ctx, NWE_NET_COARSE);
    CHECK(scale > 0.f && fabsf(log2f(scale) - roundf(log2f(scale))) == 0.f);        /* a power of two */
    CHECK(nwe_debug_set_fold(ctx, 0) == NWE_OK);
    CHECK(nwe_set_network(ctx, NWE_NET_FINE, D, W, IN_XYZ, IN_DIR, -1, (const float *const *)w, (const float *const *)b) == NWE_OK);
    CHECK(nwe_packed_bytes(ctx, NWE_NET_FINE) == 352 * 1024);                       /* the reference's formulation */
    CHECK(nwe_set_network(ctx, 2, D, W, IN_XYZ, IN_DIR, -1, (const float *const *)w, (const float *const *)b) == NWE_ERR_INVALID);
    CHECK(strlen(nwe_last_error(ctx)) > 0);
    float t[4] = {0.f, 1.f / 3, 2.f / 3, 1.f}, omt[4] = {1.f, 2.f / 3, 1.f / 3, 0.f};
    CHECK(nwe_set_sampling(ctx, t, omt, 4, NULL, 0) == NWE_OK);
    CHECK(nwe_set_sampling(ctx, t, omt, 1, NULL, 0) == NWE_ERR_UNSUPPORTED);
    nwe_outputs out;
    memset(&out, 0, sizeof out);
    out.struct_bytes = sizeof out;
    CHECK(nwe_render_rays(ctx, NULL, 0, NWE_PREC_F16X3, &out, NULL) == NWE_ERR_STATE);  /* host-only context cannot render */
    out.struct_bytes = sizeof out - sizeof(void *);
    CHECK(nwe_render_rays(ctx, NULL, 0, NWE_PREC_F16X3, &out, NULL) == NWE_ERR_INVALID); /* an older header's layout */
    CHECK(nwe_render_tiled(NULL, 0, NULL, 0, 0, 0, 1, 1, 0, 0, 0.1f, 10.f, 0, NULL, NULL, NULL, NULL, NULL) == NWE_ERR_INVALID);
    CHECK(nwe_debug_last_plan(ctx) == -1 && nwe_last_kernel_ms(ctx) < 0.f);
    CHECK(strlen(nwe_last_warning(ctx)) == 0 && nwe_debug_peer_access(ctx, ctx) == -1);   /* host-only contexts have no device */
    {
        float ms2[2]; int64_t rays2[2];
        CHECK(nwe_last_launch_parts(ctx, ms2, rays2) == NWE_ERR_STATE && ms2[0] < 0.f && rays2[0] == 0);
        CHECK(nwe_last_launch_parts(ctx, NULL, rays2) == NWE_ERR_INVALID);
    }
    /* the packed stream and bias table come back whole (and the copies refuse a wrong size) */
    {
        const int64_t nb = nwe_packed_bytes(ctx, NWE_NET_FINE), nbias = nwe_packed_bias_count(ctx, NWE_NET_FINE);
        unsigned char *buf = (unsigned char *)malloc((size_t)nb);
        float *bias = (float *)malloc(sizeof(float) * (size_t)nbias);
        CHECK(buf && bias);
        CHECK(nwe_packed_copy(ctx, NWE_NET_FINE, buf, nb) == NWE_OK && nwe_packed_copy(ctx, NWE_NET_FINE, buf, nb - 1) == NWE_ERR_INVALID);
        CHECK(nwe_packed_bias_copy(ctx, NWE_NET_FINE, bias, nbias) == NWE_OK && nwe_packed_bias_copy(ctx, NWE_NET_FINE, bias, nbias + 1) == NWE_ERR_INVALID);
        free(buf); free(bias);
    }
    /* every remaining entry point is at least referenced, so that a missing symbol fails the link */
    typedef void (*fn)(void);
    fn refs[] = {(fn)nwe_render, (fn)nwe_create_rays, (fn)nwe_to8b, (fn)nwe_packed_copy, (fn)nwe_packed_bias_copy,
                 (fn)nwe_debug_set_fine_depths, (fn)nwe_debug_set_raw, (fn)nwe_debug_set_coarse_weights, (fn)nwe_set_white_background,
                 (fn)nwe_set_train_tables, (fn)nwe_debug_set_decomposition, (fn)nwe_debug_set_stamps, (fn)nwe_selftest,
                 (fn)nwe_last_warning, (fn)nwe_debug_peer_access,
                 (fn)nwe_set_network_no_view_dirs, (fn)nwe_last_launch_parts};
    for (unsigned i = 0; i < sizeof refs / sizeof refs[0]; ++i) CHECK(refs[i] != (fn)0);
    nwe_destroy(ctx);
    for (int l = 0; l < D + 4; ++l) { free(w[l]); free(b[l]); }
    printf("abi_smoke ok\n");
    return 0;
}
