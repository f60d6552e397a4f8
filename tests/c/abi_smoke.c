/* Plain-C consumer of include/oa_icp.h: proves the header is valid C and that the library can be driven without
 * Python.  Without a GPU it checks that oa_create fails loudly (OA_E_NO_DEVICE); with one it aligns a tiny cloud.
 * Build: gcc -std=c99 -I include tests/c/abi_smoke.c -L object_alignment_amd -loa_icp -lm -o abi_smoke */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oa_icp.h"

int main(void)
{
    printf("version: %s\n", oa_version());
    oa_ctx *ctx = NULL;
    int n_dev = oa_device_count();
    int rc = oa_create(&ctx, 0);
    if (n_dev <= 0) {
        if (rc != OA_E_NO_DEVICE || ctx != NULL) { printf("expected OA_E_NO_DEVICE, got %d\n", rc); return 1; }
        printf("no device: %s\n", oa_last_error());
        printf("ABI_SMOKE_OK nodevice\n");
        return 0;
    }
    if (rc != OA_OK) { printf("oa_create: %s\n", oa_last_error()); return 1; }

    enum { N = 4000 };
    float *tgt = (float *)malloc(sizeof(float) * 3 * N), *src = (float *)malloc(sizeof(float) * 3 * N);
    unsigned s = 12345u;
    for (int i = 0; i < 3 * N; ++i) { s = s * 1664525u + 1013904223u; tgt[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
    /* source = target rotated by -0.05 rad about z and shifted: ICP must bring it back */
    const double a = -0.05, c = cos(a), sn = sin(a);
    for (int i = 0; i < N; ++i) {
        const double x = tgt[3 * i] - 0.01, y = tgt[3 * i + 1] + 0.02, z = tgt[3 * i + 2] - 0.015;
        src[3 * i] = (float)(c * x - sn * y); src[3 * i + 1] = (float)(sn * x + c * y); src[3 * i + 2] = (float)z;
    }
    const float eye[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    if (oa_set_target(ctx, tgt, N, 0) || oa_set_source(ctx, src, N, 0, NULL, 0, 1, 0, 1) || oa_set_matrices(ctx, eye, eye)) {
        printf("setup failed: %s\n", oa_last_error());
        return 1;
    }
    oa_settings st;
    memset(&st, 0, sizeof st);
    st.iters = 40; st.use_target = 1; st.with_scale = 0; st.early_exit = 1; st.thresh = 0.5; st.target_d = 1e-4;
    oa_report rep;
    if (oa_run(ctx, &st, &rep) != OA_OK) { printf("oa_run: %s\n", oa_last_error()); return 1; }
    float mw[16];
    oa_get_matrix_world(ctx, mw);
    printf("iters %d converged %d K %lld mean_dist %.3g  t = (%.4f %.4f %.4f)\n", rep.iters_done, rep.converged,
           (long long)rep.last_K, rep.mean_dist, mw[3], mw[7], mw[11]);
    /* expected matrix_world = inverse of the motion applied above: rotation +0.05 about z, then the shift undone */
    const double ex = fabs(mw[0] - cos(0.05)) + fabs(mw[1] + sin(0.05)) + fabs(mw[3] - 0.01) + fabs(mw[7] + 0.02) + fabs(mw[11] - 0.015);
    oa_destroy(ctx);
    if (!rep.converged || ex > 1e-3) { printf("unexpected result (err %.3g)\n", ex); return 1; }

    /* the same job through a multi-device context (oa_create(&ctx, devices, n_dev)): every visible GPU, or
     * -- on a one-GPU box -- device 0 listed twice; no torch, no launcher, the exchange lives inside the library */
    int devs[64];
    int n_multi = n_dev >= 2 ? (n_dev > 64 ? 64 : n_dev) : 2;
    for (int i = 0; i < n_multi; ++i) devs[i] = n_dev >= 2 ? i : 0;
    oa_ctx *mctx = NULL;
    if (oa_create_multi(&mctx, devs, n_multi) != OA_OK) { printf("oa_create_multi: %s\n", oa_last_error()); return 1; }
    if (oa_num_devices(mctx) != n_multi) { printf("oa_num_devices\n"); return 1; }
    if (oa_set_target(mctx, tgt, N, 0) || oa_set_source(mctx, src, N, 0, NULL, 0, 1, 0, 1) || oa_set_matrices(mctx, eye, eye)) {
        printf("multi setup failed: %s\n", oa_last_error());
        return 1;
    }
    if (oa_num_selected(mctx) != N) { printf("multi: %lld selected\n", (long long)oa_num_selected(mctx)); return 1; }
    oa_report mrep;
    if (oa_run(mctx, &st, &mrep) != OA_OK) { printf("multi oa_run: %s\n", oa_last_error()); return 1; }
    float mw2[16];
    oa_get_matrix_world(mctx, mw2);
    double dm = 0.0;
    for (int i = 0; i < 16; ++i) dm += fabs((double)mw2[i] - (double)mw[i]);
    printf("multi (%d devices): iters %d converged %d K %lld  |dM|_1 vs single = %.3g\n", n_multi, mrep.iters_done,
           mrep.converged, (long long)mrep.last_K, dm);
    oa_destroy(mctx);
    free(tgt); free(src);
    if (mrep.iters_done != rep.iters_done || mrep.last_K != rep.last_K || dm > 1e-5) { printf("multi-device result differs\n"); return 1; }
    printf("ABI_SMOKE_OK device\n");
    return 0;
}
