/* C driver for the sanitizer passes (tools/sanitize_host.sh): the threaded host paths of liboa_icp.so without Python in the
 * process -- multi-device contexts with one host thread and stream per child (OA_MULTI_THREADS=1 OA_MULTI_OWN_STREAMS=1 in the
 * environment), threaded uploads, early-exit loops (the agreement on the enqueued count), the modal step, per-point outputs,
 * a host thread that fails in the middle of a loop (OA_FAULT_FAIL_GROUP), contexts created and destroyed while others live (the
 * process-wide allocation cache), and a second OS thread driving its own context at the same time.
 * Build: see tools/sanitize_host.sh (clang -fsanitize=thread|address,undefined, linked against the instrumented library). */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oa_icp.h"

enum { N = 6000 };
static float tgt[3 * N], src[3 * N];
static const float eye[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };

static void make_clouds(void)
{
    unsigned s = 12345u;
    for (int i = 0; i < 3 * N; ++i) { s = s * 1664525u + 1013904223u; tgt[i] = (float)(s >> 8) / 8388608.0f - 1.0f; }
    const double a = -0.05, c = cos(a), sn = sin(a);
    for (int i = 0; i < N; ++i) {
        const double x = tgt[3 * i] - 0.01, y = tgt[3 * i + 1] + 0.02, z = tgt[3 * i + 2] - 0.015;
        src[3 * i] = (float)(c * x - sn * y); src[3 * i + 1] = (float)(sn * x + c * y); src[3 * i + 2] = (float)z;
    }
}

static int fails = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAILED: %s (%s)\n", what, oa_last_error()); ++fails; } } while (0)

static oa_settings settings(int iters, int early)
{
    oa_settings st;
    memset(&st, 0, sizeof st);
    st.iters = iters; st.use_target = 1; st.early_exit = early; st.thresh = 0.5; st.target_d = 1e-4;
    return st;
}

static void *single_context_thread(void *arg)
{
    (void)arg;
    for (int rep = 0; rep < 6; ++rep) {
        oa_ctx *c = NULL;
        if (oa_create(&c, 0) != OA_OK) { ++fails; return NULL; }
        oa_settings st = settings(30, 1);
        oa_report rep_;
        int ok = !oa_set_target(c, tgt, N, 0) && !oa_set_source(c, src, N, 0, NULL, 0, 1, 0, 1) && !oa_set_matrices(c, eye, eye)
                 && !oa_run(c, &st, &rep_) && rep_.converged;
        if (!ok) { printf("FAILED: single-context thread, rep %d (%s)\n", rep, oa_last_error()); ++fails; }
        oa_destroy(c);
    }
    return NULL;
}

int main(void)
{
    if (oa_device_count() <= 0) { printf("SAN_DRIVER_SKIPPED no device\n"); return 0; }
    make_clouds();
    pthread_t other;
    pthread_create(&other, NULL, single_context_thread, NULL);      /* its own contexts, concurrently with everything below */

    for (int n_dev = 2; n_dev <= 4; n_dev += 2) {
        int devs[4] = { 0, 0, 0, 0 };
        oa_ctx *m = NULL;
        CHECK(oa_create_multi(&m, devs, n_dev) == OA_OK, "oa_create_multi");
        if (!m) continue;
        CHECK(oa_set_exchange(m, OA_EXCHANGE_MAILBOX) == OA_OK, "oa_set_exchange");
        CHECK(!oa_set_target(m, tgt, N, 0) && !oa_set_source(m, src, N, 0, NULL, 0, 1, 0, 1), "uploads");
        double threads = 0;
        oa_get_stat(m, OA_STAT_HOST_THREADS, &threads);
        oa_report rep;
        for (int k = 0; k < 40; ++k) {                               /* early-exit loops: the threads agree on the count */
            oa_settings st = settings(40, 1);
            CHECK(oa_set_matrices(m, eye, eye) == OA_OK, "oa_set_matrices");
            CHECK(oa_run(m, &st, &rep) == OA_OK && rep.converged, "oa_run (early exit)");
            double lo = 0, hi = 0;
            oa_get_stat(m, OA_STAT_ENQUEUED_MIN, &lo); oa_get_stat(m, OA_STAT_ENQUEUED_MAX, &hi);
            CHECK(lo == hi && lo >= rep.iters_done, "enqueued counts agree");
        }
        CHECK(oa_set_matrices(m, eye, eye) == OA_OK, "oa_set_matrices");
        for (int k = 0; k < 5; ++k) {                                /* the modal step */
            oa_settings st = settings(1, 0);
            double M[16], s6[6];
            CHECK(oa_iterate(m, &st, M, s6) == OA_OK, "oa_iterate");
        }
        {                                                            /* per-point outputs merged from the shards */
            double *A = (double *)malloc(sizeof(double) * 3 * N), *B = (double *)malloc(sizeof(double) * 3 * N), ds[2];
            int64_t K = 0;
            CHECK(oa_make_pairs(m, 0.5, 1, A, B, N, &K, ds) == OA_OK && K > N / 2, "oa_make_pairs");
            free(A); free(B);
        }
        printf("multi context of %d children, %.0f host threads: last run %d iterations, K %lld\n", n_dev, threads, rep.iters_done, (long long)rep.last_K);
        oa_destroy(m);
    }
    {                                                                /* a host thread fails mid-loop: the others stop, the call returns */
        int devs[3] = { 0, 0, 0 };
        setenv("OA_FAULT_FAIL_GROUP", "1", 1); setenv("OA_FAULT_FAIL_ITER", "3", 1);      /* (read by oa_create_multi) */
        oa_ctx *m = NULL;
        CHECK(oa_create_multi(&m, devs, 3) == OA_OK, "oa_create_multi (fault)");
        unsetenv("OA_FAULT_FAIL_GROUP"); unsetenv("OA_FAULT_FAIL_ITER");
        if (m) {
            oa_set_exchange(m, OA_EXCHANGE_MAILBOX);
            oa_set_target(m, tgt, N, 0); oa_set_source(m, src, N, 0, NULL, 0, 1, 0, 1); oa_set_matrices(m, eye, eye);
            oa_settings st = settings(40, 1);
            oa_report rep;
            const int rc = oa_run(m, &st, &rep);
            CHECK(rc == OA_E_HIP, "injected failure is reported");
            oa_set_matrices(m, eye, eye);
            CHECK(oa_run(m, &st, &rep) == OA_OK && rep.converged, "the context runs the next loop");
            oa_destroy(m);
        }
    }
    pthread_join(other, NULL);
    oa_release_cached_memory();
    printf(fails ? "SAN_DRIVER_FAILED %d\n" : "SAN_DRIVER_OK\n", fails);
    return fails ? 1 : 0;
}
