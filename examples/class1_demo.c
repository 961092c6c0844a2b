/* class1_demo.c -- the drop-in boundary used from plain C: a Class 1 (optimal transport)
 * problem solved end to end through include/ipd_amg.h, no Python, no MATLAB.
 *
 *   make -C codes_of_ipd_ssn_amg_method_amd/csrc example
 *   ./examples/class1_demo [N]          (needs an MI355X; there is no CPU fallback)
 *
 * Mirrors Class1/APD_SsN_Class1.m: data (c, r, l, p, q, gama) -> warmup_class1 (:59) ->
 * the APD / semismooth-Newton loop with inner_solver = 4 (:101-275).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ipd_amg.h"

static double urand(unsigned long long* s) { /* 53-bit LCG draw in (0,1): synthetic data only */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(*s >> 11) + 0.5) / 9007199254740992.0;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc__ = (call);                                                   \
        if (rc__ != IPD_OK) {                                                \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, ipd_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 256;
    const int64_t m = N, n = N, mn = m * n;
    unsigned long long seed = 12345;
    double* c = malloc(sizeof(double) * (size_t)mn);
    double* r = malloc(sizeof(double) * (size_t)n);
    double* l = malloc(sizeof(double) * (size_t)m);
    double* p = malloc(sizeof(double) * (size_t)m);
    double* q = malloc(sizeof(double) * (size_t)n);
    double sr = 0.0, sl = 0.0;
    for (int64_t k = 0; k < mn; ++k) c[k] = urand(&seed);
    for (int64_t j = 0; j < n; ++j) { r[j] = urand(&seed); q[j] = 1.0; sr += r[j]; }
    for (int64_t i = 0; i < m; ++i) { l[i] = urand(&seed); p[i] = 1.0; sl += l[i]; }
    for (int64_t i = 0; i < m; ++i) l[i] *= sr / sl;            /* <r,1> = <l,1> */

    ipd_ctx* ctx = NULL;
    ipd_rng* rng = NULL;
    ipd_apd* ws = NULL;
    CHECK(ipd_ctx_create(0, &ctx));
    CHECK(ipd_rng_create(5489u, &rng));                          /* rng('default') */
    ipd_apd_data d = {0};
    d.cls = 1; d.m = m; d.n = n; d.c = c; d.r = r; d.l = l; d.p = p; d.q = q;
    d.gama = NULL; d.gama_scalar = INFINITY;
    CHECK(ipd_apd_create(ctx, &d, &ws));
    CHECK(ipd_apd_warmup(ws, 0.0, 100));                         /* warmup_class1(...,0,1e2) */

    ipd_apd_opts opts;
    ipd_apd_opts_init(1, &opts);
    ipd_amg_opts amg;
    ipd_amg_opts_init(&amg);                                     /* APD_SsN_Class1.m:87-88 */
    amg.retol = 1e-11; amg.bigph = 1; amg.maxit = 30; amg.theta = 0.25; amg.smoth = 5;
    amg.cycle = 'w'; amg.isnsp = 1; amg.inter = 1;
    ipd_apd_result res;
    CHECK(ipd_apd_run(ws, &opts, &amg, rng, opts.maxit, &res));

    double* x = malloc(sizeof(double) * (size_t)mn);
    double* lam = malloc(sizeof(double) * (size_t)(m + n));
    CHECK(ipd_apd_get_state(ws, x, NULL, lam, NULL));
    double viol = 0.0, xmin = 0.0;                               /* ||A x - b||, min x on the host */
    for (int64_t j = 0; j < n; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < m; ++i) s += x[i + j * m];
        viol += (s - r[j]) * (s - r[j]);
    }
    for (int64_t i = 0; i < m; ++i) {
        double s = 0.0;
        for (int64_t j = 0; j < n; ++j) s += x[i + j * m];
        viol += (s - l[i]) * (s - l[i]);
    }
    for (int64_t k = 0; k < mn; ++k) if (x[k] < xmin) xmin = x[k];
    printf("class1_demo N=%lld converged=%d k=%d fval=%.9f rr=%.3e feas=%.3e xmin=%.1e "
           "SumAMG=%lld FailAMG=%lld\n",
           (long long)N, res.converged, res.k, res.fval, res.rr, sqrt(viol), xmin,
           (long long)res.sum_amg, (long long)res.fail_amg);
    ipd_apd_destroy(ws);
    ipd_rng_destroy(rng);
    ipd_ctx_destroy(ctx);
    free(c); free(r); free(l); free(p); free(q); free(x); free(lam);
    return res.converged ? 0 : 2;
}
