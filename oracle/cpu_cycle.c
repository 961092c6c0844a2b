/* TEST INFRASTRUCTURE / CPU BASELINE -- never linked into libipdamg.so, never on the product path.
 *
 * C restatement of the reference's solve-phase loop on a FIXED hierarchy, for timing on the GPU
 * box's host cores (bench.py `cpu_baseline`, 1 thread and all cores) and checked against the
 * Python oracle by tests/test_cpu_cycle.py (parity unpinned against real MATLAB, like the oracle
 * itself: the reference ships no golden outputs).
 *
 *   AMG/Class_AMG.m:95-107   x = x + MG_cycle(b - A*x)         -> ipdo_cycles
 *   AMG/MG_Vcycle.m:12-45    smooth, restrict, recurse, prolong, smooth, PCG on the coarsest level
 *   AMG/MG_Wcycle.m:13-46    the same with two recursive corrections (:28-30)
 *   PCG.m:68-87              Shewchuk B3 with the Jacobi preconditioner (2-argument defaults :18-23)
 *
 * Like the reference (and oracle/ipd_oracle.py) the smoother applies EXPLICIT matrices:
 * g = r - A e; e += R g with Rk{1} the bigraph Gauss-Seidel inverse (Class_AMG.m:56-59) and
 * Rk{k>1} = 0.5 D^-1 (:84); the post-smoother applies R' (MG_Vcycle.m:34).  Every matrix is CSR
 * with int32 indices; row loops are OpenMP-parallel, sums inside a row and every dot product are
 * sequential in ascending index order (as MATLAB's, for one thread; with T threads the dot
 * products are reduced over T partial sums, which is what a threaded MATLAB BLAS does too).
 *
 *   gcc -O3 -fopenmp -shared -fPIC oracle/cpu_cycle.c -o oracle/libipd_cpu_cycle.so -lm
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int nr, nc;
    const int* rp;
    const int* ci;
    const double* va;
} csr;

#define MAXLEV 40
typedef struct {
    int J;
    csr A[MAXLEV], R[MAXLEV], Rt[MAXLEV], P[MAXLEV], Pt[MAXLEV]; /* P[k]: N_{k-1} x N_k, k >= 2 */
    double* Axi[MAXLEV];
    double xx[MAXLEV];
    double* diag; /* coarsest level */
    int nu, isnsp;
} hier;

static hier H;

static void spmv(const csr* M, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < M->nr; ++i) {
        double s = 0.0;
        for (int t = M->rp[i]; t < M->rp[i + 1]; ++t) s += M->va[t] * x[M->ci[t]];
        y[i] = s;
    }
}

static double dot(const double* a, const double* b, int n) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static double vsum(const double* a, int n) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int i = 0; i < n; ++i) s += a[i];
    return s;
}

void ipdo_reset(int J, int nu, int isnsp) {
    for (int k = 0; k < MAXLEV; ++k) {
        free(H.Axi[k]);
        H.Axi[k] = NULL;
    }
    free(H.diag);
    memset(&H, 0, sizeof(H));
    H.J = J;
    H.nu = nu;
    H.isnsp = isnsp;
}

/* which: 0 A_k, 1 Rk{k}, 2 Rk{k}', 3 Prok{k} (k >= 2), 4 Prok{k}'.  Arrays stay owned by the caller. */
void ipdo_set(int k, int which, int nr, int nc, const int* rp, const int* ci, const double* va) {
    csr m = {nr, nc, rp, ci, va};
    if (which == 0) {
        H.A[k] = m;
        /* A*1 and 1'*A*1 (MG_Vcycle.m:15-16) */
        H.Axi[k] = (double*)malloc(sizeof(double) * (size_t)nr);
        double* one = (double*)malloc(sizeof(double) * (size_t)nc);
        for (int i = 0; i < nc; ++i) one[i] = 1.0;
        spmv(&m, one, H.Axi[k]);
        double s = 0.0;
        for (int i = 0; i < nr; ++i) s += H.Axi[k][i];
        H.xx[k] = s;
        free(one);
        if (k == H.J) {
            H.diag = (double*)malloc(sizeof(double) * (size_t)nr);
            for (int i = 0; i < nr; ++i) {
                H.diag[i] = 0.0;
                for (int t = rp[i]; t < rp[i + 1]; ++t)
                    if (ci[t] == i) H.diag[i] = va[t];
            }
        }
    } else if (which == 1)
        H.R[k] = m;
    else if (which == 2)
        H.Rt[k] = m;
    else if (which == 3)
        H.P[k] = m;
    else
        H.Pt[k] = m;
}

/* nu sweeps e += R (r - A e) with the kernel-space correction (MG_Vcycle.m:15-25) */
static void smooth(int k, const csr* R, const double* r, double* e, double* g, double* t) {
    const csr* A = &H.A[k];
    const int N = A->nr;
    for (int s = 0; s < H.nu; ++s) {
        spmv(A, e, g);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) g[i] = r[i] - g[i];
        if (H.isnsp) {
            const double c = vsum(g, N) / H.xx[k];
            const double* axi = H.Axi[k];
#pragma omp parallel for schedule(static)
            for (int i = 0; i < N; ++i) g[i] = g[i] - axi[i] * c;
            spmv(R, g, t);
#pragma omp parallel for schedule(static)
            for (int i = 0; i < N; ++i) e[i] = e[i] + (c + t[i]);
        } else {
            spmv(R, g, t);
#pragma omp parallel for schedule(static)
            for (int i = 0; i < N; ++i) e[i] = e[i] + t[i];
        }
    }
}

/* PCG(A, r): tol 1e-11, maxit 1e4, Jacobi (PCG.m:18-23,68-87) */
static void pcg(int k, const double* rhs, double* d) {
    const csr* A = &H.A[k];
    const int N = A->nr;
    double* r = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    double *p = r + N, *q = p + N;
    double dn = 0.0;
    for (int i = 0; i < N; ++i) {
        d[i] = 0.0;
        r[i] = rhs[i];
        p[i] = r[i] / H.diag[i];
        dn += r[i] * p[i];
    }
    const double d0 = dn;
    int it = 0;
    while (it < 10000 && dn > 1e-11 * 1e-11 * d0) {
        spmv(A, p, q);
        double qp = 0.0;
        for (int i = 0; i < N; ++i) qp += q[i] * p[i];
        const double alpha = dn / qp;
        const double dold = dn;
        dn = 0.0;
        for (int i = 0; i < N; ++i) {
            d[i] += alpha * p[i];
            r[i] -= alpha * q[i];
            q[i] = r[i] / H.diag[i];
            dn += r[i] * q[i];
        }
        const double beta = dn / dold;
        for (int i = 0; i < N; ++i) p[i] = q[i] + beta * p[i];
        ++it;
    }
    free(r);
}

static void cycle(int k, int wc, const double* r, double* e, int keep) {
    const int N = H.A[k].nr;
    if (k == H.J) {
        pcg(k, r, e);
        return;
    }
    double* g = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    double *t = g + N, *rr = t + N;
    if (!keep) memset(e, 0, sizeof(double) * (size_t)N);
    smooth(k, &H.R[k], r, e, g, t);                       /* :14-25 */
    spmv(&H.A[k], e, rr);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) rr[i] = r[i] - rr[i];     /* :27 */
    const int Nc = H.A[k + 1].nr;
    double* rc = (double*)malloc(sizeof(double) * 2 * (size_t)Nc);
    double* ec = rc + Nc;
    spmv(&H.Pt[k + 1], rr, rc);
    cycle(k + 1, wc, rc, ec, 0);                          /* :29 */
    if (wc) cycle(k + 1, wc, rc, ec, 1);                  /* MG_Wcycle.m:30 */
    spmv(&H.P[k + 1], ec, t);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) e[i] = e[i] + t[i];       /* :31 */
    smooth(k, &H.Rt[k], r, e, g, t);                      /* :33-41 */
    free(rc);
    free(g);
}

/* `cycles` loop bodies of Class_AMG.m:96-103 on x; returns the seconds they took, res[c] = ||A x - b||
 * after cycle c (res[0] before the first). */
double ipdo_cycles(const double* b, double* x, int cycles, int wcycle, int threads, double* res) {
    omp_set_num_threads(threads > 0 ? threads : 1);
    const csr* A = &H.A[1];
    const int N = A->nr;
    double* r = (double*)malloc(sizeof(double) * 2 * (size_t)N);
    double* e = r + N;
    spmv(A, x, r);
    for (int i = 0; i < N; ++i) r[i] = b[i] - r[i];
    if (res) res[0] = sqrt(dot(r, r, N));
    const double t0 = omp_get_wtime();
    for (int c = 0; c < cycles; ++c) {
        spmv(A, x, r);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) r[i] = b[i] - r[i];   /* :96 */
        cycle(1, wcycle, r, e, 0);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) x[i] = x[i] + e[i];   /* :98,101 */
        spmv(A, x, e);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i) e[i] = e[i] - b[i];
        const double nr = sqrt(dot(e, e, N));             /* :103 */
        if (res) res[c + 1] = nr;
    }
    const double el = omp_get_wtime() - t0;
    free(r);
    return el;
}

int ipdo_max_threads(void) { return omp_get_max_threads(); }
