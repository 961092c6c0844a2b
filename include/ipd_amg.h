/*
 * ipd_amg.h -- C ABI of libipdamg, the MI355X-native (gfx950, hand-written HIP)
 * implementation of the AMG V/W-cycle solve and the ASAt KKT assembly of
 * zihang-student/Codes-of-IPD-SsN-AMG-method.
 *
 * The reference has no FFI layer: its boundary is the MATLAB function
 * signature set itself.  Each entry point below names the
 * MATLAB signature (reference file:line) it replaces; the MEX gateway that a
 * maintainer adds on the reference side is `codes_of_ipd_ssn_amg_method_amd/
 * mex/ipd_mex.cpp`, described in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, a negative IPD_E_* code on failure;
 *    ipd_last_error() returns the thread-local message of the last failure
 *    (the MEX shim forwards it through mexErrMsgIdAndTxt, mirroring MATLAB's
 *    error()).
 *  - sparse matrices cross the boundary in MATLAB's own layout: compressed
 *    sparse column, 0-based int64 indices (binary compatible with mwIndex),
 *    float64 values, row indices ascending, no explicit zeros.
 *  - dense vectors are float64, matrices column-major; logical vectors are one
 *    byte per entry (mxLogical).
 *  - the caller owns every input buffer; nothing is retained after the call
 *    except by an explicit handle (ipd_amg, ipd_dmat).  Output matrices are
 *    library-owned host buffers released with ipd_csc_free().
 *  - pointers are HOST pointers unless the function name ends in `_dev`.
 *  - all arithmetic is IEEE float64; there is no CPU fallback: every entry
 *    point fails with IPD_E_HIP when no gfx950 device is usable.
 */
#ifndef IPD_AMG_H
#define IPD_AMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPD_VERSION 100 /* 0.1.0 */

enum {
    IPD_OK = 0,
    IPD_E_ARG = -1,      /* bad argument (the reference's error() guards)    */
    IPD_E_HIP = -2,      /* HIP runtime failure / no device                  */
    IPD_E_NOMEM = -3,
    IPD_E_LIMIT = -4,    /* size above a documented library limit            */
    IPD_E_NUMERIC = -5,  /* e.g. coarsening stalled, zero pivot              */
    IPD_E_UNSUPPORTED = -6,
    IPD_E_COMM = -7      /* RCCL failure                                     */
};

typedef struct ipd_ctx ipd_ctx;   /* one per process+GPU: device, stream, workspace   */
typedef struct ipd_rng ipd_rng;   /* MATLAB-compatible rand stream (mt19937ar)        */
typedef struct ipd_amg ipd_amg;   /* device-resident hierarchy: Ack, Prok, Rk, J      */
typedef struct ipd_dmat ipd_dmat; /* device-resident sparse matrix (CSR, int32+fp64)  */

/* Host view of a MATLAB sparse matrix (read-only input). */
typedef struct ipd_csc {
    int64_t nrows, ncols, nnz;
    const int64_t* jc; /* ncols+1 column pointers */
    const int64_t* ir; /* nnz row indices         */
    const double* pr;  /* nnz values              */
} ipd_csc;

/* Library-owned output matrix (same layout); release with ipd_csc_free. */
typedef struct ipd_csc_out {
    int64_t nrows, ncols, nnz;
    int64_t* jc;
    int64_t* ir;
    double* pr;
} ipd_csc_out;

/* amg_options struct (Class1/APD_SsN_Class1.m:87-88, AMG/Class_AMG.m:20-34).
 * Unset numeric fields use the sentinel -1 ("isempty" in MATLAB) and receive
 * Class_AMG's empty-field defaults. */
typedef struct ipd_amg_opts {
    double retol;   /* default 1e-12 */
    int32_t bigph;  /* default 0     */
    int32_t maxit;  /* default 50    */
    double theta;   /* default 1/4   */
    int32_t smoth;  /* default 3     */
    int32_t cycle;  /* 'v' or 'w' (ASCII); default 'v'; any other value: no correction (quirk A-8) */
    int32_t isnsp;  /* default 0     */
    int32_t inter;  /* default 1 (ignored for inter<2, transfer.m:54 quirk)  */
    int64_t fnode;  /* required >0 when bigph                                */
} ipd_amg_opts;

/* pcg_options struct (PCG.m:18-27). */
typedef struct ipd_pcg_opts {
    double retol;   /* default 1e-11 */
    int64_t maxit;  /* default 10000 */
    int32_t precd;  /* 1 none, 2 Jacobi, 3 SSOR (w = 1.5), 4 ichol(H) with MATLAB's defaults
                       (IC(0), PCG.m:46), 5 SSOR on the bigraph blocks              */
    int64_t nf;     /* precd 5: size of the F block (pcg_options.nf, PCG.m:55)      */
} ipd_pcg_opts;

/* prob_data struct (Class1/APD_SsN_Class1.m:154-156, Class2/APD_SsN_Class2.m:163-166). */
typedef struct ipd_prob {
    int64_t m, n;      /* length(p), length(q)                                */
    double bk1, tk;
    const double* p;   /* m */
    const double* q;   /* n */
    const double* t;   /* diag(T), n+m entries, or NULL for T = 0             */
    const ipd_csc* H0; /* (n+m) x (n+m), output of ASAt                       */
    const double* z;   /* n+m  (n+m+1 for AMG4POT)                            */
    const uint8_t* s;  /* m*n logical, AMG4POT only                           */
    const double* phi; /* m*n, AMG4POT only                                   */
} ipd_prob;

/* ---- library / context -------------------------------------------------- */
int ipd_version(void);
const char* ipd_last_error(void);
/* Number of HIP devices this process sees (hipGetDeviceCount); 0 without a usable GPU.  bench.py's
 * ranks map LOCAL_RANK onto it before a context exists.                                          */
int ipd_device_count(int32_t* count);
int ipd_ctx_create(int device, ipd_ctx** out);
void ipd_ctx_destroy(ipd_ctx* ctx);
int ipd_ctx_sync(ipd_ctx* ctx);
void ipd_csc_free(ipd_csc_out* mat);
void ipd_amg_opts_init(ipd_amg_opts* o); /* all fields "empty"               */
void ipd_pcg_opts_init(ipd_pcg_opts* o);

/* ---- rand stream (Hybrid_AMG.m:40,69; AMG/mis_set.m:31,35) --------------- */
/* mt19937ar seeded like MATLAB rng(seed); seed 5489 == MATLAB's default.    */
int ipd_rng_create(uint32_t seed, ipd_rng** out);
/* replays caller-supplied doubles (e.g. MATLAB's own rand output); running
 * out of numbers is IPD_E_ARG.                                              */
int ipd_rng_create_replay(const double* values, int64_t count, ipd_rng** out);
void ipd_rng_destroy(ipd_rng* rng);
int ipd_rng_rand(ipd_rng* rng, int64_t count, double* out); /* rand(count,1) */
int64_t ipd_rng_consumed(const ipd_rng* rng);

/* ---- L0/L1: matrix-free A operators and KKT assembly --------------------- */
/* y = Ax(x,p,q)            Ax.m:2    ; y has n+m entries                     */
int ipd_ax(ipd_ctx*, const double* x, const double* p, const double* q,
           int64_t m, int64_t n, double* y);
/* z = Aty(y,p,q)           Aty.m:2   ; z has m*n entries                     */
int ipd_aty(ipd_ctx*, const double* y, const double* p, const double* q,
            int64_t m, int64_t n, double* z);
/* H = ASAt(s,p,q)          ASAt.m:2                                          */
int ipd_asat(ipd_ctx*, const uint8_t* s, const double* p, const double* q,
             int64_t m, int64_t n, ipd_csc_out* H);
/* y = invAAt(x,p,q,sg1,sg2) invAAt.m:1 (nargin handling is the shim's job)   */
int ipd_inv_aat(ipd_ctx*, const double* x, const double* p, const double* q,
                int64_t m, int64_t n, double sg1, double sg2, double* y);
/* y = invHHt(v,p,q,sg,phi)  Class2/invHHt.m:1 ; v,y have n+m+1 entries       */
int ipd_inv_hht(ipd_ctx*, const double* v, const double* p, const double* q,
                int64_t m, int64_t n, double sg, const double* phi, double* y);

/* ---- L2: AMG setup pieces ------------------------------------------------ */
/* S = strength(A,which)    AMG/strength.m:1                                  */
int ipd_strength(ipd_ctx*, const ipd_csc* A, int which, ipd_csc_out* S);
/* [indC,indF] = cf_split(S) AMG/cf_split.m:1 (SubG is rebuilt by the shim)   */
int ipd_cf_split(ipd_ctx*, const ipd_csc* S, uint8_t* indC, uint8_t* indF);
/* [isC,isF,As] = mis_set(A,theta)  AMG/mis_set.m:1 ; As may be NULL          */
int ipd_mis_set(ipd_ctx*, const ipd_csc* A, double theta, ipd_rng* rng,
                uint8_t* isC, uint8_t* isF, ipd_csc_out* As);
/* [Ac,Pro] = transfer(A,amg_options) AMG/transfer.m:1 ; level = global J     */
int ipd_transfer(ipd_ctx*, const ipd_csc* A, const ipd_amg_opts* o, int level,
                 ipd_rng* rng, ipd_csc_out* Ac, ipd_csc_out* Pro, uint8_t* indC);

/* ---- L3: hierarchy, cycles, PCG ----------------------------------------- */
/* setup phase of Class_AMG (AMG/Class_AMG.m:41-85); A symmetric              */
int ipd_amg_setup(ipd_ctx*, const ipd_csc* A, const ipd_amg_opts* o, ipd_rng* rng,
                  ipd_amg** out);
void ipd_amg_destroy(ipd_amg* h);
int ipd_amg_num_levels(const ipd_amg* h);                       /* J          */
int ipd_amg_level_dims(const ipd_amg* h, int k, int64_t* rows, int64_t* nnz);
/* download Ack{k} (k>=1), Prok{k} (k>=2), C-mask of level k-1 -> k (k>=2)    */
int ipd_amg_get_A(const ipd_amg* h, int k, ipd_csc_out* A);
int ipd_amg_get_P(const ipd_amg* h, int k, ipd_csc_out* P);
int ipd_amg_get_cmask(const ipd_amg* h, int k, uint8_t* isC /* rows of level k-1 */);
/* solve phase of Class_AMG (AMG/Class_AMG.m:86-109).  rel_resk/rhok need
 * maxit+1 entries (may be NULL); *it = number of cycles.                     */
int ipd_amg_solve(ipd_amg* h, const double* b, const double* guess, double* x,
                  int32_t* it, double* rel_res, double* rel_resk, double* rhok);
/* e = MG_Vcycle(r,isnsp,k)   AMG/MG_Vcycle.m:2 ; k is 1-based               */
int ipd_amg_vcycle(ipd_amg* h, const double* r, int isnsp, int k, double* e);
/* e = MG_Wcycle(r,isnsp,k,e) AMG/MG_Wcycle.m:2 ; e_inout NULL-able input    */
int ipd_amg_wcycle(ipd_amg* h, const double* r, int isnsp, int k,
                   const double* e_in, double* e_out);
/* [x,it,rel_res,rel_resk,rhok] = Class_AMG(A,b,amg_options) Class_AMG.m:1   */
int ipd_class_amg(ipd_ctx*, const ipd_csc* A, const double* b, const double* guess,
                  const ipd_amg_opts* o, ipd_rng* rng, double* x, int32_t* it,
                  double* rel_res, double* rel_resk, double* rhok);
/* [x,it,rel_res,rel_resk,rhok] = twogrid_bigph(A,b,amg_options) AMG/twogrid_bigph.m:1:
 * the two-level special case (exactly one coarse level, coarse PCG with maxit 100); empty
 * fields take twogrid_bigph.m:11-15's defaults (retol 0, maxit 50, smoth 3, isnsp 0);
 * amg_options.fnode is required.  rel_resk/rhok need maxit+1 slots.                     */
int ipd_twogrid_bigph(ipd_ctx*, const ipd_csc* A, const double* b, const double* guess,
                      const ipd_amg_opts* o, double* x, int32_t* it, double* rel_res,
                      double* rel_resk, double* rhok);
/* [x,it,rel_res,rel_resk,rhok] = twogrid(A,b,amg_options) AMG/twogrid.m:1: bigph = 0 (default)
 * smooths with .5*D^-1 and coarsens with mis_set(A,1/4) (needs `rng`); bigph = 1 as above    */
int ipd_twogrid(ipd_ctx*, const ipd_csc* A, const double* b, const double* guess,
                const ipd_amg_opts* o, ipd_rng* rng, double* x, int32_t* it, double* rel_res,
                double* rel_resk, double* rhok);
/* [d,it,res,resk] = PCG(H,e,pcg_options)   PCG.m:1 ; resk needs maxit slots
 * or NULL                                                                    */
int ipd_pcg(ipd_ctx*, const ipd_csc* H, const double* e, const double* guess,
            const ipd_pcg_opts* o, double* d, int64_t* it, double* res, double* resk);

/* ---- L4: problem-level solvers ------------------------------------------ */
/* [blocks,sizes,p,r] = components(A)  components.m:1 ; 0-based outputs,
 * components numbered by smallest member (dmperm's order is unpinned).       */
int ipd_components(ipd_ctx*, const ipd_csc* A, int64_t* blocks, int64_t* sizes,
                   int64_t* p, int64_t* r, int64_t* ncomp);
/* Replays a recorded visiting order of the components (components.m:35-39 takes it from
 * dmperm, whose order is undocumented; Hybrid_AMG.m:55-57 visits in that order, which fixes
 * info(2) and the order in which rand is consumed): smallest_members[k] = the smallest member
 * (0-based) of the component to visit k-th, i.e. min(ps(rs(k):rs(k+1)-1))-1 of a MATLAB
 * recording.  One-shot: applies to the next components / Hybrid_AMG / AMG4POT (and variants)
 * call on this context.  NULL clears it.  Members stay ascending inside a component.        */
int ipd_ctx_set_component_order(ipd_ctx*, const int64_t* smallest_members, int64_t ncomp);
/* [zeta,itamg,resamg,info] = Hybrid_AMG(prob_data,amg_options) Hybrid_AMG.m:1 */
int ipd_hybrid_amg(ipd_ctx*, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                   double* zeta, int32_t* itamg, double* resamg, int64_t info[2]);
/* [zeta,...] = AMG4POT(prob_data,amg_options,'amg')  Class2/AMG4POT.m:1       */
/* [zeta,itpcg,respcg,info] = aug_PCG(prob_data,pcg_options) aug_PCG.m:1 and
 * PCG4POT(prob_data,pcg_options) Class2/PCG4POT.m:1 (inner_solver = 3): Jacobi-PCG on the
 * system augmented with one kernel vector per connected component                          */
int ipd_aug_pcg(ipd_ctx*, const ipd_prob* pd, const ipd_pcg_opts* o, double* zeta,
                int64_t* itpcg, double* respcg, int64_t info[2]);
int ipd_pcg4pot(ipd_ctx*, const ipd_prob* pd, const ipd_pcg_opts* o, double* zeta,
                int64_t* itpcg, double* respcg, int64_t info[2]);
/* [zeta,itamg,resamg,info] = Hybrid_twogrid(prob_data,amg_options) Hybrid_twogrid.m:1 and
 * AMG4POT(prob_data,amg_options,'twogrid') Class2/AMG4POT.m:48-51 (inner_solver = 5)       */
int ipd_hybrid_twogrid(ipd_ctx*, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                       double* zeta, int32_t* itamg, double* resamg, int64_t info[2]);
int ipd_amg4pot_twogrid(ipd_ctx*, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                        double* zeta, int32_t* itamg, double* resamg, int64_t info[2]);
int ipd_amg4pot(ipd_ctx*, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                double* zeta, int32_t* itamg, double* resamg, int64_t info[2]);

/* ---- device-resident path (inputs already in HBM) ------------------------ */
/* raw device memory on the context's GPU */
int ipd_dmalloc(ipd_ctx*, size_t bytes, void** dptr);
int ipd_dfree(ipd_ctx*, void* dptr);
int ipd_h2d(ipd_ctx*, void* dst_dev, const void* src_host, size_t bytes);
int ipd_d2h(ipd_ctx*, void* dst_host, const void* src_dev, size_t bytes);
/* device matrices */
int ipd_dmat_upload(ipd_ctx*, const ipd_csc* A, int symmetric, ipd_dmat** out);
int ipd_dmat_download(ipd_ctx*, const ipd_dmat* A, ipd_csc_out* out);
int ipd_dmat_dims(const ipd_dmat* A, int64_t* rows, int64_t* cols, int64_t* nnz);
void ipd_dmat_destroy(ipd_dmat* A);
/* C = A*B for sparse operands, every C(i,j) summed in ascending inner index with separately
 * rounded multiply and add -- the order of MATLAB's sparse mtimes as `Pro'*A*Pro`
 * (AMG/transfer.m:66) uses it; exact zeros are dropped from the result                      */
int ipd_dmat_multiply(ipd_ctx*, const ipd_dmat* A, const ipd_dmat* B, ipd_dmat** out);
/* y = A*x on device vectors (CSR row walk)                                   */
int ipd_spmv_dev(ipd_ctx*, const ipd_dmat* A, const double* x_dev, double* y_dev);
int ipd_ax_dev(ipd_ctx*, const double* x_dev, const double* p_dev, const double* q_dev,
               int64_t m, int64_t n, double* y_dev);
int ipd_aty_dev(ipd_ctx*, const double* y_dev, const double* p_dev, const double* q_dev,
                int64_t m, int64_t n, double* z_dev);
int ipd_asat_dev(ipd_ctx*, const uint8_t* s_dev, const double* p_dev, const double* q_dev,
                 int64_t m, int64_t n, ipd_dmat** H);
int ipd_amg_setup_dev(ipd_ctx*, const ipd_dmat* A, const ipd_amg_opts* o, ipd_rng* rng,
                      ipd_amg** out);
int ipd_amg_solve_dev(ipd_amg* h, const double* b_dev, const double* guess_dev,
                      double* x_dev, int32_t* it, double* rel_res, double* rel_resk,
                      double* rhok);
/* Hybrid_AMG with H0 already on the device (output of ipd_asat_dev)          */
int ipd_hybrid_amg_dev(ipd_ctx*, const ipd_dmat* H0, const double* t_dev, const double* p_dev,
                       const double* q_dev, int64_t m, int64_t n, double bk1, double tk,
                       const double* z_dev, const ipd_amg_opts* o, ipd_rng* rng,
                       double* zeta_dev, int32_t* itamg, double* resamg, int64_t info[2]);

/* X = A \ B, MATLAB's mldivide for a sparse symmetric positive definite A (CSC, both triangles)
 * and a dense column-major B (n x nrhs): the reference's cold-path direct solves
 * `zeta = Jk \ (-Fk_old)` (Class1/APD_SsN_Class1.m:148, Class2/APD_SsN_Class2.m:155) and
 * `W = -Aff \ Afc` (AMG/transfer.m:58).  MATLAB uses CHOLMOD (closed source); this is a blocked
 * dense Cholesky on the device, n <= 16384.  IPD_E_NUMERIC when A is not positive definite.     */
int ipd_spd_solve(ipd_ctx*, const ipd_csc* A, const double* B, int64_t nrhs, double* X);

/* ---- L5: APD / semismooth-Newton drivers and A-ADMM warm starts ----------- */
/* SURVEY.md section 8 rows f1/f2.  The reference's drivers are MATLAB *scripts*
 * (Class1/APD_SsN_Class1.m, Class2/APD_SsN_Class2.m): their boundary is the
 * workspace they load (`c,r,l,p,q,gama` resp. `C,r,l,p,q,mu,phi`) and the
 * variables they leave behind (`xk, lk, fxk, KKT_xk, KKT_lk, SsN_itnum,
 * PCG_itnum, SumAMG, ...`).  `ipd_apd` holds that workspace in HBM; every mn-
 * sized vector stays on the device for the whole run.                        */
typedef struct ipd_apd ipd_apd;

typedef struct ipd_apd_data {
    int32_t cls;          /* 1 = transport-like (Class 1), 2 = partial OT (Class 2)   */
    int64_t m, n;
    const double* c;      /* mn, column-major cost                                     */
    const double* r;      /* n                                                         */
    const double* l;      /* m                                                         */
    const double* p;      /* m                                                         */
    const double* q;      /* n                                                         */
    const double* gama;   /* class 1: mn upper bounds, or NULL -> gama_scalar          */
    double gama_scalar;   /* class 1: scalar bound, +Inf allowed (data1-*.mat)         */
    double mu;            /* class 2: transported mass                                 */
    const double* phi;    /* class 2: mn                                               */
} ipd_apd_data;

/* Script constants (APD_SsN_Class1.m:35-36, APD_SsN_Class2.m:35-36).          */
typedef struct ipd_apd_opts {
    int32_t maxit;        /* 100                                                       */
    double kkt_tol;       /* 1e-6                                                      */
    int32_t ssn_it;       /* 50                                                        */
    double ssn_tol1;      /* class 1: 1e-11, class 2: 1e-10                            */
    double nu, delta;     /* 0.2, 0.9                                                  */
    int32_t ll_max;       /* 500                                                       */
    int32_t prob;         /* class 1 only: `prob` of :19-23; 3 selects the merit of :186 */
    int32_t inner_solver; /* :66-71: 4 AMG (default), 5 two-grid, 3 aug_PCG / PCG4POT,
                             2 plain PCG on Jk, 1 direct solve `Jk \ (-Fk_old)`           */
    double pcg_retol;     /* pcg_options of :81,84: 1e-11                               */
    int64_t pcg_maxit;    /*                        1e4                                 */
} ipd_apd_opts;
void ipd_apd_opts_init(int32_t cls, ipd_apd_opts* o);

/* One record per Newton step: what the reference prints with format `cc`/`bb`
 * (APD_SsN_Class1.m:91-92,215-236) plus the size of the active set.            */
typedef struct ipd_ssn_rec {
    int32_t k, ssn_it, ll, itamg;
    int64_t E, info0, info1;
    double Fk_norm, resamg, bk1, tk;
} ipd_ssn_rec;

typedef struct ipd_apd_result {
    int32_t converged;    /* `CONV at it = k`                                          */
    int32_t k;            /* APD iterations done so far                                */
    double fval;          /* fxk(k+1) = c'*xk                                          */
    double kkt[4];        /* KKT_xk, KKT_lk, KKT_yk, KKT_zk of the last iterate        */
    double rr;            /* max relative KKT residual (:265)                          */
    int64_t sum_amg, total_amg, fail_amg, max_amg;   /* SumAMG ... MaxAMG (:94-97)     */
    int32_t restarts;
    int64_t nrec;         /* Newton-step records available through ipd_apd_records     */
} ipd_apd_result;

int ipd_apd_create(ipd_ctx*, const ipd_apd_data* d, ipd_apd** out);
void ipd_apd_destroy(ipd_apd* h);
/* lengths of uk (mn, or mn+n+m for class 2) and lk (n+m, or n+m+1)                 */
int ipd_apd_dims(const ipd_apd* h, int64_t* len_u, int64_t* len_lam);
/* [xk,lk] = warmup_class1(c,r,l,p,q,gama,res,maxit) (Class1/warmup_class1.m:2) and
 * [uk,lk] = warmup_class2(c,r,l,p,q,mu,phi,res,maxit) (Class2/warmup_class2.m:1):
 * the result becomes the driver state (xk = vk = xk0, lk = lk0, bk = 1; Class1 :59-60).
 * maxit < 0 means `inf`; the nargin/res rules of :3-20 are applied.              */
int ipd_apd_warmup(ipd_apd* h, double res, int64_t maxit);
/* Workspace access.  u = xk (mn) for class 1, uk = [xk;yk;zk] (mn+n+m) for class 2;
 * lam = lk (n+m resp. n+m+1).  NULL pointers are skipped.                        */
int ipd_apd_set_state(ipd_apd* h, const double* u, const double* v, const double* lam, double bk);
int ipd_apd_get_state(ipd_apd* h, double* u, double* v, double* lam, double* bk);
/* Runs up to `iters` further APD iterations (`for k = 1:maxit`, Class1 :101-275,
 * Class2 :95-285) with inner_solver = 4 (Hybrid_AMG resp. AMG4POT 'amg'); stops
 * early at `CONV` or at opts->maxit.                                              */
int ipd_apd_run(ipd_apd* h, const ipd_apd_opts* o, const ipd_amg_opts* amg, ipd_rng* rng,
                int32_t iters, ipd_apd_result* res);
/* Histories: which = 0 fxk, 1 KKT_xk, 2 KKT_lk, 3 KKT_yk, 4 KKT_zk (k+1 entries),
 * 5 SsN_itnum (k entries).  Returns the number of entries written in *count.    */
int ipd_apd_history(const ipd_apd* h, int32_t which, double* out, int64_t cap, int64_t* count);
int ipd_apd_records(const ipd_apd* h, ipd_ssn_rec* out, int64_t cap, int64_t* count);
/* Row f3 of SURVEY.md section 8 (hierarchy reuse across Newton steps).  The reference sets the
 * hierarchy up at every Newton step (Hybrid_AMG.m:40-41, AMG/Class_AMG.m:41-85); when a step's
 * system equals the previous step's (same active set, T, bk1, tk) its setups share the previous
 * hierarchies' rand-independent levels 1-2, bit for bit the same result (IPD_NO_STEP_DONOR=1
 * rebuilds everything).  *steps = Newton steps solved with AMG, *same_system = those whose system
 * repeated, *donated = setups that took levels from the previous step.  NULLs are skipped.    */
int ipd_apd_reuse_stats(const ipd_apd* h, int64_t* steps, int64_t* same_system, int64_t* donated);
/* Building blocks of one APD iteration, exposed for parity tests and callers that
 * keep the outer loop: `begin` fixes k and forms ak, bk1, tk, wk, wlk (:113-126);
 * `eval` evaluates zk = (wk - H'*lam)/tk, s, Fk = bk1*lam - H*prox(zk) - wlk and the
 * line-search merit cFk (:139-144,182-196) in ONE pass over wk.
 * vals = {bk1, tk, ak, |Fk|, cFk, E}.                                            */
int ipd_apd_begin(ipd_apd* h, int32_t k, double vals[3]);
int ipd_apd_eval(ipd_apd* h, const double* lam, uint8_t* s_out, double* t_out, double* Fk_out,
                 double vals[6]);
/* HIP-event timing of `reps` eval passes on the current workspace (bench.py).   */
int ipd_apd_bench_eval(ipd_apd* h, int32_t reps, double* total_ms, double* bytes_per_pass);

/* Matrix-free level-1 operator.  If level 1 of `h` is exactly Hybrid_AMG's
 * rescaled operator Ae = bk1*Q0^2 + (Q0*T*Q0 + Q0*H0*Q0)/tk for these p, q, tk (checked entry by
 * entry against A_1's CSR values), the level-1 Gauss-Seidel sweeps read one BIT per entry (the
 * active-set mask) instead of 12 bytes.  *attached = 0 leaves the CSR kernels in place.
 * ipd_hybrid_amg(_dev) and ipd_amg4pot attach it themselves.                               */
int ipd_amg_attach_mask_operator(ipd_amg* h, const double* p_dev, const double* q_dev,
                                 int64_t m, int64_t n, double tk, int32_t* attached);
/* The same check, for the level-resident solve kernel only: its level 1 <-> 2 transfers
 * W(j,i) = s_ij*beta_i*rho_j (AMG/transfer.m:19-25 applied to that operator) then read the bit mask
 * too, whatever the size; the sweeps of the launch path keep their CSR rows.  *attached = 0 when the
 * hierarchy does not run in that kernel or P does not have the form.  The solvers do this themselves. */
int ipd_amg_attach_mask_transfers(ipd_amg* h, const double* p_dev, const double* q_dev,
                                  int64_t m, int64_t n, double tk, int32_t* attached);
/* Level 2 of the level-resident kernel in polynomial form, composed over a whole visit: for a three-level
 * hierarchy with a one-row tail and V cycles (the metric's workload) the smoth pre-sweeps, the residual, the
 * restriction, the tail's prolongation and the smoth post-sweeps of a visit of level 2 (AMG/MG_Vcycle.m:14-41
 * with Class_AMG.m:84's Jacobi smoother) are ONE dense affine map of r_2 (and of the tail's PCG result), packed
 * here by smoth dense products on the f64 matrix cores; a V cycle is then 24 chip-wide hand-offs instead of 33.
 * Same linear operator as the sweeps, different association (rounding at the 1e-16 level per pass).  Packing
 * costs about 0.5 ms at 1024 rows: for many cycles on ONE hierarchy, not for a solve of such a system (one or
 * two cycles) -- the solvers never attach it.  *attached = 0: the hierarchy does not run in that kernel form. */
int ipd_amg_attach_level2_poly(ipd_amg* h, int32_t* attached);

/* ---- measurement hooks (bench.py) ---------------------------------------- */
/* Runs `cycles` iterations of the Class_AMG loop body (residual, one V/W
 * cycle, norm) on the fixed hierarchy without convergence exit, timed with HIP
 * events on the context's stream.  Returns total milliseconds and, per cycle,
 * the algorithmic byte count B_V of SURVEY.md section 8d.                     */
int ipd_amg_bench_cycles(ipd_amg* h, const double* b_dev, double* x_dev, int cycles,
                         double* total_ms, double* bytes_per_cycle);
/* Times `reps` smoother sweeps of level k (1 <= k < J) with HIP events: the
 * per-launch duration of the dominant kernel.  launches_per_sweep is 2 on the
 * bigraph Gauss-Seidel level (F half, C half) and 1 on Jacobi levels.          */
int ipd_amg_bench_sweeps(ipd_amg* h, int k, int reps, double* total_ms,
                         int* launches_per_sweep, double* bytes_per_sweep);
/* Times `reps` launches of the single-workgroup sub-cycle kernel (levels k_sub..J out of
 * LDS) on a zero right-hand side; stamps = 100 MHz clock at start / image loaded / cycle
 * begins / cycle done inside the last launch, then clocks spent in: wave-level sub-cycles,
 * block-level sweeps, residual+restriction, prolongation.  *k_sub = 0: no such kernel.  */
int ipd_amg_bench_subcycle(ipd_amg* h, int reps, double* total_ms, int32_t* k_sub,
                           int64_t stamps[8]);
/* SURVEY 8d byte model of the hierarchy: per-level S(A_k), S(P_k), ...       */
int ipd_amg_cycle_bytes(const ipd_amg* h, double* bytes_per_cycle);
/* How the solve phase of this hierarchy (AMG/Class_AMG.m:86-109) runs: *mode = 0 one launch
 * per phase, 1 the whole solve in one workgroup (small hierarchies), 2 the whole solve in ONE
 * launch of *grid co-resident workgroups that keep levels 1-2 in registers (csrc/ipd_resident.h:
 * three-level hierarchies of the dense regimes, and hierarchies of 4+ levels whose levels >= 3 fit
 * one workgroup's LDS -- then *grid counts that tail workgroup too); *timeouts = launches of mode 2
 * that gave up and were redone in mode 0.                                                      */
int ipd_amg_solve_mode(const ipd_amg* h, int32_t* mode, int32_t* grid, int32_t* timeouts);
/* Mode 2 only: *levels = levels kept by the resident workgroups (2, 3, or 4: the mask-form kernel's deep
 * mode with levels 3 and 4 in polynomial form), *tail_root = level at which the rest starts (levels + 1);
 * zeros otherwise.                                                                                  */
int ipd_amg_resident_levels(const ipd_amg* h, int32_t* levels, int32_t* tail_root);
/* Mode 2 only: name = the kernel instantiation a solve of this hierarchy launches, as a rocprofv3
 * kernel trace spells it ("k_resident<16,16,0>", "k_resident_big<32>"; "" in the other modes; cap =
 * size of the buffer); *handoffs / *cycles = chip-wide hand-offs (tagged-granule exchanges, visits of
 * the remote tail included) and Class_AMG loop bodies (AMG/Class_AMG.m:96-105) of the LAST launch;
 * *mask_transfers != 0: the level 1 <-> 2 transfers run from the active-set bit mask.  Any output
 * pointer but name may be NULL.                                                                    */
int ipd_amg_resident_kernel(const ipd_amg* h, char* name, int32_t cap, int64_t* handoffs,
                            int32_t* cycles, int32_t* mask_transfers);
/* forms[k], k < count (level k = 1..J; forms[0] = 0): how level k runs where it is held in a single
 * workgroup's LDS image (bit mask over the images packed for this hierarchy; 64 / 128: level 3-4 / level 2 of a
 * resident kernel in polynomial form, held by the resident workgroups): 1 thread-per-row sweeps,
 * 2 the same with dense rows in registers, 4 one-wave sweeps, 8 one-wave polynomial form (the nu sweeps,
 * residual and transfers of a visit as two dense passes), 16 block-wide polynomial form (49..144 rows,
 * operators streamed from L2); 0: in no image (launches or the resident workgroups' registers).      */
int ipd_amg_level_forms(const ipd_amg* h, int32_t* forms, int32_t count);
/* Test hook: the block-wide polynomial operator of level k (forms bit 16) as packed for the images:
 * column-major with *ld rows, columns [Mr (N8) | Me (N8) | Mc (Nc8)] and then the column W, N8 / Nc8 =
 * *n / *nc rounded up to 8; `out` needs ld * (2 N8 + Nc8 + 1) doubles (cap = its size).  Rows < n:
 * e' = Mr r + Me e (+ Mc e_c in the second pass) + W 1'r; rows n .. n + nc - 1: the restricted residual
 * (AMG/MG_Vcycle.m:14-41 as two dense maps).  IPD_E_ARG without such an operator. */
int ipd_amg_poly_operator(const ipd_amg* h, int32_t k, double* out, int64_t cap, int32_t* ld, int32_t* n,
                          int32_t* nc);
/* Mode 2 only: `cycles` loop bodies in one launch with in-kernel stamps of workgroup 0:
 * stamps[0] shader clocks spent waiting in hand-off sweeps, [1] shader clocks of the launch,
 * [2] hand-offs, [3] 100 MHz ticks of the launch, [4] clocks in the barrier ahead of the
 * publish (the wave's wait for the workgroup's slowest row), [5] in the store phase,
 * [6] in the closing barrier, [7] in the CSR row walks of the two transfers, [8] in the tail
 * level's solve; the rest is the row dot products.  total_ms: HIP events.          */
int ipd_amg_bench_resident(ipd_amg* h, const double* b_dev, double* x_dev, int cycles,
                           double* total_ms, int64_t stamps[10]);

/* Wall-clock attribution of the driver's phases when IPD_PROFILE=1 (each phase is then
 * bracketed by stream synchronisations).  Slots: 0 ASAt, 1 build Ae, 2 components,
 * 3 AMG setup, 4 AMG solve, 5 small-block direct solves, 6 evaluation passes,
 * 7 begin/end passes.  Returns the number of slots.                             */
int ipd_prof_read(double* seconds, int64_t* calls, int32_t reset);

/* ---- multi-GPU row-block sharding (RCCL over xGMI) ----------------------- */
#define IPD_COMM_ID_BYTES 128
int ipd_comm_get_unique_id(uint8_t id[IPD_COMM_ID_BYTES]);
int ipd_comm_init(ipd_ctx*, const uint8_t id[IPD_COMM_ID_BYTES], int rank, int nranks);
int ipd_comm_finalize(ipd_ctx*);
/* Rank and size as RCCL reports them for the context's communicator, and the all-gathers issued
 * on it so far: grouped launches and the vectors inside them; reset != 0 zeroes the counts.   */
int ipd_comm_stats(ipd_ctx*, int32_t* rank, int32_t* nranks, int64_t* allgather_calls,
                   int64_t* allgather_vectors, int32_t reset);
/* Row-block sharded variant of ipd_amg_bench_cycles: every rank holds the same
 * hierarchy, owns rows [rank*N_k/G, (rank+1)*N_k/G) of every level, and the
 * iterate is re-assembled with ncclAllGather after each smoother half-sweep.  */
int ipd_amg_bench_cycles_sharded(ipd_amg* h, const double* b_dev, double* x_dev, int cycles,
                                 double* total_ms, double* bytes_per_cycle);

#ifdef __cplusplus
}
#endif
#endif /* IPD_AMG_H */
