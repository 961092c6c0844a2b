// Device-resident AMG hierarchy: the reference's `global Ack Prok J smoth_it Rk`
// (AMG/Class_AMG.m:42-47) as one handle.
#pragma once

#include "ipd_internal.h"

// Filled-in amg_options (Class_AMG.m:26-34 defaults applied).
struct AmgOpts {
    double retol = 1e-12;
    int bigph = 0;
    int maxit = 50;
    double theta = 0.25;
    int smoth = 3;
    int cycle = 'v';
    int isnsp = 0;
    int inter = 1;
    long long fnode = -1;
    // AMG/twogrid_bigph.m as a special case of the hierarchy: exactly two levels whatever the
    // size (:25-38), coarse solve PCG(Ac, rrc, struct('retol',[],'maxit',1e2,'precd',2)) (:72-73)
    bool twogrid = false;
    long long pcg_maxit = 10000;   // MG_Vcycle.m:43 PCG(A,r): PCG.m:20 defaults
    // AMG4POT's two solves run side by side (Class2/AMG4POT.m:46-47): at most one of them finds the compute units
    // for a resident kernel of 200+ workgroups, the other runs as launches -- the planner then keeps the LDS
    // image the launches do best with (rooted at level 4) instead of the one the resident kernel does best with
    bool concurrent_pair = false;
};
AmgOpts amg_fill_twogrid_defaults(const ipd_amg_opts* o);
AmgOpts amg_fill_defaults(const ipd_amg_opts* o);

struct Level {
    int N = 0;          // rows of A
    Csr A;              // Ack{k}
    Csr P;              // Prok{k}   : N_{k-1} x N_k   (k >= 2; stored on the coarse level)
    Csr Pt;             // Prok{k}'  : N_k x N_{k-1}   (restriction, CSR)
    Csr T1;             // Prok{k}'*A_{k-1} : the first product of the Galerkin triple (transfer.m:66),
                        // kept for the fused residual + restriction  r_k = P'r - (P'A) e
    uint8_t* cmask = nullptr;  // isC of level k-1 (N_{k-1} bytes), k >= 2
    // smoother Rk{k}: level 1 with bigph -> forward Gauss-Seidel on the [F|C] blocks
    // (dinv = 1/diag, nf = fnode); otherwise weighted Jacobi (dinv = 0.5/diag, nf = 0)
    double* dinv = nullptr;
    int nf = 0;
    double* Axi = nullptr;  // A*1   (MG_Vcycle.m:15)
    double* xx = nullptr;   // device scalar 1'*A*1
    // work vectors of the cycle
    double* r = nullptr;    // right-hand side of this level
    double* e = nullptr;    // iterate
    double* e2 = nullptr;   // ping-pong partner of e
    double* w = nullptr;    // first-half result of a Gauss-Seidel sweep (without the +c shift)
    double* rr = nullptr;   // residual r - A e
    double* scal = nullptr; // small device scalars (sum r, partial dots ...)
    int lanes = 64;         // lanes per row used by the row kernels of this level
};

struct CycleState;  // ipd_cycle.hip

struct ipd_amg {
    ipd_ctx* ctx = nullptr;
    std::shared_ptr<CycleState> cyc;  // launch geometry + partial-sum buffers
    std::unique_ptr<Arena> arena;
    // Hierarchy whose rand-independent part this one shares (levels 1 and 2 of a bigraph
    // hierarchy: A_1, its smoother data and padded copy, P_2, P_2', A_2 -- the level-1 transfer
    // draws no random numbers, AMG/transfer.m:19-25): constant device arrays are pointed at, not
    // copied; work vectors and everything from level 2's C/F split down are this hierarchy's own.
    std::shared_ptr<ipd_amg> donor;
    AmgOpts opts;
    int J = 0;
    std::vector<Level> L;  // 1-based like the MATLAB cells; L[0] unused
    // solve-phase buffers (level-1 sized)
    double* x = nullptr;
    double* b = nullptr;
    double* res = nullptr;   // A x - b
    double* hist = nullptr;  // device copy of [res0, res, rnorm]
    // PCG work vectors for the coarsest level
    double* pcg_work = nullptr;
};

// ipd_setup.hip
void amg_strength_mask(ipd_ctx* ctx, const Csr& A, double theta, uint8_t* strong, int* degi,
                       int* rowcnt);
void amg_mis_set(ipd_ctx* ctx, const Csr& A, double theta, ipd_rng* rng, uint8_t* isC,
                 uint8_t* isF, uint8_t* strong_out /*nnz bytes or NULL*/);
void amg_transfer(ipd_ctx* ctx, Arena& dst, const Csr& A, const AmgOpts& o, int level,
                  ipd_rng* rng, Csr* Ac, Csr* P, Csr* Pt, uint8_t* cmask /*A.nr bytes*/,
                  Csr* T1out = nullptr /* Pt*A kept in dst when asked for */);
ipd_amg* amg_setup(ipd_ctx* ctx, const Csr& A, const AmgOpts& o, ipd_rng* rng,
                   const std::shared_ptr<ipd_amg>& donor = nullptr);
int amg_coarsest_threshold(int N);

// ipd_cycle.hip
void amg_prepare_levels(ipd_amg* h);  // dinv, Axi, xx, work vectors
void amg_cycle(ipd_amg* h, int k, int isnsp, bool wcycle, bool keep_e);
bool amg_attach_maskop(ipd_amg* h, const double* p_dev, const double* q_dev, int m, int n, double tk,
                       bool policy, bool transfers_only = false);
// a component's mask form is only of use beyond k_resident's 2048 rows (the mask-form kernel's deep mode)
static constexpr int RES_MASK_MIN_ROWS = 2048;
void amg_solve_dev(ipd_amg* h, const double* b_dev, const double* guess_dev, double* x_dev,
                   int32_t* it, double* rel_res, double* rel_resk, double* rhok);
void pcg_dev(ipd_ctx* ctx, const Csr& H, const double* e, const double* guess, double tol,
             long long maxit, int precd, double* d, long long* it, double* res, double* resk_host,
             long long nf = 0);

// ipd_hybrid.hip: problem-level solvers on device-resident data
struct HybridOut {
    int itamg = 0;
    double resamg = 0.0;
    long long num_comp = 0, it_num = 0;
};
// Row f3, reuse across Newton steps: the hierarchies of the previous Hybrid_AMG / AMG4POT call of
// a driver, kept alive so that a step whose system is the SAME (same active set, T, bk1, tk: the
// caller says so in `same`) shares their rand-independent levels 1-2 through amg_setup's donor
// mechanism.  Guesses and levels >= 3 still draw the stream's next numbers, so results and the
// count of consumed numbers are those of a full setup, bit for bit.
struct StepDonors {
    std::vector<std::shared_ptr<ipd_amg>> prev;   // in order of use (large components)
    bool same = false;                            // set by the caller before each call
    long long steps = 0, same_steps = 0, shared = 0;   // calls, calls with `same`, donated setups
};
// opts.twogrid selects Hybrid_twogrid.m (twogrid_bigph on every large component)
void hybrid_amg_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                    const double* q, int m, int n, double bk1, double tk, const double* z,
                    const AmgOpts& opts, ipd_rng* rng, double* zeta, HybridOut* out,
                    StepDonors* step = nullptr);
void amg4pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                 const double* q, int m, int n, double bk1, double tk, const double* z,
                 const uint8_t* s, const double* phi, const AmgOpts& opts, ipd_rng* rng,
                 double* zeta, HybridOut* out, StepDonors* step = nullptr);
// aug_PCG.m / Class2/PCG4POT.m (inner_solver = 3): PCG on the kernel-augmented system
void aug_pcg_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, double tol, long long maxit,
                 double* zeta, HybridOut* out);
void pcg4pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                 const double* phi, double tol, long long maxit, double* zeta, HybridOut* out);
// Class 2, inner_solver = 1 / 2: direct solve resp. PCG on the bordered Jacobian itself
// (APD_SsN_Class2.m:152-161)
void direct_pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                    int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                    const double* phi, double* zeta);
void pcg_pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                 const double* phi, double tol, long long maxit, double* zeta, HybridOut* out);
// ipd_dense.hip: dense Cholesky (MATLAB's `\` on the cold paths' SPD systems)
void dense_chol_factor(ipd_ctx* ctx, double* A, int n, int ld);              // lower triangle, in place
void dense_chol_solve(ipd_ctx* ctx, const double* L, int n, int ld, double* B, int nrhs, int ldb);
void spd_solve_dev(ipd_ctx* ctx, const Csr& A, const double* b, double* x);  // x = A \ b
// Jk = bk1*I + (T + H0)/tk (APD_SsN_Class1.m:151, inner_solver = 2)
void build_jk(ipd_ctx* ctx, Arena& dst, const Csr& H0, const double* tdiag, double bk1, double tk,
              Csr* J);
