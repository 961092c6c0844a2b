// Solve phase on the device: the V/W cycle, its smoothers, the coarsest-level
// Jacobi-PCG and the Class_AMG stationary iteration.
//   AMG/MG_Vcycle.m:9-45, AMG/MG_Wcycle.m:10-46, PCG.m:68-87, AMG/Class_AMG.m:86-109.
//
// Kernel design (all HBM/L2-bandwidth or latency bound; no MFMA -- sparse fp64):
//  * one CSR row walk per smoother sweep.  The reference applies an explicit
//    smoother matrix (g = r - A e; e += R g, Rk{1} = forward Gauss-Seidel on the
//    bipartite blocks, Rk{k>1} = 0.5 D^-1).  Algebraically R*(r - A e) for the
//    block-triangular Rk{1} is a forward (F then C) Gauss-Seidel half-sweep pair:
//    the second half reads the first half's result, so one pass over A per sweep
//    (S(A_1) bytes, the minimum) replaces SpMV(A)+SpMV(R).  Rk{1}' is the backward
//    (C then F) pair.
//  * the kernel-augmented smoother (isnsp, MG_Vcycle.m:15-21) needs xig = 1'(r-Ae)
//    BEFORE the sweep; we use 1'(r - A e) = 1'r - (A1)'e (A symmetric), with the
//    two sums carried as per-block partials written by whichever kernel produced
//    r and e, so no extra pass or launch is needed and the result is
//    run-to-run deterministic (no float atomics).
//  * rows are split over L lanes (4..1024) chosen per level from nnz/row so that
//    short rows do not idle a wave and long rows still fill the chip.
// Solve-phase results differ from the oracle only by summation order: tests
// compare residual histories to 1e-10.
#include "ipd_amg_internal.h"

#include <cmath>

static constexpr int BT = 1024;  // threads per block of every phase kernel

// ---------------------------------------------------------------------------
// device-side level descriptor
// ---------------------------------------------------------------------------
struct LevelDev {
    int N, nf, L, G;  // rows, F-block size (0 = Jacobi), lanes/row, blocks per launch
    const int* rp;
    const int* ci;
    const double* va;
    const double* dinv;
    const double* Axi;
    const double* xx;
    double* r;
    double* rr;
    double* rsum;  // partial sums of r, nrsum entries valid
};

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// sum over the whole 1024-thread block, result in every thread
__device__ __forceinline__ double block_sum(double v, double* red /*16 doubles of LDS*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < BT / 64; ++k) s += red[k];
    return s;
}

// sum within aligned groups of L threads (L = 4..1024, power of two); every
// thread of the block must call it.  Result valid in the group's first thread.
__device__ __forceinline__ double group_sum(double v, int L, double* red /*16 doubles*/) {
    if (L <= 64) {
        for (int d = L >> 1; d > 0; d >>= 1) v += __shfl_xor(v, d);
        return v;
    }
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    const int wpg = L >> 6;  // waves per group
    const int g0 = (threadIdx.x / L) * wpg;
    double s = 0.0;
    for (int k = 0; k < wpg; ++k) s += red[g0 + k];
    return s;
}

// ---------------------------------------------------------------------------
// phases (device functions so that a single-workgroup fused kernel can chain them)
// ---------------------------------------------------------------------------
struct SmoothArgs {
    LevelDev lv;
    int row0, row1;  // rows updated by this launch
    int u0, u1;      // columns in [u0,u1) read `win` (first-half result), others `eold`
    const double* eold;
    const double* win;
    double* enew;
    double* wout;
    int isnsp;
    const double* part_old;  // partial sums of Axi.*eold
    int npart_old;           // 0 -> eold == 0
    int nrsum;
    double* part_out;        // G entries, partial sums of Axi.*enew over this launch's rows
};

__device__ __forceinline__ void phase_smooth(const SmoothArgs& a, int b, double* red) {
    const LevelDev& lv = a.lv;
    const int tid = threadIdx.x;
    double c = 0.0;
    if (a.isnsp) {  // c = xig/xx with xig = 1'r - (A1)'e          MG_Vcycle.m:18-19
        double s = 0.0;
        for (int k = tid; k < a.nrsum; k += BT) s += lv.rsum[k];
        for (int k = tid; k < a.npart_old; k += BT) s -= a.part_old[k];
        c = block_sum(s, red) / lv.xx[0];
    }
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const int nrows = a.row1 - a.row0;
    const int niter = (nrows + lv.G * gpb - 1) / (lv.G * gpb);
    double pacc = 0.0;
    for (int it = 0; it < niter; ++it) {
        const int row = a.row0 + (it * lv.G + b) * gpb + g;
        const bool valid = row < a.row1;
        double s = 0.0;
        if (valid) {
            const int e0 = lv.rp[row], e1 = lv.rp[row + 1];
            for (int t = e0 + gl; t < e1; t += L) {
                const int j = lv.ci[t];
                const double y = (j >= a.u0 && j < a.u1) ? a.win[j] : a.eold[j];
                s += lv.va[t] * y;
            }
        }
        s = group_sum(s, L, red);
        if (valid && gl == 0) {
            const double eo = a.eold[row];
            double g_i = lv.r[row] - s;
            double axi = 0.0;
            if (a.isnsp) {
                axi = lv.Axi[row];
                g_i -= axi * c;
            }
            const double wv = eo + lv.dinv[row] * g_i;  // e + R*(g - Axi*c)
            const double en = wv + c;                    //   ... + xi*c
            if (a.wout) a.wout[row] = wv;
            a.enew[row] = en;
            pacc += axi * en;
        }
    }
    if (a.isnsp) {
        const double tot = block_sum(pacc, red);
        if (tid == 0) a.part_out[b] = tot;
    }
}

// rr = r - A e                                                   MG_Vcycle.m:27
__device__ __forceinline__ void phase_resid(const LevelDev& lv, const double* e, int b,
                                            double* red) {
    const int tid = threadIdx.x;
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const int niter = (lv.N + lv.G * gpb - 1) / (lv.G * gpb);
    for (int it = 0; it < niter; ++it) {
        const int row = (it * lv.G + b) * gpb + g;
        const bool valid = row < lv.N;
        double s = 0.0;
        if (valid)
            for (int t = lv.rp[row] + gl; t < lv.rp[row + 1]; t += L) s += lv.va[t] * e[lv.ci[t]];
        s = group_sum(s, L, red);
        if (valid && gl == 0) lv.rr[row] = lv.r[row] - s;
    }
}

// generic y = M*x row walk used by restriction (M = P') and prolongation (M = P)
struct XferArgs {
    int nrows, L, G;
    const int* rp;
    const int* ci;
    const double* va;
    const double* x;
    double* y;        // restriction: y = M x ; prolongation: y += M x
    double* zero_out; // restriction: coarse iterate to clear (may be NULL)
    int add;          // 1 = prolongation
    const double* axi;  // partial dot weights (NULL -> weight 1: plain sum of y)
    double* part_out;   // G partial sums of weight.*y (NULL -> skip)
};

__device__ __forceinline__ void phase_xfer(const XferArgs& a, int b, double* red) {
    const int tid = threadIdx.x;
    const int L = a.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const int niter = (a.nrows + a.G * gpb - 1) / (a.G * gpb);
    double pacc = 0.0;
    for (int it = 0; it < niter; ++it) {
        const int row = (it * a.G + b) * gpb + g;
        const bool valid = row < a.nrows;
        double s = 0.0;
        if (valid)
            for (int t = a.rp[row] + gl; t < a.rp[row + 1]; t += L) s += a.va[t] * a.x[a.ci[t]];
        s = group_sum(s, L, red);
        if (valid && gl == 0) {
            const double v = a.add ? a.y[row] + s : s;
            a.y[row] = v;
            if (a.zero_out) a.zero_out[row] = 0.0;
            pacc += (a.axi ? a.axi[row] : 1.0) * v;
        }
    }
    if (a.part_out) {
        const double tot = block_sum(pacc, red);
        if (tid == 0) a.part_out[b] = tot;
    }
}

// top level of Class_AMG: x_new = x + e ; r = b - A x_new ; partials of r'r and 1'r
struct TopArgs {
    LevelDev lv;
    const double* b;
    const double* x;
    const double* e;  // NULL -> x_new = x
    double* xnew;
    double* nrm_part;  // G partial sums of r_i^2
};

__device__ __forceinline__ void phase_top(const TopArgs& a, int b, double* red) {
    const LevelDev& lv = a.lv;
    const int tid = threadIdx.x;
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const int niter = (lv.N + lv.G * gpb - 1) / (lv.G * gpb);
    double p2 = 0.0, p1 = 0.0;
    for (int it = 0; it < niter; ++it) {
        const int row = (it * lv.G + b) * gpb + g;
        const bool valid = row < lv.N;
        double s = 0.0;
        if (valid) {
            for (int t = lv.rp[row] + gl; t < lv.rp[row + 1]; t += L) {
                const int j = lv.ci[t];
                const double xj = a.e ? a.x[j] + a.e[j] : a.x[j];
                s += lv.va[t] * xj;
            }
        }
        s = group_sum(s, L, red);
        if (valid && gl == 0) {
            const double ri = a.b[row] - s;
            lv.r[row] = ri;
            a.xnew[row] = a.e ? a.x[row] + a.e[row] : a.x[row];
            p2 += ri * ri;
            p1 += ri;
        }
    }
    const double t2 = block_sum(p2, red);
    const double t1 = block_sum(p1, red);
    if (tid == 0) {
        a.nrm_part[b] = t2;
        lv.rsum[b] = t1;
    }
}

__global__ __launch_bounds__(BT) void k_smooth(SmoothArgs a) {
    __shared__ double red[16];
    phase_smooth(a, blockIdx.x, red);
}
__global__ __launch_bounds__(BT) void k_resid(LevelDev lv, const double* e) {
    __shared__ double red[16];
    phase_resid(lv, e, blockIdx.x, red);
}
__global__ __launch_bounds__(BT) void k_xfer(XferArgs a) {
    __shared__ double red[16];
    phase_xfer(a, blockIdx.x, red);
}
__global__ __launch_bounds__(BT) void k_top(TopArgs a) {
    __shared__ double red[16];
    phase_top(a, blockIdx.x, red);
}

// hist[0] = res0 (set on the first call), hist[1] = res, hist[2] = previous res,
// hist[3] = rel_res, hist[4] = rhok                        Class_AMG.m:89,103-105
__global__ __launch_bounds__(BT) void k_conv(const double* nrm_part, int npart, double* hist,
                                             int first) {
    __shared__ double red[16];
    double s = 0.0;
    for (int k = threadIdx.x; k < npart; k += BT) s += nrm_part[k];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) {
        const double res = sqrt(tot);
        if (first) {
            hist[0] = res;
            hist[1] = res;
            hist[2] = res;
            hist[3] = 1.0;
            hist[4] = 0.0;
        } else {
            const double prev = hist[1];
            hist[2] = prev;
            hist[1] = res;
            hist[3] = res / hist[0];
            hist[4] = res / prev;
        }
    }
}

// sum of a vector into one slot (entry point of ipd_amg_vcycle / wcycle)
__global__ __launch_bounds__(BT) void k_vec_sum(const double* v, int n, double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += BT) s += v[k];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

__global__ __launch_bounds__(BT) void k_dot_sum(const double* a, const double* b, int n,
                                                double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += BT) s += a[k] * b[k];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

// ---------------------------------------------------------------------------
// PCG (Shewchuk B3) in one workgroup                              PCG.m:68-87
// ---------------------------------------------------------------------------
// The hot use is the coarsest level (N <= 1+fix(M^(1/3)), i.e. <= 17 rows): the
// whole solve is latency, so it runs inside one workgroup with no host round
// trips.  Vectors live in global scratch (L1/L2 resident).  precd: 1 none, 2 Jacobi.
struct PcgArgs {
    int N, L;
    const int* rp;
    const int* ci;
    const double* va;
    const double* rhs;
    const double* guess;  // NULL -> zeros
    double* d;            // solution
    double* work;         // 4*N doubles: r, p, q, diag
    double tol;
    long long maxit;
    int precd;
    double* out;          // out[0] = it, out[1] = res ; then resk[0..min(it,nresk))
    long long nresk;
};

__device__ __forceinline__ void pcg_block(const PcgArgs& a, double* red) {
    const int tid = threadIdx.x;
    const int N = a.N, L = a.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    double* r = a.work;
    double* p = a.work + N;
    double* q = a.work + 2 * (size_t)N;
    double* dg = a.work + 3 * (size_t)N;
    const int niter = (N + gpb - 1) / gpb;
    // r = e - H*d0 ; diag ; p = M^-1 r ; delta_new = r'p                     :68-70
    double acc = 0.0;
    for (int it = 0; it < niter; ++it) {
        const int row = it * gpb + g;
        const bool valid = row < N;
        double s = 0.0, dd = 0.0;
        if (valid)
            for (int t = a.rp[row] + gl; t < a.rp[row + 1]; t += L) {
                const int j = a.ci[t];
                if (a.guess) s += a.va[t] * a.guess[j];
                if (j == row) dd = a.va[t];
            }
        s = group_sum(s, L, red);
        dd = group_sum(dd, L, red);
        if (valid && gl == 0) {
            const double ri = a.rhs[row] - s;
            const double pi = a.precd == 2 ? ri / dd : ri;
            r[row] = ri;
            dg[row] = dd;
            p[row] = pi;
            a.d[row] = a.guess ? a.guess[row] : 0.0;
            acc += ri * pi;
        }
    }
    double delta_new = block_sum(acc, red);
    const double delta_0 = delta_new;
    const double thresh = a.tol * a.tol * delta_0;
    long long it_count = 0;
    while (it_count < a.maxit && delta_new > thresh) {                          // :76
        const double delta_old = delta_new;
        __syncthreads();
        acc = 0.0;
        for (int it = 0; it < niter; ++it) {  // q = H*p ; q'p
            const int row = it * gpb + g;
            const bool valid = row < N;
            double s = 0.0;
            if (valid)
                for (int t = a.rp[row] + gl; t < a.rp[row + 1]; t += L) s += a.va[t] * p[a.ci[t]];
            s = group_sum(s, L, red);
            if (valid && gl == 0) {
                q[row] = s;
                acc += s * p[row];
            }
        }
        const double qp = block_sum(acc, red);
        const double alpha = delta_old / qp;                                    // :78
        acc = 0.0;
        for (int row = tid; row < N; row += BT) {
            a.d[row] += alpha * p[row];
            const double ri = r[row] - alpha * q[row];                          // :79
            r[row] = ri;
            const double wi = a.precd == 2 ? ri / dg[row] : ri;                 // :80
            q[row] = wi;  // q is free again: holds w
            acc += ri * wi;
        }
        delta_new = block_sum(acc, red);                                        // :81
        const double beta = delta_new / delta_old;                              // :82
        for (int row = tid; row < N; row += BT) p[row] = q[row] + beta * p[row];  // :83
        ++it_count;
        if (tid == 0 && a.out && it_count <= a.nresk)
            a.out[1 + it_count] = sqrt(fabs(delta_new / delta_0));              // :85
    }
    if (tid == 0 && a.out) {
        a.out[0] = (double)it_count;
        a.out[1] = sqrt(fabs(delta_new / delta_0));                             // :87 (0/0 -> NaN)
    }
    __syncthreads();
}

__global__ __launch_bounds__(BT) void k_pcg(PcgArgs a) {
    __shared__ double red[16];
    pcg_block(a, red);
}

static int pick_lanes(long long nnz, int nrows, int blocks_target) {
    // lanes per row: 4..8 entries per lane, widened while the launch
    // would leave most of the chip idle
    if (nrows <= 0) return 4;
    const double avg = (double)nnz / (double)nrows;
    int L = 4;
    while (L < 1024 && (double)L * 8.0 <= avg) L <<= 1;
    while (L < 1024 && (long long)nrows * L < (long long)blocks_target * BT / 2 && (double)L < avg)
        L <<= 1;
    return L;
}

void pcg_dev(ipd_ctx* ctx, const Csr& H, const double* e, const double* guess, double tol,
             long long maxit, int precd, double* d, long long* it, double* res,
             double* resk_host) {
    IPD_REQUIRE(H.nr == H.nc, IPD_E_ARG, "PCG: H must be square");
    IPD_REQUIRE(precd == 1 || precd == 2, IPD_E_UNSUPPORTED,
                "PCG: only precd 1 (none) and 2 (Jacobi) run on the device; 3,4,5 are cold paths");
    Arena& tmp = *ctx->scratch;
    const long long nresk = resk_host ? std::min<long long>(maxit, 1 << 20) : 0;
    PcgArgs a;
    a.N = H.nr;
    a.L = std::min(pick_lanes(H.nnz, H.nr, 1), 64);
    a.rp = H.rp;
    a.ci = H.ci;
    a.va = H.va;
    a.rhs = e;
    a.guess = guess;
    a.d = d;
    a.work = tmp.alloc<double>(4 * (size_t)H.nr);
    a.tol = tol;
    a.maxit = maxit;
    a.precd = precd;
    a.out = tmp.alloc<double>((size_t)(2 + nresk));
    a.nresk = nresk;
    hipLaunchKernelGGL(k_pcg, dim3(1), dim3(BT), 0, ctx->stream, a);
    IPD_KERNEL_CHECK();
    if (it || res || resk_host) {
        double head[2];
        ctx->fetch(a.out, head, 2);
        if (it) *it = (long long)head[0];
        if (res) *res = head[1];
        if (resk_host && head[0] > 0)
            ctx->fetch(a.out + 2, resk_host, (size_t)std::min<long long>((long long)head[0], nresk));
    }
}

// ---------------------------------------------------------------------------
// host-side orchestration
// ---------------------------------------------------------------------------
struct LevelRun {  // per-level run state kept next to Level
    LevelDev dev;
    double* part[2];  // Axi'.e partials: [parity][2*G]
    int parity = 0;
    int npart = 0;    // valid partials describing the current e (0 -> e == 0)
    int nrsum = 0;
    XferArgs restrict_args;  // r_{k+1} = P' rr_k   (stored on level k)
    XferArgs prolong_args;   // e_k += P e_{k+1}
    PcgArgs pcg;             // coarsest only
};

struct CycleState {
    std::vector<LevelRun> run;  // 1-based
    double* nrm_part = nullptr;
    double* hist = nullptr;
    double* x2 = nullptr;
};

static CycleState* state_of(ipd_amg* h) { return h->cyc.get(); }

__global__ void k_level_prepare(int N, int nf, const int* __restrict__ rp,
                                const int* __restrict__ ci, const double* __restrict__ va,
                                double* __restrict__ dinv, double* __restrict__ Axi) {
    // one wave per row: diagonal -> Rk, row sum -> A*1
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < N; r += nwaves) {
        double s = 0.0, dg = 0.0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            s += va[t];
            if (ci[t] == r) dg = va[t];
        }
        s = wave_sum(s);
        dg = wave_sum(dg);
        if (lane == 0) {
            Axi[r] = s;
            // Class_AMG.m:56-59 (1./diag) for the bigraph GS, :72/:84 (0.5*(1./diag)) otherwise
            dinv[r] = nf > 0 ? 1.0 / dg : 0.5 * (1.0 / dg);
        }
    }
}

void amg_prepare_levels(ipd_amg* h) {
    ipd_ctx* ctx = h->ctx;
    Arena& ar = *h->arena;
    std::unique_ptr<CycleState> st(new CycleState());
    st->run.resize((size_t)h->J + 1);
    const int cu = ctx->num_cu;
    for (int k = 1; k <= h->J; ++k) {
        Level& lv = h->L[k];
        const int N = lv.A.nr;
        lv.N = N;
        lv.nf = (k == 1 && h->opts.bigph) ? (int)h->opts.fnode : 0;
        IPD_REQUIRE(lv.nf < N, IPD_E_ARG, "fnode must be smaller than the matrix size");
        lv.dinv = ar.alloc<double>((size_t)N);
        lv.Axi = ar.alloc<double>((size_t)N);
        lv.xx = ar.alloc<double>(1);
        lv.r = ar.alloc<double>((size_t)N);
        lv.e = ar.alloc<double>((size_t)N);
        lv.e2 = ar.alloc<double>((size_t)N);
        lv.w = ar.alloc<double>((size_t)N);
        lv.rr = ar.alloc<double>((size_t)N);
        hipLaunchKernelGGL(k_level_prepare, dim3(std::max(1, std::min(cdiv(N, 4), 4096))), dim3(256),
                           0, ctx->stream, N, lv.nf, lv.A.rp, lv.A.ci, lv.A.va, lv.dinv, lv.Axi);
        IPD_KERNEL_CHECK();
        hipLaunchKernelGGL(k_vec_sum, dim3(1), dim3(BT), 0, ctx->stream, lv.Axi, N, lv.xx);
        IPD_KERNEL_CHECK();
        // launch geometry: for a GS level the work per launch is half the matrix
        const int rows_per_launch = lv.nf > 0 ? std::max(1, N / 2) : N;
        const long long nnz_per_launch = lv.nf > 0 ? std::max(1, lv.A.nnz / 2) : lv.A.nnz;
        lv.lanes = pick_lanes(nnz_per_launch, rows_per_launch, cu);
        const int G = (int)std::max<long long>(
            1, std::min<long long>(cu, ((long long)rows_per_launch * lv.lanes + BT - 1) / BT));
        LevelRun& rn = st->run[(size_t)k];
        rn.dev.N = N;
        rn.dev.nf = lv.nf;
        rn.dev.L = lv.lanes;
        rn.dev.G = G;
        rn.dev.rp = lv.A.rp;
        rn.dev.ci = lv.A.ci;
        rn.dev.va = lv.A.va;
        rn.dev.dinv = lv.dinv;
        rn.dev.Axi = lv.Axi;
        rn.dev.xx = lv.xx;
        rn.dev.r = lv.r;
        rn.dev.rr = lv.rr;
        rn.dev.rsum = ar.alloc<double>((size_t)cu + 1);
        rn.part[0] = ar.alloc<double>(2 * (size_t)cu + 2);
        rn.part[1] = ar.alloc<double>(2 * (size_t)cu + 2);
    }
    for (int k = 1; k < h->J; ++k) {
        Level& fine = h->L[k];
        Level& coarse = h->L[k + 1];
        LevelRun& rn = st->run[(size_t)k];
        // restriction: rows of P' (coarse rows)
        XferArgs ra;
        ra.nrows = coarse.Pt.nr;
        ra.L = pick_lanes(coarse.Pt.nnz, coarse.Pt.nr, cu);
        ra.G = (int)std::max<long long>(
            1, std::min<long long>(cu, ((long long)ra.nrows * ra.L + BT - 1) / BT));
        ra.rp = coarse.Pt.rp;
        ra.ci = coarse.Pt.ci;
        ra.va = coarse.Pt.va;
        ra.x = fine.rr;
        ra.y = coarse.r;
        ra.zero_out = coarse.e;
        ra.add = 0;
        ra.axi = nullptr;
        ra.part_out = st->run[(size_t)k + 1].dev.rsum;
        rn.restrict_args = ra;
        XferArgs pa;
        pa.nrows = coarse.P.nr;
        pa.L = pick_lanes(coarse.P.nnz, coarse.P.nr, cu);
        pa.G = (int)std::max<long long>(
            1, std::min<long long>(cu, ((long long)pa.nrows * pa.L + BT - 1) / BT));
        pa.rp = coarse.P.rp;
        pa.ci = coarse.P.ci;
        pa.va = coarse.P.va;
        pa.x = coarse.e;
        pa.y = fine.e;
        pa.zero_out = nullptr;
        pa.add = 1;
        pa.axi = fine.Axi;
        pa.part_out = nullptr;  // chosen per call (parity)
        rn.prolong_args = pa;
    }
    {   // coarsest level: PCG(A,r) with the 2-argument defaults (PCG.m:18-23)
        Level& cl = h->L[h->J];
        PcgArgs a;
        a.N = cl.A.nr;
        a.L = std::min(pick_lanes(cl.A.nnz, cl.A.nr, 1), 64);
        a.rp = cl.A.rp;
        a.ci = cl.A.ci;
        a.va = cl.A.va;
        a.rhs = cl.r;
        a.guess = nullptr;
        a.d = cl.e;
        a.work = ar.alloc<double>(4 * (size_t)cl.A.nr);
        a.tol = 1e-11;
        a.maxit = 10000;
        a.precd = 2;
        a.out = nullptr;
        a.nresk = 0;
        st->run[(size_t)h->J].pcg = a;
    }
    st->nrm_part = ar.alloc<double>((size_t)cu + 1);
    st->hist = ar.alloc<double>(8);
    st->x2 = ar.alloc<double>((size_t)h->L[1].A.nr);
    h->x = ar.alloc<double>((size_t)h->L[1].A.nr);
    h->b = ar.alloc<double>((size_t)h->L[1].A.nr);
    h->cyc = std::shared_ptr<CycleState>(st.release());
}

// one smoother sweep on level k: Jacobi = one launch, bigraph GS = two half launches
static void launch_sweep(ipd_amg* h, CycleState* st, int k, int isnsp, bool post) {
    ipd_ctx* ctx = h->ctx;
    Level& lv = h->L[k];
    LevelRun& rn = st->run[(size_t)k];
    const int G = rn.dev.G;
    SmoothArgs a;
    a.lv = rn.dev;
    a.eold = lv.e;
    a.enew = lv.e2;
    a.win = lv.w;
    a.wout = lv.w;
    a.isnsp = isnsp;
    a.part_old = rn.part[rn.parity];
    a.npart_old = rn.npart;
    a.nrsum = rn.nrsum;
    double* pnew = rn.part[rn.parity ^ 1];
    if (lv.nf == 0) {
        a.row0 = 0;
        a.row1 = lv.N;
        a.u0 = a.u1 = 0;
        a.wout = nullptr;
        a.part_out = pnew;
        hipLaunchKernelGGL(k_smooth, dim3(G), dim3(BT), 0, ctx->stream, a);
        IPD_KERNEL_CHECK();
        rn.npart = isnsp ? G : 0;
    } else {
        // pre: F rows then C rows (Rk{1});  post: C rows then F rows (Rk{1}')
        const int f0 = post ? lv.nf : 0, f1 = post ? lv.N : lv.nf;  // first half rows
        const int s0 = post ? 0 : lv.nf, s1 = post ? lv.nf : lv.N;  // second half rows
        a.row0 = f0;
        a.row1 = f1;
        a.u0 = a.u1 = 0;
        a.part_out = pnew;
        hipLaunchKernelGGL(k_smooth, dim3(G), dim3(BT), 0, ctx->stream, a);
        IPD_KERNEL_CHECK();
        a.row0 = s0;
        a.row1 = s1;
        a.u0 = f0;
        a.u1 = f1;
        a.wout = nullptr;
        a.part_out = pnew + G;
        hipLaunchKernelGGL(k_smooth, dim3(G), dim3(BT), 0, ctx->stream, a);
        IPD_KERNEL_CHECK();
        rn.npart = isnsp ? 2 * G : 0;
    }
    rn.parity ^= 1;
    std::swap(lv.e, lv.e2);
}

// Solves A_k e = r_k approximately; r in L[k].r, result in L[k].e.
// keep_e: start from the current L[k].e (second leg of a W cycle).
void amg_cycle(ipd_amg* h, int k, int isnsp, bool wcycle, bool keep_e) {
    ipd_ctx* ctx = h->ctx;
    CycleState* st = state_of(h);
    IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
    Level& lv = h->L[k];
    LevelRun& rn = st->run[(size_t)k];
    if (k == h->J) {                                   // MG_Vcycle.m:43 / MG_Wcycle.m:44
        PcgArgs a = rn.pcg;
        a.rhs = lv.r;
        a.d = lv.e;
        hipLaunchKernelGGL(k_pcg, dim3(1), dim3(BT), 0, ctx->stream, a);
        IPD_KERNEL_CHECK();
        return;
    }
    if (!keep_e) rn.npart = 0;  // e == 0 (cleared by the parent's restriction or the caller)
    const int nu = h->opts.smoth;
    for (int s = 0; s < nu; ++s) launch_sweep(h, st, k, isnsp, false);          // :14-25
    hipLaunchKernelGGL(k_resid, dim3(rn.dev.G), dim3(BT), 0, ctx->stream, rn.dev,
                       (const double*)lv.e);                                     // :27
    IPD_KERNEL_CHECK();
    {
        XferArgs ra = rn.restrict_args;
        ra.zero_out = h->L[k + 1].e;
        ra.part_out = isnsp ? st->run[(size_t)k + 1].dev.rsum : nullptr;
        hipLaunchKernelGGL(k_xfer, dim3(ra.G), dim3(BT), 0, ctx->stream, ra);
        IPD_KERNEL_CHECK();
        st->run[(size_t)k + 1].nrsum = isnsp ? ra.G : 0;
    }
    amg_cycle(h, k + 1, isnsp, wcycle, false);                                   // :29
    // MG_Wcycle.m:30 -- the second correction; on the coarsest level it repeats the
    // identical zero-guess PCG solve, so it is skipped there (same bits).
    if (wcycle && k + 1 < h->J) amg_cycle(h, k + 1, isnsp, wcycle, true);
    {
        XferArgs pa = rn.prolong_args;                                           // :31
        pa.x = h->L[k + 1].e;
        pa.y = lv.e;
        double* pnew = rn.part[rn.parity ^ 1];
        pa.part_out = isnsp ? pnew : nullptr;
        hipLaunchKernelGGL(k_xfer, dim3(pa.G), dim3(BT), 0, ctx->stream, pa);
        IPD_KERNEL_CHECK();
        rn.parity ^= 1;
        rn.npart = isnsp ? pa.G : 0;
    }
    for (int s = 0; s < nu; ++s) launch_sweep(h, st, k, isnsp, true);           // :33-41
}

static void launch_top(ipd_amg* h, CycleState* st, const double* b, const double* x,
                       const double* e, double* xnew, bool first) {
    ipd_ctx* ctx = h->ctx;
    LevelRun& rn = st->run[1];
    TopArgs a;
    a.lv = rn.dev;
    a.b = b;
    a.x = x;
    a.e = e;
    a.xnew = xnew;
    a.nrm_part = st->nrm_part;
    hipLaunchKernelGGL(k_top, dim3(rn.dev.G), dim3(BT), 0, ctx->stream, a);
    IPD_KERNEL_CHECK();
    rn.nrsum = rn.dev.G;
    hipLaunchKernelGGL(k_conv, dim3(1), dim3(BT), 0, ctx->stream, (const double*)st->nrm_part,
                       rn.dev.G, st->hist, first ? 1 : 0);
    IPD_KERNEL_CHECK();
}

// Class_AMG.m:86-109
void amg_solve_dev(ipd_amg* h, const double* b_dev, const double* guess_dev, double* x_dev,
                   int32_t* it_out, double* rel_res_out, double* rel_resk, double* rhok) {
    ipd_ctx* ctx = h->ctx;
    CycleState* st = state_of(h);
    IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
    const AmgOpts& o = h->opts;
    const int N = h->L[1].A.nr;
    double* xa = h->x;
    double* xb = st->x2;
    if (guess_dev)
        IPD_HIP(hipMemcpyAsync(xa, guess_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
    else
        IPD_HIP(hipMemsetAsync(xa, 0, sizeof(double) * (size_t)N, ctx->stream));
    launch_top(h, st, b_dev, xa, nullptr, xb, true);                            // :89
    std::swap(xa, xb);
    double hh[5];
    ctx->fetch(st->hist, hh, 5);
    const bool wc = o.cycle == 'w', vc = o.cycle == 'v';
    int it = 0;
    double rel_res = 0.0;
    if (hh[0] == 0.0) {                                                          // :91-92
        if (rel_resk) rel_resk[0] = 0.0;
        if (rhok) rhok[0] = INFINITY;
    } else {
        it = 1;                                                                  // :94
        double last_rel = 1.0;
        if (rel_resk) rel_resk[0] = 1.0;
        if (rhok) rhok[0] = NAN;
        while (last_rel > o.retol && it <= o.maxit) {                            // :95
            const double* ecorr = nullptr;
            if (vc || wc) {
                IPD_HIP(hipMemsetAsync(h->L[1].e, 0, sizeof(double) * (size_t)N, ctx->stream));
                amg_cycle(h, 1, o.isnsp, wc, false);                             // :96-102
                ecorr = h->L[1].e;
            }
            launch_top(h, st, b_dev, xa, ecorr, xb, false);                      // :103-105
            std::swap(xa, xb);
            ctx->fetch(st->hist, hh, 5);
            rel_res = hh[3];
            last_rel = rel_res;
            if (rel_resk) rel_resk[it] = rel_res;
            if (rhok) rhok[it] = hh[4];
            ++it;
            if (hh[4] > 1.0) break;                                              // :106
        }
        it -= 1;                                                                 // :108
    }
    if (x_dev)
        IPD_HIP(hipMemcpyAsync(x_dev, xa, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
    if (xa != h->x) std::swap(h->x, st->x2);  // keep h->x pointing at the current iterate
    if (it_out) *it_out = it;
    if (rel_res_out) *rel_res_out = rel_res;
    ctx->sync();
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int ipd_amg_solve_dev(ipd_amg* h, const double* b_dev, const double* guess_dev,
                                 double* x_dev, int32_t* it, double* rel_res, double* rel_resk,
                                 double* rhok) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && b_dev && x_dev, IPD_E_ARG, "NULL argument");
        CallScope scope(h->ctx);
        amg_solve_dev(h, b_dev, guess_dev, x_dev, it, rel_res, rel_resk, rhok);
    });
}

extern "C" int ipd_amg_solve(ipd_amg* h, const double* b, const double* guess, double* x,
                             int32_t* it, double* rel_res, double* rel_resk, double* rhok) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && b && x, IPD_E_ARG, "NULL argument");
        ipd_ctx* ctx = h->ctx;
        CallScope scope(ctx);
        const size_t N = (size_t)h->L[1].A.nr;
        double* db = ctx->scratch->alloc<double>(N);
        double* dg = nullptr;
        double* dx = ctx->scratch->alloc<double>(N);
        ctx->upload(db, b, N);
        if (guess) {
            dg = ctx->scratch->alloc<double>(N);
            ctx->upload(dg, guess, N);
        }
        amg_solve_dev(h, db, dg, dx, it, rel_res, rel_resk, rhok);
        ctx->fetch(dx, x, N);
    });
}

static void run_cycle_api(ipd_amg* h, const double* r, int isnsp, int k, const double* e_in,
                          double* e_out, bool wc) {
    IPD_REQUIRE(h && r && e_out, IPD_E_ARG, "NULL argument");
    IPD_REQUIRE(k >= 1 && k <= h->J, IPD_E_ARG, "level k out of range");
    ipd_ctx* ctx = h->ctx;
    CallScope scope(ctx);
    CycleState* st = state_of(h);
    IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
    Level& lv = h->L[k];
    const size_t N = (size_t)lv.A.nr;
    ctx->upload(lv.r, r, N);
    LevelRun& rn = st->run[(size_t)k];
    hipLaunchKernelGGL(k_vec_sum, dim3(1), dim3(BT), 0, ctx->stream, (const double*)lv.r, (int)N,
                       rn.dev.rsum);
    IPD_KERNEL_CHECK();
    rn.nrsum = 1;
    bool keep = false;
    if (e_in && wc) {
        ctx->upload(lv.e, e_in, N);
        if (isnsp && k < h->J) {
            rn.parity = 0;
            hipLaunchKernelGGL(k_dot_sum, dim3(1), dim3(BT), 0, ctx->stream, (const double*)lv.Axi,
                               (const double*)lv.e, (int)N, rn.part[0]);
            IPD_KERNEL_CHECK();
            rn.npart = 1;
        }
        keep = true;
    } else {
        IPD_HIP(hipMemsetAsync(lv.e, 0, sizeof(double) * N, ctx->stream));
    }
    amg_cycle(h, k, isnsp, wc, keep);
    ctx->fetch(h->L[k].e, e_out, N);
}

extern "C" int ipd_amg_vcycle(ipd_amg* h, const double* r, int isnsp, int k, double* e) {
    return ipd_guard([&] { run_cycle_api(h, r, isnsp, k, nullptr, e, false); });
}

extern "C" int ipd_amg_wcycle(ipd_amg* h, const double* r, int isnsp, int k, const double* e_in,
                              double* e_out) {
    return ipd_guard([&] { run_cycle_api(h, r, isnsp, k, e_in, e_out, true); });
}

extern "C" int ipd_class_amg(ipd_ctx* ctx, const ipd_csc* A, const double* b, const double* guess,
                             const ipd_amg_opts* o, ipd_rng* rng, double* x, int32_t* it,
                             double* rel_res, double* rel_resk, double* rhok) {
    ipd_amg* h = nullptr;
    int rc = ipd_amg_setup(ctx, A, o, rng, &h);
    if (rc != IPD_OK) return rc;
    rc = ipd_amg_solve(h, b, guess, x, it, rel_res, rel_resk, rhok);
    ipd_amg_destroy(h);
    return rc;
}

extern "C" int ipd_pcg(ipd_ctx* ctx, const ipd_csc* H, const double* e, const double* guess,
                       const ipd_pcg_opts* o, double* d, int64_t* it, double* res, double* resk) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && H && e && d, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        Arena& tmp = *ctx->scratch;
        double tol = 1e-11;
        long long maxit = 10000;
        int precd = 2;  // PCG.m:24-27 defaults
        if (o) {
            if (o->retol >= 0) tol = o->retol;
            if (o->maxit >= 0) maxit = o->maxit;
            if (o->precd >= 0) precd = o->precd;
        }
        Csr h;
        csr_upload_from_csc(ctx, tmp, H, true, &h);
        const size_t N = (size_t)h.nr;
        double* de = tmp.alloc<double>(N);
        double* dd = tmp.alloc<double>(N);
        double* dg = nullptr;
        ctx->upload(de, e, N);
        if (guess) {
            dg = tmp.alloc<double>(N);
            ctx->upload(dg, guess, N);
        }
        long long its = 0;
        pcg_dev(ctx, h, de, dg, tol, maxit, precd, dd, &its, res, resk);
        if (it) *it = its;
        ctx->fetch(dd, d, N);
    });
}

// ---------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------
// S(X) = 12 nnz + 4 (rows+1) + 8 rows + 8 cols per CSR SpMV.
static double spmv_bytes(const Csr& m) {
    return 12.0 * m.nnz + 4.0 * (m.nr + 1) + 8.0 * m.nr + 8.0 * m.nc;
}

// B_V with the fused Gauss-Seidel form (one S(A_1) per level-1 sweep, the stated
// minimum): per level (2 nu + 1) S(A_k) + S(P) + S(P') + 6 nu 8 N_k, weighted by
// the visit count (1 for V, 2^(k-1) for W), + coarsest PCG + the outer loop's
// residual S(A_1) + 32 M.
static double cycle_bytes(const ipd_amg* h) {
    const bool wc = h->opts.cycle == 'w';
    const double nu = h->opts.smoth;
    double total = 0.0;
    double visits = 1.0;
    for (int k = 1; k < h->J; ++k) {
        const Level& lv = h->L[k];
        const Level& cl = h->L[k + 1];
        const double per = (2 * nu + 1) * spmv_bytes(lv.A) + spmv_bytes(cl.P) + spmv_bytes(cl.Pt) +
                           6 * nu * 8.0 * lv.A.nr;
        total += visits * per;
        if (wc && k + 1 < h->J) visits *= 2.0;
    }
    total += visits * 2.0 * spmv_bytes(h->L[h->J].A);  // >= 1 PCG iteration + initial residual
    total += spmv_bytes(h->L[1].A) + 32.0 * h->L[1].A.nr;
    return total;
}

extern "C" int ipd_amg_cycle_bytes(const ipd_amg* h, double* bytes_per_cycle) {
    if (!h || !bytes_per_cycle) return IPD_E_ARG;
    *bytes_per_cycle = cycle_bytes(h);
    return IPD_OK;
}

extern "C" int ipd_amg_bench_cycles(ipd_amg* h, const double* b_dev, double* x_dev, int cycles,
                                    double* total_ms, double* bytes_per_cycle) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && b_dev && x_dev && cycles > 0 && total_ms, IPD_E_ARG, "bad argument");
        ipd_ctx* ctx = h->ctx;
        CallScope scope(ctx);
        CycleState* st = state_of(h);
        IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
        const int N = h->L[1].A.nr;
        const bool wc = h->opts.cycle == 'w';
        const bool anyc = wc || h->opts.cycle == 'v';
        double* xa = h->x;
        double* xb = st->x2;
        IPD_HIP(hipMemcpyAsync(xa, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        hipEvent_t ev0, ev1;
        IPD_HIP(hipEventCreate(&ev0));
        IPD_HIP(hipEventCreate(&ev1));
        launch_top(h, st, b_dev, xa, nullptr, xb, true);
        std::swap(xa, xb);
        IPD_HIP(hipEventRecord(ev0, ctx->stream));
        for (int c = 0; c < cycles; ++c) {
            const double* ecorr = nullptr;
            if (anyc) {
                IPD_HIP(hipMemsetAsync(h->L[1].e, 0, sizeof(double) * (size_t)N, ctx->stream));
                amg_cycle(h, 1, h->opts.isnsp, wc, false);
                ecorr = h->L[1].e;
            }
            launch_top(h, st, b_dev, xa, ecorr, xb, false);
            std::swap(xa, xb);
        }
        IPD_HIP(hipEventRecord(ev1, ctx->stream));
        IPD_HIP(hipEventSynchronize(ev1));
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        IPD_HIP(hipEventDestroy(ev0));
        IPD_HIP(hipEventDestroy(ev1));
        IPD_HIP(hipMemcpyAsync(x_dev, xa, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        ctx->sync();
        *total_ms = ms;
        if (bytes_per_cycle) *bytes_per_cycle = cycle_bytes(h);
    });
}
