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

#include "ipd_cycle_dev.h"
#include "ipd_cycle_phases.h"

// Dynamic LDS = the staged gather vector (N doubles) when STAGED, else nothing.
template <bool STAGED, bool PAD>
__global__ __launch_bounds__(BT) void k_smooth(SmoothArgs a) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_smooth<STAGED, PAD>(a, blockIdx.x, gridDim.x, &lds, xs_dyn);
}
template <bool STAGED, bool PAD>
__global__ __launch_bounds__(BT) void k_resid(LevelDev lv, const double* e, int row0, int row1) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_resid<STAGED, PAD>(lv, e, row0, row1, blockIdx.x, gridDim.x, &lds, xs_dyn);
}
template <bool STAGED>
__global__ __launch_bounds__(BT) void k_xfer(XferArgs a) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_xfer<STAGED>(a, blockIdx.x, gridDim.x, &lds, xs_dyn);
}
template <bool STAGED, bool PAD>
__global__ __launch_bounds__(BT) void k_top(TopArgs a) {
    __shared__ PhaseLds lds;
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    phase_top<STAGED, PAD>(a, blockIdx.x, gridDim.x, &lds, xs_dyn);
}

// padded off-diagonal copy of a CSR matrix: one wave per row
__global__ __launch_bounds__(256) void k_offdiag_maxlen(int N, const int* __restrict__ rp,
                                                        const int* __restrict__ ci,
                                                        int* __restrict__ maxlen) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < N; r += nwaves) {
        int hasd = 0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) hasd |= (ci[t] == r);
        hasd = __any(hasd) ? 1 : 0;
        if (lane == 0) atomicMax(maxlen, rp[r + 1] - rp[r] - hasd);
    }
}

__global__ __launch_bounds__(256) void k_pad_build(int N, int S, const int* __restrict__ rp,
                                                   const int* __restrict__ ci,
                                                   const double* __restrict__ va,
                                                   unsigned short* __restrict__ pci,
                                                   double* __restrict__ pva,
                                                   double* __restrict__ diag) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < N; r += nwaves) {
        const int b = rp[r], e = rp[r + 1];
        int dpos = 0x7fffffff;
        for (int t = b + lane; t < e; t += 64)
            if (ci[t] == r) dpos = t;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) dpos = min(dpos, __shfl_xor(dpos, d));
        const bool hasd = dpos != 0x7fffffff;
        const size_t base = (size_t)r * S;
        for (int t = b + lane; t < e; t += 64) {
            if (t == dpos) continue;
            const int k = (t - b) - ((hasd && dpos < t) ? 1 : 0);
            pci[base + k] = (unsigned short)ci[t];
            pva[base + k] = va[t];
        }
        const int len = (e - b) - (hasd ? 1 : 0);
        for (int k = len + lane; k < S; k += 64) {
            pci[base + k] = 0;
            pva[base + k] = 0.0;
        }
        if (lane == 0) diag[r] = hasd ? va[dpos] : 0.0;
    }
}

// hist[0] = res0 (set on the first call), hist[1] = res, hist[2] = previous res,
// hist[3] = rel_res, hist[4] = rhok                        Class_AMG.m:89,103-105
struct ConvArgs {
    const double* r;
    int n;
    double* hist;
    int first;
};

__device__ __forceinline__ void conv_block(const ConvArgs& a, double* red) {
    double s = 0.0;
    for (int k0 = threadIdx.x; k0 < a.n; k0 += 4 * BT) {  // 4 independent loads in flight
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * BT;
            v[u] = a.r[k < a.n ? k : a.n - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += (k0 + u * BT < a.n) ? v[u] * v[u] : 0.0;
    }
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) {
        double* hist = a.hist;
        const double res = sqrt(tot);
        if (a.first) {
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

__global__ __launch_bounds__(BT) void k_conv(ConvArgs a) {
    __shared__ double red[16];
    conv_block(a, red);
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

// ---------------------------------------------------------------------------
// fused single-workgroup program
// ---------------------------------------------------------------------------
// Phases whose row range fits one workgroup (a few thousand nonzeros) cost far more as
// launches (2.3 us floor + 3-10 us of latency each, and only 1-8 CUs busy) than as
// work.  The host therefore strings consecutive small phases -- e.g. the ten Gauss-
// Seidel half sweeps, the residual and the restriction of a small fine level, or
// restriction + coarsest PCG + prolongation -- into ONE launch of this kernel: one
// workgroup interprets the descriptor list, with a workgroup barrier between phases.
// Descriptors travel as kernel arguments (no upload, captured by value in graphs).
struct ResidDesc {
    LevelDev lv;
    const double* e;
    int row0, row1;
};
enum : int { PH_SMOOTH = 1, PH_RESID, PH_XFER, PH_TOP, PH_PCG, PH_CONV };
struct PhaseDesc {
    int type;
    int pad_;
    union U {
        SmoothArgs s;
        ResidDesc r;
        XferArgs x;
        TopArgs t;
        PcgArgs p;
        ConvArgs c;
    } u;
};
static constexpr int FUSED_MAX = 16;
struct FusedProg {
    int n;
    int pad_;
    PhaseDesc d[FUSED_MAX];
};
static_assert(sizeof(FusedProg) <= 3900, "fused program must fit the 4 KiB kernel-argument segment");

__global__ __launch_bounds__(BT) void k_fused(FusedProg prog) {
    __shared__ PhaseLds lds;
    __shared__ double red[16];
    extern __shared__ __attribute__((aligned(16))) double xs_dyn[];
    for (int i = 0; i < prog.n; ++i) {
        const PhaseDesc& d = prog.d[i];
        switch (d.type) {
            case PH_SMOOTH:
                if (d.u.s.lv.S > 0)
                    phase_smooth<true, true>(d.u.s, 0, 1, &lds, xs_dyn);
                else
                    phase_smooth<true, false>(d.u.s, 0, 1, &lds, xs_dyn);
                break;
            case PH_RESID:
                if (d.u.r.lv.S > 0)
                    phase_resid<true, true>(d.u.r.lv, d.u.r.e, d.u.r.row0, d.u.r.row1, 0, 1, &lds,
                                            xs_dyn);
                else
                    phase_resid<true, false>(d.u.r.lv, d.u.r.e, d.u.r.row0, d.u.r.row1, 0, 1, &lds,
                                             xs_dyn);
                break;
            case PH_XFER:
                phase_xfer<true>(d.u.x, 0, 1, &lds, xs_dyn);
                break;
            case PH_TOP:
                if (d.u.t.lv.S > 0)
                    phase_top<true, true>(d.u.t, 0, 1, &lds, xs_dyn);
                else
                    phase_top<true, false>(d.u.t, 0, 1, &lds, xs_dyn);
                break;
            case PH_PCG:
                pcg_block(d.u.p, red);
                break;
            case PH_CONV:
                conv_block(d.u.c, red);
                break;
            default:
                break;
        }
        __syncthreads();
    }
}

// lanes per row: 3..6 entries per lane (one ROW_U batch), widened while the launch
// would leave most of the chip idle
static int pick_lanes(long long nnz, int nrows, int blocks_target) {
    if (nrows <= 0) return 1;
    const double avg = (double)nnz / (double)nrows;
    int L = 1;  // short rows: one lane walks the whole row in a single ROW_U batch
    while (L < BT && (double)L * 6.0 < avg) L <<= 1;
    // widen while most of the chip would idle (tools/ubench_small.hip: a 1024-row launch of
    // short rows costs the same 6.5 us on 1, 4 or 16 workgroups, so spreading is free and
    // keeps one CU's load-issue rate from becoming the limit)
    while (L < BT && (long long)nrows * L < (long long)blocks_target * BT / 2 &&
           (double)L * 2.0 <= avg)
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
// whole Class_AMG solve phase in ONE workgroup
// ---------------------------------------------------------------------------
// Realistic Newton systems have tiny hierarchies (every level a few thousand nonzeros,
// SURVEY F4/F5): a W cycle is then several hundred dependent micro-phases and the
// multi-launch path is bound by launch latency and by the host (measured 1.9 ms per
// W cycle at M = 1000).  Here one workgroup interprets the V/W recursion itself
// (MG_Vcycle.m:12-45, MG_Wcycle.m:13-46), the stationary iteration and its stopping
// rules (Class_AMG.m:86-109): one launch and one read-back per solve.
static constexpr int SOLVE_ML = 24;
struct SolveLevel {
    LevelDev lv;
    double* e;
    double* e2;
    double* w;
    XferArgs rest;  // r_{k+1} = P' rr_k      (valid for k < J)
    XferArgs prol;  // e_k += P e_{k+1}
    int nnzA, nnzP;  // sizes for the LDS cache copy
};
struct SolveDesc {
    int J, nu, isnsp, wcycle, anycycle, maxit;
    int k_lds;        // levels k_lds..J (and the transfers between them) are cached in LDS
    int k_tiny;       // levels k_tiny..J have <= 64 rows: their whole sub-cycle runs in ONE wave
    int stage_bytes;  // size of the gather staging area at the start of dynamic LDS
    double retol;
    PcgArgs pcg;
    SolveLevel L[SOLVE_ML + 1];
};

struct SolveCtx {  // per-thread copies of uniform state
    const SolveDesc* D;
    PhaseLds* lds;
    double* red;
    double* xs;
    unsigned swapmask;  // bit k: the current iterate of level k lives in e2
    unsigned zeromask;  // bit k: the iterate of level k is identically zero (not materialised)
};

__device__ __forceinline__ double* sol_e(const SolveCtx& c, int k) {
    return ((c.swapmask >> k) & 1u) ? c.D->L[k].e2 : c.D->L[k].e;
}
__device__ __forceinline__ double* sol_e2(const SolveCtx& c, int k) {
    return ((c.swapmask >> k) & 1u) ? c.D->L[k].e : c.D->L[k].e2;
}

__device__ __forceinline__ void sol_smooth_call(SolveCtx& c, const SmoothArgs& a) {
    phase_smooth<true, false>(a, 0, 1, c.lds, c.xs);  // descriptors carry S = 0: CSR walk only
    __syncthreads();
}

__device__ __forceinline__ void sol_sweep(SolveCtx& c, int k, bool post) {
    const SolveLevel& L = c.D->L[k];
    SmoothArgs a;
    a.lv = L.lv;
    a.eold = sol_e(c, k);
    a.enew = sol_e2(c, k);
    a.win = L.w;
    a.wout = L.w;
    a.isnsp = c.D->isnsp;
    a.staged = 1;
    a.eold_zero = (c.zeromask >> k) & 1u;
    const int nf = L.lv.nf, N = L.lv.N;
    if (nf == 0) {
        a.row0 = 0;
        a.row1 = N;
        a.u0 = a.u1 = 0;
        a.wout = nullptr;
        sol_smooth_call(c, a);
    } else {
        const int f0 = post ? nf : 0, f1 = post ? N : nf;
        const int s0 = post ? 0 : nf, s1 = post ? nf : N;
        a.row0 = f0;
        a.row1 = f1;
        a.u0 = a.u1 = 0;
        sol_smooth_call(c, a);
        a.row0 = s0;
        a.row1 = s1;
        a.u0 = f0;
        a.u1 = f1;
        a.wout = nullptr;
        sol_smooth_call(c, a);
    }
    c.swapmask ^= (1u << k);
    c.zeromask &= ~(1u << k);
}

// ---- wave-level sub-cycle: levels with <= 64 rows, everything LDS-resident ------------
// A W cycle visits level k 2^(k-1) times, so most of its phases run on the deepest,
// tiniest levels (a dozen rows).  There a 1024-thread phase is all fixed cost (barriers,
// descriptor reads), so ONE wave runs the whole sub-cycle below level k_tiny: lane i owns
// row i, vectors live in LDS, a wave is its own barrier (LDS operations of one wave
// execute in order; the fence only stops the compiler from reordering them).
__device__ __forceinline__ void tiny_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double tiny_rowdot(const int* rp, const int* ci, const double* va,
                                              int row, bool valid, const double* x) {
    double s = 0.0;
    if (valid)
        for (int t = rp[row]; t < rp[row + 1]; ++t) s += va[t] * x[ci[t]];
    return s;
}

__device__ __forceinline__ void tiny_sweep(SolveCtx& c, int k) {  // Jacobi levels only (k >= 2)
    const SolveLevel& L = c.D->L[k];
    const int i = threadIdx.x, N = L.lv.N;
    const bool valid = i < N, ez = (c.zeromask >> k) & 1u;
    const double* e = sol_e(c, k);
    double* en = sol_e2(c, k);
    const double eo = (valid && !ez) ? e[i] : 0.0;
    const double rv = valid ? L.lv.r[i] : 0.0;
    const double ax = valid ? L.lv.Axi[i] : 0.0;
    double cc = 0.0;
    if (c.D->isnsp) cc = wave_sum(rv - ax * eo) / L.lv.xx[0];
    const double sd = ez ? 0.0 : tiny_rowdot(L.lv.rp, L.lv.ci, L.lv.va, i, valid, e);
    if (valid) en[i] = eo + L.lv.dinv[i] * (rv - sd - ax * cc) + cc;
    tiny_sync();
    c.swapmask ^= (1u << k);
    c.zeromask &= ~(1u << k);
}

__device__ __forceinline__ void tiny_pcg(SolveCtx& c, int k) {  // PCG.m:68-87, Jacobi, zero guess
    const SolveLevel& L = c.D->L[k];
    const PcgArgs& a = c.D->pcg;
    const int i = threadIdx.x, N = L.lv.N;
    const bool valid = i < N;
    double* pv = a.work;  // p shared through LDS
    double dg = 1.0;
    if (valid)
        for (int t = L.lv.rp[i]; t < L.lv.rp[i + 1]; ++t)
            if (L.lv.ci[t] == i) dg = L.lv.va[t];
    double r = valid ? L.lv.r[i] : 0.0;
    double p = a.precd == 2 ? r / dg : r;
    double d = 0.0;
    double delta_new = wave_sum(valid ? r * p : 0.0);
    const double delta_0 = delta_new, thresh = a.tol * a.tol * delta_0;
    long long it = 0;
    while (it < a.maxit && delta_new > thresh) {
        const double delta_old = delta_new;
        if (valid) pv[i] = p;
        tiny_sync();
        const double q = tiny_rowdot(L.lv.rp, L.lv.ci, L.lv.va, i, valid, pv);
        tiny_sync();
        const double alpha = delta_old / wave_sum(valid ? q * p : 0.0);
        d += alpha * p;
        r -= alpha * q;
        const double w = a.precd == 2 ? r / dg : r;
        delta_new = wave_sum(valid ? r * w : 0.0);
        p = w + (delta_new / delta_old) * p;
        ++it;
    }
    if (valid) sol_e(c, k)[i] = d;
    tiny_sync();
    c.zeromask &= ~(1u << k);
}

// sub-cycle rooted at level k0 >= k_tiny (r_{k0} is in LDS); executed by wave 0 only
__device__ __forceinline__ void tiny_cycle(SolveCtx& c, int k0, bool keep0) {
    const SolveDesc* D = c.D;
    const int J = D->J, nu = D->nu, i = threadIdx.x;
    unsigned visited = 0;
    int k = k0;
    bool entering = true, keep = keep0;
    for (int guard = 0; guard < (1 << 22); ++guard) {
        if (entering) {
            if (k == J) {
                tiny_pcg(c, J);
                if (k == k0) return;
                entering = false;
                k = J - 1;
                continue;
            }
            const SolveLevel& L = D->L[k];
            if (!keep) {
                c.zeromask |= (1u << k);
                if (nu == 0) {
                    if (i < L.lv.N) sol_e(c, k)[i] = 0.0;
                    tiny_sync();
                    c.zeromask &= ~(1u << k);
                }
            }
            for (int s = 0; s < nu; ++s) tiny_sweep(c, k);
            {   // residual, then restriction into the child's right-hand side
                const bool valid = i < L.lv.N;
                const double sd = tiny_rowdot(L.lv.rp, L.lv.ci, L.lv.va, i, valid, sol_e(c, k));
                if (valid) L.lv.rr[i] = L.lv.r[i] - sd;
                tiny_sync();
                const bool cv = i < L.rest.nrows;
                const double rc = tiny_rowdot(L.rest.rp, L.rest.ci, L.rest.va, i, cv, L.lv.rr);
                if (cv) L.rest.y[i] = rc;
                tiny_sync();
            }
            visited &= ~(1u << (k + 1));
            k = k + 1;
            keep = false;
        } else {
            const bool again = D->wcycle && (k + 1 < J) && !((visited >> (k + 1)) & 1u);
            if (again) {
                visited |= (1u << (k + 1));
                k = k + 1;
                keep = true;
                entering = true;
                continue;
            }
            const SolveLevel& L = D->L[k];
            {
                const bool valid = i < L.lv.N;
                double* e = sol_e(c, k);
                const double sd = tiny_rowdot(L.prol.rp, L.prol.ci, L.prol.va, i, valid, sol_e(c, k + 1));
                if (valid) e[i] = e[i] + sd;
                tiny_sync();
            }
            for (int s = 0; s < nu; ++s) tiny_sweep(c, k);
            if (k == k0) return;
            k = k - 1;
        }
    }
}

// one V or W cycle rooted at level k0 on r_{k0} (in L[k0].lv.r); the correction ends up in
// sol_e(c, k0).  keep0: start from the current iterate of level k0 (MG_Wcycle.m:30).
__device__ __forceinline__ void sol_cycle(SolveCtx& c, int k0 = 1, bool keep0 = false) {
    const SolveDesc* D = c.D;
    const int J = D->J, nu = D->nu;
    unsigned visited = 0;  // bit k: level k has completed one visit under its current parent
    int k = k0;
    bool entering = true, keep = keep0;
    for (int guard = 0; guard < (1 << 22); ++guard) {
        if (entering && k >= D->k_tiny && k > 1) {
            // the whole sub-cycle below here runs in wave 0; 2*nu sweeps per visit leave the
            // e/e2 roles of every level unchanged, so the other waves need no state update
            if (threadIdx.x < 64) {
                SolveCtx t = c;
                tiny_cycle(t, k, keep);
            }
            __syncthreads();
            c.zeromask &= ~(1u << k);
            if (k == k0) return;
            entering = false;
            k = k - 1;
            continue;
        }
        if (entering) {
            if (k == J) {  // coarsest: PCG(A, r)                         MG_Vcycle.m:43
                PcgArgs a = D->pcg;
                a.rhs = D->L[J].lv.r;
                a.d = sol_e(c, J);
                pcg_block(a, c.red);
                __syncthreads();
                c.zeromask &= ~(1u << J);
                if (J == k0) return;
                entering = false;
                k = J - 1;
                continue;
            }
            if (!keep) {
                c.zeromask |= (1u << k);
                if (nu == 0) {  // no sweep will write the iterate: materialise the zero
                    double* e = sol_e(c, k);
                    for (int i = threadIdx.x; i < D->L[k].lv.N; i += BT) e[i] = 0.0;
                    __syncthreads();
                    c.zeromask &= ~(1u << k);
                }
            }
            for (int s = 0; s < nu; ++s) sol_sweep(c, k, false);          // :14-25
            {
                const LevelDev& lv = D->L[k].lv;                          // :27
                phase_resid<true, false>(lv, sol_e(c, k), 0, lv.N, 0, 1, c.lds, c.xs);
                __syncthreads();
                phase_xfer<true>(D->L[k].rest, 0, 1, c.lds, c.xs);
                __syncthreads();
            }
            visited &= ~(1u << (k + 1));
            k = k + 1;
            keep = false;
            entering = true;
        } else {  // back in level k from its child k+1
            const bool again = D->wcycle && (k + 1 < J) && !((visited >> (k + 1)) & 1u);
            if (again) {  // MG_Wcycle.m:30: second correction starting from the first one
                visited |= (1u << (k + 1));
                k = k + 1;
                keep = true;
                entering = true;
                continue;
            }
            XferArgs pa = D->L[k].prol;                                    // :31
            pa.x = sol_e(c, k + 1);
            pa.y = sol_e(c, k);
            phase_xfer<true>(pa, 0, 1, c.lds, c.xs);
            __syncthreads();
            for (int s = 0; s < nu; ++s) sol_sweep(c, k, true);           // :33-41
            if (k == k0) return;
            k = k - 1;
        }
    }
}

__device__ __forceinline__ void sol_top(SolveCtx& c, const double* b, const double* x,
                                        const double* e, double* xnew, double* hist, int first) {
    TopArgs a;
    a.lv = c.D->L[1].lv;
    a.b = b;
    a.x = x;
    a.e = e;
    a.xnew = xnew;
    a.row0 = 0;
    a.row1 = a.lv.N;
    a.staged = 1;
    phase_top<true, false>(a, 0, 1, c.lds, c.xs);
    __syncthreads();
    ConvArgs ca;
    ca.r = a.lv.r;
    ca.n = a.lv.N;
    ca.hist = hist;
    ca.first = first;
    conv_block(ca, c.red);
    __syncthreads();
}

// Copies levels k_lds..J (matrices, transfers, work vectors) into dynamic LDS and patches the
// LDS descriptor `LD` to point at the copies.  They are tiny, but a W cycle visits level k
// 2^(k-1) times, so their phases must not pay global-memory latency.  Every thread walks the
// same carve sequence; pointers are patched by thread 0.
__device__ __forceinline__ void sol_cache_levels(const SolveDesc* D, SolveDesc* LD, char* dyn_raw) {
        size_t off = (size_t)D->stage_bytes + ((sizeof(SolveDesc) + 15) / 16) * 16;
        auto carve = [&](size_t bytes) {
            char* p = dyn_raw + off;
            off += (bytes + 15) / 16 * 16;
            return p;
        };
        auto copy_i = [&](const int* src, size_t n) {
            int* d = reinterpret_cast<int*>(carve(n * 4));
            for (size_t i = threadIdx.x; i < n; i += BT) d[i] = src[i];
            return d;
        };
        auto copy_d = [&](const double* src, size_t n) {
            double* d = reinterpret_cast<double*>(carve(n * 8));
            for (size_t i = threadIdx.x; i < n; i += BT) d[i] = src[i];
            return d;
        };
        auto copy_h = [&](const unsigned short* src, size_t n) {
            unsigned short* d = reinterpret_cast<unsigned short*>(carve(n * 2));
            for (size_t i = threadIdx.x; i < n; i += BT) d[i] = src[i];
            return d;
        };
        const bool t0 = threadIdx.x == 0;
        for (int k = D->k_lds; k <= D->J; ++k) {
            const SolveLevel& G = D->L[k];
            SolveLevel& T = LD->L[k];
            const size_t N = (size_t)G.lv.N;
            const int* rp = copy_i(G.lv.rp, N + 1);
            const int* ci = copy_i(G.lv.ci, (size_t)G.nnzA);
            const double* va = copy_d(G.lv.va, (size_t)G.nnzA);
            const double* dinv = copy_d(G.lv.dinv, N);
            const double* Axi = copy_d(G.lv.Axi, N);
            const double* xx = copy_d(G.lv.xx, 1);
            const unsigned short* pci = nullptr;
            const double* pva = nullptr;
            const double* diag = nullptr;
            if (G.lv.S > 0) {
                pci = copy_h(G.lv.pci, N * G.lv.S);
                pva = copy_d(G.lv.pva, N * G.lv.S);
                diag = copy_d(G.lv.diag, N);
            }
            double* r = reinterpret_cast<double*>(carve(N * 8));
            double* rr = reinterpret_cast<double*>(carve(N * 8));
            double* e = reinterpret_cast<double*>(carve(N * 8));
            double* e2 = reinterpret_cast<double*>(carve(N * 8));
            double* w = reinterpret_cast<double*>(carve(N * 8));
            if (t0) {
                T.lv.rp = rp;
                T.lv.ci = ci;
                T.lv.va = va;
                T.lv.dinv = dinv;
                T.lv.Axi = Axi;
                T.lv.xx = xx;
                T.lv.pci = pci;
                T.lv.pva = pva;
                T.lv.diag = diag;
                T.lv.r = r;
                T.lv.rr = rr;
                T.e = e;
                T.e2 = e2;
                T.w = w;
            }
            if (k < D->J) {  // transfers between two cached levels
                const size_t Nc = (size_t)G.rest.nrows;
                const int* trp = copy_i(G.rest.rp, Nc + 1);
                const int* tci = copy_i(G.rest.ci, (size_t)G.nnzP);
                const double* tva = copy_d(G.rest.va, (size_t)G.nnzP);
                const int* prp = copy_i(G.prol.rp, N + 1);
                const int* pci2 = copy_i(G.prol.ci, (size_t)G.nnzP);
                const double* pva2 = copy_d(G.prol.va, (size_t)G.nnzP);
                if (t0) {
                    T.rest.rp = trp;
                    T.rest.ci = tci;
                    T.rest.va = tva;
                    T.prol.rp = prp;
                    T.prol.ci = pci2;
                    T.prol.va = pva2;
                }
            }
        }
        __syncthreads();
        if (t0) {  // vectors that cross level boundaries, and the coarsest PCG
            for (int k = 1; k < D->J; ++k) {
                LD->L[k].rest.x = LD->L[k].lv.rr;
                LD->L[k].rest.y = LD->L[k + 1].lv.r;
            }
            const int J = D->J;
            if (J >= D->k_lds) {
                LD->pcg.rp = LD->L[J].lv.rp;
                LD->pcg.ci = LD->L[J].lv.ci;
                LD->pcg.va = LD->L[J].lv.va;
            }
        }
        if (D->J >= D->k_lds) {
            double* work = reinterpret_cast<double*>(carve(4 * (size_t)D->L[D->J].lv.N * 8));
            if (t0) LD->pcg.work = work;
        }
        __syncthreads();
}

// out[0] = it, out[1] = rel_res, out[2] = res0; rel_resk at out[4 ..], rhok at out[4+maxit+2 ..]
// fixed_cycles > 0: run exactly that many loop bodies without the stopping rules (bench hook)
template <bool CACHED>
__global__ __launch_bounds__(BT) void k_solve_small(const SolveDesc* __restrict__ D_global,
                                                    const double* __restrict__ b, double* xa,
                                                    double* xb, double* hist, double* out,
                                                    int fixed_cycles) {
    __shared__ PhaseLds lds;
    __shared__ double red[16];
    extern __shared__ __attribute__((aligned(16))) char dyn_raw[];
    // dynamic LDS: [ staging vector | descriptor copy | cached levels ]
    const SolveDesc* D = D_global;
    SolveDesc* LD = reinterpret_cast<SolveDesc*>(dyn_raw + D->stage_bytes);
    if (CACHED) {
        const int* src = reinterpret_cast<const int*>(D);
        int* dst = reinterpret_cast<int*>(LD);
        for (int i = threadIdx.x; i < (int)(sizeof(SolveDesc) / 4); i += BT) dst[i] = src[i];
    }
    __syncthreads();
    if (CACHED) sol_cache_levels(D, LD, dyn_raw);
    SolveCtx c;
    // without cached levels the descriptor stays in global memory: its (uniform) fields
    // are then fetched with scalar loads and live in SGPRs instead of VGPRs
    c.D = CACHED ? LD : D_global;
    c.lds = &lds;
    c.red = red;
    c.xs = reinterpret_cast<double*>(dyn_raw);
    c.swapmask = 0;
    c.zeromask = 0;
    D = c.D;
    const int N = D->L[1].lv.N;
    const int maxit = D->maxit;
    double* const x_home = xa;
    double* relk = out + 4;
    double* rhok = out + 4 + (maxit + 2);
    sol_top(c, b, xa, nullptr, xb, hist, 1);                              // Class_AMG.m:89
    {
        double* t = xa;
        xa = xb;
        xb = t;
    }
    const double res0 = hist[0];
    int it = 0;
    double rel_res = 0.0;
    if (fixed_cycles > 0) {
        for (int cyc = 0; cyc < fixed_cycles; ++cyc) {
            const double* ecorr = nullptr;
            if (D->anycycle) {
                sol_cycle(c);
                ecorr = sol_e(c, 1);
            }
            sol_top(c, b, xa, ecorr, xb, hist, 0);
            double* t = xa;
            xa = xb;
            xb = t;
        }
        it = fixed_cycles;
        rel_res = hist[3];
    } else if (res0 == 0.0) {                                             // :91-92
        if (threadIdx.x == 0) {
            relk[0] = 0.0;
            rhok[0] = INFINITY;
        }
    } else {
        it = 1;                                                           // :94
        double last_rel = 1.0;
        if (threadIdx.x == 0) {
            relk[0] = 1.0;
            rhok[0] = NAN;
        }
        while (last_rel > D->retol && it <= maxit) {                      // :95
            const double* ecorr = nullptr;
            if (D->anycycle) {
                sol_cycle(c);                                             // :96-102
                ecorr = sol_e(c, 1);
            }
            sol_top(c, b, xa, ecorr, xb, hist, 0);                        // :103-105
            double* t = xa;
            xa = xb;
            xb = t;
            rel_res = hist[3];
            const double rho = hist[4];
            if (threadIdx.x == 0) {
                relk[it] = rel_res;
                rhok[it] = rho;
            }
            last_rel = rel_res;
            ++it;
            if (rho > 1.0) break;                                         // :106
            __syncthreads();  // hist is rewritten by the next conv_block
        }
        it -= 1;                                                          // :108
    }
    __syncthreads();
    if (xa != x_home)
        for (int i = threadIdx.x; i < N; i += BT) x_home[i] = xa[i];
    if (threadIdx.x == 0) {
        out[0] = (double)it;
        out[1] = rel_res;
        out[2] = res0;
    }
}

// Sub-cycle rooted at level k_lds >= 2 of a hierarchy whose upper levels run as multi-workgroup
// launches: ONE workgroup, every level from the root down cached in LDS.  r_{root} is read from
// and the correction written to the global vectors the surrounding launches use.
__global__ __launch_bounds__(BT) void k_subcycle(const SolveDesc* __restrict__ D_global, int keep) {
    __shared__ PhaseLds lds;
    __shared__ double red[16];
    extern __shared__ __attribute__((aligned(16))) char dyn_raw[];
    const SolveDesc* D = D_global;
    SolveDesc* LD = reinterpret_cast<SolveDesc*>(dyn_raw + D->stage_bytes);
    {
        const int* src = reinterpret_cast<const int*>(D);
        int* dst = reinterpret_cast<int*>(LD);
        for (int i = threadIdx.x; i < (int)(sizeof(SolveDesc) / 4); i += BT) dst[i] = src[i];
    }
    __syncthreads();
    sol_cache_levels(D, LD, dyn_raw);
    const int k0 = D->k_lds, N0 = D->L[k0].lv.N;
    {
        double* r = LD->L[k0].lv.r;
        double* e = LD->L[k0].e;
        const double* gr = D->L[k0].lv.r;
        const double* ge = D->L[k0].e;
        for (int i = threadIdx.x; i < N0; i += BT) {
            r[i] = gr[i];
            if (keep) e[i] = ge[i];
        }
    }
    __syncthreads();
    SolveCtx c;
    c.D = LD;
    c.lds = &lds;
    c.red = red;
    c.xs = reinterpret_cast<double*>(dyn_raw);
    c.swapmask = 0;
    c.zeromask = 0;
    sol_cycle(c, k0, keep != 0);
    __syncthreads();
    const double* res = sol_e(c, k0);
    double* ge = D->L[k0].e;
    for (int i = threadIdx.x; i < N0; i += BT) ge[i] = res[i];
}

#include "ipd_cycle_host.h"
