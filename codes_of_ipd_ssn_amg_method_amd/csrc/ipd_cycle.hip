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

#include "ipd_cycle_host.h"
