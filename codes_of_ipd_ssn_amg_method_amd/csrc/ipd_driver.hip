// APD / semismooth-Newton drivers and A-ADMM warm starts on device-resident vectors
// (SURVEY.md section 8 rows f1, f2):
//   Class1/APD_SsN_Class1.m:101-275, Class2/APD_SsN_Class2.m:95-285,
//   Class1/warmup_class1.m:22-96,    Class2/warmup_class2.m:19-100.
//
// Every mn-sized vector (c, xk, vk, wk, phi, gama, the mask s) lives in HBM for the whole
// run; the host keeps only the scalars of the scripts (ak, bk, tk, norms, counters).
//
// The mn-sized work of one Newton step is ONE streaming pass (`OpEvalT`): from wk and the
// multiplier it forms zk = (wk - H'lk)/tk on the fly (never stored), writes the byte mask
// s = (0 <= zk <= gama), and reduces Ax(prox(zk)), |prox(zk)|^2, phi'prox(zk) -- i.e. Fk,
// the active set and the Armijo merit of one trial point come out of 8 bytes read + 1 byte
// written per entry, where the scripts make ~10 passes (Aty, zk, prox, Ax, norms, s).
// These passes are HBM-bound: 9 B/entry (class 1), 17 B/entry with phi (class 2).
//
// Layout: x(i,j) = x[i + j*m] (MATLAB's reshape(x,m,n)); a wave owns 64 rows x (16*reps)
// columns, lanes run along i (coalesced), 16 independent loads in flight per lane and array.
// Row sums accumulate in registers, the 16 column sums of a chunk come out of a register
// butterfly (k_tiles); partial sums are combined by small epilogue kernels in a fixed order
// (deterministic, no float atomics), which also do all O(m+n) vector work.
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <limits>

#include "ipd_amg_internal.h"
#include "ipd_cycle_dev.h"

namespace {

constexpr int TR = 256;        // tile rows
constexpr int TC = 16;         // tile columns per sub-tile
constexpr int NSC = 8;         // scalar accumulators per workgroup
constexpr int MK = 8;          // line-search trial points evaluated by one merit pass

struct Geo {
    int m, n, nib, njg, reps;  // njg column groups of reps*TC columns
};

static Geo make_geo(int m, int n) {
    Geo g;
    g.m = m;
    g.n = n;
    g.nib = cdiv(m, TR);
    const int njb = cdiv(n, TC);
    int reps = 1;
    while (reps < 8 && (long long)g.nib * cdiv(njb, reps * 2) >= 4096) reps *= 2;
    g.reps = reps;
    g.njg = cdiv(njb, reps);
    return g;
}

// problem constants shared by all passes
struct Prob {
    int cls2;            // 0: class 1 (box prox), 1: class 2 (max(0,.), phi row, tails)
    int m, n;
    const double* p;
    const double* q;
    const double* c;
    const double* phi;   // class 2
    const double* gama;  // class 1, NULL -> gs
    double gs;
};

__device__ __forceinline__ double prox_of(const Prob& P, double x, double g) {
    // prox = @(x) min(max(0,x),gama) (Class1 :32) / max(0,x) (Class2 :29)
    const double t = x > 0.0 ? x : 0.0;
    return P.cls2 ? t : (t < g ? t : g);
}

// multiplier, optionally a line-search trial point lam + step*zeta (:189,200)
struct Lam {
    const double* base;
    const double* zeta;  // NULL -> base
    double step;
    __device__ __forceinline__ double at(int t) const {
        return zeta ? base[t] + step * zeta[t] : base[t];
    }
};

// ---------------------------------------------------------------------------
// tile walker
// ---------------------------------------------------------------------------
// 16 values per lane (one per column of a chunk) -> the 16 column sums over the wave's 64 rows.
// A butterfly reduce-scatter over lane bits 0..3 halves the live values at every step (8+4+2+1
// exchanges) and two more exchanges add the four 16-lane rows: 17 exchanges instead of 16 full
// wave reductions (tools/ubench_eval.hip: 5.8 -> 6.1 TB/s at m=n=4096, 2.2 -> 2.6 at 1024).
// On return lane l < 16 holds the sum of column colsum_index(l).
__device__ __forceinline__ double colsum16(const double (&xv)[TC], int lane) {
    double a8[8], a4[4], a2[2], a1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bool hi = lane & 1;
        a8[k] = (hi ? xv[k + 8] : xv[k]) + __shfl_xor(hi ? xv[k] : xv[k + 8], 1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool hi = lane & 2;
        a4[k] = (hi ? a8[k + 4] : a8[k]) + __shfl_xor(hi ? a8[k] : a8[k + 4], 2);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool hi = lane & 4;
        a2[k] = (hi ? a4[k + 2] : a4[k]) + __shfl_xor(hi ? a4[k] : a4[k + 2], 4);
    }
    {
        const bool hi = lane & 8;
        a1 = (hi ? a2[1] : a2[0]) + __shfl_xor(hi ? a2[0] : a2[1], 8);
    }
    a1 += __shfl_xor(a1, 16);
    a1 += __shfl_xor(a1, 32);
    return a1;
}
__device__ __forceinline__ int colsum_index(int lane) {
    return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
}

template <class Op>
__global__ __launch_bounds__(256) void k_tiles(const Op op, const Geo g,
                                               double* __restrict__ lpart,
                                               double* __restrict__ rpart,
                                               double* __restrict__ spart) {
    // Every wave owns 64 consecutive rows and walks its columns alone: row sums stay in a
    // register, the column sums over the wave's rows come out of a register butterfly -- no
    // LDS tile, no barrier inside the loop (the first version transposed 256x16 tiles through
    // LDS and stalled three times per tile: 2.0 TB/s at m=n=4096).
    __shared__ double red[4 * NSC];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int ib = blockIdx.x, jg = blockIdx.y;
    const int i = ib * TR + tid;
    const bool in_i = i < g.m;
    const int ic = in_i ? i : g.m - 1;
    const double pi = in_i ? op.P.p[i] : 0.0;
    typename Op::RowC rc;
    op.row_const(rc, ic);
    double sc[NSC];
#pragma unroll
    for (int k = 0; k < NSC; ++k) sc[k] = 0.0;
    double lacc = 0.0;
    double* const rrow = rpart + ((size_t)ib * 4 + wv) * g.n;
    for (int rep = 0; rep < g.reps; ++rep) {
        const int j0 = (jg * g.reps + rep) * TC;
        if (j0 >= g.n) break;  // uniform
        double xv[TC];
        // loads are issued Op::FC columns at a time: enough bytes in flight per lane without
        // spilling the ops that stream six or seven arrays
#pragma unroll
        for (int c0 = 0; c0 < TC; c0 += Op::FC) {
            typename Op::Raw raw[Op::FC];
            typename Op::ColC cc[Op::FC];
#pragma unroll
            for (int jj = 0; jj < Op::FC; ++jj) {
                const int j = min(j0 + c0 + jj, g.n - 1);
                op.fetch(raw[jj], (size_t)j * g.m + ic);
                // per-column uniforms (scalar loads) up front, with a clamped index: inside the
                // bounds check below they could not be hoisted and serialised every column
                op.col_const(cc[jj], j);
            }
#pragma unroll
            for (int jj = 0; jj < Op::FC; ++jj) {
                const int j = j0 + c0 + jj;
                double x = 0.0;
                if (in_i && j < g.n) {
                    x = op.compute(raw[jj], rc, cc[jj], (size_t)j * g.m + i, sc);
                    lacc += x * cc[jj].q;
                }
                xv[c0 + jj] = x * pi;
            }
        }
        if (Op::NEED_AX) {
            const double cs = colsum16(xv, lane);
            const int col = j0 + colsum_index(lane);
            if (lane < 16 && col < g.n) rrow[col] = cs;
        }
    }
    if (Op::NEED_AX && in_i) lpart[(size_t)jg * g.m + i] = lacc;
#pragma unroll
    for (int k = 0; k < NSC; ++k) {
        const double w = wave_sum(sc[k]);
        if (lane == 0) red[wv * NSC + k] = w;
    }
    __syncthreads();
    if (tid < NSC) {
        const int blk = jg * gridDim.x + ib;
        spart[(size_t)blk * NSC + tid] =
            red[tid] + red[NSC + tid] + red[2 * NSC + tid] + red[3 * NSC + tid];
    }
}

// --- APD_SsN_Class1.m:125 / Class2 :119  wk = -wc + bk*(uk+ak*vk)/ak^2 ; partials of H*uk
struct OpBegin {
    static constexpr bool NEED_AX = true;
    static constexpr int FC = 16;
    Prob P;
    const double* u;
    const double* v;
    double* w;
    double ak, bk, ak2;
    struct Raw {
        double c, u, v, phi;
    };
    struct RowC {};
    struct ColC {
        double q;
    };
    __device__ void row_const(RowC&, int) const {}
    __device__ void col_const(ColC& c, int j) const { c.q = P.q[j]; }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.c = P.c[idx];
        r.u = u[idx];
        r.v = v[idx];
        r.phi = P.cls2 ? P.phi[idx] : 0.0;
    }
    __device__ double compute(const Raw& r, const RowC&, const ColC&, size_t idx, double* sc) const {
        w[idx] = -r.c + bk * (r.u + ak * r.v) / ak2;
        sc[3] += r.phi * r.u;
        return r.u;
    }
};

// --- :139-144,182-196  zk, s, prox(zk) and its reductions at one multiplier.
// Compile-time variants (class 2 / vector gama / the prob-3 merit): the pass is HBM-bound only
// if the per-entry state stays small -- a runtime-flagged version kept 48 loads and three extra
// accumulators alive per lane and ran at a third of the bandwidth.
template <bool CLS2, bool GVEC, bool M3>
struct OpEvalT {
    static constexpr bool NEED_AX = true;
    static constexpr int FC = 16;
    Prob P;
    const double* w;
    uint8_t* s;
    Lam lam;
    double itk;  // 1/tk
    struct Raw {
        double w, phi, g;
    };
    struct RowC {
        double pi, y2, lamL;
    };
    struct ColC {
        double y1, q;
    };
    __device__ void row_const(RowC& rc, int i) const {
        rc.pi = P.p[i];
        rc.y2 = lam.at(P.n + i);
        rc.lamL = CLS2 ? lam.at(P.m + P.n) : 0.0;
    }
    __device__ void col_const(ColC& c, int j) const {
        c.y1 = lam.at(j);
        c.q = P.q[j];
    }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.w = w[idx];
        if (CLS2) r.phi = P.phi[idx];
        if (GVEC) r.g = P.gama[idx];
    }
    __device__ double compute(const Raw& r, const RowC& rc, const ColC& cc, size_t idx,
                              double* sc) const {
        double aty = rc.pi * cc.y1 + rc.y2 * cc.q;                  // Aty.m:12-13
        if (CLS2) aty = aty + rc.lamL * r.phi;                      // Class2 :139
        const double z = itk * (r.w - aty);
        const double g = GVEC ? r.g : P.gs;
        const double t = z > 0.0 ? z : 0.0;
        const double px = CLS2 ? t : (t < g ? t : g);               // prox (Class1 :32, Class2 :29)
        const bool act = CLS2 ? (z >= 0.0) : (z >= 0.0 && z <= g);
        s[idx] = act ? 1 : 0;
        sc[0] += px * px;
        if (M3) {
            sc[1] += z * z;
            sc[2] += (z - px) * (z - px);
        }
        if (CLS2) sc[3] += r.phi * px;
        sc[4] += act ? 1.0 : 0.0;
        return px;
    }
};

// --- :199-207  |prox(zk)|^2 at MK trial multipliers lk_old + step[k]*zeta in one pass over wk.
// A rejected Armijo test is followed by dozens of further trials (delta = 0.9; measured mean 57
// when the first one fails): they only need the merit, not Fk or the mask, so MK of them share
// one read of wk.  Per trial the arithmetic is that of OpEval, operation for operation.
struct OpMerit {
    static constexpr bool NEED_AX = false;
    static constexpr int FC = 16;
    Prob P;
    const double* w;
    const double* lam;
    const double* zeta;
    double step[MK];
    double itk;
    struct Raw {
        double w, phi, g;
    };
    struct RowC {
        double pi, l2, z2;
    };
    struct ColC {
        double l1, z1, q;
    };
    __device__ void row_const(RowC& rc, int i) const {
        rc.pi = P.p[i];
        rc.l2 = lam[P.n + i];
        rc.z2 = zeta[P.n + i];
    }
    __device__ void col_const(ColC& c, int j) const {
        c.l1 = lam[j];
        c.z1 = zeta[j];
        c.q = P.q[j];
    }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.w = w[idx];
        r.phi = P.cls2 ? P.phi[idx] : 0.0;
        r.g = P.gama ? P.gama[idx] : P.gs;
    }
    __device__ double compute(const Raw& r, const RowC& rc, const ColC& cc, size_t, double* sc) const {
        const double l1 = cc.l1, z1 = cc.z1, qj = cc.q;
        const int M = P.m + P.n;
        const double lL = P.cls2 ? lam[M] : 0.0, zL = P.cls2 ? zeta[M] : 0.0;
#pragma unroll
        for (int k = 0; k < MK; ++k) {
            double aty = rc.pi * (l1 + step[k] * z1) + (rc.l2 + step[k] * rc.z2) * qj;
            if (P.cls2) aty = aty + (lL + step[k] * zL) * r.phi;
            const double z = itk * (r.w - aty);
            const double px = prox_of(P, z, r.g);
            sc[k] += px * px;
        }
        return 0.0;
    }
};

// --- :239-242,253-254  uk1 = prox(zk), vk1, and the KKT residuals of (uk1, lk1)
template <bool FROM_W>
struct OpEnd {
    static constexpr bool NEED_AX = true;
    static constexpr int FC = 8;
    Prob P;
    const double* w;      // FROM_W
    const double* uold;   // FROM_W: current iterate ; else: the iterate to measure
    double* unew;         // FROM_W
    double* v;            // FROM_W
    Lam lam;
    double itk, ak;
    struct Raw {
        double w, u, c, phi, g;
    };
    struct RowC {
        double pi, y2;
    };
    struct ColC {
        double y1, q;
    };
    __device__ void row_const(RowC& rc, int i) const {
        rc.pi = P.p[i];
        rc.y2 = lam.at(P.n + i);
    }
    __device__ void col_const(ColC& c, int j) const {
        c.y1 = lam.at(j);
        c.q = P.q[j];
    }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.w = FROM_W ? w[idx] : 0.0;
        r.u = uold[idx];
        r.c = P.c[idx];
        r.phi = P.cls2 ? P.phi[idx] : 0.0;
        r.g = P.gama ? P.gama[idx] : P.gs;
    }
    __device__ double compute(const Raw& r, const RowC& rc, const ColC& cc, size_t idx,
                              double* sc) const {
        double aty = rc.pi * cc.y1 + rc.y2 * cc.q;
        if (P.cls2) aty = aty + lam.at(P.m + P.n) * r.phi;
        double u1 = r.u;
        if (FROM_W) {
            const double z = itk * (r.w - aty);
            u1 = prox_of(P, z, r.g);
            unew[idx] = u1;
            v[idx] = u1 + (u1 - r.u) / ak;
        }
        const double d = u1 - prox_of(P, u1 - r.c - aty, r.g);       // :242 / Class2 :227
        sc[0] += d * d;
        sc[1] += r.c * u1;
        sc[3] += r.phi * u1;
        return u1;
    }
};

// ---------------------------------------------------------------------------
// epilogues (one workgroup of 1024 threads)
// ---------------------------------------------------------------------------
struct Parts {
    Geo g;
    const double* lpart;
    const double* rpart;
    const double* spart;
    int nblk;
};

// sum of the partials of one entry of A*x, in a fixed order; loads go out 32 at a time (a
// plain loop pays one L2/HBM round trip per partial: 64 of them made this epilogue 47 us)
__device__ __forceinline__ double sum_strided(const double* __restrict__ base, int count,
                                              size_t stride) {
    double s = 0.0;
    int k = 0;
    for (; k + 32 <= count; k += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = base[(size_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
    }
    for (; k + 8 <= count; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < count; ++k) s += base[(size_t)k * stride];
    return s;
}

__device__ __forceinline__ double ax_entry(const Parts& pt, int t) {
    if (t < pt.g.n) return sum_strided(pt.rpart + t, 4 * pt.g.nib, (size_t)pt.g.n);
    return sum_strided(pt.lpart + (t - pt.g.n), pt.g.njg, (size_t)pt.g.m);
}

__device__ __forceinline__ double scal_total(const Parts& pt, int slot, double* red) {
    double a = 0.0;
    for (int b = threadIdx.x; b < pt.nblk; b += BT) a += pt.spart[(size_t)b * NSC + slot];
    return block_sum(a, red);
}

// device-side scalar results (read back through pinned memory)
struct ApdScal {
    double normF2, lam2, wlk_lam, prox2, z2, zmp2, count, fold_zeta;  // eval
    double kx2, ky2, kz2, kl2, fx;                                     // end
};

struct BeginFin {
    Parts pt;
    Prob P;
    const double* u;
    const double* v;
    double* w;
    const double* lam;
    const double* b;
    double* wlk;
    double ak, bk, ak2, bk1, ibk, mu;
};

__global__ __launch_bounds__(BT) void k_begin_fin(const BeginFin a) {
    __shared__ double red[16];
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const double phix = a.P.cls2 ? scal_total(a.pt, 3, red) : 0.0;
    for (int t = threadIdx.x; t < M; t += BT) {
        double Hu = ax_entry(a.pt, t);
        if (a.P.cls2) {
            Hu = Hu + a.u[mn + t];                                         // Ax(xk)+[yk;zk]
            a.w[mn + t] = -0.0 + a.bk * (a.u[mn + t] + a.ak * a.v[mn + t]) / a.ak2;
        }
        a.wlk[t] = a.bk1 * (a.lam[t] - a.ibk * (Hu - a.b[t])) - a.b[t];   // :126 / Class2 :120
    }
    if (a.P.cls2 && threadIdx.x == 0)
        a.wlk[M] = a.bk1 * (a.lam[M] - a.ibk * (phix - a.b[M])) - a.b[M];
}

struct EvalFin {
    Parts pt;
    Prob P;
    const double* w;
    Lam lam;
    const double* wlk;
    const double* Fold;  // with lam.zeta: Fk_old for ress = |Fk_old'*zeta| (:198)
    double* lam_out;
    double* F;
    double* tmask;       // class 2: t = zk(mn+1:end) >= 0 as 0/1 doubles (diag of T)
    double bk1, itk;
    ApdScal* out;
    double* fpart;   // per-block partial sums of the epilogue
    int* counter;    // ticket of the epilogue blocks (0 between launches)
};

// The epilogue of the evaluation pass runs once per line-search trial, so it is spread over
// cdiv(L, 128) small workgroups (one entry of Fk per thread: its 64+ partial sums arrive in two
// round trips instead of one workgroup walking 0.5 MB -- measured 36 us -> see DESIGN.md);
// the last workgroup to finish (ticket) adds the per-block scalars in block order.
constexpr int EB = 128;
constexpr int NFS = 6;   // f2, l2, wl, fz, tp2, tz2

__device__ __forceinline__ double sum128(double v, double* red) {   // result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1];
}
__device__ __forceinline__ double scal_total128(const Parts& pt, int slot, double* red) {
    double a = 0.0;
    for (int b = threadIdx.x; b < pt.nblk; b += EB) a += pt.spart[(size_t)b * NSC + slot];
    return sum128(a, red);
}

__global__ __launch_bounds__(EB) void k_eval_fin(const EvalFin a) {
    __shared__ double red[2];
    __shared__ int is_last;
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const int L = M + (a.P.cls2 ? 1 : 0);
    const int t = blockIdx.x * EB + threadIdx.x;
    // phi'*prox(zk) is needed by the one thread that owns the last entry of Fk (class 2)
    double phix = 0.0;
    if (a.P.cls2 && blockIdx.x == M / EB) phix = scal_total128(a.pt, 3, red);
    double f2 = 0.0, l2 = 0.0, wl = 0.0, fz = 0.0, tp2 = 0.0, tz2 = 0.0;
    if (t < L) {
        const double lt = a.lam.at(t);
        double Hp;
        if (t < M) {
            Hp = ax_entry(a.pt, t);
            if (a.P.cls2) {
                const double z = a.itk * (a.w[mn + t] - lt);               // Htlk tail = lk(1:m+n)
                const double pz = z > 0.0 ? z : 0.0;
                if (a.tmask) a.tmask[t] = z >= 0.0 ? 1.0 : 0.0;
                Hp = Hp + pz;                                              // Class2 :141
                tp2 = pz * pz;
                tz2 = z * z;
            }
        } else {
            Hp = phix;
        }
        const double f = a.bk1 * lt - Hp - a.wlk[t];                       // :144
        a.F[t] = f;
        if (a.lam_out) a.lam_out[t] = lt;
        f2 = f * f;
        l2 = lt * lt;
        wl = a.wlk[t] * lt;
        if (a.lam.zeta && a.Fold) fz = a.Fold[t] * a.lam.zeta[t];
    }
    double vals[NFS] = {f2, l2, wl, fz, tp2, tz2};
#pragma unroll
    for (int k = 0; k < NFS; ++k) {
        const double sk = sum128(vals[k], red);
        if (threadIdx.x == 0) a.fpart[(size_t)blockIdx.x * NFS + k] = sk;
    }
    __threadfence();
    if (threadIdx.x == 0) is_last = (atomicAdd(a.counter, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const double prox2 = scal_total128(a.pt, 0, red);
    const double z2s = scal_total128(a.pt, 1, red);
    const double zmp2 = scal_total128(a.pt, 2, red);
    const double cnt = scal_total128(a.pt, 4, red);
    // the partials arrive in one burst (one thread walking them was a chain of dependent loads); they
    // are still added in block order
    __shared__ double fp_lds[4 * EB];
    const int nfp = (int)gridDim.x * NFS;
    for (int e = threadIdx.x; e < nfp && e < 4 * EB; e += EB) fp_lds[e] = a.fpart[e];
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[NFS];
        for (int k = 0; k < NFS; ++k) tot[k] = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b)
            for (int k = 0; k < NFS; ++k)
                tot[k] += ((int)(b * NFS + k) < 4 * EB) ? fp_lds[b * NFS + k] : a.fpart[(size_t)b * NFS + k];
        a.out->normF2 = tot[0];
        a.out->lam2 = tot[1];
        a.out->wlk_lam = tot[2];
        a.out->fold_zeta = tot[3];
        a.out->prox2 = prox2 + tot[4];
        a.out->z2 = z2s + tot[5];
        a.out->zmp2 = zmp2;  // tails: z - prox(z) = min(z,0); only class 1 (prob 3) uses it
        a.out->count = cnt;
        *a.counter = 0;
    }
}

struct EndFin {
    Parts pt;
    Prob P;
    int from_w;
    const double* w;
    const double* uold;
    double* unew;
    double* v;
    Lam lam;
    const double* b;
    double itk, ak;
    ApdScal* out;
};

__global__ __launch_bounds__(BT) void k_end_fin(const EndFin a) {
    __shared__ double red[16];
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const double kx2 = scal_total(a.pt, 0, red);
    const double fx = scal_total(a.pt, 1, red);
    const double phix = scal_total(a.pt, 3, red);
    double ky2 = 0.0, kz2 = 0.0, kl2 = 0.0;
    for (int t = threadIdx.x; t < M; t += BT) {
        double Hu = ax_entry(a.pt, t);
        if (a.P.cls2) {
            const double lt = a.lam.at(t);
            double u1 = a.uold[mn + t];
            if (a.from_w) {
                const double z = a.itk * (a.w[mn + t] - lt);
                const double pz = z > 0.0 ? z : 0.0;
                a.unew[mn + t] = pz;
                a.v[mn + t] = pz + (pz - u1) / a.ak;
                u1 = pz;
            }
            Hu = Hu + u1;
            const double sh = u1 - lt;
            const double d = u1 - (sh > 0.0 ? sh : 0.0);                   // Class2 :225-226
            if (t < a.P.n)
                ky2 += d * d;
            else
                kz2 += d * d;
        }
        const double e = Hu - a.b[t];
        kl2 += e * e;
    }
    ky2 = block_sum(ky2, red);
    kz2 = block_sum(kz2, red);
    kl2 = block_sum(kl2, red);
    if (threadIdx.x == 0) {
        if (a.P.cls2) {
            const double e = phix - a.b[M];
            kl2 += e * e;
        }
        a.out->kx2 = kx2;
        a.out->ky2 = ky2;
        a.out->kz2 = kz2;
        a.out->kl2 = kl2;
        a.out->fx = fx;
    }
}

struct MeritFin {
    Parts pt;
    Prob P;
    const double* w;
    const double* lam;
    const double* zeta;
    const double* wlk;
    double step[MK];
    double bk1, tk, itk;
    double* out;   // MK merit values cFk_new
};

__global__ __launch_bounds__(BT) void k_merit_fin(const MeritFin a) {
    // All MK trial points in ONE pass over lam, zeta, wlk and ONE exchange: the per-thread sums, the
    // wave sums and the order in which the waves' sums are added are those of block_sum / scal_total
    // value by value (same bits); MK sequential passes with four block sums each took 16.5 us.
    __shared__ double redm[BT / 64][4 * MK];
    __shared__ double tot[4 * MK];
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const int L = M + (a.P.cls2 ? 1 : 0);
    double acc[4 * MK];
#pragma unroll
    for (int k = 0; k < 4 * MK; ++k) acc[k] = 0.0;
    for (int t = threadIdx.x; t < L; t += BT) {
        const double l0 = a.lam[t], zt = a.zeta[t], wlt = a.wlk[t];
        const bool tail = a.P.cls2 && t < M;
        const double wt = tail ? a.w[mn + t] : 0.0;
#pragma unroll
        for (int k = 0; k < MK; ++k) {
            const double lt = l0 + a.step[k] * zt;
            acc[k] += lt * lt;
            acc[MK + k] += wlt * lt;
            if (tail) {
                const double z = a.itk * (wt - lt);
                const double pz = z > 0.0 ? z : 0.0;
                acc[2 * MK + k] += pz * pz;
            }
        }
    }
    for (int b = threadIdx.x; b < a.pt.nblk; b += BT) {
#pragma unroll
        for (int k = 0; k < MK; ++k) acc[3 * MK + k] += a.pt.spart[(size_t)b * NSC + k];
    }
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4 * MK; ++k) {
        const double v = wave_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) redm[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4 * MK) {
        double sgm = 0.0;
#pragma unroll
        for (int w = 0; w < BT / 64; ++w) sgm += redm[w][threadIdx.x];
        tot[threadIdx.x] = sgm;
    }
    __syncthreads();
    if (threadIdx.x < MK) {
        const int k = threadIdx.x;
        const double l2 = tot[k], wl = tot[MK + k], tp2 = tot[2 * MK + k];
        const double prox2 = tot[3 * MK + k] + tp2;
        a.out[k] = a.bk1 / 2.0 * l2 - wl + 0.5 * a.tk * prox2;   // :201-204
    }
}

// Row f3: is this Newton step's system the previous one's?  Compares the active-set mask (and,
// class 2, the 0/1 diagonal of T) with the copies kept from the last step, refreshes the copies
// and raises *changed on any difference.  8-byte words; the caller pads the mask to a multiple.
__global__ void k_words_changed(size_t nw, const unsigned long long* __restrict__ cur,
                                unsigned long long* __restrict__ prev, int* __restrict__ changed) {
    bool diff = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nw;
         i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long c = cur[i];
        if (c != prev[i]) {
            diff = true;
            prev[i] = c;
        }
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

__global__ void k_negate(int n, const double* __restrict__ x, double* __restrict__ y) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        y[i] = -x[i];
}

// ---------------------------------------------------------------------------
// warm start: warmup_class1.m:57-77 / warmup_class2.m:57-85, two passes per iteration
// ---------------------------------------------------------------------------
struct WarmScal {
    double ak, gk, muf, etafk, sgk, ibk, akbk, ak2, tt_etafk /* etafk+tt */, prox_scale /* ak^2/etagk */;
    double gkmu;  // gk + muf*ak
    double iak1;  // 1+ak
};

// pass A: dd = etafk*wuk - ak^2*(wc + cAlk + sgk*cAw); partials of H*dd
struct OpWarmA {
    static constexpr bool NEED_AX = true;
    static constexpr int FC = 4;
    Prob P;
    const double* x;
    const double* v;
    const double* wk;
    const double* pik;
    const double* l2;     // multiplier block of the splitting constraint u = w
    const double* hl1;    // hlk(1:L)
    const double* b;
    double* dd;
    WarmScal s;
    struct Raw {
        double x, v, w, pi, l2, c, phi;
    };
    struct RowC {
        double pi, h2, b2;
    };
    struct ColC {
        double bj, hj, q;
    };
    __device__ void row_const(RowC& rc, int i) const {
        rc.pi = P.p[i];
        rc.h2 = hl1[P.n + i];
        rc.b2 = b[P.n + i];
    }
    __device__ void col_const(ColC& c, int j) const {
        c.bj = b[j];
        c.hj = hl1[j];
        c.q = P.q[j];
    }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.x = x[idx];
        r.v = v[idx];
        r.w = wk[idx];
        r.pi = pik[idx];
        r.l2 = l2[idx];
        r.c = P.c[idx];
        r.phi = P.cls2 ? P.phi[idx] : 0.0;
    }
    __device__ double compute(const Raw& r, const RowC& rc, const ColC& cc, size_t idx,
                              double* sc) const {
        const double wux = (s.ak * s.gk * r.v + s.gkmu * r.x) / s.etafk;              // :63
        const double hl2 = r.l2 - s.ibk * (r.x - r.w) + s.akbk * (-(r.pi - r.w));     // :65
        double atb = rc.pi * cc.bj + rc.b2 * cc.q;                                     // Atb
        double aty = rc.pi * cc.hj + rc.h2 * cc.q;
        const int M = P.m + P.n;
        double cAlk;
        if (P.cls2) {
            atb = atb + b[M] * r.phi;                                                  // Htb
            cAlk = hl2 + (aty + hl1[M] * r.phi);                                       // Class2 :66-67
        } else {
            cAlk = aty + hl2;                                                          // Class1 :66
        }
        const double cAw = -atb - r.w;
        const double d = s.etafk * wux - s.ak2 * (r.c + cAlk + s.sgk * cAw);          // :67
        dd[idx] = d;
        sc[3] += r.phi * d;
        return d;
    }
};

// pass B: xk1, vk1, wk1, pik1 and the splitting multiplier; partials of H*xk1.  H*vk1 follows in
// the epilogue from linearity, H*vk1 = H*xk1 + (H*xk1 - H*xk)/ak, without another pass.
struct OpWarmB {
    static constexpr bool NEED_AX = true;
    static constexpr int FC = 4;
    Prob P;
    const double* dd;
    const double* ff;     // invAAt / invHHt result (L entries)
    double* x;
    double* v;
    double* wk;
    double* pik;
    double* l2;
    WarmScal s;
    struct Raw {
        double d, x, w, pi, l2, phi, g;
    };
    struct RowC {
        double pi, f2;
    };
    struct ColC {
        double fj, q;
    };
    __device__ void row_const(RowC& rc, int i) const {
        rc.pi = P.p[i];
        rc.f2 = ff[P.n + i];
    }
    __device__ void col_const(ColC& c, int j) const {
        c.fj = ff[j];
        c.q = P.q[j];
    }
    __device__ void fetch(Raw& r, size_t idx) const {
        r.d = dd[idx];
        r.x = x[idx];
        r.w = wk[idx];
        r.pi = pik[idx];
        r.l2 = l2[idx];
        r.phi = P.cls2 ? P.phi[idx] : 0.0;
        r.g = P.gama ? P.gama[idx] : P.gs;
    }
    __device__ double compute(const Raw& r, const RowC& rc, const ColC& cc, size_t idx,
                              double* sc) const {
        double aty = rc.pi * cc.fj + rc.f2 * cc.q;
        if (P.cls2) aty = aty + ff[P.m + P.n] * r.phi;
        const double x1 = (r.d - aty) / s.tt_etafk;                                    // :70
        const double v1 = x1 + (x1 - r.x) / s.ak;                                      // :71
        const double wwk = (s.ak * r.pi + r.w) / s.iak1;                               // :62
        const double bl2 = r.l2 + s.akbk * (v1 - r.pi);                                // :72
        const double w1 = prox_of(P, wwk - s.prox_scale * (-bl2), r.g);                // :73
        const double pi1 = w1 + (w1 - r.w) / s.ak;                                     // :74
        const double l21 = r.l2 + s.akbk * (v1 - pi1);                                 // :75
        x[idx] = x1;
        v[idx] = v1;
        wk[idx] = w1;
        pik[idx] = pi1;
        l2[idx] = l21;
        sc[3] += r.phi * x1;
        return x1;
    }
};

// epilogue 1 of the warm start: Hdd = [Ax(dd_x)+dd_tail ; phi'dd_x], tails of dd
struct WarmFinA {
    Parts pt;
    Prob P;
    const double* x;
    const double* v;
    const double* wk;
    const double* pik;
    const double* l2;
    const double* hl1;
    const double* b;
    double* dd;
    double* Hdd;
    WarmScal s;
};

__global__ __launch_bounds__(BT) void k_warm_fin_a(const WarmFinA a) {
    __shared__ double red[16];
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const double phid = a.P.cls2 ? scal_total(a.pt, 3, red) : 0.0;
    const WarmScal& s = a.s;
    for (int t = threadIdx.x; t < M; t += BT) {
        double h = ax_entry(a.pt, t);
        if (a.P.cls2) {
            const size_t id = mn + t;
            const double wux = (s.ak * s.gk * a.v[id] + s.gkmu * a.x[id]) / s.etafk;
            const double hl2 = a.l2[id] - s.ibk * (a.x[id] - a.wk[id]) +
                               s.akbk * (-(a.pik[id] - a.wk[id]));
            const double cAlk = hl2 + a.hl1[t];          // [Aty(..)+..*phi ; hlk(1:n+m)] tail
            const double cAw = -a.b[t] - a.wk[id];       // Htb tail = b(1:n+m)
            const double d = s.etafk * wux - s.ak2 * (0.0 + cAlk + s.sgk * cAw);
            a.dd[id] = d;
            h = h + d;                                   // Ax(dd_x) + dd(mn+1:end)
        }
        a.Hdd[t] = h;
    }
    if (a.P.cls2 && threadIdx.x == 0) a.Hdd[M] = phid;
}

// hlk(1:L) = lk1 - 1/bk*(H*uk - b)        (the z0 block of :65 adds nothing)
__global__ __launch_bounds__(BT) void k_warm_hl1(int L, const double* __restrict__ lk1,
                                                 const double* __restrict__ Hu,
                                                 const double* __restrict__ b, double ibk,
                                                 double akbk, double* __restrict__ hl1) {
    for (int t = threadIdx.x; t < L; t += BT)
        hl1[t] = lk1[t] - ibk * (Hu[t] - b[t]) + akbk * 0.0;
}

// epilogue 2: tails of pass B, H*xk1, H*vk1 and lk1(1:L) += ak/bk*(H*vk1 - b)
struct WarmFinB {
    Parts px;   // partials of xk1
    Prob P;
    const double* dd;
    const double* ff;
    double* x;
    double* v;
    double* wk;
    double* pik;
    double* l2;
    const double* b;
    double* lk1;
    double* Hx;
    WarmScal s;
};

__global__ __launch_bounds__(BT) void k_warm_fin_b(const WarmFinB a) {
    __shared__ double red[16];
    const int M = a.P.m + a.P.n;
    const size_t mn = (size_t)a.P.m * a.P.n;
    const double phix = a.P.cls2 ? scal_total(a.px, 3, red) : 0.0;
    const WarmScal& s = a.s;
    for (int t = threadIdx.x; t < M; t += BT) {
        double hx = ax_entry(a.px, t);
        if (a.P.cls2) {
            const size_t id = mn + t;
            const double x0 = a.x[id], w0 = a.wk[id], pi0 = a.pik[id], l20 = a.l2[id];
            const double x1 = (a.dd[id] - a.ff[t]) / s.tt_etafk;           // [..; ff(1:m+n)] tail
            const double v1 = x1 + (x1 - x0) / s.ak;
            const double wwk = (s.ak * pi0 + w0) / s.iak1;
            const double bl2 = l20 + s.akbk * (v1 - pi0);
            const double arg = wwk - s.prox_scale * (-bl2);
            const double w1 = arg > 0.0 ? arg : 0.0;
            const double pi1 = w1 + (w1 - w0) / s.ak;
            a.x[id] = x1;
            a.v[id] = v1;
            a.wk[id] = w1;
            a.pik[id] = pi1;
            a.l2[id] = l20 + s.akbk * (v1 - pi1);
            hx = hx + x1;
        }
        const double hv = hx + (hx - a.Hx[t]) / s.ak;
        a.Hx[t] = hx;
        a.lk1[t] = a.lk1[t] + s.akbk * (hv - a.b[t]);                      // :75
    }
    if (a.P.cls2 && threadIdx.x == 0) {
        const double phiv = phix + (phix - a.Hx[M]) / s.ak;
        a.Hx[M] = phix;
        a.lk1[M] = a.lk1[M] + s.akbk * (phiv - a.b[M]);
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct EvalRes {
    double normF = 0, lam2 = 0, wlk_lam = 0, prox2 = 0, z2 = 0, zmp2 = 0, fold_zeta = 0;
    long long E = 0;
};

struct ipd_apd {
    ipd_ctx* ctx = nullptr;
    std::unique_ptr<Arena> arena;
    int cls = 1, m = 0, n = 0, M = 0, L = 0;
    size_t mn = 0, U = 0;
    Prob P{};
    Geo geo{};
    double mu = 0.0;
    // problem data
    double *c = nullptr, *gama = nullptr, *phi = nullptr, *p = nullptr, *q = nullptr, *b = nullptr;
    // workspace
    double *u = nullptr, *u2 = nullptr, *v = nullptr, *w = nullptr;
    double *lam = nullptr, *lam_a = nullptr, *lam_b = nullptr, *wlk = nullptr;
    double *F_a = nullptr, *F_b = nullptr, *zeta = nullptr, *negF = nullptr, *tmask = nullptr;
    uint8_t* s = nullptr;
    double *lpart = nullptr, *rpart = nullptr, *spart = nullptr;
    ApdScal* dscal = nullptr;
    double* fpart = nullptr;
    int* counter = nullptr;
    double* merit = nullptr;
    bool merit3 = false;   // class 1, prob 3: the merit needs |zk|^2 and |zk - prox(zk)|^2 (:186)
    double *phi_l = nullptr, *phi_part = nullptr;  // Ax(phi), partial sums of |phi|^2
    int phi_npart = 0;
    // script variables
    int k = 0;
    double bk = 1.0;
    double ak = 0, bk1 = 0, tk = 0;  // of the iteration opened by begin()
    bool have_kkt = false;
    double kkt0[4] = {0, 0, 0, 0};   // KKT_xk(1), KKT_lk(1), KKT_yk(1), KKT_zk(1)
    double kkt[4] = {0, 0, 0, 0};
    double fval = 0.0;
    bool converged = false;
    std::vector<double> h_fx, h_kx, h_kl, h_ky, h_kz, h_ssn;
    std::vector<ipd_ssn_rec> recs;
    long long sum_amg = 0, total_amg = 0, fail_amg = 0, max_amg = 0;
    int restarts = 0;
    // row f3: hierarchies of the previous Newton step and what its system was built from
    StepDonors step;
    bool step_reuse = true, have_prev = false;
    double prev_bk1 = 0.0, prev_tk = 0.0;
    size_t s_words = 0;
    unsigned long long *s_prev = nullptr, *t_prev = nullptr;
    int* step_changed = nullptr;
    int nblk() const { return geo.nib * geo.njg; }
};

namespace {

Parts parts_of(const ipd_apd* h) {
    Parts pt;
    pt.g = h->geo;
    pt.lpart = h->lpart;
    pt.rpart = h->rpart;
    pt.spart = h->spart;
    pt.nblk = h->nblk();
    return pt;
}

template <class Op>
void launch_tiles(ipd_apd* h, const Op& op) {
    const Geo& g = h->geo;
    hipLaunchKernelGGL(k_tiles<Op>, dim3(g.nib, g.njg), dim3(256), 0, h->ctx->stream, op, g,
                       h->lpart, h->rpart, h->spart);
    IPD_KERNEL_CHECK();
}

ApdScal fetch_scal(ipd_apd* h) {
    ApdScal s;
    h->ctx->fetch_bytes(h->dscal, &s, sizeof(ApdScal));
    return s;
}

// wk, wlk of iteration k                                       Class1 :113-126, Class2 :110-120
void apd_begin(ipd_apd* h, int k) {
    ProfScope ps(h->ctx, PROF_BEGIN_END);
    const double kk = (double)k;
    h->ak = std::sqrt(kk * kk * h->bk);
    h->bk1 = h->bk / (1.0 + h->ak);
    h->tk = h->bk * (1.0 + h->ak) / (h->ak * h->ak);
    OpBegin op;
    op.P = h->P;
    op.u = h->u;
    op.v = h->v;
    op.w = h->w;
    op.ak = h->ak;
    op.bk = h->bk;
    op.ak2 = h->ak * h->ak;
    launch_tiles(h, op);
    BeginFin f;
    f.pt = parts_of(h);
    f.P = h->P;
    f.u = h->u;
    f.v = h->v;
    f.w = h->w;
    f.lam = h->lam;
    f.b = h->b;
    f.wlk = h->wlk;
    f.ak = h->ak;
    f.bk = h->bk;
    f.ak2 = h->ak * h->ak;
    f.bk1 = h->bk1;
    f.ibk = 1.0 / h->bk;
    f.mu = h->mu;
    hipLaunchKernelGGL(k_begin_fin, dim3(1), dim3(BT), 0, h->ctx->stream, f);
    IPD_KERNEL_CHECK();
}

template <bool CLS2, bool GVEC, bool M3>
void launch_eval_t(ipd_apd* h, const Lam& lam, double itk) {
    OpEvalT<CLS2, GVEC, M3> op;
    op.P = h->P;
    op.w = h->w;
    op.s = h->s;
    op.lam = lam;
    op.itk = itk;
    launch_tiles(h, op);
}
void launch_eval(ipd_apd* h, const Lam& lam, double itk) {
    if (h->cls == 2) return launch_eval_t<true, false, false>(h, lam, itk);
    const bool gv = h->gama != nullptr, m3 = h->merit3;
    if (gv)
        m3 ? launch_eval_t<false, true, true>(h, lam, itk) : launch_eval_t<false, true, false>(h, lam, itk);
    else
        m3 ? launch_eval_t<false, false, true>(h, lam, itk) : launch_eval_t<false, false, false>(h, lam, itk);
}

// one pass at lam_base (+ step*zeta): F -> F_out, multiplier -> lam_out, mask -> h->s
EvalRes apd_eval(ipd_apd* h, const double* lam_base, const double* zeta, double step,
                 double* lam_out, double* F_out, const double* F_old) {
    ProfScope ps(h->ctx, PROF_EVAL);
    const Lam lamv{lam_base, zeta, step};
    const double itk = 1.0 / h->tk;
    launch_eval(h, lamv, itk);
    EvalFin f;
    f.pt = parts_of(h);
    f.P = h->P;
    f.w = h->w;
    f.lam = lamv;
    f.wlk = h->wlk;
    f.Fold = F_old;
    f.lam_out = lam_out;
    f.F = F_out;
    f.tmask = h->tmask;
    f.bk1 = h->bk1;
    f.itk = itk;
    f.out = h->dscal;
    f.fpart = h->fpart;
    f.counter = h->counter;
    hipLaunchKernelGGL(k_eval_fin, dim3(cdiv(h->L, EB)), dim3(EB), 0, h->ctx->stream, f);
    IPD_KERNEL_CHECK();
    const ApdScal s = fetch_scal(h);
    EvalRes r;
    r.normF = std::sqrt(s.normF2);
    r.lam2 = s.lam2;
    r.wlk_lam = s.wlk_lam;
    r.prox2 = s.prox2;
    r.z2 = s.z2;
    r.zmp2 = s.zmp2;
    r.fold_zeta = s.fold_zeta;
    r.E = (long long)(s.count + 0.5);
    return r;
}

// merit cFk at lam_base + step[k]*zeta for MK steps (class 1 prob < 3 and class 2)
void apd_merit(ipd_apd* h, const double* lam_base, const double* zeta, const double step[MK],
               double merit[MK]) {
    ProfScope ps(h->ctx, PROF_EVAL);
    OpMerit op;
    op.P = h->P;
    op.w = h->w;
    op.lam = lam_base;
    op.zeta = zeta;
    op.itk = 1.0 / h->tk;
    MeritFin f;
    f.pt = parts_of(h);
    f.P = h->P;
    f.w = h->w;
    f.lam = lam_base;
    f.zeta = zeta;
    f.wlk = h->wlk;
    f.bk1 = h->bk1;
    f.tk = h->tk;
    f.itk = op.itk;
    f.out = h->merit;
    for (int k = 0; k < MK; ++k) op.step[k] = f.step[k] = step[k];
    launch_tiles(h, op);
    hipLaunchKernelGGL(k_merit_fin, dim3(1), dim3(BT), 0, h->ctx->stream, f);
    IPD_KERNEL_CHECK();
    h->ctx->fetch(h->merit, merit, MK);
}

// uk1/vk1 (from_w) and the KKT residuals of the iterate at multiplier `lam`
void apd_end(ipd_apd* h, bool from_w, const double* src_u, const double* lam, double out_kkt[4],
             double* fx) {
    ProfScope ps(h->ctx, PROF_BEGIN_END);
    const Lam L{lam, nullptr, 0.0};
    if (from_w) {
        OpEnd<true> op;
        op.P = h->P;
        op.w = h->w;
        op.uold = h->u;
        op.unew = h->u2;
        op.v = h->v;
        op.lam = L;
        op.itk = 1.0 / h->tk;
        op.ak = h->ak;
        launch_tiles(h, op);
    } else {
        OpEnd<false> op;
        op.P = h->P;
        op.w = nullptr;
        op.uold = src_u;
        op.unew = nullptr;
        op.v = nullptr;
        op.lam = L;
        op.itk = 0.0;
        op.ak = 1.0;
        launch_tiles(h, op);
    }
    EndFin f;
    f.pt = parts_of(h);
    f.P = h->P;
    f.from_w = from_w ? 1 : 0;
    f.w = h->w;
    f.uold = from_w ? h->u : src_u;
    f.unew = h->u2;
    f.v = h->v;
    f.lam = L;
    f.b = h->b;
    f.itk = from_w ? 1.0 / h->tk : 0.0;
    f.ak = from_w ? h->ak : 1.0;
    f.out = h->dscal;
    hipLaunchKernelGGL(k_end_fin, dim3(1), dim3(BT), 0, h->ctx->stream, f);
    IPD_KERNEL_CHECK();
    const ApdScal s = fetch_scal(h);
    out_kkt[0] = std::sqrt(s.kx2);
    out_kkt[1] = std::sqrt(s.kl2);
    out_kkt[2] = std::sqrt(s.ky2);
    out_kkt[3] = std::sqrt(s.kz2);
    *fx = s.fx;
}

double max_rr(const ipd_apd* h, const double kk[4]) {
    double r = std::max(kk[0] / (1.0 + h->kkt0[0]), kk[1] / (1.0 + h->kkt0[1]));
    if (h->cls == 2) {
        r = std::max(r, kk[2] / (1.0 + h->kkt0[2]));
        r = std::max(r, kk[3] / (1.0 + h->kkt0[3]));
    }
    return r;
}

void push_hist(ipd_apd* h) {
    h->h_fx.push_back(h->fval);
    h->h_kx.push_back(h->kkt[0]);
    h->h_kl.push_back(h->kkt[1]);
    h->h_ky.push_back(h->kkt[2]);
    h->h_kz.push_back(h->kkt[3]);
}

void ensure_kkt(ipd_apd* h) {
    if (h->have_kkt) return;
    apd_end(h, false, h->u, h->lam, h->kkt, &h->fval);   // Class1 :63-65, Class2 :44-48
    for (int i = 0; i < 4; ++i) h->kkt0[i] = h->kkt[i];
    h->h_fx.clear();
    h->h_kx.clear();
    h->h_kl.clear();
    h->h_ky.clear();
    h->h_kz.clear();
    h->h_ssn.clear();
    push_hist(h);
    h->have_kkt = true;
}

void copy_dev(ipd_apd* h, double* dst, const double* src, size_t n) {
    IPD_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->ctx->stream));
}

// Row f3: true when the system of this Newton step (active set, class 2's T, bk1, tk; p and q
// never change) is the one the previous step's hierarchies were built for.
bool step_same_system(ipd_apd* h, double bk1, double tk) {
    ipd_ctx* ctx = h->ctx;
    IPD_HIP(hipMemsetAsync(h->step_changed, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_words_changed, dim3(std::max(1, std::min(cdiv((int)std::min<size_t>(h->s_words, 1u << 30), 256), 1024))),
                       dim3(256), 0, ctx->stream, h->s_words,
                       reinterpret_cast<const unsigned long long*>(h->s), h->s_prev, h->step_changed);
    if (h->tmask)
        hipLaunchKernelGGL(k_words_changed, dim3(cdiv(h->M, 256)), dim3(256), 0, ctx->stream,
                           (size_t)h->M, reinterpret_cast<const unsigned long long*>(h->tmask),
                           h->t_prev, h->step_changed);
    IPD_KERNEL_CHECK();
    const bool changed = ctx->fetch1(h->step_changed) != 0;
    const bool same = h->have_prev && !changed && bk1 == h->prev_bk1 && tk == h->prev_tk;
    h->have_prev = true;
    h->prev_bk1 = bk1;
    h->prev_tk = tk;
    return same;
}

// one APD iteration                                       Class1 :101-275, Class2 :95-285
void apd_iterate(ipd_apd* h, const ipd_apd_opts& o, const AmgOpts& amg, ipd_rng* rng) {
    const int solver = o.inner_solver;
    ipd_ctx* ctx = h->ctx;
    const int k = ++h->k;
    const bool c2 = h->cls == 2;
    double resk = std::max(h->kkt[0], h->kkt[1]);
    if (c2) resk = std::max(std::max(resk, h->kkt[2]), h->kkt[3]);
    apd_begin(h, k);
    const double bk1 = h->bk1, tk = h->tk;
    const double ssn_tol = std::max(bk1 / ((double)k * (double)k), o.ssn_tol1);   // :123
    const bool merit3 = !c2 && o.prob >= 3;
    h->merit3 = merit3;
    auto merit = [&](const EvalRes& e) {
        const double f0 = bk1 / 2.0 * e.lam2 - e.wlk_lam;                         // :182
        return merit3 ? f0 + 0.5 * tk * (e.z2 - e.zmp2) : f0 + 0.5 * tk * e.prox2;
    };
    // :128-131  lk_new = lk ; Fk_new
    double* lam_new = h->lam_a;
    double* lam_old = h->lam_b;
    double* F_new = h->F_a;
    double* F_old = h->F_b;
    EvalRes e_new = apd_eval(h, h->lam, nullptr, 0.0, lam_new, F_new, nullptr);
    int ssn_it = 0;
    long long pmin = std::numeric_limits<long long>::max(), pmax = 0, psum = 0;
    while (e_new.normF > ssn_tol) {                                               // :137
        ++ssn_it;
        std::swap(lam_new, lam_old);
        std::swap(F_new, F_old);
        const EvalRes e_old = e_new;   // same multiplier, same pass: s, Fk_old, cFk_old (:139-144)
        Csr H0;
        HybridOut ho;
        {
            CallScope scope(ctx);
            {
                ProfScope ps(ctx, PROF_ASAT);
                kkt_asat(ctx, *ctx->scratch, h->s, h->p, h->q, h->m, h->n, &H0);   // :142
            }
            hipLaunchKernelGGL(k_negate, dim3(cdiv(h->L, 256)), dim3(256), 0, ctx->stream, h->L,
                               (const double*)F_old, h->negF);                     // z = -Fk_old
            IPD_KERNEL_CHECK();
            StepDonors* step = nullptr;
            if (h->step_reuse && solver == 4) {
                step = &h->step;
                step->same = step_same_system(h, bk1, tk);
                ++step->steps;
                step->same_steps += step->same;
            } else {
                h->have_prev = false;
                h->step.prev.clear();
            }
            if (solver == 1) {                                     // :146-148  zeta = Jk \ (-Fk_old)
                if (c2) {                                          // Class2 :152-156
                    direct_pot_dev(ctx, H0, h->tmask, h->p, h->q, h->m, h->n, bk1, tk, h->negF, h->s,
                                   h->phi, h->zeta);
                } else {
                    Csr Jk;
                    build_jk(ctx, *ctx->scratch, H0, nullptr, bk1, tk, &Jk);
                    spd_solve_dev(ctx, Jk, h->negF, h->zeta);
                }
                ho.itamg = 1;                                      // itpcg = 1; respcg = 0; info = [0,0]
                ho.resamg = 0.0;
            } else if (solver == 2 && c2) {                        // Class2 :157-161  PCG on the bordered Jk
                pcg_pot_dev(ctx, H0, h->tmask, h->p, h->q, h->m, h->n, bk1, tk, h->negF, h->s, h->phi,
                            o.pcg_retol, o.pcg_maxit, h->zeta, &ho);
            } else if (solver == 2) {                              // :149-152  PCG(Jk,-Fk_old)
                Csr Jk;
                build_jk(ctx, *ctx->scratch, H0, nullptr, bk1, tk, &Jk);
                long long it2 = 0;
                double res2 = 0.0;
                pcg_dev(ctx, Jk, h->negF, nullptr, o.pcg_retol, o.pcg_maxit, 2, h->zeta, &it2, &res2,
                        nullptr);
                ho.itamg = (int)std::min<long long>(it2, 2147483647LL);
                ho.resamg = res2;
            } else if (solver == 3) {                              // :157-159 / Class2 :167-169
                if (c2)
                    pcg4pot_dev(ctx, H0, h->tmask, h->p, h->q, h->m, h->n, bk1, tk, h->negF, h->s,
                                h->phi, o.pcg_retol, o.pcg_maxit, h->zeta, &ho);
                else
                    aug_pcg_dev(ctx, H0, nullptr, h->p, h->q, h->m, h->n, bk1, tk, h->negF,
                                o.pcg_retol, o.pcg_maxit, h->zeta, &ho);
            } else if (c2) {                                       // 4: 'amg', 5: 'twogrid'
                amg4pot_dev(ctx, H0, h->tmask, h->p, h->q, h->m, h->n, bk1, tk, h->negF, h->s,
                            h->phi, amg, rng, h->zeta, &ho, step);                 // Class2 :171
            } else {
                hybrid_amg_dev(ctx, H0, nullptr, h->p, h->q, h->m, h->n, bk1, tk, h->negF, amg,
                               rng, h->zeta, &ho, step);                           // Class1 :161
            }
        }
        const int itpcg = ho.itamg;
        if (solver >= 4) {
            if (itpcg == amg.maxit)
                ++h->fail_amg;                                                     // :163-169
            else
                h->max_amg = std::max<long long>(h->max_amg, itpcg);
            if (itpcg > 0) ++h->total_amg;
        }
        pmin = std::min<long long>(pmin, itpcg);
        pmax = std::max<long long>(pmax, itpcg);
        psum += itpcg;
        // :182-211 line search
        const double cF_old = merit(e_old);
        int ll = 0;
        double step = 1.0;
        e_new = apd_eval(h, lam_old, h->zeta, step, lam_new, F_new, F_old);
        const double ress = std::fabs(e_new.fold_zeta);                            // :198
        if (merit(e_new) > cF_old - o.nu * step * ress && !merit3 && o.ll_max > 0) {      // :199
            // the trials that follow only decide on the merit: MK of them per pass over wk,
            // then one full evaluation at the accepted (or last) step
            bool found = false;
            while (!found && ll < o.ll_max) {
                double st[MK], mv[MK];
                for (int k = 0; k < MK; ++k) st[k] = std::pow(o.delta, (double)(ll + 1 + k));
                apd_merit(h, lam_old, h->zeta, st, mv);
                for (int k = 0; k < MK && !found; ++k) {
                    const int cand = ll + 1 + k;
                    if (cand > o.ll_max) break;
                    if (!(mv[k] > cF_old - o.nu * st[k] * ress) || cand == o.ll_max) {
                        ll = cand;
                        found = true;
                    }
                }
                if (!found) ll += MK;
            }
            ll = std::min(ll, (int)o.ll_max);
            step = std::pow(o.delta, (double)ll);
            e_new = apd_eval(h, lam_old, h->zeta, step, lam_new, F_new, F_old);
        } else {
            while (merit(e_new) > cF_old - o.nu * step * ress) {                   // :199
                ++ll;
                step = std::pow(o.delta, (double)ll);
                e_new = apd_eval(h, lam_old, h->zeta, step, lam_new, F_new, F_old);
                if (ll == o.ll_max) break;
            }
        }
        ipd_ssn_rec rec;
        rec.k = k;
        rec.ssn_it = ssn_it;
        rec.ll = ll;
        rec.itamg = itpcg;
        rec.E = e_old.E;
        rec.info0 = ho.num_comp;
        rec.info1 = ho.it_num;
        rec.Fk_norm = e_new.normF;
        rec.resamg = ho.resamg;
        rec.bk1 = bk1;
        rec.tk = tk;
        h->recs.push_back(rec);
        if (e_new.normF <= ssn_tol) break;                                         // :213
        const double stall = c2 ? ssn_tol : ssn_tol / 100.0;                       // :219 / Class2 :207
        if (std::fabs(e_old.normF - e_new.normF) < stall) break;
        if (ssn_it == o.ssn_it) break;                                             // :226
    }
    // :239-254
    double kk[4], fx;
    apd_end(h, true, nullptr, lam_new, kk, &fx);
    double rr = max_rr(h, kk);
    double bk_next = bk1;
    if (bk1 < 1e-8 && rr > resk) {                                                 // :245 restart
        copy_dev(h, h->v, h->u, h->U);                                             // vk1 = xk
        if (c2) {
            bk_next = 10.0 * bk1;                                                  // Class2 :255
        } else {
            // rand (:246) through the stream's fill(): a replayed stream, its exhaustion error and
            // the count of consumed numbers are honoured (next_double() bypassed all three)
            double v = 0.0;
            rng->fill(&v, 1);
            bk_next = v;
        }
        ++h->restarts;
        apd_end(h, false, h->u, h->lam, kk, &fx);
        rr = max_rr(h, kk);
    } else {
        std::swap(h->u, h->u2);
        copy_dev(h, h->lam, lam_new, (size_t)h->L);
    }
    h->bk = bk_next;
    for (int i = 0; i < 4; ++i) h->kkt[i] = kk[i];
    h->fval = fx;
    push_hist(h);
    h->h_ssn.push_back((double)ssn_it);
    h->sum_amg += psum;
    if (rr <= o.kkt_tol) h->converged = true;                                      // :266
}

void fill_result(const ipd_apd* h, ipd_apd_result* r) {
    r->converged = h->converged ? 1 : 0;
    r->k = h->k;
    r->fval = h->fval;
    r->kkt[0] = h->kkt[0];
    r->kkt[1] = h->kkt[1];
    r->kkt[2] = h->kkt[2];
    r->kkt[3] = h->kkt[3];
    r->rr = h->have_kkt ? max_rr(h, h->kkt) : 0.0;
    r->sum_amg = h->sum_amg;
    r->total_amg = h->total_amg;
    r->fail_amg = h->fail_amg;
    r->max_amg = h->max_amg;
    r->restarts = h->restarts;
    r->nrec = (int64_t)h->recs.size();
}

// A-ADMM warm start                       warmup_class1.m:22-96 / warmup_class2.m:19-100
void apd_warmup(ipd_apd* h, double res, long long maxit) {
    (void)res;  // the reference's residual test is commented out (:82-93): only maxit acts
    ipd_ctx* ctx = h->ctx;
    CallScope scope(ctx);
    Arena& tmp = *ctx->scratch;
    const size_t U = h->U;
    const int L = h->L;
    double* x = h->u;
    double* v = h->v;
    double* dd = h->w;
    double* wk = tmp.alloc<double>(U);
    double* pik = tmp.alloc<double>(U);
    double* l2 = tmp.alloc<double>(U);
    double* lk1 = tmp.alloc<double>((size_t)L);
    double* hl1 = tmp.alloc<double>((size_t)L);
    double* Hx = tmp.alloc<double>((size_t)L);
    double* Hdd = tmp.alloc<double>((size_t)L);
    double* ff = tmp.alloc<double>((size_t)L);
    for (double* a : {x, v, wk, pik, l2}) IPD_HIP(hipMemsetAsync(a, 0, U * sizeof(double), ctx->stream));
    for (double* a : {lk1, Hx}) IPD_HIP(hipMemsetAsync(a, 0, (size_t)L * sizeof(double), ctx->stream));
    const double muf = 0.0;
    double gk = 1.0, bk = 1.0;
    for (long long it = 1; it <= maxit; ++it) {
        const double ak = bk, bk1 = bk / (1.0 + ak);
        const double gk1 = (gk + muf * ak) / (1.0 + ak);
        const double etafk = (1.0 + ak) * gk + muf * ak;
        const double sgk = 1.0 / bk1, etagk = (1.0 + ak) * bk;
        const double tt = sgk * ak * ak, sg = 1.0 + etafk / tt;
        WarmScal s;
        s.ak = ak;
        s.gk = gk;
        s.muf = muf;
        s.etafk = etafk;
        s.sgk = sgk;
        s.ibk = 1.0 / bk;
        s.akbk = ak / bk;
        s.ak2 = ak * ak;
        s.tt_etafk = etafk + tt;
        s.prox_scale = ak * ak / etagk;
        s.gkmu = gk + muf * ak;
        s.iak1 = 1.0 + ak;
        hipLaunchKernelGGL(k_warm_hl1, dim3(1), dim3(BT), 0, ctx->stream, L, (const double*)lk1,
                           (const double*)Hx, (const double*)h->b, s.ibk, s.akbk, hl1);
        IPD_KERNEL_CHECK();
        OpWarmA a;
        a.P = h->P;
        a.x = x;
        a.v = v;
        a.wk = wk;
        a.pik = pik;
        a.l2 = l2;
        a.hl1 = hl1;
        a.b = h->b;
        a.dd = dd;
        a.s = s;
        launch_tiles(h, a);
        WarmFinA fa;
        fa.pt = parts_of(h);
        fa.P = h->P;
        fa.x = x;
        fa.v = v;
        fa.wk = wk;
        fa.pik = pik;
        fa.l2 = l2;
        fa.hl1 = hl1;
        fa.b = h->b;
        fa.dd = dd;
        fa.Hdd = Hdd;
        fa.s = s;
        hipLaunchKernelGGL(k_warm_fin_a, dim3(1), dim3(BT), 0, ctx->stream, fa);
        IPD_KERNEL_CHECK();
        {
            CallScope inner(ctx);
            if (h->cls == 2)
                kkt_inv_hht_pre(ctx, Hdd, h->p, h->q, h->m, h->n, sg, h->phi_l, h->phi_part,
                                h->phi_npart, ff);                                // Class2 :72
            else
                kkt_inv_aat(ctx, Hdd, h->p, h->q, h->m, h->n, sg, sg, ff);        // Class1 :70
            OpWarmB b;
            b.P = h->P;
            b.dd = dd;
            b.ff = ff;
            b.x = x;
            b.v = v;
            b.wk = wk;
            b.pik = pik;
            b.l2 = l2;
            b.s = s;
            launch_tiles(h, b);
            WarmFinB fb;
            fb.px = parts_of(h);
            fb.P = h->P;
            fb.dd = dd;
            fb.ff = ff;
            fb.x = x;
            fb.v = v;
            fb.wk = wk;
            fb.pik = pik;
            fb.l2 = l2;
            fb.b = h->b;
            fb.lk1 = lk1;
            fb.Hx = Hx;
            fb.s = s;
            hipLaunchKernelGGL(k_warm_fin_b, dim3(1), dim3(BT), 0, ctx->stream, fb);
            IPD_KERNEL_CHECK();
        }
        gk = gk1;
        bk = bk1;
    }
    // driver state: xk = xk0, vk = xk, lk = lk0, bk = 1      (Class1 :59-60,35)
    copy_dev(h, h->v, h->u, U);
    copy_dev(h, h->lam, lk1, (size_t)L);
    ctx->sync();
    h->bk = 1.0;
    h->k = 0;
    h->have_kkt = false;
    h->converged = false;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" void ipd_apd_opts_init(int32_t cls, ipd_apd_opts* o) {
    if (!o) return;
    o->maxit = 100;
    o->kkt_tol = 1e-6;
    o->ssn_it = 50;
    o->ssn_tol1 = cls == 2 ? 1e-10 : 1e-11;
    o->nu = 0.2;
    o->delta = 0.9;
    o->ll_max = 500;
    o->prob = 2;
    o->inner_solver = 4;
    o->pcg_retol = 1e-11;
    o->pcg_maxit = 10000;
}

extern "C" int ipd_apd_create(ipd_ctx* ctx, const ipd_apd_data* d, ipd_apd** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && d && out, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(d->cls == 1 || d->cls == 2, IPD_E_ARG, "cls must be 1 or 2");
        IPD_REQUIRE(d->m > 0 && d->n > 0 && d->m <= 16384 && d->n <= 16384, IPD_E_LIMIT,
                    "m, n must be in [1, 16384]");
        IPD_REQUIRE(d->c && d->r && d->l && d->p && d->q, IPD_E_ARG, "NULL problem vector");
        IPD_REQUIRE(d->cls == 1 || d->phi, IPD_E_ARG, "class 2 needs phi");
        ctx->set_device();
        std::unique_ptr<ipd_apd> h(new ipd_apd);
        h->ctx = ctx;
        h->arena.reset(new Arena(&ctx->pool));
        Arena& A = *h->arena;
        const int m = (int)d->m, n = (int)d->n;
        h->cls = d->cls;
        h->m = m;
        h->n = n;
        h->M = m + n;
        h->L = h->M + (d->cls == 2 ? 1 : 0);
        h->mn = (size_t)m * n;
        h->U = h->mn + (d->cls == 2 ? (size_t)h->M : 0);
        h->mu = d->mu;
        h->geo = make_geo(m, n);
        const size_t mn = h->mn, U = h->U;
        const int L = h->L;
        h->c = A.alloc<double>(mn);
        h->p = A.alloc<double>((size_t)m);
        h->q = A.alloc<double>((size_t)n);
        h->b = A.alloc<double>((size_t)L);
        ctx->upload(h->c, d->c, mn);
        ctx->upload(h->p, d->p, (size_t)m);
        ctx->upload(h->q, d->q, (size_t)n);
        ctx->upload(h->b, d->r, (size_t)n);            // b = [r;l(;mu)]  (:33 / Class2 :29)
        ctx->upload(h->b + n, d->l, (size_t)m);
        if (d->cls == 2) {
            ctx->upload(h->b + h->M, &d->mu, 1);
            h->phi = A.alloc<double>(mn);
            ctx->upload(h->phi, d->phi, mn);
            h->tmask = A.alloc<double>((size_t)h->M);
            h->phi_l = A.alloc<double>((size_t)h->M);
            h->phi_part = A.alloc<double>(1024);
        } else if (d->gama) {
            h->gama = A.alloc<double>(mn);
            ctx->upload(h->gama, d->gama, mn);
        }
        h->u = A.alloc<double>(U);
        h->u2 = A.alloc<double>(U);
        h->v = A.alloc<double>(U);
        h->w = A.alloc<double>(U);
        for (double* a : {h->u, h->u2, h->v, h->w})
            IPD_HIP(hipMemsetAsync(a, 0, U * sizeof(double), ctx->stream));
        h->lam = A.alloc<double>((size_t)L);
        h->lam_a = A.alloc<double>((size_t)L);
        h->lam_b = A.alloc<double>((size_t)L);
        h->wlk = A.alloc<double>((size_t)L);
        h->F_a = A.alloc<double>((size_t)L);
        h->F_b = A.alloc<double>((size_t)L);
        h->zeta = A.alloc<double>((size_t)L);
        h->negF = A.alloc<double>((size_t)L);
        for (double* a : {h->lam, h->lam_a, h->lam_b, h->wlk, h->F_a, h->F_b, h->zeta, h->negF})
            IPD_HIP(hipMemsetAsync(a, 0, (size_t)L * sizeof(double), ctx->stream));
        h->s_words = (mn + 7) / 8;                      // compared in 8-byte words (row f3)
        h->s = A.alloc<uint8_t>(h->s_words * 8);
        IPD_HIP(hipMemsetAsync(h->s, 0, h->s_words * 8, ctx->stream));
        h->s_prev = A.alloc<unsigned long long>(h->s_words);
        if (d->cls == 2) h->t_prev = A.alloc<unsigned long long>((size_t)h->M);
        h->step_changed = A.alloc<int>(1);
        {
            const char* e = getenv("IPD_NO_STEP_DONOR");
            h->step_reuse = !(e && e[0] == '1');
        }
        const Geo& g = h->geo;
        h->lpart = A.alloc<double>((size_t)g.njg * m);
        h->rpart = A.alloc<double>((size_t)g.nib * 4 * n);
        IPD_HIP(hipMemsetAsync(h->rpart, 0, sizeof(double) * (size_t)g.nib * 4 * n, ctx->stream));
        h->spart = A.alloc<double>((size_t)g.nib * g.njg * NSC);
        h->dscal = reinterpret_cast<ApdScal*>(A.alloc<double>(sizeof(ApdScal) / sizeof(double) + 1));
        h->fpart = A.alloc<double>((size_t)cdiv(L, EB) * NFS);
        h->merit = A.alloc<double>(MK);
        h->counter = A.alloc<int>(4);
        IPD_HIP(hipMemsetAsync(h->counter, 0, 16, ctx->stream));
        Prob& P = h->P;
        P.cls2 = d->cls == 2 ? 1 : 0;
        P.m = m;
        P.n = n;
        P.p = h->p;
        P.q = h->q;
        P.c = h->c;
        P.phi = h->phi;
        P.gama = h->gama;
        P.gs = d->cls == 2 ? std::numeric_limits<double>::infinity() : d->gama_scalar;
        if (d->cls == 2) {
            CallScope scope(ctx);
            h->phi_npart = kkt_phi_consts(ctx, h->phi, h->p, h->q, m, n, h->phi_l, h->phi_part);
        }
        ctx->sync();
        *out = h.release();
    });
}

extern "C" void ipd_apd_destroy(ipd_apd* h) {
    if (!h) return;
    try {
        h->ctx->set_device();
        h->ctx->sync();
    } catch (...) {
    }
    delete h;
}

extern "C" int ipd_apd_dims(const ipd_apd* h, int64_t* len_u, int64_t* len_lam) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        if (len_u) *len_u = (int64_t)h->U;
        if (len_lam) *len_lam = (int64_t)h->L;
    });
}

extern "C" int ipd_apd_warmup(ipd_apd* h, double res, int64_t maxit) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        const bool inf = maxit < 0;
        IPD_REQUIRE(!(res == 0.0 && inf), IPD_E_ARG, "res = 0 and maxit = inf");   // :10-12
        long long its = inf ? 500 : (long long)maxit;                                // :18-20
        apd_warmup(h, res, its);
    });
}

extern "C" int ipd_apd_set_state(ipd_apd* h, const double* u, const double* v, const double* lam,
                                 double bk) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        IPD_REQUIRE(bk > 0.0, IPD_E_ARG, "bk must be positive");
        h->ctx->set_device();
        if (u) h->ctx->upload(h->u, u, h->U);
        if (v) h->ctx->upload(h->v, v, h->U);
        if (lam) h->ctx->upload(h->lam, lam, (size_t)h->L);
        h->bk = bk;
        h->k = 0;
        h->have_kkt = false;
        h->converged = false;
        h->recs.clear();
        h->sum_amg = h->total_amg = h->fail_amg = h->max_amg = 0;
        h->restarts = 0;
    });
}

extern "C" int ipd_apd_get_state(ipd_apd* h, double* u, double* v, double* lam, double* bk) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        h->ctx->set_device();
        if (u) h->ctx->fetch(h->u, u, h->U);
        if (v) h->ctx->fetch(h->v, v, h->U);
        if (lam) h->ctx->fetch(h->lam, lam, (size_t)h->L);
        if (bk) *bk = h->bk;
    });
}

extern "C" int ipd_apd_run(ipd_apd* h, const ipd_apd_opts* o, const ipd_amg_opts* amg,
                           ipd_rng* rng, int32_t iters, ipd_apd_result* res) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && rng, IPD_E_ARG, "NULL argument");
        h->ctx->comp_order.clear();   // a recorded component order belongs to ONE Newton system
        ipd_apd_opts oo;
        if (o)
            oo = *o;
        else
            ipd_apd_opts_init(h->cls, &oo);
        IPD_REQUIRE(oo.maxit > 0 && oo.ssn_it > 0 && oo.ll_max >= 0 && oo.delta > 0.0 &&
                        oo.delta < 1.0,
                    IPD_E_ARG, "bad driver options");
        IPD_REQUIRE(oo.inner_solver >= 1 && oo.inner_solver <= 5, IPD_E_ARG,
                    "inner_solver must be 1 (direct), 2 (PCG), 3 (aug_PCG / PCG4POT), 4 (AMG) or 5 (two-grid)");
        const AmgOpts ao = oo.inner_solver == 5 ? amg_fill_twogrid_defaults(amg) : amg_fill_defaults(amg);
        h->ctx->set_device();
        ensure_kkt(h);
        for (int it = 0; it < iters && h->k < oo.maxit && !h->converged; ++it)
            apd_iterate(h, oo, ao, rng);
        h->ctx->sync();
        if (res) fill_result(h, res);
    });
}

extern "C" int ipd_apd_history(const ipd_apd* h, int32_t which, double* out, int64_t cap,
                               int64_t* count) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && count, IPD_E_ARG, "NULL argument");
        const std::vector<double>* src = nullptr;
        switch (which) {
            case 0: src = &h->h_fx; break;
            case 1: src = &h->h_kx; break;
            case 2: src = &h->h_kl; break;
            case 3: src = &h->h_ky; break;
            case 4: src = &h->h_kz; break;
            case 5: src = &h->h_ssn; break;
            default: throw IpdError(IPD_E_ARG, "history selector must be 0..5");
        }
        const int64_t nn = std::min<int64_t>(cap, (int64_t)src->size());
        if (out)
            for (int64_t i = 0; i < nn; ++i) out[i] = (*src)[(size_t)i];
        *count = (int64_t)src->size();
    });
}

extern "C" int ipd_apd_reuse_stats(const ipd_apd* h, int64_t* steps, int64_t* same_system,
                                   int64_t* donated) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        if (steps) *steps = h->step.steps;
        if (same_system) *same_system = h->step.same_steps;
        if (donated) *donated = h->step.shared;
    });
}

extern "C" int ipd_apd_records(const ipd_apd* h, ipd_ssn_rec* out, int64_t cap, int64_t* count) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && count, IPD_E_ARG, "NULL argument");
        const int64_t nn = std::min<int64_t>(cap, (int64_t)h->recs.size());
        if (out)
            for (int64_t i = 0; i < nn; ++i) out[i] = h->recs[(size_t)i];
        *count = (int64_t)h->recs.size();
    });
}

extern "C" int ipd_apd_begin(ipd_apd* h, int32_t k, double vals[3]) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && k >= 1, IPD_E_ARG, "bad argument");
        h->ctx->set_device();
        apd_begin(h, k);
        h->ctx->sync();
        if (vals) {
            vals[0] = h->bk1;
            vals[1] = h->tk;
            vals[2] = h->ak;
        }
    });
}

extern "C" int ipd_apd_eval(ipd_apd* h, const double* lam, uint8_t* s_out, double* t_out,
                            double* Fk_out, double vals[6]) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && lam, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(h->tk > 0.0, IPD_E_ARG, "ipd_apd_begin must run first");
        ipd_ctx* ctx = h->ctx;
        ctx->set_device();
        ctx->upload(h->lam_a, lam, (size_t)h->L);
        const EvalRes e = apd_eval(h, h->lam_a, nullptr, 0.0, nullptr, h->F_a, nullptr);
        if (s_out) ctx->fetch(h->s, s_out, h->mn);
        if (t_out && h->tmask) ctx->fetch(h->tmask, t_out, (size_t)h->M);
        if (Fk_out) ctx->fetch(h->F_a, Fk_out, (size_t)h->L);
        if (vals) {
            vals[0] = h->bk1;
            vals[1] = h->tk;
            vals[2] = h->ak;
            vals[3] = e.normF;
            vals[4] = h->bk1 / 2.0 * e.lam2 - e.wlk_lam + 0.5 * h->tk * e.prox2;
            vals[5] = (double)e.E;
        }
    });
}

extern "C" int ipd_apd_bench_eval(ipd_apd* h, int32_t reps, double* total_ms,
                                  double* bytes_per_pass) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && reps > 0 && total_ms, IPD_E_ARG, "bad argument");
        IPD_REQUIRE(h->tk > 0.0, IPD_E_ARG, "ipd_apd_begin must run first");
        ipd_ctx* ctx = h->ctx;
        ctx->set_device();
        const Lam lamv{h->lam, nullptr, 0.0};
        const double itk = 1.0 / h->tk;
        hipEvent_t e0, e1;
        IPD_HIP(hipEventCreate(&e0));
        IPD_HIP(hipEventCreate(&e1));
        launch_eval(h, lamv, itk);
        IPD_HIP(hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < reps; ++r) launch_eval(h, lamv, itk);
        IPD_HIP(hipEventRecord(e1, ctx->stream));
        IPD_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, e0, e1));
        IPD_HIP(hipEventDestroy(e0));
        IPD_HIP(hipEventDestroy(e1));
        *total_ms = ms;
        // algorithmic bytes: wk read + mask written (+ phi, + gama when they are vectors),
        // the multiplier/p/q vectors and the partial sums
        if (bytes_per_pass) {
            const double mn = (double)h->mn;
            double by = 8.0 * mn + mn + 16.0 * h->M;
            if (h->phi) by += 8.0 * mn;
            if (h->gama) by += 8.0 * mn;
            by += 8.0 * ((double)h->geo.njg * h->m + 4.0 * h->geo.nib * h->n);
            *bytes_per_pass = by;
        }
    });
}
