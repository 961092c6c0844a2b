// Device-side phases of the V/W cycle (included by ipd_cycle.hip only).
//
// Every phase is a __device__ function executed by ONE 1024-thread workgroup `b`
// of `G`, so that thin __global__ wrappers give the multi-workgroup kernels and a
// single-workgroup fused kernel can chain the same code.
//
// Latency structure.  These launches move at most ~12 MB and last a few
// microseconds, so the number of DEPENDENT memory round trips, not the byte
// count, sets their duration (tools/ubench_rows.hip, tools/ubench_smooth.hip: an
// empty launch costs 2.3 us back to back, a 12.6 MB L2-resident stream 2.6 us, a
// naive CSR row walk 4.3-7 us).  Each phase is arranged as ONE global round trip
// plus LDS work:
//   1. the matrix entries of the lane, the owner's r/dinv/Axi and every vector the
//      phase gathers from are requested up front, in one burst;
//   2. big regular levels use a padded copy of the off-diagonal part (uniform row
//      stride, 16-bit columns, 16-byte aligned) so that entry addresses follow from
//      the row number alone -- no row-pointer round trip -- and are fetched with
//      8/16-byte vector loads;
//   3. the gathered vector is staged in LDS, so the gather that depends on the
//      column indices never goes back to memory;
//   4. the kernel-space scalar c = 1'(r - A e)/xx (MG_Vcycle.m:18-19) is evaluated
//      by every workgroup from the vectors themselves through the identity
//      1'(r - A e) = 1'r - (A1)'e (A symmetric): no partial-sum hand-off between
//      launches, deterministic, and reduced in the same LDS exchange as the rows.
#pragma once

// matrix entries in flight per lane and batch (one 4-entry vector of the padded format).
// 8 was slower on every workload (m=n=1024 Class 1 run 1.58 -> 1.55 s, tree-mask W cycle
// 0.532 -> 0.503 ms): short rows fill 3-6 of the slots and the rest are clamped dummy loads.
static constexpr int ROW_U = 4;
static constexpr int STAGE_MAX = 7680;   // vector entries staged in LDS (60 KiB)

// LDS scratch of a phase
struct PhaseLds {
    double row[BT / 64];
    double sca[BT / 64];
    double out[2 * (BT / 64)];
};

// One batch of a row: ROW_U (column, value) pairs held by a lane (absent: column 0, value 0).
struct RowBatch {
    int j[ROW_U];
    double a[ROW_U];
};

// All loads below are UNCONDITIONAL with clamped addresses: a load under a per-lane
// condition becomes its own exec-masked basic block, the compiler then waits for it
// before issuing the next one, and a batch degenerates into one memory round trip
// per entry (seen in the ISA: s_and_saveexec / global_load / s_waitcnt chains).
// Entries that do not exist get column 0 and value 0, so they add 0 * x[0].

// CSR: lane `gl` of `L` holds entries t, t+L, ... of [t, e1)
__device__ __forceinline__ void batch_load_csr(RowBatch& bt, const int* __restrict__ ci,
                                               const double* __restrict__ va, int t, int e1,
                                               int L) {
    int jj[ROW_U];
    double aa[ROW_U];
#pragma unroll
    for (int u = 0; u < ROW_U; ++u) {
        const int tt = t + u * L;
        const int tc = tt < e1 ? tt : 0;
        jj[u] = ci[tc];
        aa[u] = va[tc];
    }
#pragma unroll
    for (int u = 0; u < ROW_U; ++u) {
        const bool ok = t + u * L < e1;
        bt.j[u] = ok ? jj[u] : 0;
        bt.a[u] = ok ? aa[u] : 0.0;
    }
}

// padded rows: lane holds the 4-entry vectors v, v+L of the row starting at `base`
__device__ __forceinline__ void batch_load_pad(RowBatch& bt, const unsigned short* __restrict__ pci,
                                               const double* __restrict__ pva, size_t base, int v,
                                               int nvec, int L) {
#pragma unroll
    for (int u = 0; u < ROW_U / 4; ++u) {
        const int vv = v + u * L;
        if (vv < nvec) {  // wave-uniform for all but the last vector of a row
            const size_t off = base + 4 * (size_t)vv;
            const ushort4 c4 = *reinterpret_cast<const ushort4*>(pci + off);
            const double2 a01 = *reinterpret_cast<const double2*>(pva + off);
            const double2 a23 = *reinterpret_cast<const double2*>(pva + off + 2);
            bt.j[4 * u + 0] = c4.x;
            bt.j[4 * u + 1] = c4.y;
            bt.j[4 * u + 2] = c4.z;
            bt.j[4 * u + 3] = c4.w;
            bt.a[4 * u + 0] = a01.x;
            bt.a[4 * u + 1] = a01.y;
            bt.a[4 * u + 2] = a23.x;
            bt.a[4 * u + 3] = a23.y;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bt.j[4 * u + q] = 0;
                bt.a[4 * u + q] = 0.0;
            }
        }
    }
}

template <class XV>
__device__ __forceinline__ double batch_dot(const RowBatch& bt, XV xval) {
    double y[ROW_U];
#pragma unroll
    for (int u = 0; u < ROW_U; ++u) y[u] = xval(bt.j[u]);
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < ROW_U; ++u) s += bt.a[u] * y[u];
    return s;
}

// Per-lane cursor over one matrix row in either format.
struct RowCursor {
    // CSR
    int e0, e1;
    // padded
    size_t base;
    int nvec;
    double dg;  // diagonal (padded format keeps it apart), owner lane only
};

template <bool PAD>
__device__ __forceinline__ void row_open(const LevelDev& lv, int row, bool valid, bool owner,
                                         int gl, int L, RowCursor& rc, RowBatch& bt) {
    const int rowc = valid ? row : 0;  // clamped: loads stay unconditional
    if (PAD) {
        rc.base = (size_t)rowc * lv.S;
        rc.nvec = valid ? lv.S / 4 : 0;
        rc.dg = owner ? lv.diag[rowc] : 0.0;
        batch_load_pad(bt, lv.pci, lv.pva, rc.base, gl, rc.nvec, L);
    } else {
        rc.e0 = lv.rp[rowc];
        rc.e1 = valid ? lv.rp[rowc + 1] : rc.e0;
        rc.dg = 0.0;
        batch_load_csr(bt, lv.ci, lv.va, rc.e0 + gl, rc.e1, L);
    }
}

// finish the row: first batch (already loaded) + the remaining ones
template <bool PAD, class XV>
__device__ __forceinline__ double row_finish(const LevelDev& lv, const RowCursor& rc,
                                             const RowBatch& first, int gl, int L, XV xval) {
    double s = batch_dot(first, xval);
    if (PAD) {
        for (int v = gl + (ROW_U / 4) * L; v < rc.nvec; v += (ROW_U / 4) * L) {
            RowBatch bt;
            batch_load_pad(bt, lv.pci, lv.pva, rc.base, v, rc.nvec, L);
            s += batch_dot(bt, xval);
        }
    } else {
        for (int t = rc.e0 + gl + ROW_U * L; t < rc.e1; t += ROW_U * L) {
            RowBatch bt;
            batch_load_csr(bt, lv.ci, lv.va, t, rc.e1, L);
            s += batch_dot(bt, xval);
        }
    }
    return s;
}

// Reduce `s` over the L-thread group (result in all its threads) and, when
// `with_scalar`, `v` over the whole block (result in *vsum).  One barrier pair at
// most; none when L <= 64 and no scalar is requested.
__device__ __forceinline__ double reduce_rows(double s, int L, bool with_scalar, double v,
                                              double* vsum, PhaseLds* lds) {
    if (L <= 64)
        s = subwave_sum(s, L);
    else
        s = wave_sum(s);
    if (L > 64 || with_scalar) {
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (with_scalar) v = wave_sum(v);
        __syncthreads();
        if (lane == 0) {
            lds->row[w] = s;
            if (with_scalar) lds->sca[w] = v;
        }
        __syncthreads();
        if (L > 64) {
            const int wpg = L >> 6, g0 = (threadIdx.x / L) * wpg;
            double t = 0.0;
            for (int k = 0; k < wpg; ++k) t += lds->row[g0 + k];
            s = t;
        }
        if (with_scalar) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < BT / 64; ++k) t += lds->sca[k];
            *vsum = t;
        }
    }
    return s;
}

// block totals of v0 (and v1) written by thread 0 (one barrier pair)
__device__ __forceinline__ void block_totals_to(double v0, double* dst0, double v1, double* dst1,
                                                PhaseLds* lds) {
    v0 = wave_sum(v0);
    if (dst1) v1 = wave_sum(v1);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        lds->out[threadIdx.x >> 6] = v0;
        if (dst1) lds->out[BT / 64 + (threadIdx.x >> 6)] = v1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int k = 0; k < BT / 64; ++k) {
            t0 += lds->out[k];
            if (dst1) t1 += lds->out[BT / 64 + k];
        }
        *dst0 = t0;
        if (dst1) *dst1 = t1;
    }
}

// Visit j = tid, tid+BT, ... < N, VEC_U visits per trip with their loads issued together.
// Measured on MI355X (m=n=1024 Class 1 run / regime-D V cycle / tree-mask W cycle):
//   VEC_U = 8: 1.85 s / 0.239 ms / 0.648 ms     4: 1.70 / 0.222 / 0.572
//           2: 1.60 / 0.214 / 0.540             1: 1.58 / 0.211 / 0.532
// (1024-thread blocks: the narrower the better.)  With 512-thread blocks a thread visits twice
// as many entries and two per trip is the best balance:
//   VEC_U = 1: 1.52 s / 0.188 ms / 0.434 ms     2: 1.49 / 0.181 / 0.428     4: 1.50 / 0.185 / 0.448
// Wide trips lose because their clamped unconditional loads cost more issue slots and registers
// than the overlap buys.
#ifndef IPD_VEC_U
#define IPD_VEC_U 2
#endif
static constexpr int VEC_U = IPD_VEC_U;
template <class LOAD, class USE>
__device__ __forceinline__ void vec_pass(int N, LOAD load, USE use) {
    for (int j0 = threadIdx.x; j0 < N; j0 += BT * VEC_U) {
        decltype(load(0)) v[VEC_U];
#pragma unroll
        for (int u = 0; u < VEC_U; ++u) {
            const int jj = j0 + u * BT;
            v[u] = load(jj < N ? jj : N - 1);  // unconditional, clamped (see batch_load_*)
        }
#pragma unroll
        for (int u = 0; u < VEC_U; ++u) {
            const int jj = j0 + u * BT;
            if (jj < N) use(jj, v[u]);
        }
    }
}

// wave-uniform row index -> scalar register (L >= 64: a wave works on one row)
__device__ __forceinline__ int uniform_if(int v, bool uni) {
    return uni ? __builtin_amdgcn_readfirstlane(v) : v;
}

// ---------------------------------------------------------------------------
// smoother sweep (Jacobi, or one half of the bigraph Gauss-Seidel sweep)
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
    int staged;     // gather vector staged in LDS (N <= STAGE_MAX)
    int eold_zero;  // eold is identically zero and is not read (first sweep of a visit)
};

template <bool STAGED, bool PAD>
__device__ __forceinline__ void phase_smooth(const SmoothArgs& a, int b, int G, PhaseLds* lds,
                                             double* xs) {
    const LevelDev& lv = a.lv;
    const int tid = threadIdx.x;
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const bool uni = L >= 64;
    const int nrows = a.row1 - a.row0;
    const int niter = (nrows + G * gpb - 1) / (G * gpb);
    const double* __restrict__ eold = a.eold;
    const double* __restrict__ win = a.win;
    const int u0 = a.u0, u1 = a.u1;
    const bool ez = a.eold_zero != 0;
    const bool skip = ez && u0 >= u1;  // nothing to gather: A*e == 0
    const bool nsp = a.isnsp != 0;
    auto xglobal = [&](int j) { return (j >= u0 && j < u1) ? win[j] : (ez ? 0.0 : eold[j]); };
    auto xlds = [&](int j) { return xs[j]; };

    // ---- one burst of independent requests -----------------------------------
    int row = uniform_if(a.row0 + b * gpb + g, uni);
    bool valid = row < a.row1;
    bool owner = valid && gl == 0;
    RowCursor rc;
    RowBatch bt;
    row_open<PAD>(lv, row, valid && !skip, owner && !skip, gl, L, rc, bt);
    double rv = 0.0, dv = 0.0, axi = 0.0, eo = 0.0;
    if (owner) {
        rv = lv.r[row];
        dv = lv.dinv[row];
        if (nsp) axi = lv.Axi[row];
        if (!ez) eo = eold[row];  // row is never inside [u0,u1)
    }
    const double xxv = nsp ? lv.xx[0] : 1.0;
    // pieces of xig = 1'r - (A1)'e_old, and the staged gather vector
    double cpart = 0.0;
    if (nsp || (STAGED && !skip)) {
        struct Q {
            double e, w, r, a;
        };
        const bool st = STAGED && !skip;
        vec_pass(
            lv.N,
            [&](int j) {
                Q q;
                q.e = ez ? 0.0 : eold[j];
                q.w = (st && u1 > u0) ? win[j] : 0.0;  // uniform condition: no per-lane branch
                q.r = nsp ? lv.r[j] : 0.0;
                q.a = nsp ? lv.Axi[j] : 0.0;
                return q;
            },
            [&](int j, const Q& q) {
                if (st) xs[j] = (j >= u0 && j < u1) ? q.w : q.e;
                cpart += q.r - q.a * q.e;
            });
    }
    if (STAGED && !skip) __syncthreads();

    double c = 0.0;
    bool need_c = nsp;
    for (int it = 0; it < niter; ++it) {
        if (it > 0) {
            row = uniform_if(a.row0 + (it * G + b) * gpb + g, uni);
            valid = row < a.row1;
            owner = valid && gl == 0;
            row_open<PAD>(lv, row, valid && !skip, owner && !skip, gl, L, rc, bt);
            if (owner) {
                rv = lv.r[row];
                dv = lv.dinv[row];
                if (nsp) axi = lv.Axi[row];
                eo = ez ? 0.0 : eold[row];
            }
        }
        double s = 0.0;
        if (!skip) {
            s = STAGED ? row_finish<PAD>(lv, rc, bt, gl, L, xlds)
                       : row_finish<PAD>(lv, rc, bt, gl, L, xglobal);
        }
        double xig = 0.0;
        s = reduce_rows(s, L, need_c, cpart, &xig, lds);
        if (need_c) {
            c = xig / xxv;  // MG_Vcycle.m:19
            need_c = false;
        }
        if (owner) {
            if (PAD) s += rc.dg * eo;         // the diagonal term of (A e)_row
            const double g_i = rv - s - axi * c;
            const double wv = eo + dv * g_i;  // e + R*(g - Axi*c)
            if (a.wout) a.wout[row] = wv;
            a.enew[row] = wv + c;             //   ... + xi*c
        }
    }
}

// ---------------------------------------------------------------------------
// rr = r - A e                                                   MG_Vcycle.m:27
// ---------------------------------------------------------------------------
template <bool STAGED, bool PAD>
__device__ __forceinline__ void phase_resid(const LevelDev& lv, const double* __restrict__ e,
                                            int row0, int row1, int b, int G, PhaseLds* lds,
                                            double* xs) {
    const int tid = threadIdx.x;
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const bool uni = L >= 64;
    const int niter = (row1 - row0 + G * gpb - 1) / (G * gpb);
    auto xglobal = [&](int j) { return e[j]; };
    auto xlds = [&](int j) { return xs[j]; };
    int row = uniform_if(row0 + b * gpb + g, uni);
    bool valid = row < row1;
    bool owner = valid && gl == 0;
    RowCursor rc;
    RowBatch bt;
    row_open<PAD>(lv, row, valid, owner, gl, L, rc, bt);
    double rv = 0.0, eo = 0.0;
    if (owner) {
        rv = lv.r[row];
        eo = e[row];
    }
    if (STAGED) {
        vec_pass(lv.N, [&](int j) { return e[j]; }, [&](int j, double v) { xs[j] = v; });
        __syncthreads();
    }
    for (int it = 0; it < niter; ++it) {
        if (it > 0) {
            row = uniform_if(row0 + (it * G + b) * gpb + g, uni);
            valid = row < row1;
            owner = valid && gl == 0;
            row_open<PAD>(lv, row, valid, owner, gl, L, rc, bt);
            if (owner) {
                rv = lv.r[row];
                eo = e[row];
            }
        }
        double s = STAGED ? row_finish<PAD>(lv, rc, bt, gl, L, xlds)
                          : row_finish<PAD>(lv, rc, bt, gl, L, xglobal);
        double dummy;
        s = reduce_rows(s, L, false, 0.0, &dummy, lds);
        if (owner) {
            if (PAD) s += rc.dg * eo;
            lv.rr[row] = rv - s;
        }
    }
}

// ---------------------------------------------------------------------------
// y = M*x row walk: restriction (M = P') and prolongation (M = P, y += M x); CSR
// ---------------------------------------------------------------------------
struct XferArgs {
    int nrows, ncols, L, G;
    int row0, row1;  // rows produced by this launch (a rank's slice when sharded)
    const int* rp;
    const int* ci;
    const double* va;
    const double* x;
    double* y;   // restriction: y = M x ; prolongation: y += M x
    int add;     // 1 = prolongation
    int staged;
};

template <bool STAGED>
__device__ __forceinline__ void phase_xfer(const XferArgs& a, int b, int G, PhaseLds* lds,
                                           double* xs) {
    const int tid = threadIdx.x;
    const int L = a.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const bool uni = L >= 64;
    const int niter = (a.row1 - a.row0 + G * gpb - 1) / (G * gpb);
    const double* __restrict__ x = a.x;
    auto xglobal = [&](int j) { return x[j]; };
    auto xlds = [&](int j) { return xs[j]; };
    LevelDev lv;  // only the CSR fields are used by row_open/row_finish<false>
    lv.rp = a.rp;
    lv.ci = a.ci;
    lv.va = a.va;
    int row = uniform_if(a.row0 + b * gpb + g, uni);
    bool valid = row < a.row1;
    bool owner = valid && gl == 0;
    RowCursor rc;
    RowBatch bt;
    row_open<false>(lv, row, valid, owner, gl, L, rc, bt);
    double y0 = 0.0;
    if (owner && a.add) y0 = a.y[row];
    if (STAGED) {
        vec_pass(a.ncols, [&](int j) { return x[j]; }, [&](int j, double v) { xs[j] = v; });
        __syncthreads();
    }
    for (int it = 0; it < niter; ++it) {
        if (it > 0) {
            row = uniform_if(a.row0 + (it * G + b) * gpb + g, uni);
            valid = row < a.row1;
            owner = valid && gl == 0;
            row_open<false>(lv, row, valid, owner, gl, L, rc, bt);
            y0 = (owner && a.add) ? a.y[row] : 0.0;
        }
        double s = STAGED ? row_finish<false>(lv, rc, bt, gl, L, xlds)
                          : row_finish<false>(lv, rc, bt, gl, L, xglobal);
        double dummy;
        s = reduce_rows(s, L, false, 0.0, &dummy, lds);
        if (owner) a.y[row] = y0 + s;
    }
}

// ---------------------------------------------------------------------------
// fused residual + restriction:  r_c = P'(r - A e) = P'r - (P'A) e            MG_Vcycle.m:27
// ---------------------------------------------------------------------------
// T1 = P'A is the first product of the Galerkin triple (transfer.m:66) and is kept by the setup,
// so the coarse right-hand side is ONE row walk over the rows of P' (against r) and of T1
// (against e) instead of a residual launch followed by a restriction launch.
struct RrcArgs {
    XferArgs p;          // rows of P' (CSR), x = r (fine right-hand side), y = coarse right-hand side
    const int* rp2;      // T1 = P'A (CSR, same rows)
    const int* ci2;
    const double* va2;
    const double* e;     // fine iterate
};

template <bool STAGED>
__device__ __forceinline__ void phase_rrc(const RrcArgs& a, int b, int G, PhaseLds* lds, double* xs) {
    const int tid = threadIdx.x;
    const int L = a.p.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const bool uni = L >= 64;
    const int ncols = a.p.ncols;
    const int niter = (a.p.row1 - a.p.row0 + G * gpb - 1) / (G * gpb);
    const double* __restrict__ r = a.p.x;
    const double* __restrict__ e = a.e;
    double* xe = xs + ncols;
    auto rglobal = [&](int j) { return r[j]; };
    auto eglobal = [&](int j) { return e[j]; };
    auto rlds = [&](int j) { return xs[j]; };
    auto elds = [&](int j) { return xe[j]; };
    LevelDev lp, lt;   // only the CSR fields are used by row_open/row_finish<false>
    lp.rp = a.p.rp;
    lp.ci = a.p.ci;
    lp.va = a.p.va;
    lt.rp = a.rp2;
    lt.ci = a.ci2;
    lt.va = a.va2;
    int row = uniform_if(a.p.row0 + b * gpb + g, uni);
    bool valid = row < a.p.row1;
    bool owner = valid && gl == 0;
    RowCursor rcp, rct;
    RowBatch btp, btt;
    row_open<false>(lp, row, valid, owner, gl, L, rcp, btp);
    row_open<false>(lt, row, valid, owner, gl, L, rct, btt);
    if (STAGED) {
        struct Q {
            double r, e;
        };
        vec_pass(
            ncols,
            [&](int j) {
                Q q;
                q.r = r[j];
                q.e = e[j];
                return q;
            },
            [&](int j, const Q& q) {
                xs[j] = q.r;
                xe[j] = q.e;
            });
        __syncthreads();
    }
    for (int it = 0; it < niter; ++it) {
        if (it > 0) {
            row = uniform_if(a.p.row0 + (it * G + b) * gpb + g, uni);
            valid = row < a.p.row1;
            owner = valid && gl == 0;
            row_open<false>(lp, row, valid, owner, gl, L, rcp, btp);
            row_open<false>(lt, row, valid, owner, gl, L, rct, btt);
        }
        double s1 = STAGED ? row_finish<false>(lp, rcp, btp, gl, L, rlds)
                           : row_finish<false>(lp, rcp, btp, gl, L, rglobal);
        double s2 = STAGED ? row_finish<false>(lt, rct, btt, gl, L, elds)
                           : row_finish<false>(lt, rct, btt, gl, L, eglobal);
        double dummy;
        s1 = reduce_rows(s1 - s2, L, false, 0.0, &dummy, lds);
        if (owner) a.p.y[row] = s1;
    }
}

// ---------------------------------------------------------------------------
// top of the Class_AMG loop: x_new = x + e ; r = b - A x_new  (k_conv then forms ||r||)
// ---------------------------------------------------------------------------
struct TopArgs {
    LevelDev lv;
    const double* b;
    const double* x;
    const double* e;   // NULL -> x_new = x
    double* xnew;
    int row0, row1;  // rows produced by this launch
    int staged;
};

template <bool STAGED, bool PAD>
__device__ __forceinline__ void phase_top(const TopArgs& a, int b, int G, PhaseLds* lds,
                                          double* xs) {
    const LevelDev& lv = a.lv;
    const int tid = threadIdx.x;
    const int L = lv.L, gpb = BT / L;
    const int g = tid / L, gl = tid - g * L;
    const bool uni = L >= 64;
    const int niter = (a.row1 - a.row0 + G * gpb - 1) / (G * gpb);
    const double* __restrict__ x = a.x;
    const double* __restrict__ e = a.e;
    auto xglobal = [&](int j) { return e ? x[j] + e[j] : x[j]; };
    auto xlds = [&](int j) { return xs[j]; };
    int row = uniform_if(a.row0 + b * gpb + g, uni);
    bool valid = row < a.row1;
    bool owner = valid && gl == 0;
    RowCursor rc;
    RowBatch bt;
    row_open<PAD>(lv, row, valid, owner, gl, L, rc, bt);
    double bv = 0.0, xo = 0.0;
    if (owner) {
        bv = a.b[row];
        xo = xglobal(row);
    }
    if (STAGED) {
        vec_pass(lv.N, [&](int j) { return xglobal(j); }, [&](int j, double v) { xs[j] = v; });
        __syncthreads();
    }
    for (int it = 0; it < niter; ++it) {
        if (it > 0) {
            row = uniform_if(a.row0 + (it * G + b) * gpb + g, uni);
            valid = row < a.row1;
            owner = valid && gl == 0;
            row_open<PAD>(lv, row, valid, owner, gl, L, rc, bt);
            if (owner) {
                bv = a.b[row];
                xo = xglobal(row);
            }
        }
        double s = STAGED ? row_finish<PAD>(lv, rc, bt, gl, L, xlds)
                          : row_finish<PAD>(lv, rc, bt, gl, L, xglobal);
        double dummy;
        s = reduce_rows(s, L, false, 0.0, &dummy, lds);
        if (owner) {
            if (PAD) s += rc.dg * xo;
            const double ri = bv - s;
            lv.r[row] = ri;
            a.xnew[row] = xo;
        }
    }
}
