// Whole Class_AMG solve of a REALISTIC hierarchy in one workgroup (included by ipd_cycle.hip).
//
// The active sets of real Newton steps are tree-like: level 1 has M = m+n <= 2048 rows of 2-7
// entries, level 2 about half of that, and from level 3 down (a few hundred rows) everything fits
// in LDS and already runs as the single-workgroup sub-cycle (k_subcycle / blk_cycle).  What was
// left on the multi-launch path were the ~50 phases of levels 1-2 per W cycle: 5-6 us each,
// all latency (a launch boundary plus two or three dependent memory round trips for a few KB of
// matrix), 34 % of a driver run, plus a read-back per cycle for the stopping rules.
//
// Here ONE workgroup keeps going for the whole solve:
//   * levels 1 and 2: a thread owns up to MID_RPT rows (row = tid + j*BT); the first MID_RC entries
//     of each row sit in registers for the whole solve (the rest, for the rare long rows, is read
//     from L2).  The vectors of these two levels stay in global memory (16 KB each: L1/L2-hot, and
//     one CU is their only reader and writer): LDS belongs to level 3, whose matrix alone takes
//     70-150 KB of it.  A sweep is one burst of gathers and ONE barrier (two for the bigraph
//     Gauss-Seidel level: F rows, then C rows);
//   * level 3 down: the LDS image of the sub-cycle, loaded once per solve instead of once per
//     visit, and blk_cycle / tiny_cycle as they are;
//   * transfers between levels 1, 2, 3: CSR rows from L2 (short rows), vectors in LDS;
//   * the stationary iteration and its stopping rules (Class_AMG.m:86-109) in the kernel:
//     one launch and one read-back per solve.
// Arithmetic per row is the multi-launch kernels' (phase_smooth, phase_resid, phase_xfer,
// phase_top); a row's dot product is accumulated in column order by one thread.
//
// STATUS: opt-in (IPD_MID=1), correct (same cycle counts and residuals as the multi-launch path,
// tests/test_gpu_mid.py) but not faster on the systems it was built for -- see the note at its
// planning code in ipd_cycle_host.h and DESIGN.md section 6.
#pragma once

static constexpr int MID_RPT = 4;   // rows per thread on level 1 (up to MID_RPT*BT rows)
static constexpr int MID_RPT2 = 2;  // rows per thread on level 2
static constexpr int MID_RC = 4;    // entries of a row kept in registers (6: 90 spilled registers)

struct MidLevel {
    int N, nf, Nc;
    const int* rp;
    const int* ci;
    const double* va;
    const double* dinv;
    const double* Axi;
    const double* xx;
    const int* Rrp;   // restriction P' to the child (Nc rows)
    const int* Rci;
    const double* Rva;
    const int* Prp;   // prolongation P from the child (N rows)
    const int* Pci;
    const double* Pva;
};
static constexpr int MID_CH = 32;    // entries per chunk of a chunked level-3 row
struct MidDesc {
    MidLevel L1, L2;
    // level 3 when it is too big for the LDS image (a few hundred rows of 10-70 entries): its rows
    // are cut into chunks of MID_CH entries, a thread forms a chunk's partial dot product from L2,
    // the row owners add their rows' chunks in order; the image then starts at level 4
    MidLevel L3;
    const int* row_ch;    // [N3+1] first chunk of a row; row_ch[N3] = number of chunks
    const int* ch_t0;     // first entry of a chunk
    unsigned l3_off;      // byte offset in dynamic LDS of e3a | e3b | r3 | chunk sums
    int nch_max;
    int J, nu, isnsp, wcycle, anycycle, maxit;
    double retol;
    double *e1, *e1b, *w1, *r1, *e2, *e2b, *r2;   // vectors of levels 1 and 2 (global)
};

struct MidRow {
    unsigned cp[MID_RC / 2];   // columns (< 65536), two per register
    double v[MID_RC];
    int t0, len;
    double dinv, axi;
};
__device__ __forceinline__ int mid_col(MidRow& R, int u) {
    // opaque: keeps the unpacked columns from being hoisted out of the solve loops (a register each)
    if (!(u & 1)) asm volatile("" : "+v"(R.cp[u >> 1]));
    return (int)((u & 1) ? (R.cp[u >> 1] >> 16) : (R.cp[u >> 1] & 0xffffu));
}

// chunk tables of a level: row_ch = exclusive scan of ceil(len/MID_CH), ch_t0 from it
__global__ void k_mid_chunk_counts(int N, const int* __restrict__ rp, int* __restrict__ cnt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        cnt[i] = (rp[i + 1] - rp[i] + MID_CH - 1) / MID_CH;
}
__global__ void k_mid_chunk_fill(int N, const int* __restrict__ rp, const int* __restrict__ row_ch,
                                 int* __restrict__ ch_t0) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int c = row_ch[i];
        for (int t = rp[i]; t < rp[i + 1]; t += MID_CH) ch_t0[c++] = t;
    }
}

// partial dot product of one chunk (entries [t0, t1), at most MID_CH of them) with x in LDS
__device__ __forceinline__ double mid_chunk_dot(const int* __restrict__ ci, const double* __restrict__ va,
                                                int t0, int t1, AS3 const double* x) {
    double s = 0.0;
    for (int t = t0; t < t1; t += 8) {
        int cc[8];
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + u < t1 ? t + u : t0;
            cc[u] = ci[tt];
            vv[u] = va[tt];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (t + u < t1) ? vv[u] * x[cc[u]] : 0.0;
    }
    return s;
}

// (the level's arrays are passed as plain pointers: a descriptor taken by reference through the
// kernel's lambdas ended up in scratch memory)
__device__ __forceinline__ MidRow mid_row_load(const int* __restrict__ rp, const int* __restrict__ ci,
                                               const double* __restrict__ va,
                                               const double* __restrict__ dinv,
                                               const double* __restrict__ Axi, int row, bool valid) {
    MidRow R;
    const int rr = valid ? row : 0;
    R.t0 = rp[rr];
    R.len = valid ? rp[rr + 1] - R.t0 : 0;
    int cu[MID_RC];
#pragma unroll
    for (int u = 0; u < MID_RC; ++u) {
        const bool in = u < R.len;
        const int cc = ci[in ? R.t0 + u : R.t0];
        const double vv = va[in ? R.t0 + u : R.t0];
        cu[u] = in ? cc : 0;
        R.v[u] = in ? vv : 0.0;
    }
#pragma unroll
    for (int u = 0; u < MID_RC / 2; ++u) R.cp[u] = (unsigned)cu[2 * u] | ((unsigned)cu[2 * u + 1] << 16);
    R.dinv = dinv[rr];
    R.axi = Axi[rr];
    return R;
}

// row . x with x in LDS; SEL: columns in [u0,u1) read xw instead (second half of a Gauss-Seidel
// sweep: the first half's result without the shift), ZERO: the other columns read 0 (zero start)
template <bool SEL>
__device__ __forceinline__ double mid_row_dot(const int* __restrict__ gci, const double* __restrict__ gva,
                                              MidRow& R, const double* x, const double* xw, int u0,
                                              int u1, bool zero) {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < MID_RC; ++u) {
        const int c = mid_col(R, u);
        double xv;
        if (SEL) {
            const bool inw = c >= u0 && c < u1;
            const double* src = inw ? xw : x;
            xv = src[c];
            if (zero && !inw) xv = 0.0;
        } else {
            xv = x[c];
        }
        const double term = R.v[u] * xv;
        s = (u < R.len) ? s + term : s;
    }
    for (int t = R.t0 + MID_RC; t < R.t0 + R.len; ++t) {   // rare long rows: the rest from L2
        const int c = gci[t];
        double xv;
        if (SEL) {
            const bool inw = c >= u0 && c < u1;
            xv = inw ? xw[c] : (zero ? 0.0 : x[c]);
        } else {
            xv = x[c];
        }
        s += gva[t] * xv;
    }
    return s;
}

// CSR row (global) . x (global or LDS), sequential
template <class XP>
__device__ __forceinline__ double mid_csr_dot(const int* __restrict__ rp, const int* __restrict__ ci,
                                              const double* __restrict__ va, int row, bool valid,
                                              XP x) {
    double s = 0.0;
    if (valid) {
        int t = rp[row];
        const int end = rp[row + 1];
        for (; t + 4 <= end; t += 4) {
            const int c0 = ci[t], c1 = ci[t + 1], c2 = ci[t + 2], c3 = ci[t + 3];
            const double v0 = va[t], v1 = va[t + 1], v2 = va[t + 2], v3 = va[t + 3];
            s += v0 * x[c0];
            s += v1 * x[c1];
            s += v2 * x[c2];
            s += v3 * x[c3];
        }
        for (; t < end; ++t) s += va[t] * x[ci[t]];
    }
    return s;
}

// block total of a per-thread value: one barrier; every thread gets the sum (wave order)
__device__ __forceinline__ double mid_block_sum(double v, AS3 double* part) {
    const double w = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = w;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < BT / 64; ++k) s += part[k];
    __syncthreads();   // part is reused by the next reduction
    return s;
}

// out[] as k_solve_small: out[0] = it, out[1] = rel_res, out[2] = res0; rel_resk at out[4 ..],
// rhok at out[4+maxit+2 ..].  fixed_cycles > 0: that many loop bodies, no stopping rules.
template <bool L3C>   // L3C: level 3 chunked here, the LDS image starts at level 4; else at level 3
__global__ __launch_bounds__(BT) void k_solve_mid(const SolveDesc* __restrict__ D_global, const MidDesc M,
                                                  const double* __restrict__ b, double* xg, double* out,
                                                  int fixed_cycles) {
    __shared__ PhaseLds lds;
    __shared__ double red[16];
    __shared__ double blkpart[48 + SOLVE_ML + 1];
    __shared__ double midpart[16];
    extern __shared__ __attribute__((aligned(16))) char dyn_raw[];
    SolveDesc* LD = sol_load_image(D_global, dyn_raw);   // levels 3..J: constants + work vectors
    SolveCtx c;
    c.D = LD;
    c.lds = &lds;
    c.red = red;
    c.xs = reinterpret_cast<double*>(dyn_raw);
    c.swapmask = 0;
    c.zeromask = 0;
    c.part = blkpart;
    c.sumr = blkpart + 48;
    c.dbg = nullptr;
    AS3 double* part = as_lds(midpart);
    const int tid = threadIdx.x;
    const int N1 = M.L1.N, N2 = M.L2.N, nf = M.L1.nf, N3 = M.L2.Nc;
    const int nu = M.nu, J = M.J, maxit = M.maxit;
    const bool nsp = M.isnsp != 0, wcyc = M.wcycle != 0, anyc = M.anycycle != 0;
    const double retol = M.retol;
    const int* ci1 = M.L1.ci;
    const double* va1 = M.L1.va;
    const int* ci2 = M.L2.ci;
    const double* va2 = M.L2.va;
    const int *R1rp = M.L1.Rrp, *R1ci = M.L1.Rci, *P1rp = M.L1.Prp, *P1ci = M.L1.Pci;
    const double *R1va = M.L1.Rva, *P1va = M.L1.Pva;
    const int *R2rp = M.L2.Rrp, *R2ci = M.L2.Rci, *P2rp = M.L2.Prp, *P2ci = M.L2.Pci;
    const double *R2va = M.L2.Rva, *P2va = M.L2.Pva;
    // vectors of levels 1 and 2 (global; this workgroup is their only user)
    double* e1 = M.e1;      // iterate of level 1
    double* e1b = M.e1b;    // its ping-pong partner
    double* w1 = M.w1;      // first-half results / x + e / residual
    double* r1 = M.r1;      // right-hand side of level 1 (the outer residual)
    double* e2 = M.e2;
    double* e2b = M.e2b;
    double* r2 = M.r2;
    // level 3: right-hand side in the image (root of the sub-cycle) or, chunked, vectors of its own
    AS3 double* e3 = (AS3 double*)(dyn_raw + M.l3_off);
    AS3 double* e3b = e3 + M.L3.N;
    AS3 double* r3 = L3C ? e3b + M.L3.N : as_lds(LD->L[3].lv.r);
    AS3 double* psum = e3b + 2 * M.L3.N;
    AS3 double* r4 = as_lds(LD->L[L3C ? 4 : 3].lv.r);
    const int N4 = M.L3.Nc;
    const int* rp3 = M.L3.rp;
    const int* ci3 = M.L3.ci;
    const double* va3 = M.L3.va;
    const double *dinv3 = M.L3.dinv, *axi3 = M.L3.Axi;
    const int *R3rp = M.L3.Rrp, *R3ci = M.L3.Rci, *P3rp = M.L3.Prp, *P3ci = M.L3.Pci;
    const double *R3va = M.L3.Rva, *P3va = M.L3.Pva;
    const int* row_ch = M.row_ch;
    const int* ch_t0 = M.ch_t0;
    const int nch = L3C ? row_ch[N3] : 0;
    const double xx3 = (L3C && nsp) ? M.L3.xx[0] : 1.0;
    double sumr3 = 0.0, axe3 = 0.0;
    bool ez3 = true;
    (void)r4;

    // ---- rows of this thread -> registers -------------------------------------------------------
    MidRow A1[MID_RPT], A2[MID_RPT2];
#pragma unroll
    for (int j = 0; j < MID_RPT; ++j) {
        const int row = tid + j * BT;
        A1[j] = mid_row_load(M.L1.rp, ci1, va1, M.L1.dinv, M.L1.Axi, row, row < N1);
    }
#pragma unroll
    for (int j = 0; j < MID_RPT2; ++j) {
        const int row = tid + j * BT;
        A2[j] = mid_row_load(M.L2.rp, ci2, va2, M.L2.dinv, M.L2.Axi, row, row < N2);
    }
    const double xx1 = nsp ? M.L1.xx[0] : 1.0, xx2 = nsp ? M.L2.xx[0] : 1.0;
    double sumr1 = 0.0, axe1 = 0.0;   // 1'r and (A1)'e of level 1 (kernel-space correction)
    double sumr2 = 0.0, axe2 = 0.0;
    bool ez1 = true, ez2 = true;      // the iterate is identically zero (not materialised)

    // r1 = b - A (x [+ e1]); x <- x [+ e1]; returns ||r1||                 Class_AMG.m:89,96-103
    auto top = [&](bool add) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < MID_RPT; ++j) {
            const int row = tid + j * BT;
            if (row < N1) {
                const double xo = add ? xg[row] + e1[row] : xg[row];
                w1[row] = xo;
                if (add) xg[row] = xo;
            }
        }
        __syncthreads();
        double n2 = 0.0, sr = 0.0;
#pragma unroll
        for (int j = 0; j < MID_RPT; ++j) {
            const int row = tid + j * BT;
            if (row < N1) {
                const double s = mid_row_dot<false>(ci1, va1, A1[j], w1, w1, 0, 0, false);
                const double ri = b[row] - s;
                r1[row] = ri;
                n2 += ri * ri;
                sr += ri;
            }
        }
        const double tot = mid_block_sum(n2, part);
        sumr1 = nsp ? mid_block_sum(sr, part) : 0.0;
        ez1 = true;
        axe1 = 0.0;
        return sqrt(tot);
    };

    // one bigraph Gauss-Seidel sweep on level 1 (nf > 0) or Jacobi sweep (nf == 0)
    auto sweep1 = [&](bool post) __attribute__((always_inline)) {
        const double cc = nsp ? (sumr1 - (ez1 ? 0.0 : axe1)) / xx1 : 0.0;      // MG_Vcycle.m:18-19
        double acc = 0.0;
        if (nf == 0) {
#pragma unroll
            for (int j = 0; j < MID_RPT; ++j) {
                const int row = tid + j * BT;
                if (row < N1) {
                    const double eo = ez1 ? 0.0 : e1[row];
                    const double sd = ez1 ? 0.0 : mid_row_dot<false>(ci1, va1, A1[j], e1, e1, 0, 0, false);
                    const double v = eo + A1[j].dinv * (r1[row] - sd - A1[j].axi * cc) + cc;
                    e1b[row] = v;
                    acc += A1[j].axi * v;
                }
            }
        } else {
            const int f0 = post ? nf : 0, f1 = post ? N1 : nf;     // first half rows
#pragma unroll
            for (int j = 0; j < MID_RPT; ++j) {
                const int row = tid + j * BT;
                if (row < N1 && row >= f0 && row < f1) {
                    const double eo = ez1 ? 0.0 : e1[row];
                    const double sd = ez1 ? 0.0 : mid_row_dot<false>(ci1, va1, A1[j], e1, e1, 0, 0, false);
                    const double wv = eo + A1[j].dinv * (r1[row] - sd - A1[j].axi * cc);
                    w1[row] = wv;
                    const double v = wv + cc;
                    e1b[row] = v;
                    acc += A1[j].axi * v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MID_RPT; ++j) {
                const int row = tid + j * BT;
                if (row < N1 && !(row >= f0 && row < f1)) {
                    const double eo = ez1 ? 0.0 : e1[row];
                    const double sd = mid_row_dot<true>(ci1, va1, A1[j], e1, w1, f0, f1, ez1);
                    const double v = eo + A1[j].dinv * (r1[row] - sd - A1[j].axi * cc) + cc;
                    e1b[row] = v;
                    acc += A1[j].axi * v;
                }
            }
        }
        axe1 = nsp ? mid_block_sum(acc, part) : 0.0;
        if (!nsp) __syncthreads();
        double* t = e1;
        e1 = e1b;
        e1b = t;
        ez1 = false;
    };

    auto sweep2 = [&]() __attribute__((always_inline)) {
        const double cc = nsp ? (sumr2 - (ez2 ? 0.0 : axe2)) / xx2 : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < MID_RPT2; ++j) {
            const int row = tid + j * BT;
            if (row < N2) {
                const double eo = ez2 ? 0.0 : e2[row];
                const double sd = ez2 ? 0.0 : mid_row_dot<false>(ci2, va2, A2[j], e2, e2, 0, 0, false);
                const double v = eo + A2[j].dinv * (r2[row] - sd - A2[j].axi * cc) + cc;
                e2b[row] = v;
                acc += A2[j].axi * v;
            }
        }
        axe2 = nsp ? mid_block_sum(acc, part) : 0.0;
        if (!nsp) __syncthreads();
        double* t = e2;
        e2 = e2b;
        e2b = t;
        ez2 = false;
    };

    // ---- chunked level 3 (L3C) ---------------------------------------------------------------
    // A3 * x: chunk sums by all threads, then (after the barrier) a row's chunks added in order
    auto l3_chunks = [&](AS3 const double* x) __attribute__((always_inline)) {
        for (int k = tid; k < nch; k += BT) {
            const int t0 = ch_t0[k];
            // the chunk ends at the next chunk's start or at the end of the matrix
            const int t1 = k + 1 < nch ? ch_t0[k + 1] : rp3[N3];
            psum[k] = mid_chunk_dot(ci3, va3, t0, min(t1, t0 + MID_CH), x);
        }
        __syncthreads();
    };
    auto l3_row = [&](int row) __attribute__((always_inline)) {
        double sd = 0.0;
        for (int k = row_ch[row]; k < row_ch[row + 1]; ++k) sd += psum[k];
        return sd;
    };
    auto sweep3 = [&]() __attribute__((always_inline)) {
        const double cc = nsp ? (sumr3 - (ez3 ? 0.0 : axe3)) / xx3 : 0.0;
        if (!ez3) l3_chunks(e3);
        double acc = 0.0;
        for (int row = tid; row < N3; row += BT) {
            const double eo = ez3 ? 0.0 : e3[row];
            const double sd = ez3 ? 0.0 : l3_row(row);
            const double ax = axi3[row];
            const double v = eo + dinv3[row] * (r3[row] - sd - ax * cc) + cc;
            e3b[row] = v;
            acc += ax * v;
        }
        axe3 = nsp ? mid_block_sum(acc, part) : 0.0;
        if (!nsp) __syncthreads();
        AS3 double* t = e3;
        e3 = e3b;
        e3b = t;
        ez3 = false;
    };
    auto visit3 = [&](bool keep) __attribute__((always_inline)) {
        if (!keep) {
            ez3 = true;
            if (nu == 0) {
                for (int row = tid; row < N3; row += BT) e3[row] = 0.0;
                ez3 = false;
                axe3 = 0.0;
                __syncthreads();
            }
        }
        for (int s = 0; s < nu; ++s) sweep3();
        l3_chunks(e3);                                                     // rr = r - A e  :27
        for (int row = tid; row < N3; row += BT) e3b[row] = r3[row] - l3_row(row);
        __syncthreads();
        for (int i = tid; i < N4; i += BT) r4[i] = mid_csr_dot(R3rp, R3ci, R3va, i, true, e3b);
        __syncthreads();
        for (int v4 = 0; v4 < ((wcyc && 4 < J) ? 2 : 1); ++v4) sol_cycle(c, 4, v4 == 1);
        {
            AS3 const double* e4 = lds_e(c, 4);
            double acc = 0.0;
            for (int row = tid; row < N3; row += BT) {
                const double v = e3[row] + mid_csr_dot(P3rp, P3ci, P3va, row, true, e4);
                e3[row] = v;
                acc += axi3[row] * v;
            }
            axe3 = nsp ? mid_block_sum(acc, part) : 0.0;
            if (!nsp) __syncthreads();
        }
        for (int s = 0; s < nu; ++s) sweep3();
    };
    (void)visit3;

    // one visit of level 2 and everything below it                          MG_Vcycle.m:12-41
    auto visit2 = [&](bool keep) __attribute__((always_inline)) {
        if (!keep) {
            ez2 = true;
            if (nu == 0) {
                for (int row = tid; row < N2; row += BT) e2[row] = 0.0;
                ez2 = false;
                axe2 = 0.0;
                __syncthreads();
            }
        }
        for (int s = 0; s < nu; ++s) sweep2();
        // rr = r - A e (into the free buffer), r3 = P' rr                               :27
#pragma unroll
        for (int j = 0; j < MID_RPT2; ++j) {
            const int row = tid + j * BT;
            if (row < N2) e2b[row] = r2[row] - mid_row_dot<false>(ci2, va2, A2[j], e2, e2, 0, 0, false);
        }
        __syncthreads();
        {
            double sr = 0.0;
            for (int i = tid; i < N3; i += BT) {
                const double v = mid_csr_dot(R2rp, R2ci, R2va, i, true, e2b);
                r3[i] = v;
                sr += v;
            }
            if (L3C && nsp)
                sumr3 = mid_block_sum(sr, part);
            else
                __syncthreads();
        }
        AS3 const double* e3res;
        if constexpr (L3C) {
            for (int v3 = 0; v3 < ((wcyc && 3 < J) ? 2 : 1); ++v3) visit3(v3 == 1);        // :29 / MG_Wcycle.m:30
            e3res = e3;
        } else {
            // :29, and MG_Wcycle.m:30's second correction (one call site: the sub-cycle is large)
            for (int v3 = 0; v3 < ((wcyc && 3 < J) ? 2 : 1); ++v3) sol_cycle(c, 3, v3 == 1);
            e3res = lds_e(c, 3);
        }
        {   // e2 += P e3                                                                :31
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < MID_RPT2; ++j) {
                const int row = tid + j * BT;
                if (row < N2) {
                    const double v = e2[row] + mid_csr_dot(P2rp, P2ci, P2va, row, true, e3res);
                    e2[row] = v;
                    acc += A2[j].axi * v;
                }
            }
            axe2 = nsp ? mid_block_sum(acc, part) : 0.0;
            if (!nsp) __syncthreads();
        }
        for (int s = 0; s < nu; ++s) sweep2();
    };

    // MG_Vcycle / MG_Wcycle from level 1 down; the correction ends in e1
    auto cycle = [&]() __attribute__((always_inline)) {
        ez1 = true;
        if (nu == 0) {
            for (int row = tid; row < N1; row += BT) e1[row] = 0.0;
            ez1 = false;
            axe1 = 0.0;
            __syncthreads();
        }
        for (int s = 0; s < nu; ++s) sweep1(false);
#pragma unroll
        for (int j = 0; j < MID_RPT; ++j) {
            const int row = tid + j * BT;
            if (row < N1) w1[row] = r1[row] - mid_row_dot<false>(ci1, va1, A1[j], e1, e1, 0, 0, false);
        }
        __syncthreads();
        {
            double sr = 0.0;
#pragma unroll
            for (int j = 0; j < MID_RPT2; ++j) {
                const int row = tid + j * BT;
                if (row < N2) {
                    const double v = mid_csr_dot(R1rp, R1ci, R1va, row, true, w1);
                    r2[row] = v;
                    sr += v;
                }
            }
            sumr2 = nsp ? mid_block_sum(sr, part) : 0.0;
            if (!nsp) __syncthreads();
        }
        for (int leg = 0; leg < (wcyc ? 2 : 1); ++leg) visit2(leg == 1);
        {   // e1 += P e2
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < MID_RPT; ++j) {
                const int row = tid + j * BT;
                if (row < N1) {
                    const double v = e1[row] + mid_csr_dot(P1rp, P1ci, P1va, row, true, e2);
                    e1[row] = v;
                    acc += A1[j].axi * v;
                }
            }
            axe1 = nsp ? mid_block_sum(acc, part) : 0.0;
            if (!nsp) __syncthreads();
        }
        for (int s = 0; s < nu; ++s) sweep1(true);
    };

    // ---- Class_AMG.m:86-109 ---------------------------------------------------------------------
    double* relk = out + 4;
    double* rhok = out + 4 + (maxit + 2);
    const bool fixed = fixed_cycles > 0;
    int it = 0, done = 0;
    double res = 0.0, res0 = 0.0, rel_res = 0.0, last_rel = 1.0;
    bool first = true;
    for (;;) {
        const double rnow = top(!first && anyc);
        if (first) {
            first = false;
            res0 = res = rnow;
            if (!fixed) {
                if (res0 == 0.0) {
                    if (tid == 0) {
                        relk[0] = 0.0;
                        rhok[0] = INFINITY;
                    }
                    break;
                }
                it = 1;
                if (tid == 0) {
                    relk[0] = 1.0;
                    rhok[0] = NAN;
                }
            }
        } else {
            const double prev = res;
            res = rnow;
            rel_res = res / res0;
            const double rho = res / prev;
            if (fixed) {
                ++done;
            } else {
                if (tid == 0) {
                    relk[it] = rel_res;
                    rhok[it] = rho;
                }
                last_rel = rel_res;
                ++it;
                if (rho > 1.0) break;
            }
        }
        if (fixed ? done >= fixed_cycles : !(last_rel > retol && it <= maxit)) break;
        if (anyc) cycle();
    }
    if (fixed)
        it = fixed_cycles;
    else if (res0 != 0.0)
        it -= 1;
    if (tid == 0) {
        out[0] = (double)it;
        out[1] = rel_res;
        out[2] = res0;
        out[3] = 0.0;
    }
}
