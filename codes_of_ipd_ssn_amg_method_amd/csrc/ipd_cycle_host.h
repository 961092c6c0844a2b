// Host-side orchestration of the cycle (included by ipd_cycle.hip only): launch
// geometry, the recursive V/W schedule, the Class_AMG loop, the C ABI and the
// measurement hooks.
#pragma once

#include <cstdlib>

struct LevelRun {  // per-level run state kept next to Level
    LevelDev dev;
    bool e_zero = true;      // the iterate is identically zero and is not materialised
    int staged = 0;          // N <= STAGE_MAX: gather vectors go through LDS
    int maxoff = 0;          // longest off-diagonal row (k_level_prepare)
    XferArgs restrict_args;  // r_{k+1} = P' rr_k   (stored on level k)
    XferArgs prolong_args;   // e_k += P e_{k+1}
    PcgArgs pcg;             // coarsest only
};

struct CycleState {
    std::vector<LevelRun> run;  // 1-based
    double* hist = nullptr;
    // row-block sharding: `shard_ranks` owners per row range.  In emulate mode
    // one process plays all owners back to back on the shared vectors (the all-gather is
    // then implicit) -- used by the single-GPU test of the slicing logic.
    int shard_ranks = 1;
    int shard_rank = 0;
    bool shard_emulate = false;
    int shard_min_rows = 256;
    int num_cu = 256;
    // fused single-workgroup program under construction (flushed before any big launch)
    FusedProg pending;
    size_t pending_lds = 0;
    bool fuse_enabled = true;
    CycleState() { pending.n = 0; }
    // single-workgroup whole-solve kernel (small hierarchies): device descriptor + outputs
    SolveDesc* d_solve = nullptr;
    double* solve_out = nullptr;
    size_t solve_lds = 0;
    bool small_ok = false;
    bool solve_cached = false;
    // matrix-free level-1 operator (bit mask + scale vectors), see k_smooth_mask
    bool mask_ok = false;
    MaskOp maskop{};
    // single-workgroup sub-cycle rooted at level k_sub (0 = none), see k_subcycle
    SolveDesc* d_sub = nullptr;
    bool sub_semi_root = false;    // d_sub's root level is semi-cached (rows from L2)
    std::vector<int> level_forms;  // per level, over all images packed: see ipd_amg_level_forms
    struct PolyOp {                // block-wide polynomial operators packed for the images (ipd_amg_poly_operator)
        const double* M = nullptr;
        const double* W = nullptr;
        int LD = 0, N = 0, Nc = 0;
    };
    std::vector<PolyOp> poly_ops;
    SolveDesc* d_sub4 = nullptr;   // image rooted at level 4 for the resident kernel's `three` mode
    size_t sub4_lds = 0;           // (packed beside d_sub when that one is rooted at level 3)
    SolveDesc* d_sub3 = nullptr;   // image rooted at level 3 for the resident kernel alone (k_sub == 0)
    size_t sub3_lds = 0;
    SolveDesc* d_sub5 = nullptr;   // image rooted at level 5 for the mask-form kernel's deep mode with level 4 resident
    size_t sub5_lds = 0;
    int k_sub = 0;
    size_t sub_lds = 0;
    // dynamic LDS an image's operator copy needs on top of its *_lds (SolveDesc::bm_src; 0: none)
    size_t sub_bm = 0, sub3_bm = 0, sub4_bm = 0;
    double* x2 = nullptr;
    // level-resident solve kernel (ipd_resident.h): the whole Class_AMG loop in one launch of
    // res_G co-resident workgroups that keep the matrices of levels 1-2 in registers
    bool res_ok = false;
    bool res_remote = false;   // levels >= 3 served by a tail workgroup (see ResDesc::remote)
    int res_ke3 = 0;           // > 0: level 3 resident as well (entries per lane of its rows), tail rooted at 4
    size_t res_block_bytes = 0;
    ResDesc res_desc{};
    int res_G = 0;
    int res_ke = 0;          // entries per lane of a padded row (template argument)
    size_t res_lds = 0;
    double* res_out = nullptr;
    unsigned char* res_block = nullptr;   // [gran0 | gran1 | tmo]: zeroed before every launch
    int res_timeouts = 0;    // launches whose bounded spins gave up (then: multi-launch path)
    long long res_last_handoffs = 0;   // hand-offs and cycles of the last launch (ipd_amg_resident_kernel)
    int res_last_cycles = 0;
    int res_capacity = -1;   // workgroups of the chosen instantiation the device holds at once (-1: not asked yet)
    // level 1 of up to 4096 rows: the mask-form resident kernel (ipd_resident_big.h), set up by
    // amg_attach_maskop once the bit mask of level 1 is there
    bool resb = false;
    bool res_off = false;    // IPD_NO_RESIDENT=1 when the hierarchy was set up
    ResBigDesc resb_desc;
    int resb_ke2 = 16;
    bool res_poly2 = false;  // k_resident<16,16,0,true>: level 2 composed over a visit (ipd_amg_attach_level2_poly)
    bool resb_deep = false;  // realistic hierarchy: level 3 in polynomial form, remote tail at level 4 (RPW = 2)
    bool resb_poly4 = false; // ... level 4 in polynomial form as well, remote tail at level 5
    hipGraphExec_t gexec[2] = {nullptr, nullptr};  // captured Class_AMG loop bodies (x->x2, x2->x)
    const double* gb = nullptr;                    // right-hand side the graphs were captured for
    ~CycleState() {
        for (auto& g : gexec)
            if (g) (void)hipGraphExecDestroy(g);
    }
};

static CycleState* state_of(ipd_amg* h) { return h->cyc.get(); }

__global__ void k_level_prepare(int N, int nf, const int* __restrict__ rp,
                                const int* __restrict__ ci, const double* __restrict__ va,
                                double* __restrict__ dinv, double* __restrict__ Axi,
                                int* __restrict__ maxoff) {
    // one wave per row: diagonal -> Rk, row sum -> A*1, longest off-diagonal row -> maxoff
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    int longest = 0;
    for (int r = wave; r < N; r += nwaves) {
        double s = 0.0, dg = 0.0;
        int hasd = 0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            s += va[t];
            if (ci[t] == r) {
                dg = va[t];
                hasd = 1;
            }
        }
        s = wave_sum(s);
        dg = wave_sum(dg);
        hasd = __any(hasd) ? 1 : 0;
        longest = max(longest, rp[r + 1] - rp[r] - hasd);
        if (lane == 0) {
            Axi[r] = s;
            // Class_AMG.m:56-59 (1./diag) for the bigraph GS, :72/:84 (0.5*(1./diag)) otherwise
            dinv[r] = nf > 0 ? 1.0 / dg : 0.5 * (1.0 / dg);
        }
    }
    // (thousands of waves on one address: 35 of the kernel's 41 us were this atomic; most waves find the
    // maximum already there)
    if (lane == 0 && longest > 0 && longest > __hip_atomic_load(maxoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(maxoff, longest);
}

// All levels of a hierarchy in two launches instead of two per level (the hierarchy is rebuilt at every
// Newton step): workgroup b of k_levels_prepare belongs to the level whose block range holds b, workgroup
// k of k_levels_sum adds A*1 of level k in k_vec_sum's order (same bits).
constexpr int PREP_ML = 24;
struct PrepLevels {
    int n;
    int first_block[PREP_ML + 1];
    int N[PREP_ML], nf[PREP_ML];
    const int* rp[PREP_ML];
    const int* ci[PREP_ML];
    const double* va[PREP_ML];
    double* dinv[PREP_ML];
    double* Axi[PREP_ML];
    double* xx[PREP_ML];
    int* maxoff[PREP_ML];
};
__global__ void k_levels_prepare(const PrepLevels P) {
    int k = 0;
    while (k + 1 < P.n && (int)blockIdx.x >= P.first_block[k + 1]) ++k;
    const int nb = P.first_block[k + 1] - P.first_block[k], lb = blockIdx.x - P.first_block[k];
    const int N = P.N[k], nf = P.nf[k];
    const int* __restrict__ rp = P.rp[k];
    const int* __restrict__ ci = P.ci[k];
    const double* __restrict__ va = P.va[k];
    const int lane = threadIdx.x & 63;
    const int wave = (lb * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (nb * blockDim.x) >> 6;
    int longest = 0;
    for (int r = wave; r < N; r += nwaves) {
        double s = 0.0, dg = 0.0;
        int hasd = 0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            s += va[t];
            if (ci[t] == r) {
                dg = va[t];
                hasd = 1;
            }
        }
        s = wave_sum(s);
        dg = wave_sum(dg);
        hasd = __any(hasd) ? 1 : 0;
        longest = max(longest, rp[r + 1] - rp[r] - hasd);
        if (lane == 0) {
            P.Axi[k][r] = s;
            P.dinv[k][r] = nf > 0 ? 1.0 / dg : 0.5 * (1.0 / dg);
        }
    }
    // one atomic per workgroup (one per wave on one address cost 35 of the kernel's 41 us)
    __shared__ int bmax;
    if (threadIdx.x == 0) bmax = 0;
    __syncthreads();
    if (lane == 0 && longest > 0) atomicMax(&bmax, longest);
    __syncthreads();
    if (threadIdx.x == 0 && bmax > 0) atomicMax(P.maxoff[k], bmax);
}
__global__ __launch_bounds__(BT) void k_levels_sum(const PrepLevels P) {
    __shared__ double red[16];
    const int k = blockIdx.x;
    const double* v = P.Axi[k];
    double s = 0.0;
    for (int t = threadIdx.x; t < P.N[k]; t += BT) s += v[t];
    const double tot = block_sum(s, red);
    if (threadIdx.x == 0) P.xx[k][0] = tot;
}

static int pick_blocks(int nrows, int L, int cu);

// Builds the padded off-diagonal copy when the level is big and regular enough
// (see ipd_cycle_phases.h, item 2) and adapts the launch geometry to it.
static void pad_flush(ipd_ctx* ctx, PadBatch* b) {
    if (b->n == 0) return;
    int rows = 1;
    for (int q = 0; q < b->n; ++q) rows = std::max(rows, b->N[q]);
    hipLaunchKernelGGL(k_pad_build_batch, dim3(std::max(1, std::min(cdiv(rows, 4), 4096)), b->n), dim3(256), 0,
                       ctx->stream, *b);
    IPD_KERNEL_CHECK();
    b->n = 0;
}
static void build_padded(ipd_ctx* ctx, Arena& ar, const Csr& A, int rows_per_launch, int cu,
                         int maxlen /* longest off-diagonal row, from k_level_prepare */,
                         LevelDev* dev, PadBatch* batch) {
    dev->S = 0;
    dev->pci = nullptr;
    dev->pva = nullptr;
    dev->diag = nullptr;
    const char* env = std::getenv("IPD_NO_PAD");
    if (env && env[0] == '1') return;
    if (A.nr > 65535 || A.nr == 0) return;
    const double avg_off = (double)(A.nnz - A.nr) / (double)A.nr;
    // small levels too: one dependent round trip less per launch (measured -6 % solve time on
    // the m=n=1024 Class 1 run)
    if (avg_off < 0.5) return;
    const int S = (maxlen + 3) / 4 * 4;
    if (S == 0 || (double)S > 1.3 * avg_off + 16.0) return;
    unsigned short* pci = ar.alloc<unsigned short>((size_t)A.nr * S);
    double* pva = ar.alloc<double>((size_t)A.nr * S);
    double* diag = ar.alloc<double>((size_t)A.nr);
    {   // (launched with the other levels' copies: pad_flush)
        if (batch->n == PAD_BATCH) pad_flush(ctx, batch);
        const int q = batch->n++;
        batch->N[q] = A.nr;
        batch->S[q] = S;
        batch->rp[q] = A.rp;
        batch->ci[q] = A.ci;
        batch->va[q] = A.va;
        batch->pci[q] = pci;
        batch->pva[q] = pva;
        batch->diag[q] = diag;
    }
    dev->S = S;
    dev->pci = pci;
    dev->pva = pva;
    dev->diag = diag;
    // one batch (ROW_U entries = 2 vectors) per lane, widened until the chip is filled
    const int nvec = S / 4;
    int L = 4;
    // batches per lane aimed at before the chip-filling rule below widens again.  Regime D at
    // m=n=2048 (2048-entry rows, bandwidth-bound): 1: 0.376 ms per V cycle, 2: 0.342, 4: 0.332,
    // 8/16: 0.332; m=n=1024 unchanged (0.197), m=n=4096 Class 1 run 5.37 -> 5.29 s
    // (with 512-thread blocks: 4: 0.337, 8: 0.324-0.330, 16: 0.325)
    const int batches = 8;
    while (L < BT && L * (ROW_U / 4) * batches < nvec) L <<= 1;
    const double fill = 1.0;   // one workgroup per CU (0.5 left half the chip idle on a 1024-row level: 6.16 -> 5.79 us)
    while (L < BT && (double)rows_per_launch * L < fill * cu * BT && L < nvec) L <<= 1;
    dev->L = L;
    dev->G = pick_blocks(rows_per_launch, L, cu);
}

static int pick_blocks(int nrows, int L, int cu) {
    return (int)std::max<long long>(1, std::min<long long>(cu, ((long long)nrows * L + BT - 1) / BT));
}


// Packs the polynomial form of level k (k_bpoly_*, ipd_cycle.hip) into the hierarchy's arena: LD-row
// column-major [Mr | Me | Mc] for the single-workgroup images, or (rows) the row-major layout the
// resident kernels' third level takes.
struct BPolyDev {
    double* M = nullptr;
    double* W = nullptr;
    int LD = 0;
    BPolyEntry e{};   // the pack's operands (scratch: valid until the call scope ends)
};
static BPolyDev pack_bpoly(ipd_ctx* ctx, ipd_amg* h, CycleState* st, int k, int isnsp, int LD, bool rows,
                           int rows_seg = 512, int rows_ld = RES_P3_LD) {
    Arena& ar = *h->arena;
    BPolyDev b;
    const Level& lv = h->L[k];
    const Csr& P = h->L[k + 1].P;
    const LevelDev& gd = st->run[(size_t)k].dev;
    const size_t N = (size_t)lv.A.nr, Nc = (size_t)P.nc, N8 = (N + 7) / 8 * 8, Nc8 = (Nc + 7) / 8 * 8;
    const size_t Np = (N + 15) / 16 * 16, Ncp = (Nc + 15) / 16 * 16, xcols = 2 * Np + 16;
    BPolyEntry e;
    e.Arp = lv.A.rp;
    e.Aci = lv.A.ci;
    e.Ava = lv.A.va;
    e.Prp = P.rp;
    e.Pci = P.ci;
    e.Pva = P.va;
    e.dinv = gd.dinv;
    e.Axi = gd.Axi;
    e.xx = gd.xx;
    e.N = (int)N;
    e.Nc = (int)Nc;
    e.Np = (int)Np;
    e.Ncp = (int)Ncp;
    e.nu = h->opts.smoth;
    e.isnsp = isnsp;
    e.LD = LD;
    // one zeroed block of scratch: A, S, P, T1, Pw[0], Pw[1], Y, dv, u, cs
    const size_t sc = 4 * Np * Np + 2 * Np * Ncp + Np * xcols + 3 * Np;
    double* blk = zeroed<double>(ctx, sc);
    e.A = blk;
    e.S = e.A + Np * Np;
    e.P = e.S + Np * Np;
    e.T1 = e.P + Np * Ncp;
    e.Pw[0] = e.T1 + Np * Ncp;
    e.Pw[1] = e.Pw[0] + Np * Np;
    e.Y = e.Pw[1] + Np * Np;
    e.dv = e.Y + Np * xcols;
    e.u = e.dv + Np;
    e.cs = e.u + Np;
    const size_t ncols = 2 * N8 + Nc8;
    const size_t out = rows ? (N + Nc) * (size_t)rows_ld + (N + Nc) : (size_t)LD * (ncols + 1);
    b.LD = LD;
    b.M = ar.alloc<double>(out);
    b.W = rows ? b.M + (N + Nc) * (size_t)rows_ld : b.M + (size_t)LD * ncols;
    IPD_HIP(hipMemsetAsync(b.M, 0, out * sizeof(double), ctx->stream));
    e.M = b.M;
    e.W = b.W;
    e.rows = rows ? b.M : nullptr;
    e.rows_seg = rows_seg;
    e.rows_ld = rows_ld;
    hipLaunchKernelGGL(k_bpoly_scatter, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, e);
    IPD_KERNEL_CHECK();
    hipLaunchKernelGGL(k_bpoly_colsum, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, e);
    IPD_KERNEL_CHECK();
    const int nS = (int)((N * N + 255) / 256), nT1 = (int)((Ncp / 16) * (Np / 16));   // one tile per workgroup
    hipLaunchKernelGGL(k_bpoly_S_T1, dim3((unsigned)(nS + nT1)), dim3(256), 0, ctx->stream, e, nS);
    IPD_KERNEL_CHECK();
    int cur = 0;
    const int nT = (int)((Np / 16) * (Np / 16));
    for (int s = 2; s <= e.nu; ++s) {
        const int nw = s == e.nu ? (int)((N + 3) / 4) : 0;
        hipLaunchKernelGGL(k_bpoly_step, dim3((unsigned)(nT + nw)), dim3(256), 0, ctx->stream, e, s, cur, nT);
        IPD_KERNEL_CHECK();
        cur ^= 1;
    }
    const int nZ = (int)((Ncp / 16) * (xcols / 16)), nC = (int)((Np / 16) * (Ncp / 16));
    const int nK = (int)((N * N + N + 255) / 256);
    hipLaunchKernelGGL(k_bpoly_final, dim3((unsigned)(nZ + nC + nK)), dim3(256), 0, ctx->stream, e, nZ, nC);
    IPD_KERNEL_CHECK();
    b.e = e;
    return b;
}

// ---- level-resident solve kernel: eligibility and launch ---------------------------------
// Eligible: three levels -- a bigraph Gauss-Seidel level 1 and a Jacobi level 2 with padded
// rows of at most 1024 entries, at most 2048 rows each, and a tail level of at most 64 rows --
// i.e. the dense regimes (regime D), where each launch of the multi-launch path is
// latency-bound.  IPD_NO_RESIDENT=1 switches it off, IPD_RESIDENT_G overrides the grid.
static void plan_resident(ipd_amg* h, CycleState* st) {
    st->res_ok = false;
    if (const char* e = std::getenv("IPD_NO_RESIDENT"); e && e[0] == '1') {
        st->res_off = true;   // (remembered: the mask-form kernel is set up later, by amg_attach_maskop)
        return;
    }
    if (st->small_ok || h->J < 3 || h->opts.twogrid) return;
    const Level& l1 = h->L[1];
    const Level& l2 = h->L[2];
    const Level& l3 = h->L[3];
    // Rows live in registers: the padded stride is only a layout here, so a level whose rows are too
    // uneven for the launches' padded copy (hubs) gets a private copy with its longest row as stride
    // (built below, once the hierarchy is known to be taken).  d1 / d2 carry the stride either way.
    LevelDev d1 = st->run[1].dev;
    LevelDev d2 = st->run[2].dev;
    const bool nopriv = false;
    bool priv1 = false, priv2 = false;
    if (d1.S <= 0 && st->run[1].maxoff > 0 && !nopriv) {
        d1.S = (st->run[1].maxoff + 3) / 4 * 4;
        priv1 = true;
    }
    if (d2.S <= 0 && st->run[2].maxoff > 0 && !nopriv) {
        d2.S = (st->run[2].maxoff + 3) / 4 * 4;
        priv2 = true;
    }
    const int N1 = l1.A.nr, N2 = l2.A.nr, Nt = l3.A.nr, nf = l1.nf, nc = N1 - nf;
    if (const char* dbg = std::getenv("IPD_DEBUG_LEVELS"); dbg && dbg[0] == '1')
        std::fprintf(stderr, "[ipd] resident plan: J=%d nf=%d nc=%d S1=%d S2=%d S3=%d N4=%d Nt=%d k_sub=%d sub_lds=%zu\n", h->J,
                     nf, nc, d1.S, d2.S, st->run[3].dev.S, h->J >= 4 ? h->L[4].A.nr : 0, Nt, st->k_sub, st->sub_lds);
    if (nf <= 0 || nc <= 0 || d1.S <= 0 || d2.S <= 0) return;
    if (N1 > 4 * BT || N2 > 4 * BT || nf > 2 * BT || nc > 2 * BT || Nt < 1) return;
    // everything below level 2: a tail of <= 64 rows solved redundantly by every workgroup (three
    // levels), or -- deeper hierarchies -- the single-workgroup sub-cycle rooted at level 3 run by ONE
    // extra workgroup out of its LDS image (the image the multi-launch path launches k_subcycle with)
    const bool local_tail = h->J == 3 && Nt <= RES_TAIL_MAX;
    bool remote = false, three = false;
    int ke3 = 0;
    const SolveDesc* tail_img = nullptr;   // the remote tail's LDS image and its dynamic LDS size
    size_t tail_lds = 0, tail_bm = 0;   // tail_bm: room for the image's operator copy (SolveDesc::bm_src)
    if (!local_tail) {
        const char* nr = std::getenv("IPD_NO_RESIDENT_REMOTE");
        const bool cyc = h->opts.cycle == 'w' || h->opts.cycle == 'v';
        remote = !(nr && nr[0] == '1') && h->J >= 4 && Nt <= BT && cyc &&
                 ((st->k_sub == 3 && st->d_sub) || (st->k_sub == 0 && st->d_sub3));
        // Level 3 in the registers of the resident workgroups as well, the tail rooted at level 4: for
        // a level 3 too big for the tail's LDS (a few hundred rows of 15-100 entries), and preferred to
        // the tail rooted at level 3 whenever an image rooted at level 4 exists (the tail's legs are the
        // serial part of a cycle: ~22 us each from level 4, ~100 us from level 3).
        const char* n3 = std::getenv("IPD_NO_RESIDENT_THREE");
        tail_img = st->k_sub == 0 ? st->d_sub3 : st->d_sub;
        tail_lds = st->k_sub == 0 ? st->sub3_lds : st->sub_lds;
        tail_bm = st->k_sub == 0 ? st->sub3_bm : st->sub_bm;
        const bool img4 = (st->k_sub == 4 && st->d_sub) || (st->k_sub == 3 && st->d_sub4);
        if (!(nr && nr[0] == '1') && !(n3 && n3[0] == '1') && h->J >= 5 && img4 && cyc) {
            // (its rows are usually too uneven for the launches' padded copy -- hubs -- but in registers
            // the stride is only a layout: a private copy with the longest row as stride, below)
            const int S3 = st->run[3].dev.S > 0 ? st->run[3].dev.S : (st->run[3].maxoff + 3) / 4 * 4;
            const int N4 = h->L[4].A.nr;
            if (S3 > 0 && S3 <= 512 && Nt <= BT && N4 <= BT && N2 <= RES_NMAX / 2 &&
                std::max(d1.S, d2.S) <= 512 && Nt + std::max(cdiv(std::max(nf, nc), RES_WAVES), cdiv(N2, RES_WAVES)) <= 2 * BT) {
                three = remote = true;
                ke3 = S3 <= 256 ? 4 : 8;
                if (st->k_sub == 3) {
                    tail_img = st->d_sub4;
                    tail_lds = st->sub4_lds;
                    tail_bm = st->sub4_bm;
                }
            }
        }
        // Four levels with a level 3 too big for any LDS image and a coarsest level of at most 64 rows
        // (dense masks early in a run, the bench's tree / hub masks): level 3 resident, level 4 solved
        // by every workgroup as the local tail -- no tail workgroup.
        // (a V cycle visits the tail once: there the local tail stays ahead of a tail workgroup rooted at a
        // level 3 in block-wide polynomial form -- tree mask 0.096 against 0.102 ms; a W cycle is the other
        // way round, 0.198 against 0.191)
        if (remote && !three && h->J == 4 && h->opts.cycle == 'v' && !(n3 && n3[0] == '1') &&
            h->L[4].A.nr <= RES_TAIL_MAX) {
            const int S3 = st->run[3].dev.S > 0 ? st->run[3].dev.S : (st->run[3].maxoff + 3) / 4 * 4;
            if (S3 > 0 && S3 <= 512 && Nt <= BT && N2 <= RES_NMAX / 2 && std::max(d1.S, d2.S) <= 512 &&
                Nt + std::max(cdiv(std::max(nf, nc), RES_WAVES), cdiv(N2, RES_WAVES)) <= 2 * BT)
                remote = false;
        }
        if (!remote && !(nr && nr[0] == '1') && !(n3 && n3[0] == '1') && h->J == 4 && cyc &&
            h->L[4].A.nr <= RES_TAIL_MAX) {
            const int S3 = st->run[3].dev.S > 0 ? st->run[3].dev.S : (st->run[3].maxoff + 3) / 4 * 4;
            if (S3 > 0 && S3 <= 512 && Nt <= BT && N2 <= RES_NMAX / 2 && std::max(d1.S, d2.S) <= 512 &&
                Nt + std::max(cdiv(std::max(nf, nc), RES_WAVES), cdiv(N2, RES_WAVES)) <= 2 * BT) {
                three = true;
                ke3 = S3 <= 256 ? 4 : 8;
            }
        }
        if (!remote && !three) return;
    }
    const int smax = std::max(d1.S, d2.S);
    int ke = 4;
    while (64 * ke < smax) ke <<= 1;
    if (ke > 16) return;
    if (three && ke > 8) return;   // (the third row slice does not fit beside two 16-entry ones)
    int G = std::max(cdiv(std::max(nf, nc), RES_WAVES), cdiv(N2, RES_WAVES));
    if (const char* e = std::getenv("IPD_RESIDENT_G")) G = std::max(G, std::atoi(e));
    // every workgroup owns at least one row of every block (the hand-off protocol needs it)
    if (G + (remote ? 1 : 0) > st->num_cu || G > std::min(std::min(nf, nc), N2)) return;
    const int Nin = three ? h->L[4].A.nr : Nt;   // rows of the remote tail's root level / of the local tail
    if (remote && Nin > RES_WAVES * G) return;   // one row of the restriction to it per wave
    if (three && Nt + G > 2 * BT) return;        // level-3 hand-offs: N3 + G granules, two per thread
    // level 3 in polynomial form (ResDesc::p3rows): remote tail, one restriction row per workgroup at most,
    // at most four rows of level 3 per workgroup
    const bool poly3 = three && remote && h->opts.smoth >= 1 && h->L[4].A.nr <= G && h->L[4].A.nr <= 128 && Nt <= 4 * G && Nt <= BT &&
                       !(std::getenv("IPD_NO_POLY") && std::getenv("IPD_NO_POLY")[0] == '1');
    if (poly3) ke3 = 1;
    // level 4 resident as well (ResDesc::p4rows), the tail workgroup rooted at level 5
    const int N5r = h->J >= 6 ? h->L[5].A.nr : 0;
    const bool poly4 = poly3 && st->d_sub5 && h->L[1].A.nr <= RES_NMAX && N5r >= 1 && N5r <= 64 && N5r <= G &&
                       h->L[4].A.nr <= RES_P4_SEG && h->L[4].A.nr + G <= BT;
    if (poly4) {
        tail_img = st->d_sub5;
        tail_lds = st->sub5_lds;
        tail_bm = 0;   // (entered at level 5: the copied level is one the resident workgroups hold)
    }
    if (!remote || tail_lds + tail_bm > (size_t)156 * 1024) tail_bm = 0;
    const size_t lds = remote ? std::max<size_t>(RES_LDS_BYTES, tail_lds + tail_bm) : RES_LDS_BYTES;
    if (lds > 156 * 1024) return;
    Arena& ar = *h->arena;
    ResDesc D{};
    auto lev = [](const LevelDev& d) {
        ResLevelDesc L;
        L.N = d.N;
        L.nf = d.nf;
        L.S = d.S;
        L.pci = d.pci;
        L.pva = d.pva;
        L.diag = d.diag;
        L.dinv = d.dinv;
        L.Axi = d.Axi;
        L.xx = d.xx;
        return L;
    };
    auto csr = [](const Csr& m) {
        ResCsr c;
        c.rp = m.rp;
        c.ci = m.ci;
        c.va = m.va;
        return c;
    };
    auto private_pad = [&](const Csr& A, LevelDev& d) {   // k_pad_build with stride d.S
        unsigned short* pci = ar.alloc<unsigned short>((size_t)A.nr * d.S);
        double* pva = ar.alloc<double>((size_t)A.nr * d.S);
        double* dg = ar.alloc<double>((size_t)A.nr);
        hipLaunchKernelGGL(k_pad_build, dim3(std::max(1, std::min(cdiv(A.nr, 4), 4096))), dim3(256), 0,
                           h->ctx->stream, A.nr, d.S, A.rp, A.ci, A.va, pci, pva, dg);
        IPD_KERNEL_CHECK();
        d.pci = pci;
        d.pva = pva;
        d.diag = dg;
    };
    if (priv1) private_pad(l1.A, d1);
    if (priv2) private_pad(l2.A, d2);
    D.L1 = lev(d1);
    D.L2 = lev(d2);
    D.Pt2 = csr(l2.Pt);
    D.P2 = csr(l2.P);
    D.Pt3 = csr(l3.Pt);
    D.P3 = csr(l3.P);
    D.A3 = csr(l3.A);
    D.Nt = Nin;
    D.three = three ? 1 : 0;
    D.tail_root = poly4 ? 5 : three ? 4 : 3;
    D.A4 = csr(three ? h->L[4].A : l3.A);
    if (three) {
        LevelDev d3 = st->run[3].dev;
        if (poly3) {
            const BPolyDev pb = pack_bpoly(h->ctx, h, st, 3, h->opts.isnsp, 0, true);
            D.p3rows = pb.M;
            D.p3w = pb.W;
            st->level_forms.resize((size_t)h->J + 1, 0);
            st->level_forms[3] |= 64;
            if (poly4) {
                const BPolyDev pb4 = pack_bpoly(h->ctx, h, st, 4, h->opts.isnsp, 0, true, RES_P4_SEG, RES_P4_LD);
                D.p4rows = pb4.M;
                D.p4w = pb4.W;
                D.N5 = N5r;
                st->level_forms[4] |= 64;
            }
        } else if (d3.S <= 0) {   // private padded copy of level 3, stride = its longest row
            d3.S = (st->run[3].maxoff + 3) / 4 * 4;
            private_pad(l3.A, d3);
        }
        D.L3 = lev(d3);
        D.Pt4 = csr(h->L[4].Pt);
        D.P4 = csr(h->L[4].P);
    } else {
        D.L3 = lev(d2);   // unused
        D.Pt4 = csr(l3.Pt);
        D.P4 = csr(l3.P);
    }
    D.nu = h->opts.smoth;
    D.isnsp = h->opts.isnsp;
    D.wcycle = h->opts.cycle == 'w';
    D.anycycle = (h->opts.cycle == 'w' || h->opts.cycle == 'v');
    D.maxit = h->opts.maxit;
    D.retol = h->opts.retol;
    D.pcg_maxit = h->opts.pcg_maxit;
    // bigraph transfers P = [W; I]: the kernel adds the identity entries instead of walking them
    // (a bigraph level 1 built by amg_transfer has them by construction -- k_bigph_fill writes the rows of I --
    // which saves the check and its round trip on every hierarchy of a run)
    D.wident = 0;
    if (N2 == nc && h->opts.bigph) {
        D.wident = 1;
    } else if (N2 == nc) {
        int* bad = zeroed<int>(h->ctx, 1);
        hipLaunchKernelGGL(k_res_check_ident, dim3(cdiv(N2, 256)), dim3(256), 0, h->ctx->stream, nf, N2,
                           csr(l2.P), csr(l2.Pt), bad);
        IPD_KERNEL_CHECK();
        D.wident = h->ctx->fetch1(bad) == 0 ? 1 : 0;
    }
    D.localfirst = 1;
    D.pollsleep = 1;   // (0..2 sleeps between polls made no difference, from 3 on it was worse)
    D.presleep = 13;   // measured: 0 -> 0.0869, 8 -> 0.0796, 12..14 -> 0.0770, 16 -> 0.0784 ms per V cycle (a failing poll delays the publishes it waits for)
    const size_t gbytes = (size_t)RES_GRAN_MAX * 16;
    st->res_block_bytes = 2 * gbytes + 16 + (remote ? 4 * gbytes + 16 : 0);
    st->res_block = reinterpret_cast<unsigned char*>(ar.alloc_bytes(st->res_block_bytes));
    D.gran0 = st->res_block;
    D.gran1 = st->res_block + gbytes;
    D.tmo = reinterpret_cast<unsigned*>(st->res_block + 2 * gbytes);
    D.remote = remote ? 1 : 0;
    D.sub = remote ? tail_img : nullptr;
    D.tail_bm = (remote && tail_bm > 0) ? 1 : 0;
    D.tin = remote ? st->res_block + 2 * gbytes + 16 : st->res_block;        // never touched without
    D.tout = remote ? st->res_block + 4 * gbytes + 16 : st->res_block;       // a remote tail
    D.tctl = remote ? reinterpret_cast<unsigned*>(st->res_block + 6 * gbytes + 16) : D.tmo;
    D.dbg = nullptr;
    D.dbg_skip_seq = 0;
    if (const char* e = std::getenv("IPD_RES_DEBUG_SKIP_PUBLISH")) D.dbg_skip_seq = (unsigned)std::max(0, std::atoi(e));
    st->res_desc = D;
    st->res_remote = remote;
    st->resb_poly4 = poly4;
    st->res_ke3 = ke3;
    st->res_G = G;
    st->res_ke = ke;
    st->res_lds = lds;
    st->res_out = ar.alloc<double>(4 + 2 * ((size_t)std::max(h->opts.maxit, 0) + 2));
    st->res_ok = true;
}

// Resident kernels need all their workgroups on the chip at once (one per CU): as many of them may
// run side by side as their grids fit into the device's CUs -- two of 128 workgroups on an MI355X
// (AMG4POT's two concurrent solves, bench.py --batch 2) -- and a further one waits for a free slot
// (a solve lasts a millisecond or two; running it as launches beside two resident kernels slows all
// three: --batch 4 fell from 2 x 20 M to 17.6 M DoF*cycles/s) and takes the multi-launch path only
// if none frees up within 50 ms.  (The spins are bounded, so an over-commitment could only cost the
// launch, never hang.)
struct ResidentSlots {
    std::mutex mu;
    std::condition_variable cv;
    int used[64] = {0};
    bool acquire(int device, int workgroups, int cus, int wait_ms) {
        std::unique_lock<std::mutex> lock(mu);
        if (workgroups > cus) return false;
        const bool got = cv.wait_for(lock, std::chrono::milliseconds(wait_ms),
                                     [&] { return used[device & 63] + workgroups <= cus; });
        if (!got) return false;
        used[device & 63] += workgroups;
        return true;
    }
    void release(int device, int workgroups) {
        {
            std::lock_guard<std::mutex> lock(mu);
            used[device & 63] -= workgroups;
        }
        cv.notify_all();
    }
};
static ResidentSlots& resident_slots() {
    static ResidentSlots s;
    return s;
}
struct ResidentLease {
    int device, wgs;
    bool ok;
    ResidentLease(int d, int w, int cus, int wait_ms)
        : device(d), wgs(w), ok(resident_slots().acquire(d, w, cus, wait_ms)) {}
    ~ResidentLease() {
        if (ok) resident_slots().release(device, wgs);
    }
};

// Runs the whole solve (fixed_cycles == 0) or exactly fixed_cycles loop bodies on the
// iterate in x (in: guess, out: result).  Returns false when the kernel could not be used
// (another resident kernel is running, or a spin gave up): x is then unspecified and the
// caller takes the multi-launch path.  `ms`: device time of the launch (HIP events), optional.
static bool run_resident(ipd_amg* h, CycleState* st, const double* b_dev, double* x, int fixed_cycles,
                         std::vector<double>* out_host, float* ms, long long* dbg_dev = nullptr) {
    ipd_ctx* ctx = h->ctx;
    const int grid = st->res_G + (st->res_remote ? 1 : 0);
    if (ctx->res_penalty > 0) {   // an earlier launch of this context gave up: stay on the launches for a while
        --ctx->res_penalty;
        return false;
    }
    // A remote-tail launch is 129 workgroups at M = 2048: two of them do not fit side by side, and a
    // realistic solve is a few milliseconds of mostly serial sub-cycle work -- waiting for the other
    // solve (AMG4POT's two right-hand sides) would serialise them, so the loser runs as launches
    // beside it at once.  The dense three-level launches (128 workgroups, two fit) keep waiting.
    ResidentLease lease(ctx->device, grid, st->num_cu, st->res_remote ? 0 : 50);
    if (!lease.ok) return false;
    ResDesc D = st->res_desc;
    D.dbg = dbg_dev;
    if (const char* dl = std::getenv("IPD_DEBUG_LEVELS"); dl && dl[0] == '1')
        std::fprintf(stderr, "[ipd] resident launch: grid %d ke %d ke3 %d xm %d wident %d three %d remote %d\n", grid,
                     st->res_ke, st->res_ke3, D.xm, D.wident, D.three, D.remote);
    st->res_desc.dbg_skip_seq = 0;   // the test hook fires on ONE launch
    IPD_HIP(hipMemsetAsync(st->res_block, 0, st->res_block_bytes, ctx->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms) {
        for (hipEvent_t& ev : ctx->tev)
            if (!ev) IPD_HIP(hipEventCreate(&ev));
        e0 = ctx->tev[0];
        e1 = ctx->tev[1];
        IPD_HIP(hipEventRecord(e0, ctx->stream));
    }
    // The workgroups spin on one another, so ALL of them must be on the chip at once: the grid is
    // checked against what the device can hold of this instantiation (registers, LDS: one workgroup
    // per CU) before the first launch; an oversized grid takes the multi-launch path for good.
    bool fits = true;
    if (st->resb) {
        ResBigDesc B = st->resb_desc;
        B.dbg_skip_seq = D.dbg_skip_seq;
#define IPD_RESB_LAUNCH(KE2, RPW, DEEP)                                                             \
    do {                                                                                            \
        IPD_OPTIN_LDS(ctx, (k_resident_big<KE2, RPW, DEEP>), 156 * 1024);                           \
        if (st->res_capacity < 0) {                                                                 \
            int nb_ = 0;                                                                            \
            IPD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (k_resident_big<KE2, RPW, DEEP>), BT, st->res_lds)); \
            st->res_capacity = nb_ * st->num_cu;                                                    \
        }                                                                                           \
        if (grid > st->res_capacity) {                                                              \
            fits = false;                                                                           \
            break;                                                                                  \
        }                                                                                           \
        hipLaunchKernelGGL((k_resident_big<KE2, RPW, DEEP>), dim3(grid), dim3(BT), st->res_lds, ctx->stream, B, \
                           b_dev, x, st->res_out, fixed_cycles);                                    \
    } while (0)
#ifdef IPD_DEV_ONLY_RES16
        (void)B;
        fits = false;
#else
        if (st->resb_deep) {
            if (st->resb_ke2 == 4)
                IPD_RESB_LAUNCH(4, 2, true);
            else
                IPD_RESB_LAUNCH(8, 2, true);
        } else if (st->resb_ke2 == 16)
            IPD_RESB_LAUNCH(16, 1, false);
        else
            IPD_RESB_LAUNCH(32, 1, false);
#endif
#undef IPD_RESB_LAUNCH
    } else
#define IPD_RES_LAUNCH4(KE, KE3, P2)                                                                   \
    do {                                                                                            \
        IPD_OPTIN_LDS(ctx, (k_resident<KE, KE, KE3, P2>), 156 * 1024);                                  \
        if (st->res_capacity < 0) {                                                                 \
            int nb_ = 0;                                                                            \
            IPD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (k_resident<KE, KE, KE3, P2>), BT, st->res_lds)); \
            st->res_capacity = nb_ * st->num_cu;                                                    \
        }                                                                                           \
        if (grid > st->res_capacity) {                                                              \
            fits = false;                                                                           \
            break;                                                                                  \
        }                                                                                           \
        hipLaunchKernelGGL((k_resident<KE, KE, KE3, P2>), dim3(grid), dim3(BT), st->res_lds, ctx->stream, \
                           D, b_dev, x, st->res_out, fixed_cycles);                                 \
    } while (0)
#define IPD_RES_LAUNCH(KE, KE3) IPD_RES_LAUNCH4(KE, KE3, false)
#if defined(IPD_DEV_ONLY_BIG)   // (development: compile the mask-form kernels alone, see tools/kernel_regs.py)
    (void)D;
    fits = false;
#elif defined(IPD_DEV_ONLY_RES16)   // (... or the metric's instantiations alone)
    if (st->res_poly2)
        IPD_RES_LAUNCH4(16, 0, true);
    else
        IPD_RES_LAUNCH(16, 0);
#else
    if (st->res_poly2) {
        IPD_RES_LAUNCH4(16, 0, true);
    } else if (st->res_ke3 == 0) {
        if (st->res_ke == 4)
            IPD_RES_LAUNCH(4, 0);
        else if (st->res_ke == 8)
            IPD_RES_LAUNCH(8, 0);
        else
            IPD_RES_LAUNCH(16, 0);
    } else if (st->res_ke3 == 1) {
        if (st->res_ke == 4)
            IPD_RES_LAUNCH(4, 1);
        else
            IPD_RES_LAUNCH(8, 1);
    } else if (st->res_ke == 4) {
        if (st->res_ke3 == 4)
            IPD_RES_LAUNCH(4, 4);
        else
            IPD_RES_LAUNCH(4, 8);
    } else {
        if (st->res_ke3 == 4)
            IPD_RES_LAUNCH(8, 4);
        else
            IPD_RES_LAUNCH(8, 8);
    }
#endif
#undef IPD_RES_LAUNCH
#undef IPD_RES_LAUNCH4
    if (!fits) {
        st->res_ok = false;
        return false;
    }
    IPD_KERNEL_CHECK();
    if (ms) IPD_HIP(hipEventRecord(e1, ctx->stream));
    const size_t nout = 4 + 2 * ((size_t)std::max(h->opts.maxit, 0) + 2);
    std::vector<double> out(nout);
    ctx->fetch(st->res_out, out.data(), nout);   // waits for the kernel (through the host mailbox: no stream synchronisation)
    if (ms) {
        IPD_HIP(hipEventSynchronize(e1));
        IPD_HIP(hipEventElapsedTime(ms, e0, e1));
    }
    if (out[3] != 0.0) {   // a bounded spin gave up somewhere (any workgroup: the kernel reports the
        // time-out word, not only workgroup 0's own view): not every workgroup was resident
        ++st->res_timeouts;
        ++ctx->res_giveups;
        ctx->res_penalty = 32 << std::min(ctx->res_giveups - 1, 6);
        if (st->res_timeouts >= 2) st->res_ok = false;
        return false;
    }
    st->res_last_handoffs = (long long)out[nout - 1];
    st->res_last_cycles = fixed_cycles > 0 ? fixed_cycles : (int)out[0];
    if (out_host) *out_host = std::move(out);
    return true;
}

void amg_prepare_levels(ipd_amg* h) {
    ipd_ctx* ctx = h->ctx;
    Arena& ar = *h->arena;
    std::unique_ptr<CycleState> st(new CycleState());
    st->run.resize((size_t)h->J + 1);
    const int cu = ctx->num_cu;
    // first pass: per-level vectors and the longest off-diagonal row of every level (one
    // readback for all levels), second pass: padded copies and launch geometry
    int* maxoff = zeroed<int>(ctx, (size_t)h->J + 1);
    // levels whose constant data come from the donor hierarchy (see ipd_amg::donor)
    const ipd_amg* donor = h->donor.get();
    const CycleState* dst_ = donor ? donor->cyc.get() : nullptr;
    auto shared_level = [&](int k) { return dst_ && k <= 2 && k <= donor->J; };
    PrepLevels prep;
    prep.n = 0;
    prep.first_block[0] = 0;
    for (int k = 1; k <= h->J; ++k) {
        Level& lv = h->L[k];
        const int N = lv.A.nr;
        lv.N = N;
        lv.nf = (k == 1 && h->opts.bigph) ? (int)h->opts.fnode : 0;
        IPD_REQUIRE(lv.nf < N, IPD_E_ARG, "fnode must be smaller than the matrix size");
        if (shared_level(k)) {
            const Level& dl = donor->L[k];
            lv.dinv = dl.dinv;
            lv.Axi = dl.Axi;
            lv.xx = dl.xx;
            lv.r = ar.alloc<double>((size_t)N);
            lv.e = ar.alloc<double>((size_t)N);
            lv.e2 = ar.alloc<double>((size_t)N);
            lv.w = ar.alloc<double>((size_t)N);
            lv.rr = ar.alloc<double>((size_t)N);
            continue;
        }
        lv.dinv = ar.alloc<double>((size_t)N);
        lv.Axi = ar.alloc<double>((size_t)N);
        lv.xx = ar.alloc<double>(1);
        lv.r = ar.alloc<double>((size_t)N);
        lv.e = ar.alloc<double>((size_t)N);
        lv.e2 = ar.alloc<double>((size_t)N);
        lv.w = ar.alloc<double>((size_t)N);
        lv.rr = ar.alloc<double>((size_t)N);
        if (prep.n < PREP_ML) {
            const int q = prep.n++;
            prep.N[q] = N;
            prep.nf[q] = lv.nf;
            prep.rp[q] = lv.A.rp;
            prep.ci[q] = lv.A.ci;
            prep.va[q] = lv.A.va;
            prep.dinv[q] = lv.dinv;
            prep.Axi[q] = lv.Axi;
            prep.xx[q] = lv.xx;
            prep.maxoff[q] = maxoff + k;
            prep.first_block[q + 1] = prep.first_block[q] + std::max(1, std::min(cdiv(N, 16), 4096));
        } else {
            hipLaunchKernelGGL(k_level_prepare, dim3(std::max(1, std::min(cdiv(N, 4), 4096))), dim3(256),
                               0, ctx->stream, N, lv.nf, lv.A.rp, lv.A.ci, lv.A.va, lv.dinv, lv.Axi,
                               maxoff + k);
            IPD_KERNEL_CHECK();
            hipLaunchKernelGGL(k_vec_sum, dim3(1), dim3(BT), 0, ctx->stream, (const double*)lv.Axi, N,
                               lv.xx);
            IPD_KERNEL_CHECK();
        }
    }
    if (prep.n > 0) {
        hipLaunchKernelGGL(k_levels_prepare, dim3(prep.first_block[prep.n]), dim3(256), 0, ctx->stream, prep);
        IPD_KERNEL_CHECK();
        hipLaunchKernelGGL(k_levels_sum, dim3(prep.n), dim3(BT), 0, ctx->stream, prep);
        IPD_KERNEL_CHECK();
    }
    std::vector<int> hmax((size_t)h->J + 1);
    ctx->fetch(maxoff, hmax.data(), (size_t)h->J + 1);
    PadBatch pads;   // the levels' padded copies: one launch after the loop
    for (int k = 1; k <= h->J; ++k) {
        Level& lv = h->L[k];
        const int N = lv.N;
        // launch geometry: for a GS level the work per launch is half the matrix
        const int rows_per_launch = lv.nf > 0 ? std::max(1, N / 2) : N;
        const long long nnz_per_launch = lv.nf > 0 ? std::max(1, lv.A.nnz / 2) : lv.A.nnz;
        lv.lanes = pick_lanes(nnz_per_launch, rows_per_launch, cu);
        LevelRun& rn = st->run[(size_t)k];
        rn.dev.N = N;
        rn.dev.nf = lv.nf;
        rn.dev.L = lv.lanes;
        rn.dev.G = pick_blocks(rows_per_launch, lv.lanes, cu);
        rn.dev.rp = lv.A.rp;
        rn.dev.ci = lv.A.ci;
        rn.dev.va = lv.A.va;
        rn.dev.dinv = lv.dinv;
        rn.dev.Axi = lv.Axi;
        rn.dev.xx = lv.xx;
        rn.dev.r = lv.r;
        rn.dev.rr = lv.rr;
        rn.staged = N <= STAGE_MAX ? 1 : 0;
        {
            const char* ns = std::getenv("IPD_NO_STAGE");
            if (ns && ns[0] == '1') rn.staged = 0;
        }
        if (shared_level(k)) {   // the donor's padded copy and the geometry that goes with it
            const LevelDev& dd = dst_->run[(size_t)k].dev;
            rn.dev.S = dd.S;
            rn.dev.pci = dd.pci;
            rn.dev.pva = dd.pva;
            rn.dev.diag = dd.diag;
            rn.dev.L = dd.L;
            rn.dev.G = dd.G;
            lv.lanes = donor->L[k].lanes;
        } else {
            build_padded(ctx, ar, lv.A, rows_per_launch, cu, hmax[(size_t)k], &rn.dev, &pads);
        rn.maxoff = hmax[(size_t)k];
        }
    }
    pad_flush(ctx, &pads);
    for (int k = 1; k < h->J; ++k) {
        Level& fine = h->L[k];
        Level& coarse = h->L[k + 1];
        LevelRun& rn = st->run[(size_t)k];
        XferArgs ra;  // restriction: rows of P' (coarse rows), gathers the fine residual
        ra.nrows = coarse.Pt.nr;
        ra.ncols = coarse.Pt.nc;
        ra.L = pick_lanes(coarse.Pt.nnz, coarse.Pt.nr, cu);
        ra.G = pick_blocks(ra.nrows, ra.L, cu);
        ra.rp = coarse.Pt.rp;
        ra.ci = coarse.Pt.ci;
        ra.va = coarse.Pt.va;
        ra.x = fine.rr;
        ra.y = coarse.r;
        ra.add = 0;
        ra.row0 = 0;
        ra.row1 = ra.nrows;
        ra.staged = ra.ncols <= STAGE_MAX ? 1 : 0;
        rn.restrict_args = ra;
        XferArgs pa;  // prolongation: rows of P (fine rows), gathers the coarse correction
        pa.nrows = coarse.P.nr;
        pa.ncols = coarse.P.nc;
        pa.L = pick_lanes(coarse.P.nnz, coarse.P.nr, cu);
        pa.G = pick_blocks(pa.nrows, pa.L, cu);
        pa.rp = coarse.P.rp;
        pa.ci = coarse.P.ci;
        pa.va = coarse.P.va;
        pa.x = coarse.e;
        pa.y = fine.e;
        pa.add = 1;
        pa.row0 = 0;
        pa.row1 = pa.nrows;
        pa.staged = pa.ncols <= STAGE_MAX ? 1 : 0;
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
        a.maxit = h->opts.pcg_maxit;
        a.precd = 2;
        a.out = nullptr;
        a.nresk = 0;
        st->run[(size_t)h->J].pcg = a;
    }
    st->num_cu = cu;
    st->fuse_enabled = true;
    st->hist = ar.alloc<double>(8);
    st->x2 = ar.alloc<double>((size_t)h->L[1].A.nr);
    h->x = ar.alloc<double>((size_t)h->L[1].A.nr);
    h->b = ar.alloc<double>((size_t)h->L[1].A.nr);
    // ---- single-workgroup kernels -------------------------------------------------------
    // (a) whole solve in one workgroup when every level is small; (b) otherwise the sub-cycle
    // below the first level from which everything fits in LDS runs as one launch per visit.
    auto small_level = [&](int k) {
        const Level& lv = h->L[k];
        // one workgroup is one CU: beyond ~1000 short rows per level the multi-launch
        // path (many CUs per phase) wins again (measured: M = 1000 W-cycle solve 9.5 ms
        // here vs 17 ms multi-launch; M = 2048: 8.0 ms here vs 5.6 ms multi-launch)
        return lv.A.nr <= 1024 && lv.A.nnz <= 40000 && (k < 2 || lv.P.nnz <= 40000);
    };
    auto r16 = [](size_t b) { return (b + 15) / 16 * 16; };
    auto fill_desc = [&](SolveDesc* sd) {
        std::memset(sd, 0, sizeof(SolveDesc));
        sd->J = h->J;
        sd->nu = h->opts.smoth;
        sd->isnsp = h->opts.isnsp;
        sd->wcycle = h->opts.cycle == 'w';
        sd->anycycle = (h->opts.cycle == 'w' || h->opts.cycle == 'v');
        sd->maxit = h->opts.maxit;
        sd->retol = h->opts.retol;
        sd->pcg = st->run[(size_t)h->J].pcg;
        for (int k = 1; k <= h->J; ++k) {
            SolveLevel& sl = sd->L[k];
            sl.lv = st->run[(size_t)k].dev;
            sl.lv.S = 0;  // the single-workgroup kernels walk the CSR arrays only
            // in one workgroup a row is walked by few lanes: re-pick without widening
            const Level& lv = h->L[k];
            {
                const double avg = (double)lv.A.nnz / std::max(lv.A.nr, 1);
                int L = 1;
                while (L < 64 && (double)L * 6.0 < avg) L <<= 1;
                sl.lv.L = L;
            }
            sl.lv.G = 1;
            sl.e = lv.e;
            sl.e2 = lv.e2;
            sl.w = lv.w;
            sl.nnzA = lv.A.nnz;
            sl.nnzP = k < h->J ? h->L[k + 1].P.nnz : 0;
            if (k < h->J) {
                sl.rest = st->run[(size_t)k].restrict_args;
                sl.prol = st->run[(size_t)k].prolong_args;
                for (XferArgs* xa : {&sl.rest, &sl.prol}) {
                    const double avg = (double)(xa == &sl.rest ? h->L[k + 1].Pt.nnz : h->L[k + 1].P.nnz) /
                                       std::max(xa->nrows, 1);
                    int L = 1;
                    while (L < 64 && (double)L * 6.0 < avg) L <<= 1;
                    xa->L = L;
                    xa->G = 1;
                    xa->staged = 1;
                    xa->row0 = 0;
                    xa->row1 = xa->nrows;
                }
            }
        }
    };
    // bottom run of levels with <= 32 rows (k >= 2): candidates for the wave-level sub-cycle
    // (33..64 rows run faster block-wide with 16 lanes per row than in one wave) -- <= 48 rows when
    // the one-wave levels take the polynomial form (tiny_cycle: a visit is two dense passes whatever
    // the row count; IPD_NO_POLY=1: sweeps)
    int tiny_lo = h->J + 1;
    bool use_poly = h->opts.smoth >= 1 && (h->opts.cycle == 'w' || h->opts.cycle == 'v') &&
                    !(std::getenv("IPD_NO_POLY") && std::getenv("IPD_NO_POLY")[0] == '1') &&
                    !(std::getenv("IPD_NO_BLK") && std::getenv("IPD_NO_BLK")[0] == '1');
    // (polynomial form: a level whose stacked operator [e'; r_c] has more than 32 rows -- one lane per row in
    // a single wave -- runs block-wide instead, out of LDS all the same: is_lpoly below)
    bool use_lpoly = !(std::getenv("IPD_NO_BLK") && std::getenv("IPD_NO_BLK")[0] == '1');
    auto find_tiny_lo = [&](int rows_max) {
        int lo = h->J + 1;
        for (int k = h->J; k >= 2; --k) {
            if (h->L[k].A.nr > rows_max) break;
            if (rows_max > 32 && use_lpoly && k < h->J && h->L[k].A.nr + h->L[k + 1].A.nr > 32) break;
            lo = k;
        }
        return lo;
    };
    tiny_lo = find_tiny_lo(use_poly ? 48 : 32);
    auto r8 = [](size_t n) { return (n + 7) / 8 * 8; };
    auto poly_ld = [](size_t rows) -> size_t { return rows <= 32 ? 32 : (rows <= 48 ? 48 : 64); };
    auto is_poly = [&](int k) {
        return use_poly && k >= tiny_lo && k < h->J && h->L[k].A.nr + h->L[k + 1].A.nr <= 64;
    };
    // the thread-per-row / wave sub-cycles keep the residual in the free iterate buffer and never
    // use the Gauss-Seidel scratch vector: 5 vectors per cached level instead of 7
    bool lean_vectors = true;
    {
        const char* nb = std::getenv("IPD_NO_BLK");
        if (nb && nb[0] == '1') lean_vectors = false;
    }
    const bool use_lmap = lean_vectors;
    // small, nearly full thread-per-row levels: dense copy instead of the CSR arrays (see SolveLevel::blk_dense)
    const bool use_bdense = lean_vectors && !(std::getenv("IPD_NO_BLKDENSE") && std::getenv("IPD_NO_BLKDENSE")[0] == '1');
    auto is_lpoly = [&](int k) {
        return use_poly && use_lpoly && lean_vectors && k >= 2 && k < tiny_lo && k < h->J && h->L[k].A.nr <= 48 &&
               h->L[k].A.nr + h->L[k + 1].A.nr <= 64;
    };
    // thread-per-row levels of 33..144 rows in block-wide polynomial form (SolveLevel::gM); use_poly may
    // still be withdrawn below, hence the reference
    const bool use_bpoly = lean_vectors && !(std::getenv("IPD_NO_BPOLY") && std::getenv("IPD_NO_BPOLY")[0] == '1');
    auto bpoly_ld = [&](int k) { return h->L[k].A.nr + h->L[k + 1].A.nr <= 128 ? 128 : 256; };
    auto is_bpoly = [&](int k) {
        if (!use_poly || !use_bpoly || k < 2 || k >= h->J || k >= tiny_lo) return false;
        const long long N = h->L[k].A.nr, Nc = h->L[k + 1].A.nr;
        // (up to 224 rows below a level 1 of more than 2048 rows: there level 4 has 150-200 rows, often dense --
        // 23 k entries do not fit an LDS image, the operators of this form stay in L2 -- and the mask-form
        // resident kernel needs its tail rooted at level 4, ipd_resident_big.h DEEP)
        const long long nmax = h->L[1].A.nr > RES_NMAX ? 224 : 144;
        return N > 32 && N <= nmax && N + Nc <= 256 && 2 * ((N + 7) / 8 * 8) + (Nc + 7) / 8 * 8 <= 512 && !is_lpoly(k);
    };
    std::vector<BPolyDev> bpoly_dev((size_t)h->J + 2);
    auto ensure_bpoly = [&](int k, int nu, int isnsp) -> const BPolyDev& {
        (void)nu;
        BPolyDev& b = bpoly_dev[(size_t)k];
        if (!b.M) {
            b = pack_bpoly(ctx, h, st.get(), k, isnsp, bpoly_ld(k), false);
            st->poly_ops.resize((size_t)h->J + 1);
            CycleState::PolyOp& po = st->poly_ops[(size_t)k];
            po.M = b.M;
            po.W = b.W;
            po.LD = b.LD;
            po.N = h->L[k].A.nr;
            po.Nc = h->L[k + 1].A.nr;
        }
        return b;
    };
    auto is_bdense = [&](int k) {
        if (!use_bdense || k < 2 || k >= h->J || k >= tiny_lo || is_bpoly(k)) return false;
        const long long N = h->L[k].A.nr;
        return N > 32 && N <= 96 && bdense_pad((int)N) / bdense_lanes((int)N) <= BDENSE_Q &&
               3LL * h->L[k].A.nnz >= N * N;
    };
    // LDS cache plan: deepest levels first, while they fit; returns the first cached level
    auto plan_lds = [&](size_t stage, size_t* used_out) {
        size_t used = stage + SOL_HEAD + 256;
        const size_t budget = 150 * 1024;
        int k_lds = h->J + 1;
        for (int k = h->J; k >= 1; --k) {
            const Level& lv = h->L[k];
            const size_t N = (size_t)lv.A.nr;
            size_t bytes;
            if (is_poly(k)) {
                // polynomial form: [M2a; ..] and [M1; ..] stacked with the restriction, M1 P, w; three
                // vectors; none of the level's CSR arrays (its parent applies the transfers to and from it)
                const size_t Nc = (size_t)h->L[k + 1].A.nr, LD = poly_ld(N + Nc);
                bytes = 2 * (8 * LD * r8(N)) + 8 * LD * r8(Nc) + 8 * LD + 3 * r16(8 * r8(N)) + 32;
            } else if (is_lpoly(k)) {
                const size_t Nc = (size_t)h->L[k + 1].A.nr, LD = 64;
                bytes = 2 * (8 * LD * r8(N)) + 8 * LD * r8(Nc) + 8 * LD + 3 * r16(8 * r8(N)) + 32 + 8 * (8 * LD + 8);
            } else if (is_bpoly(k)) {
                // block-wide polynomial form: the operators stay in global memory; three vectors and the
                // partial sums of a pass
                bytes = 3 * r16(8 * r8(N)) + 48 + 8 * (8 * (size_t)bpoly_ld(k) + 8);
            } else {
                // (the thread-per-row sub-cycle deals BT threads to the rows: a level of more than BT rows cannot
                // be held that way -- it fits the budget once its child's operators stay in L2, block-wide
                // polynomial form of a 150-224-row level 4 below a 576-row level 3)
                if (k >= 2 && N > (size_t)BT) break;
                bytes = r16(4 * (N + 1)) +
                        (is_bdense(k) ? r16(8 * N * (size_t)bdense_ld((int)N))
                                      : r16(4 * (size_t)lv.A.nnz) + r16(8 * (size_t)lv.A.nnz)) +
                        ((lean_vectors && k >= 2) ? 5 : 7) *
                            r16(8 * (k >= tiny_lo ? r8(N) : is_bdense(k) ? (size_t)bdense_pad((int)N) : N)) +
                        16;
                if (k < h->J) {
                    const size_t Nc = (size_t)h->L[k + 1].A.nr, np = (size_t)h->L[k + 1].P.nnz;
                    bytes += r16(4 * (Nc + 1)) + r16(4 * (N + 1)) + 2 * (r16(4 * np) + r16(8 * np));
                }
                if (k == h->J) bytes += r16(4 * 8 * N);
                if (use_lmap && !is_bdense(k) && k >= 2 && k < tiny_lo && N <= (size_t)BT) bytes += r16(4 * (BT + 1));   // lane map
                if (k >= 3 && (is_bpoly(k - 1) || is_lpoly(k - 1))) bytes += 5 * 64;   // its vectors are padded to whole 8-entry blocks
                if (k >= tiny_lo) {   // dense copies of the tiny levels
                    bytes += r16(8 * N * N);
                    if (k < h->J) bytes += 2 * r16(8 * N * (size_t)h->L[k + 1].A.nr);
                }
            }
            if (used + bytes > budget) break;
            used += bytes;
            k_lds = k;
        }
        *used_out = used;
        return k_lds;
    };
    if (use_poly) {   // not at the price of a level that would otherwise be cached
        size_t u = 0;
        const int with_poly = plan_lds(16, &u);
        const int lo_poly = tiny_lo;
        use_poly = false;
        tiny_lo = find_tiny_lo(32);
        const int without = plan_lds(16, &u);
        if (with_poly <= without) {
            use_poly = true;
            tiny_lo = lo_poly;
        } else if (use_lpoly) {
            // the block-wide form out of LDS pads its operators to 64 rows: where that is what does not
            // fit, the one-wave form (48 rows) may still
            use_lpoly = false;
            use_poly = true;
            tiny_lo = find_tiny_lo(48);
            if (plan_lds(16, &u) > without) {
                use_poly = false;
                tiny_lo = find_tiny_lo(32);
            }
        }
    }
    auto tiny_from = [&](int k_lds) {   // tiny levels: <= 32 rows, cached, Jacobi (k >= 2)
        return std::max(tiny_lo, std::max(2, k_lds));
    };
    auto blk_from = [&](int k_lds) {    // cached Jacobi levels: thread-per-row sub-cycle
        const char* nb = std::getenv("IPD_NO_BLK");
        if ((nb && nb[0] == '1') || k_lds > h->J) return h->J + 1;
        return std::max(2, k_lds);
    };
    // Lays levels k_from..J out behind the staging area, packs the image on the device and
    // returns it (the descriptor the kernels take); *lds_total = dynamic LDS bytes to request.
    auto build_image = [&](SolveDesc* sd, int k_from, size_t stage, size_t* lds_total, size_t* bm_extra = nullptr) {
        const bool lean = lean_vectors && sd->k_blk <= std::max(2, k_from);
        std::vector<PackEntry> packs;
        std::vector<unsigned> relocs;
        size_t off = stage + SOL_HEAD;   // LDS offset (from dyn_raw) of the next carve
        auto carve = [&](size_t bytes) {
            const size_t o = off;
            off += r16(bytes);
            return o;
        };
        auto set_off = [&](auto& field, size_t o) {
            using T = std::remove_reference_t<decltype(field)>;
            field = reinterpret_cast<T>(o);
            relocs.push_back((unsigned)(reinterpret_cast<char*>(&field) - reinterpret_cast<char*>(sd)));
        };
        auto put = [&](auto& field, size_t n) {   // constant array: copied into the image
            using T = std::remove_reference_t<decltype(field)>;
            using E = std::remove_cv_t<std::remove_pointer_t<T>>;
            const size_t o = carve(n * sizeof(E));
            packs.push_back(PackEntry{(const void*)field, (unsigned)(o - stage), (unsigned)(n * sizeof(E))});
            set_off(field, o);
        };
        int bp_ld_max = 0;
        for (int k = k_from; k <= h->J; ++k) {     // constants first: they form the image
            SolveLevel& T = sd->L[k];
            const size_t N = (size_t)T.lv.N;
            if (k == sd->k_semi) continue;         // matrix, transfers, dinv, Axi stay in global memory
            if (((is_poly(k) && k >= sd->k_tiny) || (is_lpoly(k) && k < sd->k_tiny)) && sd->k_blk <= k) {   // polynomial form: no CSR arrays (see plan_lds)
                put(T.lv.xx, 1);
                T.lv.rp = T.lv.ci = nullptr;
                T.lv.va = T.lv.dinv = T.lv.Axi = nullptr;
                T.rest.rp = T.rest.ci = T.prol.rp = T.prol.ci = nullptr;
                T.rest.va = T.prol.va = nullptr;
                continue;
            }
            if (is_bpoly(k) && sd->k_blk <= k && k < sd->k_tiny) {   // block-wide polynomial form (see plan_lds)
                put(T.lv.xx, 1);
                T.lv.rp = T.lv.ci = nullptr;
                T.lv.va = T.lv.dinv = T.lv.Axi = nullptr;
                T.rest.rp = T.rest.ci = T.prol.rp = T.prol.ci = nullptr;
                T.rest.va = T.prol.va = nullptr;
                const BPolyDev& b = ensure_bpoly(k, sd->nu, sd->isnsp);
                T.gM = b.M;
                T.gW = b.W;
                T.gLD = b.LD;
                bp_ld_max = std::max(bp_ld_max, b.LD);
                continue;
            }
            put(T.lv.rp, N + 1);
            if (is_bdense(k) && sd->k_blk <= k && k < sd->k_tiny) {   // dense copy (carved below) instead of ci / va
                T.blk_dense = 1;
                T.lv.ci = nullptr;
                T.lv.va = nullptr;
            } else {
                put(T.lv.ci, (size_t)T.nnzA);
                put(T.lv.va, (size_t)T.nnzA);
            }
            put(T.lv.dinv, N);
            put(T.lv.Axi, N);
            put(T.lv.xx, 1);
            if (k < h->J) {
                const size_t Nc = (size_t)T.rest.nrows;
                put(T.rest.rp, Nc + 1);
                put(T.rest.ci, (size_t)T.nnzP);
                put(T.rest.va, (size_t)T.nnzP);
                put(T.prol.rp, N + 1);
                put(T.prol.ci, (size_t)T.nnzP);
                put(T.prol.va, (size_t)T.nnzP);
            }
        }
        std::vector<LmapEntry> lmaps;
        for (int k = std::max(k_from, sd->k_blk); k < std::min(sd->k_tiny, h->J + 1); ++k) {
            if (!use_lmap || k == sd->k_semi || k < 2 || h->L[k].A.nr > BT || k == h->J || sd->L[k].blk_dense || sd->L[k].gM || is_lpoly(k)) continue;
            SolveLevel& T = sd->L[k];
            const size_t o = carve(4 * (BT + 1));
            lmaps.push_back(LmapEntry{h->L[k].A.rp, h->L[k].A.nr, (unsigned)(o - stage)});
            set_off(T.lmap, o);
        }
        std::vector<DenseEntry> dense;
        std::vector<PolyEntry> polys;
        size_t poly_lds = 0;
        for (int k = k_from; k <= h->J; ++k) {
            if (!sd->L[k].blk_dense) continue;
            const Csr& m = h->L[k].A;
            const int ld = bdense_ld(m.nr);
            const size_t o = carve(8 * (size_t)m.nr * ld);
            dense.push_back(DenseEntry{m.rp, m.ci, m.va, m.nr, m.nc, (unsigned)(o - stage), ld});
            set_off(sd->L[k].dA, o);
        }
        auto add_poly = [&](int k, size_t LD) {
            SolveLevel& T = sd->L[k];
            const Level& lv = h->L[k];
            const size_t N = (size_t)lv.A.nr;
            const Csr& P = h->L[k + 1].P;
            const size_t Nc = (size_t)P.nc;
            const LevelDev& gd = st->run[(size_t)k].dev;   // global pointers (T's are LDS offsets by now)
            PolyEntry pe;
            pe.Arp = lv.A.rp;
            pe.Aci = lv.A.ci;
            pe.Ava = lv.A.va;
            pe.Prp = P.rp;
            pe.Pci = P.ci;
            pe.Pva = P.va;
            pe.dinv = gd.dinv;
            pe.Axi = gd.Axi;
            pe.xx = gd.xx;
            pe.N = (int)N;
            pe.Nc = (int)Nc;
            pe.nu = sd->nu;
            pe.isnsp = sd->isnsp;
            pe.LD = (int)LD;
            T.pLD = (int)LD;
            size_t o = carve(8 * LD * r8(N));
            pe.offMr = (unsigned)(o - stage);
            set_off(T.pMr, o);
            o = carve(8 * LD * r8(N));
            pe.offMe = (unsigned)(o - stage);
            set_off(T.pMe, o);
            o = carve(8 * LD * r8(Nc));
            pe.offMc = (unsigned)(o - stage);
            set_off(T.pMc, o);
            o = carve(8 * LD);
            pe.offW = (unsigned)(o - stage);
            set_off(T.pW, o);
            polys.push_back(pe);
            poly_lds = std::max(poly_lds, 8 * (5 * N * N + 2 * N * Nc + 4 * N) + 64);
        };
        for (int k = std::max(k_from, sd->k_blk); k < std::min(sd->k_tiny, h->J); ++k)
            if (is_lpoly(k) && k != sd->k_semi) {   // block-wide out of LDS: leading dimension 64, one row per lane
                add_poly(k, 64);
                bp_ld_max = std::max(bp_ld_max, 64);
            }
        for (int k = std::max(k_from, sd->k_tiny); k <= h->J; ++k) {
            SolveLevel& T = sd->L[k];
            const Level& lv = h->L[k];
            const size_t N = (size_t)lv.A.nr;
            if (is_poly(k) && sd->k_blk <= k && k != sd->k_semi) {
                add_poly(k, poly_ld(N + (size_t)h->L[k + 1].P.nc));
                continue;
            }
            auto add = [&](const double*& field, const Csr& m) {
                const size_t o = carve(8 * (size_t)m.nr * m.nc);
                dense.push_back(DenseEntry{m.rp, m.ci, m.va, m.nr, m.nc, (unsigned)(o - stage), 0});
                set_off(field, o);
            };
            add(T.dA, lv.A);
            if (k < h->J) {
                add(T.dP, h->L[k + 1].P);
                add(T.dPt, h->L[k + 1].Pt);
            }
            (void)N;
        }
        const size_t image_bytes = off - stage;
        for (int k = k_from; k <= h->J; ++k) {     // work vectors: carved, not copied
            SolveLevel& T = sd->L[k];
            // (one-wave levels: zero-padded to whole 8-entry blocks, see sol_load_image)
            // (dense thread-per-row levels: zero-padded to whole groups of four entries per lane, dense_row_dot)
            // (block-wide polynomial levels and their children: whole 8-entry blocks as well, bpoly_pass)
            const bool pad8 = k >= sd->k_tiny || T.gM || T.pMr || (k > k_from && (sd->L[k - 1].gM || sd->L[k - 1].pMr));
            const size_t N = T.blk_dense ? (size_t)bdense_pad(T.lv.N) : pad8 ? r8((size_t)T.lv.N) : (size_t)T.lv.N;
            set_off(T.lv.r, carve(N * 8));
            set_off(T.e, carve(N * 8));
            set_off(T.e2, carve(N * 8));
            if (lean && k >= 2) {   // never dereferenced on these levels (see lean_vectors)
                T.lv.rr = T.e2;
                relocs.push_back((unsigned)(reinterpret_cast<char*>(&T.lv.rr) - reinterpret_cast<char*>(sd)));
                T.w = T.e2;
                relocs.push_back((unsigned)(reinterpret_cast<char*>(&T.w) - reinterpret_cast<char*>(sd)));
            } else {
                set_off(T.lv.rr, carve(N * 8));
                set_off(T.w, carve(N * 8));
            }
        }
        if (bp_ld_max) set_off(sd->bp_part, carve(8 * (8 * (size_t)bp_ld_max + 8)));
        for (int k = std::max(1, k_from - 1); k < h->J; ++k) {   // vectors that cross levels
            SolveLevel& T = sd->L[k];
            if (k >= k_from) {
                T.rest.x = T.lv.rr;
                relocs.push_back((unsigned)(reinterpret_cast<char*>(&T.rest.x) - reinterpret_cast<char*>(sd)));
            }
            T.rest.y = sd->L[k + 1].lv.r;
            relocs.push_back((unsigned)(reinterpret_cast<char*>(&T.rest.y) - reinterpret_cast<char*>(sd)));
        }
        {
            sd->pcg.rp = sd->L[h->J].lv.rp;
            sd->pcg.ci = sd->L[h->J].lv.ci;
            sd->pcg.va = sd->L[h->J].lv.va;
            for (auto* f : {(const void**)&sd->pcg.rp, (const void**)&sd->pcg.ci, (const void**)&sd->pcg.va})
                relocs.push_back((unsigned)(reinterpret_cast<char*>(f) - reinterpret_cast<char*>(sd)));
            set_off(sd->pcg.work, carve(4 * (size_t)sd->L[h->J].lv.N * 8));
        }
        IPD_REQUIRE(relocs.size() <= (size_t)RELOC_MAX, IPD_E_LIMIT, "LDS image: too many relocations");
        sd->image_bytes = (int)image_bytes;
        sd->dbg_skip = std::getenv("IPD_DEBUG_SKIP") ? std::atoi(std::getenv("IPD_DEBUG_SKIP")) : 0;
        sd->lds_total = (int)r16(off);
        off = r16(off);
        sd->nreloc = (int)relocs.size();
        *lds_total = off;
        // One block-wide polynomial level's operator as an LDS copy (SolveDesc::bm_src), for the launches that can
        // afford bm_bytes more dynamic LDS (the resident kernels' tail workgroup): the deepest such level whose
        // stacked operator has at most 128 rows and fits behind the work vectors.
        sd->bm_src = nullptr;
        sd->bm_level = sd->bm_ld = sd->bm_off = sd->bm_bytes = 0;
        if (bm_extra) {
            *bm_extra = 0;
            for (int k = h->J - 1; k >= std::max(2, k_from); --k) {
                const SolveLevel& T = sd->L[k];
                if (!T.gM || T.gLD != 128) continue;
                const size_t N = (size_t)T.lv.N, Nc = (size_t)h->L[k + 1].A.nr, rows = N + Nc;
                if (rows > 128) continue;
                const size_t ld = (rows + 1) & ~size_t(1), ncols = 8 * (2 * ((N + 7) / 8) + (Nc + 7) / 8);
                const size_t need = 8 * ld * (ncols + 1);   // (ld even: a multiple of 16; the vector W behind the columns)
                if (off + need > (size_t)156 * 1024) continue;
                double* cp = ar.alloc<double>(ld * (ncols + 1));
                hipLaunchKernelGGL(k_bm_compact, dim3((unsigned)ncols + 1), dim3(128), 0, ctx->stream, T.gM, 128, cp,
                                   (int)ld, T.gW, (int)rows);
                IPD_KERNEL_CHECK();
                sd->bm_src = cp;
                sd->bm_level = k;
                sd->bm_ld = (int)ld;
                sd->bm_off = (int)off;
                sd->bm_bytes = (int)need;
                *bm_extra = need;
                break;
            }
        }
        char* img = reinterpret_cast<char*>(ar.alloc_bytes(image_bytes));
        // the image head and the pack descriptors go up in ONE copy: [head | packs | dense | lmaps | polys] in
        // a scratch block, the head then moves into the image as one more entry of k_pack_image
        auto r16b = [](size_t v) { return (v + 15) & ~size_t(15); };
        const size_t o_packs = r16b(SOL_HEAD), o_dense = o_packs + r16b((packs.size() + 1) * sizeof(PackEntry)),
                     o_lmaps = o_dense + r16b(dense.size() * sizeof(DenseEntry)),
                     o_polys = o_lmaps + r16b(lmaps.size() * sizeof(LmapEntry)),
                     o_end = o_polys + r16b(polys.size() * sizeof(PolyEntry));
        char* stg = reinterpret_cast<char*>(ctx->scratch->alloc_bytes(o_end));
        std::vector<char> hb(o_end, 0);
        std::memcpy(hb.data(), sd, sizeof(SolveDesc));
        std::memcpy(hb.data() + r16(sizeof(SolveDesc)), relocs.data(), relocs.size() * sizeof(unsigned));
        {
            PackEntry he{};
            he.src = stg;
            he.dst_off = 0;
            he.bytes = (unsigned)SOL_HEAD;
            packs.push_back(he);
        }
        std::memcpy(hb.data() + o_packs, packs.data(), packs.size() * sizeof(PackEntry));
        if (!dense.empty()) std::memcpy(hb.data() + o_dense, dense.data(), dense.size() * sizeof(DenseEntry));
        if (!lmaps.empty()) std::memcpy(hb.data() + o_lmaps, lmaps.data(), lmaps.size() * sizeof(LmapEntry));
        if (!polys.empty()) std::memcpy(hb.data() + o_polys, polys.data(), polys.size() * sizeof(PolyEntry));
        ctx->upload_bytes(stg, hb.data(), o_end);
        hipLaunchKernelGGL(k_pack_image, dim3((unsigned)packs.size()), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const PackEntry*>(stg + o_packs), img);
        IPD_KERNEL_CHECK();
        if (!dense.empty()) {
            hipLaunchKernelGGL(k_pack_dense, dim3((unsigned)dense.size()), dim3(256), 0, ctx->stream,
                               reinterpret_cast<const DenseEntry*>(stg + o_dense), img);
            IPD_KERNEL_CHECK();
        }
        if (!lmaps.empty()) {
            hipLaunchKernelGGL(k_pack_lmap, dim3((unsigned)lmaps.size()), dim3(BT), 0, ctx->stream,
                               reinterpret_cast<const LmapEntry*>(stg + o_lmaps), img);
            IPD_KERNEL_CHECK();
        }
        if (!polys.empty()) {
            IPD_OPTIN_LDS(ctx, k_pack_poly, 156 * 1024);
            hipLaunchKernelGGL(k_pack_poly, dim3((unsigned)polys.size()), dim3(BT), poly_lds, ctx->stream,
                               reinterpret_cast<const PolyEntry*>(stg + o_polys), img);
            IPD_KERNEL_CHECK();
        }
        st->level_forms.resize((size_t)h->J + 1, 0);
        for (int k = std::max(k_from, sd->k_blk); k <= h->J; ++k) {
            const SolveLevel& T = sd->L[k];
            if (k == sd->k_semi) continue;
            st->level_forms[(size_t)k] |= T.gM ? 16 : T.pMr ? (k >= sd->k_tiny ? 8 : 32) : k >= sd->k_tiny ? 4 : T.blk_dense ? 2 : 1;
        }
        return reinterpret_cast<SolveDesc*>(img);
    };
    IPD_OPTIN_LDS(ctx, k_solve_small<true>, 156 * 1024);
    IPD_OPTIN_LDS(ctx, k_solve_small<false>, 156 * 1024);
    IPD_OPTIN_LDS(ctx, k_subcycle, 156 * 1024);
    // Level 2 as a semi-cached level (r, e, e2 in LDS; matrix rows from L2) with levels 3..J fully
    // cached: returns the dynamic LDS needed behind a staging area of `stage` bytes, 0 = no
    auto semi_plan = [&](size_t stage) -> size_t {
        if (!lean_vectors || h->J < 3 || h->J > SOLVE_ML) return 0;
        const Level& l2 = h->L[2];
        if (l2.A.nr > BT || l2.A.nr <= 64 || (double)l2.A.nnz > 12.0 * l2.A.nr ||
            (double)h->L[3].P.nnz > 12.0 * l2.A.nr)
            return 0;
        for (int k = 3; k <= h->J; ++k)
            if (!small_level(k)) return 0;
        size_t used = 0;
        const int k_lds = plan_lds(stage, &used);
        if (k_lds != 3) return 0;   // <= 2: level 2 fits entirely; > 3: a deeper level does not
        const size_t need = used + 3 * r16(8 * (size_t)l2.A.nr);
        return need <= 150 * 1024 ? need : 0;
    };
    if (!st->small_ok) {
        const char* ns = std::getenv("IPD_NO_SMALL");
        bool ok = !(ns && ns[0] == '1') && h->J <= SOLVE_ML;
        size_t maxlen = 1;
        for (int k = 1; k <= h->J && ok; ++k) {
            ok = ok && small_level(k);
            maxlen = std::max(maxlen, (size_t)h->L[k].A.nr);
        }
        if (ok) {
            std::unique_ptr<SolveDesc> sd(new SolveDesc());
            fill_desc(sd.get());
            const size_t stage = r16(sizeof(double) * maxlen);
            size_t used = 0;
            int k_lds = plan_lds(stage, &used);
            sd->k_lds = k_lds;
            st->solve_cached = k_lds <= h->J;
            sd->k_tiny = tiny_from(k_lds);
            sd->k_blk = blk_from(k_lds);
            sd->stage_bytes = (int)stage;
            st->solve_lds = used;
            if (st->solve_cached) {
                st->d_solve = build_image(sd.get(), k_lds, stage, &st->solve_lds);
            } else {
                st->d_solve = reinterpret_cast<SolveDesc*>(ar.alloc_bytes(sizeof(SolveDesc)));
                ctx->upload_bytes(st->d_solve, sd.get(), sizeof(SolveDesc));
            }
            st->solve_out = ar.alloc<double>(4 + 2 * ((size_t)std::max(h->opts.maxit, 0) + 2));
            st->small_ok = true;
        }
    }
    {
        const char* ns = std::getenv("IPD_NO_SUBCYCLE");
        const bool want = !(ns && ns[0] == '1') && !st->small_ok && h->J <= SOLVE_ML && h->J >= 3 &&
                          (h->opts.cycle == 'w' || h->opts.cycle == 'v');
        bool semi_done = false;
        if (want && semi_plan(16) != 0) {   // (b1) the sub-cycle is rooted at the semi-cached level 2
            const size_t stage = 16;
            std::unique_ptr<SolveDesc> sd(new SolveDesc());
            fill_desc(sd.get());
            sd->k_lds = 2;
            sd->k_semi = 2;
            sd->k_tiny = tiny_from(3);
            sd->k_blk = blk_from(2);
            sd->stage_bytes = (int)stage;
            sd->root_r = h->L[2].r;
            sd->root_e = h->L[2].e;
            st->k_sub = 2;
            st->d_sub = build_image(sd.get(), 2, stage, &st->sub_lds, &st->sub_bm);
            semi_done = true;
        }
        if (want && !semi_done) {
            // first level from which every level is small ...
            int k_small = h->J + 1;
            for (int k = h->J; k >= 2 && small_level(k); --k) k_small = k;
            // Level 1 of 2049..4096 rows, six levels or more: the mask-form resident kernel's deep mode keeps
            // levels 3 AND 4 in polynomial form in its workgroups and roots its tail workgroup at level 5
            // (ipd_resident_big.h, POLY4) -- ONE image, rooted at level 5, serves it and the launches (which
            // then run level 4 as launches: the fall-back).  (An image rooted at level 4 for the launches
            // beside one rooted at level 5 for the resident kernel packed levels 5..J twice: 0.2 ms per hierarchy.)
            const int nf1 = h->L[1].nf, nc1 = h->L[1].A.nr - nf1;
            const bool root5 = h->J >= 6 && h->L[1].A.nr > RES_NMAX && nf1 > 0 && nf1 <= RB_HALF && nc1 <= RB_HALF &&
                               h->L[2].A.nr == nc1 && h->L[3].A.nr <= RB_N3MAX && h->L[4].A.nr <= RB_N4MAX &&
                               h->L[5].A.nr <= RB_N5MAX && h->opts.smoth >= 1 && !h->opts.twogrid && !h->opts.concurrent_pair &&
                               !(std::getenv("IPD_NO_RES_POLY4") && std::getenv("IPD_NO_RES_POLY4")[0] == '1') &&
                               !(std::getenv("IPD_NO_RESIDENT_DEEP") && std::getenv("IPD_NO_RESIDENT_DEEP")[0] == '1') &&
                               !(std::getenv("IPD_NO_RESIDENT_BIG") && std::getenv("IPD_NO_RESIDENT_BIG")[0] == '1') &&
                               !(std::getenv("IPD_NO_RESIDENT") && std::getenv("IPD_NO_RESIDENT")[0] == '1');
            if (root5) k_small = std::max(k_small, 5);
            if (k_small < h->J) {
                // ... and everything below it fits in LDS (the stage area holds N_root doubles)
                for (int kroot = k_small; kroot < h->J; ++kroot) {
                    // the generic phases (and their staging vector) only run when IPD_NO_BLK is set
                    const size_t stage = lean_vectors ? 16 : r16(sizeof(double) * (size_t)h->L[kroot].A.nr);
                    size_t used = 0;
                    const int k_lds = plan_lds(stage, &used);
                    // the root itself does not fit beside the deeper levels but has at most BT rows (a
                    // level 3 of 170-310 rows with 40-100 entries each in the m=n=1024 runs): it becomes a
                    // semi-cached root -- vectors in LDS, rows walked from L2 by several lanes each
                    // (glb_rowdot_range), 2-3 us per sweep against 5 us for the launch it replaces
                    bool semi_root = false;
                    // (only with short rows, <= 12 entries on average like the semi-cached level 2: measured on
                    // the Newton systems of the m=n=1024 Class 1 run, a level 3 of 2-3 k entries gains 5-9 % per W
                    // cycle as launches and opens the hierarchy to the resident kernel's remote tail, -15...-23 %;
                    // with 4 k entries it loses 12 %, with 9-17 k entries a sweep from L2 through one CU costs
                    // more than the launch: 0.72 -> 1.09 ms, 0.54 -> 1.21 ms per W cycle)
                    if (k_lds == kroot + 1 && kroot >= 3 && lean_vectors && h->L[kroot].A.nr <= BT &&
                        (double)h->L[kroot].A.nnz <= 12.0 * h->L[kroot].A.nr &&
                        (double)h->L[kroot + 1].P.nnz <= 12.0 * h->L[kroot].A.nr &&
                        used + 3 * r16(8 * (size_t)h->L[kroot].A.nr) <= 150 * 1024) {
                        semi_root = true;
                    }
                    if (k_lds > kroot && !semi_root) continue;
                    std::unique_ptr<SolveDesc> sd(new SolveDesc());
                    fill_desc(sd.get());
                    sd->k_lds = kroot;
                    if (semi_root) sd->k_semi = kroot;
                    sd->k_tiny = tiny_from(kroot + (semi_root ? 1 : 0));
                    sd->k_blk = blk_from(kroot);
                    sd->stage_bytes = (int)stage;
                    sd->root_r = h->L[kroot].r;
                    sd->root_e = h->L[kroot].e;
                    st->k_sub = kroot;
                    st->sub_semi_root = semi_root;
                    st->d_sub = build_image(sd.get(), kroot, stage, &st->sub_lds, &st->sub_bm);
                    break;
                }
            }
        }
    }
    // (b2) Where level 3 is only a semi-cached root (its rows come from L2), the level-resident kernel
    // does better with level 3 in registers and its tail rooted at level 4 (plan_resident, `three`:
    // 0.50-0.51 against 0.57-0.61 ms per W cycle on the Newton systems of the m=n=1024 Class 1 run), so a
    // second image rooted at level 4 is packed for it.  (Where levels 3..J fit the image as they are,
    // the tail rooted at level 3 stays 3-6 % ahead: 0.49-0.51 against 0.51-0.54 ms.)
    // With level 3 in polynomial form (plan_resident, poly3: a visit of it is three hand-offs instead of
    // thirteen) the same holds wherever that form applies, semi-cached root or not: 0.33-0.36 -> see DESIGN.
    const bool poly3_likely =
        h->J >= 5 && h->opts.smoth >= 1 && h->L[1].nf > 0 &&
        h->L[4].A.nr <= 128 &&
        h->L[4].A.nr <= std::max(cdiv(std::max(h->L[1].nf, h->L[1].A.nr - h->L[1].nf), RES_WAVES), cdiv(h->L[2].A.nr, RES_WAVES)) &&
        !(std::getenv("IPD_NO_POLY") && std::getenv("IPD_NO_POLY")[0] == '1');
    if (st->k_sub == 3 && (st->sub_semi_root || poly3_likely) && st->d_sub && h->J >= 5 && h->J <= SOLVE_ML && h->L[3].A.nr <= BT &&
        h->L[4].A.nr <= BT && st->run[3].maxoff <= 512 && h->L[1].nf > 0 &&
        !(std::getenv("IPD_NO_RESIDENT_THREE") && std::getenv("IPD_NO_RESIDENT_THREE")[0] == '1') &&
        !(std::getenv("IPD_NO_RESIDENT") && std::getenv("IPD_NO_RESIDENT")[0] == '1')) {
        const size_t stage = 16;
        size_t used = 0;
        bool ok = lean_vectors;
        for (int k = 4; k <= h->J && ok; ++k) ok = small_level(k);
        if (ok && plan_lds(stage, &used) <= 4) {
            std::unique_ptr<SolveDesc> sd(new SolveDesc());
            fill_desc(sd.get());
            sd->k_lds = 4;
            sd->k_tiny = tiny_from(4);
            sd->k_blk = blk_from(4);
            sd->stage_bytes = (int)stage;
            sd->root_r = h->L[4].r;
            sd->root_e = h->L[4].e;
            st->d_sub4 = build_image(sd.get(), 4, stage, &st->sub4_lds, &st->sub4_bm);
        }
    }
    // (b3) No sub-cycle at all because level 3's interpolation is big (P_3 with more than 40 k entries:
    // a dense 1024 x 50 block early in a run), although levels 3..J themselves are small: the launch
    // path would gain nothing from an image whose restriction and prolongation stay launches, but the
    // resident kernel's remote tail does not use P_3 from the image -- its workgroups apply it -- so an
    // image rooted at level 3 is packed for it alone.
    if (st->k_sub == 0 && !st->small_ok && h->J >= 4 && h->J <= SOLVE_ML && lean_vectors && h->L[3].A.nr <= BT &&
        h->L[1].nf > 0 && (h->opts.cycle == 'w' || h->opts.cycle == 'v') &&
        !(std::getenv("IPD_NO_RESIDENT") && std::getenv("IPD_NO_RESIDENT")[0] == '1') &&
        !(std::getenv("IPD_NO_SUBCYCLE") && std::getenv("IPD_NO_SUBCYCLE")[0] == '1')) {
        bool ok = true;
        for (int k = 3; k <= h->J && ok; ++k)
            ok = h->L[k].A.nr <= 1024 && h->L[k].A.nnz <= 40000 && (k == 3 || h->L[k].P.nnz <= 40000);
        const size_t stage = 16;
        size_t used = 0;
        if (ok && plan_lds(stage, &used) <= 3) {
            std::unique_ptr<SolveDesc> sd(new SolveDesc());
            fill_desc(sd.get());
            sd->k_lds = 3;
            sd->k_tiny = tiny_from(3);
            sd->k_blk = blk_from(3);
            sd->stage_bytes = (int)stage;
            sd->root_r = h->L[3].r;
            sd->root_e = h->L[3].e;
            st->d_sub3 = build_image(sd.get(), 3, stage, &st->sub3_lds, &st->sub3_bm);
        }
    }
    // (b4) the image rooted at level 5 (see root5 above) is the one the deep mode's tail workgroup takes
    if (st->k_sub == 5 && st->d_sub && !st->sub_semi_root && h->J >= 6 && h->L[1].A.nr > RES_NMAX) {
        st->d_sub5 = st->d_sub;
        st->sub5_lds = st->sub_lds;
    }
    // (b5) ... and k_resident's POLY3 mode (level 1 of at most 2048 rows) keeps level 4 in polynomial form in its
    // workgroups as well when there are six levels or more (ResDesc::p4rows).  Its tail workgroup takes the
    // image rooted at level 4 that the POLY3 mode uses anyway and enters it at level 5 (an image of its own,
    // rooted at level 5, packed levels 5..J a second time: +85 us per hierarchy, more than the cycles gained).
    if (!st->d_sub5 && poly3_likely && h->J >= 6 && h->L[1].A.nr <= RES_NMAX &&
        ((st->k_sub == 3 && st->d_sub4) || (st->k_sub == 4 && st->d_sub && !st->sub_semi_root)) &&
        h->L[4].A.nr <= RES_P4_SEG && h->L[5].A.nr <= 64 &&
        !(std::getenv("IPD_NO_RES_POLY4") && std::getenv("IPD_NO_RES_POLY4")[0] == '1')) {
        st->d_sub5 = st->k_sub == 3 ? st->d_sub4 : st->d_sub;
        st->sub5_lds = st->k_sub == 3 ? st->sub4_lds : st->sub_lds;
    }
    plan_resident(h, st.get());
    if (const char* dbg = std::getenv("IPD_DEBUG_LEVELS"); dbg && dbg[0] == '1') {
        std::fprintf(stderr, "[ipd] J=%d small=%d k_sub=%d resident=%d(G=%d,KE=%d) levels:", h->J,
                     (int)st->small_ok, st->k_sub, (int)st->res_ok, st->res_G, st->res_ke);
        for (int k = 1; k <= h->J; ++k) std::fprintf(stderr, " %d/%d", h->L[k].A.nr, h->L[k].A.nnz);
        std::fprintf(stderr, "\n");
    }
    h->cyc = std::shared_ptr<CycleState>(st.release());
}

// ---------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------
#define IPD_LAUNCH_SP(kern, staged, pad, grid, dyn, ...)                                          \
    do {                                                                                          \
        if (staged) {                                                                             \
            if (pad)                                                                              \
                hipLaunchKernelGGL((kern<true, true>), dim3(grid), dim3(BT), dyn, ctx->stream,     \
                                   __VA_ARGS__);                                                  \
            else                                                                                  \
                hipLaunchKernelGGL((kern<true, false>), dim3(grid), dim3(BT), dyn, ctx->stream,    \
                                   __VA_ARGS__);                                                  \
        } else {                                                                                  \
            if (pad)                                                                              \
                hipLaunchKernelGGL((kern<false, true>), dim3(grid), dim3(BT), 0, ctx->stream,      \
                                   __VA_ARGS__);                                                  \
            else                                                                                  \
                hipLaunchKernelGGL((kern<false, false>), dim3(grid), dim3(BT), 0, ctx->stream,     \
                                   __VA_ARGS__);                                                  \
        }                                                                                         \
        IPD_KERNEL_CHECK();                                                                       \
    } while (0)

// ---- fused-program emitter -----------------------------------------------------------
// A phase is "small" when one workgroup covers its rows in ONE pass and its matrix slice
// is a few thousand entries: then it costs 1-3 us inside a fused program against >= 5 us
// as a launch of its own.  Larger phases lose inside a single workgroup (one CU issues
// ~60 B/clk of loads: tools/ubench_small.hip) and stay separate launches.
static bool phase_is_small(const CycleState* st, int rows, int L, double nnz_est, int stage_len) {
    return st->fuse_enabled && stage_len <= STAGE_MAX && (long long)rows * L <= (long long)BT &&
           nnz_est <= 6000.0;
}

static void flush_fused(ipd_ctx* ctx, CycleState* st) {
    if (st->pending.n == 0) return;
    hipLaunchKernelGGL(k_fused, dim3(1), dim3(BT), st->pending_lds, ctx->stream, st->pending);
    IPD_KERNEL_CHECK();
    st->pending.n = 0;
    st->pending_lds = 0;
}

static PhaseDesc& push_phase(ipd_ctx* ctx, CycleState* st, int type, int stage_len) {
    if (st->pending.n == FUSED_MAX) flush_fused(ctx, st);
    PhaseDesc& d = st->pending.d[st->pending.n++];
    d.type = type;
    d.pad_ = 0;
    st->pending_lds = std::max(st->pending_lds, sizeof(double) * (size_t)stage_len);
    return d;
}

// Runs `launch(r0, r1)` over the row range [lo, hi) of a level.  Unsharded: one call.
// Sharded: this rank's slice only, followed by one grouped RCCL all-gather of the
// vectors the launch produced (each rank wrote its own slice of every one of them).
template <class F>
static void run_rows(ipd_ctx* ctx, CycleState* st, int lo, int hi, F launch,
                     std::initializer_list<double*> produced) {
    flush_fused(ctx, st);  // big launch: everything queued before it must run first
    const int G = st->shard_ranks;
    const int rows = hi - lo;
    if (G <= 1 || rows % G != 0 || rows < st->shard_min_rows) {  // replicated level
        launch(lo, hi);
        return;
    }
    const int cnt = rows / G;
    if (st->shard_emulate) {
        for (int vr = 0; vr < G; ++vr) launch(lo + vr * cnt, lo + (vr + 1) * cnt);
        return;
    }
    launch(lo + st->shard_rank * cnt, lo + (st->shard_rank + 1) * cnt);
    double* bases[4];
    int nv = 0;
    for (double* v : produced)
        if (v) bases[nv++] = v + lo;
    comm_allgather_inplace(ctx, bases, nv, cnt);
}

static void launch_smooth(ipd_ctx* ctx, const SmoothArgs& a, int cu) {
    const size_t dyn = a.staged ? sizeof(double) * (size_t)a.lv.N : 0;
    const int grid = pick_blocks(a.row1 - a.row0, a.lv.L, cu);
    IPD_LAUNCH_SP(k_smooth, a.staged, a.lv.S > 0, grid, dyn, a);
}

static void launch_xfer(ipd_ctx* ctx, const XferArgs& a, int cu) {
    const size_t dyn = a.staged ? sizeof(double) * (size_t)a.ncols : 0;
    const int grid = pick_blocks(a.row1 - a.row0, a.L, cu);
    if (a.staged)
        hipLaunchKernelGGL(k_xfer<true>, dim3(grid), dim3(BT), dyn, ctx->stream, a);
    else
        hipLaunchKernelGGL(k_xfer<false>, dim3(grid), dim3(BT), 0, ctx->stream, a);
    IPD_KERNEL_CHECK();
}

static void launch_resid(ipd_ctx* ctx, const LevelRun& rn, const double* e, int r0, int r1,
                         int cu) {
    const size_t dyn = rn.staged ? sizeof(double) * (size_t)rn.dev.N : 0;
    const int grid = pick_blocks(r1 - r0, rn.dev.L, cu);
    IPD_LAUNCH_SP(k_resid, rn.staged, rn.dev.S > 0, grid, dyn, rn.dev, e, r0, r1);
}

// one smoother sweep on level k: Jacobi = one launch, bigraph GS = two half launches
static void launch_sweep(ipd_amg* h, CycleState* st, int k, int isnsp, bool post) {
    ipd_ctx* ctx = h->ctx;
    Level& lv = h->L[k];
    LevelRun& rn = st->run[(size_t)k];
    SmoothArgs a;
    a.lv = rn.dev;
    a.eold = lv.e;
    a.enew = lv.e2;
    a.win = lv.w;
    a.wout = lv.w;
    a.isnsp = isnsp;
    a.staged = rn.staged;
    a.eold_zero = rn.e_zero ? 1 : 0;
    const int cu = st->num_cu;
    const int rows_launch = lv.nf > 0 ? std::max(lv.nf, lv.N - lv.nf) : lv.N;
    const bool small = a.staged && phase_is_small(st, rows_launch, a.lv.L,
                                                  (double)lv.A.nnz * rows_launch / std::max(lv.N, 1),
                                                  lv.N);
    auto go = [&](int r0, int r1) {
        a.row0 = r0;
        a.row1 = r1;
        if (small)
            push_phase(ctx, st, PH_SMOOTH, lv.N).u.s = a;
        else
            launch_smooth(ctx, a, cu);
    };
    if (small) {  // replicated on every rank, queued into the fused program
        if (lv.nf == 0) {
            a.u0 = a.u1 = 0;
            a.wout = nullptr;
            go(0, lv.N);
        } else {
            const int f0 = post ? lv.nf : 0, f1 = post ? lv.N : lv.nf;
            const int s0 = post ? 0 : lv.nf, s1 = post ? lv.nf : lv.N;
            a.u0 = a.u1 = 0;
            go(f0, f1);
            a.u0 = f0;
            a.u1 = f1;
            a.wout = nullptr;
            go(s0, s1);
        }
    } else if (lv.nf == 0) {
        a.u0 = a.u1 = 0;
        a.wout = nullptr;
        run_rows(ctx, st, 0, lv.N, go, {a.enew});
    } else if (k == 1 && st->mask_ok) {
        // bit-mask operator: same two half sweeps, 1 bit per matrix entry; sharded runs give each
        // owner its block of the half's rows (a row range inside one half is all the kernel needs)
        const MaskOp& mo = st->maskop;
        const size_t dyn = sizeof(double) * 64 * (size_t)std::max(mo.nwf, mo.nwc);
        auto half = [&](int r0, int r1) {
            a.row0 = r0;
            a.row1 = r1;
            const int nwh = (r0 < mo.nf) ? mo.nwf : mo.nwc;
            const int grid = std::max(1, cdiv(r1 - r0, std::min(MASK_RW, 64 / nwh) * (BT / 64)));
            hipLaunchKernelGGL(k_smooth_mask, dim3(grid), dim3(BT), dyn, ctx->stream, a, mo);
            IPD_KERNEL_CHECK();
        };
        const int f0 = post ? lv.nf : 0, f1 = post ? lv.N : lv.nf;
        const int s0 = post ? 0 : lv.nf, s1 = post ? lv.nf : lv.N;
        a.u0 = a.u1 = 0;
        run_rows(ctx, st, f0, f1, half, {a.enew, a.wout});
        a.u0 = f0;
        a.u1 = f1;
        a.wout = nullptr;
        run_rows(ctx, st, s0, s1, half, {a.enew});
    } else {
        // pre: F rows then C rows (Rk{1});  post: C rows then F rows (Rk{1}')
        const int f0 = post ? lv.nf : 0, f1 = post ? lv.N : lv.nf;  // first half rows
        const int s0 = post ? 0 : lv.nf, s1 = post ? lv.nf : lv.N;  // second half rows
        a.u0 = a.u1 = 0;
        run_rows(ctx, st, f0, f1, go, {a.enew, a.wout});
        a.u0 = f0;
        a.u1 = f1;
        a.wout = nullptr;
        run_rows(ctx, st, s0, s1, go, {a.enew});
    }
    rn.e_zero = false;
    std::swap(lv.e, lv.e2);
}

__global__ void k_maskop_scales(int nf, int nc, const double* __restrict__ p,
                                const double* __restrict__ q, double itk,
                                double* __restrict__ alpha, double* __restrict__ beta) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nf + nc; t += gridDim.x * blockDim.x) {
        if (t < nf)
            alpha[t] = q[t] * q[t] * itk;
        else
            beta[t - nf] = p[t - nf] * p[t - nf];
    }
}

// Derives the bit-mask form of level 1 from its CSR arrays; keeps the CSR kernels (returns
// false) unless A_1 is exactly Hybrid_AMG's rescaled operator for these p, q, tk.
//
// When it pays (`policy` true: the solvers' own call): the mask sweep moves 13x fewer bytes but is
// the slower launch while the level is latency-bound -- regime D at m = n = 1024 (2.1 M entries):
// 5.7 us against 5.2 us for the padded CSR sweep -- and the faster one once the CSR sweep is
// bandwidth-bound -- m = n = 2048 (8.4 M entries): 8.3 us against 13.4 us.  The solvers therefore
// attach it from 4 M entries on; IPD_MASKOP=1 lowers that to 16 entries per row, IPD_NO_MASKOP=1
// switches it off.  An explicit ipd_amg_attach_mask_operator call is not subject to the policy.
bool amg_attach_maskop(ipd_amg* h, const double* p_dev, const double* q_dev, int m, int n, double tk,
                       bool policy, bool transfers_only) {
    ipd_ctx* ctx = h->ctx;
    CycleState* st = state_of(h);
    if (!st) return false;
    const Level& lv = h->L[1];
    // the level-resident kernel takes its level 1 <-> 2 transfers from the mask whatever the size
    const bool for_resident = st->res_ok && !st->res_desc.three && st->res_desc.wident && h->J == 3 &&
                              true;
    bool sweeps_too = !transfers_only;
    const bool big_forced = std::getenv("IPD_RESIDENT_BIG") && std::getenv("IPD_RESIDENT_BIG")[0] == '1';
    // Realistic hierarchy with a level 1 beyond k_resident's 2048 rows (the Newton systems of the m = n = 2048
    // runs): candidate for the mask-form kernel's DEEP mode (ipd_resident_big.h) -- it needs the bit mask
    // whatever the population of the rows
    const SolveDesc* deep_img = nullptr;
    size_t deep_img_lds = 0;
    bool deep_cand = false;
    {
        const char* nrs = std::getenv("IPD_NO_RESIDENT");
        const char* nbg = std::getenv("IPD_NO_RESIDENT_BIG");
        const char* ndp = std::getenv("IPD_NO_RESIDENT_DEEP");
        const bool cyc = h->opts.cycle == 'w' || h->opts.cycle == 'v';
        if ((st->k_sub == 4 && st->d_sub) || (st->k_sub == 5 && st->d_sub5)) {   // (rooted at 5: POLY4 only, below)
            deep_img = st->d_sub;
            deep_img_lds = st->sub_lds;
        } else if (st->k_sub == 3 && st->d_sub4) {
            deep_img = st->d_sub4;
            deep_img_lds = st->sub4_lds;
        }
        deep_cand = !(nrs && nrs[0] == '1') && !(nbg && nbg[0] == '1') && !(ndp && ndp[0] == '1') && !st->res_off &&
                    !st->res_ok && !st->resb && !st->small_ok && !h->opts.twogrid && cyc && h->opts.smoth >= 1 &&
                    h->J >= 5 && (n + m > RES_NMAX || big_forced) && n <= RB_HALF && m <= RB_HALF && lv.nf == n &&
                    lv.N == m + n && h->L[2].A.nr == m && h->L[3].A.nr <= RB_N3MAX && h->L[4].A.nr <= RB_N4MAX &&
                    deep_img != nullptr && std::max(RB_LDS_BYTES, deep_img_lds) <= (size_t)156 * 1024;
    }
    if (transfers_only && !for_resident && !big_forced && !deep_cand) return false;
    if (policy) {
        const char* on = std::getenv("IPD_MASKOP");
        if (!(on && on[0] == '1') && (double)lv.A.nnz < 4.0e6) {
            if (!for_resident && !deep_cand) return false;
            sweeps_too = false;   // below the size where the mask SWEEPS of the launch path pay
        }
    }
    if (h->J < 2 || lv.nf != n || lv.N != m + n || tk == 0.0) return false;
    // a row of the mask costs nw word walks whatever its population: with fewer than ~16 entries
    // per row the padded CSR sweep always beats it
    if ((double)lv.A.nnz < 16.0 * lv.N) {
        if (!deep_cand) return false;
        sweeps_too = false;
    }
    if (std::max(m, n) > 4096) return false;   // a row's mask words must fit one wave (64 words)
    Arena& ar = *h->arena;
    MaskOp mo;
    mo.nf = n;
    mo.nc = m;
    mo.nwf = cdiv(m, 64);
    mo.nwc = cdiv(n, 64);
    unsigned long long* fb = ar.alloc<unsigned long long>((size_t)n * mo.nwf);
    unsigned long long* cb = ar.alloc<unsigned long long>((size_t)m * mo.nwc);
    double* alpha = ar.alloc<double>((size_t)n);
    double* beta = ar.alloc<double>((size_t)m);
    double* diag = ar.alloc<double>((size_t)lv.N);
    int* bad = ctx->scratch->alloc<int>(1);
    IPD_HIP(hipMemsetAsync(fb, 0, sizeof(unsigned long long) * (size_t)n * mo.nwf, ctx->stream));
    IPD_HIP(hipMemsetAsync(cb, 0, sizeof(unsigned long long) * (size_t)m * mo.nwc, ctx->stream));
    IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_maskop_scales, dim3(cdiv(lv.N, 256)), dim3(256), 0, ctx->stream, n, m, p_dev,
                       q_dev, 1.0 / tk, alpha, beta);
    hipLaunchKernelGGL(k_maskop_build, dim3(std::max(1, std::min(cdiv(lv.N, 4), 2048))), dim3(256), 0,
                       ctx->stream, lv.N, n, lv.A.rp, lv.A.ci, lv.A.va, (const double*)alpha,
                       (const double*)beta, mo.nwf, mo.nwc, fb, cb, diag, bad);
    IPD_KERNEL_CHECK();
    if (ctx->fetch1(bad) != 0) return false;
    mo.fbits = fb;
    mo.cbits = cb;
    mo.alpha = alpha;
    mo.beta = beta;
    mo.diag = diag;
    if (for_resident && lv.nf <= RES_NMAX / 2 && h->L[2].A.nr == m) {
        // W(j,i) = s_ij beta_i rho_j: rho from the row sums (isnsp: rows normalised to sum 1, transfer.m:22-24)
        // or alpha_j / A_jj, then every entry of P checked against the form
        double* rho = ar.alloc<double>((size_t)n);
        IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_res_xmask_rho, dim3(cdiv(n, 4)), dim3(256), 0, ctx->stream, n, m, h->opts.isnsp,
                           (const unsigned long long*)fb, mo.nwf, (const double*)alpha, (const double*)beta,
                           (const double*)diag, h->L[2].P.rp, h->L[2].P.ci, h->L[2].P.va, rho, bad);
        IPD_KERNEL_CHECK();
        if (ctx->fetch1(bad) == 0) {
            ResDesc& D = st->res_desc;
            D.xm = 1;
            D.xm_nwf = mo.nwf;
            D.xm_nwc = mo.nwc;
            D.xm_fbits = fb;
            D.xm_cbits = cb;
            D.xm_beta = beta;
            D.xm_rho = rho;
        }
    }
    // Level 1 beyond k_resident's 2048 rows (m = n = 2048: BASELINE config 4's size), three levels with a
    // one-row tail: the mask-form resident kernel (ipd_resident_big.h).  IPD_RESIDENT_BIG=1 prefers it
    // wherever it applies (tests), IPD_NO_RESIDENT_BIG=1 switches it off.
    {
        const char* nrs = std::getenv("IPD_NO_RESIDENT");
        const char* nbg = std::getenv("IPD_NO_RESIDENT_BIG");
        const char* fbg = std::getenv("IPD_RESIDENT_BIG");
        const bool forced = fbg && fbg[0] == '1';
        const bool cyc = h->opts.cycle == 'w' || h->opts.cycle == 'v';
        const int G = cdiv(std::max(n, m), RES_WAVES);
        if (!(nrs && nrs[0] == '1') && !st->res_off && !(nbg && nbg[0] == '1') &&
            (forced || (!st->res_ok && n + m > RES_NMAX)) && !st->small_ok &&
            !st->resb && h->J == 3 && h->L[3].A.nr == 1 && n <= RB_HALF && m <= RB_HALF && h->L[2].A.nr == m && cyc &&
            !h->opts.twogrid && G <= st->num_cu && G <= std::min(n, m) && h->opts.smoth >= 1) {
            LevelDev d2 = st->run[2].dev;
            if (d2.S <= 0 && st->run[2].maxoff > 0) {   // private padded copy, stride = the longest row
                d2.S = (st->run[2].maxoff + 3) / 4 * 4;
                const Csr& A2 = h->L[2].A;
                unsigned short* pci = ar.alloc<unsigned short>((size_t)A2.nr * d2.S);
                double* pva = ar.alloc<double>((size_t)A2.nr * d2.S);
                double* dg = ar.alloc<double>((size_t)A2.nr);
                hipLaunchKernelGGL(k_pad_build, dim3(std::max(1, std::min(cdiv(A2.nr, 4), 4096))), dim3(256), 0,
                                   ctx->stream, A2.nr, d2.S, A2.rp, A2.ci, A2.va, pci, pva, dg);
                IPD_KERNEL_CHECK();
                d2.pci = pci;
                d2.pva = pva;
                d2.diag = dg;
            }
            double* rho = const_cast<double*>(st->res_desc.xm_rho);
            bool rho_ok = st->res_desc.xm != 0;
            if (!rho_ok) {
                rho = ar.alloc<double>((size_t)n);
                IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
                hipLaunchKernelGGL(k_res_xmask_rho, dim3(cdiv(n, 4)), dim3(256), 0, ctx->stream, n, m, h->opts.isnsp,
                                   (const unsigned long long*)fb, mo.nwf, (const double*)alpha, (const double*)beta,
                                   (const double*)diag, h->L[2].P.rp, h->L[2].P.ci, h->L[2].P.va, rho, bad);
                IPD_KERNEL_CHECK();
                rho_ok = ctx->fetch1(bad) == 0;
            }
            if (rho_ok && d2.S > 0 && d2.S <= 64 * 32) {
                ResBigDesc B{};
                B.nf = n;
                B.nc = m;
                B.N2 = m;
                B.S2 = d2.S;
                B.pci2 = d2.pci;
                B.pva2 = d2.pva;
                B.diag2 = d2.diag;
                B.dinv2 = d2.dinv;
                B.Axi2 = d2.Axi;
                B.xx2 = d2.xx;
                B.diag1 = diag;
                B.dinv1 = st->run[1].dev.dinv;
                B.Axi1 = st->run[1].dev.Axi;
                B.xx1 = st->run[1].dev.xx;
                B.fbits = fb;
                B.cbits = cb;
                B.nwf = mo.nwf;
                B.nwc = mo.nwc;
                B.alpha = alpha;
                B.beta = beta;
                B.rho = rho;
                B.P3.rp = h->L[3].P.rp;
                B.P3.ci = h->L[3].P.ci;
                B.P3.va = h->L[3].P.va;
                B.A3.rp = h->L[3].A.rp;
                B.A3.ci = h->L[3].A.ci;
                B.A3.va = h->L[3].A.va;
                B.nu = h->opts.smoth;
                B.isnsp = h->opts.isnsp;
                B.wcycle = h->opts.cycle == 'w';
                B.anycycle = 1;
                B.maxit = h->opts.maxit;
                B.retol = h->opts.retol;
                B.pcg_maxit = h->opts.pcg_maxit;
                B.pollsleep = 1;
                B.presleep = 13;
                const size_t gbytes = (size_t)RB_GRAN * 16;
                B.ranks = 1;   // rank groups with a granule buffer each (test hook, see ResBigDesc::ranks)
                if (const char* e = std::getenv("IPD_RESIDENT_RANKS")) B.ranks = std::max(1, std::min(8, std::atoi(e)));
                if (B.ranks > G) B.ranks = 1;
                st->res_block_bytes = (size_t)B.ranks * 2 * gbytes + 16;
                st->res_block = reinterpret_cast<unsigned char*>(ar.alloc_bytes(st->res_block_bytes));
                B.gran = st->res_block;
                B.tmo = reinterpret_cast<unsigned*>(st->res_block + (size_t)B.ranks * 2 * gbytes);
                B.dbg_skip_seq = 0;
                st->resb_desc = B;
                st->resb_ke2 = d2.S <= 64 * 16 ? 16 : 32;
                st->resb = true;
                st->res_remote = false;
                st->res_ke3 = 0;
                st->res_G = G;
                st->res_lds = RB_LDS_BYTES;
                st->res_capacity = -1;
                st->res_desc.dbg_skip_seq = 0;
                if (const char* e = std::getenv("IPD_RES_DEBUG_SKIP_PUBLISH")) st->res_desc.dbg_skip_seq = (unsigned)std::max(0, std::atoi(e));
                if (!st->res_out) st->res_out = ar.alloc<double>(4 + 2 * ((size_t)std::max(h->opts.maxit, 0) + 2));
                st->res_ok = true;
            }
        }
    }
    // DEEP mode of the mask-form kernel: realistic hierarchies (five levels and more) whose level 1 exceeds
    // k_resident's 2048 rows.  Level 2 as short register slices, level 3 in polynomial form (pack_bpoly in the
    // RB_P3_SEG row layout), the LDS image rooted at level 4 for the tail workgroup; G <= 255 workgroups
    // (the tail needs a compute unit of its own), two rows of each block per wave.
    if (deep_cand && !st->resb) {
        const int N3 = h->L[3].A.nr, N4 = h->L[4].A.nr;
        int G = std::max(std::max(cdiv(std::max(n, m), 2 * RES_WAVES), cdiv(N3, 4)), std::max(N4, 128));
        if (const char* e = std::getenv("IPD_RESIDENT_G")) G = std::max(G, std::atoi(e));
        LevelDev d2 = st->run[2].dev;
        if (d2.S <= 0 && st->run[2].maxoff > 0) d2.S = -((st->run[2].maxoff + 3) / 4 * 4);   // private copy wanted
        const int S2 = std::abs(d2.S);
        // level 4 resident as well (POLY4), the tail rooted at level 5: the only form an image rooted at level 5 serves
        const int N5 = h->J >= 6 ? h->L[5].A.nr : 0;
        const bool poly4 = st->d_sub5 && st->k_sub == 5 && N4 <= 2 * G && N5 >= 1 && N5 <= G && N5 <= RB_N5MAX &&
                           N4 + G <= BT && std::max(RB_LDS_BYTES, st->sub5_lds) <= (size_t)156 * 1024;
        if (G + 1 <= st->num_cu && G <= std::min(n, m) && S2 > 0 && S2 <= 64 * 8 && (poly4 || st->k_sub != 5)) {
            if (d2.S < 0) {
                d2.S = S2;
                const Csr& A2 = h->L[2].A;
                unsigned short* pci = ar.alloc<unsigned short>((size_t)A2.nr * d2.S);
                double* pva = ar.alloc<double>((size_t)A2.nr * d2.S);
                double* dg = ar.alloc<double>((size_t)A2.nr);
                hipLaunchKernelGGL(k_pad_build, dim3(std::max(1, std::min(cdiv(A2.nr, 4), 4096))), dim3(256), 0,
                                   ctx->stream, A2.nr, d2.S, A2.rp, A2.ci, A2.va, pci, pva, dg);
                IPD_KERNEL_CHECK();
                d2.pci = pci;
                d2.pva = pva;
                d2.diag = dg;
            }
            double* rho = ar.alloc<double>((size_t)n);
            IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(k_res_xmask_rho, dim3(cdiv(n, 4)), dim3(256), 0, ctx->stream, n, m, h->opts.isnsp,
                               (const unsigned long long*)fb, mo.nwf, (const double*)alpha, (const double*)beta,
                               (const double*)diag, h->L[2].P.rp, h->L[2].P.ci, h->L[2].P.va, rho, bad);
            IPD_KERNEL_CHECK();
            if (ctx->fetch1(bad) == 0) {
                const BPolyDev pb = pack_bpoly(ctx, h, st, 3, h->opts.isnsp, 0, true, RB_P3_SEG, RB_P3_LD);
                st->level_forms.resize((size_t)h->J + 1, 0);
                st->level_forms[3] |= 64;
                BPolyDev pb4;
                if (poly4) {
                    pb4 = pack_bpoly(ctx, h, st, 4, h->opts.isnsp, 0, true, RB_P4_SEG, RB_P4_LD);
                    st->level_forms[4] |= 64;
                }
                ResBigDesc B{};
                B.nf = n;
                B.nc = m;
                B.N2 = m;
                B.S2 = d2.S;
                B.pci2 = d2.pci;
                B.pva2 = d2.pva;
                B.diag2 = d2.diag;
                B.dinv2 = d2.dinv;
                B.Axi2 = d2.Axi;
                B.xx2 = d2.xx;
                B.diag1 = diag;
                B.dinv1 = st->run[1].dev.dinv;
                B.Axi1 = st->run[1].dev.Axi;
                B.xx1 = st->run[1].dev.xx;
                B.fbits = fb;
                B.cbits = cb;
                B.nwf = mo.nwf;
                B.nwc = mo.nwc;
                B.alpha = alpha;
                B.beta = beta;
                B.rho = rho;
                B.P3.rp = h->L[3].P.rp;   // (unused in DEEP mode)
                B.P3.ci = h->L[3].P.ci;
                B.P3.va = h->L[3].P.va;
                B.A3.rp = h->L[3].A.rp;
                B.A3.ci = h->L[3].A.ci;
                B.A3.va = h->L[3].A.va;
                B.N3 = N3;
                B.N4 = N4;
                B.Pt3.rp = h->L[3].Pt.rp;
                B.Pt3.ci = h->L[3].Pt.ci;
                B.Pt3.va = h->L[3].Pt.va;
                B.P3d.rp = h->L[3].P.rp;
                B.P3d.ci = h->L[3].P.ci;
                B.P3d.va = h->L[3].P.va;
                B.p3rows = pb.M;
                B.p3w = pb.W;
                B.N5 = poly4 ? N5 : 0;
                B.p4rows = pb4.M;
                B.p4w = pb4.W;
                B.nu = h->opts.smoth;
                B.isnsp = h->opts.isnsp;
                B.wcycle = h->opts.cycle == 'w';
                B.anycycle = 1;
                B.maxit = h->opts.maxit;
                B.retol = h->opts.retol;
                B.pcg_maxit = h->opts.pcg_maxit;
                B.pollsleep = 1;
                B.presleep = 13;
                const size_t gbytes = (size_t)RB_GRAN * 16, tbytes = (size_t)RES_GRAN_MAX * 16;
                st->res_block_bytes = 2 * gbytes + 16 + 4 * tbytes + 16;
                st->res_block = reinterpret_cast<unsigned char*>(ar.alloc_bytes(st->res_block_bytes));
                B.gran = st->res_block;
                B.tmo = reinterpret_cast<unsigned*>(st->res_block + 2 * gbytes);
                B.dbg_skip_seq = 0;
                B.sub = poly4 ? st->d_sub5 : deep_img;
                if (poly4) deep_img_lds = st->sub5_lds;
                B.tin = st->res_block + 2 * gbytes + 16;
                B.tout = st->res_block + 2 * gbytes + 16 + 2 * tbytes;
                B.tctl = reinterpret_cast<unsigned*>(st->res_block + 2 * gbytes + 16 + 4 * tbytes);
                st->resb_desc = B;
                st->resb_ke2 = d2.S <= 64 * 4 ? 4 : 8;
                st->resb = true;
                st->resb_deep = true;
                st->resb_poly4 = poly4;
                st->res_remote = true;
                st->res_ke3 = 1;
                st->res_G = G;
                st->res_lds = std::max(RB_LDS_BYTES, deep_img_lds);
                st->res_capacity = -1;
                st->res_desc.dbg_skip_seq = 0;
                if (const char* e = std::getenv("IPD_RES_DEBUG_SKIP_PUBLISH")) st->res_desc.dbg_skip_seq = (unsigned)std::max(0, std::atoi(e));
                if (!st->res_out) st->res_out = ar.alloc<double>(4 + 2 * ((size_t)std::max(h->opts.maxit, 0) + 2));
                st->res_ok = true;
            }
        }
    }
    if (!sweeps_too) return st->res_desc.xm != 0 || st->resb;
    st->maskop = mo;
    st->mask_ok = true;
    // captured graphs (if any) were recorded with the CSR sweeps
    for (auto& g : st->gexec)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    return true;
}

// Level 2 of the level-resident kernel in polynomial form, composed over a whole visit (ResDesc::p2rows):
// three levels with a one-row tail, V cycle, 16-entry slices -- the metric's workload.  Packing costs five
// dense products of N2^3 (0.5 ms at N2 = 1024) against 18 us saved per cycle: it pays where many cycles run
// on one hierarchy (bench.py's fixed-hierarchy throughput), never in a solve of such a system, which takes one
// or two cycles -- so the solvers do not attach it themselves.
static bool amg_attach_poly2(ipd_amg* h) {
    ipd_ctx* ctx = h->ctx;
    CycleState* st = state_of(h);
    if (!st || !st->res_ok || st->resb || st->res_poly2 || st->res_remote || st->res_ke3 != 0 || st->res_ke != 16) return false;
    const ResDesc& R = st->res_desc;
    const int N2 = R.L2.N, nf = R.L1.nf;
    if (h->J != 3 || R.Nt != 1 || R.three || h->opts.cycle != 'v' || h->opts.smoth < 1 || N2 > RES_NMAX / 2 ||
        nf > RES_NMAX / 2 || N2 > 64 * 16)
        return false;
    const int seg = RES_NMAX / 2, ld = 2 * seg + 128;
    const BPolyDev pb = pack_bpoly(ctx, h, st, 2, h->opts.isnsp, 0, true, seg, ld);
    const int Np = pb.e.Np, nT = (Np / 16) * (Np / 16);
    // (IPD_OPTIN_LDS is not needed: the tiles use static LDS only)
    hipLaunchKernelGGL(k_bpoly_compose, dim3((unsigned)(nT + (N2 + 3) / 4)), dim3(256), 0, ctx->stream, pb.e, nT);
    IPD_KERNEL_CHECK();
    st->res_desc.p2rows = pb.M;
    st->res_desc.p2w = pb.W;
    st->res_desc.p2seg = seg;
    st->res_desc.p2ld = ld;
    st->res_poly2 = true;
    st->res_capacity = -1;
    st->level_forms.resize((size_t)h->J + 1, 0);
    st->level_forms[2] |= 128;
    ctx->sync();   // the pack's scratch operands die with the call scope
    return true;
}

extern "C" int ipd_amg_attach_level2_poly(ipd_amg* h, int32_t* attached) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        h->ctx->set_device();
        CallScope scope(h->ctx);
        const bool ok = amg_attach_poly2(h);
        if (attached) *attached = ok ? 1 : 0;
    });
}

extern "C" int ipd_amg_attach_mask_operator(ipd_amg* h, const double* p_dev, const double* q_dev,
                                            int64_t m, int64_t n, double tk, int32_t* attached) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && p_dev && q_dev && m > 0 && n > 0, IPD_E_ARG, "bad argument");
        h->ctx->set_device();
        CallScope scope(h->ctx);
        const bool ok = amg_attach_maskop(h, p_dev, q_dev, (int)m, (int)n, tk, false, false);
        if (attached) *attached = ok ? 1 : 0;
    });
}

extern "C" int ipd_amg_attach_mask_transfers(ipd_amg* h, const double* p_dev, const double* q_dev,
                                             int64_t m, int64_t n, double tk, int32_t* attached) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && p_dev && q_dev && m > 0 && n > 0, IPD_E_ARG, "bad argument");
        h->ctx->set_device();
        CallScope scope(h->ctx);
        const bool ok = amg_attach_maskop(h, p_dev, q_dev, (int)m, (int)n, tk, false, true);
        if (attached) *attached = ok ? 1 : 0;
    });
}

// Solves A_k e = r_k approximately; r in L[k].r, result in L[k].e.
// keep_e: start from the current L[k].e (second leg of a W cycle); otherwise the
// start is e = 0, which is never materialised (the first sweep does not read it).
void amg_cycle(ipd_amg* h, int k, int isnsp, bool wcycle, bool keep_e) {
    ipd_ctx* ctx = h->ctx;
    CycleState* st = state_of(h);
    IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
    Level& lv = h->L[k];
    LevelRun& rn = st->run[(size_t)k];
    const int cu = st->num_cu;
    if (st->k_sub == k) {  // everything from here down: one workgroup, LDS-resident (replicated)
        flush_fused(ctx, st);
        hipLaunchKernelGGL(k_subcycle, dim3(1), dim3(BT), st->sub_lds, ctx->stream,
                           (const SolveDesc*)st->d_sub, keep_e ? 1 : 0);
        IPD_KERNEL_CHECK();
        rn.e_zero = false;
        return;
    }
    if (k == h->J) {                                   // MG_Vcycle.m:43 / MG_Wcycle.m:44
        PcgArgs a = rn.pcg;                            // replicated on every rank
        a.rhs = lv.r;
        a.d = lv.e;
        if (st->fuse_enabled) {
            push_phase(ctx, st, PH_PCG, 0).u.p = a;
        } else {
            hipLaunchKernelGGL(k_pcg, dim3(1), dim3(BT), 0, ctx->stream, a);
            IPD_KERNEL_CHECK();
        }
        return;
    }
    const int nu = h->opts.smoth;
    if (!keep_e) {
        rn.e_zero = true;
        if (nu == 0) {  // no sweep will overwrite the iterate: materialise the zero
            flush_fused(ctx, st);
            IPD_HIP(hipMemsetAsync(lv.e, 0, sizeof(double) * (size_t)lv.N, ctx->stream));
            rn.e_zero = false;
        }
    }
    for (int s = 0; s < nu; ++s) launch_sweep(h, st, k, isnsp, false);          // :14-25
    const bool resid_small = rn.staged && phase_is_small(st, lv.N, rn.dev.L, (double)lv.A.nnz, lv.N);
    XferArgs ra = rn.restrict_args;
    const bool rest_small =
        ra.staged && phase_is_small(st, ra.nrows, ra.L, (double)h->L[k + 1].Pt.nnz, ra.ncols);
    const char* nrrc = std::getenv("IPD_NO_RRC");
    const bool no_rrc = nrrc && nrrc[0] == '1';
    const Csr& T1 = h->L[k + 1].T1;
    // Fused where the two launches are latency-bound (measured: tree-mask W cycle 0.432 -> 0.413 ms,
    // realistic Newton systems -2...-3.5 %); once T1 is megabytes the pair is bandwidth-bound and the
    // fused walk (CSR T1, 12 B per entry, against the padded A, 10 B) is the slower one (regime D at
    // m=n=2048: 0.321 -> 0.342 ms), so large T1 keep the two launches.
    if (!no_rrc && !resid_small && !rest_small && T1.rp && T1.nr == ra.nrows && T1.nnz <= (1 << 18)) {
        // r_{k+1} = P'r - (P'A) e: one launch instead of residual + restriction           :27
        RrcArgs rc;
        rc.p = ra;
        rc.p.x = lv.r;
        rc.p.L = pick_lanes(T1.nnz + h->L[k + 1].Pt.nnz, ra.nrows, cu);
        rc.rp2 = T1.rp;
        rc.ci2 = T1.ci;
        rc.va2 = T1.va;
        rc.e = lv.e;
        const bool staged = 2 * (size_t)ra.ncols <= (size_t)STAGE_MAX && rn.staged;
        const size_t dyn = staged ? 2 * sizeof(double) * (size_t)ra.ncols : 0;
        run_rows(ctx, st, 0, ra.nrows,
                 [&](int r0, int r1) {
                     rc.p.row0 = r0;
                     rc.p.row1 = r1;
                     const int grid = pick_blocks(r1 - r0, rc.p.L, cu);
                     if (staged)
                         hipLaunchKernelGGL(k_rrc<true>, dim3(grid), dim3(BT), dyn, ctx->stream, rc);
                     else
                         hipLaunchKernelGGL(k_rrc<false>, dim3(grid), dim3(BT), 0, ctx->stream, rc);
                     IPD_KERNEL_CHECK();
                 },
                 {ra.y});
    } else {
        if (resid_small) {                                                              // :27
            ResidDesc& rd = push_phase(ctx, st, PH_RESID, lv.N).u.r;
            rd.lv = rn.dev;
            rd.e = lv.e;
            rd.row0 = 0;
            rd.row1 = lv.N;
        } else {
            run_rows(ctx, st, 0, lv.N,
                     [&](int r0, int r1) { launch_resid(ctx, rn, lv.e, r0, r1, cu); }, {lv.rr});
        }
        if (rest_small)
            push_phase(ctx, st, PH_XFER, ra.ncols).u.x = ra;
        else
            run_rows(ctx, st, 0, ra.nrows,
                     [&](int r0, int r1) {
                         ra.row0 = r0;
                         ra.row1 = r1;
                         launch_xfer(ctx, ra, cu);
                     },
                     {ra.y});
    }
    amg_cycle(h, k + 1, isnsp, wcycle, false);                                   // :29
    // MG_Wcycle.m:30 -- the second correction; on the coarsest level it repeats the
    // identical zero-guess PCG solve, so it is skipped there (same bits).
    if (wcycle && k + 1 < h->J) amg_cycle(h, k + 1, isnsp, wcycle, true);
    {
        XferArgs pa = rn.prolong_args;                                           // :31
        pa.x = h->L[k + 1].e;
        pa.y = lv.e;
        if (pa.staged && phase_is_small(st, pa.nrows, pa.L, (double)h->L[k + 1].P.nnz, pa.ncols))
            push_phase(ctx, st, PH_XFER, pa.ncols).u.x = pa;
        else
        run_rows(ctx, st, 0, pa.nrows,
                 [&](int r0, int r1) {
                     pa.row0 = r0;
                     pa.row1 = r1;
                     launch_xfer(ctx, pa, cu);
                 },
                 {pa.y});
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
    a.staged = rn.staged;
    const size_t dyn = a.staged ? sizeof(double) * (size_t)rn.dev.N : 0;
    if (a.staged && phase_is_small(st, rn.dev.N, rn.dev.L, (double)h->L[1].A.nnz, rn.dev.N)) {
        a.row0 = 0;
        a.row1 = rn.dev.N;
        push_phase(ctx, st, PH_TOP, rn.dev.N).u.t = a;
    } else {
        run_rows(ctx, st, 0, rn.dev.N,
                 [&](int r0, int r1) {
                     a.row0 = r0;
                     a.row1 = r1;
                     const int grid = pick_blocks(r1 - r0, rn.dev.L, st->num_cu);
                     IPD_LAUNCH_SP(k_top, a.staged, rn.dev.S > 0, grid, dyn, a);
                 },
                 {rn.dev.r, xnew});
    }
    ConvArgs ca;
    ca.r = rn.dev.r;
    ca.n = rn.dev.N;
    ca.hist = st->hist;
    ca.first = first ? 1 : 0;
    if (st->fuse_enabled) {
        push_phase(ctx, st, PH_CONV, 0).u.c = ca;
    } else {
        hipLaunchKernelGGL(k_conv, dim3(1), dim3(BT), 0, ctx->stream, ca);
        IPD_KERNEL_CHECK();
    }
    flush_fused(ctx, st);  // the loop body ends here: nothing stays queued across calls
}

// one Class_AMG loop body (Class_AMG.m:96-105): x_out = x_in + cycle(b - A x_in)
static void enqueue_loop_body(ipd_amg* h, CycleState* st, const double* b, const double* xin,
                              double* xout) {
    const bool wc = h->opts.cycle == 'w', vc = h->opts.cycle == 'v';
    const double* ecorr = nullptr;
    if (vc || wc) {
        amg_cycle(h, 1, h->opts.isnsp, wc, false);
        ecorr = h->L[1].e;
    }
    launch_top(h, st, b, xin, ecorr, xout, false);
}

static void ensure_graphs(ipd_amg* h, CycleState* st, const double* b_dev);

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
    if (st->small_ok && st->shard_ranks == 1) {
        // small hierarchy: the whole solve phase is one single-workgroup launch
        if (st->solve_cached)
            hipLaunchKernelGGL(k_solve_small<true>, dim3(1), dim3(BT), st->solve_lds, ctx->stream,
                               (const SolveDesc*)st->d_solve, b_dev, xa, xb, st->hist,
                               st->solve_out, 0);
        else
            hipLaunchKernelGGL(k_solve_small<false>, dim3(1), dim3(BT), st->solve_lds, ctx->stream,
                               (const SolveDesc*)st->d_solve, b_dev, xa, xb, st->hist,
                               st->solve_out, 0);
        IPD_KERNEL_CHECK();
        const size_t nout = 4 + 2 * ((size_t)o.maxit + 2);
        std::vector<double> out(nout);
        ctx->fetch(st->solve_out, out.data(), nout);
        const int its = (int)out[0];
        if (rel_resk) std::memcpy(rel_resk, out.data() + 4, sizeof(double) * ((size_t)its + 1));
        if (rhok) std::memcpy(rhok, out.data() + 4 + (o.maxit + 2), sizeof(double) * ((size_t)its + 1));
        if (x_dev)
            IPD_HIP(hipMemcpyAsync(x_dev, xa, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        if (it_out) *it_out = its;
        if (rel_res_out) *rel_res_out = out[1];
        ctx->sync();
        return;
    }
    if (st->res_ok && st->shard_ranks == 1) {
        // dense regime: the whole solve phase is one launch of co-resident workgroups
        std::vector<double> out;
        if (run_resident(h, st, b_dev, xa, 0, &out, nullptr)) {
            const int its = (int)out[0];
            if (rel_resk) std::memcpy(rel_resk, out.data() + 4, sizeof(double) * ((size_t)its + 1));
            if (rhok) std::memcpy(rhok, out.data() + 4 + (o.maxit + 2), sizeof(double) * ((size_t)its + 1));
            if (x_dev)
                IPD_HIP(hipMemcpyAsync(x_dev, xa, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                       ctx->stream));
            if (it_out) *it_out = its;
            if (rel_res_out) *rel_res_out = out[1];
            ctx->sync();
            return;
        }
        // not usable right now: restore the initial guess and take the multi-launch path
        if (guess_dev)
            IPD_HIP(hipMemcpyAsync(xa, guess_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                   ctx->stream));
        else
            IPD_HIP(hipMemsetAsync(xa, 0, sizeof(double) * (size_t)N, ctx->stream));
    }
    launch_top(h, st, b_dev, xa, nullptr, xb, true);                            // :89
    std::swap(xa, xb);
    double hh[5];
    ctx->fetch(st->hist, hh, 5);
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
            enqueue_loop_body(h, st, b_dev, xa, xb);                             // :96-105
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
    bool keep = false;
    if (e_in && wc) {  // MG_Wcycle(r,isnsp,k,e): start from the caller's iterate
        ctx->upload(lv.e, e_in, N);
        rn.e_zero = false;
        keep = true;
    }
    amg_cycle(h, k, isnsp, wc, keep);
    flush_fused(ctx, st);
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
        long long nf = 0;
        if (o) {
            if (o->retol >= 0) tol = o->retol;
            if (o->maxit >= 0) maxit = o->maxit;
            if (o->precd >= 0) precd = o->precd;
            nf = o->nf;
        }
        Csr hm;
        csr_upload_from_csc(ctx, tmp, H, false, &hm);  // true rows of H
        const size_t N = (size_t)hm.nr;
        double* de = tmp.alloc<double>(N);
        double* dd = tmp.alloc<double>(N);
        double* dg = nullptr;
        ctx->upload(de, e, N);
        if (guess) {
            dg = tmp.alloc<double>(N);
            ctx->upload(dg, guess, N);
        }
        long long its = 0;
        pcg_dev(ctx, hm, de, dg, tol, maxit, precd, dd, &its, res, resk, nf);
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

// Mode 2 only: how many levels the resident workgroups keep in registers (2 or 3) and the level
// the tail is rooted at (3: the local tail of a three-level hierarchy or the remote tail workgroup's
// sub-cycle root; 4: remote tail below a resident level 3); zeros otherwise.
extern "C" int ipd_amg_resident_levels(const ipd_amg* h, int32_t* levels, int32_t* tail_root) {
    return ipd_guard([&] {
        IPD_REQUIRE(h, IPD_E_ARG, "NULL handle");
        const CycleState* st = h->cyc.get();
        const bool on = st && st->res_ok;
        if (levels) *levels = on ? (st->resb_poly4 ? 4 : st->res_ke3 > 0 ? 3 : 2) : 0;
        if (tail_root) *tail_root = on ? (st->resb_poly4 ? 5 : st->res_ke3 > 0 ? 4 : 3) : 0;
    });
}

// Which resident kernel this hierarchy's solve phase launches (mode 2 of ipd_amg_solve_mode) -- the
// instantiation's name as it appears in a rocprofv3 kernel trace, "" otherwise -- and what its last
// launch did: chip-wide hand-offs (tagged-granule exchanges, plus visits of the remote tail) and loop
// bodies.  bench.py derives hand-offs per cycle from these instead of re-deriving the kernel from sizes.
extern "C" int ipd_amg_resident_kernel(const ipd_amg* h, char* name, int32_t cap, int64_t* handoffs,
                                       int32_t* cycles, int32_t* mask_transfers) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && name && cap > 0, IPD_E_ARG, "bad argument");
        const CycleState* st = h->cyc.get();
        char buf[64] = "";
        if (st && st->res_ok) {
            if (st->resb)
                std::snprintf(buf, sizeof buf, "k_resident_big<%d,%d,%s>", st->resb_ke2, st->resb_deep ? 2 : 1,
                              st->resb_deep ? "true" : "false");
            else
                std::snprintf(buf, sizeof buf, st->res_poly2 ? "k_resident<%d,%d,%d,true>" : "k_resident<%d,%d,%d>",
                              st->res_ke, st->res_ke, st->res_ke3);
        }
        std::snprintf(name, (size_t)cap, "%s", buf);
        if (handoffs) *handoffs = st ? st->res_last_handoffs : 0;
        if (cycles) *cycles = st ? st->res_last_cycles : 0;
        // level 1 <-> 2 transfers from the bit mask: always in the mask-form kernel, ResDesc::xm otherwise
        if (mask_transfers) *mask_transfers = (st && st->res_ok && (st->resb || st->res_desc.xm)) ? 1 : 0;
    });
}

// How the levels held in the LDS images of this hierarchy run (bit mask over all images packed):
// 1 thread-per-row sweeps, 2 the same with dense rows in registers, 4 one-wave sweeps, 8 one-wave
// polynomial form, 16 block-wide polynomial form; 0: the level is in no image.
extern "C" int ipd_amg_level_forms(const ipd_amg* h, int32_t* forms, int32_t count) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && forms && count >= 0, IPD_E_ARG, "bad argument");
        const CycleState* st = h->cyc.get();
        for (int k = 0; k < count; ++k)
            forms[k] = (st && (size_t)k < st->level_forms.size()) ? st->level_forms[(size_t)k] : 0;
    });
}

// Test hook: the block-wide polynomial operator of level k as packed for the images, column-major with
// *ld rows: columns [Mr (N8) | Me (N8) | Mc (Nc8)] then the column W (N8 = N rounded up to 8); needs
// ld * (2 N8 + Nc8 + 1) doubles.  IPD_E_ARG when level k has no such operator.
extern "C" int ipd_amg_poly_operator(const ipd_amg* h, int32_t k, double* out, int64_t cap, int32_t* ld,
                                     int32_t* n, int32_t* nc) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && out && ld && n && nc, IPD_E_ARG, "NULL argument");
        const CycleState* st = h->cyc.get();
        IPD_REQUIRE(st && k >= 1 && (size_t)k < st->poly_ops.size() && st->poly_ops[(size_t)k].M, IPD_E_ARG,
                    "level has no block-wide polynomial operator");
        const CycleState::PolyOp& po = st->poly_ops[(size_t)k];
        const int64_t N8 = (po.N + 7) / 8 * 8, Nc8 = (po.Nc + 7) / 8 * 8;
        const int64_t need = (int64_t)po.LD * (2 * N8 + Nc8 + 1);
        IPD_REQUIRE(cap >= need, IPD_E_ARG, "buffer too small");
        h->ctx->fetch(po.M, out, (size_t)need);   // W lies right behind M (pack_bpoly)
        *ld = po.LD;
        *n = po.N;
        *nc = po.Nc;
    });
}

extern "C" int ipd_amg_solve_mode(const ipd_amg* h, int32_t* mode, int32_t* grid, int32_t* timeouts) {
    if (!h || !mode) return IPD_E_ARG;
    const CycleState* st = h->cyc.get();
    if (!st) return IPD_E_ARG;
    *mode = st->small_ok ? 1 : (st->res_ok ? 2 : 0);
    if (grid) *grid = st->res_ok ? st->res_G + (st->res_remote ? 1 : 0) : (st->small_ok ? 1 : 0);
    if (timeouts) *timeouts = st->res_timeouts;
    return IPD_OK;
}

extern "C" int ipd_amg_bench_resident(ipd_amg* h, const double* b_dev, double* x_dev, int cycles,
                                      double* total_ms, int64_t stamps[10]) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && b_dev && x_dev && cycles > 0 && total_ms && stamps, IPD_E_ARG, "bad argument");
        ipd_ctx* ctx = h->ctx;
        CallScope scope(ctx);
        CycleState* st = state_of(h);
        IPD_REQUIRE(st && st->res_ok, IPD_E_ARG, "hierarchy does not run in resident mode");
        const int N = h->L[1].A.nr;
        long long* dbg = ctx->scratch->alloc<long long>(16);
        IPD_HIP(hipMemsetAsync(dbg, 0, 128, ctx->stream));
        IPD_HIP(hipMemcpyAsync(h->x, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        float msf = 0.f;
        IPD_REQUIRE(run_resident(h, st, b_dev, h->x, cycles, nullptr, &msf, dbg), IPD_E_HIP,
                    "resident kernel gave up (not every workgroup was resident)");
        IPD_HIP(hipMemcpyAsync(x_dev, h->x, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        long long hs[10];
        ctx->fetch(dbg, hs, 10);
        for (int i = 0; i < 10; ++i) stamps[i] = hs[i];
        *total_ms = msf;
    });
}

// Captures the two loop bodies (x -> x2 and x2 -> x) as HIP graphs: one graph launch
// per cycle instead of ~40 kernel launches, so the host never paces the device.
static void ensure_graphs(ipd_amg* h, CycleState* st, const double* b_dev) {
    if (st->gexec[0] && st->gb == b_dev) return;
    ipd_ctx* ctx = h->ctx;
    for (auto& g : st->gexec)
        if (g) {
            IPD_HIP(hipGraphExecDestroy(g));
            g = nullptr;
        }
    double* xs[2] = {h->x, st->x2};
    for (int v = 0; v < 2; ++v) {
        hipGraph_t graph = nullptr;
        IPD_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        try {
            enqueue_loop_body(h, st, b_dev, xs[v], xs[v ^ 1]);
        } catch (...) {
            (void)hipStreamEndCapture(ctx->stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        IPD_HIP(hipStreamEndCapture(ctx->stream, &graph));
        hipError_t e = hipGraphInstantiate(&st->gexec[v], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        IPD_HIP(e);
    }
    st->gb = b_dev;
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
        if (st->small_ok) {  // one launch runs all the cycles (no stopping rules)
            IPD_HIP(hipMemcpyAsync(h->x, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                   ctx->stream));
            hipEvent_t e0, e1;
            IPD_HIP(hipEventCreate(&e0));
            IPD_HIP(hipEventCreate(&e1));
            IPD_HIP(hipEventRecord(e0, ctx->stream));
            if (st->solve_cached)
                hipLaunchKernelGGL(k_solve_small<true>, dim3(1), dim3(BT), st->solve_lds,
                                   ctx->stream, (const SolveDesc*)st->d_solve, b_dev, h->x, st->x2,
                                   st->hist, st->solve_out, cycles);
            else
                hipLaunchKernelGGL(k_solve_small<false>, dim3(1), dim3(BT), st->solve_lds,
                                   ctx->stream, (const SolveDesc*)st->d_solve, b_dev, h->x, st->x2,
                                   st->hist, st->solve_out, cycles);
            IPD_KERNEL_CHECK();
            IPD_HIP(hipEventRecord(e1, ctx->stream));
            IPD_HIP(hipEventSynchronize(e1));
            float msf = 0.f;
            IPD_HIP(hipEventElapsedTime(&msf, e0, e1));
            IPD_HIP(hipEventDestroy(e0));
            IPD_HIP(hipEventDestroy(e1));
            IPD_HIP(hipMemcpyAsync(x_dev, h->x, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                   ctx->stream));
            ctx->sync();
            *total_ms = msf;
            if (bytes_per_cycle) *bytes_per_cycle = cycle_bytes(h);
            return;
        }
        if (st->res_ok) {  // one launch of co-resident workgroups runs all the cycles
            IPD_HIP(hipMemcpyAsync(h->x, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                   ctx->stream));
            float msf = 0.f;
            if (run_resident(h, st, b_dev, h->x, cycles, nullptr, &msf)) {
                IPD_HIP(hipMemcpyAsync(x_dev, h->x, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                                       ctx->stream));
                ctx->sync();
                *total_ms = msf;
                if (bytes_per_cycle) *bytes_per_cycle = cycle_bytes(h);
                return;
            }
        }
        const char* ng = std::getenv("IPD_NO_GRAPH");
        const bool use_graph = !(ng && ng[0] == '1');
        IPD_HIP(hipMemcpyAsync(h->x, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        // initial residual (Class_AMG.m:89); x stays in h->x
        launch_top(h, st, b_dev, h->x, nullptr, st->x2, true);
        IPD_HIP(hipMemcpyAsync(h->x, st->x2, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        if (use_graph) ensure_graphs(h, st, b_dev);
        double* xs[2] = {h->x, st->x2};
        hipEvent_t ev0, ev1;
        IPD_HIP(hipEventCreate(&ev0));
        IPD_HIP(hipEventCreate(&ev1));
        IPD_HIP(hipEventRecord(ev0, ctx->stream));
        int v = 0;
        for (int c = 0; c < cycles; ++c) {
            if (use_graph)
                IPD_HIP(hipGraphLaunch(st->gexec[v], ctx->stream));
            else
                enqueue_loop_body(h, st, b_dev, xs[v], xs[v ^ 1]);
            v ^= 1;
        }
        IPD_HIP(hipEventRecord(ev1, ctx->stream));
        IPD_HIP(hipEventSynchronize(ev1));
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        IPD_HIP(hipEventDestroy(ev0));
        IPD_HIP(hipEventDestroy(ev1));
        IPD_HIP(hipMemcpyAsync(x_dev, xs[v], sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        ctx->sync();
        *total_ms = ms;
        if (bytes_per_cycle) *bytes_per_cycle = cycle_bytes(h);
    });
}

// Times `reps` smoother sweeps of level k (pre-smoothing direction) with HIP events on
// the context's stream: the per-launch duration of the dominant kernel (k_smooth).
// launches_per_sweep = 2 for the bigraph Gauss-Seidel level, 1 for Jacobi levels;
// bytes_per_sweep = S(A_k) + 6*8*N_k (fused-GS form).
extern "C" int ipd_amg_bench_subcycle(ipd_amg* h, int reps, double* total_ms, int32_t* k_sub,
                                      int64_t stamps[8]) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && reps > 0 && total_ms, IPD_E_ARG, "bad argument");
        ipd_ctx* ctx = h->ctx;
        ctx->set_device();
        CycleState* st = state_of(h);
        IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
        if (k_sub) *k_sub = st->k_sub;
        *total_ms = 0.0;
        if (!st->k_sub) return;
        CallScope scope(ctx);
        long long* dbg = ctx->scratch->alloc<long long>(16);
        IPD_HIP(hipMemsetAsync(dbg, 0, 128, ctx->stream));
        // patch the debug pointer into the image header
        const size_t off = offsetof(SolveDesc, dbg);
        // (a stamp is two s_memrealtime reads and a read-modify-write of global memory, ~0.5 us each: the
        // per-stage figures are for proportions)
        ctx->upload_bytes(reinterpret_cast<char*>(st->d_sub) + off, &dbg, sizeof(dbg));
        {   // a right-hand side that is not zero (a zero one ends every coarse PCG at once)
            std::vector<double> rr((size_t)h->L[st->k_sub].N);
            unsigned lcg = 12345u;
            for (auto& v : rr) {
                lcg = lcg * 1664525u + 1013904223u;
                v = (double)(lcg >> 8) / (double)(1u << 24) - 0.5;
            }
            ctx->upload(h->L[st->k_sub].r, rr.data(), rr.size());
        }
        hipEvent_t e0, e1;
        IPD_HIP(hipEventCreate(&e0));
        IPD_HIP(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_subcycle, dim3(1), dim3(BT), st->sub_lds, ctx->stream,
                           (const SolveDesc*)st->d_sub, 0);
        IPD_HIP(hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_subcycle, dim3(1), dim3(BT), st->sub_lds, ctx->stream,
                               (const SolveDesc*)st->d_sub, 0);
        IPD_HIP(hipEventRecord(e1, ctx->stream));
        IPD_HIP(hipEventSynchronize(e1));
        IPD_KERNEL_CHECK();
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, e0, e1));
        IPD_HIP(hipEventDestroy(e0));
        IPD_HIP(hipEventDestroy(e1));
        *total_ms = ms;
        long long hs[16];
        ctx->fetch(dbg, hs, 16);
        if (stamps && hs[3] > hs[2])   // shader clock (MHz) seen by the cycle: s_memtime ticks / 10 ns
            stamps[0] = hs[8] * 100 / (hs[3] - hs[2]), hs[0] = stamps[0];
        if (stamps)
            for (int i = 0; i < 8; ++i) stamps[i] = hs[i];
        long long* none = nullptr;
        ctx->upload_bytes(reinterpret_cast<char*>(st->d_sub) + off, &none, sizeof(none));
    });
}

extern "C" int ipd_amg_bench_sweeps(ipd_amg* h, int k, int reps, double* total_ms,
                                    int* launches_per_sweep, double* bytes_per_sweep) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && total_ms && reps > 0, IPD_E_ARG, "bad argument");
        IPD_REQUIRE(k >= 1 && k < h->J, IPD_E_ARG, "level must be a smoothed level (1 <= k < J)");
        ipd_ctx* ctx = h->ctx;
        CallScope scope(ctx);
        CycleState* st = state_of(h);
        Level& lv = h->L[k];
        LevelRun& rn = st->run[(size_t)k];
        fill_f64(ctx, lv.r, 1.0, (size_t)lv.N);
        rn.e_zero = true;
        for (int w = 0; w < 4; ++w) launch_sweep(h, st, k, h->opts.isnsp, false);
        flush_fused(ctx, st);
        hipEvent_t ev0, ev1;
        IPD_HIP(hipEventCreate(&ev0));
        IPD_HIP(hipEventCreate(&ev1));
        IPD_HIP(hipEventRecord(ev0, ctx->stream));
        for (int s = 0; s < reps; ++s) launch_sweep(h, st, k, h->opts.isnsp, false);
        flush_fused(ctx, st);
        IPD_HIP(hipEventRecord(ev1, ctx->stream));
        IPD_HIP(hipEventSynchronize(ev1));
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        IPD_HIP(hipEventDestroy(ev0));
        IPD_HIP(hipEventDestroy(ev1));
        *total_ms = ms;
        if (launches_per_sweep) *launches_per_sweep = lv.nf > 0 ? 2 : 1;
        if (bytes_per_sweep) *bytes_per_sweep = spmv_bytes(lv.A) + 6 * 8.0 * lv.A.nr;
    });
}

// Row-block sharded loop body (eager launches; RCCL calls are not graph-captured).
extern "C" int ipd_amg_bench_cycles_sharded(ipd_amg* h, const double* b_dev, double* x_dev,
                                            int cycles, double* total_ms,
                                            double* bytes_per_cycle) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && b_dev && x_dev && cycles > 0 && total_ms, IPD_E_ARG, "bad argument");
        ipd_ctx* ctx = h->ctx;
        CallScope scope(ctx);
        CycleState* st = state_of(h);
        IPD_REQUIRE(st, IPD_E_ARG, "hierarchy has no cycle state");
        const char* emu = std::getenv("IPD_SHARD_EMULATE");
        const int emu_ranks = emu ? std::atoi(emu) : 0;
        struct Restore {
            CycleState* st;
            ~Restore() {
                st->shard_ranks = 1;
                st->shard_rank = 0;
                st->shard_emulate = false;
            }
        } restore{st};
        if (emu_ranks > 1) {
            st->shard_ranks = emu_ranks;
            st->shard_emulate = true;
        } else {
            st->shard_ranks = comm_size(ctx);
            st->shard_rank = comm_rank(ctx);
            IPD_REQUIRE(st->shard_ranks == 1 || ctx->comm, IPD_E_COMM, "call ipd_comm_init first");
        }
        const int N = h->L[1].A.nr;
        IPD_HIP(hipMemcpyAsync(h->x, x_dev, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        launch_top(h, st, b_dev, h->x, nullptr, st->x2, true);
        IPD_HIP(hipMemcpyAsync(h->x, st->x2, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        double* xs[2] = {h->x, st->x2};
        hipEvent_t ev0, ev1;
        IPD_HIP(hipEventCreate(&ev0));
        IPD_HIP(hipEventCreate(&ev1));
        IPD_HIP(hipEventRecord(ev0, ctx->stream));
        int v = 0;
        for (int c = 0; c < cycles; ++c) {
            enqueue_loop_body(h, st, b_dev, xs[v], xs[v ^ 1]);
            v ^= 1;
        }
        IPD_HIP(hipEventRecord(ev1, ctx->stream));
        IPD_HIP(hipEventSynchronize(ev1));
        float ms = 0.f;
        IPD_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        IPD_HIP(hipEventDestroy(ev0));
        IPD_HIP(hipEventDestroy(ev1));
        IPD_HIP(hipMemcpyAsync(x_dev, xs[v], sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice,
                               ctx->stream));
        ctx->sync();
        *total_ms = ms;
        if (bytes_per_cycle) *bytes_per_cycle = cycle_bytes(h);
    });
}
