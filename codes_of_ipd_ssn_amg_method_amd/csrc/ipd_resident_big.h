// Level-resident solve kernel for level 1 of up to 4096 rows (included by ipd_cycle.hip only).
//
// k_resident (ipd_resident.h) keeps the FULL vectors of levels 1-2 in every workgroup's LDS and the
// rows of both levels in registers: at m = n = 2048 (BASELINE config 4's size: M = 4096, regime D) the
// vectors alone would be 9 x 32 KB and a level-1 row 2049 entries.  Here
//   * level 1 is the bit mask (amg_attach_maskop verified A_1 = Hybrid_AMG's rescaled operator entry by
//     entry): A(j, nf+i) = -alpha_j beta_i s_ij, so a row is 32 BITS per lane (one register) against a
//     pre-scaled LDS vector; the transfers are the mask too (W(j,i) = s_ij beta_i rho_j);
//   * level 2 (up to 2048 x 2048, dense in regime D: 33 MB) sits in registers, 32 entries per lane;
//   * LDS holds only what rows GATHER from: the scaled iterate of level 1 and the scaled x (32 KB each),
//     the iterate of level 2, one staging vector (rho.*rr for the restriction, beta.*e_2 for the
//     prolongation) and the one-column transfer to the one-row tail: 112 KB.  Right-hand sides,
//     residuals and the unscaled iterate live as the owning wave's scalars;
//   * sums over all rows (1'r, 1'(r - A e) = 1'r - (A1)'e, the norms, the tail's restriction) are taken
//     from the hand-off itself: every thread holds (A1)_j of the granules it receives, or -- where only
//     a sum is needed -- the workgroups exchange their partial sums (G or 2G granules instead of N).
// Protocol (tagged 16-byte granules, two buffers by step parity, bounded spins, give-up word), the
// one-row tail's PCG and the stationary iteration with its stopping rules are k_resident's.
//
// Round 4 -- REALISTIC hierarchies at M = 4096 (template argument DEEP; the Newton systems of the m = n =
// 2048 driver run: levels about 4096 / 2048 / 640 / 190 / 55 / 15, 12 k / 8 k / 3 k entries).  Level 1 and
// the level 1 <-> 2 transfers stay the bit mask (a row is one register whatever its population -- the
// hub rows of a dense early mask and the three-entry rows of a late one cost the same 32 LDS gathers),
// level 2 is short CSR slices in registers (KE2 = 4 or 8), level 3 (up to 1024 rows) runs in polynomial
// form exactly as k_resident's POLY3 level (rows [M2a | M1] and the stacked restriction row in registers,
// two entries per thread and segment: three hand-offs per visit), and everything from level 4 down is
// the remote tail workgroup (res_tail_workgroup: k_subcycle's LDS image rooted at level 4).  The tail
// workgroup needs a compute unit of its own, so the rows are dealt to G <= 255 workgroups: a wave then
// owns RPW = 2 rows of each block (rows w and w + 8 of the workgroup's run).
// Reference: AMG/Class_AMG.m:86-109, AMG/MG_Vcycle.m:12-45, AMG/MG_Wcycle.m:13-46, PCG.m:68-87.
#pragma once

static constexpr int RB_NMAX = 8 * BT;                 // rows of level 1
static constexpr int RB_HALF = 4 * BT;                 // rows of a block of level 1 / of level 2
static constexpr int RB_GRAN = RB_NMAX;                // granules per hand-off buffer
static constexpr int RB_RPW_MAX = 2;                   // rows of a block per wave
static constexpr int RB_N3MAX = 2 * BT;                // rows of the polynomial level 3 (DEEP)
static constexpr int RB_N4MAX = BT / 2;                // rows of the remote tail's root level (DEEP)
static constexpr int RB_P3_SEG = RB_N3MAX;             // p3rows layout: [Mr (RB_P3_SEG) | Me (RB_P3_SEG) | Mc (RB_N4MAX)]
static constexpr int RB_P3_LD = 2 * RB_P3_SEG + RB_N4MAX;
static constexpr int RB_N5MAX = BT / 4;                // rows of the tail's root level when level 4 is resident too (POLY4)
static constexpr int RB_P4_SEG = RB_N4MAX;             // p4rows layout: [Mr (RB_P4_SEG) | Me (RB_P4_SEG) | Mc (RB_N5MAX)]
static constexpr int RB_P4_LD = 2 * RB_P4_SEG + RB_N5MAX;
// LDS (doubles): E2, TU, P3C / RR2, E1S, XS, reductions, publish slots, own-row constants, fail word;
// DEEP: R3, E3 (RB_N3MAX each), E4, R4 (RB_N4MAX each), E5 (RB_N5MAX), 128 partial sums of the polynomial passes
static constexpr size_t RB_LDS_DOUBLES = (size_t)2 * RB_NMAX + 3 * RB_HALF + 2 * RES_WAVES +
                                         2 * 8 * RB_RPW_MAX + 20 * 8 * RB_RPW_MAX + 8 +
                                         2 * RB_N3MAX + 2 * RB_N4MAX + RB_N5MAX + 128;
static constexpr size_t RB_LDS_BYTES = sizeof(double) * RB_LDS_DOUBLES;

struct ResBigDesc {
    int nf, nc, N2, S2;
    const unsigned short* pci2;   // level 2: padded rows (stride S2), 16-bit columns
    const double* pva2;
    const double *diag2, *dinv2, *Axi2, *xx2;
    const double *diag1, *dinv1, *Axi1, *xx1;
    const unsigned long long* fbits;   // [nf][nwf] over the C nodes
    const unsigned long long* cbits;   // [nc][nwc] over the F rows
    int nwf, nwc;
    const double* alpha;   // nf
    const double* beta;    // nc
    const double* rho;     // nf
    ResCsr P3, A3;         // level 2 -> the one-row tail, the tail's 1 x 1 operator
    int nu, isnsp, wcycle, anycycle, maxit;
    double retol;
    long long pcg_maxit;
    unsigned char* gran;   // 2 x RB_GRAN granules
    unsigned* tmo;
    int presleep, pollsleep;
    unsigned dbg_skip_seq;
    // Rank groups (IPD_RESIDENT_RANKS=R, three-level mode): the workgroups are split into R
    // contiguous groups, each with a granule buffer of its OWN (R copies behind one another); a publish writes
    // the granule into every group's copy and a sweep reads only its own group's.  Nothing else is shared
    // between workgroups, so this is the data path of a row-block sharded run whose groups sit on R GPUs with
    // peer-mapped buffers over xGMI (a publish = one write-through store per peer, polls stay local); on one
    // GPU it is an emulation, bit-identical to ranks = 1 (tests/test_gpu_resident_big.py).
    int ranks;
    // DEEP: level 3 in polynomial form + remote tail rooted at level 4 (see the header comment)
    int N3, N4;
    ResCsr Pt3, P3d;          // level 2 <-> 3: rows of P3' (N3 x N2) and of P3 (N2 x N3)
    const double* p3rows;     // [N3 + N4][RB_P3_LD]: pack_bpoly's row layout with RB_P3_SEG segments
    const double* p3w;        // N3 + N4: the rows' factors of 1'r_3
    // POLY4 (N5 > 0): level 4 in polynomial form in the resident workgroups as well (at most two rows each and
    // one restriction row: N4 <= 2 G, N5 <= G), the tail workgroup rooted at level 5.  With the tail rooted at
    // level 4 a W cycle streams level 4's operators (0.9 MB at 200 rows) sixteen times through ONE compute
    // unit: 176 of 444 us on the Newton systems of the m = n = 2048 run.
    int N5;
    const double* p4rows;     // [N4 + N5][RB_P4_LD]
    const double* p4w;        // N4 + N5
    // ... and what its tail workgroup needs (res_tail_workgroup's fields of ResDesc)
    const SolveDesc* sub;     // LDS image of levels 4..J
    unsigned char* tin;       // 2 x RES_GRAN_MAX granules by visit parity: r_4 for the tail
    unsigned char* tout;      // ... its answer e_4
    unsigned* tctl;           // [0] != 0: the solve is over, the tail workgroup leaves
};

__device__ __forceinline__ void rb_publish(__amdgpu_buffer_rsrc_t rs, unsigned seq, int gidx, double v, int ranks = 1) {
    for (int r = 0; r < ranks; ++r)   // one copy of the buffer per rank group (ResBigDesc::ranks)
        __builtin_amdgcn_raw_buffer_store_b128(res_pack(v, seq), rs,
                                               r * (2 * RB_GRAN * 16) + (int)(seq & 1) * (RB_GRAN * 16) + gidx * 16, 0,
                                               16 /* sc1: write-through */);
}
template <int NJ>
__device__ __forceinline__ bool rb_sweep(__amdgpu_buffer_rsrc_t rs, unsigned seq, int n, bool dead, unsigned* tmo,
                                         double (&v)[NJ], int pollsleep, int group = 0) {
    const int base = group * (2 * RB_GRAN * 16) + (int)(seq & 1) * (RB_GRAN * 16);
    const int j0 = threadIdx.x;
    unsigned spins = 0;
    bool bad = false;
    if (!dead) {
        for (;;) {
            res_v4u gq[NJ];
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                const int j = j0 + u * BT;
                gq[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (j < n ? j : 0) * 16, 0, 16 /* sc1 */);
            }
            bool ok = true;
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                const int j = j0 + u * BT;
                ok &= (j >= n) | ((gq[u].y == seq) & (gq[u].w == seq));
                v[u] = __hiloint2double((int)gq[u].z, (int)gq[u].x);
            }
            if (__all(ok)) break;
            if (++spins > RES_SPIN_MAX ||
                ((spins & 255) == 255 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                bad = true;
                break;
            }
            for (int ps = 0; ps < pollsleep; ++ps) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    }
    return bad;
}

// out[]: the layout of k_resident / k_solve_small.
template <int KE2, int RPW, bool DEEP>
__global__ __launch_bounds__(BT, 2) void k_resident_big(const ResBigDesc D, const double* __restrict__ bvec,
                                                        double* xg, double* out, int fixed_cycles) {
    static_assert(RPW >= 1 && RPW <= RB_RPW_MAX, "rows per wave");
    constexpr int PUBW = RES_WAVES * RPW;   // rows of a block per workgroup
    extern __shared__ __attribute__((aligned(16))) char rb_smem[];
    double* sm = reinterpret_cast<double*>(rb_smem);
    const char* smb = rb_smem;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, G = gridDim.x - (DEEP ? 1 : 0);
    if (DEEP && b == G) {   // the tail workgroup: levels 4..J out of its LDS image
        __shared__ PhaseLds tail_lds;
        __shared__ double tail_red[16];
        __shared__ double tail_part[48 + SOLVE_ML + 1];
        __shared__ int tail_stat[RES_WAVES];
        ResDesc T{};
        T.sub = D.sub;
        T.tail_root = D.N5 > 0 ? 5 : 4;
        T.Nt = D.N5 > 0 ? D.N5 : D.N4;
        T.N5 = D.N5;             // (res_tail_workgroup: the inbox of a tail rooted at level 5 has N5 rows)
        T.remote = 1;
        T.three = 1;
        T.wcycle = D.wcycle;
        T.p3rows = D.p3rows;   // (non-NULL: the tail answers with e_4, the resident workgroups apply M1 P4)
        T.tin = D.tin;
        T.tout = D.tout;
        T.tctl = D.tctl;
        T.tmo = D.tmo;
        T.L2.N = D.N2;
        T.L3.N = D.N3;
        res_tail_workgroup(T, rb_smem, &tail_lds, tail_red, tail_part, tail_stat);
        return;
    }
    const int nf = D.nf, nc = D.nc, N1 = nf + nc, N2 = D.N2;
    const int N3 = DEEP ? D.N3 : 0, N4 = DEEP ? D.N4 : 0, N5 = DEEP ? D.N5 : 0;
    const bool poly4 = DEEP && N5 > 0;
    // LDS map (doubles); E2 is a gather target of the register rows: below 64 KB
    constexpr int oE2 = 0, oTU = RB_HALF, oP3C = 2 * RB_HALF, oE1S = 3 * RB_HALF, oXS = oE1S + RB_NMAX;
    constexpr int oRR2 = oP3C;   // DEEP: rr_2 for everybody (there is no one-row tail column then)
    constexpr int oRED = oXS + RB_NMAX, oPUB = oRED + 2 * RES_WAVES, oOWN = oPUB + 2 * 8 * RB_RPW_MAX;
    constexpr int oFAIL = oOWN + 20 * 8 * RB_RPW_MAX;
    constexpr int oR3L = oFAIL + 8, oE3L = oR3L + RB_N3MAX, oE4 = oE3L + RB_N3MAX, oR4L = oE4 + RB_N4MAX;
    constexpr int oE5 = oR4L + RB_N4MAX, oPS = oE5 + RB_N5MAX;   // (oPS: 128 doubles)
    int* fail = reinterpret_cast<int*>(sm + oFAIL);
    double* red = sm + oRED;

    // ---- rows of this wave: rows w + 8 p (p < RPW) of the workgroup's run of each block -------------
    const int loF = (int)(((long long)b * nf) / G), hiF = (int)(((long long)(b + 1) * nf) / G);
    const int loC = (int)(((long long)b * nc) / G), hiC = (int)(((long long)(b + 1) * nc) / G);   // C index (0-based)
    bool vF[RPW], vC[RPW];   // level-2 row of the wave = its C node
    int rF[RPW], rCi[RPW];
#pragma unroll
    for (int p = 0; p < RPW; ++p) {
        const int rowF = loF + w + RES_WAVES * p, rowCi = loC + w + RES_WAVES * p;
        vF[p] = rowF < hiF;
        vC[p] = rowCi < hiC;
        rF[p] = vF[p] ? rowF : 0;
        rCi[p] = vC[p] ? rowCi : 0;
    }
    // level-2 row slices -> registers; mask bits of the F rows (over C nodes) and of the C rows (over F rows)
    unsigned c2[RPW][KE2 / 2];
    double a2[RPW][KE2];
    {
        ResLevelDesc L2;
        L2.N = N2;
        L2.nf = 0;
        L2.S = D.S2;
        L2.pci = D.pci2;
        L2.pva = D.pva2;
        L2.diag = D.diag2;
        L2.dinv = D.dinv2;
        L2.Axi = D.Axi2;
        L2.xx = D.xx2;
#pragma unroll
        for (int p = 0; p < RPW; ++p) res_load_slice<KE2>(L2, rCi[p], vC[p], lane, c2[p], a2[p]);
    }
    unsigned bitsF[RPW], bitsC[RPW];   // bit q <-> entry lane + 64 q
#pragma unroll
    for (int p = 0; p < RPW; ++p) {
        bitsF[p] = bitsC[p] = 0;
        for (int q0 = 0; q0 < 32; q0 += 8) {   // eight words of each row per burst
            unsigned long long wf[8], wc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                wf[q] = D.fbits[(size_t)rF[p] * D.nwf + (q0 + q < D.nwf ? q0 + q : 0)];
                wc[q] = D.cbits[(size_t)rCi[p] * D.nwc + (q0 + q < D.nwc ? q0 + q : 0)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                bitsF[p] |= (vF[p] && q0 + q < D.nwf) ? (unsigned)((wf[q] >> lane) & 1ull) << (q0 + q) : 0u;
                bitsC[p] |= (vC[p] && q0 + q < D.nwc) ? (unsigned)((wc[q] >> lane) & 1ull) << (q0 + q) : 0u;
            }
        }
    }
    // constants of the own rows (LDS: a register pair each would stay live for the whole solve)
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            const int o = oOWN + w + RES_WAVES * p;
            sm[o + 0 * PUBW] = D.diag1[rF[p]];
            sm[o + 1 * PUBW] = D.dinv1[rF[p]];
            sm[o + 2 * PUBW] = bvec[rF[p]];
            sm[o + 3 * PUBW] = D.Axi1[rF[p]];
            sm[o + 4 * PUBW] = D.alpha[rF[p]];
            sm[o + 5 * PUBW] = D.rho[rF[p]];
            sm[o + 6 * PUBW] = D.diag1[nf + rCi[p]];
            sm[o + 7 * PUBW] = D.dinv1[nf + rCi[p]];
            sm[o + 8 * PUBW] = bvec[nf + rCi[p]];
            sm[o + 9 * PUBW] = D.Axi1[nf + rCi[p]];
            sm[o + 10 * PUBW] = D.beta[rCi[p]];
            sm[o + 11 * PUBW] = D.diag2[rCi[p]];
            sm[o + 12 * PUBW] = D.dinv2[rCi[p]];
            sm[o + 13 * PUBW] = D.Axi2[rCi[p]];
        }
    }
#define RB_OWN(k, p) sm[oOWN + (k) * PUBW + w + RES_WAVES * (p)]
#define RB_dgF(p) RB_OWN(0, p)
#define RB_dvF(p) RB_OWN(1, p)
#define RB_bF(p) RB_OWN(2, p)
#define RB_axF(p) RB_OWN(3, p)
#define RB_alF(p) RB_OWN(4, p)
#define RB_rhF(p) RB_OWN(5, p)
#define RB_dgC(p) RB_OWN(6, p)
#define RB_dvC(p) RB_OWN(7, p)
#define RB_bC(p) RB_OWN(8, p)
#define RB_axC(p) RB_OWN(9, p)
#define RB_btC(p) RB_OWN(10, p)
#define RB_dg2(p) RB_OWN(11, p)
#define RB_dv2(p) RB_OWN(12, p)
#define RB_ax2(p) RB_OWN(13, p)
// ... and their right-hand sides / the iterate x: uniform per wave, read once or twice per half sweep
#define RB_xF(p) RB_OWN(14, p)
#define RB_xC(p) RB_OWN(15, p)
#define RB_rF(p) RB_OWN(16, p)
#define RB_rC(p) RB_OWN(17, p)
#define RB_r2(p) RB_OWN(18, p)
    const bool nsp = D.isnsp != 0;
    const double xx1 = nsp ? D.xx1[0] : 1.0, xx2 = nsp ? D.xx2[0] : 1.0;
    // per-thread constants of the granules this thread receives -- granule j = tid + u BT (u < 4) of a
    // block is the block's row j: scale (alpha on F rows, beta on C rows) and (A1) of level 1, (A1) of level 2
    double sF4[4], aF4[4], sC4[4], aC4[4], ax2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = tid + u * BT;
        sF4[u] = j < nf ? D.alpha[j] : 0.0;
        aF4[u] = j < nf ? D.Axi1[j] : 0.0;
        sC4[u] = j < nc ? D.beta[j] : 0.0;
        aC4[u] = j < nc ? D.Axi1[nf + j] : 0.0;
        ax2[u] = j < N2 ? D.Axi2[j] : 0.0;
    }
    // sums of (A1) over the two blocks of level 1 (the shift of the half that is NOT handed off), the
    // tail's column P3 (dense) and kappa = (A1_2)' P3
    double saxF = 0.0, saxC = 0.0, kappa = 0.0, h33 = 0.0;
    {
        double pF = 0.0, pC = 0.0, pk = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pF += aF4[u];
            pC += aC4[u];
        }
        if (!DEEP) {
            for (int j = tid; j < N2; j += BT) {
                double v = 0.0;
                for (int t = D.P3.rp[j]; t < D.P3.rp[j + 1]; ++t)
                    if (D.P3.ci[t] == 0) v = D.P3.va[t];
                sm[oP3C + j] = v;
                pk += D.Axi2[j] * v;
            }
            for (int t = D.A3.rp[0]; t < D.A3.rp[1]; ++t)
                if (D.A3.ci[t] == 0) h33 = D.A3.va[t];
        }
        pF = wave_sum(pF);
        pC = wave_sum(pC);
        pk = wave_sum(pk);
        if (lane == 0) {
            red[w] = pF;
            red[RES_WAVES + w] = pC;
        }
        __syncthreads();
        saxF = res_red8(red);
        saxC = res_red8(red + RES_WAVES);
        __syncthreads();
        if (lane == 0) red[w] = pk;
        __syncthreads();
        kappa = res_red8(red);
        __syncthreads();
    }
    // ---- DEEP: level 3 in polynomial form.  Rows lo3..hi3-1 of [M2a | M1] (at most four) and restriction
    // row b (N4 <= G: one per workgroup), entries tid and BT + tid of a segment per thread; the rows of P3'
    // that produce this workgroup's r_3 entries (wave q < n3own) and the rows of P3 of the own level-2 rows
    const int lo3 = DEEP ? (int)(((long long)b * N3) / G) : 0, hi3 = DEEP ? (int)(((long long)(b + 1) * N3) / G) : 0;
    const int n3own = hi3 - lo3;
    const int lo4 = poly4 ? (int)(((long long)b * N4) / G) : 0, hi4 = poly4 ? (int)(((long long)(b + 1) * N4) / G) : 0;
    const int n4own = hi4 - lo4;
    int pt3e0 = 0, pt3e1 = 0, p3e0[RPW], p3e1[RPW];
#pragma unroll
    for (int p = 0; p < RPW; ++p) p3e0[p] = p3e1[p] = 0;
    if (DEEP) {
        if (tid < 5) {
            const int row = tid < 4 ? lo3 + tid : N3 + b;
            sm[oPS + 48 + tid] = (tid < 4 ? row < hi3 : b < N4) ? D.p3w[row] : 0.0;
        }
        if (w < n3own) {
            pt3e0 = D.Pt3.rp[lo3 + w];
            pt3e1 = D.Pt3.rp[lo3 + w + 1];
        }
#pragma unroll
        for (int p = 0; p < RPW; ++p)
            if (vC[p]) {
                p3e0[p] = D.P3d.rp[rCi[p]];
                p3e1[p] = D.P3d.rp[rCi[p] + 1];
            }
        for (int j = tid; j < RB_N3MAX; j += BT) sm[oR3L + j] = sm[oE3L + j] = 0.0;
        for (int j = tid; j < RB_N4MAX; j += BT) sm[oE4 + j] = sm[oR4L + j] = 0.0;
        for (int j = tid; j < RB_N5MAX; j += BT) sm[oE5 + j] = 0.0;
        if (poly4 && tid < 3) {
            const int row = tid < 2 ? lo4 + tid : N4 + b;
            sm[oPS + 112 + tid] = (tid < 2 ? row < hi4 : b < N5) ? D.p4w[row] : 0.0;
        }
    }
    // x: scaled gather copy for everybody, the own rows' values as wave scalars
    for (int j = tid; j < N1; j += BT) {
        const double sc = j < nf ? D.alpha[j] : D.beta[j - nf];
        sm[oXS + j] = sc * xg[j];
        sm[oE1S + j] = 0.0;
    }
    for (int j = tid; j < N2; j += BT) sm[oE2 + j] = 0.0;
    double eF[RPW], eC[RPW], e2v[RPW];
#pragma unroll
    for (int p = 0; p < RPW; ++p) {
        if (lane == 0) {
            RB_xF(p) = vF[p] ? xg[rF[p]] : 0.0;
            RB_xC(p) = vC[p] ? xg[nf + rCi[p]] : 0.0;
            RB_rF(p) = RB_rC(p) = RB_r2(p) = 0.0;
        }
        eF[p] = eC[p] = e2v[p] = 0.0;
    }
    if (tid == 0) *fail = 0;
    __syncthreads();

    // Addresses are re-derived from FRESH copies of the thread's indices at every use site (RB_FRESH shadows
    // tid / w / lane by copies laundered through an empty volatile asm): left alone, LLVM reassociates every
    // `index + constant` of the own-row slots, the hand-off stores and the polynomial rows into a loop-invariant
    // part and hoists it out of the cycle loop -- a hundred address registers (the LDS map ends beyond the 64 KB
    // an immediate offset reaches, so each address is a register of its own) that push the data the loop
    // needs into scratch.
#define RB_FRESH                                                                                   \
    const int tid0_ = tid, w0_ = w, lane0_ = lane;                                                 \
    int tid = tid0_, w = w0_, lane = lane0_;                                                       \
    asm volatile("" : "+v"(tid), "+v"(w), "+v"(lane))
    const int ranks = DEEP ? 1 : (D.ranks > 1 ? D.ranks : 1);
    const int mygroup = (int)(((long long)b * ranks) / G);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(D.gran, 0, ranks * 2 * RB_GRAN * 16, 0x00020000);
    unsigned seq = 0;
    bool dead = false;

    // sum of the LDS vector at `off` over the set bits of `bits` (entry lane + 64 q <-> bit q)
    auto masked_sum = [&](unsigned bits, int off, int n) __attribute__((always_inline)) {
        RB_FRESH;
        double s0 = 0.0, s1 = 0.0;
        const int nq = (n + 63) >> 6;
        // entry lane + 64 q sits at a constant distance from entry `lane`: one address register and immediate
        // offsets (an index clamped to n cost a register per gather, all of them hoisted out of the cycle loop
        // and spilled).  Entries beyond n lie inside the vector's LDS slot and their mask bits are zero.
        const double* base = sm + off + lane;
        for (int q0 = 0; q0 < nq; q0 += 8) {   // eight gathers in flight (all 32 at once spilled registers)
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = base[64 * (q0 + q)];
            const unsigned bq = bits >> q0;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                s0 += ((bq >> q) & 1u) ? x[q] : 0.0;
                s1 += ((bq >> (q + 1)) & 1u) ? x[q + 1] : 0.0;
            }
        }
        return wave_sum(s0 + s1);
    };

    // hand-off: barrier, wave 0 publishes sm[oPUB ..] (block A: granules gA.., cA rows; block B: the values
    // at oPUB + PUBW ..), sweep of n granules, STORE(j, v) per granule of the thread, block sums of p0 / p1
#define RB_HANDOFF(NJ, n, gA, cA, gB, cB, STORE, want_sums, t0, t1)                                  \
    do {                                                                                           \
        double hv_[NJ];                                                                            \
        RB_FRESH;                                                                                  \
        ++seq;                                                                                     \
        if (*fail) dead = true;   /* (the previous hand-off's give-up: read under the barrier) */  \
        __syncthreads();                                                                           \
        if (w == 0) {                                                                              \
            const int l8_ = lane & (PUBW - 1);                                                     \
            const bool second_ = lane >= PUBW;                                                     \
            if (lane < 2 * PUBW && l8_ < (second_ ? (cB) : (cA)) &&                                \
                !(seq == D.dbg_skip_seq && b == G - 1))                                            \
                rb_publish(rs, seq, (second_ ? (gB) : (gA)) + l8_, sm[oPUB + lane], ranks);        \
        }                                                                                          \
        for (int ps_ = 0; ps_ < D.presleep; ++ps_) __builtin_amdgcn_s_sleep(1);                    \
        if (rb_sweep<NJ>(rs, seq, (n), dead, D.tmo, hv_, D.pollsleep, mygroup)) {                   \
            *fail = 1;                                                                             \
            if (lane == 0) __hip_atomic_store(D.tmo, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        }                                                                                          \
        double p0 = 0.0, p1 = 0.0;                                                                 \
        _Pragma("unroll") for (int u_ = 0; u_ < NJ; ++u_) {                                        \
            const int j = tid + u_ * BT;                                                           \
            if (j < (n)) {                                                                         \
                const double v = hv_[u_];                                                          \
                STORE;                                                                             \
            }                                                                                      \
        }                                                                                          \
        if (want_sums) {                                                                           \
            p0 = wave_sum(p0);                                                                     \
            if ((want_sums) > 1) p1 = wave_sum(p1);                                                \
            if (lane == 0) {                                                                       \
                sm[oRED + w] = p0;                                                                       \
                if ((want_sums) > 1) sm[oRED + RES_WAVES + w] = p1;                                      \
            }                                                                                      \
        }                                                                                          \
        __syncthreads();                                                                           \
        if (want_sums) {                                                                           \
            t0 = res_red8(sm + oRED);                                                                    \
            if ((want_sums) > 1) t1 = res_red8(sm + oRED + RES_WAVES);                                   \
        }                                                                                          \
    } while (0)
    // exchange of K <= 2 partial sums per workgroup: t0 / t1 = their totals (same order everywhere)
#define RB_PARTIALS(K, v0, v1, t0, t1)                                                             \
    do {                                                                                           \
        RB_FRESH;                                                                                  \
        if (w == 0 && lane == 0) {                                                                 \
            sm[oPUB + 0] = (v0);                                                                   \
            sm[oPUB + 1] = (v1);                                                                   \
        }                                                                                          \
        RB_HANDOFF(2, (K) * G, (K) * b, (K), 0, 0, { if ((K) == 1 || (j & 1) == 0) p0 += v; else p1 += v; }, (K), t0, t1); \
    } while (0)

    double c1 = 0.0, c2s = 0.0, sumr1 = 0.0, sumr2 = 0.0, xig2 = 0.0;
    double dum0 = 0.0, dum1 = 0.0;
    (void)dum0;
    (void)dum1;

    // block sum of a per-wave value held by lane 0 of the waves with a row (others contribute 0)
    auto own_pair_sum = [&](double vA, double vB, double& tA, double& tB) __attribute__((always_inline)) {
        RB_FRESH;
        if (lane == 0) {
            sm[oRED + w] = vA;
            sm[oRED + RES_WAVES + w] = vB;
        }
        __syncthreads();
        tA = res_red8(sm + oRED);
        tB = res_red8(sm + oRED + RES_WAVES);
        __syncthreads();
    };

    // r = b - A x on the own rows, ||r|| and 1'r by partial sums; E1 := 0            Class_AMG.m:89,96,103
    auto top = [&]() __attribute__((always_inline)) {
        RB_FRESH;
        double a2_ = 0.0, a1_ = 0.0;
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            const double sF = masked_sum(bitsF[p], oXS + nf, nc), sC = masked_sum(bitsC[p], oXS, nf);
            const double rFv = vF[p] ? RB_bF(p) - (RB_dgF(p) * RB_xF(p) - RB_alF(p) * sF) : 0.0;
            const double rCv = vC[p] ? RB_bC(p) - (RB_dgC(p) * RB_xC(p) - RB_btC(p) * sC) : 0.0;
            if (lane == 0) {
                RB_rF(p) = rFv;
                RB_rC(p) = rCv;
            }
            eF[p] = eC[p] = 0.0;
            a2_ += rFv * rFv + rCv * rCv;
            a1_ += rFv + rCv;
        }
        double q2 = 0.0, q1 = 0.0;
        own_pair_sum(a2_, a1_, q2, q1);
        double nrm2 = 0.0;
        RB_PARTIALS(2, q2, q1, nrm2, sumr1);
        c1 = nsp ? sumr1 / xx1 : 0.0;
        for (int j = tid; j < N1; j += BT) sm[oE1S + j] = 0.0;
        __syncthreads();
        return sqrt(nrm2);
    };

    // one half of a bigraph Gauss-Seidel sweep on level 1                  MG_Vcycle.m:15-21,34-38
    double afirst = 0.0;   // (A1)'w over the half handed off first
    auto half1 = [&](bool frows, bool first, bool ezero) __attribute__((always_inline)) {
        RB_FRESH;
        const double cc = c1;
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            const bool valid = frows ? vF[p] : vC[p];
            double s = 0.0;
            if (!(ezero && first)) s = frows ? masked_sum(bitsF[p], oE1S + nf, nc) : masked_sum(bitsC[p], oE1S, nf);
            const double eo = ezero ? 0.0 : (frows ? eF[p] : eC[p]);
            const double ae = (frows ? RB_dgF(p) : RB_dgC(p)) * eo - (frows ? RB_alF(p) : RB_btC(p)) * s;   // (A e)_row
            const double g_i = (frows ? RB_rF(p) : RB_rC(p)) - ae - (frows ? RB_axF(p) : RB_axC(p)) * cc;
            const double wv = eo + (frows ? RB_dvF(p) : RB_dvC(p)) * g_i;
            const double pub = first ? wv : wv + cc;
            if (lane == 0) sm[oPUB + w + RES_WAVES * p] = pub;
            if (first) {
                if (frows) eF[p] = valid ? wv : 0.0; else eC[p] = valid ? wv : 0.0;
            } else if (frows) {
                eF[p] = valid ? pub : 0.0;
                eC[p] = vC[p] ? eC[p] + cc : 0.0;
            } else {
                eC[p] = valid ? pub : 0.0;
                eF[p] = vF[p] ? eF[p] + cc : 0.0;
            }
        }
        const int blk0 = frows ? 0 : nf, nblk = frows ? nf : nc;
        const int g0 = frows ? loF : loC, cnt = frows ? hiF - loF : hiC - loC;
        if (first) {
            RB_HANDOFF(4, nblk, g0, cnt, 0, 0,
                       {
                           sm[oE1S + blk0 + j] = (frows ? sF4[u_] : sC4[u_]) * v;
                           p0 += (frows ? aF4[u_] : aC4[u_]) * v;
                       },
                       (nsp ? 1 : 0), afirst, dum1);
        } else {
            const int oth0 = frows ? nf : 0, noth = frows ? nc : nf;
            double asec = 0.0;
            RB_HANDOFF(4, nblk, g0, cnt, 0, 0,
                       {
                           sm[oE1S + blk0 + j] = (frows ? sF4[u_] : sC4[u_]) * v;
                           p0 += (frows ? aF4[u_] : aC4[u_]) * v;
                       },
                       (nsp ? 1 : 0), asec, dum1);
            // the other half moves by c as well (its rows' scaled copies here, the own scalars above)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jo = tid + q * BT;
                if (jo < noth) sm[oE1S + oth0 + jo] += (frows ? sC4[q] : sF4[q]) * cc;
            }
            __syncthreads();
            const double saxo = frows ? saxC : saxF;
            c1 = nsp ? (sumr1 - (asec + afirst + cc * saxo)) / xx1 : 0.0;
        }
    };
    auto sweep1 = [&](bool post, bool ezero) __attribute__((always_inline)) {
        RB_FRESH;
        half1(!post, true, ezero);    // pre: F rows first (Rk{1}); post: C rows first (Rk{1}')
        half1(post, false, ezero);
    };

    // weighted-Jacobi sweep on level 2                                   MG_Vcycle.m:15-21; Class_AMG.m:84
    auto sweep2 = [&](bool ezero) __attribute__((always_inline)) {
        RB_FRESH;
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            double s = 0.0, eo = 0.0;
            if (!ezero) {
                s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2[p], a2[p], smb));
                eo = e2v[p];
                s += RB_dg2(p) * eo;
            }
            const double g_i = RB_r2(p) - s - RB_ax2(p) * c2s;
            const double en = eo + RB_dv2(p) * g_i + c2s;
            if (lane == 0) sm[oPUB + w + RES_WAVES * p] = en;
            e2v[p] = vC[p] ? en : 0.0;
        }
        double asum = 0.0;
        RB_HANDOFF(4, N2, loC, hiC - loC, 0, 0,
                   {
                       sm[oE2 + j] = v;
                       p0 += ax2[u_] * v;
                   },
                   (nsp ? 1 : 0), asum, dum1);
        xig2 = sumr2 - asum;
        c2s = nsp ? xig2 / xx2 : 0.0;
    };

    // ---- DEEP: the polynomial level 3 and the remote tail below it -------------------------------------
    const auto rtin = __builtin_amdgcn_make_buffer_rsrc(D.tin, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    const auto rtout = __builtin_amdgcn_make_buffer_rsrc(D.tout, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    unsigned tseq = 0;
    double sumr3 = 0.0;
    // hand-off among the level-3 rows: N3 values + one "ack" granule per workgroup (a workgroup without a
    // row of level 3 still has to say that it has finished the step: two-buffer protocol)
#define RB_HANDOFF3(STORE3, want_sums, t0)                                                           \
    do {                                                                                           \
        RB_FRESH;                                                                                  \
        if (w == 0 && lane == 0) sm[oPUB + PUBW] = 0.0;                                            \
        RB_HANDOFF(3, N3 + G, lo3, n3own, N3 + b, 1, { if (j < N3) { STORE3; } }, want_sums, t0, dum1); \
    } while (0)
    // ... whose ack granule of workgroup b < N4 carries r_4[b] (POLY4: the restricted residual goes to everybody)
#define RB_HANDOFF3R(ACKV, STORE3, STORE4, want_sums, t0)                                            \
    do {                                                                                           \
        RB_FRESH;                                                                                  \
        if (w == 0 && lane == 0) sm[oPUB + PUBW] = (ACKV);                                         \
        RB_HANDOFF(3, N3 + G, lo3, n3own, N3 + b, 1,                                               \
                   { if (j < N3) { STORE3; } else if (j - N3 < N4) { const int j4 = j - N3; STORE4; } }, want_sums, t0, dum1); \
    } while (0)
    // the same among the rows of level 4 (N4 + G <= BT granules: one per thread)
#define RB_HANDOFF4(STORE4)                                                                          \
    do {                                                                                           \
        RB_FRESH;                                                                                  \
        if (w == 0 && lane == 0) sm[oPUB + PUBW] = 0.0;                                            \
        RB_HANDOFF(1, N4 + G, lo4, n4own, N4 + b, 1, { if (j < N4) { STORE4; } }, 0, dum0, dum1);  \
    } while (0)
    // the sums of this workgroup's rows against [r_3; e_3] (+ (M1 P4) e_4 in the second pass) and their factor
    // of 1'r_3 -> sm[oPS + 40 + q]
    // (the rows' coefficients -- entries tid and BT + tid of the two segments, entry tid of (M1 P4) -- are
    // fetched from L2 at every pass, 21 coalesced loads in flight per thread: held in registers for the whole
    // solve, as k_resident's POLY3 does with one entry per segment, these 48 registers pushed the per-granule
    // constants of the hand-offs into scratch, reloaded at every hand-off)
    auto poly3_rows = [&](int nrows, bool post) __attribute__((always_inline)) {
        RB_FRESH;
        double m3r[5][2], m3e[5][2], m3c[4];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int row = q < 4 ? lo3 + q : N3 + b;
            const bool okr = q < nrows && (q < 4 ? row < hi3 : b < N4);
            const double* pr = D.p3rows + (size_t)(okr ? row : 0) * RB_P3_LD;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = tid + u * BT;
                const double vr = pr[j], ve = pr[RB_P3_SEG + j];   // (columns beyond N3: zeros of the pack)
                m3r[q][u] = okr ? vr : 0.0;
                m3e[q][u] = okr ? ve : 0.0;
            }
            if (q < 4) {
                const double vc = pr[2 * RB_P3_SEG + (tid < RB_N4MAX ? tid : 0)];
                m3c[q] = (okr && post && tid < N4) ? vc : 0.0;
            }
        }
        double xr[2], xe[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = tid + u * BT;
            xr[u] = sm[oR3L + j];     // (zero beyond N3)
            xe[u] = sm[oE3L + j];
        }
        const double xc = (post && tid < N4) ? sm[oE4 + tid] : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (q < nrows) {
                double t = __builtin_fma(m3e[q][0], xe[0], m3r[q][0] * xr[0]);
                t = __builtin_fma(m3r[q][1], xr[1], t);
                t = __builtin_fma(m3e[q][1], xe[1], t);
                if (q < 4) t = __builtin_fma(m3c[q], xc, t);
                const double pq = wave_sum(t);
                if (lane == 0) sm[oPS + 8 * q + w] = pq;
            }
        }
        __syncthreads();
        if (tid < nrows) {
            double sq = 0.0;
#pragma unroll
            for (int ww = 0; ww < RES_WAVES; ++ww) sq += sm[oPS + 8 * tid + ww];
            sm[oPS + 40 + tid] = __builtin_fma(sm[oPS + 48 + tid], sumr3, sq);
        }
        __syncthreads();
    };
    // POLY4: the sums of this workgroup's rows of level 4 (at most two, and restriction row b) against
    // [r_4; e_4] (+ (M1 P5) e_5 in the second pass) -> sm[oPS + 104 + q]; one entry per thread and segment
    double sumr4 = 0.0;
    auto poly4_rows = [&](int nrows, bool post) __attribute__((always_inline)) {
        RB_FRESH;
        double m4r[3], m4e[3], m4c[2];
        const int t4 = tid < RB_N4MAX ? tid : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int row = q < 2 ? lo4 + q : N4 + b;
            const bool okr = q < nrows && (q < 2 ? row < hi4 : b < N5) && tid < RB_N4MAX;
            const double* pr = D.p4rows + (size_t)(okr ? row : 0) * RB_P4_LD;
            const double vr = pr[t4], ve = pr[RB_P4_SEG + t4];
            m4r[q] = okr ? vr : 0.0;
            m4e[q] = okr ? ve : 0.0;
            if (q < 2) {
                const double vc = pr[2 * RB_P4_SEG + (tid < RB_N5MAX ? tid : 0)];
                m4c[q] = (okr && post && tid < N5) ? vc : 0.0;
            }
        }
        const double xr = sm[oR4L + t4], xe = sm[oE4 + t4];
        const double xc = (post && tid < N5) ? sm[oE5 + tid] : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < nrows) {
                double t = __builtin_fma(m4e[q], xe, m4r[q] * xr);
                if (q < 2) t = __builtin_fma(m4c[q], xc, t);
                const double pq = wave_sum(t);
                if (lane == 0) sm[oPS + 64 + 8 * q + w] = pq;
            }
        }
        __syncthreads();
        if (tid < nrows) {
            double sq = 0.0;
#pragma unroll
            for (int ww = 0; ww < RES_WAVES; ++ww) sq += sm[oPS + 64 + 8 * tid + ww];
            sm[oPS + 104 + tid] = __builtin_fma(sm[oPS + 112 + tid], sumr4, sq);
        }
        __syncthreads();
    };
    // waits for the tail's answer (n values) and leaves it at sm[off ..]
    auto tail_answer = [&](int n, int off) __attribute__((always_inline)) {
        RB_FRESH;
        double hv[1];
        int st = 0;
        if (!dead) st = res_wait_slow<1>(rtout, tseq, n, D.tmo, nullptr, hv);
        if (st) {
            *fail = 1;
            if (lane == 0) __hip_atomic_store(D.tmo, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < n) sm[off + tid] = (!dead && !st) ? hv[0] : 0.0;
        __syncthreads();
        if (*fail) dead = true;
    };
    // one visit of level 4 (POLY4) and, through the tail workgroup, of everything below it
    auto visit4 = [&]() __attribute__((always_inline)) {
        RB_FRESH;
        ++tseq;
        poly4_rows(3, false);                                        // e_4' and r_5 = P5'(r_4 - A_4 e_4')
        if (tid == 0 && b < N5)
            __builtin_amdgcn_raw_buffer_store_b128(res_pack(sm[oPS + 106], tseq), rtin,
                                                   (int)(tseq & 1) * (RES_GRAN_MAX * 16) + b * 16, 0, 16 /* sc1 */);
        if (tid < 2) sm[oPUB + tid] = sm[oPS + 104 + tid];
        RB_HANDOFF4({ sm[oE4 + j] = v; });
        tail_answer(N5, oE5);                                        // e_5
        poly4_rows(2, true);                                         // e_4'' = M2a r + M1 e' + (M1 P5) e_5
        if (tid < 2) sm[oPUB + tid] = sm[oPS + 104 + tid];
        RB_HANDOFF4({ sm[oE4 + j] = v; });
    };
    // one visit of level 3 and of everything below it
    auto visit3 = [&]() __attribute__((always_inline)) {
        RB_FRESH;
        // e' = M2a r + M1 e and the restricted residual of e' in one pass                 MG_Vcycle.m:20-29
        poly3_rows(5, false);
        if (poly4) {   // r_4 to everybody in the ack granules of level 3's hand-off; e_4 := 0
            if (tid < 4) sm[oPUB + tid] = sm[oPS + 40 + tid];
            double s4 = 0.0;
            RB_HANDOFF3R((b < N4 ? sm[oPS + 44] : 0.0), { sm[oE3L + j] = v; },
                         { sm[oR4L + j4] = v; sm[oE4 + j4] = 0.0; p0 += v; }, (nsp ? 1 : 0), s4);
            sumr4 = nsp ? s4 : 0.0;
            for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) visit4();              // MG_Wcycle.m:28-30
        } else {
            ++tseq;
            if (tid == 0 && b < N4)
                __builtin_amdgcn_raw_buffer_store_b128(res_pack(sm[oPS + 44], tseq), rtin,
                                                       (int)(tseq & 1) * (RES_GRAN_MAX * 16) + b * 16, 0, 16 /* sc1 */);
            if (tid < 4) sm[oPUB + tid] = sm[oPS + 40 + tid];
            RB_HANDOFF3({ sm[oE3L + j] = v; }, 0, dum0);
            tail_answer(N4, oE4);                                    // e_4
        }
        poly3_rows(4, true);                                         // e'' = M2a r + M1 e' + (M1 P4) e_4   :31-41
        if (tid < 4) sm[oPUB + tid] = sm[oPS + 40 + tid];
        RB_HANDOFF3({ sm[oE3L + j] = v; }, 0, dum0);
    };

    // one visit of level 2 and of everything below it
    auto visit2 = [&](bool keep) __attribute__((always_inline)) {
        RB_FRESH;
        const int nu = D.nu;
        for (int s = 0; s < nu; ++s) sweep2(!keep && s == 0);
        // rr = r - A e on the own rows                                                     MG_Vcycle.m:27
        double rr[RPW];
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            const double s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2[p], a2[p], smb)) + RB_dg2(p) * e2v[p];
            rr[p] = vC[p] ? RB_r2(p) - s : 0.0;
        }
        if (!DEEP) {   // r_3 = P3' rr by partial sums, the one-row tail by every thread
            double mine = 0.0;
#pragma unroll
            for (int p = 0; p < RPW; ++p) mine += vC[p] ? sm[oP3C + rCi[p]] * rr[p] : 0.0;
            double part = 0.0, dumA = 0.0;
            own_pair_sum(mine, 0.0, part, dumA);
            double r3 = 0.0;
            RB_PARTIALS(1, part, 0.0, r3, dum1);
            // PCG.m:68-87 on the 1 x 1 system, by every thread
            double r = r3;
            double pp = r / h33, d = 0.0;
            double delta_new = r * pp;
            const double thresh = 1e-11 * 1e-11 * delta_new;
            for (long long it = 0; it < D.pcg_maxit && delta_new > thresh; ++it) {
                const double delta_old = delta_new;
                const double q = h33 * pp;
                const double alpha = delta_old / (q * pp);
                d += alpha * pp;
                r = r - alpha * q;
                const double wi = r / h33;
                delta_new = r * wi;
                pp = wi + (delta_new / delta_old) * pp;
            }
            // e_2 += P3 e_3 on everybody's copy and on the own scalars; 1'(r - A e) moves by -d kappa
            for (int j = tid; j < N2; j += BT) sm[oE2 + j] = sm[oE2 + j] + sm[oP3C + j] * d;
#pragma unroll
            for (int p = 0; p < RPW; ++p) e2v[p] = vC[p] ? e2v[p] + sm[oP3C + rCi[p]] * d : 0.0;
            xig2 = xig2 - d * kappa;
            c2s = nsp ? xig2 / xx2 : 0.0;
            __syncthreads();
        } else {
            // rr_2 to everybody (the rows of P3' gather from it), r_3 = P3' rr_2 by the owners of level 3's rows
            if (lane == 0) {
#pragma unroll
                for (int p = 0; p < RPW; ++p) sm[oPUB + w + RES_WAVES * p] = rr[p];
            }
            RB_HANDOFF(4, N2, loC, hiC - loC, 0, 0, { sm[oRR2 + j] = v; }, 0, dum0, dum1);
            const double s3 = w < n3own ? res_csr_rowdot(D.Pt3, pt3e0, pt3e1, lane, sm, oRR2) : 0.0;
            if (lane == 0) sm[oPUB + w] = s3;
            RB_HANDOFF3({ sm[oR3L + j] = v; sm[oE3L + j] = 0.0; p0 += v; }, (nsp ? 1 : 0), sumr3);
            if (!nsp) sumr3 = 0.0;
            for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) visit3();              // MG_Wcycle.m:28-30
            // e_2 += P3 e_3 on the own rows                                            MG_Vcycle.m:31
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const double sP = res_csr_rowdot(D.P3d, p3e0[p], p3e1[p], lane, sm, oE3L);
                const double en = e2v[p] + sP;
                e2v[p] = vC[p] ? en : 0.0;
                if (lane == 0) sm[oPUB + w + RES_WAVES * p] = en;
            }
            double asum = 0.0;
            RB_HANDOFF(4, N2, loC, hiC - loC, 0, 0,
                       {
                           sm[oE2 + j] = v;
                           p0 += ax2[u_] * v;
                       },
                       (nsp ? 1 : 0), asum, dum1);
            xig2 = sumr2 - asum;
            c2s = nsp ? xig2 / xx2 : 0.0;
        }
        for (int s = 0; s < nu; ++s) sweep2(false);
    };

    // MG_Vcycle / MG_Wcycle from level 1 down; the correction ends in E1S / eF, eC
    auto cycle = [&]() __attribute__((always_inline)) {
        RB_FRESH;
        const int nu = D.nu;
        for (int s = 0; s < nu; ++s) sweep1(false, s == 0);
        {   // rr = r - A e; the F part goes out scaled by rho (the restriction's operand), the C part stays
            double rrC[RPW];
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const double sF = masked_sum(bitsF[p], oE1S + nf, nc), sC = masked_sum(bitsC[p], oE1S, nf);
                const double rrF = RB_rF(p) - (RB_dgF(p) * eF[p] - RB_alF(p) * sF);
                rrC[p] = RB_rC(p) - (RB_dgC(p) * eC[p] - RB_btC(p) * sC);
                if (lane == 0) sm[oPUB + w + RES_WAVES * p] = RB_rhF(p) * rrF;
            }
            RB_HANDOFF(4, nf, loF, hiF - loF, 0, 0, { sm[oTU + j] = v; }, 0, dum0, dum1);
            // r_2 = P' rr: row c of P' is [W(:,c)', 1 at the C node]; E2 := 0; 1'r_2 by partial sums
            double mine = 0.0;
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const double s2 = RB_btC(p) * masked_sum(bitsC[p], oTU, nf) + rrC[p];
                const double r2 = vC[p] ? s2 : 0.0;
                if (lane == 0) RB_r2(p) = r2;
                e2v[p] = 0.0;
                mine += r2;
            }
            double part = 0.0, dumA = 0.0;
            own_pair_sum(mine, 0.0, part, dumA);
            RB_PARTIALS(1, part, 0.0, sumr2, dum1);
            xig2 = sumr2;
            c2s = nsp ? sumr2 / xx2 : 0.0;
        }
        for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) visit2(leg == 1);      // MG_Wcycle.m:28-30
        {   // e_1 += P e_2: F rows rho_j sum_i s_ij beta_i e2_i, C rows the identity      MG_Vcycle.m:31
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = tid + q * BT;
                if (j < N2) sm[oTU + j] = sC4[q] * sm[oE2 + j];   // (level 2 = the C nodes: beta_j)
            }
            __syncthreads();
            double nF[RPW], nC[RPW];
#pragma unroll
            for (int p = 0; p < RPW; ++p) {
                const double sF = RB_rhF(p) * masked_sum(bitsF[p], oTU, N2);
                nF[p] = eF[p] + sF;
                nC[p] = eC[p] + e2v[p];
                eF[p] = vF[p] ? nF[p] : 0.0;
                eC[p] = vC[p] ? nC[p] : 0.0;
            }
            // the new iterate goes out block by block (a thread's constants are per block)
            double asF = 0.0, asC = 0.0;
            if (lane == 0) {
#pragma unroll
                for (int p = 0; p < RPW; ++p) sm[oPUB + w + RES_WAVES * p] = nF[p];
            }
            RB_HANDOFF(4, nf, loF, hiF - loF, 0, 0,
                       {
                           sm[oE1S + j] = sF4[u_] * v;
                           p0 += aF4[u_] * v;
                       },
                       (nsp ? 1 : 0), asF, dum1);
            if (lane == 0) {
#pragma unroll
                for (int p = 0; p < RPW; ++p) sm[oPUB + w + RES_WAVES * p] = nC[p];
            }
            RB_HANDOFF(4, nc, loC, hiC - loC, 0, 0,
                       {
                           sm[oE1S + nf + j] = sC4[u_] * v;
                           p0 += aC4[u_] * v;
                       },
                       (nsp ? 1 : 0), asC, dum1);
            c1 = nsp ? (sumr1 - (asF + asC)) / xx1 : 0.0;
        }
        for (int s = 0; s < nu; ++s) sweep1(true, false);
    };

    auto add_correction = [&]() __attribute__((always_inline)) {   // x += e                          Class_AMG.m:98,101
        RB_FRESH;
        for (int j = tid; j < N1; j += BT) sm[oXS + j] = sm[oXS + j] + sm[oE1S + j];
#pragma unroll
        for (int p = 0; p < RPW; ++p)
            if (lane == 0) {
                RB_xF(p) = RB_xF(p) + eF[p];
                RB_xC(p) = RB_xC(p) + eC[p];
            }
        __syncthreads();
    };

    // ---- Class_AMG.m:86-109 ---------------------------------------------------------------------
    const int maxit = D.maxit;
    double* relk = out + 4;
    double* rhok = out + 4 + (maxit + 2);
    const bool writer = b == 0 && tid == 0;
    const bool fixed = fixed_cycles > 0;
    int it = 0, done = 0;
    double rel_res = 0.0, last_rel = 1.0, res = 0.0, res0 = 0.0, prev = 0.0;
    bool firstp = true;
    for (;;) {
        const double rnow = top();                                                // :89 / :103
        if (firstp) {
            firstp = false;
            res0 = res = rnow;
            if (!fixed) {
                if (res0 == 0.0) {                                                // :91-92
                    if (writer) {
                        relk[0] = 0.0;
                        rhok[0] = INFINITY;
                    }
                    break;
                }
                it = 1;                                                           // :94
                if (writer) {
                    relk[0] = 1.0;
                    rhok[0] = NAN;
                }
            }
        } else {
            prev = res;
            res = rnow;
            rel_res = res / res0;                                                 // :104
            const double rho = res / prev;                                        // :105
            if (fixed) {
                ++done;
            } else {
                if (writer) {
                    relk[it] = rel_res;
                    rhok[it] = rho;
                }
                last_rel = rel_res;
                ++it;
                if (rho > 1.0) break;                                             // :106
            }
        }
        if (dead) break;
        if (fixed ? done >= fixed_cycles : !(last_rel > D.retol && it <= maxit)) break;   // :95
        if (D.anycycle) {
            cycle();                                                              // :97-102
            add_correction();
        }
    }
    if (fixed)
        it = fixed_cycles;
    else if (res0 != 0.0)
        it -= 1;                                                                  // :108
    if (DEEP && b == 0 && tid == 0)   // release the tail workgroup
        __hip_atomic_store(D.tctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {   // every workgroup writes its own rows of x
#pragma unroll
        for (int p = 0; p < RPW; ++p) {
            if (vF[p]) xg[rF[p]] = RB_xF(p);
            if (vC[p]) xg[nf + rCi[p]] = RB_xC(p);
        }
    }
    if (writer) {
        const unsigned anytmo = __hip_atomic_load(D.tmo, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        out[0] = (double)it;
        out[1] = rel_res;
        out[2] = res0;
        out[3] = (dead || anytmo != 0) ? 1.0 : 0.0;
        out[4 + 2 * (maxit + 2) - 1] = (double)(seq + tseq);   // hand-offs of this launch (see k_resident)
    }
#undef RB_HANDOFF3
#undef RB_HANDOFF3R
#undef RB_HANDOFF4
#undef RB_FRESH
#undef RB_PARTIALS
#undef RB_HANDOFF
#undef RB_OWN
#undef RB_dgF
#undef RB_dvF
#undef RB_bF
#undef RB_axF
#undef RB_alF
#undef RB_rhF
#undef RB_dgC
#undef RB_dvC
#undef RB_bC
#undef RB_axC
#undef RB_btC
#undef RB_dg2
#undef RB_dv2
#undef RB_ax2
#undef RB_xF
#undef RB_xC
#undef RB_rF
#undef RB_rC
#undef RB_r2
}
