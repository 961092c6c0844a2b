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
// Reference: AMG/Class_AMG.m:86-109, AMG/MG_Vcycle.m:12-45, AMG/MG_Wcycle.m:13-46, PCG.m:68-87.
#pragma once

static constexpr int RB_NMAX = 8 * BT;                 // rows of level 1
static constexpr int RB_HALF = 4 * BT;                 // rows of a block of level 1 / of level 2
static constexpr int RB_GRAN = RB_NMAX;                // granules per hand-off buffer
static constexpr size_t RB_LDS_BYTES =
    sizeof(double) * ((size_t)2 * RB_NMAX + 3 * RB_HALF + 2 * RES_WAVES + 2 * RES_WAVES + 16 * RES_WAVES + 16);

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
};

__device__ __forceinline__ void rb_publish(__amdgpu_buffer_rsrc_t rs, unsigned seq, int gidx, double v) {
    __builtin_amdgcn_raw_buffer_store_b128(res_pack(v, seq), rs, (int)(seq & 1) * (RB_GRAN * 16) + gidx * 16, 0,
                                           16 /* sc1: write-through */);
}
template <int NJ>
__device__ __forceinline__ bool rb_sweep(__amdgpu_buffer_rsrc_t rs, unsigned seq, int n, bool dead, unsigned* tmo,
                                         double (&v)[NJ], int pollsleep) {
    const int base = (int)(seq & 1) * (RB_GRAN * 16);
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
template <int KE2>
__global__ __launch_bounds__(BT, 2) void k_resident_big(const ResBigDesc D, const double* __restrict__ bvec,
                                                        double* xg, double* out, int fixed_cycles) {
    extern __shared__ __attribute__((aligned(16))) char rb_smem[];
    double* sm = reinterpret_cast<double*>(rb_smem);
    const char* smb = rb_smem;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, G = gridDim.x;
    const int nf = D.nf, nc = D.nc, N1 = nf + nc, N2 = D.N2;
    // LDS map (doubles); E2 is a gather target of the register rows: below 64 KB
    constexpr int oE2 = 0, oTU = RB_HALF, oP3C = 2 * RB_HALF, oE1S = 3 * RB_HALF, oXS = oE1S + RB_NMAX;
    constexpr int oRED = oXS + RB_NMAX, oPUB = oRED + 2 * RES_WAVES, oOWN = oPUB + 2 * RES_WAVES;
    int* fail = reinterpret_cast<int*>(sm + oOWN + 16 * RES_WAVES);
    double* red = sm + oRED;

    // ---- rows of this wave ------------------------------------------------------------------
    const int loF = (int)(((long long)b * nf) / G), hiF = (int)(((long long)(b + 1) * nf) / G);
    const int loC = (int)(((long long)b * nc) / G), hiC = (int)(((long long)(b + 1) * nc) / G);   // C index (0-based)
    const int rowF = loF + w, rowCi = loC + w;
    const bool vF = rowF < hiF, vC = rowCi < hiC;   // level-2 row of the wave = its C node
    const int rF = vF ? rowF : 0, rCi = vC ? rowCi : 0;
    // level-2 row slice -> registers; mask bits of the F row (over C nodes) and of the C row (over F rows)
    unsigned c2[KE2 / 2];
    double a2[KE2];
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
        res_load_slice<KE2>(L2, rCi, vC, lane, c2, a2);
    }
    unsigned bitsF = 0, bitsC = 0;   // bit q <-> entry lane + 64 q
    for (int q0 = 0; q0 < 32; q0 += 8) {   // eight words of each row per burst
        unsigned long long wf[8], wc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            wf[q] = D.fbits[(size_t)rF * D.nwf + (q0 + q < D.nwf ? q0 + q : 0)];
            wc[q] = D.cbits[(size_t)rCi * D.nwc + (q0 + q < D.nwc ? q0 + q : 0)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bitsF |= (vF && q0 + q < D.nwf) ? (unsigned)((wf[q] >> lane) & 1ull) << (q0 + q) : 0u;
            bitsC |= (vC && q0 + q < D.nwc) ? (unsigned)((wc[q] >> lane) & 1ull) << (q0 + q) : 0u;
        }
    }
    // constants of the own rows (LDS: a register pair each would stay live for the whole solve)
    if (lane == 0) {
        sm[oOWN + 0 * RES_WAVES + w] = D.diag1[rF];
        sm[oOWN + 1 * RES_WAVES + w] = D.dinv1[rF];
        sm[oOWN + 2 * RES_WAVES + w] = bvec[rF];
        sm[oOWN + 3 * RES_WAVES + w] = D.Axi1[rF];
        sm[oOWN + 4 * RES_WAVES + w] = D.alpha[rF];
        sm[oOWN + 5 * RES_WAVES + w] = D.rho[rF];
        sm[oOWN + 6 * RES_WAVES + w] = D.diag1[nf + rCi];
        sm[oOWN + 7 * RES_WAVES + w] = D.dinv1[nf + rCi];
        sm[oOWN + 8 * RES_WAVES + w] = bvec[nf + rCi];
        sm[oOWN + 9 * RES_WAVES + w] = D.Axi1[nf + rCi];
        sm[oOWN + 10 * RES_WAVES + w] = D.beta[rCi];
        sm[oOWN + 11 * RES_WAVES + w] = D.diag2[rCi];
        sm[oOWN + 12 * RES_WAVES + w] = D.dinv2[rCi];
        sm[oOWN + 13 * RES_WAVES + w] = D.Axi2[rCi];
    }
#define RB_dgF sm[oOWN + 0 * RES_WAVES + w]
#define RB_dvF sm[oOWN + 1 * RES_WAVES + w]
#define RB_bF sm[oOWN + 2 * RES_WAVES + w]
#define RB_axF sm[oOWN + 3 * RES_WAVES + w]
#define RB_alF sm[oOWN + 4 * RES_WAVES + w]
#define RB_rhF sm[oOWN + 5 * RES_WAVES + w]
#define RB_dgC sm[oOWN + 6 * RES_WAVES + w]
#define RB_dvC sm[oOWN + 7 * RES_WAVES + w]
#define RB_bC sm[oOWN + 8 * RES_WAVES + w]
#define RB_axC sm[oOWN + 9 * RES_WAVES + w]
#define RB_btC sm[oOWN + 10 * RES_WAVES + w]
#define RB_dg2 sm[oOWN + 11 * RES_WAVES + w]
#define RB_dv2 sm[oOWN + 12 * RES_WAVES + w]
#define RB_ax2 sm[oOWN + 13 * RES_WAVES + w]
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
        for (int j = tid; j < N2; j += BT) {
            double v = 0.0;
            for (int t = D.P3.rp[j]; t < D.P3.rp[j + 1]; ++t)
                if (D.P3.ci[t] == 0) v = D.P3.va[t];
            sm[oP3C + j] = v;
            pk += D.Axi2[j] * v;
        }
        for (int t = D.A3.rp[0]; t < D.A3.rp[1]; ++t)
            if (D.A3.ci[t] == 0) h33 = D.A3.va[t];
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
    // x: scaled gather copy for everybody, the own rows' values as wave scalars
    for (int j = tid; j < N1; j += BT) {
        const double sc = j < nf ? D.alpha[j] : D.beta[j - nf];
        sm[oXS + j] = sc * xg[j];
        sm[oE1S + j] = 0.0;
    }
    for (int j = tid; j < N2; j += BT) sm[oE2 + j] = 0.0;
    double xF = vF ? xg[rF] : 0.0, xC = vC ? xg[nf + rCi] : 0.0;
    double rFv = 0.0, rCv = 0.0, eF = 0.0, eC = 0.0, r2v = 0.0, e2v = 0.0;
    if (tid == 0) *fail = 0;
    __syncthreads();

    const auto rs = __builtin_amdgcn_make_buffer_rsrc(D.gran, 0, 2 * RB_GRAN * 16, 0x00020000);
    unsigned seq = 0;
    bool dead = false;

    // sum of the LDS vector at `off` over the set bits of `bits` (entry lane + 64 q <-> bit q)
    auto masked_sum = [&](unsigned bits, int off, int n) __attribute__((always_inline)) {
        double s0 = 0.0, s1 = 0.0;
        const int nq = (n + 63) >> 6;
        for (int q0 = 0; q0 < nq; q0 += 8) {   // eight gathers in flight (all 32 at once spilled registers)
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = lane + 64 * (q0 + q);
                x[q] = sm[off + (j < n ? j : 0)];
            }
            const unsigned bq = bits >> q0;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                s0 += ((bq >> q) & 1u) ? x[q] : 0.0;
                s1 += ((bq >> (q + 1)) & 1u) ? x[q + 1] : 0.0;
            }
        }
        return wave_sum(s0 + s1);
    };

    // hand-off: barrier, wave 0 publishes sm[oPUB ..] (block A: granules gA.., cA rows; block B), sweep of
    // n granules, STORE(j, v) per granule of the thread, block sums of p0 / p1 on request
#define RB_HANDOFF(NJ, n, gA, cA, gB, cB, STORE, want_sums, t0, t1)                                  \
    do {                                                                                           \
        double hv_[NJ];                                                                            \
        ++seq;                                                                                     \
        __syncthreads();                                                                           \
        if (w == 0) {                                                                              \
            const int l8_ = lane & (RES_WAVES - 1);                                                \
            const bool second_ = lane >= RES_WAVES;                                                \
            if (lane < 2 * RES_WAVES && l8_ < (second_ ? (cB) : (cA)) &&                           \
                !(seq == D.dbg_skip_seq && b == G - 1))                                            \
                rb_publish(rs, seq, (second_ ? (gB) : (gA)) + l8_, sm[oPUB + lane]);               \
        }                                                                                          \
        for (int ps_ = 0; ps_ < D.presleep; ++ps_) __builtin_amdgcn_s_sleep(1);                    \
        if (rb_sweep<NJ>(rs, seq, (n), dead, D.tmo, hv_, D.pollsleep)) {                            \
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
                red[w] = p0;                                                                       \
                if ((want_sums) > 1) red[RES_WAVES + w] = p1;                                      \
            }                                                                                      \
        }                                                                                          \
        __syncthreads();                                                                           \
        if (want_sums) {                                                                           \
            t0 = res_red8(red);                                                                    \
            if ((want_sums) > 1) t1 = res_red8(red + RES_WAVES);                                   \
        }                                                                                          \
        if (*fail) dead = true;                                                                    \
    } while (0)
    // exchange of K <= 2 partial sums per workgroup: t0 / t1 = their totals (same order everywhere)
#define RB_PARTIALS(K, v0, v1, t0, t1)                                                             \
    do {                                                                                           \
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
        if (lane == 0) {
            red[w] = vA;
            red[RES_WAVES + w] = vB;
        }
        __syncthreads();
        tA = res_red8(red);
        tB = res_red8(red + RES_WAVES);
        __syncthreads();
    };

    // r = b - A x on the own rows, ||r|| and 1'r by partial sums; E1 := 0            Class_AMG.m:89,96,103
    auto top = [&]() __attribute__((always_inline)) {
        const double sF = masked_sum(bitsF, oXS + nf, nc), sC = masked_sum(bitsC, oXS, nf);
        rFv = vF ? RB_bF - (RB_dgF * xF - RB_alF * sF) : 0.0;
        rCv = vC ? RB_bC - (RB_dgC * xC - RB_btC * sC) : 0.0;
        eF = eC = 0.0;
        double q2 = 0.0, q1 = 0.0;
        own_pair_sum((vF ? rFv * rFv : 0.0) + (vC ? rCv * rCv : 0.0), (vF ? rFv : 0.0) + (vC ? rCv : 0.0), q2, q1);
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
        const bool valid = frows ? vF : vC;
        double s = 0.0;
        if (!(ezero && first)) s = frows ? masked_sum(bitsF, oE1S + nf, nc) : masked_sum(bitsC, oE1S, nf);
        const double eo = ezero ? 0.0 : (frows ? eF : eC);
        const double ae = (frows ? RB_dgF : RB_dgC) * eo - (frows ? RB_alF : RB_btC) * s;        // (A e)_row
        const double g_i = (frows ? rFv : rCv) - ae - (frows ? RB_axF : RB_axC) * c1;
        const double wv = eo + (frows ? RB_dvF : RB_dvC) * g_i;
        const int blk0 = frows ? 0 : nf, nblk = frows ? nf : nc;
        const int g0 = frows ? loF : loC, cnt = frows ? hiF - loF : hiC - loC;
        if (first) {
            if (lane == 0) sm[oPUB + w] = wv;
            if (frows) eF = valid ? wv : 0.0; else eC = valid ? wv : 0.0;
            RB_HANDOFF(4, nblk, g0, cnt, 0, 0,
                       {
                           sm[oE1S + blk0 + j] = (frows ? sF4[u_] : sC4[u_]) * v;
                           p0 += (frows ? aF4[u_] : aC4[u_]) * v;
                       },
                       (nsp ? 1 : 0), afirst, dum1);
        } else {
            const double cc = c1;
            const double en_own = wv + cc;
            if (lane == 0) sm[oPUB + w] = en_own;
            if (frows) { eF = valid ? en_own : 0.0; eC = vC ? eC + cc : 0.0; } else { eC = valid ? en_own : 0.0; eF = vF ? eF + cc : 0.0; }
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
        half1(!post, true, ezero);    // pre: F rows first (Rk{1}); post: C rows first (Rk{1}')
        half1(post, false, ezero);
    };

    // weighted-Jacobi sweep on level 2                                   MG_Vcycle.m:15-21; Class_AMG.m:84
    auto sweep2 = [&](bool ezero) __attribute__((always_inline)) {
        double s = 0.0, eo = 0.0;
        if (!ezero) {
            s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2, a2, smb));
            eo = e2v;
            s += RB_dg2 * eo;
        }
        const double g_i = r2v - s - RB_ax2 * c2s;
        const double en = eo + RB_dv2 * g_i + c2s;
        if (lane == 0) sm[oPUB + w] = en;
        e2v = vC ? en : 0.0;
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

    // one visit of level 2 and of the one-row tail below it
    auto visit2 = [&](bool keep) __attribute__((always_inline)) {
        const int nu = D.nu;
        for (int s = 0; s < nu; ++s) sweep2(!keep && s == 0);
        {   // rr = r - A e on the own row; r_3 = P3' rr by partial sums                  MG_Vcycle.m:27
            const double s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2, a2, smb)) + RB_dg2 * e2v;
            const double rr = vC ? r2v - s : 0.0;
            double part = 0.0, dumA = 0.0;
            own_pair_sum(vC ? sm[oP3C + rCi] * rr : 0.0, 0.0, part, dumA);
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
            // e_2 += P3 e_3 on everybody's copy and on the own scalar; 1'(r - A e) moves by -d kappa
            for (int j = tid; j < N2; j += BT) sm[oE2 + j] = sm[oE2 + j] + sm[oP3C + j] * d;
            e2v = vC ? e2v + sm[oP3C + rCi] * d : 0.0;
            xig2 = xig2 - d * kappa;
            c2s = nsp ? xig2 / xx2 : 0.0;
            __syncthreads();
        }
        for (int s = 0; s < nu; ++s) sweep2(false);
    };

    // MG_Vcycle / MG_Wcycle from level 1 down; the correction ends in E1S / eF, eC
    auto cycle = [&]() __attribute__((always_inline)) {
        const int nu = D.nu;
        for (int s = 0; s < nu; ++s) sweep1(false, s == 0);
        {   // rr = r - A e; the F part goes out scaled by rho (the restriction's operand), the C part stays
            const double sF = masked_sum(bitsF, oE1S + nf, nc), sC = masked_sum(bitsC, oE1S, nf);
            const double rrF = rFv - (RB_dgF * eF - RB_alF * sF);
            const double rrC = rCv - (RB_dgC * eC - RB_btC * sC);
            if (lane == 0) sm[oPUB + w] = RB_rhF * rrF;
            RB_HANDOFF(4, nf, loF, hiF - loF, 0, 0, { sm[oTU + j] = v; }, 0, dum0, dum1);
            // r_2 = P' rr: row c of P' is [W(:,c)', 1 at the C node]; E2 := 0; 1'r_2 by partial sums
            const double s2 = RB_btC * masked_sum(bitsC, oTU, nf) + rrC;
            r2v = vC ? s2 : 0.0;
            e2v = 0.0;
            double part = 0.0, dumA = 0.0;
            own_pair_sum(r2v, 0.0, part, dumA);
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
            const double sF = RB_rhF * masked_sum(bitsF, oTU, N2);
            const double nF = eF + sF, nC = eC + e2v;
            eF = vF ? nF : 0.0;
            eC = vC ? nC : 0.0;
            // the new iterate goes out block by block (a thread's constants are per block)
            double asF = 0.0, asC = 0.0;
            if (lane == 0) sm[oPUB + w] = nF;
            RB_HANDOFF(4, nf, loF, hiF - loF, 0, 0,
                       {
                           sm[oE1S + j] = sF4[u_] * v;
                           p0 += aF4[u_] * v;
                       },
                       (nsp ? 1 : 0), asF, dum1);
            if (lane == 0) sm[oPUB + w] = nC;
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
        for (int j = tid; j < N1; j += BT) sm[oXS + j] = sm[oXS + j] + sm[oE1S + j];
        xF = xF + eF;
        xC = xC + eC;
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
    if (lane == 0) {   // every workgroup writes its own rows of x
        if (vF) xg[rF] = xF;
        if (vC) xg[nf + rCi] = xC;
    }
    if (writer) {
        const unsigned anytmo = __hip_atomic_load(D.tmo, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        out[0] = (double)it;
        out[1] = rel_res;
        out[2] = res0;
        out[3] = (dead || anytmo != 0) ? 1.0 : 0.0;
        out[4 + 2 * (maxit + 2) - 1] = (double)seq;   // hand-offs of this launch (see k_resident)
    }
#undef RB_PARTIALS
#undef RB_HANDOFF
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
}
