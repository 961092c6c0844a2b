// Level-resident solve kernel (included by ipd_cycle.hip only).
//
// The multi-launch path pays one kernel boundary plus 2-3 dependent memory round trips per
// half sweep: 4.6-5.0 us per k_smooth launch at m=n=1024, rho=1, where the 12.7 MB a launch
// streams would take 1.6 us at HBM peak (profiles/r1_kernel_stats.csv).  The whole hierarchy of
// that regime (levels 2048 / 1024 / 1: 21 MB + 10.5 MB of padded rows) fits in the chip's
// registers, so this kernel keeps it there for the WHOLE Class_AMG solve:
//
//   * G <= 256 workgroups of 512 threads, one per CU, all co-resident; a wave owns one row of
//     each row block (level 1: F rows and C rows of the bigraph Gauss-Seidel, level 2: Jacobi),
//     lane l holds entries l, l+64, ... of the padded row (16-bit column, fp64 value) in VGPRs;
//   * every workgroup keeps the full vectors of both levels (x, e, r, r - A e, A*1) in LDS, so a
//     row dot product is LDS gathers + one DPP wave sum, no memory traffic at all;
//   * the only global traffic is the hand-off of each half sweep's result: a row's new value is
//     published as ONE 16-byte write-through (sc1) store of two self-tagged 8-byte granules
//     {lo, tag, hi, tag} (MI355X guide, Guideline 16 R2: the data is the flag) and every
//     workgroup sweeps all granules of the step with sc1 loads until every tag carries the
//     step number.  Two buffers by step parity: a workgroup writes step t+2 only after it has
//     seen all of step t+1, which every workgroup publishes only after it has read all of t
//     (every workgroup owns at least one row of every block, so "all of t+1" includes everyone).
//     tools/ubench_exchange.hip prices the step at 1.9 us (G = 128) against 4.6-5.0 us for the
//     launch it replaces (profiles/r2_ubench_exchange.txt);
//   * the kernel-space scalar c = 1'(r - A e)/xx of the next sweep is reduced in the same sweep
//     phase that stores the new iterate (no extra barrier), the transfers to and from level 2 walk
//     the CSR rows of P'/P from L2 (twice per cycle), and the tail level (<= 64 rows, 1 row in
//     the dense regime) is solved redundantly by every workgroup, so it needs no hand-off;
//   * the stationary iteration and its stopping rules (Class_AMG.m:86-109) run in the kernel:
//     every workgroup forms the same norm from the same LDS copy in the same order and so takes
//     the same decision; one launch and one read-back per solve.
//
// Every spin is bounded: a workgroup that gives up raises `tmo`, every later sweep of every
// workgroup gives up at once, and the host falls back to the multi-launch path.
//
// Arithmetic per row is the multi-launch kernels' (phase_smooth / phase_resid / phase_xfer /
// phase_top), only the order inside a row's dot product differs (lane-strided entries).
#pragma once

typedef unsigned int res_v4u __attribute__((ext_vector_type(4)));

static constexpr int RES_WAVES = BT / 64;     // row slots per block and workgroup (one wave per row)
static constexpr int RES_TAIL_MAX = 64;       // rows of the redundantly solved tail level
static constexpr int RES_NMAX = 4 * BT;        // rows per level (fixed LDS slots)
static constexpr int RES_GRAN_MAX = RES_NMAX;  // granules per hand-off buffer
static constexpr size_t RES_LDS_BYTES = sizeof(double) * ((size_t)9 * RES_NMAX + RES_NMAX / 2 + 3 * RES_TAIL_MAX + 16 * RES_WAVES + 12);
static constexpr unsigned RES_SPIN_MAX = 1u << 18;
static constexpr int RES_P3_LD = 1152;         // row stride of ResDesc::p3rows: 512 + 512 + 128
static constexpr int RES_P4_SEG = 128;         // ... of ResDesc::p4rows: 128 + 128 + 64
static constexpr int RES_P4_LD = 2 * RES_P4_SEG + 64;

struct ResLevelDesc {
    int N, nf, S;
    const unsigned short* pci;
    const double* pva;
    const double* diag;
    const double* dinv;
    const double* Axi;
    const double* xx;
};
struct ResCsr {
    const int* rp;
    const int* ci;
    const double* va;
};
struct ResDesc {
    ResLevelDesc L1, L2;
    ResLevelDesc L3;  // third resident level (Jacobi, <= BT rows), `three` != 0 only
    ResCsr Pt2, P2;   // level 1 <-> 2: restriction rows (N2 x N1), prolongation rows (N1 x N2)
    ResCsr Pt3, P3;   // level 2 <-> 3 (the tail level of a three-level hierarchy)
    ResCsr Pt4, P4;   // level 3 <-> 4 (`three` only: the remote tail is rooted at level 4)
    ResCsr A3;        // tail operator (CSR), local tail only
    ResCsr A4;        // ... when level 3 is resident too and level 4 is the (local) tail
    int three;        // levels 1-3 resident (hierarchies whose level 3 does not fit the tail's LDS)
    int tail_root;    // remote tail: 3 or 4
    // Level 3 in polynomial form (template argument KE3 == 1, remote tail only; pack_bpoly with rows):
    // row i < N3 of p3rows is [M2a | M1](i,:) and row N3 + c is the restriction row c stacked on it,
    // entries 0..N3-1 applied to r_3, 512..512+N3-1 to e_3 and (rows < N3) 1024..1024+N4-1 = (M1 P4)(i,:)
    // to the tail's e_4 (row stride RES_P3_LD); p3w their factors of 1'r_3.  A visit of level 3 is then
    // THREE hand-offs (e' and the restricted residual; the tail's answer e_4 -- the tail workgroup does not
    // prolongate in this mode --; e'') instead of thirteen (ten sweeps, the residual, the restriction, the
    // prolongation).
    const double* p3rows;
    const double* p3w;
    // Level 2 in polynomial form, COMPOSED over a whole visit (template argument POLY2; three levels with a
    // one-row tail, V cycle -- the metric's workload; ipd_amg_attach_level2_poly).  The nu pre-sweeps, the
    // residual, the restriction to the one-row tail, its prolongation and the nu post-sweeps of a visit that
    // starts from e = 0 (AMG/MG_Vcycle.m:14-41) are the affine map
    //     e_2 = B r_2 + wB (1'r_2) + mp e_3,   e_3 = PCG(h33, s'r_2 + ws (1'r_2)),
    // B = M1 M2a + M2a, wB = M1 w + w, mp = M1 p3, s = P3' - (P3'A) M2a (pack_bpoly's operators, one more dense
    // product): row i of p2rows holds B(i,:) at [p2seg ..), mp_i at [2 p2seg], row N2 holds s; p2w = [wB; ws].
    // A visit of level 2 is then ONE hand-off instead of ten.
    const double* p2rows;
    const double* p2w;
    int p2seg, p2ld;
    // Level 4 in polynomial form in the resident workgroups as well (round 4, POLY3 hierarchies of six levels
    // and more; N5 > 0): row b < N4 of [M2a | M1] and the restriction row N4 + b (b < N5) per workgroup, fetched
    // from L2 at every pass (row stride RES_P4_LD: [Mr (128) | Me (128) | Mc (64)]); the restricted residual of
    // level 3 goes to EVERYBODY in the ack granules of level 3's hand-off, and the tail workgroup is rooted
    // at level 5 (ResDesc::sub then holds levels 5..J, tail_root = 5).  With the tail at level 4 its two legs
    // per visit of level 3 were 50 us of serial work on the late Newton systems (4 per W cycle: 200 of 296 us).
    int N5;
    const double* p4rows;
    const double* p4w;
    // Remote tail (hierarchies with more than three levels): workgroup gridDim.x - 1 holds the LDS
    // image of the single-workgroup sub-cycle rooted at level 3 (k_subcycle's code and data) and
    // serves the visits of everything below level 2: the other workgroups hand it r_3 = P3' rr_2
    // (tin, Nt granules) and receive the prolongated correction P3 e_3 (tout, N2 granules).
    int remote;
    const SolveDesc* sub;     // image of levels 3..J (build_image), NULL without remote tail
    unsigned char* tin;       // 2 x RES_GRAN_MAX granules by visit parity
    unsigned char* tout;
    unsigned* tctl;           // [0] != 0: the solve is over, the tail workgroup leaves
    // Level 1 <-> 2 transfers from the active-set bit mask (amg_attach_maskop, three-level hierarchies
    // with bigraph transfers only): W(j,i) = s_ij beta_i rho_j (AMG/transfer.m:19-25 on Hybrid_AMG's
    // rescaled operator), so a row of P' or P is 1 bit per entry -- 16 bits per lane, held in one
    // register -- against a pre-scaled LDS vector, instead of a CSR row walked from L2 (12.6 MB per
    // cycle, 5 us of a 72 us cycle).  xm != 0: fbits / cbits as MaskOp (row-major 64-bit words).
    int xm, xm_nwf, xm_nwc;
    const unsigned long long* xm_fbits;
    const unsigned long long* xm_cbits;
    const double* xm_beta;   // nc: the C node's factor
    const double* xm_rho;    // nf: the F row's factor (1 / row sum with isnsp)
    int localfirst;   // zero-start first sweeps formed locally (see k_resident); 0: handed off like the rest
    int tail_bm;      // the launch's dynamic LDS has room for the tail image's operator copy (SolveDesc::bm_src)
    int wident;       // P = [W; I] verified (k_res_check_ident): identity entries are added, not walked
    int Nt;           // rows of the tail level (local tail) or of the remote tail's root level
    int nu, isnsp, wcycle, anycycle, maxit;
    double retol;
    long long pcg_maxit;
    unsigned char* gran0;
    unsigned char* gran1;
    int presleep;       // s_sleep(1) repetitions between a publish and the first poll of its sweep
    int pollsleep;      // ... and between two polls
    unsigned* tmo;      // [0] != 0: a bounded spin gave up (value = step number)
    long long* dbg;     // optional stamps (diagnostic build of the bench): see k_resident
    unsigned dbg_skip_seq;   // test hook (IPD_RES_DEBUG_SKIP_PUBLISH=<step>): the last workgroup omits its
                             // publish of that step, so every sweep of the step gives up; 0 = off
};

// one fp64 value as two self-tagged 8-byte granules
__device__ __forceinline__ res_v4u res_pack(double v, unsigned tag) {
    res_v4u g;
    g.x = (unsigned)__double2loint(v);
    g.y = tag;
    g.z = (unsigned)__double2hiint(v);
    g.w = tag;
    return g;
}

__device__ __forceinline__ void res_publish(__amdgpu_buffer_rsrc_t rs, unsigned seq, int gidx, double v) {
    __builtin_amdgcn_raw_buffer_store_b128(res_pack(v, seq), rs,
                                           (int)(seq & 1) * (RES_GRAN_MAX * 16) + gidx * 16, 0,
                                           16 /* sc1: write-through */);
}

// Sweeps the n granules of hand-off `seq` (n <= NJ*BT); granule j goes to thread j % BT, pass
// u = j / BT.  Returns the values in v[u] and true when the bounded spin gave up; the caller
// stores the values after the barrier it places (all waves have then finished the step's
// reads of the vectors that are about to change).
template <int NJ>
__device__ __forceinline__ bool res_sweep(__amdgpu_buffer_rsrc_t rs, unsigned seq, int n, bool dead,
                                          unsigned* tmo, double (&v)[NJ], int pollsleep = 1) {
    const int base = (int)(seq & 1) * (RES_GRAN_MAX * 16);
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
                ((spins & 255) == 255 &&
                 __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                bad = true;
                break;
            }
            for (int ps = 0; ps < pollsleep; ++ps) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    }
    return bad;
}

// The same sweep for the slow hand-offs of the remote tail (tens of microseconds): long sleeps
// between polls, no give-up count -- it ends when every tag carries `seq` (returns 0), when the
// time-out word is raised (1) or when the exit word is (2; tail workgroup only, ctl may be NULL).
template <int NJ>
__device__ __forceinline__ int res_wait_slow(__amdgpu_buffer_rsrc_t rs, unsigned seq, int n,
                                             const unsigned* tmo, const unsigned* ctl, double (&v)[NJ]) {
    const int base = (int)(seq & 1) * (RES_GRAN_MAX * 16);
    const int j0 = threadIdx.x;
    for (unsigned spins = 0;; ++spins) {
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
        if (__all(ok)) return 0;
        if ((spins & 15) == 15) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 1;
            if (ctl && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return 2;
            if (spins > (1u << 21)) return 1;   // ~1 s: a sub-cycle leg takes 0.02-0.2 ms
        }
        __builtin_amdgcn_s_sleep(16);
        asm volatile("" ::: "memory");
    }
}

// lane-strided slice of one padded row: entries lane, lane+64, ... ; the 16-bit columns are kept
// as LDS byte offsets (8*column <= 16376), two per register
template <int KE>
__device__ __forceinline__ void res_load_slice(const ResLevelDesc& L, int row, bool valid, int lane,
                                               unsigned (&c)[KE / 2], double (&a)[KE]) {
    unsigned cc[KE];
#pragma unroll
    for (int q = 0; q < KE; ++q) {
        const int e = lane + 64 * q;
        const bool ok = valid && e < L.S;
        const size_t off = ok ? (size_t)row * L.S + e : 0;
        const unsigned short cj = L.pci[off];
        const double aa = L.pva[off];
        cc[q] = ok ? 8u * cj : 0u;
        a[q] = ok ? aa : 0.0;
    }
#pragma unroll
    for (int q = 0; q < KE / 2; ++q) c[q] = cc[2 * q] | (cc[2 * q + 1] << 16);
}

// row dot product against the LDS vector at byte offset OFFB (a compile-time constant below
// 64 KB: one ds_read_b64 with an immediate offset per entry)
template <int KE, int OFFB>
__device__ __forceinline__ double res_rowdot(unsigned (&c)[KE / 2], const double (&a)[KE],
                                             const char* smb) {
    double y[KE];
#pragma unroll
    for (int q = 0; q < KE / 2; ++q) {
        // opaque: keeps the unpacked offsets from being hoisted out of the cycle loops (they
        // would cost a register per matrix entry for the whole kernel)
        asm volatile("" : "+v"(c[q]));
        const unsigned lo = c[q] & 0xffffu, hi = c[q] >> 16;
        y[2 * q] = *reinterpret_cast<const double*>(smb + OFFB + lo);
        y[2 * q + 1] = *reinterpret_cast<const double*>(smb + OFFB + hi);
    }
    // four partial sums (entries q, q+4, ...): the dependent add chain is a quarter as long
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int q = 0; q < KE; q += 4) {
        s0 += a[q] * y[q];
        s1 += a[q + 1] * y[q + 1];
        s2 += a[q + 2] * y[q + 2];
        s3 += a[q + 3] * y[q + 3];
    }
    return (s0 + s1) + (s2 + s3);
}

// CSR row of a transfer operator (global, L2-resident) against an LDS vector, one wave per row
__device__ __forceinline__ double res_csr_rowdot(const ResCsr& M, int e0, int e1, int lane,
                                                 const double* sm, int off) {
    double s = 0.0;
    for (int t = e0 + lane; t < e1; t += 64 * 8) {   // 8 entries per lane in flight: a 1025-entry row = 3 trips
        int jj[8];
        double aa[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int tt = t + 64 * u;
            const int tc = tt < e1 ? tt : e0;
            jj[u] = M.ci[tc];
            aa[u] = M.va[tc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (t + 64 * u < e1) ? aa[u] * sm[off + jj[u]] : 0.0;
    }
    return wave_sum(s);
}

// *bad != 0 unless the level 1 <-> 2 transfers have the bigraph form P = [W; I] (AMG/transfer.m:19-25)
// entry for entry: every row of P' (level-2 row c) ends with the identity entry (column nf + c, value
// 1) and row nf + c of P is that identity entry alone.  The kernel then adds the identity parts
// itself: a 1025-entry row of P' is two 512-entry trips instead of three, and the C rows of P cost
// no trip at all.
__global__ void k_res_check_ident(int nf, int N2, ResCsr P, ResCsr Pt, int* __restrict__ bad) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N2; c += gridDim.x * blockDim.x) {
        const int e1 = Pt.rp[c + 1], p0 = P.rp[nf + c];
        const bool ok = e1 > Pt.rp[c] && Pt.ci[e1 - 1] == nf + c && Pt.va[e1 - 1] == 1.0 &&
                        P.rp[nf + c + 1] - p0 == 1 && P.ci[p0] == c && P.va[p0] == 1.0;
        if (!ok) atomicOr(bad, 1);
    }
}

// rho of the mask-form transfers and the check of P against W(j,i) = s_ij beta_i rho_j (one wave per F row
// j: its row of P holds exactly the row's mask entries, in column order, each within 1e-12 of the form)
__global__ __launch_bounds__(256) void k_res_xmask_rho(int nf, int nc, int isnsp,
                                                       const unsigned long long* __restrict__ fbits, int nwf,
                                                       const double* __restrict__ alpha,
                                                       const double* __restrict__ beta,
                                                       const double* __restrict__ diag, const int* __restrict__ prp,
                                                       const int* __restrict__ pci, const double* __restrict__ pva,
                                                       double* __restrict__ rho, int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= nf) return;
    double sb = 0.0;
    int cnt = 0;
    for (int w = 0; w < nwf; ++w) {
        const unsigned long long bits = fbits[(size_t)j * nwf + w];
        const int i = w * 64 + lane;
        if ((bits >> lane) & 1ull) sb += beta[i];
        cnt += __popcll(bits);
    }
    sb = wave_sum(sb);
    const double r = isnsp ? 1.0 / sb : alpha[j] / diag[j];
    if (lane == 0) rho[j] = r;
    bool wrong = (prp[j + 1] - prp[j]) != cnt;
    for (int t = prp[j] + lane; t < prp[j + 1] && !wrong; t += 64) {
        const int i = pci[t];
        const bool bit = i >= 0 && i < nc && ((fbits[(size_t)j * nwf + (i >> 6)] >> (i & 63)) & 1ull);
        const double ref = beta[i < nc ? i : 0] * r;
        if (!bit || !(fabs(pva[t] - ref) <= 1e-12 * fabs(ref))) wrong = true;
    }
    if (wrong) atomicExch(bad, 1);
}

// Block sums of up to two per-thread partials through red[0..2*RES_WAVES): the caller has
// written red[w] / red[RES_WAVES + w] before the barrier that precedes this call.
__device__ __forceinline__ double res_red8(const double* red) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < RES_WAVES; ++k) t += red[k];
    return t;
}

// The tail workgroup of a remote-tail launch: k_subcycle's body as a server.  It loads the LDS
// image of levels 3..J once, then for every visit waits for r_3 (D.Nt <= BT granules in tin), runs
// the V or W sub-cycle rooted at level 3 out of LDS (both legs of MG_Wcycle.m:28-30 when level 3 is
// not the coarsest), and publishes P3 * e_3 for all rows of level 2 (tout).  It leaves when the
// other workgroups raise the exit word (end of the solve) or the time-out word.
__device__ __forceinline__ void res_tail_workgroup(const ResDesc& D, char* dyn_raw, PhaseLds* lds,
                                                   double* red, double* blkpart, int* stat) {
    const int tid = threadIdx.x, w = tid >> 6;
    SolveDesc* LD = sol_load_image(D.sub, dyn_raw);
    SolveCtx c;
    c.D = LD;
    c.lds = lds;
    c.red = red;
    c.xs = reinterpret_cast<double*>(dyn_raw);
    c.swapmask = 0;
    c.zeromask = 0;
    c.part = blkpart;
    c.sumr = blkpart + 48;
    c.dbg = nullptr;
    c.bm_lds = 0;
    if (D.tail_bm && LD->bm_bytes) {   // one block-wide level's operator into LDS for the whole solve (SolveDesc::bm_src)
        const uint4* src = reinterpret_cast<const uint4*>(LD->bm_src);
        uint4* dst = reinterpret_cast<uint4*>(dyn_raw + LD->bm_off);
        for (int i = tid; i < LD->bm_bytes / 16; i += BT) dst[i] = src[i];
        __syncthreads();
        c.bm_lds = (unsigned)(size_t)(dyn_raw + LD->bm_off);
    }
    const int k0 = D.tail_root, N3 = k0 == 5 ? D.N5 : D.Nt, N2 = k0 == 3 ? D.L2.N : D.L3.N;   // inbox / outbox rows
    const ResCsr& Pout = k0 == 3 ? D.P3 : D.P4;
    const bool two_legs = D.wcycle && k0 < LD->J;
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(D.tin, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(D.tout, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    long long busy = 0;
    for (unsigned tseq = 1;; ++tseq) {
        double v[1];
        const int st = res_wait_slow<1>(rin, tseq, N3, D.tmo, D.tctl, v);
        if ((tid & 63) == 0) stat[w] = st;
        __syncthreads();
        int any = 0;
#pragma unroll
        for (int k = 0; k < RES_WAVES; ++k) any |= stat[k];
        if (any) {                             // uniform: every wave reads the same eight words
            if (D.dbg && tid == 0) D.dbg[9] = busy;   // (diagnostic build of the bench: clocks between a request's arrival and its answer's stores)
            return;
        }
        const long long tb0 = (D.dbg && tid == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0;
        if (tid < N3) LD->L[k0].lv.r[tid] = v[0];
        __syncthreads();
        sol_cycle(c, k0, false);
        __syncthreads();
        if (two_legs) {
            sol_cycle(c, k0, true);
            __syncthreads();
        }
        const double* e3 = sol_e(c, k0);
        const int base = (int)(tseq & 1) * (RES_GRAN_MAX * 16);
        if (D.p3rows) {   // polynomial level 3: its workgroups apply M1 P4 themselves, the answer is e_4
            if (tid < N3)
                __builtin_amdgcn_raw_buffer_store_b128(res_pack(e3[tid], tseq), rout, base + tid * 16, 0, 16 /* sc1 */);
        } else
        for (int j = tid; j < N2; j += BT) {   // e_2 += P e_3 is finished by the receivers     MG_Vcycle.m:31
            double sd = 0.0;
            for (int t = Pout.rp[j]; t < Pout.rp[j + 1]; ++t) sd += Pout.va[t] * e3[Pout.ci[t]];
            __builtin_amdgcn_raw_buffer_store_b128(res_pack(sd, tseq), rout, base + j * 16, 0, 16 /* sc1 */);
        }
        if (D.dbg && tid == 0) busy += (long long)__builtin_amdgcn_s_memtime() - tb0;
        __syncthreads();                       // stat and e3 are rewritten by the next visit
    }
}

// out[0] = it, out[1] = rel_res, out[2] = res0; rel_resk at out[4 ..], rhok at out[4+maxit+2 ..]
// (the layout of k_solve_small; the last slot: hand-offs of the launch).  fixed_cycles > 0: exactly that many loop bodies, no stopping
// rules (bench hook).  dbg (optional, 16 words): [0] shader clocks spent waiting in sweeps by
// workgroup 0, [1] clocks of the whole loop, [2] number of hand-offs, [3] 100 MHz ticks of the loop,
// [4] clocks in the barrier before the publish, [5] in the store phase, [6] in the closing barrier.
template <int KE1, int KE2, int KE3 = 0, bool POLY2 = false>
__global__ __launch_bounds__(BT, 2) void k_resident(const ResDesc D, const double* __restrict__ bvec,
                                                    double* xg, double* out, int fixed_cycles) {
    static_assert(!POLY2 || KE3 == 0, "POLY2: three-level hierarchies");
    constexpr bool THREE = KE3 > 0;
    constexpr bool POLY3 = KE3 == 1;   // level 3 in polynomial form (ResDesc::p3rows)
    constexpr int K3 = (THREE && !POLY3) ? KE3 : 2;
    extern __shared__ __attribute__((aligned(16))) char res_smem[];
    double* sm = reinterpret_cast<double*>(res_smem);
    const char* smb = res_smem;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x, G = gridDim.x - (D.remote ? 1 : 0);   // G: workgroups of levels 1-2
    const int N1 = D.L1.N, N2 = D.L2.N, nf = D.L1.nf, nc = N1 - nf, Nt = D.Nt;
    if (D.remote && b == G) {   // the tail workgroup
        __shared__ PhaseLds tail_lds;
        __shared__ double tail_red[16];
        __shared__ double tail_part[48 + SOLVE_ML + 1];
        __shared__ int tail_stat[RES_WAVES];
        res_tail_workgroup(D, res_smem, &tail_lds, tail_red, tail_part, tail_stat);
        return;
    }
    // LDS map (doubles): fixed slots of RES_NMAX entries, the gather targets in the first 64 KB so
    // that a gather is one ds_read_b64 with an immediate offset (no address arithmetic to hoist)
    constexpr int oX = 0, oE1 = RES_NMAX, oE2 = 2 * RES_NMAX, oRR1 = 3 * RES_NMAX;
    constexpr int oR1 = 4 * RES_NMAX, oAX1 = 5 * RES_NMAX, oR2 = 6 * RES_NMAX, oRR2 = 7 * RES_NMAX;
    constexpr int oAX2 = 8 * RES_NMAX;
    constexpr int oRHO = 9 * RES_NMAX;                 // mask-form transfers: rho of the F rows (nf <= RES_NMAX / 2)
    constexpr int oBETA = oAX2 + RES_NMAX / 2;         // ... beta of the C nodes: upper half of the AX2 slot (!THREE)
    constexpr int oU = oRR2;                           // ... beta .* e_2: lower half of the RR2 slot (free after the visit)
    constexpr int oR3 = 9 * RES_NMAX + RES_NMAX / 2, oE3 = oR3 + RES_TAIL_MAX, oP3 = oE3 + RES_TAIL_MAX;
    constexpr int oRED = oP3 + RES_TAIL_MAX;          // 2*RES_WAVES doubles
    constexpr int oPUB = oRED + 2 * RES_WAVES;        // values the waves publish this step (2 blocks)
    constexpr int oOWN = oPUB + 2 * RES_WAVES;        // 10 scalars of each wave's rows
    int* fail = reinterpret_cast<int*>(sm + oOWN + 10 * RES_WAVES);
    long long* dbg_acc = reinterpret_cast<long long*>(sm + oOWN + 10 * RES_WAVES + 1);   // 8 words
    // entry ranges of this wave's rows of the transfer operators: read once, a walk then starts
    // with its entries instead of a dependent trip for the row pointers
    int* rowp = reinterpret_cast<int*>(sm + oOWN + 10 * RES_WAVES + 12);                 // 12 ints per wave
    // third resident level (THREE): its vectors sit in the upper halves of level 2's slots (N2 <= 1024,
    // N3 <= 512); E3 is a gather target and must lie below 64 KB
    constexpr int oE3L = oE2 + RES_NMAX / 2, oR3L = oRR2 + RES_NMAX / 2, oRR3L = oRR2 + 3 * RES_NMAX / 4;
    // POLY4 (the RR3L region holds 512 doubles, e_4 takes 128): r_4, e_5, partial sums and the rows' factors
    constexpr int oR4L = oRR3L + 128, oE5L = oRR3L + 256, oPS4 = oRR3L + 320;
    constexpr int oAX3L = oAX2 + RES_NMAX / 2;
    const int N3 = THREE ? D.L3.N : 0;
    double* red = sm + oRED;

    // ---- rows of this wave ------------------------------------------------------------------
    const int loF = (int)(((long long)b * nf) / G), hiF = (int)(((long long)(b + 1) * nf) / G);
    const int loC = nf + (int)(((long long)b * nc) / G), hiC = nf + (int)(((long long)(b + 1) * nc) / G);
    const int lo2 = (int)(((long long)b * N2) / G), hi2 = (int)(((long long)(b + 1) * N2) / G);
    const int rowF = loF + w, rowC = loC + w, row2 = lo2 + w;
    const bool vF = rowF < hiF, vC = rowC < hiC, v2 = row2 < hi2;
    const int rF = vF ? rowF : 0, rC = vC ? rowC : 0, r2 = v2 ? row2 : 0;
    const int lo3 = THREE ? (int)(((long long)b * N3) / G) : 0, hi3 = THREE ? (int)(((long long)(b + 1) * N3) / G) : 0;
    const int row3 = lo3 + w;
    const bool v3 = THREE && row3 < hi3;
    const int r3 = v3 ? row3 : 0;

    // ---- matrix slices -> registers (the only read of the matrices in the whole solve) --------
    unsigned cF[KE1 / 2], cC[KE1 / 2], c2[KE2 / 2];
    double aF[KE1], aC[KE1], a2[KE2];
    res_load_slice<KE1>(D.L1, rF, vF, lane, cF, aF);
    res_load_slice<KE1>(D.L1, rC, vC, lane, cC, aC);
    if (POLY2) {
#pragma unroll
        for (int q = 0; q < KE2; ++q) {
            const int e = lane + 64 * q;
            const double bv = D.p2rows[(size_t)r2 * D.p2ld + D.p2seg + (e < N2 ? e : 0)];
            a2[q] = (v2 && e < N2) ? bv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < KE2 / 2; ++q) c2[q] = 0u;
    } else {
        res_load_slice<KE2>(D.L2, r2, v2, lane, c2, a2);
    }
    unsigned c3[K3 / 2];
    double a3[K3];
    if (THREE && !POLY3) res_load_slice<K3>(D.L3, r3, v3, lane, c3, a3);
    // polynomial form: entries tid and 512 + tid of the workgroup's rows lo3..hi3-1 (at most four) and of
    // restriction row b (the remote tail's root level has at most G rows: one per workgroup)
    double m3r[5], m3e[5], m3c[4];
#pragma unroll
    for (int q = 0; q < 5; ++q) m3r[q] = m3e[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) m3c[q] = 0.0;
    if (POLY3) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int row = q < 4 ? lo3 + q : N3 + b;
            const bool ok = (q < 4 ? row < hi3 : b < Nt) && tid < N3;
            if (ok) {
                m3r[q] = D.p3rows[(size_t)row * RES_P3_LD + tid];
                m3e[q] = D.p3rows[(size_t)row * RES_P3_LD + 512 + tid];
            }
            if (q < 4 && row < hi3 && tid < Nt) m3c[q] = D.p3rows[(size_t)row * RES_P3_LD + 1024 + tid];
        }
        if (tid < 5) {
            const int row = tid < 4 ? lo3 + tid : N3 + b;
            sm[oE3 + tid] = (tid < 4 ? row < hi3 : b < Nt) ? D.p3w[row] : 0.0;
        }
        if (D.N5 > 0 && tid < 2) {   // POLY4: the factors of 1'r_4 of row b and of restriction row N4 + b
            const int row = tid == 0 ? b : Nt + b;
            sm[oPS4 + 20 + tid] = (tid == 0 ? b < Nt : b < D.N5) ? D.p4w[row] : 0.0;
        }
    }
    // the rows' own scalars live in LDS (a register pair each would stay live for the whole solve)
    if (lane == 0) {
        sm[oOWN + 0 * RES_WAVES + w] = D.L1.diag[rF];
        sm[oOWN + 1 * RES_WAVES + w] = D.L1.dinv[rF];
        sm[oOWN + 2 * RES_WAVES + w] = bvec[rF];
        sm[oOWN + 3 * RES_WAVES + w] = D.L1.diag[rC];
        sm[oOWN + 4 * RES_WAVES + w] = D.L1.dinv[rC];
        sm[oOWN + 5 * RES_WAVES + w] = bvec[rC];
        sm[oOWN + 6 * RES_WAVES + w] = POLY2 ? D.p2w[r2] : D.L2.diag[r2];                                    // (POLY2: wB)
        sm[oOWN + 7 * RES_WAVES + w] = POLY2 ? D.p2rows[(size_t)r2 * D.p2ld + 2 * D.p2seg] : D.L2.dinv[r2];   // (POLY2: mp)
        sm[oOWN + 8 * RES_WAVES + w] = (THREE && !POLY3) ? D.L3.diag[r3] : 0.0;
        sm[oOWN + 9 * RES_WAVES + w] = (THREE && !POLY3) ? D.L3.dinv[r3] : 0.0;
        rowp[12 * w + 0] = D.Pt2.rp[r2];
        rowp[12 * w + 1] = v2 ? D.Pt2.rp[r2 + 1] - D.wident : D.Pt2.rp[r2];
        rowp[12 * w + 2] = D.P2.rp[rF];
        rowp[12 * w + 3] = vF ? D.P2.rp[rF + 1] : D.P2.rp[rF];
        rowp[12 * w + 4] = D.P2.rp[rC];
        rowp[12 * w + 5] = vC ? D.P2.rp[rC + 1] : D.P2.rp[rC];
        // remote tail: row b + G*w of the restriction to its root level, if there is one
        const int rin = b + G * w;
        const ResCsr& Pin = D.tail_root == 4 ? D.Pt4 : D.Pt3;
        rowp[12 * w + 6] = (D.remote && rin < Nt) ? Pin.rp[rin] : 0;
        rowp[12 * w + 7] = (D.remote && rin < Nt) ? Pin.rp[rin + 1] : 0;
        // third resident level: its own row of P3' (restriction 2 -> 3), this wave's level-2 row of P3
        rowp[12 * w + 8] = v3 ? D.Pt3.rp[r3] : 0;
        rowp[12 * w + 9] = v3 ? D.Pt3.rp[r3 + 1] : 0;
        rowp[12 * w + 10] = (THREE && v2) ? D.P3.rp[r2] : 0;
        rowp[12 * w + 11] = (THREE && v2) ? D.P3.rp[r2 + 1] : 0;
    }
#define dgF sm[oOWN + 0 * RES_WAVES + w]
#define dvF sm[oOWN + 1 * RES_WAVES + w]
#define bF sm[oOWN + 2 * RES_WAVES + w]
#define dgC sm[oOWN + 3 * RES_WAVES + w]
#define dvC sm[oOWN + 4 * RES_WAVES + w]
#define bC sm[oOWN + 5 * RES_WAVES + w]
#define dg2 sm[oOWN + 6 * RES_WAVES + w]
#define dv2 sm[oOWN + 7 * RES_WAVES + w]
#define dg3 sm[oOWN + 8 * RES_WAVES + w]
#define dv3 sm[oOWN + 9 * RES_WAVES + w]
    const bool nsp = D.isnsp != 0;
    const double xx1 = nsp ? D.L1.xx[0] : 1.0, xx2 = nsp ? D.L2.xx[0] : 1.0;
    const double xx3 = (THREE && nsp) ? D.L3.xx[0] : 1.0;
    for (int j = tid; j < N1; j += BT) {
        sm[oX + j] = xg[j];
        sm[oE1 + j] = 0.0;
        sm[oAX1 + j] = D.L1.Axi[j];
    }
    for (int j = tid; j < N2; j += BT) {
        sm[oE2 + j] = 0.0;
        sm[oAX2 + j] = D.L2.Axi[j];
    }
    if (THREE)
        for (int j = tid; j < N3; j += BT) {
            sm[oE3L + j] = 0.0;
            sm[oAX3L + j] = D.L3.Axi[j];
        }
    // One-row tail (the dense regimes): its transfer operator is one column, kept densely in the
    // unused upper half of the RR2 slot (restriction and prolongation use the same numbers), and
    // its operator is one number: the tail then costs two LDS passes instead of five dependent
    // trips to L2 per visit.
    const bool tail1 = !THREE && Nt == 1 && N2 <= RES_NMAX / 2;
    constexpr int oP3C = oRR2 + RES_NMAX / 2;
    // First sweep of a visit (zero start): e = D^-1 (r - (A 1) c) needs no matrix row, so every
    // workgroup forms ALL its entries itself from the r it has just received -- no hand-off.  The
    // inverse diagonals of the F rows of level 1 and of level 2 sit in the unused upper halves of
    // the R2 / E2 slots (same condition as tail1's column: at most RES_NMAX / 2 rows).
    const bool lfirst = D.localfirst && D.nu >= 1 && N2 <= RES_NMAX / 2 && nf <= RES_NMAX / 2;
    const bool lfirst2 = lfirst && !THREE;   // (DV2 shares the upper half of the E2 slot with E3)
    constexpr int oDV1 = oR2 + RES_NMAX / 2, oDV2 = oE2 + RES_NMAX / 2;
    if (lfirst) {
        for (int j = tid; j < nf; j += BT) sm[oDV1 + j] = D.L1.dinv[j];
        if (lfirst2)
            for (int j = tid; j < N2; j += BT) sm[oDV2 + j] = D.L2.dinv[j];
    }
    // mask-form transfers: this wave's F-row bits over the C nodes (low half) and its C-row bits over the
    // F rows (high half); bit q of a half <-> entry lane + 64 q, the layout of the register slices
    const bool xm = !THREE && D.xm != 0 && nc == N2 && D.wident != 0;
    unsigned xbits = 0;
    if (xm) {
        // (all 32 words requested in one burst: a loop of dependent loads cost ~1 us per word)
        unsigned long long wf[16], wc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            wf[q] = D.xm_fbits[(size_t)rF * D.xm_nwf + (q < D.xm_nwf ? q : 0)];
            wc[q] = D.xm_cbits[(size_t)(rC - nf) * D.xm_nwc + (q < D.xm_nwc ? q : 0)];
        }
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            lo |= (vF && q < D.xm_nwf) ? (unsigned)((wf[q] >> lane) & 1ull) << q : 0u;
            hi |= (vC && q < D.xm_nwc) ? (unsigned)((wc[q] >> lane) & 1ull) << q : 0u;
        }
        xbits = lo | (hi << 16);
        for (int j = tid; j < nf; j += BT) sm[oRHO + j] = D.xm_rho[j];
        for (int j = tid; j < nc; j += BT) sm[oBETA + j] = D.xm_beta[j];
    }
    double h33 = 0.0;
    if (tail1) {
        for (int j = tid; j < N2; j += BT) {
            double v = 0.0;
            if (POLY2) {
                v = D.p2rows[(size_t)N2 * D.p2ld + j];   // the stacked restriction row s
            } else {
                for (int t = D.P3.rp[j]; t < D.P3.rp[j + 1]; ++t)
                    if (D.P3.ci[t] == 0) v = D.P3.va[t];
            }
            sm[oP3C + j] = v;
        }
        for (int t = D.A3.rp[0]; t < D.A3.rp[1]; ++t)
            if (D.A3.ci[t] == 0) h33 = D.A3.va[t];
    }
    if (POLY2)   // the dense rows of B are walked in whole 64-entry steps: zeros behind r_2
        for (int j = N2 + tid; j < RES_NMAX / 2; j += BT) sm[oR2 + j] = 0.0;
    if (tid == 0) *fail = 0;
    __syncthreads();

    const auto rs = __builtin_amdgcn_make_buffer_rsrc(D.gran0, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    unsigned seq = 0;       // number of the last hand-off
    bool dead = false;      // a spin gave up somewhere: skip every further wait
    const bool dbg = D.dbg != nullptr && b == 0 && tid == 0;
    if (dbg) {
        dbg_acc[0] = dbg_acc[3] = dbg_acc[4] = dbg_acc[5] = dbg_acc[6] = dbg_acc[7] = 0;
        dbg_acc[1] = __builtin_amdgcn_s_memtime();
        dbg_acc[2] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- hand-off wrapper: sweep + barrier + store + (optional) block sums + barrier ------------
    // STORE(j, v) is called for every granule of the thread; EXTRA() runs once per thread in the
    // store phase (fix-ups on rows the thread does not sweep); both may add to p0 / p1, whose block
    // totals are returned in t0 / t1.
    // The waves have left their values in sm[oPUB + w] (first block) / sm[oPUB + RES_WAVES + w]
    // (second block); after the barrier -- which also ends the step's reads of the vectors that
    // are about to change -- wave 0 publishes them: one store instruction, one 128-byte segment per
    // block (a store per wave would be eight 16-byte partial-line writes: measured 2.6 us of waiting
    // per hand-off against 1.4).  gA/cA, gB/cB: first granule and row count of the two blocks.
#define RES_HANDOFF(NJ, n, gA, cA, gB, cB, STORE, EXTRA, want_sums, t0, t1)                        \
    do {                                                                                           \
        double hv_[NJ];                                                                            \
        ++seq;                                                                                     \
        /* a give-up of the previous hand-off (its closing barrier has ordered the flag): read here, where the */ \
        /* LDS round trip hides under the barrier, not behind the closing barrier on the critical path          */ \
        if (*fail) dead = true;                                                                    \
        if (dbg) dbg_acc[3] -= __builtin_amdgcn_s_memtime();                                       \
        __syncthreads();                                                                           \
        if (dbg) dbg_acc[3] += __builtin_amdgcn_s_memtime();                                       \
        if (w == 0) {                                                                              \
            const int l8_ = lane & (RES_WAVES - 1);                                                \
            const bool second_ = lane >= RES_WAVES;                                                \
            if (lane < 2 * RES_WAVES && l8_ < (second_ ? (cB) : (cA)) &&                           \
                !(seq == D.dbg_skip_seq && b == G - 1))                                            \
                res_publish(rs, seq, (second_ ? (gB) : (gA)) + l8_, sm[oPUB + lane]);              \
        }                                                                                          \
        if (dbg) dbg_acc[0] -= __builtin_amdgcn_s_memtime();                                       \
        for (int ps_ = 0; ps_ < D.presleep; ++ps_) __builtin_amdgcn_s_sleep(1);                    \
        if (res_sweep<NJ>(rs, seq, (n), dead, D.tmo, hv_, D.pollsleep)) {                             \
            *fail = 1;                                                                             \
            if (lane == 0) __hip_atomic_store(D.tmo, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        }                                                                                          \
        if (dbg) {                                                                                 \
            const long long t_ = __builtin_amdgcn_s_memtime();                                     \
            dbg_acc[0] += t_;                                                                      \
            dbg_acc[4] -= t_;                                                                      \
        }                                                                                          \
        double p0 = 0.0, p1 = 0.0;                                                                 \
        _Pragma("unroll") for (int u_ = 0; u_ < NJ; ++u_) {                                        \
            const int j = tid + u_ * BT;                                                           \
            if (j < (n)) {                                                                         \
                const double v = hv_[u_];                                                          \
                STORE;                                                                             \
            }                                                                                      \
        }                                                                                          \
        EXTRA;                                                                                     \
        if (want_sums) {                                                                           \
            p0 = wave_sum(p0);                                                                     \
            if ((want_sums) > 1) p1 = wave_sum(p1);                                                \
            if (lane == 0) {                                                                       \
                red[w] = p0;                                                                       \
                if ((want_sums) > 1) red[RES_WAVES + w] = p1;                                      \
            }                                                                                      \
        }                                                                                          \
        if (dbg) {                                                                                 \
            const long long t_ = __builtin_amdgcn_s_memtime();                                     \
            dbg_acc[4] += t_;                                                                      \
            dbg_acc[5] -= t_;                                                                      \
        }                                                                                          \
        __syncthreads();                                                                           \
        if (dbg) dbg_acc[5] += __builtin_amdgcn_s_memtime();                                       \
        if (want_sums) {                                                                           \
            t0 = res_red8(red);                                                                    \
            if ((want_sums) > 1) t1 = res_red8(red + RES_WAVES);                                   \
        }                                                                                          \
    } while (0)

    // sum of the LDS vector at `off` over the set bits of `bits` (entry lane + 64 q <-> bit q), n entries
    auto masked_sum = [&](unsigned bits, int off, int n) __attribute__((always_inline)) {
        double s0 = 0.0, s1 = 0.0;
        // (entry lane + 64 q at a constant distance from entry `lane`: one address register and immediate offsets;
        // entries beyond n lie inside the vector's LDS slot -- n <= RES_NMAX / 2 -- and their mask bits are zero)
        (void)n;
        const double* base = sm + off + lane;
        for (int q0 = 0; q0 < 16; q0 += 8) {   // eight gathers in flight (sixteen cost registers the row slices need)
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
    double c1 = 0.0, c2s = 0.0;     // kernel-space scalars of the next sweep on level 1 / 2
    double sumr2p = 0.0;            // POLY2: 1'r_2 of the visit
    const double p2ws = POLY2 ? D.p2w[N2] : 0.0;
    double res = 0.0, res0 = 0.0, prev = 0.0;
    double dum0 = 0.0, dum1 = 0.0;
    (void)dum0;
    (void)dum1;

    // r = b - A x (rows of this wave), ||r||, c1 for a zero start; E1 := 0        Class_AMG.m:89,96,103
    auto top = [&]() __attribute__((always_inline)) {
        const double sF = wave_sum(res_rowdot<KE1, 8 * oX>(cF, aF, smb));
        const double sC = wave_sum(res_rowdot<KE1, 8 * oX>(cC, aC, smb));
        if (lane == 0) {
            sm[oPUB + w] = bF - (sF + dgF * sm[oX + rF]);
            sm[oPUB + RES_WAVES + w] = bC - (sC + dgC * sm[oX + rC]);
        }
        double nrm2 = 0.0, sumr = 0.0;
        RES_HANDOFF(4, N1, loF, hiF - loF, loC, hiC - loC,
                    { sm[oR1 + j] = v; sm[oE1 + j] = 0.0; p0 += v * v; p1 += v; }, {}, 2, nrm2, sumr);
        c1 = nsp ? sumr / xx1 : 0.0;
        if (lfirst) {   // first half (F rows) of the first pre-smoothing sweep: half1(true, true, true)
            for (int j = tid; j < nf; j += BT) {
                const double g_i = sm[oR1 + j] - sm[oAX1 + j] * c1;
                sm[oE1 + j] = sm[oDV1 + j] * g_i;
            }
            __syncthreads();
        }
        return sqrt(nrm2);
    };

    // one half of a bigraph Gauss-Seidel sweep on level 1.  `first`: rows of the first half (the
    // other half still holds the old iterate); second half: + the shift by c of both halves and
    // the scalar of the next sweep.                         MG_Vcycle.m:15-21,34-38; Class_AMG.m:56-59
    auto half1 = [&](bool frows, bool first, bool ezero) __attribute__((always_inline)) {
        double s = 0.0, eo = 0.0;
        const int row = frows ? rowF : rowC;
        const bool valid = frows ? vF : vC;
        const int rr_ = valid ? row : 0;
        // the row's own scalars first: their LDS round trips overlap the gathers of the row (read after the
        // wave sum they were one more dependent LDS latency on every half sweep's critical path)
        if (!ezero) eo = sm[oE1 + rr_];
        const double dg_ = frows ? dgF : dgC, dv_ = frows ? dvF : dvC;
        const double r_own = sm[oR1 + rr_], ax_own = sm[oAX1 + rr_];
        if (!(ezero && first)) s = wave_sum(frows ? res_rowdot<KE1, 8 * oE1>(cF, aF, smb) : res_rowdot<KE1, 8 * oE1>(cC, aC, smb));
        s += dg_ * eo;
        const double g_i = r_own - s - ax_own * c1;
        const double wv = eo + dv_ * g_i;
        const int blk0 = frows ? 0 : nf, nblk = frows ? nf : nc;
        if (lane == 0) sm[oPUB + w] = wv;
        const int g0 = (frows ? loF : loC) - blk0, cnt = frows ? hiF - loF : hiC - loC;
        if (first) {
            RES_HANDOFF(2, nblk, g0, cnt, 0, 0, { sm[oE1 + blk0 + j] = v; }, {}, 0, dum0, dum1);
        } else {
            // other half: w -> w + c ; this half: wv + c ; scalar of the next sweep
            const int oth0 = frows ? nf : 0, noth = frows ? nc : nf;
            const double cc = c1;
            double xig = 0.0;
            RES_HANDOFF(2, nblk, g0, cnt, 0, 0,
                        {
                            const double en = v + cc;
                            sm[oE1 + blk0 + j] = en;
                            p0 += sm[oR1 + blk0 + j] - sm[oAX1 + blk0 + j] * en;
                        },
                        {
                            for (int jo = tid; jo < noth; jo += BT) {
                                const double en = sm[oE1 + oth0 + jo] + cc;
                                sm[oE1 + oth0 + jo] = en;
                                p0 += sm[oR1 + oth0 + jo] - sm[oAX1 + oth0 + jo] * en;
                            }
                        },
                        (nsp ? 1 : 0), xig, dum1);
            c1 = nsp ? xig / xx1 : 0.0;
        }
    };
    auto sweep1 = [&](bool post, bool ezero) __attribute__((always_inline)) {
        if (!(lfirst && ezero && !post)) half1(!post, true, ezero);   // else: done by top()    // pre: F rows first (Rk{1}); post: C rows first (Rk{1}')
        half1(post, false, ezero);
    };

    // weighted-Jacobi sweep on level 2                                   MG_Vcycle.m:15-21; Class_AMG.m:84
    auto sweep2 = [&](bool ezero) __attribute__((always_inline)) {
        double s = 0.0, eo = 0.0;
        if (!ezero) {
            s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2, a2, smb));
            eo = sm[oE2 + r2];
            s += dg2 * eo;
        }
        const double g_i = sm[oR2 + r2] - s - sm[oAX2 + r2] * c2s;
        const double wv = eo + dv2 * g_i;
        if (lane == 0) sm[oPUB + w] = wv;
        const double cc = c2s;
        double xig = 0.0;
        RES_HANDOFF(4, N2, lo2, hi2 - lo2, 0, 0,
                    {
                        const double en = v + cc;
                        sm[oE2 + j] = en;
                        p0 += sm[oR2 + j] - sm[oAX2 + j] * en;
                    },
                    {}, (nsp ? 1 : 0), xig, dum1);
        c2s = nsp ? xig / xx2 : 0.0;
    };

    // tail level: restriction, Jacobi-PCG (PCG.m:68-87, zero guess), prolongation -- all of it by
    // every workgroup on its own LDS copies, so no hand-off                     MG_Vcycle.m:27-31,43
    const auto rtin = __builtin_amdgcn_make_buffer_rsrc(D.tin, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    const auto rtout = __builtin_amdgcn_make_buffer_rsrc(D.tout, 0, 2 * RES_GRAN_MAX * 16, 0x00020000);
    unsigned tseq = 0;      // number of the last visit of the remote tail
    // Remote tail: the residual of the level above the tail's root (LDS offset oRRs) is restricted row
    // by row -- row i of the restriction by wave i / G of workgroup i % G (Nt <= BT <= 8 G rows) --
    // straight into the tail workgroup's inbox; then everybody waits for the prolongated correction
    // (Nout granules) and finishes e += P e_tail on its copy (offsets oEd, oRd, oAXd), with the scalar
    // of the next sweep.
    auto remote_tail = [&](const ResCsr& Pin, int oRRs, int oEd, int oRd, int oAXd, double xxd, int Nout,
                           double& cnext) __attribute__((always_inline)) {
        ++tseq;
        const int rin = b + G * w;
        if (rin < Nt) {
            const double s3 = res_csr_rowdot(Pin, rowp[12 * w + 6], rowp[12 * w + 7], lane, sm, oRRs);
            if (lane == 0)
                __builtin_amdgcn_raw_buffer_store_b128(res_pack(s3, tseq), rtin,
                                                       (int)(tseq & 1) * (RES_GRAN_MAX * 16) + rin * 16, 0,
                                                       16 /* sc1 */);
        }
        double hv[4];
        int st = 0;
        if (!dead) st = res_wait_slow<4>(rtout, tseq, Nout, D.tmo, nullptr, hv);
        if (st) {
            *fail = 1;
            if (lane == 0) __hip_atomic_store(D.tmo, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double p0 = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = tid + u * BT;
            if (j < Nout && !dead && !st) {
                const double en = sm[oEd + j] + hv[u];
                sm[oEd + j] = en;
                p0 += sm[oRd + j] - sm[oAXd + j] * en;
            }
        }
        if (nsp) {
            p0 = wave_sum(p0);
            if (lane == 0) red[w] = p0;
        }
        __syncthreads();
        if (nsp) cnext = res_red8(red) / xxd;
        if (*fail) dead = true;
    };
    // Local tail (at most 64 rows, solved redundantly by every workgroup on its own LDS copies, so no
    // hand-off): restriction of the level above (residual at oRRs), Jacobi-PCG (PCG.m:68-87, zero
    // guess), prolongation into the level above (oEd, with oRd / oAXd for the next scalar).
    auto local_tail = [&](const ResCsr& PtT, const ResCsr& AT, const ResCsr& PT, int oRRs, int oEd, int oRd,
                          int oAXd, double xxd, int Nabove, double& cnext) __attribute__((always_inline)) {
        if (Nt == 1) {
            // one row: its entries are dealt to all the waves, the eight partial sums are added in
            // wave order (a single wave walking 1024 entries took four dependent trips)
            const int e0 = PtT.rp[0], e1 = PtT.rp[1];
            double s = 0.0;
            for (int t = e0 + tid; t < e1; t += BT * 2) {
                const int t1 = t + BT;
                const int ja = PtT.ci[t], jb = PtT.ci[t1 < e1 ? t1 : e0];
                const double aa = PtT.va[t], ab = PtT.va[t1 < e1 ? t1 : e0];
                s += aa * sm[oRRs + ja];
                s += t1 < e1 ? ab * sm[oRRs + jb] : 0.0;
            }
            s = wave_sum(s);
            if (lane == 0) red[w] = s;
            __syncthreads();
            if (tid == 0) sm[oR3] = res_red8(red);
        } else {
            for (int i = w; i < Nt; i += RES_WAVES) {
                const double s = res_csr_rowdot(PtT, PtT.rp[i], PtT.rp[i + 1], lane, sm, oRRs);
                if (lane == 0) sm[oR3 + i] = s;
            }
        }
        __syncthreads();
        if (w == 0) {
            const int i = lane < Nt ? lane : 0;
            const bool valid = lane < Nt;
            const int e0 = AT.rp[i], e1 = AT.rp[i + 1];
            double dd = 0.0;
            for (int t = e0; t < e1; ++t)
                if (AT.ci[t] == i) dd = AT.va[t];
            double r = valid ? sm[oR3 + i] : 0.0;
            double p = valid ? r / dd : 0.0;
            double d = 0.0;
            double delta_new = wave_sum(valid ? r * p : 0.0);
            const double thresh = 1e-11 * 1e-11 * delta_new;
            long long it = 0;
            while (it < D.pcg_maxit && delta_new > thresh) {
                const double delta_old = delta_new;
                if (valid) sm[oP3 + i] = p;
                tiny_sync();
                double q = 0.0;
                if (valid)
                    for (int t = e0; t < e1; ++t) q += AT.va[t] * sm[oP3 + AT.ci[t]];
                tiny_sync();
                const double qp = wave_sum(valid ? q * p : 0.0);
                const double alpha = delta_old / qp;
                d += alpha * p;
                r = r - alpha * q;
                const double wi = valid ? r / dd : 0.0;
                delta_new = wave_sum(valid ? r * wi : 0.0);
                p = wi + (delta_new / delta_old) * p;
                ++it;
            }
            if (valid) sm[oE3 + i] = d;
        }
        __syncthreads();
        double p0 = 0.0;
        for (int j = tid; j < Nabove; j += BT) {
            double s = 0.0;
            for (int t = PT.rp[j]; t < PT.rp[j + 1]; ++t) s += PT.va[t] * sm[oE3 + PT.ci[t]];
            const double en = sm[oEd + j] + s;
            sm[oEd + j] = en;
            p0 += sm[oRd + j] - sm[oAXd + j] * en;
        }
        if (nsp) {
            p0 = wave_sum(p0);
            if (lane == 0) red[w] = p0;
        }
        __syncthreads();
        if (nsp) cnext = res_red8(red) / xxd;
        };
    auto tail = [&]() __attribute__((always_inline)) {
        if (D.remote) {
            remote_tail(D.Pt3, oRR2, oE2, oR2, oAX2, xx2, N2, c2s);
            return;
        }
        if (tail1) {
            double s = 0.0;
            for (int j = tid; j < N2; j += BT) s += sm[oP3C + j] * sm[oRR2 + j];       // r_3 = P' rr
            s = wave_sum(s);
            if (lane == 0) red[w] = s;
            __syncthreads();
            // PCG.m:68-87 on the 1 x 1 system, by every thread (the arithmetic of pcg_single)
            double r = res_red8(red);
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
            __syncthreads();   // red is rewritten below
            double p0 = 0.0;
            for (int j = tid; j < N2; j += BT) {                                       // e_2 += P e_3
                const double en = sm[oE2 + j] + sm[oP3C + j] * d;
                sm[oE2 + j] = en;
                p0 += sm[oR2 + j] - sm[oAX2 + j] * en;
            }
            if (nsp) {
                p0 = wave_sum(p0);
                if (lane == 0) red[w] = p0;
            }
            __syncthreads();
            if (nsp) c2s = res_red8(red) / xx2;
            return;
        }
        local_tail(D.Pt3, D.A3, D.P3, oRR2, oE2, oR2, oAX2, xx2, N2, c2s);
    };

    // ---- third resident level (THREE): Jacobi like level 2, N3 <= BT rows dealt in contiguous runs to the
    // workgroups (0-4 rows each).  A workgroup without a row still has to say that it has finished a
    // step (the two-buffer protocol lets a buffer be rewritten once everybody has published the step in
    // between), so every workgroup publishes one extra "ack" granule, N3 + b, with its rows.
    double c3s = 0.0, sumr3 = 0.0;
#define RES_HANDOFF3(STORE3, want_sums, t0)                                                              \
    do {                                                                                                 \
        if (w == 0 && lane == 0) sm[oPUB + RES_WAVES] = 0.0;                                             \
        RES_HANDOFF(2, N3 + G, lo3, hi3 - lo3, N3 + b, 1, { if (j < N3) { STORE3; } }, {}, want_sums, t0, dum1); \
    } while (0)
    // POLY4: the ack granule of workgroup b < N4 carries r_4[b] (the restricted residual goes to everybody) ...
#define RES_HANDOFF3R(ACKV, STORE3, STORE4, want_sums, t0)                                               \
    do {                                                                                                 \
        if (w == 0 && lane == 0) sm[oPUB + RES_WAVES] = (ACKV);                                          \
        RES_HANDOFF(2, N3 + G, lo3, hi3 - lo3, N3 + b, 1,                                                \
                    { if (j < N3) { STORE3; } else if (j - N3 < Nt) { const int j4 = j - N3; STORE4; } }, {}, want_sums, t0, dum1); \
    } while (0)
    // ... and the hand-off among the rows of level 4: row b of workgroup b < N4, an ack granule of everybody
#define RES_HANDOFF4(STORE4)                                                                             \
    do {                                                                                                 \
        if (w == 0 && lane == 0) sm[oPUB + RES_WAVES] = 0.0;                                             \
        RES_HANDOFF(1, Nt + G, b, (b < Nt ? 1 : 0), Nt + b, 1, { if (j < Nt) { STORE4; } }, {}, 0, dum0, dum1); \
    } while (0)
    auto sweep3 = [&](bool ezero) __attribute__((always_inline)) {
        if (THREE) {
            double s = 0.0, eo = 0.0;
            if (!ezero) {
                s = wave_sum(res_rowdot<K3, 8 * oE3L>(c3, a3, smb));
                eo = sm[oE3L + r3];
                s += dg3 * eo;
            }
            const double g_i = sm[oR3L + r3] - s - sm[oAX3L + r3] * c3s;
            const double wv = eo + dv3 * g_i;
            if (lane == 0) sm[oPUB + w] = wv;
            const double cc = c3s;
            double xig = 0.0;
            RES_HANDOFF3({
                             const double en = v + cc;
                             sm[oE3L + j] = en;
                             p0 += sm[oR3L + j] - sm[oAX3L + j] * en;
                         },
                         (nsp ? 1 : 0), xig);
            c3s = nsp ? xig / xx3 : 0.0;
        }
    };
    // one visit of level 3 and, through the remote tail rooted at level 4, of everything below it
    // polynomial form: the sums of this workgroup's rows against [r_3; e_3] (+ their factor of 1'r_3)
    // -> sm[oR3 + 40 + q]; the caller's next barrier publishes them
    auto poly3_rows = [&](int nrows, bool post) __attribute__((always_inline)) {
        const double xr = tid < N3 ? sm[oR3L + tid] : 0.0, xe = tid < N3 ? sm[oE3L + tid] : 0.0;
        const double xc = (post && tid < Nt) ? sm[oRR3L + tid] : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (q < nrows) {
                double t = __builtin_fma(m3e[q], xe, m3r[q] * xr);
                if (q < 4) t = __builtin_fma(m3c[q], xc, t);
                const double pq = wave_sum(t);
                if (lane == 0) sm[oR3 + 8 * q + w] = pq;
            }
        }
        __syncthreads();
        if (tid < nrows) {
            double sq = 0.0;
#pragma unroll
            for (int ww = 0; ww < RES_WAVES; ++ww) sq += sm[oR3 + 8 * tid + ww];
            sm[oR3 + 40 + tid] = __builtin_fma(sm[oE3 + tid], sumr3, sq);
        }
        __syncthreads();
    };
    // POLY4: workgroup b's row of level 4 (b < N4 = Nt) and restriction row N4 + b (b < N5) against [r_4; e_4]
    // (+ (M1 P5) e_5 in the second pass) -> sm[oPS4 + 16 + q]; coefficients from L2 at every pass, one entry per
    // thread and segment (threads 0..127).  e_4 lives where the tail's answer of the POLY3 mode does (oRR3L).
    const int N5 = POLY3 ? D.N5 : 0;
    const bool poly4 = POLY3 && N5 > 0;
    double sumr4 = 0.0;
    auto poly4_rows = [&](int nrows, bool post) __attribute__((always_inline)) {
        // rows q = 0: row b of [M2a | M1] (b < N4), q = 1: restriction row N4 + b (b < N5; first pass only)
        double m4r[2], m4e[2], m4c;
        const int t4 = tid < RES_P4_SEG ? tid : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = q == 0 ? b : Nt + b;
            const bool okr = q < nrows && (q == 0 ? b < Nt : b < N5) && tid < RES_P4_SEG;
            const double* pr = D.p4rows + (size_t)(okr ? row : 0) * RES_P4_LD;
            const double vr = pr[t4], ve = pr[RES_P4_SEG + t4];
            m4r[q] = okr ? vr : 0.0;
            m4e[q] = okr ? ve : 0.0;
            if (q == 0) {
                const double vc = pr[2 * RES_P4_SEG + (tid < 64 ? tid : 0)];
                m4c = (okr && post && tid < N5) ? vc : 0.0;
            }
        }
        const double xr = tid < Nt ? sm[oR4L + t4] : 0.0, xe = tid < Nt ? sm[oRR3L + t4] : 0.0;
        const double xc = (post && tid < N5) ? sm[oE5L + tid] : 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q < nrows) {
                double t = __builtin_fma(m4e[q], xe, m4r[q] * xr);
                if (q == 0) t = __builtin_fma(m4c, xc, t);
                const double pq = wave_sum(t);
                if (lane == 0) sm[oPS4 + 8 * q + w] = pq;
            }
        }
        __syncthreads();
        if (tid < nrows) {
            double sq = 0.0;
#pragma unroll
            for (int ww = 0; ww < RES_WAVES; ++ww) sq += sm[oPS4 + 8 * tid + ww];
            sm[oPS4 + 16 + tid] = __builtin_fma(sm[oPS4 + 20 + tid], sumr4, sq);
        }
        __syncthreads();
    };
    auto visit3 = [&](bool keep) __attribute__((always_inline)) {
        if (POLY3) {
            // e' = M2a r + M1 e and the restricted residual of e' in one pass                 MG_Vcycle.m:20-29
            poly3_rows(5, false);
            if (poly4) {   // r_4 to everybody (ack granules), e_4 := 0; then level 4's own visits      MG_Wcycle.m:28-30
                if (tid < 4) sm[oPUB + tid] = sm[oR3 + 40 + tid];
                double s4 = 0.0;
                RES_HANDOFF3R((b < Nt ? sm[oR3 + 44] : 0.0), { sm[oE3L + j] = v; },
                              { sm[oR4L + j4] = v; sm[oRR3L + j4] = 0.0; p0 += v; }, (nsp ? 1 : 0), s4);
                sumr4 = nsp ? s4 : 0.0;
                for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) {
                    ++tseq;
                    poly4_rows(2, false);                              // e_4' (row b) and r_5[b] = P5'(r_4 - A_4 e_4')
                    if (tid == 0 && b < N5)
                        __builtin_amdgcn_raw_buffer_store_b128(res_pack(sm[oPS4 + 17], tseq), rtin,
                                                               (int)(tseq & 1) * (RES_GRAN_MAX * 16) + b * 16, 0,
                                                               16 /* sc1 */);
                    if (tid == 0) sm[oPUB] = sm[oPS4 + 16];
                    RES_HANDOFF4({ sm[oRR3L + j] = v; });
                    double hv5[1];
                    int st5 = 0;
                    if (!dead) st5 = res_wait_slow<1>(rtout, tseq, N5, D.tmo, nullptr, hv5);   // e_5
                    if (st5) {
                        *fail = 1;
                        if (lane == 0) __hip_atomic_store(D.tmo, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (tid < N5) sm[oE5L + tid] = (!dead && !st5) ? hv5[0] : 0.0;
                    __syncthreads();
                    if (*fail) dead = true;
                    poly4_rows(1, true);                               // e_4'' = M2a r + M1 e' + (M1 P5) e_5
                    if (tid == 0) sm[oPUB] = sm[oPS4 + 16];
                    RES_HANDOFF4({ sm[oRR3L + j] = v; });
                }
            } else {
                ++tseq;
                if (tid == 0 && b < Nt)
                    __builtin_amdgcn_raw_buffer_store_b128(res_pack(sm[oR3 + 44], tseq), rtin,
                                                           (int)(tseq & 1) * (RES_GRAN_MAX * 16) + b * 16, 0,
                                                           16 /* sc1 */);
                if (tid < 4) sm[oPUB + tid] = sm[oR3 + 40 + tid];
                RES_HANDOFF3({ sm[oE3L + j] = v; }, 0, dum0);
                double hv[1];
                int st = 0;
                if (!dead) st = res_wait_slow<1>(rtout, tseq, Nt, D.tmo, nullptr, hv);   // e_4 (Nt <= G <= BT values)
                if (st) {
                    *fail = 1;
                    if (lane == 0) __hip_atomic_store(D.tmo, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (tid < Nt) sm[oRR3L + tid] = (!dead && !st) ? hv[0] : 0.0;
                __syncthreads();
                if (*fail) dead = true;
            }
            poly3_rows(4, true);                                         // e'' = M2a r + M1 e' + (M1 P4) e_4   :31-41
            if (tid < 4) sm[oPUB + tid] = sm[oR3 + 40 + tid];
            RES_HANDOFF3({ sm[oE3L + j] = v; }, 0, dum0);
            (void)keep;
        } else if (THREE) {
            const int nu = D.nu;
            for (int s = 0; s < nu; ++s) sweep3(!keep && s == 0);
            {   // rr = r - A e                                                       MG_Vcycle.m:27
                const double s = wave_sum(res_rowdot<K3, 8 * oE3L>(c3, a3, smb)) + dg3 * sm[oE3L + r3];
                if (lane == 0) sm[oPUB + w] = sm[oR3L + r3] - s;
                RES_HANDOFF3({ sm[oRR3L + j] = v; }, 0, dum0);
            }
            if (D.remote)
                remote_tail(D.Pt4, oRR3L, oE3L, oR3L, oAX3L, xx3, N3, c3s);
            else
                local_tail(D.Pt4, D.A4, D.P4, oRR3L, oE3L, oR3L, oAX3L, xx3, N3, c3s);
            for (int s = 0; s < nu; ++s) sweep3(false);
        }
    };

    // one visit of level 2 and everything below it
    auto visit2 = [&](bool keep) __attribute__((always_inline)) {
        const int nu = D.nu;
        if (POLY2) {   // the whole visit as one composed pass (ResDesc::p2rows): ONE hand-off
            // r_3 = s'r_2 + ws (1'r_2) and the one-row tail's PCG (PCG.m:68-87) by every workgroup
            double s3 = 0.0;
            for (int j = tid; j < N2; j += BT) s3 += sm[oP3C + j] * sm[oR2 + j];
            s3 = wave_sum(s3);
            if (lane == 0) red[w] = s3;
            __syncthreads();
            double r = res_red8(red) + p2ws * sumr2p;
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
            // e_2 = B r_2 + wB (1'r_2) + mp e_3 on the own row: a dense row against R2 (entry lane + 64 q at a
            // constant distance: immediate offsets, no column registers)
            const double* rb = sm + oR2 + lane;
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int q = 0; q < KE2; q += 4) {
                t0 += a2[q] * rb[64 * q];
                t1 += a2[q + 1] * rb[64 * (q + 1)];
                t2 += a2[q + 2] * rb[64 * (q + 2)];
                t3 += a2[q + 3] * rb[64 * (q + 3)];
            }
            const double val = wave_sum((t0 + t1) + (t2 + t3)) + dg2 * sumr2p + dv2 * d;
            if (lane == 0) sm[oPUB + w] = val;
            RES_HANDOFF(4, N2, lo2, hi2 - lo2, 0, 0, { sm[oE2 + j] = v; }, {}, 0, dum0, dum1);
            (void)keep;
            (void)nu;
            return;
        }
        for (int s = (lfirst2 && !keep) ? 1 : 0; s < nu; ++s) sweep2(!keep && s == 0);
        // rr = r - A e                                                           MG_Vcycle.m:27
        {
            const double s = wave_sum(res_rowdot<KE2, 8 * oE2>(c2, a2, smb)) + dg2 * sm[oE2 + r2];
            if (lane == 0) sm[oPUB + w] = sm[oR2 + r2] - s;
            RES_HANDOFF(4, N2, lo2, hi2 - lo2, 0, 0, { sm[oRR2 + j] = v; }, {}, 0, dum0, dum1);
        }
        if (dbg) dbg_acc[7] -= __builtin_amdgcn_s_memtime();
        if (THREE) {
            {   // r_3 = P3' rr_2 ; E3 := 0 ; c for the zero start
                const double s = res_csr_rowdot(D.Pt3, rowp[12 * w + 8], rowp[12 * w + 9], lane, sm, oRR2);
                if (lane == 0) sm[oPUB + w] = s;
                double sumr = 0.0;
                RES_HANDOFF3({ sm[oR3L + j] = v; sm[oE3L + j] = 0.0; p0 += v; }, (nsp ? 1 : 0), sumr);
                c3s = nsp ? sumr / xx3 : 0.0;
                sumr3 = sumr;
            }
            for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) visit3(leg == 1);   // MG_Wcycle.m:28-30
            {   // e_2 += P3 e_3                                                     MG_Vcycle.m:31
                const double sP = res_csr_rowdot(D.P3, rowp[12 * w + 10], rowp[12 * w + 11], lane, sm, oE3L);
                if (lane == 0) sm[oPUB + w] = sm[oE2 + r2] + sP;
                double xig = 0.0;
                RES_HANDOFF(4, N2, lo2, hi2 - lo2, 0, 0,
                            { sm[oE2 + j] = v; p0 += sm[oR2 + j] - sm[oAX2 + j] * v; }, {}, (nsp ? 1 : 0), xig, dum1);
                c2s = nsp ? xig / xx2 : 0.0;
            }
        } else {
            tail();
        }
        if (dbg) dbg_acc[7] += __builtin_amdgcn_s_memtime();
        for (int s = 0; s < nu; ++s) sweep2(false);
    };

    // MG_Vcycle / MG_Wcycle from level 1 down; the correction ends in E1
    auto cycle = [&]() __attribute__((always_inline)) {
        const int nu = D.nu;
        for (int s = 0; s < nu; ++s) sweep1(false, s == 0);
        {   // rr = r - A e on both blocks
            const double sF = wave_sum(res_rowdot<KE1, 8 * oE1>(cF, aF, smb)) + dgF * sm[oE1 + rF];
            const double sC = wave_sum(res_rowdot<KE1, 8 * oE1>(cC, aC, smb)) + dgC * sm[oE1 + rC];
            if (lane == 0) {
                sm[oPUB + w] = sm[oR1 + rF] - sF;
                sm[oPUB + RES_WAVES + w] = sm[oR1 + rC] - sC;
            }
            if (xm) {   // the F part arrives pre-scaled by rho for the mask-form restriction below
                RES_HANDOFF(4, N1, loF, hiF - loF, loC, hiC - loC,
                            { sm[oRR1 + j] = j < nf ? v * sm[oRHO + j] : v; }, {}, 0, dum0, dum1);
            } else {
                RES_HANDOFF(4, N1, loF, hiF - loF, loC, hiC - loC, { sm[oRR1 + j] = v; }, {}, 0, dum0, dum1);
            }
        }
        {   // r_2 = P' rr ; E2 := 0 ; c for the zero start
            if (dbg) dbg_acc[6] -= __builtin_amdgcn_s_memtime();
            // row r2 of P' is [W(:,r2)' , 1 at nf + r2]; rowC == nf + row2 (level 2 = the C nodes)
            const double s = xm ? sm[oBETA + r2] * masked_sum(xbits >> 16, oRR1, nf) + sm[oRR1 + rC]
                                : res_csr_rowdot(D.Pt2, rowp[12 * w + 0], rowp[12 * w + 1], lane, sm, oRR1) +
                                      (D.wident ? sm[oRR1 + rC] : 0.0);
            if (dbg) dbg_acc[6] += __builtin_amdgcn_s_memtime();
            if (lane == 0) sm[oPUB + w] = s;
            double sumr = 0.0;
            RES_HANDOFF(4, N2, lo2, hi2 - lo2, 0, 0, { sm[oR2 + j] = v; sm[oE2 + j] = 0.0; p0 += v; }, {},
                        (nsp ? 1 : 0), sumr, dum1);
            c2s = nsp ? sumr / xx2 : 0.0;
            sumr2p = sumr;
            if (lfirst2 && !POLY2) {   // sweep2(true) of the first visit, same thread-to-entry map and sums
                const double cc = c2s;
                double p0 = 0.0;
                for (int j = tid; j < N2; j += BT) {
                    const double g_i = sm[oR2 + j] - sm[oAX2 + j] * cc;
                    const double en = sm[oDV2 + j] * g_i + cc;
                    sm[oE2 + j] = en;
                    p0 += sm[oR2 + j] - sm[oAX2 + j] * en;
                }
                if (nsp) {
                    p0 = wave_sum(p0);
                    if (lane == 0) red[w] = p0;
                }
                __syncthreads();
                if (nsp) c2s = res_red8(red) / xx2;
                __syncthreads();   // red is rewritten by the next hand-off
            }
        }
        for (int leg = 0; leg < (D.wcycle ? 2 : 1); ++leg) visit2(leg == 1);      // MG_Wcycle.m:28-30
        {   // e_1 += P e_2                                                        MG_Vcycle.m:31
            if (dbg) dbg_acc[6] -= __builtin_amdgcn_s_memtime();
            // F rows: W(rowF,:) against E2 (A's columns nf + i are level-2 indices i); C rows: identity
            if (xm) {   // beta .* e_2 once per workgroup, then one masked sum per F row
                for (int j = tid; j < N2; j += BT) sm[oU + j] = sm[oBETA + j] * sm[oE2 + j];
                __syncthreads();
            }
            const double sF = xm ? sm[oRHO + rF] * masked_sum(xbits & 0xffffu, oU, N2)
                                 : res_csr_rowdot(D.P2, rowp[12 * w + 2], rowp[12 * w + 3], lane, sm, oE2);
            const double sC = D.wident ? sm[oE2 + r2]
                                       : res_csr_rowdot(D.P2, rowp[12 * w + 4], rowp[12 * w + 5], lane, sm, oE2);
            if (dbg) dbg_acc[6] += __builtin_amdgcn_s_memtime();
            if (lane == 0) {
                sm[oPUB + w] = sm[oE1 + rF] + sF;
                sm[oPUB + RES_WAVES + w] = sm[oE1 + rC] + sC;
            }
            double xig = 0.0;
            RES_HANDOFF(4, N1, loF, hiF - loF, loC, hiC - loC,
                        { sm[oE1 + j] = v; p0 += sm[oR1 + j] - sm[oAX1 + j] * v; }, {}, (nsp ? 1 : 0), xig, dum1);
            c1 = nsp ? xig / xx1 : 0.0;
        }
        for (int s = 0; s < nu; ++s) sweep1(true, false);
    };

    auto add_correction = [&]() __attribute__((always_inline)) {   // x += e                                       Class_AMG.m:98,101
        for (int j = tid; j < N1; j += BT) sm[oX + j] = sm[oX + j] + sm[oE1 + j];
        __syncthreads();
    };

    // ---- Class_AMG.m:86-109 (one call site of every step: the kernel is large) -----------------
    const int maxit = D.maxit;
    double* relk = out + 4;
    double* rhok = out + 4 + (maxit + 2);
    const bool writer = b == 0 && tid == 0;
    const bool fixed = fixed_cycles > 0;
    int it = 0, done = 0;
    double rel_res = 0.0, last_rel = 1.0;
    bool first = true;
    for (;;) {
        const double rnow = top();                                                // :89 / :103
        if (first) {
            first = false;
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
    if (D.remote && b == 0 && tid == 0)   // release the tail workgroup
        __hip_atomic_store(D.tctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0)
        for (int j = tid; j < N1; j += BT) xg[j] = sm[oX + j];
    if (writer) {
        out[0] = (double)it;
        out[1] = rel_res;
        out[2] = res0;
        // any workgroup's give-up, not only this one's: a workgroup that gave up keeps publishing
        // (tagged, but computed from values it never received), so the iterate is void even when
        // workgroup 0 itself saw every hand-off arrive
        const unsigned anytmo = __hip_atomic_load(D.tmo, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        out[3] = (dead || anytmo != 0) ? 1.0 : 0.0;
        // the last slot of the rhok block is never written by the iteration (it <= maxit): hand-offs of this
        // launch, chip-wide ones and visits of the remote tail (ipd_amg_resident_kernel)
        out[4 + 2 * (maxit + 2) - 1] = (double)(seq + tseq);
    }
    if (dbg) {
        D.dbg[0] = dbg_acc[0];
        D.dbg[1] = __builtin_amdgcn_s_memtime() - dbg_acc[1];
        D.dbg[2] = seq;
        D.dbg[3] = __builtin_amdgcn_s_memrealtime() - dbg_acc[2];
        D.dbg[4] = dbg_acc[3];
        D.dbg[5] = dbg_acc[4];
        D.dbg[6] = dbg_acc[5];
        D.dbg[7] = dbg_acc[6];
        D.dbg[8] = dbg_acc[7];
    }
#undef RES_HANDOFF3
#undef RES_HANDOFF3R
#undef RES_HANDOFF4
#undef RES_HANDOFF
#undef dgF
#undef dvF
#undef bF
#undef dgC
#undef dvC
#undef bC
#undef dg2
#undef dv2
#undef dg3
#undef dv3
}
