// AMG setup on the device: Ruge-Stueben strength, C/F splitting (mis_set, the live
// one, and cf_split, the one north_star names), interpolation, Galerkin product.
//   AMG/strength.m:6-18, AMG/mis_set.m:9-67, AMG/cf_split.m:6-16,
//   AMG/transfer.m:17-66, AMG/Class_AMG.m:41-85.
//
// Everything in this TU is integer/compare work or strictly ordered fp64
// arithmetic (no FMA contraction, sequential accumulation in ascending index),
// so C/F masks, Pro and Ac are BIT-IDENTICAL to the oracle's on every level.
// The kernels are latency-bound at realistic sizes (N <= 4096, nnz 1e3..1e6):
// the design goal is few launches and no float atomics, not bandwidth.
#pragma clang fp contract(off)

#include "ipd_amg_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

static inline int rows_grid(int nr) { return std::max(1, std::min(cdiv(nr, 4), 4096)); }
static inline int elems_grid(long long n) {
    return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096));
}

#define WAVE_ROWS(r, nr)                                                    \
    const int lane = threadIdx.x & 63;                                      \
    const int wave__ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;        \
    const int nwaves__ = (gridDim.x * blockDim.x) >> 6;                     \
    for (int r = wave__; r < (nr); r += nwaves__)

#define THREAD_ELEMS(i, n)                                                  \
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n);            \
         i += gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------
// strength                                                (AMG/strength.m:7-18)
// ---------------------------------------------------------------------------
// max_row(i) = max over the row of D-A; the diagonal of D-A is an implicit zero,
// so the maximum is never negative; "<= 0 -> Inf" (strength.m:9-10).
__global__ __launch_bounds__(256) void k_rowmax(int nr, const int* __restrict__ rp,
                                                const int* __restrict__ ci,
                                                const double* __restrict__ va,
                                                double* __restrict__ maxrow,
                                                double* __restrict__ diag) {
    WAVE_ROWS(r, nr) {
        double mx = 0.0, dg = 0.0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            const int j = ci[t];
            const double v = va[t];
            if (j == r)
                dg = v;
            else
                mx = fmax(mx, -v);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            mx = fmax(mx, __shfl_xor(mx, d));
            dg += __shfl_xor(dg, d);  // at most one lane holds the diagonal
        }
        if (lane == 0) {
            maxrow[r] = mx > 0.0 ? mx : INFINITY;
            diag[r] = dg;
        }
    }
}

// strong(t) = [ -a_ij / min(max_row(i), max_row(j)) >= theta ], j != i   (mis_set.m:25)
// degi = column counts of the mask (mis_set.m:28), rowcnt = row counts (mis_set.m:67)
__global__ __launch_bounds__(256) void k_strong(int nr, const int* __restrict__ rp,
                                                const int* __restrict__ ci,
                                                const double* __restrict__ va,
                                                const double* __restrict__ maxrow, double theta,
                                                uint8_t* __restrict__ strong,
                                                int* __restrict__ degi, int* __restrict__ rowcnt) {
    WAVE_ROWS(r, nr) {
        const double mr = maxrow[r];
        int cnt = 0;
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            const int j = ci[t];
            bool f = false;
            if (j != r) {
                const double sv = (-va[t]) / fmin(mr, maxrow[j]);
                f = sv >= theta;
            }
            strong[t] = f ? 1 : 0;
            if (f) {
                atomicAdd(&degi[j], 1);
                ++cnt;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
        if (lane == 0) rowcnt[r] = cnt;
    }
}

void amg_strength_mask(ipd_ctx* ctx, const Csr& A, double theta, uint8_t* strong, int* degi,
                       int* rowcnt) {
    Arena& tmp = *ctx->scratch;
    double* maxrow = tmp.alloc<double>((size_t)A.nr);
    double* diag = tmp.alloc<double>((size_t)A.nr);
    IPD_HIP(hipMemsetAsync(degi, 0, sizeof(int) * (size_t)std::max(A.nr, 1), ctx->stream));
    hipLaunchKernelGGL(k_rowmax, dim3(rows_grid(A.nr)), dim3(256), 0, ctx->stream, A.nr, A.rp, A.ci,
                       A.va, maxrow, diag);
    IPD_KERNEL_CHECK();
    hipLaunchKernelGGL(k_strong, dim3(rows_grid(A.nr)), dim3(256), 0, ctx->stream, A.nr, A.rp, A.ci,
                       A.va, maxrow, theta, strong, degi, rowcnt);
    IPD_KERNEL_CHECK();
}

// strength VALUES for ipd_strength (zero where dropped; compacted afterwards)
__global__ __launch_bounds__(256) void k_strength_values(int nr, const int* __restrict__ rp,
                                                         const int* __restrict__ ci,
                                                         const double* __restrict__ va,
                                                         const double* __restrict__ maxrow,
                                                         int which, double* __restrict__ out) {
    WAVE_ROWS(r, nr) {
        const double mr = maxrow[r];
        for (int t = rp[r] + lane; t < rp[r + 1]; t += 64) {
            const int j = ci[t];
            double sv = 0.0;
            if (j != r) sv = (-va[t]) / (which == 1 ? mr : fmin(mr, maxrow[j]));
            out[t] = sv;
        }
    }
}

// ---------------------------------------------------------------------------
// mis_set                                                  (AMG/mis_set.m:25-67)
// ---------------------------------------------------------------------------
__global__ void k_flag_pos(int n, const int* __restrict__ v, int* __restrict__ flag) {
    THREAD_ELEMS(i, n) flag[i] = v[i] > 0 ? 1 : 0;
}

// deg(idx) = deg(idx) + 0.1*rand(sum(idx),1)  (:35);  isF(deg==0) = true (:40)
__global__ void k_deg_init(int n, const int* __restrict__ degi, const int* __restrict__ rank,
                           const double* __restrict__ randv, double* __restrict__ deg,
                           uint8_t* __restrict__ isC, uint8_t* __restrict__ isF,
                           uint8_t* __restrict__ isU, uint8_t* __restrict__ isS) {
    THREAD_ELEMS(i, n) {
        const int d = degi[i];
        double dv = 0.0;
        if (d > 0) {
            const double tie = 0.1 * randv[rank[i]];
            dv = (double)d + tie;
        }
        deg[i] = dv;
        isC[i] = 0;
        isF[i] = d == 0 ? 1 : 0;
        isU[i] = 1;
        isS[i] = d > 0 ? 1 : 0;   // isS = deg > 0 of the first round (:47)
    }
}

// edges (i,j), i<j, of triu(As(S,S),1): the smaller degree loses; ties keep the
// smaller index (:49-52).  Every write stores 0, so the races are benign.
__global__ __launch_bounds__(256) void k_mis_sel_kill(int nr, const int* __restrict__ rp,
                                                      const int* __restrict__ ci,
                                                      const uint8_t* __restrict__ strong,
                                                      const double* __restrict__ deg,
                                                      uint8_t* __restrict__ isS,
                                                      int* __restrict__ counts) {
    if (blockIdx.x == 0 && threadIdx.x < 2) counts[threadIdx.x] = 0;   // summed by k_mis_settle
    WAVE_ROWS(i, nr) {
        const double di = deg[i];
        if (di > 0.0) {
            for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) {
                const int j = ci[t];
                if (strong[t] && j > i) {
                    const double dj = deg[j];
                    if (dj > 0.0) {
                        if (di >= dj)
                            isS[j] = 0;
                        else
                            isS[i] = 0;
                    }
                }
            }
        }
    }
}

// The rest of a round in one launch, one wave per node (:53-59 and the loop test :42):
//   isC(isS) = true;  [i,~] = find(As(:,isC)); isF(i) = true;  isU = ~(isF|isC);  deg(~isU) = 0
// A neighbour is in C after this round iff it was before or survived the selection, and both
// flags are final when this kernel starts, so no node waits for another one's commit.  The
// selection of the next round, isS = deg > 0 = isU, goes to a second buffer because isS is still
// being read here.
__global__ __launch_bounds__(256) void k_mis_settle(int nr, const int* __restrict__ rp,
                                                    const int* __restrict__ ci,
                                                    const uint8_t* __restrict__ strong,
                                                    const uint8_t* __restrict__ isS,
                                                    uint8_t* __restrict__ isC,
                                                    uint8_t* __restrict__ isF,
                                                    uint8_t* __restrict__ isU,
                                                    double* __restrict__ deg,
                                                    uint8_t* __restrict__ isS_next,
                                                    int* __restrict__ counts) {
    int nc = 0, nu = 0;
    WAVE_ROWS(i, nr) {
        bool hit = false;
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) {
            const int j = ci[t];
            if (strong[t] && (isC[j] || isS[j])) hit = true;
        }
        hit = __any(hit);
        if (lane == 0) {
            const bool c = isC[i] || isS[i];
            const bool f = isF[i] || hit;
            const bool u = !(c || f);
            if (c) isC[i] = 1;
            if (f) isF[i] = 1;
            isU[i] = u ? 1 : 0;
            isS_next[i] = u ? 1 : 0;
            if (!u) deg[i] = 0.0;
            nc += c;
            nu += u;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (nc) atomicAdd(&counts[0], nc);
        if (nu) atomicAdd(&counts[1], nu);
    }
}

__global__ void k_mis_absorb(int n, uint8_t* __restrict__ isU, uint8_t* __restrict__ isC) {
    THREAD_ELEMS(i, n) if (isU[i]) {
        isC[i] = 1;
        isU[i] = 0;
    }
}

// iso = sum(As,2)==0; isC(iso) = true; isF(iso) = false   (:67)
__global__ void k_mis_iso(int n, const int* __restrict__ rowcnt, uint8_t* __restrict__ isC,
                          uint8_t* __restrict__ isF) {
    THREAD_ELEMS(i, n) if (rowcnt[i] == 0) {
        isC[i] = 1;
        isF[i] = 0;
    }
}


// ---------------------------------------------------------------------------
// mis_set of a SMALL level in one launch                   (AMG/mis_set.m:25-67)
// ---------------------------------------------------------------------------
// Levels >= 2 of the drivers' Newton systems have a few hundred rows and a few thousand entries.
// There the launch-per-step form above is ~20 launches and 4 host round trips per level (strength,
// degree flags + scan, random numbers, every round of the selection, the clean-up, the C index
// scan): 100-150 us of launch and round-trip latency around ~10 us of work.  For levels of at most
// 1024 rows and MIS_SMALL_NNZ entries ONE workgroup does all of it, thread i = node i: the rows'
// strong neighbours are listed once in LDS (16-bit indices), the rounds run on those lists with the
// degrees and flags in LDS, and the loop test of mis_set.m:42 is taken on the device.  The random
// numbers of mis_set.m:35 are handed in as the NEXT N numbers of the stream; the kernel uses the
// first `nconn` of them (as the reference does) and reports nconn, and the host then consumes exactly
// that many (ipd_rng state saved and restored around the peek).  Same statements, same order of
// evaluation per entry as k_rowmax / k_strong / k_deg_init / k_mis_sel_kill / k_mis_settle /
// k_mis_absorb / k_mis_iso / k_u8_to_flag / k_count_bad_split + the scans: identical bits.
static constexpr int MIS_SMALL_ROWS = 1024;
static constexpr int MIS_SMALL_NNZ = 40000;
struct MisSmallArgs {
    int N, N0;
    const int* rp;
    const int* ci;
    const double* va;
    double theta;
    const double* randv;      // N values: the stream's next numbers
    uint8_t* strong;          // out: nnz flags
    double* maxrow;           // out (interpolation needs them again, transfer.m:49-51)
    double* diag;
    uint8_t* isC;             // out
    uint8_t* isF;             // out
    int* cidx;                // out: N + 1 entries, cidx[N] = Nc
    volatile unsigned* box;   // mailbox: {status, nconn, Nc, bad, rounds}; status 1 = degenerate branch (:30-34)
    unsigned ticket;
};
__device__ __forceinline__ int mis_block_exscan(int v, int* wsum, int* total) {   // 1024 threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int sgm = wsum[k];
        if (k < w) woff += sgm;
        tot += sgm;
    }
    *total = tot;
    return woff + x - v;
}
__global__ __launch_bounds__(1024) void k_mis_small(const MisSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char mis_raw[];
    __shared__ int wsum[16];
    __shared__ int s_cnt[2];
    const int N = a.N;
    // L lanes per node (the largest of 1, 2, 4, 8 with N L <= 1024): a row's entries -- in memory for the
    // strength pass, its strong-neighbour list in LDS for the rounds -- are strided over the group, so a
    // hub row of 60-300 entries is a few trips, not a chain of as many; lane 0 of the group owns the node
    int L = 1;
    while (L < 8 && N * (L * 2) <= 1024) L <<= 1;
    const int i = threadIdx.x / L, sub = threadIdx.x % L;
    const bool valid = i < N, owner = valid && sub == 0;
    double* maxrow = reinterpret_cast<double*>(mis_raw);            // N
    double* deg = maxrow + MIS_SMALL_ROWS;                          // N
    int* degi = reinterpret_cast<int*>(deg + MIS_SMALL_ROWS);       // N
    int* scnt = degi + MIS_SMALL_ROWS;                              // N: strong neighbours listed so far
    uint8_t* fC = reinterpret_cast<uint8_t*>(scnt + MIS_SMALL_ROWS);
    uint8_t* fF = fC + MIS_SMALL_ROWS;
    uint8_t* fU = fF + MIS_SMALL_ROWS;
    uint8_t* fS = fU + MIS_SMALL_ROWS;
    uint8_t* fS2 = fS + MIS_SMALL_ROWS;
    unsigned short* sci = reinterpret_cast<unsigned short*>(fS2 + MIS_SMALL_ROWS);   // strong neighbours, row i at [r0, ..)
    const int r0 = valid ? a.rp[i] : 0, r1 = valid ? a.rp[i + 1] : 0;
    auto group_or = [&](bool v) {
        int x = v ? 1 : 0;
        for (int d = 1; d < L; d <<= 1) x |= __shfl_xor(x, d);
        return x != 0;
    };
    // ---- strength.m:7-10 (k_rowmax); eight entries per lane and trip, all loads of a trip in flight
    {
        double mx = 0.0, dg = 0.0;
        for (int t0 = r0 + sub; t0 < r1; t0 += 8 * L) {
            int jj[8];
            double vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * L < r1 ? t0 + u * L : r0;
                jj[u] = a.ci[t];
                vv[u] = a.va[t];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u * L < r1) {
                    if (jj[u] == i)
                        dg = vv[u];
                    else
                        mx = fmax(mx, -vv[u]);
                }
            }
        }
        for (int d = 1; d < L; d <<= 1) {
            mx = fmax(mx, __shfl_xor(mx, d));
            dg += __shfl_xor(dg, d);   // at most one lane holds the diagonal
        }
        if (owner) {
            const double m = mx > 0.0 ? mx : INFINITY;
            maxrow[i] = m;
            a.maxrow[i] = m;
            a.diag[i] = dg;
            degi[i] = 0;
            scnt[i] = 0;
        }
    }
    __syncthreads();
    // ---- mis_set.m:25-29 (k_strong): the mask, its column counts (deg) and row counts; the strong
    // neighbours of row i are listed at sci[r0 ..) in any order (the rounds only ask whether ANY / EVERY
    // neighbour has a property, so the order of the list does not matter)
    {
        const double mr = valid ? maxrow[i] : 1.0;
        for (int t0 = r0 + sub; t0 < r1; t0 += 8 * L) {
            int jj[8];
            double vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * L < r1 ? t0 + u * L : r0;
                jj[u] = a.ci[t];
                vv[u] = a.va[t];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u * L < r1) {
                    const int j = jj[u];
                    bool f = false;
                    if (j != i) {
                        const double sv = (-vv[u]) / fmin(mr, maxrow[j]);
                        f = sv >= a.theta;
                    }
                    a.strong[t0 + u * L] = f ? 1 : 0;
                    if (f) {
                        atomicAdd(&degi[j], 1);
                        sci[r0 + atomicAdd(&scnt[i], 1)] = (unsigned short)j;
                    }
                }
            }
        }
    }
    __syncthreads();   // degi, scnt and the lists are final
    const int rowcnt = valid ? scnt[i] : 0;
    const int d = valid ? degi[i] : 0;
    int nconn = 0;
    const int rank = mis_block_exscan((owner && d > 0) ? 1 : 0, wsum, &nconn);
    if ((double)nconn < 0.25 * sqrt((double)N)) {              // :30-34: the host takes this (rare) branch
        if (threadIdx.x == 0) {
            a.box[16] = 1u;
            a.box[17] = (unsigned)nconn;
            __threadfence_system();
            a.box[0] = a.ticket;
        }
        return;
    }
    // ---- :35-40 (k_deg_init)
    if (owner) {
        double dv = 0.0;
        if (d > 0) {
            const double tie = 0.1 * a.randv[rank];
            dv = (double)d + tie;
        }
        deg[i] = dv;
        fC[i] = 0;
        fF[i] = d == 0 ? 1 : 0;
        fU[i] = 1;
        fS[i] = d > 0 ? 1 : 0;
    }
    __syncthreads();
    // ---- :42-65: the rounds
    const int s0 = r0, s1 = r0 + rowcnt;
    int sumC = 0, sumU = N, rounds = 0;
    uint8_t* cur = fS;
    uint8_t* nxt = fS2;
    while ((double)sumC < (double)N / 2.0 && sumU > a.N0 && rounds <= N + 8) {
        ++rounds;
        if (valid) {                                           // k_mis_sel_kill (:49-52)
            const double di = deg[i];
            if (di > 0.0)
                for (int t = s0 + sub; t < s1; t += L) {
                    const int j = sci[t];
                    if (j > i) {
                        const double dj = deg[j];
                        if (dj > 0.0) {
                            if (di >= dj)
                                cur[j] = 0;
                            else
                                cur[i] = 0;
                        }
                    }
                }
        }
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        bool hit = false;                                      // k_mis_settle (:53-59)
        if (valid)
            for (int t = s0 + sub; t < s1; t += L) {
                const int j = sci[t];
                if (fC[j] || cur[j]) hit = true;
            }
        hit = group_or(hit);
        int c1 = 0, u1 = 0;
        if (owner) {
            const bool c = fC[i] || cur[i];
            const bool f = fF[i] || hit;
            const bool u = !(c || f);
            c1 = c ? 1 : 0;
            u1 = u ? 1 : 0;
            nxt[i] = u ? 1 : 0;
            fU[i] = u ? 1 : 0;
            if (f) fF[i] = 1;
            if (!u) deg[i] = 0.0;
        }
        // (fC is read by the neighbours in this phase: committed after the barrier below)
        const unsigned long long bc = __ballot(c1 != 0), bu = __ballot(u1 != 0);
        if ((threadIdx.x & 63) == 0) {
            if (bc) atomicAdd(&s_cnt[0], __popcll(bc));
            if (bu) atomicAdd(&s_cnt[1], __popcll(bu));
        }
        __syncthreads();
        if (owner && c1) fC[i] = 1;
        sumC = s_cnt[0];
        sumU = s_cnt[1];
        uint8_t* tsw = cur;
        cur = nxt;
        nxt = tsw;
        __syncthreads();
        if (sumU <= a.N0) {                                    // :61-64 (k_mis_absorb)
            if (owner && fU[i]) {
                fC[i] = 1;
                fU[i] = 0;
            }
            sumU = 0;
            __syncthreads();
        }
    }
    // ---- :67 (k_mis_iso), then the C index scan and the consistency count of transfer.m:46-47
    int isc = 0, isf = 0;
    if (owner) {
        isc = fC[i];
        isf = fF[i];
        if (rowcnt == 0) {
            isc = 1;
            isf = 0;
        }
        a.isC[i] = (uint8_t)isc;
        a.isF[i] = (uint8_t)isf;
    }
    int Nc = 0, bad = 0;
    const int cpos = mis_block_exscan(owner ? isc : 0, wsum, &Nc);
    mis_block_exscan((owner && (isc != 0) == (isf != 0)) ? 1 : 0, wsum, &bad);
    if (owner) a.cidx[i] = cpos;
    if (threadIdx.x == 0) {
        a.cidx[N] = Nc;
        a.box[16] = 0u;
        a.box[17] = (unsigned)nconn;
        a.box[18] = (unsigned)Nc;
        a.box[19] = (unsigned)bad;
        a.box[20] = (unsigned)rounds;
        __threadfence_system();
        a.box[0] = a.ticket;
    }
}

// -> true: done (Nc, bad filled, cidx / maxrow / diag written); false: not taken (size, mailbox off) or
// the degenerate branch of :30-34 came up, and the caller runs the launch-per-step form
static bool mis_set_small(ipd_ctx* ctx, const Csr& A, double theta, ipd_rng* rng, uint8_t* isC, uint8_t* isF,
                          uint8_t* strong, double* maxrow, double* diag, int* cidx, int* Nc, int* bad) {
    const int N = A.nr;
    if (N > MIS_SMALL_ROWS || A.nnz > MIS_SMALL_NNZ || N < 1) return false;
    if (const char* e = getenv("IPD_NO_MIS_SMALL"); e && e[0] == '1') return false;
    unsigned ticket = 0;
    if (!ctx->mailbox_begin(&ticket)) return false;
    Arena& tmp = *ctx->scratch;
    // peek at the stream's next N numbers (state restored below; a replay stream may hold fewer)
    std::vector<double> rv((size_t)N, 0.0);
    {
        const bool rp = rng->replay;
        const int64_t have = rp ? std::max<int64_t>(0, (int64_t)rng->values.size() - rng->consumed) : N;
        const int64_t take = std::min<int64_t>(N, have);
        uint32_t mt[624];
        std::memcpy(mt, rng->mt, sizeof(mt));
        const int mti = rng->mti;
        const int64_t consumed = rng->consumed;
        if (take > 0) rng->fill(rv.data(), take);
        std::memcpy(rng->mt, mt, sizeof(mt));
        rng->mti = mti;
        rng->consumed = consumed;
    }
    double* drand = tmp.alloc<double>((size_t)N);
    ctx->upload(drand, rv.data(), (size_t)N);
    MisSmallArgs a;
    a.N = N;
    a.N0 = std::min((int)std::floor(std::sqrt((double)N)) + 1, 25);   // :12
    a.rp = A.rp;
    a.ci = A.ci;
    a.va = A.va;
    a.theta = theta;
    a.randv = drand;
    a.strong = strong;
    a.maxrow = maxrow;
    a.diag = diag;
    a.isC = isC;
    a.isF = isF;
    a.cidx = cidx;
    a.box = ctx->mailbox;
    a.ticket = ticket;
    const size_t lds = 16 * (size_t)MIS_SMALL_ROWS + 8 * (size_t)MIS_SMALL_ROWS + 5 * (size_t)MIS_SMALL_ROWS +
                       2 * (size_t)std::max(A.nnz, 1) + 64;
    IPD_OPTIN_LDS(ctx, k_mis_small, 156 * 1024);
    hipLaunchKernelGGL(k_mis_small, dim3(1), dim3(1024), lds, ctx->stream, a);
    IPD_KERNEL_CHECK();
    unsigned w[5] = {0, 0, 0, 0, 0};
    ctx->mailbox_wait(ticket, w, sizeof(w));
    if (w[0] != 0) return false;                     // degenerate branch: nothing consumed yet
    IPD_REQUIRE((int)w[4] <= N + 8, IPD_E_NUMERIC, "mis_set: no progress");
    std::vector<double> used((size_t)w[1]);
    rng->fill(used.data(), (int64_t)w[1]);           // mis_set.m:35 consumes sum(deg > 0) numbers
    *Nc = (int)w[2];
    *bad = (int)w[3];
    return true;
}

void amg_mis_set(ipd_ctx* ctx, const Csr& A, double theta, ipd_rng* rng, uint8_t* isC,
                 uint8_t* isF, uint8_t* strong_out) {
    IPD_REQUIRE(rng, IPD_E_ARG, "mis_set needs a rand stream");
    IPD_REQUIRE(theta > 0, IPD_E_ARG, "mis_set: theta must be positive");
    const int N = A.nr;
    Arena& tmp = *ctx->scratch;
    uint8_t* strong = strong_out ? strong_out : tmp.alloc<uint8_t>((size_t)A.nnz);
    int* degi = tmp.alloc<int>((size_t)N + 1);
    int* rowcnt = tmp.alloc<int>((size_t)N + 1);
    int* flag = tmp.alloc<int>((size_t)N + 1);
    int* rank = tmp.alloc<int>((size_t)N + 2);
    double* deg = tmp.alloc<double>((size_t)N);
    uint8_t* isU = tmp.alloc<uint8_t>((size_t)N);
    uint8_t* isS = tmp.alloc<uint8_t>((size_t)N);
    int* counts = tmp.alloc<int>(2);
    const int N0 = std::min((int)std::floor(std::sqrt((double)N)) + 1, 25);  // :12
    amg_strength_mask(ctx, A, theta, strong, degi, rowcnt);                  // :25-29
    const int g = elems_grid(N);
    hipLaunchKernelGGL(k_flag_pos, dim3(g), dim3(256), 0, ctx->stream, N, degi, flag);
    IPD_KERNEL_CHECK();
    const int nconn = exclusive_scan_total(ctx, flag, rank, N);
    if ((double)nconn < 0.25 * std::sqrt((double)N)) {                       // :30-34
        std::vector<double> rv((size_t)N0);
        rng->fill(rv.data(), N0);
        std::vector<uint8_t> hc((size_t)N, 0), hf((size_t)N, 1);
        for (int k = 0; k < N0; ++k) {
            long long pick = (long long)std::ceil(rv[k] * (double)N) - 1;
            if (pick < 0) pick = 0;  // rand never returns exactly 0; guard anyway
            if (pick >= N) pick = N - 1;
            hc[(size_t)pick] = 1;
            hf[(size_t)pick] = 0;
        }
        ctx->upload(isC, hc.data(), (size_t)N);
        ctx->upload(isF, hf.data(), (size_t)N);
        return;
    }
    std::vector<double> rv((size_t)nconn);
    rng->fill(rv.data(), nconn);                                             // :35
    double* drand = tmp.alloc<double>((size_t)nconn);
    ctx->upload(drand, rv.data(), (size_t)nconn);
    uint8_t* isS2 = tmp.alloc<uint8_t>((size_t)N);
    hipLaunchKernelGGL(k_deg_init, dim3(g), dim3(256), 0, ctx->stream, N, degi, rank, drand, deg,
                       isC, isF, isU, isS);
    IPD_KERNEL_CHECK();
    int sumC = 0, sumU = N;
    int rounds = 0;
    while ((double)sumC < (double)N / 2.0 && sumU > N0) {                    // :42
        IPD_REQUIRE(++rounds <= N + 8, IPD_E_NUMERIC, "mis_set: no progress");
        // two launches per round: the edge-wise selection, then everything that follows it
        hipLaunchKernelGGL(k_mis_sel_kill, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp,
                           A.ci, strong, deg, isS, counts);
        hipLaunchKernelGGL(k_mis_settle, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp, A.ci,
                           strong, isS, isC, isF, isU, deg, isS2, counts);
        IPD_KERNEL_CHECK();
        std::swap(isS, isS2);
        int hc[2];
        ctx->fetch(counts, hc, 2);
        sumC = hc[0];
        sumU = hc[1];
        if (sumU <= N0) {                                                    // :61-64
            hipLaunchKernelGGL(k_mis_absorb, dim3(g), dim3(256), 0, ctx->stream, N, isU, isC);
            IPD_KERNEL_CHECK();
            sumU = 0;
        }
    }
    hipLaunchKernelGGL(k_mis_iso, dim3(g), dim3(256), 0, ctx->stream, N, rowcnt, isC, isF);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// cf_split                                                (AMG/cf_split.m:6-16)
// ---------------------------------------------------------------------------
// The sequential greedy pass (k = 1..N: an unvisited k becomes C and its
// neighbours F) yields the lexicographically-first maximal independent set:
// k is C iff no lower-indexed neighbour is C.  Parallel form, one workgroup:
// an undecided node becomes F as soon as a lower neighbour is C, and C as soon
// as every lower neighbour is F.  Decisions are final, so in-place updates and
// any interleaving give the identical (bit-exact) result.
__global__ __launch_bounds__(1024) void k_cf_split(int n, const int* __restrict__ rp,
                                                   const int* __restrict__ ci,
                                                   uint8_t* __restrict__ state /*0 U,1 C,2 F*/,
                                                   int* __restrict__ rounds_out) {
    __shared__ int pending;
    int rounds = 0;
    while (true) {
        if (threadIdx.x == 0) pending = 0;
        __syncthreads();
        bool mine = false;
        for (int k = threadIdx.x; k < n; k += 1024) {
            if (state[k] != 0) continue;
            bool anyC = false, anyU = false;
            for (int t = rp[k]; t < rp[k + 1]; ++t) {
                const int j = ci[t];
                if (j >= k) break;  // columns ascend: only lower neighbours matter
                const uint8_t sj = state[j];
                anyC |= (sj == 1);
                anyU |= (sj == 0);
            }
            if (anyC)
                state[k] = 2;
            else if (!anyU)
                state[k] = 1;
            else
                mine = true;
        }
        if (mine) pending = 1;
        __syncthreads();
        ++rounds;
        const int p = pending;
        __syncthreads();
        if (!p || rounds > n + 2) break;  // every round decides >= 1 node; the bound is a hang guard
    }
    if (threadIdx.x == 0) *rounds_out = rounds;
}

__global__ void k_state_to_masks(int n, const uint8_t* __restrict__ state,
                                 uint8_t* __restrict__ isC, uint8_t* __restrict__ isF) {
    THREAD_ELEMS(i, n) {
        isC[i] = state[i] == 1;
        isF[i] = state[i] == 2;
    }
}

static void amg_cf_split(ipd_ctx* ctx, const Csr& S, uint8_t* isC, uint8_t* isF) {
    Arena& tmp = *ctx->scratch;
    uint8_t* state = tmp.alloc<uint8_t>((size_t)S.nr);
    int* rounds = tmp.alloc<int>(1);
    IPD_HIP(hipMemsetAsync(state, 0, (size_t)std::max(S.nr, 1), ctx->stream));
    hipLaunchKernelGGL(k_cf_split, dim3(1), dim3(1024), 0, ctx->stream, S.nr, S.rp, S.ci, state,
                       rounds);
    IPD_KERNEL_CHECK();
    hipLaunchKernelGGL(k_state_to_masks, dim3(elems_grid(S.nr)), dim3(256), 0, ctx->stream, S.nr,
                       state, isC, isF);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// interpolation                                          (AMG/transfer.m:19-63)
// ---------------------------------------------------------------------------
// level 1 of a bigraph: F = first nf rows, W = (-Aff)\Afc with Aff diagonal
// (transfer.m:20-25); one lane per row, sequential, so the row sum used by the
// isnsp normalisation (:22-24) is accumulated in ascending column order.
__global__ __launch_bounds__(256) void k_bigph_count(int N, int nf, const int* __restrict__ rp,
                                                     const int* __restrict__ ci, int* rowlen,
                                                     const ScanTail st, int* __restrict__ badp) {
    // st.out != NULL: biased counts with the flag in bit 30, scanned and posted by the launch's tail;
    // otherwise plain counts (k_bigph_fill scans them) and the flag at *badp
    WAVE_ROWS(i, N) {
        if (i >= nf) {
            if (lane == 0) {
                if (st.out)
                    scan_put(rowlen, i, 1);
                else
                    rowlen[i] = 1;
            }
            continue;
        }
        int c = 0;
        bool bad = false;   // Aff is not diagonal
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) {
            const int j = ci[t];
            if (j >= nf)
                ++c;
            else if (j != i)
                bad = true;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        bad = __any(bad);
        if (lane == 0) {
            if (st.out)
                scan_put(rowlen, i, c, bad);
            else {
                rowlen[i] = c;
                if (bad) *badp = 1;
            }
        }
    }
    scan_tail(st);   // P's row pointers; total and flag to the host
}

// One wave per row.  Lanes write the entries in parallel; the row sum of the isnsp
// normalisation is accumulated by lane 0 alone, sequentially in ascending column order
// (recomputing the quotients it sums, so it does not depend on the other lanes' stores).
__global__ __launch_bounds__(256) void k_bigph_fill(int N, int nf, int isnsp,
                                                    const int* __restrict__ rp,
                                                    const int* __restrict__ ci,
                                                    const double* __restrict__ va,
                                                    const int* prp,
                                                    int* __restrict__ pci, double* __restrict__ pva,
                                                    uint8_t* __restrict__ cmask,
                                                    const int* __restrict__ head_cnt, int* head_rp,
                                                    int* head_total) {
    __shared__ ScanHeadLds L;   // head_cnt != NULL: plain counts, scanned here (scan_head; N <= SCAN_HEAD_MAX)
    if (head_cnt) {
        scan_head(head_cnt, N, head_rp, head_total, L);
        prp = L.rp;
    }
    WAVE_ROWS(i, N) {
        const int pos0 = prp[i];
        if (i >= nf) {
            if (lane == 0) {
                pci[pos0] = i - nf;
                pva[pos0] = 1.0;
                cmask[i] = 1;
            }
            continue;
        }
        const int b = rp[i], e = rp[i + 1];
        // the diagonal, and the first entry of the C block (columns ascend: a suffix)
        double dii = 0.0;
        int first = e;
        for (int t = b + lane; t < e; t += 64) {
            const int j = ci[t];
            if (j == i) dii = va[t];
            if (j >= nf) first = min(first, t);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            dii += __shfl_xor(dii, d);  // one lane holds it
            first = min(first, __shfl_xor(first, d));
        }
        const double nd = -dii;
        double s = 1.0;
        if (isnsp == 1) {
            double acc = 0.0;
            if (lane == 0)
                for (int t = first; t < e; ++t) acc = acc + va[t] / nd;
            s = __shfl(acc, 0);
        }
        for (int t = first + lane; t < e; t += 64) {
            const double w = va[t] / nd;
            pci[pos0 + (t - first)] = ci[t] - nf;
            pva[pos0 + (t - first)] = isnsp == 1 ? w / s : w;
        }
        if (lane == 0) cmask[i] = 0;
    }
}

// General level (transfer.m:41-63).  One single-wave workgroup per row keeps two
// dense coarse rows in LDS: acc1 = W1(i,:) = Afc(i,:)/(-a_ii), acc2 = W2(i,:) =
// sum_k X(i,k) W1(k,:) with X = ((-Dff)\(Aff.*(I+As_FF))), k ascending; the row of
// W is W1 + 0.5*W2 (the always-true test at transfer.m:54).
__global__ __launch_bounds__(256) void k_build_W(int N, int Nc, const int* __restrict__ rp,
                                                const int* __restrict__ ci,
                                                const double* __restrict__ va,
                                                const double* __restrict__ diag,
                                                const uint8_t* __restrict__ strong,
                                                const uint8_t* __restrict__ isC,
                                                const uint8_t* __restrict__ isF,
                                                const int* __restrict__ cidx,
                                                double* __restrict__ dense,
                                                int* __restrict__ rowcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* acc2 = reinterpret_cast<double*>(smem_raw);
    double* acc1 = acc2 + Nc;
    const int lane = threadIdx.x, T = blockDim.x;  // 64 or 256 threads per row
    __shared__ int wcnt[4];
    for (int i = blockIdx.x; i < N; i += gridDim.x) {
        double* drow = dense + (size_t)i * Nc;
        if (isC[i]) {  // identity row of P = [W; I]
            const int me = cidx[i];
            for (int c = lane; c < Nc; c += T) drow[c] = (c == me) ? 1.0 : 0.0;
            if (lane == 0) rowcnt[i] = 1;
            continue;
        }
        for (int c = lane; c < Nc; c += T) {
            acc1[c] = 0.0;
            acc2[c] = 0.0;
        }
        __syncthreads();
        const double ndi = -diag[i];
        const int b = rp[i], e = rp[i + 1];
        for (int t = b + lane; t < e; t += T) {
            const int j = ci[t];
            if (isC[j]) acc1[cidx[j]] = va[t] / ndi;
        }
        for (int t = b; t < e; ++t) {
            const int k = ci[t];
            if (isF[k] && (k == i || strong[t])) {
                const double x = va[t] / ndi;
                const double ndk = -diag[k];
                for (int u = rp[k] + lane; u < rp[k + 1]; u += T) {
                    const int j = ci[u];
                    if (isC[j]) {
                        const double w1 = va[u] / ndk;
                        const double prod = x * w1;
                        const int c = cidx[j];
                        acc2[c] = acc2[c] + prod;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        int nz = 0;
        for (int c = lane; c < Nc; c += T) {
            const double half = 0.5 * acc2[c];
            const double v = acc1[c] + half;
            drow[c] = v;
            nz += (v != 0.0);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
        if ((lane & 63) == 0) wcnt[lane >> 6] = nz;
        __syncthreads();
        if (lane == 0) {
            int tot = 0;
            for (int w = 0; w < (T >> 6); ++w) tot += wcnt[w];
            rowcnt[i] = tot;   // (plain counts: the compaction or a scan launch turns them into row pointers)
        }
        __syncthreads();
    }
}

// The same rows with ONE WAVE per row, pipelined (the form of k_spgemm_rows_w).  k_build_W's loop over the
// strong F neighbours k is a chain of four dependent global round trips and a barrier per neighbour -- column,
// then diag / row range of k, then k's entries, then their C flags and indices: 0.5 us each, 44 us for the
// 90-entry rows of a 100-row level.  Here the lanes read the metadata of 64 entries of row i at once; the
// strong F neighbours are listed in LDS in ascending order, a row of k longer than 64 entries as up to four
// consecutive 64-entry pieces; and the pieces D ahead -- already filtered to C columns and divided by -a_kk --
// are in flight while the current D are applied.  Every acc2[c] still receives x(i,k) * w1(k,c) one term at a
// time in ascending k (the pieces of one k touch distinct columns), and a wave's LDS operations execute in
// program order: the bits equal k_build_W's.
constexpr int BW_PIECES = 4;   // 64-entry pieces of a neighbour's row that are prefetched (the rest: a plain loop)
template <int D>
__global__ __launch_bounds__(64) void k_build_W_w(int N, int Nc, const int* __restrict__ rp,
                                                  const int* __restrict__ ci,
                                                  const double* __restrict__ va,
                                                  const double* __restrict__ diag,
                                                  const uint8_t* __restrict__ strong,
                                                  const uint8_t* __restrict__ isC,
                                                  const uint8_t* __restrict__ isF,
                                                  const int* __restrict__ cidx,
                                                  double* __restrict__ dense,
                                                  int* __restrict__ rowcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* acc2 = reinterpret_cast<double*>(smem_raw);
    double* acc1 = acc2 + Nc;
    __shared__ int s_yb[64 * BW_PIECES], s_yn[64 * BW_PIECES], s_te[64 * BW_PIECES];
    __shared__ double s_x[64 * BW_PIECES], s_nd[64 * BW_PIECES];
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < N; i += gridDim.x) {
        double* drow = dense + (size_t)i * Nc;
        if (isC[i]) {  // identity row of P = [W; I]
            const int me = cidx[i];
            for (int c = lane; c < Nc; c += 64) drow[c] = (c == me) ? 1.0 : 0.0;
            if (lane == 0) rowcnt[i] = 1;
            continue;
        }
        for (int c = lane; c < Nc; c += 64) {
            acc1[c] = 0.0;
            acc2[c] = 0.0;
        }
        __syncthreads();
        const double ndi = -diag[i];
        const int b = rp[i], e = rp[i + 1];
        for (int e0 = b; e0 < e; e0 += 64) {
            const int t = e0 + lane;
            const bool mine = t < e;
            const int kk = mine ? ci[t] : 0;
            const double av = mine ? va[t] : 0.0;
            const bool kC = mine && isC[kk];
            const bool take = mine && isF[kk] && (kk == i || strong[t]);
            const double xq = av / ndi;                    // W1(i, .) entry or X(i, k)
            if (kC) acc1[cidx[kk]] = xq;
            // the strong F neighbours of this batch in ascending order, a row of more than 64 entries as pieces
            const int yb0 = take ? rp[kk] : 0;
            const int ylen = take ? rp[kk + 1] - yb0 : 0;
            const int npc = take ? min(BW_PIECES, (ylen + 63) >> 6) : 0;
            int r0 = npc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(r0, d);
                if (lane >= d) r0 += y;
            }
            const int cnt = __shfl(r0, 63);
            r0 -= npc;
            if (take) {
                const double ndk = -diag[kk];
                for (int c = 0; c < npc; ++c) {
                    s_yb[r0 + c] = yb0 + 64 * c;
                    s_yn[r0 + c] = min(64, ylen - 64 * c);
                    s_te[r0 + c] = (c == npc - 1 && ylen > 64 * BW_PIECES) ? yb0 + ylen : 0;
                    s_x[r0 + c] = xq;
                    s_nd[r0 + c] = ndk;
                }
            }
            __syncthreads();
            int jA[D], jB[D], tA[D], tB[D];
            double vA[D], vB[D], xA[D], xB[D];
            // pieces u0 .. u0+D-1: C column index (or -1) and w1 = a_kj / -a_kk of the lane's entry; x(i,k) and the
            // long-row mark ride along, so that applying a piece is one LDS read-add-write and nothing else
            // (branch-free in two rounds -- all the pieces' entries, then all their C flags and indices -- so that the
            // D pieces' dependent loads overlap: under `if (lane < n) { j = ...; if (isC[j]) ... }` each piece
            // waited for its own two round trips in turn, 1.5 us per call)
            auto load = [&](int u0, int* jj, double* vv, double* xs, int* ts) __attribute__((always_inline)) {
                int jt[D], nn[D];
                double nd[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int u = min(u0 + d, cnt - 1);   // uniform; cnt > 0 here
                    const int bb = s_yb[u];
                    nn[d] = u0 + d < cnt ? s_yn[u] : 0;
                    xs[d] = s_x[u];
                    ts[d] = u0 + d < cnt ? s_te[u] : 0;
                    nd[d] = s_nd[u];
                    const int idx = bb + min(lane, max(nn[d], 1) - 1);
                    jt[d] = ci[idx];
                    vv[d] = va[idx];
                }
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const bool c = isC[jt[d]] != 0;
                    const int cd = cidx[jt[d]];
                    jj[d] = (lane < nn[d] && c) ? cd : -1;
                    vv[d] = vv[d] / nd[d];
                }
            };
            auto apply = [&](int u0, const int* jj, const double* vv, const double* xs, const int* ts)
                             __attribute__((always_inline)) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int u = u0 + d;   // uniform
                    if (u < cnt) {
                        const double x = xs[d];
                        if (jj[d] >= 0) {
                            const double prod = x * vv[d];
                            acc2[jj[d]] = acc2[jj[d]] + prod;
                        }
                        const int te = ts[d];
                        if (te) {   // the rest of a very long row (distinct columns: lane order is free)
                            const double ndk = s_nd[u];
                            for (int q = s_yb[u] + 64 + lane; q < te; q += 64) {
                                const int j = ci[q];
                                if (isC[j]) {
                                    const double w1 = va[q] / ndk;
                                    const double prod = x * w1;
                                    const int c = cidx[j];
                                    acc2[c] = acc2[c] + prod;
                                }
                            }
                        }
                    }
                }
            };
            if (cnt > 0) {
                load(0, jA, vA, xA, tA);
                for (int u0 = 0; u0 < cnt; u0 += 2 * D) {
                    load(u0 + D, jB, vB, xB, tB);
                    apply(u0, jA, vA, xA, tA);
                    load(u0 + 2 * D, jA, vA, xA, tA);
                    apply(u0 + D, jB, vB, xB, tB);
                }
            }
            __syncthreads();   // (the lists are rewritten by the next batch)
        }
        int nz = 0;
        for (int c = lane; c < Nc; c += 64) {
            const double half = 0.5 * acc2[c];
            const double v = acc1[c] + half;
            drow[c] = v;
            nz += (v != 0.0);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d);
        if (lane == 0) rowcnt[i] = nz;
        __syncthreads();
    }
}

// The same rows through the ordered product of ipd_sparse.hip, for levels whose rows are long
// (filled-in level 2 under dense masks): W1 and X are written as CSR matrices over all N rows
// (C rows empty), W2 = X*W1 is one csr_spgemm (which switches to register tiles when that is
// faster), and the rows of W are put together in a dense scratch.  Every W2(i,c) still receives
// x(i,k)*w1(k,c) one term at a time in ascending k, so the bits equal k_build_W's.
__global__ __launch_bounds__(256) void k_w_split_count(int N, const int* __restrict__ rp,
                                                      const int* __restrict__ ci,
                                                      const uint8_t* __restrict__ strong,
                                                      const uint8_t* __restrict__ isC,
                                                      const uint8_t* __restrict__ isF,
                                                      int* cnt1,
                                                      int* cntx, const ScanTail st) {
    WAVE_ROWS(i, N) {
        int n1 = 0, nx = 0;
        if (!isC[i])
            for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) {
                const int j = ci[t];
                n1 += isC[j] != 0;
                nx += isF[j] && (j == i || strong[t]);
            }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            n1 += __shfl_xor(n1, d);
            nx += __shfl_xor(nx, d);
        }
        if (lane == 0) {
            if (st.out) {
                scan_put(cnt1, i, n1);
                scan_put(cntx, i, nx);
            } else {
                cnt1[i] = n1;
                cntx[i] = nx;
            }
        }
    }
    scan_tail(st);   // the row pointers of W1 and X, both totals in one mailbox message
}

__global__ __launch_bounds__(256) void k_w_split_fill(int N, const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const double* __restrict__ va,
                                                     const double* __restrict__ diag,
                                                     const uint8_t* __restrict__ strong,
                                                     const uint8_t* __restrict__ isC,
                                                     const uint8_t* __restrict__ isF,
                                                     const int* __restrict__ cidx,
                                                     const int* rp1,
                                                     int* __restrict__ ci1, double* __restrict__ va1,
                                                     const int* rpx,
                                                     int* __restrict__ cix, double* __restrict__ vax,
                                                     const int* __restrict__ head1,
                                                     const int* __restrict__ headx, int* rp1_out,
                                                     int* rpx_out) {
    // head1 != NULL: rp1 / rpx are still k_w_split_count's plain counts, scanned here (scan_head; N <= SCAN_HEAD_MAX)
    __shared__ ScanHeadLds L1, Lx;
    if (head1) {
        scan_head(head1, N, rp1_out, nullptr, L1);
        scan_head(headx, N, rpx_out, nullptr, Lx);
        rp1 = L1.rp;
        rpx = Lx.rp;
    }
    WAVE_ROWS(i, N) {
        if (isC[i]) continue;
        const double ndi = -diag[i];
        int b1 = rp1[i], bx = rpx[i];
        const int b = rp[i], e = rp[i + 1];
        for (int t0 = b; t0 < e; t0 += 64) {
            const int t = t0 + lane;
            const int j = t < e ? ci[t] : 0;
            const bool f1 = t < e && isC[j];
            const bool fx = t < e && isF[j] && (j == i || strong[t]);
            const double v = t < e ? va[t] / ndi : 0.0;
            const unsigned long long m1 = __ballot(f1), mx = __ballot(fx);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (f1) {
                const int pos = b1 + __popcll(m1 & below);
                ci1[pos] = cidx[j];
                va1[pos] = v;
            }
            if (fx) {
                const int pos = bx + __popcll(mx & below);
                cix[pos] = j;
                vax[pos] = v;
            }
            b1 += __popcll(m1);
            bx += __popcll(mx);
        }
    }
}

// dense rows hold W1; add half of W2 on the F rows, write the identity entry on the C rows
__global__ __launch_bounds__(256) void k_w_combine(int N, int Nc, const uint8_t* __restrict__ isC,
                                                  const int* __restrict__ cidx,
                                                  const int* __restrict__ rp2,
                                                  const int* __restrict__ ci2,
                                                  const double* __restrict__ va2,
                                                  double* __restrict__ dense) {
    WAVE_ROWS(i, N) {
        double* drow = dense + (size_t)i * Nc;
        if (isC[i]) {
            if (lane == 0) drow[cidx[i]] = 1.0;
            continue;
        }
        for (int t = rp2[i] + lane; t < rp2[i + 1]; t += 64) {
            const int c = ci2[t];
            const double half = 0.5 * va2[t];
            drow[c] = drow[c] + half;
        }
    }
}

__global__ void k_u8_to_flag(int n, const uint8_t* __restrict__ a, int* __restrict__ f) {
    THREAD_ELEMS(i, n) f[i] = a[i] ? 1 : 0;
}
__global__ void k_count_bad_split(int n, const uint8_t* __restrict__ isC,
                                  const uint8_t* __restrict__ isF, int* __restrict__ bad) {
    THREAD_ELEMS(i, n) if ((isC[i] != 0) == (isF[i] != 0)) atomicAdd(bad, 1);
}

// dense rows -> CSR, one wave per row.  normF != NULL: D = diag(W*1); W = D\W on the F rows (transfer.m:60-62) --
// the row sum runs over the stored entries one at a time in ascending column order (the order MATLAB's sum over
// a sparse row takes), read out of the lanes that hold them.
__device__ __forceinline__ double su_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__global__ __launch_bounds__(256) void k_dense_compact2(int nr, int nc,
                                                        const double* __restrict__ dense,
                                                        const int* rp,
                                                        int* __restrict__ ci,
                                                        double* __restrict__ va,
                                                        const int* __restrict__ head_cnt, int* head_rp,
                                                        int* head_total,
                                                        const uint8_t* __restrict__ normF) {
    __shared__ ScanHeadLds L;   // head_cnt != NULL: plain counts, scanned here (scan_head; nr <= SCAN_HEAD_MAX)
    if (head_cnt) {
        scan_head(head_cnt, nr, head_rp, head_total, L);
        rp = L.rp;
    }
    WAVE_ROWS(i, nr) {
        int base = rp[i];
        const double* drow = dense + (size_t)i * nc;
        double s = 1.0;
        const bool norm = normF && normF[i];
        if (norm) {
            double acc = 0.0;
            for (int j0 = 0; j0 < nc; j0 += 64) {
                const int j = j0 + lane;
                const double v = j < nc ? drow[j] : 0.0;
                unsigned long long mask = __ballot(v != 0.0);
                while (mask) {
                    const int l = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    acc = acc + su_readlane(v, l);
                }
            }
            s = acc;
        }
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const double v = j < nc ? drow[j] : 0.0;
            const bool nzf = v != 0.0;
            const unsigned long long mask = __ballot(nzf);
            if (nzf) {
                const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                ci[pos] = j;
                va[pos] = norm ? v / s : v;
            }
            base += __popcll(mask);
        }
    }
}

// ideal interpolation (transfer.m:57-58): Aff and Afc as dense row-major arrays, one wave per row
__global__ __launch_bounds__(256) void k_ideal_split(int N, int Nf, int Nc, const int* __restrict__ rp,
                                                     const int* __restrict__ ci,
                                                     const double* __restrict__ va,
                                                     const uint8_t* __restrict__ isF,
                                                     const int* __restrict__ fidx,
                                                     const int* __restrict__ cidx,
                                                     double* __restrict__ Aff, double* __restrict__ Afc) {
    WAVE_ROWS(i, N) {
        if (!isF[i]) continue;
        const size_t r = (size_t)fidx[i];
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) {
            const int j = ci[t];
            if (isF[j])
                Aff[r * Nf + fidx[j]] = va[t];
            else
                Afc[r * Nc + cidx[j]] = va[t];
        }
    }
}
// rows of Pro in the original ordering: F rows = -(Aff \ Afc), C rows = identity (Pro(p,:) = P, :63)
__global__ void k_ideal_rows(int N, int Nc, const uint8_t* __restrict__ isF, const int* __restrict__ fidx,
                             const int* __restrict__ cidx, const double* __restrict__ X,
                             double* __restrict__ dense) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)N * Nc;
         e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / Nc), c = (int)(e % Nc);
        dense[e] = isF[i] ? -X[(size_t)fidx[i] * Nc + c] : (c == cidx[i] ? 1.0 : 0.0);
    }
}


void amg_transfer(ipd_ctx* ctx, Arena& dst, const Csr& A, const AmgOpts& o, int level,
                  ipd_rng* rng, Csr* Ac, Csr* Pout, Csr* Ptout, uint8_t* cmask, Csr* T1out) {
    IPD_REQUIRE(A.nr == A.nc, IPD_E_ARG, "transfer: A must be square");
    IPD_REQUIRE(&dst != ctx->scratch.get(), IPD_E_ARG, "transfer: dst must not be the scratch arena");
    CallScope scope(ctx);  // temporaries die with this call; results live in dst
    Arena& tmp = *ctx->scratch;
    const int N = A.nr;
    Csr P;
    P.nr = N;
    bool lazy = false;        // the interpolation's entry count stays on the device (see "lazy counts" below)
    int* counts = nullptr;    // device: entries of P, P'A, Ac
    if (level == 1 && o.bigph) {                                             // transfer.m:19-25
        const int nf = (int)o.fnode;
        IPD_REQUIRE(nf > 0 && nf < N, IPD_E_ARG, "transfer: fnode must satisfy 0 < fnode < N");
        P.nc = N - nf;
        P.rp = dst.alloc<int>((size_t)N + 1);
        // lazy counts (see below): the fill scans the plain row lengths itself, P's arrays are sized by A's entry
        // count, and count and "Aff is not diagonal" flag are fetched with the products' counts
        lazy = level < 40 && ctx->xfer_hint[level][0] > 0 && ctx->xfer_hint[level][1] > 0 &&
               ctx->xfer_hint[level][2] > 0 && N <= SCAN_HEAD_MAX && (size_t)N * (size_t)P.nc <= SPGEMM_LAZY_MAX &&
               (size_t)P.nc * (size_t)P.nc <= SPGEMM_LAZY_MAX;
        const int* head = nullptr;
        if (lazy) {
            counts = zeroed<int>(ctx, 6);
            int* rowlen = tmp.alloc<int>((size_t)N + 1);
            hipLaunchKernelGGL(k_bigph_count, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, nf,
                               A.rp, A.ci, rowlen, ScanTail(), counts + 3);
            IPD_KERNEL_CHECK();
            P.nnz = (int)std::min<size_t>((size_t)A.nnz + (size_t)N, (size_t)N * (size_t)P.nc);
            head = rowlen;
        } else {   // entry count and flag: one launch, ONE round trip
            int* rowlen = zeroed<int>(ctx, (size_t)N + 1);   // (biased counts: see ScanTail)
            TailTotal tt(ctx, rowlen, P.rp, N);
            hipLaunchKernelGGL(k_bigph_count, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, nf,
                               A.rp, A.ci, rowlen, tt.t, (int*)nullptr);
            IPD_KERNEL_CHECK();
            int h2[2] = {0, 0};
            tt.wait(h2);
            P.nnz = h2[0];
            IPD_REQUIRE(h2[1] == 0, IPD_E_UNSUPPORTED,
                        "transfer: bigph level 1 needs a diagonal Aff block (transfer.m:20-21)");
        }
        P.ci = dst.alloc<int>((size_t)P.nnz);
        P.va = dst.alloc<double>((size_t)P.nnz);
        hipLaunchKernelGGL(k_bigph_fill, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, nf,
                           o.isnsp, A.rp, A.ci, A.va, (const int*)P.rp, P.ci, P.va, cmask, head, P.rp,
                           head ? counts : (int*)nullptr);
        IPD_KERNEL_CHECK();
    } else {                                                                 // transfer.m:41-63
        uint8_t* isC = cmask;
        uint8_t* isF = tmp.alloc<uint8_t>((size_t)N);
        uint8_t* strong = tmp.alloc<uint8_t>((size_t)std::max(A.nnz, 1));
        int* cidx = tmp.alloc<int>((size_t)N + 2);  // cidx[N] = Nc, cidx[N+1] = #bad
        double* maxrow = tmp.alloc<double>((size_t)N);
        double* diag = tmp.alloc<double>((size_t)N);
        int meta[2] = {0, 0};
        const bool small_done = mis_set_small(ctx, A, o.theta, rng, isC, isF, strong, maxrow, diag, cidx,
                                              &meta[0], &meta[1]);
        if (!small_done) {
            amg_mis_set(ctx, A, o.theta, rng, isC, isF, strong);             // :41
            int* flag = tmp.alloc<int>((size_t)N + 1);
            IPD_HIP(hipMemsetAsync(cidx + N + 1, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(k_u8_to_flag, dim3(elems_grid(N)), dim3(256), 0, ctx->stream, N, isC,
                               flag);
            hipLaunchKernelGGL(k_count_bad_split, dim3(elems_grid(N)), dim3(256), 0, ctx->stream, N, isC,
                               isF, cidx + N + 1);
            IPD_KERNEL_CHECK();
            exclusive_scan_i32(ctx, flag, cidx, N);
            ctx->fetch(cidx + N, meta, 2);
        }
        const int Nc = meta[0];
        IPD_REQUIRE(meta[1] == 0, IPD_E_NUMERIC,
                    "mis_set left nodes in neither/both of the C and F sets (SURVEY A-6)");
        IPD_REQUIRE(Nc > 0, IPD_E_NUMERIC, "transfer: empty coarse set");
        IPD_REQUIRE((size_t)Nc * 16 <= 128 * 1024, IPD_E_LIMIT,
                    "transfer: more than 8192 coarse nodes on a non-bigraph level");
        P.nc = Nc;
        // lazy counts (see below): the previous hierarchy left estimates for this level and the dense bounds
        // of P, P'A and Ac are small
        lazy = level >= 1 && level < 40 && ctx->xfer_hint[level][0] > 0 && ctx->xfer_hint[level][1] > 0 &&
               ctx->xfer_hint[level][2] > 0 && (size_t)N * (size_t)Nc <= SPGEMM_LAZY_MAX &&
               (size_t)Nc * (size_t)Nc <= SPGEMM_LAZY_MAX;
        if (lazy) counts = zeroed<int>(ctx, 6);   // entries of P, P'A, Ac; a flag (bigraph level: Aff not diagonal); longest row of P'A
        if (!small_done) {
            hipLaunchKernelGGL(k_rowmax, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp, A.ci,
                               A.va, maxrow, diag);
            IPD_KERNEL_CHECK();
        }
        const size_t dense_elems = (size_t)N * (size_t)Nc;
        IPD_REQUIRE(dense_elems * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                    "transfer: dense interpolation scratch above 2 GiB");
        // very long rows (filled-in level 2 under dense masks): the product form (see k_w_split_count), whose
        // product can run on register tiles; otherwise one kernel (k_build_W_w)
        bool split = (double)A.nnz / std::max(N, 1) >= 256.0;
        if (const char* e = getenv("IPD_INTERP")) split = !strcmp(e, "split");
        // (the product form adds into rows that start out as zeros)
        double* dense = (split && o.inter < 2) ? zeroed<double>(ctx, dense_elems) : tmp.alloc<double>(dense_elems);
        // P's row pointers.  With a lazy count the compaction scans the plain counts on its way in (scan_head).
        // Otherwise the host needs the total: the product form and the ideal interpolation count the rows in a
        // 256-thread launch whose tail scans and posts it (ScanTail); k_build_W's one-wave workgroups are followed
        // by a scan launch.
        const bool head_ok = lazy && N <= SCAN_HEAD_MAX;
        const bool wtail = !head_ok && (o.inter >= 2 || split);
        int* rowcnt = wtail ? zeroed<int>(ctx, (size_t)N + 1) : tmp.alloc<int>((size_t)N + 1);
        P.rp = dst.alloc<int>((size_t)N + 1);
        ScanTail pt;
        std::unique_ptr<TailTotal> ptt;
        if (wtail) {
            if (lazy)
                pt = scan_tail_lazy(rowcnt, P.rp, N, counts);
            else {
                ptt.reset(new TailTotal(ctx, rowcnt, P.rp, N));
                pt = ptt->t;
            }
        }
        if (o.inter >= 2) {                                                  // :57-58  W = -Aff \ Afc
            // MATLAB solves with the sparse Aff (CHOLMOD); here a dense Cholesky of the F-F block
            // (a principal block of the SPD level matrix) with the Nc columns of Afc as right-hand
            // sides (csrc/ipd_dense.hip).  Cold path.
            const int Nf = N - Nc;
            IPD_REQUIRE((size_t)Nf * Nf * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                        "transfer: dense Aff of the ideal interpolation above 2 GiB");
            int* fflag = tmp.alloc<int>((size_t)N + 1);
            int* fidx = tmp.alloc<int>((size_t)N + 1);
            hipLaunchKernelGGL(k_u8_to_flag, dim3(elems_grid(N)), dim3(256), 0, ctx->stream, N, isF, fflag);
            IPD_KERNEL_CHECK();
            exclusive_scan_i32(ctx, fflag, fidx, N);
            double* Aff = tmp.alloc<double>((size_t)std::max(Nf, 1) * std::max(Nf, 1));
            double* Afc = tmp.alloc<double>((size_t)std::max(Nf, 1) * Nc);
            IPD_HIP(hipMemsetAsync(Aff, 0, sizeof(double) * (size_t)Nf * Nf, ctx->stream));
            IPD_HIP(hipMemsetAsync(Afc, 0, sizeof(double) * (size_t)Nf * Nc, ctx->stream));
            if (Nf > 0) {
                hipLaunchKernelGGL(k_ideal_split, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, Nf, Nc,
                                   A.rp, A.ci, A.va, isF, fidx, cidx, Aff, Afc);
                IPD_KERNEL_CHECK();
                dense_chol_factor(ctx, Aff, Nf, Nf);
                dense_chol_solve(ctx, Aff, Nf, Nf, Afc, Nc, Nc);
            }
            hipLaunchKernelGGL(k_ideal_rows, dim3((int)std::min<size_t>((dense_elems + 255) / 256, 8192)),
                               dim3(256), 0, ctx->stream, N, Nc, isF, fidx, cidx, (const double*)Afc, dense);
            IPD_KERNEL_CHECK();
            dense_rowcount(ctx, N, Nc, Nc, dense, rowcnt, pt);
        } else if (split) {
            Csr W1, X, W2;
            W1.nr = N;
            W1.nc = Nc;
            X.nr = X.nc = N;
            W1.rp = tmp.alloc<int>((size_t)N + 1);
            X.rp = tmp.alloc<int>((size_t)N + 1);
            if (head_ok) {
                // no round trip: W1 and X are sub-patterns of A (arrays sized by A's entry count, the fill scans
                // the plain counts itself) and the product's count stays on the device
                int* cnt1 = tmp.alloc<int>((size_t)N + 1);
                int* cntx = tmp.alloc<int>((size_t)N + 1);
                hipLaunchKernelGGL(k_w_split_count, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp,
                                   A.ci, strong, isC, isF, cnt1, cntx, ScanTail());
                IPD_KERNEL_CHECK();
                W1.nnz = X.nnz = std::max(A.nnz, 1);   // (allocation bound)
                W1.ci = tmp.alloc<int>((size_t)W1.nnz);
                W1.va = tmp.alloc<double>((size_t)W1.nnz);
                X.ci = tmp.alloc<int>((size_t)X.nnz);
                X.va = tmp.alloc<double>((size_t)X.nnz);
                hipLaunchKernelGGL(k_w_split_fill, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp,
                                   A.ci, A.va, diag, strong, isC, isF, cidx, (const int*)W1.rp, W1.ci, W1.va,
                                   (const int*)X.rp, X.ci, X.va, (const int*)cnt1, (const int*)cntx, W1.rp, X.rp);
                IPD_KERNEL_CHECK();
                W1.nnz = X.nnz = std::max(A.nnz / 2, 1);   // (estimates for the product's kernel choice)
                csr_spgemm(ctx, tmp, X, W1, &W2, tmp.alloc<int>(1));
            } else {
                int* cnt1 = zeroed<int>(ctx, (size_t)N + 1);
                int* cntx = zeroed<int>(ctx, (size_t)N + 1);
                {   // counts, both scans and both totals: one launch, one round trip
                    TailTotal wt(ctx, cnt1, W1.rp, N);
                    wt.t.in2 = cntx;
                    wt.t.out2 = X.rp;
                    hipLaunchKernelGGL(k_w_split_count, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp,
                                       A.ci, strong, isC, isF, cnt1, cntx, wt.t);
                    IPD_KERNEL_CHECK();
                    int t[2] = {0, 0};
                    wt.wait(t);
                    W1.nnz = t[0];
                    X.nnz = t[1];
                }
                W1.ci = tmp.alloc<int>((size_t)std::max(W1.nnz, 1));
                W1.va = tmp.alloc<double>((size_t)std::max(W1.nnz, 1));
                X.ci = tmp.alloc<int>((size_t)std::max(X.nnz, 1));
                X.va = tmp.alloc<double>((size_t)std::max(X.nnz, 1));
                hipLaunchKernelGGL(k_w_split_fill, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, A.rp,
                                   A.ci, A.va, diag, strong, isC, isF, cidx, (const int*)W1.rp, W1.ci, W1.va,
                                   (const int*)X.rp, X.ci, X.va, (const int*)nullptr, (const int*)nullptr,
                                   (int*)nullptr, (int*)nullptr);
                IPD_KERNEL_CHECK();
                csr_spgemm(ctx, tmp, X, W1, &W2);
            }
            csr_expand_dense(ctx, W1, dense, Nc);
            hipLaunchKernelGGL(k_w_combine, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, Nc, isC,
                               cidx, W2.rp, W2.ci, W2.va, dense);
            IPD_KERNEL_CHECK();
            dense_rowcount(ctx, N, Nc, Nc, dense, rowcnt, pt);
        } else {
            const char* bwe = getenv("IPD_INTERP");
            if (bwe && !strcmp(bwe, "block")) {   // (the barrier-per-neighbour form, kept for the bit-for-bit tests)
                const bool wide = (double)A.nnz / std::max(N, 1) >= 96.0;
                IPD_OPTIN_LDS(ctx, k_build_W, 128 * 1024);
                hipLaunchKernelGGL(k_build_W, dim3(std::min(N, 16384)), dim3(wide ? 256 : 64),
                                   (size_t)Nc * 16, ctx->stream, N, Nc, A.rp, A.ci, A.va, diag, strong,
                                   isC, isF, cidx, dense, rowcnt);
            } else {
                IPD_OPTIN_LDS(ctx, k_build_W_w<8>, 128 * 1024);
                hipLaunchKernelGGL(k_build_W_w<8>, dim3(std::min(N, 16384)), dim3(64),
                                   (size_t)Nc * 16, ctx->stream, N, Nc, A.rp, A.ci, A.va, diag, strong,
                                   isC, isF, cidx, dense, rowcnt);
            }
            IPD_KERNEL_CHECK();
        }
        const int* head = nullptr;   // plain counts the compaction scans itself
        if (head_ok) {   // dense bound, no round trip: the count is fetched with the products' below
            P.nnz = (int)((size_t)N * (size_t)Nc);
            head = rowcnt;
        } else if (lazy) {
            P.nnz = (int)((size_t)N * (size_t)Nc);
            if (!wtail) exclusive_scan_i32(ctx, rowcnt, P.rp, N, counts);
        } else if (wtail) {
            int two[2] = {0, 0};
            ptt->wait(two);
            P.nnz = two[0];
        } else {
            P.nnz = exclusive_scan_total(ctx, rowcnt, P.rp, N);
        }
        P.ci = dst.alloc<int>((size_t)std::max(P.nnz, 1));
        P.va = dst.alloc<double>((size_t)std::max(P.nnz, 1));
        // compaction, and D = diag(W*1); W = D\W on the F rows (transfer.m:60-62) in the same launch
        hipLaunchKernelGGL(k_dense_compact2, dim3(rows_grid(N)), dim3(256), 0, ctx->stream, N, Nc,
                           dense, (const int*)P.rp, P.ci, P.va, head, P.rp, head ? counts : (int*)nullptr,
                           o.isnsp == 1 ? (const uint8_t*)isF : (const uint8_t*)nullptr);
        IPD_KERNEL_CHECK();
    }
    // Ac = Pro'*A*Pro, evaluated left to right                               transfer.m:66
    Csr Pt, T1, C;
    csr_transpose(ctx, dst, P, &Pt);
    const int Ncc = P.nc;
    // Lazy counts (round 4): where the dense bounds are small the entry counts of P'A and Ac (and of P, above)
    // stay on the device until all three are fetched in ONE round trip at the end -- four host round trips per
    // level became two.  The products' kernel choice meanwhile runs on the previous hierarchy's counts of the
    // same level (ipd_ctx::xfer_hint; both kernels give the same bits, tests/test_gpu_product.py).
    int* const hint = (level >= 1 && level < 40) ? ctx->xfer_hint[level] : nullptr;
    const bool lazy_prod = hint && hint[1] > 0 && hint[2] > 0 && (lazy || P.nnz >= 0) &&
                           (size_t)Ncc * (size_t)N <= SPGEMM_LAZY_MAX && (size_t)Ncc * (size_t)Ncc <= SPGEMM_LAZY_MAX;
    if (lazy_prod) {
        int* c3 = lazy ? counts : zeroed<int>(ctx, 6);
        Csr Pe = P, Pte = Pt;
        if (lazy) Pe.nnz = Pte.nnz = std::max(1, std::min(hint[0], P.nnz));   // (estimates for the heuristic only)
        csr_spgemm(ctx, T1out ? dst : tmp, Pte, A, &T1, c3 + 1, nullptr, c3 + 4);
        Csr T1e = T1;
        T1e.nnz = std::max(1, std::min(hint[1], T1.nnz));
        LazyPost post;   // the level's counts ride back on the last compaction
        post.src = c3;
        post.n = 5;
        csr_spgemm(ctx, dst, T1e, Pe, &C, c3 + 2, &post, nullptr, hint[3]);
        int h3[5] = {0, 0, 0, 0, 0};
        if (post.box)
            ctx->mailbox_wait(post.ticket, h3, sizeof(h3));
        else
            ctx->fetch(c3, h3, 5);
        hint[3] = h3[4];
        if (lazy) {
            P.nnz = Pt.nnz = h3[0];
            IPD_REQUIRE(h3[3] == 0, IPD_E_UNSUPPORTED,
                        "transfer: bigph level 1 needs a diagonal Aff block (transfer.m:20-21)");
        }
        T1.nnz = h3[1];
        C.nnz = h3[2];
    } else {
        IPD_REQUIRE(!lazy, IPD_E_NUMERIC, "transfer: lazy interpolation count without lazy products");
        csr_spgemm(ctx, T1out ? dst : tmp, Pt, A, &T1);
        csr_spgemm(ctx, dst, T1, P, &C);
    }
    if (hint) {
        hint[0] = P.nnz;
        hint[1] = T1.nnz;
        hint[2] = C.nnz;
    }
    if (T1out) *T1out = T1;
    *Ac = C;
    *Pout = P;
    *Ptout = Pt;
}

// ---------------------------------------------------------------------------
// hierarchy                                           (AMG/Class_AMG.m:20-85)
// ---------------------------------------------------------------------------
AmgOpts amg_fill_defaults(const ipd_amg_opts* in) {
    AmgOpts o;  // Class_AMG.m:26-34 empty-field defaults
    if (!in) return o;
    if (in->retol >= 0) o.retol = in->retol;
    if (in->bigph >= 0) o.bigph = in->bigph;
    if (in->maxit >= 0) o.maxit = in->maxit;
    if (in->theta >= 0) o.theta = in->theta;
    if (in->smoth >= 0) o.smoth = in->smoth;
    if (in->cycle >= 0) o.cycle = in->cycle;
    if (in->isnsp >= 0) o.isnsp = in->isnsp;
    if (in->inter >= 0) o.inter = in->inter;
    o.fnode = in->fnode;
    return o;
}

// twogrid_bigph.m:6-15: nargin/empty-field defaults differ from Class_AMG's
AmgOpts amg_fill_twogrid_defaults(const ipd_amg_opts* in) {
    AmgOpts o;
    o.retol = 0.0;
    o.maxit = 50;
    o.smoth = 3;
    o.isnsp = 0;
    if (in) {
        if (in->retol >= 0) o.retol = in->retol;
        if (in->maxit >= 0) o.maxit = in->maxit;
        if (in->smoth >= 0) o.smoth = in->smoth;
        if (in->isnsp >= 0) o.isnsp = in->isnsp;
        o.fnode = in->fnode;
    }
    o.bigph = 1;
    o.cycle = 'v';       // two levels: a V cycle is twogrid_it (:55-84)
    o.twogrid = true;
    o.pcg_maxit = 100;   // :72
    return o;
}

// 1 + fix(size(A,1)^(1/3)) in floating point (Class_AMG.m:76)
int amg_coarsest_threshold(int N) { return 1 + (int)std::floor(std::pow((double)N, 1.0 / 3.0)); }

// `donor`: a hierarchy of the SAME matrix and options set up earlier (AMG4POT's first right-hand
// side, Class2/AMG4POT.m:46-47).  The reference sets up twice; the two hierarchies are equal up
// to and including A_2 -- only mis_set (levels >= 2) consumes random numbers -- so that part is
// shared and the rest is built with the stream's next numbers exactly as a full setup would:
// same bits, same rand consumption, without the most expensive product of the setup.
ipd_amg* amg_setup(ipd_ctx* ctx, const Csr& A, const AmgOpts& o, ipd_rng* rng,
                   const std::shared_ptr<ipd_amg>& donor_in) {
    // a donor that itself took its levels 1-2 from another hierarchy: share with that root, so
    // that a chain of Newton steps with the same system keeps two hierarchies alive, not all
    const std::shared_ptr<ipd_amg> donor = donor_in && donor_in->donor ? donor_in->donor : donor_in;
    IPD_REQUIRE(A.nr == A.nc && A.nr > 0, IPD_E_ARG, "Class_AMG: A must be square and non-empty");
    if (o.bigph)  // Class_AMG.m:36-40
        IPD_REQUIRE(o.fnode > 0, IPD_E_ARG, "amg_options.bigph = 1 requires Nf > 0");
    IPD_REQUIRE(o.smoth >= 0 && o.maxit >= 0, IPD_E_ARG, "negative smoth/maxit");
    ctx->zreset();   // one memset for all the zero-initialised temporaries of the previous build
    std::unique_ptr<ipd_amg> h(new ipd_amg());
    h->ctx = ctx;
    h->arena.reset(new Arena(&ctx->pool));
    h->opts = o;
    h->L.resize(2);
    h->J = 1;
    const int thr = amg_coarsest_threshold(A.nr);
    // Levels 1-2 of a bigraph hierarchy depend on A and on bigph / fnode / isnsp / inter only
    // (transfer.m:19-25 draws no random numbers and uses no theta; Rk{1}, Rk{2} use none of the
    // options): exactly what is compared here, so a donor built under another theta, smoth, cycle,
    // retol or maxit is still the same levels 1-2.
    const bool share = donor && o.bigph && !o.twogrid && donor->J >= 2 && donor->opts.bigph &&
                       !donor->opts.twogrid && donor->opts.fnode == o.fnode &&
                       donor->opts.isnsp == o.isnsp && donor->opts.inter == o.inter &&
                       donor->L[1].A.nr == A.nr && donor->L[1].A.nnz == A.nnz && A.nr > thr &&
                       donor->ctx->device == ctx->device;
    if (share) {
        h->donor = donor;
        h->L[1].A = donor->L[1].A;
        h->L[1].N = A.nr;
        Level nl;
        nl.A = donor->L[2].A;
        nl.P = donor->L[2].P;
        nl.Pt = donor->L[2].Pt;
        nl.T1 = donor->L[2].T1;
        nl.cmask = donor->L[2].cmask;
        nl.N = nl.A.nr;
        h->L.push_back(nl);
        h->J = 2;
    } else {
        csr_copy(ctx, *h->arena, A, &h->L[1].A);
        h->L[1].N = A.nr;
    }
    auto more = [&] {   // Class_AMG.m:76; twogrid_bigph.m builds exactly one coarse level
        return o.twogrid ? h->J < 2 : h->L[h->J].A.nr > thr;
    };
    if (o.twogrid) IPD_REQUIRE(A.nr >= 2, IPD_E_ARG, "twogrid needs at least 2 nodes");
    while (more()) {
        IPD_REQUIRE(h->J < 40, IPD_E_NUMERIC, "Class_AMG: coarsening stalled (40 levels)");
        const Csr& Ak = h->L[h->J].A;
        Level nl;
        nl.cmask = h->arena->alloc<uint8_t>((size_t)Ak.nr);
        amg_transfer(ctx, *h->arena, Ak, o, h->J, rng, &nl.A, &nl.P, &nl.Pt, nl.cmask, &nl.T1);
        nl.N = nl.A.nr;
        IPD_REQUIRE(nl.N < Ak.nr, IPD_E_NUMERIC,
                    "Class_AMG: coarsening made no progress (the reference would loop forever)");
        h->L.push_back(nl);
        h->J += 1;
    }
    amg_prepare_levels(h.get());
    return h.release();
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int ipd_strength(ipd_ctx* ctx, const ipd_csc* A, int which, ipd_csc_out* S) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && S, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(which == 1 || which == 2, IPD_E_ARG, "strength: which must be 1 or 2");
        CallScope scope(ctx);
        Arena& tmp = *ctx->scratch;
        Csr a;
        csr_upload_from_csc(ctx, tmp, A, false, &a);
        double* maxrow = tmp.alloc<double>((size_t)a.nr);
        double* diag = tmp.alloc<double>((size_t)a.nr);
        Csr v = a;
        v.va = tmp.alloc<double>((size_t)a.nnz);
        hipLaunchKernelGGL(k_rowmax, dim3(rows_grid(a.nr)), dim3(256), 0, ctx->stream, a.nr, a.rp,
                           a.ci, a.va, maxrow, diag);
        hipLaunchKernelGGL(k_strength_values, dim3(rows_grid(a.nr)), dim3(256), 0, ctx->stream, a.nr,
                           a.rp, a.ci, a.va, maxrow, which, v.va);
        IPD_KERNEL_CHECK();
        Csr clean;
        csr_drop_zeros(ctx, tmp, v, &clean);
        csr_download_as_csc(ctx, clean, false, S);
    });
}

extern "C" int ipd_cf_split(ipd_ctx* ctx, const ipd_csc* S, uint8_t* indC, uint8_t* indF) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && S && indC && indF, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(S->nrows == S->ncols, IPD_E_ARG, "cf_split: S must be square");
        CallScope scope(ctx);
        Arena& tmp = *ctx->scratch;
        Csr s;
        csr_upload_from_csc(ctx, tmp, S, true, &s);  // graph(S) requires a symmetric S
        uint8_t* dC = tmp.alloc<uint8_t>((size_t)s.nr);
        uint8_t* dF = tmp.alloc<uint8_t>((size_t)s.nr);
        amg_cf_split(ctx, s, dC, dF);
        ctx->fetch(dC, indC, (size_t)s.nr);
        ctx->fetch(dF, indF, (size_t)s.nr);
    });
}

// strong flags (aligned with A's pattern) -> CSR pattern matrix with values 1
__global__ void k_flag_to_value(int nnz, const uint8_t* __restrict__ f, double* __restrict__ v) {
    THREAD_ELEMS(i, nnz) v[i] = f[i] ? 1.0 : 0.0;
}

extern "C" int ipd_mis_set(ipd_ctx* ctx, const ipd_csc* A, double theta, ipd_rng* rng,
                           uint8_t* isC, uint8_t* isF, ipd_csc_out* As) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && rng && isC && isF, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(A->nrows == A->ncols, IPD_E_ARG, "mis_set: A must be square");
        CallScope scope(ctx);
        Arena& tmp = *ctx->scratch;
        Csr a;
        csr_upload_from_csc(ctx, tmp, A, false, &a);
        uint8_t* dC = tmp.alloc<uint8_t>((size_t)a.nr);
        uint8_t* dF = tmp.alloc<uint8_t>((size_t)a.nr);
        uint8_t* strong = tmp.alloc<uint8_t>((size_t)std::max(a.nnz, 1));
        amg_mis_set(ctx, a, theta, rng, dC, dF, strong);
        ctx->fetch(dC, isC, (size_t)a.nr);
        ctx->fetch(dF, isF, (size_t)a.nr);
        if (As) {
            Csr v = a;
            v.va = tmp.alloc<double>((size_t)std::max(a.nnz, 1));
            hipLaunchKernelGGL(k_flag_to_value, dim3(elems_grid(a.nnz)), dim3(256), 0, ctx->stream,
                               a.nnz, strong, v.va);
            IPD_KERNEL_CHECK();
            Csr clean;
            csr_drop_zeros(ctx, tmp, v, &clean);
            csr_download_as_csc(ctx, clean, false, As);
        }
    });
}

extern "C" int ipd_transfer(ipd_ctx* ctx, const ipd_csc* A, const ipd_amg_opts* o, int level,
                            ipd_rng* rng, ipd_csc_out* Ac, ipd_csc_out* Pro, uint8_t* indC) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && Ac && Pro, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        AmgOpts opts = amg_fill_defaults(o);
        Arena out(&ctx->pool);
        Csr a, c, p, pt;
        // true rows of A (= transpose of the CSC arrays): a product like Q0*H0*Q0 is symmetric
        // only up to the last bit, and the setup must see exactly MATLAB's rows
        csr_upload_from_csc(ctx, out, A, false, &a);
        uint8_t* cmask = out.alloc<uint8_t>((size_t)a.nr);
        amg_transfer(ctx, out, a, opts, level, rng, &c, &p, &pt, cmask);
        csr_download_as_csc(ctx, c, false, Ac);
        csr_download_as_csc(ctx, pt, true, Pro);  // CSR of Pro' == CSC of Pro
        if (indC) ctx->fetch(cmask, indC, (size_t)a.nr);
        ctx->sync();
    });
}

extern "C" int ipd_amg_setup_dev(ipd_ctx* ctx, const ipd_dmat* A, const ipd_amg_opts* o,
                                 ipd_rng* rng, ipd_amg** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && out, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        *out = amg_setup(ctx, A->m, amg_fill_defaults(o), rng);
    });
}

extern "C" int ipd_amg_setup(ipd_ctx* ctx, const ipd_csc* A, const ipd_amg_opts* o, ipd_rng* rng,
                             ipd_amg** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && out, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        Arena up(&ctx->pool);
        Csr a;
        csr_upload_from_csc(ctx, up, A, false, &a);  // true rows (see ipd_transfer)
        *out = amg_setup(ctx, a, amg_fill_defaults(o), rng);
        ctx->sync();
    });
}

static int twogrid_host(ipd_ctx* ctx, const ipd_csc* A, const double* b, const double* guess,
                        const AmgOpts& ao, ipd_rng* rng, double* x, int32_t* it, double* rel_res,
                        double* rel_resk, double* rhok) {
    ipd_amg* h = nullptr;
    int rc = ipd_guard([&] {
        IPD_REQUIRE(ctx && A && b && x, IPD_E_ARG, "NULL argument");
        CallScope scope(ctx);
        Arena up(&ctx->pool);
        Csr a;
        csr_upload_from_csc(ctx, up, A, false, &a);
        h = amg_setup(ctx, a, ao, rng);
        ctx->sync();
    });
    if (rc != IPD_OK) return rc;
    rc = ipd_amg_solve(h, b, guess, x, it, rel_res, rel_resk, rhok);
    ipd_amg_destroy(h);
    return rc;
}
// [x,it,rel_res,rel_resk,rhok] = twogrid_bigph(A,b,amg_options)      AMG/twogrid_bigph.m:1
extern "C" int ipd_twogrid_bigph(ipd_ctx* ctx, const ipd_csc* A, const double* b,
                                 const double* guess, const ipd_amg_opts* o, double* x, int32_t* it,
                                 double* rel_res, double* rel_resk, double* rhok) {
    return twogrid_host(ctx, A, b, guess, amg_fill_twogrid_defaults(o), nullptr, x, it, rel_res,
                        rel_resk, rhok);
}
// [x,it,rel_res,rel_resk,rhok] = twogrid(A,b,amg_options)                  AMG/twogrid.m:1
extern "C" int ipd_twogrid(ipd_ctx* ctx, const ipd_csc* A, const double* b, const double* guess,
                           const ipd_amg_opts* o, ipd_rng* rng, double* x, int32_t* it,
                           double* rel_res, double* rel_resk, double* rhok) {
    AmgOpts ao = amg_fill_twogrid_defaults(o);
    ao.bigph = (o && o->bigph >= 0) ? o->bigph : 0;                       // twogrid.m:12
    ao.theta = 0.25;                                                      // mis_set(A,1/4), :50
    ao.inter = 1;
    if (ao.bigph && ao.fnode <= 0) {                                      // :24-26
        ipd_set_error("bigph = 1 requires fnode > 0");
        return IPD_E_ARG;
    }
    if (!ao.bigph && !rng) {
        ipd_set_error("twogrid: mis_set needs a rand stream");
        return IPD_E_ARG;
    }
    return twogrid_host(ctx, A, b, guess, ao, rng, x, it, rel_res, rel_resk, rhok);
}

extern "C" void ipd_amg_destroy(ipd_amg* h) {
    if (!h) return;
    if (h->ctx) {
        (void)hipSetDevice(h->ctx->device);
        (void)hipStreamSynchronize(h->ctx->stream);
    }
    delete h;
}

extern "C" int ipd_amg_num_levels(const ipd_amg* h) { return h ? h->J : IPD_E_ARG; }

extern "C" int ipd_amg_level_dims(const ipd_amg* h, int k, int64_t* rows, int64_t* nnz) {
    if (!h || k < 1 || k > h->J) return IPD_E_ARG;
    if (rows) *rows = h->L[k].A.nr;
    if (nnz) *nnz = h->L[k].A.nnz;
    return IPD_OK;
}

extern "C" int ipd_amg_get_A(const ipd_amg* h, int k, ipd_csc_out* A) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && A && k >= 1 && k <= h->J, IPD_E_ARG, "bad level");
        CallScope scope(h->ctx);
        csr_download_as_csc(h->ctx, h->L[k].A, false, A);
    });
}

extern "C" int ipd_amg_get_P(const ipd_amg* h, int k, ipd_csc_out* P) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && P && k >= 2 && k <= h->J, IPD_E_ARG, "bad level (Prok{k} exists for k>=2)");
        CallScope scope(h->ctx);
        csr_download_as_csc(h->ctx, h->L[k].Pt, true, P);
    });
}

extern "C" int ipd_amg_get_cmask(const ipd_amg* h, int k, uint8_t* isC) {
    return ipd_guard([&] {
        IPD_REQUIRE(h && isC && k >= 2 && k <= h->J, IPD_E_ARG, "bad level");
        CallScope scope(h->ctx);
        h->ctx->fetch(h->L[k].cmask, isC, (size_t)h->L[k - 1].A.nr);
    });
}
