// Problem-level solvers on the device:
//   Hybrid_AMG.m:12-113 (rescaling, component routing), components.m:32-55,
//   Class2/AMG4POT.m:27-55 (Sherman-Morrison around two Hybrid_AMG solves).
//
// Ae = bk1*Q0^2 + (Q0*T*Q0 + Q0*H0*Q0)/tk is assembled entry by entry with the
// reference's operation order (no FMA), so it is bit-identical to the oracle's and
// the hierarchy built on it stays bit-exact.  Connected components are found on the
// device (concurrent union-find); the O(M) bookkeeping that turns
// labels into blocks/sizes/p/r and routes components is host logic.
#pragma clang fp contract(off)

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>

#include "ipd_amg_internal.h"

#include <algorithm>
#include <cmath>
#include <numeric>

static inline int rows_grid(int nr) { return std::max(1, std::min(cdiv(nr, 4), 4096)); }
static inline int elems_grid(long long n) {
    return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096));
}

// ---------------------------------------------------------------------------
// Ae assembly                                             (Hybrid_AMG.m:17-24)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double qp_of(const double* __restrict__ p, const double* __restrict__ q,
                                        int n, int i) {
    return i < n ? q[i] : -p[i - n];  // qp = [q; -p]
}

// row lengths of Ae: the pattern of H0 plus a full diagonal; flags a zero in p or q (Hybrid_AMG.m:18-19); the last
// workgroup scans the lengths into Ae's row pointers and posts total and flag to the host
__global__ __launch_bounds__(256) void k_ae_count(int M, int n, const int* __restrict__ rp,
                                                  const int* __restrict__ ci, const double* __restrict__ p,
                                                  const double* __restrict__ q, int* rowlen,
                                                  const ScanTail st) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < M; i += nwaves) {
        bool has = false;
        for (int t = rp[i] + lane; t < rp[i + 1]; t += 64) has |= (ci[t] == i);
        const bool hasd = __any(has);
        if (lane == 0)
            scan_put(rowlen, i, rp[i + 1] - rp[i] + (hasd ? 0 : 1), (i < n ? q[i] : p[i - n]) == 0.0);
    }
    scan_tail(st);
}

__global__ __launch_bounds__(256) void k_ae_fill(int M, int n, const int* __restrict__ rp,
                                                 const int* __restrict__ ci,
                                                 const double* __restrict__ va,
                                                 const double* __restrict__ p,
                                                 const double* __restrict__ q,
                                                 const double* __restrict__ tdiag, double bk1,
                                                 double inv_tk, const int* __restrict__ orp,
                                                 int* __restrict__ oci, double* __restrict__ ova) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < M; i += nwaves) {
        const int b = rp[i], e = rp[i + 1];
        const double qi = qp_of(p, q, n, i);
        // number of stored columns below i, and whether the diagonal is stored
        int below = 0;
        bool has = false;
        for (int t = b + lane; t < e; t += 64) {
            const int j = ci[t];
            below += (j < i);
            has |= (j == i);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) below += __shfl_xor(below, d);
        has = __any(has);
        const int ob = orp[i];
        const int shift = has ? 0 : 1;
        for (int t = b + lane; t < e; t += 64) {
            const int j = ci[t];
            const double a0 = (qi * va[t]) * qp_of(p, q, n, j);  // (Q0*H0)*Q0
            const int pos = ob + (t - b) + (j > i ? shift : 0);
            oci[pos] = j;
            if (j == i) {
                const double qq = qi * qi;                       // Q = Q0*Q0
                const double kk = tdiag ? (qi * tdiag[i]) * qi : 0.0;  // K = (Q0*T)*Q0
                const double x = kk + a0;                        // K + A0
                ova[pos] = bk1 * qq + inv_tk * x;
            } else {
                ova[pos] = inv_tk * a0;
            }
        }
        if (!has && lane == 0) {
            const double qq = qi * qi;
            const double kk = tdiag ? (qi * tdiag[i]) * qi : 0.0;
            oci[ob + below] = i;
            ova[ob + below] = bk1 * qq + inv_tk * kk;
        }
    }
}

// f = Q0*z, dK = diag(K), and zeta = Q0*u at the end
__global__ void k_scale_qp(int M, int n, const double* __restrict__ p, const double* __restrict__ q,
                           const double* __restrict__ in, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x)
        out[i] = qp_of(p, q, n, i) * in[i];
}
__global__ void k_dk(int M, int n, const double* __restrict__ p, const double* __restrict__ q,
                     const double* __restrict__ tdiag, double* __restrict__ dK) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const double qi = qp_of(p, q, n, i);
        dK[i] = tdiag ? (qi * tdiag[i]) * qi : 0.0;
    }
}
static void build_Ae(ipd_ctx* ctx, Arena& dst, const Csr& H0, const double* tdiag, const double* p,
                     const double* q, int m, int n, double bk1, double tk, Csr* Ae) {
    const int M = m + n;
    IPD_REQUIRE(H0.nr == M && H0.nc == M, IPD_E_ARG, "Hybrid_AMG: H0 must be (n+m) x (n+m)");
    int* rowlen = zeroed<int>(ctx, (size_t)M + 1);   // (biased counts: see ScanTail)
    Csr a;
    a.nr = a.nc = M;
    a.rp = dst.alloc<int>((size_t)M + 1);
    TailTotal tt(ctx, rowlen, a.rp, M);   // entry count and the "p or q has a zero" flag in one message
    hipLaunchKernelGGL(k_ae_count, dim3(rows_grid(M)), dim3(256), 0, ctx->stream, M, n, H0.rp, H0.ci, p, q,
                       rowlen, tt.t);
    IPD_KERNEL_CHECK();
    int h2[2] = {0, 0};
    tt.wait(h2);
    a.nnz = h2[0];
    IPD_REQUIRE(h2[1] == 0, IPD_E_ARG, "p or q contains 0 !!!!!");  // Hybrid_AMG.m:18-19
    a.ci = dst.alloc<int>((size_t)a.nnz);
    a.va = dst.alloc<double>((size_t)a.nnz);
    const double inv_tk = 1.0 / tk;
    hipLaunchKernelGGL(k_ae_fill, dim3(rows_grid(M)), dim3(256), 0, ctx->stream, M, n, H0.rp, H0.ci,
                       H0.va, p, q, tdiag, bk1, inv_tk, a.rp, a.ci, a.va);
    IPD_KERNEL_CHECK();
    *Ae = a;
}

// ---------------------------------------------------------------------------
// connected components                                     (components.m:32-55)
// ---------------------------------------------------------------------------
// Concurrent union-find over the whole chip (the scheme of Jaiganesh & Burtscher's ECL-CC):
// parent[x] <= x always; every edge (i,j), j < i, merges the two trees by hooking the LARGER
// root under the smaller with a compare-and-swap, retrying from the returned value when another
// wave got there first; a last pass points every node at its root.  The root of a tree is the
// smallest member of the component whatever the interleaving, so the labels are deterministic.
// Three launches and no rounds: the single-workgroup hook-and-compress loop this replaces took
// 350-400 us per Newton step at m = n = 4096.
__device__ __forceinline__ int cc_root(int x, int* __restrict__ parent) {
    int curr = parent[x];
    if (curr != x) {
        int prev = x, next;
        while (curr > (next = parent[curr])) {   // path halving: only ever points further up
            parent[prev] = next;
            prev = curr;
            curr = next;
        }
    }
    return curr;
}

__device__ __forceinline__ void cc_union(int a, int b, int* __restrict__ parent) {
    int ra = cc_root(a, parent), rb = cc_root(b, parent);
    while (ra != rb) {
        if (ra < rb) {
            const int t = ra;
            ra = rb;
            rb = t;
        }
        const int seen = atomicCAS(&parent[ra], ra, rb);   // ra > rb: hook ra under rb
        if (seen == ra) break;
        ra = seen;                                          // ra was hooked meanwhile: follow it
    }
}

__global__ __launch_bounds__(256) void k_cc_init(int N, const int* __restrict__ rp,
                                                 const int* __restrict__ ci,
                                                 int* __restrict__ parent) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int first = i;   // rows are sorted: the first entry is the smallest neighbour
        if (rp[i] < rp[i + 1]) first = min(i, ci[rp[i]]);
        parent[i] = first;
    }
}

__global__ __launch_bounds__(256) void k_cc_hook(int N, const int* __restrict__ rp,
                                                 const int* __restrict__ ci,
                                                 int* __restrict__ parent) {
    // rows of up to 32 entries: one thread each; longer rows (hubs, dense masks): one wave each
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gtid; i < N; i += gsz) {
        const int b = rp[i], e = rp[i + 1];
        if (e - b > 32) continue;
        for (int t = b; t < e; ++t) {
            const int j = ci[t];
            if (j >= i) break;
            cc_union(i, j, parent);
        }
    }
    const int lane = threadIdx.x & 63;
    for (int i = gtid >> 6; i < N; i += gsz >> 6) {
        const int b = rp[i], e = rp[i + 1];
        if (e - b <= 32) continue;
        for (int t = b + lane; t < e; t += 64) {
            const int j = ci[t];
            if (j < i) cc_union(i, j, parent);
        }
    }
}

__global__ __launch_bounds__(256) void k_cc_flatten(int N, const int* __restrict__ parent,
                                                    int* __restrict__ root_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        int r = parent[i];
        while (true) {
            const int up = parent[r];
            if (up == r) break;
            r = up;
        }
        root_out[i] = r;
    }
}

struct Components {
    int ncomp = 0;
    std::vector<int> blocks;  // component of every node (numbered by smallest member)
    std::vector<int> sizes;
    std::vector<int> p;       // nodes grouped by component, ascending inside a component
    std::vector<int> r;       // boundaries, ncomp+1
};

static void find_components(ipd_ctx* ctx, const Csr& A, Components* out) {
    IPD_REQUIRE(A.nr == A.nc, IPD_E_ARG, "Adjacency matrix must be square");  // components.m:33
    const int N = A.nr;
    int* parent = ctx->scratch->alloc<int>((size_t)N);
    int* root = ctx->scratch->alloc<int>((size_t)N);
    const int g = std::max(1, std::min((N + 255) / 256, 1024));
    hipLaunchKernelGGL(k_cc_init, dim3(g), dim3(256), 0, ctx->stream, N, A.rp, A.ci, parent);
    // a wave per long row needs 64 threads per row to keep the chip busy
    const int gh = std::max(1, std::min((N + 3) / 4, 2048));
    hipLaunchKernelGGL(k_cc_hook, dim3(gh), dim3(256), 0, ctx->stream, N, A.rp, A.ci, parent);
    hipLaunchKernelGGL(k_cc_flatten, dim3(g), dim3(256), 0, ctx->stream, N, parent, root);
    IPD_KERNEL_CHECK();
    std::vector<int> par((size_t)N);
    ctx->fetch(root, par.data(), (size_t)N);
    out->blocks.assign((size_t)N, 0);
    std::vector<int> cid((size_t)N, -1);
    int nc = 0;
    for (int i = 0; i < N; ++i)
        if (par[i] == i) cid[i] = nc++;  // roots ascend == smallest members ascend
    out->ncomp = nc;
    out->sizes.assign((size_t)nc, 0);
    for (int i = 0; i < N; ++i) {
        IPD_REQUIRE(par[i] >= 0 && par[i] <= i && cid[par[i]] >= 0, IPD_E_NUMERIC,
                    "components: a node does not point at the smallest member of its tree");
        out->blocks[i] = cid[par[i]];
        out->sizes[(size_t)out->blocks[i]]++;
    }
    out->r.assign((size_t)nc + 1, 0);
    for (int c = 0; c < nc; ++c) out->r[c + 1] = out->r[c] + out->sizes[c];
    out->p.assign((size_t)N, 0);
    std::vector<int> cur(out->r.begin(), out->r.end() - 1);
    for (int i = 0; i < N; ++i) out->p[(size_t)cur[(size_t)out->blocks[i]]++] = i;
    // An injected visiting order (SURVEY A-9: MATLAB's dmperm order is undocumented, so a recorded
    // one can be replayed): component k of the result is the one whose smallest member is
    // comp_order[k].  Members stay ascending inside a component (F side first).  The order only
    // moves info(2), the loop order and with it the order in which rand is consumed.
    if (!ctx->comp_order.empty()) {
        const std::vector<int>& ord = ctx->comp_order;
        IPD_REQUIRE((int)ord.size() == nc, IPD_E_ARG,
                    "component order: as many entries as there are components expected");
        std::vector<int> newid((size_t)nc, -1);
        for (int k = 0; k < nc; ++k) {
            const int sm = ord[(size_t)k];
            IPD_REQUIRE(sm >= 0 && sm < N && par[(size_t)sm] == sm && newid[(size_t)cid[(size_t)sm]] < 0,
                        IPD_E_ARG, "component order: entries must be the smallest members of distinct components");
            newid[(size_t)cid[(size_t)sm]] = k;
        }
        Components re;
        re.ncomp = nc;
        re.blocks.resize((size_t)N);
        re.sizes.assign((size_t)nc, 0);
        for (int c = 0; c < nc; ++c) re.sizes[(size_t)newid[(size_t)c]] = out->sizes[(size_t)c];
        re.r.assign((size_t)nc + 1, 0);
        for (int c = 0; c < nc; ++c) re.r[(size_t)c + 1] = re.r[(size_t)c] + re.sizes[(size_t)c];
        re.p.assign((size_t)N, 0);
        std::vector<int> cur2(re.r.begin(), re.r.end() - 1);
        for (int i = 0; i < N; ++i) {
            const int c = newid[(size_t)out->blocks[(size_t)i]];
            re.blocks[(size_t)i] = c;
            re.p[(size_t)cur2[(size_t)c]++] = i;
        }
        *out = re;
    }
}

// One-shot: the order applies to the NEXT top-level call that visits components on this context
// (components, Hybrid_AMG, AMG4POT and their two-grid / PCG variants) and is cleared when that
// call returns.  order == NULL or ncomp == 0 clears it.
extern "C" int ipd_ctx_set_component_order(ipd_ctx* ctx, const int64_t* smallest_members, int64_t ncomp) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && ncomp >= 0 && ncomp < (int64_t(1) << 30), IPD_E_ARG, "bad argument");
        ctx->comp_order.clear();
        if (smallest_members)
            for (int64_t k = 0; k < ncomp; ++k) ctx->comp_order.push_back((int)smallest_members[k]);
    });
}
struct CompOrderScope {   // clears the injected order when the top-level call ends
    ipd_ctx* ctx;
    explicit CompOrderScope(ipd_ctx* c) : ctx(c) {}
    ~CompOrderScope() {
        if (ctx) ctx->comp_order.clear();
    }
};

// ---------------------------------------------------------------------------
// sub-matrix of one component: Aek = Ae(pk,pk), pk ascending
// ---------------------------------------------------------------------------
__global__ void k_sub_count(int nk, const int* __restrict__ pk, const int* __restrict__ rp,
                            int* __restrict__ rowlen) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nk; r += gridDim.x * blockDim.x) {
        const int i = pk[r];
        rowlen[r] = rp[i + 1] - rp[i];
    }
}
__global__ __launch_bounds__(256) void k_sub_fill(int nk, const int* __restrict__ pk,
                                                  const int* __restrict__ newidx,
                                                  const int* __restrict__ rp,
                                                  const int* __restrict__ ci,
                                                  const double* __restrict__ va,
                                                  const int* __restrict__ orp,
                                                  int* __restrict__ oci, double* __restrict__ ova) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < nk; r += nwaves) {
        const int i = pk[r];
        const int b = rp[i], ob = orp[r];
        for (int t = b + lane; t < rp[i + 1]; t += 64) {
            oci[ob + (t - b)] = newidx[ci[t]];  // a component is closed: every column is inside
            ova[ob + (t - b)] = va[t];
        }
    }
}
__global__ void k_gather(int nk, const int* __restrict__ pk, const double* __restrict__ src,
                         double* __restrict__ dst) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nk; r += gridDim.x * blockDim.x)
        dst[r] = src[pk[r]];
}
__global__ void k_scatter(int nk, const int* __restrict__ pk, const double* __restrict__ src,
                          double* __restrict__ dst) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nk; r += gridDim.x * blockDim.x)
        dst[pk[r]] = src[r];
}
__global__ void k_scale(int n, double alpha, const double* __restrict__ in,
                        double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = alpha * in[i];
}

// ---------------------------------------------------------------------------
// small components: u(pk) = Ae(pk,pk) \ f(pk)              (Hybrid_AMG.m:85-91)
// ---------------------------------------------------------------------------
// The system is block diagonal with SPD blocks of at most N0 = 100 rows: one
// workgroup per block factors its dense copy in LDS (Cholesky) and solves.
__global__ __launch_bounds__(256) void k_small_blocks(const int* __restrict__ boff,
                                                      const int* __restrict__ nodes,
                                                      const int* __restrict__ local,
                                                      const int* __restrict__ rp,
                                                      const int* __restrict__ ci,
                                                      const double* __restrict__ va,
                                                      const double* __restrict__ f,
                                                      double* __restrict__ u,
                                                      int* __restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int blk = blockIdx.x;
    const int o = boff[blk], nb = boff[blk + 1] - o;
    const int ld = nb + 1;
    double* A = reinterpret_cast<double*>(smem_raw);  // nb x ld, row-major
    double* y = A + (size_t)nb * ld;
    const int tid = threadIdx.x;
    for (int k = tid; k < nb * ld; k += 256) A[k] = 0.0;
    __syncthreads();
    for (int li = tid; li < nb; li += 256) {
        const int i = nodes[o + li];
        for (int t = rp[i]; t < rp[i + 1]; ++t) A[li * ld + local[ci[t]]] = va[t];
        y[li] = f[i];
    }
    __syncthreads();
    // right-looking Cholesky, lower triangle
    for (int k = 0; k < nb; ++k) {
        const double akk = A[k * ld + k];
        if (!(akk > 0.0)) {
            if (tid == 0) *bad = 1;
            return;  // uniform: every thread reads the same akk
        }
        const double lkk = sqrt(akk);
        __syncthreads();
        for (int i = k + tid; i < nb; i += 256) A[i * ld + k] = (i == k) ? lkk : A[i * ld + k] / lkk;
        __syncthreads();
        const int rem = nb - k - 1;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int i = k + 1 + idx / rem, j = k + 1 + idx % rem;
            if (j <= i) A[i * ld + j] -= A[i * ld + k] * A[j * ld + k];
        }
        __syncthreads();
    }
    // forward and backward substitution by thread 0 of each wave-0 lane set (nb <= 100)
    if (tid == 0) {
        for (int i = 0; i < nb; ++i) {
            double s = y[i];
            for (int j = 0; j < i; ++j) s -= A[i * ld + j] * y[j];
            y[i] = s / A[i * ld + i];
        }
        for (int i = nb - 1; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < nb; ++j) s -= A[j * ld + i] * y[j];
            y[i] = s / A[i * ld + i];
        }
    }
    __syncthreads();
    for (int li = tid; li < nb; li += 256) u[nodes[o + li]] = y[li];
}

// ---------------------------------------------------------------------------
// Hybrid_AMG
// ---------------------------------------------------------------------------
static double host_sum(ipd_ctx* ctx, const double* d, int n) {
    std::vector<double> h((size_t)n);
    ctx->fetch(d, h.data(), (size_t)n);
    double s = 0.0;
    for (double v : h) s += v;
    return s;
}

// Solves Ae u = f for one (sub)system with Class_AMG and returns its statistics.
struct MaskHint {   // p, q, tk of the rescaled system when A is the whole Ae (else p == NULL)
    const double* p = nullptr;
    const double* q = nullptr;
    int m = 0, n = 0;
    double tk = 0.0;
};

// Row f3: AMG4POT solves two systems with the same Ae (Class2/AMG4POT.m:46-47) and the reference
// sets the hierarchy up twice.  The second setup draws fresh random numbers in mis_set, so only the
// rand-independent part (Ae, the components, levels 1-2) is shared: bit-identical to two full setups.
// (One hierarchy for both right-hand sides was measured in round 2 -- it changes zeta by 6.5e-9 and is
// slower, because the two solves no longer overlap -- and removed in round 4.)
struct HybridCache {
    bool valid = false;
    Csr Ae;
    Components cc;
    // donors (default path): the first call records its hierarchies, the second call's setups
    // share their rand-independent part (amg_setup's `donor`): bit-identical results
    bool record_donors = false, use_donors = false;
    std::vector<std::shared_ptr<ipd_amg>> donors;     // offered to the setups of this call
    std::vector<std::shared_ptr<ipd_amg>> recorded;   // set up by this call
    size_t next_donor = 0;
    long long* shared_count = nullptr;                // setups that took a donor's levels
};

// Work postponed to the solve phase of a Hybrid_AMG call: AMG4POT's two calls set their
// hierarchies up one after the other (that keeps the order in which the reference consumes
// rand) and then run their solve phases concurrently on two streams.
using Deferred = std::vector<std::function<void()>>;

// Guess + setup now (consumes rand); returns the solve phase, which fills *it / *rel_res.
static std::function<void(int*, double*)> class_amg_prepare(
    ipd_ctx* ctx, const Csr& A, const double* f, AmgOpts o, int isnsp, long long fnode, double gscale,
    ipd_rng* rng, double* u_out, const MaskHint& mh = MaskHint(), HybridCache* cache = nullptr) {
    const int N = A.nr;
    o.isnsp = isnsp;
    o.fnode = fnode;
    std::vector<double> g((size_t)N);
    rng->fill(g.data(), N);                       // Hybrid_AMG.m:40 / :69  bk1*tk*rand(size(f))
    for (double& v : g) v = gscale * v;
    double* dg = ctx->scratch->alloc<double>((size_t)N);
    ctx->upload(dg, g.data(), (size_t)N);
    std::shared_ptr<ipd_amg> own;
    ipd_amg* h = nullptr;
    if (cache && (cache->record_donors || cache->use_donors)) {
        ProfScope ps(ctx, PROF_AMG_SETUP);
        std::shared_ptr<ipd_amg> donor;
        if (cache->use_donors && cache->next_donor < cache->donors.size())
            donor = cache->donors[cache->next_donor++];
        own = std::shared_ptr<ipd_amg>(amg_setup(ctx, A, o, rng, donor), ipd_amg_destroy);
        h = own.get();
        if (mh.p && o.bigph) amg_attach_maskop(h, mh.p, mh.q, mh.m, mh.n, mh.tk, true);
        if (cache->record_donors) cache->recorded.push_back(own);
        if (cache->shared_count && own->donor) ++*cache->shared_count;
    } else {
        ProfScope ps(ctx, PROF_AMG_SETUP);
        std::unique_ptr<ipd_amg, void (*)(ipd_amg*)> fresh(amg_setup(ctx, A, o, rng), ipd_amg_destroy);
        h = fresh.get();
        // matrix-free level 1 where it pays (policy in amg_attach_maskop)
        if (mh.p && o.bigph) amg_attach_maskop(h, mh.p, mh.q, mh.m, mh.n, mh.tk, true);
        own = std::shared_ptr<ipd_amg>(fresh.release(), ipd_amg_destroy);
    }
    return [ctx, h, own, f, dg, u_out](int* it, double* rel_res) {
        int32_t its = 0;
        double rr = 0.0;
        {
            ProfScope ps(ctx, PROF_AMG_SOLVE);
            amg_solve_dev(h, f, dg, u_out, &its, &rr, nullptr, nullptr);
        }
        *it = its;
        *rel_res = rr;
    };
}

// Debugging aid: IPD_DUMP_SYSTEM=<prefix> IPD_DUMP_CALLS=<lo>-<hi> writes the rescaled Newton
// systems Ae*u = f of those Hybrid_AMG calls (counted per process from 0) to
// <prefix><call>.bin: int64 M, nf, nnz; int32 rp[M+1], ci[nnz]; double va[nnz], f[M].
// tests/read_system_dump.py reads them back, so a system a driver run produced can be put
// through the oracle.
static void dump_system(ipd_ctx* ctx, const Csr& Ae, const double* f, int nf) {
    static std::atomic<int> calls{0};
    const int call = calls++;
    const char* prefix = getenv("IPD_DUMP_SYSTEM");
    if (!prefix) return;
    int lo = 0, hi = 0;
    if (const char* e = getenv("IPD_DUMP_CALLS")) {
        if (sscanf(e, "%d-%d", &lo, &hi) < 2) hi = lo;
    }
    if (call < lo || call > hi) return;
    std::vector<int> rp((size_t)Ae.nr + 1), ci((size_t)Ae.nnz);
    std::vector<double> va((size_t)Ae.nnz), hf((size_t)Ae.nr);
    ctx->fetch(Ae.rp, rp.data(), rp.size());
    if (Ae.nnz) {
        ctx->fetch(Ae.ci, ci.data(), ci.size());
        ctx->fetch(Ae.va, va.data(), va.size());
    }
    ctx->fetch(f, hf.data(), hf.size());
    const std::string path = std::string(prefix) + std::to_string(call) + ".bin";
    FILE* fp = fopen(path.c_str(), "wb");
    IPD_REQUIRE(fp, IPD_E_ARG, "IPD_DUMP_SYSTEM: cannot open the output file");
    const int64_t head[3] = {Ae.nr, nf, Ae.nnz};
    fwrite(head, sizeof(int64_t), 3, fp);
    fwrite(rp.data(), sizeof(int), rp.size(), fp);
    fwrite(ci.data(), sizeof(int), ci.size(), fp);
    fwrite(va.data(), sizeof(double), va.size(), fp);
    fwrite(hf.data(), sizeof(double), hf.size(), fp);
    fclose(fp);
}

static void hybrid_amg_cached(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                              const double* q, int m, int n, double bk1, double tk, const double* z,
                              const AmgOpts& opts, ipd_rng* rng, double* zeta, HybridOut* out,
                              HybridCache* cache, Deferred* later = nullptr);

void hybrid_amg_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                    const double* q, int m, int n, double bk1, double tk, const double* z,
                    const AmgOpts& opts, ipd_rng* rng, double* zeta, HybridOut* out,
                    StepDonors* step) {
    if (!step) {
        hybrid_amg_cached(ctx, H0, tdiag, p, q, m, n, bk1, tk, z, opts, rng, zeta, out, nullptr);
        return;
    }
    // consecutive Newton steps with the same system (Hybrid_AMG.m:40-41 sets up per call)
    HybridCache cache;
    cache.record_donors = true;
    cache.use_donors = step->same;
    if (step->same) cache.donors = step->prev;
    cache.shared_count = &step->shared;
    step->prev.clear();
    hybrid_amg_cached(ctx, H0, tdiag, p, q, m, n, bk1, tk, z, opts, rng, zeta, out, &cache);
    step->prev = std::move(cache.recorded);
}

static void hybrid_amg_cached(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                              const double* q, int m, int n, double bk1, double tk, const double* z,
                              const AmgOpts& opts, ipd_rng* rng, double* zeta, HybridOut* out,
                              HybridCache* cache, Deferred* later) {
    IPD_REQUIRE(rng, IPD_E_ARG, "Hybrid_AMG needs a rand stream");
    auto defer = [&](std::function<void()> fn) {   // run now, or in the caller's solve phase
        if (later)
            later->push_back(std::move(fn));
        else
            fn();
    };
    IPD_REQUIRE(tk != 0.0, IPD_E_ARG, "tk must be nonzero");
    const int M = m + n;
    const int N0 = 100;                                                   // Hybrid_AMG.m:51
    Arena& tmp = *ctx->scratch;
    const bool reuse = cache && cache->valid;
    Csr Ae;
    if (reuse) {
        Ae = cache->Ae;
    } else {
        ProfScope ps(ctx, PROF_BUILD_AE);
        build_Ae(ctx, tmp, H0, tdiag, p, q, m, n, bk1, tk, &Ae);        // :17-24
    }
    double* f = tmp.alloc<double>((size_t)M);
    double* u = tmp.alloc<double>((size_t)M);
    double* dK = tmp.alloc<double>((size_t)M);
    const int g = elems_grid(M);
    hipLaunchKernelGGL(k_scale_qp, dim3(g), dim3(256), 0, ctx->stream, M, n, p, q, z, f);
    hipLaunchKernelGGL(k_dk, dim3(g), dim3(256), 0, ctx->stream, M, n, p, q, tdiag, dK);
    IPD_KERNEL_CHECK();
    IPD_HIP(hipMemsetAsync(u, 0, sizeof(double) * (size_t)M, ctx->stream));
    dump_system(ctx, Ae, f, n);
    // components of A0 = Q0*H0*Q0: same pattern as H0 (qp has no zeros)     :27
    Components cc_local;
    if (!reuse) {
        ProfScope ps(ctx, PROF_COMPONENTS);
        find_components(ctx, H0, &cc_local);
        if (cache) {
            cache->cc = cc_local;
            cache->Ae = Ae;
        }
    }
    const Components& cc = reuse ? cache->cc : cc_local;
    out->num_comp = cc.ncomp;
    std::vector<double> hdK;
    if (tdiag) {
        hdK.resize((size_t)M);
        ctx->fetch(dK, hdK.data(), (size_t)M);
    }
    auto sum_dk = [&](const int* idx, int cnt) {
        if (!tdiag) return 0.0;
        double s = 0.0;
        for (int k = 0; k < cnt; ++k) s += hdK[(size_t)(idx ? idx[k] : k)];
        return s;
    };
    const double gscale = bk1 * tk;
    if (cc.ncomp == 1) {                                                  // :30-48
        const int isnsp = sum_dk(nullptr, M) != 0.0 ? 0 : 1;
        MaskHint mh;
        mh.p = p;
        mh.q = q;
        mh.m = m;
        mh.n = n;
        mh.tk = tk;
        auto run = class_amg_prepare(ctx, Ae, f, opts, isnsp, n, gscale, rng, u, mh, cache);
        defer([run, out] {
            int it = 0;
            double rr = 0.0;
            run(&it, &rr);
            out->itamg = it;
            out->resamg = rr;
            out->it_num = 1;
        });
    } else {                                                              // :50-107
        out->itamg = 0;
        out->resamg = 0.0;
        out->it_num = 0;
        std::vector<int> newidx((size_t)M);
        int* d_new = tmp.alloc<int>((size_t)M);
        // [q; p] in the order of the unknowns: a large component's own q (its F rows) and p (its C rows) are
        // gathered from it for the mask form of its level 1 (amg_attach_maskop: the sub-matrix Ae(pk, pk) of a
        // component has the same rank-one structure, with the component's entries of p and q)
        double* qp = nullptr;
        for (int k = 0; k < cc.ncomp && !qp; ++k)
            if (cc.sizes[k] > RES_MASK_MIN_ROWS && opts.bigph) {
                qp = tmp.alloc<double>((size_t)M);
                IPD_HIP(hipMemcpyAsync(qp, q, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
                IPD_HIP(hipMemcpyAsync(qp + n, p, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, ctx->stream));
            }
        for (int k = 0; k < cc.ncomp; ++k) {                              // :55 large components
            if (cc.sizes[k] <= N0) continue;
            const int nk = cc.sizes[k];
            int* d_pk = tmp.alloc<int>((size_t)nk);   // per component: the scatter may run later
            const int* pk = cc.p.data() + cc.r[k];  // ascending: F side first (quirk A-9)
            std::fill(newidx.begin(), newidx.end(), -1);
            int fnode = 0;
            for (int r = 0; r < nk; ++r) {
                newidx[(size_t)pk[r]] = r;
                fnode += pk[r] < n;                                       // :68 sum(pk<=n)
            }
            ctx->upload(d_pk, pk, (size_t)nk);
            ctx->upload(d_new, newidx.data(), (size_t)M);
            Csr Ak;
            Ak.nr = Ak.nc = nk;
            int* rowlen = tmp.alloc<int>((size_t)nk + 1);
            Ak.rp = tmp.alloc<int>((size_t)nk + 1);
            hipLaunchKernelGGL(k_sub_count, dim3(elems_grid(nk)), dim3(256), 0, ctx->stream, nk,
                               d_pk, Ae.rp, rowlen);
            IPD_KERNEL_CHECK();
            Ak.nnz = exclusive_scan_total(ctx, rowlen, Ak.rp, nk);
            Ak.ci = tmp.alloc<int>((size_t)Ak.nnz);
            Ak.va = tmp.alloc<double>((size_t)Ak.nnz);
            hipLaunchKernelGGL(k_sub_fill, dim3(rows_grid(nk)), dim3(256), 0, ctx->stream, nk, d_pk,
                               d_new, Ae.rp, Ae.ci, Ae.va, Ak.rp, Ak.ci, Ak.va);
            double* fk = tmp.alloc<double>((size_t)nk);
            double* dk = tmp.alloc<double>((size_t)nk);
            hipLaunchKernelGGL(k_gather, dim3(elems_grid(nk)), dim3(256), 0, ctx->stream, nk, d_pk,
                               f, fk);
            IPD_KERNEL_CHECK();
            const int isnsp = sum_dk(pk, nk) != 0.0 ? 0 : 1;              // :60-66
            IPD_REQUIRE(fnode > 0 && fnode < nk, IPD_E_NUMERIC,
                        "Hybrid_AMG: a large component lies on one side of the bigraph");
            MaskHint mhk;
            if (qp && nk > RES_MASK_MIN_ROWS) {
                double* qpk = tmp.alloc<double>((size_t)nk);
                hipLaunchKernelGGL(k_gather, dim3(elems_grid(nk)), dim3(256), 0, ctx->stream, nk, d_pk,
                                   (const double*)qp, qpk);
                IPD_KERNEL_CHECK();
                mhk.q = qpk;
                mhk.p = qpk + fnode;
                mhk.n = fnode;
                mhk.m = nk - fnode;
                mhk.tk = tk;
            }
            auto run = class_amg_prepare(ctx, Ak, fk, opts, isnsp, fnode, gscale, rng, dk, mhk, cache);
            defer([run, out, ctx, nk, d_pk, dk, u, k] {
                int it = 0;
                double rr = 0.0;
                run(&it, &rr);
                hipLaunchKernelGGL(k_scatter, dim3(elems_grid(nk)), dim3(256), 0, ctx->stream, nk,
                                   (const int*)d_pk, (const double*)dk, u);  // :77 u(pk) = dk
                IPD_KERNEL_CHECK();
                out->itamg = std::max(out->itamg, it);
                out->resamg = std::max(out->resamg, rr);
                out->it_num = k + 1;                                      // :80 (1-based)
            });
        }
        // small components, all together                                   :85-91
        std::vector<int> nodes, boff(1, 0), local((size_t)M, 0);
        int maxnb = 0;
        for (int k = 0; k < cc.ncomp; ++k) {
            if (cc.sizes[k] > N0) continue;
            for (int r = cc.r[k]; r < cc.r[k + 1]; ++r) {
                local[(size_t)cc.p[r]] = r - cc.r[k];
                nodes.push_back(cc.p[r]);
            }
            boff.push_back((int)nodes.size());
            maxnb = std::max(maxnb, cc.sizes[k]);
        }
        if (!nodes.empty()) {
            ProfScope ps(ctx, PROF_SMALL_BLOCKS);
            const int nblk = (int)boff.size() - 1;
            int* d_boff = tmp.alloc<int>(boff.size());
            int* d_nodes = tmp.alloc<int>(nodes.size());
            int* d_local = tmp.alloc<int>((size_t)M);
            int* bad = tmp.alloc<int>(1);
            ctx->upload(d_boff, boff.data(), boff.size());
            ctx->upload(d_nodes, nodes.data(), nodes.size());
            ctx->upload(d_local, local.data(), (size_t)M);
            IPD_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
            const size_t lds = sizeof(double) * ((size_t)maxnb * (maxnb + 1) + maxnb);
            IPD_OPTIN_LDS(ctx, k_small_blocks, 96 * 1024);
            hipLaunchKernelGGL(k_small_blocks, dim3(nblk), dim3(256), lds, ctx->stream, d_boff,
                               d_nodes, d_local, Ae.rp, Ae.ci, Ae.va, f, u, bad);
            IPD_KERNEL_CHECK();
            IPD_REQUIRE(ctx->fetch1(bad) == 0, IPD_E_NUMERIC,
                        "Hybrid_AMG: a small diagonal block is not positive definite");
        }
    }
    defer([ctx, g, M, n, p, q, u, zeta] {
        hipLaunchKernelGGL(k_scale_qp, dim3(g), dim3(256), 0, ctx->stream, M, n, p, q,
                           (const double*)u, zeta);                       // :113 zeta = Q0*u
        IPD_KERNEL_CHECK();
    });
    if (cache) cache->valid = true;
}

// ---------------------------------------------------------------------------
// AMG4POT                                              (Class2/AMG4POT.m:27-55)
// ---------------------------------------------------------------------------
__global__ void k_mask_mul(size_t len, const uint8_t* __restrict__ s,
                           const double* __restrict__ phi, double* __restrict__ out,
                           double* __restrict__ part) {
    // out = s.*phi ; part[block] = sum phi.*(s.*phi)
    __shared__ double red[4];
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) {
        const double v = s[i] ? phi[i] : 0.0;
        out[i] = v;
        acc += phi[i] * v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

static double host_dot(ipd_ctx* ctx, const double* a, const double* b, int n) {
    std::vector<double> ha((size_t)n), hb((size_t)n);
    ctx->fetch(a, ha.data(), (size_t)n);
    ctx->fetch(b, hb.data(), (size_t)n);
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += ha[(size_t)i] * hb[(size_t)i];
    return s;
}

__global__ void k_axpby(int n, double a, const double* __restrict__ x, double b,
                        const double* __restrict__ y, double* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = a * x[i] + b * y[i];
}

template <class SOLVE>
static void pot_reduce(ipd_ctx* ctx, const Csr& H0, const double* p, const double* q, int m, int n,
                       double bk1, double tk, const double* z, const uint8_t* s, const double* phi,
                       double* zeta, HybridOut* out, SOLVE solve) {
    (void)H0;
    const int M = m + n;
    const size_t mn = (size_t)m * n;
    Arena& tmp = *ctx->scratch;
    const double epss = bk1, sg = 1.0 / tk;                               // :31
    double* sphi = tmp.alloc<double>(mn);
    const int nb = (int)std::max<size_t>(1, std::min<size_t>((mn + 255) / 256, 1024));
    double* part = tmp.alloc<double>((size_t)nb);
    hipLaunchKernelGGL(k_mask_mul, dim3(nb), dim3(256), 0, ctx->stream, mn, s, phi, sphi, part);
    IPD_KERNEL_CHECK();
    const double phi_e = epss + sg * host_sum(ctx, part, nb);             // :33
    double* v = tmp.alloc<double>((size_t)M);
    kkt_ax(ctx, sphi, p, q, m, n, v);                                     // :34 v = Ax(s.*phi)
    const double z2 = ctx->fetch1(z + M);
    double* w = tmp.alloc<double>((size_t)M);
    hipLaunchKernelGGL(k_axpby, dim3(elems_grid(M)), dim3(256), 0, ctx->stream, M, 1.0, z,
                       -(sg / phi_e * z2), (const double*)v, w);          // w = z1 - sg/phi_e*z2*v
    IPD_KERNEL_CHECK();
    double* vv = tmp.alloc<double>((size_t)M);
    double* ww = tmp.alloc<double>((size_t)M);
    HybridOut o1, o2;
    solve(v, vv, &o1);                                                    // :46
    solve(w, ww, &o2);                                                    // :47
    const double vvv = host_dot(ctx, v, vv, M);
    const double vww = host_dot(ctx, v, ww, M);
    const double tt = sg * sg / (phi_e - sg * sg * vvv);                  // :53
    hipLaunchKernelGGL(k_axpby, dim3(elems_grid(M)), dim3(256), 0, ctx->stream, M, 1.0,
                       (const double*)ww, tt * vww, (const double*)vv, zeta);  // zeta1
    IPD_KERNEL_CHECK();
    const double vz1 = host_dot(ctx, v, zeta, M);
    const double zeta2 = (z2 - sg * vz1) / phi_e;                         // :54
    ctx->upload(zeta + M, &zeta2, 1);
    out->itamg = std::max(o1.itamg, o2.itamg);                            // :55
    out->resamg = std::max(o1.resamg, o2.resamg);
    out->num_comp = std::max(o1.num_comp, o2.num_comp);
    out->it_num = std::max(o1.it_num, o2.it_num);
}

void amg4pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                 const double* q, int m, int n, double bk1, double tk, const double* z,
                 const uint8_t* s, const double* phi, const AmgOpts& opts, ipd_rng* rng,
                 double* zeta, HybridOut* out, StepDonors* step) {
    const char* nc = getenv("IPD_NO_POT_CONCURRENT");
    const bool concurrent = !(nc && nc[0] == '1');
    HybridCache cache;
    if (!concurrent) {
        if (step) step->prev.clear();
        auto solve = [&](const double* rhs, double* x, HybridOut* o) {
            hybrid_amg_cached(ctx, H0, tdiag, p, q, m, n, bk1, tk, rhs, opts, rng, x, o, nullptr);
        };
        pot_reduce(ctx, H0, p, q, m, n, bk1, tk, z, s, phi, zeta, out, solve);
        ctx->sync();
        return;
    }
    // The two systems Ae*vv = v and Ae*ww = w are independent (Class2/AMG4POT.m:46-47).  Their
    // guesses and hierarchies are drawn/built one after the other -- the rand stream is consumed
    // in the reference's order -- and their solve phases then run at the same time on two
    // streams: each is a chain of latency-bound launches that leaves the device mostly idle.
    ipd_ctx* aux = ipd_ctx_aux(ctx);
    AmgOpts popts = opts;
    popts.concurrent_pair = true;
    Deferred first, second;
    bool have_first = false;
    const char* nd = getenv("IPD_NO_DONOR");
    const bool donors = !(nd && nd[0] == '1');
    if (step && !donors) step->prev.clear();
    auto solve = [&](const double* rhs, double* x, HybridOut* o) {
        if (!have_first) {
            cache.record_donors = donors;
            if (step && donors && step->same) {   // the previous Newton step had this very system
                cache.use_donors = true;
                cache.donors = step->prev;
                cache.shared_count = &step->shared;
            }
            hybrid_amg_cached(ctx, H0, tdiag, p, q, m, n, bk1, tk, rhs, popts, rng, x, o,
                              donors ? &cache : nullptr, &first);
            cache.record_donors = false;
            cache.use_donors = donors;
            cache.donors = cache.recorded;   // the second setup shares the first one's levels
            cache.next_donor = 0;
            cache.shared_count = nullptr;
            have_first = true;
            return;
        }
        ctx->sync();   // rhs, H0, ... were produced on ctx's stream
        // (Letting the second setup overlap with the first solve phase as well was measured: no
        // gain -- the setup is bound by its host round trips -- so both setups stay on this
        // thread and only the solve phases, which consume no random numbers, run concurrently.)
        CallScope aux_scope(aux);
        // an injected component order (ipd_ctx_set_component_order) holds for both solves: without the
        // shared component cache (IPD_NO_DONOR=1) the second call finds the components itself
        aux->comp_order = ctx->comp_order;
        hybrid_amg_cached(aux, H0, tdiag, p, q, m, n, bk1, tk, rhs, popts, rng, x, o,
                          donors ? &cache : nullptr, &second);
        std::exception_ptr err;
        std::thread other([&] {
            try {
                aux->set_device();
                for (auto& fn : second) fn();
                aux->sync();
            } catch (...) {
                err = std::current_exception();
            }
        });
        std::exception_ptr err0;
        try {
            for (auto& fn : first) fn();
            ctx->sync();
        } catch (...) {
            err0 = std::current_exception();
        }
        other.join();
        first.clear();    // releases the hierarchies
        second.clear();
        cache.donors.clear();
        if (step && donors)
            step->prev = std::move(cache.recorded);   // kept for the next Newton step
        else
            cache.recorded.clear();
        if (err0) std::rethrow_exception(err0);
        if (err) std::rethrow_exception(err);
    };
    pot_reduce(ctx, H0, p, q, m, n, bk1, tk, z, s, phi, zeta, out, solve);
}

// ---------------------------------------------------------------------------
// aug_PCG                                                          (aug_PCG.m:11-36)
// ---------------------------------------------------------------------------
// augAe = [Y'QK Y  Y'QK ; QK Y  Ae] with Y the component indicators and QK = bk1*Q + K/tk
// diagonal, so the first nc rows are  [d_c , qk_i for the members i of c]  and row nc+i is
// [qk_i at column c(i) , row i of Ae shifted by nc]; columns come out sorted.
__global__ void k_aug_qk(int M, int n, const double* __restrict__ p, const double* __restrict__ q,
                         const double* __restrict__ t, double bk1, double itk,
                         double* __restrict__ qk) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const double qp = i < n ? q[i] : -p[i - n];
        const double kk = t ? (qp * t[i]) * qp : 0.0;          // K = Q0*T*Q0
        qk[i] = bk1 * (qp * qp) + itk * kk;                     // aug_PCG.m:27
    }
}
__global__ void k_aug_rowlen(int nc, int M, const int* __restrict__ cr, const int* __restrict__ rp,
                             int* __restrict__ len) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nc + M; r += gridDim.x * blockDim.x)
        len[r] = r < nc ? 1 + cr[r + 1] - cr[r] : 1 + rp[r - nc + 1] - rp[r - nc];
}
// one wave per row of the augmented matrix; also the right-hand side [Y'f ; f]
__global__ __launch_bounds__(256) void k_aug_fill(int nc, int M, const int* __restrict__ cr,
                                                  const int* __restrict__ cp,
                                                  const int* __restrict__ blocks,
                                                  const double* __restrict__ qk,
                                                  const double* __restrict__ f,
                                                  const int* __restrict__ rp,
                                                  const int* __restrict__ ci,
                                                  const double* __restrict__ va,
                                                  const int* __restrict__ orp, int* __restrict__ oci,
                                                  double* __restrict__ ova, double* __restrict__ of) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < nc + M; r += nwaves) {
        const int ob = orp[r];
        if (r < nc) {
            const int b = cr[r], e = cr[r + 1];
            if (lane == 0) {   // sums in ascending member order, like the sparse products
                double d = 0.0, fs = 0.0;
                for (int t = b; t < e; ++t) {
                    d += qk[cp[t]];
                    fs += f[cp[t]];
                }
                oci[ob] = r;
                ova[ob] = d;
                of[r] = fs;
            }
            for (int t = b + lane; t < e; t += 64) {
                oci[ob + 1 + (t - b)] = nc + cp[t];
                ova[ob + 1 + (t - b)] = qk[cp[t]];
            }
        } else {
            const int i = r - nc, b = rp[i];
            if (lane == 0) {
                oci[ob] = blocks[i];
                ova[ob] = qk[i];
                of[r] = f[i];
            }
            for (int t = b + lane; t < rp[i + 1]; t += 64) {
                oci[ob + 1 + (t - b)] = nc + ci[t];
                ova[ob + 1 + (t - b)] = va[t];
            }
        }
    }
}
__global__ void k_aug_back(int M, int n, int nc, const int* __restrict__ blocks,
                           const double* __restrict__ U, const double* __restrict__ p,
                           const double* __restrict__ q, double* __restrict__ zeta) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const double qp = i < n ? q[i] : -p[i - n];
        zeta[i] = qp * (U[blocks[i]] + U[nc + i]);              // :35-36
    }
}

void aug_pcg_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, double tol, long long maxit,
                 double* zeta, HybridOut* out) {
    IPD_REQUIRE(tk != 0.0, IPD_E_ARG, "tk must be nonzero");
    const int M = m + n;
    Arena& tmp = *ctx->scratch;
    Csr Ae;
    build_Ae(ctx, tmp, H0, tdiag, p, q, m, n, bk1, tk, &Ae);
    double* f = tmp.alloc<double>((size_t)M);
    const int g = elems_grid(M);
    hipLaunchKernelGGL(k_scale_qp, dim3(g), dim3(256), 0, ctx->stream, M, n, p, q, z, f);
    IPD_KERNEL_CHECK();
    Components cc;
    find_components(ctx, H0, &cc);                                                  // :24
    const int nc = cc.ncomp;
    int* d_cr = tmp.alloc<int>((size_t)nc + 1);
    int* d_cp = tmp.alloc<int>((size_t)M);
    int* d_bl = tmp.alloc<int>((size_t)M);
    ctx->upload(d_cr, cc.r.data(), (size_t)nc + 1);
    ctx->upload(d_cp, cc.p.data(), (size_t)M);
    ctx->upload(d_bl, cc.blocks.data(), (size_t)M);
    double* qk = tmp.alloc<double>((size_t)M);
    hipLaunchKernelGGL(k_aug_qk, dim3(g), dim3(256), 0, ctx->stream, M, n, p, q, tdiag, bk1, 1.0 / tk, qk);
    Csr aug;
    aug.nr = aug.nc = nc + M;
    int* len = tmp.alloc<int>((size_t)aug.nr + 1);
    aug.rp = tmp.alloc<int>((size_t)aug.nr + 1);
    hipLaunchKernelGGL(k_aug_rowlen, dim3(elems_grid(aug.nr)), dim3(256), 0, ctx->stream, nc, M,
                       (const int*)d_cr, (const int*)Ae.rp, len);
    IPD_KERNEL_CHECK();
    aug.nnz = exclusive_scan_total(ctx, len, aug.rp, aug.nr);
    aug.ci = tmp.alloc<int>((size_t)aug.nnz);
    aug.va = tmp.alloc<double>((size_t)aug.nnz);
    double* augf = tmp.alloc<double>((size_t)aug.nr);
    double* U = tmp.alloc<double>((size_t)aug.nr);
    hipLaunchKernelGGL(k_aug_fill, dim3(rows_grid(aug.nr)), dim3(256), 0, ctx->stream, nc, M,
                       (const int*)d_cr, (const int*)d_cp, (const int*)d_bl, (const double*)qk,
                       (const double*)f, (const int*)Ae.rp, (const int*)Ae.ci, (const double*)Ae.va,
                       (const int*)aug.rp, aug.ci, aug.va, augf);
    IPD_KERNEL_CHECK();
    long long it = 0;
    double res = 0.0;
    pcg_dev(ctx, aug, augf, nullptr, tol, maxit, 2, U, &it, &res, nullptr);         // :29-34
    hipLaunchKernelGGL(k_aug_back, dim3(g), dim3(256), 0, ctx->stream, M, n, nc, (const int*)d_bl,
                       (const double*)U, p, q, zeta);
    IPD_KERNEL_CHECK();
    out->itamg = (int)std::min<long long>(it, 2147483647LL);
    out->resamg = res;
    out->num_comp = nc;
    out->it_num = 1;
}

// PCG4POT (Class2/PCG4POT.m:27-39) = amg4pot_dev's reduction with aug_PCG as the inner solve
void pcg4pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                 const double* phi, double tol, long long maxit, double* zeta, HybridOut* out) {
    auto solve = [&](const double* rhs, double* x, HybridOut* o) {
        aug_pcg_dev(ctx, H0, tdiag, p, q, m, n, bk1, tk, rhs, tol, maxit, x, o);
    };
    pot_reduce(ctx, H0, p, q, m, n, bk1, tk, z, s, phi, zeta, out, solve);
}

// ---------------------------------------------------------------------------
// Jk = bk1*I + (T + H0)/tk                              (APD_SsN_Class1.m:151)
// ---------------------------------------------------------------------------
// H0 stores a diagonal entry only on rows that have active entries, Jk on every row.
__global__ void k_jk_count(int M, const int* __restrict__ rp, const int* __restrict__ ci,
                           int* __restrict__ len) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < M; r += gridDim.x * blockDim.x) {
        bool has = false;
        for (int t = rp[r]; t < rp[r + 1]; ++t) has = has || ci[t] == r;
        len[r] = rp[r + 1] - rp[r] + (has ? 0 : 1);
    }
}
__global__ void k_jk_fill(int M, const int* __restrict__ rp, const int* __restrict__ ci,
                          const double* __restrict__ va, const double* __restrict__ t, double bk1,
                          double tk, const int* __restrict__ orp, int* __restrict__ oci,
                          double* __restrict__ ova) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < M; r += gridDim.x * blockDim.x) {
        int o = orp[r];
        bool placed = false;
        const double tr = t ? t[r] : 0.0;
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            const int c = ci[k];
            if (!placed && c > r) {   // the missing diagonal goes in before the first larger column
                oci[o] = r;
                ova[o++] = bk1 + (tr + 0.0) / tk;
                placed = true;
            }
            oci[o] = c;
            if (c == r) {
                ova[o++] = bk1 + (tr + va[k]) / tk;
                placed = true;
            } else {
                ova[o++] = va[k] / tk;
            }
        }
        if (!placed) {
            oci[o] = r;
            ova[o] = bk1 + (tr + 0.0) / tk;
        }
    }
}
void build_jk(ipd_ctx* ctx, Arena& dst, const Csr& H0, const double* tdiag, double bk1, double tk,
              Csr* J) {
    const int M = H0.nr;
    int* len = ctx->scratch->alloc<int>((size_t)M + 1);
    J->nr = J->nc = M;
    J->rp = dst.alloc<int>((size_t)M + 1);
    hipLaunchKernelGGL(k_jk_count, dim3(elems_grid(M)), dim3(256), 0, ctx->stream, M, H0.rp, H0.ci, len);
    IPD_KERNEL_CHECK();
    J->nnz = exclusive_scan_total(ctx, len, J->rp, M);
    J->ci = dst.alloc<int>((size_t)J->nnz);
    J->va = dst.alloc<double>((size_t)J->nnz);
    hipLaunchKernelGGL(k_jk_fill, dim3(elems_grid(M)), dim3(256), 0, ctx->stream, M, H0.rp, H0.ci,
                       H0.va, tdiag, bk1, tk, (const int*)J->rp, J->ci, J->va);
    IPD_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// Class 2, inner_solver = 1 and 2: the bordered Jacobian itself      (APD_SsN_Class2.m:152-161)
//   J = bk1*speye(M+1) + (cT + cH0)/tk,  cT = [T 0; 0 0],  cH0 = [H0 ss; ss' phi'*(s.*phi)],
//   ss = Ax(s.*phi)
// assembled densely ((M+1)^2 doubles: cold paths, bounded by the 2 GiB scratch limit)
// ---------------------------------------------------------------------------
__global__ void k_border(int M, int ld, const double* __restrict__ ss, double itk /* tk */, double corner,
                         double* __restrict__ D) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= M; i += gridDim.x * blockDim.x) {
        if (i < M) {
            const double v = ss[i] / itk;            // (0 + ss)/tk
            D[(size_t)i * ld + M] = v;
            D[(size_t)M * ld + i] = v;
        } else {
            D[(size_t)M * ld + M] = corner;
        }
    }
}

static double* pot_jacobian_dense(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p,
                                  const double* q, int m, int n, double bk1, double tk,
                                  const uint8_t* s, const double* phi) {
    const int M = m + n, N = M + 1;
    const size_t mn = (size_t)m * n;
    Arena& tmp = *ctx->scratch;
    IPD_REQUIRE((size_t)N * N * 8 <= (size_t(2) << 30), IPD_E_LIMIT,
                "inner_solver 1/2 (class 2): dense Jacobian above 2 GiB");
    double* sphi = tmp.alloc<double>(mn);
    const int nb = (int)std::max<size_t>(1, std::min<size_t>((mn + 255) / 256, 1024));
    double* part = tmp.alloc<double>((size_t)nb);
    hipLaunchKernelGGL(k_mask_mul, dim3(nb), dim3(256), 0, ctx->stream, mn, s, phi, sphi, part);
    IPD_KERNEL_CHECK();
    const double pssp = host_sum(ctx, part, nb);                          // phi'*(s.*phi)
    double* ss = tmp.alloc<double>((size_t)M);
    kkt_ax(ctx, sphi, p, q, m, n, ss);                                    // ss = Ax(s.*phi)
    Csr Jk;
    build_jk(ctx, tmp, H0, tdiag, bk1, tk, &Jk);                          // bk1*I + (T + H0)/tk
    double* D = tmp.alloc<double>((size_t)N * N);
    IPD_HIP(hipMemsetAsync(D, 0, sizeof(double) * (size_t)N * N, ctx->stream));
    csr_expand_dense(ctx, Jk, D, N);
    hipLaunchKernelGGL(k_border, dim3(elems_grid(N)), dim3(256), 0, ctx->stream, M, N, (const double*)ss,
                       tk, bk1 + pssp / tk, D);
    IPD_KERNEL_CHECK();
    return D;
}

// zeta = J \ z                                                      (APD_SsN_Class2.m:155)
void direct_pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                    int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                    const double* phi, double* zeta) {
    const int N = m + n + 1;
    double* D = pot_jacobian_dense(ctx, H0, tdiag, p, q, m, n, bk1, tk, s, phi);
    dense_chol_factor(ctx, D, N, N);
    if (zeta != z)
        IPD_HIP(hipMemcpyAsync(zeta, z, sizeof(double) * (size_t)N, hipMemcpyDeviceToDevice, ctx->stream));
    dense_chol_solve(ctx, D, N, N, zeta, 1, 1);
}

// [zeta,itpcg,respcg] = PCG(J,z,pcg_options)                         (APD_SsN_Class2.m:161)
void pcg_pot_dev(ipd_ctx* ctx, const Csr& H0, const double* tdiag, const double* p, const double* q,
                 int m, int n, double bk1, double tk, const double* z, const uint8_t* s,
                 const double* phi, double tol, long long maxit, double* zeta, HybridOut* out) {
    const int N = m + n + 1;
    Arena& tmp = *ctx->scratch;
    double* D = pot_jacobian_dense(ctx, H0, tdiag, p, q, m, n, bk1, tk, s, phi);
    Csr J;
    J.nr = J.nc = N;
    int* cnt = zeroed<int>(ctx, (size_t)N + 1);   // (biased counts: see ScanTail)
    J.rp = tmp.alloc<int>((size_t)N + 1);
    {
        TailTotal tt(ctx, cnt, J.rp, N);
        dense_rowcount(ctx, N, N, N, D, cnt, tt.t);
        int two[2] = {0, 0};
        tt.wait(two);
        J.nnz = two[0];
    }
    J.ci = tmp.alloc<int>((size_t)J.nnz);
    J.va = tmp.alloc<double>((size_t)J.nnz);
    dense_compact(ctx, N, N, N, D, J);
    long long it = 0;
    double res = 0.0;
    pcg_dev(ctx, J, z, nullptr, tol, maxit, 2, zeta, &it, &res, nullptr);
    out->itamg = (int)std::min<long long>(it, 2147483647LL);
    out->resamg = res;
    out->num_comp = 0;
    out->it_num = 0;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int ipd_components(ipd_ctx* ctx, const ipd_csc* A, int64_t* blocks, int64_t* sizes,
                              int64_t* p, int64_t* r, int64_t* ncomp) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && A && blocks && sizes && p && r && ncomp, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(A->nrows == A->ncols, IPD_E_ARG, "Adjacency matrix must be square");
        CallScope scope(ctx);
        CompOrderScope order_scope(ctx);
        Csr a;
        csr_upload_from_csc(ctx, *ctx->scratch, A, true, &a);  // undirected graph: pattern symmetric
        Components cc;
        find_components(ctx, a, &cc);
        *ncomp = cc.ncomp;
        for (int i = 0; i < a.nr; ++i) {
            blocks[i] = cc.blocks[(size_t)i];
            p[i] = cc.p[(size_t)i];
        }
        for (int c = 0; c < cc.ncomp; ++c) sizes[c] = cc.sizes[(size_t)c];
        for (int c = 0; c <= cc.ncomp; ++c) r[c] = cc.r[(size_t)c];
    });
}

static void check_prob(const ipd_prob* pd, bool pot) {
    IPD_REQUIRE(pd && pd->p && pd->q && pd->H0 && pd->z, IPD_E_ARG, "prob_data: NULL field");
    IPD_REQUIRE(pd->m > 0 && pd->n > 0 && pd->m + pd->n < (int64_t(1) << 30), IPD_E_ARG,
                "prob_data: bad m/n");
    if (pot) IPD_REQUIRE(pd->s && pd->phi, IPD_E_ARG, "AMG4POT needs prob_data.s and .phi");
}

static int hybrid_host(ipd_ctx* ctx, const ipd_prob* pd, const AmgOpts& ao, ipd_rng* rng,
                       double* zeta, int32_t* itamg, double* resamg, int64_t info[2]) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && zeta, IPD_E_ARG, "NULL argument");
        check_prob(pd, false);
        CallScope scope(ctx);
        CompOrderScope order_scope(ctx);
        Arena& tmp = *ctx->scratch;
        const int m = (int)pd->m, n = (int)pd->n, M = m + n;
        Csr H0;
        csr_upload_from_csc(ctx, tmp, pd->H0, true, &H0);
        double* dp = tmp.alloc<double>((size_t)m);
        double* dq = tmp.alloc<double>((size_t)n);
        double* dz = tmp.alloc<double>((size_t)M);
        double* dt = nullptr;
        ctx->upload(dp, pd->p, (size_t)m);
        ctx->upload(dq, pd->q, (size_t)n);
        ctx->upload(dz, pd->z, (size_t)M);
        if (pd->t) {
            dt = tmp.alloc<double>((size_t)M);
            ctx->upload(dt, pd->t, (size_t)M);
        }
        double* dzeta = tmp.alloc<double>((size_t)M);
        HybridOut ho;
        hybrid_amg_dev(ctx, H0, dt, dp, dq, m, n, pd->bk1, pd->tk, dz, ao, rng, dzeta, &ho);
        ctx->fetch(dzeta, zeta, (size_t)M);
        if (itamg) *itamg = ho.itamg;
        if (resamg) *resamg = ho.resamg;
        if (info) {
            info[0] = ho.num_comp;
            info[1] = ho.it_num;
        }
    });
}

extern "C" int ipd_hybrid_amg(ipd_ctx* ctx, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                              double* zeta, int32_t* itamg, double* resamg, int64_t info[2]) {
    return hybrid_host(ctx, pd, amg_fill_defaults(o), rng, zeta, itamg, resamg, info);
}
extern "C" int ipd_hybrid_twogrid(ipd_ctx* ctx, const ipd_prob* pd, const ipd_amg_opts* o,
                                  ipd_rng* rng, double* zeta, int32_t* itamg, double* resamg,
                                  int64_t info[2]) {
    return hybrid_host(ctx, pd, amg_fill_twogrid_defaults(o), rng, zeta, itamg, resamg, info);
}

extern "C" int ipd_hybrid_amg_dev(ipd_ctx* ctx, const ipd_dmat* H0, const double* t_dev,
                                  const double* p_dev, const double* q_dev, int64_t m, int64_t n,
                                  double bk1, double tk, const double* z_dev,
                                  const ipd_amg_opts* o, ipd_rng* rng, double* zeta_dev,
                                  int32_t* itamg, double* resamg, int64_t info[2]) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && H0 && p_dev && q_dev && z_dev && zeta_dev, IPD_E_ARG, "NULL argument");
        IPD_REQUIRE(m > 0 && n > 0, IPD_E_ARG, "bad m/n");
        CallScope scope(ctx);
        CompOrderScope order_scope(ctx);
        HybridOut ho;
        hybrid_amg_dev(ctx, H0->m, t_dev, p_dev, q_dev, (int)m, (int)n, bk1, tk, z_dev,
                       amg_fill_defaults(o), rng, zeta_dev, &ho);
        ctx->sync();
        if (itamg) *itamg = ho.itamg;
        if (resamg) *resamg = ho.resamg;
        if (info) {
            info[0] = ho.num_comp;
            info[1] = ho.it_num;
        }
    });
}

static int amg4pot_host(ipd_ctx* ctx, const ipd_prob* pd, const AmgOpts& ao, ipd_rng* rng,
                        double* zeta, int32_t* itamg, double* resamg, int64_t info[2]) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && zeta, IPD_E_ARG, "NULL argument");
        check_prob(pd, true);
        CallScope scope(ctx);
        CompOrderScope order_scope(ctx);
        Arena& tmp = *ctx->scratch;
        const int m = (int)pd->m, n = (int)pd->n, M = m + n;
        const size_t mn = (size_t)m * n;
        Csr H0;
        csr_upload_from_csc(ctx, tmp, pd->H0, true, &H0);
        double* dp = tmp.alloc<double>((size_t)m);
        double* dq = tmp.alloc<double>((size_t)n);
        double* dz = tmp.alloc<double>((size_t)M + 1);
        double* dphi = tmp.alloc<double>(mn);
        uint8_t* ds = tmp.alloc<uint8_t>(mn);
        double* dt = nullptr;
        ctx->upload(dp, pd->p, (size_t)m);
        ctx->upload(dq, pd->q, (size_t)n);
        ctx->upload(dz, pd->z, (size_t)M + 1);
        ctx->upload(dphi, pd->phi, mn);
        ctx->upload(ds, pd->s, mn);
        if (pd->t) {
            dt = tmp.alloc<double>((size_t)M);
            ctx->upload(dt, pd->t, (size_t)M);
        }
        double* dzeta = tmp.alloc<double>((size_t)M + 1);
        HybridOut ho;
        amg4pot_dev(ctx, H0, dt, dp, dq, m, n, pd->bk1, pd->tk, dz, ds, dphi, ao, rng, dzeta, &ho);
        ctx->fetch(dzeta, zeta, (size_t)M + 1);
        if (itamg) *itamg = ho.itamg;
        if (resamg) *resamg = ho.resamg;
        if (info) {
            info[0] = ho.num_comp;
            info[1] = ho.it_num;
        }
    });
}

extern "C" int ipd_amg4pot(ipd_ctx* ctx, const ipd_prob* pd, const ipd_amg_opts* o, ipd_rng* rng,
                           double* zeta, int32_t* itamg, double* resamg, int64_t info[2]) {
    return amg4pot_host(ctx, pd, amg_fill_defaults(o), rng, zeta, itamg, resamg, info);
}
// AMG4POT(prob_data, amg_options, 'twogrid')                      Class2/AMG4POT.m:48-51
extern "C" int ipd_amg4pot_twogrid(ipd_ctx* ctx, const ipd_prob* pd, const ipd_amg_opts* o,
                                   ipd_rng* rng, double* zeta, int32_t* itamg, double* resamg,
                                   int64_t info[2]) {
    return amg4pot_host(ctx, pd, amg_fill_twogrid_defaults(o), rng, zeta, itamg, resamg, info);
}

// [zeta,itpcg,respcg,info] = aug_PCG(prob_data,pcg_options) (aug_PCG.m:1) and
// PCG4POT(prob_data,pcg_options) (Class2/PCG4POT.m:1); precd is forced to 2 (aug_PCG.m:32)
static int aug_host(ipd_ctx* ctx, const ipd_prob* pd, const ipd_pcg_opts* o, bool pot, double* zeta,
                    int64_t* itpcg, double* respcg, int64_t info[2]) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && zeta, IPD_E_ARG, "NULL argument");
        check_prob(pd, pot);
        const double tol = (o && o->retol >= 0) ? o->retol : 1e-11;                 // PCG.m:25-26
        const long long maxit = (o && o->maxit >= 0) ? o->maxit : 10000;
        CallScope scope(ctx);
        CompOrderScope order_scope(ctx);
        Arena& tmp = *ctx->scratch;
        const int m = (int)pd->m, n = (int)pd->n, M = m + n;
        const size_t mn = (size_t)m * n;
        Csr H0;
        csr_upload_from_csc(ctx, tmp, pd->H0, true, &H0);
        double* dp = tmp.alloc<double>((size_t)m);
        double* dq = tmp.alloc<double>((size_t)n);
        double* dz = tmp.alloc<double>((size_t)M + 1);
        double* dt = nullptr;
        ctx->upload(dp, pd->p, (size_t)m);
        ctx->upload(dq, pd->q, (size_t)n);
        ctx->upload(dz, pd->z, (size_t)M + (pot ? 1 : 0));
        if (pd->t) {
            dt = tmp.alloc<double>((size_t)M);
            ctx->upload(dt, pd->t, (size_t)M);
        }
        double* dzeta = tmp.alloc<double>((size_t)M + 1);
        HybridOut ho;
        if (pot) {
            double* dphi = tmp.alloc<double>(mn);
            uint8_t* ds = tmp.alloc<uint8_t>(mn);
            ctx->upload(dphi, pd->phi, mn);
            ctx->upload(ds, pd->s, mn);
            pcg4pot_dev(ctx, H0, dt, dp, dq, m, n, pd->bk1, pd->tk, dz, ds, dphi, tol, maxit, dzeta, &ho);
        } else {
            aug_pcg_dev(ctx, H0, dt, dp, dq, m, n, pd->bk1, pd->tk, dz, tol, maxit, dzeta, &ho);
        }
        ctx->fetch(dzeta, zeta, (size_t)M + (pot ? 1 : 0));
        if (itpcg) *itpcg = ho.itamg;
        if (respcg) *respcg = ho.resamg;
        if (info) {
            info[0] = ho.num_comp;
            info[1] = ho.it_num;
        }
    });
}
extern "C" int ipd_aug_pcg(ipd_ctx* ctx, const ipd_prob* pd, const ipd_pcg_opts* o, double* zeta,
                           int64_t* itpcg, double* respcg, int64_t info[2]) {
    return aug_host(ctx, pd, o, false, zeta, itpcg, respcg, info);
}
extern "C" int ipd_pcg4pot(ipd_ctx* ctx, const ipd_prob* pd, const ipd_pcg_opts* o, double* zeta,
                           int64_t* itpcg, double* respcg, int64_t info[2]) {
    return aug_host(ctx, pd, o, true, zeta, itpcg, respcg, info);
}
