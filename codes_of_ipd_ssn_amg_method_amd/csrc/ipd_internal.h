// Internal declarations shared by the translation units of libipdamg.
// Not part of the C ABI (that is include/ipd_amg.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ipd_amg.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
struct IpdError : public std::runtime_error {
    int code;
    IpdError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void ipd_set_error(const std::string& msg);

#define IPD_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess)                                                          \
            throw IpdError(IPD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define IPD_REQUIRE(cond, code, msg)                 \
    do {                                             \
        if (!(cond)) throw IpdError((code), (msg));  \
    } while (0)

// Wraps a C-ABI body: converts exceptions into status codes + ipd_last_error().
template <class F>
static inline int ipd_guard(F&& f) {
    try {
        f();
        return IPD_OK;
    } catch (const IpdError& e) {
        ipd_set_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        ipd_set_error("out of host memory");
        return IPD_E_NOMEM;
    } catch (const std::exception& e) {
        ipd_set_error(e.what());
        return IPD_E_ARG;
    }
}

#define IPD_KERNEL_CHECK() IPD_HIP(hipGetLastError())

// ---------------------------------------------------------------------------
// device memory: chunk pool + bump arenas
// ---------------------------------------------------------------------------
// hipMalloc is slow (tens of microseconds) and the AMG setup is redone for
// every Newton step, so all temporaries and all hierarchy storage come out of
// bump arenas whose chunks are recycled through a per-context pool.
struct ChunkPool {
    struct Chunk {
        char* base;
        size_t bytes;
    };
    std::vector<Chunk> free_chunks;
    size_t total_bytes = 0;
    Chunk get(size_t min_bytes);
    void put(Chunk c) { free_chunks.push_back(c); }
    void release_all();
};

struct Arena {
    ChunkPool* pool = nullptr;
    std::vector<ChunkPool::Chunk> chunks;
    size_t cur = 0;   // index of the chunk being filled
    size_t off = 0;   // offset inside it
    explicit Arena(ChunkPool* p = nullptr) : pool(p) {}
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    ~Arena() { release(); }
    void* alloc_bytes(size_t bytes);
    template <class T>
    T* alloc(size_t n) {
        return static_cast<T*>(alloc_bytes((n ? n : 1) * sizeof(T)));
    }
    void reset() {  // keep chunks, forget contents
        cur = 0;
        off = 0;
    }
    void release();  // give chunks back to the pool
};

// ---------------------------------------------------------------------------
// device sparse matrix (CSR, int32 indices, fp64 values, sorted columns)
// ---------------------------------------------------------------------------
struct Csr {
    int nr = 0, nc = 0, nnz = 0;
    int* rp = nullptr;     // nr+1
    int* ci = nullptr;     // nnz
    double* va = nullptr;  // nnz
};

struct ipd_dmat {
    ipd_ctx* ctx = nullptr;
    std::unique_ptr<Arena> arena;  // owns the storage of `m`
    Csr m;
};

// ---------------------------------------------------------------------------
// MATLAB-compatible rand stream
// ---------------------------------------------------------------------------
struct ipd_rng {
    bool replay = false;
    uint32_t mt[624];
    int mti = 625;
    std::vector<double> values;  // replay mode
    int64_t consumed = 0;
    void seed(uint32_t s);
    uint32_t next_u32();
    double next_double();  // genrand_res53, as MATLAB's mt19937ar rand
    void fill(double* out, int64_t n);
};

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct RcclState;  // ipd_dist.cpp

struct ipd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    ChunkPool pool;
    std::unique_ptr<Arena> scratch;  // per-call temporaries (reset by CallScope)
    void* pinned = nullptr;          // host staging for small readbacks
    size_t pinned_bytes = 0;
    char* up_ring = nullptr;         // pinned staging ring of upload_bytes (ipd_core.cpp)
    size_t up_ring_bytes = 0, up_ring_off = 0;
    // mailbox for scalar readbacks: host-coherent memory a one-wave kernel writes the words and
    // then a ticket into, while the host spins on the ticket (no copy engine, no stream wait)
    volatile unsigned* mailbox = nullptr;   // [0] ticket, [16..240) payload words
    unsigned mailbox_ticket = 0;
    int num_cu = 256;
    long long asat_nnz_hint = 0;   // entries of the last ASAt result (sizes the next one's arrays)
    // entries of P, Pt*A and Ac of the last hierarchy's level k (amg_transfer's lazy counts: the kernel-choice
    // heuristics of the next hierarchy's products run on these estimates; 0 = none yet)
    int xfer_hint[40][4] = {};   // [3]: longest row of P'A
    void* asat_agg = nullptr;      // k_asat_small's chained-scan words (ipd_kkt.hip)
    // Zero pool: temporaries that must start out as zeros (transpose bitmaps, dense operand blocks, flags) are
    // bumped out of one block that a single memset clears again at the start of the next hierarchy build --
    // one fill per setup where there were fourteen.  zalloc'd memory dies when the function that took it returns
    // (amg_setup calls zreset; nothing that holds such memory may call amg_setup).
    char* zpool = nullptr;
    size_t zpool_bytes = 0, zpool_used = 0;
    int scope_depth = 0;           // CallScope nesting (the outermost scope also resets the zero pool)
    void* zalloc(size_t bytes);    // nullptr when the pool has no room (callers fall back to alloc + memset)
    void zreset();
    // injected visiting order of the connected components (ipd_ctx_set_component_order): the
    // smallest member of the component to visit k-th; empty = by smallest member ascending
    std::vector<int> comp_order;
    RcclState* comm = nullptr;
    ipd_ctx* aux = nullptr;   // second stream/arena for work that overlaps with this context's
    // Level-resident kernel (ipd_resident.h): launches of this context whose bounded spins gave up
    // (the workgroups were not all on the chip: the GPU is shared with something this process does
    // not see), and the number of solves that stay on the multi-launch path before the next attempt
    // (32 after the first give-up, doubling: a stall costs ~0.5 s, a run builds hundreds of
    // hierarchies -- counting per hierarchy would pay the stall on every one of them)
    int res_giveups = 0;
    int res_penalty = 0;
    // a pair of timing events kept for the context's lifetime (bench hooks: creating and destroying
    // a pair per call costs ~10 us of the timed region)
    hipEvent_t tev[2] = {nullptr, nullptr};

    // read back `n` elements synchronously through the pinned staging buffer
    template <class T>
    void fetch(const T* dsrc, T* hdst, size_t n) {
        fetch_bytes(dsrc, hdst, n * sizeof(T));
    }
    template <class T>
    T fetch1(const T* dsrc) {
        T v;
        fetch_bytes(dsrc, &v, sizeof(T));
        return v;
    }
    void fetch_bytes(const void* dsrc, void* hdst, size_t bytes);
    // For kernels that post their own scalar results: take a ticket, pass `mailbox` and the
    // ticket to the kernel (which stores <= 224 words at mailbox[16..] and then the ticket at
    // mailbox[0], see k_mailbox), then wait.  Returns false when the mailbox is switched off.
    bool mailbox_begin(unsigned* ticket);
    void mailbox_wait(unsigned ticket, void* hdst, size_t bytes);
    void upload_bytes(void* ddst, const void* hsrc, size_t bytes);
    template <class T>
    void upload(T* ddst, const T* hsrc, size_t n) {
        upload_bytes(ddst, hsrc, n * sizeof(T));
    }
    void sync() { IPD_HIP(hipStreamSynchronize(stream)); }
    void set_device() { IPD_HIP(hipSetDevice(device)); }
};

// Resets the scratch arena when a top-level call ends (nesting-aware).
struct CallScope {
    ipd_ctx* ctx;
    size_t cur, off;
    explicit CallScope(ipd_ctx* c) : ctx(c), cur(c->scratch->cur), off(c->scratch->off) {
        c->set_device();
        if (c->scope_depth++ == 0 && c->zpool_used > (c->zpool_bytes >> 1)) c->zreset();
    }
    ~CallScope() {
        ctx->scratch->cur = cur;
        ctx->scratch->off = off;
        --ctx->scope_depth;
    }
};

ipd_ctx* ipd_ctx_aux(ipd_ctx* ctx);   // ipd_core.cpp

// ipd_dist.cpp: RCCL communicator of the context (one rank per GPU)
void ipd_comm_cleanup(ipd_ctx* ctx);
int comm_rank(const ipd_ctx* ctx);
int comm_size(const ipd_ctx* ctx);
void comm_allgather_inplace(ipd_ctx* ctx, double* const* bases, int nvec, int count);

// Wall-clock attribution for tools/bench_driver.py (IPD_PROFILE=1): a scope synchronises the
// stream at both ends, so it is off unless asked for.
enum ProfSlot { PROF_ASAT, PROF_BUILD_AE, PROF_COMPONENTS, PROF_AMG_SETUP, PROF_AMG_SOLVE,
                PROF_SMALL_BLOCKS, PROF_EVAL, PROF_BEGIN_END, PROF_SLOTS };
bool ipd_prof_enabled();
void ipd_prof_add(int slot, double seconds);
struct ProfScope {
    ipd_ctx* ctx;
    int slot;
    double t0 = 0.0;
    ProfScope(ipd_ctx* c, int s);
    ~ProfScope();
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// scratch temporaries that start out as zeros: out of the zero pool when it has room
template <class T>
static inline T* zeroed(ipd_ctx* ctx, size_t n) {
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (void* p = ctx->zalloc(bytes)) return static_cast<T*>(p);
    T* q = ctx->scratch->alloc<T>(n);
    IPD_HIP(hipMemsetAsync(q, 0, bytes, ctx->stream));
    return q;
}

// ---------------------------------------------------------------------------
// scans that ride on the launch that produces the counts
// ---------------------------------------------------------------------------
// A launch that produces per-row counts is followed by an exclusive scan of them (the row pointers of the
// matrix being built).  Instead of a second launch, the workgroup with the highest index does the scan at the
// end of the producer: the counts array starts out as zeros (zero pool), every producer stores count + 1 with
// an agent-scope store (scan_put: written through, visible across the XCDs' L2s without a fence), and the
// scanning workgroup polls the entries until none is zero.  It is the last workgroup to be dispatched, so every
// entry it waits for belongs to a workgroup that is running or done; a bounded spin turns a missing entry into
// a negative total (the host raises) instead of a hang.  Bit 30 of an entry is a flag the producer may set
// (scan_put's third argument); the OR of the flags travels beside the total.
struct ScanTail {
    const int* in = nullptr;     // n biased counts (zeros before the launch)
    int* out = nullptr;          // n+1 row pointers (nullptr: no tail)
    const int* in2 = nullptr;    // optionally a second array of the same length
    int* out2 = nullptr;
    int n = 0;
    int* extra = nullptr;        // device copy of the total (lazy counts), or nullptr
    int extra2 = 0;              // ... extra[1] receives the second total / the OR of the flags as well
    volatile unsigned* box = nullptr;   // host mailbox (ipd_ctx::mailbox_wait): total, second total / flags
    unsigned ticket = 0;
};

// Host side of a tail whose total the host waits for: through the mailbox when it is on, else through a device
// word and a fetch.
struct TailTotal {
    ipd_ctx* ctx;
    ScanTail t;
    unsigned ticket = 0;
    bool mail = false;
    int* dev = nullptr;
    TailTotal(ipd_ctx* c, const int* in, int* out, int n) : ctx(c) {
        t.in = in;
        t.out = out;
        t.n = n;
        mail = c->mailbox_begin(&ticket);
        if (mail) {
            t.box = c->mailbox;
            t.ticket = ticket;
        } else {
            dev = c->scratch->alloc<int>(2);
            t.extra = dev;
            t.extra2 = 1;
        }
    }
    void wait(int* two) {   // two[0] = total, two[1] = second total or the OR of the flags (0 / 1)
        if (mail)
            ctx->mailbox_wait(ticket, two, 2 * sizeof(int));
        else
            ctx->fetch(dev, two, 2);
        IPD_REQUIRE(two[0] >= 0 && two[1] >= 0, IPD_E_HIP, "row-pointer scan: a count never arrived");
    }
};
// csr_spgemm(..., post): the compaction of a lazily counted product posts post->n device words starting at
// post->src to the host mailbox (box != NULL afterwards: wait for post->ticket; else fetch them)
struct LazyPost {
    const int* src = nullptr;
    int n = 0;
    volatile unsigned* box = nullptr;
    unsigned ticket = 0;
};
static inline ScanTail scan_tail_lazy(const int* in, int* out, int n, int* total_dev) {
    ScanTail t;
    t.in = in;
    t.out = out;
    t.n = n;
    t.extra = total_dev;
    return t;
}

#ifdef __HIPCC__
__device__ __forceinline__ void scan_put(int* cnt, int i, int count, bool flag = false) {
    __hip_atomic_store(cnt + i, (count + 1) | (flag ? (1 << 30) : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive scan of n plain ints by the calling workgroup (any multiple of 64 threads up to 1024): a contiguous
// chunk per thread, the chunk sums scanned across the workgroup.  Returns the total, also stored at out[n].
__device__ __forceinline__ int ipd_scan_counts(const int* in, int* out, int n, int* wsum) {
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, w = tid >> 6, nwv = T >> 6;
    const int chunk = (n + T - 1) / T;
    const int b = min(n, tid * chunk), e = min(n, b + chunk);
    int s = 0;
    for (int i = b; i < e; ++i) s += in[i];
    int x = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0, total = 0;
    for (int k = 0; k < nwv; ++k) {
        const int t = wsum[k];
        if (k < w) woff += t;
        total += t;
    }
    int run = woff + x - s;
    for (int i = b; i < e; ++i) {
        const int v = in[i];
        out[i] = run;
        run += v;
    }
    if (tid == 0) out[n] = total;
    return total;
}

constexpr int SCAN_TAIL_C = 8;   // entries per thread and pass of the scanning workgroup
struct ScanTailLds {
    int wsum[SCAN_TAIL_C * 16];   // (entries per thread) x (waves)
};

// polls, decodes and scans one array; returns the total (negative: an entry never arrived), ORs the flags.
// Entry j*T + tid of a pass sits in register j of thread tid: loads and stores are coalesced, the scan runs
// over the waves' inclusive scans of each register row.
__device__ __forceinline__ int scan_tail_one(const int* in, int* out, int n, ScanTailLds& L, int* flags_out) {
    constexpr int C = SCAN_TAIL_C;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, w = tid >> 6, nwv = T >> 6;
    int carry = 0, flags = 0;
    bool dead = false;
    for (int base = 0; base < n; base += C * T) {
        int v[C];
        unsigned spins = 0;
        bool ok;
        do {   // agent-scope loads, all in flight together; again until no entry is zero
            ok = true;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int idx = base + j * T + tid;
                v[j] = idx < n ? __hip_atomic_load(in + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
            }
#pragma unroll
            for (int j = 0; j < C; ++j) ok &= v[j] != 0;
            if (!ok) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1u << 20)) {
                    dead = true;
                    ok = true;
                }
            }
        } while (!ok);
        int x[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            flags |= v[j] & (1 << 30);
            v[j] = (v[j] & 0x3fffffff) - 1;
            int t = v[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(t, d);
                if (lane >= d) t += y;
            }
            x[j] = t;
            if (lane == 63) L.wsum[j * nwv + w] = t;
        }
        __syncthreads();
        int run = carry;   // sum of the (row, wave) blocks before the one being written
#pragma unroll
        for (int j = 0; j < C; ++j)
            for (int k = 0; k < nwv; ++k) {
                const int t = L.wsum[j * nwv + k];
                if (k == w) {
                    const int idx = base + j * T + tid;
                    if (idx < n) out[idx] = run + x[j] - v[j];
                }
                run += t;
            }
        carry = run;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
    if (__syncthreads_or(flags != 0)) *flags_out = 1;
    return __syncthreads_or(dead) ? -1 : carry;
}

// Called by EVERY thread of EVERY workgroup at the very end of the producer kernel (it holds barriers).
// Workgroups of 256 threads and more (a pass covers 8 entries per thread).
__device__ __forceinline__ void scan_tail(const ScanTail& s) {
    if (!s.out) return;
    if (blockIdx.x != gridDim.x - 1 || blockIdx.y != gridDim.y - 1) return;
    __shared__ ScanTailLds L;
    int flags = 0;
    const int t1 = scan_tail_one(s.in, s.out, s.n, L, &flags);
    const int t2 = s.in2 ? scan_tail_one(s.in2, s.out2, s.n, L, &flags) : flags;
    if (threadIdx.x == 0) {
        if (s.extra) {
            s.extra[0] = t1;
            if (s.extra2) s.extra[1] = t2;
        }
        if (s.box) {
            s.box[16] = (unsigned)t1;
            s.box[17] = (unsigned)t2;
            __threadfence_system();
            s.box[0] = s.ticket;
        }
    }
}

// The other way round, for producers with one-wave workgroups: the CONSUMER's workgroups each scan the (plain)
// counts for themselves -- n <= SCAN_HEAD_MAX ints out of L2, a microsecond -- and workgroup 0 also stores the
// row pointers and the total for whoever comes later.  256 threads.
constexpr int SCAN_HEAD_MAX = 4096;
struct ScanHeadLds {
    int rp[SCAN_HEAD_MAX + 1];
    int wsum[4];
};
__device__ __forceinline__ int scan_head(const int* __restrict__ cnt, int n, int* rp_out, int* total_out,
                                         ScanHeadLds& L, int* max_out = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int C = (n + 255) >> 8;   // a contiguous chunk per thread, at most 16 entries
    int v[SCAN_HEAD_MAX / 256];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_HEAD_MAX / 256; ++j) {
        const int idx = tid * C + j;
        v[j] = (j < C && idx < n) ? cnt[idx] : 0;
        s += v[j];
    }
    int x = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) L.wsum[w] = x;
    __syncthreads();
    int woff = 0, carry = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = L.wsum[k];
        if (k < w) woff += t;
        carry += t;
    }
    int run = woff + x - s;
#pragma unroll
    for (int j = 0; j < SCAN_HEAD_MAX / 256; ++j) {
        const int idx = tid * C + j;
        if (j < C && idx < n) L.rp[idx] = run;
        run += v[j];
    }
    if (tid == 0) L.rp[n] = carry;
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int i = tid; i <= n; i += 256) rp_out[i] = L.rp[i];
        if (tid == 0 && total_out) *total_out = carry;
        if (max_out) {   // the longest row (the next hierarchy's product model wants it: the row kernels are as slow
                         // as their longest row)
            int mx = 0;
#pragma unroll
            for (int j = 0; j < SCAN_HEAD_MAX / 256; ++j) mx = max(mx, v[j]);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
            __syncthreads();
            if (lane == 0) L.wsum[w] = mx;
            __syncthreads();
            if (tid == 0) *max_out = max(max(L.wsum[0], L.wsum[1]), max(L.wsum[2], L.wsum[3]));
        }
    }
    return carry;
}
#endif

// Opt-in to more than 64 KiB of dynamic LDS.  The attribute is per DEVICE (a process may hold
// contexts on several), so it is remembered per (kernel, device) pair, under a lock (contexts
// are used from several host threads).  ipd_core.cpp
void ipd_lds_optin(const void* kernel, int device, int bytes);
#define IPD_OPTIN_LDS(ctx, kernel, bytes) \
    ipd_lds_optin(reinterpret_cast<const void*>(kernel), (ctx)->device, (bytes))

// ---------------------------------------------------------------------------
// cross-TU device routines (all asynchronous on ctx->stream unless noted)
// ---------------------------------------------------------------------------
// ipd_sparse.hip
Csr csr_alloc(Arena& a, int nr, int nc, int nnz);
void csr_upload_from_csc(ipd_ctx* ctx, Arena& a, const ipd_csc* A, bool symmetric, Csr* out);
void csr_download_as_csc(ipd_ctx* ctx, const Csr& m, bool already_transposed, ipd_csc_out* out);
void csr_transpose(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* At);  // deterministic
void csr_spmv(ipd_ctx* ctx, const Csr& A, const double* x, double* y);
void exclusive_scan_i32(ipd_ctx* ctx, const int* in, int* out, int n,   // out has n+1 entries
                        int* total_dev = nullptr);                       // ... the total also goes there (device)
int exclusive_scan_total(ipd_ctx* ctx, const int* in, int* out, int n);
void fill_f64(ipd_ctx* ctx, double* p, double v, size_t n);
// C = X*Y with MATLAB ordering (ascending inner index, no FMA, exact zeros dropped)
// total_dev != NULL ("lazy count"): no host round trip -- C's arrays are sized by the dense bound nr*nc (the
// caller has checked SPGEMM_LAZY_MAX), C->nnz is that bound until the caller has fetched *total_dev; the nnz of
// X and Y are then only read by the kernel-choice heuristic (estimates will do: both kernels give the same bits)
constexpr size_t SPGEMM_LAZY_MAX = size_t(1) << 21;
// maxrow_dev (lazy only): the longest row of C is stored there; x_maxrow: the longest row of X if known (estimate)
void csr_spgemm(ipd_ctx* ctx, Arena& dst, const Csr& X, const Csr& Y, Csr* C, int* total_dev = nullptr,
                LazyPost* post = nullptr, int* maxrow_dev = nullptr, int x_maxrow = 0);
void csr_expand_dense(ipd_ctx* ctx, const Csr& A, double* dense, int ld);  // dense pre-zeroed
// st.out != NULL: rowcnt is zeroed<int> and the launch's tail scans the (biased) counts into st.out (nr > 0);
// an empty st leaves plain counts
void dense_rowcount(ipd_ctx* ctx, int nr, int nc, int ld, const double* dense, int* rowcnt, const ScanTail& st);
void dense_compact(ipd_ctx* ctx, int nr, int nc, int ld, const double* dense, const Csr& out);
void csr_copy(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* out);
void csr_drop_zeros(ipd_ctx* ctx, Arena& dst, const Csr& A, Csr* out);  // ipd_kkt.hip

// ipd_kkt.hip
void kkt_ax(ipd_ctx* ctx, const double* x, const double* p, const double* q, int m, int n,
            double* y);
void kkt_aty(ipd_ctx* ctx, const double* y, const double* p, const double* q, int m, int n,
             double* z);
void kkt_asat(ipd_ctx* ctx, Arena& dst, const uint8_t* s, const double* p, const double* q,
              int m, int n, Csr* H);
void kkt_inv_aat(ipd_ctx* ctx, const double* x, const double* p, const double* q, int m, int n,
                 double sg1, double sg2, double* y);
void kkt_inv_hht(ipd_ctx* ctx, const double* v, const double* p, const double* q, int m, int n,
                 double sg, const double* phi, double* y);
void kkt_inv_hht_pre(ipd_ctx* ctx, const double* v, const double* p, const double* q, int m, int n,
                     double sg, const double* l, const double* part, int npart, double* y);
int kkt_phi_consts(ipd_ctx* ctx, const double* phi, const double* p, const double* q, int m, int n,
                   double* l, double* part);
