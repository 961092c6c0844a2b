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
    volatile unsigned* mailbox = nullptr;   // [0] ticket, [16..48) payload words
    unsigned mailbox_ticket = 0;
    int num_cu = 256;
    long long asat_nnz_hint = 0;   // entries of the last ASAt result (sizes the next one's arrays)
    // entries of P, Pt*A and Ac of the last hierarchy's level k (amg_transfer's lazy counts: the kernel-choice
    // heuristics of the next hierarchy's products run on these estimates; 0 = none yet)
    int xfer_hint[40][3] = {};
    void* asat_agg = nullptr;      // k_asat_small's chained-scan words (ipd_kkt.hip)
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
    // ticket to the kernel (which stores <= 32 words at mailbox[16..] and then the ticket at
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
    }
    ~CallScope() {
        ctx->scratch->cur = cur;
        ctx->scratch->off = off;
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
void exclusive_scan_total2(ipd_ctx* ctx, const int* in1, int* out1, const int* in2, int* out2, int n,
                           int* total1, int* total2);  // same, returns out[n]
void fill_i32(ipd_ctx* ctx, int* p, int v, size_t n);
void fill_f64(ipd_ctx* ctx, double* p, double v, size_t n);
void fill_u8(ipd_ctx* ctx, uint8_t* p, uint8_t v, size_t n);
// C = X*Y with MATLAB ordering (ascending inner index, no FMA, exact zeros dropped)
// total_dev != NULL ("lazy count"): no host round trip -- C's arrays are sized by the dense bound nr*nc (the
// caller has checked SPGEMM_LAZY_MAX), C->nnz is that bound until the caller has fetched *total_dev; the nnz of
// X and Y are then only read by the kernel-choice heuristic (estimates will do: both kernels give the same bits)
constexpr size_t SPGEMM_LAZY_MAX = size_t(1) << 21;
void csr_spgemm(ipd_ctx* ctx, Arena& dst, const Csr& X, const Csr& Y, Csr* C, int* total_dev = nullptr);
void csr_expand_dense(ipd_ctx* ctx, const Csr& A, double* dense, int ld);  // dense pre-zeroed
void dense_rowcount(ipd_ctx* ctx, int nr, int nc, int ld, const double* dense, int* rowcnt);
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
