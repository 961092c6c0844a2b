// Context, memory pools, MATLAB-compatible rand stream and the small C-ABI
// entry points that do not launch kernels.
#include "ipd_internal.h"

#include <mutex>
#include <set>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <cstdlib>

static thread_local std::string g_last_error;

void ipd_set_error(const std::string& msg) { g_last_error = msg; }

// ---------------------------------------------------------------------------
// ChunkPool / Arena
// ---------------------------------------------------------------------------
static constexpr size_t kMinChunk = size_t(64) << 20;  // 64 MiB
static constexpr size_t kAlign = 256;

ChunkPool::Chunk ChunkPool::get(size_t min_bytes) {
    // best fit among free chunks
    int best = -1;
    for (int i = 0; i < (int)free_chunks.size(); ++i) {
        if (free_chunks[i].bytes >= min_bytes &&
            (best < 0 || free_chunks[i].bytes < free_chunks[best].bytes))
            best = i;
    }
    if (best >= 0) {
        Chunk c = free_chunks[best];
        free_chunks.erase(free_chunks.begin() + best);
        return c;
    }
    Chunk c;
    c.bytes = std::max(kMinChunk, (min_bytes + kAlign - 1) / kAlign * kAlign);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, c.bytes);
    if (e != hipSuccess)
        throw IpdError(IPD_E_NOMEM, std::string("hipMalloc(") + std::to_string(c.bytes) +
                                        "): " + hipGetErrorString(e));
    c.base = static_cast<char*>(p);
    total_bytes += c.bytes;
    return c;
}

void ChunkPool::release_all() {
    for (auto& c : free_chunks) (void)hipFree(c.base);
    free_chunks.clear();
}

void* Arena::alloc_bytes(size_t bytes) {
    bytes = (bytes + kAlign - 1) / kAlign * kAlign;
    while (true) {
        if (cur < chunks.size()) {
            if (off + bytes <= chunks[cur].bytes) {
                void* p = chunks[cur].base + off;
                off += bytes;
                return p;
            }
            ++cur;
            off = 0;
            continue;
        }
        chunks.push_back(pool->get(bytes));
        cur = chunks.size() - 1;
        off = 0;
    }
}

void Arena::release() {
    if (pool)
        for (auto& c : chunks) pool->put(c);
    chunks.clear();
    cur = off = 0;
}

// ---------------------------------------------------------------------------
// ipd_ctx
// ---------------------------------------------------------------------------
// Scalar readbacks (counts, norms, convergence flags) happen ~60 times per hierarchy and once
// per cycle.  Measured on MI355X (tools/ubench_fetch.hip): kernel + hipMemcpyAsync to pinned
// memory + hipStreamSynchronize 14.3 us per round trip; a one-wave kernel that stores the words
// and then a ticket into host-coherent memory while the host spins on the ticket 5.8 us.
constexpr int MAILBOX_WORDS = 224;   // payload words (a resident solve's result block is 136)

__global__ void k_mailbox(const unsigned* __restrict__ src, int nwords, volatile unsigned* box,
                          unsigned ticket) {
    const int i = threadIdx.x;
    for (int w = i; w < nwords; w += 64) box[16 + w] = src[w];
    __threadfence_system();   // the payload is visible to the host before the ticket is
    __syncthreads();
    if (i == 0) box[0] = ticket;
}

// The attribute is set while the lock is held and the pair is recorded only once it is set: a second
// host thread that reaches the same kernel at the same moment (AMG4POT's two solve phases) waits here
// instead of launching with > 64 KiB of dynamic LDS before the first thread's call has taken effect.
void ipd_lds_optin(const void* kernel, int device, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, device})) return;
    IPD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, device});
}

static bool mailbox_enabled() {
    return true;
}

bool ipd_ctx::mailbox_begin(unsigned* ticket) {
    if (!mailbox || !mailbox_enabled()) return false;
    *ticket = ++mailbox_ticket;
    return true;
}

void ipd_ctx::mailbox_wait(unsigned ticket, void* hdst, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; mailbox[0] != ticket; ++spins) {
        if ((spins & 0xffff) == 0 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            // the stream is stuck or faulted: let the runtime report it
            IPD_HIP(hipStreamSynchronize(stream));
            if (mailbox[0] != ticket)
                throw IpdError(IPD_E_HIP, "readback mailbox: the ticket never arrived");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    unsigned words[MAILBOX_WORDS];
    for (size_t w = 0; w < bytes / 4; ++w) words[w] = mailbox[16 + w];
    std::memcpy(hdst, words, bytes);
}

void ipd_ctx::fetch_bytes(const void* dsrc, void* hdst, size_t bytes) {
    if (bytes == 0) return;
    unsigned ticket = 0;
    if (bytes <= MAILBOX_WORDS * 4 && bytes % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(dsrc) & 3) == 0 && mailbox_begin(&ticket)) {
        hipLaunchKernelGGL(k_mailbox, dim3(1), dim3(64), 0, stream,
                           static_cast<const unsigned*>(dsrc), (int)(bytes / 4), mailbox, ticket);
        IPD_HIP(hipGetLastError());
        mailbox_wait(ticket, hdst, bytes);
        return;
    }
    if (bytes <= pinned_bytes) {
        IPD_HIP(hipMemcpyAsync(pinned, dsrc, bytes, hipMemcpyDeviceToHost, stream));
        IPD_HIP(hipStreamSynchronize(stream));
        std::memcpy(hdst, pinned, bytes);
    } else {
        IPD_HIP(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, stream));
        IPD_HIP(hipStreamSynchronize(stream));
    }
}

void ipd_ctx::upload_bytes(void* ddst, const void* hsrc, size_t bytes) {
    if (bytes == 0) return;
    // The source is usually a transient pageable host buffer, which the caller may free or reuse on
    // return.  Small uploads (descriptor tables, random numbers: ~20 per hierarchy) therefore go
    // through a ring of pinned staging memory: the bytes are copied into the ring on the host and
    // the copy to the device is queued WITHOUT waiting for it (a copy from pageable memory followed by
    // a stream synchronisation cost ~20 us each, 0.3-0.4 ms per hierarchy).  A slot is reused only
    // after the ring has wrapped, and wrapping waits for the stream once.
    if (up_ring && bytes <= up_ring_bytes / 8) {
        const size_t need = (bytes + 255) & ~size_t(255);
        if (up_ring_off + need > up_ring_bytes) {
            IPD_HIP(hipStreamSynchronize(stream));
            up_ring_off = 0;
        }
        char* slot = up_ring + up_ring_off;
        up_ring_off += need;
        std::memcpy(slot, hsrc, bytes);
        IPD_HIP(hipMemcpyAsync(ddst, slot, bytes, hipMemcpyHostToDevice, stream));
        return;
    }
    IPD_HIP(hipMemcpyAsync(ddst, hsrc, bytes, hipMemcpyHostToDevice, stream));
    IPD_HIP(hipStreamSynchronize(stream));
}

extern "C" int ipd_version(void) { return IPD_VERSION; }

// ---------------------------------------------------------------------------
// profiling scopes
// ---------------------------------------------------------------------------
#include <chrono>
static double g_prof_t[PROF_SLOTS];
static long long g_prof_n[PROF_SLOTS];
bool ipd_prof_enabled() {
    static const bool on = [] {
        const char* e = getenv("IPD_PROFILE");
        return e && *e && *e != '0';
    }();
    return on;
}
void ipd_prof_add(int slot, double seconds) {
    g_prof_t[slot] += seconds;
    ++g_prof_n[slot];
}
static double prof_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}
ProfScope::ProfScope(ipd_ctx* c, int s) : ctx(c), slot(s) {
    if (!ipd_prof_enabled()) return;
    (void)hipStreamSynchronize(ctx->stream);
    t0 = prof_now();
}
ProfScope::~ProfScope() {
    if (!ipd_prof_enabled()) return;
    (void)hipStreamSynchronize(ctx->stream);
    ipd_prof_add(slot, prof_now() - t0);
}
extern "C" int ipd_prof_read(double* seconds, int64_t* calls, int32_t reset) {
    for (int i = 0; i < PROF_SLOTS; ++i) {
        if (seconds) seconds[i] = g_prof_t[i];
        if (calls) calls[i] = g_prof_n[i];
        if (reset) {
            g_prof_t[i] = 0.0;
            g_prof_n[i] = 0;
        }
    }
    return PROF_SLOTS;
}

extern "C" const char* ipd_last_error(void) { return g_last_error.c_str(); }

extern "C" int ipd_device_count(int32_t* count) {
    return ipd_guard([&] {
        IPD_REQUIRE(count != nullptr, IPD_E_ARG, "ipd_device_count: count is NULL");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
        *count = ndev;
    });
}

extern "C" int ipd_ctx_create(int device, ipd_ctx** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(out != nullptr, IPD_E_ARG, "ipd_ctx_create: out is NULL");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            throw IpdError(IPD_E_HIP,
                           "no HIP device available: libipdamg has no CPU fallback "
                           "(hipGetDeviceCount: " +
                               std::string(hipGetErrorString(e)) + ")");
        IPD_REQUIRE(device >= 0 && device < ndev, IPD_E_ARG, "ipd_ctx_create: bad device index");
        IPD_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        IPD_HIP(hipGetDeviceProperties(&prop, device));
        std::unique_ptr<ipd_ctx> c(new ipd_ctx());
        c->device = device;
        c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        IPD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->scratch.reset(new Arena(&c->pool));
        c->pinned_bytes = size_t(1) << 20;
        IPD_HIP(hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault));
        {
            void* ring = nullptr;
            c->up_ring_bytes = size_t(4) << 20;
            IPD_HIP(hipHostMalloc(&ring, c->up_ring_bytes, hipHostMallocDefault));
            c->up_ring = static_cast<char*>(ring);
        }
        void* box = nullptr;
        IPD_HIP(hipHostMalloc(&box, 256 * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(box, 0, 256 * sizeof(unsigned));
        c->mailbox = static_cast<volatile unsigned*>(box);
        *out = c.release();
    });
}

// zero pool (ipd_internal.h): created on first use, 64 MiB
void* ipd_ctx::zalloc(size_t bytes) {
    constexpr size_t POOL = size_t(64) << 20;
    if (bytes > POOL / 2) return nullptr;
    if (!zpool) {
        void* p = nullptr;
        if (hipMalloc(&p, POOL) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        IPD_HIP(hipMemsetAsync(p, 0, POOL, stream));
        zpool = static_cast<char*>(p);
        zpool_bytes = POOL;
        zpool_used = 0;
    }
    const size_t need = (bytes + 255) & ~size_t(255);
    if (zpool_used + need > zpool_bytes) return nullptr;
    void* r = zpool + zpool_used;
    zpool_used += need;
    return r;
}

void ipd_ctx::zreset() {
    if (zpool && zpool_used) IPD_HIP(hipMemsetAsync(zpool, 0, zpool_used, stream));
    zpool_used = 0;
}

// lazily created companion context (same device, own stream/arenas/pinned buffer)
ipd_ctx* ipd_ctx_aux(ipd_ctx* ctx) {
    if (!ctx->aux) {
        ipd_ctx* a = nullptr;
        const int rc = ipd_ctx_create(ctx->device, &a);
        if (rc != IPD_OK) throw IpdError(rc, "cannot create the auxiliary context");
        ctx->aux = a;
    }
    return ctx->aux;
}

extern "C" void ipd_ctx_destroy(ipd_ctx* ctx) {
    if (!ctx) return;
    if (ctx->aux) {
        ipd_ctx_destroy(ctx->aux);
        ctx->aux = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ipd_comm_cleanup(ctx);
    ctx->scratch.reset();
    ctx->pool.release_all();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->up_ring) (void)hipHostFree(ctx->up_ring);
    if (ctx->asat_agg) (void)hipFree(ctx->asat_agg);
    if (ctx->zpool) (void)hipFree(ctx->zpool);
    if (ctx->mailbox) (void)hipHostFree(const_cast<unsigned*>(ctx->mailbox));
    for (hipEvent_t& ev : ctx->tev)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int ipd_ctx_sync(ipd_ctx* ctx) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        ctx->set_device();
        ctx->sync();
    });
}

extern "C" void ipd_csc_free(ipd_csc_out* m) {
    if (!m) return;
    std::free(m->jc);
    std::free(m->ir);
    std::free(m->pr);
    m->jc = m->ir = nullptr;
    m->pr = nullptr;
    m->nnz = m->nrows = m->ncols = 0;
}

extern "C" void ipd_amg_opts_init(ipd_amg_opts* o) {
    if (!o) return;
    o->retol = -1;
    o->bigph = -1;
    o->maxit = -1;
    o->theta = -1;
    o->smoth = -1;
    o->cycle = -1;
    o->isnsp = -1;
    o->inter = -1;
    o->fnode = -1;
}

extern "C" void ipd_pcg_opts_init(ipd_pcg_opts* o) {
    if (!o) return;
    o->retol = -1;
    o->maxit = -1;
    o->precd = -1;
    o->nf = 0;
}

// ---------------------------------------------------------------------------
// raw device memory
// ---------------------------------------------------------------------------
extern "C" int ipd_dmalloc(ipd_ctx* ctx, size_t bytes, void** dptr) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && dptr, IPD_E_ARG, "ipd_dmalloc: NULL argument");
        ctx->set_device();
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) throw IpdError(IPD_E_NOMEM, hipGetErrorString(e));
        *dptr = p;
    });
}

extern "C" int ipd_dfree(ipd_ctx* ctx, void* dptr) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        ctx->set_device();
        ctx->sync();
        IPD_HIP(hipFree(dptr));
    });
}

extern "C" int ipd_h2d(ipd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        ctx->set_device();
        IPD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        ctx->sync();
    });
}

extern "C" int ipd_d2h(ipd_ctx* ctx, void* dst, const void* src, size_t bytes) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        ctx->set_device();
        IPD_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
    });
}

// ---------------------------------------------------------------------------
// mt19937ar, 53-bit doubles: MATLAB's default `rand`
// ---------------------------------------------------------------------------
void ipd_rng::seed(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    mti = 624;
}

uint32_t ipd_rng::next_u32() {
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    if (mti >= 624) {
        int kk;
        for (kk = 0; kk < 624 - 397; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < 623; ++kk) {
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        mti = 0;
    }
    uint32_t y = mt[mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double ipd_rng::next_double() {
    uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

void ipd_rng::fill(double* out, int64_t n) {
    if (replay) {
        if (consumed + n > (int64_t)values.size())
            throw IpdError(IPD_E_ARG, "replay rand stream exhausted");
        std::copy(values.begin() + consumed, values.begin() + consumed + n, out);
    } else {
        for (int64_t i = 0; i < n; ++i) out[i] = next_double();
    }
    consumed += n;
}

extern "C" int ipd_rng_create(uint32_t seed, ipd_rng** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(out, IPD_E_ARG, "out is NULL");
        std::unique_ptr<ipd_rng> r(new ipd_rng());
        r->seed(seed);
        *out = r.release();
    });
}

extern "C" int ipd_rng_create_replay(const double* values, int64_t count, ipd_rng** out) {
    return ipd_guard([&] {
        IPD_REQUIRE(out && (values || count == 0) && count >= 0, IPD_E_ARG, "bad argument");
        std::unique_ptr<ipd_rng> r(new ipd_rng());
        r->replay = true;
        r->values.assign(values, values + count);
        *out = r.release();
    });
}

extern "C" void ipd_rng_destroy(ipd_rng* rng) { delete rng; }

extern "C" int ipd_rng_rand(ipd_rng* rng, int64_t count, double* out) {
    return ipd_guard([&] {
        IPD_REQUIRE(rng && out && count >= 0, IPD_E_ARG, "bad argument");
        rng->fill(out, count);
    });
}

extern "C" int64_t ipd_rng_consumed(const ipd_rng* rng) { return rng ? rng->consumed : -1; }
