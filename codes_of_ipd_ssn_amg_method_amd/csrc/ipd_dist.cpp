// Multi-GPU row-block sharding: the RCCL communicator (one rank per GPU over xGMI)
// and the in-place all-gather that re-assembles a vector whose row blocks were
// produced by different ranks.  The control plane (unique-id
// broadcast, barriers) is the caller's: bench.py uses torch.distributed for it.
#include <rccl/rccl.h>

#include "ipd_amg_internal.h"

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    long long allgather_calls = 0;   // grouped launches
    long long allgather_vectors = 0; // ncclAllGather calls inside them
};

#define IPD_NCCL(expr)                                                                    \
    do {                                                                                  \
        ncclResult_t r__ = (expr);                                                        \
        if (r__ != ncclSuccess)                                                           \
            throw IpdError(IPD_E_COMM, std::string(#expr) + ": " + ncclGetErrorString(r__)); \
    } while (0)

void ipd_comm_cleanup(ipd_ctx* ctx) {
    if (ctx && ctx->comm) {
        if (ctx->comm->comm) (void)ncclCommDestroy(ctx->comm->comm);
        delete ctx->comm;
        ctx->comm = nullptr;
    }
}

int comm_rank(const ipd_ctx* ctx) { return ctx->comm ? ctx->comm->rank : 0; }
int comm_size(const ipd_ctx* ctx) { return ctx->comm ? ctx->comm->nranks : 1; }

// Every rank wrote base[rank*count .. (rank+1)*count) of each vector; afterwards every
// rank holds all nranks*count entries.  The vectors of one call travel as ONE grouped
// RCCL launch (messages are a few KiB: the cost is latency, not bandwidth).
void comm_allgather_inplace(ipd_ctx* ctx, double* const* bases, int nvec, int count) {
    RcclState* c = ctx->comm;
    IPD_REQUIRE(c && c->comm, IPD_E_COMM, "communicator not initialised (ipd_comm_init)");
    if (c->nranks == 1 || count == 0 || nvec == 0) return;
    IPD_NCCL(ncclGroupStart());
    for (int v = 0; v < nvec; ++v)
        IPD_NCCL(ncclAllGather(bases[v] + (size_t)c->rank * count, bases[v], (size_t)count,
                               ncclDouble, c->comm, ctx->stream));
    IPD_NCCL(ncclGroupEnd());
    c->allgather_calls += 1;
    c->allgather_vectors += nvec;
}

extern "C" int ipd_comm_get_unique_id(uint8_t id[IPD_COMM_ID_BYTES]) {
    return ipd_guard([&] {
        IPD_REQUIRE(id, IPD_E_ARG, "id is NULL");
        static_assert(sizeof(ncclUniqueId) == IPD_COMM_ID_BYTES, "unique id size");
        ncclUniqueId u;
        IPD_NCCL(ncclGetUniqueId(&u));
        std::memcpy(id, &u, sizeof(u));
    });
}

extern "C" int ipd_comm_init(ipd_ctx* ctx, const uint8_t id[IPD_COMM_ID_BYTES], int rank,
                             int nranks) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx && id && nranks >= 1 && rank >= 0 && rank < nranks, IPD_E_ARG,
                    "bad communicator arguments");
        ctx->set_device();
        ipd_comm_cleanup(ctx);
        std::unique_ptr<RcclState> st(new RcclState());
        st->rank = rank;
        st->nranks = nranks;
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof(u));
        IPD_NCCL(ncclCommInitRank(&st->comm, nranks, u, rank));
        ctx->comm = st.release();
    });
}

// What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank) and how
// many all-gathers the library has issued on it since ipd_comm_init (reset = 1 zeroes them).
extern "C" int ipd_comm_stats(ipd_ctx* ctx, int32_t* rank, int32_t* nranks, int64_t* allgather_calls,
                              int64_t* allgather_vectors, int32_t reset) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        RcclState* c = ctx->comm;
        IPD_REQUIRE(c && c->comm, IPD_E_COMM, "communicator not initialised (ipd_comm_init)");
        int r = 0, n = 0;
        IPD_NCCL(ncclCommUserRank(c->comm, &r));
        IPD_NCCL(ncclCommCount(c->comm, &n));
        if (rank) *rank = r;
        if (nranks) *nranks = n;
        if (allgather_calls) *allgather_calls = c->allgather_calls;
        if (allgather_vectors) *allgather_vectors = c->allgather_vectors;
        if (reset) c->allgather_calls = c->allgather_vectors = 0;
    });
}

extern "C" int ipd_comm_finalize(ipd_ctx* ctx) {
    return ipd_guard([&] {
        IPD_REQUIRE(ctx, IPD_E_ARG, "ctx is NULL");
        ctx->set_device();
        ctx->sync();
        ipd_comm_cleanup(ctx);
    });
}
