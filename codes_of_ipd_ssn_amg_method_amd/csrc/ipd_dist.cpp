// Multi-GPU row-block sharding over RCCL (placeholder until the sharded cycle lands).
#include "ipd_amg_internal.h"

void ipd_comm_cleanup(ipd_ctx*) {}

extern "C" int ipd_comm_get_unique_id(uint8_t*) {
    ipd_set_error("RCCL sharding not built yet");
    return IPD_E_UNSUPPORTED;
}
extern "C" int ipd_comm_init(ipd_ctx*, const uint8_t*, int, int) {
    ipd_set_error("RCCL sharding not built yet");
    return IPD_E_UNSUPPORTED;
}
extern "C" int ipd_comm_finalize(ipd_ctx*) { return IPD_OK; }
extern "C" int ipd_amg_bench_cycles_sharded(ipd_amg*, const double*, double*, int, double*, double*) {
    ipd_set_error("RCCL sharding not built yet");
    return IPD_E_UNSUPPORTED;
}
