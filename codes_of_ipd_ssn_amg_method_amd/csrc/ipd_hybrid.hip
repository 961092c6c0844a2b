// Problem-level solvers (placeholder until Hybrid_AMG lands).
#include "ipd_amg_internal.h"
static int unsupported() { ipd_set_error("Hybrid_AMG path not built yet"); return IPD_E_UNSUPPORTED; }
extern "C" int ipd_components(ipd_ctx*, const ipd_csc*, int64_t*, int64_t*, int64_t*, int64_t*, int64_t*) { return unsupported(); }
extern "C" int ipd_hybrid_amg(ipd_ctx*, const ipd_prob*, const ipd_amg_opts*, ipd_rng*, double*, int32_t*, double*, int64_t*) { return unsupported(); }
extern "C" int ipd_amg4pot(ipd_ctx*, const ipd_prob*, const ipd_amg_opts*, ipd_rng*, double*, int32_t*, double*, int64_t*) { return unsupported(); }
extern "C" int ipd_hybrid_amg_dev(ipd_ctx*, const ipd_dmat*, const double*, const double*, const double*, int64_t, int64_t, double, double, const double*, const ipd_amg_opts*, ipd_rng*, double*, int32_t*, double*, int64_t*) { return unsupported(); }
