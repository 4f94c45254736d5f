// pp_stubs.hip -- entry points not implemented yet (removed one by one as kernels land).
#include "pp_internal.h"
#define STUB { return PP_ERR_UNSUPPORTED; }
extern "C" {
int32_t pp_rs_batch(pp_ctx*, uint64_t, const double*, const double*, const double*, const double*, const double*, const double*, double, int32_t*, int32_t*, pp_rs_word*, uint8_t*) STUB
int32_t pp_rs_batch_dev(pp_ctx*, uint64_t, const double*, const double*, const double*, const double*, const double*, const double*, double, int32_t*, int32_t*, pp_rs_word*, uint8_t*) STUB
int32_t pp_rs_all_paths(pp_ctx*, double, double, double, double, double, double, double, double, int32_t*, pp_rs_word*, uint64_t*, double*, double*, double*, int64_t*, uint64_t) STUB
int32_t pp_rs_optimal_path(pp_ctx*, double, double, double, double, double, double, double, double, pp_rs_word*, double*, double*, double*, int64_t*, uint64_t, uint64_t*) STUB
double pp_rs_path_cost(const pp_rs_word*, const pp_config*) { return 0.0; }
int32_t pp_analytic_batch(pp_ctx*, const pp_map*, const pp_config*, uint64_t, const double*, const double*, const double*, const double*, const double*, const double*, uint8_t*, int32_t*, int32_t*, pp_rs_word*, double*) STUB
int32_t pp_plan(pp_ctx*, const pp_map*, const pp_config*, const double*, const double*, pp_path*, pp_plan_stats*) STUB
int32_t pp_plan_batch(pp_ctx*, const pp_map*, const pp_config*, uint64_t, const double*, const double*, pp_path*, pp_plan_stats*, int32_t*) STUB
}
