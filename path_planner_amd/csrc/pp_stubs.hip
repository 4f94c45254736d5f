// pp_stubs.hip -- entry points not implemented yet (removed one by one as kernels land).
#include "pp_internal.h"
#define STUB { return PP_ERR_UNSUPPORTED; }
extern "C" {
int32_t pp_plan(pp_ctx*, const pp_map*, const pp_config*, const double*, const double*, pp_path*, pp_plan_stats*) STUB
int32_t pp_plan_batch(pp_ctx*, const pp_map*, const pp_config*, uint64_t, const double*, const double*, pp_path*, pp_plan_stats*, int32_t*) STUB
}
