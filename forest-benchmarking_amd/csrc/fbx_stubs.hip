// Temporary: entry points whose kernels are not written yet report FBX_ERR_UNSUPPORTED.
#include "fbx_common.hpp"
using namespace fbx;
#define STUB(name, ...) int name(__VA_ARGS__) { set_error(#name ": not implemented in this build"); return FBX_ERR_UNSUPPORTED; }
extern "C" {
STUB(fbx_linv_process, const fbx_design*, int64_t, const double*, double*)
STUB(fbx_linv_state, const fbx_design*, int64_t, const double*, double*)
STUB(fbx_mle_state, const fbx_design*, int64_t, const double*, const double*, double, double, double, double, int, double*, int32_t*, int32_t*)
STUB(fbx_r_operator, const fbx_design*, int64_t, const double*, const double*, double*)
STUB(fbx_state_log_likelihood, const fbx_design*, int64_t, const double*, const double*, const double*, double*)
STUB(fbx_convert, int, int, int, int64_t, const double*, int, double*)
STUB(fbx_kraus_sweep, int, int64_t, int, const double*, const double*, double*, double*, double*, double*)
STUB(fbx_kraus_sweep_dev, int, int64_t, int, const double*, const double*, double*, double*, double*, double*)
STUB(fbx_proj_choi, int, int, int64_t, const double*, double*, int32_t*)
STUB(fbx_proj_state_physical, int, int64_t, const double*, double*)
STUB(fbx_apply_choi, int, int64_t, const double*, const double*, double*)
STUB(fbx_process_fidelity, int, int64_t, const double*, const double*, double*, double*)
STUB(fbx_state_measures, int, int64_t, const double*, const double*, double*, double*, double*, double*)
}
