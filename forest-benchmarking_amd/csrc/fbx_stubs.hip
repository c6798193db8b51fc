// Temporary: entry points whose kernels are not written yet report FBX_ERR_UNSUPPORTED.
#include "fbx_common.hpp"
using namespace fbx;
#define STUB(name, ...) int name(__VA_ARGS__) { set_error(#name ": not implemented in this build"); return FBX_ERR_UNSUPPORTED; }
extern "C" {
STUB(fbx_linv_process, const fbx_design*, int64_t, const double*, double*)
}
