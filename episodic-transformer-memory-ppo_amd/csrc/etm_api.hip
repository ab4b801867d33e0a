// ABI version + error strings of libetm_hip.so.
#include "etm_common.h"

extern "C" int etm_abi_version(void) { return 1; }

extern "C" const char *etm_error_string(int code) {
  switch (code) {
    case ETM_OK: return "ok";
    case ETM_EINVAL: return "etm: invalid argument (null pointer or bad dimension)";
    case ETM_EUNSUPPORTED: return "etm: shape not supported by the gfx950 kernels (need D % 32 == 0, head_dim in {32,64,96,128}, L <= 128)";
    case ETM_EWORKSPACE: return "etm: workspace too small";
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "etm: unknown error";
}
