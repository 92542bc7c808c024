// Version / error-string entry points of libslu_hip.so.
#include "slu_common.h"

extern "C" int slu_abi_version(void) { return SLU_ABI_VERSION; }

extern "C" const char* slu_strerror(int code) {
  switch (code) {
    case SLU_OK: return "ok";
    case SLU_EINVAL: return "invalid argument (null pointer, non-positive size or inconsistent descriptor)";
    case SLU_EUNSUPPORTED: return "unsupported shape or kernel family";
    case SLU_ELAUNCH: return "HIP kernel launch failed";
    default: return "unknown slu error code";
  }
}
