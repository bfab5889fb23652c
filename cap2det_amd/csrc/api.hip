// Version / error strings of the C-ABI (include/cap2det_hip.h).
#include "c2d_common.h"

extern "C" int c2d_version(void) { return C2D_ABI_VERSION; }

extern "C" const char* c2d_error_string(int code) {
  switch (code) {
    case C2D_OK: return "ok";
    case C2D_ERR_INVALID_ARG: return "invalid argument";
    case C2D_ERR_UNSUPPORTED: return "unsupported configuration";
    case C2D_ERR_LAUNCH: return "kernel launch failed";
    case C2D_ERR_WORKSPACE: return "workspace too small";
    case C2D_ERR_DATA: return "malformed input data";
    default: return "unknown error";
  }
}
