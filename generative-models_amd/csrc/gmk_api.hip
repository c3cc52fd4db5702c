// gmk_api.hip -- version / error strings.
#include "gmk_common.h"

extern "C" int gmk_version(void) { return 1; }

extern "C" const char* gmk_strerror(int code) {
  switch (code) {
    case GMK_OK: return "ok";
    case GMK_E_SHAPE: return "gmk: unsupported or inconsistent shape";
    case GMK_E_ALIGN: return "gmk: pointer is not 16-byte aligned";
    case GMK_E_UNSUPPORTED: return "gmk: unsupported option";
    case GMK_E_WORKSPACE: return "gmk: workspace too small";
    case GMK_E_NULL: return "gmk: required pointer is NULL";
    case GMK_E_TOO_LARGE: return "gmk: input exceeds the stated maximum of this allocation-free entry point";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "gmk: unknown error";
}
